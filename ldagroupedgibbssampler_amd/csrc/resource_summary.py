#!/usr/bin/env python3
"""One line per kernel from hipcc's -Rpass-analysis=kernel-resource-usage remarks: registers, scratch, occupancy.
usage: resource_summary.py ggs_resource_usage.txt   (written by the Makefile at every build of libggs_hip.so)"""
import re
import subprocess
import sys


def demangle(names):
    try:
        out = subprocess.run(["c++filt"] + names, capture_output=True, text=True, check=True).stdout.splitlines()
        return out if len(out) == len(names) else names
    except (OSError, subprocess.CalledProcessError):
        return names


def main(path):
    txt = open(path).read()
    rows = []
    for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
        name = b.split()[0]

        def g(key):
            m = re.search(key + r": (\d+)", b)
            return int(m.group(1)) if m else -1
        rows.append((name, g("VGPRs"), g("AGPRs"), g("SGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"),
                     g(r"LDS Size \[bytes/block\]")))
    names = demangle([r[0] for r in rows])
    print("# kernel | VGPRs | AGPRs | SGPRs | scratch bytes/lane | occupancy waves/SIMD (by registers) | static LDS bytes/block")
    for n, r in sorted(zip(names, rows)):
        n = re.sub(r"\(.*$", "", n).replace("void ", "")
        print("%-60s %4d %4d %4d %5d %3d %6d" % (n, r[1], r[2], r[3], r[4], r[5], r[6]))


if __name__ == "__main__":
    main(sys.argv[1])
