"""Reads the gfx950 assembly of every instance of z_hot_kernel / z_warm_kernel and checks what their hand-counted loads rest on
(ggs_z_sliced.hpp, table_chunks): the registers an inline-assembly load writes are touched by NOTHING the compiler generated
between that load and the wait -- in the text of the whole kernel: outside the asm statements, the only instructions that may
read such a register are the selects that mask a theta piece to 0.0 beyond K while it is staged (they sit behind the wait's asm
statement and the empty asm statements that hand the registers over), and nothing may write one (check_kernel says how "between" is
read off the text).  Also: no scratch.

usage: check_table_kernels_asm.py            compiles a translation unit that instantiates all 48 kernels (hipcc -S, under a minute)
       check_table_kernels_asm.py file.s     checks an existing assembly file
Exit code 0 and a one-line summary, or 1 and the offending lines."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ldagroupedgibbssampler_amd", "csrc")
KMAX = [8 * i for i in range(1, 25)]


def generate(path_s):
    src = "#include <hip/hip_runtime.h>\n#include \"ggs_z_sliced.hpp\"\n" + "".join(
        "template __global__ void ggs::z_hot_kernel<%d>(ggs::ZParams);\ntemplate __global__ void ggs::z_warm_kernel<%d>(ggs::ZParams);\n" % (k, k) for k in KMAX)
    with tempfile.NamedTemporaryFile("w", suffix=".hip", delete=False) as f:
        f.write(src)
    try:
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-I", CSRC,
                        "-I", os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", path_s, f.name], check=True, stderr=subprocess.DEVNULL)
    finally:
        os.remove(f.name)


def regs_of(tok):
    """v5 -> {5}; v[4:7] -> {4,5,6,7}; anything else -> {}"""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def check_kernel(name, lines):
    """lines: the kernel's text, walked in layout order (the chunk loop is laid out in execution order: arrive A, stage A,
    issue A', arithmetic, stores, arrive B, stage B, issue B', arithmetic, stores).  A register an asm load writes is IN FLIGHT
    from that load until an asm wait statement reads it (the list entry's words) or, behind an asm wait, a masking select
    does (a theta piece); until then nothing outside an asm statement may read or write it.  Barriers end every flight (a
    tier's chunk loop has consumed all it requested).  Returns a list of complaints."""
    bad = []
    in_asm = False
    flying = {}                         # register -> an asm wait has been passed since its load
    n_loads = n_sel = n_take = 0
    for ln in lines:
        t = ln.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        ops = [o.strip() for o in re.split(r"[,\s]+", t.split(";")[0].strip()) if o.strip()]
        if "scratch_" in ops[0]:
            bad.append("%s: scratch access: %s" % (name, t))
            continue
        if ops[0] == "s_barrier":
            flying.clear()
            continue
        if in_asm:
            if ops[0].startswith("global_load"):
                n_loads += 1
                for r in regs_of(ops[1]):
                    flying[r] = False
            elif ops[0] == "s_waitcnt":
                for r in flying:
                    flying[r] = True
            elif ops[0] == "v_mov_b32":                       # the wait statement's own copies of the list entry
                for r in regs_of(ops[2]):
                    if r in flying:
                        n_take += 1
                        if not flying[r]:
                            bad.append("%s: asm copy of an in-flight register with no wait in front: %s" % (name, t))
                        del flying[r]
            continue
        dst = regs_of(ops[1]) if len(ops) > 1 else set()
        srcs = set()
        for o in ops[2:]:
            srcs |= regs_of(o)
        if ops[0].startswith(("global_store", "ds_write", "ds_store", "global_atomic", "v_cmp", "v_cmpx")):   # no vector destination: every operand is read
            srcs |= dst
            dst = set()
        for r in sorted(dst & set(flying)):
            bad.append("%s: writes v%d while an asm load into it is in flight: %s" % (name, r, t))
        for r in sorted(srcs & set(flying)):
            if ops[0].startswith("v_cndmask_b32") and flying[r]:
                n_sel += 1
                del flying[r]
            else:
                bad.append("%s: reads v%d while an asm load into it is in flight: %s" % (name, r, t))
    if n_loads == 0 or n_sel == 0 or n_take == 0:
        bad.append("%s: asm loads %d, staging selects %d, list words taken %d -- the check no longer matches the code" % (name, n_loads, n_sel, n_take))
    return bad


def main(argv):
    if len(argv) > 1:
        path, made = argv[1], False
    else:
        path, made = os.path.join(tempfile.gettempdir(), "ggs_table_kernels_%d.s" % os.getpid()), True
        generate(path)
    txt = open(path).read().splitlines()
    if made:
        os.remove(path)
    kernels = {}
    cur = None
    for ln in txt:
        m = re.match(r"^(_ZN3ggs1[23]z_(?:hot|warm)_kernelILi\d+EEEvNS_7ZParamsE):", ln)
        if m:
            cur = m.group(1)
            kernels[cur] = []
            continue
        if cur is not None:
            kernels[cur].append(ln)
            if ln.strip().startswith("s_endpgm"):
                cur = None
    bad = []
    for name, lines in sorted(kernels.items()):
        bad += check_kernel(name, lines)
    if len(kernels) != 2 * len(KMAX) and len(argv) <= 1:
        bad.append("expected %d kernels, found %d" % (2 * len(KMAX), len(kernels)))
    for b in bad[:40]:
        print(b)
    print("%d kernels checked, %d complaints" % (len(kernels), len(bad)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
