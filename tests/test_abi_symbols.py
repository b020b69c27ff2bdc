"""The C-ABI boundary, checked without a GPU: include/ggs_hip.h, libggs_hip.so and the ctypes
signature table must agree on the set of entry points (no compute call is made here)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ggs_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ggs_[a-z0-9_]+)\s*\(", text)))


def test_header_is_plain_c():
    """The header must compile as C (no C++/torch types in the signatures)."""
    src = '#include "ggs_hip.h"\nint main(void) { ggs_config c; (void)c; return sizeof(ggs_timings) > 0 ? 0 : 1; }\n'
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-x", "c", "-", "-fsyntax-only"],
                       input=src.encode(), capture_output=True)
    assert r.returncode == 0, r.stderr.decode()


def test_library_exports_every_declared_symbol():
    from ldagroupedgibbssampler_amd import _lib
    names = declared_functions()
    assert len(names) >= 30
    lib = ctypes.CDLL(_lib.LIB_PATH) if os.path.exists(_lib.LIB_PATH) else None
    if lib is None:
        pytest.fail("libggs_hip.so is not built: run __graft_entry__.build()")
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, "declared in ggs_hip.h but not exported: %s" % missing


def test_ctypes_table_matches_header():
    from ldagroupedgibbssampler_amd import _lib
    names = set(declared_functions())
    table = set(_lib.SIGNATURES)
    assert names == table, "header-only: %s; table-only: %s" % (sorted(names - table), sorted(table - names))
    L = _lib.load()                      # dlopen + type every symbol; needs no GPU
    assert L.ggs_abi_version() == _lib.ABI_VERSION


def test_struct_layouts_match_header():
    """ggs_config / ggs_timings / ggs_exchange_ops as ctypes sees them == as a C compiler lays them out."""
    from ldagroupedgibbssampler_amd import _lib
    prog = r'''
#include <stdio.h>
#include <stddef.h>
#include "ggs_hip.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(ggs_config), offsetof(ggs_config, alpha), offsetof(ggs_config, beta),
         offsetof(ggs_config, seed), offsetof(ggs_config, flags), sizeof(ggs_timings), offsetof(ggs_timings, sweeps),
         offsetof(ggs_timings, exchange_ms), offsetof(ggs_timings, exchange_ag_ms), sizeof(ggs_exchange_ops), offsetof(ggs_exchange_ops, ctx), offsetof(ggs_exchange_ops, all_gather_i32));
  return 0;
}'''
    exe = os.path.join(ROOT, "tests", ".abi_layout_probe")
    try:
        subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), "-x", "c", "-", "-o", exe], input=prog.encode(), check=True)
        got = [int(x) for x in subprocess.check_output([exe]).split()]
    finally:
        if os.path.exists(exe):
            os.remove(exe)
    cfg, tm, ops = _lib.GGSConfig, _lib.GGSTimings, _lib.GGSExchangeOps
    assert got == [ctypes.sizeof(cfg), cfg.alpha.offset, cfg.beta.offset, cfg.seed.offset, cfg.flags.offset,
                   ctypes.sizeof(tm), tm.sweeps.offset, tm.exchange_ms.offset, tm.exchange_ag_ms.offset, ctypes.sizeof(ops), ops.ctx.offset, ops.all_gather_i32.offset]


def test_no_gpu_means_loud_failure_not_fallback():
    """Without a device ggs_create must fail (GGS_ERR_HIP); nothing silently runs on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ldagroupedgibbssampler_amd import native
    with pytest.raises(native.GGSError) as e:
        native.GGSHandle(4, 10, 0.1, 0.1, 1)
    assert e.value.code == native.ERR_HIP


def test_host_side_java_lcg_utility(oracle):
    """ggs_java_lcg_next_ints is host code (no GPU): must equal the oracle's java.util.Random."""
    import numpy as np
    from ldagroupedgibbssampler_amd.sharded import java_lcg_initial_z
    for K in (2, 3, 16, 100, 1000):
        assert np.array_equal(java_lcg_initial_z(50000, K, 2019), oracle.jrandom_ints(2019, K, 50000))
    assert np.array_equal(java_lcg_initial_z(1000, 7, -5), oracle.jrandom_ints(-5, 7, 1000))


def test_table_kernels_never_spill():
    """z_hot_kernel and z_warm_kernel keep the operands of the chunk after next in registers that inline-assembly loads fill
    behind the compiler's back (ggs_z_sliced.hpp, table_chunks): a register of a load in flight must never be copied, so no
    instance of these kernels may use scratch, and each must fit the 128 registers a guest beside the cold kernel gets.
    Checked on the summary the build writes (csrc/ggs_resource_summary.txt: always the figures of the library that is loaded)."""
    from ldagroupedgibbssampler_amd import _lib
    _lib.build()
    rows = [l.split() for l in open(os.path.join(_lib.CSRC, "ggs_resource_summary.txt")) if l.startswith("ggs::z_hot_kernel") or l.startswith("ggs::z_warm_kernel")]
    assert len(rows) == 48, "one instance per KMAX = 8 ... 192 of each kernel"
    for name, vgprs, agprs, sgprs, scratch, occ, lds in rows:
        assert int(scratch) == 0, "%s spills %s bytes per lane" % (name, scratch)
        assert int(vgprs) + int(agprs) <= 128 and int(occ) >= 4, "%s: %s + %s registers" % (name, vgprs, agprs)


def test_table_kernels_hand_counted_loads_are_left_alone():
    """scripts/check_table_kernels_asm.py: compiles every instance of z_hot_kernel / z_warm_kernel to assembly and checks that
    between an inline-assembly load and its wait nothing the compiler generated reads or writes the load's registers (the
    first build that took the list entry's words outside the wait statement copied them in front of it and faulted)."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_table_kernels_asm.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "48 kernels checked, 0 complaints" in r.stdout
