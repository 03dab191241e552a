"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/ptm_engine.h
declares, and refuses to compute without a GPU (no silent CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from ptmcmc_amd import engine as E

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    txt = open(os.path.join(ROOT, "include", "ptm_engine.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ptm_[a-z0-9_]+)\s*\(", txt)) - {"ptm_loglike_batch_fn"})


def test_header_and_binding_agree():
    assert declared_functions() == sorted(E.EXPORTS)


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(E.LIB_PATH)
    for name in declared_functions():
        assert hasattr(lib, name), name
    assert lib.ptm_abi_version() == 3


def test_header_is_plain_c():
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.c")
        open(src, "w").write('#include "ptm_engine.h"\nint main(void){ptm_config c; c.struct_size=sizeof c; return ptm_abi_version()==PTM_ABI_VERSION?0:1;}\n')
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", src, "-o", os.path.join(d, "t.o")])


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="GPU present")
def test_no_cpu_fallback_without_gpu():
    assert E.device_count() == 0
    with pytest.raises(E.PtmError) as ei:
        E.Engine(2, 8, 1)
    assert "no gfx950" in str(ei.value)
    import numpy as np
    with pytest.raises(E.PtmError):
        E.debug_eval(E.FN_LOG, np.ones(4))


def test_config_validation_happens_before_device_use():
    cfg = E.PtmConfig()
    cfg.struct_size = 4
    h = C.c_void_p()
    assert E.load().ptm_engine_create(C.byref(cfg), C.byref(h)) == -1
    assert b"size mismatch" in E.load().ptm_last_error()


def test_bench_record_helpers_accept_every_committed_profile():
    """bench.py reads the HBM traffic of its kernel from the committed counter summaries (profiles/*_pmc_summary.json): any other JSON
    that lands under profiles/ must not break the record (a summary of a different counter set once did), and the roofline arithmetic
    of the record must hold together."""
    import glob
    import json
    import sys
    root = ROOT
    sys.path.insert(0, root)
    import bench
    for f in glob.glob(os.path.join(root, "profiles", "*.json")):
        json.load(open(f))                                   # every committed JSON parses
    t = bench.measured_traffic("sweep_mfma32_kernel<2, false, 0, false, true>")
    assert t is None or (t["bytes"] > 0 and t["read"] > 0 and t["write"] > 0)
    assert bench.measured_traffic("no_such_kernel") is None
    r = bench.roofline_record("k", 1.5, 20, 13_800_000, 16_777_216, 9.0e9, 1, t)
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.5 < r["frac"] < 0.8
    assert abs(r["step_frac"] - 9.0e9 * r["bytes_per_mh_step"] / (r["peak"] * 1e9)) < 1e-12
