"""CPU-only checks of the drop-in boundary: the C-ABI library loads (no GPU needed to dlopen it) and
exports every symbol include/ammsb.h declares; host-only entry points behave; no compute is called."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def capi():
    import __graft_entry__ as ge
    ge.build()
    import ammsb_pkg
    ammsb_pkg.load()
    from mcmc_ammsb_gpu_amd import _capi
    return _capi


def test_every_declared_symbol_is_exported(capi):
    hdr = open(os.path.join(ROOT, "include", "ammsb.h")).read()
    declared = set(re.findall(r"\b(ammsb_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "header parse failed"
    lib = C.CDLL(capi.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), "libammsb_hip.so does not export %s" % name
    assert declared == set(capi.SIGNATURES), "ctypes binding and header disagree: %s" % (
        declared ^ set(capi.SIGNATURES))


def test_host_only_entry_points(capi, orc):
    lib = capi.load()
    assert lib.ammsb_version() == 200
    assert lib.ammsb_strerror(0) == b"ok" and lib.ammsb_strerror(-1) == b"invalid argument"
    p = capi.Params(1000, 48, 0, 32, 1.0 / 48, 0.0315, 1024.0, 0.5, 1e-7, 1.0, 1.0)
    assert lib.ammsb_params_quantize(C.byref(p)) == 0
    q = orc.make_params(1000, 48, 32, alpha=np.float32(1.0 / 48))
    for f in ("alpha", "a", "b", "c", "epsilon", "eta0", "eta1"):
        assert getattr(p, f) == getattr(q, f)
    for t in (1, 7, 1000, 123456):
        assert lib.ammsb_eps_t(C.byref(p), t) == orc.lib().orc_eps_t(q, t)
    assert lib.ammsb_params_quantize(None) == -1


def test_product_fails_loudly_without_gpu(capi):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from mcmc_ammsb_gpu_amd import ops
    with pytest.raises(capi.AmmsbError):
        ops.Context(ops.make_params(100, 32))
    ctx = C.c_void_p()
    p = capi.Params(100, 32, 0, 8, 0.03, 0.0315, 1024.0, 0.5, 1e-7, 1.0, 1.0)
    assert capi.load().ammsb_ctx_create(0, C.byref(p), C.byref(ctx)) < 0  # ENODEV, never a silent CPU path


def test_product_does_not_reference_the_oracle():
    pkg = os.path.join(ROOT, "mcmc-ammsb-gpu_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cc", ".cpp", ".inc", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle_lib" not in txt and "libammsb_oracle" not in txt and "ammsb_oracle.h" not in txt, f
