"""MI355X-native SG-MCMC a-MMSB hot path (drop-in for the device work of ielhelw/mcmc-ammsb-gpu).

Layout
    csrc/               hand-written gfx950 HIP kernels + the C ABI (include/ammsb.h) -> libammsb_hip.so
    _capi.py            ctypes binding of that ABI (raises if the library is missing: no CPU fallback)
    ops.py              host mirror of the reference operators (PhiUpdater, BetaUpdater, ...)

The directory name contains '-', so import it through `ammsb_pkg.load()` at the repo root, which
registers it as the module `mcmc_ammsb_gpu_amd`.
"""
from . import _capi  # noqa: F401
from ._capi import AmmsbError  # noqa: F401
