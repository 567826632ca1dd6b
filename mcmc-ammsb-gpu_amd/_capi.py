"""ctypes view of libammsb_hip.so (the C ABI declared in include/ammsb.h).

There is no CPU fallback: if the shared object is missing or a call fails, this module raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AMMSB_HIP_LIB") or os.path.join(_HERE, "libammsb_hip.so")  # override: A/B runs of two builds

RPM_MAX_BLOCKS = 32
MAX_GROUPS = 65535
NOISE_OFF = 1
PHI_STREAMING = 2  # AMMSB_PHI_STREAMING


class AmmsbError(RuntimeError):
    pass


class Params(C.Structure):
    _fields_ = [("N", C.c_uint64), ("K", C.c_uint64), ("E", C.c_uint64),
                ("num_node_sample", C.c_uint32),
                ("alpha", C.c_float), ("a", C.c_float), ("b", C.c_float), ("c", C.c_float),
                ("epsilon", C.c_float), ("eta0", C.c_float), ("eta1", C.c_float)]


class Rpm(C.Structure):
    _fields_ = [("blocks", C.c_void_p * RPM_MAX_BLOCKS), ("rows_in_block", C.c_uint64),
                ("num_rows", C.c_uint64), ("num_cols", C.c_uint64), ("num_blocks", C.c_uint32)]


class SetDesc(C.Structure):
    _fields_ = [("slots", C.c_void_p), ("num_bins", C.c_uint64), ("prime_idx", C.c_uint32)]


class PpxSums(C.Structure):
    _fields_ = [("link_ll", C.c_double), ("nonlink_ll", C.c_double),
                ("link_cnt", C.c_uint64), ("nonlink_cnt", C.c_uint64)]


_vp, _u32, _u64, _f32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_float
_P = C.POINTER

LOOP_TIMESTAMPS = 1


class LoopConfig(C.Structure):  # ammsb_loop_config
    _fields_ = [("theta", _vp), ("beta", _vp), ("pi", _P(Rpm)), ("phi_sum", _vp),
                ("training_set", _P(SetDesc)), ("heldout_set", _P(SetDesc)),
                ("phi_seeds", _vp), ("phi_vec", _vp), ("phi_wg", _u32), ("phi_flags", _u32),
                ("beta_seeds", _vp), ("grads", _vp), ("beta_wg", _u32), ("beta_flags", _u32),
                ("edges", _vp * 2), ("nodes", _vp * 2), ("neighbors", _vp * 2), ("nbr_table", _vp * 2),
                ("nbr_seeds", _vp * 2), ("nbr_wg", _u32),
                ("csr_offsets", _vp), ("csr_targets", _vp), ("mb_seeds", _vp), ("mb_candidates", _u32),
                ("mb_workspace", _vp), ("mb_count", _vp), ("mini_batch", _u32), ("max_fan_out", _u32),
                ("max_nodes", _u32), ("max_edges", _u32), ("flags", _u32)]


class MbChoice(C.Structure):  # ammsb_mb_choice
    _fields_ = [("link", _u32), ("u", _u32), ("n", _u32), ("n_candidates", _u32)]


# name -> argtypes (all return int unless listed in _OTHER_RES)
SIGNATURES = {
    "ammsb_version": [],
    "ammsb_strerror": [C.c_int],
    "ammsb_last_error": [_vp],
    "ammsb_last_kernel_name": [_vp, C.c_int],
    "ammsb_params_quantize": [_P(Params)],
    "ammsb_eps_t": [_P(Params), _u32],
    "ammsb_ctx_create": [C.c_int, _P(Params), _P(_vp)],
    "ammsb_ctx_destroy": [_vp],
    "ammsb_theta_sum": [_vp, _vp, _vp],
    "ammsb_ctx_params": [_vp, _P(Params)],
    "ammsb_rng_init": [_vp, _vp, _u64, _u64, _u64, _vp],
    "ammsb_rng_init_mixed": [_vp, _vp, _u64, _u64, _u64, _vp],
    "ammsb_set_has": [_vp, _P(SetDesc), _vp, _u64, _vp, _vp],
    "ammsb_set_num_bins": [_u64],
    "ammsb_set_build": [_vp, _vp, _u64, _vp, _u64, _P(_u32), _vp, _vp],
    "ammsb_pi_init_gamma": [_vp, _P(Rpm), _vp, _f32, _f32, _vp, _vp],
    "ammsb_sample_neighbors": [_vp, _vp, _vp, _u32, _u32, _vp, _vp, _vp],
    "ammsb_update_phi": [_vp, _vp, _P(Rpm), _vp, _P(SetDesc), _vp, _vp, _u32, _u32, _vp, _u32, _u32,
                         _u32, _u32, _vp, _vp],
    "ammsb_update_phi_occupancy": [_vp, _u32, _P(C.c_int), _P(C.c_int)],
    "ammsb_update_pi": [_vp, _P(Rpm), _vp, _vp, _vp, _u32, _u32, _vp],
    "ammsb_beta_grads": [_vp, _vp, _vp, _P(Rpm), _P(SetDesc), _vp, _u32, _u32, _u32, _u32, _vp, _vp],
    "ammsb_can_fuse_pi_beta": [_vp, _u32, _u32],
    "ammsb_update_pi_beta_grads": [_vp, _vp, _vp, _P(Rpm), _vp, _vp, _vp, _P(SetDesc), _vp, _u32, _u32, _vp, _vp],
    "ammsb_sum_rows_f32": [_vp, _vp, _u32, _u32, _vp, _vp],
    "ammsb_update_theta": [_vp, _vp, _vp, _vp, _u32, _f32, _vp, _u32, _vp],
    "ammsb_beta_from_theta": [_vp, _vp, _vp, _vp],
    "ammsb_perplexity": [_vp, _vp, _P(Rpm), _P(SetDesc), _vp, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _vp],
    "ammsb_minibatch_link": [_vp, _vp, _vp, _u32, _u32, _vp, _vp, _vp],
    "ammsb_minibatch_candidates": [_u64, _u32],
    "ammsb_minibatch_candidates_for": [_u64, _u32, _u64],
    "ammsb_minibatch_workspace_bytes": [_u32],
    "ammsb_minibatch_nonlink": [_vp, _vp, _u32, _u32, _u32, _u32, _P(SetDesc), _P(SetDesc), _vp, _vp, _vp, _vp, _vp],
    "ammsb_loop_create": [_vp, _P(LoopConfig), _P(_vp)],
    "ammsb_loop_destroy": [_vp],
    "ammsb_loop_run": [_vp, _P(MbChoice), _P(MbChoice), _u32, _u32, _u32, _vp],
    "ammsb_loop_check": [_vp, _P(_u32)],
    "ammsb_loop_status": [_vp, _P(_u32), _P(_u32)],
    "ammsb_loop_timestamps": [_vp, _u32, _u32, _P(C.c_double), _P(C.c_double)],
    "ammsb_loop_step_stamps": [_vp, _u32, _u32, _P(C.c_double)],
    "ammsb_wg_sum_f32": [_vp, _vp, _vp, _u32, _u32, _u32, _vp],
    "ammsb_wg_sum_u32": [_vp, _vp, _vp, _u32, _u32, _u32, _vp],
    "ammsb_wg_normalize_f32": [_vp, _vp, _vp, _u32, _u32, _u32, _vp],
    "ammsb_rpm_sum_f32": [_vp, _P(Rpm), _vp, _u32, _vp],
    "ammsb_rpm_normalize_f32": [_vp, _P(Rpm), _vp, _u32, _vp],
    "ammsb_wg_sort_u32": [_vp, _vp, _vp, _u32, _vp],
    "ammsb_wg_sort_f32": [_vp, _vp, _vp, _u32, _vp],
    "ammsb_randn_fill": [_vp, _vp, _u32, _u32, _vp, _vp],
    "ammsb_rpm_fetch": [_vp, _P(Rpm), _u64, _u64, _vp, _vp],
    "ammsb_clock_probe": [_vp, _vp, _u32, _u32, _vp],
}
_OTHER_RES = {"ammsb_strerror": C.c_char_p, "ammsb_last_error": C.c_char_p, "ammsb_last_kernel_name": C.c_char_p,
              "ammsb_eps_t": C.c_float,
              "ammsb_minibatch_candidates": C.c_uint32, "ammsb_minibatch_candidates_for": C.c_uint32,
              "ammsb_minibatch_workspace_bytes": C.c_uint64, "ammsb_set_num_bins": C.c_uint64}

_lib = None


def load(path=None):
    """dlopen the HIP library and bind every symbol include/ammsb.h declares."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    so = path or LIB_PATH
    if not os.path.exists(so):
        raise AmmsbError("%s not found: build it with `make -C mcmc-ammsb-gpu_amd/csrc` "
                         "(or __graft_entry__.build()); there is no CPU fallback" % so)
    # One HIP runtime per process: torch carries its own libamdhip64, and device pointers / streams are
    # handed from torch to this library.  Loading torch first makes the library bind to that copy; the
    # other order gives two runtimes that cannot see each other's devices.
    import torch  # noqa: F401
    lib = C.CDLL(so)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.argtypes = args
        fn.restype = _OTHER_RES.get(name, C.c_int)
    if path is None:
        _lib = lib
    return lib


def check(rc, ctx=None):
    if rc != 0:
        lib = load()
        msg = lib.ammsb_strerror(rc).decode()
        if ctx:
            msg += " (" + lib.ammsb_last_error(ctx).decode() + ")"
        raise AmmsbError("ammsb call failed: %d %s" % (rc, msg))
