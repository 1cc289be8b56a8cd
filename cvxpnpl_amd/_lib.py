"""ctypes binding of the C ABI (include/cvxpnpl_amd.h).

The shared library is built in-tree by ``__graft_entry__.build()`` /
``python -m cvxpnpl_amd.build``.  There is NO CPU fallback: if the library is missing,
or no GPU is visible, the solver entry points raise.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CVXPNPL_AMD_LIB: diagnostics only (tools/phase_profile.py loads an instrumented build through it)
LIB_PATH = os.environ.get("CVXPNPL_AMD_LIB") or os.path.join(_HERE, "libcvxpnpl_amd.so")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)

# every symbol include/cvxpnpl_amd.h declares
EXPORTS = (
    "cvxpnpl_default_opts", "cvxpnpl_opts_size", "cvxpnpl_solve_batch", "cvxpnpl_solve_cost_batch", "cvxpnpl_recover_multi", "cvxpnpl_recover_multi_batch", "cvxpnpl_recover_multi_device", "cvxpnpl_assemble_batch",
    "cvxpnpl_assemble_large_batch", "cvxpnpl_assemble_large_scratch_bytes",
    "cvxpnpl_score_hypotheses", "cvxpnpl_select_best", "cvxpnpl_refit_update", "cvxpnpl_sample_minimal_sets", "cvxpnpl_assemble_subsets", "cvxpnpl_pack_results", "cvxpnpl_stream_write_value", "cvxpnpl_stream_wait_value", "cvxpnpl_stream_wait_value_bounded", "cvxpnpl_stream_wait_gave_up", "cvxpnpl_synth_batch", "cvxpnpl_pose_errors", "cvxpnpl_disambiguate",
    "cvxpnpl_workspace_bytes", "cvxpnpl_set_workspace", "cvxpnpl_release_workspace", "cvxpnpl_calibration_copy", "cvxpnpl_ipm_batch",
    "cvxpnpl_event_create", "cvxpnpl_event_record", "cvxpnpl_event_elapsed_ms", "cvxpnpl_event_destroy",
    "cvxpnpl_last_error", "cvxpnpl_last_layout", "cvxpnpl_version", "cvxpnpl_device_count",
)

VARIANT_FULL, VARIANT_RC = 0, 1
STATUS_NAMES = {0: "certified", 1: "rank>1", 2: "uncertified", 3: "nonfinite", 4: "reflection"}


class Opts(C.Structure):
    """cvxpnpl_opts_t"""
    _fields_ = [
        ("struct_size", C.c_uint32), ("eps", C.c_double), ("max_iters", C.c_int32), ("rho", C.c_double), ("alpha", C.c_double),
        ("first_check", C.c_int32), ("check_every", C.c_int32), ("res_tol", C.c_double),
        ("jacobi_sweeps", C.c_int32), ("jacobi_tol", C.c_double), ("warm_start", C.c_int32), ("rho_tail", C.c_double), ("tail_from", C.c_int32), ("lane_iters", C.c_int32), ("layout", C.c_int32),
        ("variant", C.c_int32), ("adapt_every", C.c_int32), ("adapt_from", C.c_int32), ("adapt_mu", C.c_double), ("adapt_tau", C.c_double),
        ("stall_from", C.c_int32), ("stall_lam", C.c_double), ("stall_res", C.c_double), ("stall_drop", C.c_double), ("rescue_from", C.c_int32),
        ("f32_sweeps_until", C.c_int32), ("sweep_schedule", C.c_int32), ("dual_shift", C.c_double), ("dual_refine", C.c_int32),
    ]


class LibraryMissing(RuntimeError):
    pass


_lib = None


def lib():
    """Load libcvxpnpl_amd.so (loudly)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LibraryMissing(
            f"{LIB_PATH} is missing: the HIP extension has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'` or `python -m cvxpnpl_amd.build`). "
            "cvxpnpl_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    L.cvxpnpl_default_opts.argtypes = [C.POINTER(Opts)]
    L.cvxpnpl_default_opts.restype = None
    L.cvxpnpl_opts_size.argtypes = []
    L.cvxpnpl_opts_size.restype = C.c_size_t
    if L.cvxpnpl_opts_size() != C.sizeof(Opts):  # the hand-written mirror above and the header must move in lock-step
        raise ImportError(f"cvxpnpl_amd._lib.Opts has {C.sizeof(Opts)} bytes, {LIB_PATH} expects {L.cvxpnpl_opts_size()} (include/cvxpnpl_amd.h)")
    L.cvxpnpl_solve_batch.argtypes = [C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_int32, C.POINTER(Opts), C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.cvxpnpl_solve_batch.restype = C.c_int
    L.cvxpnpl_solve_cost_batch.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.POINTER(Opts), C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.cvxpnpl_solve_cost_batch.restype = C.c_int
    L.cvxpnpl_assemble_batch.argtypes = [C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.cvxpnpl_assemble_batch.restype = C.c_int
    L.cvxpnpl_assemble_large_scratch_bytes.argtypes = [C.c_int64, C.c_int32, C.c_int32]
    L.cvxpnpl_assemble_large_scratch_bytes.restype = C.c_size_t
    L.cvxpnpl_assemble_large_batch.argtypes = [C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.cvxpnpl_assemble_large_batch.restype = C.c_int
    L.cvxpnpl_recover_multi.argtypes = [_dp, _dp, _dp, _dp, _dp]
    L.cvxpnpl_recover_multi.restype = C.c_int
    L.cvxpnpl_recover_multi_batch.argtypes = [C.c_int64, _ip, _dp, _dp, _dp, _dp, _dp, _ip, C.c_int32]
    L.cvxpnpl_recover_multi_batch.restype = C.c_int
    L.cvxpnpl_recover_multi_device.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.cvxpnpl_recover_multi_device.restype = C.c_int
    L.cvxpnpl_pack_results.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.cvxpnpl_pack_results.restype = C.c_int
    L.cvxpnpl_stream_write_value.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
    L.cvxpnpl_stream_write_value.restype = C.c_int
    L.cvxpnpl_stream_wait_value.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
    L.cvxpnpl_stream_wait_value.restype = C.c_int
    if not os.environ.get("CVXPNPL_AMD_LIB") or hasattr(L, "cvxpnpl_stream_wait_value_bounded"):  # (an A/B library of an earlier round, loaded
        # through the diagnostics variable, may predate these two entries; the product library must have them)
        L.cvxpnpl_stream_wait_value_bounded.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
        L.cvxpnpl_stream_wait_value_bounded.restype = C.c_int
        L.cvxpnpl_stream_wait_gave_up.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        L.cvxpnpl_stream_wait_gave_up.restype = C.c_int
    L.cvxpnpl_score_hypotheses.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int32,
                                           C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
    L.cvxpnpl_score_hypotheses.restype = C.c_int
    if not os.environ.get("CVXPNPL_AMD_LIB") or hasattr(L, "cvxpnpl_select_best"):  # (an A/B library of an earlier round lacks them)
        L.cvxpnpl_select_best.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                          C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.cvxpnpl_select_best.restype = C.c_int
        L.cvxpnpl_refit_update.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_double,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.cvxpnpl_refit_update.restype = C.c_int
    L.cvxpnpl_sample_minimal_sets.argtypes = [C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.cvxpnpl_sample_minimal_sets.restype = C.c_int
    L.cvxpnpl_synth_batch.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.c_double, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.cvxpnpl_synth_batch.restype = C.c_int
    L.cvxpnpl_pose_errors.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.cvxpnpl_pose_errors.restype = C.c_int
    L.cvxpnpl_disambiguate.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.cvxpnpl_disambiguate.restype = C.c_int
    L.cvxpnpl_workspace_bytes.argtypes = [C.c_int64]
    L.cvxpnpl_workspace_bytes.restype = C.c_size_t
    L.cvxpnpl_set_workspace.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    L.cvxpnpl_set_workspace.restype = C.c_int
    L.cvxpnpl_release_workspace.argtypes = [C.c_void_p, C.c_int32]
    L.cvxpnpl_release_workspace.restype = C.c_int
    L.cvxpnpl_calibration_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]
    L.cvxpnpl_calibration_copy.restype = C.c_int
    L.cvxpnpl_assemble_subsets.argtypes = [C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.cvxpnpl_assemble_subsets.restype = C.c_int
    L.cvxpnpl_ipm_batch.argtypes = [C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.cvxpnpl_ipm_batch.restype = C.c_int
    L.cvxpnpl_event_create.restype = C.c_void_p
    L.cvxpnpl_event_record.argtypes = [C.c_void_p, C.c_void_p]
    L.cvxpnpl_event_elapsed_ms.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
    L.cvxpnpl_event_destroy.argtypes = [C.c_void_p]
    L.cvxpnpl_event_destroy.restype = None
    L.cvxpnpl_last_error.restype = C.c_char_p
    L.cvxpnpl_version.restype = C.c_char_p
    L.cvxpnpl_device_count.restype = C.c_int
    _lib = L
    return L


def default_opts(**overrides):
    o = Opts()
    lib().cvxpnpl_default_opts(C.byref(o))
    for k, v in overrides.items():
        if v is None:
            continue
        if not hasattr(o, k):
            raise TypeError(f"unknown solver option {k!r}")
        setattr(o, k, v)
    return o


def last_error():
    return lib().cvxpnpl_last_error().decode()
