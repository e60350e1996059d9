"""ctypes binding of libmgcr_hip.so (the C ABI declared in include/mgcr.h).

The library is built in-tree (mgpreconditionedgcr_amd/csrc/Makefile -> libmgcr_hip.so next to
this file) and loaded from there.  There is no fallback of any kind: if the shared object is
missing, or no HIP device can be initialised, importing / initialising fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmgcr_hip.so")

OK = 0


class MgcrError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("mgcr error %d: %s" % (code, msg))
        self.code = code


class GcrParamC(C.Structure):
    """struct mgcr_gcr_param (include/mgcr.h) — mirror of GCR_Param (src/SolverParam.h:21-35)."""
    _fields_ = [("truncation", C.c_int32), ("restart", C.c_int32), ("max_iter", C.c_int32),
                ("tol", C.c_double), ("verbose", C.c_int32),
                ("left_precond", C.c_void_p), ("right_precond", C.c_void_p),
                ("use_x0", C.c_int32), ("flexible", C.c_int32), ("check_every", C.c_int32),
                ("profile_spmv", C.c_int32)]


class MgParamC(C.Structure):
    """struct mgcr_mg_param (include/mgcr.h) — mirror of MG_Param (src/SolverParam.h:38-59)."""
    _fields_ = [("ndim", C.c_int32), ("dims", C.c_int64 * 8), ("blocked", C.c_int32 * 8),
                ("subblock_dim", C.c_int64), ("n_vec", C.c_int32), ("vecs_ri", C.c_void_p),
                ("n_level", C.c_int32), ("smoother", GcrParamC), ("coarse", GcrParamC), ("damping", C.c_double),
                ("coarse_direct_rows", C.c_int32)]


ALLREDUCE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int64)
EXCHANGE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.POINTER(C.c_double)),
                          C.POINTER(C.c_int64), C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.c_int64))

_lib = None
_initialised_device = None

_vp = C.c_void_p
_dp = C.POINTER(C.c_double)
_SIGS = {
    # name: (restype, [argtypes])
    "mgcr_init": (C.c_int, [C.c_int]),
    "mgcr_finalize": (C.c_int, []),
    "mgcr_last_error": (C.c_char_p, []),
    "mgcr_version": (C.c_char_p, []),
    "mgcr_device_info": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "mgcr_synchronize": (C.c_int, []),
    "mgcr_vec_create": (C.c_int, [C.c_int64, C.POINTER(_vp)]),
    "mgcr_vec_destroy": (C.c_int, [_vp]),
    "mgcr_vec_size": (C.c_int64, [_vp]),
    "mgcr_vec_upload": (C.c_int, [_vp, _vp]),
    "mgcr_vec_download": (C.c_int, [_vp, _vp]),
    "mgcr_vec_copy": (C.c_int, [_vp, _vp]),
    "mgcr_vec_zero": (C.c_int, [_vp]),
    "mgcr_vec_set_constant": (C.c_int, [_vp, _dp]),
    "mgcr_vec_fill_rhs": (C.c_int, [_vp, C.c_uint64, C.c_int64]),
    "mgcr_dot": (C.c_int, [_vp, _vp, _dp]),
    "mgcr_norm2": (C.c_int, [_vp, _dp]),
    "mgcr_add_scaled": (C.c_int, [_vp, _vp, _dp, _vp]),
    "mgcr_axpy": (C.c_int, [_dp, _vp, _vp]),
    "mgcr_scale": (C.c_int, [_vp, _dp]),
    "mgcr_normalise": (C.c_int, [_vp]),
    "mgcr_vec_gamma5": (C.c_int, [_vp, _vp, C.c_int64]),
    "mgcr_csr_create": (C.c_int, [C.c_int64, C.c_int64, _vp, _vp, _vp, C.POINTER(_vp)]),
    "mgcr_dirac_create": (C.c_int, [_vp, _dp, C.POINTER(_vp)]),
    "mgcr_dirac_set_k": (C.c_int, [_vp, _dp]),
    "mgcr_bcsr_create_from_triplets": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, _vp, _vp, _vp, C.POINTER(_vp)]),
    "mgcr_bcsr_create": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _vp, _vp, _vp, C.POINTER(_vp)]),
    "mgcr_op_destroy": (C.c_int, [_vp]),
    "mgcr_op_dim": (C.c_int64, [_vp]),
    "mgcr_op_nrow": (C.c_int64, [_vp]),
    "mgcr_op_nnz": (C.c_int64, [_vp]),
    "mgcr_op_apply": (C.c_int, [_vp, _vp, _vp]),
    "mgcr_op_stored_bytes": (C.c_int, [_vp, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "mgcr_op_storage_format": (C.c_int, [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "mgcr_op_xr_fuse_kind": (C.c_int, [_vp, C.POINTER(C.c_int32)]),
    "mgcr_op_ell_layout": (C.c_int, [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "mgcr_csr_replace": (C.c_int, [_vp, C.c_int64, C.c_int64, _vp, _vp, _vp]),
    "mgcr_selftest_coherence": (C.c_int, [C.c_int32, C.c_int32, C.POINTER(C.c_int64)]),
    "mgcr_set_option": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_int)]),
    "mgcr_stat": (C.c_int, [C.c_char_p, C.POINTER(C.c_int64)]),
    "mgcr_gcr_solve": (C.c_int, [_vp, C.POINTER(GcrParamC), _vp, _vp, _vp, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "mgcr_gcr_create": (C.c_int, [_vp, C.POINTER(GcrParamC), C.c_int32, C.POINTER(_vp)]),
    "mgcr_gcr_set_operator": (C.c_int, [_vp, _vp]),
    "mgcr_gcr_set_x0": (C.c_int, [_vp, _vp]),
    "mgcr_gcr_set_param": (C.c_int, [_vp, C.POINTER(GcrParamC)]),
    "mgcr_gcr_solve_op": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "mgcr_mg_create": (C.c_int, [_vp, C.POINTER(MgParamC), C.POINTER(_vp)]),
    "mgcr_mg_level_info": (C.c_int, [_vp, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "mgcr_mg_restrict": (C.c_int, [_vp, C.c_int32, _vp, _vp]),
    "mgcr_mg_expand": (C.c_int, [_vp, C.c_int32, _vp, _vp]),
    "mgcr_mg_level_op": (C.c_int, [_vp, C.c_int32, C.POINTER(_vp)]),
    "mgcr_mg_download_prolongator": (C.c_int, [_vp, C.c_int32, _vp, _vp]),
    "mgcr_rccl_unique_id": (C.c_int, [_vp]),
    "mgcr_comm_create_rccl": (C.c_int, [C.c_int, C.c_int, _vp, C.POINTER(_vp)]),
    "mgcr_comm_create_host": (C.c_int, [C.c_int, C.c_int, ALLREDUCE_CB, EXCHANGE_CB, _vp, C.POINTER(_vp)]),
    "mgcr_comm_destroy": (C.c_int, [_vp]),
    "mgcr_comm_allreduce_sum": (C.c_int, [_vp, _dp, C.c_int32]),
    "mgcr_comm_allreduce_kind": (C.c_int, [_vp, C.POINTER(C.c_int32)]),
    "mgcr_op_halo_kind": (C.c_int, [_vp, C.POINTER(C.c_int32)]),
    "mgcr_comm_bench_allreduce": (C.c_int, [_vp, C.c_int32, C.c_int32, _dp]),
    "mgcr_plan_create": (C.c_int, [_vp, C.c_int64, C.c_int64, C.c_int64, _vp, _vp, C.POINTER(_vp)]),
    "mgcr_plan_info": (C.c_int, [_vp, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "mgcr_plan_peers": (C.c_int, [_vp, _vp, _vp, _vp]),
    "mgcr_plan_local_columns": (C.c_int, [_vp, _vp]),
    "mgcr_plan_send_indices": (C.c_int, [_vp, C.c_int32, _vp]),
    "mgcr_plan_halo_globals": (C.c_int, [_vp, _vp]),
    "mgcr_plan_destroy": (C.c_int, [_vp]),
    "mgcr_dcsr_create": (C.c_int, [_vp, C.c_int64, C.c_int64, C.c_int64, _vp, _vp, _vp, C.POINTER(_vp)]),
    "mgcr_dbcsr_create": (C.c_int, [_vp, C.c_int64, C.c_int64, C.c_int32, C.c_int32, _vp, _vp, _vp, C.POINTER(_vp)]),
    "mgcr_set_small_solve_rows": (C.c_int, [C.c_int64]),
    "mgcr_gcr_last_profile": (C.c_int, [_dp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "mgcr_bench_op_apply": (C.c_int, [_vp, _vp, _vp, C.c_int32, _dp]),
    "mgcr_timer_start": (C.c_int, []),
    "mgcr_timer_stop": (C.c_int, [_dp]),
}


def exported_symbols():
    """Names every entry point include/mgcr.h declares (used by the CPU-side ABI test)."""
    return sorted(_SIGS)


def lib():
    """Load libmgcr_hip.so (no device needed for loading)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` or "
                "`make -C mgpreconditionedgcr_amd/csrc` (hipcc, gfx950). There is no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != OK:
        raise MgcrError(rc, lib().mgcr_last_error().decode(errors="replace"))


def init(device=None):
    """mgcr_init on `device` (default: LOCAL_RANK or 0). Raises if no GPU is usable."""
    global _initialised_device
    if device is None:
        if _initialised_device is not None:
            return
        device = int(os.environ.get("LOCAL_RANK", "0"))
    if _initialised_device is not None:
        if _initialised_device != device:
            raise MgcrError(1, "already initialised on device %d" % _initialised_device)
        return
    check(lib().mgcr_init(device))
    _initialised_device = device


def finalize():
    global _initialised_device
    if _initialised_device is not None:
        check(lib().mgcr_finalize())
        _initialised_device = None
