"""TEST INFRASTRUCTURE — ctypes/numpy front-end of the CPU oracle (oracle/libmgcr_oracle.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product package (mgpreconditionedgcr_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmgcr_oracle.so")
_lib = None

c128 = np.complex128
_cp = np.ctypeslib.ndpointer(dtype=c128, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


class GcrParamC(C.Structure):
    _fields_ = [("truncation", C.c_int), ("restart", C.c_int), ("max_iter", C.c_int),
                ("tol", C.c_double), ("verbose", C.c_int),
                ("left_precond", C.c_void_p), ("right_precond", C.c_void_p),
                ("use_x0", C.c_int), ("flexible", C.c_int)]


def build():
    """Compile the oracle (gcc) if the .so is missing or older than its sources."""
    srcs = [os.path.join(_HERE, f) for f in ("mgcr_oracle.c", "mgcr_oracle_mg.c")]
    opt = os.path.join(_HERE, "libmgcr_cpu_opt.so")
    if (not os.path.exists(_LIB_PATH) or not os.path.exists(opt)
            or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
            or os.path.getmtime(os.path.join(_HERE, "mgcr_cpu_opt.c")) > os.path.getmtime(opt)):
        subprocess.run(["make", "-C", _HERE, "oracle"], check=True, capture_output=True)


def opt_gcr_poisson(n, restart, iters, nthreads=0):
    """Optimised CPU port (oracle/mgcr_cpu_opt.c: OpenMP, fused passes) on Poisson n^3, RHS seed 0, x0 = 0:
    returns (seconds in the iterations, history[0..iters]).  bench.py's second CPU baseline."""
    build()
    L = C.CDLL(os.path.join(_HERE, "libmgcr_cpu_opt.so"))
    L.orc_opt_gcr_poisson.argtypes = [C.c_int64, C.c_int, C.c_int, C.c_int, _f64p]
    L.orc_opt_gcr_poisson.restype = C.c_double
    hist = np.zeros(iters + 1)
    dt = L.orc_opt_gcr_poisson(n, restart, iters, nthreads, hist)
    return dt, hist


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_dot.argtypes = [C.c_int64, _cp, _cp, _cp]
        L.orc_sqnorm.argtypes = [C.c_int64, _cp]
        L.orc_sqnorm.restype = C.c_double
        L.orc_add_scaled.argtypes = [C.c_int64, _cp, _cp, _cp, _cp]
        L.orc_sub_scaled.argtypes = [C.c_int64, _cp, _cp, _cp, _cp]
        L.orc_normalise.argtypes = [C.c_int64, _cp]
        L.orc_op_csr.argtypes = [C.c_int64, C.c_int64, _i64p, _i64p, _cp]
        L.orc_op_csr.restype = C.c_void_p
        L.orc_op_dirac.argtypes = [C.c_void_p, C.c_double, C.c_double]
        L.orc_op_dirac.restype = C.c_void_p
        L.orc_op_bcsr_from_triplets.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, _i32p, _i32p, _cp]
        L.orc_op_bcsr_from_triplets.restype = C.c_void_p
        L.orc_bcsr_nblocks.argtypes = [C.c_void_p]
        L.orc_bcsr_nblocks.restype = C.c_int32
        L.orc_bcsr_export.argtypes = [C.c_void_p, _i32p, _i32p, _cp]
        L.orc_bcsr_val_at.argtypes = [C.c_void_p, C.c_int64, C.c_int64, _cp]
        L.orc_op_dim.argtypes = [C.c_void_p]
        L.orc_op_dim.restype = C.c_int64
        L.orc_set_sum_order.argtypes = [C.c_int]
        L.orc_set_device_model.argtypes = [C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_set_device_ranks.argtypes = [C.c_int, _i64p]
        L.orc_op_set_layout.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.orc_mg_set_vectors_are_prolongator.argtypes = [C.c_int]
        L.orc_op_set_rowmap.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_int, C.c_int]
        L.orc_set_device_lean.argtypes = [C.c_int, C.c_int]
        L.orc_set_device_xr_banded.argtypes = [C.c_int]
        L.orc_set_device_plane.argtypes = [C.c_int64]
        L.orc_op_nrow.argtypes = [C.c_void_p]
        L.orc_op_nrow.restype = C.c_int64
        L.orc_op_apply.argtypes = [C.c_void_p, _cp, _cp]
        L.orc_op_free.argtypes = [C.c_void_p]
        L.orc_op_gcr.argtypes = [C.c_void_p, C.POINTER(GcrParamC), C.c_int]
        L.orc_op_gcr.restype = C.c_void_p
        L.orc_op_gcr_set_operator.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_op_gcr_set_x0.argtypes = [C.c_void_p]
        L.orc_gcr_solve.argtypes = [C.c_void_p, C.POINTER(GcrParamC), _cp, _cp, _f64p, C.c_int, C.POINTER(C.c_int)]
        L.orc_gcr_solve.restype = C.c_int
        L.orc_read_text_csr.argtypes = [C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                        C.POINTER(C.c_int64), C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_read_text_csr.restype = C.c_int
        L.orc_mg_aggregates.argtypes = [C.c_int, _i64p, _i32p, C.c_int64, _i32p]
        L.orc_mg_aggregates.restype = C.c_int64
        L.orc_mg_prolongator.argtypes = [C.c_int64, C.c_int, C.c_int64, _i32p, _cp, _cp]
        L.orc_mg_restrict.argtypes = [C.c_int64, C.c_int, C.c_int64, _i32p, _cp, _cp, _cp]
        L.orc_mg_expand.argtypes = [C.c_int64, C.c_int, _i32p, _cp, _cp, _cp]
        L.orc_mg_galerkin.argtypes = [C.c_int64, _i64p, _i64p, _cp, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int64,
                                      _i32p, _cp, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_mg_galerkin.restype = C.c_int64
        L.orc_mg_create.argtypes = [C.c_void_p, C.c_int64, _i64p, _i64p, _cp, C.c_int, C.c_double, C.c_double, C.c_int,
                                    _i64p, _i32p, C.c_int64, C.c_int, _cp, C.c_int, C.POINTER(GcrParamC),
                                    C.POINTER(GcrParamC), C.c_double]
        L.orc_mg_create.restype = C.c_void_p
        L.orc_mg_level_dim.argtypes = [C.c_void_p, C.c_int]
        L.orc_mg_level_dim.restype = C.c_int64
        L.orc_mg_level_ne.argtypes = [C.c_void_p, C.c_int]
        L.orc_mg_level_nagg.argtypes = [C.c_void_p, C.c_int]
        L.orc_mg_level_nagg.restype = C.c_int64
        L.orc_mg_level_pv.argtypes = [C.c_void_p, C.c_int]
        L.orc_mg_level_pv.restype = C.POINTER(C.c_double)
        L.orc_mg_level_agg.argtypes = [C.c_void_p, C.c_int]
        L.orc_mg_level_agg.restype = C.POINTER(C.c_int32)
        L.orc_mg_level_op.argtypes = [C.c_void_p, C.c_int]
        L.orc_mg_level_op.restype = C.c_void_p
        L.orc_mg_level_restrict.argtypes = [C.c_void_p, C.c_int, _cp, _cp]
        L.orc_mg_level_expand.argtypes = [C.c_void_p, C.c_int, _cp, _cp]
        L.orc_op_mg.argtypes = [C.c_void_p, C.c_int64]
        L.orc_op_mg.restype = C.c_void_p
        L.orc_fill_rhs.argtypes = [C.c_int64, C.c_uint64, _cp]
        L.orc_poisson3d.argtypes = [C.c_int64, _i64p, _i64p, _cp]
        _lib = L
    return _lib


def _c(a):
    return np.ascontiguousarray(a, dtype=c128)


# ---------------------------------------------------------------- Field algebra
def dot(a, b):
    out = np.zeros(1, c128)
    lib().orc_dot(a.size, _c(a), _c(b), out)
    return out[0]


def sqnorm(a):
    return lib().orc_sqnorm(a.size, _c(a))


def add_scaled(a, b, alpha):
    out = np.empty(a.size, c128)
    lib().orc_add_scaled(a.size, _c(a), _c(b), np.array([alpha], c128), out)
    return out


def sub_scaled(a, b, alpha):
    out = np.empty(a.size, c128)
    lib().orc_sub_scaled(a.size, _c(a), _c(b), np.array([alpha], c128), out)
    return out


# ---------------------------------------------------------------- operators
class Op:
    """Opaque oracle operator handle; keeps the numpy buffers it borrows alive."""

    def __init__(self, handle, keep=()):
        self.h = handle
        self._keep = list(keep)

    @property
    def dim(self):
        return lib().orc_op_dim(self.h)

    def set_layout(self, ell_width, ell_lanes, tail_cap):
        """This operator's own device layout for summation order 3 (the levels of a multigrid hierarchy are stored differently)."""
        lib().orc_op_set_layout(self.h, int(ell_width), int(ell_lanes), int(tail_cap))
        return self

    def set_rowmap(self, band, per, plane=0, init_banded=False, xr_banded=False):
        """This operator's own row map for summation order 3 — what `device_order(band=, per=, plane=, init_banded=, xr_banded=)` sets
        globally, for the GCR solves ON this operator only (the levels of a multigrid hierarchy)."""
        lib().orc_op_set_rowmap(self.h, int(band), int(per), int(plane), int(bool(init_banded)), int(bool(xr_banded)))
        return self

    def __call__(self, x):
        if np.size(x) != self.dim:
            raise ValueError("operator has %d columns, field has %d entries" % (self.dim, np.size(x)))
        y = np.empty(lib().orc_op_nrow(self.h), c128)
        lib().orc_op_apply(self.h, _c(x), y)
        return y


def csr(nrow, ncol, rowptr, col, val):
    rowptr = np.ascontiguousarray(rowptr, np.int64)
    col = np.ascontiguousarray(col, np.int64)
    val = _c(val)
    return Op(lib().orc_op_csr(nrow, ncol, rowptr, col, val), keep=(rowptr, col, val))


def dirac(D, k):
    k = complex(k)
    return Op(lib().orc_op_dirac(D.h, k.real, k.imag), keep=(D,))


def bcsr_from_triplets(nbrow, nbcol, bs, rows, cols, blocks):
    rows = np.ascontiguousarray(rows, np.int32)
    cols = np.ascontiguousarray(cols, np.int32)
    blocks = _c(blocks)
    return Op(lib().orc_op_bcsr_from_triplets(nbrow, nbcol, bs, rows.size, rows, cols, blocks))


def bcsr_export(op, nbrow, bs):
    nb = lib().orc_bcsr_nblocks(op.h)
    browptr = np.empty(nbrow + 1, np.int32)
    bcol = np.empty(nb, np.int32)
    blocks = np.empty(nb * bs * bs, c128)
    lib().orc_bcsr_export(op.h, browptr, bcol, blocks)
    return browptr, bcol, blocks.reshape(nb, bs, bs)


def bcsr_val_at(op, row, col):
    out = np.zeros(1, c128)
    lib().orc_bcsr_val_at(op.h, row, col, out)
    return out[0]


def gcr_param(truncation=0, restart=0, max_iter=100, tol=1e-16, verbose=False, left=None, right=None,
              use_x0=False, flexible=False):
    """Mirror of GCR_Param(trunc, re, max_it, tol, verb, l, r) (src/SolverParam.h:33)."""
    p = GcrParamC(truncation, restart, max_iter, tol, int(verbose),
                  left.h if left is not None else None, right.h if right is not None else None,
                  int(use_x0), int(flexible))
    p._keep = (left, right)
    return p


def gcr_solve(A, param, rhs, x0=None):
    """GCR::solve(rhs, x) (src/GCR.h:158-302).  Returns (x, history, n_iter, converged)."""
    n = A.dim
    x = np.zeros(n, c128) if x0 is None else _c(x0).copy()
    cap = param.max_iter + 2
    hist = np.zeros(cap, np.float64)
    conv = C.c_int(0)
    it = lib().orc_gcr_solve(A.h, C.byref(param), _c(rhs), x, hist, cap, C.byref(conv))
    return x, hist[: it + 1].copy(), it, bool(conv.value)


def gcr_reorder_sensitivity(A, param, rhs, x0=None):
    """How much the residual history of THIS solve moves when the reference's own dot products are
    summed in another, equally valid order (reverse, pairwise).  Returns (h_ref, s, (it_min, it_max)):
    s[k] = max over the orders of |h_order(k) - h_ref(k)|, and the range of iteration counts the
    orders (index order included) needed.  A parallel reduction cannot do better than this."""
    _, ref, it0, _ = gcr_solve(A, param, rhs, x0)
    dev = np.zeros(ref.size)
    its = [it0]
    try:
        for mode in (1, 2):
            lib().orc_set_sum_order(mode)
            _, h, it, _ = gcr_solve(A, param, rhs, x0)
            its.append(it)
            m = min(h.size, ref.size)
            dev[:m] = np.maximum(dev[:m], np.abs(h[:m] - ref[:m]))
            dev[m:] = np.inf  # that order had already converged: steps beyond it are not comparable
    finally:
        lib().orc_set_sum_order(0)
    return ref, dev, (min(its), max(its))


class device_order:
    """Context manager: inside it the oracle sums its dot products in the order the HIP library does
    (orc_set_sum_order(3), model in mgcr_oracle.c) and, when `ell_width` is given, forms SpMV rows the way the
    multi-lane / CSR-tail kernels do.  `blocks`: 0 = the multi-kernel solver's grid (red_grid(n)), 1 = the
    one-workgroup solver.  `band`, `per`, `init_banded`: the RowMap of the kernels that embed the operator apply
    (gcr_dev.h:make_row_map), needed from about 182^3 rows on (`row_map(n, reach)` below computes them)."""

    def __init__(self, blocks=0, band=0, per=0, init_banded=False, ell_width=-1, ell_lanes=1, tail_cap=0, rank_offsets=None, lean=False,
                 recurrence_residual=False, xr_banded=False, plane=0):
        """rank_offsets: first row of every rank's block (+ the total) of a distributed solve — the ranks sum their rows
        separately and add the totals in rank order.  lean: x is formed the way the device's lean restart cycles form it (restart
        mode <= 16 slots without the literal preconditioner hooks), so that x is comparable bit for bit too; recurrence_residual:
        the oracle's V-cycle restricts the residual the pre-smoother's recurrence ended with, like the device's default.
        xr_banded: the residual update runs inside the windowed apply kernel (csrc/gcr_fused_xr_tile.h; `xr_banded_for` below says
        when), so |r|^2 of every step but a solve's last is summed over the banded map."""
        self.args = (int(blocks), int(band), int(per), int(bool(init_banded)), int(ell_width), int(ell_lanes), int(tail_cap))
        self.ranks = None if rank_offsets is None else np.ascontiguousarray(rank_offsets, np.int64)
        self.lean = (int(bool(lean)), int(bool(recurrence_residual)))
        self.xr_banded = int(bool(xr_banded))
        self.plane = int(plane)   # RowMap::plane of the ragged plane walk (`row_map_plane`)

    def __enter__(self):
        lib().orc_set_device_model(*self.args)
        if self.ranks is not None:
            lib().orc_set_device_ranks(self.ranks.size - 1, self.ranks)
        lib().orc_set_device_lean(*self.lean)
        lib().orc_set_device_xr_banded(self.xr_banded)
        lib().orc_set_device_plane(self.plane)
        lib().orc_set_sum_order(3)
        return self

    def __exit__(self, *exc):
        lib().orc_set_sum_order(0)
        lib().orc_set_device_lean(0, 0)
        lib().orc_set_device_xr_banded(0)
        lib().orc_set_device_plane(0)
        lib().orc_set_device_model(0, 0, 0, 0, -1, 1, 0)
        lib().orc_set_device_ranks(1, np.zeros(2, np.int64))
        return False


def _row_map3(n, reach, plane_walk=True):
    g = min(max((n + 1023) // 1024, 1), 512)
    slice_ = g * 1024 // 8
    wide = (2 * reach >= slice_) if reach > 0 else (n >= 1 << 23)
    if g >= 64 and g % 8 == 0 and wide:
        per, nb = g // 8, 8
        T = reach // 1024 if reach > 0 and reach % 1024 == 0 else 0
        if plane_walk and g == 512 and 32 <= T <= 512 and T != 64:
            per, nb = T, 64 // ((T + 7) // 8)
        band = ((n + nb - 1) // nb + 1023) // 1024 * 1024
        Tr = (reach + 1023) // 1024
        if plane_walk and g == 512 and T == 0 and reach > 0 and reach % 64 == 0 and 32 <= Tr <= 512:
            per, nb = Tr, 64 // ((Tr + 7) // 8)
            nplanes = (n + reach - 1) // reach
            return (nplanes + nb - 1) // nb * reach, per, reach
        return band, per, 0
    return 0, 0, 0


def row_map(n, reach, plane_walk=True):
    """(band, per) of gcr_dev.h:make_row_map for a solve on n rows whose operator's rows reach `reach` rows away
    (0: unknown): banded when 2 * reach >= rows one XCD covers per trip of the plain grid-stride.  Full grids (512 workgroups) whose
    reach is a multiple of 1024 rows — the plane of a 3-D grid — get the plane-walk form: per = reach / 1024 workgroups tile one plane,
    floor(64 / ceil(per / 8)) bands (per = 64: the 8-band map); a reach that is a multiple of 64 only, the ragged form (`row_map_plane`)."""
    return _row_map3(n, reach, plane_walk)[:2]


def row_map_plane(n, reach, plane_walk=True):
    """RowMap::plane: != 0 when the map is the ragged plane walk (planes of `reach` rows, a multiple of 64 but not of 1024: a band's
    ceil(reach / 1024) workgroups tile one plane, the last tile short, and step by the plane)."""
    return _row_map3(n, reach, plane_walk)[2]


def gcr_x_sensitivity(A, param, rhs, x0=None):
    """Same question for the solution: max over the summation orders of max_i |x_order[i] - x_ref[i]|
    after the same number of steps (solves that stop at different steps are not comparable: inf)."""
    x_ref, _, it0, _ = gcr_solve(A, param, rhs, x0)
    dev = 0.0
    try:
        for mode in (1, 2):
            lib().orc_set_sum_order(mode)
            x, _, it, _ = gcr_solve(A, param, rhs, x0)
            dev = np.inf if it != it0 else max(dev, float(np.abs(x - x_ref).max()))
    finally:
        lib().orc_set_sum_order(0)
    return x_ref, dev


def gcr_op(A, param, x0_mode=1):
    """GCR used as an Operator (preconditioner / smoother), src/GCR.h:62-68."""
    return Op(lib().orc_op_gcr(A.h if A is not None else None, C.byref(param), x0_mode), keep=(A, param))


# ---------------------------------------------------------------- data
def read_text_csr(path):
    """read_data (src/Parse.cpp:64-90): returns (nrow, ncol, rowptr[int64], col[int64], val)."""
    nrow, ncol, nnz = C.c_int64(), C.c_int64(), C.c_int64()
    rc = lib().orc_read_text_csr(path.encode(), C.byref(nrow), C.byref(ncol), C.byref(nnz), None, None, None)
    if rc != 0:
        raise IOError("File read is unsuccessful! (%s, rc=%d)" % (path, rc))
    rowptr = np.empty(nrow.value + 1, np.int64)
    col = np.empty(nnz.value, np.int64)
    val = np.empty(nnz.value, c128)
    rc = lib().orc_read_text_csr(path.encode(), C.byref(nrow), C.byref(ncol), C.byref(nnz),
                                 rowptr.ctypes.data, col.ctypes.data, val.ctypes.data)
    if rc != 0:
        raise IOError("parse error in %s (rc=%d)" % (path, rc))
    return nrow.value, ncol.value, rowptr, col, val


def fill_rhs(n, seed=0):
    out = np.empty(n, c128)
    lib().orc_fill_rhs(n, seed, out)
    return out


def rhs_grid(n, seed=0):
    """Pure-numpy twin of orc_fill_rhs / ref_harness fill_rhs (splitmix64 on the 0.001 grid)."""
    def sm(x):
        x = (x + np.uint64(0x9E3779B97F4A7C15))
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))
    with np.errstate(over="ignore"):
        base = np.uint64(seed) * np.uint64(0x100000001B3)
        i = np.arange(n, dtype=np.uint64)
        a = sm(base + np.uint64(2) * i)
        b = sm(base + np.uint64(2) * i + np.uint64(1))
    re = (a % np.uint64(2000)).astype(np.float64) / 1000. - 1.
    im = (b % np.uint64(2000)).astype(np.float64) / 1000. - 1.
    return (re + 1j * im).astype(c128)


def poisson3d(n):
    N = n * n * n
    nnz = 7 * N - 6 * n * n
    rowptr = np.empty(N + 1, np.int64)
    col = np.empty(nnz, np.int64)
    val = np.empty(nnz, c128)
    lib().orc_poisson3d(n, rowptr, col, val)
    return N, rowptr, col, val


# ---------------------------------------------------------------- multigrid pieces
def mg_aggregates(dims, blocked, sub):
    """Mesh::blocking (src/Mesh.h:236-298): aggregate index of every unknown. Returns (agg, nagg)."""
    dims = np.ascontiguousarray(dims, np.int64)
    blocked = np.ascontiguousarray(blocked, np.int32)
    agg = np.empty(int(np.prod(dims)), np.int32)
    nagg = lib().orc_mg_aggregates(dims.size, dims, blocked, sub, agg)
    if nagg < 0:
        raise ValueError("Dimension not exactly divisible by block size!")
    return agg, int(nagg)


def mg_prolongator(agg, nagg, vecs):
    """Block-local restriction + per-aggregate Gram-Schmidt (src/MG.h:171-198). vecs: [ne][n]."""
    vecs = _c(vecs)
    ne, n = vecs.shape
    pv = np.empty((n, ne), c128)
    lib().orc_mg_prolongator(n, ne, nagg, np.ascontiguousarray(agg, np.int32), vecs, pv)
    return pv


def mg_restrict(agg, nagg, pv, x):
    n, ne = pv.shape
    xc = np.empty(nagg * ne, c128)
    lib().orc_mg_restrict(n, ne, nagg, np.ascontiguousarray(agg, np.int32), _c(pv), _c(x), xc)
    return xc


def mg_expand(agg, pv, xc):
    n, ne = pv.shape
    x = np.empty(n, c128)
    lib().orc_mg_expand(n, ne, np.ascontiguousarray(agg, np.int32), _c(pv), _c(xc), x)
    return x


def mg_galerkin(rowptr, col, val, agg, nagg, pv, shift=None):
    """Galerkin blocks P^H A P (src/MG.h:204-281). Returns (rows, cols, blocks[nblk][ne][ne])."""
    n, ne = pv.shape
    rowptr = np.ascontiguousarray(rowptr, np.int64)
    col = np.ascontiguousarray(col, np.int64)
    val = _c(val)
    agg = np.ascontiguousarray(agg, np.int32)
    pv = _c(pv)
    k = complex(shift) if shift is not None else 0j
    hs = int(shift is not None)
    nblk = lib().orc_mg_galerkin(n, rowptr, col, val, hs, k.real, k.imag, ne, nagg, agg, pv, None, None, None)
    rows, cols = np.empty(nblk, np.int32), np.empty(nblk, np.int32)
    blocks = np.empty((nblk, ne, ne), c128)
    lib().orc_mg_galerkin(n, rowptr, col, val, hs, k.real, k.imag, ne, nagg, agg, pv,
                          rows.ctypes.data, cols.ctypes.data, blocks.ctypes.data)
    return rows, cols, blocks


class MG(Op):
    """Corrected multigrid cycle as an operator (see oracle/mgcr_oracle_mg.c)."""

    def __init__(self, A, rowptr, col, val, dims, blocked, sub, vecs, nlev, smoother, coarse, damping=1.0, shift=None, vectors_are_prolongator=False):
        """vectors_are_prolongator: `vecs` are level 0's prolongator columns as they stand (orthonormal per aggregate already): not
        orthonormalised again."""
        rowptr = np.ascontiguousarray(rowptr, np.int64)
        col = np.ascontiguousarray(col, np.int64)
        val = _c(val)
        dims = np.ascontiguousarray(dims, np.int64)
        blocked = np.ascontiguousarray(blocked, np.int32)
        vecs = _c(vecs)
        ne, n = vecs.shape
        k = complex(shift) if shift is not None else 0j
        if vectors_are_prolongator:
            lib().orc_mg_set_vectors_are_prolongator(1)
        self.mg = lib().orc_mg_create(A.h, n, rowptr, col, val, int(shift is not None), k.real, k.imag, dims.size, dims,
                                      blocked, sub, ne, vecs, nlev, C.byref(smoother), C.byref(coarse), damping)
        if not self.mg:
            raise ValueError("orc_mg_create failed")
        self.nlev = nlev
        super().__init__(lib().orc_op_mg(self.mg, n), keep=(A, rowptr, col, val, smoother, coarse))

    def level_dim(self, l):
        return lib().orc_mg_level_dim(self.mg, l)

    def level_op(self, l):
        return Op(lib().orc_mg_level_op(self.mg, l), keep=(self,))

    def prolongator(self, l):
        """(pv[n][ne], agg[n]) of level l -> l+1."""
        n, ne = self.level_dim(l), lib().orc_mg_level_ne(self.mg, l)
        pv = np.ctypeslib.as_array(lib().orc_mg_level_pv(self.mg, l), shape=(n * ne * 2,)).copy().view(c128).reshape(n, ne)
        agg = np.ctypeslib.as_array(lib().orc_mg_level_agg(self.mg, l), shape=(n,)).copy()
        return pv, agg

    def restrict(self, l, x):
        xc = np.empty(self.level_dim(l + 1), c128)
        lib().orc_mg_level_restrict(self.mg, l, _c(x), xc)
        return xc

    def expand(self, l, xc):
        x = np.empty(self.level_dim(l), c128)
        lib().orc_mg_level_expand(self.mg, l, _c(xc), x)
        return x
