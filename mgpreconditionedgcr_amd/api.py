"""Host-side mirror of the reference's Operator / Field / SolverParam interface over the C ABI.

Same names, argument meaning and error behaviour as the reference's C++ classes (paths relative
to the reference root); the data lives in HBM and every operation is a HIP kernel of
libmgcr_hip.so.  This Python layer is what tests/ and bench.py drive; the C++ twin of the same
interface is include/mgcr/*.h.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import GcrParamC, MgParamC, MgcrError, check

c128 = np.complex128


def _ri(z):
    z = complex(z)
    return (C.c_double * 2)(z.real, z.imag)


def set_option(name, value):
    """mgcr_set_option: switch between equivalent code paths ("pattern_storage", "lean_cycles"); returns the old value."""
    prev = C.c_int()
    check(_lib.lib().mgcr_set_option(name.encode(), int(value), C.byref(prev)))
    return prev.value


def stat(name):
    """mgcr_stat: a counter of the library ("resident_solves")."""
    v = C.c_int64()
    check(_lib.lib().mgcr_stat(name.encode(), C.byref(v)))
    return v.value


class Field:
    """Field<num_type> (src/Fields.h:29-71): complex-fp64 vector tagged with mesh dimensions."""

    def __init__(self, dims, data=None):
        _lib.init()
        if isinstance(dims, (int, np.integer)):
            dims = (int(dims),)
        self.dims = tuple(int(d) for d in dims)
        n = int(np.prod(self.dims, dtype=np.int64))
        h = C.c_void_p()
        check(_lib.lib().mgcr_vec_create(n, C.byref(h)))
        self.h = h
        self._n = n
        if data is not None:
            self.upload(data)

    # -- construction / queries
    @classmethod
    def like(cls, other):
        return cls(other.dims)

    def field_size(self):  # src/Fields.h:120-123
        return self._n

    def get_ndim(self):
        return len(self.dims)

    def upload(self, data):
        a = np.ascontiguousarray(data, dtype=c128).reshape(-1)
        if a.size != self._n:
            raise MgcrError(1, "Dimension mismatch.")
        check(_lib.lib().mgcr_vec_upload(self.h, a.ctypes.data))
        return self

    def to_numpy(self):
        out = np.empty(self._n, c128)
        check(_lib.lib().mgcr_vec_download(self.h, out.ctypes.data))
        return out

    def val_at(self, location):  # src/Fields.h:163-168 (host convenience, one element)
        if not 0 <= location < self._n:
            raise MgcrError(1, "Field memory access out of bound!")
        return self.to_numpy()[location]

    def set_zero(self):  # src/Fields.h:137-145
        check(_lib.lib().mgcr_vec_zero(self.h))
        return self

    def set_constant(self, c):  # src/Fields.h:146-151
        check(_lib.lib().mgcr_vec_set_constant(self.h, _ri(c)))
        return self

    def fill_rhs(self, seed=0, global_offset=0):
        """Deterministic repo-owned stand-in for init_rand (src/Fields.h:125-135)."""
        check(_lib.lib().mgcr_vec_fill_rhs(self.h, seed, global_offset))
        return self

    def assign(self, other):  # operator= src/Fields.h:256-286
        check(_lib.lib().mgcr_vec_copy(self.h, other.h))
        return self

    def copy(self):
        return Field(self.dims).assign(self)

    # -- algebra (value semantics like the reference; the fused solver kernels do not use these)
    def dot(self, other):  # src/Fields.h:216-226
        out = (C.c_double * 2)()
        check(_lib.lib().mgcr_dot(self.h, other.h, out))
        return complex(out[0], out[1])

    def squarednorm(self):  # src/Fields.h:228-235
        out = C.c_double()
        check(_lib.lib().mgcr_norm2(self.h, C.byref(out)))
        return out.value

    def norm(self):
        return float(np.sqrt(self.squarednorm()))

    def add_scaled(self, alpha, other):
        """self + other*alpha as a new Field (operator+ with operator*, src/Fields.h:192-253)."""
        out = Field(self.dims)
        check(_lib.lib().mgcr_add_scaled(out.h, self.h, _ri(alpha), other.h))
        return out

    def __add__(self, other):
        return self.add_scaled(1.0, other)

    def __sub__(self, other):
        return self.add_scaled(-1.0, other)

    def __mul__(self, a):  # src/Fields.h:245-253
        out = self.copy()
        check(_lib.lib().mgcr_scale(out.h, _ri(a)))
        return out

    def __iadd__(self, other):  # src/Fields.h:288-297
        check(_lib.lib().mgcr_axpy(_ri(1.0), other.h, self.h))
        return self

    def __isub__(self, other):
        check(_lib.lib().mgcr_axpy(_ri(-1.0), other.h, self.h))
        return self

    def normalise(self):  # src/Fields.h:237-243
        check(_lib.lib().mgcr_normalise(self.h))
        return self

    def gamma5(self, spinor_index=4):  # src/Fields.h:310-339
        """New Field with the spinor components swapped 0<->2, 1<->3 (on the device)."""
        if self.dims[spinor_index] != 4:
            raise ValueError("gamma5: dimension %d of the mesh has %d entries, not 4" % (spinor_index, self.dims[spinor_index]))
        inner = int(np.prod(self.dims[spinor_index + 1:], dtype=np.int64))
        out = Field(self.dims)
        check(_lib.lib().mgcr_vec_gamma5(self.h, out.h, inner))
        return out

    def __del__(self):
        try:
            if getattr(self, "h", None):
                _lib.lib().mgcr_vec_destroy(self.h)
                self.h = None
        except Exception:
            pass


class Operator:
    """Operator<num_type> (src/Operator.h:16-29)."""

    def __init__(self):
        self.h = None
        self._keep = []

    def get_dim(self):
        return int(_lib.lib().mgcr_op_dim(self.h))

    def get_nrow(self):
        return int(_lib.lib().mgcr_op_nrow(self.h))

    def __call__(self, f, out=None):
        """Field operator()(const Field&): applies the operator, returns a new Field."""
        if out is None:
            nrow = self.get_nrow()
            out = Field(f.dims if nrow == f.field_size() else (nrow,))
        check(_lib.lib().mgcr_op_apply(self.h, f.h, out.h))
        return out

    def stored_bytes(self):
        b, w, t = C.c_int64(), C.c_int32(), C.c_int64()
        check(_lib.lib().mgcr_op_stored_bytes(self.h, C.byref(b), C.byref(w), C.byref(t)))
        return dict(matrix_bytes=b.value, ell_width=w.value, tail_nnz=t.value)

    def ell_layout(self):
        """dict(ell_width, lanes, tail_rows, reach, tail_chunk_cap): how a Sparse's rows are dealt to threads (include/mgcr.h)."""
        w, l, t, r, cap, win = C.c_int32(), C.c_int32(), C.c_int64(), C.c_int64(), C.c_int32(), C.c_int32()
        check(_lib.lib().mgcr_op_ell_layout(self.h, C.byref(w), C.byref(l), C.byref(t), C.byref(r), C.byref(cap), C.byref(win)))
        return dict(ell_width=w.value, lanes=l.value, tail_rows=t.value, reach=r.value, tail_chunk_cap=cap.value, x_window=win.value)

    def xr_fuse_kind(self):
        """0 / 1 / 2: where a lean GCR on this operator runs its residual update (include/mgcr.h mgcr_op_xr_fuse_kind)."""
        k = C.c_int32()
        check(_lib.lib().mgcr_op_xr_fuse_kind(self.h, C.byref(k)))
        return k.value

    def storage_format(self):
        """(format, n_patterns): 0 ELL slab, 1 row-pattern dictionary (columns + values), 2 (columns only)."""
        f, n = C.c_int32(), C.c_int32()
        check(_lib.lib().mgcr_op_storage_format(self.h, C.byref(f), C.byref(n)))
        return f.value, n.value

    def bench_apply(self, x, y, reps=20):
        ms = C.c_double()
        check(_lib.lib().mgcr_bench_op_apply(self.h, x.h, y.h, reps, C.byref(ms)))
        return ms.value

    def __del__(self):
        try:
            if getattr(self, "h", None):
                _lib.lib().mgcr_op_destroy(self.h)
                self.h = None
        except Exception:
            pass


class Sparse(Operator):
    """Sparse<long> (src/Operator.h:56-101): CSR with int64 indices on the host side.  The host arrays are kept (the
    reference's get_ROW / get_COL / val_at accessors and its set-up-time algebra work on them); the device copy is what
    operator() uses."""

    def __init__(self, rows, cols, rowptr, col, val):
        super().__init__()
        _lib.init()
        rowptr = np.ascontiguousarray(rowptr, np.int64)
        col = np.ascontiguousarray(col, np.int64)
        val = np.ascontiguousarray(val, c128)
        if rowptr.size != rows + 1 or col.size != rowptr[-1] or val.size != col.size:
            raise MgcrError(1, "CSR arrays have inconsistent sizes")
        h = C.c_void_p()
        check(_lib.lib().mgcr_csr_create(rows, cols, rowptr.ctypes.data, col.ctypes.data, val.ctypes.data, C.byref(h)))
        self.h = h
        self._nnz = int(rowptr[-1])
        self._shape = (int(rows), int(cols))
        self.ROW, self.COL, self.VAL = rowptr, col, val

    @classmethod
    def from_triplets(cls, rows, cols, trip_rows, trip_cols, trip_vals):
        """Sparse(rows, cols, triplets, n) (src/Operator.h:250-294): unordered triplets, duplicates summed."""
        from . import hostalg
        return cls(rows, cols, *hostalg.csr_from_triplets(rows, cols, trip_rows, trip_cols, trip_vals))

    def get_nnz(self):  # src/Operator.h:73
        return self._nnz

    def get_ROW(self, l):  # src/Operator.h:78-79
        return int(self.ROW[l])

    def get_COL(self, l):
        return int(self.COL[l])

    def val_at(self, row, col=None):  # src/Operator.h:392-402
        if col is None:
            return complex(self.VAL[row])
        for l in range(self.ROW[row], self.ROW[row + 1]):
            if self.COL[l] == col:
                return complex(self.VAL[l])
        return 0j

    def dagger(self):
        """void Sparse::dagger() (src/Operator.h:296-328): conjugate transpose IN PLACE (rows and columns swap)."""
        from . import hostalg
        nr, nc, rp, ci, va = hostalg.csr_dagger(self._shape[0], self._shape[1], self.ROW, self.COL, self.VAL)
        rp, ci, va = np.ascontiguousarray(rp, np.int64), np.ascontiguousarray(ci, np.int64), np.ascontiguousarray(va, c128)
        # the matrix changes BEHIND the handle (mgcr_csr_replace): a DiracOp / GCR that borrowed this Sparse — the
        # reference's DiracOp keeps a Sparse* (src/Operator.h:117) — applies the daggered matrix from now on
        check(_lib.lib().mgcr_csr_replace(self.h, nr, nc, rp.ctypes.data, ci.ctypes.data, va.ctypes.data))
        self._shape, self.ROW, self.COL, self.VAL = (int(nr), int(nc)), rp, ci, va
        self._nnz = int(rp[-1])
        return self

    def __mul__(self, a):
        """Sparse operator*(complex) (src/Operator.h:535-544): a new Sparse with every value scaled."""
        from . import hostalg
        return Sparse(self._shape[0], self._shape[1], self.ROW, self.COL, hostalg.csr_scale(self.VAL, a))


class DiracOp(Operator):
    """DiracOp = Id - k*D (src/Operator.h:104-122,555-575); borrows the Sparse."""

    def __init__(self, mat, k_factor):
        super().__init__()
        h = C.c_void_p()
        check(_lib.lib().mgcr_dirac_create(mat.h, _ri(k_factor), C.byref(h)))
        self.h = h
        self._keep.append(mat)
        self.comm, self.row0 = getattr(mat, "comm", None), getattr(mat, "row0", 0)

    def set_k(self, new_k):  # src/Operator.h:116
        check(_lib.lib().mgcr_dirac_set_k(self.h, _ri(new_k)))


class HierarchicalSparse(Operator):
    """HierarchicalSparse<long,int> (src/HierarchicalSparse.h:22-48) from unsorted
    (block_row, block_col, dense block) triplets, duplicates kept (ctor :58-98)."""

    def __init__(self, block_rows, block_cols, rows, cols, blocks):
        super().__init__()
        _lib.init()
        rows = np.ascontiguousarray(rows, np.int32)
        cols = np.ascontiguousarray(cols, np.int32)
        blocks = np.ascontiguousarray(blocks, c128)
        nt = rows.size
        bs = int(round(np.sqrt(blocks.size // max(nt, 1))))
        if nt == 0 or bs * bs * nt != blocks.size:
            raise MgcrError(1, "blocks must hold ntriplets square blocks")
        h = C.c_void_p()
        check(_lib.lib().mgcr_bcsr_create_from_triplets(block_rows, block_cols, bs, nt, rows.ctypes.data,
                                                        cols.ctypes.data, blocks.ctypes.data, C.byref(h)))
        self.h = h
        self.bs = bs


class Dense(HierarchicalSparse):
    """Dense<num_type>(matrix, dim) (src/Operator.h:125-129,159-173): row-major dim x dim complex matrix; its
    operator() accumulates each row in column order.  Stored as a block-CSR operator of one block."""

    def __init__(self, matrix, dim=None):
        m = np.ascontiguousarray(matrix, c128)
        dim = int(dim) if dim is not None else int(round(np.sqrt(m.size)))
        if dim * dim != m.size:
            raise MgcrError(1, "Dense: matrix must hold dim*dim entries")
        super().__init__(1, 1, [0], [0], m.reshape(1, dim, dim))
        self.mat = m.reshape(dim, dim).copy()

    def val_at(self, row, col=None):  # src/Operator.h:45-46
        return complex(self.mat.reshape(-1)[row] if col is None else self.mat[row, col])

    # set-up-time algebra (src/Operator.h:139-190), on the host like the reference's
    def __add__(self, other):
        from . import hostalg
        return Dense(hostalg.dense_add(self.mat, other.mat))

    def __mul__(self, other):
        from . import hostalg
        return Dense(hostalg.dense_mul(self.mat, other.mat))

    def dagger(self):
        from . import hostalg
        return Dense(hostalg.dense_dagger(self.mat))


class GCR_Param:
    """GCR_Param(trunc, re, max_it, tau, verb, solver_l, solver_r) (src/SolverParam.h:21-35,85-99)."""

    def __init__(self, trunc=0, re=0, max_it=100, tau=1e-16, verb=True, solver_l=None, solver_r=None,
                 use_x0=False, flexible=False, check_every=0, profile_spmv=False):
        self.truncation, self.restart, self.max_iter, self.tol = int(trunc), int(re), int(max_it), float(tau)
        self.verbose = bool(verb)
        self.left_precond, self.right_precond = solver_l, solver_r
        self.use_x0, self.flexible, self.check_every = bool(use_x0), bool(flexible), int(check_every)
        self.profile_spmv = bool(profile_spmv)

    def _c(self):
        return GcrParamC(self.truncation, self.restart, self.max_iter, self.tol, int(self.verbose),
                         self.left_precond.h if self.left_precond is not None else None,
                         self.right_precond.h if self.right_precond is not None else None,
                         int(self.use_x0), int(self.flexible), self.check_every, int(self.profile_spmv))


class GCR(Operator):
    """GCR<num_type> (src/GCR.h:18-50): `GCR(M, param)` or `GCR(param)` + `initialise(M)`."""

    def __init__(self, M=None, gcr_param=None, x0_mode=1):
        super().__init__()
        if isinstance(M, GCR_Param) and gcr_param is None:  # GCR(GCR_Param*) src/GCR.h:30
            M, gcr_param = None, M
        _lib.init()
        self.param = gcr_param
        self.A = M
        h = C.c_void_p()
        pc = gcr_param._c()
        check(_lib.lib().mgcr_gcr_create(M.h if M is not None else None, C.byref(pc), x0_mode, C.byref(h)))
        self.h = h
        self._keep += [M, gcr_param, gcr_param.left_precond, gcr_param.right_precond]
        self.last_history = None
        self.last_iterations = None
        self.last_converged = None

    def initialise(self, M):  # src/GCR.h:31
        self.A = M
        self._keep.append(M)
        check(_lib.lib().mgcr_gcr_set_operator(self.h, M.h))

    def set_x0(self, x0):
        check(_lib.lib().mgcr_gcr_set_x0(self.h, x0.h if x0 is not None else None))

    def solve(self, rhs, x):
        """void solve(const Field& rhs, Field& x) (src/GCR.h:158-302). x is updated in place;
        the residual history / iteration count are kept in last_history / last_iterations."""
        cap = max(self.param.max_iter, 1) + 1
        hist = np.zeros(cap, np.float64)
        it, conv = C.c_int32(), C.c_int32()
        pc = self.param._c()  # the reference reads its GCR_Param* at solve time: refresh, keep the work vectors
        check(_lib.lib().mgcr_gcr_set_param(self.h, C.byref(pc)))
        check(_lib.lib().mgcr_gcr_solve_op(self.h, rhs.h, x.h, hist.ctypes.data, cap, C.byref(it), C.byref(conv)))
        self.last_iterations = it.value
        self.last_converged = bool(conv.value)
        self.last_history = hist[: it.value + 1].copy()
        return x


def legacy_dense_gcr(matrix, rhs, x, tol, max_iter, truncation, verbose=True):
    """GCR(matrix, dimension).solve(rhs, x, tol, max_iter, truncation) — the reference's legacy raw-pointer dense solve
    (src/GCR.h:70-156): r0 = rhs - A x (x0 honoured), directions truncated to the last `truncation`, stops when
    |r|^2 <= tol — absolute, tested BEFORE every step, so possibly after none — or after max_iter steps.  On the device:
    the Dense operator, GCR in truncation mode with use_x0; sqrt(tol) / |rhs| as relative tolerance is the same test.
    Returns (x, absolute residual norms per step) and prints the reference's lines when verbose."""
    A = Dense(matrix)
    d = A.mat.shape[0]
    b = Field((d,), rhs)
    xf = Field((d,), x)
    r0 = b - A(xf)
    bn2, rr = b.squarednorm(), r0.squarednorm()
    norms = np.zeros(0)
    if rr > tol and max_iter > 0 and bn2 > 0.:
        g = GCR(A, GCR_Param(int(truncation), 0, int(max_iter), float(np.sqrt(tol) / np.sqrt(bn2)), False, use_x0=True))
        g.solve(b, xf)
        norms = g.last_history[1:] * np.sqrt(bn2)
        rr = float(norms[-1] ** 2)
    elif rr > tol and max_iter > 0:
        # rhs = 0 with x0 != 0: the reference iterates on r0 = -A x0 and drives x towards 0 (its test is absolute).  The Field
        # solve's test is relative to its right-hand side, so hand it r0 AS the right-hand side (r = rhs there whatever x is,
        # src/GCR.h:189) and let it update x0 in place: the same recurrence, tolerance sqrt(tol) / |r0|
        g = GCR(A, GCR_Param(int(truncation), 0, int(max_iter), float(np.sqrt(tol) / np.sqrt(rr)), False))
        rn = np.sqrt(rr)
        g.solve(r0, xf)
        norms = g.last_history[1:] * rn
        rr = float(norms[-1] ** 2)
    if verbose:
        for k, v in enumerate(norms):
            print("Step %d residual norm = %.10e" % (k + 1, v))
        if norms.size == max_iter:
            print("GCR did not converge after %d steps! Residual norm = %.10e" % (max_iter, rr))
    return xf.to_numpy(), norms


class Mesh:
    """Mesh<num_type> (src/Mesh.h:13-64): row-major N-D index algebra (host side)."""

    def __init__(self, index_dims, num_dims=None):
        self.dims = tuple(int(d) for d in (index_dims[:num_dims] if num_dims else index_dims))

    def get_ndim(self):
        return len(self.dims)

    def get_size(self):
        return int(np.prod(self.dims, dtype=np.int64))

    def ind_loc(self, index):  # src/Mesh.h:156-165
        loc = int(index[0])
        for i in range(1, len(self.dims)):
            loc = loc * self.dims[i] + int(index[i])
        return loc

    def loc_ind(self, loc):  # alloc_loc_ind src/Mesh.h:368-398
        return tuple(int(v) for v in np.unravel_index(loc, self.dims))

    # ---- aggregates (src/Mesh.h:236-324): the reference's 4 blocked ("spacetime") dimensions --------------
    def blocking(self, subblock_dim, blocked_dimensions):
        """Mesh::blocking (src/Mesh.h:236-298): block_map[block][offset] = location of a site in the row-major
        spacetime lattice (the blocked dimensions only), blocks and in-block offsets row-major as well.  Integer
        index algebra on the host, as in the reference; the device set-up (csrc/mg_setup.hip:agg_kernel) forms
        the same aggregates directly from the row index (tests/test_gpu_mg.py::test_mesh_blocking_matches_device_aggregates)."""
        blocked = [bool(b) for b in blocked_dimensions][:len(self.dims)]
        idx = [d for d, b in enumerate(blocked) if b]
        if len(idx) != 4:
            raise ValueError("Mesh::blocking blocks exactly 4 dimensions (got %d)" % len(idx))
        sub = int(subblock_dim)
        st_dims = [self.dims[d] for d in idx]
        if any(v % sub for v in st_dims):
            raise ValueError("Dimension not exactly divisible by block size!")
        self.sub_dim = sub
        self.blocked_ind = idx
        self.block_dim = [v // sub for v in st_dims]
        coords = np.indices(st_dims).reshape(4, -1)                      # x, y, z, w of every site, row-major
        loc = np.ravel_multi_index(tuple(coords), st_dims)
        block = np.ravel_multi_index(tuple(coords // sub), self.block_dim)
        off = np.ravel_multi_index(tuple(coords % sub), [sub] * 4)
        self.block_map = np.empty((self.get_nblocks(), self.get_block_size()), np.int64)
        self.block_map[block, off] = loc
        return self

    def get_nblocks(self):
        return int(np.prod(self.block_dim))

    def get_block_dim(self):
        return list(self.block_dim)

    def get_block_size(self):
        return self.sub_dim ** 4

    def get_block_map(self, block_idx):
        return self.block_map[int(block_idx)]

    def alloc_full_index(self, spacetime_loc, spinor, colour, spacetime_dimensions, spinor_dimension):
        """Mesh::alloc_full_index (src/Mesh.h:300-324): spacetime location + spinor + colour -> N-D index."""
        st_dims = [self.dims[d] for d in range(len(self.dims)) if spacetime_dimensions[d]]
        st = np.unravel_index(int(spacetime_loc), st_dims)
        out, c = [], 0
        for d in range(len(self.dims)):
            if spacetime_dimensions[d]:
                out.append(int(st[c]))
                c += 1
            elif spinor_dimension[d]:
                out.append(int(spinor))
            else:
                out.append(int(colour))
        return tuple(out)


def gamma5(field_values, dims, spinor_index=4):
    """Field::gamma5 (src/Fields.h:310-339) of host values: output[index with spinor 0<->2, 1<->3] = field[i]
    (computed on the device: Field.gamma5)."""
    return Field(tuple(dims), np.asarray(field_values, c128).reshape(-1)).gamma5(spinor_index).to_numpy()


def _vec_double_fields(fields, spinor_index):
    plus, minus = [], []
    for v in fields:
        g = v.gamma5(spinor_index)
        plus.append((v + g) * 0.5)
        minus.append((v - g) * 0.5)
    return plus + minus


def vec_double(vecs, dims, spinor_index=4):
    """MG::vec_double (src/MG.h:316-345): [v+ ..., v- ...] with v+- = (v +- gamma5 v) * 0.5, on the device."""
    fields = [Field(tuple(dims), np.asarray(v, c128).reshape(-1)) for v in vecs]
    return np.array([f.to_numpy() for f in _vec_double_fields(fields, spinor_index)])


class MG_Param:
    """MG_Param(mesh, subblock, eigenvecs, eigen_param, solver_coarse, solver_smooth, levels,
    solver_l, solver_r) (src/SolverParam.h:38-59,142-158).

    solver_coarse / solver_smooth are GCR objects created with `GCR(GCR_Param)` exactly as in the
    reference (src/main.cpp:851-853); MG takes their GCR_Param.  Extras the reference lacks:
    `null_vectors` (use these near-null vectors instead of computing n_eigen of them),
    `spacetime` mask for meshes that are not 6-D, `damping` (reference literal: 0.1), `coarse_direct` (solve a
    coarsest level of at most that many unknowns directly instead of by solver_coarse; 0 = never, like the reference)."""

    def __init__(self, m, subblock, eigenvecs, eigen_param, solver_coarse, solver_smooth, levels=1,
                 solver_l=None, solver_r=None, spacetime=None, spinor=None, null_vectors=None, damping=1.0, coarse_direct=0):
        self.mesh = m if isinstance(m, Mesh) else Mesh(m)
        self.subblock_dim = int(subblock)
        self.n_eigen = int(eigenvecs)
        self.eigenvector_precomp_param = eigen_param
        self.coarse_solver, self.smoother_solver = solver_coarse, solver_smooth
        nd = self.mesh.get_ndim()
        default_st = [True, True, True, True, False, False]
        self.spacetime = list(spacetime) if spacetime is not None else (default_st[:nd] if nd == 6 else [True] * nd)
        self.spinor = list(spinor) if spinor is not None else ([False] * 4 + [True, False] if nd == 6 else [False] * nd)
        self.n_level = int(levels)
        self.left_precond, self.right_precond = solver_l, solver_r
        self.null_vectors = null_vectors
        self.damping = float(damping)
        self.coarse_direct = int(coarse_direct)   # > 0: a coarsest level of at most this many unknowns (<= 2048) is inverted at set-up


class MG(Operator):
    """MG<num_type> (src/MG.h:20-61): `MG(M, param)` or `MG(param)` + `initialise(M)`.

    apply = corrected V-cycle; restrict / expand as in src/MG.h:347-383."""

    def __init__(self, M=None, parameter=None):
        super().__init__()
        if isinstance(M, MG_Param) and parameter is None:
            M, parameter = None, M
        self.param = parameter
        self.m = None
        if M is not None:
            self.initialise(M)

    def near_null_vectors(self, M, start=None, alias_rhs_x=True, double=True):
        """Arnoldi::solve (src/MG.h:90-122): inverse iteration with GCR for the smallest modes,
        Gram-Schmidt between them, then chirality doubling when the mesh has a spinor dimension.
        The start vector is the repo's deterministic RHS (seed 9) instead of libc rand() unless `start` is given (the
        reference's is init_rand(9)).  `alias_rhs_x` (default, the reference's literal first loop): gcr.solve(b, b) —
        rhs and x are ONE Field, i.e. x0 = b while r0 = b (src/GCR.h:189), so each of the ten steps is
        b <- normalise(b + GCR(b)); False: each solve starts from x0 = 0, b <- normalise(GCR(b)).
        The later vectors are solved from x0 = 0 either way: the reference solves
        them into a malloc'ed, never initialised Field (src/MG.h:110, src/Fields.h:97-101), which has no defined value
        (tests/test_gpu_mg.py::test_arnoldi_vs_reference)."""
        prm = self.param
        dims = prm.mesh.dims
        gp = prm.eigenvector_precomp_param
        gcr = GCR(M, GCR_Param(gp.truncation, gp.restart, gp.max_iter, gp.tol, False))
        comm = getattr(M, "comm", None)   # distributed operator: dot products and norms are global

        def gdot(a, b_):
            return comm.dot(a, b_) if comm is not None else a.dot(b_)

        def gnormalise(f):
            return f * (1.0 / np.sqrt(gdot(f, f).real)) if comm is not None else f.normalise()

        b = Field(dims).fill_rhs(9, global_offset=getattr(M, "row0", 0)) if start is None else Field(dims, start)
        x = Field(dims)
        for _ in range(10):
            if alias_rhs_x:
                x.assign(b)
            else:
                x.set_zero()
            gcr.solve(b, x)
            b = gnormalise(b.assign(x))
        vecs = [b.copy()]
        for count in range(1, prm.n_eigen):
            x.set_zero()
            gcr.solve(vecs[-1], x)
            t = x.copy()
            for v in vecs:       # Gram-Schmidt against the vectors found so far (src/MG.h:112-118), Field algebra on the device
                t = t - v * gdot(v, t)
            vecs.append(gnormalise(t))
        if double and any(prm.spinor):
            vecs = _vec_double_fields(vecs, prm.spinor.index(True))
        return np.array([v.to_numpy() for v in vecs])

    def initialise(self, M):  # src/MG.h:131-285
        _lib.init()
        prm = self.param
        self.m = M
        vecs = prm.null_vectors if prm.null_vectors is not None else self.near_null_vectors(M)
        vecs = np.ascontiguousarray(vecs, c128)
        if vecs.ndim == 1:
            vecs = vecs[None, :]
        nd = prm.mesh.get_ndim()
        pc = MgParamC()
        pc.ndim = nd
        for d in range(nd):
            pc.dims[d] = prm.mesh.dims[d]
            pc.blocked[d] = int(bool(prm.spacetime[d]))
        pc.subblock_dim = prm.subblock_dim
        pc.n_vec = vecs.shape[0]
        pc.vecs_ri = vecs.ctypes.data
        pc.n_level = prm.n_level
        pc.smoother = prm.smoother_solver.param._c()
        pc.coarse = prm.coarse_solver.param._c()
        pc.damping = prm.damping
        pc.coarse_direct_rows = prm.coarse_direct
        h = C.c_void_p()
        check(_lib.lib().mgcr_mg_create(M.h, C.byref(pc), C.byref(h)))
        if self.h:
            _lib.lib().mgcr_op_destroy(self.h)
        self.h = h
        self._keep += [M, prm]
        self.n_levels = prm.n_level + 1

    def level_info(self, level):
        dim, ne, nagg = C.c_int64(), C.c_int32(), C.c_int64()
        check(_lib.lib().mgcr_mg_level_info(self.h, level, C.byref(dim), C.byref(ne), C.byref(nagg)))
        return dict(dim=dim.value, ne=ne.value, nagg=nagg.value)

    def restrict(self, x_fine, level=0):  # src/MG.h:366-383
        out = Field((self.level_info(level + 1)["dim"],))
        check(_lib.lib().mgcr_mg_restrict(self.h, level, x_fine.h, out.h))
        return out

    def expand(self, x_coarse, level=0):  # src/MG.h:347-364
        info = self.level_info(level)
        out = Field(self.param.mesh.dims if level == 0 else (info["dim"],))
        check(_lib.lib().mgcr_mg_expand(self.h, level, x_coarse.h, out.h))
        return out

    def level_operator(self, level):
        """Borrowed view of the level's operator (level >= 1: the Galerkin m_coarse, src/MG.h:281)."""
        h = C.c_void_p()
        check(_lib.lib().mgcr_mg_level_op(self.h, level, C.byref(h)))
        op = Operator()
        op.h = h
        op._keep.append(self)
        op.__class__ = _BorrowedOperator
        return op

    def prolongator(self, level=0):
        info = self.level_info(level)
        pv = np.empty((info["dim"], info["ne"]), c128)
        agg = np.empty(info["dim"], np.int32)
        check(_lib.lib().mgcr_mg_download_prolongator(self.h, level, pv.ctypes.data, agg.ctypes.data))
        return pv, agg


class _BorrowedOperator(Operator):
    def __del__(self):  # owned by the MG object
        self.h = None


def read_data(filename, directory=None):
    """Sparse<long> read_data(const std::string&) (src/Parse.cpp:64-90).

    The reference opens "../../data/sample_matrix/" + filename relative to the cwd; `directory`
    (or $MGCR_SAMPLE_DIR) overrides that prefix.  Format: SURVEY.md Appendix B."""
    prefix = directory if directory is not None else os.environ.get("MGCR_SAMPLE_DIR", "../../data/sample_matrix/")
    path = os.path.join(prefix, filename)
    try:
        f = open(path, "r")
    except OSError:
        print("File read is unsuccessful!")  # src/Parse.cpp:67-68 (the reference then reads garbage)
        raise
    print("File read is successful.")
    with f:
        row, col, nnz = (int(t) for t in f.readline().split())
        rowptr = np.empty(row + 1, np.int64)
        rowptr[:row] = np.array(f.readline().split(), dtype=np.int64)
        rowptr[row] = nnz  # ROW[rows] = nnz, src/Operator.h:61
        cols = np.empty(nnz, np.int64)
        vals = np.empty(nnz, c128)
        for i, line in enumerate(f):
            if i >= nnz:
                break
            a, b = line.split()
            cols[i] = int(a)
            re, im = b[1:-1].split(",")
            vals[i] = complex(float(re), float(im))
    return Sparse(row, col, rowptr, cols, vals)
