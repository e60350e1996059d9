"""Problem generators shared by tests and bench.py (host side, numpy).

poisson3d_csr: 3-D 7-point Poisson of SURVEY.md §8(d) config 2 — lexicographic rows
r = (i*n + j)*n + k, diagonal 6, off-diagonals -1 to the in-range neighbours (Dirichlet
truncation), columns ascending within a row; complex values with zero imaginary part.
"""
import numpy as np


def poisson3d_csr(n, i0=0, i1=None, ni=None):
    """Rows of planes [i0, i1) of the (ni x n x n) Poisson matrix (ni = n by default) as
    (nrow, ncol, rowptr, col, val).  Column indices are global (0 .. ni*n*n)."""
    if ni is None:
        ni = n
    if i1 is None:
        i1 = ni
    i, j, k = np.meshgrid(np.arange(i0, i1, dtype=np.int64), np.arange(n, dtype=np.int64),
                          np.arange(n, dtype=np.int64), indexing="ij")
    i, j, k = i.ravel(), j.ravel(), k.ravel()
    r = (i * n + j) * n + k
    # candidate entries in ascending column order
    cand = [(i > 0, r - n * n, -1.0), (j > 0, r - n, -1.0), (k > 0, r - 1, -1.0),
            (np.ones_like(r, bool), r, 6.0),
            (k < n - 1, r + 1, -1.0), (j < n - 1, r + n, -1.0), (i < ni - 1, r + n * n, -1.0)]
    mask = np.stack([c[0] for c in cand], axis=1)
    cols = np.stack([c[1] for c in cand], axis=1)
    vals = np.broadcast_to(np.array([c[2] for c in cand]), mask.shape)
    rowptr = np.zeros(r.size + 1, np.int64)
    np.cumsum(mask.sum(axis=1), out=rowptr[1:])
    return r.size, ni * n * n, rowptr, cols[mask], vals[mask].astype(np.complex128)


def poisson3d_box_csr(nz, ny, nx):
    """The same operator on an nz x ny x nx box (rows r = (i*ny + j)*nx + k): planes of ny * nx sites, lines of nx."""
    i, j, k = np.meshgrid(np.arange(nz, dtype=np.int64), np.arange(ny, dtype=np.int64), np.arange(nx, dtype=np.int64), indexing="ij")
    i, j, k = i.ravel(), j.ravel(), k.ravel()
    r = (i * ny + j) * nx + k
    cand = [(i > 0, r - ny * nx, -1.0), (j > 0, r - nx, -1.0), (k > 0, r - 1, -1.0),
            (np.ones_like(r, bool), r, 6.0),
            (k < nx - 1, r + 1, -1.0), (j < ny - 1, r + nx, -1.0), (i < nz - 1, r + ny * nx, -1.0)]
    mask = np.stack([c[0] for c in cand], axis=1)
    cols = np.stack([c[1] for c in cand], axis=1)
    vals = np.broadcast_to(np.array([c[2] for c in cand]), mask.shape)
    rowptr = np.zeros(r.size + 1, np.int64)
    np.cumsum(mask.sum(axis=1), out=rowptr[1:])
    return r.size, nz * ny * nx, rowptr, cols[mask], vals[mask].astype(np.complex128)


def rhs_grid(n, seed=0, offset=0):
    """numpy twin of mgcr_vec_fill_rhs: splitmix64 values on the 0.001 grid of init_rand."""
    def sm(x):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))
    with np.errstate(over="ignore"):
        base = np.uint64(seed) * np.uint64(0x100000001B3)
        i = np.arange(offset, offset + n, dtype=np.uint64)
        a = sm(base + np.uint64(2) * i)
        b = sm(base + np.uint64(2) * i + np.uint64(1))
    re = (a % np.uint64(2000)).astype(np.float64) / 1000. - 1.
    im = (b % np.uint64(2000)).astype(np.float64) / 1000. - 1.
    return (re + 1j * im).astype(np.complex128)


def random_csr(nrow, ncol, rng, min_len=0, max_len=12, long_rows=0, long_len=200):
    """Irregular CSR for parity tests: row lengths uniform in [min_len, max_len], plus
    `long_rows` rows of length ~long_len (exercises the CSR tail), sorted unique columns."""
    lens = rng.integers(min_len, max_len + 1, size=nrow)
    if long_rows:
        idx = rng.choice(nrow, size=long_rows, replace=False)
        lens[idx] = np.minimum(ncol, rng.integers(long_len // 2, long_len + 1, size=long_rows))
    lens = np.minimum(lens, ncol)
    rowptr = np.zeros(nrow + 1, np.int64)
    np.cumsum(lens, out=rowptr[1:])
    col = np.empty(rowptr[-1], np.int64)
    for r in range(nrow):
        col[rowptr[r]:rowptr[r + 1]] = np.sort(rng.choice(ncol, size=lens[r], replace=False))
    val = (rng.uniform(-1, 1, rowptr[-1]) + 1j * rng.uniform(-1, 1, rowptr[-1])).astype(np.complex128)
    return rowptr, col, val


def skewed_csr(nrow, rng, window=1 << 17, long_rows=0, long_len=2000):
    """An irregular scalar CSR for the ELL + CSR-tail hybrid (BASELINE configs[4] "irregular nnz/row"; the reference's
    general apply is src/Operator.h:330-346): 80 % of the rows have 5-9 entries, 20 % have 10-64, `long_rows` rows have
    ~long_len; columns are the row number plus a random offset within +-window (clamped to the matrix: graph-like
    locality; duplicates inside a row are possible and are separate entries, as CSR allows).  Vectorised: 8 M rows /
    100 M entries in seconds.  Returns (rowptr, col, val), square nrow x nrow."""
    lens = np.where(rng.random(nrow) < 0.8, rng.integers(5, 10, nrow), rng.integers(10, 65, nrow)).astype(np.int64)
    if long_rows:
        idx = rng.choice(nrow, size=long_rows, replace=False)
        lens[idx] = rng.integers(long_len // 2, long_len + 1, size=long_rows)
    rowptr = np.zeros(nrow + 1, np.int64)
    np.cumsum(lens, out=rowptr[1:])
    nnz = int(rowptr[-1])
    col = np.repeat(np.arange(nrow, dtype=np.int64), lens)
    col += rng.integers(-window, window + 1, nnz)
    np.clip(col, 0, nrow - 1, out=col)
    val = np.empty(nnz, np.complex128)
    val.real = rng.uniform(-1, 1, nnz)
    val.imag = rng.uniform(-1, 1, nnz)
    return rowptr, col, val
