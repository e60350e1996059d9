"""Residual update inside the windowed apply (csrc/gcr_fused_xr_tile.h) against the separate kernels: same history / x?
   python tools/xr_tile_check.py          [ENV_VAR [n]]   (children with ENV_VAR = 0 / 1; default MGCR_XR_FUSE_TILE)"""
import os
import subprocess
import sys
import tempfile

import numpy as np

CHILD = r'''
import sys
import numpy as np
sys.path.insert(0, ".")
import mgpreconditionedgcr_amd as mg
from mgpreconditionedgcr_amd import *
n, restart, its, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
N, ncol, rowptr, col, val = problems.poisson3d_csr(n)
A = Sparse(N, ncol, rowptr, col, val)
b = Field((n, n, n)).fill_rhs(0)
x = Field((n, n, n)).set_zero()
g = GCR(A, GCR_Param(0, restart, its, 1e-30, False))
g.solve(b, x)
np.save(out, np.concatenate([np.asarray(g.last_history), x.to_numpy().ravel().view(np.float64)]))
print("layout", A.ell_layout() if hasattr(A, "ell_layout") else None, "its", g.last_iterations)
'''


def run(n, restart, its, env_add):
    d = tempfile.mkdtemp()
    f = os.path.join(d, "o.npy")
    env = dict(os.environ, **env_add)
    p = subprocess.run([sys.executable, "-c", CHILD, str(n), str(restart), str(its), f], env=env, capture_output=True, text=True, timeout=300)
    if p.returncode:
        print(p.stdout[-2000:], p.stderr[-3000:])
        raise SystemExit(1)
    return np.load(f), p.stdout.strip()


VAR = sys.argv[1] if len(sys.argv) > 1 else "MGCR_XR_FUSE_TILE"
CASES = ((64, 5, 23, {"MGCR_FUSED_TILE_REACH": "1024", "MGCR_XR_FUSE_ROWS": "0", "MGCR_RESIDENT": "0", "MGCR_STEPBUILD": "0"}),
                               (64, 10, 23, {"MGCR_FUSED_TILE_REACH": "1024", "MGCR_XR_FUSE_ROWS": "0", "MGCR_RESIDENT": "0", "MGCR_STEPBUILD": "0"}),
                               (192, 5, 12, {}), (192, 3, 7, {}), (256, 5, 7, {}), (256, 10, 12, {}))
for n, restart, its, extra in CASES:
    if len(sys.argv) > 2 and n != int(sys.argv[2]):
        continue
    a, oa = run(n, restart, its, dict(extra, **{VAR: "0"}))
    b, ob = run(n, restart, its, dict(extra, **{VAR: "1"}))
    h = its + 1
    same_h = np.array_equal(a[:h], b[:h])
    same_x = np.array_equal(a[h:], b[h:])
    rel = np.max(np.abs(a[:h] - b[:h]) / np.abs(a[:h]))
    relx = np.max(np.abs(a[h:] - b[h:])) / np.max(np.abs(a[h:]))
    print(f"n={n} restart={restart} its={its}: history identical {same_h} (max rel {rel:.2e}), x identical {same_x} (rel {relx:.2e}), last hist {a[h-1]:.6e}", flush=True)
