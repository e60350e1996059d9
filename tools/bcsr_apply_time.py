"""Apply time of the unstructured block-CSR operator of the bcsr workload (bs 20, 5-64 blocks per row, 3 GB):
    python tools/bcsr_apply_time.py [tag]      (environment: MGCR_BCSR_ORDER=0 for the stored row order)"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402


def main():
    import mgpreconditionedgcr_amd as mg
    from mgpreconditionedgcr_amd import Field, HierarchicalSparse
    mg.init(0)
    tag = sys.argv[1] if len(sys.argv) > 1 else ""
    bs, nb = 20, 36000
    rng = np.random.default_rng(0)
    counts = rng.integers(5, 65, size=nb)
    rows = np.repeat(np.arange(nb, dtype=np.int32), counts)
    cols = rng.integers(0, nb, size=rows.size).astype(np.int32)
    blocks = np.empty((rows.size, bs, bs), dtype=np.complex128)
    blocks.real = rng.standard_normal((rows.size, bs, bs)).astype(np.float64) * 0.01
    blocks.imag = 0.0
    H = HierarchicalSparse(nb, nb, rows, cols, blocks)
    nbytes = rows.size * bs * bs * 16 + 2 * nb * bs * 16
    del blocks
    x, y = Field((nb * bs,)).fill_rhs(0), Field((nb * bs,))
    ms = H.bench_apply(x, y, reps=20)
    print(json.dumps({"tag": tag, "blocks": int(rows.size), "apply_ms": round(ms, 4), "TBps": round(nbytes / ms / 1e9, 3)}), flush=True)


if __name__ == "__main__":
    main()
