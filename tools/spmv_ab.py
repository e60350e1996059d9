"""Stand-alone stencil SpMV, cold and back to back, at 128^3 and 256^3 (A/B runs of library variants):
    python tools/spmv_ab.py [tag]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402


def main():
    import mgpreconditionedgcr_amd as mg
    from mgpreconditionedgcr_amd import Field, Sparse, problems
    mg.init(0)
    tag = sys.argv[1] if len(sys.argv) > 1 else ""
    for n in (128, 256):
        N, ncol, rowptr, col, val = problems.poisson3d_csr(n)
        A = Sparse(N, ncol, rowptr, col, val)
        del rowptr, col, val
        xf, yf = Field((n, n, n)).fill_rhs(0), Field((n, n, n))
        cold, cp = bench.cold_apply_ms(mg, A, xf, yf, 30, Field)
        warm = A.bench_apply(xf, yf, reps=50)
        print(json.dumps({"tag": tag, "n": n, "cold_us": round(cold["median"] * 1e3, 2), "cold_min_us": round(cold["min"] * 1e3, 2),
                          "b2b_us": round(warm * 1e3, 2), "copy_cold_us": round(cp["median"] * 1e3, 2)}), flush=True)
        del A, xf, yf


if __name__ == "__main__":
    main()
