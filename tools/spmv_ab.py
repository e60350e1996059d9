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
        # the same with a READ-ONLY sweep (a dot product over 2 x 256 MiB): cold caches without the sweep's dirty lines
        import ctypes
        nf = 16 * 1024 * 1024
        fa, fb = Field((nf,)).set_zero(), Field((nf,)).set_zero()
        t_c = ctypes.c_double()
        rs, rc = [], []
        for _ in range(30):
            fa.dot(fb)
            mg.lib().mgcr_timer_start()
            A(xf, out=yf)
            mg.lib().mgcr_timer_stop(ctypes.byref(t_c))
            rs.append(t_c.value)
        for _ in range(30):
            fa.dot(fb)
            mg.lib().mgcr_timer_start()
            yf.assign(xf)
            mg.lib().mgcr_timer_stop(ctypes.byref(t_c))
            rc.append(t_c.value)
        del fa, fb
        rs.sort(); rc.sort()
        print(json.dumps({"tag": tag, "n": n, "read_sweep_cold_us": round(rs[15] * 1e3, 2), "read_sweep_copy_us": round(rc[15] * 1e3, 2)}), flush=True)
        print(json.dumps({"tag": tag, "n": n, "cold_us": round(cold["median"] * 1e3, 2), "cold_min_us": round(cold["min"] * 1e3, 2),
                          "b2b_us": round(warm * 1e3, 2), "copy_cold_us": round(cp["median"] * 1e3, 2)}), flush=True)
        del A, xf, yf


if __name__ == "__main__":
    main()
