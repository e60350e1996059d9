#!/usr/bin/env python3
"""HBM bytes per apply of a stand-alone SpMV workload from two rocprofv3 PMC passes (separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs of
`python3 bench.py --workload irregular_spmv` with MGCR_BENCH_IRREGULAR_WINDOW=<w>), merged into profiles/pmc_traffic.json under
"workloads" (bench.py prints it as roofline.traffic while the kernel-source fingerprint matches):

    python tools/pmc_workload.py <name> <fetch dir> <write dir> <pmc_traffic.json> [<tcc dir>]

Per kernel: (2 x FETCH_SIZE + WRITE_SIZE) x 1024 bytes per dispatch (gfx950 correction, MI355X_MICROARCH.md); the apply = one dispatch of
each SpMV kernel the workload launches (slab kernel, chunked tail, long-row tail)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.pmc_traffic import load  # noqa: E402

SPMV = ("ell_spmv_rowthread", "ell_spmv_window", "ell_spmv_lanes", "csr_tail_chunk_kernel", "csr_tail_kernel")


def apply_bytes(kern):
    """One apply = one dispatch of each kernel the WHOLE apply launches.  The workload also times the parts alone (spmv_part), which
    shows up as further kernels: with a window kernel that multiplies the tail itself (template flag TAIL = true) the apply is
    that kernel (+ the long-row tail kernel); otherwise slab kernel + chunked tail (+ long-row tail)."""
    fused = [k for k in kern if k.startswith("ell_spmv_window") and k.rstrip(">").endswith("true")]
    if fused:
        names = fused + [k for k in kern if k.startswith("csr_tail_kernel")]
    else:
        names = [k for k in kern if not (k.startswith("ell_spmv_window") and k.rstrip(">").endswith("true"))]
    return sum(kern[k]["hbm_bytes_per_dispatch"] for k in names)


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--recompute":   # re-derive hbm_bytes_per_apply of every workload from its per-kernel entries
        d = json.load(open(sys.argv[2]))
        for w in (d.get("workloads") or {}).values():
            w["hbm_bytes_per_apply"] = apply_bytes(w["kernels"])
        json.dump(d, open(sys.argv[2], "w"), indent=1)
        return
    name, dfetch, dwrite, out_json = sys.argv[1:5]
    fe, wr = load(dfetch, "FETCH_SIZE"), load(dwrite, "WRITE_SIZE")
    kern, total = {}, 0.0
    for k in sorted(set(fe) | set(wr)):
        if not k.startswith(SPMV):
            continue
        nf, sf = fe.get(k, [0, 0.0])
        nw, sw = wr.get(k, [0, 0.0])
        b = (2.0 * (sf / nf if nf else 0.0) + (sw / nw if nw else 0.0)) * 1024.0
        kern[k] = {"dispatches": max(nf, nw), "hbm_bytes_per_dispatch": b}
        total += b
    total = apply_bytes(kern)
    rec = {"hbm_bytes_per_apply": total, "kernels": kern,
           "note": "PMC FETCH_SIZE / WRITE_SIZE passes (tools/pmc_workload.py): sum over the apply's kernels of (2 x FETCH + WRITE) x 1024 per dispatch"}
    if len(sys.argv) > 5:   # L2 request counters of the same command (one more pass)
        tcc = {}
        for c in ("TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum", "TCP_TCC_READ_REQ_sum"):
            for k, (n, v) in load(sys.argv[5], c).items():
                if k.startswith(SPMV) and n:
                    tcc.setdefault(k, {})[c] = v / n
        rec["l2_requests_per_dispatch"] = tcc
    import bench
    rec["src_sha16"] = bench.source_sha16()
    try:
        d = json.load(open(out_json))
    except Exception:
        d = {}
    d.setdefault("workloads", {})[name] = rec
    json.dump(d, open(out_json, "w"), indent=1)
    print(json.dumps({name: rec}, indent=1))


if __name__ == "__main__":
    main()
