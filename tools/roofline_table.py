#!/usr/bin/env python3
"""Per-kernel achieved-HBM-rate table from a rocprofv3 --kernel-trace --stats CSV of `bench.py`
(Poisson n^3, restart 5).  Bytes are the models of DESIGN.md section 3 for the stored layout
(stencil view of the row-pattern dictionary, lean restart cycles, fused SpMV + dots).

    python tools/roofline_table.py profiles/r01_bench_kernel_stats_v3.csv [n]
"""
import csv
import re
import sys

path = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 128
# --pmc <fetch dir> <write dir>: two more columns — HBM-side bytes per dispatch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
# of the same command ((2 x FETCH + WRITE) x 1024: gfx950 correction, MI355X_MICROARCH.md) and their ratio to the model
pmc = None
if "--pmc" in sys.argv:
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tools.pmc_traffic import load
    k = sys.argv.index("--pmc")
    fe, wr = load(sys.argv[k + 1], "FETCH_SIZE"), load(sys.argv[k + 2], "WRITE_SIZE")
    pmc = {}
    for name in set(fe) | set(wr):
        nf, sf = fe.get(name, [0, 0.0])
        nw, sw = wr.get(name, [0, 0.0])
        pmc[name] = (2.0 * (sf / nf if nf else 0.0) + (sw / nw if nw else 0.0)) * 1024.0
N = n ** 3
nnz = 7 * N - 6 * n * n
V = 16 * N
slab = 7 * ((N + 63) // 64 * 64)
B_pat = ((N + 63) // 64) * 8 * 8 + 7 * 20          # stencil view: 8 presence words of 8 B per wave of 64 rows + the slot table
B_spmv = B_pat + 2 * V
R = 5
rows = list(csv.DictReader(open(path)))
print("| kernel | calls | avg µs | bytes / launch (model) | GB/s | of 8 TB/s |" + (" HBM-side bytes (PMC) | PMC / model |" if pmc else ""))
print("|---|---|---|---|---|---|" + ("---|---|" if pmc else ""))
for r in rows:
    name = r["Name"].split("(")[0].replace("void mgcr::", "").replace("mgcr::", "")
    us = float(r["AverageNs"]) / 1e3
    b = None
    m = re.match(r"multidot_kernel<(\d+)", name)
    if m:
        b = (1 + int(m.group(1))) * V
    m = re.match(r"step_apply_kernel<\d+, \d+, (\d+)[,>]", name) or re.match(r"step_apply_tile_kernel<\d+, (?:true|false), (\d+)[,>]", name)
    if m:
        b = B_spmv + int(m.group(1)) * V            # SpMV + lim direction streams, Ar written once
    m = re.match(r"step_apply_xr_tile_kernel<\d+, (?:true|false), (\d+), (true|false)>", name)
    if m:
        # residual update inside the windowed apply: r and Ap read, r' and A r' written, the other lim - 1 direction streams
        # (Ap is the newest of them: in registers when the last template flag is true, else read again — an L2 hit, not counted)
        b = B_pat + 4 * V + (int(m.group(1)) - 1) * V
    m = re.match(r"step_build_kernel<\d+, \d+, (\d+), (true|false), (true|false)(?:, (?:true|false))?>", name)
    if m:
        lim = int(m.group(1))
        b = B_spmv - V + (2 * lim + 2) * V               # SpMV without the write of Ar; lim streams twice, r re-read, Ap written
        if m.group(3) == "true":
            b = B_spmv - V + (3 * lim + 5) * V           # the step that closes a cycle: + lim p streams, x read and written, P0 written
        if m.group(2) == "true":
            b += 2 * V                                   # + the next step's residual update: r read, r' written
    m = re.match(r"build_lean_kernel<(\d+)[,>]", name)
    if m:
        b = (3 + int(m.group(1))) * V               # r, Ar, lim Aps read; Ap written
    m = re.match(r"build_close_kernel<(\d+)", name)
    if m:
        b = (2 * int(m.group(1)) + 6) * V           # R ps + R Aps + dir + Ar + x read; p, Ap, x written
    m = re.match(r"build_kernel<(\d+), true, true, (true|false), (true|false)>", name)
    if m:
        lim = int(m.group(1))
        b = (4 + 2 * lim) * V + (2 * V if m.group(3) == "true" else 0)
    if name.startswith("xr_update_kernel<true"):
        b = 3 * V
    if name.startswith("xr_update_kernel<false"):
        b = 6 * V
    m = re.match(r"init_apply(_tile)?_kernel", name)
    if m:
        b = B_spmv                                   # SpMV of step 0 + its dot products (no extra streams when b is r0)
    if name.startswith(("ell_spmv_rowthread", "pat_spmv", "sten_spmv")):
        b = B_spmv
    # (copy_kernel: not listed — the run mixes copies of one vector with the 512 MiB cache sweeps of the cold-apply measurement;
    # bench.py reports the copy of one vector on its own: spmv.copy_of_one_vector_ms)
    if b is None or int(r["Calls"]) < 4:
        continue
    gbs = b / us / 1e3
    extra = ""
    if pmc:
        t = pmc.get(name)
        extra = " %.1f MB | %.2f |" % (t / 1e6, t / b) if t else " | |"
    print("| `%s` | %s | %.1f | %.1f MB | %.0f | %.2f |%s" % (name, r["Calls"], us, b / 1e6, gbs, gbs / 8000, extra))
