#!/usr/bin/env python3
"""HBM traffic per kernel and per GCR phase from two rocprofv3 PMC passes of bench.py
(MI355X_MICROARCH.md, HBM / rocprofv3 section: separate --pmc FETCH_SIZE and --pmc WRITE_SIZE runs;
units KiB per dispatch; gfx950 correction: bytes read = 2 x FETCH_SIZE).

    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras
    rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r02_pmc_hbm_traffic.csv profiles/pmc_traffic.json

The JSON is stamped with the fingerprint of the kernel sources it was measured on (bench.source_sha16): bench.py prints
roofline.traffic only while that fingerprint matches the sources it runs.

Writes the per-kernel table (CSV) and the JSON bench.py reads for roofline.traffic: average corrected
HBM bytes per launch of the kernels of each phase of a GCR iteration (xr | apply+dots | build).
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def load(d, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0].replace("mgcr::", "")
            a = acc[name]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return acc


def phase_of(name):
    if name.startswith("xr_update_kernel"):
        return "xr"
    if name.startswith(("step_apply_kernel", "step_apply_tile_kernel", "step_apply_xr", "step_build_kernel", "multidot_kernel")):   # (step_build: apply + dots + build in one launch)
        return "apply_dots"
    if name.startswith(("pat_spmv", "ell_spmv", "csr_tail", "sten_spmv")):
        return "spmv"   # stand-alone applies (set-up, bench.py's replay / cold-cache loops)
    if name.startswith(("build_lean_kernel", "build_close_kernel", "build_kernel")):
        return "build"
    return None


def main():
    dfetch, dwrite, out_csv, out_json = sys.argv[1:5]
    n = int(sys.argv[5]) if len(sys.argv) > 5 else 128
    fe, wr = load(dfetch, "FETCH_SIZE"), load(dwrite, "WRITE_SIZE")
    rows = []
    phases = defaultdict(lambda: [0, 0.0])
    for name in sorted(set(fe) | set(wr)):
        nf, sf = fe.get(name, [0, 0.0])
        nw, sw = wr.get(name, [0, 0.0])
        calls = max(nf, nw)
        favg = sf / nf if nf else 0.0
        wavg = sw / nw if nw else 0.0
        b = (2.0 * favg + wavg) * 1024.0
        rows.append((name, calls, favg, wavg, b))
        ph = phase_of(name)
        if ph:
            phases[ph][0] += calls
            phases[ph][1] += b * calls
    with open(out_csv, "w") as f:
        f.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) -- python3 bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5\n")
        f.write("# MI355X, Poisson %d^3.  Units: KiB per dispatch as reported; gfx950 correction (MI355X_MICROARCH.md HBM section): reads = 2 x FETCH_SIZE\n" % n)
        f.write("kernel,dispatches,FETCH_SIZE_KiB_avg,WRITE_SIZE_KiB_avg,hbm_bytes_per_launch_corrected\n")
        for name, calls, favg, wavg, b in rows:
            f.write('"%s",%d,%.1f,%.1f,%.0f\n' % (name, calls, favg, wavg, b))
    import bench
    js = {"n": n, "src_sha16": bench.source_sha16(),
          "derivation": "(2*FETCH_SIZE + WRITE_SIZE)*1024 per dispatch from %s (separate --pmc passes, gfx950 FETCH_SIZE x2 correction), "
                        "averaged over the launches of the kernels of each phase" % os.path.basename(out_csv),
          "phase_hbm_bytes_per_launch": {k: v[1] / v[0] for k, v in phases.items() if v[0]},
          "phase_launches": {k: v[0] for k, v in phases.items()},
          # per kernel too: bench.py weights these by the mix of kernels its timed iterations launched (iteration k of a restart
          # cycle orthogonalises against k directions), so that `traffic` and `achieved` describe the same launches
          "kernel_hbm_bytes_per_launch": {name: b for name, calls, favg, wavg, b in rows if phase_of(name)}}
    json.dump(js, open(out_json, "w"), indent=1)
    print(json.dumps(js, indent=1))


if __name__ == "__main__":
    main()
