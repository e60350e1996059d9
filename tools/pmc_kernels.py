#!/usr/bin/env python3
"""HBM-side bytes per dispatch of every kernel of a command, from two rocprofv3 PMC passes (--pmc FETCH_SIZE / --pmc WRITE_SIZE, run
separately; (2 x FETCH + WRITE) x 1024: gfx950 correction, MI355X_MICROARCH.md):   python tools/pmc_kernels.py <fetch dir> <write dir> [min MB]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.pmc_traffic import load  # noqa: E402

fe, wr = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
lo = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
print("| kernel | dispatches | read MB | written MB | total MB per dispatch |")
print("|---|---|---|---|---|")
rows = []
for k in set(fe) | set(wr):
    nf, sf = fe.get(k, [0, 0.0])
    nw, sw = wr.get(k, [0, 0.0])
    r = 2.0 * (sf / nf if nf else 0.0) * 1024.0 / 1e6
    w = (sw / nw if nw else 0.0) * 1024.0 / 1e6
    rows.append((r + w, k, max(nf, nw), r, w))
for t, k, n, r, w in sorted(rows, reverse=True):
    if t >= lo:
        print("| `%s` | %d | %.1f | %.1f | %.1f |" % (k[:110], n, r, w, t))
