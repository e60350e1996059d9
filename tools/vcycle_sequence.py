"""Kernels of the LAST V-cycle of a rocprofv3 kernel trace of tools/vcycle_prof.py, in launch order, with the gap to the
previous kernel's end:  python tools/vcycle_sequence.py gpurun_out/vc/vc_results.db [min_us]"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.
    rows = db.execute("select name, start, end, grid_x, workgroup_x from kernels order by start").fetchall()
    n0 = max(r[3] for r in rows if "expand_add" in r[0])
    ex = [i for i, r in enumerate(rows) if "expand_add" in r[0] and r[3] == n0]
    per = ex[-1] - ex[-2]
    cyc = rows[len(rows) - per:]
    prev_end = None
    print("| # | kernel | grid x wg | us | gap before, us |")
    print("|---|---|---|---|---|")
    for i, r in enumerate(cyc):
        name = re.match(r"([A-Za-z_0-9]+)", r[0].replace("void mgcr::", "").replace("mgcr::", "")).group(1)
        targs = re.search(r"<(.*)>", r[0])
        dur = (r[2] - r[1]) / 1e3
        gap = (r[1] - prev_end) / 1e3 if prev_end else 0.
        prev_end = r[2]
        if dur >= min_us:
            print("| %d | `%s%s` | %d x %d | %.1f | %.1f |" % (i, name, "<" + targs.group(1) + ">" if targs else "", r[3] // max(r[4], 1), r[4], dur, gap))


if __name__ == "__main__":
    main()
