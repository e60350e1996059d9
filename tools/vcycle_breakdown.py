"""Per-level kernel breakdown of ONE V-cycle from a rocprofv3 kernel trace of tools/vcycle_prof.py (rocpd sqlite database).

    rocprofv3 --kernel-trace -d gpurun_out/vc -o vc -- python3 tools/vcycle_prof.py
    python tools/vcycle_breakdown.py gpurun_out/vc/vc_results.db > profiles/r01_vcycle_breakdown.md

The last cycle of the trace is cut out (period = distance between the two last fine-level prolongation kernels) and its
kernels are attributed to a level by the number of rows they work on (launch size)."""
import collections
import re
import sqlite3
import sys


def short(n):
    n = n.replace("void mgcr::", "").replace("mgcr::", "")
    m = re.match(r"([A-Za-z_0-9]+)", n)
    return m.group(1)


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, start, end, grid_x, workgroup_x from kernels order by start").fetchall()
    n0 = max(r[3] for r in rows if "expand_add" in r[0])
    ex = [i for i, r in enumerate(rows) if "expand_add" in r[0] and r[3] == n0]
    per = ex[-1] - ex[-2]
    cyc = rows[len(rows) - per:]
    wall = (cyc[-1][2] - cyc[0][1]) / 1e3
    # levels by rows: solver kernels run min(rows / 1024, 512) workgroups of 1024 threads, SpMV / transfer kernels one thread per row
    sizes = sorted({r[3] for r in cyc if "expand_add" in r[0] or "restrict" in r[0]}, reverse=True)
    fine = [n0] + [s for s in sizes if s != n0]
    # rows per level: n0, n0/8, n0/64 ...
    lev_rows = [n0 // (8 ** l) for l in range(3)]

    def level_of(r):
        g, name = r[3], r[0]
        dur = (r[2] - r[1]) / 1e3
        if g <= 1024:
            return 9   # one-workgroup bookkeeping kernels of the nested solves (any level)
        if "restrict" in name:   # launched with one thread per COARSE unknown: belongs to the finer level
            for l in range(2):
                if g == lev_rows[l + 1]:
                    return l
        for l, nr in enumerate(lev_rows):
            if g == nr:
                return l
        # reduction-shaped kernels (<= 512 x 1024 threads): tell the levels apart by their duration class
        if g == 512 * 1024:
            return 0 if dur > 60 else 1
        if g == lev_rows[2]:
            return 2
        return 2 if g <= lev_rows[2] else 1

    agg = collections.OrderedDict()
    tot = collections.Counter()
    for r in cyc:
        l = level_of(r)
        key = (l, short(r[0]))
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += (r[2] - r[1]) / 1e3
        tot[l] += (r[2] - r[1]) / 1e3
    print("# One V-cycle, Poisson 256^3, 3 levels (2^3 aggregates), smoother 2 sweeps of GCR(10), coarsest GCR(10) 50 iterations")
    print()
    print("%d kernels, %.0f us wall under rocprofv3 (sum of kernel durations %.0f us)" % (len(cyc), wall, sum(tot.values())))
    print()
    for l in sorted(tot):
        if l == 9:
            print("## one-workgroup bookkeeping kernels of the nested solves (all levels): %.0f us" % tot[l])
        else:
            print("## level %d (%d rows): %.0f us" % (l, lev_rows[l], tot[l]))
        print()
        print("| kernel | launches | total us | avg us |")
        print("|---|---|---|---|")
        for (ll, name), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            if ll == l:
                print("| `%s` | %d | %.1f | %.1f |" % (name, c, t, t / c))
        print()


if __name__ == "__main__":
    main()
