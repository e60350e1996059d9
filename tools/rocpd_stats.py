"""rocprofv3 (ROCm 7.2) writes its kernel trace as a rocpd sqlite database by default; this turns it into the
per-kernel statistics CSV (`Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs, StdDev`) that
`rocprofv3 --stats --output-format csv` prints and that tools/roofline_table.py reads.

    python tools/rocpd_stats.py gpurun_out/prof10/r10_results.db > profiles/r01_bench_kernel_stats_v10.csv
"""
import csv
import sqlite3
import statistics
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    agg = {}
    for name, start, end in db.execute("select name, start, end from kernels"):
        agg.setdefault(name, []).append(end - start)
    total = sum(sum(v) for v in agg.values())
    rows = [(n, len(v), sum(v), sum(v) / len(v), 100.0 * sum(v) / total, min(v), max(v), statistics.pstdev(v)) for n, v in agg.items()]
    rows.sort(key=lambda r: -r[2])
    w = csv.writer(sys.stdout, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    w.writerows(rows)


if __name__ == "__main__":
    main()
