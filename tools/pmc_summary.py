#!/usr/bin/env python3
"""Per-kernel averages of whatever counters a set of rocprofv3 --pmc output directories hold, plus --stats durations.
    python tools/pmc_summary.py <dir> [<dir> ...]      prints: kernel, dispatches, counter = average per dispatch"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    return re.sub(r"^void ", "", name).split("(")[0].replace("mgcr::", "")


def main():
    for d in sys.argv[1:]:
        acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                a = acc[short(r["Kernel_Name"])][r["Counter_Name"]]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
        for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                a = acc[short(r["Name"])]["avg_duration_us"]
                a[0] = int(r["Calls"])
                a[1] = float(r["AverageNs"]) / 1e3 * int(r["Calls"])
        if acc:
            print("==", d)
        for k in sorted(acc):
            print("  %-70s" % k[:70], "  ".join("%s=%.4g (n=%d)" % (c, v[1] / max(v[0], 1), v[0]) for c, v in sorted(acc[k].items())))


if __name__ == "__main__":
    main()
