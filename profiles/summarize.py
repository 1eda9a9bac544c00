#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel_stats / counter_collection) into small committed summaries.
usage: summarize.py stats <kernel_stats.csv> | pmc <counter_collection.csv>"""
import collections
import csv
import sys


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0]


def stats(path):
    print("%-34s %6s %12s %12s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for r in csv.DictReader(open(path)):
        print("%-34s %6s %12.1f %12.2f %8.2f" % (short(r["Name"])[:34], r["Calls"], float(r["TotalDurationNs"]) / 1e3,
                                                float(r["AverageNs"]) / 1e3, float(r["Percentage"])))


def pmc(path):
    agg = collections.defaultdict(lambda: [0, 0.0])
    cname = None
    for r in csv.DictReader(open(path)):
        cname = r["Counter_Name"]
        k = short(r["Kernel_Name"])
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    print("counter %s (KB as reported by rocprofv3; gfx950: FETCH_SIZE under-reports wide streaming reads 2x)" % cname)
    print("%-34s %6s %14s %14s" % ("kernel", "calls", "total_KB", "per_call_KB"))
    for k, (n, v) in sorted(agg.items(), key=lambda x: -x[1][1]):
        print("%-34s %6d %14.1f %14.2f" % (k[:34], n, v, v / n))


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc}[sys.argv[1]](sys.argv[2])
