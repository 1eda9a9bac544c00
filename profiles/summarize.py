#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel_stats / counter_collection) into small committed summaries.
usage: summarize.py stats <kernel_stats.csv> | pmc <counter_collection.csv>"""
import collections
import csv
import sys


def short(name):
    name = name.replace("void ", "").split("(")[0]
    # rt_traverse<COUNT, CHAIN, PHASED, GROUP>: GROUP = false keeps the three-argument name of rounds 1-2
    return name.replace(", false>", ">") if name.count(",") == 3 and name.endswith(", false>") else name


def stats(path):
    print("%-34s %6s %12s %12s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for r in csv.DictReader(open(path)):
        print("%-34s %6s %12.1f %12.2f %8.2f" % (short(r["Name"])[:34], r["Calls"], float(r["TotalDurationNs"]) / 1e3,
                                                float(r["AverageNs"]) / 1e3, float(r["Percentage"])))


def pmc(path):
    """per kernel and counter: calls, total, per call (one CSV may hold several counters of one pass)"""
    agg = collections.defaultdict(lambda: [0, 0.0])
    counters = []
    for r in csv.DictReader(open(path)):
        c = r["Counter_Name"]
        if c not in counters:
            counters.append(c)
        k = (short(r["Kernel_Name"]), c)
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    print("counters %s (rocprofv3 values summed over a kernel's dispatches; FETCH_SIZE / WRITE_SIZE in KB; gfx950: FETCH_SIZE "
          "reports half the bytes of wide streaming reads)" % " ".join(counters))
    kernels = sorted({k[0] for k in agg}, key=lambda k: -max(agg[(k, c)][1] for c in counters if (k, c) in agg))
    for c in counters:
        print("-- %s" % c)
        print("%-40s %6s %16s %16s" % ("kernel", "calls", "total", "per_call"))
        for k in kernels:
            if (k, c) in agg:
                n, v = agg[(k, c)]
                print("%-40s %6d %16.1f %16.2f" % (k[:40], n, v, v / n))


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc}[sys.argv[1]](sys.argv[2])
