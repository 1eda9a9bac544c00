#!/usr/bin/env python3
"""Per time bin of a rocprofv3 kernel trace: queues with kernels, kernels started, fraction of the bin with no kernel
resident, with a traversal kernel resident, mean kernels resident -- to find the steady state of a frames-in-flight run.
usage: tools/trace_bins.py <kernel_trace.csv> [bin_ms]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
binw = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 5e6
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in rows)
t0, t1 = ev[0][0], max(e[1] for e in ev)
nb = int((t1 - t0) / binw) + 1
import collections
q = [set() for _ in range(nb)]
n = [0] * nb
busy = [0.0] * nb
trav = [0.0] * nb
occ = [0.0] * nb
# union busy via sweep
pts = sorted([(s, 1) for s, e, k, qq in ev] + [(e, -1) for s, e, k, qq in ev])
cur, last = 0, pts[0][0]
def add(arr, a, b, w=1.0):
    i = int((a - t0) / binw)
    while a < b:
        edge = min(b, t0 + (i + 1) * binw)
        arr[i] += (edge - a) * w
        a = edge
        i += 1
for t, d in pts:
    if cur > 0:
        add(busy, last, t)
        add(occ, last, t, cur)
    last, cur = t, cur + d
tp = sorted([(s, 1) for s, e, k, qq in ev if "rt_traverse" in k] + [(e, -1) for s, e, k, qq in ev if "rt_traverse" in k])
cur, last = 0, tp[0][0]
for t, d in tp:
    if cur > 0:
        add(trav, last, t)
    last, cur = t, cur + d
for s, e, k, qq in ev:
    i = int((s - t0) / binw)
    q[i].add(qq)
    n[i] += 1
print("bin start ms  queues  kernels  idle  traversal-resident  mean-resident")
for i in range(nb):
    print("%10.1f  %6d  %7d  %5.2f  %8.2f  %8.2f" % (i * binw / 1e6, len(q[i]), n[i], 1 - busy[i] / binw, trav[i] / binw, occ[i] / binw))

# the timeline of one queue in the densest part of the trace: kernel, start, gap to the queue's previous kernel, duration
if len(sys.argv) > 3:
    span = float(sys.argv[3]) * 1e6
    best = max(range(nb), key=lambda i: n[i])
    a = t0 + best * binw
    qs = collections.Counter(qq for s, e, k, qq in ev if a <= s < a + span and "rt_shade" in k)
    qid = qs.most_common(1)[0][0]
    print("queue %s from %.1f ms on:" % (qid, best * binw / 1e6))
    prev = None
    for s, e, k, qq in ev:
        if qq != qid or s < a or s >= a + span:
            continue
        name = k.split("(")[0].replace("void ", "").replace("psm::", "")[:44]
        print("%9.1f us  +gap %7.1f  dur %8.1f  %s" % ((s - a) / 1e3, (s - prev) / 1e3 if prev else 0, (e - s) / 1e3, name))
        prev = e
