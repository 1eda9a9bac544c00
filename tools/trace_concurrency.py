#!/usr/bin/env python3
"""Kernel concurrency of a rocprofv3 --kernel-trace run of bench.py with frames in flight: how many kernels run at
once, how long each kernel type is resident, and the timeline of one queue (launch gaps, stretched kernels).
usage: tools/trace_concurrency.py <kernel_trace.csv> [window_ms] [queue_id]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 25e6
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
             r["Kernel_Name"].split("(")[0].replace("void ", "").replace("psm::", ""), r["Queue_Id"]) for r in rows)
t1 = max(e[1] for e in ev)
lo = t1 - win
sel = [e for e in ev if e[0] >= lo]
pts = sorted([(s, 1, n) for s, e, n, q in sel] + [(e, -1, n) for s, e, n, q in sel])
cur, last, hist, curk, tk = 0, pts[0][0], collections.Counter(), collections.Counter(), collections.Counter()
for t, d, n in pts:
    hist[cur] += t - last
    for k, v in curk.items():
        if v > 0:
            tk[k] += t - last
    last, cur = t, cur + d
    curk[n] += d
tot = sum(hist.values())
print("last %.1f ms of the trace: %d kernels on queues %s" % (win / 1e6, len(sel), sorted({e[3] for e in sel})))
print("fraction of time with k kernels resident: " + "  ".join("%d: %.3f" % (k, hist[k] / tot) for k in sorted(hist)))
dur, cnt = collections.Counter(), collections.Counter()
for s, e, n, q in sel:
    dur[n] += e - s
    cnt[n] += 1
print("%-40s %6s %10s %10s %10s" % ("kernel", "n", "total ms", "avg ms", "resident"))
for k, v in sorted(dur.items(), key=lambda x: -x[1])[:14]:
    print("%-40s %6d %10.2f %10.3f %10.3f" % (k[:40], cnt[k], v / 1e6, v / 1e6 / cnt[k], tk[k] / tot))
if len(sys.argv) > 3:
    prev = None
    for s, e, n, q in [x for x in sel if x[3] == sys.argv[3]][:120]:
        print("%9.1f us  +gap %7.1f  dur %8.1f  %s" % ((s - lo) / 1e3, (s - prev) / 1e3 if prev else 0, (e - s) / 1e3, n))
        prev = e
