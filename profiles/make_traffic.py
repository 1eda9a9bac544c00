#!/usr/bin/env python3
"""Turn the PMC passes of profiles/collect.sh into the per-launch figures bench.py's `roofline` object quotes.

usage: profiles/make_traffic.py <out.json> <tag> <scene> <width> <height> [<tag> <scene> <width> <height> ...]
reads gpurun_out/<tag>_{fetch,write,sq,tcc}/**/counter_collection.csv and gpurun_out/<tag>_kt1/**/kernel_stats.csv

Per traversal kernel (the dominant kernel), averaged over its launches:
  fetch_bytes_raw   FETCH_SIZE x 1024          (gfx950: half the bytes of wide streaming reads, MI355X_MICROARCH.md HBM)
  write_bytes       WRITE_SIZE x 1024
  traffic_bytes     2 x fetch_bytes_raw + write_bytes   (the guide's gfx950 correction; divergent 16-byte reads are not
                    calibrated there, so the raw figure is kept alongside)
  sq                SQ_* counters per launch (VALU instructions, busy / wave quad-cycles, active lane-cycles)
  l2_hit            TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(tag, sub, want):
    files = glob.glob(os.path.join(ROOT, "gpurun_out", "%s_%s" % (tag, sub), "**", "*counter_collection.csv"), recursive=True)
    out = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    if not files:
        return out
    for r in csv.DictReader(open(files[0])):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0]
        if want not in k:
            continue
        a = out[k][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return out


def kernel_avg_us(tag, want):
    files = glob.glob(os.path.join(ROOT, "gpurun_out", "%s_kt1" % tag, "**", "*kernel_stats.csv"), recursive=True)
    res = {}
    if files:
        for r in csv.DictReader(open(files[0])):
            k = r["Name"].replace("void ", "").split("(")[0]
            if want in k:
                res[k] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3}
    return res


def main():
    out_path = sys.argv[1]
    entries = []
    args = sys.argv[2:]
    for q in range(0, len(args), 4):
        tag, scene, w, h = args[q], args[q + 1], int(args[q + 2]), int(args[q + 3])
        fetch, write = per_kernel(tag, "fetch", "rt_traverse"), per_kernel(tag, "write", "rt_traverse")
        sq, tcc = per_kernel(tag, "sq", "rt_traverse"), per_kernel(tag, "tcc", "rt_traverse")
        times = kernel_avg_us(tag, "rt_traverse")
        for k in sorted(fetch):
            n, v = fetch[k]["FETCH_SIZE"]
            e = {"scene": scene, "width": w, "height": h, "kernel": k, "launches_sampled": n,
                 "fetch_bytes_per_launch_raw": v * 1024.0 / n}
            if k in write:
                e["write_bytes_per_launch"] = write[k]["WRITE_SIZE"][1] * 1024.0 / write[k]["WRITE_SIZE"][0]
                e["traffic_bytes_per_launch"] = 2.0 * e["fetch_bytes_per_launch_raw"] + e["write_bytes_per_launch"]
            if k in sq:
                e["sq_per_launch"] = {c: a[1] / a[0] for c, a in sq[k].items()}
                s = e["sq_per_launch"]
                if s.get("SQ_ACTIVE_INST_VALU"):
                    e["valu_lane_utilisation"] = s.get("SQ_THREAD_CYCLES_VALU", 0.0) / (s["SQ_ACTIVE_INST_VALU"] * 64.0)
            if k in tcc and "TCC_HIT_sum" in tcc[k]:
                hit, miss = tcc[k]["TCC_HIT_sum"][1], tcc[k]["TCC_MISS_sum"][1]
                e["l2_hit"] = hit / max(hit + miss, 1.0)
            if k in times:
                e["kernel_trace_avg_us"] = times[k]["avg_us"]
                e["kernel_trace_calls"] = times[k]["calls"]
            e["source"] = ("rocprofv3 --pmc passes of profiles/collect.sh %s (FETCH_SIZE, WRITE_SIZE, SQ_*, TCC_* each in its own "
                           "pass, --lanes 1), bench.py --scene %s --width %d --height %d" % (tag, scene, w, h))
            entries.append(e)
        # the schedule of the timed region (frames in flight): the SQ pass of the bench as timed, per traversal kernel
        sq4 = per_kernel(tag, "sq4", "rt_traverse")
        for k in sorted(sq4):
            c = sq4[k]
            if "SQ_ACTIVE_INST_VALU" not in c:
                continue
            act, thr, ins = c["SQ_ACTIVE_INST_VALU"], c.get("SQ_THREAD_CYCLES_VALU", [0, 0.0]), c.get("SQ_INSTS_VALU", [0, 0.0])
            entries.append({"scene": scene, "width": w, "height": h, "kernel": k, "pass": "as timed (frames in flight)",
                            "launches_sampled": act[0], "valu_insts_total": ins[1], "valu_lane_utilisation": thr[1] / (act[1] * 64.0),
                            "source": "rocprofv3 --pmc SQ_* pass of profiles/collect.sh %s with the bench's default frames in flight "
                                      "(5 frames of the timed schedule, then the bench's counting and serial kernel passes)" % tag})
    json.dump({"entries": entries}, open(out_path, "w"), indent=1)
    print("wrote %d entries to %s" % (len(entries), out_path))


if __name__ == "__main__":
    main()
