#!/usr/bin/env python3
"""Turn the PMC passes of profiles/collect.sh into the per-launch figures bench.py's `roofline` object quotes.

usage: profiles/make_traffic.py <out.json> <tag> <scene> <width> <height> [<tag> <scene> <width> <height> ...]
reads gpurun_out/<tag>_{fetch,write,sq,tcc}/**/counter_collection.csv and gpurun_out/<tag>_{kt,kt1,kt1a}/**/kernel_stats.csv
(round 3: the PMC passes profile the bench AS TIMED, so the hand-over kernel of the timed region, rt_traverse<false, false,
true>, has counters of its own next to the single-launch kernel of the serial passes)

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


def norm(k):
    """rt_traverse<COUNT, CHAIN, PHASED>"""
    return k.replace("void ", "").split("(")[0]


def per_kernel(tag, sub, want):
    files = glob.glob(os.path.join(ROOT, "gpurun_out", "%s_%s" % (tag, sub), "**", "*counter_collection.csv"), recursive=True)
    out = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    if not files:
        return out
    for r in csv.DictReader(open(files[0])):
        k = norm(r["Kernel_Name"])
        if want not in k:
            continue
        a = out[k][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return out


def kernel_avg_us(tag, want, sub="kt1"):
    files = glob.glob(os.path.join(ROOT, "gpurun_out", "%s_%s" % (tag, sub), "**", "*kernel_stats.csv"), recursive=True)
    res = {}
    if files:
        for r in csv.DictReader(open(files[0])):
            k = norm(r["Name"])
            if want in k:
                res[k] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3}
    return res


def main():
    out_path = sys.argv[1]
    entries = []
    args = sys.argv[2:]
    for q in range(0, len(args), 4):
        tag, scene, w, h = args[q], args[q + 1], int(args[q + 2]), int(args[q + 3])
        # hand-over kernel: the passes of the bench as timed; single-launch kernels: the --lanes 1 passes (whole rounds only)
        def both(sub):
            a, b = per_kernel(tag, sub, "rt_traverse"), per_kernel(tag, sub + "1", "rt_traverse")
            out = {k: v for k, v in a.items() if k.endswith("true>")}
            out.update({k: v for k, v in (b if b else a).items() if not k.endswith("true>")})
            return out
        fetch, write, sq, tcc = both("fetch"), both("write"), both("sq"), both("tcc")
        alone = kernel_avg_us(tag, "rt_traverse", "kt1")        # --lanes 1: single-launch kernel, every launch alone
        alone.update({k: v for k, v in kernel_avg_us(tag, "rt_traverse", "kt1a").items() if k.endswith("true>")})  # hand-over kernel alone
        flight = kernel_avg_us(tag, "rt_traverse", "kt")        # as timed: launches of different frames overlap
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
            if k in alone:
                e["alone_avg_us"] = alone[k]["avg_us"]
                e["alone_calls"] = alone[k]["calls"]
            if k in flight:
                e["in_flight_avg_us"] = flight[k]["avg_us"]
                e["in_flight_calls"] = flight[k]["calls"]
            e["source"] = ("rocprofv3 --pmc passes of profiles/collect.sh %s (FETCH_SIZE, WRITE_SIZE, SQ_*, TCC_* each in its own "
                           "pass of the bench as timed; launches serialised by the profiler), bench.py --scene %s --width %d --height %d; "
                           "alone_avg_us from the --lanes 1 kernel traces, in_flight_avg_us from the trace of the bench as timed" % (
                               tag, scene, w, h))
            entries.append(e)
        # the shading kernel's VALU instruction count per launch (SQ pass as timed): with the traversal kernels' it says how much
        # of the chip's VALU issue capacity a timed step uses (bench.py: valu_issue_frac_of_step)
        for k, c in sorted(per_kernel(tag, "sq", "rt_shade").items()):
            if "SQ_INSTS_VALU" in c:
                entries.append({"scene": scene, "width": w, "height": h, "kernel": k, "launches_sampled": c["SQ_INSTS_VALU"][0],
                                "sq_per_launch": {n: a[1] / a[0] for n, a in c.items()},
                                "source": "rocprofv3 --pmc SQ_* pass of profiles/collect.sh %s (bench as timed)" % tag})
    # provenance: the kernel sources the passes ran with (collect.sh wrote their hashes on the GPU box) and the commit this is
    # filed under; bench.py recomputes the hashes from the sources it runs from and flags a replay of stale counters
    prov = {"sources_sha256": {}, "commit": None, "tags": sorted({args[q] for q in range(0, len(args), 4)})}
    for tag in prov["tags"]:
        path = os.path.join(ROOT, "gpurun_out", "%s_sources.sha256" % tag)
        if os.path.exists(path):
            for line in open(path):
                h, name = line.split()
                if prov["sources_sha256"].setdefault(name, h) != h:
                    raise SystemExit("make_traffic: %s differs between the tags' passes" % name)
    try:
        import subprocess
        prov["commit"] = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "HEAD"], text=True).strip()
        prov["commit_note"] = "HEAD when make_traffic.py ran" + ("; the tree had uncommitted changes in csrc/" if subprocess.check_output(
            ["git", "-C", ROOT, "status", "--porcelain", "prismarine-core_amd/csrc"], text=True).strip() else "")
    except Exception:
        pass
    json.dump({"provenance": prov, "entries": entries}, open(out_path, "w"), indent=1)
    print("wrote %d entries to %s" % (len(entries), out_path))


if __name__ == "__main__":
    main()
