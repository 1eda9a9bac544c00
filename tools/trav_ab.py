#!/usr/bin/env python3
"""A/B of the traversal schedules (psm_rt_set_traverse_mode) on the ray sets of one frame, in ONE process with
interleaved rounds (cdna_hip_programming.md 5.4 rule 24): per bounce round, wall time of an intersection() running
alone on the device (host timer around `reps` back-to-back calls + sync) and the sum of its launches' HIP-event
durations. No oracle: product path only.
usage (GPU box): python tools/trav_ab.py [--scene sponza_like|stress] [--width W --height H] [--reps N] [--rounds K]"""
import argparse
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
psm = importlib.import_module("prismarine-core_amd")
scenes = importlib.import_module("prismarine-core_amd.scenes")

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="sponza_like")
ap.add_argument("--tris", type=int, default=0)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--rounds", type=int, default=3, help="interleaved measurement rounds")
ap.add_argument("--configs", default="", help="comma list of config names to run (default all)")
args = ap.parse_args()

CONFIGS = [
    ("whole", "whole", {}),
    ("cap96", "phased", {"caps": [96]}),
    ("cap64x2", "phased", {"caps": [64, 64]}),
    ("live8", "adaptive", {"min_live": 8, "min_steps": 8}),
    ("live12", "adaptive", {"min_live": 12, "min_steps": 8}),
    ("live16", "adaptive", {"min_live": 16, "min_steps": 8}),
    ("live20", "adaptive", {"min_live": 20, "min_steps": 8}),
    ("live24", "adaptive", {"min_live": 24, "min_steps": 8}),
    ("live32", "adaptive", {"min_live": 32, "min_steps": 8}),
    ("live16s24", "adaptive", {"min_live": 16, "min_steps": 24}),
    ("live16f16k", "adaptive", {"min_live": 16, "min_steps": 8, "final_rays": 16384}),
    ("live16f1k", "adaptive", {"min_live": 16, "min_steps": 8, "final_rays": 1024}),
    ("live16l4", "adaptive", {"min_live": 16, "min_steps": 8, "max_launches": 4}),
]
if args.configs:
    want = args.configs.split(",")
    CONFIGS = [c for c in CONFIGS if c[0] in want]

sc = getattr(scenes, args.scene)(**({"n_tris": args.tris} if args.tris else {}))
W, H = args.width, args.height
ctx = psm.Context(0)
th = psm.TriangleHierarchy(ctx)
th.allocate(sc["tris"].shape[0])
th.loadTriangles(sc["tris"], sc["normals"], sc["mats"])
th.build()
ms = psm.MaterialSet()
for m in sc["materials"]:
    ms.addSubmat(m)
rt = psm.Pipeline(ctx, seed=1000)
rt.resizeBuffers(W, H)
rt.resize(W, H)
cam = scenes.camera_matrices(sc["eye"], sc["view"], W, H)
rt.camera_matrices(cam[0], cam[1])
rt.applyMaterials(ms)
sets = []
for rnd in range(16):
    if rt.getRayCount() <= 0:
        break
    sets.append(rt.download_rays())
    rt.intersection(th)
    rt.shade()
print("ray sets:", [len(s) for s in sets], flush=True)


def select(mode, kw):
    if mode == "phased":
        rt.setTraversePhases(kw["caps"], min_rays=0)
    elif mode == "adaptive":
        rt.setTraverseAdaptive(min_rays=0, **kw)
    rt.setTraverseMode(mode)


res = {}
for rep in range(args.rounds):
    for si, rays in enumerate(sets):
        rt.upload_rays(rays)
        for name, mode, kw in CONFIGS:
            select(mode, kw)
            rt.resetHits(); rt.intersection(th, force=True); ctx.sync()  # warm
            ctx.stats_enable(True, False); ctx.stats_reset()
            t0 = time.perf_counter()
            for _ in range(args.reps):
                rt.resetHits(); rt.intersection(th, force=True)
            ctx.sync()
            wall = (time.perf_counter() - t0) / args.reps * 1e3
            st = ctx.stats()
            ctx.stats_enable(False, False)
            res.setdefault((name, si), []).append((wall, st.traverse_ms / args.reps, st.traverse_launches // args.reps))

print("%-12s" % "config" + "".join("  round%d wall/ev ms (n)" % i for i in range(len(sets))) + "   sum wall")
for name, mode, kw in CONFIGS:
    line, tot = "%-12s" % name, 0.0
    for si in range(len(sets)):
        r = res[(name, si)]
        wall = min(x[0] for x in r)
        ev = min(x[1] for x in r)
        tot += wall
        line += "   %6.3f / %6.3f (%2d)" % (wall, ev, r[0][2])
    print(line + "   %7.3f" % tot, flush=True)
