#!/usr/bin/env python3
"""What would one traversal launch over the rays of several frames buy? ("rounds of several in-flight frames
batched into one launch".) Takes the ray queues of F frames at bounce round r (camera seeds differ), times the single-launch
traversal of each queue alone and of their concatenation in one queue (HIP events, launches alone on the chip).
usage (GPU box): python tools/merge_probe.py"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
psm = importlib.import_module("prismarine-core_amd")
scenes = importlib.import_module("prismarine-core_amd.scenes")
scene = scenes.sponza_like()
w, h = 1920, 1080
ctx = psm.Context(0)
th = psm.TriangleHierarchy(ctx)
th.allocate(scene["tris"].shape[0])
th.loadTriangles(scene["tris"], scene["normals"], scene["mats"])
th.build()
ms = psm.MaterialSet()
for m in scene["materials"]:
    ms.addSubmat(m)
cam = scenes.camera_matrices(scene["eye"], scene["view"], w, h)
rt = psm.Pipeline(ctx, seed=3)
rt.resizeBuffers(w, h); rt.resize(w, h)
rt.applyMaterials(ms)


def timed_traverse(rays, mode="whole", reps=3):
    rt.setTraverseMode(mode)
    best = 1e9
    for _ in range(reps):
        rt.upload_rays(rays)
        ctx.stats_enable(True, False); ctx.stats_reset()
        rt.intersection(th, force=True)
        ctx.sync()
        best = min(best, ctx.stats().traverse_ms)
    ctx.stats_enable(False, False)
    return best


for rnd in (0, 1, 2):
    queues = []
    for f in range(4):
        rt.camera_matrices(cam[0], cam[1], time=1000 + 17 * f)
        for k in range(rnd):
            rt.intersection(th)
            rt.shade(time=500 + k + 31 * f)
        rt.getRayCount()
        queues.append(rt.download_rays())
    limit = 4 * w * h
    F = 4
    while sum(q.shape[0] for q in queues[:F]) > limit:
        F -= 1
    for mode in ("whole", "adaptive"):
        alone = [timed_traverse(q, mode) for q in queues[:F]]
        merged = timed_traverse(np.concatenate(queues[:F]), mode)
        n = sum(q.shape[0] for q in queues[:F])
        print("round %d, %d frames, %8d rays, %-8s: alone %s = %.3f ms, merged %.3f ms (%.1f %%), %.3f -> %.3f ns/ray" % (
            rnd, F, n, mode, " + ".join("%.3f" % a for a in alone), sum(alone), merged, 100.0 * (merged / sum(alone) - 1.0),
            sum(alone) * 1e6 / n, merged * 1e6 / n))
