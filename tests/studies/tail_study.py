#!/usr/bin/env python3
"""Per-round traversal time vs work and vs the longest ray: is a launch bounded by throughput or by the
dependent chain of its slowest wave? Lives under tests/ because it uses the oracle (to count per-ray steps):
only tests/, smoke() and bench.py's cpu_baseline leg may touch oracle/.
usage (on a GPU box): python tests/studies/tail_study.py"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import oracle as O
psm = importlib.import_module("prismarine-core_amd")
scenes = importlib.import_module("prismarine-core_amd.scenes")
W, H = int(os.environ.get("W", 1920)), int(os.environ.get("H", 1080))
sc = scenes.sponza_like()
ctx = psm.Context(0)
th = psm.TriangleHierarchy(ctx); th.allocate(sc["tris"].shape[0]); th.loadTriangles(sc["tris"], sc["normals"], sc["mats"]); th.build()
ms = psm.MaterialSet()
for m in sc["materials"]: ms.addSubmat(m)
rt = psm.Pipeline(ctx, seed=1000); rt.resizeBuffers(W, H); rt.resize(W, H)
cam = scenes.camera_matrices(sc["eye"], sc["view"], W, H)
ob = O.build_scene(sc["tris"])
rt.camera_matrices(cam[0], cam[1]); rt.applyMaterials(ms)
reps = 3
print("round rays gpu_ms  Mrays/s  visits_total  mean  p99  max  wave_max_mean  est_lone_wave_us_per_step")
for rnd in range(16):
    n = rt.getRayCount()
    if n <= 0: break
    rays = rt.download_rays()
    rt.resetHits(); rt.intersection(th, force=True); ctx.sync()
    ctx.stats_enable(True, False); ctx.stats_reset()
    for _ in range(reps): rt.resetHits(); rt.intersection(th, force=True)
    t = ctx.stats().traverse_ms / reps
    ctx.stats_enable(False, False)
    v, tt = O.traverse_visits(ob["nodes"], sc["tris"], ob["M"], rays["origin"], rays["direct"], 16)
    steps = v.astype(np.int64)
    pad = (-len(steps)) % 64
    wv = np.concatenate([steps, np.zeros(pad, np.int64)]).reshape(-1, 64).max(1)
    print("%2d %8d %.3f %8.1f %12d %6.1f %5d %5d %8.1f   %.3f" % (rnd, n, t, n / t / 1e3, steps.sum(), steps.mean(), np.percentile(steps, 99), steps.max(), wv.mean(), t * 1e3 / max(steps.max(), 1)))
    rt.resetHits(); rt.intersection(th, force=True)
    rt.shade()
