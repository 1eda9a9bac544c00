#!/usr/bin/env python3
"""Traversal micro-benchmark: primary + second-round rays of one Sponza-class frame, traversal only."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
psm = importlib.import_module("prismarine-core_amd")
scenes = importlib.import_module("prismarine-core_amd.scenes")
W, H = 1920, 1080
sc = scenes.sponza_like()
ctx = psm.Context(0)
th = psm.TriangleHierarchy(ctx); th.allocate(sc["tris"].shape[0]); th.loadTriangles(sc["tris"], sc["normals"], sc["mats"]); th.build()
ms = psm.MaterialSet()
for m in sc["materials"]: ms.addSubmat(m)
rt = psm.Pipeline(ctx, seed=1000); rt.resizeBuffers(W, H); rt.resize(W, H)
cam = scenes.camera_matrices(sc["eye"], sc["view"], W, H)
rt.camera_matrices(cam[0], cam[1]); rt.applyMaterials(ms)
sets = [rt.download_rays()]
rt.intersection(th); rt.shade(); sets.append(rt.download_rays())
reps = int(os.environ.get("REPS", "5"))
for name, rays in zip(("primary", "round2"), sets):
    rt.upload_rays(rays)
    rt.resetHits(); rt.intersection(th, force=True); ctx.sync()
    ctx.stats_enable(True, False); ctx.stats_reset()
    for _ in range(reps): rt.resetHits(); rt.intersection(th, force=True)
    st = ctx.stats()
    ms_ = st.traverse_ms / reps
    print("%s %s rays=%d  %.3f ms  %.1f Mrays/s" % (os.environ.get("TAG", ""), name, len(rays), ms_, len(rays) / ms_ / 1e3))
    ctx.stats_enable(False, False)

# coherence experiment: the same round-2 rays, reordered on the host
if os.environ.get("SORT_EXP"):
    rays = sets[1]
    o = rays["origin"]; d = rays["direct"]
    lo, hi = o.min(0), o.max(0)
    q = np.clip(((o - lo) / (hi - lo + 1e-9) * 1023).astype(np.uint64), 0, 1023)
    def part(x):
        x = (x | (x << 16)) & 0x030000FF; x = (x | (x << 8)) & 0x0300F00F; x = (x | (x << 4)) & 0x030C30C3; x = (x | (x << 2)) & 0x09249249
        return x
    mort = part(q[:, 0]) | (part(q[:, 1]) << 1) | (part(q[:, 2]) << 2)
    octant = ((d[:, 0] > 0).astype(np.uint64) | ((d[:, 1] > 0).astype(np.uint64) << 1) | ((d[:, 2] > 0).astype(np.uint64) << 2))
    typ = ((rays["bitfield"] >> 1) & 3).astype(np.uint64)
    for name, key in (("morton", mort), ("octant+morton", (octant << 30) | mort), ("type+octant+morton", (typ << 33) | (octant << 30) | mort),
                      ("random", np.random.RandomState(0).permutation(len(rays)).astype(np.uint64))):
        order = np.argsort(key, kind="stable")
        rt.upload_rays(rays[order])
        rt.resetHits(); rt.intersection(th, force=True); ctx.sync()
        ctx.stats_enable(True, False); ctx.stats_reset()
        for _ in range(reps): rt.resetHits(); rt.intersection(th, force=True)
        st = ctx.stats(); ms_ = st.traverse_ms / reps
        print("%s round2 sorted by %s: %.3f ms  %.1f Mrays/s" % (os.environ.get("TAG", ""), name, ms_, len(rays) / ms_ / 1e3))
        ctx.stats_enable(False, False)

if os.environ.get("SIZE_EXP"):
    rays = sets[1]
    for frac in (1, 2, 4, 8, 16, 32, 64, 256):
        sub = rays[: len(rays) // frac]
        rt.upload_rays(sub)
        rt.resetHits(); rt.intersection(th, force=True); ctx.sync()
        ctx.stats_enable(True, False); ctx.stats_reset()
        for _ in range(reps): rt.resetHits(); rt.intersection(th, force=True)
        st = ctx.stats(); ms_ = st.traverse_ms / reps
        print("%s round2 first 1/%d (%d rays): %.3f ms  %.1f Mrays/s" % (os.environ.get("TAG", ""), frac, len(sub), ms_, len(sub) / ms_ / 1e3))
        ctx.stats_enable(False, False)

if os.environ.get("TINY_EXP"):
    rays = sets[1]
    for cnt_ in (64, 1024, 8192):
        sub = rays[:cnt_]
        rt.upload_rays(sub)
        rt.resetHits(); rt.intersection(th, force=True); ctx.sync()
        ctx.stats_enable(True, False); ctx.stats_reset()
        for _ in range(20): rt.resetHits(); rt.intersection(th, force=True)
        st = ctx.stats(); ms_ = st.traverse_ms / 20
        print("%s tiny %d rays: %.4f ms" % (os.environ.get("TAG", ""), cnt_, ms_))
        ctx.stats_enable(False, False)
