#!/usr/bin/env python3
"""Prototype: S frames in flight on S contexts (streams) of one GPU vs the same frames one after another."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
psm = importlib.import_module("prismarine-core_amd")
scenes = importlib.import_module("prismarine-core_amd.scenes")
W, H = 1920, 1080
S = int(os.environ.get("S", 4))
FR = int(os.environ.get("FRAMES", 16))
sc = scenes.sponza_like()
cam = scenes.camera_matrices(sc["eye"], sc["view"], W, H)
P = []
for s in range(S):
    ctx = psm.Context(0)
    th = psm.TriangleHierarchy(ctx); th.allocate(sc["tris"].shape[0]); th.loadTriangles(sc["tris"], sc["normals"], sc["mats"])
    ms = psm.MaterialSet()
    for m in sc["materials"]: ms.addSubmat(m)
    rt = psm.Pipeline(ctx, seed=1000 + s); rt.resizeBuffers(W, H); rt.resize(W, H)
    rt.applyMaterials(ms)
    P.append((ctx, th, ms, rt))

def start(p):
    ctx, th, ms, rt = p
    th.markDirty(); th.build()
    rt.camera_matrices(cam[0], cam[1])

def round_(p):
    """one bounce round; False when the frame is over"""
    ctx, th, ms, rt = p
    if rt.getRayCount() <= 0:
        return False
    rt.intersection(th); rt.applyMaterials(ms); rt.shade(reload=False)
    return True

def sequential(frames):
    p = P[0]
    rays = 0
    for f in range(frames):
        start(p)
        for j in range(16):
            rays += max(p[3].getRayCount(), 0)
            if not round_(p): break
        p[3].sample()
    p[0].sync()
    return rays

def concurrent(frames):
    rays = 0
    for f0 in range(0, frames, S):
        live = list(P[: min(S, frames - f0)])
        for p in live: start(p)
        for j in range(16):
            nxt = []
            for p in live:
                p[3]._reload() if j else None
                rays += max(p[3].getRayCount(), 0)
                if round_(p): nxt.append(p)
            live = nxt
            if not live: break
        for p in P[: min(S, frames - f0)]: p[3].sample()
    for p in P: p[0].sync()
    return rays

for name, fn in (("sequential", sequential), ("concurrent S=%d" % S, concurrent)):
    fn(S)  # warm-up
    t0 = time.perf_counter(); rays = fn(FR); dt = time.perf_counter() - t0
    print("%s: %d frames %.2f ms/frame  %.1f Mrays/s" % (name, FR, dt * 1e3 / FR, rays / dt / 1e6))
