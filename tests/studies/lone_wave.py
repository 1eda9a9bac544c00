#!/usr/bin/env python3
"""GPU study (uses the oracle for per-ray step counts: lives under tests/): what a traversal wave's step costs as a function
of how many waves share its SIMD and of the scene's size -- the chain a lone wave walks (a frame alone, a tile's launches and
every round's tail end with such waves) against the issue-bound step of a full chip. One launch of the counting single-launch
kernel per row; cycles per wave-step = shader-clock ticks summed over the launch's waves / their wave-steps (the kernel's own
s_memtime stamps), so launch overhead is not in it.
usage: python tests/studies/lone_wave.py   (on the GPU box, from the repo root)"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle as O

psm = importlib.import_module("prismarine-core_amd")
scenes = importlib.import_module("prismarine-core_amd.scenes")


def rays_for(sc, n, rng):
    tri = sc["tris"]
    tid = rng.randint(0, tri.shape[0], n)
    wgt = rng.dirichlet((1, 1, 1), n).astype(np.float32)
    target = (tri[tid] * wgt[:, :, None]).sum(1)
    ext = (tri.reshape(-1, 3).max(0) - tri.reshape(-1, 3).min(0)).max()
    origin = (target + rng.normal(0, 1, (n, 3)) * 0.15 * ext).astype(np.float32)
    rays = np.zeros(n, psm.RAY_DT)
    rays["origin"], rays["direct"], rays["color"] = origin, (target - origin).astype(np.float32), 1.0
    rays["bitfield"] = 1 | (3 << 8)
    rays["texel"] = np.arange(n) % 64
    rays["pkey"] = np.arange(n)
    return rays


def main():
    ctx = psm.Context(0)
    rng = np.random.RandomState(5)
    cases = [("cornell, 32 triangles", scenes.cornell(open_top=True), 20000),
             ("sponza-like, 6 007 triangles", scenes.sponza_like(n_tris=6007), 40000),
             ("sponza-like, 262 267 triangles (C3)", scenes.sponza_like(), 200000)]
    print("%-38s %8s %8s %10s %12s %12s" % ("scene", "waves", "rays", "max steps", "cycles/step", "ns/step"))
    for name, sc, nprobe in cases:
        th = psm.TriangleHierarchy(ctx)
        th.allocate(sc["tris"].shape[0])
        th.loadTriangles(sc["tris"], sc["normals"], sc["mats"])
        th.build()
        ob = O.build_scene(sc["tris"])
        probe = rays_for(sc, nprobe, rng)
        v, _ = O.traverse_visits(ob["nodes"], sc["tris"], ob["M"], probe["origin"], probe["direct"], 8)
        order = np.argsort(-v.astype(np.int64), kind="stable")
        rt = psm.Pipeline(ctx)
        rt.resizeBuffers(1024, 1024)
        rt.setTraverseMode(psm.TRAVERSE_WHOLE if hasattr(psm, "TRAVERSE_WHOLE") else 0)
        for waves in (1, 4, 64, 1024, 4096, 8192):
            n = waves * 64
            idx = order[np.arange(n) % min(len(order), max(n, 64))] if n <= len(order) else order[np.arange(n) % len(order)]
            rays = probe[idx].copy()
            rays["pkey"] = np.arange(n)
            best = None
            for rep in range(3):
                rt.upload_rays(rays)
                ctx.stats_enable(False, True)
                ctx.stats_reset()
                rt.intersection(th)
                ctx.sync() if hasattr(ctx, "sync") else None
                st = ctx.stats()
                ctx.stats_enable(False, False)
                if st.wave_steps:
                    cps = st.wave_clock_ticks / st.wave_steps
                    best = cps if best is None else min(best, cps)
            print("%-38s %8d %8d %10d %12.0f %12.0f" % (name, waves, n, int(v[idx].max()), best, best / 2.35), flush=True)
        rt.close()
        th.close()


if __name__ == "__main__":
    main()
