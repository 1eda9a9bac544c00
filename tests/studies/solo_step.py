#!/usr/bin/env python3
"""GPU study (uses the oracle for per-ray step counts: lives under tests/): what a wave-step costs when ONE lane of the wave has
a ray -- the lane-per-ray step (solo_max 0) against the solo gear (solo_max 1) -- as a function of how many such waves share
the chip. Every wave gets one of the longest rays of a probe set in lane 0 and 63 rays that miss the scene's box.
cycles per wave-step = shader-clock ticks summed over the launch's waves / their wave-steps (the kernel's own s_memtime stamps).
usage: python tests/studies/solo_step.py   (on the GPU box, from the repo root)"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "studies"))
import numpy as np
from oracle import oracle as O
from lone_wave import rays_for

psm = importlib.import_module("prismarine-core_amd")
scenes = importlib.import_module("prismarine-core_amd.scenes")


def main():
    ctx = psm.Context(0)
    rng = np.random.RandomState(5)
    cases = [("sponza-like, 6 007 triangles", scenes.sponza_like(n_tris=6007), 40000),
             ("sponza-like, 262 267 triangles (C3)", scenes.sponza_like(), 200000)]
    print("%-38s %5s %8s %10s %12s %12s" % ("scene", "solo", "waves", "max steps", "cycles/step", "ns/step"))
    for name, sc, nprobe in cases:
        th = psm.TriangleHierarchy(ctx)
        th.allocate(sc["tris"].shape[0])
        th.loadTriangles(sc["tris"], sc["normals"], sc["mats"])
        th.build()
        ob = O.build_scene(sc["tris"])
        probe = rays_for(sc, nprobe, rng)
        v, _ = O.traverse_visits(ob["nodes"], sc["tris"], ob["M"], probe["origin"], probe["direct"], 8)
        order = np.argsort(-v.astype(np.int64), kind="stable")
        ext = float(np.abs(sc["tris"]).max())
        rt = psm.Pipeline(ctx)
        rt.resizeBuffers(1024, 1024)
        rt.setTraverseMode("whole")
        for waves in (1, 64, 1024, 2048, 4096, 8192):
            n = waves * 64
            rays = np.zeros(n, psm.RAY_DT)
            rays["origin"] = (100.0 * ext, 100.0 * ext, 100.0 * ext)     # outside, pointing away: no node step at all
            rays["direct"] = (1.0, 1.0, 1.0)
            rays["color"] = 1.0
            pick = probe[order[np.arange(waves) % min(len(order), 4096)]]
            for f in ("origin", "direct"):
                rays[f][::64] = pick[f]
            rays["bitfield"] = 1 | (3 << 8)
            rays["texel"] = np.arange(n) % 64
            rays["pkey"] = np.arange(n)
            for solo in (0, 1):
                rt.setTraverseSolo(solo)
                best = None
                for rep in range(3):
                    rt.upload_rays(rays)
                    ctx.stats_enable(False, True)
                    ctx.stats_reset()
                    rt.intersection(th)
                    st = ctx.stats()
                    ctx.stats_enable(False, False)
                    if st.wave_steps:
                        cps = st.wave_clock_ticks / st.wave_steps
                        best = cps if best is None else min(best, cps)
                    if os.environ.get("PSM_SOLO_DBG") and solo and rep == 2:   # the instrumented experiment build (PSM_SOLO_DEBUG=2)
                        print("      per node visit: load wait %.0f, slab + ballot %.0f, scalar step %.0f ticks (visits %d, wave-steps %d)" % (
                            st.stack_drops / st.node_visits, st.iter_caps / st.node_visits, st.baked_drops / st.node_visits, st.node_visits, st.wave_steps))
                print("%-38s %5d %8d %10d %12.0f %12.0f" % (name, solo, waves, int(v[order[:min(waves, 4096)]].max()), best, best / 2.35), flush=True)
        rt.close()
        th.close()


if __name__ == "__main__":
    main()
