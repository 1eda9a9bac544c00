#!/usr/bin/env python3
"""GPU study (uses the oracle: lives under tests/): the solo gear against the oracle, ray by ray, for solo_max 0..4 --
how many rays differ, which lanes, which fields. usage: python tests/studies/solo_debug.py"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle as O
psm = importlib.import_module("prismarine-core_amd")
scenes = importlib.import_module("prismarine-core_amd.scenes")

def bits(a): return np.ascontiguousarray(a).view(np.uint32)

ctx = psm.Context(0)
for name, sc, w, h in (("cornell", scenes.cornell(), 1280, 720), ("sponza 20011", scenes.sponza_like(n_tris=20011), 320, 180)):
    th = psm.TriangleHierarchy(ctx); th.allocate(sc["tris"].shape[0]); th.loadTriangles(sc["tris"], sc["normals"], sc["mats"]); th.build()
    ob = O.build_scene(sc["tris"])
    cam = scenes.camera_matrices(sc["eye"], sc["view"], w, h)
    for counting in (False, True):
        for solo in range(5):
            rt = psm.Pipeline(ctx); rt.resizeBuffers(w, h); rt.resize(w, h)
            rt.setTraverseSolo(solo)
            rt.camera_matrices(cam[0], cam[1], time=4242)
            rays = rt.download_rays()
            ctx.stats_enable(False, counting); ctx.stats_reset()
            rt.intersection(th)
            st = ctx.stats()
            ctx.stats_enable(False, False)
            gh, gc = rt.download_hits(w * h)
            oh, oc, octr = O.traverse(ob["nodes"], sc["tris"], ob["M"], rays["origin"], rays["direct"], 8)
            bad = (gc != oc) | (gh["tri"][:, 0] != oh["tri"][:, 0]) | (bits(gh["t"][:, 0]) != bits(oh["t"][:, 0]))
            idx = np.nonzero(bad)[0]
            print("%s counting=%d solo=%d: %d of %d rays differ; V %d/%d T %d/%d" % (name, counting, solo, len(idx), w * h, st.node_visits, octr.node_visits, st.tri_tests, octr.tri_tests), flush=True)
            for i in idx[:6]:
                print("   ray %d lane %d wave %d: count %d/%d tri %s / %s  t %r / %r" % (i, i % 64, i // 64, gc[i], oc[i], gh["tri"][i, :3], oh["tri"][i, :3], gh["t"][i, 0], oh["t"][i, 0]))
            if len(idx):
                w_ = idx // 64
                print("   waves with differences: %d; rays per such wave: %s; lanes histogram (first 16): %s" % (len(np.unique(w_)), np.bincount(np.bincount(w_)[np.unique(w_)])[:8], np.bincount(idx % 64, minlength=64)[:16]))
            rt.close()
    th.close()
