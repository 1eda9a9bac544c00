#!/usr/bin/env python3
"""What a frame on its own pays for asking the host after every bounce round (study, MI355X).

The reference's loop (Viewer.cpp:303-309) reads the ray count back before every round; psm_lanes_render does the same through a
pinned word and an event, so between the end of a round's scan kernel and the start of the next traversal lies one host round
trip. This script times C3 frames one at a time in two ways:
  polled   the product's schedule (FrameBatch with one lane: psm_lanes_render)
  blind    the same frame with its rounds queued WITHOUT waiting for the counts: every launch is sized for the largest count a
           round can have and the kernels clamp to the queue's real total, which they read on the device (rt_traverse / rt_shade:
           min(nrays, bases[nb])). The number of rounds is taken from the polled run (a real schedule would have to undo a round
           that turns out to be void).
and checks that both accumulate the same image. usage: tools/spec_rounds.py [frames] [width height]"""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
    psm = importlib.import_module("prismarine-core_amd")
    scenes = importlib.import_module("prismarine-core_amd.scenes")
    sc = scenes.sponza_like()
    cam = scenes.camera_matrices(sc["eye"], sc["view"], w, h)
    n = sc["tris"].shape[0]

    batch = psm.FrameBatch(1, w, h, device=0, seed=1000)
    batch.allocate(n)
    batch.loadTriangles(sc["tris"], sc["normals"], sc["mats"])
    ms = psm.MaterialSet()
    for m in sc["materials"]:
        ms.addSubmat(m)
    batch.applyMaterials(ms)
    seeds = batch.frame_seeds(frames + 4)
    batch.trace(cam[0], cam[1], seeds[:4])          # warm-up (graph capture, allocations)
    batch.clearSampler()
    batch.sync()
    t0 = time.perf_counter()
    res = batch.trace(cam[0], cam[1], seeds[4:])
    batch.sync()
    polled = (time.perf_counter() - t0) / frames
    rounds = max(r[0] for r in res)
    img_polled = batch.snapHdr()
    print("polled: %.3f ms per frame, %d rounds per frame, %.0f rays per frame" % (polled * 1e3, rounds, sum(r[1] for r in res) / frames))

    ctx = psm.Context(0)
    th = psm.TriangleHierarchy(ctx)
    th.allocate(n)
    th.loadTriangles(sc["tris"], sc["normals"], sc["mats"])
    rt = psm.Pipeline(ctx, seed=1000)
    rt.resizeBuffers(w, h)
    rt.resize(w, h)
    rt.applyMaterials(ms)
    lib, C = psm.lib(), psm.C
    ci = np.ascontiguousarray(cam[0], np.float32).reshape(16)
    pi = np.ascontiguousarray(cam[1], np.float32).reshape(16)

    def frame(seed_state):
        # the draws of psm_lanes_render's frame: one for the camera, one per round (the CRT rand() stand-in)
        st = [seed_state]

        def draw():
            st[0] = (st[0] * 214013 + 2531011) & 0xFFFFFFFF
            return (st[0] >> 16) & 0x7FFF
        th.markDirty()
        th.build()
        ctx.check(lib.psm_rt_camera(rt._h, psm._p(ci), psm._p(pi), C.c_uint32(draw())), "camera")
        for _ in range(rounds):
            rt.set_ray_count(w * h)      # an upper bound: the kernels clamp to the queue's total on the device
            rt._obj = th
            ctx.check(lib.psm_rt_traverse(rt._h, th._h), "traverse")
            ctx.check(lib.psm_rt_shade(rt._h, th._h, C.c_uint32(draw())), "shade")
        rt.sample()

    for f in range(4):
        frame(int(seeds[f]))
    rt.clearSampler()
    ctx.sync()
    t0 = time.perf_counter()
    for f in range(frames):
        frame(int(seeds[4 + f]))
    ctx.sync()
    blind = (time.perf_counter() - t0) / frames
    img_blind = rt.snapHdr()
    same = np.array_equal(img_polled[..., 3], img_blind[..., 3]) and np.allclose(img_polled[..., :3], img_blind[..., :3], rtol=1e-4, atol=1e-5)
    print("blind:  %.3f ms per frame (%.1f %% of polled); images %s" % (blind * 1e3, 100.0 * blind / polled, "equal" if same else "DIFFER"))


if __name__ == "__main__":
    main()
