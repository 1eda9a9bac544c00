#!/usr/bin/env python3
"""GPU study: what do the waves that END LAST in a round do? Needs the experiment build with the wave log
(csrc: -DPSM_EXP_WAVELOG=1 -> variants/libpsm_wavelog.so; PSM_HIP_LIB points at it): every wave of a single-launch traversal
writes when it started and ended (s_memrealtime, 100 MHz), when it went into the solo gear, and how many wave-steps it spent
with 2, 3-4, 5-8, 9-16 and 17+ lanes with work. One frame alone (AUTO -> one launch per round), C3, 1920x1080.
usage: PSM_HIP_LIB=.../libpsm_wavelog.so python tests/studies/wave_log.py"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np

psm = importlib.import_module("prismarine-core_amd")
scenes = importlib.import_module("prismarine-core_amd.scenes")

W, H = 1920, 1080
sc = scenes.sponza_like()
ctx = psm.Context(0)
th = psm.TriangleHierarchy(ctx); th.allocate(sc["tris"].shape[0]); th.loadTriangles(sc["tris"], sc["normals"], sc["mats"]); th.build()
ms = psm.MaterialSet()
for m in sc["materials"]:
    ms.addSubmat(m)
rt = psm.Pipeline(ctx, seed=1000); rt.resizeBuffers(W, H); rt.resize(W, H)
cam = scenes.camera_matrices(sc["eye"], sc["view"], W, H)
nw_max = (4 * W * H + 63) // 64 + 1024
buf = ctx.buf_alloc(nw_max * 32)
ptr, _ = ctx.buf_ptr(buf)
os.environ["PSM_EXP_WAVELOG_PTR"] = hex(ptr)
for solo in (1, 0):
    rt.setTraverseSolo(solo)
    rt.setSeed(1000)
    rt.camera_matrices(cam[0], cam[1]); rt.applyMaterials(ms)
    print("solo_max %d" % solo)
    for rnd in range(16):
        n = rt.getRayCount()
        if n <= 0:
            break
        nw = (n + 63) // 64
        ctx.buf_upload(buf, np.zeros(nw_max * 8, np.uint32))
        rt.intersection(th); ctx.sync()
        rec = ctx.buf_download(buf, np.uint32, nw * 8).reshape(nw, 8).astype(np.int64)
        rec = rec[rec[:, 1] != 0]                       # (blocks of the launch grid's padding beyond the rays' waves write nothing here)
        t0, t1, ts = rec[:, 0], rec[:, 1], rec[:, 2]
        base = t0.min()
        end = (t1 - base) % (1 << 32) / 100.0          # us after the first wave's start
        start = (t0 - base) % (1 << 32) / 100.0
        solo_at = (ts - base) % (1 << 32) / 100.0
        total = end.max()
        order = np.argsort(-end)
        print(" round %d: %d rays, %d waves, launch %.0f us; waves still running at 50 / 60 / 70 / 80 / 90 %% of it: %s" % (
            rnd, n, nw, total, [int((end > f * total).sum()) for f in (0.5, 0.6, 0.7, 0.8, 0.9)]))
        last = order[:16]
        for i in last[:8]:
            print("   wave %6d: start %6.0f us, into the gear at %6.0f, end %6.0f; wave-steps at 2 / 3-4 / 5-8 / 9-16 / 17+ lanes: %s" % (
                i, start[i], solo_at[i], end[i], rec[i, 3:8].tolist()))
        # over the last 200 waves: where did their time after 50 % of the launch go?
        tail = order[:200]
        print("   the 200 last waves: mean start %.0f us, mean gear entry %.0f, mean end %.0f; mean steps %s" % (
            start[tail].mean(), solo_at[tail].mean(), end[tail].mean(), rec[tail, 3:8].mean(0).round(1).tolist()))
        rt.shade()
rt.close(); th.close()
