#!/usr/bin/env python3
"""Sort alone: HIP-event time of psm_sort_u64_u32 for every implementation (psm_sort_set_algorithm) at the key
counts of C3 and C5 -- uniform random 61-bit keys and the Morton codes of the two bench scenes (S-sponza-like, S-stress:
what the build actually sorts; the hybrid sort's chunks depend on how the keys spread over its sixteen-bit bins).
usage (GPU box): python3 tools/sort_bench.py [--no-stress] [--no-uniform] [--algos 0,2]   (PSM_SORT_TUNE="S_small,S_large,threads" for radix_local's shape)"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
psm = importlib.import_module("prismarine-core_amd")
scenes = importlib.import_module("prismarine-core_amd.scenes")
NAMES = {0: "three-kernel x8", 1: "onesweep", 2: "hybrid"}
BYTES = {0: 256, 1: 200, 2: 200}   # SURVEY 8(d) prices every sort at 200 B/key; the three-kernel passes move 256


def morton_keys(ctx, scene):
    th = psm.TriangleHierarchy(ctx)
    th.allocate(scene["tris"].shape[0])
    th.loadTriangles(scene["tris"], scene["normals"], scene["mats"])
    th.stage("bounds", None)
    th.stage("morton")
    n = th.info().leaf_count
    keys = th.download(psm.BVH_KEYS, np.uint64, n)
    th.close()
    return keys


def bench(ctx, rs, label, keys):
    n = keys.shape[0]
    vals = np.arange(n, dtype=np.uint32)
    hk, hv = ctx.buf_alloc(n * 8), ctx.buf_alloc(n * 4)
    ref = None
    for algo in ALGOS:
        rs.setAlgorithm(algo)
        times = []
        for rep in range(8):
            ctx.buf_upload(hk, keys); ctx.buf_upload(hv, vals)
            ctx.stats_enable(True, False); ctx.stats_reset()
            rs.sort(hk, hv, n)
            ctx.sync()
            st = ctx.stats()
            if rep:
                times.append(st.sort_ms)
        ctx.stats_enable(False, False)
        gk = ctx.buf_download(hk, np.uint64, n)
        gv = ctx.buf_download(hv, np.uint32, n)
        if ref is None:
            ref = (gk, gv)
            assert (gk[1:] >= gk[:-1]).all()
        else:
            assert np.array_equal(gk, ref[0]) and np.array_equal(gv, ref[1]), "algorithm %d differs from algorithm 0" % algo
        best, med = min(times), sorted(times)[len(times) // 2]
        print("%-22s n %9d  %-16s best %.4f ms  median %.4f ms  (%.0f Mkeys/s, %.0f GB/s at %d B/key)%s" % (
            label, n, NAMES[algo], best, med, n / best / 1e3, n * BYTES[algo] / best / 1e6, BYTES[algo],
            "  [fell back: a chunk overflowed]" if rs.getAlgorithm() == (2, 0) else ""), flush=True)
    rs.setAlgorithm(2)
    ctx.buf_free(hk); ctx.buf_free(hv)


ALGOS = [int(x) for x in sys.argv[sys.argv.index('--algos') + 1].split(',')] if '--algos' in sys.argv else [0, 1, 2]
ctx = psm.Context(0)
rs = psm.RadixSort(ctx)
print("# PSM_SORT_TUNE=%s" % os.environ.get("PSM_SORT_TUNE", "(default 1024,2048,1024,4096,4096,512)"))
for n in (() if "--no-uniform" in sys.argv else (262267, 2_000_000, 9_999_616)):
    rng = np.random.RandomState(1)
    keys = (rng.randint(0, 2 ** 62, size=n, dtype=np.int64).astype(np.uint64)) >> np.uint64(1)
    bench(ctx, rs, "uniform 61-bit", keys)
bench(ctx, rs, "Morton, S-sponza-like", morton_keys(ctx, scenes.sponza_like()))
if "--no-stress" not in sys.argv:
    bench(ctx, rs, "Morton, S-stress", morton_keys(ctx, scenes.stress()))
