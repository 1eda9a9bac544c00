#!/usr/bin/env python3
"""Sort alone: HIP-event time of psm_sort_u64_u32 for both implementations (psm_sort_set_algorithm) at the key
counts of C3 and C5, Morton-like keys. usage (GPU box): python tools/sort_bench.py"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
psm = importlib.import_module("prismarine-core_amd")
ctx = psm.Context(0)
rs = psm.RadixSort(ctx)
for n in (262267, 2_000_000, 9_999_616):
    rng = np.random.RandomState(1)
    keys = (rng.randint(0, 2 ** 62, size=n, dtype=np.int64).astype(np.uint64)) >> np.uint64(1)
    vals = np.arange(n, dtype=np.uint32)
    hk, hv = ctx.buf_alloc(n * 8), ctx.buf_alloc(n * 4)
    for algo in (0, 1):
        rs.setAlgorithm(algo)
        best = 1e9
        for rep in range(6):
            ctx.buf_upload(hk, keys); ctx.buf_upload(hv, vals)
            ctx.stats_enable(True, False); ctx.stats_reset()
            rs.sort(hk, hv, n)
            ctx.sync()
            st = ctx.stats()
            if rep:
                best = min(best, st.sort_ms)
        ctx.stats_enable(False, False)
        gk = ctx.buf_download(hk, np.uint64, n)
        assert (gk[1:] >= gk[:-1]).all()
        print("n %9d  %-20s %.4f ms  (%.0f Mkeys/s, %.0f GB/s at %d B/key)" % (n, ["three-kernel", "onesweep"][algo], best, n / best / 1e3,
                                                                              n * (200 if algo == 1 else 256) / best / 1e6, 200 if algo == 1 else 256))
    rs.setAlgorithm(0)
    ctx.buf_free(hk); ctx.buf_free(hv)
