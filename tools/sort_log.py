#!/usr/bin/env python3
"""GPU study: where does a radix_local workgroup spend its time? Needs the experiment build with the phase log
(make -C prismarine-core_amd/csrc sortlog -> variants/libpsm_sortlog.so; PSM_HIP_LIB points at it): thread 0 of every workgroup
writes s_memtime at its phase boundaries. Morton codes of the two bench scenes through psm_sort_u64_u32 (hybrid).
usage: PSM_HIP_LIB=$PWD/prismarine-core_amd/csrc/variants/libpsm_sortlog.so python3 tools/sort_log.py [--no-stress]"""
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
psm = importlib.import_module("prismarine-core_amd")
scenes = importlib.import_module("prismarine-core_amd.scenes")


def morton_keys(ctx, scene):
    th = psm.TriangleHierarchy(ctx)
    th.allocate(scene["tris"].shape[0])
    th.loadTriangles(scene["tris"], scene["normals"], scene["mats"])
    th.stage("bounds", None); th.stage("morton")
    n = th.info().leaf_count
    keys = th.download(psm.BVH_KEYS, np.uint64, n)
    th.close()
    return keys


ctx = psm.Context(0)
rs = psm.RadixSort(ctx)
lib = psm.lib()
lib.psm_sortlog_set.argtypes = [C.c_void_p]
for name, sc in (("S-sponza-like", scenes.sponza_like()),) + (() if "--no-stress" in sys.argv else (("S-stress", scenes.stress()),)):
    keys = morton_keys(ctx, sc)
    n = keys.shape[0]
    S = 1024 if n <= (1 << 19) else 2048
    if os.environ.get("PSM_SORT_TUNE"):
        t = [int(v) for v in os.environ["PSM_SORT_TUNE"].split(",")]
        S = t[0] if n <= (1 << 19) else t[1]
    nwg = (n + S - 1) // S
    hk, hv, hl = ctx.buf_alloc(n * 8), ctx.buf_alloc(n * 4), ctx.buf_alloc(nwg * 32 * 8)
    for rep in range(3):
        ctx.buf_upload(hk, keys); ctx.buf_upload(hv, np.arange(n, dtype=np.uint32))
        ctx.buf_upload(hl, np.zeros(nwg * 32, np.uint64))
        assert lib.psm_sortlog_set(C.c_void_p(ctx.buf_ptr(hl)[0])) == 0
        rs.sort(hk, hv, n); ctx.sync()
    lib.psm_sortlog_set(None)
    log = ctx.buf_download(hl, np.uint64, nwg * 32).reshape(nwg, 32).astype(np.int64)
    log = log[log[:, 31] != 0]
    size = log[:, 30]
    t = log[:, :30]
    nst = (t != 0).sum(1)                      # stamps written: 3 + 3 per pass
    passes = (nst - 3) // 3
    d = lambda a, b: (t[:, b] - t[:, a])
    print("%s: %d keys, %d workgroups with a chunk; chunk size mean %.0f max %d; passes mean %.2f (min %d max %d)" % (
        name, n, log.shape[0], size.mean(), size.max(), passes.mean(), passes.min(), passes.max()))
    print("  cycles (s_memtime), mean over workgroups: window load + boundaries %.0f, keys to registers + diff %.0f" % (d(0, 1).mean(), d(1, 2).mean()))
    rank = np.array([d(2 + 3 * p if p else 2, 3 + 3 * p)[passes > p].mean() for p in range(int(passes.max()))])
    pre = np.array([d(3 + 3 * p, 4 + 3 * p)[passes > p].mean() for p in range(int(passes.max()))])
    sca = np.array([d(4 + 3 * p, 5 + 3 * p)[passes > p].mean() for p in range(int(passes.max()))])
    print("  per pass: (reload +) rank %s\n            prefix %s\n            scatter %s" % (rank.round(0), pre.round(0), sca.round(0)))
    last = np.array([t[i, nst[i] - 1] for i in range(t.shape[0])])
    print("  write-out %.0f; whole workgroup %.0f cycles" % ((log[:, 31] - last).mean(), (log[:, 31] - t[:, 0]).mean()))
    ctx.buf_free(hk); ctx.buf_free(hv); ctx.buf_free(hl)
