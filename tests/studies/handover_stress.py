"""GPU study (uses the oracle: lives under tests/): the hand-over schedules of the GROUP traversal launch, counting kernels,
repeated in one process -- the script that bisected the stale non-temporal loads of DESIGN.md 5.4.
usage: python tests/studies/handover_stress.py [adaptive|phased] [repetitions]   (on the GPU box, from the repo root)"""
import sys, importlib, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
psm = importlib.import_module("prismarine-core_amd")
scenes = importlib.import_module("prismarine-core_amd.scenes")
from oracle import oracle
import test_gpu_parity as T
ctx = psm.Context(0)
mode, kw = sys.argv[1] if len(sys.argv) > 1 else "adaptive", {"min_live": 16, "min_steps": 4, "final_rays": 64, "max_launches": 4}
if mode == "phased": kw = {"caps": [3, 9]}
rng = np.random.RandomState(11)
sc = [scenes.sponza_like(n_tris=6007), scenes.cornell(open_top=True), scenes.sponza_like(n_tris=20011)]
cap = max(s_["tris"].shape[0] for s_ in sc)
arena = psm.Arena(ctx, 3, cap)
ths, rts, rays_l, built = [], [], [], []
for k, s_ in enumerate(sc):
    th = psm.TriangleHierarchy(ctx); th.allocate(cap, arena, k)
    th.loadTriangles(s_["tris"], s_["normals"], s_["mats"]); th.build()
    ob = oracle.build_scene(s_["tris"])
    n = [30000, 777, 50001][k]
    tri = s_["tris"]; tid = rng.randint(0, tri.shape[0], n)
    wgt = rng.dirichlet((1, 1, 1), n).astype(np.float32)
    target = (tri[tid] * wgt[:, :, None]).sum(1)
    origin = (target + rng.normal(0, 1, (n, 3)) * 2.0 + np.array([0, 2, 0])).astype(np.float32)
    rays = np.zeros(n, psm.RAY_DT)
    rays["origin"], rays["direct"], rays["color"] = origin, (target - origin).astype(np.float32), 1.0
    rays["bitfield"] = 1 | (3 << 8); rays["texel"] = np.arange(n) % 100; rays["pkey"] = np.arange(n)
    rt = psm.Pipeline(ctx); rt.resizeBuffers(128, 128)
    T._select_schedule(rt, mode, kw); rt.upload_rays(rays)
    ths.append(th); rts.append(rt); rays_l.append(rays); built.append(ob)
orc = []
for k, s_ in enumerate(sc):
    oh, oc, ostat = oracle.traverse(built[k]["nodes"], s_["tris"], built[k]["M"], rays_l[k]["origin"], rays_l[k]["direct"], 8)
    v, _ = oracle.traverse_visits(built[k]["nodes"], s_["tris"], built[k]["M"], rays_l[k]["origin"], rays_l[k]["direct"], 8)
    orc.append((oh, oc, v))
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
    for k in range(3): rts[k].upload_rays(rays_l[k])
    ctx.stats_enable(False, True); ctx.stats_reset()
    psm.traverse_group(rts, ths)
    st = ctx.stats()
    for k in range(3):
        oh, oc, v = orc[k]
        gh, gc = rts[k].download_hits(rays_l[k].shape[0])
        bad = np.nonzero((gc != oc) | (gh["tri"][:, 0] != oh["tri"][:, 0]))[0]
        print(mode, "rep", rep, "pipeline", k, "mismatches", len(bad), bad[:12], "oracle steps of them", v[bad[:12]], "gpu t", gh["t"][bad[:6], 0], "gpu tri", gh["tri"][bad[:6], 0], "median steps all", int(np.median(v)), flush=True)
    print("V,T", st.node_visits, st.tri_tests, flush=True)
