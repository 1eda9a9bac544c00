"""Debug aid for tests/test_gpu_fuzz.py::test_fuzzed_frames_shade_like_the_oracle: runs one seed on the GPU, stops at the first round
whose hit chains differ from the oracle's and writes the rays, both answers and the scene to gpurun_out/fuzz_<seed>.npz."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fuzz as F   # noqa: E402
from oracle import oracle as O   # noqa: E402

psm = importlib.import_module("prismarine-core_amd")
scenes = importlib.import_module("prismarine-core_amd.scenes")
O.build()
seed = int(sys.argv[1])
ctx = psm.Context(0)
tris, _, _, tags = F.fuzz_case(seed, visible=True)
rng = np.random.RandomState(9000 + seed)
n = tris.shape[0]
nk = rng.randint(1, 7)
materials = []
for _ in range(nk):
    d = tuple(float(x) for x in rng.choice([0.0, 0.2, 0.73, 1.0, 1.5], 3)) if rng.rand() < 0.4 else tuple(float(x) for x in rng.uniform(0, 1, 3))
    rough = float(rng.choice([0.0, 0.05, 0.5, 0.9, 1.0]))
    metal = float(rng.choice([0.0, 0.0, 0.3, 1.0]))
    em = tuple(float(x) for x in rng.uniform(0, 8, 3)) if rng.rand() < 0.25 else (0.0, 0.0, 0.0)
    materials.append({"diffuse": d + (1.0,), "specular": (0.0, rough, metal, 0.0), "emissive": em + (1.0,)})
mats = rng.randint(0, nk, n).astype(np.int32)
kind = rng.randint(0, 3)
if kind == 0:
    normals = scenes.prepare_normals(tris)
else:
    raw = rng.normal(0, 1, (n, 3, 3)).astype(np.float32)
    if kind == 2:
        raw[rng.rand(n) < 0.5] = 0.0
    normals = scenes.prepare_normals(tris, raw)
lo, hi = tris.reshape(-1, 3).min(0), tris.reshape(-1, 3).max(0)
span = np.maximum(hi - lo, np.float32(1e-20))
centre = tris[rng.randint(0, n)].mean(0).astype(np.float32)
away = rng.normal(0, 1, 3)
away /= np.linalg.norm(away)
eye = (centre + away * np.linalg.norm(span) * rng.uniform(0.2, 2.0)).astype(np.float32)
w, h = int(rng.randint(17, 120)), int(rng.randint(9, 80))
th = psm.TriangleHierarchy(ctx)
th.allocate(n)
th.loadTriangles(tris, normals, mats)
th.build()
rt = psm.Pipeline(ctx)
rt.resizeBuffers(w, h)
rt.resize(w, h)
ms = psm.MaterialSet()
for mm in materials:
    ms.addSubmat(mm)
cam = scenes.camera_matrices(eye, centre, w, h)
ob = O.build_scene(tris)
marr = scenes.materials_array(materials)
cfg = O.make_cfg(w, h, material_count=len(marr))
lights = O.default_lights(1)
rt.camera_matrices(cam[0], cam[1], time=seed)
orays, ocoord, osum, oflag = O.camera(cfg, cam[0], cam[1], seed)
rt.applyMaterials(ms)
print("seed", seed, "n", n, "kept", ob["count"], w, h, tags)
for rnd in range(4):
    if orays.shape[0] < 32:
        break
    rt.intersection(th)
    oh, oc, _ = O.traverse(ob["nodes"], tris, ob["M"], orays["origin"], orays["direct"], 8)
    gh, gc = rt.download_hits(orays.shape[0])
    bad = np.nonzero(gc != oc)[0]
    for k in range(8):
        m = (oc > k) & (gc > k)
        bad = np.union1d(bad, np.nonzero(m & ((gh["tri"][:, k] != oh["tri"][:, k]) | (gh["t"][:, k].view(np.uint32) != oh["t"][:, k].view(np.uint32))))[0])
    print("round", rnd, "rays", orays.shape[0], "hits", int((oc > 0).sum()), "differing rays", bad.size)
    if bad.size:
        for i in bad[:10]:
            print(" ray", i, "gpu count", gc[i], "oracle count", oc[i], "gpu", gh[i][:max(gc[i], 1)], "oracle", oh[i][:max(oc[i], 1)],
                  "origin", orays["origin"][i], "direct", orays["direct"][i])
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        np.savez(os.path.join(ROOT, "gpurun_out", "fuzz_%d.npz" % seed), tris=tris, origin=orays["origin"], direct=orays["direct"], gc=gc, oc=oc,
                 gh=gh, oh=oh, bad=bad, M=ob["M"], nodes=ob["nodes"])
        break
    t = 300 + 7 * rnd + seed
    rt.shade(time=t)
    orays = O.shade(cfg, lights, marr, mats, tris, normals, t, orays, oh, oc, osum, oflag)
