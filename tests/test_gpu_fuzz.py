"""Differential fuzzing of the build and the traversal: random triangle soups with the pathologies meshes have -- flat in an
axis, duplicated and degenerate triangles, coordinates on a coarse grid (long runs of equal Morton codes), a large common
offset (few mantissa bits left), anisotropic extents, tiny and huge scales, negative zeros -- and random rays that start inside,
far away and on vertices, with zero and negative-zero direction components. The bar is the parity tests' own: every stage of
the build and every hit chain bit for bit as the oracle's (hlbvh/*.comp, raytracing/directTraverse.comp as restated in
oracle/psm_oracle.c), node visits, triangle tests, stack drops and iteration caps equal.

The seeds that run by default take a few seconds; PSM_FUZZ_SEEDS="a:b" runs seeds a .. b-1, "a,b,c" those seeds (a study on the GPU box:
round 5 ran 0:400, profiles/r05_fuzz.txt)."""
import os

import numpy as np
import pytest

from test_gpu_parity import _built_equals_oracle, _check_build, _hits_equal, _load, _rays_equal, _select_schedule, _trace_both
from util import bits

pytestmark = pytest.mark.gpu


def _seeds():
    spec = os.environ.get("PSM_FUZZ_SEEDS")
    if spec:
        if ":" in spec:
            a, b = spec.split(":")
            return list(range(int(a), int(b)))
        return [int(x) for x in spec.split(",")]
    return list(range(16))


def fuzz_case(seed, visible=False):
    """(tris float32 [n, 3, 3], origin [m, 3], direct [m, 3], tags) of one seed. `visible`: a soup a camera sees something of --
    hundreds of triangles or more, each a few per cent of the box, scales within what INFINITY = 10000 (constants.glsl:82) lets hit."""
    rng = np.random.RandomState(1000 + seed)
    n = int(rng.choice([1, 2, 3, 5, 33, 64, 65, 257, 1000, 4099, 20000], p=[.04, .04, .04, .04, .08, .08, .08, .15, .2, .15, .1]))
    tags = []
    extent = np.float32(10.0) ** rng.uniform(-1.5, 1.5, 3).astype(np.float32)       # anisotropic scene box
    size = np.float32(10.0) ** np.float32(rng.uniform(-2.5, 0.3))                     # triangle size relative to the box
    if visible:
        n = int(rng.choice([257, 1000, 4099]))
        extent = np.float32(10.0) ** rng.uniform(-0.5, 0.5, 3).astype(np.float32)
        size = np.float32(10.0) ** np.float32(rng.uniform(-1.3, -0.4))
    centre = rng.uniform(-1, 1, (n, 1, 3)).astype(np.float32)
    if rng.rand() < 0.3:                                                              # clustered: most triangles in a few clumps
        k = rng.randint(1, 6)
        clumps = rng.uniform(-1, 1, (k, 3)).astype(np.float32)
        centre = (clumps[rng.randint(0, k, n)] + rng.normal(0, 0.02, (n, 3))).astype(np.float32)[:, None, :]
        tags.append("clustered")
    tris = (centre + rng.uniform(-1, 1, (n, 3, 3)).astype(np.float32) * size) * extent
    if rng.rand() < 0.25:                                                             # flat: zero extent in an axis
        ax = rng.randint(0, 3)
        tris[:, :, ax] = np.float32(rng.choice([0.0, -0.0, 1.0, -3.5, 1e4]))
        tags.append("flat%d" % ax)
    if rng.rand() < 0.3:                                                              # coordinates on a grid: runs of equal codes
        q = np.float32(extent.max() / rng.choice([4, 16, 64, 1024]))
        tris = (np.round(tris / q) * q).astype(np.float32)
        tags.append("grid")
    if rng.rand() < 0.3 and n > 3:                                                    # duplicates (equal-distance hit chains)
        dup = rng.randint(0, n, max(1, n // 4))
        tris[rng.randint(0, n, dup.size)] = tris[dup]
        tags.append("dup")
    if rng.rand() < 0.3 and n > 3:                                                    # degenerate: points and needles
        deg = rng.randint(0, n, max(1, n // 8))
        tris[deg, 1] = tris[deg, 0]
        half = deg[: deg.size // 2]
        tris[half, 2] = tris[half, 0]
        tags.append("degenerate")
    r = rng.rand()
    if visible:
        if r < 0.7:
            tris = tris * np.float32(10.0) ** np.float32(rng.uniform(-2.5, 2.5))
        r = 1.0
    if r < 0.15:                                                                      # global scale, far from 1
        s = np.float32(10.0) ** np.float32(rng.uniform(-9, -3))
        tris = tris * s
        tags.append("scale%.0e" % s)
    elif r < 0.3:
        s = np.float32(10.0) ** np.float32(rng.uniform(3, 9))
        tris = tris * s
        tags.append("scale%.0e" % s)
    if rng.rand() < 0.2:                                                              # a common offset eats the mantissa
        off = (rng.uniform(-1, 1, 3) * 10.0 ** rng.uniform(2, 6)).astype(np.float32)
        tris = tris + off * np.float32(np.abs(tris).max() / 10.0)
        tags.append("offset")
    if rng.rand() < 0.2:
        tris[np.abs(tris) < np.float32(np.abs(tris).max() * 0.05)] *= np.float32(-0.0) if rng.rand() < 0.5 else np.float32(0.0)
        tags.append("zeros")
    tris = np.ascontiguousarray(tris.astype(np.float32))
    assert np.isfinite(tris).all()

    m = 4096
    lo, hi = tris.reshape(-1, 3).min(0), tris.reshape(-1, 3).max(0)
    span = np.maximum(hi - lo, np.float32(1e-30))
    origin = (lo + rng.rand(m, 3).astype(np.float32) * span).astype(np.float32)
    target = (tris[rng.randint(0, n, m)] * rng.dirichlet((1, 1, 1), m).astype(np.float32)[:, :, None]).sum(1)
    k = m // 8
    origin[0:k] = (lo - span * np.float32(rng.uniform(0.1, 3.0)))                     # outside, one corner's side
    origin[k:2 * k] += (rng.normal(0, 1, (k, 3)) * span * 2).astype(np.float32)       # around the box
    direct = (target - origin).astype(np.float32)                                      # aimed at a point of a triangle
    direct[2 * k:3 * k] = rng.normal(0, 1, (k, 3)).astype(np.float32)                  # anywhere
    direct[3 * k:4 * k, rng.randint(0, 3)] = 0.0                                       # a zero component
    direct[4 * k:5 * k, rng.randint(0, 3)] = -0.0
    ax = rng.randint(0, 3)
    direct[5 * k:6 * k] = 0.0
    direct[5 * k:6 * k, ax] = np.where(rng.rand(k) < 0.5, -1.0, 1.0)                   # axis-parallel
    origin[6 * k:7 * k] = tris[rng.randint(0, n, k), rng.randint(0, 3, k)]             # starts on a vertex
    direct[6 * k:7 * k] = rng.normal(0, 1, (k, 3)).astype(np.float32)
    direct[7 * k:] *= np.float32(10.0) ** rng.uniform(-15, 15, (m - 7 * k, 1)).astype(np.float32)   # any length
    direct = np.nan_to_num(direct, nan=0.0, posinf=3e38, neginf=-3e38).astype(np.float32)
    return tris, np.ascontiguousarray(origin), np.ascontiguousarray(direct), tags


@pytest.mark.parametrize("seed", _seeds())
def test_fuzzed_soup_builds_and_traces_like_the_oracle(psm, ctx, oracle, scenes, seed):
    tris, origin, direct, tags = fuzz_case(seed)
    n = tris.shape[0]
    sc = {"tris": tris, "normals": np.zeros_like(tris), "mats": np.zeros(n, np.int32)}
    sc["normals"][:, :, 1] = 1.0
    _check_build(psm, ctx, oracle, sc)
    gh, gc, st, oh, oc, octr = _trace_both(psm, ctx, oracle, tris, origin, direct)
    _hits_equal(gh, gc, oh, oc)
    assert (st.node_visits, st.tri_tests, st.stack_drops, st.iter_caps) == (octr.node_visits, octr.tri_tests, octr.stack_drops, octr.iter_caps), tags


def fuzz_opt(rng):
    """A build `optimization` matrix (TriangleHierarchy.inl:226-267): a rotation, an anisotropic scale and a shift, in double."""
    a, b = rng.uniform(-np.pi, np.pi, 2)
    ry = np.array([[np.cos(a), 0, np.sin(a), 0], [0, 1, 0, 0], [-np.sin(a), 0, np.cos(a), 0], [0, 0, 0, 1]])
    rx = np.array([[1, 0, 0, 0], [0, np.cos(b), -np.sin(b), 0], [0, np.sin(b), np.cos(b), 0], [0, 0, 0, 1]])
    sc = np.diag(list(10.0 ** rng.uniform(-1, 1, 3)) + [1.0])
    m = ry @ rx @ sc
    m[:3, 3] = rng.uniform(-3, 3, 3)
    return np.ascontiguousarray(m, np.float64).reshape(16)


@pytest.mark.parametrize("seed", _seeds())
def test_fuzzed_soup_with_an_optimisation_matrix(psm, ctx, oracle, scenes, seed):
    """The same soups under a random optimisation matrix: both fit transforms in double (inverse(opt), then the bounds of the
    transformed scene), keys, tree and boxes as the oracle's; primary hits of the tree built that way."""
    tris, origin, direct, tags = fuzz_case(seed)
    rng = np.random.RandomState(5000 + seed)
    opt = fuzz_opt(rng)
    n = tris.shape[0]
    sc = {"tris": tris, "normals": np.zeros_like(tris), "mats": np.zeros(n, np.int32)}
    sc["normals"][:, :, 1] = 1.0
    ob = _check_build(psm, ctx, oracle, sc, opt)
    th = _load(psm, ctx, sc)
    th.build(opt)
    m = origin.shape[0]
    rays = np.zeros(m, psm.RAY_DT)
    rays["origin"], rays["direct"], rays["color"] = origin, direct, 1.0
    rays["bitfield"] = 1 | (3 << 8)
    rays["texel"] = np.arange(m) % 100
    rays["pkey"] = np.arange(m)
    rt = psm.Pipeline(ctx)
    rt.resizeBuffers(128, 128)
    rt.upload_rays(rays)
    assert rt.intersection(th) == 1
    gh, gc = rt.download_hits(m)
    oh, oc, _ = oracle.traverse(ob["nodes"], tris, ob["M"], origin, direct, 8)
    _hits_equal(gh, gc, oh, oc)
    rt.close()
    th.close()


@pytest.mark.parametrize("seed", _seeds())
def test_fuzzed_frames_shade_like_the_oracle(psm, ctx, oracle, scenes, seed):
    """Whole bounce rounds on the soups: random materials (black, brighter than 1, rough to mirror, dielectric to full metal, some
    emissive), normals that are the face's, random unit vectors or zero (loader.comp:119-128 falls back), a ragged image size.
    Hit chains and the next round's ray queue slot for slot bit for bit, deposit counts exactly, sums to float-atomic order."""
    tris, _, _, tags = fuzz_case(seed, visible=True)
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
    centre = tris[rng.randint(0, n)].mean(0).astype(np.float32)                 # the camera looks at a triangle ...
    away = rng.normal(0, 1, 3)
    away /= np.linalg.norm(away)
    eye = (centre + away * np.linalg.norm(span) * rng.uniform(0.2, 2.0)).astype(np.float32)   # ... from inside or outside the soup
    scene = {"tris": tris, "normals": normals, "mats": mats, "materials": materials, "eye": eye, "view": centre}
    w, h = int(rng.randint(17, 120)), int(rng.randint(9, 80))
    th = _load(psm, ctx, scene)
    th.build()
    rt = psm.Pipeline(ctx)
    rt.resizeBuffers(w, h)
    rt.resize(w, h)
    ms = psm.MaterialSet()
    for mm in materials:
        ms.addSubmat(mm)
    cam = scenes.camera_matrices(scene["eye"], scene["view"], w, h)
    ob = oracle.build_scene(tris)
    marr = scenes.materials_array(materials)
    cfg = oracle.make_cfg(w, h, material_count=len(marr))
    lights = oracle.default_lights(1)
    rt.camera_matrices(cam[0], cam[1], time=seed)
    orays, ocoord, osum, oflag = oracle.camera(cfg, cam[0], cam[1], seed)
    rt.applyMaterials(ms)
    for rnd in range(4):
        if orays.shape[0] < 32:
            break
        assert rt.getRayCount() == orays.shape[0]
        ctx.stats_enable(False, True)
        ctx.stats_reset()
        rt.intersection(th)
        drops = ctx.stats().chain_pool_drops
        ctx.stats_enable(False, False)
        oh, oc, _ = oracle.traverse(ob["nodes"], tris, ob["M"], orays["origin"], orays["direct"], 8)
        gh, gc = rt.download_hits(orays.shape[0])
        if drops:
            # More chained hits than the chain pool holds -- currentRayLimit / 2 entries, the reference's whole hit buffer
            # (Pipeline.inl:193; its allocation is an unchecked atomic increment, directTraverse.comp:230): a soup of coplanar triangles
            # seen face on gives most rays a chain of eight. The product's rule: a chain that no longer fits is cut to its head (the
            # nearest hit with the largest triangle id, reorderTriangles' first), the ray count and everything else are unaffected,
            # `chain_pool_drops` counts the cut chains. Which rays are cut is the order of the pool's atomic: the round ends the case.
            cut = (gc == 1) & (oc > 1)
            assert int(cut.sum()) == drops and int((np.maximum(oc, 1) - 1)[~cut].sum()) <= max(2 * w * h, 1024) < int((np.maximum(oc, 1) - 1).sum())
            oc = np.where(cut, 1, oc)
            _hits_equal(gh, gc, oh, oc)
            break
        _hits_equal(gh, gc, oh, oc)
        t = 300 + 7 * rnd + seed
        rt.shade(time=t)
        orays = oracle.shade(cfg, lights, marr, mats, tris, normals, t, orays, oh, oc, osum, oflag)
        assert rt.raycountCache == orays.shape[0], (rnd, tags)
        _rays_equal(rt.download_rays(), orays)
        sm, c, f = rt.download_texels()
        fin = np.isfinite(osum[:, :3])
        assert np.array_equal(np.isfinite(sm[:, :3]), fin)
        np.testing.assert_allclose(sm[:, :3][fin], osum[:, :3][fin], rtol=1e-5, atol=1e-6)
        assert np.array_equal(sm[:, 3], osum[:, 3])
    rt.close()
    th.close()


@pytest.mark.parametrize("seed", _seeds())
def test_fuzzed_keys_sort_stably(psm, ctx, oracle, seed):
    """The hybrid sort against numpy's stable order on keys whose sixteen-bit bins are as uneven as a generator can make them:
    a few bins hold most keys (some longer than a chunk's LDS: the slow path and the fall-back), the rest are sparse; runs of equal
    keys; keys that differ in one digit only; 63- and 64-bit keys."""
    rng = np.random.RandomState(13000 + seed)
    n = int(rng.choice([0, 1, 63, 4096, 4097, 20000, 100003, 300007]))
    nb = max(1, int(rng.choice([1, 2, 5, 40, 3000, 60000])))
    bins = rng.randint(0, 65536, nb).astype(np.uint64)
    wgt = rng.pareto(0.7, nb) + 1e-3
    hi = bins[rng.choice(nb, n, p=wgt / wgt.sum())] if n else np.zeros(0, np.uint64)
    low = rng.randint(0, 2 ** 47, n, dtype=np.int64).astype(np.uint64)
    mode = rng.randint(0, 4)
    if mode == 1 and n:
        low &= np.uint64(0xFF) << np.uint64(8 * rng.randint(0, 6))          # one digit differs
    elif mode == 2 and n:
        low = low[rng.randint(0, max(1, n // 50), n)]                         # long runs of equal keys
    keys = (hi << np.uint64(48 if rng.rand() < 0.5 else 47)) | low
    vals = rng.permutation(n).astype(np.uint32)
    rs = psm.RadixSort(ctx)
    try:
        gk, gv = rs.sort_arrays(keys, vals)
    finally:
        rs.setAlgorithm(2)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(gk, keys[order]) and np.array_equal(gv, vals[order])


@pytest.mark.parametrize("seed", _seeds())
def test_fuzzed_refit_matches_the_oracle(psm, ctx, oracle, scenes, seed):
    """psm_bvh_refit on the soups: vertices jittered inside the build's bounds, some triangles collapsed to points (a refit keeps
    them: aabbmaker's degeneracy tests decide what a BUILD keeps) -- boxes of oracle.refit bit for bit, topology untouched."""
    tris, _, _, tags = fuzz_case(seed)
    n = tris.shape[0]
    sc = {"tris": tris, "normals": np.zeros_like(tris), "mats": np.zeros(n, np.int32)}
    sc["normals"][:, :, 1] = 1.0
    ob = oracle.build_scene(tris)
    th = _load(psm, ctx, sc)
    th.build()
    if ob["count"] < 2:
        th.close()
        return
    _built_equals_oracle(psm, oracle, th, ob)
    rng = np.random.RandomState(17000 + seed)
    lo, hi = tris.reshape(-1, 3).min(0), tris.reshape(-1, 3).max(0)
    moved = np.clip(tris + rng.normal(0, 0.05, tris.shape).astype(np.float32) * (hi - lo), lo, hi).astype(np.float32)
    if n > 8:
        p = rng.randint(0, n, n // 8)
        moved[p, 1] = moved[p, 0]
        moved[p, 2] = moved[p, 0]
    moved = np.ascontiguousarray(moved)
    th.clearTribuffer()
    th.loadTriangles(moved, sc["normals"], sc["mats"])
    th.refit()
    _built_equals_oracle(psm, oracle, th, oracle.refit(ob, moved))
    th.close()


def fuzz_schedule(rng):
    """A traversal schedule psm_rt_set_traverse_mode / _phases / _adaptive / _solo accept, drawn at random."""
    kind = rng.randint(0, 3)
    kw = {"solo": int(rng.randint(0, 5))}
    if kind == 0:
        return "whole", kw
    if kind == 1:
        kw["caps"] = [int(x) for x in rng.randint(1, 60, rng.randint(1, 8))]
        return "phased", kw
    kw.update(min_live=int(rng.randint(2, 65)), min_steps=int(rng.randint(0, 40)), final_rays=int(rng.choice([0, 64, 1000, 65536])),
              max_launches=int(rng.randint(2, 16)))
    return "adaptive", kw


@pytest.mark.parametrize("seed", _seeds())
def test_fuzzed_schedules_on_fuzzed_soups(psm, ctx, oracle, scenes, seed):
    """Every way a ray can be carried through its traversal -- one launch, launches capped at random step counts, hand-over below a
    random number of live lanes, the solo gear from 0 to 4 rays -- on the soups and their rays (long chains of coplanar duplicates
    included, which cannot hand over): hits, chains, V, T, drops and caps as the oracle's uninterrupted loop."""
    tris, origin, direct, tags = fuzz_case(seed)
    rng = np.random.RandomState(21000 + seed)
    mode, kw = fuzz_schedule(rng)
    n, m = tris.shape[0], origin.shape[0]
    sc = {"tris": tris, "normals": np.zeros_like(tris), "mats": np.zeros(n, np.int32)}
    sc["normals"][:, :, 1] = 1.0
    th = _load(psm, ctx, sc)
    th.build()
    ob = oracle.build_scene(tris)
    rays = np.zeros(m, psm.RAY_DT)
    rays["origin"], rays["direct"], rays["color"] = origin, direct, 1.0
    rays["bitfield"] = 1 | (3 << 8)
    rays["texel"] = np.arange(m) % 100
    rays["pkey"] = np.arange(m)
    rt = psm.Pipeline(ctx)
    rt.resizeBuffers(128, 128)
    _select_schedule(rt, mode, kw)
    rt.upload_rays(rays)
    ctx.stats_enable(False, True)
    ctx.stats_reset()
    assert rt.intersection(th) == 1
    st = ctx.stats()
    ctx.stats_enable(False, False)
    gh, gc = rt.download_hits(m)
    oh, oc, octr = oracle.traverse(ob["nodes"], tris, ob["M"], origin, direct, 8)
    assert st.chain_pool_drops == 0
    _hits_equal(gh, gc, oh, oc)
    assert (st.node_visits, st.tri_tests, st.stack_drops, st.iter_caps) == (octr.node_visits, octr.tri_tests, octr.stack_drops, octr.iter_caps), (mode, kw, tags)
    rt.close()
    th.close()


@pytest.mark.parametrize("seed", _seeds())
def test_fuzzed_soup_with_non_finite_vertices(psm, ctx, oracle, scenes, seed):
    """A few vertices at +-infinity or NaN (a broken export): the bounds reduction, the fit transform and aabbmaker's tests meet
    them in min / max / compare positions whose semantics psm_math.h restates from GLSL as the oracle does. Same bits, stage by stage."""
    tris, origin, direct, tags = fuzz_case(seed)
    rng = np.random.RandomState(25000 + seed)
    n = tris.shape[0]
    k = max(1, n // 50)
    bad = rng.choice([np.inf, -np.inf, np.nan], k)
    tris[rng.randint(0, n, k), rng.randint(0, 3, k), rng.randint(0, 3, k)] = bad.astype(np.float32)
    sc = {"tris": tris, "normals": np.zeros_like(tris), "mats": np.zeros(n, np.int32)}
    sc["normals"][:, :, 1] = 1.0
    _check_build(psm, ctx, oracle, sc)
    gh, gc, st, oh, oc, octr = _trace_both(psm, ctx, oracle, tris, origin[:1024], direct[:1024])
    _hits_equal(gh, gc, oh, oc)
    assert (st.node_visits, st.tri_tests) == (octr.node_visits, octr.tri_tests)
