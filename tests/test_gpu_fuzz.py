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

from test_gpu_parity import _built_equals_oracle, _check_build, _global_tri, _hits_equal, _load, _rays_equal, _select_schedule, _trace_both
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


def fuzz_case(seed, visible=False, big=False):
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
    if big:   # many workgroups, sort tiles and segment-tree levels: runs of equal codes that span them, bins longer than a chunk
        n = int(rng.choice([200003, 524288, 1000003]))
        size = np.float32(10.0) ** np.float32(rng.uniform(-3.5, -1.5))
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


def fuzz_scene(seed, scenes):
    """A renderable scene of one seed -- a `visible` soup, random materials (black, brighter than 1, rough to mirror, dielectric to full
    metal, some emissive), normals that are the face's, random unit vectors or zero (loader.comp:119-128 falls back), in four of ten
    cases textures in random parts over texcoords far outside 0..1, a camera that looks at a triangle -- and a ragged image size."""
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
    texcoords = textures = None
    if rng.rand() < 0.4:   # any texture in any part (a height map as an albedo, an albedo as a normal map), texcoords far outside 0..1
        textures = scenes.procedural_textures()
        texcoords = np.ascontiguousarray((rng.uniform(-1, 1, (n, 3, 2)) * 10.0 ** rng.uniform(-1, 1.5)).astype(np.float32))
        for mm in materials:
            for part in ("diffusePart", "specularPart", "bumpPart", "emissivePart"):
                if rng.rand() < 0.5:
                    mm[part] = int(rng.randint(1, 6))
        tags.append("tex")
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
    scene = {"tris": tris, "normals": normals, "mats": mats, "materials": materials, "eye": eye, "view": centre, "texcoords": texcoords}
    w, h = int(rng.randint(17, 120)), int(rng.randint(9, 80))
    scene["textures"] = textures
    scene["name"] = "fuzz%d" % seed
    return scene, w, h, tags


@pytest.mark.parametrize("seed", _seeds())
def test_fuzzed_frames_shade_like_the_oracle(psm, ctx, oracle, scenes, seed):
    """Whole bounce rounds on the soups: random materials (black, brighter than 1, rough to mirror, dielectric to full metal, some
    emissive), normals that are the face's, random unit vectors or zero (loader.comp:119-128 falls back), a ragged image size.
    Hit chains and the next round's ray queue slot for slot bit for bit, deposit counts exactly, sums to float-atomic order."""
    scene, w, h, tags = fuzz_scene(seed, scenes)
    tris, normals, mats, materials, texcoords, textures = (scene[k] for k in ("tris", "normals", "mats", "materials", "texcoords", "textures"))
    th = _load(psm, ctx, scene)
    th.build()
    rt = psm.Pipeline(ctx)
    rt.resizeBuffers(w, h)
    rt.resize(w, h)
    ms = psm.MaterialSet()
    for mm in materials:
        ms.addSubmat(mm)
    if textures:
        ts = psm.TextureSet()
        for slot in sorted(textures):
            assert ts.loadTexture(textures[slot]) == slot
        ms.setTextureSet(ts)
    cam = scenes.camera_matrices(scene["eye"], scene["view"], w, h)
    ob = oracle.build_scene(tris)
    marr = scenes.materials_array(materials)
    cfg = oracle.make_cfg(w, h, material_count=len(marr))
    if textures:
        oracle.set_textures(cfg, texcoords, textures)
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
    """The three sorts (mostly the hybrid one) against numpy's stable order on keys whose sixteen-bit bins are as uneven as a generator can make them:
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
    rs.setAlgorithm(int(rng.choice([0, 1, 2, 2])))     # eight passes, one-sweep, hybrid
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


@pytest.mark.parametrize("seed", _seeds())
def test_fuzzed_soup_cut_into_chained_hierarchies(psm, ctx, oracle, scenes, seed):
    """SURVEY f4, multi-BVH: the soup cut into two to four hierarchies (the last one repeating triangles of the others: equal-distance
    hits that come from different hierarchies), intersection() called once per hierarchy over the same rays, each under a random
    schedule -- the chain after every call as oracle.traverse_chain's (directTraverse.comp:219-249, 335-346)."""
    tris, origin, direct, tags = fuzz_case(seed)
    rng = np.random.RandomState(29000 + seed)
    n = tris.shape[0]
    if n < 8:
        return
    dup = rng.randint(0, n, max(1, n // 6))
    tris = np.ascontiguousarray(np.concatenate([tris, tris[dup]]))
    k = int(rng.randint(2, 5))
    cuts = np.sort(rng.choice(np.arange(1, n), k - 2, replace=False)) if k > 2 else np.zeros(0, np.int64)
    edges = [0] + [int(c) for c in cuts] + [n, n + dup.size]
    parts = [np.arange(edges[i], edges[i + 1]) for i in range(len(edges) - 1)]
    ths, obs = [], []
    for ix in parts:
        sc = {"tris": np.ascontiguousarray(tris[ix]), "normals": np.zeros((ix.size, 3, 3), np.float32), "mats": np.zeros(ix.size, np.int32)}
        sc["normals"][:, :, 1] = 1.0
        th = _load(psm, ctx, sc)
        th.build()
        ths.append(th)
        obs.append(oracle.build_scene(sc["tris"]))
    m = origin.shape[0]
    rays = np.zeros(m, psm.RAY_DT)
    rays["origin"], rays["direct"], rays["color"] = origin, direct, 1.0
    rays["bitfield"] = 1 | (3 << 8)
    rays["texel"] = np.arange(m) % 100
    rays["pkey"] = np.arange(m)
    rt = psm.Pipeline(ctx)
    rt.resizeBuffers(128, 128)
    rt.upload_rays(rays)
    oh, oc = None, None
    for j, ix in enumerate(parts):
        _select_schedule(rt, *fuzz_schedule(rng))
        ctx.stats_enable(False, True)
        ctx.stats_reset()
        rt.intersection(ths[j])
        drops = ctx.stats().chain_pool_drops
        ctx.stats_enable(False, False)
        if j == 0:
            oh, oc, _ = oracle.traverse(obs[0]["nodes"], tris[ix], obs[0]["M"], origin, direct, 8)
        else:
            oracle.traverse_chain(obs[j]["nodes"], np.ascontiguousarray(tris[ix]), obs[j]["M"], origin, direct, oh, oc, int(ix[0]), 8)
        gh, gc = rt.download_hits(m)
        if drops:   # the chain pool is full (every call of a round allocates its chains anew): cut chains keep their head, see the frames test
            cut = (gc == 1) & (oc > 1)
            assert int(cut.sum()) == drops
            _hits_equal(_global_tri(gh, gc, parts), gc, oh, np.where(cut, 1, oc))
            break
        _hits_equal(_global_tri(gh, gc, parts), gc, oh, oc)
    rt.close()
    for th in ths:
        th.close()


@pytest.mark.parametrize("seed", _seeds())
def test_fuzzed_frames_on_lanes_equal_the_oracles_frames(psm, oracle, scenes, seed):
    """psm_lanes_render under random conditions -- 1..5 lanes, 1..9 frames, depth 1..16, with and without the rebuild per frame -- on
    the fuzzed scenes: the accumulated image, the deposit counts and the rounds and rays of every frame as the same frames rendered one
    after another by the oracle."""
    scene, w, h, tags = fuzz_scene(seed, scenes)
    rng = np.random.RandomState(33000 + seed)
    lanes, frames = int(rng.randint(1, 6)), int(rng.randint(1, 10))
    depth, rebuild = int(rng.choice([1, 2, 3, 16])), bool(rng.rand() < 0.7)
    w, h = min(w, 64), min(h, 48)
    batch = psm.FrameBatch(lanes, w, h, seed=seed + 5)
    batch.allocate(scene["tris"].shape[0])
    batch.loadTriangles(scene["tris"], scene["normals"], scene["mats"], scene["texcoords"])
    ms = psm.MaterialSet()
    for mm in scene["materials"]:
        ms.addSubmat(mm)
    if scene["textures"]:
        ts = psm.TextureSet()
        for slot in sorted(scene["textures"]):
            ts.loadTexture(scene["textures"][slot])
        ms.setTextureSet(ts)
    batch.applyMaterials(ms)
    if not rebuild:
        for ln in batch.lanes:
            ln.th.build()
    for ln in batch.lanes:
        ln.ctx.stats_enable(False, True)
        ln.ctx.stats_reset()
    per_frame = batch.render(frames, scene["eye"], scene["view"], depth=depth, rebuild=rebuild)
    img = batch.snapHdr()
    if sum(ln.ctx.stats().chain_pool_drops for ln in batch.lanes):   # a full chain pool cuts chains (frames test): which ones is a matter of timing
        batch.close()
        return
    sc = dict(scene)
    if not sc["textures"]:
        sc.pop("textures")
        sc.pop("texcoords")
    ref, st = oracle.render_frames(sc, w, h, frames=frames, seed=seed + 5, depth=depth, frame_streams=True)
    fin = np.isfinite(ref[..., :3])
    assert np.array_equal(np.isfinite(img[..., :3]), fin)
    np.testing.assert_allclose(img[..., :3][fin], ref[..., :3][fin], rtol=1e-4, atol=1e-5)
    assert np.array_equal(img[..., 3], ref[..., 3])
    assert len(per_frame) == frames
    assert sum(r for _, r in per_frame) == st["rays"] and sum(k for k, _ in per_frame) == len(st["rounds"]), (lanes, frames, depth, rebuild, tags)
    batch.close()


@pytest.mark.parametrize("seed", _seeds())
def test_fuzzed_tile_sharding_equals_the_whole_frame(psm, ctx, oracle, scenes, seed):
    """SURVEY 8(e) on the fuzzed scenes: the frame dealt in 8-row bands over 2..6 ranks under random band weights (ranks without a
    band included), every rank's camera queue as the oracle's, the bounce loop in lock step on the global ray count, the tiles
    gathered into one rank's pipeline (psm_dist_gather_tiles' one launch over the ranks' dense tiles): the unsharded frame's sums
    to float-atomic order, its deposit counts exactly."""
    import importlib
    pdist = importlib.import_module("prismarine-core_amd.dist")
    scene, w, h, tags = fuzz_scene(seed, scenes)
    rng = np.random.RandomState(37000 + seed)
    world = int(rng.randint(2, 7))
    weights = None if rng.rand() < 0.4 else [int(x) for x in rng.randint(0, 4, world)]
    if weights is not None and sum(weights) == 0:
        weights[int(rng.randint(0, world))] = 1
    root = int(rng.randint(0, world))
    th = _load(psm, ctx, scene)
    th.build()
    ms = psm.MaterialSet()
    for mm in scene["materials"]:
        ms.addSubmat(mm)
    if scene["textures"]:
        ts = psm.TextureSet()
        for slot in sorted(scene["textures"]):
            ts.loadTexture(scene["textures"][slot])
        ms.setTextureSet(ts)
    cam = scenes.camera_matrices(scene["eye"], scene["view"], w, h)
    cfg = oracle.make_cfg(w, h, material_count=len(scene["materials"]))

    def lockstep(pipes):
        gens = [psm.sharded_rounds(rt, th, ms, depth=4) for rt in pipes]
        local = [next(g) for g in gens]
        alive = [True] * len(gens)
        while any(alive):
            total = sum(local)
            for i, g in enumerate(gens):
                if alive[i]:
                    try:
                        local[i] = g.send(total)
                    except StopIteration:
                        alive[i] = False

    ctx.stats_enable(False, True)
    ctx.stats_reset()
    full = psm.Pipeline(ctx, seed=seed + 1)
    full.resizeBuffers(w, h)
    full.resize(w, h)
    full.camera_matrices(cam[0], cam[1])
    lockstep([full])
    want, _, _ = full.download_texels()
    pipes = []
    for r in range(world):
        rt = psm.Pipeline(ctx, seed=seed + 1)
        rt.resizeBuffers(w, h)
        rt.resize(w, h)
        rt.setTileInterleaved(r, world, weights)
        assert rt.tile_texels() == pdist.owned_texels(r, world, w, h, weights)
        rt.camera_matrices(cam[0], cam[1], time=None)
        pipes.append(rt)
    t0, _ = oracle.rand_next(seed + 1)
    for r, rt in enumerate(pipes):
        orays, *_ = oracle.camera_interleaved(cfg, cam[0], cam[1], t0, r, world, weights)
        _rays_equal(rt.download_rays(), orays)
    assert sum(rt.tile_texels() for rt in pipes) == w * h
    lockstep(pipes)
    drops = ctx.stats().chain_pool_drops
    ctx.stats_enable(False, False)
    if not drops:   # (a full chain pool cuts chains by timing: frames test)
        per = pdist.largest_tile_texels(world, w, h, weights) * 16
        hall = ctx.buf_alloc(per * world)
        pall, _ = ctx.buf_ptr(hall)
        for r in range(world):
            pipes[r].pack_texels_dev(pall + r * per)
        pipes[root].unpack_tiles_dev(world, root, pall, per // 4)
        got, _, _ = pipes[root].download_texels()
        fin = np.isfinite(want[:, :3])
        assert np.array_equal(np.isfinite(got[:, :3]), fin)
        np.testing.assert_allclose(got[:, :3][fin], want[:, :3][fin], rtol=1e-5, atol=1e-6)
        assert np.array_equal(got[:, 3], want[:, 3])
        ctx.buf_free(hall)
    for rt in pipes + [full]:
        rt.close()
    th.close()


@pytest.mark.parametrize("seed", _seeds())
def test_fuzzed_viewer_frames_with_lights_sky_and_sampling(psm, ctx, oracle, scenes, seed):
    """Whole frames the way the viewer renders them (Viewer.cpp:296-312) under random settings: one to three spherical lights of
    random place, size and colour (one of them below the horizon now and then), an equirect sky of random texels, the 360-degree
    camera, a ray grid up to twice the window (sampler.comp:37-97), a depth limit, one to three accumulated frames. The display image
    within 1e-4 of the oracle's, its sample counts exactly."""
    scene, w, h, tags = fuzz_scene(seed, scenes)
    rng = np.random.RandomState(41000 + seed)
    frames, depth = int(rng.randint(1, 4)), int(rng.choice([1, 2, 4, 16]))
    ss = int(rng.choice([1, 1, 2]))
    dw, dh = min(w, 72), min(h, 48)
    gw, gh = dw * ss, dh * ss
    L = oracle.default_lights(int(rng.randint(1, 4)))
    for i in range(L.shape[0]):
        if i > 0 or rng.rand() < 0.5:
            v = rng.normal(0, 1, 3)
            L[i]["lightVector"] = (v[0], v[1], v[2], float(10.0 ** rng.uniform(0.5, 3)))
            L[i]["lightColor"] = tuple(float(x) for x in rng.uniform(0, 200, 3)) + (float(rng.uniform(0.5, 50)),)
            L[i]["lightOffset"] = tuple(float(x) for x in rng.uniform(-0.5, 0.5, 3)) + (0.0,)
            L[i]["lightAmbient"] = tuple(float(x) for x in rng.uniform(0, 0.1, 3)) + (0.0,)
    sky = rng.randint(0, 256, (int(rng.randint(1, 40)), int(rng.randint(1, 70)), 4)).astype(np.uint8) if rng.rand() < 0.4 else None
    mode360 = rng.rand() < 0.25
    th = _load(psm, ctx, scene)
    th.build()
    rt = psm.Pipeline(ctx)
    rt.resizeBuffers(gw, gh)
    rt.resize(dw, dh)
    ms = psm.MaterialSet()
    for mm in scene["materials"]:
        ms.addSubmat(mm)
    if scene["textures"]:
        ts = psm.TextureSet()
        for slot in sorted(scene["textures"]):
            ts.loadTexture(scene["textures"][slot])
        ms.setTextureSet(ts)
    rt.setLights(L)
    if sky is not None:
        rt.setSkybox(sky)
    if mode360:
        rt.switchMode()
    rt.setSeed(seed + 11)
    ctx.stats_enable(False, True)
    ctx.stats_reset()
    for _ in range(frames):
        psm.render_frame(rt, th, ms, scene["eye"], scene["view"], depth=depth)
    drops = ctx.stats().chain_pool_drops
    ctx.stats_enable(False, False)
    img = rt.snapHdr()
    assert img.shape == (dh, dw, 4)
    if not drops:
        sc = dict(scene)
        if not sc["textures"]:
            sc.pop("textures")
            sc.pop("texcoords")
        ref, st = oracle.render_frames(sc, gw, gh, frames=frames, seed=seed + 11, depth=depth, lights=L, skybox=sky, display=(dw, dh), enable360=mode360)
        fin = np.isfinite(ref[..., :3])
        assert np.array_equal(np.isfinite(img[..., :3]), fin)
        np.testing.assert_allclose(img[..., :3][fin], ref[..., :3][fin], rtol=1e-4, atol=1e-5)
        assert np.array_equal(img[..., 3], ref[..., 3])
    if sky is not None:
        rt.setSkybox(None)
    rt.close()
    th.close()


def _big_seeds():
    spec = os.environ.get("PSM_FUZZ_BIG_SEEDS")
    if spec:
        a, b = spec.split(":")
        return list(range(int(a), int(b)))
    return list(range(4))


@pytest.mark.parametrize("seed", _big_seeds())
def test_fuzzed_big_soups(psm, ctx, oracle, scenes, seed):
    """The soups at 200 003 .. 1 000 003 triangles: hundreds of workgroups per kernel, hundreds of sort tiles and LDS chunks, segment
    trees twenty levels deep -- with the grid, duplicate and cluster pathologies that make runs of equal Morton codes span them and
    push the hybrid sort into its slow path. Every build stage bit for bit, then the rays."""
    tris, origin, direct, tags = fuzz_case(seed + 100000, big=True)
    n = tris.shape[0]
    sc = {"tris": tris, "normals": np.zeros_like(tris), "mats": np.zeros(n, np.int32)}
    sc["normals"][:, :, 1] = 1.0
    _check_build(psm, ctx, oracle, sc)
    gh, gc, st, oh, oc, octr = _trace_both(psm, ctx, oracle, tris, origin, direct)
    _hits_equal(gh, gc, oh, oc)
    assert (st.node_visits, st.tri_tests, st.stack_drops, st.iter_caps) == (octr.node_visits, octr.tri_tests, octr.stack_drops, octr.iter_caps), tags
