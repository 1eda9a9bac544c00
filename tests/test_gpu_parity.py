"""GPU parity tests: HIP path (through the C ABI) vs the CPU oracle on the same seeded inputs.

Bit-exact for Morton codes, sort order, BVH topology + boxes and hit chains; bit-exact for the ray
queues the shading kernel emits; accumulated radiance within 1e-4 relative (the only GPU/CPU
difference left is the order of the float atomic adds into a texel).
"""
import os

import numpy as np
import pytest

from util import bits, canonical_nodes

pytestmark = pytest.mark.gpu


def _load(psm, ctx, scene):
    th = psm.TriangleHierarchy(ctx)
    th.allocate(scene["tris"].shape[0])
    th.loadTriangles(scene["tris"], scene["normals"], scene["mats"], scene.get("texcoords"))
    return th


def _scene(scenes, name):
    if name.endswith("+tex"):
        return scenes.textured(_scene(scenes, name[:-4]))
    if name.endswith("+blackmetal"):
        # a black full-metal wall (albedo 0, metallic 1: its reflection colour is clamp(0 / 0) = NaN, and the reference's createRay
        # KEEPS a NaN colour: shadinglib.glsl / rayslib.glsl:162-203) and a material brighter than 1 -- not `ordinary`
        # (psm_rt_set_materials), so rt_shade builds both lobes of every hit
        sc = _scene(scenes, name[:-11])
        sc["materials"] = list(sc["materials"])
        sc["materials"][0] = dict(sc["materials"][0], diffuse=(0.0, 0.0, 0.0, 1.0), specular=(0.0, 0.3, 1.0, 0.0))
        sc["materials"][1] = dict(sc["materials"][1], diffuse=(1.5, 0.2, 0.1, 1.0))
        return sc
    if name.endswith("+clearmetal"):
        # every fourth triangle twice, both copies a full-metal material whose diffuse constant says alpha 0 (would the
        # equal-distance chain composite to albedo 0 and the dropped lobe's colour be NaN, against the `ordinary material` rule of
        # psm_rt_set_materials?). It does not: without a diffuse texture a hit's albedo is vec4(diffuse.xyz, 1) (fetchDiffuse,
        # surface.comp:155-161), the chain's head is opaque and nothing composites. The case holds that: 500-1000 chains of two
        # hits per round, queues slot for slot as the oracle's.
        sc = dict(_scene(scenes, name[:-11]))
        sc["materials"] = list(sc["materials"]) + [dict(sc["materials"][0], diffuse=(0.8, 0.8, 0.8, 0.0), specular=(0.0, 0.3, 1.0, 0.0))]
        dup = np.arange(0, sc["tris"].shape[0], 4)
        sc["tris"] = np.ascontiguousarray(np.concatenate([sc["tris"], sc["tris"][dup]], 0))
        sc["normals"] = np.ascontiguousarray(np.concatenate([sc["normals"], sc["normals"][dup]], 0))
        sc["mats"] = np.ascontiguousarray(np.concatenate([sc["mats"], np.full(dup.size, len(sc["materials"]) - 1, np.int32)]))
        sc["mats"][dup] = len(sc["materials"]) - 1
        return sc
    if name == "cornell":
        return scenes.cornell()
    if name == "cornell_open":
        return scenes.cornell(open_top=True)
    if name == "sponza_small":
        return scenes.sponza_like(n_tris=20011)
    if name == "sponza":
        return scenes.sponza_like()
    raise KeyError(name)


# ---------------------------------------------------------------------------- sort
@pytest.mark.parametrize("algo", [0, 1, 2], ids=["three-kernel", "onesweep", "hybrid"])
@pytest.mark.parametrize("n", [0, 1, 2, 255, 256, 257, 1000, 1024, 1025, 4095, 4096, 4097, 100003, 2 ** 17 + 1, 300007, 2 ** 19 + 3, 2 ** 21 + 77])
def test_sort_matches_oracle(psm, ctx, oracle, n, algo):
    """Every sort implementation (psm_sort_set_algorithm): the histogram / scan / scatter kernels per pass, one histogram
    sweep + look-back scatter, and the hybrid sort (two global passes, the rest in LDS; the short keys below all fall
    into ONE of its sixteen-bit bins, so from 32 k keys on this also runs the chunk that does not fit LDS)."""
    rng = np.random.RandomState(n + 1)
    keys = rng.randint(0, 2 ** 63 - 1, size=n, dtype=np.int64).astype(np.uint64)
    if n > 10:
        keys[rng.randint(0, n, n // 3)] = keys[rng.randint(0, n, n // 3)]  # ties: stability matters
        keys[: n // 8] &= np.uint64(0xFFFF)                                # short keys
    vals = np.arange(n, dtype=np.uint32)
    rs = psm.RadixSort(ctx)
    rs.setAlgorithm(algo)
    try:
        gk, gv = rs.sort_arrays(keys, vals)
    finally:
        rs.setAlgorithm(2)
    ok, ov = oracle.radix_sort(keys, vals.astype(np.int32))
    assert np.array_equal(gk, ok)
    assert np.array_equal(gv, ov.astype(np.uint32))
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(gk, keys[order]) and np.array_equal(gv, vals[order])


def test_sort_all_equal_and_sorted_inputs(psm, ctx):
    rs = psm.RadixSort(ctx)
    n = 5000
    k = np.full(n, 0x123456789ABCDEF, np.uint64)
    gk, gv = rs.sort_arrays(k, np.arange(n, dtype=np.uint32))
    assert np.array_equal(gv, np.arange(n, dtype=np.uint32))
    k = np.arange(n, dtype=np.uint64)[::-1].copy() << np.uint64(40)
    gk, gv = rs.sort_arrays(k, np.arange(n, dtype=np.uint32))
    assert np.array_equal(gk, np.sort(k)) and np.array_equal(gv, np.arange(n, dtype=np.uint32)[::-1])


def _stable(keys, vals):
    order = np.argsort(keys, kind="stable")
    return keys[order], vals[order]


def test_hybrid_sort_of_morton_codes_stays_hybrid(psm, ctx, oracle, scenes):
    """The Morton codes of the Sponza-class scene through the generic entry point (bins = key bits 48..63): the longest bin
    (1 977 keys) fits a chunk, nothing overflows, the context stays on the hybrid sort."""
    ob = oracle.build_scene(scenes.sponza_like()["tris"])
    keys = ob["keys_unsorted"]
    vals = np.arange(keys.size, dtype=np.uint32)[::-1].copy()   # values need not be ascending: stability is by position
    rs = psm.RadixSort(ctx)
    gk, gv = rs.sort_arrays(keys, vals)
    ek, ev = _stable(keys, vals)
    assert np.array_equal(gk, ek) and np.array_equal(gv, ev)
    assert np.array_equal(gk, ob["keys"])
    assert rs.getAlgorithm() == (2, 2)


@pytest.mark.parametrize("case", ["bin_of_3072", "bin_of_3073", "bin_spans_three_stretches", "bins_end_on_stretch_ends",
                                  "all_equal", "two_values", "high_bit_set", "low_digits_only", "sorted", "reversed"])
def test_hybrid_sort_chunk_edges(psm, ctx, case):
    """radix_local's chunking (small sorts: stretches of S = 1 024 bin starts, 4 096 keys of LDS): a bin of exactly CAP - S
    keys still fits, one key more goes through global memory (and sends the context back to the eight passes), a bin over
    several stretches leaves workgroups without a chunk, bins that end exactly where a stretch ends, keys that differ in one
    digit only."""
    rng = np.random.RandomState(11)
    n = 20000
    low = rng.randint(0, 2 ** 48, size=n, dtype=np.int64).astype(np.uint64)
    overflow = False
    if case in ("bin_of_3072", "bin_of_3073"):
        big = 3072 if case == "bin_of_3072" else 3073
        # bins of 1 key up to position 1023, then the big bin starting exactly at 1023 (the last start of stretch 0)
        bins = np.concatenate([np.arange(1023), np.full(big, 5000), 6000 + rng.randint(0, 3000, n - 1023 - big)])
        overflow = case == "bin_of_3073"
    elif case == "bin_spans_three_stretches":
        bins = np.concatenate([np.arange(900), np.full(3000, 5000), 6000 + np.arange(n - 3900) // 7])
    elif case == "bins_end_on_stretch_ends":
        bins = np.arange(n) // 512
    elif case == "all_equal":
        bins = np.full(n, 77); low[:] = 12345
        overflow = True     # one bin of 20 000 keys (nothing to sort in it, as the workgroup finds out: no digit differs)
    elif case == "two_values":
        bins = np.full(n, 77); low[:] = rng.randint(0, 2, n).astype(np.uint64) << np.uint64(23)
        overflow = True
    elif case == "high_bit_set":
        bins = rng.randint(0, 65536, n)
    elif case == "low_digits_only":
        bins = np.zeros(n, np.int64); low = rng.randint(0, 256, n).astype(np.uint64)
        overflow = True
    elif case in ("sorted", "reversed"):
        bins = np.sort(rng.randint(0, 65536, n)); low = np.sort(low)
        if case == "reversed":
            bins = bins[::-1].copy(); low = low[::-1].copy()
    keys = (bins.astype(np.uint64) << np.uint64(48)) | low
    if case not in ("sorted", "reversed", "bin_of_3072", "bin_of_3073", "bin_spans_three_stretches", "bins_end_on_stretch_ends"):
        keys = keys[rng.permutation(n)]
    elif case in ("bin_of_3072", "bin_of_3073", "bin_spans_three_stretches", "bins_end_on_stretch_ends"):
        keys = keys[rng.permutation(n)]   # the global passes bring the bins back together, in this layout
    vals = rng.randint(0, 2 ** 32, size=n, dtype=np.int64).astype(np.uint32)
    rs = psm.RadixSort(ctx)
    gk, gv = rs.sort_arrays(keys, vals)
    ek, ev = _stable(keys, vals)
    assert np.array_equal(gk, ek) and np.array_equal(gv, ev)
    assert rs.getAlgorithm() == (2, 0 if overflow else 2)
    if overflow:   # the context now sorts with the eight passes; asking for the hybrid sort again re-arms it
        gk, gv = rs.sort_arrays(keys, vals)
        assert np.array_equal(gk, ek) and np.array_equal(gv, ev)
        rs.setAlgorithm(2)
        assert rs.getAlgorithm() == (2, 2)


# ---------------------------------------------------------------------------- build
def _check_build(psm, ctx, oracle, scene, opt=None):
    tris = scene["tris"]
    ob = oracle.build_scene(tris, opt)
    th = _load(psm, ctx, scene)
    th.stage("bounds", opt)
    info = th.info()
    assert np.array_equal(bits(np.array(info.bounds_min)), bits(ob["mn"]))
    assert np.array_equal(bits(np.array(info.bounds_max)), bits(ob["mx"]))
    assert np.array_equal(bits(np.array(info.transform)), bits(ob["M"]))
    th.stage("morton")
    info = th.info()
    n = ob["count"]
    assert info.leaf_count == n
    assert np.array_equal(th.download(psm.BVH_KEYS, np.uint64, n), ob["keys_unsorted"])
    assert np.array_equal(th.download(psm.BVH_INDICES, np.uint32, n), np.arange(n, dtype=np.uint32))
    lb = th.download(psm.BVH_LEAF_BOX, np.uint32, 4 * n).reshape(n, 4)
    assert np.array_equal(lb, ob["leafs"]["box"])
    assert np.array_equal(th.download(psm.BVH_LEAF_TRI, np.int32, n), ob["leafs"]["pdata"][:, 3])
    th.stage("sort")
    assert np.array_equal(th.download(psm.BVH_KEYS, np.uint64, n), ob["keys"])
    assert np.array_equal(th.download(psm.BVH_INDICES, np.uint32, n), ob["idx"].astype(np.uint32))
    th.stage("emit")
    info = th.info()
    if n >= 2:
        link = th.download(psm.BVH_LINK, np.int32, 2 * (n - 1)).reshape(n - 1, 2)
        pb = th.download(psm.BVH_PAIR_BOX, np.uint32, 8 * (n - 1)).reshape(n - 1, 8)
        rg = th.download(psm.BVH_RANGE, np.int32, 2 * (n - 1)).reshape(n - 1, 2)
        assert info.root == oracle.find_split(ob["keys"], 0, n - 1)
        nodes = canonical_nodes(info.root, link, pb, rg, oracle.NODE_DT)
        assert nodes.shape == ob["nodes"].shape
        assert np.array_equal(nodes["pdata"], ob["nodes"]["pdata"])   # topology, ranges, triangle ids
        assert np.array_equal(nodes["box"], ob["nodes"]["box"])       # fp16 boxes after refit
    else:
        assert info.root == -1
    th.close()
    return ob


@pytest.mark.parametrize("name", ["cornell", "sponza_small", "sponza"])
def test_build_matches_oracle(psm, ctx, oracle, scenes, name):
    _check_build(psm, ctx, oracle, _scene(scenes, name))


def test_build_with_ties_degenerates_and_opt(psm, ctx, oracle, scenes):
    rng = np.random.RandomState(7)
    base = scenes.sponza_like(n_tris=3001)["tris"]
    tris = np.concatenate([base, base[:700], base[100:400], base[100:400]], 0)  # duplicate triangles: Morton ties
    tris[5] = tris[5][0]          # degenerate (all three vertices equal) -> skipped by aabbmaker.comp:160
    tris[77] = tris[77][1]
    tris = tris[rng.permutation(tris.shape[0])]
    sc = {"tris": np.ascontiguousarray(tris), "normals": scenes.prepare_normals(tris),
          "mats": np.zeros(tris.shape[0], np.int32)}
    ob = _check_build(psm, ctx, oracle, sc)
    assert ob["count"] == tris.shape[0] - 2
    assert (ob["keys"][1:] == ob["keys"][:-1]).sum() > 500
    a = 0.3
    opt = np.array([[np.cos(a), 0, np.sin(a), 0.5], [0, 1.3, 0, -1.0], [-np.sin(a), 0, np.cos(a), 2.0], [0, 0, 0, 1]],
                   np.float64).reshape(16)
    _check_build(psm, ctx, oracle, sc, opt)


def _built_equals_oracle(psm, oracle, th, ob):
    """the state a finished build leaves behind: transform, sorted keys + order, topology, ranges, triangle ids, boxes"""
    info = th.info()
    n = ob["count"]
    assert info.leaf_count == n
    assert np.array_equal(bits(np.array(info.transform)), bits(ob["M"]))
    assert np.array_equal(th.download(psm.BVH_KEYS, np.uint64, n), ob["keys"])
    assert np.array_equal(th.download(psm.BVH_INDICES, np.uint32, n), ob["idx"].astype(np.uint32))
    link = th.download(psm.BVH_LINK, np.int32, 2 * (n - 1)).reshape(n - 1, 2)
    pb = th.download(psm.BVH_PAIR_BOX, np.uint32, 8 * (n - 1)).reshape(n - 1, 8)
    rg = th.download(psm.BVH_RANGE, np.int32, 2 * (n - 1)).reshape(n - 1, 2)
    assert info.root == oracle.find_split(ob["keys"], 0, n - 1)
    # the traversal records (written by the build: every node puts its own id into its parent's record) against the
    # reference-shaped records (links found by findSplit, produced on demand): same links, same twelve box coordinates
    n32 = th.download(psm.BVH_NODE32, np.uint32, 8 * (n - 1)).reshape(n - 1, 8)
    assert np.array_equal(n32[:, 6:8].view(np.int32), link.astype(np.int32))
    def half(a, k):   # k-th fp16 of a row of packed words
        return (a[:, k // 2] >> (16 * (k % 2))) & 0xFFFF
    for side, first in ((0, 0), (4, 6)):      # left box: halves 0..5 of the record, right box: 6..11
        for j, src in enumerate((0, 1, 2, 4, 5, 6)):   # mn.x mn.y mn.z | mx.x mx.y mx.z of the uvec4 pair (mn.xy mn.zw mx.xy mx.zw)
            assert np.array_equal(half(n32, first + j), half(pb[:, side:side + 4], src))
    nodes = canonical_nodes(info.root, link, pb, rg, oracle.NODE_DT)
    assert np.array_equal(nodes["pdata"], ob["nodes"]["pdata"])
    assert np.array_equal(nodes["box"], ob["nodes"]["box"])


def test_rebuild_replayed_as_captured_graph_is_bit_exact(psm, ctx, oracle, scenes):
    """psm_bvh_build replays a rebuild as one captured hipGraph from the second build of a triangle count on. Every
    build -- plain, the capturing one, replays, with another optimisation matrix, after the triangle set changed, after
    another sorter grew the context's sort buffers under the graph, with the other sort algorithm, with graphs off --
    must leave exactly the oracle's tree."""
    sc = _scene(scenes, "sponza_small")
    ob = oracle.build_scene(sc["tris"])
    th = _load(psm, ctx, sc)
    for _ in range(4):           # plain, capture + replay, replay, replay
        th.markDirty()
        th.build()
        _built_equals_oracle(psm, oracle, th, ob)
    a = 0.3
    opt = np.array([[np.cos(a), 0, np.sin(a), 0.5], [0, 1.3, 0, -1.0], [-np.sin(a), 0, np.cos(a), 2.0], [0, 0, 0, 1]], np.float64).reshape(16)
    ob_opt = oracle.build_scene(sc["tris"], opt)
    th.markDirty()
    th.build(opt)                # replay of the same graph: the matrix travels through device memory
    _built_equals_oracle(psm, oracle, th, ob_opt)
    # another sorter on the same context grows the shared sort buffers: the graph's pointers are stale and it is re-captured
    rs = psm.RadixSort(ctx)
    k = np.random.RandomState(3).randint(0, 2**63, size=sc["tris"].shape[0] * 3, dtype=np.int64).astype(np.uint64)
    gk, _ = rs.sort_arrays(k, np.arange(k.size, dtype=np.uint32))
    assert np.array_equal(gk, np.sort(k))
    for _ in range(3):
        th.markDirty()
        th.build()
        _built_equals_oracle(psm, oracle, th, ob)
    for algo in (1, 0):          # the one-sweep sort, the eight-pass sort: another graph each
        rs.setAlgorithm(algo)
        for _ in range(3):
            th.markDirty()
            th.build()
            _built_equals_oracle(psm, oracle, th, ob)
    rs.setAlgorithm(2)
    th.setBuildGraph(False)
    th.markDirty()
    th.build()
    _built_equals_oracle(psm, oracle, th, ob)
    th.setBuildGraph(True)
    # a different triangle set in the same hierarchy
    half = sc["tris"].shape[0] // 2
    sc2 = {"tris": np.ascontiguousarray(sc["tris"][:half]), "normals": np.ascontiguousarray(sc["normals"][:half]),
           "mats": np.ascontiguousarray(sc["mats"][:half])}
    ob2 = oracle.build_scene(sc2["tris"])
    th.clearTribuffer()
    th.loadTriangles(sc2["tris"], sc2["normals"], sc2["mats"])
    for _ in range(3):
        th.markDirty()
        th.build()
        _built_equals_oracle(psm, oracle, th, ob2)
    th.close()


def test_build_of_a_clustered_scene_overflows_the_hybrid_sort_and_falls_back(psm, ctx, oracle, scenes):
    """A scene whose bounds are set by two far-away outliers while everything else sits in a thousandth of them: its Morton
    codes share their top bits, a sixteen-bit bin holds thousands of keys, the hybrid sort's chunk does not fit LDS. The build
    must be bit-exact all the same -- through the workgroup's global-memory path the first time, through the eight-pass sort
    the context falls back to afterwards, plain and as a re-captured graph -- and psm_sort_get_algorithm must say so."""
    base = scenes.sponza_like(n_tris=20011)["tris"]
    lo, hi = base.reshape(-1, 3).min(0), base.reshape(-1, 3).max(0)
    tris = ((base - lo) * np.float32(0.001) + lo).astype(np.float32)        # the scene, shrunk into a corner of its own bounds
    tris = np.ascontiguousarray(np.concatenate([tris, base[:2] + (hi - lo) * np.float32(0.9)], 0))   # two triangles far out keep the bounds wide
    sc = {"tris": tris, "normals": scenes.prepare_normals(tris), "mats": np.zeros(tris.shape[0], np.int32)}
    ob = oracle.build_scene(tris)
    top = (ob["keys"] >> np.uint64(47)).astype(np.int64)
    assert np.bincount(top - top.min()).max() > 4096      # one bin is longer than a chunk's LDS
    rs = psm.RadixSort(ctx)
    assert rs.getAlgorithm() == (2, 2)
    th = _load(psm, ctx, sc)
    for k in range(4):           # first build: plain launches, hybrid with the slow chunk; then the fallback, then its graph
        th.markDirty()
        th.build()
        _built_equals_oracle(psm, oracle, th, ob)
        assert rs.getAlgorithm() == (2, 0)
    th.close()
    rs.setAlgorithm(2)           # (the fixture does this for the next test anyway)
    # a well-spread scene on the re-armed context stays hybrid
    sc2 = scenes.sponza_like(n_tris=20011)
    th = _load(psm, ctx, sc2)
    ob2 = oracle.build_scene(sc2["tris"])
    for k in range(3):
        th.markDirty()
        th.build()
        _built_equals_oracle(psm, oracle, th, ob2)
    assert rs.getAlgorithm() == (2, 2)
    th.close()


def test_refit_only_keeps_the_tree_and_matches_the_oracle(psm, ctx, oracle, scenes):
    """psm_bvh_refit (refit-only dynamic updates): the triangles of a built hierarchy are reloaded, moved -- the node
    records afterwards are the build's topology with the boxes of oracle.refit (aabbmaker.comp:165-194 + refit.comp:21-114), bit
    for bit, twice in a row (a refit of a refit), rays hit what the oracle's traversal of that tree hits, a rebuild afterwards is
    a build again; without a build, or with another triangle count, the call is refused."""
    sc = scenes.sponza_like(n_tris=20011)
    th = psm.TriangleHierarchy(ctx)
    th.allocate(sc["tris"].shape[0])
    th.loadTriangles(sc["tris"], sc["normals"], sc["mats"])
    with pytest.raises(psm.PsmError):
        th.refit()                                   # nothing built yet
    th.build()
    ob = oracle.build_scene(sc["tris"])
    _built_equals_oracle(psm, oracle, th, ob)
    rng = np.random.RandomState(4)
    tris = sc["tris"]
    lo, hi = tris.reshape(-1, 3).min(0), tris.reshape(-1, 3).max(0)
    for step in range(2):
        tris = np.clip(tris + rng.normal(0, 0.03, tris.shape), lo, hi).astype(np.float32)   # (inside the build's bounds: its transform stays)
        th.clearTribuffer()
        th.loadTriangles(tris, scenes.prepare_normals(tris), sc["mats"])
        th.refit()
        ob = oracle.refit(ob, tris)
        _built_equals_oracle(psm, oracle, th, ob)
    # traversal of the refitted tree
    w, h = 96, 54
    rt = psm.Pipeline(ctx, seed=5)
    rt.resizeBuffers(w, h)
    rt.resize(w, h)
    cam = scenes.camera_matrices(sc["eye"], sc["view"], w, h)
    rt.camera_matrices(cam[0], cam[1], time=3)
    rays = rt.download_rays()
    rt.intersection(th, force=True)
    gh, gc = rt.download_hits(rays.shape[0])
    oh, oc, _ = oracle.traverse(ob["nodes"], tris, ob["M"], rays["origin"], rays["direct"], 8)
    _hits_equal(gh, gc, oh, oc)
    rt.close()
    # another triangle count: no refit; a rebuild is a build
    th.clearTribuffer()
    th.loadTriangles(tris[:-5], scenes.prepare_normals(tris[:-5]), sc["mats"][:-5])
    with pytest.raises(psm.PsmError):
        th.refit()
    th.markDirty()
    th.build()
    _built_equals_oracle(psm, oracle, th, oracle.build_scene(tris[:-5]))
    th.close()


@pytest.mark.parametrize("n", [1, 2, 3])
def test_build_tiny(psm, ctx, oracle, scenes, n):
    tris = scenes.cornell()["tris"][:n]
    sc = {"tris": tris, "normals": scenes.prepare_normals(tris), "mats": np.zeros(n, np.int32)}
    _check_build(psm, ctx, oracle, sc)


# ---------------------------------------------------------------------------- camera + traverse
def _setup_frame(psm, ctx, scenes, scene, w, h):
    th = _load(psm, ctx, scene)
    th.build()
    rt = psm.Pipeline(ctx)
    rt.resizeBuffers(w, h)
    rt.resize(w, h)
    ms = psm.MaterialSet()
    for m in scene["materials"]:
        ms.addSubmat(m)
    if scene.get("textures"):
        ts = psm.TextureSet()
        for slot in sorted(scene["textures"]):
            assert ts.loadTexture(scene["textures"][slot]) == slot
        ms.setTextureSet(ts)
    cam = scenes.camera_matrices(scene["eye"], scene["view"], w, h)
    return th, rt, ms, cam


def _rays_equal(g, o):
    assert g.shape == o.shape
    for f in ("origin", "direct", "color"):
        # bit for bit; a NaN (the colour of a black full-metal surface's reflection ray) where the oracle has a NaN -- the two
        # machines give 0 / 0 different signs and payloads
        gn, on = np.isnan(g[f]), np.isnan(o[f])
        assert np.array_equal(gn, on), f
        assert np.array_equal(bits(g[f])[~on], bits(o[f])[~on]), f
    for f in ("bitfield", "texel", "pkey"):
        assert np.array_equal(g[f], o[f]), f


def _hits_equal(gh, gc, oh, oc):
    assert np.array_equal(gc, oc)
    for k in range(8):
        m = oc > k
        assert np.array_equal(gh["tri"][m, k], oh["tri"][m, k])
        for f in ("u", "v", "t"):
            assert np.array_equal(bits(gh[f][m, k]), bits(oh[f][m, k])), (f, k)


@pytest.mark.parametrize("name,w,h", [("cornell", 1280, 720), ("sponza_small", 320, 180), ("sponza", 480, 270)])
def test_primary_rays_and_hits_bit_exact(psm, ctx, oracle, scenes, name, w, h):
    """BASELINE config 2 (cornell 1280x720 primary rays) and the Sponza-class scene."""
    scene = _scene(scenes, name)
    th, rt, ms, cam = _setup_frame(psm, ctx, scenes, scene, w, h)
    ob = oracle.build_scene(scene["tris"])
    cfg = oracle.make_cfg(w, h, material_count=len(scene["materials"]))
    ctx.stats_enable(False, True)
    ctx.stats_reset()
    rt.camera_matrices(cam[0], cam[1], time=4242)
    orays, ocoord, osum, oflag = oracle.camera(cfg, cam[0], cam[1], 4242)
    grays = rt.download_rays()
    _rays_equal(grays, orays)
    s, c, f = rt.download_texels()
    assert np.array_equal(bits(c), bits(ocoord)) and np.array_equal(f, oflag)
    assert rt.intersection(th) == 1
    gh, gc = rt.download_hits(w * h)
    oh, oc, octr = oracle.traverse(ob["nodes"], scene["tris"], ob["M"], orays["origin"], orays["direct"], 8)
    _hits_equal(gh, gc, oh, oc)
    st = ctx.stats()
    assert st.node_visits == octr.node_visits and st.tri_tests == octr.tri_tests
    assert st.stack_drops == octr.stack_drops and st.iter_caps == octr.iter_caps
    assert st.rays_traced == w * h
    ctx.stats_enable(False, False)
    rt.close()
    th.close()


def test_traverse_random_rays_with_chains(psm, ctx, oracle, scenes):
    """Rays aimed at shared edges / duplicated triangles: equal-distance hit chains (directTraverse.comp:287-305)."""
    rng = np.random.RandomState(3)
    base = scenes.sponza_like(n_tris=6007)["tris"]
    tris = np.ascontiguousarray(np.concatenate([base, base[:1500]], 0))   # coplanar duplicates -> chains of 2
    sc = {"tris": tris, "normals": scenes.prepare_normals(tris), "mats": np.zeros(tris.shape[0], np.int32),
          "materials": scenes.cornell()["materials"], "eye": np.zeros(3, np.float32), "view": np.ones(3, np.float32)}
    th = _load(psm, ctx, sc)
    th.build()
    ob = oracle.build_scene(tris)
    n = 20000
    tid = rng.randint(0, tris.shape[0], n)
    w = rng.dirichlet((1, 1, 1), n).astype(np.float32)
    w[: n // 4, 2] = 0  # exactly on an edge
    w[: n // 4] /= w[: n // 4].sum(1, keepdims=True)
    target = (tris[tid] * w[:, :, None]).sum(1)
    origin = (target + rng.normal(0, 1, (n, 3)) * 3.0 + np.array([0, 4, 0])).astype(np.float32)
    direct = (target - origin).astype(np.float32)
    rays = np.zeros(n, psm.RAY_DT)
    rays["origin"], rays["direct"], rays["color"] = origin, direct, 1.0
    rays["bitfield"] = 1 | (3 << 8)
    rays["texel"] = np.arange(n) % 100
    rays["pkey"] = np.arange(n)
    rt = psm.Pipeline(ctx)
    rt.resizeBuffers(128, 128)
    rt.upload_rays(rays)
    assert rt.intersection(th) == 1
    gh, gc = rt.download_hits(n)
    oh, oc, _ = oracle.traverse(ob["nodes"], tris, ob["M"], origin, direct, 8)
    assert (oc > 1).sum() > 100
    _hits_equal(gh, gc, oh, oc)
    rt.close()
    th.close()


def _trace_both(psm, ctx, oracle, tris, origin, direct):
    sc = {"tris": tris, "normals": np.zeros_like(tris), "mats": np.zeros(tris.shape[0], np.int32)}
    sc["normals"][:, :, 1] = 1.0
    th = _load(psm, ctx, sc)
    th.build()
    ob = oracle.build_scene(tris)
    n = origin.shape[0]
    rays = np.zeros(n, psm.RAY_DT)
    rays["origin"], rays["direct"], rays["color"] = origin, direct, 1.0
    rays["bitfield"] = 1 | (3 << 8)
    rays["texel"] = np.arange(n) % 100
    rays["pkey"] = np.arange(n)
    rt = psm.Pipeline(ctx)
    rt.resizeBuffers(128, 128)
    rt.upload_rays(rays)
    ctx.stats_enable(False, True)
    ctx.stats_reset()
    assert rt.intersection(th) == 1
    st = ctx.stats()
    ctx.stats_enable(False, False)
    gh, gc = rt.download_hits(n)
    oh, oc, octr = oracle.traverse(ob["nodes"], tris, ob["M"], origin, direct, 8)
    rt.close()
    th.close()
    return gh, gc, st, oh, oc, octr


def test_traverse_axis_aligned_inside_and_far_rays_bit_exact(psm, ctx, oracle, scenes):
    """The slab test divides by the direction (directTraverse.comp:372-373): rays with zero components (1/0 = inf,
    0 * inf = NaN inside min/max), rays that start inside the geometry's boxes, rays from very far away, very long and
    very short direction vectors, and rays that point away from everything. Hits, chains, V and T as the oracle's."""
    rng = np.random.RandomState(11)
    tris = np.ascontiguousarray(scenes.sponza_like(n_tris=12007)["tris"])
    lo, hi = tris.reshape(-1, 3).min(0), tris.reshape(-1, 3).max(0)
    n = 16384
    origin = (lo + rng.rand(n, 3) * (hi - lo)).astype(np.float32)           # inside the scene's bounds
    direct = rng.normal(0, 1, (n, 3)).astype(np.float32)
    k = n // 8
    direct[0 * k:1 * k, 0] = 0.0                                              # one zero component
    direct[1 * k:2 * k, 1:] = 0.0                                             # two: axis-parallel rays
    direct[1 * k:2 * k, 0] = np.where(rng.rand(k) < 0.5, -1.0, 1.0)
    direct[2 * k:3 * k] *= np.float32(1e-18)                                  # tiny directions (normalised by the kernel)
    direct[3 * k:4 * k] *= np.float32(1e15)                                   # huge ones
    origin[4 * k:5 * k] = (origin[4 * k:5 * k] - direct[4 * k:5 * k] / np.linalg.norm(direct[4 * k:5 * k], axis=1, keepdims=True)
                           * np.float32(5000.0)).astype(np.float32)          # starts 5000 units away, aims at the scene
    origin[5 * k:6 * k] += np.float32(20000.0)                                # beyond INFINITY, pointing anywhere
    v = tris[rng.randint(0, tris.shape[0], k)]                                # starts exactly on a vertex of a triangle
    origin[6 * k:7 * k] = v[:, 0]
    direct[7 * k:8 * k, 2] = -0.0                                             # negative zero component
    gh, gc, st, oh, oc, octr = _trace_both(psm, ctx, oracle, tris, origin, direct)
    _hits_equal(gh, gc, oh, oc)
    assert (st.node_visits, st.tri_tests, st.stack_drops, st.iter_caps) == (octr.node_visits, octr.tri_tests, octr.stack_drops, octr.iter_caps)
    # tiny / huge directions, far origins and on-vertex origins do hit things; what a zero component does (1/0 = inf in the
    # slab test, 0 * inf = NaN, minNum / maxNum semantics) is whatever the restated arithmetic says -- identically on both sides
    assert (oc[2 * k:4 * k] > 0).sum() > 1000 and (oc[4 * k:5 * k] > 0).sum() > 1000 and (oc[6 * k:7 * k] > 0).all()


def test_traverse_stack_overflow_iteration_cap_and_long_chains_bit_exact(psm, ctx, oracle):
    """Faithful-by-default limits: the 16-entry stack whose overflowing pushes are dropped
    (directTraverse.comp:451-462), the 8192-iteration cap (:383) and the equal-distance chain of at most 8 entries
    (:294). 2^19 slivers in one plane, sorted along x, make a tree 19 levels deep; rays grazing the plane run through
    every padded box and hit nothing, so the stack overflows and the loop runs into its cap. 12 coplanar copies of one
    triangle make chains of 12."""
    rng = np.random.RandomState(5)
    N = 1 << 19
    x0 = np.arange(N, dtype=np.float64) / N
    tris = np.zeros((N, 3, 3), np.float32)
    tris[:, 0] = np.stack([x0, np.zeros(N), np.full(N, -0.5)], 1)
    tris[:, 1] = np.stack([x0 + 0.8 / N, np.zeros(N), np.full(N, -0.5)], 1)
    tris[:, 2] = np.stack([x0, np.zeros(N), np.full(N, 0.5)], 1)
    lid = np.array([[[0.2, 0.6, -0.3], [0.8, 0.6, -0.3], [0.5, 0.6, 0.3]]], np.float32)
    tris = np.ascontiguousarray(np.concatenate([tris, np.repeat(lid, 12, 0)], 0))
    n = 2048
    h = n // 2
    origin = np.zeros((n, 3), np.float32)
    direct = np.zeros((n, 3), np.float32)
    origin[:h] = np.stack([np.full(h, -0.25), rng.rand(h) * 2e-4 + 1e-5, rng.rand(h) * 0.6 - 0.3], 1)   # grazing rays
    direct[:h] = np.stack([np.ones(h), rng.normal(0, 1e-6, h), rng.normal(0, 1e-3, h)], 1)
    t = np.stack([0.35 + rng.rand(h) * 0.3, np.full(h, 0.6), rng.rand(h) * 0.2 - 0.15], 1)              # rays onto the lid
    origin[h:] = t + np.stack([rng.normal(0, 0.2, h), np.full(h, 0.5), rng.normal(0, 0.2, h)], 1)
    direct[h:] = t - origin[h:]
    origin, direct = origin.astype(np.float32), direct.astype(np.float32)
    gh, gc, st, oh, oc, octr = _trace_both(psm, ctx, oracle, tris, origin, direct)
    assert octr.stack_drops > 1000 and octr.iter_caps >= h   # the case is what it says
    assert int(oc.max()) == 8 and (oc == 8).sum() > h // 2   # chains of 12 are cut at BAKED_CAP
    _hits_equal(gh, gc, oh, oc)
    assert (st.node_visits, st.tri_tests, st.stack_drops, st.iter_caps) == (octr.node_visits, octr.tri_tests, octr.stack_drops, octr.iter_caps)


@pytest.mark.parametrize("scale", [0.004, 35.0])
def test_traverse_scaled_scene_bit_exact(psm, ctx, oracle, scenes, scale):
    """INFINITY is 10000 in the reference (constants.glsl:82) and the unit-cube transform makes the ray-parameter
    scale dirlenInv smaller or larger than 1: the `<= INFINITY - PZERO` tests of directTraverse.comp:425-429 bite
    differently in a scene a few millimetres across and in one whose far hits lie beyond 10000."""
    base = scenes.sponza_like(n_tris=20011)
    scene = dict(base)
    scene["tris"] = (base["tris"] * np.float32(scale)).astype(np.float32)
    scene["eye"] = (np.asarray(base["eye"], np.float32) * np.float32(scale)).astype(np.float32)
    scene["view"] = base["view"]
    w, h = 128, 72
    th, rt, ms, cam = _setup_frame(psm, ctx, scenes, scene, w, h)
    ob = oracle.build_scene(scene["tris"])
    mats = scenes.materials_array(scene["materials"])
    cfg = oracle.make_cfg(w, h, material_count=len(mats))
    lights = oracle.default_lights(1)
    rt.camera_matrices(cam[0], cam[1], time=3)
    orays, ocoord, osum, oflag = oracle.camera(cfg, cam[0], cam[1], 3)
    rt.applyMaterials(ms)
    hits_seen = 0
    for rnd in range(3):
        if orays.shape[0] < 32:
            break
        _rays_equal(rt.download_rays(), orays)
        ctx.stats_enable(False, True)
        ctx.stats_reset()
        rt.intersection(th)
        st = ctx.stats()
        ctx.stats_enable(False, False)
        oh, oc, octr = oracle.traverse(ob["nodes"], scene["tris"], ob["M"], orays["origin"], orays["direct"], 8)
        gh, gc = rt.download_hits(orays.shape[0])
        _hits_equal(gh, gc, oh, oc)
        assert (st.node_visits, st.tri_tests) == (octr.node_visits, octr.tri_tests)
        hits_seen += int((oc > 0).sum())
        rt.shade(time=70 + rnd)
        orays = oracle.shade(cfg, lights, mats, scene["mats"], scene["tris"], scene["normals"], 70 + rnd, orays, oh, oc, osum, oflag)
    assert hits_seen > 1000
    rt.close()
    th.close()


TRAVERSE_SCHEDULES = [
    ("whole", {}),
    ("phased", {"caps": [1]}),
    ("phased", {"caps": [3, 5, 7]}),
    ("phased", {"caps": [16, 16, 16, 16, 16, 16, 16]}),
    ("phased", {"caps": [40]}),
    ("adaptive", {"min_live": 16, "min_steps": 8, "final_rays": 256, "max_launches": 8}),
    ("adaptive", {"min_live": 64, "min_steps": 0, "final_rays": 0, "max_launches": 15}),   # hand over at the first idle lane
    ("adaptive", {"min_live": 2, "min_steps": 1, "final_rays": 0, "max_launches": 3}),
    ("adaptive", {"min_live": 24, "min_steps": 4, "final_rays": 64, "max_launches": 4}),
    ("whole", {"solo": 0}),      # the solo gear (psm_rt_set_traverse_solo) off, at its default (1: every entry above and below), and wider
    ("whole", {"solo": 2}),
    ("whole", {"solo": 4}),
    ("adaptive", {"min_live": 16, "min_steps": 8, "final_rays": 256, "max_launches": 8, "solo": 4}),
    ("adaptive", {"min_live": 12, "min_steps": 8, "final_rays": 65536, "max_launches": 3, "solo": 0}),
    ("phased", {"caps": [24, 24], "solo": 3}),
]


def _select_schedule(rt, mode, kw):
    kw = dict(kw)
    if "solo" in kw:
        rt.setTraverseSolo(kw.pop("solo"))
    if mode == "phased":
        rt.setTraversePhases(kw["caps"], min_rays=0)
    elif mode == "adaptive":
        rt.setTraverseAdaptive(min_rays=0, **kw)
    rt.setTraverseMode(mode)


@pytest.mark.parametrize("mode,kw", TRAVERSE_SCHEDULES, ids=lambda v: v if isinstance(v, str) else "-".join("%s" % x for x in v.values()).replace(" ", ""))
def test_every_traversal_schedule_is_bit_exact(psm, ctx, oracle, scenes, mode, kw):
    """psm_rt_set_traverse_mode: every kernel schedule that ships -- one launch, fixed-cap phases, ballot-triggered
    hand-over with persistent resume waves -- gives the hits, chains and
    V / T counters of the oracle's uninterrupted per-ray loop (directTraverse.comp:333-484)."""
    scene = scenes.sponza_like(n_tris=20011)
    w, h = 160, 90
    th, rt, ms, cam = _setup_frame(psm, ctx, scenes, scene, w, h)
    ob = oracle.build_scene(scene["tris"])
    _select_schedule(rt, mode, kw)
    rt.camera_matrices(cam[0], cam[1], time=11)
    rt.applyMaterials(ms)
    for rnd in range(3):
        rays = rt.download_rays()
        ctx.stats_enable(False, True)
        ctx.stats_reset()
        rt.intersection(th, force=True)
        st = ctx.stats()
        ctx.stats_enable(False, False)
        gh, gc = rt.download_hits(rays.shape[0])
        oh, oc, octr = oracle.traverse(ob["nodes"], scene["tris"], ob["M"], rays["origin"], rays["direct"], 8)
        _hits_equal(gh, gc, oh, oc)
        assert (st.node_visits, st.tri_tests) == (octr.node_visits, octr.tri_tests)
        rt.shade(time=40 + rnd)
    rt.close()
    th.close()


@pytest.mark.parametrize("mode,kw", [("adaptive", {"min_live": 16, "min_steps": 8, "final_rays": 64, "max_launches": 8}),
                                     ("phased", {"caps": [5, 9]})], ids=["adaptive", "phased"])
def test_traversal_schedules_keep_equal_distance_chains(psm, ctx, oracle, scenes, mode, kw):
    """Rays that carry an equal-distance chain of two or more hits cannot hand over (their chain lives in registers):
    they finish in the launch they are in. Duplicated coplanar triangles make thousands of them."""
    rng = np.random.RandomState(5)
    base = scenes.sponza_like(n_tris=6007)["tris"]
    tris = np.ascontiguousarray(np.concatenate([base, base[:1500]], 0))
    sc = {"tris": tris, "normals": scenes.prepare_normals(tris), "mats": np.zeros(tris.shape[0], np.int32),
          "materials": scenes.cornell()["materials"], "eye": np.zeros(3, np.float32), "view": np.ones(3, np.float32)}
    th = _load(psm, ctx, sc)
    th.build()
    ob = oracle.build_scene(tris)
    n = 20000
    tid = rng.randint(0, tris.shape[0], n)
    wgt = rng.dirichlet((1, 1, 1), n).astype(np.float32)
    target = (tris[tid] * wgt[:, :, None]).sum(1)
    origin = (target + rng.normal(0, 1, (n, 3)) * 3.0 + np.array([0, 4, 0])).astype(np.float32)
    direct = (target - origin).astype(np.float32)
    rays = np.zeros(n, psm.RAY_DT)
    rays["origin"], rays["direct"], rays["color"] = origin, direct, 1.0
    rays["bitfield"] = 1 | (3 << 8)
    rays["texel"] = np.arange(n) % 100
    rays["pkey"] = np.arange(n)
    rt = psm.Pipeline(ctx)
    rt.resizeBuffers(128, 128)
    _select_schedule(rt, mode, kw)
    rt.upload_rays(rays)
    assert rt.intersection(th) == 1
    gh, gc = rt.download_hits(n)
    oh, oc, _ = oracle.traverse(ob["nodes"], tris, ob["M"], origin, direct, 8)
    assert (oc > 1).sum() > 100
    _hits_equal(gh, gc, oh, oc)
    rt.close()
    th.close()


# ---------------------------------------------------------------------------- shade + full frames
@pytest.mark.parametrize("name,w,h", [("cornell_open", 96, 96), ("sponza_small", 160, 90),
                                      ("cornell_open+tex", 96, 96), ("sponza_small+tex", 160, 90),
                                      ("cornell_open+blackmetal", 96, 96), ("sponza_small+blackmetal", 160, 90),
                                      ("cornell_open+clearmetal", 96, 96),
                                      ("cornell_open", 97, 61), ("cornell", 33, 17)])   # ragged: no multiple of a wave, a workgroup or a band
def test_shade_rounds_bit_exact_queues(psm, ctx, oracle, scenes, name, w, h):
    """+tex: SURVEY f2 -- texcoords, the sampler table and every texture part of surface.comp:100-161."""
    scene = _scene(scenes, name)
    th, rt, ms, cam = _setup_frame(psm, ctx, scenes, scene, w, h)
    ob = oracle.build_scene(scene["tris"])
    mats = scenes.materials_array(scene["materials"])
    cfg = oracle.make_cfg(w, h, material_count=len(mats))
    if scene.get("textures"):
        oracle.set_textures(cfg, scene["texcoords"], scene["textures"])
    lights = oracle.default_lights(1)
    rt.camera_matrices(cam[0], cam[1], time=99)
    orays, ocoord, osum, oflag = oracle.camera(cfg, cam[0], cam[1], 99)
    rt.applyMaterials(ms)
    for rnd in range(6):
        if orays.shape[0] < 32:
            break
        assert rt.getRayCount() == orays.shape[0]
        rt.intersection(th)
        oh, oc, _ = oracle.traverse(ob["nodes"], scene["tris"], ob["M"], orays["origin"], orays["direct"], 8)
        gh, gc = rt.download_hits(orays.shape[0])
        _hits_equal(gh, gc, oh, oc)
        t = 1000 + rnd
        rt.shade(time=t)
        orays = oracle.shade(cfg, lights, mats, scene["mats"], scene["tris"], scene["normals"], t, orays, oh, oc, osum, oflag)
        assert rt.raycountCache == orays.shape[0], rnd
        _rays_equal(rt.download_rays(), orays)
        s, c, f = rt.download_texels()
        np.testing.assert_allclose(s[:, :3], osum[:, :3], rtol=1e-5, atol=1e-6)
        assert np.array_equal(s[:, 3], osum[:, 3])  # deposit counts
    assert rnd >= 2
    rt.close()
    th.close()


def test_queue_overflow_drops_past_the_ray_limit_in_queue_order(psm, ctx, oracle, scenes):
    """currentRayLimit = min(4 w h, 4096^2) (Pipeline.inl:187-189). A full queue of rays that all hit bright diffuse walls
    emits more rays than the limit holds; canonical rule: the queue keeps the first `limit` rays in queue
    order, the rest are dropped (and counted). Queue after the round slot for slot as the oracle's."""
    scene = scenes.cornell()     # closed box: every ray hits something
    w, h = 32, 32
    th, rt, ms, cam = _setup_frame(psm, ctx, scenes, scene, w, h)
    ob = oracle.build_scene(scene["tris"])
    mats = scenes.materials_array(scene["materials"])
    cfg = oracle.make_cfg(w, h, material_count=len(mats))
    limit = 4 * w * h
    assert cfg.ray_limit == limit
    lights = oracle.default_lights(1)
    parts = [oracle.camera(cfg, cam[0], cam[1], 500 + q) for q in range(4)]
    orays = np.concatenate([p_[0] for p_ in parts])      # 4096 rays = the limit, four jittered copies of the camera rays
    assert orays.shape[0] == limit
    orays["pkey"] = np.arange(limit)
    ocoord, osum, oflag = parts[0][1], parts[0][2], parts[0][3]
    rt.camera_matrices(cam[0], cam[1], time=500)      # texel coords / sums / flags as the oracle's first camera pass
    rt.applyMaterials(ms)
    rt.upload_rays(orays)
    ctx.stats_enable(False, True)
    ctx.stats_reset()
    dropped = 0
    for rnd in range(3):
        rt.intersection(th)
        oh, oc, _ = oracle.traverse(ob["nodes"], scene["tris"], ob["M"], orays["origin"], orays["direct"], 8)
        gh, gc = rt.download_hits(orays.shape[0])
        _hits_equal(gh, gc, oh, oc)
        rt.shade(time=900 + rnd)
        orays = oracle.shade(cfg, lights, mats, scene["mats"], scene["tris"], scene["normals"], 900 + rnd, orays, oh, oc, osum, oflag)
        assert rt.raycountCache == orays.shape[0]
        _rays_equal(rt.download_rays(), orays)
        if orays.shape[0] == limit:
            dropped += 1
    st = ctx.stats()
    ctx.stats_enable(False, False)
    assert dropped >= 1 and st.ray_limit_drops > 0     # the queue did overflow
    rt.close()
    th.close()


def test_ray_count_larger_than_the_queue_is_clamped_on_the_device(psm, ctx, scenes):
    """psm_rt_set_ray_count accepts any count up to currentRayLimit; the segmented queue of a later round holds fewer
    rays than that. A count beyond the queue's total must not be resolved to slots past its last segment: the kernels clamp
    it to what the queue holds (bases[nb]), so the round equals the one run with the true count."""
    scene = scenes.sponza_like(n_tris=20011)
    w, h = 96, 54
    outs = []
    for bump in (0, 777, 4 * w * h):
        th, rt, ms, cam = _setup_frame(psm, ctx, scenes, scene, w, h)
        rt.camera_matrices(cam[0], cam[1], time=31)
        rt.applyMaterials(ms)
        rt.intersection(th)
        rt.shade(time=5)
        n = rt.raycountCache                      # the real length of the (segmented) queue
        assert 32 < n < 4 * w * h
        rt.set_ray_count(min(4 * w * h, n + bump))
        rt.intersection(th, force=True)
        rt.shade(time=6)
        assert rt.raycountCache > 32
        rays = rt.download_rays()
        tsum, _, _ = rt.download_texels()
        outs.append((n, rays, tsum))
        rt.close()
        th.close()
    for n, rays, tsum in outs[1:]:
        assert n == outs[0][0]
        _rays_equal(rays, outs[0][1])
        assert np.array_equal(tsum[:, 3], outs[0][2][:, 3])
        np.testing.assert_allclose(tsum[:, :3], outs[0][2][:, :3], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name,w,h,frames", [("cornell_open", 64, 64, 3), ("sponza_small", 128, 72, 2),
                                             ("cornell_open+tex", 64, 64, 3), ("sponza_small+tex", 128, 72, 2),
                                             ("cornell_open", 37, 29, 2)])
def test_accumulated_radiance(psm, ctx, oracle, scenes, name, w, h, frames):
    """Viewer.cpp:296-312 call order, several frames; accumulated radiance within 1e-4 relative."""
    scene = _scene(scenes, name)
    th, rt, ms, cam = _setup_frame(psm, ctx, scenes, scene, w, h)
    rt.setSeed(31337)
    for _ in range(frames):
        psm.render_frame(rt, th, ms, scene["eye"], scene["view"])
    img = rt.snapHdr()
    ref, stats = oracle.render_frames(scene, w, h, frames=frames, seed=31337, nthreads=8)
    assert stats["rays"] > w * h * frames
    assert ref[..., :3].max() > 0.1
    np.testing.assert_allclose(img[..., :3], ref[..., :3], rtol=1e-4, atol=1e-5)
    assert np.array_equal(img[..., 3], ref[..., 3])
    rt.close()
    th.close()


OBJ_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "obj")


@pytest.mark.parametrize("stem", ["box", "rbox", "sphere", "Cow", "shelf"])
def test_obj_files_build_hits_and_radiance(psm, ctx, oracle, scenes, stem):
    """Models the builder did not write (tests/golden/obj: the reference's Resources/toys data files -- Blender / 3ds Max exports
    with v / vt / vn faces, groups and four materials) and one textured model of this repo's (map_Kd / map_Bump / map_Ke / map_Ks),
    from file to image: the HLBVH bit for bit (bounds, Morton codes, order, topology, boxes), the primary rays' hit chains at
    64x64 bit for bit, and three frames of accumulated radiance within 1e-4 -- textures through TextureSet slots and the
    texcoords read from `vt` (stored 1 - v, loader.comp:97-99) where the file has them."""
    scene = scenes.read_obj(os.path.join(OBJ_DIR, stem + ".obj"))
    assert (stem == "shelf") == bool(scene.get("textures"))
    _check_build(psm, ctx, oracle, scene)
    w = h = 64
    th, rt, ms, cam = _setup_frame(psm, ctx, scenes, scene, w, h)
    ob = oracle.build_scene(scene["tris"])
    rt.camera_matrices(cam[0], cam[1], time=7)
    rt.applyMaterials(ms)
    rays = rt.download_rays()
    rt.intersection(th)
    gh, gc = rt.download_hits(rays.shape[0])
    oh, oc, _ = oracle.traverse(ob["nodes"], scene["tris"], ob["M"], rays["origin"], rays["direct"], 8)
    _hits_equal(gh, gc, oh, oc)
    assert (oc > 0).mean() > 0.1                      # the default camera sees the model
    rt.setSeed(4711)
    rt.clearSampler()
    for _ in range(3):
        psm.render_frame(rt, th, ms, scene["eye"], scene["view"])
    img = rt.snapHdr()
    ref, stats = oracle.render_frames(scene, w, h, frames=3, seed=4711, nthreads=8)
    assert ref[..., :3].max() > 0.05
    np.testing.assert_allclose(img[..., :3], ref[..., :3], rtol=1e-4, atol=1e-5)
    assert np.array_equal(img[..., 3], ref[..., 3])
    rt.close()
    th.close()


def test_gltf_scene_from_file_to_image(psm, ctx, oracle, scenes):
    """The reference viewer's only input format (Source/Examples/Viewer.cpp:66-279), tests/golden/gltf/court.gltf: interleaved
    and planar views, a padded stride, 16-bit indices at an odd half-word, a primitive without normals, one without indices, two
    buffers, T * S * R and matrix nodes, an instance, textures by slot. Every (node, primitive) goes through psm_bvh_load_mesh --
    the gather kernel de-indexes, reads by accessor, transforms, falls back to face normals -- and must equal the oracle's
    loader restatement bit for bit (positions, normals, texcoords, material ids, in loading order); then the HLBVH and the
    primary hits bit for bit and three frames of radiance within 1e-4."""
    import importlib
    from util import gltf_soup
    gltf = importlib.import_module("prismarine-core_amd.gltf")
    gs = gltf.read_gltf(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gltf", "court.gltf"))
    scene = gltf_soup(oracle, gs)
    n = gs["triangle_count"]
    th = psm.TriangleHierarchy(ctx)
    th.allocate(n)
    ms = psm.MaterialSet()
    ts = gltf.load_into(gs, th, ms)
    assert th.triangleCount == n == scene["tris"].shape[0] and ms.getMaterialCount() == 5 and len(ts.textures) == 3
    # the primitives of one glTF buffer share ONE device pool (a fresh view per primitive used to mean an upload each)
    addr = lambda a: a.__array_interface__["data"][0]
    used = {addr(g["vertices"]) for g in gs["instances"]} | {addr(g["indices"]) for g in gs["instances"] if g.get("indices") is not None}
    assert th.pool_uploads == len(used) <= 2, (th.pool_uploads, len(used))
    assert np.array_equal(bits(th.download(psm.BVH_POSITIONS, np.float32, 9 * n)), bits(scene["tris"].reshape(-1)))
    assert np.array_equal(bits(th.download(psm.BVH_NORMALS, np.float32, 9 * n)), bits(scene["normals"].reshape(-1)))
    assert np.array_equal(bits(th.download(psm.BVH_TEXCOORDS, np.float32, 6 * n)), bits(scene["texcoords"].reshape(-1)))
    assert np.array_equal(th.download(psm.BVH_MATERIALS, np.int32, n), scene["mats"])
    _check_build(psm, ctx, oracle, scene)            # (the same soup through loadTriangles: every stage of the build)
    th.build()
    ob = oracle.build_scene(scene["tris"])
    assert np.array_equal(th.download(psm.BVH_KEYS, np.uint64, ob["count"]), ob["keys"])
    w, h = 64, 48
    rt = psm.Pipeline(ctx)
    rt.resizeBuffers(w, h)
    rt.resize(w, h)
    cam = scenes.camera_matrices(scene["eye"], scene["view"], w, h)
    rt.camera_matrices(cam[0], cam[1], time=7)
    rt.applyMaterials(ms)
    rays = rt.download_rays()
    rt.intersection(th)
    gh, gc = rt.download_hits(rays.shape[0])
    oh, oc, _ = oracle.traverse(ob["nodes"], scene["tris"], ob["M"], rays["origin"], rays["direct"], 8)
    _hits_equal(gh, gc, oh, oc)
    assert (oc > 0).mean() > 0.5
    rt.setSeed(4711)
    rt.clearSampler()
    for _ in range(3):
        psm.render_frame(rt, th, ms, scene["eye"], scene["view"])
    img = rt.snapHdr()
    ref, stats = oracle.render_frames(scene, w, h, frames=3, seed=4711, nthreads=8)
    assert ref[..., :3].max() > 0.05
    np.testing.assert_allclose(img[..., :3], ref[..., :3], rtol=1e-4, atol=1e-5)
    assert np.array_equal(img[..., 3], ref[..., 3])
    # which way up: row 0 of snapHdr() is the picture's BOTTOM row (camera.comp:61: texel row 0 is NDC y = -1; the reference's
    # HdrImage alike) -- the camera looks down on the court from above, so the last rows are sky and the first are floor -- and a
    # PFM file stores its rows bottom to top: write_pfm writes them as they come (it used to flip them: upside-down snapshots)
    sky = np.float32([0.5, 0.7, 1.0])
    np.testing.assert_allclose(img[-1, :, :3], np.broadcast_to(sky, (w, 3)), rtol=1e-5)
    assert np.abs(img[0, :, :3] - sky).max(-1).min() > 0.05
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        psm.write_pfm(os.path.join(d, "s.pfm"), img)
        raw = open(os.path.join(d, "s.pfm"), "rb").read()
        body = np.frombuffer(raw[raw.index(b"-1.0\n") + 5:], "<f4").reshape(h, w, 3)
        assert np.array_equal(body[0], img[0, :, :3]) and np.array_equal(psm.read_pfm(os.path.join(d, "s.pfm")), img[..., :3])
    rt.close()
    th.close()


def test_cpp_header_layer_loads_a_gltf_scene_like_the_viewer(psm, ctx, oracle, tmp_path):
    """include/Prismarine drop-in headers: the viewer's glTF loop (Viewer.cpp:133-277) in C++ -- raw buffers, BufferViewSet,
    a TriangleArrayInstance + AccessorSet per primitive, the node walk in (stand-in) glm doubles, setTransform + loadMesh per
    (node, primitive) -- leaves in the hierarchy exactly what gltf.read_gltf + the oracle's loader give: positions, normals,
    texcoords and material ids bit for bit, in loading order (tests/cpp/gltf_load.cpp; the parsed file goes over in a flat
    binary form written here from the JSON, the step the reference leaves to tinygltf)."""
    import base64
    import importlib
    import json
    import struct
    import subprocess
    from util import gltf_soup
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "gltf_load")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-I", os.path.join(root, "include"), "-DPSM_NO_SYSTEM_GLM",
                           os.path.join(root, "tests", "cpp", "gltf_load.cpp"), "-o", exe,
                           "-L", os.path.join(root, "prismarine-core_amd"), "-lpsm_hip",
                           "-Wl,-rpath," + os.path.join(root, "prismarine-core_amd")])
    path = os.path.join(root, "tests", "golden", "gltf", "court.gltf")
    g = json.load(open(path))
    blob = bytearray()
    i32 = lambda *v: blob.extend(struct.pack("<%di" % len(v), *v))
    f64 = lambda v: blob.extend(struct.pack("<%dd" % len(v), *v))
    i32(len(g["buffers"]))
    for b in g["buffers"]:
        raw = base64.b64decode(b["uri"].split(",", 1)[1]) if b["uri"].startswith("data:") else open(os.path.join(os.path.dirname(path), b["uri"]), "rb").read()
        raw = raw + b"\0" * (-len(raw) % 4)
        i32(len(raw))
        blob.extend(raw)
    i32(len(g["bufferViews"]))
    for v in g["bufferViews"]:
        i32(v["buffer"], v.get("byteOffset", 0), v.get("byteStride", 0))
    i32(len(g["accessors"]))
    for a in g["accessors"]:
        i32(a["bufferView"], a.get("byteOffset", 0), a["componentType"], a["count"])
    i32(len(g["meshes"]))
    for m in g["meshes"]:
        i32(len(m["primitives"]))
        for p in m["primitives"]:
            at = p["attributes"]
            i32(at.get("POSITION", -1), at.get("NORMAL", -1), at.get("TEXCOORD_0", -1), p.get("indices", -1), p.get("material", -1), p.get("mode", 4))
    i32(len(g["nodes"]))
    for nd in g["nodes"]:
        hm = "matrix" in nd
        i32(nd.get("mesh", -1), int(hm), int("translation" in nd and not hm), int("scale" in nd and not hm), int("rotation" in nd and not hm))
        f64(nd.get("matrix", [0.0] * 16)); f64(nd.get("translation", [0.0] * 3)); f64(nd.get("scale", [1.0] * 3)); f64(nd.get("rotation", [0.0] * 4))
        kids = nd.get("children", [])
        i32(len(kids), *kids)
    roots = g["scenes"][0]["nodes"]
    i32(len(roots), *roots)
    f64([1.25])
    inp, out = str(tmp_path / "model.bin"), str(tmp_path / "soup.bin")
    open(inp, "wb").write(bytes(blob))
    subprocess.check_call([exe, inp, out])
    gltf = importlib.import_module("prismarine-core_amd.gltf")
    scene = gltf_soup(oracle, gltf.read_gltf(path, mscale=1.25))
    raw = np.fromfile(out, np.uint8)
    n = int(raw[:4].view(np.int32)[0])
    assert n == scene["tris"].shape[0] == 171
    body = raw[4:].view(np.float32)
    assert np.array_equal(bits(body[:9 * n]), bits(scene["tris"].reshape(-1)))
    assert np.array_equal(bits(body[9 * n:18 * n]), bits(scene["normals"].reshape(-1)))
    assert np.array_equal(bits(body[18 * n:24 * n]), bits(scene["texcoords"].reshape(-1)))
    assert np.array_equal(body[24 * n:].view(np.int32), scene["mats"])


def test_tile_sharding_equals_full_frame(psm, ctx, scenes):
    """rendering row tiles separately (bounce loop in lock step on the global ray count)
    and merging texel sums reproduces the unsharded frame."""
    scene = scenes.cornell(open_top=True)
    w, h = 80, 64
    th, rt0, ms, cam = _setup_frame(psm, ctx, scenes, scene, w, h)

    def run(tiles):
        pipes = []
        for (y0, y1) in tiles:
            rt = psm.Pipeline(ctx, seed=5)
            rt.resizeBuffers(w, h)
            rt.resize(w, h)
            rt.setTile(y0, y1)
            rt.camera_matrices(cam[0], cam[1])
            pipes.append(rt)
        gens = [psm.sharded_rounds(rt, th, ms) for rt in pipes]
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
        sums = np.zeros((w * h, 4), np.float32)
        for rt, (y0, y1) in zip(pipes, tiles):
            s, c, f = rt.download_texels()
            sums[y0 * w:y1 * w] = s[y0 * w:y1 * w]
            rt.close()
        return sums

    full = run([(0, h)])
    tiled = run([(0, 13), (13, 40), (40, h)])
    assert full[:, :3].max() > 0.1
    np.testing.assert_allclose(tiled[:, :3], full[:, :3], rtol=1e-5, atol=1e-6)
    assert np.array_equal(tiled[:, 3], full[:, 3])
    rt0.close()
    th.close()


@pytest.mark.parametrize("weights", [None, [1, 2, 2], [3, 1, 0]], ids=["round-robin", "1-2-2", "3-1-0"])
def test_interleaved_tiles_camera_and_gather(psm, ctx, oracle, scenes, weights):
    """8-row bands dealt over 3 ranks (round-robin; weighted 1:2:2 -- the gathering rank gets fewer; 3:1:0 -- a rank
    without any band): camera queue equals the oracle's, and packing each tile's texel sums into a dense buffer +
    unpacking on the gathering pipeline reproduces the unsharded frame."""
    scene = scenes.cornell(open_top=True)
    w, h, world = 72, 52, 3  # 6.5 bands: a partial last band
    th, rt0, ms, cam = _setup_frame(psm, ctx, scenes, scene, w, h)
    cfg = oracle.make_cfg(w, h, material_count=len(scene["materials"]))
    pdist = __import__("importlib").import_module("prismarine-core_amd.dist")

    def lockstep(pipes):
        gens = [psm.sharded_rounds(rt, th, ms) for rt in pipes]
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

    full = psm.Pipeline(ctx, seed=5)
    full.resizeBuffers(w, h); full.resize(w, h)
    full.camera_matrices(cam[0], cam[1])
    lockstep([full])
    want, _, _ = full.download_texels()

    pipes = []
    for r in range(world):
        rt = psm.Pipeline(ctx, seed=5)
        rt.resizeBuffers(w, h); rt.resize(w, h)
        rt.setTileInterleaved(r, world, weights)
        assert rt.tile_texels() == pdist.owned_texels(r, world, w, h, weights)
        rt.camera_matrices(cam[0], cam[1], time=None)
        pipes.append(rt)
    # camera parity per tile (same seed => same `time` as the oracle draws)
    t0, _ = oracle.rand_next(5)
    for r, rt in enumerate(pipes):
        orays, *_ = oracle.camera_interleaved(cfg, cam[0], cam[1], t0, r, world, weights)
        _rays_equal(rt.download_rays(), orays)
    assert sum(rt.tile_texels() for rt in pipes) == w * h
    lockstep(pipes)
    # gather: tiles 1,2 are packed and unpacked into pipeline 0 (the "root")
    per = pdist.largest_tile_texels(world, w, h, weights) * 16
    hb = ctx.buf_alloc(per)
    ptr, _ = ctx.buf_ptr(hb)
    for r in range(1, world):
        pipes[r].pack_texels_dev(ptr)
        pipes[0].unpack_texels_dev(True, r, world, ptr)
    got, _, _ = pipes[0].download_texels()
    assert want[:, :3].max() > 0.1
    np.testing.assert_allclose(got[:, :3], want[:, :3], rtol=1e-5, atol=1e-6)
    assert np.array_equal(got[:, 3], want[:, 3])
    # the same gather the way psm_dist_gather_tiles does it on the gathering rank: every rank's dense tile back to back
    # (what ncclGather delivers, stride = the largest tile), then ONE launch that fills every texel rank 1 does not own
    # -- here into rank 1's pipeline, so that the skipped rank is not always 0
    hall = ctx.buf_alloc(per * world)
    pall, _ = ctx.buf_ptr(hall)
    for r in range(world):
        pipes[r].pack_texels_dev(pall + r * per)
    pipes[1].unpack_tiles_dev(world, 1, pall, per // 4)
    got1, _, _ = pipes[1].download_texels()
    np.testing.assert_allclose(got1[:, :3], want[:, :3], rtol=1e-5, atol=1e-6)
    assert np.array_equal(got1[:, 3], want[:, 3])
    with pytest.raises(psm.PsmError):
        pipes[1].unpack_tiles_dev(world, 1, pall, per // 4 - 4)   # a stride smaller than the largest tile is refused
    ctx.buf_free(hall)
    ctx.buf_free(hb)
    for rt in pipes + [full, rt0]:
        rt.close()
    th.close()


@pytest.mark.parametrize("kw", [{}, {"interleaved": False}, {"index16": True}, {"normals": False},
                                {"quads": True}, {"xform": True}])
def test_load_mesh_matches_oracle(psm, ctx, oracle, scenes, kw):
    """the HIP gather kernel behind loadMesh vs the oracle's loader.comp restatement, bit-exact."""
    kw = dict(kw)
    sc = scenes.sponza_like(n_tris=4000)
    tris = sc["tris"]
    normals = None if kw.pop("normals", True) is False else sc["normals"]
    t = None
    if kw.pop("xform", False):
        a = 0.4
        t = np.array([[np.cos(a), 0, np.sin(a), 0.3], [0, 1.7, 0, -2.0], [-np.sin(a), 0, np.cos(a), 1.0], [0, 0, 0, 1]], np.float32)
    mesh = scenes.make_indexed_mesh(tris, normals, transform=t, material_id=3, **kw)
    opos, onrm, omats = oracle.load_mesh(mesh)
    th = psm.TriangleHierarchy(ctx)
    th.allocate(tris.shape[0] + 16)
    th.loadTriangles(tris[:5], sc["normals"][:5], sc["mats"][:5])  # the mesh is appended after existing triangles
    th.loadMesh(mesh)
    n = opos.shape[0]
    assert th.triangleCount == n + 5
    gp = th.download(psm.BVH_POSITIONS, np.float32, 9 * (n + 5)).reshape(-1, 9)
    gn = th.download(psm.BVH_NORMALS, np.float32, 9 * (n + 5)).reshape(-1, 9)
    gm = th.download(psm.BVH_MATERIALS, np.int32, n + 5)
    assert np.array_equal(bits(gp[5:]), bits(opos)) and np.array_equal(bits(gn[5:]), bits(onrm))
    assert np.array_equal(gm[5:], omats) and np.array_equal(bits(gp[:5]), bits(tris[:5].reshape(5, 9)))
    # and the loaded geometry builds and traces like the soup it came from
    if not kw and t is None and normals is not None:
        th2 = psm.TriangleHierarchy(ctx)
        th2.allocate(n)
        th2.loadMesh(mesh)
        th2.build()
        ob = oracle.build_scene(tris)
        assert np.array_equal(th2.download(psm.BVH_KEYS, np.uint64, ob["count"]), ob["keys"])
        th2.close()
    th.close()


def test_load_mesh_random_descriptions_incl_out_of_range(psm, ctx, oracle):
    """Mesh descriptions now come from files (gltf.read_gltf): 60 seeded random ones -- strides smaller and larger than the
    element, overlapping views, accessor offsets before and past the pool, indices that point past the vertices, 16- and 32-bit
    indices at random loading offsets, quads, pools of odd length -- must read as the oracle's loader reads them (a word outside
    the pool is 0, as a robust GL buffer access gives; vertex/loader.comp:32-54), bit for bit, and must not fault."""
    from util import random_mesh_descriptions
    for case, mesh in enumerate(random_mesh_descriptions(20261005, 60)):
        opos, onrm, omats, otex = oracle.load_mesh(mesh, with_tex=True)
        n = opos.shape[0]
        th = psm.TriangleHierarchy(ctx)
        th.allocate(n)
        th.loadMesh(mesh)
        assert th.triangleCount == n, case
        got = [th.download(w, np.float32, k * n).reshape(n, k) for w, k in ((psm.BVH_POSITIONS, 9), (psm.BVH_NORMALS, 9), (psm.BVH_TEXCOORDS, 6))]
        for g, o, what in zip(got, (opos, onrm, otex), ("positions", "normals", "texcoords")):
            gn, on = np.isnan(g), np.isnan(o)      # a degenerate triangle's face normal is 0 / 0 on both machines
            assert np.array_equal(gn, on), (case, what)
            assert np.array_equal(bits(g)[~on], bits(o)[~on]), (case, what)
        assert np.array_equal(th.download(psm.BVH_MATERIALS, np.int32, n), omats), case
        th.close()


def test_textures_change_the_image_and_slots_can_be_freed(psm, ctx, oracle, scenes):
    """TextureSet slot reuse (TextureSet.inl:42-86): freeing the bump + emissive textures falls back to the
    untextured branches of surface.comp; an unknown / empty slot is ignored (validateTexture, :81-83)."""
    scene = _scene(scenes, "cornell_open+tex")
    w, h = 48, 40
    th, rt, ms, cam = _setup_frame(psm, ctx, scenes, scene, w, h)

    def frame():
        rt.setSeed(5)
        rt.clearSampler()
        psm.render_frame(rt, th, ms, scene["eye"], scene["view"])
        return rt.snapHdr()

    img_all = frame()
    ref_all, _ = oracle.render_frames(scene, w, h, frames=1, seed=5)
    np.testing.assert_allclose(img_all[..., :3], ref_all[..., :3], rtol=1e-4, atol=1e-5)
    ms.texset.freeTexture(2)
    ms.texset.freeTexture(3)
    img_less = frame()
    less = dict(scene)
    less["textures"] = {k: v for k, v in scene["textures"].items() if k not in (2, 3)}
    ref_less, _ = oracle.render_frames(less, w, h, frames=1, seed=5)
    np.testing.assert_allclose(img_less[..., :3], ref_less[..., :3], rtol=1e-4, atol=1e-5)
    assert np.abs(ref_less[..., :3] - ref_all[..., :3]).max() > 0.05
    plain, _ = oracle.render_frames(scenes.cornell(open_top=True), w, h, frames=1, seed=5)
    assert np.abs(plain[..., :3] - ref_all[..., :3]).max() > 0.05
    # a freed slot is reused by the next texture (TextureSet.inl:76-80)
    assert ms.texset.loadTexture(scene["textures"][3]) == 3
    rt.close()
    th.close()


def test_load_mesh_texcoord_accessor(psm, ctx, oracle, scenes):
    """loader.comp:94-99: texcoords read through their accessor, v stored as 1 - v (INVERT_TX_Y)."""
    sc = scenes.sponza_like(n_tris=3000)
    tc = scenes.planar_texcoords(sc["tris"], 0.7)
    for kw in ({}, {"interleaved": False}, {"quads": True}):
        mesh = scenes.make_indexed_mesh(sc["tris"], sc["normals"], texcoords=tc, **kw)
        opos, onrm, omats, otex = oracle.load_mesh(mesh, with_tex=True)
        if not kw:
            assert np.array_equal(bits(otex), bits((np.float32([0, 1]) + np.float32([1, -1]) * tc).reshape(-1, 6)))
        th = psm.TriangleHierarchy(ctx)
        th.allocate(opos.shape[0])
        th.loadMesh(mesh)
        n = opos.shape[0]
        gt = th.download(psm.BVH_TEXCOORDS, np.float32, 6 * n).reshape(n, 6)
        assert np.array_equal(bits(gt), bits(otex)), kw
        assert np.array_equal(bits(th.download(psm.BVH_POSITIONS, np.float32, 9 * n).reshape(n, 9)), bits(opos))
        th.close()


def _multi_scene(scenes):
    """Sponza-class scene cut into three hierarchies; the third repeats every 7th triangle of the others with
    another material, so rays meet equal-distance hits coming from different hierarchies."""
    sc = scenes.textured(scenes.sponza_like(n_tris=20011))
    n = sc["tris"].shape[0]
    dup = np.concatenate([np.arange(0, n, 7), np.arange(0, n, 35)])   # every 35th triangle twice: chains inside part 3
    cat = lambda a: np.concatenate([a, a[dup]])
    ms = dict(sc)
    ms["tris"], ms["normals"], ms["texcoords"] = cat(sc["tris"]), cat(sc["normals"]), cat(sc["texcoords"])
    ms["mats"] = np.concatenate([sc["mats"], (sc["mats"][dup] + 1) % len(sc["materials"])]).astype(np.int32)
    cut = n // 2
    parts = [np.arange(0, cut), np.arange(cut, n), np.arange(n, n + dup.size)]
    return ms, parts


def _load_parts(psm, ctx, scene, parts):
    ths = []
    for ix in parts:
        th = psm.TriangleHierarchy(ctx)
        th.allocate(ix.size)
        th.loadTriangles(scene["tris"][ix], scene["normals"][ix], scene["mats"][ix], scene["texcoords"][ix])
        th.build()
        ths.append(th)
    return ths


def _global_tri(gh, gc, parts):
    """tri | object << 27 (psm_rt_traverse) -> index into the concatenated scene arrays"""
    base = np.array([int(ix[0]) for ix in parts] + [0] * (16 - len(parts)))
    tri = gh["tri"].copy()
    valid = np.arange(8)[None, :] < gc[:, None]
    tri[valid] = base[(tri[valid] >> 27) & 15] + (tri[valid] & ((1 << 27) - 1))
    out = gh.copy()
    out["tri"] = tri
    return out


def test_multi_bvh_chained_intersection(psm, ctx, oracle, scenes):
    """intersection() with several hierarchies over one ray queue (ray.hit handed from call to call,
    directTraverse.comp:219-249,335-346,497-508): hit chains and the shaded queues bit-exact vs the oracle."""
    scene, parts = _multi_scene(scenes)
    w, h = 128, 72
    ths = _load_parts(psm, ctx, scene, parts)
    rt = psm.Pipeline(ctx)
    rt.resizeBuffers(w, h)
    rt.resize(w, h)
    ms = psm.MaterialSet()
    for m in scene["materials"]:
        ms.addSubmat(m)
    ts = psm.TextureSet()
    for slot in sorted(scene["textures"]):
        ts.loadTexture(scene["textures"][slot])
    ms.setTextureSet(ts)
    cam = scenes.camera_matrices(scene["eye"], scene["view"], w, h)
    obs = [oracle.build_scene(scene["tris"][ix]) for ix in parts]
    mats = scenes.materials_array(scene["materials"])
    cfg = oracle.make_cfg(w, h, material_count=len(mats))
    oracle.set_textures(cfg, scene["texcoords"], scene["textures"])
    lights = oracle.default_lights(1)
    rt.camera_matrices(cam[0], cam[1], time=7)
    orays, ocoord, osum, oflag = oracle.camera(cfg, cam[0], cam[1], 7)
    rt.applyMaterials(ms)
    overrides = 0
    for rnd in range(5):
        if orays.shape[0] < 32:
            break
        oh, oc, _ = oracle.traverse(obs[0]["nodes"], scene["tris"][parts[0]], obs[0]["M"], orays["origin"], orays["direct"], 8)
        rt.intersection(ths[0])
        for k in (1, 2):
            before = oh["tri"][:, 0].copy()
            oracle.traverse_chain(obs[k]["nodes"], scene["tris"][parts[k]], obs[k]["M"], orays["origin"], orays["direct"],
                                  oh, oc, int(parts[k][0]), 8)
            overrides += int((before != oh["tri"][:, 0]).sum())
            rt.intersection(ths[k])
            gh, gc = rt.download_hits(orays.shape[0])
            _hits_equal(_global_tri(gh, gc, parts), gc, oh, oc)
        t = 500 + rnd
        rt.shade(time=t)
        orays = oracle.shade(cfg, lights, mats, scene["mats"], scene["tris"], scene["normals"], t, orays, oh, oc, osum, oflag)
        assert rt.raycountCache == orays.shape[0], rnd
        _rays_equal(rt.download_rays(), orays)
    assert rnd >= 2 and overrides > 1000           # later hierarchies really replaced chain heads
    assert (oc > 1).sum() > 0                      # and equal-distance chains across hierarchies occurred
    rt.close()
    for th in ths:
        th.close()


def test_multi_bvh_radiance_and_union_equivalence(psm, ctx, oracle, scenes):
    """Whole frames over three hierarchies: radiance <= 1e-4 vs the oracle; and on a scene without coincident
    triangles the chained result is the single-hierarchy result (same nearest hits)."""
    scene, parts = _multi_scene(scenes)
    w, h, frames = 96, 54, 2
    ths = _load_parts(psm, ctx, scene, parts)
    rt = psm.Pipeline(ctx)
    rt.resizeBuffers(w, h)
    rt.resize(w, h)
    ms = psm.MaterialSet()
    for m in scene["materials"]:
        ms.addSubmat(m)
    ts = psm.TextureSet()
    for slot in sorted(scene["textures"]):
        ts.loadTexture(scene["textures"][slot])
    ms.setTextureSet(ts)

    def render(objs, sc):
        rt.setSeed(21)
        rt.clearSampler()
        for _ in range(frames):
            ms.loadToVGA()
            rt.camera(sc["eye"], sc["view"])
            for _ in range(16):
                if rt.getRayCount() <= 0:
                    break
                for th in objs:
                    rt.intersection(th)
                rt.applyMaterials(ms)
                rt.shade()
            rt.sample()
        return rt.snapHdr()

    img = render(ths, scene)
    ref, st = oracle.render_frames(scene, w, h, frames=frames, seed=21, parts=parts, nthreads=8)
    np.testing.assert_allclose(img[..., :3], ref[..., :3], rtol=1e-4, atol=1e-5)
    for th in ths:
        th.close()
    # no coincident triangles: two hierarchies == one
    plain = scenes.sponza_like(n_tris=20011)
    n = plain["tris"].shape[0]
    halves = [np.arange(0, n // 3), np.arange(n // 3, n)]
    plain["texcoords"] = np.zeros((n, 3, 2), np.float32)
    ms2 = psm.MaterialSet()
    for m in plain["materials"]:
        ms2.addSubmat(m)
    ms, two = ms2, _load_parts(psm, ctx, plain, halves)
    one = _load_parts(psm, ctx, plain, [np.arange(n)])
    a = render(two, plain)
    b = render(one, plain)
    np.testing.assert_allclose(a[..., :3], b[..., :3], rtol=1e-4, atol=1e-5)
    rt.close()
    for th in two + one:
        th.close()


def test_frame_batch_lanes_equal_sequential_frames(psm, oracle, scenes):
    """psm_lanes_render + psm_rt_sample_from: 5 frames on 3 lanes (two batches, the second one short) give the
    image of the same 5 frames rendered one after another -- by the oracle (<= 1e-4) and by one Pipeline on
    the GPU (same deposit counts, radiance to float-atomic order)."""
    scene = scenes.textured(scenes.cornell(open_top=True))
    w, h, frames, seed = 64, 48, 5, 4242
    batch = psm.FrameBatch(3, w, h, seed=seed)
    batch.allocate(scene["tris"].shape[0])
    batch.loadTriangles(scene["tris"], scene["normals"], scene["mats"], scene["texcoords"])
    ms = psm.MaterialSet()
    for m in scene["materials"]:
        ms.addSubmat(m)
    ts = psm.TextureSet()
    for slot in sorted(scene["textures"]):
        ts.loadTexture(scene["textures"][slot])
    ms.setTextureSet(ts)
    batch.applyMaterials(ms)
    per_frame = batch.render(frames, scene["eye"], scene["view"])
    img = batch.snapHdr()
    ref, st = oracle.render_frames(scene, w, h, frames=frames, seed=seed, frame_streams=True)
    np.testing.assert_allclose(img[..., :3], ref[..., :3], rtol=1e-4, atol=1e-5)
    assert np.array_equal(img[..., 3], ref[..., 3])
    assert sum(r for _, r in per_frame) == st["rays"] and sum(n for n, _ in per_frame) == len(st["rounds"])
    # the same frames, one after another, on a single Pipeline
    ln = batch.lanes[1]
    one = ln.rays
    one.clearSampler()
    master = psm.Pipeline(ln.ctx, seed=seed)   # only its rand() stream is used
    for _ in range(frames):
        one.setSeed(master._rand())
        psm.render_frame(one, ln.th, ms, scene["eye"], scene["view"])
    seq = one.snapHdr()
    np.testing.assert_allclose(img[..., :3], seq[..., :3], rtol=1e-5, atol=1e-6)
    assert np.array_equal(img[..., 3], seq[..., 3])
    master.close()
    batch.close()


@pytest.mark.parametrize("lanes,frames,w,h,depth,rebuild", [
    (1, 3, 48, 32, 16, True),     # one frame at a time: every frame's rebuild is queued while the frame before folds
    (2, 1, 48, 32, 16, True),     # fewer frames than lanes
    (4, 2, 48, 32, 16, True),
    (3, 7, 48, 32, 16, True),     # lanes end out of turn and claim their next frames
    (4, 9, 48, 32, 16, False),    # no rebuild: only the waits
    (3, 5, 48, 32, 1, True),      # one bounce round per frame
    (3, 5, 5, 5, 16, True),       # 25 primary rays: a frame ends in camera() (fewer than 32 rays, Pipeline.inl:459-461)
], ids=["1x3", "2x1", "4x2", "3x7", "4x9-norebuild", "depth1", "5x5px"])
def test_lane_scheduler_edge_cases_equal_the_oracle(psm, oracle, scenes, lanes, frames, w, h, depth, rebuild):
    """psm_lanes_render at the corners of its scheduler: the frames are claimed, rebuilt, folded and counted as the same
    frames rendered one after another by the oracle (radiance <= 1e-4, deposit counts, rounds and rays per frame)."""
    scene = scenes.cornell(open_top=True)
    seed = 977
    batch = psm.FrameBatch(lanes, w, h, seed=seed)
    batch.allocate(scene["tris"].shape[0])
    batch.loadTriangles(scene["tris"], scene["normals"], scene["mats"])
    ms = psm.MaterialSet()
    for m in scene["materials"]:
        ms.addSubmat(m)
    batch.applyMaterials(ms)
    if not rebuild:
        for ln in batch.lanes:
            ln.th.build()
    per_frame = batch.render(frames, scene["eye"], scene["view"], depth=depth, rebuild=rebuild)
    img = batch.snapHdr()
    ref, st = oracle.render_frames(scene, w, h, frames=frames, seed=seed, depth=depth, frame_streams=True)
    np.testing.assert_allclose(img[..., :3], ref[..., :3], rtol=1e-4, atol=1e-5)
    assert np.array_equal(img[..., 3], ref[..., 3])
    assert len(per_frame) == frames
    assert sum(r for _, r in per_frame) == st["rays"] and sum(n for n, _ in per_frame) == len(st["rounds"])
    batch.close()


def _three_lights(oracle):
    L = oracle.default_lights(3)
    L[1]["lightVector"] = (-0.5, 0.8, 0.6, 30.0)
    L[1]["lightColor"] = (40.0, 10.0, 5.0, 3.0)
    L[1]["lightOffset"] = (0.2, 0.0, -0.3, 0.0)
    L[2]["lightVector"] = (0.1, -1.0, 0.2, 5.0)
    L[2]["lightColor"] = (2.0, 8.0, 30.0, 1.5)
    L[2]["lightAmbient"] = (0.05, 0.02, 0.01, 0.0)
    return L


def test_three_lights_radiance(psm, ctx, oracle, scenes):
    """setLightCount / lightVector / lightColor / lightOffset (Pipeline.hpp:103-121): three spherical lights of
    different size, one below the horizon (lightCenter's sign flip, shadinglib.glsl:22-26)."""
    scene = scenes.cornell(open_top=True)
    w, h, frames = 64, 48, 2
    th, rt, ms, cam = _setup_frame(psm, ctx, scenes, scene, w, h)
    L = _three_lights(oracle)
    rt.setLights(L)
    rt.setSeed(3)
    for _ in range(frames):
        psm.render_frame(rt, th, ms, scene["eye"], scene["view"])
    img = rt.snapHdr()
    ref, _ = oracle.render_frames(scene, w, h, frames=frames, seed=3, lights=L)
    one, _ = oracle.render_frames(scene, w, h, frames=frames, seed=3)
    assert np.abs(ref[..., :3] - one[..., :3]).max() > 0.1
    np.testing.assert_allclose(img[..., :3], ref[..., :3], rtol=1e-4, atol=1e-5)
    rt.close()
    th.close()


def test_supersampled_ray_grid_and_depth_limit(psm, ctx, oracle, scenes):
    """The viewer traces a ray grid twice the window (Application.hpp:222,277): resizeBuffers(2w, 2h) with
    resize(w, h) -- sampler.comp:37-97 gathers the 2x2 texels whose jittered coordinate lands in a pixel. And the
    `for j < depth` bound of Viewer.cpp:304 cuts frames short."""
    scene = scenes.cornell(open_top=True)
    gw, gh, dw, dh, frames = 96, 64, 48, 32, 2
    th = _load(psm, ctx, scene)
    th.build()
    rt = psm.Pipeline(ctx)
    rt.resizeBuffers(gw, gh)
    rt.resize(dw, dh)
    ms = psm.MaterialSet()
    for m in scene["materials"]:
        ms.addSubmat(m)
    rt.setSeed(3)
    for _ in range(frames):
        psm.render_frame(rt, th, ms, scene["eye"], scene["view"])
    img = rt.snapHdr()
    assert img.shape == (dh, dw, 4)
    ref, st = oracle.render_frames(scene, gw, gh, frames=frames, seed=3, display=(dw, dh))
    np.testing.assert_allclose(img[..., :3], ref[..., :3], rtol=1e-4, atol=1e-5)
    assert np.array_equal(img[..., 3], ref[..., 3]) and ref[..., 3].max() > 2   # several texels per pixel
    # depth limit
    rt.clearSampler()
    rt.setSeed(3)
    rounds = psm.render_frame(rt, th, ms, scene["eye"], scene["view"], depth=2)
    ref2, st2 = oracle.render_frames(scene, gw, gh, frames=1, seed=3, display=(dw, dh), depth=2)
    assert rounds == 2 == len(st2["rounds"]) and len(st["rounds"]) > 4
    np.testing.assert_allclose(rt.snapHdr()[..., :3], ref2[..., :3], rtol=1e-4, atol=1e-5)
    rt.close()
    th.close()


def test_camera_360_mode(psm, ctx, oracle, scenes):
    """switchMode() (Pipeline.inl:128-132) -> cameraUniform.enable360 (camera.comp:48-59): primary rays over the
    whole sphere from the camera position, bit-exact; a frame's radiance within 1e-4; switching back restores the
    projected camera."""
    scene = scenes.cornell(open_top=True)
    w, h = 96, 48
    th, rt, ms, cam = _setup_frame(psm, ctx, scenes, scene, w, h)
    mats = scenes.materials_array(scene["materials"])
    cfg = oracle.make_cfg(w, h, material_count=len(mats))
    rt.switchMode()
    cfg.enable360 = 1
    rt.camera_matrices(cam[0], cam[1], time=21)
    orays, _, _, _ = oracle.camera(cfg, cam[0], cam[1], 21)
    grays = rt.download_rays()
    _rays_equal(grays, orays)
    d = grays["direct"]
    assert np.allclose(np.linalg.norm(d, axis=1), 1.0, atol=1e-4)
    assert (d[:, 1] > 0.5).any() and (d[:, 1] < -0.5).any() and (d[:, 0] > 0.5).any() and (d[:, 0] < -0.5).any()
    assert np.ptp(grays["origin"], axis=0).max() < 1e-6            # every ray starts at the eye
    rt.setSeed(8)
    rt.clearSampler()
    psm.render_frame(rt, th, ms, scene["eye"], scene["view"])
    ref, _ = oracle.render_frames(scene, w, h, frames=1, seed=8, enable360=True)
    np.testing.assert_allclose(rt.snapHdr()[..., :3], ref[..., :3], rtol=1e-4, atol=1e-5)
    rt.switchMode()
    cfg.enable360 = 0
    rt.camera_matrices(cam[0], cam[1], time=21)
    orays, _, _, _ = oracle.camera(cfg, cam[0], cam[1], 21)
    _rays_equal(rt.download_rays(), orays)
    rt.close()
    th.close()


def test_gpu_matches_committed_golden_fixture(psm, oracle, scenes):
    """The HIP path against tests/golden/cornell_open_round1_features.npz (committed oracle output, not recomputed
    here): textures + normal maps, three lights, 2x supersampled ray grid, three frames in flight on two lanes;
    and the 360-degree camera."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cornell_open_round1_features.npz"))
    sct = scenes.textured(scenes.cornell(open_top=True))
    batch = psm.FrameBatch(2, 64, 48, seed=2718, display=(32, 24))
    batch.allocate(sct["tris"].shape[0])
    batch.loadTriangles(sct["tris"], sct["normals"], sct["mats"], sct["texcoords"])
    ms = psm.MaterialSet()
    for m in sct["materials"]:
        ms.addSubmat(m)
    ts = psm.TextureSet()
    for slot in sorted(sct["textures"]):
        ts.loadTexture(sct["textures"][slot])
    ms.setTextureSet(ts)
    batch.applyMaterials(ms)
    L = _three_lights(oracle)
    batch.each(lambda r: r.setLights(L))
    per_frame = batch.render(3, sct["eye"], sct["view"])
    img = batch.snapHdr()
    assert sum(r for _, r in per_frame) == int(g["textured_rays"])
    np.testing.assert_allclose(img[..., :3], g["textured"][..., :3], rtol=1e-4, atol=1e-5)
    assert np.array_equal(img[..., 3], g["textured"][..., 3])
    batch.close()
    ctx = psm.Context(0)
    sc = scenes.cornell(open_top=True)
    th, rt, ms2, cam = _setup_frame(psm, ctx, scenes, sc, 48, 24)
    rt.switchMode()
    rt.setSeed(99)
    psm.render_frame(rt, th, ms2, sc["eye"], sc["view"])
    np.testing.assert_allclose(rt.snapHdr()[..., :3], g["pano"][..., :3], rtol=1e-4, atol=1e-5)
    rt.close()
    th.close()
    ctx.close()


def _sky_image(w=64, h=32):
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.zeros((h, w, 4), np.uint8)
    img[..., 0] = (40 + 200 * xx / (w - 1)).astype(np.uint8)
    img[..., 1] = (255 * yy / (h - 1)).astype(np.uint8)
    img[..., 2] = ((xx * 7 + yy * 13) % 256).astype(np.uint8)
    img[..., 3] = 255
    return img


def test_equirect_skybox_radiance(psm, ctx, oracle, scenes, tmp_path):
    """setSkybox -- equirect RGBA8 lookup (environment.glsl:23-26) + HDR snapshot to PFM."""
    scene = scenes.cornell(open_top=True)
    w, h, frames = 72, 56, 2
    sky = _sky_image()
    th, rt, ms, cam = _setup_frame(psm, ctx, scenes, scene, w, h)
    rt.setSkybox(sky)
    rt.setSeed(99)
    for _ in range(frames):
        psm.render_frame(rt, th, ms, scene["eye"], scene["view"])
    img = rt.snapHdr()
    ref, _ = oracle.render_frames(scene, w, h, frames=frames, seed=99, skybox=sky)
    const, _ = oracle.render_frames(scene, w, h, frames=frames, seed=99)
    assert np.abs(ref[..., :3] - const[..., :3]).max() > 0.05      # the texture really changes the image
    np.testing.assert_allclose(img[..., :3], ref[..., :3], rtol=1e-4, atol=1e-5)
    path = str(tmp_path / "snap.pfm")
    psm.write_pfm(path, img)
    assert np.array_equal(psm.read_pfm(path), img[..., :3])
    rt.setSkybox(None)
    rt.close()
    th.close()


def test_errors_are_reported(psm, ctx):
    th = psm.TriangleHierarchy(ctx)
    th.allocate(4)
    with pytest.raises(psm.PsmError):
        th.loadTriangles(np.zeros((5, 9), np.float32))  # exceeds allocate() capacity
    rt = psm.Pipeline(ctx)
    with pytest.raises(psm.PsmError):
        ctx.check(psm.lib().psm_rt_traverse(rt._h, th._h), "traverse before build")
    rt.resizeBuffers(16, 8)
    rays = np.zeros(4, psm.RAY_DT)
    rays["direct"], rays["bitfield"] = 1.0, 1 | (3 << 8)
    for bad in (16 * 8, -1, 2 ** 31 - 1):     # a ray deposits into the texel it names: one outside the grid is refused at the door
        rays["texel"][2] = bad
        with pytest.raises(psm.PsmError):
            rt.upload_rays(rays)
    rays["texel"][2] = 16 * 8 - 1
    rt.upload_rays(rays)
    rt.close()
    th.close()
    big = psm.TriangleHierarchy(ctx)
    with pytest.raises(psm.PsmError):
        big.allocate((1 << 27) + 1)   # beyond the 32-bit node offsets of the traversal kernel: refused, nothing allocated


def test_cpp_header_layer_viewer_call_order(psm, ctx, oracle, scenes, tmp_path):
    """include/Prismarine drop-in headers: the GltfViewer::process() call order (Viewer.cpp:296-312) in C++
    through psm::TriangleHierarchy / psm::Pipeline gives the oracle's image (camera matrices are computed
    by the header layer's own float glm math, so agreement is statistical, not bitwise)."""
    import os
    import struct
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "viewer_order")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(root, "include"), "-DPSM_NO_SYSTEM_GLM",
                           os.path.join(root, "tests", "cpp", "viewer_order.cpp"), "-o", exe,
                           "-L", os.path.join(root, "prismarine-core_amd"), "-lpsm_hip",
                           "-Wl,-rpath," + os.path.join(root, "prismarine-core_amd")])
    sc = scenes.cornell(open_top=True)
    w, h, frames = 64, 48, 2
    inp, out = str(tmp_path / "scene.bin"), str(tmp_path / "img.bin")
    with open(inp, "wb") as f:
        n = sc["tris"].shape[0]
        f.write(struct.pack("<i", n))
        f.write(sc["tris"].astype("<f4").tobytes())
        f.write(sc["normals"].astype("<f4").tobytes())
        f.write(sc["mats"].astype("<i4").tobytes())
        f.write(struct.pack("<i", len(sc["materials"])))
        for m in sc["materials"]:
            f.write(np.asarray(list(m["diffuse"]) + list(m["specular"]), "<f4").tobytes())
        f.write(np.asarray(sc["eye"], "<f4").tobytes())
        f.write(np.asarray(sc["view"], "<f4").tobytes())
    subprocess.check_call([exe, inp, str(w), str(h), str(frames), out])
    img = np.fromfile(out, np.float32).reshape(h, w, 4)
    ref, _ = oracle.render_frames(sc, w, h, frames=frames, seed=31337)
    assert img[..., :3].max() > 0.1
    assert abs(float(img[..., :3].mean()) - float(ref[..., :3].mean())) < 0.02 * float(ref[..., :3].mean())
    close = np.abs(img[..., :3] - ref[..., :3]).max(-1) < 2e-3 + 1e-2 * ref[..., :3].max(-1)
    assert close.mean() > 0.9
    # the same program through psm::FrameBatch: 5 frames, 3 in flight
    out2 = str(tmp_path / "img_lanes.bin")
    subprocess.check_call([exe, inp, str(w), str(h), "5", out2, "3"])
    img2 = np.fromfile(out2, np.float32).reshape(h, w, 4)
    ref2, _ = oracle.render_frames(sc, w, h, frames=5, seed=31337, frame_streams=True)
    assert abs(float(img2[..., :3].mean()) - float(ref2[..., :3].mean())) < 0.02 * float(ref2[..., :3].mean())
    close2 = np.abs(img2[..., :3] - ref2[..., :3]).max(-1) < 2e-3 + 1e-2 * ref2[..., :3].max(-1)
    assert close2.mean() > 0.9


def test_native_rccl_communicator_single_rank(psm, ctx, scenes):
    """psm_dist_* (RCCL from the C ABI, no torch): a one-rank communicator on the GPU box -- init, the per-frame tile
    gather (pack -> ncclGather -> nothing to unpack at world 1) leaving the image untouched, the int all-gather the
    sharded scheduler uses, barrier, destroy. The N > 1 data flow is the same calls with more peers."""
    pdist = __import__("importlib").import_module("prismarine-core_amd.dist")
    scene = scenes.cornell(open_top=True)
    w, h = 72, 52
    th, rt, ms, cam = _setup_frame(psm, ctx, scenes, scene, w, h)
    nd = pdist.NativeDist(ctx, 0, 1, lambda raw: raw)
    rt.setTileInterleaved(0, 1)
    rt.camera_matrices(cam[0], cam[1], time=9)
    rt.applyMaterials(ms)
    for _ in range(3):
        if rt.intersection(th) == 0:
            break
        rt.shade(time=5)
    before, _, _ = rt.download_texels()
    nd.gather_tiles(rt)
    nd.gather_tiles(rt)
    after, _, _ = rt.download_texels()
    assert before[:, :3].max() > 0.1 and np.array_equal(before, after)
    assert nd.allgather_i32([3, 1, 4, 1, 5]) == [[3, 1, 4, 1, 5]]
    nd.barrier()
    # a Pipeline whose tile is not this communicator's is refused, not gathered wrongly
    rt.setTileInterleaved(1, 3)
    with pytest.raises(psm.PsmError):
        nd.gather_tiles(rt)
    nd.close()
    rt.close()
    th.close()


@pytest.mark.parametrize("lanes,frames", [(4, 7), (1, 3), (5, 5)])
def test_native_sharded_frames_equal_unsharded(psm, ctx, scenes, lanes, frames):
    """psm_dist_render_frames (two alternating lane groups, exchanges, one gather per frame, fold in frame order) on a
    one-rank communicator gives the image psm_lanes_render gives for the same frames, and psm_dist_render_batch too."""
    pdist = __import__("importlib").import_module("prismarine-core_amd.dist")
    scene = scenes.cornell(open_top=True)
    w, h = 64, 48
    ms = psm.MaterialSet()
    for m in scene["materials"]:
        ms.addSubmat(m)

    def make():
        b = psm.FrameBatch(lanes, w, h, seed=77)
        b.allocate(scene["tris"].shape[0])
        b.loadTriangles(scene["tris"], scene["normals"], scene["mats"])
        b.applyMaterials(ms)
        return b
    ref = make()
    ref.render(frames, scene["eye"], scene["view"])
    want = ref.snapHdr()
    ref.close()
    cam = scenes.camera_matrices(scene["eye"], scene["view"], w, h)
    for pipelined in (True, False):
        b = make()
        b.each(lambda r: r.setTileInterleaved(0, 1))
        nd = pdist.NativeDist(b.lanes[0].ctx, 0, 1, lambda raw: raw)
        seeds = b.frame_seeds(frames)
        if pipelined:
            rounds = b.render_frames_sharded(nd, seeds, cam[0], cam[1])
        else:
            rounds = []
            for f0 in range(0, frames, lanes):
                rounds += b.render_batch_sharded(nd, seeds[f0:f0 + lanes], cam[0], cam[1])
        assert len(rounds) == frames and min(rounds) >= 2
        got = b.snapHdr()
        np.testing.assert_allclose(got[..., :3], want[..., :3], rtol=1e-5, atol=1e-6)
        assert np.array_equal(got[..., 3], want[..., 3])
        nd.close()
        b.close()


def test_lifecycle_resizes_tiny_images_and_empty_inputs(psm, oracle, scenes):
    """What a host does to the classes over an application's life (Viewer.cpp:56-63,231-242: resizeBuffers on every
    window resize, clearTribuffer + loadMesh per scene): contexts created and destroyed, the ray grid resized between
    frames (down to 1x1 and up again), a frame whose queue runs empty, an intersection over zero rays, a hierarchy
    cleared and refilled with another triangle count while its build graph exists -- no error, no stale state: the
    frame after all of it equals the oracle's."""
    scene = scenes.cornell(open_top=True)
    for _ in range(3):                                   # contexts come and go
        c = psm.Context(0)
        th = _load(psm, c, scene)
        th.build()
        th.close()
        c.close()
    ctx = psm.Context(0)
    th = _load(psm, ctx, scene)
    ms = psm.MaterialSet()
    for m in scene["materials"]:
        ms.addSubmat(m)
    rt = psm.Pipeline(ctx, seed=11)
    for w, h in [(40, 30), (1, 1), (7, 3), (64, 48), (40, 30)]:
        rt.resizeBuffers(w, h)
        rt.resize(w, h)
        for _ in range(2):
            rounds = psm.render_frame(rt, th, ms, scene["eye"], scene["view"])
            assert 0 <= rounds <= 16
        img = rt.snapHdr()
        assert img.shape[:2] == (h, w) and np.isfinite(img).all()
    # a queue of zero rays: intersection and shade are no-ops, the count stays zero
    rt.upload_rays(np.zeros(0, psm.RAY_DT))
    assert rt.getRayCount() == 0
    rt.intersection(th)
    rt.applyMaterials(ms)
    rt.shade(time=1)
    assert rt.raycountCache == 0
    # the hierarchy refilled with other triangle counts (its captured build graph belongs to the old count)
    for n in (5, scene["tris"].shape[0], 12):
        th.clearTribuffer()
        th.loadTriangles(scene["tris"][:n], scene["normals"][:n], scene["mats"][:n])
        for _ in range(3):
            th.markDirty()
            th.build()
        ob = oracle.build_scene(scene["tris"][:n])
        assert th.info().leaf_count == ob["count"]
        assert np.array_equal(th.download(psm.BVH_KEYS, np.uint64, ob["count"]), ob["keys"])
    # and after all of it a frame that equals the oracle's
    th.clearTribuffer()
    th.loadTriangles(scene["tris"], scene["normals"], scene["mats"])
    w, h = 40, 30
    rt2 = psm.Pipeline(ctx, seed=31337)
    rt2.resizeBuffers(w, h)
    rt2.resize(w, h)
    for _ in range(2):
        psm.render_frame(rt2, th, ms, scene["eye"], scene["view"])
    ref, _ = oracle.render_frames(scene, w, h, frames=2, seed=31337, nthreads=8)
    np.testing.assert_allclose(rt2.snapHdr()[..., :3], ref[..., :3], rtol=1e-4, atol=1e-5)
    rt.close(); rt2.close(); th.close(); ctx.close()
