"""Helpers shared by the parity tests."""
import numpy as np


def half2_key(p):
    """fp16 pair bits -> per-half sortable keys with -0 < +0 (vectorised, uint32 in/out)."""
    p = p.astype(np.uint32)
    s = p & np.uint32(0x80008000)
    m = (s >> np.uint32(15)) * np.uint32(0xFFFF)
    return p ^ (m | np.uint32(0x80008000))


def key_half2(k):
    k = k.astype(np.uint32)
    s = (~k) & np.uint32(0x80008000)
    m = (s >> np.uint32(15)) * np.uint32(0xFFFF)
    return k ^ (m | np.uint32(0x80008000))


def _min16(a, b):
    lo = np.minimum(a & 0xFFFF, b & 0xFFFF)
    hi = np.minimum(a >> 16, b >> 16)
    return (lo | (hi << 16)).astype(np.uint32)


def _max16(a, b):
    lo = np.maximum(a & 0xFFFF, b & 0xFFFF)
    hi = np.maximum(a >> 16, b >> 16)
    return (lo | (hi << 16)).astype(np.uint32)


def box_union(a, b):
    """a, b: uint32[...,4] packed fp16 boxes (mn.xy, mn.zw, mx.xy, mx.zw)."""
    ka, kb = half2_key(a), half2_key(b)
    out = np.empty_like(ka)
    out[..., 0] = _min16(ka[..., 0], kb[..., 0])
    out[..., 1] = _min16(ka[..., 1], kb[..., 1])
    out[..., 2] = _max16(ka[..., 2], kb[..., 2])
    out[..., 3] = _max16(ka[..., 3], kb[..., 3])
    return key_half2(out)


def canonical_nodes(root, link, pairbox, rng, node_dt):
    """Renumber the split-gap tree of the HIP builder into the canonical BFS HlbvhNode array
    (SURVEY a-9): root 0, the i-th internal node of a level gets children base+2i, base+2i+1,
    internal children enqueued left then right."""
    n_leaf = link.shape[0] + 1
    nodes = np.zeros(2 * n_leaf - 1, node_dt)
    parent = np.full(2 * n_leaf - 1, -1, np.int64)
    cur = np.array([root], np.int64)
    ids = np.array([0], np.int64)
    next_id = 1
    while cur.size:
        L, R = link[cur, 0].astype(np.int64), link[cur, 1].astype(np.int64)
        lid = next_id + 2 * np.arange(cur.size, dtype=np.int64)
        rid = lid + 1
        next_id += 2 * cur.size
        nodes["pdata"][ids, 0] = lid
        nodes["pdata"][ids, 1] = rid
        nodes["pdata"][ids, 3] = -1
        nodes["box"][lid] = pairbox[cur, 0:4]
        nodes["box"][rid] = pairbox[cur, 4:8]
        parent[lid] = ids
        parent[rid] = ids
        for child, cid, pos in ((L, lid, rng[cur, 0]), (R, rid, rng[cur, 1])):
            leaf = child < 0
            nodes["pdata"][cid[leaf], 0] = pos[leaf]
            nodes["pdata"][cid[leaf], 1] = pos[leaf]
            nodes["pdata"][cid[leaf], 3] = ~child[leaf]
        ch = np.stack([L, R], 1).reshape(-1)
        ci = np.stack([lid, rid], 1).reshape(-1)
        keep = ch >= 0
        cur, ids = ch[keep], ci[keep]
    nodes["pdata"][:, 2] = parent
    nodes["box"][0] = box_union(nodes["box"][1], nodes["box"][2])
    return nodes


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def gltf_soup(oracle, gscene):
    """A scene dict (tris / normals / mats / texcoords / materials / textures / eye / view) from a gltf.read_gltf result,
    every instance resolved by the ORACLE's loader restatement (vertex/loader.comp:32-152) in loading order."""
    P, N, M, T = [], [], [], []
    for inst in gscene["instances"]:
        pos, nrm, mats, tex = oracle.load_mesh(inst, with_tex=True)
        P.append(pos); N.append(nrm); M.append(mats); T.append(tex)
    tris = np.concatenate(P).reshape(-1, 3, 3)
    lo, hi = tris.reshape(-1, 3).min(0), tris.reshape(-1, 3).max(0)
    c, ext = 0.5 * (lo + hi), float((hi - lo).max())
    sc = {"name": gscene["name"], "tris": tris, "normals": np.concatenate(N).reshape(-1, 3, 3), "mats": np.concatenate(M).astype(np.int32),
          "texcoords": np.concatenate(T).reshape(-1, 3, 2), "materials": gscene["materials"],
          "eye": (c + np.asarray((0.1 * ext, 0.35 * ext, 0.6 * ext))).astype(np.float32), "view": c.astype(np.float32)}
    if gscene["textures"]:
        sc["textures"] = gscene["textures"]
    return sc


def random_mesh_descriptions(seed, count):
    """Seeded random loadMesh descriptions, sane and not: strides smaller and larger than the element, overlapping views,
    offsets before and past the pool, indices past the vertices, 16- / 32-bit indices at random loading offsets, quads."""
    rng = np.random.RandomState(seed)
    for case in range(count):
        nfl = int(rng.randint(7, 600))
        verts = rng.uniform(-4, 4, nfl).astype(np.float32)
        if case % 7 == 0:
            verts[rng.randint(0, nfl, 3)] = [0.0, -0.0, 1e-30]
        views = [(int(rng.randint(-3, nfl + 5)), int(rng.choice([0, 0, 1, 2, 3, 4, 6, 8, 11, -2]))) for _ in range(int(rng.randint(1, 4)))]
        acc = [(int(rng.randint(-2, 12)), int(rng.choice([0, 1, 2, 2, 2, 3])), int(rng.randint(0, len(views)))) for _ in range(int(rng.randint(1, 5)))]
        quads = bool(rng.randint(0, 4) == 0)
        nodes = int(rng.randint(1, 40))
        indexed = bool(rng.randint(0, 3))
        idx16 = bool(rng.randint(0, 2))
        idx = None
        if indexed:
            nw = int(rng.randint(1, 80))
            hi = 65536 if idx16 and rng.randint(0, 3) == 0 else max(nfl // 2, 2)
            idx = (rng.randint(0, hi, nw).astype(np.uint32) | (rng.randint(0, hi, nw).astype(np.uint32) << 16 if idx16 else 0)).astype(np.uint32)
        a = rng.uniform(0, 6.28)
        t = np.array([[np.cos(a), 0.1, np.sin(a), rng.uniform(-2, 2)], [0, rng.uniform(0.3, 2), 0, rng.uniform(-2, 2)],
                      [-np.sin(a), 0, np.cos(a), rng.uniform(-2, 2)], [0, 0, 0, 1]], np.float32)
        yield {"vertices": verts, "indices": idx, "accessors": acc, "views": views,
               "vertex_accessor": int(rng.randint(0, len(acc))), "normal_accessor": int(rng.randint(-1, len(acc))),
               "texcoord_accessor": int(rng.randint(-1, len(acc))), "transform": t.reshape(16),
               "transform_inv": np.linalg.inv(t.astype(np.float64)).astype(np.float32).reshape(16), "material_id": int(rng.randint(-1, 5)),
               "index16": int(idx16), "node_count": nodes, "primitive_type": int(quads), "loading_offset": int(rng.randint(0, 30))}
