"""Synthetic scenes and OBJ/MTL io for the path-tracing core.

The reference ships no scene that matches the BASELINE configs (sponza.obj is not in the
repository and its viewer only loads glTF, Source/Examples/Viewer.cpp:66-69), so the
benchmark scenes are procedural:

  cornell()      S-cornell      32 triangles, extents [-1,1]^3 (closed box + 2 blocks)
  sponza_like()  S-sponza-like  262 267 triangles, an atrium ~30 x 12 x 18 units
  stress()       S-stress       ~10 M triangles, jittered blob instances on a grid

A scene is a dict of numpy arrays:
  tris     float32 [N,3,3]  world-space triangle soup (what loader.comp appends,
                            ShadersSDK/vertex/loader.comp:115-135)
  normals  float32 [N,3,3]  per-vertex normals as stored in the normal mosaic (normalised;
                            face normal where the mesh has none, loader.comp:119-128)
  mats     int32   [N]      per-triangle material id (loader.comp:121)
  materials list of dict    diffuse / specular / emissive (VirtualMaterial, Structs.hpp:240-262)
  eye, view float32 [3]     default camera
"""
import math
import logging
import os

import numpy as np

SEED_SPONZA = 0x5EED5A
SEED_STRESS = 0x5EED10
SPONZA_TRIS = 262267


def face_normals(tris):
    e1 = tris[:, 1] - tris[:, 0]
    e2 = tris[:, 2] - tris[:, 0]
    n = np.cross(e1, e2).astype(np.float32)
    ln = np.sqrt((n * n).sum(1, keepdims=True))
    ln[ln == 0] = 1.0
    return (n / ln).astype(np.float32)


def prepare_normals(tris, normals=None):
    """loader.comp:119-128: keep a supplied normal when it is non-trivial, else the face normal."""
    fn = face_normals(tris)
    out = np.repeat(fn[:, None, :], 3, axis=1).astype(np.float32)
    if normals is not None:
        normals = np.asarray(normals, np.float32)
        ok = np.abs(normals).max(axis=2) >= 1e-4
        ln = np.sqrt((normals * normals).sum(2, keepdims=True))
        ln[ln == 0] = 1.0
        nn = (normals / ln).astype(np.float32)
        out = np.where(ok[:, :, None], nn, out).astype(np.float32)
    return np.ascontiguousarray(out)


def _quad(p0, p1, p2, p3):
    return [[p0, p1, p2], [p0, p2, p3]]


def _box(lo, hi, faces="xXyYzZ"):
    x0, y0, z0 = lo
    x1, y1, z1 = hi
    t = []
    if "x" in faces:
        t += _quad((x0, y0, z0), (x0, y0, z1), (x0, y1, z1), (x0, y1, z0))
    if "X" in faces:
        t += _quad((x1, y0, z0), (x1, y1, z0), (x1, y1, z1), (x1, y0, z1))
    if "y" in faces:
        t += _quad((x0, y0, z0), (x1, y0, z0), (x1, y0, z1), (x0, y0, z1))
    if "Y" in faces:
        t += _quad((x0, y1, z0), (x0, y1, z1), (x1, y1, z1), (x1, y1, z0))
    if "z" in faces:
        t += _quad((x0, y0, z0), (x0, y1, z0), (x1, y1, z0), (x1, y0, z0))
    if "Z" in faces:
        t += _quad((x0, y0, z1), (x1, y0, z1), (x1, y1, z1), (x0, y1, z1))
    return t


def _material(diffuse, specular=(0.0, 0.9, 0.0), emissive=(0.0, 0.0, 0.0)):
    # specular.y = roughness, specular.z = metallic (surface.comp:189)
    return {"diffuse": tuple(diffuse) + (1.0,), "specular": (0.0,) + tuple(specular[1:]) + (0.0,),
            "emissive": tuple(emissive) + (1.0,)}


def cornell(open_top=False):
    """32 triangles: box (12) + short block (10) + tall block (10); 4 materials."""
    tris, mats = [], []
    box = _box((-1, -1, -1), (1, 1, 1))
    box_m = [1, 1, 2, 2, 0, 0, 3, 3, 0, 0, 0, 0]  # -x red, +x green, floor, ceiling(light), back, front
    if open_top:
        keep = [i for i in range(12) if i not in (6, 7, 10, 11)]
        box = [box[i] for i in keep]
        box_m = [box_m[i] for i in keep]
    tris += box
    mats += box_m
    b1 = _box((-0.65, -1.0, -0.1), (-0.05, -0.4, 0.5), faces="xXYzZ")
    b2 = _box((0.1, -1.0, -0.7), (0.7, 0.2, -0.1), faces="xXYzZ")
    tris += b1 + b2
    mats += [0] * 20
    if open_top:
        # keep the triangle count at 32 with a thin shelf
        tris += _quad((-1, 0.2, -1), (-1, 0.2, -0.6), (-0.2, 0.2, -0.6), (-0.2, 0.2, -1))
        tris += _quad((0.2, 0.5, -1), (0.2, 0.5, -0.7), (1, 0.5, -0.7), (1, 0.5, -1))
        mats += [0, 0, 0, 0]
    tris = np.asarray(tris, np.float32)
    assert tris.shape[0] == 32
    materials = [_material((0.73, 0.73, 0.73)), _material((0.65, 0.05, 0.05)),
                 _material((0.12, 0.45, 0.15)), _material((0.78, 0.78, 0.78), emissive=(17, 12, 4))]
    eye = (0.0, 0.0, 0.97) if not open_top else (0.0, 1.2, 3.2)
    return {"name": "cornell", "tris": tris, "normals": prepare_normals(tris),
            "mats": np.asarray(mats, np.int32), "materials": materials,
            "eye": np.asarray(eye, np.float32), "view": np.asarray((0.0, -0.1, -1.0), np.float32)}


def _grid_patch(origin, du, dv, nu, nv, height=None):
    """nu x nv quads spanning origin + s*du + t*dv, optional height(s,t)->offset along normal."""
    s = np.linspace(0.0, 1.0, nu + 1, dtype=np.float64)
    t = np.linspace(0.0, 1.0, nv + 1, dtype=np.float64)
    S, T = np.meshgrid(s, t, indexing="ij")
    o = np.asarray(origin, np.float64)
    du = np.asarray(du, np.float64)
    dv = np.asarray(dv, np.float64)
    P = o + S[..., None] * du + T[..., None] * dv
    if height is not None:
        n = np.cross(du, dv)
        n /= np.linalg.norm(n)
        P = P + height(S, T)[..., None] * n
    p00, p10, p11, p01 = P[:-1, :-1], P[1:, :-1], P[1:, 1:], P[:-1, 1:]
    t1 = np.stack([p00, p10, p11], axis=2)
    t2 = np.stack([p00, p11, p01], axis=2)
    return np.concatenate([t1.reshape(-1, 3, 3), t2.reshape(-1, 3, 3)], 0).astype(np.float32)


def _revolve(center, profile_r, profile_y, nseg, a0=0.0, a1=2 * math.pi, axis="y"):
    """Surface of revolution: rings given by (radius, height) profile. Returns tris, normals."""
    ang = np.linspace(a0, a1, nseg + 1)
    r = np.asarray(profile_r, np.float64)[:, None]
    y = np.asarray(profile_y, np.float64)[:, None]
    X = r * np.cos(ang)[None, :]
    Z = r * np.sin(ang)[None, :]
    Y = np.broadcast_to(y, X.shape)
    P = np.stack([X, Y, Z], -1)
    dr = np.gradient(np.asarray(profile_r, np.float64))
    dy = np.gradient(np.asarray(profile_y, np.float64))
    nx = dy[:, None] * np.cos(ang)[None, :]
    nz = dy[:, None] * np.sin(ang)[None, :]
    ny = np.broadcast_to(-dr[:, None], nx.shape)
    Nn = np.stack([nx, ny, nz], -1)
    Nn /= np.maximum(np.linalg.norm(Nn, axis=-1, keepdims=True), 1e-12)
    if axis == "x":
        P = P[..., [1, 0, 2]]
        Nn = Nn[..., [1, 0, 2]]
    elif axis == "z":
        P = P[..., [0, 2, 1]]
        Nn = Nn[..., [0, 2, 1]]
    P = P + np.asarray(center, np.float64)

    def quads(A):
        a00, a10, a11, a01 = A[:-1, :-1], A[1:, :-1], A[1:, 1:], A[:-1, 1:]
        t1 = np.stack([a00, a11, a10], axis=2)
        t2 = np.stack([a00, a01, a11], axis=2)
        return np.concatenate([t1.reshape(-1, 3, 3), t2.reshape(-1, 3, 3)], 0)

    return quads(P).astype(np.float32), quads(Nn).astype(np.float32)


def sponza_like(n_tris=SPONZA_TRIS, seed=SEED_SPONZA):
    """Procedural atrium: floor, two storeys of colonnades, arches, curtains, vases; open roof."""
    rng = np.random.RandomState(seed & 0x7FFFFFFF)
    T, Nr, Mt = [], [], []

    def add(tris, mat, normals=None):
        T.append(tris)
        Nr.append(prepare_normals(tris, normals))
        Mt.append(np.full(tris.shape[0], mat, np.int32))

    LX, LZ, H1, H2 = 15.0, 9.0, 5.0, 10.0
    # floor with slight tile relief, gallery floors, outer walls, roof rim (open centre)
    add(_grid_patch((-LX, 0, -LZ), (2 * LX, 0, 0), (0, 0, 2 * LZ), 96, 60,
                    height=lambda s, t: -0.01 * ((np.floor(s * 48) + np.floor(t * 30)) % 2)), 0)
    for z0, z1 in ((-LZ, -LZ + 3.2), (LZ - 3.2, LZ)):
        add(_grid_patch((-LX, H1, z0), (2 * LX, 0, 0), (0, 0, z1 - z0), 64, 8), 1)
        add(_grid_patch((-LX, H1 - 0.3, z1 if z0 < 0 else z0), (2 * LX, 0, 0), (0, 0.3, 0), 64, 1), 1)
        add(_grid_patch((-LX, H2, z0), (2 * LX, 0, 0), (0, 0, z1 - z0), 64, 8), 1)
    add(_grid_patch((-LX, 0, -LZ), (0, H2 + 2, 0), (0, 0, 2 * LZ), 24, 36), 1)
    add(_grid_patch((LX, 0, -LZ), (0, 0, 2 * LZ), (0, H2 + 2, 0), 36, 24), 1)
    add(_grid_patch((-LX, 0, -LZ), (2 * LX, 0, 0), (0, H2 + 2, 0), 60, 24), 1)
    add(_grid_patch((-LX, 0, LZ), (0, H2 + 2, 0), (2 * LX, 0, 0), 24, 60), 1)
    # colonnades: 2 storeys x 2 rows x 12 columns, fluted profile
    cols_x = np.linspace(-LX + 2.0, LX - 2.0, 12)
    for storey, (yb, yt) in enumerate(((0.0, H1 - 0.3), (H1, H2 - 0.3))):
        for zc in (-LZ + 3.2, LZ - 3.2):
            for xc in cols_x:
                ys = np.concatenate([[0, 0.15, 0.3], np.linspace(0.35, 0.9, 12), [0.93, 0.97, 1.0]])
                rs = np.concatenate([[0.55, 0.55, 0.42], 0.36 - 0.05 * np.linspace(0, 1, 12), [0.42, 0.5, 0.5]])
                t, n = _revolve((xc, yb, zc), rs, yb * 0 + ys * (yt - yb), 40)
                add(t, 2, n)
    # arches between neighbouring columns (half tori) on both storeys
    for yb in (H1 - 0.3 - 1.2, H2 - 0.3 - 1.2):
        for zc in (-LZ + 3.2, LZ - 3.2):
            for xa, xb in zip(cols_x[:-1], cols_x[1:]):
                R = 0.5 * (xb - xa) - 0.45
                k = np.linspace(0, math.pi, 17)
                # tube of radius 0.18 swept on a half circle in the x-y plane
                ring = np.linspace(0, 2 * math.pi, 11)
                cx = 0.5 * (xa + xb)
                P = np.zeros((17, 11, 3))
                Nn = np.zeros((17, 11, 3))
                for i, a in enumerate(k):
                    c = np.array([cx + R * math.cos(a), yb + 0.6 * R * math.sin(a), zc])
                    e = np.array([math.cos(a), 0.6 * math.sin(a), 0.0])
                    e /= np.linalg.norm(e)
                    P[i] = c + 0.18 * (np.cos(ring)[:, None] * e + np.sin(ring)[:, None] * np.array([0, 0, 1.0]))
                    Nn[i] = np.cos(ring)[:, None] * e + np.sin(ring)[:, None] * np.array([0, 0, 1.0])
                a00, a10, a11, a01 = P[:-1, :-1], P[1:, :-1], P[1:, 1:], P[:-1, 1:]
                n00, n10, n11, n01 = Nn[:-1, :-1], Nn[1:, :-1], Nn[1:, 1:], Nn[:-1, 1:]
                tt = np.concatenate([np.stack([a00, a10, a11], 2).reshape(-1, 3, 3),
                                     np.stack([a00, a11, a01], 2).reshape(-1, 3, 3)], 0).astype(np.float32)
                nn = np.concatenate([np.stack([n00, n10, n11], 2).reshape(-1, 3, 3),
                                     np.stack([n00, n11, n01], 2).reshape(-1, 3, 3)], 0).astype(np.float32)
                add(tt, 2, nn)
    # draped curtains (height-field patches hanging between the storeys)
    for i in range(8):
        x0 = -LX + 3.0 + i * 3.4
        zc = (-LZ + 3.25) if i % 2 == 0 else (LZ - 3.25)
        ph = rng.uniform(0, 6.28, 3)
        amp = rng.uniform(0.08, 0.22)
        add(_grid_patch((x0, H1 + 0.2, zc), (2.6, 0, 0), (0, 3.8, 0), 72, 56,
                        height=lambda s, t, ph=ph, amp=amp: amp * (np.sin(18 * s + ph[0]) * (0.3 + 0.7 * (1 - t)) +
                                                                   0.35 * np.sin(41 * s + 7 * t + ph[1]))), 3 + (i % 3))
    # vases / spheres with fine tessellation (small triangles: 3 decades of sizes overall)
    for i in range(14):
        xc = rng.uniform(-LX + 2.5, LX - 2.5)
        zc = rng.uniform(-LZ + 4.5, LZ - 4.5)
        sc = rng.uniform(0.25, 0.7)
        ys = np.linspace(0, 1, 33)
        rs = 0.12 + 0.5 * np.sin(np.pi * ys) ** 0.8 * (1 - 0.35 * ys)
        t, n = _revolve((xc, 0.0, zc), sc * rs, sc * 1.6 * ys, 48)
        add(t, 6, n)
    tris = np.concatenate(T, 0)
    normals = np.concatenate(Nr, 0)
    mats = np.concatenate(Mt, 0)
    have = tris.shape[0]
    if have > n_tris:
        tris, normals, mats = tris[:n_tris], normals[:n_tris], mats[:n_tris]
    elif have < n_tris:
        # filler: a finely tessellated hanging banner strip with exactly the missing count
        miss = n_tris - have
        nu = max(1, int(math.sqrt(miss / 2)))
        nv = max(1, (miss // 2) // nu)
        f = _grid_patch((-6.0, H1 + 0.5, 0.0), (12.0, 0, 0), (0, 3.5, 0.4), nu, nv,
                        height=lambda s, t: 0.15 * np.sin(9 * s + 3 * t))
        extra = miss - f.shape[0]
        if extra > 0:
            g = _grid_patch((-6.0, H1 + 0.3, 0.3), (12.0, 0, 0), (0, 0.15, 0), extra, 1)[:extra]
            f = np.concatenate([f, g], 0)
        f = f[:miss]
        tris = np.concatenate([tris, f], 0)
        normals = np.concatenate([normals, prepare_normals(f)], 0)
        mats = np.concatenate([mats, np.full(miss, 4, np.int32)], 0)
    assert tris.shape[0] == n_tris, tris.shape
    materials = [
        _material((0.62, 0.58, 0.52), specular=(0, 0.7, 0.0)),   # floor stone
        _material((0.70, 0.66, 0.58), specular=(0, 0.9, 0.0)),   # walls
        _material((0.74, 0.72, 0.66), specular=(0, 0.6, 0.0)),   # columns / arches
        _material((0.60, 0.08, 0.07), specular=(0, 0.95, 0.0)),  # red curtain
        _material((0.08, 0.25, 0.50), specular=(0, 0.95, 0.0)),  # blue curtain
        _material((0.10, 0.42, 0.12), specular=(0, 0.95, 0.0)),  # green curtain
        _material((0.80, 0.62, 0.25), specular=(0, 0.25, 0.8)),  # brass vases
    ]
    return {"name": "sponza_like", "tris": np.ascontiguousarray(tris, np.float32),
            "normals": np.ascontiguousarray(normals, np.float32),
            "mats": np.ascontiguousarray(mats, np.int32), "materials": materials,
            "eye": np.asarray((-12.5, 4.2, 0.6), np.float32), "view": np.asarray((6.0, 4.6, -0.4), np.float32)}


def stress(n_tris=10_000_000, seed=SEED_STRESS):
    """Jittered copies of a ~4k-triangle blob on a 50x50 grid plus a ground plane."""
    rng = np.random.RandomState(seed & 0x7FFFFFFF)
    ys = np.linspace(0, 1, 33)
    rs = 0.05 + 0.5 * np.sin(np.pi * ys)
    blob, bn = _revolve((0, 0, 0), rs, ys, 62)  # 32*62*2 = 3968 triangles
    per = blob.shape[0]
    ground = _grid_patch((-110, 0, -110), (220, 0, 0), (0, 0, 220), 64, 64)
    copies = (n_tris - ground.shape[0]) // per
    side = int(math.ceil(math.sqrt(copies)))
    T, Nr = [ground], [prepare_normals(ground)]
    for c in range(copies):
        gx, gz = c % side, c // side
        s = rng.uniform(0.8, 1.9)
        a = rng.uniform(0, 6.28)
        R = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]], np.float32)
        off = np.array([(gx - side / 2) * 4.0 + rng.uniform(-0.8, 0.8), 0.0,
                        (gz - side / 2) * 4.0 + rng.uniform(-0.8, 0.8)], np.float32)
        T.append((blob @ R.T) * np.float32(s) + off)
        Nr.append(bn @ R.T)
    tris = np.concatenate(T, 0).astype(np.float32)
    normals = prepare_normals(tris, np.concatenate(Nr, 0))
    mats = (np.arange(tris.shape[0]) // per % 3).astype(np.int32)
    materials = [_material((0.7, 0.7, 0.7)), _material((0.7, 0.3, 0.2)), _material((0.2, 0.4, 0.7))]
    return {"name": "stress", "tris": np.ascontiguousarray(tris), "normals": normals, "mats": mats,
            "materials": materials, "eye": np.asarray((-60, 35, 70), np.float32),
            "view": np.asarray((0, 0, 0), np.float32)}


# ---------------------------------------------------------------------------
# OBJ / MTL io (f1: geometry ingestion; triangles only, v/vn/f, usemtl, Kd/Ks/Ke)
# ---------------------------------------------------------------------------

def write_obj(path, scene):
    """The scene as Wavefront OBJ + MTL: v / vn per corner, vt (u, 1 - stored v) where the scene has texcoords, usemtl runs,
    Kd / Ks / Ke (+ d) per material and map_Kd / map_Ke / map_Ks / map_Bump for its texture parts -- each image as
    `<stem>_tex<slot>.png.npy` (the pre-decoded form read_obj prefers) and, where PIL is importable, as the PNG itself."""
    tris, normals, mats = scene["tris"], scene["normals"], scene["mats"]
    stem = os.path.splitext(path)[0]
    mtl = stem + ".mtl"
    tc = scene.get("texcoords")
    part_stmt = (("diffusePart", "map_Kd"), ("emissivePart", "map_Ke"), ("specularPart", "map_Ks"), ("bumpPart", "map_Bump"))
    for slot, img in (scene.get("textures") or {}).items():
        name = "%s_tex%d.png" % (stem, slot)
        np.save(name + ".npy", np.ascontiguousarray(img, np.uint8))
        try:
            from PIL import Image
            Image.fromarray(np.ascontiguousarray(img, np.uint8), "RGBA").save(name)
        except ImportError:
            pass
    with open(mtl, "w") as f:
        for i, m in enumerate(scene["materials"]):
            f.write("newmtl m%d\nKd %.9g %.9g %.9g\nd %.9g\nKs %.9g %.9g %.9g\nKe %.9g %.9g %.9g\n" % (
                (i,) + tuple(m["diffuse"][:3]) + (m["diffuse"][3],) + tuple(m["specular"][:3]) + tuple(m["emissive"][:3])))
            for part, stmt in part_stmt:
                if m.get(part):
                    f.write("%s %s_tex%d.png\n" % (stmt, os.path.basename(stem), m[part]))
            f.write("\n")
    with open(path, "w") as f:
        f.write("mtllib %s\n" % os.path.basename(mtl))
        for p in tris.reshape(-1, 3):
            f.write("v %.9g %.9g %.9g\n" % tuple(p))
        if tc is not None:
            for p in np.asarray(tc, np.float32).reshape(-1, 2):
                f.write("vt %.9g %.9g\n" % (p[0], np.float32(1.0) - p[1]))
        for p in normals.reshape(-1, 3):
            f.write("vn %.9g %.9g %.9g\n" % tuple(p))
        cur = -1
        for t in range(tris.shape[0]):
            if mats[t] != cur:
                cur = int(mats[t])
                f.write("usemtl m%d\n" % cur)
            a = 3 * t + 1
            if tc is not None:
                f.write("f %d/%d/%d %d/%d/%d %d/%d/%d\n" % (a, a, a, a + 1, a + 1, a + 1, a + 2, a + 2, a + 2))
            else:
                f.write("f %d//%d %d//%d %d//%d\n" % (a, a, a + 1, a + 1, a + 2, a + 2))


log = logging.getLogger("prismarine.scenes")

# MTL statements that name an image, and the VirtualMaterial texture part each one feeds (surface.comp:100-161)
_MTL_MAPS = {"map_kd": "diffusePart", "map_ke": "emissivePart", "map_ks": "specularPart", "map_pr": "specularPart", "map_pm": "specularPart",
             "map_bump": "bumpPart", "bump": "bumpPart", "norm": "bumpPart"}
# statements the path has no use for: counted, logged, never silently dropped
_MTL_KNOWN_UNUSED = ("ka", "ns", "ni", "illum", "tf", "tr", "sharpness", "map_ka", "map_d", "map_ns", "disp", "decal", "refl", "pc", "pcr", "ps", "aniso", "anisor")
_OBJ_KNOWN_UNUSED = ("o", "g", "s", "vp", "l", "p", "cstype", "deg", "curv", "curv2", "surf", "parm", "trim", "hole", "scrv", "sp", "end", "con", "mg", "bevel",
                     "c_interp", "d_interp", "lod", "shadow_obj", "trace_obj", "maplib", "usemap")


def load_image_rgba8(path):
    """An image as uint8 [h, w, 4], row 0 = the image's top row. `<path>.npy` (a pre-decoded copy: what travels to a GPU box,
    where nothing is decoded) wins; otherwise the file is decoded with PIL where PIL is importable."""
    if os.path.exists(path + ".npy"):
        a = np.load(path + ".npy")
    else:
        try:
            from PIL import Image
        except ImportError as e:
            raise FileNotFoundError("%s: no pre-decoded %s.npy beside it and no PIL to decode it with" % (path, os.path.basename(path))) from e
        a = np.asarray(Image.open(path).convert("RGBA"))
    if a.ndim != 3 or a.shape[2] != 4 or a.dtype != np.uint8:
        raise ValueError("%s: expected uint8 [h, w, 4], got %s %s" % (path, a.dtype, a.shape))
    return np.ascontiguousarray(a)


def read_mtl(path, ignored=None):
    """Wavefront MTL -> (name -> index, [material dicts], {index: {part: image file}}).
    Kd -> diffuse rgb, d -> diffuse alpha, Ke -> emissive, Ks -> specular as the path reads it (.y roughness, .z metallic:
    surface.comp:189; what write_obj writes back), Pr / Pm (the PBR extension) -> roughness / metallic over Ks; map_Kd, map_Ke,
    map_Ks (map_Pr, map_Pm), map_Bump (bump, norm) -> the diffuse / emissive / specular / bump texture part (options such as
    `-bm 1` in front of the file name are skipped). Everything else lands in `ignored` (statement -> count)."""
    names, mats, maps = {}, [], {}
    cur = None
    ignored = {} if ignored is None else ignored
    if not os.path.exists(path):
        ignored["mtllib (file not found: %s)" % os.path.basename(path)] = ignored.get("mtllib (file not found: %s)" % os.path.basename(path), 0) + 1
        return names, mats, maps
    for ln, line in enumerate(open(path, errors="replace"), 1):
        p = line.split("#", 1)[0].split()
        if not p:
            continue
        key = p[0].lower()
        try:
            if key == "newmtl":
                cur = _material((0.8, 0.8, 0.8))
                names[" ".join(p[1:])] = len(mats)
                mats.append(cur)
            elif cur is None:
                raise ValueError("statement before the first newmtl")
            elif key == "kd":
                cur["diffuse"] = tuple(map(float, p[1:4])) + (cur["diffuse"][3],)
            elif key == "d":
                cur["diffuse"] = cur["diffuse"][:3] + (float(p[-1]),)
            elif key == "ks":
                cur["specular"] = tuple(map(float, p[1:4])) + (0.0,)
            elif key == "pr":
                cur["specular"] = (cur["specular"][0], float(p[1]), cur["specular"][2], 0.0)
            elif key == "pm":
                cur["specular"] = (cur["specular"][0], cur["specular"][1], float(p[1]), 0.0)
            elif key == "ke":
                cur["emissive"] = tuple(map(float, p[1:4])) + (1.0,)
            elif key in _MTL_MAPS:
                maps.setdefault(len(mats) - 1, {})[_MTL_MAPS[key]] = p[-1]    # the file name is the last token (options precede it)
            else:
                tag = key if key in _MTL_KNOWN_UNUSED else key + " (unknown)"
                ignored["mtl " + tag] = ignored.get("mtl " + tag, 0) + 1
        except (ValueError, IndexError) as e:
            raise ValueError("%s:%d: malformed `%s` (%s)" % (path, ln, line.strip(), e)) from e
    return names, mats, maps


def read_obj(path):
    """Wavefront OBJ -> a scene dict. v / vt / vn / f (v, v/vt, v//vn, v/vt/vn; negative = relative indices; polygons as
    fans), mtllib / usemtl; texcoords are stored as (u, 1 - v) -- the loader's INVERT_TX_Y, vertex/loader.comp:97-99 -- and only
    where the file has a `vt` (a face without them gets 0, 0); images named by the materials are loaded into texture slots
    1.. in order of first use (TextureSet.inl:42-86: slot 0 = none) and the materials' *Part fields point at them.
    A malformed face (fewer than three corners, an index that is 0 or out of range) raises ValueError with file and line;
    statements the path does not use are counted in scene["ignored"] and logged, not dropped in silence."""
    V, VT, VN, T, TT, TN, M = [], [], [], [], [], [], []
    names, materials, maps = {}, [], {}
    ignored = {}
    cur = 0
    base = os.path.dirname(path)

    def resolve(tok, n, what, ln):
        i = int(tok)
        j = i - 1 if i > 0 else n + i
        if i == 0 or j < 0 or j >= n:
            raise ValueError("%s:%d: %s index %d out of range (1..%d)" % (path, ln, what, i, n))
        return j

    for ln, line in enumerate(open(path, errors="replace"), 1):
        p = line.split("#", 1)[0].split()
        if not p:
            continue
        key = p[0]
        try:
            if key == "v":
                V.append(tuple(map(float, p[1:4])))
                if len(V[-1]) != 3:
                    raise ValueError("a vertex needs three coordinates")
            elif key == "vt":
                VT.append((float(p[1]), float(p[2]) if len(p) > 2 else 0.0))
            elif key == "vn":
                VN.append(tuple(map(float, p[1:4])))
            elif key == "mtllib":
                n2, m2, mp2 = read_mtl(os.path.join(base, " ".join(p[1:])), ignored)
                for k, v in mp2.items():
                    maps[len(materials) + k] = v
                for k, v in n2.items():
                    names.setdefault(k, len(materials) + v)
                materials += m2
            elif key == "usemtl":
                nm = " ".join(p[1:])
                if nm not in names:
                    names[nm] = len(materials)
                    materials.append(_material((0.8, 0.8, 0.8)))
                    ignored["usemtl of an undefined material"] = ignored.get("usemtl of an undefined material", 0) + 1
                cur = names[nm]
            elif key == "f":
                if len(p) < 4:
                    raise ValueError("a face needs at least three corners")
                idx = []
                for tok in p[1:]:
                    q = tok.split("/")
                    vi = resolve(q[0], len(V), "vertex", ln)
                    ti = resolve(q[1], len(VT), "texcoord", ln) if len(q) > 1 and q[1] else -1
                    ni = resolve(q[2], len(VN), "normal", ln) if len(q) > 2 and q[2] else -1
                    idx.append((vi, ti, ni))
                for k in range(1, len(idx) - 1):
                    tri = (idx[0], idx[k], idx[k + 1])
                    T.append([V[a] for a, _, _ in tri])
                    TT.append([(VT[b][0], 1.0 - VT[b][1]) if b >= 0 else (0.0, 0.0) for _, b, _ in tri])
                    TN.append([VN[c] if c >= 0 else (0.0, 0.0, 0.0) for _, _, c in tri])
                    M.append(cur)
            else:
                tag = key if key in _OBJ_KNOWN_UNUSED else key + " (unknown)"
                ignored["obj " + tag] = ignored.get("obj " + tag, 0) + 1
        except (ValueError, IndexError) as e:
            if str(e).startswith(path):
                raise
            raise ValueError("%s:%d: malformed `%s` (%s)" % (path, ln, line.strip(), e)) from e
    if not T:
        raise ValueError("%s: no faces" % path)
    if not materials:
        materials = [_material((0.8, 0.8, 0.8))]
    # images -> texture slots, in order of first use; a material keeps the slot number in its *Part field
    textures, slot_of = {}, {}
    for mi in sorted(maps):
        for part, fname in maps[mi].items():
            full = os.path.join(base, fname.replace("\\", "/"))
            if full not in slot_of:
                if len(textures) >= 31:
                    raise ValueError("%s: more than 31 images (samplers[MAX_TEXTURES], surface.comp:46-52)" % path)
                slot_of[full] = len(textures) + 1
                textures[slot_of[full]] = load_image_rgba8(full)
            materials[mi] = dict(materials[mi], **{part: slot_of[full]})
    tris = np.asarray(T, np.float32).reshape(-1, 3, 3)
    normals = prepare_normals(tris, np.asarray(TN, np.float32).reshape(-1, 3, 3))
    lo, hi = tris.reshape(-1, 3).min(0), tris.reshape(-1, 3).max(0)
    c, ext = 0.5 * (lo + hi), float((hi - lo).max())
    if ignored:
        log.info("%s: statements the path does not use: %s", path, ", ".join("%s x%d" % kv for kv in sorted(ignored.items())))
    sc = {"name": os.path.basename(path), "tris": tris, "normals": normals,
          "mats": np.asarray(M, np.int32), "materials": materials, "ignored": ignored,
          "eye": (c + np.asarray((0.0, 0.6 * ext, 1.4 * ext))).astype(np.float32),
          "view": c.astype(np.float32)}
    if VT:
        sc["texcoords"] = np.asarray(TT, np.float32).reshape(-1, 3, 2)
    if textures:
        sc["textures"] = textures
    return sc


def materials_array(materials):
    """VirtualMaterial[count] as a structured numpy array (128 B each, Structs.hpp:240-262)."""
    dt = np.dtype([("diffuse", "<f4", 4), ("specular", "<f4", 4), ("transmission", "<f4", 4),
                   ("emissive", "<f4", 4), ("ior", "<f4"), ("roughness", "<f4"), ("alpharef", "<f4"),
                   ("unk0f", "<f4"), ("diffusePart", "<u4"), ("specularPart", "<u4"),
                   ("bumpPart", "<u4"), ("emissivePart", "<u4"), ("flags", "<i4"), ("alphafunc", "<i4"),
                   ("binding", "<i4"), ("bitfield", "<i4"), ("iModifiers0", "<i4", 4)])
    assert dt.itemsize == 128
    a = np.zeros(len(materials), dt)
    for i, m in enumerate(materials):
        a[i]["diffuse"] = m["diffuse"]
        a[i]["specular"] = m["specular"]
        a[i]["emissive"] = m["emissive"]
        a[i]["ior"] = 1.0
        a[i]["roughness"] = 0.0001
        for part in ("diffusePart", "specularPart", "bumpPart", "emissivePart"):
            a[i][part] = m.get(part, 0)
    return a


# ---------------------------------------------------------------------------
# procedural textures + planar texcoords (no image files travel with the repo)
# ---------------------------------------------------------------------------

def planar_texcoords(tris, scale=1.5):
    """Box-projected u,v per vertex [n,3,2]: drop the face normal's dominant axis (keeps both signs and
    values outside [0,1) so GL_REPEAT is exercised)."""
    t = np.asarray(tris, np.float32).reshape(-1, 3, 3)
    n = np.cross((t[:, 1] - t[:, 0]).astype(np.float64), (t[:, 2] - t[:, 0]).astype(np.float64))
    ax = np.argmax(np.abs(n), 1)
    ua = np.where(ax == 0, 1, 0)
    va = np.where(ax == 2, 1, 2)
    idx = np.arange(t.shape[0])
    uv = np.stack([t[idx, :, ua], t[idx, :, va]], -1)
    return (uv * np.float32(scale)).astype(np.float32)


def procedural_textures():
    """{slot: uint8 [h,w,4]}: 1 colour checker with alpha, 2 tangent-space normal map, 3 emissive spots,
    4 roughness/metallic map (.y/.z, surface.comp:189), 5 grey height map (surface.comp:141-148)."""
    def grid(w, h):
        y, x = np.mgrid[0:h, 0:w]
        return (x + 0.5) / w, (y + 0.5) / h
    out = {}
    u, v = grid(64, 32)
    chk = ((np.floor(u * 8) + np.floor(v * 4)) % 2)
    out[1] = np.stack([0.25 + 0.6 * chk, 0.3 + 0.5 * u, 0.8 - 0.5 * v, 0.55 + 0.45 * chk], -1)
    u, v = grid(32, 32)
    nx, ny = 0.45 * np.sin(2 * np.pi * u * 3), 0.45 * np.cos(2 * np.pi * v * 2)
    nz = np.sqrt(np.maximum(1 - nx * nx - ny * ny, 0))
    out[2] = np.stack([0.5 + 0.5 * nx, 0.5 + 0.5 * ny, 0.5 + 0.5 * nz, np.ones_like(nx)], -1)
    u, v = grid(16, 16)
    spot = (((u - 0.5) ** 2 + (v - 0.5) ** 2) < 0.09).astype(np.float64)
    out[3] = np.stack([spot, 0.8 * spot, 0.5 * spot, np.ones_like(spot)], -1)
    u, v = grid(8, 24)
    out[4] = np.stack([np.zeros_like(u), 0.2 + 0.7 * v, 0.9 * (u > 0.5), np.ones_like(u)], -1)
    u, v = grid(48, 40)
    hgt = 0.5 + 0.5 * np.sin(2 * np.pi * u * 2) * np.sin(2 * np.pi * v * 3)
    out[5] = np.stack([hgt, hgt, hgt, np.ones_like(hgt)], -1)
    return {k: np.ascontiguousarray(np.clip(np.rint(a * 255.0), 0, 255).astype(np.uint8)) for k, a in out.items()}


def textured(scene, scale=1.5):
    """A copy of `scene` with box-projected texcoords, the procedural texture table and texture parts spread
    over its materials (every combination of diffuse / specular / bump / emissive part occurs)."""
    sc = dict(scene)
    sc["texcoords"] = planar_texcoords(scene["tris"], scale)
    sc["textures"] = procedural_textures()
    combos = [{"diffusePart": 1, "bumpPart": 2}, {"bumpPart": 5, "specularPart": 4},
              {"diffusePart": 1, "emissivePart": 3}, {"specularPart": 4, "bumpPart": 2, "emissivePart": 3}]
    mats = []
    for i, m in enumerate(scene["materials"]):
        mm = dict(m)
        mm.update(combos[i % len(combos)])
        mats.append(mm)
    sc["materials"] = mats
    sc["name"] = scene["name"] + "+tex"
    return sc


# ---------------------------------------------------------------------------
# camera matrices (Pipeline.inl:298-312: perspective(pi/3, aspect, 0.001, 1000) x lookAt)
# ---------------------------------------------------------------------------

# The reference computes its camera matrices on the host with glm in single precision (Pipeline.inl:283-284,298-312):
# lookAt, perspective and inverse below restate glm's operation order (gtc/matrix_transform.inl lookAtRH /
# perspectiveRH with the -1..1 depth range, detail/func_matrix.inl compute_inverse<4,4>) in float32, so the
# matrices are bit-for-bit the ones the reference would upload (pinned by tests/golden/glm_host_formulas.npz,
# generated with the reference's vendored glm).
_F = np.float32


def _dot3(a, b):
    return _F(_F(_F(a[0] * b[0]) + _F(a[1] * b[1])) + _F(a[2] * b[2]))


def _cross3(x, y):
    return np.array([_F(_F(x[1] * y[2]) - _F(y[1] * x[2])), _F(_F(x[2] * y[0]) - _F(y[2] * x[0])),
                     _F(_F(x[0] * y[1]) - _F(y[0] * x[1]))], np.float32)


def _normalize3(v):
    inv = _F(_F(1.0) / _F(np.sqrt(_dot3(v, v))))
    return np.array([_F(v[0] * inv), _F(v[1] * inv), _F(v[2] * inv)], np.float32)


def look_at(eye, center, up=(0.0, 1.0, 0.0)):
    """glm::lookAt (right-handed) as the row-major matrix M with M @ (p, 1) = view-space p."""
    eye, center, up = (np.asarray(v, np.float32) for v in (eye, center, up))
    f = _normalize3((center - eye).astype(np.float32))
    s = _normalize3(_cross3(f, up))
    u = _cross3(s, f)
    m = np.eye(4, dtype=np.float32)
    m[0, :3], m[1, :3], m[2, :3] = s, u, -f
    m[0, 3], m[1, 3], m[2, 3] = -_dot3(s, eye), -_dot3(u, eye), _dot3(f, eye)
    return m


def perspective(fovy, aspect, znear, zfar):
    """glm::perspective (right-handed, depth -1..1), row-major, all in float32 as glm evaluates it."""
    fovy, aspect, znear, zfar = _F(fovy), _F(aspect), _F(znear), _F(zfar)
    t = _F(math.tan(float(_F(fovy / _F(2.0)))))  # std::tan(float): correctly rounded on the platforms checked
    m = np.zeros((4, 4), np.float32)
    m[0, 0] = _F(_F(1.0) / _F(aspect * t))
    m[1, 1] = _F(_F(1.0) / t)
    m[2, 2] = -_F(_F(zfar + znear) / _F(zfar - znear))
    m[3, 2] = -1.0
    m[2, 3] = -_F(_F(_F(_F(2.0) * zfar) * znear) / _F(zfar - znear))
    return m


def inverse4(mat):
    """glm::inverse(mat4) in float32 with glm's operation order; `mat` and the result are row-major."""
    m = np.ascontiguousarray(np.asarray(mat, np.float32).T)  # m[c][r] = glm's column-major indexing

    def d(a, b, c, e):
        return _F(_F(a * b) - _F(c * e))
    c00 = d(m[2][2], m[3][3], m[3][2], m[2][3]); c02 = d(m[1][2], m[3][3], m[3][2], m[1][3]); c03 = d(m[1][2], m[2][3], m[2][2], m[1][3])
    c04 = d(m[2][1], m[3][3], m[3][1], m[2][3]); c06 = d(m[1][1], m[3][3], m[3][1], m[1][3]); c07 = d(m[1][1], m[2][3], m[2][1], m[1][3])
    c08 = d(m[2][1], m[3][2], m[3][1], m[2][2]); c10 = d(m[1][1], m[3][2], m[3][1], m[1][2]); c11 = d(m[1][1], m[2][2], m[2][1], m[1][2])
    c12 = d(m[2][0], m[3][3], m[3][0], m[2][3]); c14 = d(m[1][0], m[3][3], m[3][0], m[1][3]); c15 = d(m[1][0], m[2][3], m[2][0], m[1][3])
    c16 = d(m[2][0], m[3][2], m[3][0], m[2][2]); c18 = d(m[1][0], m[3][2], m[3][0], m[1][2]); c19 = d(m[1][0], m[2][2], m[2][0], m[1][2])
    c20 = d(m[2][0], m[3][1], m[3][0], m[2][1]); c22 = d(m[1][0], m[3][1], m[3][0], m[1][1]); c23 = d(m[1][0], m[2][1], m[2][0], m[1][1])
    f0 = np.array([c00, c00, c02, c03], np.float32); f1 = np.array([c04, c04, c06, c07], np.float32)
    f2 = np.array([c08, c08, c10, c11], np.float32); f3 = np.array([c12, c12, c14, c15], np.float32)
    f4 = np.array([c16, c16, c18, c19], np.float32); f5 = np.array([c20, c20, c22, c23], np.float32)
    v0 = np.array([m[1][0], m[0][0], m[0][0], m[0][0]], np.float32); v1 = np.array([m[1][1], m[0][1], m[0][1], m[0][1]], np.float32)
    v2 = np.array([m[1][2], m[0][2], m[0][2], m[0][2]], np.float32); v3 = np.array([m[1][3], m[0][3], m[0][3], m[0][3]], np.float32)
    i0 = (v1 * f0 - v2 * f1) + v3 * f2   # numpy float32 arrays: every product and sum rounds to float32, left to right
    i1 = (v0 * f0 - v2 * f3) + v3 * f4
    i2 = (v0 * f1 - v1 * f3) + v3 * f5
    i3 = (v0 * f2 - v1 * f4) + v2 * f5
    sa = np.array([1, -1, 1, -1], np.float32); sb = np.array([-1, 1, -1, 1], np.float32)
    inv = np.stack([i0 * sa, i1 * sb, i2 * sa, i3 * sb]).astype(np.float32)  # inv[c][r]
    row0 = np.array([inv[0][0], inv[1][0], inv[2][0], inv[3][0]], np.float32)
    dot0 = (m[0] * row0).astype(np.float32)
    det = _F(_F(dot0[0] + dot0[1]) + _F(dot0[2] + dot0[3]))
    return np.ascontiguousarray((inv * _F(_F(1.0) / det)).astype(np.float32).T)


def camera_matrices(eye, view, width, height):
    """Row-major (camInv, projInv) = inverse(lookAt), inverse(perspective) as float32[16]: cameraUniformData.camInv /
    projInv as Pipeline::camera(eye, view) uploads them (Pipeline.inl:283-284,298-312)."""
    cam = look_at(eye, view)
    proj = perspective(_F(math.pi) / _F(3.0), _F(width) / _F(height), 0.001, 1000.0)
    return np.ascontiguousarray(inverse4(cam).reshape(16)), np.ascontiguousarray(inverse4(proj).reshape(16))


# ---------------------------------------------------------------------------
# glTF-style mesh descriptions (accessors / buffer views), the input of loadMesh
# ---------------------------------------------------------------------------

def make_indexed_mesh(tris, normals=None, interleaved=True, index16=False, quads=False, transform=None, material_id=0,
                      texcoords=None):
    """Turn a triangle soup into an indexed, accessor-described mesh (welds identical vertices).
    interleaved: one buffer view with stride 6 holding position|normal per vertex; else two planar views.
    quads: pair consecutive triangles (a,b,c),(d,a,c) into 4-index primitives where possible (tests only)."""
    tris = np.asarray(tris, np.float32).reshape(-1, 3, 3)
    n = tris.shape[0]
    nr = np.zeros_like(tris) if normals is None else np.asarray(normals, np.float32).reshape(-1, 3, 3)
    cols = [tris.reshape(-1, 3), nr.reshape(-1, 3)]
    if texcoords is not None:
        cols.append(np.asarray(texcoords, np.float32).reshape(-1, 2))
    rec = np.concatenate(cols, 1)
    uniq, inv = np.unique(rec, axis=0, return_inverse=True)
    inv = inv.reshape(-1).astype(np.uint32)
    w = rec.shape[1]
    if interleaved:
        verts = uniq.reshape(-1).astype(np.float32)
        views = [(0, w)]
        accessors = [(0, 2, 0), (3, 2, 0)] + ([(6, 1, 0)] if texcoords is not None else [])
    else:
        verts = np.concatenate([uniq[:, :3].reshape(-1), uniq[:, 3:6].reshape(-1), uniq[:, 6:].reshape(-1)]).astype(np.float32)
        views = [(0, 3), (uniq.shape[0] * 3, 0)]  # stride 0 -> components + 1
        accessors = [(0, 2, 0), (0, 2, 1)]
        if texcoords is not None:
            views.append((uniq.shape[0] * 6, 0))
            accessors.append((0, 1, 2))
    node_count, prim = n, 0
    idx = inv
    if quads:
        assert n % 2 == 0
        q = inv.reshape(-1, 2, 3)
        # primitive (i0,i1,i2,i3) expands to triangles (i0,i1,i2) and (i3,i0,i2)  (loader.comp:56)
        idx = np.stack([q[:, 0, 0], q[:, 0, 1], q[:, 0, 2], q[:, 1, 0]], 1).reshape(-1).astype(np.uint32)
        node_count, prim = n // 2, 1
    if index16:
        assert uniq.shape[0] < 65536
        pad = np.concatenate([idx, np.zeros(idx.size % 2, np.uint32)])
        idx = (pad[0::2] | (pad[1::2] << np.uint32(16))).astype(np.uint32)
    t = np.eye(4, dtype=np.float32) if transform is None else np.asarray(transform, np.float32).reshape(4, 4)
    ti = np.linalg.inv(t.astype(np.float64)).astype(np.float32)
    return {"vertices": verts, "indices": idx, "accessors": accessors, "views": views, "vertex_accessor": 0,
            "normal_accessor": -1 if normals is None else 1, "texcoord_accessor": -1 if texcoords is None else 2, "transform": t.reshape(16), "transform_inv": ti.reshape(16),
            "material_id": material_id, "index16": int(index16), "node_count": node_count, "primitive_type": prim,
            "loading_offset": 0}
