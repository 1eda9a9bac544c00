"""glTF 2.0 scenes, loaded the way the reference's viewer loads them (Source/Examples/Viewer.cpp:66-279) -- host code only.

The reference's example application reads nothing but glTF: tinygltf parses the file, every glTF buffer becomes one GL buffer,
every bufferView a VirtualBufferView, every primitive a TriangleArrayInstance whose POSITION / NORMAL / TEXCOORD_0 accessors become
VirtualAccessors, materials become VirtualMaterials, and the node tree is walked with glm double matrices, one
TriangleHierarchy::loadMesh per (node, primitive). `read_gltf` restates that loop and returns what it would have handed to
loadMesh -- mesh descriptions in the form TriangleHierarchy.loadMesh (and the oracle's loader restatement) take -- so the
de-indexing, the accessor reads, the transform and the normal fallback are done by the HIP gather kernel behind psm_bvh_load_mesh,
as the reference does them in vertex/loader.comp.

Faithful to the viewer, including what a careful loader would do differently (`as_viewer=False` does those the careful way):
  * a node's matrix is  parent * (matrix * T * S * R)  (Viewer.cpp:246-253; glTF itself says T * R * S), every product a full glm
    dmat4 product; the root's parent is scale(mscale) (:240-241); the result is narrowed to float where setTransform takes it
    (VertexInstance.inl:54-58). The double algebra below follows glm's operation order and is pinned bit for bit against the
    reference's vendored glm (tests/golden/glm_gltf_transforms.npz, tests/golden/make_glm_golden.py);
  * a node that has a mesh does not pass on to its children, and the walk stops four levels below a scene root (:254-268);
  * only scene 0 is loaded (:271-277), only TRIANGLES primitives (:225), only POSITION, NORMAL and TEXCOORD_0 (:171-193);
  * every accessor of a primitive reads the buffer of its POSITION view, the indices the buffer of their own view (:176, :203);
  * offsets are in 4-byte units, rounded down (:144-145, :170); 16-bit indices by component type, anything else is read as
    32-bit (:205-213);
  * a primitive without indices keeps MeshUniformStruct's nodeCount = 1: one triangle (Structs.hpp:218);
  * a primitive without a material has material id -1, which shades as "no material" (surface.comp);
  * emissiveFactor goes into VirtualMaterial.emissive as the viewer puts it there (:114-120) -- and lights nothing: the path takes
    emission from the emissive TEXTURE only (fetchEmissive, surface.comp:104-110);
  * a texture whose image cannot be read is slot 0, "none" (TextureSet.inl:88-101).
The image files a scene names are decoded by scenes.load_image_rgba8 (a pre-decoded `<file>.npy` wins, so nothing is decoded on
a GPU box) and handed over bottom row first, the order FreeImage gives the reference's glTextureSubImage2D (TextureSet.inl:103-118).
"""
import base64
import json
import logging
import os

import numpy as np

from . import scenes

log = logging.getLogger("prismarine.gltf")

_F32 = np.float32
_COMPONENTS = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT2": 4, "MAT3": 9, "MAT4": 16}
_SHORT, _USHORT, _UINT, _FLOAT, _UBYTE = 5122, 5123, 5125, 5126, 5121
_TRIANGLES = 4
MAX_TEXTURES = 32   # samplers[MAX_TEXTURES], surface.comp:46-52; slot 0 is "none"


# ---- glm's dmat4 algebra, in glm's operation order (matrices are numpy [row, column], float64) ---------------------------

def _mul(a, b):
    """glm operator*(dmat4, dmat4): column c of the result is ((A0 b0c + A1 b1c) + A2 b2c) + A3 b3c."""
    r = np.empty((4, 4), np.float64)
    for c in range(4):
        r[:, c] = ((a[:, 0] * b[0, c] + a[:, 1] * b[1, c]) + a[:, 2] * b[2, c]) + a[:, 3] * b[3, c]
    return r


def _translate(v):
    """glm::translate(dvec3) = translate(dmat4(1), v): Result[3] = m[0] v0 + m[1] v1 + m[2] v2 + m[3]."""
    m = np.eye(4, dtype=np.float64)
    r = m.copy()
    r[:, 3] = ((m[:, 0] * v[0] + m[:, 1] * v[1]) + m[:, 2] * v[2]) + m[:, 3]
    return r


def _scale(v):
    """glm::scale(dvec3) = scale(dmat4(1), v): Result[i] = m[i] v[i]."""
    m = np.eye(4, dtype=np.float64)
    r = m.copy()
    for i in range(3):
        r[:, i] = m[:, i] * v[i]
    return r


def _mat4_cast(q):
    """glm::mat4_cast(dquat), the quaternion given as glTF and glm's memory have it: (x, y, z, w)."""
    x, y, z, w = (np.float64(t) for t in q)
    qxx, qyy, qzz = x * x, y * y, z * z
    qxz, qxy, qyz = x * z, x * y, y * z
    qwx, qwy, qwz = w * x, w * y, w * z
    one, two = np.float64(1.0), np.float64(2.0)
    r = np.eye(4, dtype=np.float64)
    r[0, 0] = one - two * (qyy + qzz); r[1, 0] = two * (qxy + qwz); r[2, 0] = two * (qxz - qwy)
    r[0, 1] = two * (qxy - qwz); r[1, 1] = one - two * (qxx + qzz); r[2, 1] = two * (qyz + qwx)
    r[0, 2] = two * (qxz + qwy); r[1, 2] = two * (qyz - qwx); r[2, 2] = one - two * (qxx + qyy)
    return r


def root_transform(mscale=1.0):
    """Viewer.cpp:240-241: dmat4 matrix(1.0); matrix *= scale(dvec3(mscale))."""
    return _mul(np.eye(4, dtype=np.float64), _scale([np.float64(mscale)] * 3))


def node_transform(parent, node, as_viewer=True):
    """Viewer.cpp:246-253 for one node (a dict of the glTF file). tinygltf reads T / R / S only where there is no matrix."""
    eye = np.eye(4, dtype=np.float64)
    mat = node.get("matrix")
    has_matrix = isinstance(mat, list) and len(mat) >= 16
    t = None if has_matrix else node.get("translation")
    s = None if has_matrix else node.get("scale")
    q = None if has_matrix else node.get("rotation")
    local = eye.copy()
    local = _mul(local, np.asarray(mat[:16], np.float64).reshape(4, 4).T if has_matrix else eye)   # glTF matrices are column-major
    tm = _translate(np.asarray(t[:3], np.float64)) if isinstance(t, list) and len(t) >= 3 else eye
    sm = _scale(np.asarray(s[:3], np.float64)) if isinstance(s, list) and len(s) >= 3 else eye
    rm = _mat4_cast(q[:4]) if isinstance(q, list) and len(q) >= 4 else eye
    for m in ((tm, sm, rm) if as_viewer else (tm, rm, sm)):
        local = _mul(local, m)
    return _mul(np.asarray(parent, np.float64), local)


def mesh_transform(transform):
    """setTransform(mat4(transform)) (VertexInstance.inl:54-58): (t, inverse(t)) as float32, row-major -- the `transform` and
    `transform_inv` of a mesh description."""
    t = np.asarray(transform, np.float64).astype(_F32)
    return np.ascontiguousarray(t.reshape(16)), np.ascontiguousarray(scenes.inverse4(t).reshape(16))


# ---- the file --------------------------------------------------------------------------------------------------------------

def _need(obj, key, what, path):
    if key not in obj:
        raise ValueError("%s: %s has no `%s`" % (path, what, key))
    return obj[key]


def _index(v, n, what, path):
    if not isinstance(v, (int, float)) or int(v) != v or not 0 <= int(v) < n:
        raise ValueError("%s: %s index %r out of range (0..%d)" % (path, what, v, n - 1))
    return int(v)


def _read_buffer(b, i, base, path):
    uri = b.get("uri")
    if uri is None:
        raise ValueError("%s: buffer %d has no uri (binary .glb containers are not what the viewer's LoadASCIIFromFile reads)" % (path, i))
    if uri.startswith("data:"):
        head, _, payload = uri.partition(",")
        if not head.endswith(";base64"):
            raise ValueError("%s: buffer %d: only base64 data URIs" % (path, i))
        raw = base64.b64decode(payload)
    else:
        full = os.path.join(base, uri)
        if not os.path.exists(full):
            raise ValueError("%s: buffer %d: %s not found" % (path, i, uri))
        raw = open(full, "rb").read()
    n = int(b.get("byteLength", len(raw)))
    if n > len(raw):
        raise ValueError("%s: buffer %d: byteLength %d but %d bytes of data" % (path, i, n, len(raw)))
    data = np.zeros((n + 3) // 4 * 4, np.uint8)   # accessors address it in 4-byte units
    data[:n] = np.frombuffer(raw, np.uint8, n)
    return data


def _texture_index(param):
    """getTextureIndex (Application.hpp:80-82): the `index` member of a texture-info object, -1 without one."""
    if isinstance(param, dict) and isinstance(param.get("index"), (int, float)):
        return int(param["index"])
    return -1


def _numbers(v, n):
    """tinygltf's Parameter::number_array: an array of numbers, or the one number (ParseParameterProperty)."""
    if isinstance(v, (int, float)) and not isinstance(v, bool):
        v = [v]
    if isinstance(v, list) and len(v) >= n and all(isinstance(t, (int, float)) and not isinstance(t, bool) for t in v):
        return [float(t) for t in v]
    return None


def _read_gltf(path, mscale, as_viewer):
    """(read_gltf below) A .gltf file -> a scene dict:
      instances   mesh descriptions in loading order, one per (node, primitive): the arguments of TriangleHierarchy.loadMesh
      materials   VirtualMaterial dicts (scenes.materials_array), texture parts as TextureSet slots
      textures    {slot: uint8 [h, w, 4] image, bottom row first}
      triangle_count, buffers, views, templates (the per-mesh primitive templates, Viewer.cpp:151-232), ignored (what the file
      holds that the path does not use, statement -> count; logged)
    Malformed input raises ValueError naming the file and the element."""
    base = os.path.dirname(os.path.abspath(path))
    try:
        g = json.loads(open(path, "rb").read().decode("utf-8"))
    except (UnicodeDecodeError, json.JSONDecodeError) as e:
        raise ValueError("%s: not a glTF JSON document (%s) -- the viewer reads .gltf text (LoadASCIIFromFile, Viewer.cpp:69)" % (path, e)) from e
    if not isinstance(g, dict):
        raise ValueError("%s: not a glTF JSON document" % path)
    ignored = {}

    def note(what, n=1):
        ignored[what] = ignored.get(what, 0) + n

    buffers = [_read_buffer(b, i, base, path) for i, b in enumerate(g.get("buffers", []))]
    # one float view and one word view per buffer, shared by every primitive that reads it (TriangleHierarchy.loadMeshes uploads
    # a pool once per distinct memory: the reference's primitives share a GL buffer the same way, Viewer.cpp:133-139)
    fviews = [b.view(_F32) for b in buffers]
    wviews = [b.view(np.uint32) for b in buffers]
    gviews = g.get("bufferViews", [])
    gacc = g.get("accessors", [])
    for i, bv in enumerate(gviews):
        _index(_need(bv, "buffer", "bufferView %d" % i, path), len(buffers), "bufferView %d: buffer" % i, path)
        st = int(bv.get("byteStride", 0))
        if st > 252 or st % 4:
            raise ValueError("%s: bufferView %d: byteStride %d (a multiple of 4 up to 252)" % (path, i, st))
    # BufferViewSet, Viewer.cpp:141-148
    views = [(int(bv.get("byteOffset", 0)) // 4, int(bv.get("byteStride", 0)) // 4) for bv in gviews]

    # ---- textures and materials, Viewer.cpp:71-137
    images = g.get("images", [])
    textures, slot_of, rt_textures = {}, {}, []
    for i, tx in enumerate(g.get("textures", [])):
        src = _index(_need(tx, "source", "texture %d" % i, path), len(images), "texture %d: source" % i, path)
        uri = images[src].get("uri")
        slot = 0
        if uri is None or uri.startswith("data:"):
            note("image without a file uri (the viewer loads textures by file name)")
        else:
            full = os.path.join(base, uri)
            if full in slot_of:
                slot = slot_of[full]                       # TextureSet::texnames
            elif not (os.path.exists(full) or os.path.exists(full + ".npy")):
                note("image file not found: %s" % uri)     # FreeImage_GetFileType fails: slot 0
            else:
                if len(textures) + 1 >= MAX_TEXTURES:
                    raise ValueError("%s: more than %d images (samplers[MAX_TEXTURES], surface.comp:46-52)" % (path, MAX_TEXTURES - 1))
                slot = slot_of[full] = len(textures) + 1
                textures[slot] = np.ascontiguousarray(scenes.load_image_rgba8(full)[::-1])
        rt_textures.append(slot)
    if g.get("samplers"):
        note("samplers (every texture is GL_LINEAR, GL_REPEAT: TextureSet.inl:113-118)", len(g["samplers"]))

    def part(param, what):
        ti = _texture_index(param)
        if ti < 0:
            return 0
        return rt_textures[_index(ti, len(rt_textures), what, path)]

    materials = []
    for i, m in enumerate(g.get("materials", [])):
        pbr = m.get("pbrMetallicRoughness", {}) if isinstance(m.get("pbrMetallicRoughness", {}), dict) else {}
        bc = _numbers(pbr.get("baseColorFactor"), 3)
        diffuse = tuple(float(_F32(t)) for t in bc[:3]) + (1.0,) if bc else (1.0, 1.0, 1.0, 1.0)
        spec = [1.0, 1.0, 1.0, 1.0]
        mf, rf = _numbers(pbr.get("metallicFactor"), 1), _numbers(pbr.get("roughnessFactor"), 1)
        if mf:
            spec[2] = float(_F32(mf[0]))
        if rf:
            spec[1] = float(_F32(rf[0]))
        ef = _numbers(m.get("emissiveFactor"), 3)
        emissive = tuple(float(_F32(t)) for t in ef[:3]) + (1.0,) if ef else (0.0, 0.0, 0.0, 0.0)
        materials.append({"diffuse": diffuse, "specular": tuple(spec), "emissive": emissive,
                          "diffusePart": part(pbr.get("baseColorTexture"), "material %d: baseColorTexture" % i),
                          "specularPart": part(pbr.get("metallicRoughnessTexture"), "material %d: metallicRoughnessTexture" % i),
                          "emissivePart": part(m.get("emissiveTexture"), "material %d: emissiveTexture" % i),
                          "bumpPart": part(m.get("normalTexture"), "material %d: normalTexture" % i)})
        for k in m:
            if k not in ("pbrMetallicRoughness", "emissiveFactor", "emissiveTexture", "normalTexture", "name", "extras"):
                note("material." + k)

    # ---- primitive templates, Viewer.cpp:151-232
    def accessor(ai, what):
        a = gacc[_index(ai, len(gacc), what, path)]
        vi = _index(_need(a, "bufferView", what, path), len(gviews), what + ": bufferView", path)
        return a, gviews[vi], vi

    templates = []
    for mi, mesh in enumerate(g.get("meshes", [])):
        prims = []
        for pi, prim in enumerate(_need(mesh, "primitives", "mesh %d" % mi, path)):
            what = "mesh %d primitive %d" % (mi, pi)
            attrs = _need(prim, "attributes", what, path)
            geom = {"accessors": [], "views": views, "vertex_accessor": -1, "normal_accessor": -1, "texcoord_accessor": -1,
                    "vertices": None, "indices": None, "index16": 0, "node_count": 1, "primitive_type": 0, "loading_offset": 0,
                    "material_id": int(prim.get("material", -1))}
            for name in sorted(attrs):           # std::map<std::string, int>: in key order
                if name not in ("POSITION", "NORMAL", "TEXCOORD_0"):
                    note("attribute " + name)
                    continue
                a, bv, vi = accessor(attrs[name], "%s %s" % (what, name))
                if a.get("componentType") != _FLOAT:
                    raise ValueError("%s: %s %s: component type %r (the loader reads 32-bit floats, loader.comp:32-50)"
                                     % (path, what, name, a.get("componentType")))
                ncomp = _COMPONENTS.get(a.get("type"), 0)
                if ncomp != (2 if name == "TEXCOORD_0" else 3):
                    raise ValueError("%s: %s %s: type %r" % (path, what, name, a.get("type")))
                geom["accessors"].append((int(a.get("byteOffset", 0)) // 4, ncomp - 1, vi))
                key = {"POSITION": "vertex_accessor", "NORMAL": "normal_accessor", "TEXCOORD_0": "texcoord_accessor"}[name]
                geom[key] = len(geom["accessors"]) - 1
                if name == "POSITION":
                    geom["vertices"] = fviews[bv["buffer"]]
                    geom["_vertex_count"] = int(a.get("count", 0))
            if geom["vertices"] is None:
                raise ValueError("%s: %s has no POSITION" % (path, what))
            if prim.get("indices") is not None and int(prim["indices"]) >= 0:
                a, bv, _ = accessor(prim["indices"], what + " indices")
                ct = a.get("componentType")
                is16 = ct in (_SHORT, _USHORT)
                if ct == _UBYTE:
                    if as_viewer:
                        note("8-bit indices (read as 32-bit words, as the viewer does)")
                    else:
                        raise ValueError("%s: %s: 8-bit indices" % (path, what))
                elif ct not in (_SHORT, _USHORT, _UINT):
                    raise ValueError("%s: %s: index component type %r" % (path, what, ct))
                geom["node_count"] = int(a.get("count", 0)) // 3
                geom["indices"] = wviews[bv["buffer"]]
                geom["index16"] = int(is16)
                geom["loading_offset"] = (int(bv.get("byteOffset", 0)) + int(a.get("byteOffset", 0))) // (2 if is16 else 4)
            elif as_viewer:
                note("primitive without indices (the viewer leaves nodeCount = 1: one triangle, Structs.hpp:218)")
            else:
                geom["node_count"] = geom["_vertex_count"] // 3
            if geom["material_id"] >= len(materials):
                raise ValueError("%s: %s: material %d of %d" % (path, what, geom["material_id"], len(materials)))
            if int(prim.get("mode", _TRIANGLES)) == _TRIANGLES:      # Viewer.cpp:225
                prims.append(geom)
            else:
                note("primitive mode %d (only TRIANGLES are loaded)" % int(prim["mode"]))
        templates.append(prims)

    # ---- the node walk, Viewer.cpp:244-277
    nodes = g.get("nodes", [])
    instances = []

    def traverse(ni, parent, recursive, seen):
        ni = _index(ni, len(nodes), "node", path)
        node = nodes[ni]
        if ni in seen:
            raise ValueError("%s: node %d is its own ancestor" % (path, ni))
        transform = node_transform(parent, node, as_viewer)
        mesh = node.get("mesh", -1)
        kids = node.get("children", []) or []
        if mesh is not None and int(mesh) >= 0:
            t, ti = mesh_transform(transform)
            for geom in templates[_index(mesh, len(templates), "node %d: mesh" % ni, path)]:
                if geom["node_count"] > 0:                                   # TriangleHierarchy.inl:174
                    inst = {k: v for k, v in geom.items() if not k.startswith("_")}
                    inst["transform"], inst["transform_inv"] = t, ti
                    instances.append(inst)
            if as_viewer:
                if kids:
                    note("children of a node with a mesh (not visited, Viewer.cpp:254-268)", len(kids))
                return
        for c in kids:
            if recursive >= 0 or not as_viewer:
                traverse(c, transform, recursive - 1, seen | {ni})
            else:
                note("nodes deeper than four levels (not visited, Viewer.cpp:266)")
        if node.get("camera") is not None:
            note("node.camera")
        if node.get("skin") is not None:
            note("node.skin")

    gscenes = g.get("scenes", [])
    if gscenes:
        root = root_transform(mscale)
        for ni in gscenes[0].get("nodes", []):
            traverse(ni, root, 2, frozenset())
        if len(gscenes) > 1:
            note("scenes beyond the first", len(gscenes) - 1)
    else:
        note("no scenes: nothing is loaded (Viewer.cpp:272)")
    for k in ("animations", "skins", "cameras", "extensionsUsed"):
        if g.get(k):
            note(k, len(g[k]))
    if ignored:
        log.info("%s: not used by the path: %s", path, ", ".join("%s x%d" % kv for kv in sorted(ignored.items())))
    return {"name": os.path.basename(path), "instances": instances, "materials": materials, "textures": textures,
            "triangle_count": sum(m["node_count"] for m in instances), "buffers": buffers, "views": views,
            "templates": templates, "ignored": ignored}


def read_gltf(path, mscale=1.0, as_viewer=True):
    """A .gltf file -> a scene dict (instances = the arguments of TriangleHierarchy.loadMesh in loading order, materials, textures by
    slot, triangle_count, buffers, views, templates, ignored): see the module's text. Malformed input of any kind -- a missing or
    mistyped member as much as an index out of range -- raises ValueError naming the file."""
    try:
        return _read_gltf(path, mscale, as_viewer)
    except (KeyError, TypeError, IndexError, AttributeError, OverflowError) as e:
        raise ValueError("%s: malformed glTF (%s: %s)" % (path, type(e).__name__, e)) from e


def load_into(scene, intersector, material_manager=None, texture_set=None):
    """What GltfViewer::init does with the parsed file (Viewer.cpp:71-137, 239-277) on the package's mirror of the reference's
    classes: the images into a TextureSet in slot order, the materials into the MaterialSet, clearTribuffer and one loadMesh per
    instance. The intersector must have been allocate()d for scene["triangle_count"] triangles. Returns the TextureSet."""
    if material_manager is not None:
        from . import TextureSet
        txs = texture_set if texture_set is not None else TextureSet()
        material_manager.setTextureSet(txs)
        remap = {0: 0}
        for slot in sorted(scene["textures"]):
            remap[slot] = txs.loadTexture(scene["textures"][slot])
        material_manager.clearSubmats()
        for m in scene["materials"]:
            m = dict(m)
            for p in ("diffusePart", "specularPart", "emissivePart", "bumpPart"):
                m[p] = remap[m.get(p, 0)]
            material_manager.addSubmat(m)
    else:
        txs = texture_set
    intersector.clearTribuffer()
    intersector.loadMeshes(scene["instances"])
    return txs
