"""prismarine-core_amd -- MI355X-native path-tracing core (hot path of EngineWorld/prismarine-core).

Python host-side mirror of the reference's header-only API (Include/Prismarine/*.hpp) over the
C ABI of include/psm_hip.h.  The compute lives in libpsm_hip.so (hand-written gfx950 HIP kernels,
prismarine-core_amd/csrc); this package is plumbing: ctypes signatures and the call order of
psm::RadixSort / psm::TriangleHierarchy / psm::Pipeline.

There is no CPU fallback: importing works anywhere (so the library's exports can be checked), but
every compute entry point needs a gfx950 device and raises PsmError otherwise.

The directory name contains a hyphen; import it with
    importlib.import_module("prismarine-core_amd")
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PSM_HIP_LIB") or os.path.join(_HERE, "libpsm_hip.so")  # override: A/B builds of the same ABI

EXPORTS = [
    "psm_device_count", "psm_ctx_create", "psm_ctx_create_on_stream", "psm_ctx_destroy", "psm_ctx_sync", "psm_ctx_copy_bandwidth", "psm_ctx_stream", "psm_last_error",
    "psm_buf_alloc", "psm_buf_free", "psm_buf_upload", "psm_buf_download", "psm_buf_ptr",
    "psm_sort_u64_u32", "psm_sort_u64_u32_dev", "psm_sort_set_algorithm", "psm_sort_get_algorithm",
    "psm_bvh_create", "psm_bvh_destroy", "psm_bvh_clear", "psm_bvh_load_triangles", "psm_bvh_set_texcoords", "psm_bvh_load_mesh", "psm_bvh_build", "psm_bvh_set_build_graph", "psm_bvh_refit",
    "psm_bvh_get_info", "psm_bvh_stage_bounds", "psm_bvh_stage_morton", "psm_bvh_stage_sort",
    "psm_bvh_stage_emit", "psm_bvh_download",
    "psm_rt_create", "psm_rt_destroy", "psm_rt_resize_buffers", "psm_rt_resize", "psm_rt_set_tile", "psm_rt_set_tile_interleaved", "psm_rt_set_tile_weighted",
    "psm_rt_set_lights", "psm_rt_set_sky", "psm_rt_set_skybox", "psm_rt_set_texture", "psm_rt_set_materials", "psm_rt_camera", "psm_rt_set_camera_mode", "psm_rt_ray_count",
    "psm_rt_traverse", "psm_rt_set_traverse_mode", "psm_rt_set_traverse_phases", "psm_rt_set_traverse_adaptive", "psm_rt_set_traverse_solo", "psm_rt_reset_hits", "psm_rt_shade", "psm_rt_sample", "psm_rt_sample_from", "psm_lanes_render", "psm_lanes_run_sharded", "psm_rt_clear_sampler", "psm_rt_snap",
    "psm_rt_get_texels_dev", "psm_rt_set_texels_dev", "psm_rt_tile_texels", "psm_rt_pack_texels_dev",
    "psm_rt_unpack_texels_dev", "psm_rt_unpack_tiles_dev", "psm_rt_ray_count_dev", "psm_rt_set_ray_count", "psm_rt_download_rays", "psm_rt_download_hits",
    "psm_rt_upload_rays", "psm_rt_download_texels",
    "psm_stats_enable", "psm_stats_reset", "psm_stats_get", "psm_stats_reference", "psm_stats_traverse_intervals",
    "psm_dist_unique_id", "psm_dist_init", "psm_dist_prepare", "psm_dist_connect", "psm_dist_connect_transport", "psm_dist_connect_hoststaged",
    "psm_dist_transport_name", "psm_dist_agree", "psm_dist_destroy", "psm_dist_rank", "psm_dist_world", "psm_dist_comm_ranks", "psm_dist_gather_tiles",
    "psm_dist_allgather_i32", "psm_dist_barrier", "psm_dist_decide", "psm_dist_render_batch", "psm_dist_render_frames", "psm_dist_emulate_tile", "psm_dist_set_band_weights",
]

TRAVERSE_AUTO, TRAVERSE_WHOLE, TRAVERSE_PHASED, TRAVERSE_ADAPTIVE = range(4)
TRAVERSE_MODES = {"auto": TRAVERSE_AUTO, "whole": TRAVERSE_WHOLE, "phased": TRAVERSE_PHASED, "adaptive": TRAVERSE_ADAPTIVE}

(BVH_KEYS, BVH_INDICES, BVH_LEAF_BOX, BVH_LEAF_TRI, BVH_PAIR_BOX, BVH_LINK, BVH_RANGE,
 BVH_SORTED_TRI, BVH_POSITIONS, BVH_NORMALS, BVH_MATERIALS, BVH_TEXCOORDS, BVH_NODE32) = range(13)

RAY_DT = np.dtype([("origin", "<f4", 3), ("direct", "<f4", 3), ("color", "<f4", 3),
                   ("bitfield", "<i4"), ("texel", "<i4"), ("pkey", "<u4")])
HIT_DT = np.dtype([("u", "<f4"), ("v", "<f4"), ("t", "<f4"), ("tri", "<i4")])
LIGHT_DT = np.dtype([("lightVector", "<f4", 4), ("lightColor", "<f4", 4),
                     ("lightOffset", "<f4", 4), ("lightAmbient", "<f4", 4)])


class PsmError(RuntimeError):
    pass


class BvhInfo(C.Structure):
    _fields_ = [("triangle_count", C.c_uint32), ("leaf_count", C.c_uint32), ("root", C.c_int32),
                ("transform", C.c_float * 16), ("bounds_min", C.c_float * 4), ("bounds_max", C.c_float * 4)]


class Accessor(C.Structure):
    _fields_ = [("offset4", C.c_int32), ("components", C.c_int32), ("buffer_view", C.c_int32)]


class BufferView(C.Structure):
    _fields_ = [("offset4", C.c_int32), ("stride4", C.c_int32)]


class MeshDesc(C.Structure):
    _fields_ = [("d_vertices", C.c_void_p), ("vertex_floats", C.c_size_t), ("d_indices", C.c_void_p),
                ("index_words", C.c_size_t), ("accessors", C.c_void_p), ("accessor_count", C.c_uint32),
                ("views", C.c_void_p), ("view_count", C.c_uint32), ("vertex_accessor", C.c_int32),
                ("normal_accessor", C.c_int32), ("texcoord_accessor", C.c_int32), ("modifier_accessor", C.c_int32),
                ("transform", C.c_float * 16), ("transform_inv", C.c_float * 16), ("material_id", C.c_int32),
                ("is_indexed", C.c_int32), ("index16", C.c_int32), ("node_count", C.c_int32),
                ("primitive_type", C.c_int32), ("loading_offset", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("rays_traced", C.c_uint64), ("node_visits", C.c_uint64), ("tri_tests", C.c_uint64),
                ("stack_drops", C.c_uint64), ("iter_caps", C.c_uint64), ("baked_drops", C.c_uint64),
                ("chain_pool_drops", C.c_uint64), ("ray_limit_drops", C.c_uint64),
                ("traverse_launches", C.c_uint32), ("traverse_ms", C.c_float), ("build_ms", C.c_float),
                ("sort_ms", C.c_float), ("shade_ms", C.c_float), ("camera_ms", C.c_float),
                ("sample_ms", C.c_float), ("rounds", C.c_uint32), ("bounds_ms", C.c_float), ("morton_ms", C.c_float),
                ("emit_ms", C.c_float), ("wave_clock_ticks", C.c_uint64), ("wave_real_ticks", C.c_uint64),
                ("wave_steps", C.c_uint64), ("waves", C.c_uint64), ("handover_launches", C.c_uint32), ("handover_ms", C.c_float)]


_lib = None


def lib():
    """Load libpsm_hip.so (built by __graft_entry__.build()). Fails loudly when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PsmError("libpsm_hip.so is missing at %s -- run `make -C prismarine-core_amd/csrc` "
                           "(or __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        _lib.psm_last_error.restype = C.c_char_p
        _lib.psm_ctx_stream.restype = C.c_void_p
        for name in EXPORTS:
            getattr(_lib, name)  # AttributeError if an export is missing
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Context:
    """One GPU, one in-order stream (the reference's single GL context, Viewer.cpp:371)."""

    def __init__(self, device=0, stream=None):
        """stream: an existing hipStream_t (int), e.g. torch.cuda.current_stream().cuda_stream, so kernels
        and RCCL collectives are ordered on one stream without host synchronisation."""
        self._h = C.c_void_p()
        if stream is None:
            rc = lib().psm_ctx_create(C.c_int(device), C.byref(self._h))
        else:
            rc = lib().psm_ctx_create_on_stream(C.c_int(device), C.c_void_p(stream), C.byref(self._h))
        if rc != 0:
            raise PsmError("psm_ctx_create(device=%d) failed with %d: a gfx950 (MI355X) device is required; "
                           "there is no CPU fallback" % (device, rc))
        self.device = device

    def check(self, rc, what=""):
        if rc != 0:
            msg = lib().psm_last_error(self._h)
            raise PsmError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else ""))

    def sync(self):
        self.check(lib().psm_ctx_sync(self._h), "psm_ctx_sync")

    @property
    def stream(self):
        return lib().psm_ctx_stream(self._h)

    def copy_bandwidth(self, nbytes=1 << 30, reps=5):
        """Measured device-to-device copy rate in GB/s of traffic (read + write): the box's achievable HBM ceiling."""
        v = C.c_double()
        self.check(lib().psm_ctx_copy_bandwidth(self._h, C.c_size_t(nbytes), C.c_int(reps), C.byref(v)), "psm_ctx_copy_bandwidth")
        return v.value

    def stats_enable(self, timing=True, counting=False):
        """timing: False / True (HIP events around every launch) / 2 (traversal launches only: light, for frames in flight)"""
        self.check(lib().psm_stats_enable(self._h, C.c_int(int(timing)), C.c_int(int(counting))), "psm_stats_enable")

    def stats_reference(self, origin=None):
        """time origin of traverse_intervals(): recorded now on this context's stream, or shared with `origin`'s"""
        self.check(lib().psm_stats_reference(self._h, (origin or self)._h), "psm_stats_reference")

    def traverse_intervals(self):
        """(start, end) in ms after the reference of every traversal launch timed since the last reset"""
        n = C.c_uint32()
        self.check(lib().psm_stats_traverse_intervals(self._h, None, C.c_uint32(0), C.byref(n)), "psm_stats_traverse_intervals")
        buf = (C.c_float * (2 * max(n.value, 1)))()
        self.check(lib().psm_stats_traverse_intervals(self._h, buf, C.c_uint32(n.value), C.byref(n)), "psm_stats_traverse_intervals")
        return [(buf[2 * i], buf[2 * i + 1]) for i in range(n.value)]

    def stats_reset(self):
        self.check(lib().psm_stats_reset(self._h), "psm_stats_reset")

    def stats(self):
        s = Stats()
        self.check(lib().psm_stats_get(self._h, C.byref(s)), "psm_stats_get")
        return s

    def close(self):
        if self._h:
            lib().psm_ctx_destroy(self._h)
            self._h = C.c_void_p()

    # buffers (the GLuint names the header layer passes around, Utils.hpp:140-150)
    def buf_alloc(self, nbytes):
        h = C.c_uint32()
        self.check(lib().psm_buf_alloc(self._h, C.c_size_t(nbytes), C.byref(h)), "psm_buf_alloc")
        return h.value

    def buf_free(self, h):
        self.check(lib().psm_buf_free(self._h, C.c_uint32(h)), "psm_buf_free")

    def buf_ptr(self, h):
        p, n = C.c_void_p(), C.c_size_t()
        self.check(lib().psm_buf_ptr(self._h, C.c_uint32(h), C.byref(p), C.byref(n)), "psm_buf_ptr")
        return p.value, n.value

    def buf_upload(self, h, arr, offset=0):
        arr = np.ascontiguousarray(arr)
        self.check(lib().psm_buf_upload(self._h, C.c_uint32(h), C.c_size_t(offset), _p(arr), C.c_size_t(arr.nbytes)),
                   "psm_buf_upload")

    def buf_download(self, h, dtype, count, offset=0):
        out = np.zeros(count, dtype)
        self.check(lib().psm_buf_download(self._h, C.c_uint32(h), C.c_size_t(offset), _p(out), C.c_size_t(out.nbytes)),
                   "psm_buf_download")
        return out


class RadixSort:
    """psm::RadixSort (Include/Prismarine/Radix.hpp:27-74)."""

    def __init__(self, ctx):
        self.ctx = ctx

    def setAlgorithm(self, algorithm):
        """2 (default): hybrid -- two global passes over the top sixteen key bits, the rest in LDS; 0: histogram / scan /
        scatter kernels for all eight passes; 1: one-sweep histograms + look-back scatter."""
        self.ctx.check(lib().psm_sort_set_algorithm(self.ctx._h, C.c_int(algorithm)), "psm_sort_set_algorithm")

    def getAlgorithm(self):
        """(asked, effective): they differ once a hybrid sort has overflowed a chunk and the context fell back to 0."""
        a, e = C.c_int(0), C.c_int(0)
        self.ctx.check(lib().psm_sort_get_algorithm(self.ctx._h, C.byref(a), C.byref(e)), "psm_sort_get_algorithm")
        return a.value, e.value

    def sort(self, keys_handle, vals_handle, size=1, descending=0):
        # `descending` is accepted and ignored like the reference's shaders do (radix/includes.glsl:50-55)
        self.ctx.check(lib().psm_sort_u64_u32(self.ctx._h, C.c_uint32(keys_handle), C.c_uint32(vals_handle),
                                              C.c_uint32(size)), "psm_sort_u64_u32")

    def sort_arrays(self, keys, vals):
        """Convenience: host arrays in, sorted host arrays out."""
        keys = np.ascontiguousarray(keys, np.uint64)
        vals = np.ascontiguousarray(vals, np.uint32)
        n = keys.shape[0]
        hk = self.ctx.buf_alloc(max(n, 1) * 8)
        hv = self.ctx.buf_alloc(max(n, 1) * 4)
        try:
            if n:
                self.ctx.buf_upload(hk, keys)
                self.ctx.buf_upload(hv, vals)
            self.sort(hk, hv, n)
            return self.ctx.buf_download(hk, np.uint64, n), self.ctx.buf_download(hv, np.uint32, n)
        finally:
            self.ctx.buf_free(hk)
            self.ctx.buf_free(hv)


class TriangleHierarchy:
    """psm::TriangleHierarchy (Include/Prismarine/TriangleHierarchy.hpp:75-94)."""

    def __init__(self, ctx):
        self.ctx = ctx
        self._h = C.c_void_p()
        self.triangleCount = 0
        self.materialID = 0
        self._dirty = False
        self.maxt = 0

    def allocate(self, count):
        if self._h:
            lib().psm_bvh_destroy(self._h)
            self._h = C.c_void_p()
        self.ctx.check(lib().psm_bvh_create(self.ctx._h, C.c_size_t(count), C.byref(self._h)), "psm_bvh_create")
        self.maxt = count
        self.clearTribuffer()

    def clearTribuffer(self):
        self.markDirty()
        self.ctx.check(lib().psm_bvh_clear(self._h), "psm_bvh_clear")
        self.triangleCount = 0

    def setMaterialID(self, mid):
        self.materialID = mid

    def loadTriangles(self, tris, normals=None, mats=None, texcoords=None):
        """loadMesh() reduced to its result: append world-space triangles (loader.comp:115-135);
        texcoords: float32 [n,3,2] (u,v per vertex) or None."""
        tris = np.ascontiguousarray(tris, np.float32).reshape(-1, 9)
        n = tris.shape[0]
        nrm = None if normals is None else np.ascontiguousarray(normals, np.float32).reshape(-1, 9)
        mm = None if mats is None else np.ascontiguousarray(mats, np.int32)
        self.ctx.check(lib().psm_bvh_load_triangles(self._h, _p(tris), _p(nrm) if nrm is not None else None,
                                                    _p(mm) if mm is not None else None, C.c_size_t(n),
                                                    C.c_int32(self.materialID)), "psm_bvh_load_triangles")
        if texcoords is not None:
            tc = np.ascontiguousarray(texcoords, np.float32).reshape(-1, 6)
            if tc.shape[0] != n:
                raise ValueError("texcoords: expected %d triangles, got %d" % (n, tc.shape[0]))
            self.ctx.check(lib().psm_bvh_set_texcoords(self._h, C.c_size_t(self.triangleCount), _p(tc), C.c_size_t(n)),
                           "psm_bvh_set_texcoords")
        self.triangleCount += n
        self.markDirty()

    def loadMesh(self, mesh):
        """loadMesh(TriangleArrayInstance*) (TriangleHierarchy.inl:173-192): `mesh` is a dict with the
        fields of VertexInstance.hpp -- vertices (float pool), indices (uint32 words or None), accessors
        [(offset4, components, bufferView)], views [(offset4, stride4)], vertex_accessor, normal_accessor,
        texcoord_accessor (v is stored as 1 - v, loader.comp:97-99),
        transform / transform_inv (row-major 4x4), material_id, index16, node_count, primitive_type,
        loading_offset. The pools are uploaded to device buffers and resolved by the HIP gather kernel."""
        self.loadMeshes([mesh])

    def loadMeshes(self, meshes):
        """loadMesh for every description of `meshes`, in order. Descriptions that share a pool (the same numpy array: the
        primitives of one glTF buffer, gltf.read_gltf) share its upload, as the reference's primitives share a GL buffer
        (Viewer.cpp:133-139)."""
        pools = {}   # (address, bytes) -> (handle, contiguous array kept alive): float and word views of one buffer share an upload
        self.pool_uploads = 0   # device pools the last loadMeshes made: one per distinct stretch of memory, however many views name it

        def dev(a, dtype):
            c = np.ascontiguousarray(a, dtype)   # (a view of the right dtype comes back as itself: same memory, same key)
            key = (c.__array_interface__["data"][0], c.nbytes)   # (both pool types are 4-byte words: the bytes are the same)
            if key not in pools:
                h = self.ctx.buf_alloc(max(c.nbytes, 4))
                pools[key] = (h, c, a)
                self.ctx.buf_upload(h, c)
                self.pool_uploads += 1
            h, c, _ = pools[key]
            return self.ctx.buf_ptr(h)[0], c.size

        try:
            for mesh in meshes:
                idx = mesh.get("indices")
                acc = (Accessor * len(mesh["accessors"]))(*[Accessor(*a) for a in mesh["accessors"]])
                views = (BufferView * len(mesh["views"]))(*[BufferView(*v) for v in mesh["views"]])
                d = MeshDesc()
                d.d_vertices, d.vertex_floats = dev(mesh["vertices"], np.float32)
                d.d_indices, d.index_words = dev(idx, np.uint32) if idx is not None else (None, 0)
                d.accessors, d.accessor_count = C.cast(acc, C.c_void_p), len(mesh["accessors"])
                d.views, d.view_count = C.cast(views, C.c_void_p), len(mesh["views"])
                d.vertex_accessor, d.normal_accessor = mesh["vertex_accessor"], mesh.get("normal_accessor", -1)
                d.texcoord_accessor, d.modifier_accessor = mesh.get("texcoord_accessor", -1), -1
                t = np.ascontiguousarray(mesh["transform"], np.float32).reshape(16)
                ti = np.ascontiguousarray(mesh["transform_inv"], np.float32).reshape(16)
                for k in range(16):
                    d.transform[k], d.transform_inv[k] = t[k], ti[k]
                d.material_id, d.is_indexed, d.index16 = mesh.get("material_id", self.materialID), int(idx is not None), int(mesh.get("index16", 0))
                d.node_count, d.primitive_type = mesh["node_count"], mesh.get("primitive_type", 0)
                d.loading_offset = mesh.get("loading_offset", 0)
                self.ctx.check(lib().psm_bvh_load_mesh(self._h, C.byref(d)), "psm_bvh_load_mesh")
                self.triangleCount += mesh["node_count"] * (2 if mesh.get("primitive_type", 0) == 1 else 1)
                self.markDirty()
        finally:
            for h, _, _ in pools.values():
                self.ctx.buf_free(h)

    def isDirty(self):
        return self._dirty

    def markDirty(self):
        self._dirty = True

    def resolve(self):
        self._dirty = False

    def build(self, optimization=None):
        if self.triangleCount <= 0 or not self._dirty:  # TriangleHierarchy.inl:214
            return
        opt = None if optimization is None else np.ascontiguousarray(optimization, np.float64).reshape(16)
        self.ctx.check(lib().psm_bvh_build(self._h, _p(opt) if opt is not None else None), "psm_bvh_build")
        self.resolve()

    def refit(self):
        """psm_bvh_refit (not in the reference): the triangles were reloaded -- same count, same order, moved -- and only
        the boxes are recomputed; the tree is the last build's."""
        self.ctx.check(lib().psm_bvh_refit(self._h), "psm_bvh_refit")
        self.resolve()

    def setBuildGraph(self, enable=True):
        """Replay rebuilds as one captured hipGraph from the second build of a triangle count on (default), or keep plain launches."""
        self.ctx.check(lib().psm_bvh_set_build_graph(self._h, C.c_int(1 if enable else 0)), "psm_bvh_set_build_graph")

    def stage(self, name, optimization=None):
        if name == "bounds":
            opt = None if optimization is None else np.ascontiguousarray(optimization, np.float64).reshape(16)
            rc = lib().psm_bvh_stage_bounds(self._h, _p(opt) if opt is not None else None)
        else:
            rc = getattr(lib(), "psm_bvh_stage_" + name)(self._h)
        self.ctx.check(rc, "psm_bvh_stage_" + name)

    def info(self):
        i = BvhInfo()
        self.ctx.check(lib().psm_bvh_get_info(self._h, C.byref(i)), "psm_bvh_get_info")
        return i

    def download(self, what, dtype, count):
        out = np.zeros(count, dtype)
        if count:
            self.ctx.check(lib().psm_bvh_download(self._h, C.c_int(what), _p(out), C.c_size_t(out.nbytes)),
                           "psm_bvh_download")
        return out

    def close(self):
        if self._h:
            lib().psm_bvh_destroy(self._h)
            self._h = C.c_void_p()


class TextureSet:
    """psm::TextureSet (Include/Prismarine/TextureSet.{hpp,inl}): slot table of material textures. A texture
    is an RGBA8 image, uint8 [h,w,4], row 0 = v 0 (GL order); slot 0 means "none" (TextureSet.inl:7-12)."""

    def __init__(self):
        self.textures = [None]
        self.freedomTextures = []
        self.revision = 1

    def loadTexture(self, image):
        """TextureSet.inl:73-86: reuse a freed slot, else append; returns the slot index materials refer to."""
        img = np.ascontiguousarray(image, np.uint8)
        if img.ndim != 3 or img.shape[2] != 4:
            raise ValueError("texture must be uint8 [h,w,4]")
        if self.freedomTextures:
            idx = self.freedomTextures.pop()
            self.textures[idx] = img
        else:
            idx = len(self.textures)
            self.textures.append(img)
        self.revision += 1
        return idx

    def freeTexture(self, idx):
        self.freedomTextures.append(idx)
        self.textures[idx] = None
        self.revision += 1

    def clearGlTextures(self):
        for i in range(1, len(self.textures)):
            self.freeTexture(i)

    def loadToVGA(self):
        pass  # uploaded by Pipeline.applyMaterials


class MaterialSet:
    """psm::MaterialSet (Include/Prismarine/MaterialSet.hpp:28-41): a host-side material array."""

    def __init__(self):
        self.submats = []
        self.loadOffset = 0
        self.texset = None

    def setTextureSet(self, txs):
        self.texset = txs

    def addSubmat(self, m):
        self.submats.append(m)
        return len(self.submats) - 1

    def setSumbat(self, i, m):
        while len(self.submats) <= i:
            self.submats.append(dict(self.submats[-1]) if self.submats else m)
        self.submats[i] = m

    def clearSubmats(self):
        self.submats = []

    def getMaterialCount(self):
        return len(self.submats)

    def setLoadingOffset(self, off):
        self.loadOffset = off

    def loadToVGA(self):
        pass  # uploaded by Pipeline.applyMaterials


class Pipeline:
    """psm::Pipeline (Include/Prismarine/Pipeline.hpp:84-137)."""

    def __init__(self, ctx, seed=1):
        from . import scenes as _sc
        self._sc = _sc
        self.ctx = ctx
        self._h = C.c_void_p()
        ctx.check(lib().psm_rt_create(ctx._h, C.byref(self._h)), "psm_rt_create")
        self.width = self.height = self.displayWidth = self.displayHeight = 256  # Pipeline.hpp:87-90
        self.raycountCache = 0
        self._rand_state = seed & 0xFFFFFFFF
        self._mat_sig = None
        self._tex_sig = None
        self.resizeBuffers(256, 256)
        self.resize(256, 256)

    # host rand() of Pipeline.inl:282,426 made explicit: the MSVC CRT LCG, seedable
    def setSeed(self, seed):
        self._rand_state = seed & 0xFFFFFFFF

    def _rand(self):
        self._rand_state = (self._rand_state * 214013 + 2531011) & 0xFFFFFFFF
        return (self._rand_state >> 16) & 0x7FFF

    def resizeBuffers(self, w, h):
        self.width, self.height = w, h
        self.ctx.check(lib().psm_rt_resize_buffers(self._h, C.c_uint32(w), C.c_uint32(h)), "psm_rt_resize_buffers")

    def resize(self, w, h):
        self.displayWidth, self.displayHeight = w, h
        self.ctx.check(lib().psm_rt_resize(self._h, C.c_uint32(w), C.c_uint32(h)), "psm_rt_resize")

    def setTile(self, y0, y1):
        self.ctx.check(lib().psm_rt_set_tile(self._h, C.c_uint32(y0), C.c_uint32(y1)), "psm_rt_set_tile")

    def setTileInterleaved(self, rank, world, weights=None):
        """8-row bands dealt to the ranks: round-robin, or weights[r] bands of every period of sum(weights) for rank r
        (psm_rt_set_tile_weighted; dist.band_pattern is the same dealing in Python)."""
        wv = None if weights is None else (C.c_uint32 * world)(*[int(v) for v in weights])
        self.ctx.check(lib().psm_rt_set_tile_weighted(self._h, C.c_uint32(rank), C.c_uint32(world), wv),
                       "psm_rt_set_tile_weighted")

    def tile_texels(self):
        n = C.c_uint32()
        self.ctx.check(lib().psm_rt_tile_texels(self._h, C.byref(n)), "psm_rt_tile_texels")
        return n.value

    def pack_texels_dev(self, dev_ptr):
        self.ctx.check(lib().psm_rt_pack_texels_dev(self._h, C.c_void_p(dev_ptr)), "psm_rt_pack_texels_dev")

    def unpack_tiles_dev(self, world, skip_rank, dev_ptr, stride_floats):
        """all the other ranks' gathered tiles (dense, stride_floats apart) into this image in one launch"""
        self.ctx.check(lib().psm_rt_unpack_tiles_dev(self._h, C.c_uint32(world), C.c_uint32(skip_rank), C.c_void_p(dev_ptr),
                                                     C.c_size_t(stride_floats)), "psm_rt_unpack_tiles_dev")

    def unpack_texels_dev(self, interleaved, a, b, dev_ptr):
        self.ctx.check(lib().psm_rt_unpack_texels_dev(self._h, C.c_int(int(interleaved)), C.c_uint32(a), C.c_uint32(b),
                                                      C.c_void_p(dev_ptr)), "psm_rt_unpack_texels_dev")

    def ray_count_dev(self, dev_ptr):
        self.ctx.check(lib().psm_rt_ray_count_dev(self._h, C.c_void_p(dev_ptr)), "psm_rt_ray_count_dev")

    def set_ray_count(self, n):
        self.ctx.check(lib().psm_rt_set_ray_count(self._h, C.c_int32(n)), "psm_rt_set_ray_count")
        self.raycountCache = n

    def setLights(self, lights):
        lights = np.ascontiguousarray(lights, LIGHT_DT)
        self.ctx.check(lib().psm_rt_set_lights(self._h, _p(lights), C.c_uint32(lights.shape[0])), "psm_rt_set_lights")

    def setSky(self, rgb):
        a = np.asarray(list(rgb)[:3] + [1.0], np.float32)
        self.ctx.check(lib().psm_rt_set_sky(self._h, _p(a)), "psm_rt_set_sky")

    def setSkybox(self, image):
        """setSkybox(GLuint) (Pipeline.hpp:93): here the equirect image itself, uint8 [h,w,4] RGBA (None =
        constant sky colour). Sampled as public/environment.glsl:23-26 does (GL_LINEAR, clamp to edge)."""
        if image is None:
            self.ctx.check(lib().psm_rt_set_skybox(self._h, None, C.c_uint32(0), C.c_uint32(0)), "psm_rt_set_skybox")
            return
        img = np.ascontiguousarray(image, np.uint8)
        self.ctx.check(lib().psm_rt_set_skybox(self._h, _p(img), C.c_uint32(img.shape[1]), C.c_uint32(img.shape[0])),
                       "psm_rt_set_skybox")

    def clearSampler(self):
        self.ctx.check(lib().psm_rt_clear_sampler(self._h), "psm_rt_clear_sampler")

    def switchMode(self):
        """Pipeline.inl:128-132: clear, then toggle the 360-degree camera (camera.comp:48-59)."""
        self.raycountCache = 0
        self.clearSampler()
        self.enable360 = 0 if getattr(self, "enable360", 0) else 1
        self.ctx.check(lib().psm_rt_set_camera_mode(self._h, C.c_int(self.enable360)), "psm_rt_set_camera_mode")

    def camera_matrices(self, cam_inv, proj_inv, time=None):
        t = self._rand() if time is None else time
        ci = np.ascontiguousarray(cam_inv, np.float32).reshape(16)
        pi = np.ascontiguousarray(proj_inv, np.float32).reshape(16)
        self.ctx.check(lib().psm_rt_camera(self._h, _p(ci), _p(pi), C.c_uint32(t)), "psm_rt_camera")
        self._reload()

    def camera(self, eye, view):
        ci, pi = self._sc.camera_matrices(eye, view, self.displayWidth, self.displayHeight)
        self.camera_matrices(ci, pi)

    def _reload(self):
        n = C.c_int32()
        self.ctx.check(lib().psm_rt_ray_count(self._h, C.byref(n)), "psm_rt_ray_count")
        self.raycountCache = n.value

    def getRayCount(self):
        return self.raycountCache if self.raycountCache >= 32 else 0  # Pipeline.inl:459-461

    def setTraverseMode(self, mode):
        """Tuning knob (psm_rt_set_traverse_mode): which kernel schedule intersection() runs as -- "auto", "whole",
        "phased", "adaptive" or a TRAVERSE_* value. Results never depend on it."""
        m = TRAVERSE_MODES[mode] if isinstance(mode, str) else int(mode)
        self.ctx.check(lib().psm_rt_set_traverse_mode(self._h, C.c_int(m)), "psm_rt_set_traverse_mode")

    def setTraversePhases(self, caps, min_rays=1 << 20):
        """psm_rt_set_traverse_phases: wave-step caps of the launches an intersection() is cut into (selects "phased")."""
        arr = (C.c_uint32 * max(len(caps), 1))(*caps)
        self.ctx.check(lib().psm_rt_set_traverse_phases(self._h, arr, C.c_uint32(len(caps)), C.c_uint32(min_rays)),
                       "psm_rt_set_traverse_phases")

    def setTraverseAdaptive(self, min_live=12, min_steps=8, final_rays=65536, max_launches=4, min_rays=1 << 19):
        """psm_rt_set_traverse_adaptive: parameters of the "adaptive" schedule (does not select it)."""
        self.ctx.check(lib().psm_rt_set_traverse_adaptive(self._h, C.c_uint32(min_live), C.c_uint32(min_steps),
                                                          C.c_uint32(final_rays), C.c_uint32(max_launches),
                                                          C.c_uint32(min_rays)), "psm_rt_set_traverse_adaptive")

    def setTraverseSolo(self, solo_max=1):
        """psm_rt_set_traverse_solo: a traversal wave left with at most solo_max rays (0..4) walks them one at a time with all its
        lanes on one ray; 0 switches the gear off. Results never depend on it."""
        self.ctx.check(lib().psm_rt_set_traverse_solo(self._h, C.c_uint32(solo_max)), "psm_rt_set_traverse_solo")

    def resetHits(self):
        """Forget the hit chains of the current queue (ray.hit = -1): the next intersection() starts afresh."""
        self.ctx.check(lib().psm_rt_reset_hits(self._h), "psm_rt_reset_hits")

    def intersection(self, obj, clearDepth=0, force=False):
        """force=True skips the local >=32 rule (tile-sharded frames decide on the global count).
        Called with several hierarchies before shade(), the hits chain across them (multi-BVH,
        directTraverse.comp:219-249,335-346); download_hits() then reports tri | object_sequence << 27."""
        if obj is None or obj.triangleCount <= 0:
            return 0
        if (self.raycountCache if force else self.getRayCount()) <= 0:
            return 0
        self._obj = obj
        self.ctx.check(lib().psm_rt_traverse(self._h, obj._h), "psm_rt_traverse")
        return 1

    def applyMaterials(self, mat):
        sig = (id(mat), len(mat.submats), mat.loadOffset)
        if sig != self._mat_sig:
            arr = self._sc.materials_array(mat.submats)
            self.ctx.check(lib().psm_rt_set_materials(self._h, _p(arr), C.c_uint32(arr.shape[0]),
                                                      C.c_int32(mat.loadOffset)), "psm_rt_set_materials")
            self._mat_sig = sig
        ts = mat.texset
        if ts is not None and (id(ts), ts.revision) != self._tex_sig:
            for i in range(1, 32):
                img = ts.textures[i] if i < len(ts.textures) else None
                if img is None:
                    self.ctx.check(lib().psm_rt_set_texture(self._h, C.c_uint32(i), None, C.c_uint32(0), C.c_uint32(0)),
                                   "psm_rt_set_texture")
                else:
                    self.ctx.check(lib().psm_rt_set_texture(self._h, C.c_uint32(i), _p(img), C.c_uint32(img.shape[1]),
                                                            C.c_uint32(img.shape[0])), "psm_rt_set_texture")
            self._tex_sig = (id(ts), ts.revision)

    def shade(self, time=None, force=False, reload=True):
        """reload=False leaves raycountCache stale: the caller learns the count elsewhere (ray_count_dev +
        an all-gather) and reports it with set_ray_count()."""
        if not force and self.getRayCount() <= 0:
            return
        t = self._rand() if time is None else time  # drawn every round so sharded ranks stay in step
        if self.raycountCache <= 0:
            return
        self.ctx.check(lib().psm_rt_shade(self._h, self._obj._h, C.c_uint32(t)), "psm_rt_shade")
        if reload:
            self._reload()

    def reclaim(self):
        pass  # Pipeline.inl:361-369 is a no-op

    def sample(self):
        self.ctx.check(lib().psm_rt_sample(self._h), "psm_rt_sample")

    def render(self):
        pass  # display quad: out of scope (render.vert/frag)

    def snapHdr(self, raw=False):
        out = np.zeros((self.displayHeight, self.displayWidth, 4), np.float32)
        self.ctx.check(lib().psm_rt_snap(self._h, _p(out), C.c_int(int(raw))), "psm_rt_snap")
        return out

    def snapRawHdr(self):
        return self.snapHdr(True)

    # parity / debug access
    def download_rays(self):
        n = self.raycountCache
        out = np.zeros(max(n, 1), RAY_DT)
        cnt = C.c_uint32()
        self.ctx.check(lib().psm_rt_download_rays(self._h, _p(out), C.c_uint32(n), C.byref(cnt)), "psm_rt_download_rays")
        return out[:n]

    def upload_rays(self, rays):
        rays = np.ascontiguousarray(rays, RAY_DT)
        self.ctx.check(lib().psm_rt_upload_rays(self._h, _p(rays), C.c_uint32(rays.shape[0])), "psm_rt_upload_rays")
        self.raycountCache = rays.shape[0]

    def download_hits(self, n):
        hits = np.zeros((max(n, 1), 8), HIT_DT)
        counts = np.zeros(max(n, 1), np.int32)
        self.ctx.check(lib().psm_rt_download_hits(self._h, _p(hits), _p(counts), C.c_uint32(n)), "psm_rt_download_hits")
        return hits[:n], counts[:n]

    def download_texels(self):
        wh = self.width * self.height
        s = np.zeros((wh, 4), np.float32)
        c = np.zeros((wh, 2), np.float32)
        f = np.zeros(wh, np.int32)
        self.ctx.check(lib().psm_rt_download_texels(self._h, _p(s), _p(c), _p(f)), "psm_rt_download_texels")
        return s, c, f

    def get_texels_dev(self, y0, y1, dev_ptr):
        self.ctx.check(lib().psm_rt_get_texels_dev(self._h, C.c_uint32(y0), C.c_uint32(y1), C.c_void_p(dev_ptr)),
                       "psm_rt_get_texels_dev")

    def set_texels_dev(self, y0, y1, dev_ptr):
        self.ctx.check(lib().psm_rt_set_texels_dev(self._h, C.c_uint32(y0), C.c_uint32(y1), C.c_void_p(dev_ptr)),
                       "psm_rt_set_texels_dev")

    def close(self):
        if self._h:
            lib().psm_rt_destroy(self._h)
            self._h = C.c_void_p()


def write_pfm(path, image):
    """Save an HDR snapshot (snapHdr(); the app writes EXR on key L, Application.hpp:324-343) as a PFM file: 'PF', width height,
    scale -1.0 (little endian), RGB float32, rows BOTTOM TO TOP -- which is the order snapHdr() returns them in: row 0 of the
    sampler's image is the picture's bottom row (camera.comp:61: texel row 0 is NDC y = -1; the reference's HdrImage and FreeImage's
    scanlines are bottom-up alike), so the rows are written as they come."""
    img = np.ascontiguousarray(image[..., :3], np.float32)
    with open(path, "wb") as f:
        f.write(("PF\n%d %d\n-1.0\n" % (img.shape[1], img.shape[0])).encode())
        f.write(img.astype("<f4").tobytes())


def read_pfm(path):
    """The image of a PFM file in snapHdr()'s row order (row 0 = the picture's bottom row)."""
    with open(path, "rb") as f:
        assert f.readline().strip() == b"PF"
        w, h = (int(v) for v in f.readline().split())
        scale = float(f.readline())
        data = np.frombuffer(f.read(), "<f4" if scale < 0 else ">f4").reshape(h, w, 3)
    return data.copy()


def write_exr(path, image):
    """Save an HDR snapshot as the app does on key L (PathTracerApplication::saveHdr, Application.hpp:324-343: FreeImage FIT_RGBAF ->
    FIF_EXR, EXR_FLOAT): an OpenEXR scan-line file, channels A B G R as 32-bit floats, one scan line per chunk, NO_COMPRESSION
    (the app asks FreeImage for PIZ; which lossless compression a file uses is invisible to its readers). The app copies row r of
    snapHdr()'s image into FreeImage scan line r, which FreeImage counts from the BOTTOM: the file's first (top) scan line is the
    image's last row. `image`: [h, w, 3 or 4] float; a missing alpha is written as 1."""
    import struct
    img = np.asarray(image, np.float32)
    h, w = img.shape[:2]
    rgba = np.ones((h, w, 4), np.float32)
    rgba[..., :img.shape[2]] = img[..., :4]

    def attr(name, typ, payload):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(payload)) + payload
    chlist = b"".join(n + b"\0" + struct.pack("<iB3xii", 2, 0, 1, 1) for n in (b"A", b"B", b"G", b"R")) + b"\0"   # 2 = FLOAT
    box = struct.pack("<iiii", 0, 0, w - 1, h - 1)
    header = (struct.pack("<ii", 20000630, 2) + attr("channels", "chlist", chlist) + attr("compression", "compression", b"\0") +
              attr("dataWindow", "box2i", box) + attr("displayWindow", "box2i", box) + attr("lineOrder", "lineOrder", b"\0") +
              attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) + attr("screenWindowCenter", "v2f", struct.pack("<ff", 0.0, 0.0)) +
              attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0")
    line_bytes = 4 * w * 4
    first = len(header) + 8 * h
    with open(path, "wb") as f:
        f.write(header)
        f.write(np.arange(h, dtype="<u8").__mul__(8 + line_bytes).__add__(first).astype("<u8").tobytes())   # offset table
        for y in range(h):
            row = rgba[h - 1 - y]                                   # top scan line first
            f.write(struct.pack("<ii", y, line_bytes))
            f.write(np.ascontiguousarray(row[:, [3, 2, 1, 0]].T, "<f4").tobytes())   # channel by channel: A, B, G, R


def read_exr(path):
    """The image of an uncompressed 32-bit-float scan-line OpenEXR file (what write_exr writes; any channel set), in snapHdr()'s
    row order: [h, w, 4] RGBA, a channel the file lacks reads 0 (alpha: 1)."""
    import struct
    data = open(path, "rb").read()
    magic, version = struct.unpack_from("<ii", data, 0)
    assert magic == 20000630 and (version & 0xFF) == 2 and not (version & 0x200), "not a scan-line OpenEXR 2 file"
    pos, attrs = 8, {}
    while data[pos] != 0:
        e = data.index(b"\0", pos); name = data[pos:e].decode(); pos = e + 1
        e = data.index(b"\0", pos); typ = data[pos:e].decode(); pos = e + 1
        (size,) = struct.unpack_from("<i", data, pos); pos += 4
        attrs[name] = (typ, data[pos:pos + size]); pos += size
    pos += 1
    assert attrs["compression"][1] == b"\0" and attrs["lineOrder"][1] == b"\0", "only uncompressed, increasing-y files"
    x0, y0, x1, y1 = struct.unpack("<iiii", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    chans, cp, cl = [], 0, attrs["channels"][1]
    while cl[cp] != 0:
        e = cl.index(b"\0", cp); nm = cl[cp:e].decode(); cp = e + 1
        ptype, = struct.unpack_from("<i", cl, cp); cp += 16
        assert ptype == 2, "only 32-bit float channels"
        chans.append(nm)
    out = np.zeros((h, w, 4), np.float32)
    out[..., 3] = 1.0
    offs = np.frombuffer(data, "<u8", h, pos)
    for y in range(h):
        o = int(offs[y])
        yy, nbytes = struct.unpack_from("<ii", data, o)
        assert nbytes == 4 * w * len(chans)
        line = np.frombuffer(data, "<f4", w * len(chans), o + 8).reshape(len(chans), w)
        for k, nm in enumerate(chans):
            if nm in "RGBA":
                out[h - 1 - (yy - y0), :, "RGBA".index(nm)] = line[k]
    return out


def sharded_rounds(rays, intersector, materials, depth=16):
    """The bounce loop of Viewer.cpp:304-310 for one tile of a sharded frame, as a generator: yields
    the local ray count and is sent the GLOBAL count (sum over tiles), so the reference's
    `getRayCount() < 32 -> stop` rule (Pipeline.inl:459-461) is applied to the whole frame and the
    sharded image equals the unsharded one."""
    rays.applyMaterials(materials)
    for _ in range(depth):
        total = yield rays.raycountCache
        if total < 32:
            break
        rays.intersection(intersector, force=True)
        rays.shade(force=True)
        rays.reclaim()


class LaneResult(C.Structure):
    _fields_ = [("rounds", C.c_uint32), ("rays", C.c_uint64)]


class FrameBatch:
    """Several frames in flight on one GPU (psm_lanes_render): `lanes` independent copies of
    GltfViewer::process() (Viewer.cpp:296-312), each on its own context / HIP stream with its own
    TriangleHierarchy and Pipeline, so one frame's traversal tail overlaps the other frames' kernels.
    lanes[0].rays is the accumulating Pipeline: after a batch the lanes are folded into it in frame order
    (psm_rt_sample_from), which gives the image the same frames rendered one after another would give.

    rand(): the accumulating stream (setSeed) hands every frame one draw; that draw seeds the frame's own
    CRT-rand() stand-in, from which its camera() and shade() calls draw (Pipeline.inl:282,426)."""

    class Lane:
        def __init__(self, ctx, th, rays):
            self.ctx, self.th, self.rays = ctx, th, rays

    def __init__(self, lanes, width, height, device=0, seed=1, streams=None, display=None, master_stream=None):
        self.n = lanes
        self.lanes = []
        for s in range(lanes):
            ctx = Context(device, stream=None if streams is None else streams[s])
            th = TriangleHierarchy(ctx)
            rt = Pipeline(ctx, seed=seed)
            rt.resizeBuffers(width, height)
            rt.resize(*(display or (width, height)))
            self.lanes.append(FrameBatch.Lane(ctx, th, rt))
        # the accumulating Pipeline only samples: it has its own context so that folding a finished frame
        # never queues behind a lane's tracing kernels
        self.master_ctx = Context(device, stream=master_stream)
        self.master = Pipeline(self.master_ctx, seed=seed)
        self.master.resizeBuffers(width, height)
        self.master.resize(*(display or (width, height)))
        self.width, self.height = width, height
        self.frames_rendered = 0

    def pipelines(self):
        """every Pipeline that traces"""
        for ln in self.lanes:
            yield ln.rays

    # -- scene: every lane holds the same scene ------------------------------------------------------
    def allocate(self, n):
        for ln in self.lanes:
            ln.th.allocate(n)

    def loadTriangles(self, tris, normals=None, mats=None, texcoords=None):
        for ln in self.lanes:
            ln.th.loadTriangles(tris, normals, mats, texcoords)

    def loadMesh(self, mesh):
        for ln in self.lanes:
            ln.th.loadMesh(mesh)

    def clearTribuffer(self):
        for ln in self.lanes:
            ln.th.clearTribuffer()

    def each(self, fn):
        """Apply a setter to every lane's Pipeline: batch.each(lambda r: r.setSkybox(img))."""
        for r in self.pipelines():
            fn(r)

    def applyMaterials(self, materials):
        for r in self.pipelines():
            r.applyMaterials(materials)

    def setSeed(self, seed):
        self.master.setSeed(seed)

    # -- frames -----------------------------------------------------------------------------------------
    def frame_seeds(self, frames):
        return [self.master._rand() for _ in range(frames)]

    def trace(self, cam_inv, proj_inv, seeds, depth=16, rebuild=True, optimization=None, fold=True):
        """len(seeds) frames, up to `lanes` in flight. fold=True: sample() each into the accumulating Pipeline in
        frame order. fold=False (len(seeds) <= lanes): frame f stays in lane f for the caller to fold.
        Returns per-frame (rounds, rays)."""
        k = len(seeds)
        if k == 0:
            return []
        n = min(self.n, k)
        rts = (C.c_void_p * n)(*[ln.rays._h for ln in self.lanes[:n]])
        bvhs = (C.c_void_p * n)(*[ln.th._h for ln in self.lanes[:n]])
        sd = (C.c_uint32 * k)(*[v & 0xFFFFFFFF for v in seeds])
        res = (LaneResult * k)()
        ci = np.ascontiguousarray(cam_inv, np.float32).reshape(16)
        pi = np.ascontiguousarray(proj_inv, np.float32).reshape(16)
        opt = None if optimization is None else np.ascontiguousarray(optimization, np.float64).reshape(16)
        rc = lib().psm_lanes_render(rts, bvhs, C.c_uint32(n), _p(ci), _p(pi), sd, C.c_uint32(k), C.c_uint32(depth),
                                    C.c_int(int(rebuild)), _p(opt) if opt is not None else None,
                                    self.master._h if fold else None, res)
        self.lanes[0].ctx.check(rc, "psm_lanes_render")
        for ln in self.lanes[:n]:
            ln.th._dirty = False
            ln.rays._obj = ln.th
        return [(res[f].rounds, res[f].rays) for f in range(k)]

    def fold(self, k):
        """sample() for frames left in lanes 0..k-1 by trace(fold=False), in frame order."""
        for ln in self.lanes[:k]:
            self.master.ctx.check(lib().psm_rt_sample_from(self.master._h, ln.rays._h), "psm_rt_sample_from")

    # -- tile-sharded frames: lanes run free until their LOCAL count parks them (dist.run_batch_sharded) ----
    def run_sharded(self, seeds=None, cam_inv=None, proj_inv=None, force_until=None, depth=16, rebuild=True):
        """seeds given: begin len(seeds) frames (frame f on lane f) and run them until every lane is parked.
        seeds None: resume the parked frames, lane s at least up to round force_until[s].
        Returns (rounds, local_counts) per lane."""
        if seeds is not None:
            k = len(seeds)
            self._k = k
            self._rts = (C.c_void_p * k)(*[ln.rays._h for ln in self.lanes[:k]])
            self._bvhs = (C.c_void_p * k)(*[ln.th._h for ln in self.lanes[:k]])
            self._state = (C.c_uint32 * k)(*[v & 0xFFFFFFFF for v in seeds])
            self._rounds = (C.c_uint32 * k)()
            self._cam = (np.ascontiguousarray(cam_inv, np.float32).reshape(16), np.ascontiguousarray(proj_inv, np.float32).reshape(16))
        k = self._k
        fu = (C.c_uint32 * k)(*([0] * k if force_until is None else [int(v) for v in force_until]))
        counts = (C.c_int32 * k)()
        rc = lib().psm_lanes_run_sharded(self._rts, self._bvhs, C.c_uint32(k), _p(self._cam[0]), _p(self._cam[1]), self._state,
                                         self._rounds, fu, C.c_uint32(depth), C.c_int(int(seeds is not None)),
                                         C.c_int(int(rebuild)), None, counts)
        self.lanes[0].ctx.check(rc, "psm_lanes_run_sharded")
        if seeds is not None:
            for ln in self.lanes[:k]:
                ln.th._dirty = False
                ln.rays._obj = ln.th
        return list(self._rounds), list(counts)

    def render_batch_sharded(self, native, seeds, cam_inv, proj_inv, depth=16, rebuild=True):
        """len(seeds) tile-sharded frames in flight, start to finish, in the C ABI (psm_dist_render_batch): rounds,
        the (round, count) exchanges, one tile gather per frame and, on rank 0, the fold in frame order."""
        k = len(seeds)
        rts = (C.c_void_p * k)(*[ln.rays._h for ln in self.lanes[:k]])
        bvhs = (C.c_void_p * k)(*[ln.th._h for ln in self.lanes[:k]])
        sd = (C.c_uint32 * k)(*[v & 0xFFFFFFFF for v in seeds])
        rounds = (C.c_uint32 * k)()
        ci = np.ascontiguousarray(cam_inv, np.float32).reshape(16)
        pi = np.ascontiguousarray(proj_inv, np.float32).reshape(16)
        rc = lib().psm_dist_render_batch(native._h, rts, bvhs, C.c_uint32(k), _p(ci), _p(pi), sd, C.c_uint32(depth), C.c_int(int(rebuild)),
                                         None, self.master._h, rounds)
        self.lanes[0].ctx.check(rc, "psm_dist_render_batch")
        for ln in self.lanes[:k]:
            ln.th._dirty = False
            ln.rays._obj = ln.th
        return list(rounds)

    def render_frames_sharded(self, native, seeds, cam_inv, proj_inv, depth=16, rebuild=True, lanes=None):
        """len(seeds) tile-sharded frames with all lanes (or the first `lanes` of them) in flight and no drain between batches
        (psm_dist_render_frames)."""
        k, n = len(seeds), (self.n if lanes is None else max(1, min(int(lanes), self.n)))
        rts = (C.c_void_p * n)(*[ln.rays._h for ln in self.lanes[:n]])
        bvhs = (C.c_void_p * n)(*[ln.th._h for ln in self.lanes[:n]])
        sd = (C.c_uint32 * max(k, 1))(*[v & 0xFFFFFFFF for v in seeds])
        rounds = (C.c_uint32 * max(k, 1))()
        ci = np.ascontiguousarray(cam_inv, np.float32).reshape(16)
        pi = np.ascontiguousarray(proj_inv, np.float32).reshape(16)
        rc = lib().psm_dist_render_frames(native._h, rts, bvhs, C.c_uint32(n), _p(ci), _p(pi), sd, C.c_uint32(k), C.c_uint32(depth),
                                          C.c_int(int(rebuild)), None, self.master._h, rounds)
        self.lanes[0].ctx.check(rc, "psm_dist_render_frames")
        for ln in self.lanes[:n]:
            ln.th._dirty = False
            ln.rays._obj = ln.th
        return list(rounds)[:k]

    def fold_one(self, lane):
        self.master.ctx.check(lib().psm_rt_sample_from(self.master._h, lane.rays._h), "psm_rt_sample_from")

    def render(self, frames, eye, view, depth=16, rebuild=True):
        """`frames` x process() with `lanes` frames in flight; returns per-frame (rounds, rays)."""
        sc = self.master._sc
        ci, pi = sc.camera_matrices(eye, view, self.master.displayWidth, self.master.displayHeight)
        out = self.trace(ci, pi, self.frame_seeds(frames), depth, rebuild)
        self.frames_rendered += frames
        return out

    def snapHdr(self, raw=False):
        return self.master.snapHdr(raw)

    def clearSampler(self):
        self.master.clearSampler()

    def contexts(self):
        for ln in self.lanes:
            yield ln.ctx

    def sync(self):
        for c in self.contexts():
            c.sync()
        self.master_ctx.sync()

    def close(self):
        for ln in self.lanes:
            ln.rays.close()
            ln.th.close()
        for ln in self.lanes:
            ln.ctx.close()
        self.lanes = []
        self.master.close()
        self.master_ctx.close()


def render_frame(rays, intersector, materials, eye, view, depth=16):
    """GltfViewer::process(), Source/Examples/Viewer.cpp:296-312 (minus display)."""
    materials.loadToVGA()
    intersector.markDirty()
    intersector.build()
    rays.camera(eye, view)
    rounds = 0
    for _ in range(depth):
        if rays.getRayCount() <= 0:
            break
        rays.intersection(intersector)
        rays.applyMaterials(materials)
        rays.shade()
        rays.reclaim()
        rounds += 1
    rays.sample()
    rays.render()
    return rounds
