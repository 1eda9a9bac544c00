/*
 * psm_hip.h -- C ABI of the MI355X-native path-tracing core (libpsm_hip.so).
 *
 * This is the drop-in boundary for the hot path of EngineWorld/prismarine-core: the three
 * shader directories ShadersSDK/{radix,hlbvh,raytracing} and the host orchestration in
 * Include/Prismarine/{Radix.hpp,TriangleHierarchy.inl,Pipeline.inl}.  Where the reference passes
 * buffers by SSBO binding number and launches by glDispatchCompute, this ABI passes plain
 * pointers / handles and sizes.  Every entry point names the reference interface it replaces.
 *
 * Conventions
 *   - every function returns 0 on success, a negative psm_status otherwise; nothing throws
 *   - one psm_ctx per GPU, one in-order HIP stream per context (the reference's single GL
 *     context + barrier after every dispatch, Utils.hpp:167-171); all calls are asynchronous
 *     on that stream unless the doc says "synchronises"
 *   - matrices are row-major float[16] / double[16]:  (M v)[i] = sum_j M[4*i+j] v[j]
 *   - "device pointer" = hipMalloc'ed memory on the context's device
 *   - there is NO CPU fallback: without a gfx950 device psm_ctx_create fails
 */
#ifndef PSM_HIP_H
#define PSM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    PSM_OK = 0,
    PSM_ERR_INVALID = -1,     /* bad argument / handle */
    PSM_ERR_HIP = -2,         /* a HIP runtime call failed; see psm_last_error */
    PSM_ERR_NO_DEVICE = -3,   /* no gfx950 device visible */
    PSM_ERR_CAPACITY = -4,    /* exceeds an allocated capacity */
    PSM_ERR_STATE = -5,       /* call order violated (e.g. traverse before build) */
    PSM_ERR_PEER = -6         /* tile-sharded frames: another rank reported a failure or did not arrive; every rank
                                 of the communicator returns from the same psm_dist_* call with an error */
} psm_status;

typedef struct psm_ctx psm_ctx;
typedef struct psm_bvh psm_bvh;
typedef struct psm_rt psm_rt;

/* ---------------------------------------------------------------------------------------------
 * context + buffers: replaces the GL utility shim, Include/Prismarine/Utils.hpp:140-178
 * (allocateBuffer<T>, glNamedBufferSubData, glGetNamedBufferSubData, dispatch)
 * ------------------------------------------------------------------------------------------- */
int psm_ctx_create(int device, psm_ctx** out);
/* same, but every launch goes to an existing HIP stream (e.g. the one a framework's collectives run on,
 * so kernels and RCCL calls are ordered without host synchronisation); the stream is not owned */
int psm_ctx_create_on_stream(int device, void* hip_stream, psm_ctx** out);
int psm_ctx_destroy(psm_ctx* ctx);
int psm_ctx_sync(psm_ctx* ctx);
/* measured HBM ceiling of the box: device-to-device copy of `bytes`, best of `reps`, in GB/s of traffic (read + write) */
int psm_ctx_copy_bandwidth(psm_ctx* ctx, size_t bytes, int reps, double* gb_per_s);                 /* synchronises (glFinish, Viewer.cpp:314) */
void* psm_ctx_stream(psm_ctx* ctx);             /* the hipStream_t every launch goes to */
const char* psm_last_error(psm_ctx* ctx);
int psm_device_count(void);

/* handle = the GLuint buffer name the header layer passes around (Utils.hpp:140-150) */
int psm_buf_alloc(psm_ctx* ctx, size_t bytes, uint32_t* handle);
int psm_buf_free(psm_ctx* ctx, uint32_t handle);
int psm_buf_upload(psm_ctx* ctx, uint32_t handle, size_t offset, const void* src, size_t bytes);
int psm_buf_download(psm_ctx* ctx, uint32_t handle, size_t offset, void* dst, size_t bytes); /* synchronises */
int psm_buf_ptr(psm_ctx* ctx, uint32_t handle, void** dev_ptr, size_t* bytes);

/* ---------------------------------------------------------------------------------------------
 * psm::RadixSort::sort, Include/Prismarine/Radix.hpp:47-74 (+ radix/{histogram,pfx-work,
 * permute}.comp): stable ascending sort of (u64 key, u32 value) pairs, result in place.
 * The reference caps n at 2 Mi (Radix.hpp:34-35); here n is bounded by memory only.
 * ------------------------------------------------------------------------------------------- */
int psm_sort_u64_u32(psm_ctx* ctx, uint32_t keys_handle, uint32_t vals_handle, uint32_t n);
int psm_sort_u64_u32_dev(psm_ctx* ctx, uint64_t* d_keys, uint32_t* d_vals, size_t n);
/* Which implementation the sort (and the hierarchy build's sort stage) runs. Results are identical.
 * 2 (default) = hybrid: the LSD passes of the TOP sixteen key bits first (histogram / scan / scatter kernels, two passes),
 * then every workgroup sorts a chunk of whole sixteen-bit bins by the remaining digits in LDS and writes it back once
 * (radix_local): 3 moves of a key through HBM and 7 launches where the reference makes 8 x 3 dispatches (Radix.hpp:57-73).
 * A chunk whose last bin does not fit LDS is sorted through global memory by its workgroup alone (correct, slow) and the
 * context then falls back to algorithm 0 for good (psm_sort_get_algorithm shows it; setting the algorithm again clears it).
 * 0 = per pass a histogram, a scan (pfx-work.comp:34-70 as its own launch) and a scatter kernel: 256 B/key, 24 launches.
 * 1 = ONE histogram sweep over the keys for all eight digits (histogram.comp:80-116 once instead of per pass) + one
 * scatter launch per pass that finds its tile's bases by decoupled look-back: 200 B/key, 10 launches -- measured slower
 * than 0 on MI355X at every size, kept selectable and under the same parity tests. A look-back spin that
 * times out is reported as PSM_ERR_STATE by the next synchronising call on the context (psm_ctx_sync,
 * psm_buf_download, psm_bvh_get_info, psm_bvh_download): the sort never hangs silently. */
int psm_sort_set_algorithm(psm_ctx* ctx, int algorithm);
/* the algorithm asked for and the one the next sort will run (they differ after a hybrid sort overflowed); either may be NULL */
int psm_sort_get_algorithm(psm_ctx* ctx, int* asked, int* effective);

/* ---------------------------------------------------------------------------------------------
 * psm::TriangleHierarchy, Include/Prismarine/TriangleHierarchy.{hpp,inl}
 * ------------------------------------------------------------------------------------------- */
/* allocate(count), TriangleHierarchy.inl:77-112 (capacity = 2*count there; here exactly max_tris).
 * max_tris <= 2^27 (PSM_ERR_CAPACITY beyond; the reference stops at ~4.19 M, TriangleHierarchy.inl:80). */
int psm_bvh_create(psm_ctx* ctx, size_t max_tris, psm_bvh** out);
int psm_bvh_destroy(psm_bvh* bvh);
/* clearTribuffer(), TriangleHierarchy.inl:161-166 */
int psm_bvh_clear(psm_bvh* bvh);
/* loadMesh(), TriangleHierarchy.inl:173-192 + vertex/loader.comp:32-152 reduced to its result:
 * appends n world-space triangles. positions: 9 floats/triangle; normals: 9 floats/triangle as
 * stored in the normal mosaic (may be NULL -> face normals, loader.comp:119-128); mats: per
 * triangle material id (NULL -> material_id for all). Host pointers. */
int psm_bvh_load_triangles(psm_bvh* bvh, const float* positions, const float* normals,
                           const int32_t* mats, size_t n, int32_t material_id);
/* per-vertex texture coordinates (texcoords mosaic, loader.comp:131): 6 floats per triangle (u,v of the
 * three vertices) for triangles [first, first+n); host pointer. Triangles never given any read as (0,0). */
int psm_bvh_set_texcoords(psm_bvh* bvh, size_t first, const float* uv, size_t n);
/* loadMesh() proper: the accessor / buffer-view virtualisation of
 * Include/Prismarine/VertexInstance.{hpp,inl} + ShadersSDK/vertex/loader.comp:32-152 as one HIP gather
 * kernel: de-index (32- or packed 16-bit indices), read by accessor, transform, normal fallback to the
 * face normal, quads -> two triangles; appended at the current triangle count. */
typedef struct {
    int32_t offset4;      /* VirtualAccessor.offset4 (in floats) */
    int32_t components;   /* VirtualAccessor.components: 0..3 = 1..4 floats (structs.glsl:245-254) */
    int32_t buffer_view;  /* VirtualAccessor.bufferView */
} psm_accessor;
typedef struct {
    int32_t offset4, stride4;  /* VirtualBufferView (structs.glsl:229-232); stride4 <= 0 -> components+1 */
} psm_buffer_view;
typedef struct {
    const float* d_vertices;       /* device pointer: the float pool every accessor indexes (binding 1) */
    size_t vertex_floats;
    const uint32_t* d_indices;     /* device pointer or NULL (binding 2); 16-bit indices packed two per word */
    size_t index_words;
    const psm_accessor* accessors; /* host arrays, copied */
    uint32_t accessor_count;
    const psm_buffer_view* views;
    uint32_t view_count;
    int32_t vertex_accessor, normal_accessor, texcoord_accessor, modifier_accessor; /* -1 = absent */
    float transform[16];           /* row-major t (setTransform, VertexInstance.inl:54-58) */
    float transform_inv[16];       /* row-major inverse(t) */
    int32_t material_id, is_indexed, index16, node_count, primitive_type /* 1 = quads */, loading_offset;
} psm_mesh_desc;
int psm_bvh_load_mesh(psm_bvh* bvh, const psm_mesh_desc* mesh);

/* build(optimization), TriangleHierarchy.inl:206-329: bounds -> fit transform -> Morton+leaves
 * -> radix sort -> emit -> boxes. opt may be NULL (identity). No host synchronisation. */
int psm_bvh_build(psm_bvh* bvh, const double* opt);
/* The reference rebuilds with ~100 dispatches and host polls per frame (TriangleHierarchy.inl:206-329); here a rebuild is
 * 34 launches (C3), and from the second build of a triangle count on they are replayed as ONE captured hipGraph (the host's share of a
 * tiled frame drops from 31 % to 8 % of the wall time). Results are identical; enable = 0 keeps plain launches (default: 1).
 * Per-stage timing (psm_stats_enable) always uses plain launches. */
int psm_bvh_set_build_graph(psm_bvh* bvh, int enable);
/* Refit only (SURVEY f4 "refit-only dynamic updates"; the reference refits as the last stage of build() only): for a hierarchy
 * that has been built and whose triangles were reloaded since -- the same number, in the same order, moved -- recompute the leaf
 * boxes (aabbmaker.comp:165-194, with the transform of the build) and every node's child boxes bottom-up (refit.comp:21-114).
 * Topology, ranges, triangle ids, sorted keys: the build's. PSM_ERR_STATE without a complete build of the current triangle count. */
int psm_bvh_refit(psm_bvh* bvh);

typedef struct {
    uint32_t triangle_count; /* uploaded triangles (tcounter, TriangleHierarchy.inl:209) */
    uint32_t leaf_count;     /* non-degenerate triangles (aabbCounter, :280) */
    int32_t root;            /* root link: >=0 split-gap id of the root, -1 = no traversable tree */
    float transform[16];     /* geometryUniform.transform as M (row-major, not transposed) */
    float bounds_min[4], bounds_max[4]; /* minmax.comp result after the -+1e-5 pad */
} psm_bvh_info;
int psm_bvh_get_info(psm_bvh* bvh, psm_bvh_info* info); /* synchronises */

/* Stage-level entry points (each = one reference dispatch group); psm_bvh_build runs them in
 * order. Exposed so parity tests can check every stage against the oracle. */
int psm_bvh_stage_bounds(psm_bvh* bvh, const double* opt);  /* minmax.comp + host reduce + fit */
int psm_bvh_stage_morton(psm_bvh* bvh);                     /* aabbmaker.comp */
int psm_bvh_stage_sort(psm_bvh* bvh);                       /* Radix.hpp:47-74 */
int psm_bvh_stage_emit(psm_bvh* bvh);                       /* build-new + child-link + refit */

/* Debug / parity downloads (synchronise). `what` (PAIR_BOX, LINK and RANGE -- the nodes in the reference's terms -- are not
 * written by a build, whose traversal reads its own 32-byte record: the first download after a build produces them from the
 * build's still-resident inputs; PSM_ERR_STATE once the next build has begun): */
enum {
    PSM_BVH_KEYS = 0,      /* uint64[leaf_count]   sorted Morton codes (unsorted before stage_sort) */
    PSM_BVH_INDICES = 1,   /* uint32[leaf_count]   MortonIndices */
    PSM_BVH_LEAF_BOX = 2,  /* uint32[4][leaf_count] leaf record boxes (packHalf2 mn.xy mn.zw mx.xy mx.zw), slot order */
    PSM_BVH_LEAF_TRI = 3,  /* int32[leaf_count]    leaf record triangle ids, slot order */
    PSM_BVH_PAIR_BOX = 4,  /* uint32[8][leaf_count-1] child boxes of internal node (split gap) s: left, right */
    PSM_BVH_LINK = 5,      /* int32[2][leaf_count-1]  child links: >=0 internal gap id, <0 leaf: ~triangle */
    PSM_BVH_RANGE = 6,     /* int32[2][leaf_count-1]  sorted-leaf range [first,last] of internal node s */
    PSM_BVH_SORTED_TRI = 7,/* int32[leaf_count]    triangle id of the k-th sorted leaf */
    PSM_BVH_POSITIONS = 8, /* float[9][triangle_count] world-space triangle soup as loaded */
    PSM_BVH_NORMALS = 9,   /* float[9][triangle_count] per-vertex normals as loaded */
    PSM_BVH_MATERIALS = 10,/* int32[triangle_count] */
    PSM_BVH_TEXCOORDS = 11,/* float[6][triangle_count] u,v per vertex */
    PSM_BVH_NODE32 = 12    /* uint32[8][leaf_count-1] the traversal record of internal node s as the build wrote it: 12 fp16 box
                              coordinates (left mn.xyz mx.xyz, right mn.xyz mx.xyz) + the two child links, which a node writes
                              into its PARENT's record */
};
int psm_bvh_download(psm_bvh* bvh, int what, void* dst, size_t bytes);

/* ---------------------------------------------------------------------------------------------
 * psm::Pipeline, Include/Prismarine/Pipeline.{hpp,inl}
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    float lightVector[4]; /* xyz direction, w distance   (Pipeline.inl:93-98) */
    float lightColor[4];  /* rgb, w radius */
    float lightOffset[4];
    float lightAmbient[4];
} psm_light; /* LightUniformStruct, Structs.hpp:165-170 */

typedef struct {
    float diffuse[4], specular[4], transmission[4], emissive[4];
    float ior, roughness, alpharef, unk0f;
    uint32_t diffusePart, specularPart, bumpPart, emissivePart;
    int32_t flags, alphafunc, binding, bitfield;
    int32_t iModifiers0[4];
} psm_material; /* VirtualMaterial, Structs.hpp:240-262 (128 bytes) */

int psm_rt_create(psm_ctx* ctx, psm_rt** out);
int psm_rt_destroy(psm_rt* rt);
/* resizeBuffers(w,h), Pipeline.inl:174-214: ray grid; ray limit = min(4*w*h, 4096*4096) */
int psm_rt_resize_buffers(psm_rt* rt, uint32_t width, uint32_t height);
/* resize(w,h), Pipeline.inl:138-172: display image (presampled / filtered) */
int psm_rt_resize(psm_rt* rt, uint32_t display_width, uint32_t display_height);
/* tile sharding (new): this context owns ray-grid rows [y0,y1). Default all rows. */
int psm_rt_set_tile(psm_rt* rt, uint32_t y0, uint32_t y1);
/* interleaved sharding: this context owns the global 8-row bands g with g % world == rank (balances
 * sky rows against geometry rows; 1.01 max/mean on the Sponza-class view vs 1.21 for 8 contiguous strips).
 * camera() then touches only the owned texels -- except on rank 0, the rank the tiles are gathered to, which
 * also prepares the jitter positions / flags of all other texels because its sample() reads the whole image. */
int psm_rt_set_tile_interleaved(psm_rt* rt, uint32_t rank, uint32_t world);
/* the same with a weighted dealing: the bands go round in periods of P = weights[0] + ... + weights[world-1] <= 64, rank r
 * owning weights[r] bands of every period, laid out by a smooth weighted round-robin (each step every rank's credit grows
 * by its weight, the largest credit -- lowest rank on ties -- takes the band and pays P) so that a rank's bands are
 * spread evenly over the image. weights NULL = 1 each = psm_rt_set_tile_interleaved. The gathering rank also unpacks,
 * fills and samples the whole image, so it is given fewer bands than the workers (8 GPUs: 2 of every 23 against 3). */
int psm_rt_set_tile_weighted(psm_rt* rt, uint32_t rank, uint32_t world, const uint32_t* weights);
/* lightColor/lightVector/lightOffset/lightAmbient + setLightCount, Pipeline.hpp:103-121 */
int psm_rt_set_lights(psm_rt* rt, const psm_light* lights, uint32_t count);
/* environment: constant colour ... */
int psm_rt_set_sky(psm_rt* rt, const float rgba[4]);
/* ... or setSkybox(), Pipeline.hpp:93 + public/environment.glsl:23-26: an equirect RGBA8 image
 * (host pointer, width*height*4 bytes, row 0 first) sampled with GL_LINEAR / clamp-to-edge as the app's
 * loadCubemap() sets it up (Application.hpp:46-54). NULL restores the constant colour. */
int psm_rt_set_skybox(psm_rt* rt, const uint8_t* rgba8, uint32_t width, uint32_t height);
/* TextureSet (TextureSet.inl:15-35,88-122): slot 1..31 of the sampler table
 * surface.comp indexes with diffusePart / specularPart / bumpPart / emissivePart (slot 0 = none,
 * MAX_TEXTURES = 32, surface.comp:46). RGBA8, GL_LINEAR, GL_REPEAT as TextureSet::loadTexture sets them;
 * host pointer, width*height*4 bytes. rgba8 == NULL frees the slot. */
int psm_rt_set_texture(psm_rt* rt, uint32_t slot, const uint8_t* rgba8, uint32_t width, uint32_t height);
/* MaterialSet::loadToVGA + bindWithContext, MaterialSet.inl:13-23 (host pointer, copied) */
int psm_rt_set_materials(psm_rt* rt, const psm_material* mats, uint32_t count, int32_t load_offset);
/* camera(persp, frontSide), Pipeline.inl:279-296 -> camera.comp. camInv/projInv are the inverse
 * matrices the reference uploads (:283-284); `time` replaces the host rand() (:282). Clears the
 * ray counters (clearRays) and this frame's texel sums. */
int psm_rt_camera(psm_rt* rt, const float cam_inv[16], const float proj_inv[16], uint32_t time);
/* cameraUniform.enable360 (switchMode(), Pipeline.inl:128-132; camera.comp:48-59): primary rays over the whole
 * sphere (equirect image) from the camera position instead of through the projection */
int psm_rt_set_camera_mode(psm_rt* rt, int enable360);
/* raycountCache after reloadQueuedRays, Pipeline.inl:325-359 (the >=32 rule of getRayCount,
 * :459-461, is applied by the header layer). Synchronises. */
int psm_rt_ray_count(psm_rt* rt, int32_t* count);
/* intersection(obj), Pipeline.inl:385-405 -> directTraverse.comp. The first call after the ray queue changed
 * (camera, shade, upload_rays, reset_hits) starts the hit chains; further calls with other hierarchies extend
 * them as the reference's ray.hit hand-over does (multi-BVH; directTraverse.comp:219-249,335-346):
 * the search starts at the distance already found and new hits overwrite the front of the chain. At most 16
 * hierarchies of < 2^27 triangles each per queue; psm_rt_shade() then interpolates each hit from the
 * hierarchy that produced it (its `bvh` argument is used when only one was traversed). */
int psm_rt_traverse(psm_rt* rt, psm_bvh* bvh);
/* Which kernel schedule an intersection() runs as. A tuning knob: hits, chains and counters never depend on it
 * (every schedule performs, per ray, the node steps and triangle tests of directTraverse.comp:333-484 in the
 * same order). A wave64 steps as long as its slowest ray; the schedules differ in how they keep lanes busy:
 *   WHOLE       one launch, 64 consecutive rays per wave, run to completion
 *   PHASED      launch k runs at most caps[k] wave-steps, then the rays still under way hand their state (node,
 *               stack, best hit) to a dense continuation queue and the next launch resumes them packed 64 to a
 *               wave; the last launch runs to completion (psm_rt_set_traverse_phases)
 *   ADAPTIVE    the same hand-over, triggered per wave by __ballot / popcount: a wave hands over as soon as fewer
 *               than min_live of its lanes have work left; resume launches are persistent waves striding over the
 *               continuation queue (psm_rt_set_traverse_adaptive)
 *   AUTO        (default) ADAPTIVE while the Pipeline is one of several frames in flight (psm_lanes_*: the other frames'
 *               kernels fill the tails the extra launches add; +5 % on C3, +9 % on C5's scene), WHOLE for a frame on its own (it is bound
 *               by its longest ray, which extra launches serialise) and for intersections under min_rays rays.
 *               Further hierarchies of a multi-BVH queue always run WHOLE. */
enum { PSM_TRAVERSE_AUTO = 0, PSM_TRAVERSE_WHOLE = 1, PSM_TRAVERSE_PHASED = 2, PSM_TRAVERSE_ADAPTIVE = 3 };
int psm_rt_set_traverse_mode(psm_rt* rt, int mode);
/* PHASED: count caps (1..7) -> count + 1 launches, for intersections over at least min_rays rays; count = 0 selects
 * WHOLE. Selects PSM_TRAVERSE_PHASED. */
int psm_rt_set_traverse_phases(psm_rt* rt, const uint32_t* caps, uint32_t count, uint32_t min_rays);
/* ADAPTIVE parameters (does not change the mode): hand over below min_live live lanes (2..64) but not before
 * min_steps wave-steps; a resume launch that finds at most final_rays rays waiting finishes them; at most
 * max_launches launches (2..15) per intersection; intersections under min_rays rays run WHOLE.
 * Defaults: 12, 8, 65536, 4, 2^19. */
int psm_rt_set_traverse_adaptive(psm_rt* rt, uint32_t min_live, uint32_t min_steps, uint32_t final_rays,
                                 uint32_t max_launches, uint32_t min_rays);
/* The solo gear of every schedule (round 4): a traversal wave that is left with at most solo_max rays (0..4; default 1) --
 * and is not in a launch that hands rays over -- stops stepping them one lane each and walks them one after the other with
 * ALL its lanes on one ray: the ray's state is wave-uniform, so the step's decisions are scalar arithmetic instead of lane
 * masks, a lane evaluates one axis of one child box (the same v_fma_mix_f32 on the same operands; tNear / tFar by quad-permute
 * DPP max / min in the order mathlib.glsl:129-193 combines them), the stack lives in the lanes of one register, a leaf's two
 * triangles are tested side by side. A round's tail consists of such waves (a bounce round's longest ray takes 659-1099
 * steps against a mean of 56) and a frame on its own, a tile's launches and every hand-over round's last launch end with
 * them. Per ray nothing changes: the node steps and triangle tests of directTraverse.comp:333-484 in the same order, hits,
 * chains and counters bit-exact. 0 switches the gear off. */
int psm_rt_set_traverse_solo(psm_rt* rt, uint32_t solo_max);
/* forget the chains of the current queue without changing it (the reference's ray.hit = -1, rayslib.glsl:149) */
int psm_rt_reset_hits(psm_rt* rt);
/* applyMaterials + shade, Pipeline.inl:407-436 -> surface.comp + rayshading.comp, then the
 * queue hand-off of reloadQueuedRays (:325-359). `time` replaces rand() (:426). */
int psm_rt_shade(psm_rt* rt, psm_bvh* bvh, uint32_t time);
/* sample(), Pipeline.inl:251-277 -> sampler.comp, deinterlace.comp, filter.comp */
int psm_rt_sample(psm_rt* rt);
/* sample() fed with the frame another Pipeline of the same ray-grid size rendered (its texel sums and jitter
 * coordinates): folds that frame into rt's accumulating image exactly as rt's own sample() would have.
 * Stream-ordered against both contexts, no host synchronisation. */
int psm_rt_sample_from(psm_rt* rt, psm_rt* src);
/* clearSampler(), Pipeline.inl:314-322 (also zeroes presampled: the GL texture starts undefined) */
int psm_rt_clear_sampler(psm_rt* rt);
/* snapHdr()/snapRawHdr(), Pipeline.inl:439-456: display_w*display_h*4 floats to host. Synchronises. */
int psm_rt_snap(psm_rt* rt, float* rgba, int raw);

/* per-texel frame radiance (sum rgb, deposit count) for the tile gather:
 * copy rows [y0,y1) to / from a device pointer (width*(y1-y0)*4 floats). */
int psm_rt_get_texels_dev(psm_rt* rt, uint32_t y0, uint32_t y1, float* d_dst);
int psm_rt_set_texels_dev(psm_rt* rt, uint32_t y0, uint32_t y1, const float* d_src);
/* the same for any tile shape: pack this context's owned texels (row-major over its owned rows) into a
 * dense device buffer of psm_rt_tile_texels() * 4 floats; unpack the dense buffer of tile (rank, world,
 * interleaved != 0) or rows [rank, world) (interleaved == 0) into the full image on the gathering rank */
int psm_rt_tile_texels(psm_rt* rt, uint32_t* count);
int psm_rt_pack_texels_dev(psm_rt* rt, float* d_dst);
int psm_rt_unpack_texels_dev(psm_rt* rt, int interleaved, uint32_t a, uint32_t b, const float* d_src);
/* the gathering rank's side of the gather in one launch: d_all holds the dense tiles of ranks 0..world-1 of an interleaved
 * sharding back to back, stride_floats apart (a multiple of 4, at least the largest tile: what a gather of equal-sized
 * buffers delivers); every texel skip_rank does not own takes its radiance from its owner's tile */
int psm_rt_unpack_tiles_dev(psm_rt* rt, uint32_t world, uint32_t skip_rank, const float* d_all, size_t stride_floats);
/* ray count hand-off without a host read-back: copy the current count to a device int32 (on the
 * context's stream); tell the library the count the host learned elsewhere (e.g. from an all-gather) */
int psm_rt_ray_count_dev(psm_rt* rt, int32_t* d_dst);
int psm_rt_set_ray_count(psm_rt* rt, int32_t count);

/* Debug / parity downloads of the current ray queue and last traversal result (synchronise). */
typedef struct {
    float origin[3], direct[3], color[3];
    int32_t bitfield, texel;
    uint32_t pkey;
} psm_ray;
typedef struct {
    float u, v, t;
    int32_t tri;
} psm_hit;
int psm_rt_download_rays(psm_rt* rt, psm_ray* dst, uint32_t max_rays, uint32_t* count);
/* hits: max_rays*8 entries (chain of ray i at [8*i, 8*i+counts[i])) */
int psm_rt_download_hits(psm_rt* rt, psm_hit* hits, int32_t* counts, uint32_t max_rays);
/* replace the current ray queue (host pointer) -- lets tests drive traverse with chosen rays. count <= currentRayLimit
 * (PSM_ERR_CAPACITY); every ray's texel must lie inside the ray grid, 0 <= texel < w * h (PSM_ERR_INVALID): shading deposits
 * a ray's radiance into the texel it names */
int psm_rt_upload_rays(psm_rt* rt, const psm_ray* src, uint32_t count);
int psm_rt_download_texels(psm_rt* rt, float* sum_rgba, float* coord_xy, int32_t* flags);

/* ---------------------------------------------------------------------------------------------
 * several frames in flight (new; DESIGN.md "lanes")
 * `frames` x GltfViewer::process() (Viewer.cpp:296-312) with up to `lanes` of them in flight: lane s =
 * (rts[s], bvhs[s]) on its own context / stream. Frame f has its own CRT-rand() stand-in, started from
 * frame_seeds[f] (one draw for camera(), one per shade(), as Pipeline.inl:282,426 draw them) and runs: build (if
 * rebuild != 0, with `opt`), camera, at most `depth` rounds of { stop if fewer than 32 rays (Pipeline.inl:459-461);
 * intersection; shade }, then sample() -- issued on `fold_into` (psm_rt_sample_from) in frame order, so the
 * accumulated image equals the frames rendered one after another. Lanes never wait for each other's rounds: a
 * frame's traversal tail overlaps the other frames' kernels, and a lane takes the next frame as soon as its own is
 * folded. fold_into may be NULL when frames <= lanes (the lanes then keep their frames for the caller to fold).
 * Materials / lights / sky / textures / tiles must have been set on every rt. One hierarchy per lane: frames that
 * intersect several hierarchies (multi-BVH) go through the per-call API. Returns when everything is idle.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    uint32_t rounds; /* shade() calls made for this frame */
    uint64_t rays;   /* rays traced */
} psm_lane_result;
int psm_lanes_render(psm_rt* const* rts, psm_bvh* const* bvhs, uint32_t lanes, const float cam_inv[16],
                     const float proj_inv[16], const uint32_t* frame_seeds, uint32_t frames, uint32_t depth,
                     int rebuild, const double* opt, psm_rt* fold_into, psm_lane_result* results /* [frames] */);
/* The same frames when they are tile-sharded over several GPUs: the `fewer than 32 rays -> stop` rule then looks
 * at each frame's GLOBAL count. A rank with >= 32 local rays knows the global count is >= 32 too, so every lane
 * runs free (as above) until its LOCAL count drops below 32 or `depth` is reached, and then parks with its queue
 * intact. Returns when all lanes are parked: rounds[s] = rounds done, counts_out[s] = local rays waiting. The host
 * exchanges those (one small all-gather per batch as a rule: the tiles of a frame run dry in the same round) and,
 * where a frame's global count says it goes on, calls again with start = 0 and force_until[s] = the round lane s
 * must reach whatever its local count (it traces its few rays, or none, drawing its rand() every round so the
 * ranks stay in step). start != 0 begins new frames: build (if rebuild) + camera, rounds[] reset.
 * rand_state[s]: the frame's CRT-rand() stand-in state, in/out across calls. */
int psm_lanes_run_sharded(psm_rt* const* rts, psm_bvh* const* bvhs, uint32_t lanes, const float cam_inv[16],
                          const float proj_inv[16], uint32_t* rand_state, uint32_t* rounds, const uint32_t* force_until,
                          uint32_t depth, int start, int rebuild, const double* opt, int32_t* counts_out);

/* ---------------------------------------------------------------------------------------------
 * tile-sharded frames across the GPUs of one node (new; SURVEY 8(b) "proposed C ABI", 8(e); the reference has no
 * multi-GPU path). One process per GPU. The path has ONE data-path collective per frame: the gather of every tile's
 * per-texel radiance to rank 0 (RCCL over xGMI), which runs sample() on the whole image. psm_dist_allgather_i32
 * carries the (round, ray count) pairs psm_lanes_run_sharded asks the host to exchange.
 *   rank 0:   psm_dist_unique_id(id);  ... hand `id` to the other ranks (any side channel) ...
 *   all:      psm_dist_init(ctx, rank, world, id, &dist);  psm_rt_set_tile_interleaved(rt, rank, world);
 *   a frame:  camera / rounds on rt ...; psm_dist_gather_tiles(dist, rt);  rank 0: psm_rt_sample(rt)
 * Every rank makes the same psm_dist_* calls in the same order. Collectives run on the communicator's own stream,
 * ordered against the Pipeline's stream by events: no host synchronisation in psm_dist_gather_tiles.
 * ------------------------------------------------------------------------------------------- */
typedef struct psm_dist psm_dist;
int psm_dist_unique_id(uint8_t id[128]);                        /* ncclGetUniqueId */
int psm_dist_init(psm_ctx* ctx, int rank, int world, const uint8_t id[128], psm_dist** out); /* ncclCommInitRank on ctx's device; collective */
/* The same in two steps, for launchers that want to agree between them: psm_dist_prepare creates this rank's LOCAL
 * resources only (stream, events; cannot block on a peer), psm_dist_connect is the collective ncclCommInitRank. A rank
 * whose prepare failed can tell the others over the side channel before anybody is inside the collective. */
int psm_dist_prepare(psm_ctx* ctx, int rank, int world, psm_dist** out);
int psm_dist_connect(psm_dist* dist, const uint8_t id[128]);
/* The transport seam: the two exchanges of the path as a table of functions. psm_dist_init / psm_dist_connect install
 * RCCL (ncclGather / ncclAllGather on the communicator's stream, stream-ordered); psm_dist_connect_transport installs
 * the caller's table instead, and psm_dist_connect_hoststaged a transport built into the library that stages through a
 * POSIX shared-memory segment (hipMemcpy to the host, sequence counters, bounded waits), so that several processes
 * SHARING ONE GPU -- which RCCL refuses -- can drive the whole sharded scheduler against real peers (tests; never
 * chosen silently). A transport function returns 0 or non-zero; it may complete on `hip_stream` (a hipStream_t) or
 * synchronously on the host. All pointers are device pointers. */
typedef struct {
    void* user;
    /* `count` floats of every rank at d_send -> root's d_recv[world * count] in rank order (d_recv is NULL elsewhere) */
    int (*gather_f32)(void* user, const float* d_send, float* d_recv, size_t count, int root, void* hip_stream);
    /* n ints of every rank at d_send -> every rank's d_recv[world * n] in rank order */
    int (*allgather_i32)(void* user, const int32_t* d_send, int32_t* d_recv, size_t n, void* hip_stream);
    void (*destroy)(void* user);            /* may be NULL */
    const char* (*last_error)(void* user);  /* may be NULL */
    const char* name;                       /* e.g. "rccl", "host-staged" */
} psm_dist_transport;
int psm_dist_connect_transport(psm_dist* dist, const psm_dist_transport* transport);
/* shm_name: a POSIX shared-memory name every rank of the group passes ("/psm-<unique>"); rank 0 creates it and unlinks
 * it once all ranks are attached. slot_bytes >= the largest tile of a gather (16 B x the texels rank 0 owns);
 * timeout_ms bounds every wait for a peer (PSM_ERR_PEER afterwards, on every rank that waits). */
int psm_dist_connect_hoststaged(psm_dist* dist, const char* shm_name, size_t slot_bytes, uint32_t timeout_ms);
const char* psm_dist_transport_name(const psm_dist* dist);     /* NULL while not connected */
int psm_dist_destroy(psm_dist* dist);
int psm_dist_rank(const psm_dist* dist);
int psm_dist_world(const psm_dist* dist);
/* the ranks the TRANSPORT itself counts in the communicator -- RCCL: ncclCommCount (psm_dist_connect fails unless it equals `world`
 * and ncclCommUserRank equals `rank`); host-staged: the processes attached to the segment; 0 while not connected. What a bench line
 * should print next to n_gpus. */
int psm_dist_comm_ranks(const psm_dist* dist);
/* pack rt's owned texels, ncclGather them to rank 0, and there unpack the tiles of ranks 1..world-1 into rt's image
 * (rt must carry psm_rt_set_tile_interleaved(rank, world) of this communicator). 16 B per texel; 1080p: 33 MB in all. */
int psm_dist_gather_tiles(psm_dist* dist, psm_rt* rt);
/* n ints from every rank to every rank (host arrays: recv holds world * n); synchronises */
int psm_dist_allgather_i32(psm_dist* dist, const int32_t* send, int32_t* recv, uint32_t n);
int psm_dist_barrier(psm_dist* dist);
/* every rank's own status (PSM_OK or its error code) -> one verdict for all: PSM_OK when every rank is fine, this rank's
 * own code when it failed, PSM_ERR_PEER when only others did. One one-int all-gather; every rank must call it. */
int psm_dist_agree(psm_dist* dist, int local_rc);
/* the global `fewer than 32 rays -> stop` rule (Pipeline.inl:459-461) from every rank's answers: all = [world][2][lanes]
 * (rounds done, local rays waiting) as gathered after psm_lanes_run_sharded -> per lane: over (the frame has ended) and
 * force_until (the round every rank must reach next). Pure host arithmetic; needs no device.
 * A rank that failed locally reports rounds = -1 for its lanes (it keeps taking part in the collectives so that nobody
 * waits for it): any negative round makes psm_dist_decide return PSM_ERR_PEER on every rank at the same exchange. */
int psm_dist_decide(uint32_t world, uint32_t lanes, const int32_t* all, uint32_t depth, int32_t* over, uint32_t* force_until);
/* `lanes` tile-sharded frames in flight, start to finish, on this rank (every rank makes the same call): build (if
 * rebuild) + camera + rounds on lanes that run free and park on their local counts (psm_lanes_run_sharded), the
 * all-gathers + psm_dist_decide until every frame has ended, then per frame, in frame order, psm_dist_gather_tiles and on
 * rank 0 psm_rt_sample_from(fold_into, lane). rts[s] must carry psm_rt_set_tile_interleaved(rank, world); fold_into is
 * rank 0's accumulating Pipeline (ignored elsewhere); rounds_out[lanes] may be NULL.
 * Failure: a rank whose own work fails (build, kernels, capacity) keeps the collective sequence -- it reports rounds = -1
 * at the next exchange and sends an empty tile to gathers already due -- so every rank leaves at the same exchange
 * with an error (its own, or PSM_ERR_PEER) instead of waiting inside a collective; a last one-int exchange makes the
 * return code agree when the failure came after the last decision. Only a failing transport call itself cannot be
 * covered. */
int psm_dist_render_batch(psm_dist* dist, psm_rt* const* rts, psm_bvh* const* bvhs, uint32_t lanes, const float cam_inv[16],
                          const float proj_inv[16], const uint32_t* frame_seeds, uint32_t depth, int rebuild, const double* opt,
                          psm_rt* fold_into, uint32_t* rounds_out);
/* `frames` tile-sharded frames with `lanes` of them in flight and NO drain between batches: the lanes form two groups
 * that alternate batches of lanes / 2 frames -- while one group exchanges, gathers and folds, the other group's frames
 * keep the chip busy -- with the same collective sequence on every rank (one communicator). Same arguments and result
 * as psm_dist_render_batch called ceil(frames / lanes) times; frame_seeds[frames], rounds_out[frames] or NULL. */
int psm_dist_render_frames(psm_dist* dist, psm_rt* const* rts, psm_bvh* const* bvhs, uint32_t lanes, const float cam_inv[16],
                           const float proj_inv[16], const uint32_t* frame_seeds, uint32_t frames, uint32_t depth, int rebuild,
                           const double* opt, psm_rt* fold_into, uint32_t* rounds_out);
/* one-GPU rehearsal of a worker rank's per-frame cost: gathers pack tile (tile_rank, tile_world) instead of the
 * communicator's own (rank, world) and unpack nothing (the image is then not a complete frame) */
int psm_dist_emulate_tile(psm_dist* dist, int tile_rank, int tile_world);
/* the dealing of the bands this communicator's gathers pack and unpack (psm_rt_set_tile_weighted's weights; NULL =
 * round-robin, the default). Every rank passes the same weights. */
int psm_dist_set_band_weights(psm_dist* dist, const uint32_t* weights);

/* ---------------------------------------------------------------------------------------------
 * statistics (PROFILE_RT replacement, Utils.hpp:27): algorithmic counters + HIP-event timing
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    uint64_t rays_traced;     /* R: rays handed to traverse since reset */
    uint64_t node_visits;     /* V (only counted while counting is enabled) */
    uint64_t tri_tests;       /* T */
    uint64_t stack_drops, iter_caps, baked_drops, chain_pool_drops, ray_limit_drops;
                              /* chain_pool_drops: equal-distance chains cut to their head because the chain pool (currentRayLimit / 2
                               * entries beside one head per ray; Pipeline.inl:193) was full */
    uint32_t traverse_launches;
    float traverse_ms;        /* sum of HIP-event durations of the traverse kernel */
    float build_ms, sort_ms, shade_ms, camera_ms, sample_ms;
    uint32_t rounds;
    float bounds_ms, morton_ms, emit_ms; /* parts of build_ms: minmax + fit, Morton codes + leaves, node emission + link + refit */
    /* clock diagnosis, counted with V and T: over all traversal waves, the sums of their lifetimes in shader-clock ticks
     * (s_memtime) and in ticks of the constant 100 MHz clock (s_memrealtime) -- their ratio x 100 MHz is the clock the chip
     * held while they ran -- the wave-steps they took and their number */
    uint64_t wave_clock_ticks, wave_real_ticks, wave_steps, waves;
    /* the part of traverse_launches / traverse_ms that are launches of the hand-over kernel (rt_traverse<*, false, true>:
     * the PHASED / ADAPTIVE schedules, what AUTO runs with frames in flight); the rest are single-launch traversals */
    uint32_t handover_launches;
    float handover_ms;
} psm_stats;
/* timing: 0 off; 1 HIP events around every launch, per stage (the rebuild then runs as plain launches instead of its
 * captured graph); 2 traversal launches only -- light enough to stay on while frames are in flight, the build keeps
 * its graph. counting: V, T and the drop counters (the counting instantiations of the kernels). */
int psm_stats_enable(psm_ctx* ctx, int timing, int counting);
int psm_stats_reset(psm_ctx* ctx);
/* one time axis for the traversal launches of several contexts of a device (frames in flight): psm_stats_reference(ctx,
 * ctx) records the origin on ctx's stream, psm_stats_reference(other, ctx) makes `other` share it (the origin context must
 * outlive the sharing ones' next reference); psm_stats_traverse_intervals then returns start, end in ms after the origin of
 * every traversal launch timed since the last reset (start_end_ms[2 * cap_launches], *count = launches recorded) */
int psm_stats_reference(psm_ctx* ctx, psm_ctx* origin);
int psm_stats_traverse_intervals(psm_ctx* ctx, float* start_end_ms, uint32_t cap_launches, uint32_t* count);
int psm_stats_get(psm_ctx* ctx, psm_stats* out); /* synchronises */

#ifdef __cplusplus
}
#endif
#endif /* PSM_HIP_H */
