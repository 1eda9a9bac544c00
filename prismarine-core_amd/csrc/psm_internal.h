// psm_internal.h -- host-side objects behind the C ABI (include/psm_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/psm_hip.h"

namespace psm {

struct TexDesc {
    const uint32_t* texels;  // RGBA8, one texel per word, row 0 first
    int w, h;
};
constexpr int MAX_TEXTURES = 32;  // surface.comp:46
constexpr int MAX_TRAV_OBJECTS = 16;  // hierarchies chained over one ray queue (multi-BVH)
constexpr int OBJ_SHIFT = 27;          // hit.w = triangle | object << 27

struct ObjGeom {  // what interpolateMeshData reads of one hierarchy (directTraverse.comp:116-147)
    const float4* tri48;
    const float* nrm;
    const int32_t* tri_mats;
    const float* uv;
};

// state of a suspended traversal (trace.hip, phased launches)
struct TravState {      // structure of arrays in ONE allocation, `capacity` entries each (one pointer: the kernel is
    char* base;         // short of scalar registers): head float4 | idx u32 | cur i32 | misc u32 | predist f32 |
    uint32_t capacity;  // lastTri i32 | stack i32[STACK_CAP][capacity]; misc = sp | it << 8 | bakedCount << 24
    __host__ __device__ float4* head() const { return (float4*)base; }
    __host__ __device__ uint32_t* idx() const { return (uint32_t*)(base + (size_t)16 * capacity); }
    __host__ __device__ int32_t* cur() const { return (int32_t*)(base + (size_t)20 * capacity); }
    __host__ __device__ uint32_t* misc() const { return (uint32_t*)(base + (size_t)24 * capacity); }
    __host__ __device__ float* predist() const { return (float*)(base + (size_t)28 * capacity); }
    __host__ __device__ int32_t* lastTri() const { return (int32_t*)(base + (size_t)32 * capacity); }
    __host__ __device__ int32_t* stack() const { return (int32_t*)(base + (size_t)36 * capacity); }
};
enum StatCat { CAT_TRAVERSE = 0, CAT_BUILD, CAT_SORT, CAT_SHADE, CAT_CAMERA, CAT_SAMPLE, CAT_BOUNDS, CAT_MORTON, CAT_EMIT,
               CAT_TRAVERSE_HANDOVER /* launches of the hand-over kernel rt_traverse<*, false, true>; also counted in CAT_TRAVERSE */, CAT_COUNT };

// device-resident counters (one block per context)
struct DevCounters {
    unsigned long long node_visits, tri_tests, stack_drops, iter_caps, baked_drops, chain_pool_drops,
        ray_limit_drops, pad;
    // clock diagnosis (counting builds): shader-clock and 100 MHz constant-clock ticks summed over traversal waves
    unsigned long long wave_clock_ticks, wave_real_ticks, wave_steps, waves;
};

struct Buf {
    void* ptr = nullptr;
    size_t bytes = 0;
};

// Which rank owns the global 8-row band g of a tile-sharded frame: owner[g % P], a periodic dealing of P = sum of the
// ranks' weights bands in which rank r gets weights[r] (all weights 1: g % world, the round-robin dealing). The period is
// laid out by a smooth weighted round-robin, so a rank's bands are spread evenly over the image whatever its weight --
// the gathering rank is given fewer bands than the others because it also unpacks, fills and samples the whole image
// (DESIGN.md 6.1). The same function in dist.py (band_pattern); the tests hold a third statement of it.
constexpr uint32_t MAX_BAND_PERIOD = 64;
struct BandMap {
    uint32_t P = 1, world = 1;
    uint8_t owner[MAX_BAND_PERIOD] = {};   // position in the period -> rank
    uint8_t before[MAX_BAND_PERIOD] = {};  // position -> positions before it with the same owner
    uint8_t cnt[MAX_BAND_PERIOD] = {1};    // rank -> its bands per period
    // the band's owner and its index among the owner's bands
    __host__ __device__ uint32_t rank_of(uint32_t g) const { return owner[g % P]; }
    __host__ __device__ uint32_t local_band(uint32_t g) const { const uint32_t p = g % P; return (g / P) * cnt[owner[p]] + before[p]; }
};
// weights NULL: all 1. False when world or the period exceed MAX_BAND_PERIOD or every weight is 0.
inline bool band_map_make(uint32_t world, const uint32_t* weights, BandMap* m) {
    if (world == 0 || world > MAX_BAND_PERIOD) return false;
    uint64_t P = 0;
    for (uint32_t r = 0; r < world; r++) P += weights ? weights[r] : 1u;
    if (P == 0 || P > MAX_BAND_PERIOD) return false;
    *m = BandMap();
    m->P = (uint32_t)P; m->world = world;
    int64_t cur[MAX_BAND_PERIOD] = {0};
    for (uint32_t r = 0; r < MAX_BAND_PERIOD; r++) m->cnt[r] = 0;
    for (uint32_t p = 0; p < m->P; p++) {
        uint32_t pick = 0;
        for (uint32_t r = 0; r < world; r++) {
            cur[r] += weights ? weights[r] : 1u;
            if (cur[r] > cur[pick]) pick = r;
        }
        cur[pick] -= (int64_t)P;
        m->owner[p] = (uint8_t)pick;
        m->before[p] = m->cnt[pick]++;
    }
    return true;
}
inline bool band_map_equal(const BandMap& a, const BandMap& b) {
    if (a.P != b.P || a.world != b.world) return false;
    for (uint32_t p = 0; p < a.P; p++) if (a.owner[p] != b.owner[p]) return false;
    return true;
}
// texels `rank` owns of a w x h image
inline uint32_t owned_texels(const BandMap& m, uint32_t rank, uint32_t w, uint32_t h) {
    uint32_t rows = 0;
    for (uint32_t g = 0; g * 8 < h; g++)
        if (m.rank_of(g) == rank) rows += (h - g * 8) < 8u ? (h - g * 8) : 8u;
    return rows * w;
}
inline uint32_t max_owned_texels(const BandMap& m, uint32_t w, uint32_t h) {
    uint32_t best = 0;
    for (uint32_t r = 0; r < m.world; r++) { const uint32_t t = owned_texels(m, r, w, h); best = t > best ? t : best; }
    return best;
}

}  // namespace psm

struct psm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = true;
    std::string err;
    std::vector<psm::Buf> bufs;  // handle = index+1
    int timing = 0;      // 0: none; 1: HIP events around every launch (the rebuild then runs as plain launches); 2: traversal launches only
    bool counting = false;
    psm::DevCounters* d_counters = nullptr;
    struct Timed {
        hipEvent_t a, b;
        int cat;
    };
    std::vector<Timed> timed;
    std::vector<hipEvent_t> free_events;
    hipEvent_t ref_event = nullptr;     // psm_stats_reference: the time origin of `intervals` (owned by ref_owner)
    psm_ctx* ref_owner = nullptr;
    std::vector<float> intervals;       // start, end (ms after ref_event) of every timed traversal launch since the last reset
    float cat_ms[psm::CAT_COUNT] = {0};
    uint32_t cat_launches[psm::CAT_COUNT] = {0};
    uint64_t rays_traced = 0;
    uint32_t rounds = 0;
    // sort scratch (grown on demand)
    uint64_t* sort_keys_tmp = nullptr;
    uint32_t* sort_vals_tmp = nullptr;
    uint32_t* sort_hist = nullptr;
    size_t sort_cap = 0, sort_hist_cap = 0;
    uint32_t sort_gen = 0;              // bumped whenever a sort buffer is reallocated (captured build graphs hold the pointers)
    int sort_algorithm = 2;             // 2 (default): hybrid -- two global passes over the top sixteen key bits, the rest in LDS; 0: histogram / scan / scatter kernels for all eight passes; 1: one-sweep
    uint32_t* sort_overflow = nullptr;  // pinned host word a hybrid sort raises when a chunk did not fit LDS (sorted through global memory: correct, slow)
    bool sort_demoted = false;          // ... seen raised: hybrid sorts of this context run as algorithm 0 from then on (psm_sort_set_algorithm clears it)
    uint32_t sort_hybrid_s_small = 1024, sort_hybrid_s_large = 2048, sort_hybrid_threads = 1024, sort_hybrid_threads_large = 512, sort_hybrid_cap_small = 4096, sort_hybrid_cap_large = 4096;  // radix_local's stretch per workgroup, its width and its LDS capacity in keys (PSM_SORT_TUNE)
    uint32_t* sort_error_word = nullptr; // device word the look-back raises on a spin timeout
};

struct psm_bvh {
    psm_ctx* ctx = nullptr;
    size_t cap = 0;
    uint32_t tri_count = 0;
    bool built = false, bounds_done = false, morton_done = false, sort_done = false;
    uint32_t topo_tris = 0;       // triangle count of the last complete build whose sorted keys, leaf slots and transform are still on the device (psm_bvh_refit); 0: none
    float* d_pos = nullptr;       // 9 floats / triangle
    float* d_nrm = nullptr;       // 9 floats / triangle
    int32_t* d_mats = nullptr;    // material id / triangle
    float* d_tex = nullptr;       // 6 floats / triangle: u,v per vertex
    float4* d_tri48 = nullptr;    // v0, e1, e2 (xyz, w unused) / triangle -- traversal layout
    uint64_t* d_keys = nullptr;   // Morton codes, slot order then sorted in place
    uint32_t* d_idx = nullptr;    // MortonIndices
    uint4* d_leafbox = nullptr;   // leaf record box, slot order
    int32_t* d_leaftri = nullptr; // leaf record triangle, slot order
    uint32_t* d_block = nullptr;  // per-block counts / bases for the leaf compaction
    // small device block: [0..15] M (float), [16..23] bounds as ordered ints, [24] leaf count,
    // [25] root link, [26..33] bounds floats after pad
    uint32_t* d_small = nullptr;
    double* d_opt = nullptr;      // optimisation matrix (16 doubles)
    double opt_host[16] = {};     // ... as last uploaded (a rebuild with the same matrix skips the 10-us copy in front of it)
    bool opt_uploaded = false;
    uint4* d_seg = nullptr;       // segment tree of sortable-key boxes, levels concatenated
    std::vector<size_t> seg_off;  // level offsets (entries)
    int32_t* d_sorted_tri = nullptr;
    uint4* d_pairbox = nullptr;   // 2 x uint4 per internal node (split gap)
    int2* d_link = nullptr;
    int2* d_range = nullptr;
    uint4* d_node32 = nullptr;    // traversal record per internal node: 12 fp16 box coords + 2 links (32 B)
    bool records_valid = false;   // d_pairbox / d_link / d_range hold the last build's records (written on demand)
    // the build as one hipGraph (34 launches at C3): captured on the second build of a triangle count, replayed afterwards
    bool use_graph = true;
    hipGraphExec_t build_graph = nullptr;
    uint32_t graph_tris = 0, graph_sort_gen = 0, plain_builds = 0;
    int graph_algo = -1;
    uint32_t* graph_error_word = nullptr;
};

struct psm_rt {
    psm_ctx* ctx = nullptr;
    uint32_t w = 0, h = 0, dw = 0, dh = 0, y0 = 0, y1 = 0;
    uint32_t tile_mode = 0, tile_rank = 0, tile_world = 1;  // 0: rows [y0,y1); 1: the 8-row bands `bands` deals to tile_rank
    psm::BandMap bands;           // mode 1: the dealing (psm_rt_set_tile_interleaved / _weighted)
    bool tile_root = true;        // this Pipeline samples the whole image: camera() also fills the texels it does not own
    uint32_t limit = 0;           // currentRayLimit
    int cur = 0;                  // current queue index
    uint32_t ray_count = 0;       // host mirror of the current queue length (valid after sync points)
    bool count_valid = true;
    // two segmented ray queues (psm_common.h RayQueue): current and next. Capacity = one segment per shading workgroup.
    float4* qA[2] = {nullptr, nullptr};  // origin.xyz, texel
    float4* qB[2] = {nullptr, nullptr};  // direct.xyz, bitfield
    float4* qC[2] = {nullptr, nullptr};  // color.xyz, pkey
    uint32_t* q_bases[2] = {nullptr, nullptr};  // segment bases, nb + 1 entries
    uint32_t q_nb[2] = {1, 1};    // segments of each queue (1 = written densely by camera() / upload)
    uint32_t* d_block = nullptr;  // per-workgroup output counts of the shading kernel
    uint32_t* d_cnt = nullptr;    // [0] current count, [1] next count, [2] chain pool cursor
    float4* hit0 = nullptr;       // head of chain per ray: u, v, t, tri
    uint32_t* hitN = nullptr;     // chain length | pool offset << 4
    float4* pool = nullptr;       // chain entries beyond the head
    uint32_t pool_cap = 0;
    float2* t_coord = nullptr;
    float4* t_sum = nullptr;
    int32_t* t_flag = nullptr;
    float4* presampled = nullptr;
    float4* filtered = nullptr;
    psm_light* d_lights = nullptr;
    uint32_t light_count = 1;
    psm_material* d_mats = nullptr;
    uint32_t mat_count = 0;
    int32_t mat_offset = 0;
    float sky[4] = {0.5f, 0.7f, 1.0f, 1.0f};
    psm::TexDesc tex_host[psm::MAX_TEXTURES] = {};
    psm::TexDesc* d_tex_table = nullptr;
    bool tex_dirty = false;
    psm_bvh* trav_objs[psm::MAX_TRAV_OBJECTS] = {};  // hierarchies traversed since the queue last changed
    psm_bvh* last_objs[psm::MAX_TRAV_OBJECTS] = {};  // ... of the last traversal(s), kept for psm_rt_download_hits after shade() reset trav_n
    int trav_n = 0;
    psm::ObjGeom* d_geoms = nullptr;
    int enable360 = 0;            // cameraUniform.enable360 (switchMode)
    // phased traversal (trace.hip): continuation queues, allocated on first use
    void* d_phase_mem = nullptr;
    psm::TravState phase_state[2] = {};
    uint32_t* d_phase_cnt = nullptr;
    uint32_t phase_set = 0;        // which of the two sets of continuation counts the next hand-over round uses
    bool phase_dirty = true;       // the sets are not known to be clear (first use; a round that failed half way)
    uint32_t phase_cap = 0;
    uint32_t phase_caps[7] = {96};  // PSM_TRAVERSE_PHASED: wave-step caps of the launches before the last one
    int phase_caps_n = 1;
    int trav_mode = 0;              // PSM_TRAVERSE_* (psm_rt_set_traverse_mode); 0 = automatic
    uint32_t adapt_min_live = 12, adapt_min_steps = 8, adapt_final_rays = 65536, adapt_max_launches = 4;  // tuned on C3, 4 frames in flight (its 2 M-ray rounds plan three launches either way; C5's 8 M-ray rounds take the fourth: -1.3 %)
    bool mats_ordinary = true;      // no material whose dropped lobe's colour can be NaN (psm_rt_set_materials): rt_shade builds one lobe per hit
    uint32_t solo_max = 1;          // a traversal wave with at most this many rays left walks them one by one, all lanes on one ray (trace.hip: solo_ray); 0: never. 1: a frame alone 3.37 -> 3.20 ms, a 1/8 tile 0.633 -> 0.611, 4 frames in flight 2.21 -> 2.19; 2: 3.24 / 0.620 / 2.19 (profiles/r04_solo_gear.txt)
    uint32_t in_flight = 1;         // lanes this Pipeline is currently scheduled with (lanes.hip)
    uint32_t phase_min_rays = 1u << 19;  // smaller intersections run as one launch (tiles: tools/run_r02_ae.sh)
    // frames in flight (lanes.hip): pinned slot + events, created on first use
    uint32_t* h_cnt = nullptr;
    hipEvent_t ev_cnt = nullptr, ev_fold = nullptr;
    uint32_t* d_sky = nullptr;    // equirect RGBA8 skybox (one texel per word) or null
    uint32_t sky_w = 0, sky_h = 0;
    int samples_lock = 4;         // SAMPLES_LOCK, constants.glsl:35
};

// a communicator of tile-sharded frames (dist.hip); the two exchanges go through `tr` (psm_dist_transport)
struct psm_dist {
    psm_ctx* ctx = nullptr;
    psm_dist_transport tr = {};
    bool connected = false;
    hipStream_t stream = nullptr;
    hipEvent_t ev_in = nullptr, ev_out = nullptr, ev_x = nullptr;
    // while psm_dist_allgather_i32 waits for its result the caller's lanes go on being served (psm_dist_render_frames sets it)
    void (*idle_fn)(void*) = nullptr;
    void* idle_user = nullptr;
    int rank = 0, world = 1;
    int comm_ranks = 0;                 // ranks the TRANSPORT says it joins (RCCL: ncclCommCount; host-staged: the processes attached to the segment); 0 while not connected
    int tile_rank = 0, tile_world = 1;  // the tile geometry gathers use: (rank, world) unless psm_dist_emulate_tile changed it
    psm::BandMap bands;                 // the dealing of the bands (psm_dist_set_band_weights); default round-robin over tile_world
    float* d_send = nullptr;   // per_floats
    float* d_recv = nullptr;   // rank 0: world * per_floats
    size_t per_floats = 0;
    int32_t* d_i32 = nullptr;  // all-gather staging: send | recv
    size_t i32_cap = 0;
};

namespace psm {

// RAII-free helpers -----------------------------------------------------------------------------
int set_err(psm_ctx* c, int code, const char* what, hipError_t e = hipSuccess);
#define PSM_HIP(ctx, call)                                                   \
    do {                                                                     \
        hipError_t e__ = (call);                                             \
        if (e__ != hipSuccess) return psm::set_err((ctx), PSM_ERR_HIP, #call, e__); \
    } while (0)

struct TimedScope {
    psm_ctx* c;
    int idx = -1;
    TimedScope(psm_ctx* ctx, int cat);
    ~TimedScope();
};

// kernels (launch wrappers) ------------------------------------------------------------------------
// key_bits: the key bits that can be set at all (64; 63 for Morton codes) -- what the hybrid sort's global passes split by
int launch_sort(psm_ctx* c, uint64_t* d_keys, uint32_t* d_vals, size_t n_max, const uint32_t* d_n, int key_bits = 64);
int sort_effective_algorithm(psm_ctx* c);
int sort_check(psm_ctx* c);
int sort_reserve(psm_ctx* c, size_t n_max);
int launch_bvh_opt_changed(psm_bvh* b);
int launch_bvh_bounds(psm_bvh* b);
int launch_bvh_morton(psm_bvh* b);
int launch_bvh_emit(psm_bvh* b);
int launch_bvh_refit_leaves(psm_bvh* b);
int launch_bvh_emit_records(psm_bvh* b);
int launch_bvh_prepare_tris(psm_bvh* b, uint32_t first, uint32_t n);
int launch_bvh_load_mesh(psm_bvh* b, const psm_mesh_desc* d, const psm_accessor* d_acc, const psm_buffer_view* d_views);
int launch_rt_camera(psm_rt* r, const float* cam_inv, const float* proj_inv, uint32_t time);
int launch_rt_traverse(psm_rt* r, psm_bvh* b);
int launch_rt_shade(psm_rt* r, psm_bvh* b, uint32_t time);
int launch_rt_sample(psm_rt* r, psm_rt* src);
// pack / unpack the dense tile of rows [a, b) (bands == NULL) or of rank a in the dealing `bands`
int launch_rt_pack(psm_rt* r, hipStream_t stream, float* d_buf, int unpack, const BandMap* bands, uint32_t a, uint32_t b);
int launch_rt_unpack_all(psm_rt* r, hipStream_t stream, const float* d_all, const BandMap& bands, uint32_t skip, size_t stride_floats);
// dist.hip: pieces of the sharded batch that lanes.hip's pipelined scheduler shares
int dist_reserve(psm_dist* d, uint32_t w, uint32_t h);
int dist_gather_placeholder(psm_dist* d);
// `defer` (may be null): the caller takes over the Pipeline's stream waiting for the gather (and, on rank 0, for the fold) --
// one bit each, LANE_WAIT_GATHER / LANE_WAIT_FOLD -- and may put its next rebuild in front of the waits (lane_flush_waits)
bool dist_frame_gather(psm_dist* d, psm_rt* rt, psm_rt* fold_into, int& local, uint32_t* defer = nullptr);
enum { LANE_WAIT_GATHER = 1u, LANE_WAIT_FOLD = 2u };
int lane_flush_waits(psm_dist* d, psm_rt* rt, uint32_t* pending);
int rt_fold(psm_rt* r, psm_rt* src, bool* defer);   // lanes.hip: sample() of r fed with src's frame
int launch_rt_gather_queue(psm_rt* r, float4* d_dense, uint32_t m);  // current queue in queue order: A | B | C, m rays each
uint32_t tile_texel_count(const psm_rt* r);
}  // namespace psm
