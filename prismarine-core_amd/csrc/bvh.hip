// bvh.hip -- HLBVH builder for gfx950: bounds, Morton + leaf records, emit, boxes.
//
// Replaces ShadersSDK/hlbvh/{minmax,aabbmaker,build-new,child-link,refit}.comp and the host
// orchestration of psm::TriangleHierarchy::build (Include/Prismarine/TriangleHierarchy.inl:206-329).
//
// MI355X-first structure (DESIGN.md "build"):
//   * no host readbacks: bounds -> fit transform (double, on device) -> Morton -> sort -> emit all
//     run back to back on one stream; the leaf count lives in device memory
//   * the reference emits one BVH level per dispatch (build-new.comp, tens of levels) and refits
//     with ONE workgroup (refit.comp); here every internal node is identified by its split gap s
//     (between sorted leaves s and s+1), finds its own range [first,last] from the sorted keys,
//     and reads its children's boxes from a min/max segment tree over the sorted leaf boxes --
//     one launch for the whole tree, no atomics, no level synchronisation.  The tree SHAPE is
//     the reference's (findSplit, build-new.comp:33-56); node numbering is an internal choice
//     (SURVEY a-9: numbering never changes results).
#include "psm_common.h"
#include "psm_internal.h"

namespace psm {

// d_small layout (uint32 words)
constexpr int SM_M = 0;        // 16 floats: fit transform M
constexpr int SM_BOUNDS = 16;  // 8 ordered ints: min xyzw, max xyzw (exact, before the pad)
constexpr int SM_COUNT = 24;   // leaf count
constexpr int SM_ROOT = 25;    // root link
constexpr int SM_BFLOAT = 26;  // 8 floats: bounds after the -+1e-5 pad
constexpr int SM_M0 = 34;      // 16 floats: first-pass transform inverse(opt)
constexpr int SM_TICKET = 50;  // workgroups of bvh_bounds that have finished (the last one computes the fit transform and clears this)

PSM_D int32_t float_to_ordered(float f) {
    int32_t i = (int32_t)f2u(f);
    return i >= 0 ? i : (i ^ 0x7fffffff);
}
PSM_D float ordered_to_float(int32_t i) { return u2f((uint32_t)(i >= 0 ? i : (i ^ 0x7fffffff))); }

// ---- stage: bounds (hlbvh/minmax.comp:50-79 + TriangleHierarchy.inl:248-267) -------------------

// Runs when the optimisation matrix is (first) uploaded, not per build: the first-pass transform inverse(opt)
// (TriangleHierarchy.inl:226-232) and the neutral elements of the bounds reduction (minmax.comp:56); every build's last bounds
// workgroup leaves the reduction words neutral again for the next one.
__global__ void bvh_init_bounds(uint32_t* sm, const double* opt) {
    if (threadIdx.x == 0) {
        double mat[16];
        gm_first_pass(opt, mat);  // TriangleHierarchy.inl:226-232
        for (int r = 0; r < 4; r++)
            for (int cc = 0; cc < 4; cc++) sm[SM_M0 + 4 * r + cc] = f2u((float)mat[4 * cc + r]);
        for (int c = 0; c < 4; c++) {
            sm[SM_BOUNDS + c] = (uint32_t)float_to_ordered(100000.f);       // minmax.comp:56
            sm[SM_BOUNDS + 4 + c] = (uint32_t)float_to_ordered(-100000.f);
        }
        sm[SM_COUNT] = 0;
        sm[SM_ROOT] = (uint32_t)-1;
        sm[SM_TICKET] = 0;
    }
}

// TriangleHierarchy.inl:257-267: mat = inverse(translate(mn) * scale(mx - mn)) * inverse(opt),
// double precision, cast to float -- evaluated on the device so the build never leaves the stream.
PSM_D void fit_transform(uint32_t* sm, const double* opt, const int32_t* bounds) {
    float mn[4], mx[4];
    for (int c = 0; c < 4; c++) {
        mn[c] = ordered_to_float(bounds[c]) - 0.00001f;  // minmax.comp:76-77
        mx[c] = ordered_to_float(bounds[4 + c]) + 0.00001f;
        sm[SM_BFLOAT + c] = f2u(mn[c]);
        sm[SM_BFLOAT + 4 + c] = f2u(mx[c]);
    }
    float scale[3], offset[3];
    for (int c = 0; c < 3; c++) { scale[c] = mx[c] - mn[c]; offset[c] = mn[c]; }
    double mat[16];
    gm_fit(scale, offset, opt, mat);
    for (int r = 0; r < 4; r++)
        for (int cc = 0; cc < 4; cc++) sm[SM_M + 4 * r + cc] = f2u((float)mat[4 * cc + r]);
}

// The bounds of the transformed vertices (minmax.comp:50-79) AND the fit transform in one launch (round 5; three launches before:
// init, reduce, fit -- 21 us of a 0.15 ms C3 build were their launch latencies): every workgroup reduces its share (wave shuffles,
// LDS, eight ordered-int atomics), takes a ticket, and the workgroup that draws the last one -- every other one's atomics are
// complete, in L2, by then -- reads the result, evaluates the fit transform and leaves the reduction words neutral for the next build.
constexpr int BOUNDS_BLOCK = 1024;   // (256 workgroups of 1024: four times the loads in flight of round 4's 256 x 256 on C5's 360 MB; 1024 workgroups of 256 are
                                     // equal alone and 4 % slower on a tile with 12 frames in flight, profiles/r05_wg_shapes_in_flight.txt)
__global__ __launch_bounds__(BOUNDS_BLOCK) void bvh_bounds(const float* __restrict__ pos, uint32_t n, uint32_t* sm, const double* opt) {
    __shared__ float red[8][BOUNDS_BLOCK / 64];
    __shared__ uint32_t s_last;
    float M[16];
#pragma unroll
    for (int i = 0; i < 16; i++) M[i] = u2f(sm[SM_M0 + i]);
    float mn[4] = {100000.f, 100000.f, 100000.f, 100000.f};
    float mx[4] = {-100000.f, -100000.f, -100000.f, -100000.f};
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
        const float* p = pos + (size_t)9 * t;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            float o[4];
            mat_vec(M, p[3 * k], p[3 * k + 1], p[3 * k + 2], 1.0f, o);
#pragma unroll
            for (int c = 0; c < 4; c++) { mn[c] = pmin(mn[c], o[c]); mx[c] = pmax(mx[c], o[c]); }
        }
    }
#pragma unroll
    for (int c = 0; c < 4; c++) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            mn[c] = pmin(mn[c], __shfl_xor(mn[c], d, 64));
            mx[c] = pmax(mx[c], __shfl_xor(mx[c], d, 64));
        }
    }
    int w = threadIdx.x >> 6;
    if (lane_id() == 0) {
#pragma unroll
        for (int c = 0; c < 4; c++) { red[c][w] = mn[c]; red[4 + c][w] = mx[c]; }
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        int c = threadIdx.x;
        float v = red[c][0];
        for (int q = 1; q < BOUNDS_BLOCK / 64; q++) v = (c < 4) ? pmin(v, red[c][q]) : pmax(v, red[c][q]);
        if (c < 4) atomicMin((int32_t*)&sm[SM_BOUNDS + c], float_to_ordered(v));
        else atomicMax((int32_t*)&sm[SM_BOUNDS + c], float_to_ordered(v));
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();   // this workgroup's eight atomics before its ticket
        s_last = atomicAdd(&sm[SM_TICKET], 1u) == gridDim.x - 1u ? 1u : 0u;
    }
    __syncthreads();
    if (s_last == 0u || threadIdx.x != 0) return;
    __threadfence();
    int32_t b[8];
    for (int c = 0; c < 8; c++) b[c] = __hip_atomic_load((int32_t*)&sm[SM_BOUNDS + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    fit_transform(sm, opt, b);
    for (int c = 0; c < 4; c++) {   // neutral again for the next build (minmax.comp:56)
        sm[SM_BOUNDS + c] = (uint32_t)float_to_ordered(100000.f);
        sm[SM_BOUNDS + 4 + c] = (uint32_t)float_to_ordered(-100000.f);
    }
    sm[SM_COUNT] = 0;
    sm[SM_ROOT] = (uint32_t)-1;
    sm[SM_TICKET] = 0;
}

// traversal layout: v0, e1 = v1 - v0, e2 = v2 - v0 (the first three operations of
// intersectTriangle, include/vertex.glsl:148-156, hoisted; same IEEE results)
__global__ __launch_bounds__(256) void bvh_prepare_tris(const float* __restrict__ pos, float4* __restrict__ tri48,
                                                        uint32_t first, uint32_t n) {
    uint32_t t = first + blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= first + n) return;
    const float* p = pos + (size_t)9 * t;
    float4 v0 = make_float4(p[0], p[1], p[2], 1.0f);
    float4 e1 = make_float4(p[3] - p[0], p[4] - p[1], p[5] - p[2], 0.0f);
    float4 e2 = make_float4(p[6] - p[0], p[7] - p[1], p[8] - p[2], 0.0f);
    tri48[(size_t)3 * t + 0] = v0;
    tri48[(size_t)3 * t + 1] = e1;
    tri48[(size_t)3 * t + 2] = e2;
}

// ---- geometry ingestion: vertex/loader.comp:32-152 ----------------------------------
struct MeshArgs {
    const float* iverts;
    uint32_t vertex_floats;
    const uint32_t* vindics;
    uint32_t index_words;
    const psm_accessor* accessors;
    const psm_buffer_view* views;
    int vertexAccessor, normalAccessor, texcoordAccessor;
    float T[16], Ti[16];
    int materialID, isIndexed, index16, nodeCount, primitiveType, loadingOffset;
    uint32_t storingOffset;
};

// readByAccessor, loader.comp:32-54 (reads beyond the pool return 0 instead of faulting)
PSM_D void read_by_accessor(const MeshArgs& a, int accessorID, uint32_t idx, float out[4]) {
    psm_accessor ac = a.accessors[accessorID];
    psm_buffer_view bv = a.views[ac.buffer_view];
    uint32_t cmps = (uint32_t)ac.components & 3u;
    uint32_t stride4 = bv.stride4 > 0 ? (uint32_t)bv.stride4 : (cmps + 1u);
    uint32_t off = idx * stride4 + (uint32_t)bv.offset4 + (uint32_t)ac.offset4;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) out[k] = (k <= cmps && off + k < a.vertex_floats) ? a.iverts[off + k] : 0.f;
}

__global__ __launch_bounds__(128) void bvh_load_mesh(MeshArgs a, float* __restrict__ pos, float* __restrict__ nrm,
                                                     int32_t* __restrict__ mats, float4* __restrict__ tri48,
                                                     float* __restrict__ tex) {
    int ct = blockIdx.x * 128 + threadIdx.x;
    if (ct >= a.nodeCount) return;
    int trp = a.primitiveType == 1 ? 4 : 3;
    v3 vertice[4], normal[4];
    float tu[4] = {0.f, 0.f, 0.f, 0.f}, tv[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < trp; i++) {
        uint32_t ptri = (uint32_t)a.loadingOffset + (uint32_t)(ct * trp + i);
        uint32_t vi = ptri;
        if (a.isIndexed != 0) {
            if (a.index16) { uint32_t wd = (ptri >> 1) < a.index_words ? a.vindics[ptri >> 1] : 0u; vi = (wd >> (16u * (ptri & 1u))) & 0xFFFFu; }  // vertex.glsl:199
            else vi = ptri < a.index_words ? a.vindics[ptri] : 0u;
        }
        float p[4], n[4] = {0.f, 0.f, 0.f, 0.f};
        read_by_accessor(a, a.vertexAccessor, vi, p);
        if (a.normalAccessor != -1) read_by_accessor(a, a.normalAccessor, vi, n);
        if (a.texcoordAccessor != -1) { float t4[4]; read_by_accessor(a, a.texcoordAccessor, vi, t4); tu[i] = t4[0]; tv[i] = t4[1]; }  // :94-96
        tv[i] = 1.0f - tv[i];  // INVERT_TX_Y, loader.comp:97-99 (set by build-spv-new.bat:31-32)
        float po[4], no[4];
        mat_vec(a.T, p[0], p[1], p[2], 1.0f, po);     // mult4(meshUniform.transform, vec4(vpos, 1)), :101
        matT_vec(a.Ti, n[0], n[1], n[2], 0.0f, no);   // mult4(meshUniform.transformInv, vec4(vnorm, 0)), :100
        vertice[i] = mk3(po[0] / po[3], po[1] / po[3], po[2] / po[3]);
        normal[i] = mk3(no[0], no[1], no[2]);
    }
    v3 offsetnormal = normalize3(cross3(vertice[1] - vertice[0], vertice[2] - vertice[0]));  // :119
    int istride = a.primitiveType == 1 ? 2 : 1;
    uint32_t tidc = a.storingOffset + (uint32_t)(ct * istride);
    for (int q = 0; q < istride; q++, tidc++) {
        const int m[3] = {q == 0 ? 0 : 3, q == 0 ? 1 : 0, 2};  // :56, quads: (0,1,2) and (3,0,2)
        mats[tidc] = a.materialID;
        v3 vv[3];
        for (int i = 0; i < 3; i++) {
            v3 nn = normal[m[i]];
            v3 an = mk3(pabs(nn.x), pabs(nn.y), pabs(nn.z));
            v3 use = (mlength3(an) >= 0.0001f && a.normalAccessor != -1) ? normalize3(nn) : normalize3(offsetnormal);  // :124-128
            vv[i] = vertice[m[i]];
            float* pp = pos + (size_t)9 * tidc + 3 * i;
            float* np = nrm + (size_t)9 * tidc + 3 * i;
            pp[0] = vv[i].x; pp[1] = vv[i].y; pp[2] = vv[i].z;
            np[0] = use.x; np[1] = use.y; np[2] = use.z;
            tex[(size_t)6 * tidc + 2 * i] = tu[m[i]];
            tex[(size_t)6 * tidc + 2 * i + 1] = tv[m[i]];
        }
        tri48[(size_t)3 * tidc + 0] = make_float4(vv[0].x, vv[0].y, vv[0].z, 1.0f);
        tri48[(size_t)3 * tidc + 1] = make_float4(vv[1].x - vv[0].x, vv[1].y - vv[0].y, vv[1].z - vv[0].z, 0.0f);
        tri48[(size_t)3 * tidc + 2] = make_float4(vv[2].x - vv[0].x, vv[2].y - vv[0].y, vv[2].z - vv[0].z, 0.0f);
    }
}

int launch_bvh_load_mesh(psm_bvh* b, const psm_mesh_desc* d, const psm_accessor* d_acc, const psm_buffer_view* d_views) {
    MeshArgs a;
    a.iverts = d->d_vertices; a.vertex_floats = (uint32_t)d->vertex_floats;
    a.vindics = d->d_indices; a.index_words = (uint32_t)d->index_words;
    a.accessors = d_acc; a.views = d_views;
    a.vertexAccessor = d->vertex_accessor; a.normalAccessor = d->normal_accessor; a.texcoordAccessor = d->texcoord_accessor;
    for (int i = 0; i < 16; i++) { a.T[i] = d->transform[i]; a.Ti[i] = d->transform_inv[i]; }
    a.materialID = d->material_id; a.isIndexed = d->is_indexed; a.index16 = d->index16; a.nodeCount = d->node_count;
    a.primitiveType = d->primitive_type; a.loadingOffset = d->loading_offset;
    a.storingOffset = b->tri_count;
    bvh_load_mesh<<<(d->node_count + 127) / 128, 128, 0, b->ctx->stream>>>(a, b->d_pos, b->d_nrm, b->d_mats, b->d_tri48, b->d_tex);
    PSM_HIP(b->ctx, hipGetLastError());
    return PSM_OK;
}

// ---- stage: Morton + leaf records (hlbvh/aabbmaker.comp:142-232, splitLimit = 0) --------------

struct LeafCalc {
    bool keep;
    uint64_t key;
    uint4 box;
};

PSM_D LeafCalc leaf_calc(const float* __restrict__ pos, uint32_t t, const float* M) {
    LeafCalc r;
    const float* p = pos + (size_t)9 * t;
    float v[3][4];
#pragma unroll
    for (int k = 0; k < 3; k++) mat_vec(M, p[3 * k], p[3 * k + 1], p[3 * k + 2], 1.0f, v[k]);
    float c[4];
#pragma unroll
    for (int k = 0; k < 4; k++) c[k] = ((v[0][k] + v[1][k]) + v[2][k]) * 0.33333333333333f;  // :159
    v3 s;
    s.x = (pabs(v[0][0] - c[0]) + pabs(v[1][0] - c[0])) + pabs(v[2][0] - c[0]);
    s.y = (pabs(v[0][1] - c[1]) + pabs(v[1][1] - c[1])) + pabs(v[2][1] - c[1]);
    s.z = (pabs(v[0][2] - c[2]) + pabs(v[1][2] - c[2])) + pabs(v[2][2] - c[2]);
    r.keep = !(len3(s) < 1.e-5f);  // :160
    float bmn[4], bmx[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        bmn[k] = pmin(pmin(v[0][k], v[1][k]), v[2][k]);
        bmx[k] = pmax(pmax(v[0][k], v[1][k]), v[2][k]);
    }
    r.keep = r.keep && greaterEqualF(bmx[0] - bmn[0], 0.f) && greaterEqualF(bmx[1] - bmn[1], 0.f) &&
             greaterEqualF(bmx[2] - bmn[2], 0.f);  // :176
    uint32_t q[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float f = floorf(pclamp(c[k], 0.0f, 0.99999f) * 2097152.0f);  // :187-189
        uint32_t u = (uint32_t)f;
        q[k] = u > 0x1FFFFFu ? 0x1FFFFFu : u;
    }
    r.key = morton3_64(q[0], q[1], q[2]);
#pragma unroll
    for (int k = 0; k < 4; k++) { bmn[k] = bmn[k] - PZERO; bmx[k] = bmx[k] + PZERO; }  // :193-194
    r.box = make_uint4(pack_half2(bmn[0], bmn[1]), pack_half2(bmn[2], bmn[3]), pack_half2(bmx[0], bmx[1]),
                       pack_half2(bmx[2], bmx[3]));
    return r;
}

// (Round 5 tried the scan of the counts in the count kernel's last workgroup -- a launch less -- with every workgroup publishing its
// count behind a release fence: on this chip a device-scope release writes the XCD's L2 back, and 9 766 of them took C5's Morton stage
// from 0.175 to 0.62 ms (C3: 13 -> 19 us). bvh_bounds, 256 workgroups, keeps its last-workgroup step; the leaf compaction its three launches.)
__global__ __launch_bounds__(256) void bvh_morton_count(const float* __restrict__ pos, uint32_t n,
                                                        const uint32_t* __restrict__ sm,
                                                        uint32_t* __restrict__ blockCounts) {
    __shared__ uint32_t wsum[4];
    float M[16];
#pragma unroll
    for (int i = 0; i < 16; i++) M[i] = u2f(sm[SM_M + i]);
    uint32_t t = blockIdx.x * 256 + threadIdx.x;
    bool keep = false;
    if (t < n) keep = leaf_calc(pos, t, M).keep;
    uint64_t b = __ballot(keep);
    if (lane_id() == 0) wsum[threadIdx.x >> 6] = (uint32_t)__popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) blockCounts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// exclusive scan of nb block counts in place; total -> *total_out
__global__ __launch_bounds__(1024) void scan_blocks(uint32_t* __restrict__ g, uint32_t nb, uint32_t* total_out) {
    // (one workgroup; a thread that walks its own contiguous chunk instead reads and writes one cache line per lane and
    // instruction: 55 us for C5's 39 k counts, 17 us this way)
    __shared__ uint32_t tmp[33];
    const uint32_t total = block_scan_array_1024(g, g, nb, tmp);
    if (threadIdx.x == 0 && total_out) *total_out = total;
}

// (Round 5 also built the leaf compaction as ONE launch with a decoupled look-back -- tile tickets, one {flag | count} word per tile
// published by a relaxed atomic store, 64 predecessors looked at per step by the tile's first wave: bit-exact, and SLOWER: C3 13 -> 31 us,
// C5 0.172 -> 0.906 ms. 39 061 tickets and as many exit counts on one address are ~23 ns each through the device-scope atomic path;
// the second read of the positions the three launches pay for is cheaper. With the release-fence variant above: two ways to lose.)
// canonical leaf slot `to` = rank among kept triangles in ascending t (SURVEY a-8)
__global__ __launch_bounds__(256) void bvh_morton_write(const float* __restrict__ pos, uint32_t n,
                                                        const uint32_t* __restrict__ sm,
                                                        const uint32_t* __restrict__ blockBase,
                                                        uint64_t* __restrict__ keys, uint32_t* __restrict__ idx,
                                                        uint4* __restrict__ leafbox, int32_t* __restrict__ leaftri) {
    __shared__ uint32_t wsum[4];
    float M[16];
#pragma unroll
    for (int i = 0; i < 16; i++) M[i] = u2f(sm[SM_M + i]);
    uint32_t t = blockIdx.x * 256 + threadIdx.x;
    LeafCalc lc;
    lc.keep = false;
    if (t < n) lc = leaf_calc(pos, t, M);
    uint64_t b = __ballot(lc.keep);
    int w = threadIdx.x >> 6;
    if (lane_id() == 0) wsum[w] = (uint32_t)__popcll(b);
    __syncthreads();
    uint32_t wbase = 0;
    for (int q = 0; q < w; q++) wbase += wsum[q];
    if (lc.keep) {
        uint32_t to = blockBase[blockIdx.x] + wbase + (uint32_t)__popcll(b & lanemask_lt());
        keys[to] = lc.key;
        idx[to] = to;  // :185
        leafbox[to] = lc.box;
        leaftri[to] = (int32_t)t;
    }
}

// Refit only: the leaf boxes of the triangles as they are NOW -- aabbmaker.comp:165-176,193-194 with the transform the
// hierarchy was built with -- into the slots the build gave them. The segment tree and bvh_emit then recompute every node's child
// boxes (refit.comp:21-114) from them; the sorted keys are untouched, so ranges, links and triangle ids come out as built.
__global__ __launch_bounds__(256) void bvh_refit_leaves(const float* __restrict__ pos, const uint32_t* __restrict__ sm,
                                                        const int32_t* __restrict__ leaftri, uint4* __restrict__ leafbox) {
    float M[16];
#pragma unroll
    for (int i = 0; i < 16; i++) M[i] = u2f(sm[SM_M + i]);
    const uint32_t s = blockIdx.x * 256 + threadIdx.x;
    if (s >= sm[SM_COUNT]) return;
    leafbox[s] = leaf_calc(pos, (uint32_t)leaftri[s], M).box;
}

// ---- stage: emit + boxes ---------------------------------------------------------------------

// sortable-key box: mins in x,y (min-reduced), maxes in z,w (max-reduced), per 16-bit lane
PSM_D uint4 box_union(uint4 a, uint4 b) {
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    uint4 r;
    r.x = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(us2, a.x), __builtin_bit_cast(us2, b.x)));
    r.y = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(us2, a.y), __builtin_bit_cast(us2, b.y)));
    r.z = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(us2, a.z), __builtin_bit_cast(us2, b.z)));
    r.w = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(us2, a.w), __builtin_bit_cast(us2, b.w)));
    return r;
}
PSM_D uint4 box_identity() { return make_uint4(0xffffffffu, 0xffffffffu, 0u, 0u); }
PSM_D uint4 box_to_key(uint4 h) { return make_uint4(half2_to_key(h.x), half2_to_key(h.y), half2_to_key(h.z), half2_to_key(h.w)); }
PSM_D uint4 key_to_box(uint4 k) { return make_uint4(key_to_half2(k.x), key_to_half2(k.y), key_to_half2(k.z), key_to_half2(k.w)); }

PSM_D uint4 shfl_down4(uint4 v, int d) {
    return make_uint4(__shfl_down(v.x, d, 64), __shfl_down(v.y, d, 64), __shfl_down(v.z, d, 64), __shfl_down(v.w, d, 64));
}

// Builds 8 levels of the min/max segment tree per launch. Level L entry j covers sorted leaves
// [j*2^L, (j+1)*2^L). Level 0 is gathered through the sorted MortonIndices (child-link.comp:34-45).
// lvl[q] = entry offset of level (L0+q) inside seg; in_count = entries of level L0 (upper bound).
struct SegLevels {
    uint32_t off[9];
};
template <bool GATHER>
__global__ __launch_bounds__(256) void bvh_segtree(uint4* __restrict__ seg, SegLevels lv, uint32_t in_count_max,
                                                   const uint32_t* __restrict__ sm, const uint32_t* __restrict__ idx,
                                                   const uint4* __restrict__ leafbox, const int32_t* __restrict__ leaftri,
                                                   int32_t* __restrict__ sorted_tri, int L0, int nvalid) {
    __shared__ uint4 sh[4];
    uint32_t count = sm[SM_COUNT];
    uint32_t in_count = GATHER ? count : ((count + (1u << L0) - 1u) >> L0);
    uint32_t k = blockIdx.x * 256 + threadIdx.x;
    uint4 v = box_identity();
    if (GATHER) {
        if (k < in_count) {
            uint32_t leaf = idx[k];
            v = box_to_key(leafbox[leaf]);
            sorted_tri[k] = leaftri[leaf];
        }
        if (k < in_count_max) seg[lv.off[0] + k] = v;
    } else {
        if (k < in_count) v = seg[lv.off[0] + k];
    }
    int l = lane_id();
#pragma unroll
    for (int s = 1; s <= 6; s++) {
        uint4 o = shfl_down4(v, 1 << (s - 1));
        v = box_union(v, o);
        if (s < nvalid && (l & ((1 << s) - 1)) == 0) {
            uint32_t j = k >> s;
            if (j < ((in_count_max + (1u << s) - 1u) >> s)) seg[lv.off[s] + j] = v;
        }
    }
    if (l == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint4 a = box_union(sh[0], sh[1]), b = box_union(sh[2], sh[3]);
        uint32_t j7 = k >> 7;
        if (7 < nvalid && j7 < ((in_count_max + 127u) >> 7)) seg[lv.off[7] + j7] = a;
        if (7 < nvalid && j7 + 1 < ((in_count_max + 127u) >> 7)) seg[lv.off[7] + j7 + 1] = b;
        uint32_t j8 = k >> 8;
        if (8 < nvalid && j8 < ((in_count_max + 255u) >> 8)) seg[lv.off[8] + j8] = box_union(a, b);
    }
}

// Levels 9 and up of a tree whose level 8 has at most 4096 entries (up to 2^20 leaves), in ONE workgroup: level 8 -- written by the
// launch before -- goes to LDS, every further level is the pairwise union of the one below, kept in LDS for the next and written
// out. (Round 5: bvh_segtree<false> took one launch per eight levels -- two for C3's 18 levels, each a 5-us gap in a 0.14 ms build.)
struct SegTop {
    uint32_t off[32];
};
__global__ __launch_bounds__(1024) void bvh_segtree_top(uint4* __restrict__ seg, SegTop lv, uint32_t n_max, int nlev) {
    __shared__ uint4 a[4096];
    __shared__ uint4 b[2048];
    uint4* prev = a;
    uint4* next = b;
    uint32_t m = (n_max + 255u) >> 8;   // entries of level 8 (upper bound; entries beyond the leaf count hold the identity)
    for (uint32_t j = threadIdx.x; j < m; j += 1024) prev[j] = seg[lv.off[8] + j];
    __syncthreads();
    for (int k = 9; k < nlev; k++) {
        const uint32_t mk = (m + 1u) >> 1;
        for (uint32_t j = threadIdx.x; j < mk; j += 1024) {
            const uint4 v = box_union(prev[2 * j], 2 * j + 1 < m ? prev[2 * j + 1] : box_identity());
            next[j] = v;
            seg[lv.off[k] + j] = v;
        }
        __syncthreads();
        uint4* t = prev; prev = next; next = t;   // (the level above is at most half as long: it fits whichever array is free)
        m = mk;
    }
}

struct SegTree {
    const uint4* seg;
    uint32_t off[32];
    uint32_t nlev;   // levels the tree has (entries of off beyond it repeat its end)
};

// union of sorted leaves [f, l]
PSM_D uint4 seg_query(const SegTree& st, uint32_t f, uint32_t l) {
    uint32_t lo = f, hi = l + 1;
    uint4 acc = box_identity();
    int level = 0;
    while (lo < hi) {
        if (lo & 1u) { acc = box_union(acc, st.seg[st.off[level] + lo]); lo++; }
        if (hi & 1u) { hi--; acc = box_union(acc, st.seg[st.off[level] + hi]); }
        lo >>= 1; hi >>= 1; level++;
    }
    return acc;
}

// findSplit, hlbvh/build-new.comp:33-56
PSM_D int find_split(const uint64_t* __restrict__ keys, int first, int last) {
    uint64_t firstCode = keys[first];
    uint64_t lastCode = keys[last];
    int split = (first + last) >> 1;
    if (firstCode != lastCode) {
        split = first;
        int commonPrefix = nlz64(firstCode ^ lastCode);
        int step = last - first;
        for (int i = 0; i < 8192; i++) {
            step = (step + 1) >> 1;
            int newSplit = split + step;
            if (newSplit < last) {
                uint64_t splitCode = keys[newSplit];
                if (nlz64(firstCode ^ splitCode) > commonPrefix) split = newSplit;
            }
            if (step <= 1) break;
        }
    }
    return min(max(split, first), last - 1);
}

// One thread per split gap s (internal node). The node whose findSplit lands on gap s is unique:
//  * keys differ across the gap: it is the radix-tree node whose range is every key sharing the
//    first delta = nlz(key[s]^key[s+1]) bits with key[s];
//  * keys equal across the gap: the gap lies inside a run of equal keys [a,b]; the run is a node
//    of the reference tree and below it findSplit halves ranges at (first+last)>>1.
// RECORDS: also write the per-node records in the reference's terms (child boxes as 2 x uvec4, links, ranges) that
// psm_bvh_download hands out for parity checks. Traversal reads node32 only, so a build writes 32 of the 80 bytes per
// node; the records are produced on demand by running the kernel again on the build's (still resident) inputs.
// RECORDS = false (the build): the traversal records. A node does not search for its children's splits -- two more binary
// searches per node, and a wave waits for the longest of its 64 -- : every node finds its PARENT, which costs two key
// reads, and writes its own id into the parent's record; a parent writes its child boxes and the links of its LEAF children.
// The parent of the node over [f, l] splits at gap l (the node is its left child) or at gap f - 1 (right child): inside a
// run of equal codes it is the range of the median recursion one level up; otherwise it is the side whose neighbour shares
// the longer prefix with the node (nlz(key[l] ^ key[l+1]) against nlz(key[f-1] ^ key[f]); they cannot be equal: the two
// neighbours would have to differ from the node in the same bit, one above and one below it). The node over [0, count-1]
// is the root. Every dword of a record has exactly one writer.
// RECORDS = true: the per-node records in the reference's terms (child boxes as 2 x uvec4, links found by findSplit as
// build-new.comp does, ranges) that psm_bvh_download hands out for parity checks, produced on demand from the build's
// (still resident) inputs; it does not touch the traversal records, so the tests can hold one against the other.
template <bool RECORDS>
__global__ __launch_bounds__(256) void bvh_emit(const uint64_t* __restrict__ keys, const int32_t* __restrict__ sorted_tri,
                                                SegTree st, uint32_t* __restrict__ sm, uint4* __restrict__ pairbox,
                                                int2* __restrict__ link, int2* __restrict__ range,
                                                uint4* __restrict__ node32) {
    int count = (int)sm[SM_COUNT];
    int s = blockIdx.x * 256 + threadIdx.x;
    if (!RECORDS && s == 0 && count < 2) sm[SM_ROOT] = (uint32_t)-1;
    // (no early return: the wave searches its widest ranges together, below, and needs all its lanes for that)
    const bool live = s < count - 1;
    // Everything a lane reads on its own lies within 64 leaves of its gap (its gallop reaches 63 keys, the tree nodes of a range of
    // up to 64 leaves cover leaves of that range only): the workgroup stages that window -- 385 sorted keys and the segment tree's
    // levels 0..6 over them, 18 KB -- in LDS with coalesced loads, and the ten or so DEPENDENT reads of a node's search and of its two
    // box queries become LDS reads (the kernel was bound by exactly that chain: 4.4 L2 requests per node at 15 % of the L2's rate).
    constexpr int WIN = 64, WKEYS = 256 + 2 * WIN + 1;
    __shared__ uint64_t wkey[WKEYS];
    __shared__ uint4 wseg[WKEYS + WKEYS / 2 + WKEYS / 4 + WKEYS / 8 + WKEYS / 16 + WKEYS / 32 + WKEYS / 64 + 14];
    const int b0 = blockIdx.x * 256;
    const int w0 = max(b0 - WIN, 0), w1 = min(b0 + 256 + WIN + 1, count);   // keys [w0, w1)
    int lvl_off[8], lvl_lo[7];   // level k's entries [w0 >> k, ((w1 - 1) >> k) + 1) start at wseg[lvl_off[k]]
    {
        int at = 0;
#pragma unroll
        for (int k = 0; k < 7; k++) {
            lvl_off[k] = at; lvl_lo[k] = w0 >> k;
            at += w1 > w0 ? (((w1 - 1) >> k) - (w0 >> k) + 1) : 0;
        }
        lvl_off[7] = at;
        for (int i = threadIdx.x; i < w1 - w0; i += 256) wkey[i] = keys[w0 + i];
#pragma unroll
        for (int k = 0; k < 7; k++)
            if (k < (int)st.nlev)   // (a small tree has fewer levels; no range reaches the ones it lacks)
                for (int i = threadIdx.x; i < lvl_off[k + 1] - lvl_off[k]; i += 256) wseg[lvl_off[k] + i] = st.seg[st.off[k] + (uint32_t)(lvl_lo[k] + i)];
        __syncthreads();
    }
    auto key_at = [&](int i) -> uint64_t { return (i >= w0 && i < w1) ? wkey[i - w0] : keys[i]; };
    if (!live) s = 0;
    const uint64_t ks = count > 0 ? key_at(s) : 0ull, ks1 = count > 1 ? key_at(s + 1) : 0ull;
    // The node's range [f, l]: the keys around the gap that share the gap's common prefix -- nlz(key ^ ks) >= delta -- or, inside
    // a run of equal codes (delta = 64 says exactly that), the run. A lane gallops up to 63 keys away on its own; one gap in 64
    // belongs to a node over more than that, and its search is what the whole wave used to wait for: up to 2 x 23 dependent key
    // reads at 10 M leaves. Those ends are now found by the WAVE, one open end at a time: 64 probes per step, the distance
    // growing (then shrinking) by a factor of 64 -- three steps for a range of 4096 keys, seven for 10 M.
    const int D = ks != ks1 ? nlz64(ks ^ ks1) : 64;
    constexpr int SOLO_REACH = 32;   // the largest gallop step a lane takes alone
    int lo = s, hi = s + 1;
    bool openL = false, openR = false;
    if (live) {
        int step = 1;
        for (;;) {   // leftmost f with nlz(key[f] ^ ks) >= D (monotone towards s); lo: known inside
            if (lo - step < 0) break;
            if (step > SOLO_REACH) { openL = true; break; }
            if (nlz64(key_at(lo - step) ^ ks) < D) break;
            lo -= step; step <<= 1;
        }
        if (!openL) {   // answer in (lo - step, lo]
            int bad = max(lo - step, -1);
            while (lo - bad > 1) {
                int mid = (lo + bad) >> 1;
                if (nlz64(key_at(mid) ^ ks) >= D) lo = mid; else bad = mid;
            }
        }
        step = 1;
        for (;;) {
            if (hi + step >= count) break;
            if (step > SOLO_REACH) { openR = true; break; }
            if (nlz64(key_at(hi + step) ^ ks) < D) break;
            hi += step; step <<= 1;
        }
        if (!openR) {
            int bad = min(hi + step, count);
            while (bad - hi > 1) {
                int mid = (hi + bad) >> 1;
                if (nlz64(key_at(mid) ^ ks) >= D) hi = mid; else bad = mid;
            }
        }
    }
    {   // the open ends, one after the other, all lanes on one search (wave-uniform state; a lane's result lands in its lo / hi)
        const int lj = lane_id();
        for (int side = 0; side < 2; side++) {
            unsigned long long open = __builtin_amdgcn_ballot_w64(side == 0 ? openL : openR);
            uint64_t last_k = 0; int last_D = -1, last_at = 0;   // neighbours inside one run of equal codes ask the same question
            while (open != 0ull) {
                const int L = __builtin_ctzll(open);
                open &= open - 1ull;
                const uint32_t klo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)ks, L), khi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(ks >> 32), L);
                const uint64_t k0 = ((uint64_t)khi << 32) | klo;
                const int D0 = __builtin_amdgcn_readlane(D, L);
                long long at = __builtin_amdgcn_readlane(side == 0 ? lo : hi, L);   // known inside
                if (D0 == 64 && last_D == 64 && k0 == last_k) {
                    at = last_at;
                } else {
                    const long long dir = side == 0 ? -1 : 1;
                    long long w = 1;
                    for (;;) {   // outwards: probes at at + dir * (j + 1) * w
                        const long long p = at + dir * (long long)(lj + 1) * w;
                        const bool in = p >= 0 && p < count && nlz64(keys[p] ^ k0) >= D0;
                        const unsigned long long m = __builtin_amdgcn_ballot_w64(in);
                        if (m == ~0ull) { at += dir * 64 * w; w *= 64; continue; }
                        at += dir * (long long)__builtin_ctzll(~m) * w;   // the last probe inside (none: `at` stays)
                        break;
                    }
                    while (w > 1) {   // inwards: the end lies within w keys beyond `at`
                        w /= 64;
                        const long long p = at + dir * (long long)(lj + 1) * w;
                        const bool in = lj < 63 && p >= 0 && p < count && nlz64(keys[p] ^ k0) >= D0;
                        const unsigned long long m = __builtin_amdgcn_ballot_w64(in);
                        at += dir * (long long)__builtin_ctzll(~m) * w;
                    }
                    last_k = k0; last_D = D0; last_at = (int)at;
                }
                if (lj == L) { if (side == 0) lo = (int)at; else hi = (int)at; }
            }
        }
    }
    int f = lo, l = hi;
    int parent = -1;       // the gap the parent splits at; -1: not known yet
    bool is_left = false;  // this node is its parent's left child
    if (live && ks == ks1) {   // inside a run of equal codes [f, l]: findSplit halves ranges at (first + last) >> 1
        for (;;) {
            int m = (f + l) >> 1;
            if (s == m) break;
            parent = m;                 // the range one level up splits at m; s lies on ...
            is_left = s < m;            // ... its left side [f, m] or its right side [m + 1, l]
            if (s < m) l = m; else f = m + 1;
        }
    }
    // The child boxes: unions of the sorted leaves [f, s] and [s + 1, l] from the min / max segment tree (refit.comp:91-98). A range
    // of up to 64 leaves is at most 7 levels of the tree, walked by the lane; a wider one -- again the one node in 64 the wave would
    // wait for, up to 2 x 24 levels -- is gathered by the WAVE: the tree nodes of a range [A, B) are known in closed form (level k
    // contributes entry ceil(A / 2^k) if that is odd and entry floor(B / 2^k) - 1 if floor(B / 2^k) is odd, while ceil < floor), so lane
    // 2k takes the low candidate of level k, lane 2k + 1 the high one, and six shuffles unite them (min / max: any order).
    uint4 cb[2];
    {
        const uint32_t qa[2] = {(uint32_t)f, (uint32_t)(s + 1)}, qb[2] = {(uint32_t)s + 1u, (uint32_t)l + 1u};   // [qa, qb)
        const int lj = lane_id();
#pragma unroll
        for (int side = 0; side < 2; side++) {
            const bool wide = live && qb[side] - qa[side] > 64u;
            cb[side] = box_identity();
            if (live && !wide) {   // seg_query over the staged levels: the range lies inside the window
                uint32_t qlo = qa[side], qhi = qb[side];
#pragma unroll
                for (int k = 0; k < 7; k++) {
                    if (qlo < qhi) {
                        if (qlo & 1u) { cb[side] = box_union(cb[side], wseg[lvl_off[k] + (int)qlo - lvl_lo[k]]); qlo++; }
                        if (qhi & 1u) { qhi--; cb[side] = box_union(cb[side], wseg[lvl_off[k] + (int)qhi - lvl_lo[k]]); }
                        qlo >>= 1; qhi >>= 1;
                    }
                }
            }
            unsigned long long open = __builtin_amdgcn_ballot_w64(wide);
            while (open != 0ull) {
                const int L = __builtin_ctzll(open);
                open &= open - 1ull;
                const uint32_t A = (uint32_t)__builtin_amdgcn_readlane((int)qa[side], L), B = (uint32_t)__builtin_amdgcn_readlane((int)qb[side], L);
                const uint32_t k = (uint32_t)lj >> 1;                       // this lane's level (A, B < 2^28: the sums below fit)
                const uint32_t lo_k = k < 28u ? (A + ((1u << k) - 1u)) >> k : 1u, hi_k = k < 28u ? B >> k : 0u;
                const bool take = lo_k < hi_k && (((lj & 1) == 0 ? lo_k : hi_k) & 1u) != 0u;
                uint4 v = take ? st.seg[st.off[k] + ((lj & 1) == 0 ? lo_k : hi_k - 1u)] : box_identity();
#pragma unroll
                for (int d = 32; d > 0; d >>= 1)
                    v = box_union(v, make_uint4(__shfl_xor(v.x, d, 64), __shfl_xor(v.y, d, 64), __shfl_xor(v.z, d, 64), __shfl_xor(v.w, d, 64)));
                if (lj == L) cb[side] = v;
            }
        }
    }
    if (!live) return;
    uint4 lb = key_to_box(cb[0]);
    uint4 rb = key_to_box(cb[1]);
    if (RECORDS) {
        // children [f,s] and [s+1,l] (splitNode, build-new.comp:70-117; leaf link child-link.comp:34-53)
        int2 lk;
        lk.x = (f == s) ? ~sorted_tri[s] : find_split(keys, f, s);
        lk.y = (s + 1 == l) ? ~sorted_tri[l] : find_split(keys, s + 1, l);
        pairbox[2 * (size_t)s + 0] = lb;
        pairbox[2 * (size_t)s + 1] = rb;
        link[s] = lk;
        range[s] = make_int2(f, l);
        return;
    }
    // traversal record (trace.hip): xyz of both child boxes (the w halves are never read by the
    // slab test) + both links = 32 bytes, one aligned pair of 16-byte loads per visit
    uint32_t* rec = (uint32_t*)(node32 + 2 * (size_t)s);
    node32[2 * (size_t)s + 0] = make_uint4(lb.x, (lb.y & 0xffffu) | (lb.z << 16), (lb.z >> 16) | (lb.w << 16), rb.x);
    *(uint2*)(rec + 4) = make_uint2((rb.y & 0xffffu) | (rb.z << 16), (rb.z >> 16) | (rb.w << 16));
    if (f == s) rec[6] = (uint32_t)~sorted_tri[s];        // leaf children: the parent's to write
    if (s + 1 == l) rec[7] = (uint32_t)~sorted_tri[l];
    // this node's own link, into its parent's record
    if (f == 0 && l == count - 1) {
        sm[SM_ROOT] = (uint32_t)s;
        return;
    }
    if (parent < 0) {
        const int dL = f > 0 ? nlz64(key_at(f - 1) ^ key_at(f)) : -1;
        const int dR = l < count - 1 ? nlz64(key_at(l) ^ key_at(l + 1)) : -1;
        is_left = dR > dL;
        parent = is_left ? l : f - 1;
    }
    ((uint32_t*)(node32 + 2 * (size_t)parent))[is_left ? 6 : 7] = (uint32_t)s;
}

// ---- launch wrappers ----------------------------------------------------------------------------

int launch_bvh_prepare_tris(psm_bvh* b, uint32_t first, uint32_t n) {
    if (n == 0) return PSM_OK;
    bvh_prepare_tris<<<(n + 255) / 256, 256, 0, b->ctx->stream>>>(b->d_pos, b->d_tri48, first, n);
    PSM_HIP(b->ctx, hipGetLastError());
    return PSM_OK;
}

// the first-pass transform and the reduction's neutral elements: when the optimisation matrix is uploaded (api.hip: bvh_upload_opt)
int launch_bvh_opt_changed(psm_bvh* b) {
    bvh_init_bounds<<<1, 64, 0, b->ctx->stream>>>(b->d_small, b->d_opt);
    PSM_HIP(b->ctx, hipGetLastError());
    return PSM_OK;
}

int launch_bvh_bounds(psm_bvh* b) {
    psm_ctx* c = b->ctx;
    TimedScope ts(c, CAT_BOUNDS);
    uint32_t n = b->tri_count;
    uint32_t grid = std::max(1u, std::min((n + BOUNDS_BLOCK - 1u) / BOUNDS_BLOCK, 256u));  // one workgroup per CU: the 8 atomics per workgroup on the same 8 words are the cost
    bvh_bounds<<<grid, BOUNDS_BLOCK, 0, c->stream>>>(b->d_pos, n, b->d_small, b->d_opt);
    PSM_HIP(c, hipGetLastError());
    return PSM_OK;
}

int launch_bvh_morton(psm_bvh* b) {
    psm_ctx* c = b->ctx;
    TimedScope ts(c, CAT_MORTON);
    uint32_t n = b->tri_count;
    uint32_t nb = (n + 255u) / 256u;
    if (nb == 0) return PSM_OK;
    bvh_morton_count<<<nb, 256, 0, c->stream>>>(b->d_pos, n, b->d_small, b->d_block);
    scan_blocks<<<1, 1024, 0, c->stream>>>(b->d_block, nb, b->d_small + SM_COUNT);
    bvh_morton_write<<<nb, 256, 0, c->stream>>>(b->d_pos, n, b->d_small, b->d_block, b->d_keys, b->d_idx,
                                                b->d_leafbox, b->d_leaftri);
    PSM_HIP(c, hipGetLastError());
    return PSM_OK;
}

int launch_bvh_refit_leaves(psm_bvh* b) {
    psm_ctx* c = b->ctx;
    TimedScope ts(c, CAT_MORTON);
    const uint32_t n = b->tri_count;   // upper bound of the leaf count
    if (n == 0) return PSM_OK;
    bvh_refit_leaves<<<(n + 255u) / 256u, 256, 0, c->stream>>>(b->d_pos, b->d_small, b->d_leaftri, b->d_leafbox);
    PSM_HIP(c, hipGetLastError());
    return PSM_OK;
}

int launch_bvh_emit(psm_bvh* b) {
    psm_ctx* c = b->ctx;
    TimedScope ts(c, CAT_EMIT);
    uint32_t n = b->tri_count;  // upper bound of the leaf count
    if (n == 0) return PSM_OK;
    int nlev = (int)b->seg_off.size() - 1;  // levels 0..nlev-1
    const bool top = nlev > 9 && ((n + 255u) >> 8) <= 4096u;   // everything above level 8 in one workgroup
    for (int L0 = 0; L0 < (top ? 8 : nlev); L0 += 8) {
        SegLevels lv;
        for (int q = 0; q < 9; q++) lv.off[q] = (uint32_t)b->seg_off[(size_t)std::min(L0 + q, nlev)];
        uint32_t in_max = (n + (1u << L0) - 1u) >> L0;
        uint32_t grid = (in_max + 255u) / 256u;
        if (L0 == 0)
            bvh_segtree<true><<<grid, 256, 0, c->stream>>>(b->d_seg, lv, in_max, b->d_small, b->d_idx, b->d_leafbox,
                                                           b->d_leaftri, b->d_sorted_tri, L0, nlev - L0);
        else
            bvh_segtree<false><<<grid, 256, 0, c->stream>>>(b->d_seg, lv, in_max, b->d_small, nullptr, nullptr, nullptr,
                                                            nullptr, L0, nlev - L0);
    }
    if (top) {
        SegTop lv;
        for (int q = 0; q < 32; q++) lv.off[q] = (uint32_t)b->seg_off[(size_t)std::min(q, nlev)];
        bvh_segtree_top<<<1, 1024, 0, c->stream>>>(b->d_seg, lv, n, nlev);
    }
    SegTree st;
    st.seg = b->d_seg;
    for (int q = 0; q < 32; q++) st.off[q] = (uint32_t)b->seg_off[(size_t)std::min(q, nlev)];
    st.nlev = (uint32_t)nlev;
    uint32_t grid = (n + 255u) / 256u;
    bvh_emit<false><<<grid, 256, 0, c->stream>>>(b->d_keys, b->d_sorted_tri, st, b->d_small, b->d_pairbox, b->d_link,
                                                 b->d_range, b->d_node32);
    b->records_valid = false;
    PSM_HIP(c, hipGetLastError());
    return PSM_OK;
}

// the per-node records of the last build (psm_bvh_download: PAIR_BOX, LINK, RANGE), produced when first asked for
int launch_bvh_emit_records(psm_bvh* b) {
    psm_ctx* c = b->ctx;
    uint32_t n = b->tri_count;
    if (n == 0 || b->records_valid) return PSM_OK;
    int nlev = (int)b->seg_off.size() - 1;
    SegTree st;
    st.seg = b->d_seg;
    for (int q = 0; q < 32; q++) st.off[q] = (uint32_t)b->seg_off[(size_t)std::min(q, nlev)];
    st.nlev = (uint32_t)nlev;
    bvh_emit<true><<<(n + 255u) / 256u, 256, 0, c->stream>>>(b->d_keys, b->d_sorted_tri, st, b->d_small, b->d_pairbox, b->d_link,
                                                             b->d_range, b->d_node32);
    PSM_HIP(c, hipGetLastError());
    b->records_valid = true;
    return PSM_OK;
}

}  // namespace psm
