// trace.hip -- closest-hit BVH traversal for gfx950.
//
// Replaces ShadersSDK/raytracing/directTraverse.comp (main :488-511, traverse :333-484) as driven
// by psm::Pipeline::intersection (Include/Prismarine/Pipeline.inl:385-405).
//
// Data layout (DESIGN.md "traversal"): an internal node is its split gap g; one visit reads
//   pairbox[2g], pairbox[2g+1]  two fp16 child boxes, 32 contiguous bytes (Nodes[x].box, Nodes[y].box, :391-392)
//   link[g]                     two child links, 8 bytes: >=0 internal gap id, <0 leaf: ~triangle
// so the reference's four dependent 32-byte AoS node fetches per visit (:391-392, :434-435, :474)
// become one 32-byte and one 8-byte load. Triangles are v0,e1,e2 as 3 x float4.
// The per-ray 16-entry stack (8 LDS + 8 global in the reference, :49-70) lives entirely in LDS,
// laid out [depth][lane] so a wave's pushes and pops never conflict.
#include "psm_common.h"
#include "psm_internal.h"

namespace psm {

constexpr int SM_M = 0;
constexpr int SM_ROOT = 25;

constexpr int TRAV_BLOCK = 128;

struct Slab {
    float hit, near, far;
};

// intersectCubeDual, include/mathlib.glsl:129-193, fp32 branch, one child
PSM_D Slab slab_child(v3 o, v3 dr, v3 norig, uint4 b) {
    v3 cmn = mk3(half_lo(b.x), half_hi(b.x), half_lo(b.y));
    v3 cmx = mk3(half_lo(b.z), half_hi(b.z), half_lo(b.w));
    float tminx = fmaf(cmn.x, dr.x, norig.x), tmaxx = fmaf(cmx.x, dr.x, norig.x);
    float tminy = fmaf(cmn.y, dr.y, norig.y), tmaxy = fmaf(cmx.y, dr.y, norig.y);
    float tminz = fmaf(cmn.z, dr.z, norig.z), tmaxz = fmaf(cmx.z, dr.z, norig.z);
    float tNear = pmax(pmax(pmin(tminx, tmaxx), pmin(tminy, tmaxy)), pmin(tminz, tmaxz));
    float tFar = pmin(pmin(pmax(tminx, tmaxx), pmax(tminy, tmaxy)), pmax(tminz, tmaxz));
    bool isCube = ((tFar + PZERO) >= tNear) && ((tFar + PZERO) >= 0.0f);
    Slab s;
    s.near = isCube ? pmin(tNear, tFar) : INF;
    s.far = isCube ? pmax(tNear, tFar) : INF;
    s.hit = ((s.near + PZERO) <= 0.0f) ? s.far : s.near;
    return s;
}

// intersectTriangle, include/vertex.glsl:140-189 (e1, e2 precomputed by bvh_prepare_tris)
PSM_D float tri_test(const float4* __restrict__ tri48, int tri, v3 orig, v3 dir, float& U, float& V) {
    float4 a = tri48[(size_t)3 * tri + 0], b = tri48[(size_t)3 * tri + 1], c = tri48[(size_t)3 * tri + 2];
    v3 v0 = mk3(a.x, a.y, a.z), e1 = mk3(b.x, b.y, b.z), e2 = mk3(c.x, c.y, c.z);
    v3 pvec = cross3(dir, e2);
    float det = dot3(e1, pvec);
    if (pabs(det) <= 0.0f) return INF;
    float invDev = 1.f / (pmax(pabs(det), 0.000001f) * psign(det));
    v3 tvec = orig - v0;
    float u = dot3(tvec, pvec) * invDev;
    if (u < -0.00001f || u > 1.00001f) return INF;
    v3 qvec = cross3(tvec, e1);
    float v = dot3(dir, qvec) * invDev;
    if (v < -0.00001f || (u + v) > 1.00001f) return INF;
    float t = dot3(e2, qvec) * invDev;
    if (!greaterEqualF(t, 0.0f)) return INF;
    U = u;
    V = v;
    return t;
}

struct Baked {
    float u, v, t;
    int tri;
};

template <bool COUNT>
__global__ __launch_bounds__(TRAV_BLOCK) void rt_traverse(const float4* __restrict__ qA, const float4* __restrict__ qB,
                                                          uint32_t nrays, const uint4* __restrict__ pairbox,
                                                          const int2* __restrict__ link,
                                                          const float4* __restrict__ tri48,
                                                          const uint32_t* __restrict__ sm, float4* __restrict__ hit0,
                                                          uint32_t* __restrict__ hitN, float4* __restrict__ pool,
                                                          uint32_t pool_cap, uint32_t* __restrict__ cnt,
                                                          DevCounters* __restrict__ ctr) {
    __shared__ int stack[STACK_CAP][TRAV_BLOCK];
    uint32_t i = blockIdx.x * TRAV_BLOCK + threadIdx.x;
    const int tid = threadIdx.x;
    bool alive = i < nrays;
    uint32_t nV = 0, nT = 0, nDrop = 0, nCap = 0, nBakedDrop = 0;

    float4 A = alive ? qA[i] : make_float4(0, 0, 0, 0);
    float4 B = alive ? qB[i] : make_float4(1, 0, 0, 0);
    v3 origin = mk3(A.x, A.y, A.z);
    v3 direct = normalize3(mk3(B.x, B.y, B.z));  // :350

    float M[16];
#pragma unroll
    for (int k = 0; k < 16; k++) M[k] = u2f(sm[SM_M + k]);
    int root = (int)sm[SM_ROOT];

    float to4[4], td4[4];
    mat_vec(M, origin.x, origin.y, origin.z, 1.0f, to4);   // :353
    matT_vec(M, direct.x, direct.y, direct.z, 1.0f, td4);  // :354
    v3 torig = mk3(to4[0], to4[1], to4[2]);
    v3 tdir = mk3(td4[0], td4[1], td4[2]);
    float dirlen = len3(tdir) / pmax(len3(direct), 0.000001f);
    float dirlenInv = 1.f / pmax(dirlen, 0.000001f);
    v3 dirproj = normalize3(tdir);

    // root slab test, intersectCubeSingle (mathlib.glsl:107-126) against [-1e-5, 1+1e-5]^3, :365
    float rootNear, rootD;
    {
        v3 dr = mk3(1.0f / dirproj.x, 1.0f / dirproj.y, 1.0f / dirproj.z);
        v3 no = mk3(-torig.x * dr.x, -torig.y * dr.y, -torig.z * dr.z);
        const float lo = -0.00001f, hi = 1.00001f;
        float a0 = fmaf(lo, dr.x, no.x), a1 = fmaf(hi, dr.x, no.x);
        float b0 = fmaf(lo, dr.y, no.y), b1 = fmaf(hi, dr.y, no.y);
        float c0 = fmaf(lo, dr.z, no.z), c1 = fmaf(hi, dr.z, no.z);
        float tNear = pmax(pmax(pmin(a0, a1), pmin(b0, b1)), pmin(c0, c1));
        float tFar = pmin(pmin(pmax(a0, a1), pmax(b0, b1)), pmax(c0, c1));
        bool isCube = greaterEqualF(tFar, tNear) && greaterEqualF(tFar, 0.0f);
        float nr = isCube ? pmin(tNear, tFar) : INF;
        float fr = isCube ? pmax(tNear, tFar) : INF;
        rootNear = nr;
        rootD = isCube ? (lessF(nr, 0.0f) ? fr : nr) : INF;
    }
    float toffset = pmax(rootNear, 0.f);
    v3 origined = mk3(torig.x + dirproj.x * toffset, torig.y + dirproj.y * toffset, torig.z + dirproj.z * toffset);
    v3 divident = mk3(1.f / dirproj.x, 1.f / dirproj.y, 1.f / dirproj.z);
    v3 norig = mk3(-origined.x * divident.x, -origined.y * divident.y, -origined.z * divident.z);

    bool validBox = alive && root >= 0 && lessF(rootD, INF) && lessF(rootD * dirlenInv, INF) && greaterEqualF(rootD, 0.0f);

    // hit state (TResult + bakedStack, :27-45)
    float predist = INF;
    int lastTri = -1;
    int bakedCount = 0;
    Baked head = {0.f, 0.f, INF, -1};
    Baked extra[BAKED_CAP - 1];

    const float IP = INF - PZERO;
    int cur = root;
    int sp = 0;
    int it = 0;
    for (; it < MAX_ITERS; it++) {
        if (!validBox) break;
        const uint4* pb = pairbox + 2 * (size_t)cur;
        uint4 lb = pb[0], rb = pb[1];
        int2 lk = link[cur];
        if (COUNT) nV++;
        Slab L = slab_child(origined, divident, norig, lb);
        Slab R = slab_child(origined, divident, norig, rb);
        bool leftNear = lessEqualF(L.near, R.near);  // :414
        bool ogL = (L.hit <= IP) && (L.hit * dirlenInv <= IP) && (L.hit > -PZERO) && (L.near <= IP) &&
                   (L.near * dirlenInv <= IP) && (((L.near + toffset) * dirlenInv - PZERO) <= predist);
        bool ogR = (R.hit <= IP) && (R.hit * dirlenInv <= IP) && (R.hit > -PZERO) && (R.near <= IP) &&
                   (R.near * dirlenInv <= IP) && (((R.near + toffset) * dirlenInv - PZERO) <= predist);
        bool leafL = ogL && lk.x < 0, leafR = ogR && lk.y < 0;
        bool intL = ogL && lk.x >= 0, intR = ogR && lk.y >= 0;

        if (leafL || leafR) {  // :441-448 -> testIntersectionPacked :261-309
            bool leftOrder = (leafL && leafR) ? leftNear : leafL;
            int triL = leafL ? ~lk.x : -1, triR = leafR ? ~lk.y : -1;
            int tx = leftOrder ? triL : triR, ty = leftOrder ? triR : triL;
            bool vx = leftOrder ? leafL : leafR, vy = leftOrder ? leafR : leafL;
            bool validx = (tx >= 0) && (tx != lastTri) && vx;
            bool validy = (ty >= 0) && (ty != lastTri) && vy && (tx != ty);
            if (!validx) {
                int t = tx; tx = ty; ty = t;
                bool q = validx; validx = validy; validy = q;
            }
#pragma unroll
            for (int pass = 0; pass < 2; pass++) {
                int tri = pass == 0 ? tx : ty;
                bool valid = pass == 0 ? validx : validy;
                if (valid) {
                    float u = 0.f, v = 0.f;
                    float d = tri_test(tri48, tri, origin, direct, u, v);
                    if (COUNT) nT++;
                    bool near = lessF(d, INF) && lessEqualF(d, predist) && greaterEqualF(d, 0.0f);
                    if (near) {
                        if (!equalF(d, predist)) bakedCount = 0;
                        predist = d;
                        lastTri = tri;
                        int at = bakedCount++;
                        if (at == 0) { head.u = u; head.v = v; head.t = d; head.tri = tri; }
                        else if (at < BAKED_CAP) { extra[at - 1].u = u; extra[at - 1].v = v; extra[at - 1].t = d; extra[at - 1].tri = tri; }
                        else if (COUNT) nBakedDrop++;
                    }
                }
            }
        }
        bool descend = intL || intR;
        if (descend) {  // :451-462
            bool leftOrder = (intL && intR) ? leftNear : intL;
            int lr0 = intL ? lk.x : -1, lr1 = intR ? lk.y : -1;
            if (!leftOrder) { int t = lr0; lr0 = lr1; lr1 = t; }
            if (lr1 != -1 && lr0 != lr1) {
                if (sp < STACK_CAP) stack[sp++][tid] = lr1;
                else if (COUNT) nDrop++;
            }
            cur = lr0;
        } else {  // :467-476
            sp--;
            if (sp >= 0) cur = stack[sp][tid];
            else validBox = false;
        }
    }
    if (COUNT && it >= MAX_ITERS && validBox) nCap++;

    if (alive) {
        uint32_t count = 0, off = 0;
        if (bakedCount <= 1) {
            count = (uint32_t)bakedCount;  // reorderTriangles is the identity on 0/1 entries
        } else {
            // reorderTriangles, :74-112 (rare path: equal-distance chains)
            Baked bk[BAKED_CAP];
            int n = bakedCount > BAKED_CAP ? BAKED_CAP : bakedCount;
            bk[0] = head;
            for (int k = 1; k < n; k++) bk[k] = extra[k - 1];
            for (int iround = 1; iround < n; iround++) {
                for (int index = 0; index < n - iround; index++) {
                    Baked a = bk[index], b = bk[index + 1];
                    bool lessIdx = a.tri <= b.tri;
                    bool deeper = lessF(a.t, b.t);
                    if (lessIdx || deeper) { bk[index] = b; bk[index + 1] = a; }
                }
            }
            int clean = 0;
            for (int iround = 0; iround < BAKED_CAP; iround++) {
                if (iround >= n - 1) break;
                if (bk[iround + 1].tri != bk[iround].tri) bk[clean++] = bk[iround];
            }
            if (n > 0 && clean <= BAKED_CAP) bk[clean++] = bk[n - 1];
            head = bk[0];
            count = (uint32_t)clean;
            if (count > 1) {
                off = atomicAdd(&cnt[2], count - 1);
                if (off + (count - 1) > pool_cap) {
                    if (COUNT) atomicAdd(&ctr->chain_pool_drops, 1ull);
                    count = 1;
                } else {
                    for (uint32_t k = 1; k < count; k++)
                        pool[off + k - 1] = make_float4(bk[k].u, bk[k].v, bk[k].t, __int_as_float(bk[k].tri));
                }
            }
        }
        if (count == 0) hit0[i] = make_float4(0.f, 0.f, INF, __int_as_float(-1));
        else hit0[i] = make_float4(head.u, head.v, head.t, __int_as_float(head.tri));
        hitN[i] = count | (off << 4);
    }
    if (COUNT) {
        uint32_t v = wave_sum(nV), t = wave_sum(nT), d = wave_sum(nDrop), c = wave_sum(nCap), b = wave_sum(nBakedDrop);
        if (lane_id() == 0) {
            if (v) atomicAdd(&ctr->node_visits, (unsigned long long)v);
            if (t) atomicAdd(&ctr->tri_tests, (unsigned long long)t);
            if (d) atomicAdd(&ctr->stack_drops, (unsigned long long)d);
            if (c) atomicAdd(&ctr->iter_caps, (unsigned long long)c);
            if (b) atomicAdd(&ctr->baked_drops, (unsigned long long)b);
        }
    }
}

int launch_rt_traverse(psm_rt* r, psm_bvh* b) {
    psm_ctx* c = r->ctx;
    uint32_t n = r->ray_count;
    if (n == 0) return PSM_OK;
    uint32_t grid = (n + TRAV_BLOCK - 1) / TRAV_BLOCK;
    TimedScope ts(c, CAT_TRAVERSE);
    if (c->counting)
        rt_traverse<true><<<grid, TRAV_BLOCK, 0, c->stream>>>(r->qA[r->cur], r->qB[r->cur], n, b->d_pairbox, b->d_link,
                                                               b->d_tri48, b->d_small, r->hit0, r->hitN, r->pool,
                                                               r->pool_cap, r->d_cnt, c->d_counters);
    else
        rt_traverse<false><<<grid, TRAV_BLOCK, 0, c->stream>>>(r->qA[r->cur], r->qB[r->cur], n, b->d_pairbox, b->d_link,
                                                                b->d_tri48, b->d_small, r->hit0, r->hitN, r->pool,
                                                                r->pool_cap, r->d_cnt, c->d_counters);
    PSM_HIP(c, hipGetLastError());
    c->rays_traced += n;
    return PSM_OK;
}

}  // namespace psm
