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
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "psm_common.h"
#include "psm_internal.h"

namespace psm {

constexpr int SM_M = 0;
constexpr int SM_ROOT = 25;

#ifndef PSM_EXP_WAVELOG
#define PSM_EXP_WAVELOG 0   // 1: an experiment build whose traversal waves log when they ended and how many wave-steps they spent at 1, 2, 3-4, 5-8, 9-16, 17+ lanes with work
#endif
#ifndef PSM_TRAV_BLOCK
#define PSM_TRAV_BLOCK 64
#endif
// one wave per workgroup: its slot (registers, 4 KB of stack) is free the moment the wave ends instead of when its partner
// does -- equal with frames in flight, 1.7 % faster for a frame alone (3.41 -> 3.35 ms); 256: 2 % slower (round 2)
constexpr int TRAV_BLOCK = PSM_TRAV_BLOCK;

// workgroups are dealt to the XCDs in runs of XCD_RUN (a power of two; the grid is a multiple of 8 runs), rt_traverse
#ifndef PSM_XCD_GROUP
#define PSM_XCD_GROUP 64
#endif
constexpr uint32_t XCD_RUN = PSM_XCD_GROUP;

// (leaf tests run once half of a wave's lanes with work wait for one: a third is equal, two thirds and more lose 2-18 %,
// profiles/r03_park_trigger.txt)

struct Slab {
    float hit, near;
    bool cube;
};

// intersectCubeDual, include/mathlib.glsl:129-193, fp32 branch, one child. The reference returns near = far = INFINITY
// for a missed box (:186-187), which then fails `hit <= INFINITY - PZERO` (directTraverse.comp:425): here the miss is
// the flag `cube` and near / hit are the raw values, which is the same decision with two selects fewer.
PSM_D Slab slab_child(v3 dr, v3 norig, float mnx, float mny, float mnz, float mxx, float mxy, float mxz) {
    float tminx = fmaf(mnx, dr.x, norig.x), tmaxx = fmaf(mxx, dr.x, norig.x);
    float tminy = fmaf(mny, dr.y, norig.y), tmaxy = fmaf(mxy, dr.y, norig.y);
    float tminz = fmaf(mnz, dr.z, norig.z), tmaxz = fmaf(mxz, dr.z, norig.z);
    float tNear = smaxf(smaxf(sminf(tminx, tmaxx), sminf(tminy, tmaxy)), sminf(tminz, tmaxz));
    float tFar = sminf(sminf(smaxf(tminx, tmaxx), smaxf(tminy, tmaxy)), smaxf(tminz, tmaxz));
    float tfp = tFar + PZERO;
    Slab s;
    // the reference's two tests, tfp >= tNear and tfp >= 0 (:189), as one against the larger of the two: maxNum drops a NaN
    // tNear, but tNear is NaN only when all six plane distances are, and then tFar and tfp are NaN too and the test fails
    // either way (one compare and one mask AND less per box)
    s.cube = tfp >= smaxf(tNear, 0.0f);
    s.near = sminf(tNear, tFar);
    float far = smaxf(tNear, tFar);
    // the reference's `near + PZERO <= 0` (:192): a sum of two floats rounds to zero only when it is zero, and its sign is
    // the exact sum's, so the test is `near <= -PZERO` for every near (NaN: false either way) -- one add less per box
    s.hit = (s.near <= -PZERO) ? far : s.near;
    return s;
}

// accepted child, directTraverse.comp:416-430. Of the reference's six tests two are implied by the others for every
// input: near <= hit (hit is near or far), so `near <= INF - PZERO` follows from `hit <= INF - PZERO`; and dirlenInv is
// in [0, 1e6] or NaN, so where `hit * dirlenInv <= INF - PZERO` holds, `near * dirlenInv <= INF - PZERO` can only fail
// for near = -inf with dirlenInv = 0, where the predist test is NaN <= predist = false as well.
// The two tests on hit that remain, `hit <= IP` and `hit * dirlenInv <= IP` (IP = INF - PZERO), are one compare against a
// per-ray constant (hit_limit below): next to `hit > -PZERO` only hits in (-PZERO, +inf] matter; a rounded product with a
// factor >= 0 is monotone in hit, so the hits that pass both tests are exactly those up to the largest one that does.
// (bitwise & on purpose: the operands are computations, && would branch around them -- measured: +7 VALU, +2 branches)
PSM_D bool child_ok(const Slab& c, float hitMax, float dirlenInv, float toffset, float predist) {
    return c.cube & (c.hit <= hitMax) & (c.hit > -PZERO) & (((c.near + toffset) * dirlenInv - PZERO) <= predist);
}

// The largest h with `h <= IP && h * dirlenInv <= IP` (float arithmetic, as directTraverse.comp:425-426 evaluates them).
// dirlenInv <= 1: the product of a hit >= 0 is at most the hit, so IP itself (a hit in (-PZERO, 0) passes both tests and
// the limit alike). dirlenInv > 1: the quotient, moved by the ulp or two that the two roundings can be off -- the loops
// establish the definition, the quotient only makes them short. NaN: NaN, every compare fails as `hit * NaN <= IP` does.
PSM_D float hit_limit(float dirlenInv) {
    const float IP = INF - PZERO;
    if (!(dirlenInv > 1.0f)) return dirlenInv == dirlenInv ? IP : dirlenInv;
    float t = IP / dirlenInv;   // in (IP * 1e-6, IP): positive and normal, its neighbours are its bit pattern +- 1
    while (t * dirlenInv > IP) t = __uint_as_float(__float_as_uint(t) - 1u);
    for (;;) {
        const float up = __uint_as_float(__float_as_uint(t) + 1u);
        if (!(up * dirlenInv <= IP)) break;
        t = up;
    }
    return t;
}

// intersectTriangle, include/vertex.glsl:140-189 (e1, e2 precomputed by bvh_prepare_tris)
PSM_D float tri_test(const float4* __restrict__ tri48, int tri, v3 orig, v3 dir, float& U, float& V) {
    // (cached: with the non-temporal hint the frame takes 3.03 ms instead of 2.40 -- the 12.6 MB of C3's triangle records live in L2)
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

// The wave's lane mask of a flag (the flag stays a flag: __ballot(int) would first widen it to an integer).
PSM_D unsigned long long lane_mask(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// ---- SOLO: the whole wave walks ONE ray ------------------------------------------------------------------------------
//
// A round's tail is made of waves in which a single ray is left (the per-ray step counts have a long thin tail: mean 56,
// p99 124, maximum 659-1099 on C3's bounce rounds, and two such rays rarely share a wave): the wave then issues the ~140
// instructions of a node step -- and every vector-compare -> scalar-AND -> vector-select hop of its lane-mask logic --
// for one live lane, ~2450 cycles per step for a wave alone on its SIMD (profiles/r03_lone_wave.txt), and a frame on its
// own, a tile's launches and every hand-over round's last launch wait for exactly these chains (4 x 0.68 ms of a 3.35 ms
// frame). When at most `solo_max` lanes of a wave have work left (and the launch does not hand rays over), the wave
// changes gear: it takes the rays one after the other, and for each one
//   * the ray's constants and state are broadcast (v_readlane) -- wave-uniform from here on, so every decision of the
//     step (accepted? leaf? nearer child? push, pop, done?) is SCALAR arithmetic and a scalar branch: no lane masks;
//   * lane j of each group of eight loads ONE word of the 32-byte node record and takes a second one from a neighbour by a
//     quad permute: lanes 0,1,2 hold the left child's x,y,z slabs, lanes 4,5,6 the right child's, lanes 3 and 7 the links; a lane
//     evaluates its two plane distances (the same v_fma_mix_f32 on the same operands), tNear / tFar are two quad-permute
//     DPP v_max / v_min each, IN THE ORDER slab_child combines them (lane 0: ((x, y), z)), the acceptance tests run in lanes
//     0 and 4 at once and reach the scalar unit as two bits of one ballot;
//   * the stack lives in the lanes of one VGPR (v_writelane / v_readlane at the scalar depth): no LDS round trip in a pop;
//   * a leaf's two triangles are tested side by side in lanes 0 and 1, and accepted one after the other in the
//     reference's order (directTraverse.comp:261-309) by scalar code; only the ray's own lane touches its hit chain.
// Per ray the node steps and triangle tests, their operands and their order are those of the lane-per-ray step above
// (hits, chains, V and T bit-exact: every traversal test runs through this body, test_every_traversal_schedule_is_bit_exact
// with solo_max 0..4); the step is ~25 vector + ~30 scalar instructions and one memory round trip.
PSM_D float rl_f(float x, int lane) { return u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(x), lane)); }
PSM_D int rl_i(int x, int lane) { return __builtin_amdgcn_readlane(x, lane); }
template <int P0, int P1, int P2, int P3>
PSM_D float quad_perm(float x) {   // lane i of every quad reads lane P_i of its quad
    return u2f((uint32_t)__builtin_amdgcn_mov_dpp((int)f2u(x), P0 | (P1 << 2) | (P2 << 4) | (P3 << 6), 0xF, 0xF, true));
}

struct SoloCounters {
    uint32_t nV, nT, nDrop, nCap, nBakedDrop;
};

// Runs the ray of lane `L` (wave-uniform) to its end. Out, in lane L only: the hit chain (head, extra, bakedCount); the
// caller marks the lane done.
// The ray travels from its lane to the wave through 23 words of LDS (`xch`, written by solo_park): the floats come back as
// vector registers that hold the same value in every lane (operands of the plane and triangle arithmetic), the integers as
// scalars (v_readfirstlane) -- the kernel has 78 scalar registers, the ray's thirteen constants do not fit beside the loop's.
constexpr int XCH_WORDS = 32;
constexpr int SOLO_MAX = 4;     // rays a wave takes into the solo gear at most (LDS: SOLO_MAX blocks of XCH_WORDS words per wave)
// the lane-per-ray loop's variables of one ray, parked in LDS for solo_ray
PSM_D void solo_park(uint32_t* xch, const v3 origin, const v3 direct, const v3 divident, const v3 norig, const float dirlenInv, const float hitMax,
                     const float toffset, const float predist, const int lastTri, const int cur, const int sp, const int it, const int pl,
                     const int pr, const bool pLeftNear, const bool parked) {
    xch[0] = f2u(origin.x); xch[1] = f2u(origin.y); xch[2] = f2u(origin.z); xch[3] = f2u(dirlenInv);
    xch[4] = f2u(direct.x); xch[5] = f2u(direct.y); xch[6] = f2u(direct.z); xch[7] = f2u(hitMax);
    xch[8] = f2u(divident.x); xch[9] = f2u(divident.y); xch[10] = f2u(divident.z); xch[11] = f2u(toffset);
    xch[12] = f2u(norig.x); xch[13] = f2u(norig.y); xch[14] = f2u(norig.z); xch[15] = f2u(predist);
    xch[16] = (uint32_t)lastTri; xch[17] = (uint32_t)cur; xch[18] = (uint32_t)sp; xch[19] = (uint32_t)it;
    xch[20] = (uint32_t)pl; xch[21] = (uint32_t)pr; xch[22] = (pLeftNear ? 1u : 0u) | (parked ? 2u : 0u);
}
template <bool COUNT>
PSM_D void solo_ray(const int L, const uint32_t* __restrict__ node_dw, const float4* __restrict__ tri48, const int* lds_stack_col, const uint32_t* xch,
                    const int lj, int& bakedCount, Baked& head, Baked* extra, SoloCounters& ctr, unsigned long long& wave_steps) {
    const bool mine = lj == L;
    const int j8 = lj & 7, q = lj & 3;
    const v3 O = mk3(u2f(xch[0]), u2f(xch[1]), u2f(xch[2]));
    const v3 D = mk3(u2f(xch[4]), u2f(xch[5]), u2f(xch[6]));
    const float dirlenInv = u2f(xch[3]), hitMax = u2f(xch[7]), toffset = u2f(xch[11]);
    float predist = u2f(xch[15]);
    // this lane's axis of the slab test (lanes 3 and 7 of a group of eight hold the links: any axis)
    const int ax = q < 3 ? q : 2;
    const float my_dr = u2f(xch[8 + ax]), my_no = u2f(xch[12 + ax]);
    int lastTri = __builtin_amdgcn_readfirstlane((int)xch[16]), cur = __builtin_amdgcn_readfirstlane((int)xch[17]);
    int sp = __builtin_amdgcn_readfirstlane((int)xch[18]), it = __builtin_amdgcn_readfirstlane((int)xch[19]);
    const int pl = __builtin_amdgcn_readfirstlane((int)xch[20]), pr = __builtin_amdgcn_readfirstlane((int)xch[21]);
    const int fl = __builtin_amdgcn_readfirstlane((int)xch[22]);
    const bool pLeftNear = (fl & 1) != 0, parked = (fl & 2) != 0;
    // this lane's part of a node record. A child's box is six halfs in three words, [mn.x mn.y] [mn.z mx.x] [mx.y mx.z]; of a
    // group of eight lanes, lanes 0,1,2 take the left child's x,y,z slabs, lanes 4,5,6 the right child's, lanes 3 and 7 the
    // two links. Lane x loads word 0, lane y word 2, lane z word 1: every slab lane then finds one of its two planes in the LOW
    // half of its own word (mn.x, mx.y, mn.z) and the other one in the HIGH half of the word its neighbour z, x, y loaded
    // (mx.x, mn.y, mx.z) -- one load and one quad permute per lane, and which of the two is the min plane does not matter:
    // the slab test takes min and max of the two distances (mathlib.glsl:152-157).
    const uint32_t child = (uint32_t)(j8 >> 2);
    const uint32_t wsel4 = 4u * (q == 3 ? 6u + child : child * 3u + (q == 0 ? 0u : (q == 1 ? 2u : 1u)));   // byte offset of this lane's word
    // the ray's stack: entry k in lane k
    int stk = lj < sp ? lds_stack_col[(lj < STACK_CAP ? lj : 0) * TRAV_BLOCK + L] : 0;
    // the node step leaves its accepted leaves as (leafL, leafR, lkx, lky, leftNear); the links are decoded here, where one
    // step in eight needs them
    uint32_t leafL = pl < 0 ? 1u : 0u, leafR = pr < 0 ? 1u : 0u, leftNear = pLeftNear ? 1u : 0u;
    int lkx = pl, lky = pr;
    (void)parked;
    for (;;) {
        if ((leafL | leafR) != 0u) {  // testIntersectionPacked, :261-309
            const int pl = leafL ? lkx : 0, pr = leafR ? lky : 0;
            const bool pLeftNear = leftNear != 0u;
            const bool lo = (pl < 0) && (pLeftNear || pr >= 0);
            const int tx = ~(lo ? pl : pr), ty = ~(lo ? pr : pl);
            const bool validx = (tx >= 0) && (tx != lastTri);
            const bool validy = (ty >= 0) && (ty != lastTri) && (tx != ty);
            const int triA = validx ? tx : ty;
            const bool valid = validx || validy, again = validx && validy;
            if (valid) {
                float u = 0.f, v = 0.f, d = INF;
                if (lj == 0 || (lj == 1 && again)) d = tri_test(tri48, lj == 0 ? triA : ty, O, D, u, v);
#pragma unroll
                for (int pass = 0; pass < 2; pass++) {
                    if (pass == 1 && !again) break;
                    const float sd = rl_f(d, pass), su = rl_f(u, pass), sv = rl_f(v, pass);
                    const int stri = pass == 0 ? triA : ty;
                    if (COUNT) ctr.nT += mine ? 1u : 0u;
                    // (predist is a vector register with one value in all lanes: the decisions reach the scalar unit as lane masks)
                    const bool near = lane_mask(lessF(sd, INF) && lessEqualF(sd, predist) && greaterEqualF(sd, 0.0f)) != 0ull;
                    if (near) {
                        const bool fresh = !equalF(sd, predist);
                        predist = sd;
                        lastTri = stri;
                        if (mine) {
                            if (fresh) bakedCount = 0;
                            int at = bakedCount++;
                            if (at == 0) { head.u = su; head.v = sv; head.t = sd; head.tri = stri; }
                            else if (at < BAKED_CAP) { extra[at - 1].u = su; extra[at - 1].v = sv; extra[at - 1].t = sd; extra[at - 1].tri = stri; }
                            else if (COUNT) ctr.nBakedDrop++;
                        }
                    }
                }
            }
            leafL = 0u; leafR = 0u;
        }
        if (sp < 0) break;
        it++;
        const bool lastIter = it >= MAX_ITERS;  // :383
        // (a 32-bit byte offset on the scalar base: the node array is < 4 GiB by the 2^27-triangle cap)
        const uint32_t wd = *(const uint32_t*)((const char*)node_dw + (((uint32_t)cur << 5) | wsel4));
        const uint32_t wn = (uint32_t)__builtin_amdgcn_mov_dpp((int)wd, 2 | (0 << 2) | (1 << 4) | (3 << 6), 0xF, 0xF, true);   // x <- z, y <- x, z <- y
        if (COUNT) { ctr.nV += mine ? 1u : 0u; wave_steps++; }
        // intersectCubeDual, mathlib.glsl:129-193: slab_child with one axis of one child per lane
        const float tmin = fmaf(half_lo(wd), my_dr, my_no), tmax = fmaf(half_hi(wn), my_dr, my_no);   // (or tmax, tmin: see above)
        const float lo_ = sminf(tmin, tmax), hi_ = smaxf(tmin, tmax);
        const float tNear = smaxf(smaxf(lo_, quad_perm<1, 2, 0, 3>(lo_)), quad_perm<2, 0, 1, 3>(lo_));   // lane 0 / 4: ((x, y), z)
        const float tFar = sminf(sminf(hi_, quad_perm<1, 2, 0, 3>(hi_)), quad_perm<2, 0, 1, 3>(hi_));
        const float near = sminf(tNear, tFar);
        const float hit = (near <= -PZERO) ? smaxf(tNear, tFar) : near;
        lkx = rl_i((int)wd, 3); lky = rl_i((int)wd, 7);
        // child_ok (directTraverse.comp:416-430 as rt_traverse tests it), its four tests as four lane masks ANDed in the scalar unit
        // (as one boolean expression they come back through a 0 / 1 vector register: two instructions more)
        const uint32_t okm = (uint32_t)(lane_mask((tFar + PZERO) >= smaxf(tNear, 0.0f)) & lane_mask(hit <= hitMax) & lane_mask(hit > -PZERO) &
                                        lane_mask((((near + toffset) * dirlenInv) - PZERO) <= predist));
        // :414 in lane 0: lessEqualF(L.near, R.near) = (R.near - L.near) > -PZERO, the right child's value fetched from lane 4
        // by the subtraction itself (DPP row_shl:4)
        const float Rnear = u2f((uint32_t)__builtin_amdgcn_mov_dpp((int)f2u(near), 0x104, 0xF, 0xF, true));
        leftNear = (uint32_t)lane_mask((Rnear - near) > -PZERO) & 1u;
        // from here on the step of rt_traverse in scalar registers (flags as 0 / 1 integers: one s_and / s_xor each)
        const uint32_t ogL = okm & 1u, ogR = (okm >> 4) & 1u;
        leafL = ogL & ((uint32_t)lkx >> 31); leafR = ogR & ((uint32_t)lky >> 31);   // :441-448: accepted leaves wait for the block above
        const uint32_t intL = ogL ^ leafL, intR = ogR ^ leafR;   // accepted and not a leaf
        const uint32_t leftFirst = intL & (leftNear | (intR ^ 1u));  // :451-462
        const int first = leftFirst ? lkx : lky, second = leftFirst ? lky : lkx;
        if ((intL & intR) != 0u && lkx != lky) {
            if (sp < STACK_CAP) { stk = lj == sp ? second : stk; sp++; }   // (v_writelane_b32 has no builtin in this clang: one compare + select)
            else if (COUNT) ctr.nDrop += mine ? 1u : 0u;
        }
        cur = first;
        if ((intL | intR) == 0u) {    // :467-476
            sp--;
            if (sp >= 0) cur = rl_i(stk, sp);
        }
        if (COUNT) ctr.nCap += (sp >= 0 && lastIter && mine) ? 1u : 0u;
        sp = lastIter ? -1 : sp;
    }
}

// CHAIN: a further intersection() over the same rays with another hierarchy (multi-BVH). The
// search starts from the distance of the chain the ray already carries (traverse(), :335-346) and the hits
// it bakes overwrite the front of that chain, the rest of the old chain staying linked behind them
// (includeChain, :219-249). Triangle ids are tagged with the object's sequence number (bits 27..30).
// PHASED: wave64 ray compaction. A wave runs in lock step, so it costs as much as its slowest ray (mean 56
// steps per bounce ray, mean per-wave maximum 118): most lanes idle behind a few long rays. A PHASED launch
// lets a wave hand over: when fewer than `min_live` of its lanes still have work (__ballot / popcount, wave
// uniform), or after `cap` wave-steps, the rays that are not done write their traversal state (node, stack,
// best hit) to a dense continuation queue -- one atomic per wave, lanes ranked by __ballot -- and the next
// launch resumes them packed 64 to a wave. Resume launches are persistent: a fixed grid of waves strides over
// the continuation queue, 64 rays at a time. Per ray nothing changes -- the same node steps and triangle
// tests in the same order -- so hits, chains and counters are bit-exact; only the idle lanes go.
// All arguments of the kernel in one struct. The hot loop needs a few of them (queue, node and triangle arrays);
// the rest -- result arrays, counters, the hand-over state -- are read where they are used, through the kernarg
// segment, behind an opaque move (cold_args): held in scalar registers from the kernel's entry they would cost the
// loop ~30 SGPRs, whose spills take two VGPRs from a kernel that is allowed 64 (8 waves per SIMD).
constexpr size_t MAX_PHASES = 16;   // launches of a hand-over round at most; two sets of continuation counts alternate between rounds

struct TravArgs {
    const float4 *qA, *qB;        // hot
    const uint32_t* qbases;
    uint32_t qnb, nrays;
    const uint4* node32;
    const float4* tri48;
    const uint32_t* sm;
    uint32_t cap, min_live, min_steps, final_rays;  // Phase, hot part
    uint32_t solo_max;            // a wave with at most this many rays left walks them one by one (solo_ray); 0: never
#if PSM_EXP_WAVELOG
    uint32_t* wavelog;            // experiment build (tests/studies/wave_log.py): 8 words per wave of a fresh-ray launch
#endif
    const uint32_t* in_count;
    float4* hit0;                 // cold from here on
    uint32_t* hitN;
    float4* pool;
    uint32_t* cnt;
    DevCounters* ctr;
    uint32_t pool_cap, obj_tag;
    TravState in, out;
    uint32_t* out_count;
    uint32_t* zero_cnt;           // the last launch of a hand-over round clears the OTHER set of continuation counts (next round's)
};

PSM_D const TravArgs* cold_args() {
    unsigned long long zero;  // an opaque 0: whatever is loaded through the sum cannot be hoisted above this point
    asm volatile("s_mov_b64 %0, 0" : "=s"(zero));
    return (const TravArgs*)((const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + zero);
}

template <bool COUNT, bool CHAIN, bool PHASED>
__global__ __launch_bounds__(TRAV_BLOCK, 8) void rt_traverse(TravArgs ka) {
    const float4* __restrict__ qA = ka.qA;
    const float4* __restrict__ qB = ka.qB;
    const uint32_t* __restrict__ qbases = ka.qbases;
    const uint32_t qnb = ka.qnb;
    // never past what the queue holds: a host-supplied count (psm_rt_set_ray_count) larger than the queue's total would
    // resolve to slots beyond its last segment
    const uint32_t nrays = min(ka.nrays, qbases[qnb]);
    const uint4* __restrict__ node32 = ka.node32;
    const float4* __restrict__ tri48 = ka.tri48;
    const uint32_t* __restrict__ sm = ka.sm;
    struct { uint32_t cap, min_live, min_steps, final_rays; const uint32_t* in_count; } ph = {ka.cap, ka.min_live, ka.min_steps, ka.final_rays, ka.in_count};
    __shared__ int stack[STACK_CAP][TRAV_BLOCK];
    __shared__ uint32_t xch[TRAV_BLOCK / 64][SOLO_MAX][XCH_WORDS];   // solo_ray: rays on their way from their lanes to the wave
    const int tid = threadIdx.x;
    __builtin_assume(tid >= 0 && tid < TRAV_BLOCK);   // (the launch bound does not tell the optimiser: with one wave per workgroup `tid & ~63` is 0 and `tid >> 6` is 0 -- a register and a spill less)
    const bool resume = PHASED && ph.in_count != nullptr;
    const uint32_t total = resume ? *ph.in_count : nrays;
    // with few rays left there is nothing to pack them with: the launch finishes them
    const uint32_t min_live = (PHASED && !(resume && total <= ph.final_rays)) ? ph.min_live : 0u;
    const uint32_t cap = (PHASED && !(resume && total <= ph.final_rays)) ? ph.cap : 0xFFFFFFFFu;
    const int capI = (int)min(cap, 0x7FFFFFFFu), minLive1 = (int)min_live - 1, minSteps = (int)min(ph.min_steps, 0x7FFFFFFFu);
    // (a launch that hands rays over does so below min_live lanes: the solo gear is for launches that finish their rays)
    const int soloMax = (PHASED && (min_live != 0u || cap != 0xFFFFFFFFu)) ? 0 : (int)min(ka.solo_max, (uint32_t)SOLO_MAX);   // 0: the loop ends with nobody left
    uint32_t nV = 0, nT = 0, nDrop = 0, nCap = 0, nBakedDrop = 0;
    unsigned long long dg_t0 = 0, dg_r0 = 0, dg_steps = 0;
    if (COUNT) { dg_t0 = __builtin_amdgcn_s_memtime(); dg_r0 = __builtin_amdgcn_s_memrealtime(); }
#if PSM_EXP_WAVELOG
    uint32_t wl_steps[6] = {0, 0, 0, 0, 0, 0};
    const unsigned long long wl_t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long wl_solo_before = 0;
#endif
    if (PHASED && blockIdx.x == 0 && tid < (int)MAX_PHASES) {   // (instead of a memset launch in front of every round: 40 us on a lane's critical path with frames in flight)
        uint32_t* z = cold_args()->zero_cnt;
        if (z) z[tid] = 0u;
    }
    const int root = (int)sm[SM_ROOT];
    // fresh rays: one ray per thread of the grid. Workgroups b and b + 8 of a grid share an XCD (round-robin dispatch,
    // observed, a speed matter only) and each XCD has its own 4 MB L2: the grid is dealt so that an XCD walks runs of
    // XCD_RUN consecutive workgroups' rays (4096 rays: two rows of texels, neighbouring parts of the tree), the runs
    // themselves round-robin. One contiguous eighth of the queue per XCD was 6 % slower (C5: 28 %): the expensive
    // part of the image lands on one XCD. Runs of 4096 rays are 1-2 % faster than plain round-robin serial, equal in
    // flight; 1024 rays: 4 % slower in flight, 16 k: 1 %, 64 k: 5 % (profiles/r03_cache_policy_ab.txt: x8, x128, x512
    // of 128-thread workgroups). Resume: every wave strides over the continuation queue.
    constexpr uint32_t XCD_GROUP = XCD_RUN;
    const uint32_t bq = blockIdx.x >> 3;   // position within the XCD's sequence
    const uint32_t vb_ = ((bq / XCD_GROUP) * 8u + (blockIdx.x & 7u)) * XCD_GROUP + (bq % XCD_GROUP);
    const uint32_t vblock = resume ? blockIdx.x : vb_;
    for (uint32_t batch = vblock * TRAV_BLOCK + (uint32_t)(tid & ~63); batch < total; batch += gridDim.x * TRAV_BLOCK) {
    const uint32_t slot = batch + (uint32_t)(tid & 63);
    bool alive = slot < total;
    uint32_t i = slot;
    if (resume) i = alive ? cold_args()->in.idx()[slot] : 0u;

    float4 A = make_float4(0, 0, 0, 0), B = make_float4(1, 0, 0, 0);
    float M[16];
    {
        const uint32_t loc = alive ? queue_loc(qbases, qnb, nrays, i) : 0u;  // the queue is segmented (psm_common.h)
        if (alive) { A = ld_stream(&qA[loc]); B = ld_stream(&qB[loc]); }
#pragma unroll
        for (int k = 0; k < 16; k++) M[k] = u2f(sm[SM_M + k]);
    }
    v3 origin = mk3(A.x, A.y, A.z);
    v3 direct = normalize3(mk3(B.x, B.y, B.z));  // :350

    float to4[4], td4[4];
    mat_vec(M, origin.x, origin.y, origin.z, 1.0f, to4);   // :353
    matT_vec(M, direct.x, direct.y, direct.z, 1.0f, td4);  // :354
    v3 torig = mk3(to4[0], to4[1], to4[2]);
    v3 tdir = mk3(td4[0], td4[1], td4[2]);
    float dirlen = len3(tdir) / pmax(len3(direct), 0.000001f);
    float dirlenInv = 1.f / pmax(dirlen, 0.000001f);
    const float hitMax = hit_limit(dirlenInv);
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
        float tNear = smaxf(smaxf(sminf(a0, a1), sminf(b0, b1)), sminf(c0, c1));
        float tFar = sminf(sminf(smaxf(a0, a1), smaxf(b0, b1)), smaxf(c0, c1));
        bool isCube = greaterEqualF(tFar, tNear) && greaterEqualF(tFar, 0.0f);
        float nr = isCube ? sminf(tNear, tFar) : INF;
        float fr = isCube ? smaxf(tNear, tFar) : INF;
        rootNear = nr;
        rootD = isCube ? (lessF(nr, 0.0f) ? fr : nr) : INF;
    }
    float toffset = smaxf(rootNear, 0.f);
    v3 origined = mk3(torig.x + dirproj.x * toffset, torig.y + dirproj.y * toffset, torig.z + dirproj.z * toffset);
    v3 divident = mk3(1.f / dirproj.x, 1.f / dirproj.y, 1.f / dirproj.z);
    v3 norig = mk3(-origined.x * divident.x, -origined.y * divident.y, -origined.z * divident.z);

    const bool validRay = alive && root >= 0 && lessF(rootD, INF) && lessF(rootD * dirlenInv, INF) && greaterEqualF(rootD, 0.0f);

    // hit state (TResult + bakedStack, :27-45)
    float predist = INF;
    if (CHAIN) {
        const TravArgs* K = cold_args();
        if (alive && (K->hitN[i] & 15u) != 0u) predist = K->hit0[i].z;
    }
    int lastTri = -1;
    int bakedCount = 0;
    Baked head = {0.f, 0.f, INF, -1};
    Baked extra[BAKED_CAP - 1];

    // sp: entries on the ray's stack, -1 once the ray has no node step left (the flag lives in the register the step
    // updates anyway: carried as a lane mask of its own it cost the scalar unit a three-instruction merge per step)
    int cur = root;
    int sp = validRay ? 0 : -1;
    int it = 0;
    if (resume && alive) {  // pick the ray up where the previous launch left it
        const TravState in = cold_args()->in;
        uint32_t m = ld_stream(&in.misc()[slot]);
        cur = (int)ld_stream((const uint32_t*)&in.cur()[slot]);
        sp = (int)(m & 255u);
        it = (int)((m >> 8) & 0xFFFFu);
        bakedCount = (int)(m >> 24);
        predist = __uint_as_float(ld_stream((const uint32_t*)&in.predist()[slot]));
        lastTri = (int)ld_stream((const uint32_t*)&in.lastTri()[slot]);
        float4 hd = ld_stream(&in.head()[slot]);
        head.u = hd.x; head.v = hd.y; head.t = hd.z; head.tri = __float_as_int(hd.w);
#pragma unroll 4
        for (int k = 0; k < sp; k++) stack[k][tid] = (int)ld_stream((const uint32_t*)&in.stack()[(size_t)k * in.capacity + slot]);
    }
    int wsteps = 0;
    // handed over: its result is written by the launch that finishes the ray
    bool suspendedFlag = false;
    // Leaf tests are deferred, not reordered: a lane that reaches a leaf parks its triangle pair (pl, pr) and
    // sits out the node steps of the others until enough lanes of the wave are parked (or nobody can step),
    // then all of them run the triangle block together. Per ray the sequence of node steps and triangle
    // tests is exactly the reference's (:441-476) -- a parked lane does nothing in between -- but the wave
    // issues the ~150-instruction triangle block for ~half its live lanes instead of for 3-4 of them.
    int pl = 0, pr = 0;  // parked leaf links as stored in the node record (~triangle id: negative), 0 = none
    int pLeftNear = 0;   // (an integer in a vector register, written under the step's own exec mask: as a flag it was a lane mask merged by the scalar unit at every step)
    // (a lane waits for its leaf tests iff pl | pr < 0: read from the two links where it is needed -- as a flag of its own it was a
    // lane mask the scalar unit merged at every step; with sp: (sp | pl | pr) >= 0 says "a node step is due" in one compare)
    bool stepping = sp >= 0; // a node step is due: sp >= 0 && !parkedNow (assigned for the whole wave at once, no merge)
    for (;;) {
        if (stepping) {
            {
                it++;
                const bool lastIter = it >= MAX_ITERS;  // :383, see below
                // 32-bit byte offset (the node array is < 4 GiB: 2^27 nodes): one shift, the load adds it to the scalar base
                const uint4* np = (const uint4*)((const char*)node32 + ((uint32_t)cur << 5));
                uint4 n0 = np[0], n1 = np[1];
                int2 lk = make_int2((int)n1.z, (int)n1.w);
                if (COUNT) nV++;
                Slab L = slab_child(divident, norig, half_lo(n0.x), half_hi(n0.x), half_lo(n0.y), half_hi(n0.y), half_lo(n0.z), half_hi(n0.z));
                Slab R = slab_child(divident, norig, half_lo(n0.w), half_hi(n0.w), half_lo(n1.x), half_hi(n1.x), half_lo(n1.y), half_hi(n1.y));
                // straight-line selects from here on (bitwise & on the flags: no short-circuit branches)
                const bool leftNear = lessEqualF(L.near, R.near);  // :414 (only read when both children are accepted)
                const bool ogL = child_ok(L, hitMax, dirlenInv, toffset, predist), ogR = child_ok(R, hitMax, dirlenInv, toffset, predist);
                const bool lfL = lk.x < 0, lfR = lk.y < 0;
                // the flags are lane masks: && / || on values already computed are mask ANDs / ORs (bitwise & | would
                // promote them to integers and materialise them in VGPRs)
                const bool leafL = ogL && lfL, leafR = ogR && lfR;
                pl = leafL ? lk.x : 0;   // accepted leaves, :441-448: tested below (kept undecoded: one select each)
                pr = leafR ? lk.y : 0;
                pLeftNear = leftNear ? 1 : 0;
                const bool intL = ogL != leafL, intR = ogR != leafR;  // accepted and not a leaf (one mask XOR, no second compare)
                const bool leftFirst = intL && (leftNear || !intR);  // :451-462: both ? leftNear : intL
                const int first = leftFirst ? lk.x : lk.y, second = leftFirst ? lk.y : lk.x;
                if (intL && intR && (lk.x != lk.y)) {
                    if (sp < STACK_CAP) stack[sp++][tid] = second;
                    else if (COUNT) nDrop++;
                }
                cur = first;              // the nearer accepted internal child (a dead value when there is none)
                if (!(intL || intR)) {    // :467-476
                    sp--;        // -1: the stack is empty, the ray has no node step left
                    if (sp >= 0) cur = stack[sp][tid];
                }
                // :383: the loop runs MAX_ITERS iterations at most; a ray that still has work after its last one stops here
                // (flagged with the step itself instead of in a branch of its own before the next one: its parked leaves,
                // if any, are still tested below, exactly as when the flag was raised one wave-step later)
                if (COUNT) nCap += (sp >= 0 && lastIter) ? 1u : 0u;
                sp = lastIter ? -1 : sp;
            }
        }
        // wave-uniform bookkeeping, in scalar registers: np lanes wait for a leaf test, nl lanes have work of any kind
        stepping = (sp | pl | pr) >= 0;
        const int np = __popcll(lane_mask((pl | pr) < 0));
        const int nl = np + __popcll(lane_mask(stepping));
        if (COUNT) dg_steps++;
#if PSM_EXP_WAVELOG
        wl_steps[nl <= 1 ? 0 : (nl <= 2 ? 1 : (nl <= 4 ? 2 : (nl <= 8 ? 3 : (nl <= 16 ? 4 : 5))))]++;
#endif
        // Keep stepping the others while fewer than half of the lanes with work wait for a leaf test (with nobody able to
        // step, np == nl: the tests run; with nobody left at all, nl == 0, the wave is done) -- unless the launch's cap is
        // reached or too few lanes have work left to be worth a wave (PHASED). Three differences whose signs are the three
        // conditions, so that the decision is one AND and one compare in the scalar unit instead of a chain of selects
        // (the step is bound by scalar issue as much as by vector issue: ~75 instructions of each).
        int handover = 0;   // >= 0 (PHASED): hand over after the parked tests
        if (PHASED) {
            wsteps++;
            const int thr1 = wsteps >= minSteps ? minLive1 : -1;   // nl > thr1  <=>  nl >= min_live once min_steps have run
            handover = (thr1 - nl) & (wsteps - capI);              // < 0: enough lanes live and below the cap
            if (((max(np << 1, soloMax) - nl) & handover) < 0) continue;
        } else if (nl > max(np << 1, soloMax)) continue;
        const bool capHit = PHASED && handover >= 0;
        // nobody left -- or at most solo_max rays and nobody to hand them to: the wave walks those one at a time, all lanes on
        // one ray (below the loop). One more difference in the AND above and this compare are all the node step pays for it.
        if (nl <= soloMax) break;
        if ((pl | pr) < 0) {  // testIntersectionPacked, :261-309
            // both leaves: the nearer one first (:441-448); otherwise the one that is a leaf (pl, pr are 0 when not)
            const bool lo = (pl < 0) && (pLeftNear != 0 || pr >= 0);
            const int tx = ~(lo ? pl : pr), ty = ~(lo ? pr : pl);  // triangle ids, -1 = none (~0)
            const bool validx = (tx >= 0) && (tx != lastTri);
            const bool validy = (ty >= 0) && (ty != lastTri) && (tx != ty);
            int tri = validx ? tx : ty;       // first test: x, or y straight away when x is skipped
            bool valid = validx || validy;
            bool again = validx && validy;    // second test: y after x
#pragma unroll 1
            for (int pass = 0; pass < 2; pass++) {
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
                tri = ty;
                valid = again;
                again = false;
            }
            pl = 0; pr = 0;
        }
        if (capHit) {
            // every parked test has just run. Rays with work left and a chain of at most one hit hand their state
            // to the next launch (a longer chain lives in registers / scratch: such a ray, < 0.1 %, finishes here)
            const bool susp = sp >= 0 && bakedCount <= 1;
            const unsigned long long sb = lane_mask(susp);
            if (sb != 0ull) {
                const TravArgs* K = cold_args();
                const TravState out = K->out;
                uint32_t base = 0;
                const int leader = __ffsll((long long)sb) - 1;
                if (lane_id() == leader) base = atomicAdd(K->out_count, (uint32_t)__popcll(sb));
                base = __shfl(base, leader);
                if (susp) {
                    const uint32_t o = base + (uint32_t)__popcll(sb & ((1ull << lane_id()) - 1ull));
                    st_stream(&out.idx()[o], i);
                    st_stream((uint32_t*)&out.cur()[o], (uint32_t)cur);
                    st_stream(&out.misc()[o], (uint32_t)sp | ((uint32_t)it << 8) | ((uint32_t)bakedCount << 24));
                    st_stream((uint32_t*)&out.predist()[o], __float_as_uint(predist));
                    st_stream((uint32_t*)&out.lastTri()[o], (uint32_t)lastTri);
                    st_stream(&out.head()[o], make_float4(head.u, head.v, head.t, __int_as_float(head.tri)));
                    for (int k = 0; k < sp; k++) st_stream((uint32_t*)&out.stack()[(size_t)k * out.capacity + o], (uint32_t)stack[k][tid]);
                    sp = -1;
                    suspendedFlag = true;
                }
            }
            if (lane_mask(sp >= 0) == 0ull) break;
        }
        stepping = sp >= 0;   // nobody is parked now
    }
    // The solo gear (solo_ray): the rays the loop above has left are walked one at a time, all lanes on one ray. (Behind the
    // loop, not in it: inside, its values would be live across the node steps, which have no register to spare.)
    {
        // (a lane waits for its leaf tests iff it holds a parked link: read from pl / pr, so that the loop's flag does not have to
        // be kept for its exits)
        const bool parkedL = pl < 0 || pr < 0;
        unsigned long long work = lane_mask(sp >= 0 || parkedL);   // none after a complete run or a hand-over
        if (work != 0ull) {
            // (the lane number behind an opaque zero, so that it is computed HERE: shared with an earlier lane_id() it -- and the lane
            // roles the solo gear derives from it -- would be held in registers across the node steps, which have none to spare)
            uint32_t zero;
            asm volatile("s_mov_b32 %0, 0" : "=s"(zero));
            const int lj = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, zero));
            // every ray first leaves its lane (after this the loop's per-lane variables are dead: the solo gear needs the registers)
            if (sp >= 0 || parkedL)
                solo_park(&xch[tid >> 6][__popcll(work & ((1ull << lj) - 1ull))][0], origin, direct, divident, norig, dirlenInv, hitMax, toffset, predist,
                          lastTri, cur, sp, it, pl, pr, pLeftNear != 0, parkedL);
            // one wave: its LDS operations execute in program order; the fences keep the compiler from moving the reads up
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            SoloCounters sc = {nV, nT, nDrop, nCap, nBakedDrop};
#if PSM_EXP_WAVELOG
            wl_solo_before = __builtin_amdgcn_s_memrealtime();
#endif
            // (s_setprio 3 for the wave while it is in the gear changed nothing: 2.185 / 3.167 / 0.603 ms against 2.186 / 3.167 / 0.603,
            // profiles/r04_solo_gear.txt -- the one-ray waves are not starved by the bulk's waves, they run when the chip has emptied)
            for (int k = 0; work != 0ull; k++) {
                const int Ls = __builtin_ctzll(work);
                work &= work - 1ull;
                solo_ray<COUNT>(Ls, (const uint32_t*)node32, tri48, &stack[0][tid & ~63], &xch[tid >> 6][k][0], lj, bakedCount, head, extra, sc, dg_steps);
            }
            if (COUNT) { nV = sc.nV; nT = sc.nT; nDrop = sc.nDrop; nCap = sc.nCap; nBakedDrop = sc.nBakedDrop; }
        }
    }
    if (PHASED && suspendedFlag) alive = false;  // handed over
    const TravArgs* K = cold_args();
    float4* __restrict__ hit0 = K->hit0;
    uint32_t* __restrict__ hitN = K->hitN;
    float4* __restrict__ pool = K->pool;
    uint32_t* __restrict__ cnt = K->cnt;
    DevCounters* __restrict__ ctr = K->ctr;
    const uint32_t pool_cap = K->pool_cap, obj_tag = K->obj_tag;

    if (CHAIN) {
        if (alive && bakedCount > 0) {
            Baked bk[BAKED_CAP];
            int n = bakedCount > BAKED_CAP ? BAKED_CAP : bakedCount;
            bk[0] = head;
            for (int k = 1; k < n; k++) bk[k] = extra[k - 1];
            int clean = n;
            if (n > 1) {  // reorderTriangles, :74-112
                for (int iround = 1; iround < n; iround++) {
                    for (int index = 0; index < n - iround; index++) {
                        Baked a = bk[index], b = bk[index + 1];
                        bool lessIdx = a.tri <= b.tri;
                        bool deeper = lessF(a.t, b.t);
                        if (lessIdx || deeper) { bk[index] = b; bk[index + 1] = a; }
                    }
                }
                clean = 0;
                for (int iround = 0; iround < BAKED_CAP; iround++) {
                    if (iround >= n - 1) break;
                    if (bk[iround + 1].tri != bk[iround].tri) bk[clean++] = bk[iround];
                }
                if (clean <= BAKED_CAP) bk[clean++] = bk[n - 1];
            }
            uint32_t oldN = hitN[i];
            uint32_t oldCount = oldN & 15u, oldOff = oldN >> 4;
            uint32_t k = (uint32_t)clean;
            uint32_t count = k > oldCount ? k : oldCount, off = 0;
            if (count > 1) {
                off = atomicAdd(&cnt[2], count - 1);
                if (off + (count - 1) > pool_cap) {
                    if (COUNT) atomicAdd(&ctr->chain_pool_drops, 1ull);
                    count = 1;
                } else {
                    for (uint32_t j = 1; j < count; j++)
                        pool[off + j - 1] = j < k ? make_float4(bk[j].u, bk[j].v, bk[j].t, __int_as_float(bk[j].tri | (int)obj_tag))
                                                  : pool[oldOff + j - 1];
                }
            }
            hit0[i] = make_float4(bk[0].u, bk[0].v, bk[0].t, __int_as_float(bk[0].tri | (int)obj_tag));
            hitN[i] = count | (off << 4);
        }
    } else if (alive) {
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
        if (count == 0) st_stream(&hit0[i], make_float4(0.f, 0.f, INF, __int_as_float(-1)));
        else st_stream(&hit0[i], make_float4(head.u, head.v, head.t, __int_as_float(head.tri)));
        st_stream(&hitN[i], count | (off << 4));
    }
#if PSM_EXP_WAVELOG
    if (!resume && cold_args()->wavelog && lane_id() == 0) {
        uint32_t* rec = cold_args()->wavelog + (size_t)8 * (blockIdx.x * (TRAV_BLOCK / 64) + (tid >> 6));
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        rec[0] = (uint32_t)wl_t0; rec[1] = (uint32_t)t1;                       // 100 MHz ticks (low words)
        rec[2] = wl_solo_before ? (uint32_t)wl_solo_before : (uint32_t)t1;     // when the wave went into the solo gear (or ended)
        rec[3] = wl_steps[1]; rec[4] = wl_steps[2]; rec[5] = wl_steps[3]; rec[6] = wl_steps[4]; rec[7] = wl_steps[5];
    }
#endif
    if (!resume) break;
    }  // batches
    if (COUNT) {
        DevCounters* ctr = cold_args()->ctr;
        uint32_t v = wave_sum(nV), t = wave_sum(nT), d = wave_sum(nDrop), c = wave_sum(nCap), b = wave_sum(nBakedDrop);
        if (lane_id() == 0) {
            atomicAdd(&ctr->wave_clock_ticks, (unsigned long long)(__builtin_amdgcn_s_memtime() - dg_t0));
            atomicAdd(&ctr->wave_real_ticks, (unsigned long long)(__builtin_amdgcn_s_memrealtime() - dg_r0));
            atomicAdd(&ctr->wave_steps, dg_steps);
            atomicAdd(&ctr->waves, 1ull);
            if (v) atomicAdd(&ctr->node_visits, (unsigned long long)v);
            if (t) atomicAdd(&ctr->tri_tests, (unsigned long long)t);
            if (d) atomicAdd(&ctr->stack_drops, (unsigned long long)d);
            if (c) atomicAdd(&ctr->iter_caps, (unsigned long long)c);
            if (b) atomicAdd(&ctr->baked_drops, (unsigned long long)b);
        }
    }
}


constexpr uint32_t RESUME_GRID_CAP = 256 * 32 / (TRAV_BLOCK / 64);  // 256 CUs x 32 resident waves: a resume launch never needs more blocks

// continuation queues of the hand-over schedules: one entry per ray of the Pipeline's currentRayLimit
static int ensure_phase_buffers(psm_rt* r) {
    psm_ctx* c = r->ctx;
    if (r->d_phase_mem && r->phase_cap == r->limit) return PSM_OK;
    if (r->d_phase_mem) { (void)hipStreamSynchronize(c->stream); (void)hipFree(r->d_phase_mem); r->d_phase_mem = nullptr; }
    const size_t L = r->limit;
    const size_t per = L * (4 * 5 + 16 + 4 * STACK_CAP);  // idx, cur, misc, predist, lastTri, head, stack
    PSM_HIP(c, hipMalloc(&r->d_phase_mem, 2 * per + 2 * sizeof(uint32_t) * MAX_PHASES));
    char* base = (char*)r->d_phase_mem;
    for (int k = 0; k < 2; k++) {
        r->phase_state[k].base = base + k * per;
        r->phase_state[k].capacity = (uint32_t)L;
    }
    r->d_phase_cnt = (uint32_t*)(base + 2 * per);   // two sets: a round counts in one and its last launch clears the other
    r->phase_dirty = true;
    r->phase_cap = (uint32_t)L;
    return PSM_OK;
}

// One launch of a hand-over schedule: the wave-step cap (0xFFFFFFFF: none) and the live-lane threshold (0: none)
struct PhasePlan {
    uint32_t cap, min_live;
};

// Which kernel(s) an intersection() over n rays runs as (psm_rt_set_traverse_mode). Results never depend on it.
static int plan_traverse(const psm_rt* r, uint32_t n, bool chain, std::vector<PhasePlan>& plan) {
    plan.clear();
    int mode = r->trav_mode;
    if (chain) return PSM_TRAVERSE_WHOLE;  // later hierarchies of a multi-BVH queue: rt_traverse<*, CHAIN>
    // AUTO, measured on MI355X: with several frames in flight the ballot-triggered hand-over wins
    // (2.65 against 2.79 ms per C3 frame, 14.0 against 15.3 on C5's scene: fewer VALU instructions, and the other frames'
    // kernels fill the tails its extra launches add); a frame on its own is bound by its longest ray, which the extra
    // launches serialise, so it runs one launch; so do rounds under phase_min_rays rays (tiles).
    if (mode == PSM_TRAVERSE_AUTO) mode = r->in_flight > 1 ? PSM_TRAVERSE_ADAPTIVE : PSM_TRAVERSE_WHOLE;
    if (mode == PSM_TRAVERSE_WHOLE || n < r->phase_min_rays) return PSM_TRAVERSE_WHOLE;
    if (mode == PSM_TRAVERSE_PHASED) {
        for (int k = 0; k < r->phase_caps_n; k++) plan.push_back(PhasePlan{r->phase_caps[k], 0u});
    } else {
        // adaptive: a wave that hands over leaves at most min_live - 1 rays behind, so launch k has at most
        // n * ((min_live - 1) / 64)^k rays: stop planning where that bound is small enough to finish
        uint64_t bound = n;
        const uint32_t T = r->adapt_min_live;
        while (plan.size() + 1 < r->adapt_max_launches && plan.size() + 2 < MAX_PHASES && bound > r->adapt_final_rays) {
            plan.push_back(PhasePlan{0xFFFFFFFFu, T});
            bound = ((bound + 63) / 64) * (T - 1);
        }
    }
    return plan.empty() ? PSM_TRAVERSE_WHOLE : mode;
}

int launch_rt_traverse(psm_rt* r, psm_bvh* b) {
    psm_ctx* c = r->ctx;
    uint32_t n = r->ray_count;
    if (n == 0) return PSM_OK;
    // the first intersection() after the queue changed starts the chains; later ones extend them
    const bool chain = r->trav_n > 0;
    if (r->trav_n >= MAX_TRAV_OBJECTS) return set_err(c, PSM_ERR_CAPACITY, "more than 16 hierarchies traversed for one ray queue");
    if (b->tri_count > (1u << OBJ_SHIFT)) return set_err(c, PSM_ERR_CAPACITY, "hierarchy too large for the object tag (2^27 triangles)");
    const uint32_t tag = (uint32_t)r->trav_n << OBJ_SHIFT;
    r->last_objs[r->trav_n] = b;
    r->trav_objs[r->trav_n++] = b;
    std::vector<PhasePlan> plan;
    const int mode = plan_traverse(r, n, chain, plan);
    {
        uint32_t grid = ((n + TRAV_BLOCK - 1) / TRAV_BLOCK + (8u * XCD_RUN - 1u)) & ~(8u * XCD_RUN - 1u);  // a multiple of 8 XCDs x XCD_RUN (rt_traverse: vblock)
        TravArgs ta = {};
        ta.qA = r->qA[r->cur]; ta.qB = r->qB[r->cur]; ta.qbases = r->q_bases[r->cur]; ta.qnb = r->q_nb[r->cur]; ta.nrays = n;
        ta.node32 = b->d_node32; ta.tri48 = b->d_tri48; ta.sm = b->d_small;
        ta.hit0 = r->hit0; ta.hitN = r->hitN; ta.pool = r->pool; ta.cnt = r->d_cnt; ta.ctr = c->d_counters;
        ta.pool_cap = r->pool_cap; ta.obj_tag = tag;
        ta.cap = 0xFFFFFFFFu;
        ta.solo_max = r->solo_max;
#if PSM_EXP_WAVELOG
        { const char* e = getenv("PSM_EXP_WAVELOG_PTR"); ta.wavelog = e ? (uint32_t*)(uintptr_t)strtoull(e, nullptr, 0) : nullptr; }
#endif
        if (mode == PSM_TRAVERSE_WHOLE) {
            TimedScope ts(c, CAT_TRAVERSE);
            if (chain) {
                if (c->counting) rt_traverse<true, true, false><<<grid, TRAV_BLOCK, 0, c->stream>>>(ta);
                else rt_traverse<false, true, false><<<grid, TRAV_BLOCK, 0, c->stream>>>(ta);
            } else {
                if (c->counting) rt_traverse<true, false, false><<<grid, TRAV_BLOCK, 0, c->stream>>>(ta);
                else rt_traverse<false, false, false><<<grid, TRAV_BLOCK, 0, c->stream>>>(ta);
            }
        } else {
            // launch p hands unfinished rays to launch p+1 through a dense continuation queue; the last launch runs
            // to completion. Resume launches are persistent (a fixed grid strides over the queue), sized by the
            // host's upper bound of the rays that can be waiting; the device count decides what they do.
            int rc = ensure_phase_buffers(r);
            if (rc != PSM_OK) return rc;
            const size_t np = plan.size() + 1;
            if (r->phase_dirty) PSM_HIP(c, hipMemsetAsync(r->d_phase_cnt, 0, 2 * sizeof(uint32_t) * MAX_PHASES, c->stream));   // first use, or a round that failed half way
            r->phase_dirty = true;
            uint32_t* const pcnt = r->d_phase_cnt + r->phase_set * MAX_PHASES;
            uint32_t* const pother = r->d_phase_cnt + (r->phase_set ^ 1u) * MAX_PHASES;
            r->phase_set ^= 1u;
            uint64_t bound = n;
            for (size_t p = 0; p < np; p++) {
                TravArgs ph = ta;
                ph.zero_cnt = p + 1 == np ? pother : nullptr;
                ph.cap = p < plan.size() ? plan[p].cap : 0xFFFFFFFFu;
                ph.min_live = p < plan.size() ? plan[p].min_live : 0u;
                ph.min_steps = r->adapt_min_steps;
                ph.final_rays = mode == PSM_TRAVERSE_ADAPTIVE ? r->adapt_final_rays : 0u;
                ph.in_count = p == 0 ? nullptr : pcnt + (p - 1);
                ph.in = r->phase_state[(p + 1) & 1];
                ph.out = r->phase_state[p & 1];
                ph.out_count = pcnt + p;
                uint32_t g = grid;
                if (p > 0) {
                    uint64_t need = (bound + TRAV_BLOCK - 1) / TRAV_BLOCK;
                    g = (uint32_t)(need < RESUME_GRID_CAP ? need : RESUME_GRID_CAP);
                    if (g == 0) g = 1;
                }
                TimedScope ts(c, CAT_TRAVERSE_HANDOVER);
                if (c->counting) rt_traverse<true, false, true><<<g, TRAV_BLOCK, 0, c->stream>>>(ph);
                else rt_traverse<false, false, true><<<g, TRAV_BLOCK, 0, c->stream>>>(ph);
                if (ph.min_live > 0 && ph.cap == 0xFFFFFFFFu) bound = ((bound + 63) / 64) * (ph.min_live - 1);
            }
            PSM_HIP(c, hipGetLastError());
            r->phase_dirty = false;
        }
    }
    PSM_HIP(c, hipGetLastError());
    c->rays_traced += n;
    return PSM_OK;
}


}  // namespace psm
