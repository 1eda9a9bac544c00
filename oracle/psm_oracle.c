/*
 * psm_oracle.c -- CPU restatement of the prismarine-core hot path (build + trace).
 * See psm_oracle.h: TEST INFRASTRUCTURE ONLY, PARITY UNPINNED BY THE REFERENCE.
 *
 * Reference files restated (paths relative to /root/reference):
 *   ShadersSDK/include/{morton,mathlib,vertex,structs,constants}.glsl
 *   ShadersSDK/hlbvh/{minmax,aabbmaker,build-new,child-link,refit}.comp
 *   ShadersSDK/radix/{histogram,pfx-work,permute}.comp  (collapse to a stable LSD sort)
 *   ShadersSDK/raytracing/directTraverse.comp
 *   Include/Prismarine/TriangleHierarchy.inl:206-329, Radix.hpp:47-74
 */
#include "psm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* canonical scalar helpers (DESIGN.md "canonical arithmetic")         */
/* ------------------------------------------------------------------ */
#include "psm_oracle_internal.h"

/* packHalf2x16 rounding: round-to-nearest-even (SURVEY a-8 canonical rule) */
uint16_t psmo_f32_to_f16(float f) {
    uint32_t x = f2u(f);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t e = (x >> 23) & 0xffu;
    uint32_t m = x & 0x7fffffu;
    if (e == 255u) return (uint16_t)(sign | 0x7c00u | (m ? (0x200u | (m >> 13)) : 0u));
    int32_t E = (int32_t)e - 127 + 15;
    if (E >= 31) return (uint16_t)(sign | 0x7c00u);
    if (E <= 0) {
        if (E < -10) return (uint16_t)sign;
        m |= 0x800000u;
        uint32_t shift = (uint32_t)(14 - E);
        uint32_t hm = m >> shift;
        uint32_t rem = m & ((1u << shift) - 1u);
        uint32_t half = 1u << (shift - 1u);
        if (rem > half || (rem == half && (hm & 1u))) hm++;
        return (uint16_t)(sign | hm);
    }
    uint32_t h = ((uint32_t)E << 10) | (m >> 13);
    uint32_t rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;
    return (uint16_t)(sign | h);
}

float psmo_f16_to_f32(uint16_t h) {
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1fu;
    uint32_t m = h & 0x3ffu;
    if (e == 0) {
        if (m == 0) return u2f(sign);
        /* subnormal: m * 2^-24 */
        float v = (float)m * 5.9604644775390625e-08f;
        return (sign ? -v : v);
    }
    if (e == 31) return u2f(sign | 0x7f800000u | (m << 13));
    return u2f(sign | ((e - 15 + 127) << 23) | (m << 13));
}

/* include/morton.glsl:37-51 */
static inline uint64_t part1by2_64(uint32_t a) {
    uint64_t x = a & 0x1fffffull;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}
uint64_t psmo_morton3_64(uint32_t x, uint32_t y, uint32_t z) {
    return part1by2_64(x) | (part1by2_64(y) << 1) | (part1by2_64(z) << 2);
}

/* ------------------------------------------------------------------ */
/* build                                                               */
/* ------------------------------------------------------------------ */

static inline void load_tri_xformed(const float* tris, int t, const float M[16], float v[3][4]) {
    for (int k = 0; k < 3; k++) {
        float p[4] = {tris[9 * t + 3 * k + 0], tris[9 * t + 3 * k + 1], tris[9 * t + 3 * k + 2], 1.0f};
        mat_vec(M, p, v[k]);
    }
}

/* hlbvh/minmax.comp:50-79 + host reduce TriangleHierarchy.inl:248-255.
 * min/max are exact and order independent; every partial carries the -+1e-5
 * pad (minmax.comp:76-77) so the reduced result is (exact min) - 1e-5f.
 * The shader reduces in an order of its own (two primitives per thread, a tree in shared memory, a host loop over the
 * workgroups); for that to have ONE result when a coordinate is NaN -- GLSL leaves min / max of a NaN undefined -- min and max
 * here are minNum / maxNum, what the v_min_f32 / v_max_f32 of the reference's target hardware compute (and its min3 / max3
 * path, minmax.comp:30-34): a NaN coordinate is ignored, whatever the order. For every other input the result is the same
 * as with `y < x ? y : x` (zeros of either sign give the same padded bound). */
void psmo_minmax(const float* tris, int n, const float M[16], float mn[4], float mx[4]) {
    for (int c = 0; c < 4; c++) { mn[c] = 100000.f; mx[c] = -100000.f; }
    for (int t = 0; t < n; t++) {
        float v[3][4];
        load_tri_xformed(tris, t, M, v);
        for (int c = 0; c < 4; c++) {
            float lo = smin(smin(v[0][c], v[1][c]), v[2][c]);
            float hi = smax(smax(v[0][c], v[1][c]), v[2][c]);
            mn[c] = smin(mn[c], lo);
            mx[c] = smax(mx[c], hi);
        }
    }
    for (int c = 0; c < 4; c++) { mn[c] = mn[c] - 0.00001f; mx[c] = mx[c] + 0.00001f; }
}

/* ---- the host formulas of TriangleHierarchy.inl:226-232,257-267 with glm's own operation order --------------------
 * g-matrices are glm's dmat4 in memory: column-major, g[4*c + r]. Every function below restates the arithmetic of the
 * glm the reference vendors (detail/func_matrix.inl compute_inverse<4,4>, detail/type_mat4x4.inl operator*,
 * gtc/matrix_transform.inl translate / scale), operation by operation, so signs of zeros come out the same too;
 * pinned against that glm by tests/golden/glm_host_formulas.npz. */
static void glm_mul4d(const double* a, const double* b, double* o) { /* Result[c] = A0*B[c][0] + A1*B[c][1] + A2*B[c][2] + A3*B[c][3] */
    double t[16];
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++)
            t[4 * c + r] = ((a[0 + r] * b[4 * c + 0] + a[4 + r] * b[4 * c + 1]) + a[8 + r] * b[4 * c + 2]) + a[12 + r] * b[4 * c + 3];
    memcpy(o, t, sizeof(t));
}
static void glm_inverse4d(const double* m, double* o) {
#define G(c, r) m[4 * (c) + (r)]
    double c00 = G(2,2) * G(3,3) - G(3,2) * G(2,3), c02 = G(1,2) * G(3,3) - G(3,2) * G(1,3), c03 = G(1,2) * G(2,3) - G(2,2) * G(1,3);
    double c04 = G(2,1) * G(3,3) - G(3,1) * G(2,3), c06 = G(1,1) * G(3,3) - G(3,1) * G(1,3), c07 = G(1,1) * G(2,3) - G(2,1) * G(1,3);
    double c08 = G(2,1) * G(3,2) - G(3,1) * G(2,2), c10 = G(1,1) * G(3,2) - G(3,1) * G(1,2), c11 = G(1,1) * G(2,2) - G(2,1) * G(1,2);
    double c12 = G(2,0) * G(3,3) - G(3,0) * G(2,3), c14 = G(1,0) * G(3,3) - G(3,0) * G(1,3), c15 = G(1,0) * G(2,3) - G(2,0) * G(1,3);
    double c16 = G(2,0) * G(3,2) - G(3,0) * G(2,2), c18 = G(1,0) * G(3,2) - G(3,0) * G(1,2), c19 = G(1,0) * G(2,2) - G(2,0) * G(1,2);
    double c20 = G(2,0) * G(3,1) - G(3,0) * G(2,1), c22 = G(1,0) * G(3,1) - G(3,0) * G(1,1), c23 = G(1,0) * G(2,1) - G(2,0) * G(1,1);
    const double f0[4] = {c00, c00, c02, c03}, f1[4] = {c04, c04, c06, c07}, f2[4] = {c08, c08, c10, c11};
    const double f3[4] = {c12, c12, c14, c15}, f4[4] = {c16, c16, c18, c19}, f5[4] = {c20, c20, c22, c23};
    const double v0[4] = {G(1,0), G(0,0), G(0,0), G(0,0)}, v1[4] = {G(1,1), G(0,1), G(0,1), G(0,1)};
    const double v2[4] = {G(1,2), G(0,2), G(0,2), G(0,2)}, v3[4] = {G(1,3), G(0,3), G(0,3), G(0,3)};
    const double sa[4] = {1, -1, 1, -1}, sb[4] = {-1, 1, -1, 1};
    double inv[16];
    for (int k = 0; k < 4; k++) {
        inv[0 + k] = ((v1[k] * f0[k] - v2[k] * f1[k]) + v3[k] * f2[k]) * sa[k];
        inv[4 + k] = ((v0[k] * f0[k] - v2[k] * f3[k]) + v3[k] * f4[k]) * sb[k];
        inv[8 + k] = ((v0[k] * f1[k] - v1[k] * f3[k]) + v3[k] * f5[k]) * sa[k];
        inv[12 + k] = ((v0[k] * f2[k] - v1[k] * f4[k]) + v2[k] * f5[k]) * sb[k];
    }
    double d0 = G(0,0) * inv[0], d1 = G(0,1) * inv[4], d2 = G(0,2) * inv[8], d3 = G(0,3) * inv[12];
    double one_over_det = 1.0 / ((d0 + d1) + (d2 + d3));
    for (int i = 0; i < 16; i++) o[i] = inv[i] * one_over_det;
#undef G
}
static void glm_identity4d(double* g) {
    for (int i = 0; i < 16; i++) g[i] = (i % 5 == 0) ? 1.0 : 0.0;
}
static void glm_from_rowmajor(const double* r, double* g) {
    for (int c = 0; c < 4; c++) for (int q = 0; q < 4; q++) g[4 * c + q] = r[4 * q + c];
}
/* float(g) read back row-major: what value_ptr(transpose(mat4(g))) uploads */
static void glm_to_rowmajor_f(const double* g, float* M) {
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) M[4 * r + c] = (float)g[4 * c + r];
}

/* first-pass transform, TriangleHierarchy.inl:226-232: dmat4 mat(1.0); mat *= inverse(optimization) */
void psmo_inverse_opt(const double opt[16], float M[16]) {
    double go[16], gi[16], mat[16];
    glm_from_rowmajor(opt, go);
    glm_inverse4d(go, gi);
    glm_identity4d(mat);
    glm_mul4d(mat, gi, mat);
    glm_to_rowmajor_f(mat, M);
}

/* TriangleHierarchy.inl:257-267: scale = mx-mn, offset = mn (float);
 * mat(1.0); mat *= inverse(translate(dvec3(offset)) * scale(dvec3(scale))); mat *= inverse(dmat4(opt)); cast to float.
 * transformInv = inverse(mat4(mat)) in float (:266): computed here in double from the float matrix -- it is not read
 * by any kernel of the path. */
void psmo_fit_transform(const float mn[4], const float mx[4], const double opt[16], float M[16],
                        float Minv[16]) {
    float scale[3], offset[3];
    for (int c = 0; c < 3; c++) { scale[c] = mx[c] - mn[c]; offset[c] = mn[c]; }
    double I[16], T[16], S[16], TS[16], iTS[16], go[16], iopt[16], mat[16];
    glm_identity4d(I);
    /* translate(m, v): Result = m; Result[3] = m[0]*v[0] + m[1]*v[1] + m[2]*v[2] + m[3] */
    memcpy(T, I, sizeof(T));
    for (int r = 0; r < 4; r++)
        T[12 + r] = ((I[0 + r] * (double)offset[0] + I[4 + r] * (double)offset[1]) + I[8 + r] * (double)offset[2]) + I[12 + r];
    /* scale(m, v): Result[i] = m[i] * v[i], Result[3] = m[3] */
    for (int r = 0; r < 4; r++) {
        S[0 + r] = I[0 + r] * (double)scale[0];
        S[4 + r] = I[4 + r] * (double)scale[1];
        S[8 + r] = I[8 + r] * (double)scale[2];
        S[12 + r] = I[12 + r];
    }
    glm_mul4d(T, S, TS);
    glm_inverse4d(TS, iTS);
    glm_from_rowmajor(opt, go);
    glm_inverse4d(go, iopt);
    glm_identity4d(mat);
    glm_mul4d(mat, iTS, mat);
    glm_mul4d(mat, iopt, mat);
    glm_to_rowmajor_f(mat, M);
    double gm[16], gim[16];
    for (int c = 0; c < 4; c++) for (int r = 0; r < 4; r++) gm[4 * c + r] = (double)M[4 * r + c];
    glm_inverse4d(gm, gim);
    glm_to_rowmajor_f(gim, Minv);
}

/* hlbvh/aabbmaker.comp:142-232 at splitLimit = 0 (:139-140).
 * Canonical leaf slot: rank among kept triangles in ascending t (SURVEY a-8). */
int psmo_morton_leaves(const float* tris, int n, const float M[16], uint64_t* keys, int32_t* idx,
                       psmo_node* leafs) {
    int to = 0;
    for (int t = 0; t < n; t++) {
        float v[3][4];
        load_tri_xformed(tris, t, M, v);
        float c[4];
        for (int k = 0; k < 4; k++) c[k] = ((v[0][k] + v[1][k]) + v[2][k]) * 0.33333333333333f;
        float s[3];
        for (int k = 0; k < 3; k++)
            s[k] = (fabsf(v[0][k] - c[k]) + fabsf(v[1][k] - c[k])) + fabsf(v[2][k] - c[k]);
        if (len3(s) < 1.e-5f) continue; /* :160 */
        float bmn[4], bmx[4];
        for (int k = 0; k < 4; k++) {
            bmn[k] = pmin(pmin(v[0][k], v[1][k]), v[2][k]);
            bmx[k] = pmax(pmax(v[0][k], v[1][k]), v[2][k]);
        }
        /* :176 greaterEqualF(branges, 0) on x,y,z */
        if (!(greaterEqualF(bmx[0] - bmn[0], 0.f) && greaterEqualF(bmx[1] - bmn[1], 0.f) &&
              greaterEqualF(bmx[2] - bmn[2], 0.f)))
            continue;
        uint32_t q[3];
        for (int k = 0; k < 3; k++) {
            float f = floorf(pclamp(c[k], 0.0f, 0.99999f) * 2097152.0f);
            uint32_t u = (uint32_t)f;
            q[k] = u > 0x1FFFFFu ? 0x1FFFFFu : u;
        }
        keys[to] = psmo_morton3_64(q[0], q[1], q[2]);
        idx[to] = to;
        for (int k = 0; k < 4; k++) { bmn[k] = bmn[k] - PSMO_PZERO; bmx[k] = bmx[k] + PSMO_PZERO; }
        pack_half4(bmn, &leafs[to].box[0]);
        pack_half4(bmx, &leafs[to].box[2]);
        leafs[to].pdata[0] = to;
        leafs[to].pdata[1] = to;
        leafs[to].pdata[2] = -1;
        leafs[to].pdata[3] = t;
        to++;
    }
    return to;
}

/* Radix.hpp:47-74 + radix/{histogram,pfx-work,permute}.comp: 8 passes of 8 bits, stable, ascending,
 * (u64 key, u32 value).  Restated as the literal LSD counting sort. */
void psmo_radix_sort(uint64_t* keys, int32_t* vals, int n) {
    if (n <= 1) return;
    uint64_t* tk = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)n);
    int32_t* tv = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
    for (int pass = 0; pass < 8; pass++) {
        size_t count[257];
        memset(count, 0, sizeof(count));
        int sh = pass * 8;
        for (int i = 0; i < n; i++) count[((keys[i] >> sh) & 0xff) + 1]++;
        for (int d = 0; d < 256; d++) count[d + 1] += count[d];
        for (int i = 0; i < n; i++) {
            size_t p = count[(keys[i] >> sh) & 0xff]++;
            tk[p] = keys[i];
            tv[p] = vals[i];
        }
        memcpy(keys, tk, sizeof(uint64_t) * (size_t)n);
        memcpy(vals, tv, sizeof(int32_t) * (size_t)n);
    }
    free(tk);
    free(tv);
}

/* hlbvh/build-new.comp:18-23 */
static inline int nlz64(uint64_t x) {
    if (x == 0) return 64;
    int n = 0;
    while (!(x & 0x8000000000000000ull)) { x <<= 1; n++; }
    return n;
}

/* hlbvh/build-new.comp:33-56 */
int psmo_find_split(const uint64_t* keys, int first, int last, uint64_t* key_reads) {
    uint64_t firstCode = keys[first];
    uint64_t lastCode = keys[last];
    uint64_t reads = 2;
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
                reads++;
                int splitPrefix = nlz64(firstCode ^ splitCode);
                if (splitPrefix > commonPrefix) split = newSplit;
            }
            if (step <= 1) break;
        }
    }
    if (key_reads) *key_reads += reads;
    if (split < first) split = first;
    if (split > last - 1) split = last - 1;
    return split;
}

static const uint32_t HALF_P1000 = 0x63D0u; /* packHalf(1000.0)  */
static const uint32_t HALF_N1000 = 0xE3D0u; /* packHalf(-1000.0) */

/* fp16 min/max with -0 < +0 (order-independent refit) */
static inline int32_t half_key(uint16_t h) { return (h & 0x8000u) ? -(int32_t)(h & 0x7fffu) - 1 : (int32_t)h; }
static inline uint16_t half_min(uint16_t a, uint16_t b) { return half_key(b) < half_key(a) ? b : a; }
static inline uint16_t half_max(uint16_t a, uint16_t b) { return half_key(a) < half_key(b) ? b : a; }
static inline uint32_t half2_min(uint32_t a, uint32_t b) {
    return (uint32_t)half_min((uint16_t)a, (uint16_t)b) | ((uint32_t)half_min((uint16_t)(a >> 16), (uint16_t)(b >> 16)) << 16);
}
static inline uint32_t half2_max(uint32_t a, uint32_t b) {
    return (uint32_t)half_max((uint16_t)a, (uint16_t)b) | ((uint32_t)half_max((uint16_t)(a >> 16), (uint16_t)(b >> 16)) << 16);
}

/* build-new.comp:120-153 (level loop driven by TriangleHierarchy.inl:304-313),
 * child-link.comp:16-59, refit.comp:21-114.
 * Canonical numbering = BFS (SURVEY a-9): root 0; the i-th queued node of a level gets
 * children base+2i, base+2i+1; internal children are enqueued left then right.
 * Returns the node count. */
int psmo_build_nodes(const uint64_t* keys, const int32_t* idx, psmo_node* leafs, int n,
                     psmo_node* nodes, int* levels, uint64_t* key_reads) {
    if (levels) *levels = 0;
    if (n <= 0) return 0;
    int* cur = (int*)malloc(sizeof(int) * (size_t)(n + 1));
    int* nxt = (int*)malloc(sizeof(int) * (size_t)(n + 1));
    int* leafIndices = (int*)malloc(sizeof(int) * (size_t)(n + 1));
    int ncur = 0, nleaf = 0, lcounter = 0;
    /* root, build-new.comp:128-141 */
    {
        int hid = lcounter++;
        cur[ncur++] = hid;
        nodes[hid].box[0] = nodes[hid].box[1] = HALF_P1000 | (HALF_P1000 << 16);
        nodes[hid].box[2] = nodes[hid].box[3] = HALF_N1000 | (HALF_N1000 << 16);
        nodes[hid].pdata[0] = 0;
        nodes[hid].pdata[1] = n - 1;
        nodes[hid].pdata[2] = -1;
        nodes[hid].pdata[3] = -1;
    }
    int nlev = 0;
    while (ncur > 0) {
        int nnxt = 0;
        int base = lcounter;
        for (int i = 0; i < ncur; i++) {
            int prID = cur[i];
            psmo_node parent = nodes[prID];
            if (parent.pdata[0] == parent.pdata[1]) continue; /* splitNode :74 */
            int split = psmo_find_split(keys, parent.pdata[0], parent.pdata[1], key_reads);
            int hid = base + 2 * i;
            /* all queued nodes are internal except possibly a 1-leaf root */
            int tr[4] = {parent.pdata[0], split, split + 1, parent.pdata[1]};
            for (int c = 0; c < 2; c++) {
                int h = hid + c;
                nodes[h].box[0] = nodes[h].box[1] = HALF_P1000 | (HALF_P1000 << 16);
                nodes[h].box[2] = nodes[h].box[3] = HALF_N1000 | (HALF_N1000 << 16);
                nodes[h].pdata[0] = tr[2 * c];
                nodes[h].pdata[1] = tr[2 * c + 1];
                nodes[h].pdata[2] = prID;
                nodes[h].pdata[3] = -1;
                int isLeaf = (tr[2 * c + 1] - tr[2 * c]) < 1;
                if (isLeaf) leafIndices[nleaf++] = h; else nxt[nnxt++] = h;
            }
            nodes[prID].pdata[0] = hid;
            nodes[prID].pdata[1] = hid + 1;
            if (lcounter < hid + 2) lcounter = hid + 2;
        }
        int* tmp = cur; cur = nxt; nxt = tmp;
        ncur = nnxt;
        nlev++;
    }
    if (levels) *levels = nlev;
    /* child-link.comp:27-53 */
    for (int g = 0; g < nleaf; g++) {
        int id = leafIndices[g];
        psmo_node child = nodes[id];
        int leafID = idx[child.pdata[0]];
        leafs[leafID].pdata[2] = id;
        memcpy(child.box, leafs[leafID].box, sizeof(child.box));
        int ry = child.pdata[1];
        child.pdata[0] = ry;
        child.pdata[1] = ry;
        child.pdata[3] = leafs[leafID].pdata[3];
        nodes[id] = child;
    }
    /* refit.comp:63-110: bottom-up union; the BFS numbering puts children after parents,
     * so a reverse sweep visits children first. min/max are exact => any schedule agrees. */
    for (int id = lcounter - 1; id >= 0; id--) {
        psmo_node* nd = &nodes[id];
        if (nd->pdata[0] != nd->pdata[1]) {
            const psmo_node* ln = &nodes[nd->pdata[0]];
            const psmo_node* rn = &nodes[nd->pdata[1]];
            nd->box[0] = half2_min(ln->box[0], rn->box[0]);
            nd->box[1] = half2_min(ln->box[1], rn->box[1]);
            nd->box[2] = half2_max(ln->box[2], rn->box[2]);
            nd->box[3] = half2_max(ln->box[3], rn->box[3]);
        }
    }
    free(cur); free(nxt); free(leafIndices);
    return lcounter;
}

/* Refit only: the triangles have moved, the tree stays. Leaf boxes as aabbmaker.comp:165-176,193-194 computes them,
 * with the transform M of the build, for the triangle each leaf slot holds (a leaf stays a leaf even if its triangle would now
 * fail aabbmaker's degeneracy tests: those decide what a BUILD keeps); then child-link.comp:38-45 (the leaf node takes its
 * leaf's box) and refit.comp:63-110 (bottom-up union) over the nodes as built. */
void psmo_refit(const float* tris, const float M[16], psmo_node* leafs, int nleafs, psmo_node* nodes, int nnodes) {
    for (int s = 0; s < nleafs; s++) {
        float v[3][4], bmn[4], bmx[4];
        load_tri_xformed(tris, leafs[s].pdata[3], M, v);
        for (int k = 0; k < 4; k++) {
            bmn[k] = pmin(pmin(v[0][k], v[1][k]), v[2][k]) - PSMO_PZERO;
            bmx[k] = pmax(pmax(v[0][k], v[1][k]), v[2][k]) + PSMO_PZERO;
        }
        pack_half4(bmn, &leafs[s].box[0]);
        pack_half4(bmx, &leafs[s].box[2]);
        if (leafs[s].pdata[2] >= 0 && leafs[s].pdata[2] < nnodes) memcpy(nodes[leafs[s].pdata[2]].box, leafs[s].box, sizeof(leafs[s].box));
    }
    for (int id = nnodes - 1; id >= 0; id--) {   /* BFS numbering: children after parents */
        psmo_node* nd = &nodes[id];
        if (nd->pdata[0] != nd->pdata[1]) {
            const psmo_node* ln = &nodes[nd->pdata[0]];
            const psmo_node* rn = &nodes[nd->pdata[1]];
            nd->box[0] = half2_min(ln->box[0], rn->box[0]);
            nd->box[1] = half2_min(ln->box[1], rn->box[1]);
            nd->box[2] = half2_max(ln->box[2], rn->box[2]);
            nd->box[3] = half2_max(ln->box[3], rn->box[3]);
        }
    }
}

/* TriangleHierarchy::build, TriangleHierarchy.inl:206-329 */
int psmo_build(const float* tris, int n, const double opt[16], float M[16], uint64_t* keys,
               int32_t* idx, psmo_node* leafs, psmo_node* nodes) {
    float M0[16], Minv[16], mn[4], mx[4];
    psmo_inverse_opt(opt, M0);
    psmo_minmax(tris, n, M0, mn, mx);
    psmo_fit_transform(mn, mx, opt, M, Minv);
    int cnt = psmo_morton_leaves(tris, n, M, keys, idx, leafs);
    if (cnt <= 0) return 0;
    psmo_radix_sort(keys, idx, cnt);
    psmo_build_nodes(keys, idx, leafs, cnt, nodes, NULL, NULL);
    return cnt;
}

/* ------------------------------------------------------------------ */
/* geometry ingestion, vertex/loader.comp:32-152                       */
/* ------------------------------------------------------------------ */
static void read_by_accessor(const psmo_mesh_desc* d, int accessorID, uint32_t idx, float out[4]) { /* :32-54 */
    const psmo_accessor* ac = &d->accessors[accessorID];
    const psmo_buffer_view* bv = &d->views[ac->buffer_view];
    uint32_t cmps = (uint32_t)ac->components & 3u;
    uint32_t stride4 = bv->stride4 > 0 ? (uint32_t)bv->stride4 : (cmps + 1u);
    uint32_t off = idx * stride4 + (uint32_t)bv->offset4 + (uint32_t)ac->offset4;
    for (uint32_t k = 0; k < 4; k++) out[k] = (k <= cmps && off + k < d->vertex_floats) ? d->vertices[off + k] : 0.f;
}

int psmo_load_mesh(const psmo_mesh_desc* d, int storing_offset, float* pos, float* nrm, int32_t* mats) {
    return psmo_load_mesh_tex(d, storing_offset, pos, nrm, mats, NULL);
}

int psmo_load_mesh_tex(const psmo_mesh_desc* d, int storing_offset, float* pos, float* nrm, int32_t* mats, float* tex) {
    int trp = d->primitive_type == 1 ? 4 : 3;
    int istride = d->primitive_type == 1 ? 2 : 1;
    for (int ct = 0; ct < d->node_count; ct++) {
        float vertice[4][3], normal[4][3], tuv[4][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
        for (int i = 0; i < trp; i++) {
            uint32_t ptri = (uint32_t)d->loading_offset + (uint32_t)(ct * trp + i);
            uint32_t vi = ptri;
            if (d->is_indexed != 0) {
                if (d->index16) { uint32_t wd = (ptri >> 1) < d->index_words ? d->indices[ptri >> 1] : 0u; vi = (wd >> (16u * (ptri & 1u))) & 0xFFFFu; }
                else vi = ptri < d->index_words ? d->indices[ptri] : 0u;
            }
            float p[4], n[4] = {0.f, 0.f, 0.f, 0.f};
            read_by_accessor(d, d->vertex_accessor, vi, p);
            if (d->normal_accessor != -1) read_by_accessor(d, d->normal_accessor, vi, n);
            if (d->texcoord_accessor != -1) { float t4[4]; read_by_accessor(d, d->texcoord_accessor, vi, t4); tuv[i][0] = t4[0]; tuv[i][1] = t4[1]; } /* :94-96 */
            tuv[i][1] = 1.0f - tuv[i][1]; /* INVERT_TX_Y, :97-99 (build-spv-new.bat:31-32) */
            float pv[4] = {p[0], p[1], p[2], 1.0f}, nv[4] = {n[0], n[1], n[2], 0.0f}, po[4], no[4];
            mat_vec(d->transform, pv, po);       /* :101 */
            matT_vec(d->transform_inv, nv, no);  /* :100 */
            for (int k = 0; k < 3; k++) { vertice[i][k] = po[k] / po[3]; normal[i][k] = no[k]; }
        }
        float e1[3] = {vertice[1][0] - vertice[0][0], vertice[1][1] - vertice[0][1], vertice[1][2] - vertice[0][2]};
        float e2[3] = {vertice[2][0] - vertice[0][0], vertice[2][1] - vertice[0][1], vertice[2][2] - vertice[0][2]};
        float cr[3], offsetnormal[3];
        cross3(e1, e2, cr);
        normalize3(cr, offsetnormal); /* :119 */
        for (int q = 0; q < istride; q++) {
            int tidc = storing_offset + ct * istride + q;
            const int m[3] = {q == 0 ? 0 : 3, q == 0 ? 1 : 0, 2}; /* :56 */
            mats[tidc] = d->material_id;
            for (int i = 0; i < 3; i++) {
                const float* nn = normal[m[i]];
                float an[3] = {fabsf(nn[0]), fabsf(nn[1]), fabsf(nn[2])};
                float use[3];
                if (pmax(an[0], pmax(an[1], an[2])) >= 0.0001f && d->normal_accessor != -1) normalize3(nn, use);
                else normalize3(offsetnormal, use); /* :124-128 */
                for (int k = 0; k < 3; k++) { pos[9 * tidc + 3 * i + k] = vertice[m[i]][k]; nrm[9 * tidc + 3 * i + k] = use[k]; }
                if (tex) { tex[6 * tidc + 2 * i] = tuv[m[i]][0]; tex[6 * tidc + 2 * i + 1] = tuv[m[i]][1]; }
            }
        }
    }
    return d->node_count * istride;
}

/* ------------------------------------------------------------------ */
/* traversal, raytracing/directTraverse.comp                           */
/* ------------------------------------------------------------------ */

/* include/mathlib.glsl:107-126 */
static float intersectCubeSingle(const float o[3], const float ray[3], const float cmn[3],
                                 const float cmx[3], float* near, float* far) {
    float t1[3], t2[3];
    for (int k = 0; k < 3; k++) {
        float dr = 1.0f / ray[k];
        float norig = -o[k] * dr;
        float tMin = fmaf(cmn[k], dr, norig);
        float tMax = fmaf(cmx[k], dr, norig);
        t1[k] = smin(tMin, tMax);
        t2[k] = smax(tMin, tMax);
    }
    float tNear = smax(smax(t1[0], t1[1]), t1[2]);
    float tFar = smin(smin(t2[0], t2[1]), t2[2]);
    int isCube = greaterEqualF(tFar, tNear) && greaterEqualF(tFar, 0.0f);
    float inf = PSMO_INFINITY;
    *near = isCube ? smin(tNear, tFar) : inf;
    *far = isCube ? smax(tNear, tFar) : inf;
    return isCube ? (lessF(*near, 0.0f) ? *far : *near) : inf;
}

/* include/mathlib.glsl:129-193, fp32 branch (canonical = fp32 slab
 * maths on fp16-stored boxes). One child at a time; the dual form is component-wise. */
static float intersectCubeChild(const float o[3], const float dr[3], const float cmn[4],
                                const float cmx[4], float* near, float* far) {
    float t1[3], t2[3];
    for (int k = 0; k < 3; k++) {
        float norig = -o[k] * dr[k];
        float tMin = fmaf(cmn[k], dr[k], norig);
        float tMax = fmaf(cmx[k], dr[k], norig);
        t1[k] = smin(tMin, tMax);
        t2[k] = smax(tMin, tMax);
    }
    float tNear = smax(smax(t1[0], t1[1]), t1[2]);
    float tFar = smin(smin(t2[0], t2[1]), t2[2]);
    float inf = PSMO_INFINITY;
    int isCube = ((tFar + PSMO_PZERO) >= tNear) && ((tFar + PSMO_PZERO) >= 0.0f);
    *near = isCube ? smin(tNear, tFar) : inf;
    *far = isCube ? smax(tNear, tFar) : inf;
    return ((*near + PSMO_PZERO) <= 0.0f) ? *far : *near;
}

/* include/vertex.glsl:140-189 */
static float intersectTriangle(const float* tris, const float orig[3], const float dir[3], int tri,
                               float* U, float* V, int valid, psmo_counters* ctr) {
    float T = PSMO_INFINITY;
    if (tri == -1) valid = 0;
    if (valid) {
        const float* v0 = &tris[9 * tri + 0];
        const float* v1 = &tris[9 * tri + 3];
        const float* v2 = &tris[9 * tri + 6];
        if (ctr) ctr->tri_tests++;
        float e1[3] = {v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2]};
        float e2[3] = {v2[0] - v0[0], v2[1] - v0[1], v2[2] - v0[2]};
        float pvec[3];
        cross3(dir, e2, pvec);
        float det = dot3(e1, pvec);
        if (fabsf(det) <= 0.0f) valid = 0;
        if (valid) {
            float invDev = 1.f / (pmax(fabsf(det), 0.000001f) * psign(det));
            float tvec[3] = {orig[0] - v0[0], orig[1] - v0[1], orig[2] - v0[2]};
            float u = dot3(tvec, pvec) * invDev;
            if (u < -0.00001f || u > 1.00001f) valid = 0;
            if (valid) {
                float qvec[3];
                cross3(tvec, e1, qvec);
                float v = dot3(dir, qvec) * invDev;
                if (v < -0.00001f || (u + v) > 1.00001f) valid = 0;
                if (valid) {
                    float t = dot3(e2, qvec) * invDev;
                    if (greaterEqualF(t, 0.0f) && valid) {
                        T = t;
                        *U = u;
                        *V = v;
                    }
                }
            }
        }
    }
    return T;
}

typedef struct {
    float predist;
    int triangleID;
    int bakedCount;
    psmo_hit baked[PSMO_BAKED_CAP];
} tstate;

static void bake_push(tstate* st, float u, float v, float t, int tri, psmo_counters* ctr) {
    int at = st->bakedCount++;
    if (at < PSMO_BAKED_CAP) {
        st->baked[at].u = u; st->baked[at].v = v; st->baked[at].t = t; st->baked[at].tri = tri;
    } else if (ctr) {
        ctr->baked_drops++; /* reference writes out of bounds here (:294); canonical = drop */
    }
}

/* directTraverse.comp:261-309, non-AMD branch (single-triangle test) */
static void testIntersectionPacked(tstate* st, const float* tris, const float orig[3],
                                   const float dir[3], int tx, int ty, int vx, int vy,
                                   psmo_counters* ctr) {
    int validx = (tx >= 0) && (tx != -1) && (tx != st->triangleID) && vx;
    int validy = (ty >= 0) && (ty != -1) && (ty != st->triangleID) && vy;
    validy = validy && (tx != ty);
    if (!validx) {
        int t = tx; tx = ty; ty = t;
        t = validx; validx = validy; validy = t;
    }
    if (validx || validy) {
        float ux = 0.f, vxx = 0.f, uy = 0.f, vyy = 0.f;
        float dx = intersectTriangle(tris, orig, dir, tx, &ux, &vxx, validx, ctr);
        float dy = intersectTriangle(tris, orig, dir, ty, &uy, &vyy, validy, ctr);
        int near = validx && lessF(dx, PSMO_INFINITY) && lessEqualF(dx, st->predist) && greaterEqualF(dx, 0.0f);
        if (near) {
            if (!equalF(dx, st->predist)) st->bakedCount = 0;
            st->predist = dx;
            st->triangleID = tx;
            bake_push(st, ux, vxx, dx, tx, ctr);
        }
        near = validy && lessF(dy, PSMO_INFINITY) && lessEqualF(dy, st->predist) && greaterEqualF(dy, 0.0f);
        if (near) {
            if (!equalF(dy, st->predist)) st->bakedCount = 0;
            st->predist = dy;
            st->triangleID = ty;
            bake_push(st, uy, vyy, dy, ty, ctr);
        }
    }
}

/* directTraverse.comp:74-112 */
static void reorderTriangles(tstate* st) {
    if (st->bakedCount > PSMO_BAKED_CAP) st->bakedCount = PSMO_BAKED_CAP;
    int n = st->bakedCount;
    for (int iround = 1; iround < n; iround++) {
        for (int index = 0; index < n - iround; index++) {
            psmo_hit a = st->baked[index], b = st->baked[index + 1];
            int lessIdx = a.tri <= b.tri;
            int deeper = lessF(a.t, b.t);
            if (lessIdx || deeper) { st->baked[index] = b; st->baked[index + 1] = a; }
        }
    }
    int clean = 0;
    for (int iround = 0; iround < PSMO_BAKED_CAP; iround++) {
        if (iround >= n - 1) break;
        if (st->baked[iround + 1].tri != st->baked[iround].tri) st->baked[clean++] = st->baked[iround];
    }
    if (n > 0 && clean <= PSMO_BAKED_CAP) st->baked[clean++] = st->baked[n - 1];
    st->bakedCount = clean;
}

/* directTraverse.comp:333-484 for a fresh ray (ray.hit == -1). Returns the chain length;
 * out[0] is the head of the chain ("hit triangle index"). */
int psmo_traverse(const psmo_node* nodes, const float* tris, const float M[16],
                  const float origin[3], const float direct_in[3], psmo_hit out[PSMO_BAKED_CAP],
                  psmo_counters* ctr) {
    return psmo_traverse_from(nodes, tris, M, origin, direct_in, PSMO_INFINITY, out, ctr);
}

/* traverse() entered with the distance of the chain the ray already carries (:335-336, multi-BVH) */
int psmo_traverse_from(const psmo_node* nodes, const float* tris, const float M[16],
                       const float origin[3], const float direct_in[3], float start_dist,
                       psmo_hit out[PSMO_BAKED_CAP], psmo_counters* ctr) {
    tstate st;
    st.predist = start_dist;
    st.triangleID = -1;
    st.bakedCount = 0;
    for (int i = 0; i < PSMO_BAKED_CAP; i++) { st.baked[i].u = 0; st.baked[i].v = 0; st.baked[i].t = PSMO_INFINITY; st.baked[i].tri = -1; }
    int deferredStack[PSMO_STACK_CAP];
    int deferredPtr = 0;
    deferredStack[0] = -1;

    float direct[3];
    normalize3(direct_in, direct); /* :350 */

    float o4[4] = {origin[0], origin[1], origin[2], 1.0f};
    float d4[4] = {direct[0], direct[1], direct[2], 1.0f};
    float torig4[4], tdir4[4];
    mat_vec(M, o4, torig4);   /* :353 projectVoxels */
    matT_vec(M, d4, tdir4);   /* :354 */
    float dirlen = len3(tdir4) / pmax(len3(direct), 0.000001f);
    float dirlenInv = 1.f / pmax(dirlen, 0.000001f);
    float dirproj[3];
    normalize3(tdir4, dirproj);

    float near = PSMO_INFINITY, far = PSMO_INFINITY;
    const float cmn[3] = {-0.00001f, -0.00001f, -0.00001f};
    const float cmx[3] = {1.00001f, 1.00001f, 1.00001f};
    float d = intersectCubeSingle(torig4, dirproj, cmn, cmx, &near, &far);
    float toffset = smax(near, 0.f);
    float origined[3], divident[3];
    for (int k = 0; k < 3; k++) {
        origined[k] = torig4[k] + dirproj[k] * toffset;
        divident[k] = 1.f / dirproj[k];
    }

    int idx = 0, found = -1;
    int validBox = lessF(d, PSMO_INFINITY) && lessF(d * dirlenInv, PSMO_INFINITY) && greaterEqualF(d, 0.0f);
    psmo_node node = nodes[idx];
    int skipUpstream = 0;
    const float IP = PSMO_INFINITY - PSMO_PZERO;
    int i;
    for (i = 0; i < PSMO_MAX_ITERS; i++) {
        if (!validBox) break;
        int notLeaf = node.pdata[0] != node.pdata[1];
        if (notLeaf) {
            if (ctr) ctr->node_visits++;
            const psmo_node* L = &nodes[node.pdata[0]];
            const psmo_node* R = &nodes[node.pdata[1]];
            float lmn[4], lmx[4], rmn[4], rmx[4];
            unpack_half4(&L->box[0], lmn); unpack_half4(&L->box[2], lmx);
            unpack_half4(&R->box[0], rmn); unpack_half4(&R->box[2], rmx);
            float nears[2], fars[2], hits[2];
            hits[0] = intersectCubeChild(origined, divident, lmn, lmx, &nears[0], &fars[0]);
            hits[1] = intersectCubeChild(origined, divident, rmn, rmx, &nears[1], &fars[1]);
            int leftNear = lessEqualF(nears[0], nears[1]);
            int og[2];
            for (int c = 0; c < 2; c++) {
                og[c] = (hits[c] <= IP) && (hits[c] * dirlenInv <= IP) && (hits[c] > -PSMO_PZERO) &&
                        (nears[c] <= IP) && (nears[c] * dirlenInv <= IP) &&
                        (((nears[c] + toffset) * dirlenInv - PSMO_PZERO) <= st.predist) &&
                        (node.pdata[0] != -1) && (node.pdata[1] != -1);
                /* the esc terms of :429 are vestigial */
            }
            int lp[4] = {-1, -1, -1, -1}, rp[4] = {-1, -1, -1, -1};
            if (og[0]) memcpy(lp, L->pdata, sizeof(lp));
            if (og[1]) memcpy(rp, R->pdata, sizeof(rp));
            int ov[2] = {og[0] && (lp[0] != lp[1]), og[1] && (rp[0] != rp[1])}; /* nodes */
            int lf[2] = {og[0] && (lp[0] == lp[1]), og[1] && (rp[0] == rp[1])}; /* leafs */
            if (lf[0] || lf[1]) {
                int leftOrder = (lf[0] && lf[1]) ? leftNear : lf[0];
                if (leftOrder)
                    testIntersectionPacked(&st, tris, origin, direct, lp[3], rp[3], lf[0], lf[1], ctr);
                else
                    testIntersectionPacked(&st, tris, origin, direct, rp[3], lp[3], lf[1], lf[0], ctr);
            }
            int anyOverlap = ov[0] || ov[1];
            if (anyOverlap) {
                int leftOrder = (ov[0] && ov[1]) ? leftNear : ov[0];
                int lr0 = ov[0] ? node.pdata[0] : -1;
                int lr1 = ov[1] ? node.pdata[1] : -1;
                if (!leftOrder) { int t = lr0; lr0 = lr1; lr1 = t; }
                if (lr1 != -1 && lr0 != lr1) {
                    if (deferredPtr < PSMO_STACK_CAP) deferredStack[deferredPtr++] = lr1;
                    else if (ctr) ctr->stack_drops++;
                }
                found = lr0;
            }
            skipUpstream = skipUpstream || anyOverlap;
        }
        /* :467-476 */
        {
            int ptr = skipUpstream ? deferredPtr : --deferredPtr;
            idx = ptr >= 0 ? (skipUpstream ? found : deferredStack[ptr]) : -1;
            validBox = validBox && idx >= 0 && ptr >= 0;
            if (validBox) node = nodes[idx];
        }
        skipUpstream = 0;
    }
    if (i >= PSMO_MAX_ITERS && validBox && ctr) ctr->iter_caps++;
    reorderTriangles(&st);
    for (int k = 0; k < st.bakedCount; k++) out[k] = st.baked[k];
    return st.bakedCount;
}

#ifdef _OPENMP
#include <omp.h>
#endif

void psmo_traverse_batch_ex(const psmo_node* nodes, const float* tris, const float M[16],
                            const float* origins, const float* directs, int nrays, psmo_hit* hits,
                            int32_t* counts, psmo_counters* ctr, int nthreads, uint32_t* per_ray_visits,
                            uint32_t* per_ray_tests);

void psmo_traverse_batch(const psmo_node* nodes, const float* tris, const float M[16],
                         const float* origins, const float* directs, int nrays, psmo_hit* hits,
                         int32_t* counts, psmo_counters* ctr, int nthreads) {
    psmo_traverse_batch_ex(nodes, tris, M, origins, directs, nrays, hits, counts, ctr, nthreads, NULL, NULL);
}

/* same, plus optional per-ray visit / triangle-test counts (divergence studies) */
void psmo_traverse_batch_ex(const psmo_node* nodes, const float* tris, const float M[16],
                            const float* origins, const float* directs, int nrays, psmo_hit* hits,
                            int32_t* counts, psmo_counters* ctr, int nthreads, uint32_t* per_ray_visits,
                            uint32_t* per_ray_tests) {
    psmo_counters total;
    memset(&total, 0, sizeof(total));
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel
#endif
    {
        psmo_counters local;
        memset(&local, 0, sizeof(local));
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (int r = 0; r < nrays; r++) {
            psmo_hit tmp[PSMO_BAKED_CAP];
            uint64_t v0 = local.node_visits, t0 = local.tri_tests;
            int c = psmo_traverse(nodes, tris, M, &origins[3 * r], &directs[3 * r], tmp, &local);
            counts[r] = c;
            if (per_ray_visits) per_ray_visits[r] = (uint32_t)(local.node_visits - v0);
            if (per_ray_tests) per_ray_tests[r] = (uint32_t)(local.tri_tests - t0);
            if (hits) {
                for (int k = 0; k < PSMO_BAKED_CAP; k++) {
                    if (k < c) hits[(size_t)r * PSMO_BAKED_CAP + k] = tmp[k];
                    else { psmo_hit z = {0.f, 0.f, PSMO_INFINITY, -1}; hits[(size_t)r * PSMO_BAKED_CAP + k] = z; }
                }
            }
        }
#ifdef _OPENMP
#pragma omp critical
#endif
        {
            total.node_visits += local.node_visits;
            total.tri_tests += local.tri_tests;
            total.stack_drops += local.stack_drops;
            total.iter_caps += local.iter_caps;
            total.baked_drops += local.baked_drops;
        }
    }
    if (ctr) *ctr = total;
}

/* A further intersection() of the same rays with another hierarchy (directTraverse.comp:219-249,335-346,
 * 497-508): the search starts at the head distance of the existing chain; the hits it bakes overwrite the
 * front of that chain and what is left of the old chain stays linked behind them. Triangle ids are recorded
 * as tri_base + local id so one concatenated triangle array serves the shading stage. The reference's
 * "hit index 0 counts as no chain" slip (hid > 0, :225,229) is not reproduced. */
void psmo_traverse_chain_batch(const psmo_node* nodes, const float* tris, const float M[16],
                               const float* origins, const float* directs, int nrays, psmo_hit* hits,
                               int32_t* counts, int tri_base, psmo_counters* ctr, int nthreads) {
    psmo_counters total;
    memset(&total, 0, sizeof(total));
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel
#endif
    {
        psmo_counters local;
        memset(&local, 0, sizeof(local));
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (int r = 0; r < nrays; r++) {
            psmo_hit tmp[PSMO_BAKED_CAP];
            psmo_hit* chain = &hits[(size_t)r * PSMO_BAKED_CAP];
            float start = counts[r] > 0 ? chain[0].t : PSMO_INFINITY;
            int k = psmo_traverse_from(nodes, tris, M, &origins[3 * r], &directs[3 * r], start, tmp, &local);
            for (int j = 0; j < k; j++) { chain[j] = tmp[j]; chain[j].tri += tri_base; }
            if (k > counts[r]) counts[r] = k;
        }
#ifdef _OPENMP
#pragma omp critical
#endif
        {
            total.node_visits += local.node_visits;
            total.tri_tests += local.tri_tests;
            total.stack_drops += local.stack_drops;
            total.iter_caps += local.iter_caps;
            total.baked_drops += local.baked_drops;
        }
    }
    if (ctr) *ctr = total;
}

/* Independent cross-check (not a restatement): closest triangle by testing all of them
 * with the same triangle test, smallest t wins. */
int psmo_brute_force(const float* tris, int ntris, const float origin[3], const float direct_in[3],
                     psmo_hit* best) {
    float direct[3];
    normalize3(direct_in, direct);
    int found = 0;
    best->t = PSMO_INFINITY; best->tri = -1; best->u = 0; best->v = 0;
    for (int t = 0; t < ntris; t++) {
        float u = 0, v = 0;
        float T = intersectTriangle(tris, origin, direct, t, &u, &v, 1, NULL);
        if (T < PSMO_INFINITY - PSMO_PZERO && T < best->t) {
            best->t = T; best->tri = t; best->u = u; best->v = v;
            found = 1;
        }
    }
    return found;
}
