// psm_math.h -- canonical arithmetic for the gfx950 kernels (DESIGN.md "canonical arithmetic").
//
// Every function is one IEEE binary32 operation sequence in the order written; the library is
// compiled with -ffp-contract=off so nothing is fused except explicit fmaf().  The GLSL the
// reference is written in leaves min/max NaN behaviour, dot/normalize evaluation order and all
// transcendental precision implementation-defined; these definitions pin them.
#pragma once
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PSM_HD __host__ __device__ __forceinline__
#define PSM_D __device__ __forceinline__

namespace psm {

constexpr float PZERO = 0.0005f;      // include/constants.glsl:72
constexpr float INF = 10000.0f;       // include/constants.glsl:82
constexpr float GAP = PZERO * 2.f;    // include/shadinglib.glsl:8
constexpr int STACK_CAP = 16;         // directTraverse.comp:40-41
constexpr int BAKED_CAP = 8;          // directTraverse.comp:42
constexpr int MAX_ITERS = 8192;       // directTraverse.comp:383

PSM_HD uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
PSM_HD float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }

// GLSL min/max: min(x,y) = y<x ? y : x ; max(x,y) = x<y ? y : x
PSM_HD float pmin(float x, float y) { return (y < x) ? y : x; }
PSM_HD float pmax(float x, float y) { return (x < y) ? y : x; }
// slab tests: IEEE-754 minNum / maxNum (a NaN operand is ignored) with -0 < +0 = v_min_f32 / v_max_f32
__device__ __forceinline__ float sminf(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ float smaxf(float a, float b) { return __builtin_fmaxf(a, b); }
PSM_HD float pclamp(float x, float lo, float hi) { return pmin(pmax(x, lo), hi); }
PSM_HD float psign(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }
PSM_HD float pabs(float x) { return u2f(f2u(x) & 0x7fffffffu); }

struct v3 {
    float x, y, z;
};
PSM_HD v3 mk3(float x, float y, float z) { return v3{x, y, z}; }
PSM_HD v3 operator+(v3 a, v3 b) { return v3{a.x + b.x, a.y + b.y, a.z + b.z}; }
PSM_HD v3 operator-(v3 a, v3 b) { return v3{a.x - b.x, a.y - b.y, a.z - b.z}; }
PSM_HD v3 operator*(v3 a, v3 b) { return v3{a.x * b.x, a.y * b.y, a.z * b.z}; }
PSM_HD v3 operator*(v3 a, float s) { return v3{a.x * s, a.y * s, a.z * s}; }
PSM_HD float dot3(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
PSM_HD float len3(v3 a) { return sqrtf(dot3(a, a)); }
PSM_HD v3 normalize3(v3 a) {
    float inv = 1.0f / sqrtf(dot3(a, a));
    return v3{a.x * inv, a.y * inv, a.z * inv};
}
PSM_HD v3 cross3(v3 a, v3 b) {
    return v3{a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
PSM_HD v3 fma3(v3 a, float s, v3 c) { return v3{fmaf(a.x, s, c.x), fmaf(a.y, s, c.y), fmaf(a.z, s, c.z)}; }
PSM_HD float mlength3(v3 c) { return pmax(c.x, pmax(c.y, c.z)); }
PSM_HD float mixf(float x, float y, float a) { return x * (1.0f - a) + y * a; }

// include/mathlib.glsl:10-14
PSM_HD bool lessEqualF(float a, float b) { return (b - a) > -PZERO; }
PSM_HD bool lessF(float a, float b) { return (b - a) >= PZERO; }
PSM_HD bool greaterEqualF(float a, float b) { return (a - b) > -PZERO; }
PSM_HD bool equalF(float a, float b) { return pabs(a - b) < PZERO; }

// row-major 4x4: (M v)[i] = ((m0 x + m1 y) + m2 z) + m3 w   (mult4(mat, vec), mathlib.glsl:78-81)
PSM_HD void mat_vec(const float* M, float x, float y, float z, float w, float* o) {
#pragma unroll
    for (int i = 0; i < 4; i++) o[i] = ((M[4 * i + 0] * x + M[4 * i + 1] * y) + M[4 * i + 2] * z) + M[4 * i + 3] * w;
}
// (M^T v)[i]   (mult4(vec, mat), mathlib.glsl:73-76)
PSM_HD void matT_vec(const float* M, float x, float y, float z, float w, float* o) {
#pragma unroll
    for (int i = 0; i < 4; i++) o[i] = ((M[0 + i] * x + M[4 + i] * y) + M[8 + i] * z) + M[12 + i] * w;
}

// ---- fp16 storage: packHalf2x16 = round-to-nearest-even, unpack exact --------------------
PSM_D uint32_t pack_half2(float a, float b) {
    __half ha = __float2half_rn(a), hb = __float2half_rn(b);
    return (uint32_t)__builtin_bit_cast(uint16_t, ha) | ((uint32_t)__builtin_bit_cast(uint16_t, hb) << 16);
}
PSM_D float half_lo(uint32_t p) { return __half2float(__builtin_bit_cast(__half, (uint16_t)(p & 0xffffu))); }
PSM_D float half_hi(uint32_t p) { return __half2float(__builtin_bit_cast(__half, (uint16_t)(p >> 16))); }

// fp16 bits <-> unsigned sortable key with -0 < +0 (order-independent refit)
PSM_HD uint32_t half2_to_key(uint32_t p) {
    uint32_t s = p & 0x80008000u;
    uint32_t m = (s >> 15) * 0xffffu;  // 0xffff per negative half
    return p ^ (m | 0x80008000u);
}
PSM_HD uint32_t key_to_half2(uint32_t k) {
    uint32_t s = (~k) & 0x80008000u;   // key top bit clear => negative
    uint32_t m = (s >> 15) * 0xffffu;
    return k ^ (m | 0x80008000u);
}

// ---- Morton, include/morton.glsl:37-51 -----------------------------------------------------
PSM_HD uint64_t part1by2_64(uint32_t a) {
    uint64_t x = a & 0x1fffffull;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}
PSM_HD uint64_t morton3_64(uint32_t x, uint32_t y, uint32_t z) {
    return part1by2_64(x) | (part1by2_64(y) << 1) | (part1by2_64(z) << 2);
}
PSM_HD int nlz64(uint64_t x) { return x == 0 ? 64 : __builtin_clzll(x); }

// ---- RNG, include/random.glsl:11-46 ---------------------------------------------------------
PSM_HD uint32_t hash32(uint32_t x) {
    x += (x << 10u);
    x ^= (x >> 6u);
    x += (x << 3u);
    x ^= (x >> 11u);
    x += (x << 15u);
    return x;
}
struct Rng {
    uint32_t smp, clocks, time5;
    PSM_HD float next() {
        uint32_t hs = clocks;
        clocks = hash32(clocks + 1u);
        uint32_t h = hash32(smp ^ hash32(hs) ^ hash32(time5));
        float f = u2f((h & 0x007FFFFFu) | 0x3F800000u);
        return f - 1.0f;  // fract() of a value in [1,2)
    }
};
PSM_HD uint32_t child_key(uint32_t pkey, uint32_t site) { return hash32(pkey ^ hash32(site)); }

// ---- pinned transcendental functions ---------------------------------------------------------
PSM_HD void sincos_reduce(float x, float& r, int& q) {
    int j = (int)(x * 1.27323954473516f);
    if (j & 1) j += 1;
    float y = (float)j;
    r = ((x - y * 0.78515625f) - y * 2.4187564849853515625e-4f) - y * 3.77489497744594108e-8f;
    q = (j >> 1) & 3;
}
PSM_HD float sin_poly(float r) {
    float z = r * r;
    return r + r * z * (-1.6666654611e-1f + z * (8.3321608736e-3f + z * -1.9515295891e-4f));
}
PSM_HD float cos_poly(float r) {
    float z = r * r;
    return (1.0f - 0.5f * z) + z * z * (4.166664568298827e-2f + z * (-1.388731625493765e-3f + z * 2.443315711809948e-5f));
}
PSM_HD float psin(float x) {
    float s = 1.0f;
    if (x < 0.0f) { s = -1.0f; x = -x; }
    float r; int q;
    sincos_reduce(x, r, q);
    float v = (q & 1) ? cos_poly(r) : sin_poly(r);
    if (q & 2) v = -v;
    return s * v;
}
PSM_HD float pcos(float x) {
    if (x < 0.0f) x = -x;
    float r; int q;
    sincos_reduce(x, r, q);
    float v = (q & 1) ? sin_poly(r) : cos_poly(r);
    if (q == 1 || q == 2) v = -v;
    return v;
}
PSM_HD float plog2(float x) {
    uint32_t b = f2u(x);
    int e = (int)((b >> 23) & 0xffu) - 127;
    float m = u2f((b & 0x7fffffu) | 0x3f800000u);
    if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
    float f = m - 1.0f;
    float z = f * f;
    float p = 7.0376836292e-2f;
    p = p * f + -1.1514610310e-1f;
    p = p * f + 1.1676998740e-1f;
    p = p * f + -1.2420140846e-1f;
    p = p * f + 1.4249322787e-1f;
    p = p * f + -1.6668057665e-1f;
    p = p * f + 2.0000714765e-1f;
    p = p * f + -2.4999993993e-1f;
    p = p * f + 3.3333331174e-1f;
    float y = f * z * p;
    y = y + -0.5f * z;
    float ln = f + y;
    return (float)e + ln * 1.44269504088896341f;
}
PSM_HD float pexp2(float t) {
    if (t < -125.0f) return 0.0f;
    if (t > 125.0f) t = 125.0f;
    float fl = floorf(t);
    int i = (int)fl;
    float fr = t - fl;
    if (fr > 0.5f) { i += 1; fr = fr - 1.0f; }
    float p = 1.535336188319500e-4f;
    p = p * fr + 1.339887440266574e-3f;
    p = p * fr + 9.618437357674640e-3f;
    p = p * fr + 5.550332471162809e-2f;
    p = p * fr + 2.402264791363012e-1f;
    p = p * fr + 6.931472028550421e-1f;
    float px = 1.0f + fr * p;
    return px * u2f((uint32_t)(i + 127) << 23);
}
PSM_HD float ppow(float x, float y) {
    if (!(x > 0.0f)) return 0.0f;
    return pexp2(y * plog2(x));
}

// atan / atan2 / asin for the equirect sky lookup (public/environment.glsl:23-26)
PSM_HD float patan_pos(float x) {  // x >= 0
    float y0 = 0.0f;
    if (x > 2.414213562373095f) { y0 = 1.5707963267948966f; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y0 = 0.7853981633974483f; x = (x - 1.0f) / (x + 1.0f); }
    float z = x * x;
    float p = 8.05374449538e-2f;
    p = p * z + -1.38776856032e-1f;
    p = p * z + 1.99777106478e-1f;
    p = p * z + -3.33329491539e-1f;
    return y0 + (p * z * x + x);
}
PSM_HD float patan2(float y, float x) {
    const float PI_F = 3.14159265358979323846f;
    if (x == 0.0f && y == 0.0f) return 0.0f;
    float ax = pabs(x), ay = pabs(y);
    float a = (ax == 0.0f) ? 1.5707963267948966f : patan_pos(ay / ax);
    if (x < 0.0f) a = PI_F - a;
    return (y < 0.0f) ? -a : a;
}
PSM_HD float pasin(float x) {
    float c = pclamp(x, -1.0f, 1.0f);
    return patan2(c, sqrtf((1.0f - c) * (1.0f + c)));
}

// ---- double 4x4 helpers for the fit transform (TriangleHierarchy.inl:226-232,257-267) ----------------------
// The reference evaluates these formulas on the host with glm in double. gmat = glm's dmat4 memory (column-major,
// g[4c + r]); gm_mul / gm_inverse follow glm's operation order (type_mat4x4.inl operator*, func_matrix.inl
// compute_inverse<4,4>) so the float matrix the kernels read is bit for bit the one the reference uploads.
PSM_HD void gm_mul(const double* a, const double* b, double* o) {
    double t[16];
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++)
            t[4 * c + r] = ((a[r] * b[4 * c] + a[4 + r] * b[4 * c + 1]) + a[8 + r] * b[4 * c + 2]) + a[12 + r] * b[4 * c + 3];
    for (int i = 0; i < 16; i++) o[i] = t[i];
}
PSM_HD void gm_inverse(const double* m, double* o) {
    double c00 = m[10] * m[15] - m[14] * m[11], c02 = m[6] * m[15] - m[14] * m[7], c03 = m[6] * m[11] - m[10] * m[7];
    double c04 = m[9] * m[15] - m[13] * m[11], c06 = m[5] * m[15] - m[13] * m[7], c07 = m[5] * m[11] - m[9] * m[7];
    double c08 = m[9] * m[14] - m[13] * m[10], c10 = m[5] * m[14] - m[13] * m[6], c11 = m[5] * m[10] - m[9] * m[6];
    double c12 = m[8] * m[15] - m[12] * m[11], c14 = m[4] * m[15] - m[12] * m[7], c15 = m[4] * m[11] - m[8] * m[7];
    double c16 = m[8] * m[14] - m[12] * m[10], c18 = m[4] * m[14] - m[12] * m[6], c19 = m[4] * m[10] - m[8] * m[6];
    double c20 = m[8] * m[13] - m[12] * m[9], c22 = m[4] * m[13] - m[12] * m[5], c23 = m[4] * m[9] - m[8] * m[5];
    const double f0[4] = {c00, c00, c02, c03}, f1[4] = {c04, c04, c06, c07}, f2[4] = {c08, c08, c10, c11};
    const double f3[4] = {c12, c12, c14, c15}, f4[4] = {c16, c16, c18, c19}, f5[4] = {c20, c20, c22, c23};
    const double v0[4] = {m[4], m[0], m[0], m[0]}, v1[4] = {m[5], m[1], m[1], m[1]};
    const double v2[4] = {m[6], m[2], m[2], m[2]}, v3[4] = {m[7], m[3], m[3], m[3]};
    double inv[16];
    for (int k = 0; k < 4; k++) {
        const double sa = (k & 1) ? -1.0 : 1.0, sb = -sa;
        inv[k] = ((v1[k] * f0[k] - v2[k] * f1[k]) + v3[k] * f2[k]) * sa;
        inv[4 + k] = ((v0[k] * f0[k] - v2[k] * f3[k]) + v3[k] * f4[k]) * sb;
        inv[8 + k] = ((v0[k] * f1[k] - v1[k] * f3[k]) + v3[k] * f5[k]) * sa;
        inv[12 + k] = ((v0[k] * f2[k] - v1[k] * f4[k]) + v2[k] * f5[k]) * sb;
    }
    double one_over_det = 1.0 / ((m[0] * inv[0] + m[1] * inv[4]) + (m[2] * inv[8] + m[3] * inv[12]));
    for (int i = 0; i < 16; i++) o[i] = inv[i] * one_over_det;
}
PSM_HD void gm_identity(double* g) {
    for (int i = 0; i < 16; i++) g[i] = (i % 5 == 0) ? 1.0 : 0.0;
}
// row-major double[16] (the C ABI's convention) -> glm memory, and glm memory -> the float matrix uploaded row-major
PSM_HD void gm_from_rowmajor(const double* r, double* g) {
    for (int c = 0; c < 4; c++) for (int q = 0; q < 4; q++) g[4 * c + q] = r[4 * q + c];
}
// dmat4 mat(1.0); mat *= inverse(optimization)   (TriangleHierarchy.inl:229-230)
PSM_HD void gm_first_pass(const double* opt_rowmajor, double* mat) {
    double go[16], gi[16];
    gm_from_rowmajor(opt_rowmajor, go);
    gm_inverse(go, gi);
    gm_identity(mat);
    gm_mul(mat, gi, mat);
}
// dmat4 mat(1.0); mat *= inverse(translate(dvec3(offset)) * scale(dvec3(scale))); mat *= inverse(dmat4(optimization))
PSM_HD void gm_fit(const float* scale, const float* offset, const double* opt_rowmajor, double* mat) {
    double I[16], T[16], S[16], TS[16], iTS[16], go[16], iopt[16];
    gm_identity(I);
    for (int i = 0; i < 16; i++) T[i] = I[i];
    for (int r = 0; r < 4; r++)  // translate(m, v): Result[3] = m[0] v0 + m[1] v1 + m[2] v2 + m[3]
        T[12 + r] = ((I[r] * (double)offset[0] + I[4 + r] * (double)offset[1]) + I[8 + r] * (double)offset[2]) + I[12 + r];
    for (int r = 0; r < 4; r++) {  // scale(m, v): Result[i] = m[i] v[i]
        S[r] = I[r] * (double)scale[0];
        S[4 + r] = I[4 + r] * (double)scale[1];
        S[8 + r] = I[8 + r] * (double)scale[2];
        S[12 + r] = I[12 + r];
    }
    gm_mul(T, S, TS);
    gm_inverse(TS, iTS);
    gm_from_rowmajor(opt_rowmajor, go);
    gm_inverse(go, iopt);
    gm_identity(mat);
    gm_mul(mat, iTS, mat);
    gm_mul(mat, iopt, mat);
}

}  // namespace psm
