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

// ---- double 4x4 helpers for the fit transform (TriangleHierarchy.inl:257-267) ---------------
PSM_HD void inverse4d(const double* m, double* o) {
    double a00 = m[0], a01 = m[1], a02 = m[2], a03 = m[3];
    double a10 = m[4], a11 = m[5], a12 = m[6], a13 = m[7];
    double a20 = m[8], a21 = m[9], a22 = m[10], a23 = m[11];
    double a30 = m[12], a31 = m[13], a32 = m[14], a33 = m[15];
    double b00 = a00 * a11 - a01 * a10, b01 = a00 * a12 - a02 * a10;
    double b02 = a00 * a13 - a03 * a10, b03 = a01 * a12 - a02 * a11;
    double b04 = a01 * a13 - a03 * a11, b05 = a02 * a13 - a03 * a12;
    double b06 = a20 * a31 - a21 * a30, b07 = a20 * a32 - a22 * a30;
    double b08 = a20 * a33 - a23 * a30, b09 = a21 * a32 - a22 * a31;
    double b10 = a21 * a33 - a23 * a31, b11 = a22 * a33 - a23 * a32;
    double det = b00 * b11 - b01 * b10 + b02 * b09 + b03 * b08 - b04 * b07 + b05 * b06;
    double id = 1.0 / det;
    o[0] = (a11 * b11 - a12 * b10 + a13 * b09) * id;
    o[1] = (-a01 * b11 + a02 * b10 - a03 * b09) * id;
    o[2] = (a31 * b05 - a32 * b04 + a33 * b03) * id;
    o[3] = (-a21 * b05 + a22 * b04 - a23 * b03) * id;
    o[4] = (-a10 * b11 + a12 * b08 - a13 * b07) * id;
    o[5] = (a00 * b11 - a02 * b08 + a03 * b07) * id;
    o[6] = (-a30 * b05 + a32 * b02 - a33 * b01) * id;
    o[7] = (a20 * b05 - a22 * b02 + a23 * b01) * id;
    o[8] = (a10 * b10 - a11 * b08 + a13 * b06) * id;
    o[9] = (-a00 * b10 + a01 * b08 - a03 * b06) * id;
    o[10] = (a30 * b04 - a31 * b02 + a33 * b00) * id;
    o[11] = (-a20 * b04 + a21 * b02 - a23 * b00) * id;
    o[12] = (-a10 * b09 + a11 * b07 - a12 * b06) * id;
    o[13] = (a00 * b09 - a01 * b07 + a02 * b06) * id;
    o[14] = (-a30 * b03 + a31 * b01 - a32 * b00) * id;
    o[15] = (a20 * b03 - a21 * b01 + a22 * b00) * id;
}
PSM_HD void mul4d(const double* a, const double* b, double* o) {
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double s = 0.0;
            for (int k = 0; k < 4; k++) s += a[4 * i + k] * b[4 * k + j];
            o[4 * i + j] = s;
        }
}

}  // namespace psm
