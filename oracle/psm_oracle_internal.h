/* psm_oracle_internal.h -- canonical scalar helpers shared by the oracle sources.
 * TEST INFRASTRUCTURE ONLY (see psm_oracle.h). */
#ifndef PSM_ORACLE_INTERNAL_H
#define PSM_ORACLE_INTERNAL_H
#include "psm_oracle.h"
#include <math.h>
#include <string.h>

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* GLSL min/max: min(x,y) = y<x ? y : x ; max(x,y) = x<y ? y : x */
static inline float pmin(float x, float y) { return (y < x) ? y : x; }
static inline float pmax(float x, float y) { return (x < y) ? y : x; }
/* slab tests: IEEE-754 minNum / maxNum (a NaN operand is ignored) with -0 < +0 -- what the
 * v_min_f32 / v_max_f32 of the reference's own target hardware compute (GLSL leaves NaN undefined) */
static inline float smin(float x, float y) {
    if (x != x) return y;
    if (y != y) return x;
    if (x == y) return signbit(x) ? x : y;
    return (y < x) ? y : x;
}
static inline float smax(float x, float y) {
    if (x != x) return y;
    if (y != y) return x;
    if (x == y) return signbit(x) ? y : x;
    return (x < y) ? y : x;
}
static inline float pclamp(float x, float lo, float hi) { return pmin(pmax(x, lo), hi); }
static inline float psign(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }
static inline float dot3(const float* a, const float* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
static inline float len3(const float* a) { return sqrtf(dot3(a, a)); }
static inline void normalize3(const float* a, float* o) {
    float inv = 1.0f / sqrtf(dot3(a, a));
    o[0] = a[0] * inv; o[1] = a[1] * inv; o[2] = a[2] * inv;
}
static inline void cross3(const float* a, const float* b, float* o) {
    float x = a[1] * b[2] - b[1] * a[2];
    float y = a[2] * b[0] - b[2] * a[0];
    float z = a[0] * b[1] - b[0] * a[1];
    o[0] = x; o[1] = y; o[2] = z;
}

/* include/mathlib.glsl:10-14 */
static inline int lessEqualF(float a, float b) { return (b - a) > -PSMO_PZERO; }
static inline int lessF(float a, float b) { return (b - a) >= PSMO_PZERO; }
static inline int greaterEqualF(float a, float b) { return (a - b) > -PSMO_PZERO; }
static inline int equalF(float a, float b) { return fabsf(a - b) < PSMO_PZERO; }

/* mult4(mat, vec) = M v, include/mathlib.glsl:78-81 with the host's transposed upload
 * (TriangleHierarchy.inl:265). Canonical order ((m0 x + m1 y) + m2 z) + m3 w. */
static inline void mat_vec(const float M[16], const float v[4], float o[4]) {
    for (int i = 0; i < 4; i++)
        o[i] = ((M[4 * i + 0] * v[0] + M[4 * i + 1] * v[1]) + M[4 * i + 2] * v[2]) + M[4 * i + 3] * v[3];
}
/* mult4(vec, mat) = M^T v, include/mathlib.glsl:73-76 */
static inline void matT_vec(const float M[16], const float v[4], float o[4]) {
    for (int i = 0; i < 4; i++)
        o[i] = ((M[0 + i] * v[0] + M[4 + i] * v[1]) + M[8 + i] * v[2]) + M[12 + i] * v[3];
}

/* packHalf2(vec4) -> uvec2, include/mathlib.glsl:332-334 */
static inline void pack_half4(const float v[4], uint32_t o[2]) {
    o[0] = (uint32_t)psmo_f32_to_f16(v[0]) | ((uint32_t)psmo_f32_to_f16(v[1]) << 16);
    o[1] = (uint32_t)psmo_f32_to_f16(v[2]) | ((uint32_t)psmo_f32_to_f16(v[3]) << 16);
}
static inline void unpack_half4(const uint32_t p[2], float o[4]) {
    o[0] = psmo_f16_to_f32((uint16_t)(p[0] & 0xffffu));
    o[1] = psmo_f16_to_f32((uint16_t)(p[0] >> 16));
    o[2] = psmo_f16_to_f32((uint16_t)(p[1] & 0xffffu));
    o[3] = psmo_f16_to_f32((uint16_t)(p[1] >> 16));
}

#endif
