/*
 * psm_oracle_shade.c -- CPU restatement of the wavefront loop (camera, surface+shade, sample).
 * See psm_oracle.h: TEST INFRASTRUCTURE ONLY, PARITY UNPINNED BY THE REFERENCE.
 *
 * Reference files restated (paths relative to /root/reference):
 *   ShadersSDK/raytracing/{camera,surface,rayshading,sampler,deinterlace,filter}.comp
 *   ShadersSDK/include/{random,rayslib,shadinglib,structs}.glsl, public/environment.glsl
 *   ShadersSDK/raytracing/directTraverse.comp:116-217 (interpolateMeshData)
 *   Include/Prismarine/Pipeline.inl:251-436
 *
 * Canonical rules (SURVEY a-12..a-16):
 *   - RNG stream id of a ray is its path key (pkey), not its queue position
 *   - next queue = for each input ray in order: [current, diffuse, reflection, shadow]
 *   - radiance is summed per texel directly (no colour-chain linked lists)
 *   - transcendental functions are the fixed polynomials below (GLSL leaves their
 *     precision implementation-defined; pinning them makes CPU and GPU bit-identical)
 *   - texture-less materials only (SURVEY a-14); sky is a constant colour
 */
#include "psm_oracle_internal.h"

#include <stdlib.h>

#define GAP (PSMO_PZERO * 2.f) /* shadinglib.glsl:8 */
#define TWO_PI_F 6.2831853071795864769252867665590057683943f
#define SQRT_OF_ONE_THIRD_F 0.5773502691896257645091487805019574556476f

/* ------------------------------------------------------------------ */
/* pinned transcendental functions                                     */
/* ------------------------------------------------------------------ */
static void sincos_reduce(float x, float* r, int* q) {
    int j = (int)(x * 1.27323954473516f); /* 4/pi */
    if (j & 1) j += 1;
    float y = (float)j;
    *r = ((x - y * 0.78515625f) - y * 2.4187564849853515625e-4f) - y * 3.77489497744594108e-8f;
    *q = (j >> 1) & 3;
}
static float sin_poly(float r) {
    float z = r * r;
    return r + r * z * (-1.6666654611e-1f + z * (8.3321608736e-3f + z * -1.9515295891e-4f));
}
static float cos_poly(float r) {
    float z = r * r;
    return (1.0f - 0.5f * z) + z * z * (4.166664568298827e-2f + z * (-1.388731625493765e-3f + z * 2.443315711809948e-5f));
}
float psmo_sinf(float x) {
    float s = 1.0f;
    if (x < 0.0f) { s = -1.0f; x = -x; }
    float r; int q;
    sincos_reduce(x, &r, &q);
    float v = (q & 1) ? cos_poly(r) : sin_poly(r);
    if (q & 2) v = -v;
    return s * v;
}
float psmo_cosf(float x) {
    if (x < 0.0f) x = -x;
    float r; int q;
    sincos_reduce(x, &r, &q);
    float v = (q & 1) ? sin_poly(r) : cos_poly(r);
    if (q == 1 || q == 2) v = -v;
    return v;
}
static float log2_pinned(float x) {
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
static float exp2_pinned(float t) {
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
float psmo_powf(float x, float y) {
    if (!(x > 0.0f)) return 0.0f;
    return exp2_pinned(y * log2_pinned(x));
}

static float atan_pos(float x) {
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
float psmo_atan2f(float y, float x) {
    const float PI_F = 3.14159265358979323846f;
    if (x == 0.0f && y == 0.0f) return 0.0f;
    float ax = fabsf(x), ay = fabsf(y);
    float a = (ax == 0.0f) ? 1.5707963267948966f : atan_pos(ay / ax);
    if (x < 0.0f) a = PI_F - a;
    return (y < 0.0f) ? -a : a;
}
float psmo_asinf(float x) {
    float c = pclamp(x, -1.0f, 1.0f);
    return psmo_atan2f(c, sqrtf((1.0f - c) * (1.0f + c)));
}

/* public/environment.glsl:23-26 readEnv: equirect lookup, RGBA8 texture, GL_LINEAR, clamp to edge
 * (Source/Examples/Application.hpp:46-54). Canonical bilinear filter: fp32 weights. */
static void read_env(const psmo_frame_cfg* cfg, const float* r, float out[3]) {
    if (!cfg->sky_tex || cfg->sky_w <= 0 || cfg->sky_h <= 0) { out[0] = cfg->sky[0]; out[1] = cfg->sky[1]; out[2] = cfg->sky[2]; return; }
    const float PI_F = 3.14159265358979323846f;
    float nr[3];
    normalize3(r, nr);
    float u = fmaf(psmo_atan2f(nr[2], nr[0]) / PI_F, 0.5f, 0.5f);
    float v = fmaf((psmo_asinf(nr[1]) * 2.0f) / PI_F, 0.5f, 0.5f);
    float x = u * (float)cfg->sky_w - 0.5f, y = v * (float)cfg->sky_h - 0.5f;
    float fx = floorf(x), fy = floorf(y);
    float a = x - fx, b = y - fy;
    int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    x0 = x0 < 0 ? 0 : (x0 > cfg->sky_w - 1 ? cfg->sky_w - 1 : x0);
    x1 = x1 < 0 ? 0 : (x1 > cfg->sky_w - 1 ? cfg->sky_w - 1 : x1);
    y0 = y0 < 0 ? 0 : (y0 > cfg->sky_h - 1 ? cfg->sky_h - 1 : y0);
    y1 = y1 < 0 ? 0 : (y1 > cfg->sky_h - 1 ? cfg->sky_h - 1 : y1);
    const uint8_t* t = cfg->sky_tex;
    for (int c = 0; c < 3; c++) {
        float t00 = (float)t[4 * (y0 * cfg->sky_w + x0) + c] / 255.0f, t10 = (float)t[4 * (y0 * cfg->sky_w + x1) + c] / 255.0f;
        float t01 = (float)t[4 * (y1 * cfg->sky_w + x0) + c] / 255.0f, t11 = (float)t[4 * (y1 * cfg->sky_w + x1) + c] / 255.0f;
        float top = t00 * (1.0f - a) + t10 * a, bot = t01 * (1.0f - a) + t11 * a;
        out[c] = top * (1.0f - b) + bot * b;
    }
}

/* ------------------------------------------------------------------ */
/* RNG, include/random.glsl:11-46                                      */
/* ------------------------------------------------------------------ */
uint32_t psmo_hash(uint32_t x) {
    x += (x << 10u);
    x ^= (x >> 6u);
    x += (x << 3u);
    x ^= (x >> 11u);
    x += (x << 15u);
    return x;
}
typedef struct { uint32_t smp, clocks, time5; } rng_t;
static float rng_next(rng_t* g) {
    uint32_t hs = g->clocks;
    g->clocks = psmo_hash(g->clocks + 1u);
    uint32_t h = psmo_hash(g->smp ^ psmo_hash(hs) ^ psmo_hash(g->time5));
    float f = u2f((h & 0x007FFFFFu) | 0x3F800000u);
    return f - floorf(f);
}

/* host rand() stand-in (Pipeline.inl:282,426): the MSVC CRT LCG the reference's
 * Windows build ran on; made explicit so a seed pins the whole frame. */
uint32_t psmo_rand_next(uint32_t* state) {
    *state = *state * 214013u + 2531011u;
    return (*state >> 16) & 0x7fffu;
}

/* ray bitfield, include/structs.glsl:73-78 */
static inline int bf_get(int bf, int off, int bits) { return (bf >> off) & ((1 << bits) - 1); }
static inline int bf_set(int bf, int v, int off, int bits) {
    int mask = ((1 << bits) - 1) << off;
    return (bf & ~mask) | ((v << off) & mask);
}
#define R_ACTIVE(b) bf_get(b, 0, 1)
#define R_TYPE(b) bf_get(b, 1, 2)
#define R_DL(b) bf_get(b, 3, 1)
#define R_TARGET(b) bf_get(b, 4, 4)
#define R_BOUNCE(b) bf_get(b, 8, 4)
#define R_BASIS(b) bf_get(b, 12, 1)
#define S_ACTIVE(b, v) b = bf_set(b, v, 0, 1)
#define S_TYPE(b, v) b = bf_set(b, v, 1, 2)
#define S_DL(b, v) b = bf_set(b, v, 3, 1)
#define S_TARGET(b, v) b = bf_set(b, v, 4, 4)
#define S_BOUNCE(b, v) b = bf_set(b, v, 8, 4)
#define S_BASIS(b, v) b = bf_set(b, v, 12, 1)

static inline float mlength3(const float* c) { return pmax(c[0], pmax(c[1], c[2])); }
static inline float mixf(float x, float y, float a) { return x * (1.0f - a) + y * a; }

/* ------------------------------------------------------------------ */
/* camera, raytracing/camera.comp:22-101                               */
/* ------------------------------------------------------------------ */
/* camera over a list of row bands (band b covers rows band_y[b] .. band_y[b]+band_h[b]-1, height <= 8).
 * canonical queue order: bands in list order, 8-wide tiles left to right, row-major inside a tile */
static int camera_bands(const psmo_frame_cfg* cfg, const float camInv[16], const float projInv[16],
                        uint32_t time, const int* band_y, const int* band_h, int nbands, psmo_ray* rays,
                        float* texel_coord, float* texel_sum, int32_t* texel_flag) {
    int w = cfg->width;
    float invw = 1.0f / (float)cfg->width, invh = 1.0f / (float)cfg->height;
    int n = 0;
    for (int b = 0; b < nbands; b++)
    for (int rem = 0; rem < band_h[b] * w; rem++) {
        int by = band_y[b], bh = band_h[b];
        int tx = rem / (8 * bh);
        if (tx * 8 >= w) tx = (w - 1) / 8;
        int tw = (w - tx * 8) < 8 ? (w - tx * 8) : 8;
        int r2 = rem - tx * 8 * bh;
        int y = by + r2 / tw, x = tx * 8 + r2 % tw;
        int idx = y * w + x;
        rng_t g = {(uint32_t)idx, 0u, time << 5};
        float rx = rng_next(&g);
        float ry = rng_next(&g);
        float cx = ((float)x + pclamp(rx, 0.00001f, 0.99999f)) * invw;
        float cy = ((float)y + pclamp(ry, 0.00001f, 0.99999f)) * invh;
        texel_coord[2 * idx + 0] = cx;
        texel_coord[2 * idx + 1] = cy;
        float ndc[2] = {cx * 2.0f - 1.0f, cy * 2.0f - 1.0f};
        float pf[4] = {ndc[0], ndc[1], 0.999f, 1.0f}, pn[4] = {ndc[0], ndc[1], 0.0f, 1.0f};
        float t0[4], co[4], orig[4];
        mat_vec(projInv, pf, t0); mat_vec(camInv, t0, co);
        mat_vec(projInv, pn, t0); mat_vec(camInv, t0, orig);
        float cw = co[3], ow = orig[3];
        for (int k = 0; k < 4; k++) { co[k] = co[k] / cw; orig[k] = orig[k] / ow; }
        float dd[3] = {co[0] - orig[0], co[1] - orig[1], co[2] - orig[2]};
        float dir[3];
        normalize3(dd, dir);
        if (cfg->enable360 == 1) { /* camera.comp:48-59: equirect directions, two fixed quaternion turns */
            const float PI_F = 3.1415926535897932384626422832795028841971f;
            float picx = (ndc[0] * -1.f) * PI_F, picy = (ndc[1] * 0.5f) * PI_F;
            float v[3] = {psmo_cosf(picy) * psmo_cosf(picx), psmo_cosf(picy) * psmo_sinf(picx), psmo_sinf(picy)};
            const float axes[2][3] = {{0.f, 0.f, -1.f}, {1.f, 0.f, 0.f}};
            for (int q = 0; q < 2; q++) {
                float half = PI_F / 4.f;
                float sh = psmo_sinf(half), ch = psmo_cosf(half);
                float qv[3] = {axes[q][0] * sh, axes[q][1] * sh, axes[q][2] * sh};
                float c1[3], t[3], c2[3];
                cross3(v, qv, c1);
                for (int k = 0; k < 3; k++) t[k] = c1[k] + ch * v[k];
                cross3(t, qv, c2);
                for (int k = 0; k < 3; k++) v[k] = v[k] + 2.0f * c2[k];
            }
            float o4[4] = {0.f, 0.f, 0.f, 1.f}, d4[4] = {v[0], v[1], v[2], 0.f}, od[4];
            mat_vec(camInv, o4, orig);
            mat_vec(camInv, d4, od);
            for (int k = 0; k < 3; k++) dir[k] = od[k];
        }
        psmo_ray r;
        for (int k = 0; k < 3; k++) { r.origin[k] = orig[k]; r.direct[k] = dir[k]; r.color[k] = 1.0f; }
        int bf = 0;
        S_ACTIVE(bf, 1); S_TYPE(bf, 0); S_DL(bf, 0); S_BOUNCE(bf, 4); S_BASIS(bf, 1);
        /* pre-collect a zero sample so the texel counts (:99) */
        texel_sum[4 * idx + 0] = 0.f; texel_sum[4 * idx + 1] = 0.f;
        texel_sum[4 * idx + 2] = 0.f; texel_sum[4 * idx + 3] = 1.f;
        texel_flag[idx] = 1;
        /* createRayIdx -> createRayStrict: bounce-1 (rayslib.glsl:130-156,205-222) */
        S_BOUNCE(bf, R_BOUNCE(bf) - 1);
        r.bitfield = bf;
        r.texel = idx;
        r.pkey = (uint32_t)idx;
        rays[n++] = r;
    }
    return n;
}

/* rows [y0,y1) in bands of 8 from y0 */
int psmo_camera(const psmo_frame_cfg* cfg, const float camInv[16], const float projInv[16],
                uint32_t time, int y0, int y1, psmo_ray* rays, float* texel_coord,
                float* texel_sum, int32_t* texel_flag) {
    int nb = (y1 - y0 + 7) / 8;
    int* by = (int*)malloc(sizeof(int) * (size_t)(nb + 1));
    int* bh = (int*)malloc(sizeof(int) * (size_t)(nb + 1));
    for (int b = 0; b < nb; b++) { by[b] = y0 + 8 * b; bh[b] = (y1 - by[b]) < 8 ? (y1 - by[b]) : 8; }
    int n = camera_bands(cfg, camInv, projInv, time, by, bh, nb, rays, texel_coord, texel_sum, texel_flag);
    free(by); free(bh);
    return n;
}

/* The dealing of the 8-row bands of a tile-sharded frame (not in the reference, which has one GPU; the build's own
 * definition, csrc/psm_internal.h BandMap): band g belongs to rank pattern[g % P], P = sum of the weights, laid out by a
 * smooth weighted round-robin -- every step each rank's credit grows by its weight, the largest credit (lowest rank on
 * ties) takes the band and pays P. weights NULL = 1 each = g % world. Returns P (0: invalid). */
int psmo_band_pattern(int world, const uint32_t* weights, uint8_t pattern[64]) {
    if (world < 1 || world > 64) return 0;
    long long P = 0, cur[64] = {0};
    for (int r = 0; r < world; r++) P += weights ? weights[r] : 1u;
    if (P < 1 || P > 64) return 0;
    for (int p = 0; p < (int)P; p++) {
        int pick = 0;
        for (int r = 0; r < world; r++) {
            cur[r] += weights ? weights[r] : 1u;
            if (cur[r] > cur[pick]) pick = r;
        }
        cur[pick] -= P;
        pattern[p] = (uint8_t)pick;
    }
    return (int)P;
}

/* sharding by bands: rank owns the global 8-row bands the dealing gives it */
int psmo_camera_weighted(const psmo_frame_cfg* cfg, const float camInv[16], const float projInv[16],
                         uint32_t time, int rank, int world, const uint32_t* weights, psmo_ray* rays, float* texel_coord,
                         float* texel_sum, int32_t* texel_flag) {
    int h = cfg->height;
    int ng = (h + 7) / 8, nb = 0;
    uint8_t pat[64];
    int P = psmo_band_pattern(world, weights, pat);
    if (P == 0) return -1;
    int* by = (int*)malloc(sizeof(int) * (size_t)(ng + 1));
    int* bh = (int*)malloc(sizeof(int) * (size_t)(ng + 1));
    for (int g = 0; g < ng; g++)
        if (pat[g % P] == rank) { by[nb] = 8 * g; bh[nb] = (h - 8 * g) < 8 ? (h - 8 * g) : 8; nb++; }
    int n = camera_bands(cfg, camInv, projInv, time, by, bh, nb, rays, texel_coord, texel_sum, texel_flag);
    free(by); free(bh);
    return n;
}

/* round-robin dealing: rank owns the global 8-row bands g with g % world == rank */
int psmo_camera_interleaved(const psmo_frame_cfg* cfg, const float camInv[16], const float projInv[16],
                            uint32_t time, int rank, int world, psmo_ray* rays, float* texel_coord,
                            float* texel_sum, int32_t* texel_flag) {
    return psmo_camera_weighted(cfg, camInv, projInv, time, rank, world, NULL, rays, texel_coord, texel_sum, texel_flag);
}

/* ------------------------------------------------------------------ */
/* surface: interpolateMeshData (directTraverse.comp:116-217) + surface.comp:165-195 */
/* ------------------------------------------------------------------ */
typedef struct {
    float uvt[4];
    float normal_trav[3]; /* hit.normalHeight as written by traverse */
    float normal[3];      /* after surface.comp */
    float albedo[4], emission[4], mr[4];
    int active;           /* HitActived */
} surf_t;

/* validateTexture, surface.comp:81-83 */
static int valid_tex(const psmo_frame_cfg* cfg, uint32_t binding) {
    return binding != 0u && binding != 0xFFFFFFFFu && binding < 32u && cfg->textures[binding].rgba8 &&
           cfg->textures[binding].w > 0;
}

/* fetchTexture, surface.comp:85-95: RGBA8 unorm, GL_LINEAR + GL_REPEAT, fp32 weights; non-finite reads 0 */
static void fetch_tex(const psmo_texture* t, float u, float v, int ox, int oy, float* o) {
    float uu = u + (float)ox / (float)t->w, vv = v + (float)oy / (float)t->h;
    if (!(fabsf(uu) < INFINITY) || !(fabsf(vv) < INFINITY)) { o[0] = o[1] = o[2] = o[3] = 0.f; return; }
    uu = uu - floorf(uu); vv = vv - floorf(vv);
    float x = uu * (float)t->w - 0.5f, y = vv * (float)t->h - 0.5f;
    float fx = floorf(x), fy = floorf(y);
    float a = x - fx, b = y - fy;
    int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    x0 = x0 < 0 ? x0 + t->w : x0; y0 = y0 < 0 ? y0 + t->h : y0;
    x1 = x1 >= t->w ? x1 - t->w : x1; y1 = y1 >= t->h ? y1 - t->h : y1;
    x0 = x0 >= t->w ? x0 - t->w : x0; y0 = y0 >= t->h ? y0 - t->h : y0;
    const uint8_t* p00 = t->rgba8 + 4 * ((size_t)y0 * t->w + x0), *p10 = t->rgba8 + 4 * ((size_t)y0 * t->w + x1);
    const uint8_t* p01 = t->rgba8 + 4 * ((size_t)y1 * t->w + x0), *p11 = t->rgba8 + 4 * ((size_t)y1 * t->w + x1);
    for (int c = 0; c < 4; c++) {
        float t00 = (float)p00[c] / 255.0f, t10 = (float)p10[c] / 255.0f;
        float t01 = (float)p01[c] / 255.0f, t11 = (float)p11[c] / 255.0f;
        float top = t00 * (1.0f - a) + t10 * a, bot = t01 * (1.0f - a) + t11 * a;
        o[c] = top * (1.0f - b) + bot * b;
    }
}

static void surface_eval(const psmo_frame_cfg* cfg, const psmo_material* mats, const int32_t* tri_mats,
                         const float* tris, const float* normals, const psmo_hit* h, surf_t* s) {
    int tri = h->tri;
    s->uvt[0] = h->u; s->uvt[1] = h->v; s->uvt[2] = h->t; s->uvt[3] = 0.f;
    const float* v0 = &tris[9 * tri], *v1 = v0 + 3, *v2 = v0 + 6;
    const float* n0 = &normals[9 * tri], *n1 = n0 + 3, *n2 = n0 + 6;
    float vs[3] = {(1.0f - h->u) - h->v, h->u, h->v};
    float d1[3] = {v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2]};
    float d2[3] = {v2[0] - v0[0], v2[1] - v0[1], v2[2] - v0[2]};
    float cr[3], nor[3], nrm[3];
    cross3(d1, d2, cr);
    normalize3(cr, nor);
    for (int k = 0; k < 3; k++) nrm[k] = (vs[0] * n0[k] + vs[1] * n1[k]) + vs[2] * n2[k];
    normalize3(nrm, nrm); /* lessF(length, 0) is never true (:187) */
    float sg = psign(dot3(nrm, nor));
    for (int k = 0; k < 3; k++) { nrm[k] = nrm[k] * sg; s->normal_trav[k] = nrm[k]; }
    /* surface.comp:176-186 with no normal map: tbn * (0,0,1) = normalize(normal) */
    float t1[3];
    normalize3(nrm, t1);
    normalize3(t1, s->normal);
    int matID = tri_mats[tri] - cfg->material_offset;
    s->active = !(matID >= cfg->material_count || matID < 0);
    for (int k = 0; k < 4; k++) { s->albedo[k] = 0.f; s->emission[k] = 0.f; s->mr[k] = 0.f; }
    if (s->active) {
        const psmo_material* m = &mats[matID];
        static const float zero_tc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const float* tc = cfg->texcoords ? cfg->texcoords + (size_t)6 * tri : zero_tc;
        float tu = (vs[0] * tc[0] + vs[1] * tc[2]) + vs[2] * tc[4];
        float tv = (vs[0] * tc[1] + vs[1] * tc[3]) + vs[2] * tc[5];
        float diff[4] = {pmax(m->diffuse[0], 0.f), pmax(m->diffuse[1], 0.f), pmax(m->diffuse[2], 0.f), 1.0f};
        float emis[4] = {0.f, 0.f, 0.f, 0.f}; /* fetchEmissive: textures only (:110-116) */
        float spc[4] = {m->specular[0], m->specular[1], m->specular[2], m->specular[3]};
        if (valid_tex(cfg, m->diffusePart)) fetch_tex(&cfg->textures[m->diffusePart], tu, tv, 0, 0, diff);
        if (valid_tex(cfg, m->emissivePart)) fetch_tex(&cfg->textures[m->emissivePart], tu, tv, 0, 0, emis);
        if (valid_tex(cfg, m->specularPart)) fetch_tex(&cfg->textures[m->specularPart], tu, tv, 0, 0, spc);
        if (valid_tex(cfg, m->bumpPart)) {
            const psmo_texture* bt = &cfg->textures[m->bumpPart];
            /* tangent, directTraverse.comp:190-209 */
            float du1 = tc[2] - tc[0], du2 = tc[4] - tc[0];
            float dv1 = tc[3] - tc[1], dv2 = tc[5] - tc[1];
            float e0x = du1, e0y = du2, e1x = dv2, e1y = dv1 * -1.0f;
            if (fabsf(e0x) < 0.000001f && fabsf(e0y) < 0.000001f) { e0x = 1.f; e0y = 0.f; }
            if (fabsf(e1x) < 0.000001f && fabsf(e1y) < 0.000001f) { e1x = 1.f; e1y = 0.f; }
            float f = 1.f / (e0x * e1x + e0y * e1y);
            if (isnan(f)) f = 0.f;
            if (isinf(f)) f = 10000.f;
            float tang[3], tangent[3];
            for (int k = 0; k < 3; k++) tang[k] = fmaf(e1x, d1[k], e1y * d2[k]) * f;
            float ts = psign(dot3(tang, nor));
            for (int k = 0; k < 3; k++) tangent[k] = tang[k] - nrm[k] * ts;
            normalize3(tangent, tangent);
            /* getNormalMapping, surface.comp:138-153 */
            float nm4[4], nm[3];
            fetch_tex(bt, tu, tv, 0, 0, nm4);
            if (equalF(nm4[0], nm4[1]) && equalF(nm4[0], nm4[2])) { /* grey: a height map */
                float h00[4], h01[4], h10[4];
                fetch_tex(bt, tu, tv, 0, 0, h00);
                fetch_tex(bt, tu, tv, 1, 0, h01);
                fetch_tex(bt, tu, tv, 0, 1, h10);
                float z0 = h00[0] * 2.0f, z1 = h01[0] * 2.0f, z2 = h10[0] * 2.0f;
                float pa[3] = {1.0f - 0.0f, 0.0f - 0.0f, z1 - z0}, pb[3] = {0.0f - 0.0f, 1.0f - 0.0f, z2 - z0}, c2[3];
                cross3(pa, pb, c2);
                normalize3(c2, nm);
            } else {
                float raw[3] = {mixf(0.f, fmaf(nm4[0], 2.0f, -1.0f), 1.0f), mixf(0.f, fmaf(nm4[1], 2.0f, -1.0f), 1.0f),
                                mixf(1.f, fmaf(nm4[2], 2.0f, -1.0f), 1.0f)};
                normalize3(raw, nm);
            }
            /* surface.comp:176-186 */
            float ns[3], tg[3], bc[3], bt3[3], w[3], o[3];
            normalize3(nrm, ns);
            normalize3(tangent, tg);
            cross3(ns, tg, bc);
            normalize3(bc, bt3);
            normalize3(nm, w);
            for (int k = 0; k < 3; k++) o[k] = (tg[k] * w[0] + bt3[k] * w[1]) + ns[k] * w[2];
            normalize3(o, s->normal);
        }
        float emis2[4] = {emis[0] * 2.f, emis[1] * 2.f, emis[2] * 2.f, 1.0f};
        float mr[4] = {spc[1], spc[2], 0.f, 0.f};
        uint32_t p[2];
        pack_half4(diff, p); unpack_half4(p, s->albedo);
        pack_half4(emis2, p); unpack_half4(p, s->emission);
        pack_half4(mr, p); unpack_half4(p, s->mr);
    }
}

/* rayshading.comp:25-28 */
static void composite(const float* src, const float* dst, float* o) {
    float oa = src[3] + dst[3] * (1.0f - src[3]);
    float den = pmax(oa, 0.00001f);
    for (int k = 0; k < 3; k++)
        o[k] = pclamp((src[k] * src[3] + dst[k] * dst[3] * (1.0f - src[3])) / den, 0.f, 1.f);
    o[3] = pclamp(oa, 0.f, 1.f);
}

/* include/random.glsl:48-69 */
static void randomCosine(rng_t* g, const float* normal, float* o) {
    float up = sqrtf(rng_next(g));
    float over = sqrtf(1.f - up * up);
    float around = rng_next(g) * TWO_PI_F;
    float p0[3] = {0, 0, 1};
    if (fabsf(normal[0]) < SQRT_OF_ONE_THIRD_F) { p0[0] = 1; p0[1] = 0; p0[2] = 0; }
    else if (fabsf(normal[1]) < SQRT_OF_ONE_THIRD_F) { p0[0] = 0; p0[1] = 1; p0[2] = 0; }
    float c1[3], p1[3], p2[3];
    cross3(normal, p0, c1);
    normalize3(c1, p1);
    cross3(normal, p1, p2);
    float ca = psmo_cosf(around) * over, sa = psmo_sinf(around) * over;
    float v[3];
    for (int k = 0; k < 3; k++) v[k] = fmaf(normal[k], up, fmaf(p1[k], ca, p2[k] * sa));
    normalize3(v, o);
}
/* include/random.glsl:71-76 */
static void randomDirectionInSphere(rng_t* g, float* o) {
    float up = fmaf(rng_next(g), 2.0f, -1.0f);
    float over = sqrtf(1.f - up * up);
    float around = rng_next(g) * TWO_PI_F;
    float v[3] = {up, psmo_cosf(around) * over, psmo_sinf(around) * over};
    normalize3(v, o);
}
/* shadinglib.glsl:22-26 */
static void lightCenter(const psmo_light* L, float* o) {
    float lv[3];
    normalize3(L->lightVector, lv);
    float s = (L->lightVector[1] < 0.0f) ? -1.0f : 1.0f;
    for (int k = 0; k < 3; k++) o[k] = fmaf(lv[k] * s, L->lightVector[3], L->lightOffset[k] + 0.0f);
}
/* shadinglib.glsl:32-48 */
static float intersectSphere(const float* origin, const float* ray, const float* c, float radius) {
    float ts[3] = {origin[0] - c[0], origin[1] - c[1], origin[2] - c[2]};
    float a = dot3(ray, ray);
    float b = 2.0f * dot3(ts, ray);
    float cc = dot3(ts, ts) - radius * radius;
    float disc = fmaf(b, b, -4.0f * a * cc);
    float t = PSMO_INFINITY;
    if (disc > 0.0f) {
        float da = 0.5f / a;
        float sq = sqrtf(disc);
        float t1 = (-b - sq) * da;
        float t2 = (-b + sq) * da;
        float mn = pmin(t1, t2), mx = pmax(t1, t2);
        t = mx >= 0.0f ? (mn >= 0.0f ? mn : mx) : t;
    }
    return t;
}

typedef struct {
    float origin[3], direct[3], color[3], final[3];
    int bf;
} wray; /* working copy of RayRework */

/* createRay, rayslib.glsl:162-203 (+createRayStrict :130-156, _collect :59-79) */
static void deposit(const float* fin, int texel, float* texel_sum, int32_t* texel_flag) {
    float c[3] = {pmax(fin[0], 0.f), pmax(fin[1], 0.f), pmax(fin[2], 0.f)};
    int bad = 0;
    for (int k = 0; k < 3; k++) if (isnan(c[k]) || isinf(c[k])) bad = 1;
    if (mlength3(c) < 10000.f && !bad) {
        for (int k = 0; k < 3; k++) texel_sum[4 * texel + k] += c[k];
        texel_sum[4 * texel + 3] += 1.0f;
        texel_flag[texel] = 1;
    }
}
static void create_ray(const psmo_frame_cfg* cfg, wray* r, int texel, uint32_t pkey, psmo_ray* out,
                       int* nout, float* texel_sum, int32_t* texel_flag) {
    int invalid = R_ACTIVE(r->bf) == 0 || R_BOUNCE(r->bf) <= 0 || mlength3(r->color) < 0.0001f;
    if (mlength3(r->final) >= 0.0001f && R_ACTIVE(r->bf) == 0) deposit(r->final, texel, texel_sum, texel_flag);
    if (invalid) return;
    S_BASIS(r->bf, 0);
    int bounce = R_BOUNCE(r->bf) - 1;
    S_BOUNCE(r->bf, bounce);
    if (*nout >= cfg->ray_limit) return; /* canonical overflow rule: drop past currentRayLimit */
    psmo_ray* o = &out[(*nout)++];
    for (int k = 0; k < 3; k++) { o->origin[k] = r->origin[k]; o->direct[k] = r->direct[k]; o->color[k] = r->color[k]; }
    o->bitfield = r->bf;
    o->texel = texel;
    o->pkey = pkey;
}

static inline uint32_t child_key(uint32_t pkey, uint32_t site) { return psmo_hash(pkey ^ psmo_hash(site)); }

/* rayshading.comp:48-278 with surface.comp fused in front */
int psmo_shade(const psmo_frame_cfg* cfg, const psmo_light* lights, const psmo_material* mats,
               const int32_t* tri_mats, const float* tris, const float* normals, uint32_t time,
               const psmo_ray* rays, int nrays, const psmo_hit* hits, const int32_t* counts,
               psmo_ray* out_rays, float* texel_sum, int32_t* texel_flag) {
    int nout = 0;
    for (int it = 0; it < nrays; it++) {
        const psmo_ray* in = &rays[it];
        rng_t g = {in->pkey, 0u, time << 5};
        wray ray;
        for (int k = 0; k < 3; k++) { ray.origin[k] = in->origin[k]; ray.direct[k] = in->direct[k]; ray.color[k] = in->color[k]; ray.final[k] = 0.f; }
        ray.bf = in->bitfield;
        int skipping = 0;
        int n = counts[it];
        const psmo_hit* chain = &hits[(size_t)it * PSMO_BAKED_CAP];
        surf_t sf[PSMO_BAKED_CAP];
        for (int k = 0; k < n; k++) surface_eval(cfg, mats, tri_mats, tris, normals, &chain[k], &sf[k]);

        /* hit composite, :60-116 */
        float uvt_t = PSMO_INFINITY;
        float c_albedo[4] = {0, 0, 0, 0}, c_emission[4] = {0, 0, 0, 0}, c_mr[4] = {0, 0, 0, 0}, c_normal[3] = {0, 0, 0};
        int next = -1;
        if (n > 0) {
            int k = 0;
            while (!sf[k].active && k + 1 < n) k++;
            uvt_t = sf[k].uvt[2];
            if (!sf[k].active) {
                for (int c = 0; c < 3; c++) c_normal[c] = sf[k].normal_trav[c];
                next = -1;
            } else {
                for (int c = 0; c < 4; c++) { c_albedo[c] = sf[k].albedo[c]; c_emission[c] = sf[k].emission[c]; c_mr[c] = sf[k].mr[c]; }
                for (int c = 0; c < 3; c++) c_normal[c] = sf[k].normal[c];
                next = (k + 1 < n) ? k + 1 : -1;
            }
        }
        for (int i = 0; i < 8; i++) {
            if (next == -1) break;
            const surf_t* h = &sf[next];
            if (!equalF(uvt_t, h->uvt[2])) break;
            if (!h->active) { next = (next + 1 < n) ? next + 1 : -1; continue; }
            float comp[4];
            uvt_t = h->uvt[2];
            composite(c_albedo, h->albedo, comp);
            float aw = h->albedo[3];
            for (int c = 0; c < 4; c++) c_albedo[c] = comp[c];
            for (int c = 0; c < 3; c++) c_normal[c] = mixf(c_normal[c], h->normal[c], aw);
            for (int c = 0; c < 4; c++) { c_mr[c] = mixf(c_mr[c], h->mr[c], aw); c_emission[c] = mixf(c_emission[c], h->emission[c], aw); }
            if (c_albedo[3] > 0.99999f) break;
            next = (next + 1 < n) ? next + 1 : -1;
        }

        /* physical lights, :119-138 */
        int lc = -1;
        int type = R_TYPE(ray.bf);
        if (R_DL(ray.bf) > 0 && (type == 1 || type == 2) && !skipping) {
            int nl = cfg->light_count < 16 ? cfg->light_count : 16;
            for (int i = 0; i < nl; i++) {
                float ctr[3];
                lightCenter(&lights[i], ctr);
                float dt = intersectSphere(ray.origin, ray.direct, ctr, lights[i].lightColor[3] + GAP);
                float t = 1.0f * dt;
                if (lessF(dt, PSMO_INFINITY) && lessEqualF(t, uvt_t)) lc = i;
            }
        }
        if (lc >= 0 && (R_TARGET(ray.bf) == lc || type != 2)) {
            for (int k = 0; k < 3; k++) { ray.final[k] = ray.color[k] * pmax(lights[lc].lightColor[k], 0.f); ray.color[k] = ray.color[k] * 0.0f; }
            S_ACTIVE(ray.bf, 0);
            skipping = 1;
        }
        /* background, :141-152 (constant sky) */
        if (greaterEqualF(uvt_t, PSMO_INFINITY) && type != 2 && !skipping) {
            float envc[3];
            read_env(cfg, ray.direct, envc);
            for (int k = 0; k < 3; k++) { ray.final[k] = ray.color[k] * envc[k]; ray.color[k] = ray.color[k] * 0.0f; }
            S_ACTIVE(ray.bf, 0);
            skipping = 1;
        }
        /* :155-161 */
        normalize3(ray.direct, ray.direct);
        for (int k = 0; k < 3; k++) ray.origin[k] = ray.origin[k] + ray.direct[k] * uvt_t;
        if (R_ACTIVE(ray.bf) < 1 || n == 0) skipping = 1;

        /* :164-180 */
        float normal[3];
        {
            float dn = dot3(c_normal, ray.direct);
            for (int k = 0; k < 3; k++) normal[k] = (dn < 0.0f) ? c_normal[k] : -c_normal[k];
        }
        float refly = c_mr[0];
        float pw = pclamp(psmo_powf(fabsf(dot3(ray.direct, normal)), 1.400f - 1.f), 0.0f, 1.0f);
        float sc[3];
        float sm = sqrtf(c_mr[1]);
        for (int k = 0; k < 3; k++) sc[k] = mixf(mixf(1.f, 0.05f, pw), c_albedo[k], sm);
        float emis = mlength3(c_emission);
        float spca = pclamp(mlength3(sc), 0.0f, 1.0f);
        float prom = 1.0f - c_albedo[3];
        float aprom = (type == 2) ? prom : ((rng_next(&g) < prom) ? 1.f : 0.f);

        wray diffuseRay = ray, reflectionRay = ray, emissiveRay = ray;
        for (int k = 0; k < 3; k++) { diffuseRay.final[k] *= 0.0f; reflectionRay.final[k] *= 0.0f; emissiveRay.final[k] *= 0.0f; }
        if (!skipping) for (int k = 0; k < 3; k++) ray.final[k] *= 0.f;

        if (R_ACTIVE(ray.bf) > 0 && !skipping) {
            for (int k = 0; k < 3; k++) ray.final[k] *= 0.0f;
            /* diffuse(), shadinglib.glsl:106-119 */
            {
                wray* r = &diffuseRay;
                float d[3];
                for (int k = 0; k < 3; k++) r->color[k] *= c_albedo[k];
                randomCosine(&g, normal, d);
                normalize3(d, r->direct);
                for (int k = 0; k < 3; k++) r->origin[k] = fmaf(r->direct[k], GAP, r->origin[k]);
                S_ACTIVE(r->bf, R_TYPE(r->bf) == 2 ? 0 : R_ACTIVE(r->bf));
                S_BOUNCE(r->bf, R_BOUNCE(r->bf) < 2 ? R_BOUNCE(r->bf) : 2);
                S_TYPE(r->bf, 1);
                S_DL(r->bf, 0);
            }
            /* reflection(), shadinglib.glsl:139-148 */
            {
                wray* r = &reflectionRay;
                float col[3];
                for (int k = 0; k < 3; k++) col[k] = pclamp(sc[k] / spca, 0.0f, 1.0f);
                float dn = dot3(normal, r->direct);
                float refl[3], rc[3], mx[3];
                for (int k = 0; k < 3; k++) refl[k] = r->direct[k] - 2.0f * dn * normal[k];
                randomCosine(&g, normal, rc);
                float a = pclamp(refly * rng_next(&g), 0.0f, 1.0f);
                for (int k = 0; k < 3; k++) mx[k] = mixf(refl[k], rc[k], a);
                normalize3(mx, r->direct);
                for (int k = 0; k < 3; k++) r->color[k] *= col[k];
                for (int k = 0; k < 3; k++) r->origin[k] = fmaf(r->direct[k], GAP, r->origin[k]);
                S_DL(r->bf, (R_TYPE(r->bf) == 1) ? 0 : 1);
                S_TYPE(r->bf, 0);
                S_BOUNCE(r->bf, R_BOUNCE(r->bf) < 3 ? R_BOUNCE(r->bf) : 3);
                S_ACTIVE(r->bf, R_TYPE(r->bf) == 2 ? 0 : R_ACTIVE(r->bf));
            }
            /* emissive(), shadinglib.glsl:127-137 */
            {
                wray* r = &emissiveRay;
                float d[3];
                for (int k = 0; k < 3; k++) r->final[k] = pmax(r->color[k] * c_emission[k], 0.0f);
                for (int k = 0; k < 3; k++) r->final[k] = (R_TYPE(r->bf) == 1) ? 0.0f : pmax(r->final[k], 0.0f);
                for (int k = 0; k < 3; k++) r->color[k] *= 0.0f;
                randomCosine(&g, normal, d);
                normalize3(d, r->direct);
                for (int k = 0; k < 3; k++) r->origin[k] = fmaf(r->direct[k], GAP, r->origin[k]);
                S_BOUNCE(r->bf, 0);
                S_ACTIVE(r->bf, 0);
                S_DL(r->bf, 0);
            }
            /* promised(), shadinglib.glsl:121-125 */
            S_BOUNCE(ray.bf, R_BOUNCE(ray.bf) + 1);
            for (int k = 0; k < 3; k++) ray.origin[k] = fmaf(ray.direct[k], GAP, ray.origin[k]);
            for (int k = 0; k < 3; k++) { ray.color[k] *= aprom; ray.final[k] *= aprom; }
        } else {
            for (int k = 0; k < 3; k++) { reflectionRay.color[k] *= 0.0f; emissiveRay.color[k] *= 0.0f; diffuseRay.color[k] *= 0.0f; diffuseRay.final[k] *= 0.0f; }
        }
        if (!skipping) {
            for (int k = 0; k < 3; k++) {
                diffuseRay.color[k] *= 1.0f - aprom;
                diffuseRay.final[k] *= 1.0f - aprom;
                reflectionRay.color[k] *= 1.0f - aprom;
                emissiveRay.final[k] *= (1.0f - aprom) * (1.0f - pclamp(spca, 0.0f, 1.0f));
            }
        }
        if (R_BASIS(ray.bf) == 1 && aprom < 0.1f) S_BASIS(ray.bf, 0); /* :226-231 (last3d unused) */

        /* reclaim current ray, :235-251 */
        {
            int bounce = R_BOUNCE(ray.bf) - 1;
            for (int k = 0; k < 3; k++) { ray.final[k] = pmax(0.0f, ray.final[k]); ray.color[k] = pmax(0.0f, ray.color[k]); }
            if (bounce < 0 || mlength3(ray.color) < 0.0001f || n == 0) S_ACTIVE(ray.bf, 0);
            S_BOUNCE(ray.bf, bounce >= 0 ? bounce : 0);
            /* storeRay + addRayToList, rayslib.glsl:110-121,81-97 */
            if (mlength3(ray.final) >= 0.0001f && R_ACTIVE(ray.bf) == 0) deposit(ray.final, in->texel, texel_sum, texel_flag);
            if (R_ACTIVE(ray.bf) == 1 && nout < cfg->ray_limit) {
                psmo_ray* o = &out_rays[nout++];
                for (int k = 0; k < 3; k++) { o->origin[k] = ray.origin[k]; o->direct[k] = ray.direct[k]; o->color[k] = ray.color[k]; }
                o->bitfield = ray.bf; o->texel = in->texel; o->pkey = in->pkey;
            }
        }
        /* emit new rays, :263-275 */
        if (!skipping) {
            float coef = pclamp((rng_next(&g) < spca) ? 1.0f : 0.0f, 0.0f, 1.0f);
            for (int k = 0; k < 3; k++) { reflectionRay.color[k] *= coef; diffuseRay.color[k] *= 1.0f - coef; }
            /* directLight(0, diffuseRay, 1, normal), shadinglib.glsl:75-93 */
            wray shadowRay = diffuseRay;
            {
                wray* r = &shadowRay;
                S_ACTIVE(r->bf, R_TYPE(r->bf) == 2 ? 0 : R_ACTIVE(r->bf));
                S_DL(r->bf, 1);
                S_TYPE(r->bf, 2);
                S_TARGET(r->bf, 0);
                S_BOUNCE(r->bf, R_BOUNCE(r->bf) < 1 ? R_BOUNCE(r->bf) : 1);
                float ctr[3], sd[3], sl[3], lpath[3], ldirect[3];
                lightCenter(&lights[0], ctr);
                randomDirectionInSphere(&g, sd);
                for (int k = 0; k < 3; k++) sl[k] = fmaf(sd[k], lights[0].lightColor[3] - 0.0001f, ctr[k]);
                for (int k = 0; k < 3; k++) lpath[k] = sl[k] - r->origin[k];
                normalize3(lpath, ldirect);
                float dv[3] = {ctr[0] - r->origin[0], ctr[1] - r->origin[1], ctr[2] - r->origin[2]};
                float dist = len3(dv);
                float q = lights[0].lightColor[3] / dist;
                float weight = 1.0f - sqrtf(1.0f - pclamp(dot3(ldirect, normal) * 2.f * (q * q), 0.f, 1.f));
                for (int k = 0; k < 3; k++) r->origin[k] = fmaf(r->direct[k], -GAP, r->origin[k]);
                for (int k = 0; k < 3; k++) r->direct[k] = ldirect[k];
                for (int k = 0; k < 3; k++) r->color[k] *= 1.0f * weight;
                for (int k = 0; k < 3; k++) r->final[k] *= 0.f;
                for (int k = 0; k < 3; k++) r->origin[k] = fmaf(r->direct[k], GAP, r->origin[k]);
            }
            /* emitRay x3 (shadinglib.glsl:191-195), applyLight (:181-189) */
            create_ray(cfg, &diffuseRay, in->texel, child_key(in->pkey, 1u), out_rays, &nout, texel_sum, texel_flag);
            create_ray(cfg, &reflectionRay, in->texel, child_key(in->pkey, 2u), out_rays, &nout, texel_sum, texel_flag);
            {
                float ce = pclamp(emis, 0.0f, 1.0f);
                for (int k = 0; k < 3; k++) { emissiveRay.color[k] *= ce; emissiveRay.final[k] *= ce; }
                create_ray(cfg, &emissiveRay, in->texel, child_key(in->pkey, 4u), out_rays, &nout, texel_sum, texel_flag);
            }
            {
                int off = (R_TYPE(diffuseRay.bf) == 2) || (dot3(c_normal, shadowRay.direct) < 0.f);
                S_ACTIVE(shadowRay.bf, off ? 0 : R_ACTIVE(shadowRay.bf));
                create_ray(cfg, &shadowRay, in->texel, child_key(in->pkey, 3u), out_rays, &nout, texel_sum, texel_flag);
            }
        }
    }
    return nout;
}

/* ------------------------------------------------------------------ */
/* sampler.comp:37-97 + deinterlace/filter (plain copies, :16-25)      */
/* ------------------------------------------------------------------ */
void psmo_sample(const psmo_frame_cfg* cfg, const float* texel_coord, const float* texel_sum,
                 const int32_t* texel_flag, float* presampled, float* filtered) {
    int w = cfg->width, h = cfg->height, dw = cfg->display_width, dh = cfg->display_height;
    float ax = (float)w / (float)dw, ay = (float)h / (float)dh;
    int sclx = (int)ceilf(ax), scly = (int)ceilf(ay);
    for (int it = 0; it < dw * dh; it++) {
        int px = it % dw, py = it / dw;
        int samplecount = 0;
        int bx = (int)((float)px * ax), by = (int)((float)py * ay);
        float newc[3] = {0, 0, 0};
        for (int x = -1; x <= sclx; x++) {
            for (int y = -1; y <= scly; y++) {
                int cx = bx + x, cy = by + y;
                if (cx >= 0 && cx < w && cy >= 0 && cy < h) {
                    int ts = cy * w + cx;
                    if (!texel_flag[ts]) continue;
                    float sx = texel_coord[2 * ts + 0] * (float)dw;
                    float sy = texel_coord[2 * ts + 1] * (float)dh;
                    float dx = (sx - (float)px) + 0.00001f, dy = (sy - (float)py) + 0.00001f;
                    if (dx >= 0.0f && dx < 1.0f && dy >= 0.0f && dy < 1.0f) {
                        samplecount++;
                        for (int k = 0; k < 3; k++) newc[k] += texel_sum[4 * ts + k];
                    }
                }
            }
        }
        if (samplecount > 0) {
            for (int k = 0; k < 3; k++) newc[k] = newc[k] / (float)samplecount;
            float* xs = &presampled[4 * it];
            float next = xs[3] + (float)samplecount;
            float prev = xs[3];
            float divisor = prev / next;
            for (int k = 0; k < 3; k++) xs[k] = fmaf(xs[k], divisor, newc[k] * (1.0f - divisor));
            float lock = (float)(cfg->samples_lock - 1);
            xs[3] = (cfg->samples_lock > 0) ? pmin(next, lock) : next;
        }
        for (int k = 0; k < 4; k++) filtered[4 * it + k] = presampled[4 * it + k];
    }
}
