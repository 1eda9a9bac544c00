// shade.hip -- wavefront loop stages for gfx950: camera, surface+shade, queue hand-off, sample.
//
// Replaces ShadersSDK/raytracing/{camera,surface,rayshading,sampler,deinterlace,filter}.comp with
// include/{random,rayslib,shadinglib}.glsl, as driven by psm::Pipeline::{camera,applyMaterials,
// shade,reloadQueuedRays,sample} (Include/Prismarine/Pipeline.inl:251-436).
//
// Differences in mechanism, not in results (DESIGN.md "wavefront loop"):
//   * rays live in three float4 SoA queues (origin|texel, direct|bitfield, color|pkey) instead of
//     80-byte AoS RayRework records addressed through index lists (rayslib.glsl:12-46)
//   * surface.comp is fused in front of rayshading.comp (no 112-byte HitRework round trip)
//   * the next queue is built by an ordered compaction (block scan -> staging -> ordered copy), so
//     queue order is the canonical [current, diffuse, reflection, shadow] per input ray instead of
//     the arrival order of wave-aggregated atomics (ballotlib.glsl:106-131)
//   * radiance is summed per texel with float atomics instead of colour-chain linked lists
//     (rayslib.glsl:59-79 + sampler.comp:15-35)
//   * the RNG stream id is the ray's path key, not its queue slot (SURVEY a-15)
#include "psm_common.h"
#include "psm_internal.h"

namespace psm {

constexpr float TWO_PI_F = 6.2831853071795864769252867665590057683943f;
constexpr float SQRT_OF_ONE_THIRD_F = 0.5773502691896257645091487805019574556476f;
static_assert(QUEUE_SEG == SHADE_BLOCK * 4, "one segment holds a shading workgroup's output rays");

// ray bitfield, include/structs.glsl:73-78
PSM_HD int bf_get(int bf, int off, int bits) { return (bf >> off) & ((1 << bits) - 1); }
PSM_HD int bf_set(int bf, int v, int off, int bits) {
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

struct Mat16 {
    float m[16];
};

// tile = the texels a context owns. mode 0: rows [a, b) in bands of 8 from a; mode 1: the global 8-row bands the
// dealing `map` gives to rank a (BandMap, psm_internal.h). local_row = index of the row among the owned rows.
struct Tile {
    uint32_t mode, a, b, h;
    BandMap map;
    uint8_t pos[MAX_BAND_PERIOD];  // mode 1: the positions of the period that rank a owns, ascending
};
struct TileRow {
    bool owned;
    uint32_t by, bh, local_base;  // band start row, band height, owned rows before the band
};
PSM_HD TileRow tile_row(const Tile& t, uint32_t y) {
    TileRow r;
    if (t.mode == 0) {
        r.owned = y >= t.a && y < t.b;
        uint32_t band = r.owned ? ((y - t.a) >> 3) : 0;
        r.by = t.a + band * 8;
        r.bh = (t.b - r.by) < 8u ? (t.b - r.by) : 8u;
        r.local_base = band * 8;
    } else {
        uint32_t g = y >> 3;
        r.owned = t.map.rank_of(g) == t.a;
        r.by = g * 8;
        r.bh = (t.h - r.by) < 8u ? (t.h - r.by) : 8u;
        r.local_base = t.map.local_band(g) * 8;
    }
    return r;
}

// owned texel k (row-major over the owned rows) -> its row; the inverse of tile_row().local_base
PSM_HD uint32_t tile_owned_row(const Tile& t, uint32_t local_row) {
    if (t.mode == 0) return t.a + local_row;
    const uint32_t lb = local_row >> 3, mine = t.map.cnt[t.a];
    return ((lb / mine) * t.map.P + t.pos[lb % mine]) * 8 + (local_row & 7u);
}

// jittered sample position of a texel, camera.comp:27-35: a pure function of (texel, time)
PSM_D float2 texel_coord(uint32_t idx, uint32_t x, uint32_t y, uint32_t time, float invw, float invh) {
    Rng g{idx, 0u, time << 5};
    float rx = g.next();
    float ry = g.next();
    return make_float2(((float)x + pclamp(rx, 0.00001f, 0.99999f)) * invw,   // :35
                       ((float)y + pclamp(ry, 0.00001f, 0.99999f)) * invh);
}

// ---- camera, raytracing/camera.comp:22-101 ------------------------------------------------------
// One thread per OWNED texel: a rank of a tile-sharded frame touches only its own rows (texel arrays and ray queue).
__global__ __launch_bounds__(256) void rt_camera(Mat16 camInv, Mat16 projInv, uint32_t time, uint32_t w, uint32_t h,
                                                 Tile tile, uint32_t nrays, float4* __restrict__ qA,
                                                 float4* __restrict__ qB, float4* __restrict__ qC,
                                                 float2* __restrict__ t_coord, float4* __restrict__ t_sum,
                                                 int32_t* __restrict__ t_flag, uint32_t* __restrict__ cnt,
                                                 uint32_t* __restrict__ qbases, int enable360) {
    uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k == 0) { cnt[0] = nrays; cnt[1] = 0; cnt[2] = 0; qbases[0] = 0; qbases[1] = nrays; }  // the primary queue is one dense segment
    if (k >= nrays) return;
    uint32_t x = k % w, y = tile_owned_row(tile, k / w);
    uint32_t idx = y * w + x;
    float invw = 1.0f / (float)w, invh = 1.0f / (float)h;
    float2 cxy = texel_coord(idx, x, y, time, invw, invh);
    float cx = cxy.x, cy = cxy.y;
    t_coord[idx] = cxy;
    t_sum[idx] = make_float4(0.f, 0.f, 0.f, 1.f);  // pre-collected zero sample (:99)
    t_flag[idx] = 1;
    TileRow tr = tile_row(tile, y);
    float nx = cx * 2.0f - 1.0f, ny = cy * 2.0f - 1.0f;
    float t0[4], co[4], orig[4];
    mat_vec(projInv.m, nx, ny, 0.999f, 1.0f, t0);
    mat_vec(camInv.m, t0[0], t0[1], t0[2], t0[3], co);
    mat_vec(projInv.m, nx, ny, 0.0f, 1.0f, t0);
    mat_vec(camInv.m, t0[0], t0[1], t0[2], t0[3], orig);
    float cw = co[3], ow = orig[3];
#pragma unroll
    for (int k = 0; k < 4; k++) { co[k] = co[k] / cw; orig[k] = orig[k] / ow; }
    v3 dir = normalize3(mk3(co[0] - orig[0], co[1] - orig[1], co[2] - orig[2]));
    if (enable360 == 1) {  // camera.comp:48-59: equirect directions, two fixed quaternion turns
        const float PI_F = 3.1415926535897932384626422832795028841971f;
        float picx = (nx * -1.f) * PI_F, picy = (ny * 0.5f) * PI_F;
        v3 v = mk3(pcos(picy) * pcos(picx), pcos(picy) * psin(picx), psin(picy));
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const float half = PI_F / 4.f;
            const float sh = psin(half), ch = pcos(half);
            v3 qv = q == 0 ? mk3(0.f * sh, 0.f * sh, -1.f * sh) : mk3(1.f * sh, 0.f * sh, 0.f * sh);
            v3 c1 = cross3(v, qv);
            v3 t = mk3(c1.x + ch * v.x, c1.y + ch * v.y, c1.z + ch * v.z);
            v3 c2 = cross3(t, qv);
            v = mk3(v.x + 2.0f * c2.x, v.y + 2.0f * c2.y, v.z + 2.0f * c2.z);
        }
        float od[4];
        mat_vec(camInv.m, 0.f, 0.f, 0.f, 1.f, orig);
        mat_vec(camInv.m, v.x, v.y, v.z, 0.f, od);
        dir = mk3(od[0], od[1], od[2]);
    }
    int bf = 0;
    S_ACTIVE(bf, 1); S_TYPE(bf, 0); S_DL(bf, 0); S_BOUNCE(bf, 4); S_BASIS(bf, 1);
    S_BOUNCE(bf, R_BOUNCE(bf) - 1);  // createRayIdx -> createRayStrict, rayslib.glsl:130-156
    // queue order: bands of 8 rows, 8-wide tiles left to right, row-major inside a tile, so one wave
    // of primary rays covers an 8x8 pixel tile (queue order never changes results: RNG is per pkey)
    uint32_t tx = x >> 3;
    uint32_t tw = min(8u, w - tx * 8);
    uint32_t q = tr.local_base * w + tx * 8 * tr.bh + (y - tr.by) * tw + (x - tx * 8);
    st_stream(&qA[q], make_float4(orig[0], orig[1], orig[2], __int_as_float((int)idx)));
    st_stream(&qB[q], make_float4(dir.x, dir.y, dir.z, __int_as_float(bf)));
    st_stream(&qC[q], make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(idx)));
}

// The gathering rank of a sharded frame samples the whole image (sampler.comp reads every texel's jitter position and
// flag): it fills the texels it does not own here; their sums arrive with the gather (psm_rt_unpack_texels_dev).
__global__ __launch_bounds__(256) void rt_camera_rest(uint32_t time, uint32_t w, uint32_t h, Tile tile,
                                                      float2* __restrict__ t_coord, float4* __restrict__ t_sum,
                                                      int32_t* __restrict__ t_flag) {
    uint32_t idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= w * h) return;
    uint32_t x = idx % w, y = idx / w;
    if (tile_row(tile, y).owned) return;
    t_coord[idx] = texel_coord(idx, x, y, time, 1.0f / (float)w, 1.0f / (float)h);
    t_sum[idx] = make_float4(0.f, 0.f, 0.f, 1.f);
    t_flag[idx] = 1;
}

// ---- surface: interpolateMeshData (directTraverse.comp:116-217) + surface.comp:165-195 -----------
struct Surf {
    float t;
    v3 normal_trav, normal;
    float albedo[4], emission[4], mr[4];
    bool active;
};

PSM_D void unpack4(uint32_t lo, uint32_t hi, float* o) {
    o[0] = half_lo(lo); o[1] = half_hi(lo); o[2] = half_lo(hi); o[3] = half_hi(hi);
}

struct SurfSrc {
    const float4* tri48;
    const float* nrm;
    const int32_t* tri_mats;
    const psm_material* mats;
    const float* uv;       // 6 floats / triangle
    const TexDesc* tex;    // MAX_TEXTURES slots
    const ObjGeom* geoms;  // MULTI: per-object geometry, indexed by the tag in hit.w
    int mat_offset, mat_count;
};

// validateTexture, surface.comp:81-83
PSM_D bool valid_tex(const TexDesc* __restrict__ tex, uint32_t binding) {
    return binding != 0u && binding != 0xFFFFFFFFu && binding < (uint32_t)MAX_TEXTURES && tex[binding].w > 0;
}

// fetchTexture, surface.comp:85-95: RGBA8 unorm, GL_LINEAR, GL_REPEAT (TextureSet.inl:113-118), fp32 weights;
// NaN/Inf results read as 0 (a non-finite coordinate therefore reads 0)
PSM_D void fetch_tex(const TexDesc t, float u, float v, int ox, int oy, float* o) {
    float uu = u + (float)ox / (float)t.w, vv = v + (float)oy / (float)t.h;
    if (!(pabs(uu) < INF) || !(pabs(vv) < INF)) { o[0] = o[1] = o[2] = o[3] = 0.f; return; }
    uu = uu - floorf(uu); vv = vv - floorf(vv);
    float x = uu * (float)t.w - 0.5f, y = vv * (float)t.h - 0.5f;
    float fx = floorf(x), fy = floorf(y);
    float a = x - fx, b = y - fy;
    int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    x0 = x0 < 0 ? x0 + t.w : x0; y0 = y0 < 0 ? y0 + t.h : y0;
    x1 = x1 >= t.w ? x1 - t.w : x1; y1 = y1 >= t.h ? y1 - t.h : y1;
    x0 = x0 >= t.w ? x0 - t.w : x0; y0 = y0 >= t.h ? y0 - t.h : y0;  // fract() rounding up to 1.0
    uint32_t p00 = t.texels[y0 * t.w + x0], p10 = t.texels[y0 * t.w + x1];
    uint32_t p01 = t.texels[y1 * t.w + x0], p11 = t.texels[y1 * t.w + x1];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        float t00 = (float)((p00 >> (8 * c)) & 255u) / 255.0f, t10 = (float)((p10 >> (8 * c)) & 255u) / 255.0f;
        float t01 = (float)((p01 >> (8 * c)) & 255u) / 255.0f, t11 = (float)((p11 >> (8 * c)) & 255u) / 255.0f;
        float top = t00 * (1.0f - a) + t10 * a, bot = t01 * (1.0f - a) + t11 * a;
        o[c] = top * (1.0f - b) + bot * b;
    }
}

template <bool TEX, bool MULTI>
PSM_D Surf surface_eval(float4 hit, const SurfSrc& src) {
    Surf s;
    int tri = __float_as_int(hit.w);
    ObjGeom g = {src.tri48, src.nrm, src.tri_mats, src.uv};
    if (MULTI) {  // the hierarchy whose mosaics were bound when this hit was interpolated
        g = src.geoms[(tri >> OBJ_SHIFT) & (MAX_TRAV_OBJECTS - 1)];
        tri &= (1 << OBJ_SHIFT) - 1;
    }
    float u = hit.x, v = hit.y;
    s.t = hit.z;
    float4 b = g.tri48[(size_t)3 * tri + 1], c = g.tri48[(size_t)3 * tri + 2];
    v3 d1 = mk3(b.x, b.y, b.z), d2 = mk3(c.x, c.y, c.z);
    const float* n = g.nrm + (size_t)9 * tri;   // (cached on purpose, like the triangle records: with the hint 2.40 -> 2.47 ms per frame)
    float vs0 = (1.0f - u) - v, vs1 = u, vs2 = v;
    v3 nor = normalize3(cross3(d1, d2));
    v3 nn = mk3((vs0 * n[0] + vs1 * n[3]) + vs2 * n[6], (vs0 * n[1] + vs1 * n[4]) + vs2 * n[7],
                (vs0 * n[2] + vs1 * n[5]) + vs2 * n[8]);
    nn = normalize3(nn);
    float sg = psign(dot3(nn, nor));
    nn = nn * sg;
    s.normal_trav = nn;
    s.normal = normalize3(normalize3(nn));  // surface.comp:176-186 with no normal map
    int matID = g.tri_mats[tri] - src.mat_offset;
    s.active = !(matID >= src.mat_count || matID < 0);
#pragma unroll
    for (int k = 0; k < 4; k++) { s.albedo[k] = 0.f; s.emission[k] = 0.f; s.mr[k] = 0.f; }
    if (s.active) {
        const psm_material* m = src.mats + matID;
        float diff[4] = {pmax(m->diffuse[0], 0.f), pmax(m->diffuse[1], 0.f), pmax(m->diffuse[2], 0.f), 1.0f};
        float emis[4] = {0.f, 0.f, 0.f, 0.f};
        float spc[4] = {m->specular[0], m->specular[1], m->specular[2], m->specular[3]};
        if (TEX) {
        const float* tc = g.uv + (size_t)6 * tri;
        float tu = (vs0 * tc[0] + vs1 * tc[2]) + vs2 * tc[4], tv = (vs0 * tc[1] + vs1 * tc[3]) + vs2 * tc[5];
        if (valid_tex(src.tex, m->diffusePart)) fetch_tex(src.tex[m->diffusePart], tu, tv, 0, 0, diff);     // :155-161
        if (valid_tex(src.tex, m->emissivePart)) fetch_tex(src.tex[m->emissivePart], tu, tv, 0, 0, emis);   // :110-116
        if (valid_tex(src.tex, m->specularPart)) fetch_tex(src.tex[m->specularPart], tu, tv, 0, 0, spc);    // :102-108
        if (valid_tex(src.tex, m->bumpPart)) {
            const TexDesc bt = src.tex[m->bumpPart];
            // tangent, directTraverse.comp:190-209
            float du1 = tc[2] - tc[0], du2 = tc[4] - tc[0];
            float dv1 = tc[3] - tc[1], dv2 = tc[5] - tc[1];
            float e0x = du1, e0y = du2, e1x = dv2, e1y = dv1 * -1.0f;
            if (pabs(e0x) < 0.000001f && pabs(e0y) < 0.000001f) { e0x = 1.f; e0y = 0.f; }
            if (pabs(e1x) < 0.000001f && pabs(e1y) < 0.000001f) { e1x = 1.f; e1y = 0.f; }
            float f = 1.f / (e0x * e1x + e0y * e1y);
            if (isnan(f)) f = 0.f;
            if (isinf(f)) f = 10000.f;
            v3 tang = mk3(fmaf(e1x, d1.x, e1y * d2.x) * f, fmaf(e1x, d1.y, e1y * d2.y) * f, fmaf(e1x, d1.z, e1y * d2.z) * f);
            float ts = psign(dot3(tang, nor));
            v3 tangent = normalize3(tang - nn * ts);
            // getNormalMapping, surface.comp:138-153
            float nm4[4];
            fetch_tex(bt, tu, tv, 0, 0, nm4);
            v3 nm;
            if (equalF(nm4[0], nm4[1]) && equalF(nm4[0], nm4[2])) {  // grey: a height map
                float h00[4], h01[4], h10[4];
                fetch_tex(bt, tu, tv, 0, 0, h00);
                fetch_tex(bt, tu, tv, 1, 0, h01);
                fetch_tex(bt, tu, tv, 0, 1, h10);
                float z0 = h00[0] * 2.0f, z1 = h01[0] * 2.0f, z2 = h10[0] * 2.0f;
                nm = normalize3(cross3(mk3(1.0f - 0.0f, 0.0f - 0.0f, z1 - z0), mk3(0.0f - 0.0f, 1.0f - 0.0f, z2 - z0)));
            } else {
                nm = normalize3(mk3(mixf(0.f, fmaf(nm4[0], 2.0f, -1.0f), 1.0f), mixf(0.f, fmaf(nm4[1], 2.0f, -1.0f), 1.0f),
                                    mixf(1.f, fmaf(nm4[2], 2.0f, -1.0f), 1.0f)));
            }
            // surface.comp:176-186
            v3 normal_s = normalize3(nn);
            v3 tangent_s = normalize3(tangent);
            v3 bitangent = normalize3(cross3(normal_s, tangent_s));
            v3 w = normalize3(nm);
            s.normal = normalize3(mk3((tangent_s.x * w.x + bitangent.x * w.y) + normal_s.x * w.z,
                                      (tangent_s.y * w.x + bitangent.y * w.y) + normal_s.y * w.z,
                                      (tangent_s.z * w.x + bitangent.z * w.y) + normal_s.z * w.z));
        }
        }  // TEX
        unpack4(pack_half2(diff[0], diff[1]), pack_half2(diff[2], diff[3]), s.albedo);
        unpack4(pack_half2(emis[0] * 2.f, emis[1] * 2.f), pack_half2(emis[2] * 2.f, 1.0f), s.emission);
        unpack4(pack_half2(spc[1], spc[2]), pack_half2(0.f, 0.f), s.mr);
    }
    return s;
}

// include/random.glsl:48-69 (u0, u1: the two random() draws it takes, in that order)
PSM_D v3 randomCosineU(float u0, float u1, v3 normal) {
    float up = sqrtf(u0);
    float over = sqrtf(1.f - up * up);
    float around = u1 * TWO_PI_F;
    v3 p0 = mk3(0, 0, 1);
    if (pabs(normal.x) < SQRT_OF_ONE_THIRD_F) p0 = mk3(1, 0, 0);
    else if (pabs(normal.y) < SQRT_OF_ONE_THIRD_F) p0 = mk3(0, 1, 0);
    v3 p1 = normalize3(cross3(normal, p0));
    v3 p2 = cross3(normal, p1);
    float ca = pcos(around) * over, sa = psin(around) * over;
    v3 v = mk3(fmaf(normal.x, up, fmaf(p1.x, ca, p2.x * sa)), fmaf(normal.y, up, fmaf(p1.y, ca, p2.y * sa)),
               fmaf(normal.z, up, fmaf(p1.z, ca, p2.z * sa)));
    return normalize3(v);
}
PSM_D v3 randomCosine(Rng& g, v3 normal) {
    float u0 = g.next();
    float u1 = g.next();
    return randomCosineU(u0, u1, normal);
}
// include/random.glsl:71-76 (u0, u1: its two draws)
PSM_D v3 randomDirectionInSphereU(float u0, float u1) {
    float up = fmaf(u0, 2.0f, -1.0f);
    float over = sqrtf(1.f - up * up);
    float around = u1 * TWO_PI_F;
    return normalize3(mk3(up, pcos(around) * over, psin(around) * over));
}
// shadinglib.glsl:22-26
PSM_D v3 lightCenter(const psm_light& L) {
    v3 lv = normalize3(mk3(L.lightVector[0], L.lightVector[1], L.lightVector[2]));
    float s = (L.lightVector[1] < 0.0f) ? -1.0f : 1.0f;
    return mk3(fmaf(lv.x * s, L.lightVector[3], L.lightOffset[0] + 0.0f), fmaf(lv.y * s, L.lightVector[3], L.lightOffset[1] + 0.0f),
               fmaf(lv.z * s, L.lightVector[3], L.lightOffset[2] + 0.0f));
}
// shadinglib.glsl:32-48
PSM_D float intersectSphere(v3 origin, v3 ray, v3 c, float radius) {
    v3 ts = origin - c;
    float a = dot3(ray, ray);
    float b = 2.0f * dot3(ts, ray);
    float cc = dot3(ts, ts) - radius * radius;
    float disc = fmaf(b, b, -4.0f * a * cc);
    float t = INF;
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

// readEnv, public/environment.glsl:23-26: equirect RGBA8, GL_LINEAR, clamp to edge (fp32 weights)
PSM_D v3 read_env(const uint32_t* __restrict__ sky, int sw, int sh, v3 r) {
    const float PI_F = 3.14159265358979323846f;
    v3 nr = normalize3(r);
    float u = fmaf(patan2(nr.z, nr.x) / PI_F, 0.5f, 0.5f);
    float v = fmaf((pasin(nr.y) * 2.0f) / PI_F, 0.5f, 0.5f);
    float x = u * (float)sw - 0.5f, y = v * (float)sh - 0.5f;
    float fx = floorf(x), fy = floorf(y);
    float a = x - fx, b = y - fy;
    int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    x0 = x0 < 0 ? 0 : (x0 > sw - 1 ? sw - 1 : x0);
    x1 = x1 < 0 ? 0 : (x1 > sw - 1 ? sw - 1 : x1);
    y0 = y0 < 0 ? 0 : (y0 > sh - 1 ? sh - 1 : y0);
    y1 = y1 < 0 ? 0 : (y1 > sh - 1 ? sh - 1 : y1);
    uint32_t p00 = sky[y0 * sw + x0], p10 = sky[y0 * sw + x1], p01 = sky[y1 * sw + x0], p11 = sky[y1 * sw + x1];
    float o[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float t00 = (float)((p00 >> (8 * c)) & 255u) / 255.0f, t10 = (float)((p10 >> (8 * c)) & 255u) / 255.0f;
        float t01 = (float)((p01 >> (8 * c)) & 255u) / 255.0f, t11 = (float)((p11 >> (8 * c)) & 255u) / 255.0f;
        float top = t00 * (1.0f - a) + t10 * a, bot = t01 * (1.0f - a) + t11 * a;
        o[c] = top * (1.0f - b) + bot * b;
    }
    return mk3(o[0], o[1], o[2]);
}

struct WRay {
    v3 origin, direct, color, fin;
    int bf;
};

// _collect, rayslib.glsl:59-79 (direct per-texel sum)
PSM_D void deposit(v3 fin, int texel, float4* __restrict__ t_sum, int32_t* __restrict__ t_flag) {
    v3 c = mk3(pmax(fin.x, 0.f), pmax(fin.y, 0.f), pmax(fin.z, 0.f));
    bool bad = isnan(c.x) || isnan(c.y) || isnan(c.z) || isinf(c.x) || isinf(c.y) || isinf(c.z);
    if (mlength3(c) < 10000.f && !bad) {
        float* p = (float*)&t_sum[texel];
        atomicAdd(p + 0, c.x);
        atomicAdd(p + 1, c.y);
        atomicAdd(p + 2, c.z);
        atomicAdd(p + 3, 1.0f);
        t_flag[texel] = 1;
    }
}

struct OutRay {
    float4 A, B, C;
};

// createRay, rayslib.glsl:162-203 (+createRayStrict :130-156): returns true when the ray is queued
PSM_D bool create_ray(WRay& r, int texel, uint32_t pkey, OutRay& o, float4* __restrict__ t_sum,
                      int32_t* __restrict__ t_flag) {
    bool invalid = R_ACTIVE(r.bf) == 0 || R_BOUNCE(r.bf) <= 0 || mlength3(r.color) < 0.0001f;
    if (mlength3(r.fin) >= 0.0001f && R_ACTIVE(r.bf) == 0) deposit(r.fin, texel, t_sum, t_flag);
    if (invalid) return false;
    S_BASIS(r.bf, 0);
    S_BOUNCE(r.bf, R_BOUNCE(r.bf) - 1);
    o.A = make_float4(r.origin.x, r.origin.y, r.origin.z, __int_as_float(texel));
    o.B = make_float4(r.direct.x, r.direct.y, r.direct.z, __int_as_float(r.bf));
    o.C = make_float4(r.color.x, r.color.y, r.color.z, __uint_as_float(pkey));
    return true;
}

struct ShadeArgs {
    RayQueue q;  // the rays to shade
    const float4* hit0;
    const uint32_t* hitN;
    const float4* pool;
    SurfSrc src;
    const psm_light* lights;
    float4 *sA, *sB, *sC;
    uint32_t* blockCounts;
    float4* t_sum;
    int32_t* t_flag;
    uint32_t nrays;
    uint32_t time;
    int light_count;
    float sky[3];
    const uint32_t* sky_tex;
    int sky_w, sky_h;
};

// surface.comp + rayshading.comp:48-278
// TEX = false is the same kernel with the sampler table known to be empty (validateTexture fails for every
// part): the texture-less frame keeps its registers and occupancy
template <bool TEX, bool MULTI, bool BOTH>
__global__ __launch_bounds__(SHADE_BLOCK) void rt_shade(ShadeArgs a) {
    __shared__ uint32_t scan_tmp[8];
    uint32_t it = blockIdx.x * SHADE_BLOCK + threadIdx.x;
    // the (at most) four rays a ray puts on the next queue, each in its own slot -- current, diffuse, reflection, shadow:
    // the canonical order -- with a flag; every index below is a constant, so the slots live in registers, not in scratch
    OutRay outs[4];
    bool have[4] = {false, false, false, false};
    const uint32_t nrays = min(a.nrays, a.q.bases[a.q.nb]);  // never past what the queue holds (psm_rt_set_ray_count)
    if (it < nrays) {
        const uint32_t loc = queue_loc(a.q.bases, a.q.nb, nrays, it);
        float4 A = ld_stream(&a.q.A[loc]), B = ld_stream(&a.q.B[loc]), C = ld_stream(&a.q.C[loc]);
        int in_texel = __float_as_int(A.w);
        uint32_t in_pkey = __float_as_uint(C.w);
        Rng g{in_pkey, 0u, a.time << 5};
        WRay ray;
        ray.origin = mk3(A.x, A.y, A.z);
        ray.direct = mk3(B.x, B.y, B.z);
        ray.color = mk3(C.x, C.y, C.z);
        ray.fin = mk3(0.f, 0.f, 0.f);
        ray.bf = __float_as_int(B.w);
        bool skipping = false;
        uint32_t hn = ld_stream(&a.hitN[it]);
        int n = (int)(hn & 15u);
        uint32_t poff = hn >> 4;

        // hit composite, rayshading.comp:60-116
        float uvt_t = INF;
        float c_albedo[4] = {0, 0, 0, 0}, c_emission[4] = {0, 0, 0, 0}, c_mr[4] = {0, 0, 0, 0};
        v3 c_normal = mk3(0, 0, 0);
        // one walk over the chain: the first active hit seeds the composite (:60-86), the hits at the same
        // distance behind it are blended in (:88-116)
        bool found = false;
        for (int k = 0; k < n; k++) {
            Surf h = surface_eval<TEX, MULTI>(k == 0 ? ld_stream(&a.hit0[it]) : a.pool[poff + k - 1], a.src);
            if (!found) {
                uvt_t = h.t;
                if (!h.active) { c_normal = h.normal_trav; continue; }
                found = true;
#pragma unroll
                for (int c = 0; c < 4; c++) { c_albedo[c] = h.albedo[c]; c_emission[c] = h.emission[c]; c_mr[c] = h.mr[c]; }
                c_normal = h.normal;
                continue;
            }
            if (!equalF(uvt_t, h.t)) break;
            if (!h.active) continue;
            // composite(), rayshading.comp:25-28
            float oa = c_albedo[3] + h.albedo[3] * (1.0f - c_albedo[3]);
            float den = pmax(oa, 0.00001f);
            float comp[4];
#pragma unroll
            for (int c = 0; c < 3; c++)
                comp[c] = pclamp((c_albedo[c] * c_albedo[3] + h.albedo[c] * h.albedo[3] * (1.0f - c_albedo[3])) / den, 0.f, 1.f);
            comp[3] = pclamp(oa, 0.f, 1.f);
            float aw = h.albedo[3];
            uvt_t = h.t;
#pragma unroll
            for (int c = 0; c < 4; c++) c_albedo[c] = comp[c];
            c_normal = mk3(mixf(c_normal.x, h.normal.x, aw), mixf(c_normal.y, h.normal.y, aw), mixf(c_normal.z, h.normal.z, aw));
#pragma unroll
            for (int c = 0; c < 4; c++) { c_mr[c] = mixf(c_mr[c], h.mr[c], aw); c_emission[c] = mixf(c_emission[c], h.emission[c], aw); }
            if (c_albedo[3] > 0.99999f) break;
        }

        // physical lights, :119-138
        int lc = -1;
        int type = R_TYPE(ray.bf);
        if (R_DL(ray.bf) > 0 && (type == 1 || type == 2) && !skipping) {
            int nl = a.light_count < 16 ? a.light_count : 16;
            for (int i = 0; i < nl; i++) {
                v3 ctr = lightCenter(a.lights[i]);
                float dt = intersectSphere(ray.origin, ray.direct, ctr, a.lights[i].lightColor[3] + GAP);
                float t = 1.0f * dt;
                if (lessF(dt, INF) && lessEqualF(t, uvt_t)) lc = i;
            }
        }
        if (lc >= 0 && (R_TARGET(ray.bf) == lc || type != 2)) {
            ray.fin = mk3(ray.color.x * pmax(a.lights[lc].lightColor[0], 0.f), ray.color.y * pmax(a.lights[lc].lightColor[1], 0.f),
                          ray.color.z * pmax(a.lights[lc].lightColor[2], 0.f));
            ray.color = ray.color * 0.0f;
            S_ACTIVE(ray.bf, 0);
            skipping = true;
        }
        // background, :141-152 (constant sky)
        if (greaterEqualF(uvt_t, INF) && type != 2 && !skipping) {
            v3 envc = a.sky_tex ? read_env(a.sky_tex, a.sky_w, a.sky_h, ray.direct) : mk3(a.sky[0], a.sky[1], a.sky[2]);
            ray.fin = mk3(ray.color.x * envc.x, ray.color.y * envc.y, ray.color.z * envc.z);
            ray.color = ray.color * 0.0f;
            S_ACTIVE(ray.bf, 0);
            skipping = true;
        }
        // :155-161
        ray.direct = normalize3(ray.direct);
        ray.origin = mk3(ray.origin.x + ray.direct.x * uvt_t, ray.origin.y + ray.direct.y * uvt_t, ray.origin.z + ray.direct.z * uvt_t);
        if (R_ACTIVE(ray.bf) < 1 || n == 0) skipping = true;

        // :164-180
        v3 normal;
        {
            float dn = dot3(c_normal, ray.direct);
            normal = (dn < 0.0f) ? c_normal : mk3(-c_normal.x, -c_normal.y, -c_normal.z);
        }
        float refly = c_mr[0];
        float pw = pclamp(ppow(pabs(dot3(ray.direct, normal)), 1.400f - 1.f), 0.0f, 1.0f);
        float sm_ = sqrtf(c_mr[1]);
        float diel = mixf(1.f, 0.05f, pw);
        v3 sc = mk3(mixf(diel, c_albedo[0], sm_), mixf(diel, c_albedo[1], sm_), mixf(diel, c_albedo[2], sm_));
        float emis = mlength3(mk3(c_emission[0], c_emission[1], c_emission[2]));
        float spca = pclamp(mlength3(sc), 0.0f, 1.0f);
        float prom = 1.0f - c_albedo[3];
        float aprom = (type == 2) ? prom : ((g.next() < prom) ? 1.f : 0.f);

        // From here on the reference builds three copies of the ray (diffuse, reflection, emissive) and a fourth (shadow) from
        // the diffuse one, edits them in branches and merges them again (:195-275). A ray that is `skipping` emits none of
        // them (:263), and `!skipping` implies an active ray (:159-161 above), so everything about the secondary rays lives
        // in ONE region that only non-skipping rays enter, each ray built from the fields it actually changes -- same
        // operations in the same order on the same values, bit for bit, without the per-field selects of the merges.
        const bool go = !skipping;
        const v3 hitp = ray.origin;       // the hit point, :155-156
        const v3 incol = ray.color;       // the ray's colour after the light / background tests (unchanged for a `go` ray)
        const int inbf = ray.bf;
        if (go) {
            ray.fin = ray.fin * 0.f;
            ray.fin = ray.fin * 0.0f;     // :195 (two products in the reference: the sign of a zero survives them alike)
            // promised(), shadinglib.glsl:121-125
            S_BOUNCE(ray.bf, R_BOUNCE(ray.bf) + 1);
            ray.origin = fma3(ray.direct, GAP, ray.origin);
            ray.color = ray.color * aprom;
            ray.fin = ray.fin * aprom;
        }
        if (R_BASIS(ray.bf) == 1 && aprom < 0.1f) S_BASIS(ray.bf, 0);

        // reclaim current ray, :235-251
        {
            int bounce = R_BOUNCE(ray.bf) - 1;
            ray.fin = mk3(pmax(0.0f, ray.fin.x), pmax(0.0f, ray.fin.y), pmax(0.0f, ray.fin.z));
            ray.color = mk3(pmax(0.0f, ray.color.x), pmax(0.0f, ray.color.y), pmax(0.0f, ray.color.z));
            if (bounce < 0 || mlength3(ray.color) < 0.0001f || n == 0) S_ACTIVE(ray.bf, 0);
            S_BOUNCE(ray.bf, bounce >= 0 ? bounce : 0);
            if (mlength3(ray.fin) >= 0.0001f && R_ACTIVE(ray.bf) == 0) deposit(ray.fin, in_texel, a.t_sum, a.t_flag);
            if (R_ACTIVE(ray.bf) == 1) {
                OutRay& o = outs[0];
                have[0] = true;
                o.A = make_float4(ray.origin.x, ray.origin.y, ray.origin.z, __int_as_float(in_texel));
                o.B = make_float4(ray.direct.x, ray.direct.y, ray.direct.z, __int_as_float(ray.bf));
                o.C = make_float4(ray.color.x, ray.color.y, ray.color.z, __uint_as_float(in_pkey));
            }
        }
        // the secondary rays, :195-210, :219-224, :263-275
        if (go) {
            const float om = 1.0f - aprom;
            // the ten random() draws of the region in the reference's order -- diffuse (#2, #3), reflection (#4, #5, #6),
            // emissive (#7, #8: its direction is drawn and never used, the ray is born inactive and only deposits `final`),
            // the lobe (#9), the shadow ray (#10, #11) -- taken first, so that each ray can be finished (and its registers
            // released into its output slot) before the next one is begun
            const float u2 = g.next(), u3 = g.next(), u4 = g.next(), u5 = g.next(), u6 = g.next();
            (void)g.next();
            (void)g.next();
            const float coef = pclamp((g.next() < spca) ? 1.0f : 0.0f, 0.0f, 1.0f);   // :267
            const float u10 = g.next(), u11 = g.next();

            // Of the diffuse and the reflection ray exactly one survives the lobe pick (:267-268: coef is 0 or 1, the other
            // ray's colour is multiplied by 0 and createRay drops it, and the shadow ray -- a copy of the diffuse one -- with
            // it; none of them deposits: their `final` is 0). So the cosine-weighted direction both need is evaluated ONCE,
            // from the draws of the ray that survives (#2, #3 for the diffuse ray, #4, #5 for the reflection), and only the
            // surviving rays are built -- the same operations on the same values for everything that reaches the queue.
            const bool refl_lobe = coef != 0.0f;
            // ... as long as that colour is an ordinary number: x * 0 is NaN for x = NaN or Inf, and createRay KEEPS a NaN colour
            // (`mlength(color) < 0.0001` is false for it) -- a black full-metal surface does that to its reflection ray (albedo 0,
            // metallic 1: the colour is clamp(0 / 0)), a colour that overflowed does it to the diffuse and shadow rays, and the
            // reference queues those rays. BOTH = true builds both lobes, the second one from its own draws, and lets createRay
            // decide, as the reference does; launch_rt_shade picks it whenever the materials at hand can produce such a colour
            // (any texture; a material that is not `ordinary`, api.hip: psm_rt_set_materials). BOTH = false is the same code with
            // the second turn known to build rays createRay would drop.
            auto lobe = [&](const bool lobe_refl) {
            const v3 rc = randomCosineU(lobe_refl ? u4 : u2, lobe_refl ? u5 : u3, normal);
            v3 sdir;   // this secondary ray's direction
            {
                // reflection(), shadinglib.glsl:139-148: mix(reflect(dir, n), randomCosine(n), clamp(roughness * random()))
                float dn = dot3(normal, ray.direct);
                v3 refl = mk3(ray.direct.x - 2.0f * dn * normal.x, ray.direct.y - 2.0f * dn * normal.y, ray.direct.z - 2.0f * dn * normal.z);
                float al = pclamp(refly * u6, 0.0f, 1.0f);
                v3 mixed = mk3(mixf(refl.x, rc.x, al), mixf(refl.y, rc.y, al), mixf(refl.z, rc.z, al));
                sdir = normalize3(lobe_refl ? mixed : rc);   // diffuse(): normalize(randomCosine(n)), :106-119
            }
            if (!lobe_refl) {
                // ---- diffuse ray: diffuse(), shadinglib.glsl:106-119
                WRay dr;
                dr.color = incol * mk3(c_albedo[0], c_albedo[1], c_albedo[2]);
                dr.color = dr.color * om;                       // :219
                dr.color = dr.color * (1.0f - coef);            // :268
                dr.direct = sdir;
                dr.origin = fma3(dr.direct, GAP, hitp);
                dr.fin = mk3(0.f, 0.f, 0.f);
                dr.bf = inbf;
                S_ACTIVE(dr.bf, R_TYPE(dr.bf) == 2 ? 0 : R_ACTIVE(dr.bf));
                S_BOUNCE(dr.bf, R_BOUNCE(dr.bf) < 2 ? R_BOUNCE(dr.bf) : 2);
                S_TYPE(dr.bf, 1);
                S_DL(dr.bf, 0);
                // ---- shadow ray = directLight(0, diffuseRay, 1, normal), shadinglib.glsl:75-93, then applyLight :181-189
                {
                    WRay sr = dr;
                    S_ACTIVE(sr.bf, R_TYPE(sr.bf) == 2 ? 0 : R_ACTIVE(sr.bf));
                    S_DL(sr.bf, 1);
                    S_TYPE(sr.bf, 2);
                    S_TARGET(sr.bf, 0);
                    S_BOUNCE(sr.bf, R_BOUNCE(sr.bf) < 1 ? R_BOUNCE(sr.bf) : 1);
                    v3 ctr = lightCenter(a.lights[0]);
                    v3 sd = randomDirectionInSphereU(u10, u11);
                    v3 sl = fma3(sd, a.lights[0].lightColor[3] - 0.0001f, ctr);
                    v3 ldirect = normalize3(sl - sr.origin);
                    float dist = len3(ctr - sr.origin);
                    float q = a.lights[0].lightColor[3] / dist;
                    float weight = 1.0f - sqrtf(1.0f - pclamp(dot3(ldirect, normal) * 2.f * (q * q), 0.f, 1.f));
                    sr.origin = fma3(sr.direct, -GAP, sr.origin);
                    sr.direct = ldirect;
                    sr.color = sr.color * (1.0f * weight);
                    sr.origin = fma3(sr.direct, GAP, sr.origin);
                    bool off = (R_TYPE(dr.bf) == 2) || (dot3(c_normal, sr.direct) < 0.f);
                    S_ACTIVE(sr.bf, off ? 0 : R_ACTIVE(sr.bf));
                    have[3] = create_ray(sr, in_texel, child_key(in_pkey, 3u), outs[3], a.t_sum, a.t_flag);
                }
                have[1] = create_ray(dr, in_texel, child_key(in_pkey, 1u), outs[1], a.t_sum, a.t_flag);
            } else {
                // ---- reflection ray: reflection(), shadinglib.glsl:139-148
                WRay rr;
                rr.direct = sdir;
                v3 col = mk3(pclamp(sc.x / spca, 0.0f, 1.0f), pclamp(sc.y / spca, 0.0f, 1.0f), pclamp(sc.z / spca, 0.0f, 1.0f));
                rr.color = incol * col;
                rr.color = rr.color * om;                   // :220
                rr.color = rr.color * coef;                 // :267
                rr.origin = fma3(rr.direct, GAP, hitp);
                rr.fin = mk3(0.f, 0.f, 0.f);
                rr.bf = inbf;
                S_DL(rr.bf, (R_TYPE(rr.bf) == 1) ? 0 : 1);
                S_TYPE(rr.bf, 0);
                S_BOUNCE(rr.bf, R_BOUNCE(rr.bf) < 3 ? R_BOUNCE(rr.bf) : 3);
                S_ACTIVE(rr.bf, R_TYPE(rr.bf) == 2 ? 0 : R_ACTIVE(rr.bf));
                have[2] = create_ray(rr, in_texel, child_key(in_pkey, 2u), outs[2], a.t_sum, a.t_flag);
            }
            };
            lobe(refl_lobe);              // the surviving lobe
            if (BOTH) lobe(!refl_lobe);   // ... and the other one, whose colour is 0 or NaN
            // ---- emissive: only its deposit (createRay of an inactive ray, rayslib.glsl:162-203)
            {
                v3 ef = mk3(pmax(incol.x * c_emission[0], 0.0f), pmax(incol.y * c_emission[1], 0.0f), pmax(incol.z * c_emission[2], 0.0f));
                if (R_TYPE(inbf) == 1) ef = mk3(0.f, 0.f, 0.f);
                else ef = mk3(pmax(ef.x, 0.0f), pmax(ef.y, 0.0f), pmax(ef.z, 0.0f));
                ef = ef * ((1.0f - aprom) * (1.0f - pclamp(spca, 0.0f, 1.0f)));   // :222-223
                ef = ef * pclamp(emis, 0.0f, 1.0f);                               // :271
                if (mlength3(ef) >= 0.0001f) deposit(ef, in_texel, a.t_sum, a.t_flag);
            }
        }
    }
    // ordered compaction into this workgroup's segment of the next queue
    uint32_t total;
    const uint32_t nout = (uint32_t)have[0] + (uint32_t)have[1] + (uint32_t)have[2] + (uint32_t)have[3];
    uint32_t base = block_scan_excl<SHADE_BLOCK>(nout, scan_tmp, &total);
    size_t at = (size_t)blockIdx.x * QUEUE_SEG + base;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (have[k]) {
            st_stream(&a.sA[at], outs[k].A);
            st_stream(&a.sB[at], outs[k].B);
            st_stream(&a.sC[at], outs[k].C);
            at++;
        }
    }
    if (threadIdx.x == 0) a.blockCounts[blockIdx.x] = total;
}

// reloadQueuedRays, Pipeline.inl:325-359: next count (clamped to currentRayLimit), pool cursor reset -- and the segment
// bases of the next queue (exclusive scan of the workgroups' output counts; bases[nb] = total)
template <int NT>
__global__ __launch_bounds__(NT) void rt_scan_blocks(const uint32_t* __restrict__ counts, uint32_t nb, uint32_t limit,
                                                       uint32_t* __restrict__ bases, uint32_t* __restrict__ cnt,
                                                       DevCounters* __restrict__ ctr, uint32_t* __restrict__ host_cnt) {
    __shared__ uint32_t tmp[33];
    const uint32_t tid = threadIdx.x;
    const uint32_t total = block_scan_array<NT>(counts, bases, nb, tmp);
    if (tid == 0) {
        bases[nb] = total;
        uint32_t next = total < limit ? total : limit;  // canonical overflow rule: rays past currentRayLimit are dropped
        if (total > limit) atomicAdd(&ctr->ray_limit_drops, (unsigned long long)(total - limit));
        cnt[1] = next;
        cnt[0] = next;
        cnt[2] = 0;  // chain pool cursor
        // the count the host's scheduler waits for (frames in flight), straight into its pinned slot: no copy launch
        // between this kernel and the event the host polls
        if (host_cnt) *(volatile uint32_t*)host_cnt = next;
    }
}

// the current queue in queue order (debug / parity download): A | B | C, m rays each
__global__ __launch_bounds__(256) void rt_gather_queue(RayQueue q, uint32_t total, uint32_t m, float4* __restrict__ dense) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    total = min(total, q.bases[q.nb]);
    if (i >= m || i >= total) return;
    const uint32_t loc = queue_loc(q.bases, q.nb, total, i);
    dense[i] = q.A[loc];
    dense[(size_t)m + i] = q.B[loc];
    dense[2 * (size_t)m + i] = q.C[loc];
}

// ---- sampler.comp:37-97 + deinterlace/filter copies ----------------------------------------------
__global__ __launch_bounds__(256) void rt_sample(uint32_t w, uint32_t h, uint32_t dw, uint32_t dh,
                                                 const float2* __restrict__ t_coord, const float4* __restrict__ t_sum,
                                                 const int32_t* __restrict__ t_flag, float4* __restrict__ presampled,
                                                 float4* __restrict__ filtered, int samples_lock) {
    uint32_t it = blockIdx.x * 256 + threadIdx.x;
    if (it >= dw * dh) return;
    int px = (int)(it % dw), py = (int)(it / dw);
    float ax = (float)w / (float)dw, ay = (float)h / (float)dh;
    int sclx = (int)ceilf(ax), scly = (int)ceilf(ay);
    int bx = (int)((float)px * ax), by = (int)((float)py * ay);
    int samplecount = 0;
    float n0 = 0.f, n1 = 0.f, n2 = 0.f;
    for (int x = -1; x <= sclx; x++) {
        for (int y = -1; y <= scly; y++) {
            int cx = bx + x, cy = by + y;
            if (cx >= 0 && cx < (int)w && cy >= 0 && cy < (int)h) {
                int ts = cy * (int)w + cx;
                // (the position test first, the flag only for a texel whose sample lands in this pixel: of the nine
                // candidates of a 1:1 grid one does, and the other eight no longer cost a flag read -- same condition)
                float2 co = t_coord[ts];
                float sx = co.x * (float)dw, sy = co.y * (float)dh;
                float dx = (sx - (float)px) + 0.00001f, dy = (sy - (float)py) + 0.00001f;
                if (dx >= 0.0f && dx < 1.0f && dy >= 0.0f && dy < 1.0f && t_flag[ts]) {
                    samplecount++;
                    float4 s = ld_stream(&t_sum[ts]);
                    n0 += s.x; n1 += s.y; n2 += s.z;
                }
            }
        }
    }
    float4 xs = ld_stream(&presampled[it]);
    if (samplecount > 0) {
        float sc = (float)samplecount;
        n0 = n0 / sc; n1 = n1 / sc; n2 = n2 / sc;
        float next = xs.w + sc;
        float prev = xs.w;
        float divisor = prev / next;
        xs.x = fmaf(xs.x, divisor, n0 * (1.0f - divisor));
        xs.y = fmaf(xs.y, divisor, n1 * (1.0f - divisor));
        xs.z = fmaf(xs.z, divisor, n2 * (1.0f - divisor));
        xs.w = (samples_lock > 0) ? pmin(next, (float)(samples_lock - 1)) : next;
        st_stream(&presampled[it], xs);
    }
    st_stream(&filtered[it], xs);
}

// dense <-> full-image copy of a tile's per-texel radiance (tile gather): one thread per texel OF THE TILE
// (dense index k = owned row * w + x), not per texel of the image
__global__ __launch_bounds__(256) void rt_pack(float4* __restrict__ t_sum, float4* __restrict__ buf, uint32_t w, uint32_t ntile,
                                               Tile tile, int unpack) {
    uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= ntile) return;
    uint32_t lr = k / w, x = k - lr * w;
    uint32_t idx = tile_owned_row(tile, lr) * w + x;
    if (unpack) t_sum[idx] = buf[k];
    else buf[k] = t_sum[idx];
}

// the gathering rank's side of the tile gather in ONE launch: `all` holds the dense tiles of ranks 0..world-1 of the
// dealing `map` back to back (stride floats4 each, what ncclGather delivers); every texel that `skip` does not
// own takes its radiance from its owner's tile. (One rt_pack launch per peer walked the whole image seven times.)
__global__ __launch_bounds__(256) void rt_unpack_all(float4* __restrict__ t_sum, const float4* __restrict__ all, uint32_t w,
                                                     uint32_t h, BandMap map, uint32_t skip, size_t stride) {
    uint32_t idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= w * h) return;
    uint32_t x = idx % w, y = idx / w;
    uint32_t g = y >> 3;
    uint32_t q = map.rank_of(g);
    if (q == skip) return;
    size_t d = (size_t)(map.local_band(g) * 8 + (y - g * 8)) * w + x;   // tile_row(): owned rows before the band + row in the band
    t_sum[idx] = all[(size_t)q * stride + d];
}

// ---- launch wrappers ------------------------------------------------------------------------------
static Tile tile_of(uint32_t mode, uint32_t a, uint32_t b, uint32_t h, const BandMap* map) {
    Tile t = {};
    t.mode = mode; t.a = a; t.b = b; t.h = h;
    if (mode && map) {
        t.map = *map;
        uint32_t n = 0;
        for (uint32_t p = 0; p < map->P; p++) if (map->owner[p] == a) t.pos[n++] = (uint8_t)p;
    }
    return t;
}
static Tile make_tile(const psm_rt* r) {
    return r->tile_mode ? tile_of(1u, r->tile_rank, r->tile_world, r->h, &r->bands) : tile_of(0u, r->y0, r->y1, r->h, nullptr);
}

uint32_t tile_texel_count(const psm_rt* r) {
    if (r->tile_mode == 0) return (r->y1 - r->y0) * r->w;
    return owned_texels(r->bands, r->tile_rank, r->w, r->h);
}

int launch_rt_pack(psm_rt* r, hipStream_t stream, float* d_buf, int unpack, const BandMap* bands, uint32_t a, uint32_t b) {
    psm_ctx* c = r->ctx;
    Tile t = tile_of(bands ? 1u : 0u, a, b, r->h, bands);
    uint32_t n = bands ? owned_texels(*bands, a, r->w, r->h) : (b - a) * r->w;  // texels of the tile
    if (n == 0) return PSM_OK;
    rt_pack<<<(n + 255) / 256, 256, 0, stream>>>(r->t_sum, (float4*)d_buf, r->w, n, t, unpack);
    PSM_HIP(c, hipGetLastError());
    return PSM_OK;
}

int launch_rt_unpack_all(psm_rt* r, hipStream_t stream, const float* d_all, const BandMap& bands, uint32_t skip, size_t stride_floats) {
    psm_ctx* c = r->ctx;
    uint32_t n = r->w * r->h;
    rt_unpack_all<<<(n + 255) / 256, 256, 0, stream>>>(r->t_sum, (const float4*)d_all, r->w, r->h, bands, skip, stride_floats / 4);
    PSM_HIP(c, hipGetLastError());
    return PSM_OK;
}

int launch_rt_camera(psm_rt* r, const float* cam_inv, const float* proj_inv, uint32_t time) {
    psm_ctx* c = r->ctx;
    Mat16 ci, pi;
    for (int i = 0; i < 16; i++) { ci.m[i] = cam_inv[i]; pi.m[i] = proj_inv[i]; }
    uint32_t n = r->w * r->h;
    TimedScope ts(c, CAT_CAMERA);
    uint32_t nrays = tile_texel_count(r);
    if (nrays)
        rt_camera<<<(nrays + 255) / 256, 256, 0, c->stream>>>(ci, pi, time, r->w, r->h, make_tile(r), nrays, r->qA[r->cur], r->qB[r->cur],
                                                          r->qC[r->cur], r->t_coord, r->t_sum, r->t_flag, r->d_cnt, r->q_bases[r->cur], r->enable360);
    else
        { PSM_HIP(c, hipMemsetAsync(r->d_cnt, 0, 3 * sizeof(uint32_t), c->stream)); PSM_HIP(c, hipMemsetAsync(r->q_bases[r->cur], 0, 2 * sizeof(uint32_t), c->stream)); }
    if (r->tile_root && nrays < n)
        rt_camera_rest<<<(n + 255) / 256, 256, 0, c->stream>>>(time, r->w, r->h, make_tile(r), r->t_coord, r->t_sum, r->t_flag);
    PSM_HIP(c, hipGetLastError());
    r->ray_count = nrays;
    r->count_valid = true;
    r->trav_n = 0;
    r->q_nb[r->cur] = 1;
    return PSM_OK;
}

static RayQueue current_queue(const psm_rt* r) {
    return RayQueue{r->qA[r->cur], r->qB[r->cur], r->qC[r->cur], r->q_bases[r->cur], r->q_nb[r->cur]};
}

int launch_rt_gather_queue(psm_rt* r, float4* d_dense, uint32_t m) {
    psm_ctx* c = r->ctx;
    if (m == 0) return PSM_OK;
    rt_gather_queue<<<(m + 255) / 256, 256, 0, c->stream>>>(current_queue(r), r->ray_count, m, d_dense);
    PSM_HIP(c, hipGetLastError());
    return PSM_OK;
}

int launch_rt_shade(psm_rt* r, psm_bvh* b, uint32_t time) {
    psm_ctx* c = r->ctx;
    uint32_t n = r->ray_count;
    if (n == 0) return PSM_OK;
    uint32_t nb = (n + SHADE_BLOCK - 1) / SHADE_BLOCK;
    ShadeArgs a;
    a.q = current_queue(r);
    a.hit0 = r->hit0; a.hitN = r->hitN; a.pool = r->pool;
    a.src.tri48 = b->d_tri48; a.src.nrm = b->d_nrm; a.src.tri_mats = b->d_mats; a.src.uv = b->d_tex;
    a.src.mats = r->d_mats; a.src.tex = r->d_tex_table; a.lights = r->d_lights;
    const int nxt_q = r->cur ^ 1;
    a.sA = r->qA[nxt_q]; a.sB = r->qB[nxt_q]; a.sC = r->qC[nxt_q];  // every workgroup writes its own segment of the next queue
    a.blockCounts = r->d_block;
    a.t_sum = r->t_sum; a.t_flag = r->t_flag;
    a.nrays = n; a.time = time;
    a.src.mat_offset = r->mat_offset; a.src.mat_count = (int)r->mat_count; a.light_count = (int)r->light_count;
    a.sky[0] = r->sky[0]; a.sky[1] = r->sky[1]; a.sky[2] = r->sky[2];
    a.sky_tex = r->d_sky; a.sky_w = (int)r->sky_w; a.sky_h = (int)r->sky_h;
    int nxt = r->cur ^ 1;
    const bool multi = r->trav_n > 1;
    a.src.geoms = r->d_geoms;
    if (multi) {  // hits carry an object tag: hand the kernel every traversed hierarchy's mosaics
        ObjGeom g[MAX_TRAV_OBJECTS] = {};
        for (int i = 0; i < r->trav_n; i++)
        {
            const psm_bvh* o = r->trav_objs[i];
            g[i] = ObjGeom{o->d_tri48, o->d_nrm, o->d_mats, o->d_tex};
        }
        PSM_HIP(c, hipMemcpyAsync(r->d_geoms, g, sizeof(g), hipMemcpyHostToDevice, c->stream));
        PSM_HIP(c, hipStreamSynchronize(c->stream));
    }
    r->trav_n = 0;  // the queue changes: the next intersection() starts new chains
    {
        TimedScope ts(c, CAT_SHADE);
        bool any_tex = false;
        for (int i = 1; i < MAX_TEXTURES; i++) any_tex = any_tex || r->tex_host[i].texels != nullptr;
        // (both lobes of every hit are built where a dropped lobe's colour could be NaN instead of 0: textures -- any texel may be
        // black metal or transparent --, materials that are not `ordinary`; see the kernel)
        if (any_tex && multi) rt_shade<true, true, true><<<nb, SHADE_BLOCK, 0, c->stream>>>(a);
        else if (any_tex) rt_shade<true, false, true><<<nb, SHADE_BLOCK, 0, c->stream>>>(a);
        else if (!r->mats_ordinary && multi) rt_shade<false, true, true><<<nb, SHADE_BLOCK, 0, c->stream>>>(a);
        else if (!r->mats_ordinary) rt_shade<false, false, true><<<nb, SHADE_BLOCK, 0, c->stream>>>(a);
        else if (multi) rt_shade<false, true, false><<<nb, SHADE_BLOCK, 0, c->stream>>>(a);
        else rt_shade<false, false, false><<<nb, SHADE_BLOCK, 0, c->stream>>>(a);
        rt_scan_blocks<1024>   // (256 threads: the same with frames in flight, slower alone)
           <<<1, 1024, 0, c->stream>>>(r->d_block, nb, r->limit, r->q_bases[nxt], r->d_cnt, c->d_counters, r->h_cnt);
    }
    r->q_nb[nxt] = nb;
    PSM_HIP(c, hipGetLastError());
    r->cur = nxt;
    r->count_valid = false;
    c->rounds++;
    return PSM_OK;
}

int launch_rt_sample(psm_rt* r, psm_rt* src) {
    psm_ctx* c = r->ctx;
    uint32_t n = r->dw * r->dh;
    TimedScope ts(c, CAT_SAMPLE);
    rt_sample<<<(n + 255) / 256, 256, 0, c->stream>>>(r->w, r->h, r->dw, r->dh, src->t_coord, src->t_sum, src->t_flag,
                                                      r->presampled, r->filtered, r->samples_lock);
    PSM_HIP(c, hipGetLastError());
    return PSM_OK;
}

}  // namespace psm
