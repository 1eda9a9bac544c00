/*
 * psm_oracle.h -- CPU restatement of the prismarine-core hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and there only as the checker.
 *
 * PARITY UNPINNED BY THE REFERENCE: the reference (EngineWorld/prismarine-core)
 * ships no tests, golden vectors or fixtures, and its GLSL compute path cannot
 * run here (no GL context, no glslc).  This file restates the GLSL text line
 * by line (file:line cited at every function) under the canonical determinism
 * rules written down in DESIGN.md; it is cross-checked by independent
 * properties in tests/ (naive bit-loop Morton, brute-force closest hit,
 * std-sort stability, BVH invariants), by second readings of the GLSL that
 * do not come from this file (tests/independent_{build,traverse,
 * camera_sampler,shade}.py), and its two host formulas are pinned by the
 * reference's own vendored glm (oracle/ref_glm).
 *
 * Conventions
 *   - matrices are row-major float[16]: (M v)[i] = sum_j M[4*i+j] v[j]
 *   - triangles are a soup: 9 floats per triangle (v0 xyz, v1 xyz, v2 xyz)
 *   - built with -O2 -ffp-contract=off: every float op below is one IEEE
 *     binary32 operation in the order written
 */
#ifndef PSM_ORACLE_H
#define PSM_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSMO_PZERO 0.0005f        /* include/constants.glsl:72 */
#define PSMO_INFINITY 10000.0f    /* include/constants.glsl:82 */
#define PSMO_STACK_CAP 16         /* directTraverse.comp:40-41 (8 LDS + 8 global) */
#define PSMO_BAKED_CAP 8          /* directTraverse.comp:42 */
#define PSMO_MAX_ITERS 8192       /* directTraverse.comp:383 */

/* HlbvhNode, include/structs.glsl:165-175 / Structs.hpp:125-128 */
typedef struct {
    uint32_t box[4];   /* packHalf2(mn).xy, packHalf2(mx).xy */
    int32_t pdata[4];  /* x,y children or x==y leaf; z parent; w triangle */
} psmo_node;

typedef struct {
    float u, v, t;
    int32_t tri;
} psmo_hit;

typedef struct {
    uint64_t node_visits;  /* V: internal-node visits (two child boxes fetched) */
    uint64_t tri_tests;    /* T: triangle vertex fetches */
    uint64_t stack_drops;  /* far children dropped at STACK_CAP */
    uint64_t iter_caps;    /* rays that ran into MAX_ITERS */
    uint64_t baked_drops;  /* equal-distance entries beyond BAKED_CAP */
} psmo_counters;

/* ---- small numerics (exported for KATs) ---- */
uint16_t psmo_f32_to_f16(float f);
float psmo_f16_to_f32(uint16_t h);
uint64_t psmo_morton3_64(uint32_t x, uint32_t y, uint32_t z);
uint32_t psmo_hash(uint32_t x);
float psmo_sinf(float x);
float psmo_cosf(float x);
float psmo_powf(float x, float y);
float psmo_atan2f(float y, float x);
float psmo_asinf(float x);

/* ---- build ---- */
void psmo_minmax(const float* tris, int n, const float M[16], float mn[4], float mx[4]);
void psmo_fit_transform(const float mn[4], const float mx[4], const double opt[16],
                        float M[16], float Minv[16]);
void psmo_inverse_opt(const double opt[16], float M[16]);
int psmo_morton_leaves(const float* tris, int n, const float M[16], uint64_t* keys,
                       int32_t* idx, psmo_node* leafs);
void psmo_radix_sort(uint64_t* keys, int32_t* vals, int n);
int psmo_find_split(const uint64_t* keys, int first, int last, uint64_t* key_reads);
int psmo_build_nodes(const uint64_t* keys, const int32_t* idx, psmo_node* leafs, int n,
                     psmo_node* nodes, int* levels, uint64_t* key_reads);
/* whole TriangleHierarchy::build; returns leaf count; nodes sized 2*n */
void psmo_refit(const float* tris, const float M[16], psmo_node* leafs, int nleafs, psmo_node* nodes, int nnodes);
int psmo_build(const float* tris, int n, const double opt[16], float M[16], uint64_t* keys,
               int32_t* idx, psmo_node* leafs, psmo_node* nodes);

/* ---- geometry ingestion: vertex/loader.comp:32-152 ---- */
typedef struct { int32_t offset4, components, buffer_view; } psmo_accessor;
typedef struct { int32_t offset4, stride4; } psmo_buffer_view;
typedef struct {
    const float* vertices; size_t vertex_floats;
    const uint32_t* indices; size_t index_words;
    const psmo_accessor* accessors; uint32_t accessor_count;
    const psmo_buffer_view* views; uint32_t view_count;
    int32_t vertex_accessor, normal_accessor, texcoord_accessor, modifier_accessor;
    float transform[16], transform_inv[16];
    int32_t material_id, is_indexed, index16, node_count, primitive_type, loading_offset;
} psmo_mesh_desc;
/* appends to pos/nrm (9 floats per triangle) and mats starting at triangle `storing_offset`; returns triangles written */
int psmo_load_mesh(const psmo_mesh_desc* d, int storing_offset, float* pos, float* nrm, int32_t* mats);
/* same, also writing the texcoord mosaic (6 floats per triangle: u,v per vertex; v stored as 1 - v) */
int psmo_load_mesh_tex(const psmo_mesh_desc* d, int storing_offset, float* pos, float* nrm, int32_t* mats, float* tex);

/* ---- trace ---- */
int psmo_traverse(const psmo_node* nodes, const float* tris, const float M[16],
                  const float origin[3], const float direct[3], psmo_hit out[PSMO_BAKED_CAP],
                  psmo_counters* ctr);
void psmo_traverse_batch(const psmo_node* nodes, const float* tris, const float M[16],
                         const float* origins, const float* directs, int nrays,
                         psmo_hit* hits /* nrays*8 */, int32_t* counts, psmo_counters* ctr,
                         int nthreads);
void psmo_traverse_batch_ex(const psmo_node* nodes, const float* tris, const float M[16],
                            const float* origins, const float* directs, int nrays, psmo_hit* hits,
                            int32_t* counts, psmo_counters* ctr, int nthreads, uint32_t* per_ray_visits,
                            uint32_t* per_ray_tests);
int psmo_traverse_from(const psmo_node* nodes, const float* tris, const float M[16],
                       const float origin[3], const float direct[3], float start_dist,
                       psmo_hit out[PSMO_BAKED_CAP], psmo_counters* ctr);
/* multi-BVH: extend the chains in hits/counts (in/out) with another hierarchy's hits, ids offset by tri_base */
void psmo_traverse_chain_batch(const psmo_node* nodes, const float* tris, const float M[16],
                               const float* origins, const float* directs, int nrays, psmo_hit* hits,
                               int32_t* counts, int tri_base, psmo_counters* ctr, int nthreads);
int psmo_brute_force(const float* tris, int ntris, const float origin[3],
                     const float direct[3], psmo_hit* best);

/* ---- wavefront loop ---- */
typedef struct {
    float origin[3];
    float direct[3];
    float color[3];
    int32_t bitfield;
    int32_t texel;
    uint32_t pkey;
} psmo_ray;

typedef struct {
    float diffuse[4];
    float specular[4];
    float transmission[4];
    float emissive[4];
    float ior, roughness, alpharef, unk0f;
    uint32_t diffusePart, specularPart, bumpPart, emissivePart;
    int32_t flags, alphafunc, binding, bitfield;
    int32_t iModifiers0[4];
} psmo_material; /* VirtualMaterial, Structs.hpp:240-262 (128 B) */

typedef struct {
    float lightVector[4];
    float lightColor[4];
    float lightOffset[4];
    float lightAmbient[4];
} psmo_light; /* LightUniformStruct, Structs.hpp:165-170 */

typedef struct {
    const uint8_t* rgba8;          /* NULL = empty slot */
    int w, h;
} psmo_texture; /* RGBA8, GL_LINEAR, GL_REPEAT (TextureSet.inl:113-118) */

typedef struct {
    int width, height;             /* ray grid (sceneRes) */
    int display_width, display_height;
    int light_count;
    int material_offset, material_count;
    float sky[4];                  /* constant environment colour */
    int ray_limit;                 /* currentRayLimit, Pipeline.inl:187-189 */
    int samples_lock;              /* SAMPLES_LOCK, constants.glsl:35 (4) */
    const uint8_t* sky_tex;        /* optional equirect RGBA8 skybox (NULL = constant sky) */
    int sky_w, sky_h;
    const float* texcoords;        /* optional, 6 floats / triangle (u,v per vertex); NULL = all (0,0) */
    psmo_texture textures[32];     /* sampler table, surface.comp:46-52; slot 0 unused */
    int enable360;                 /* cameraUniform.enable360 (switchMode, Pipeline.inl:128-132) */
} psmo_frame_cfg;

int psmo_camera(const psmo_frame_cfg* cfg, const float camInv[16], const float projInv[16],
                uint32_t time, int y0, int y1, psmo_ray* rays, float* texel_coord,
                float* texel_sum, int32_t* texel_flag);
int psmo_band_pattern(int world, const uint32_t* weights, uint8_t pattern[64]);
int psmo_camera_weighted(const psmo_frame_cfg* cfg, const float camInv[16], const float projInv[16],
                         uint32_t time, int rank, int world, const uint32_t* weights, psmo_ray* rays, float* texel_coord,
                         float* texel_sum, int32_t* texel_flag);
int psmo_camera_interleaved(const psmo_frame_cfg* cfg, const float camInv[16], const float projInv[16],
                            uint32_t time, int rank, int world, psmo_ray* rays, float* texel_coord,
                            float* texel_sum, int32_t* texel_flag);
int psmo_shade(const psmo_frame_cfg* cfg, const psmo_light* lights, const psmo_material* mats,
               const int32_t* tri_mats, const float* tris, const float* normals, uint32_t time,
               const psmo_ray* rays, int nrays, const psmo_hit* hits, const int32_t* counts,
               psmo_ray* out_rays, float* texel_sum, int32_t* texel_flag);
void psmo_sample(const psmo_frame_cfg* cfg, const float* texel_coord, const float* texel_sum,
                 const int32_t* texel_flag, float* presampled, float* filtered);
uint32_t psmo_rand_next(uint32_t* state);

#ifdef __cplusplus
}
#endif
#endif
