// api.hip -- the C ABI of include/psm_hip.h: contexts, buffers, object lifetime, stage order.
// Host logic only; the kernels live in sort.hip, bvh.hip, trace.hip, shade.hip.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <new>

#include "psm_common.h"
#include "psm_internal.h"

namespace psm {

constexpr int SM_COUNT = 24;
constexpr int SM_ROOT = 25;
constexpr int SM_BFLOAT = 26;
constexpr int SM_WORDS = 64;

int set_err(psm_ctx* c, int code, const char* what, hipError_t e) {
    if (c) {
        c->err = what ? what : "";
        if (e != hipSuccess) {
            c->err += ": ";
            c->err += hipGetErrorString(e);
        }
    }
    return code;
}

TimedScope::TimedScope(psm_ctx* ctx, int cat) : c(ctx) {
    if (!c->timing || (c->timing == 2 && cat != CAT_TRAVERSE && cat != CAT_TRAVERSE_HANDOVER)) return;
    psm_ctx::Timed t;
    for (hipEvent_t* ev : {&t.a, &t.b}) {
        if (!c->free_events.empty()) {
            *ev = c->free_events.back();
            c->free_events.pop_back();
        } else if (hipEventCreate(ev) != hipSuccess) {
            return;
        }
    }
    t.cat = cat;
    (void)hipEventRecord(t.a, c->stream);
    c->timed.push_back(t);
    idx = (int)c->timed.size() - 1;
}
TimedScope::~TimedScope() {
    if (idx >= 0) (void)hipEventRecord(c->timed[(size_t)idx].b, c->stream);
}

static void collect_timing(psm_ctx* c) {
    for (auto& t : c->timed) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) {
            c->cat_ms[t.cat] += ms;
            c->cat_launches[t.cat]++;
            if (t.cat == CAT_TRAVERSE_HANDOVER) { c->cat_ms[CAT_TRAVERSE] += ms; c->cat_launches[CAT_TRAVERSE]++; }
            if (c->ref_event && (t.cat == CAT_TRAVERSE || t.cat == CAT_TRAVERSE_HANDOVER)) {
                float t0 = 0.f;
                if (hipEventElapsedTime(&t0, c->ref_event, t.a) == hipSuccess) { c->intervals.push_back(t0); c->intervals.push_back(t0 + ms); }
                else (void)hipGetLastError();
            }
        }
        c->free_events.push_back(t.a);
        c->free_events.push_back(t.b);
    }
    c->timed.clear();
}

template <typename T>
static int dev_alloc(psm_ctx* c, T** p, size_t count) {
    *p = nullptr;
    if (count == 0) count = 1;
    PSM_HIP(c, hipMalloc((void**)p, count * sizeof(T)));
    return PSM_OK;
}
template <typename T>
static void dev_free(T*& p) {
    if (p) (void)hipFree(p);
    p = nullptr;
}

}  // namespace psm

using namespace psm;

extern "C" {

int psm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static int ctx_create(int device, void* ext_stream, bool use_ext, psm_ctx** out);
int psm_ctx_create(int device, psm_ctx** out) { return ctx_create(device, nullptr, false, out); }
int psm_ctx_create_on_stream(int device, void* hip_stream, psm_ctx** out) { return ctx_create(device, hip_stream, true, out); }

static int ctx_create(int device, void* ext_stream, bool use_ext, psm_ctx** out) {
    if (!out) return PSM_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return PSM_ERR_NO_DEVICE;
    if (device < 0 || device >= n) return PSM_ERR_INVALID;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return PSM_ERR_HIP;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        std::fprintf(stderr, "psm: device %d is %s; this library is built for gfx950 only\n", device, prop.gcnArchName);
        return PSM_ERR_NO_DEVICE;
    }
    psm_ctx* c = new (std::nothrow) psm_ctx();
    if (!c) return PSM_ERR_INVALID;
    c->device = device;
    c->own_stream = !use_ext;
    if (use_ext) c->stream = (hipStream_t)ext_stream;
    if (hipSetDevice(device) != hipSuccess ||
        (!use_ext && hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) ||
        hipMalloc((void**)&c->d_counters, sizeof(DevCounters)) != hipSuccess ||
        hipMemsetAsync(c->d_counters, 0, sizeof(DevCounters), c->stream) != hipSuccess) {
        delete c;
        return PSM_ERR_HIP;
    }
    if (const char* t = std::getenv("PSM_SORT_TUNE")) {   // study knob (tools/sort_bench.py): "S_small,S_large,threads[,cap_small,cap_large[,threads_large]]" of radix_local
        unsigned a = 0, b = 0, th = 0, cs = 4096, cl = 4096, thl = 0;
        const int got = std::sscanf(t, "%u,%u,%u,%u,%u,%u", &a, &b, &th, &cs, &cl, &thl);
        if (got >= 3 && a >= 64 && a < cs && b >= 64 && b < cl && (th == 512 || th == 1024) && (thl == 0 || thl == 512 || thl == 1024)) {
            c->sort_hybrid_s_small = a; c->sort_hybrid_s_large = b; c->sort_hybrid_threads = th; c->sort_hybrid_threads_large = thl ? thl : th;
            c->sort_hybrid_cap_small = cs; c->sort_hybrid_cap_large = cl;
        }
    }
    *out = c;
    return PSM_OK;
}

int psm_ctx_destroy(psm_ctx* c) {
    if (!c) return PSM_ERR_INVALID;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    collect_timing(c);
    for (auto e : c->free_events) (void)hipEventDestroy(e);
    if (c->ref_event && c->ref_owner == c) (void)hipEventDestroy(c->ref_event);
    for (auto& b : c->bufs) if (b.ptr) (void)hipFree(b.ptr);
    dev_free(c->sort_keys_tmp);
    dev_free(c->sort_vals_tmp);
    dev_free(c->sort_hist);
    if (c->sort_overflow) (void)hipHostFree(c->sort_overflow);
    dev_free(c->d_counters);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return PSM_OK;
}

// Achievable HBM ceiling of this box: device-to-device copy of `bytes` (read + write = 2 x bytes of traffic),
// best of `reps`, timed with HIP events on the context's stream (report the roofline fraction
// against a measured ceiling as well as the vendor peak).
int psm_ctx_copy_bandwidth(psm_ctx* c, size_t bytes, int reps, double* gb_per_s) {
    if (!c || !gb_per_s || bytes == 0 || reps <= 0) return PSM_ERR_INVALID;
    (void)hipSetDevice(c->device);
    void *a = nullptr, *b = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = PSM_OK;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess ||
        hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) rc = PSM_ERR_HIP;
    float best = 0.f;
    if (rc == PSM_OK) {
        (void)hipMemsetAsync(a, 1, bytes, c->stream);
        (void)hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, c->stream);  // warm-up
        for (int i = 0; i < reps && rc == PSM_OK; i++) {
            (void)hipEventRecord(e0, c->stream);
            (void)hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, c->stream);
            (void)hipEventRecord(e1, c->stream);
            if (hipEventSynchronize(e1) != hipSuccess) { rc = PSM_ERR_HIP; break; }
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms > 0.f && (best == 0.f || ms < best)) best = ms;
        }
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    if (rc != PSM_OK) return set_err(c, rc, "psm_ctx_copy_bandwidth");
    *gb_per_s = best > 0.f ? 2.0 * (double)bytes / ((double)best * 1e-3) / 1e9 : 0.0;
    return PSM_OK;
}

int psm_ctx_sync(psm_ctx* c) {
    if (!c) return PSM_ERR_INVALID;
    PSM_HIP(c, hipStreamSynchronize(c->stream));
    return sort_check(c);  // the radix sort's look-back raises a device word instead of hanging
}
void* psm_ctx_stream(psm_ctx* c) { return c ? (void*)c->stream : nullptr; }
const char* psm_last_error(psm_ctx* c) { return c ? c->err.c_str() : "null context"; }

// ---- buffers ------------------------------------------------------------------------------------
int psm_buf_alloc(psm_ctx* c, size_t bytes, uint32_t* handle) {
    if (!c || !handle) return PSM_ERR_INVALID;
    (void)hipSetDevice(c->device);
    Buf b;
    b.bytes = bytes;
    PSM_HIP(c, hipMalloc(&b.ptr, bytes ? bytes : 1));
    for (size_t i = 0; i < c->bufs.size(); i++) {
        if (!c->bufs[i].ptr) {
            c->bufs[i] = b;
            *handle = (uint32_t)i + 1;
            return PSM_OK;
        }
    }
    c->bufs.push_back(b);
    *handle = (uint32_t)c->bufs.size();
    return PSM_OK;
}
static Buf* get_buf(psm_ctx* c, uint32_t h) {
    if (!c || h == 0 || h > c->bufs.size() || !c->bufs[h - 1].ptr) return nullptr;
    return &c->bufs[h - 1];
}
int psm_buf_free(psm_ctx* c, uint32_t h) {
    Buf* b = get_buf(c, h);
    if (!b) return c ? set_err(c, PSM_ERR_INVALID, "psm_buf_free: bad handle") : PSM_ERR_INVALID;
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(b->ptr);
    b->ptr = nullptr;
    b->bytes = 0;
    return PSM_OK;
}
int psm_buf_upload(psm_ctx* c, uint32_t h, size_t offset, const void* src, size_t bytes) {
    Buf* b = get_buf(c, h);
    if (!b || !src || offset + bytes > b->bytes) return c ? set_err(c, PSM_ERR_INVALID, "psm_buf_upload: bad handle or range") : PSM_ERR_INVALID;
    PSM_HIP(c, hipMemcpyAsync((char*)b->ptr + offset, src, bytes, hipMemcpyHostToDevice, c->stream));
    PSM_HIP(c, hipStreamSynchronize(c->stream));  // src may be pageable host memory
    return PSM_OK;
}
int psm_buf_download(psm_ctx* c, uint32_t h, size_t offset, void* dst, size_t bytes) {
    Buf* b = get_buf(c, h);
    if (!b || !dst || offset + bytes > b->bytes) return c ? set_err(c, PSM_ERR_INVALID, "psm_buf_download: bad handle or range") : PSM_ERR_INVALID;
    PSM_HIP(c, hipMemcpyAsync(dst, (char*)b->ptr + offset, bytes, hipMemcpyDeviceToHost, c->stream));
    PSM_HIP(c, hipStreamSynchronize(c->stream));
    return sort_check(c);
}
int psm_buf_ptr(psm_ctx* c, uint32_t h, void** p, size_t* bytes) {
    Buf* b = get_buf(c, h);
    if (!b) return c ? set_err(c, PSM_ERR_INVALID, "psm_buf_ptr: bad handle") : PSM_ERR_INVALID;
    if (p) *p = b->ptr;
    if (bytes) *bytes = b->bytes;
    return PSM_OK;
}

// ---- sort -----------------------------------------------------------------------------------------
int psm_sort_set_algorithm(psm_ctx* c, int algorithm) {
    if (!c || algorithm < 0 || algorithm > 2) return PSM_ERR_INVALID;
    c->sort_algorithm = algorithm;
    c->sort_demoted = false;   // asking for the hybrid sort again gives it another chance
    if (c->sort_overflow) {
        (void)hipStreamSynchronize(c->stream);
        *c->sort_overflow = 0u;
    }
    return PSM_OK;
}
int psm_sort_get_algorithm(psm_ctx* c, int* asked, int* effective) {
    if (!c) return PSM_ERR_INVALID;
    if (asked) *asked = c->sort_algorithm;
    if (effective) *effective = sort_effective_algorithm(c);
    return PSM_OK;
}
int psm_sort_u64_u32_dev(psm_ctx* c, uint64_t* d_keys, uint32_t* d_vals, size_t n) {
    if (!c || (n && (!d_keys || !d_vals))) return PSM_ERR_INVALID;
    (void)hipSetDevice(c->device);
    return launch_sort(c, d_keys, d_vals, n, nullptr);
}
int psm_sort_u64_u32(psm_ctx* c, uint32_t hk, uint32_t hv, uint32_t n) {
    Buf* k = get_buf(c, hk);
    Buf* v = get_buf(c, hv);
    if (!k || !v) return c ? set_err(c, PSM_ERR_INVALID, "psm_sort: bad handle") : PSM_ERR_INVALID;
    if ((size_t)n * 8 > k->bytes || (size_t)n * 4 > v->bytes) return set_err(c, PSM_ERR_CAPACITY, "psm_sort: n exceeds buffer");
    return psm_sort_u64_u32_dev(c, (uint64_t*)k->ptr, (uint32_t*)v->ptr, n);
}

// ---- TriangleHierarchy ----------------------------------------------------------------------------
int psm_bvh_destroy(psm_bvh* b) {
    if (!b) return PSM_ERR_INVALID;
    (void)hipSetDevice(b->ctx->device);
    (void)hipStreamSynchronize(b->ctx->stream);
    dev_free(b->d_pos); dev_free(b->d_nrm); dev_free(b->d_mats); dev_free(b->d_tri48); dev_free(b->d_tex);
    dev_free(b->d_keys); dev_free(b->d_idx); dev_free(b->d_leafbox); dev_free(b->d_leaftri);
    dev_free(b->d_block); dev_free(b->d_small); dev_free(b->d_opt); dev_free(b->d_seg);
    dev_free(b->d_sorted_tri); dev_free(b->d_pairbox); dev_free(b->d_link); dev_free(b->d_range); dev_free(b->d_node32);
    if (b->build_graph) (void)hipGraphExecDestroy(b->build_graph);
    delete b;
    return PSM_OK;
}

int psm_bvh_create(psm_ctx* c, size_t max_tris, psm_bvh** out) {
    if (!c || !out || max_tris == 0) return PSM_ERR_INVALID;
    // 2^27 triangles (the reference: ~4.19 M, TriangleHierarchy.inl:80): the traversal kernel addresses its 32-byte node
    // records with a 32-bit byte offset, and 9 floats per triangle stay below 2^31 elements
    if (max_tris > (1ull << 27)) return set_err(c, PSM_ERR_CAPACITY, "psm_bvh_create: max_tris exceeds 2^27 (32-bit node offsets)");
    (void)hipSetDevice(c->device);
    psm_bvh* b = new (std::nothrow) psm_bvh();
    if (!b) return PSM_ERR_INVALID;
    b->ctx = c;
    b->cap = max_tris;
    size_t n = max_tris;
    // segment tree level offsets: level L has ceil(n / 2^L) entries
    size_t off = 0;
    for (size_t cnt = n;; cnt = (cnt + 1) / 2) {
        b->seg_off.push_back(off);
        off += cnt;
        if (cnt == 1) break;
    }
    b->seg_off.push_back(off);
    int rc = PSM_OK;
    auto A = [&](int r) { if (rc == PSM_OK) rc = r; };
    A(dev_alloc(c, &b->d_pos, 9 * n)); A(dev_alloc(c, &b->d_nrm, 9 * n)); A(dev_alloc(c, &b->d_mats, n));
    A(dev_alloc(c, &b->d_tri48, 3 * n));
    A(dev_alloc(c, &b->d_tex, 6 * n)); A(dev_alloc(c, &b->d_keys, n)); A(dev_alloc(c, &b->d_idx, n));
    A(dev_alloc(c, &b->d_leafbox, n)); A(dev_alloc(c, &b->d_leaftri, n));
    A(dev_alloc(c, &b->d_block, (n + 255) / 256 + 1)); A(dev_alloc(c, &b->d_small, (size_t)SM_WORDS));
    A(dev_alloc(c, &b->d_opt, (size_t)16)); A(dev_alloc(c, &b->d_seg, off));
    A(dev_alloc(c, &b->d_sorted_tri, n)); A(dev_alloc(c, &b->d_pairbox, 2 * n)); A(dev_alloc(c, &b->d_link, n));
    A(dev_alloc(c, &b->d_range, n));
    A(dev_alloc(c, &b->d_node32, 2 * n));
    if (rc != PSM_OK) { psm_bvh_destroy(b); return rc; }
    if (hipMemsetAsync(b->d_small, 0, SM_WORDS * 4, c->stream) != hipSuccess ||
        hipMemsetAsync(b->d_tex, 0, 6 * n * sizeof(float), c->stream) != hipSuccess) { psm_bvh_destroy(b); return PSM_ERR_HIP; }
    *out = b;
    return PSM_OK;
}

int psm_bvh_clear(psm_bvh* b) {
    if (!b) return PSM_ERR_INVALID;
    b->tri_count = 0;
    b->built = b->bounds_done = b->morton_done = b->sort_done = false;
    return PSM_OK;
}

int psm_bvh_load_triangles(psm_bvh* b, const float* positions, const float* normals, const int32_t* mats, size_t n,
                           int32_t material_id) {
    if (!b || (n && !positions)) return PSM_ERR_INVALID;
    psm_ctx* c = b->ctx;
    (void)hipSetDevice(c->device);
    if (b->tri_count + n > b->cap) return set_err(c, PSM_ERR_CAPACITY, "psm_bvh_load_triangles: exceeds allocate() capacity");
    if (n == 0) return PSM_OK;
    uint32_t first = b->tri_count;
    PSM_HIP(c, hipMemcpyAsync(b->d_pos + (size_t)9 * first, positions, n * 9 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    std::vector<float> fn;
    if (!normals) {
        // loader.comp:119-128: no normal accessor -> normalize(normalize(cross(v1-v0, v2-v0)))
        fn.resize(n * 9);
        for (size_t t = 0; t < n; t++) {
            const float* p = positions + 9 * t;
            v3 e1 = mk3(p[3] - p[0], p[4] - p[1], p[5] - p[2]), e2 = mk3(p[6] - p[0], p[7] - p[1], p[8] - p[2]);
            v3 nn = normalize3(normalize3(cross3(e1, e2)));
            for (int k = 0; k < 3; k++) { fn[9 * t + 3 * k] = nn.x; fn[9 * t + 3 * k + 1] = nn.y; fn[9 * t + 3 * k + 2] = nn.z; }
        }
        normals = fn.data();
    }
    PSM_HIP(c, hipMemcpyAsync(b->d_nrm + (size_t)9 * first, normals, n * 9 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    std::vector<int32_t> mm;
    if (!mats) { mm.assign(n, material_id); mats = mm.data(); }
    PSM_HIP(c, hipMemcpyAsync(b->d_mats + first, mats, n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    PSM_HIP(c, hipStreamSynchronize(c->stream));  // sources may be pageable / temporaries
    b->tri_count += (uint32_t)n;
    b->built = b->bounds_done = b->morton_done = b->sort_done = false;
    return launch_bvh_prepare_tris(b, first, (uint32_t)n);
}

int psm_bvh_set_texcoords(psm_bvh* b, size_t first, const float* uv, size_t n) {
    if (!b || (n && !uv)) return PSM_ERR_INVALID;
    psm_ctx* c = b->ctx;
    (void)hipSetDevice(c->device);
    if (first + n > b->cap) return set_err(c, PSM_ERR_CAPACITY, "psm_bvh_set_texcoords: range exceeds capacity");
    if (n == 0) return PSM_OK;
    PSM_HIP(c, hipMemcpyAsync(b->d_tex + 6 * first, uv, n * 6 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    PSM_HIP(c, hipStreamSynchronize(c->stream));
    return PSM_OK;
}

int psm_bvh_load_mesh(psm_bvh* b, const psm_mesh_desc* d) {
    if (!b || !d) return PSM_ERR_INVALID;
    psm_ctx* c = b->ctx;
    (void)hipSetDevice(c->device);
    if (d->node_count <= 0) return PSM_OK;  // TriangleHierarchy.inl:174
    if (!d->d_vertices || !d->accessors || !d->views || d->vertex_accessor < 0 ||
        (uint32_t)d->vertex_accessor >= d->accessor_count || d->normal_accessor >= (int32_t)d->accessor_count ||
        d->texcoord_accessor >= (int32_t)d->accessor_count ||
        (d->is_indexed && !d->d_indices))
        return set_err(c, PSM_ERR_INVALID, "psm_bvh_load_mesh: bad mesh description");
    for (uint32_t i = 0; i < d->accessor_count; i++)
        if (d->accessors[i].buffer_view < 0 || (uint32_t)d->accessors[i].buffer_view >= d->view_count)
            return set_err(c, PSM_ERR_INVALID, "psm_bvh_load_mesh: accessor refers to a missing buffer view");
    size_t tris = (size_t)d->node_count * (d->primitive_type == 1 ? 2 : 1);
    if (b->tri_count + tris > b->cap) return set_err(c, PSM_ERR_CAPACITY, "psm_bvh_load_mesh: exceeds allocate() capacity");
    psm_accessor* d_acc = nullptr;
    psm_buffer_view* d_views = nullptr;
    PSM_HIP(c, hipMalloc((void**)&d_acc, d->accessor_count * sizeof(psm_accessor)));
    if (hipMalloc((void**)&d_views, d->view_count * sizeof(psm_buffer_view)) != hipSuccess) { (void)hipFree(d_acc); return set_err(c, PSM_ERR_HIP, "hipMalloc"); }
    int rc = PSM_OK;
    if (hipMemcpyAsync(d_acc, d->accessors, d->accessor_count * sizeof(psm_accessor), hipMemcpyHostToDevice, c->stream) != hipSuccess ||
        hipMemcpyAsync(d_views, d->views, d->view_count * sizeof(psm_buffer_view), hipMemcpyHostToDevice, c->stream) != hipSuccess)
        rc = set_err(c, PSM_ERR_HIP, "psm_bvh_load_mesh: upload of the accessor tables failed");
    if (rc == PSM_OK) rc = launch_bvh_load_mesh(b, d, d_acc, d_views);
    (void)hipStreamSynchronize(c->stream);  // the tables are temporaries
    (void)hipFree(d_acc);
    (void)hipFree(d_views);
    if (rc == PSM_OK) {
        b->tri_count += (uint32_t)tris;
        b->built = b->bounds_done = b->morton_done = b->sort_done = false;
    }
    return rc;
}

// the optimisation matrix of the next build on the device (NULL: identity). A frame loop rebuilds with the same matrix every
// time: the 10-us copy in front of the rebuild is made only when the matrix has changed -- and then behind a synchronisation,
// because the copy reads the hierarchy's own host copy, which an earlier copy still in flight must not see overwritten.
static int bvh_upload_opt(psm_bvh* b, const double* opt) {
    psm_ctx* c = b->ctx;
    static const double ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    const double* want = opt ? opt : ident;
    if (b->opt_uploaded && memcmp(b->opt_host, want, sizeof(b->opt_host)) == 0) return PSM_OK;
    PSM_HIP(c, hipStreamSynchronize(c->stream));
    b->opt_uploaded = false;
    memcpy(b->opt_host, want, sizeof(b->opt_host));
    PSM_HIP(c, hipMemcpyAsync(b->d_opt, b->opt_host, 16 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    { int rc = launch_bvh_opt_changed(b); if (rc != PSM_OK) return rc; }   // first-pass transform + the bounds reduction's neutral elements: per matrix, not per build
    b->opt_uploaded = true;
    return PSM_OK;
}

int psm_bvh_stage_bounds(psm_bvh* b, const double* opt) {
    if (!b) return PSM_ERR_INVALID;
    psm_ctx* c = b->ctx;
    (void)hipSetDevice(c->device);
    if (b->tri_count == 0) return set_err(c, PSM_ERR_STATE, "build: no triangles");
    { int rc = bvh_upload_opt(b, opt); if (rc != PSM_OK) return rc; }
    b->built = false;  // a new build has begun: the node records of the last one can no longer be produced (psm_bvh_download)
    b->topo_tris = 0;  // ... nor can it be refitted: its transform and keys are being replaced
    int rc = launch_bvh_bounds(b);
    if (rc == PSM_OK) b->bounds_done = true;
    return rc;
}
int psm_bvh_stage_morton(psm_bvh* b) {
    if (!b) return PSM_ERR_INVALID;
    (void)hipSetDevice(b->ctx->device);
    if (!b->bounds_done) return set_err(b->ctx, PSM_ERR_STATE, "morton before bounds");
    int rc = launch_bvh_morton(b);
    if (rc == PSM_OK) b->morton_done = true;
    return rc;
}
int psm_bvh_stage_sort(psm_bvh* b) {
    if (!b) return PSM_ERR_INVALID;
    (void)hipSetDevice(b->ctx->device);
    if (!b->morton_done) return set_err(b->ctx, PSM_ERR_STATE, "sort before morton");
    int rc = launch_sort(b->ctx, b->d_keys, b->d_idx, b->tri_count, b->d_small + SM_COUNT, 63);   // Morton codes: 3 x 21 bits
    if (rc == PSM_OK) b->sort_done = true;
    return rc;
}
int psm_bvh_stage_emit(psm_bvh* b) {
    if (!b) return PSM_ERR_INVALID;
    (void)hipSetDevice(b->ctx->device);
    if (!b->sort_done) return set_err(b->ctx, PSM_ERR_STATE, "emit before sort");
    int rc = launch_bvh_emit(b);
    if (rc == PSM_OK) { b->built = true; b->topo_tris = b->tri_count; }
    return rc;
}

// Refit only (no entry point of the reference does this on its own: its refit is the last stage of build()): the
// triangles of a hierarchy that was built have moved -- same count, reloaded in the same order -- and only the boxes are
// recomputed: leaf boxes (aabbmaker.comp:165-194) with the build's transform, every node's child boxes bottom-up
// (refit.comp:21-114). Topology, ranges and triangle ids stay the build's (a triangle the build dropped as degenerate stays
// dropped, one that has become degenerate keeps its leaf). C3: 0.04 ms against the rebuild's 0.14.
int psm_bvh_refit(psm_bvh* b) {
    if (!b) return PSM_ERR_INVALID;
    psm_ctx* c = b->ctx;
    (void)hipSetDevice(c->device);
    if (b->tri_count == 0 || b->topo_tris != b->tri_count)
        return set_err(c, PSM_ERR_STATE, "psm_bvh_refit: no complete build of this triangle count to refit (build first; reload the same number of triangles)");
    TimedScope ts(c, CAT_BUILD);
    int rc = launch_bvh_refit_leaves(b);
    if (rc == PSM_OK) rc = launch_bvh_emit(b);
    if (rc == PSM_OK) b->bounds_done = b->morton_done = b->sort_done = b->built = true;
    return rc;
}
static int bvh_build_plain(psm_bvh* b, const double* opt) {
    TimedScope ts(b->ctx, CAT_BUILD);
    int rc = psm_bvh_stage_bounds(b, opt);
    if (rc == PSM_OK) rc = psm_bvh_stage_morton(b);
    if (rc == PSM_OK) rc = psm_bvh_stage_sort(b);
    if (rc == PSM_OK) rc = psm_bvh_stage_emit(b);
    return rc;
}

static void bvh_drop_graph(psm_bvh* b) {
    if (b->build_graph) (void)hipGraphExecDestroy(b->build_graph);
    b->build_graph = nullptr;
}

// Capture the build's launches (bounds, Morton + leaves, the sort's passes, segment tree, emit: 34 at C3) on the
// context's stream into one executable graph. Every argument is a device pointer of this hierarchy / context or a
// function of the triangle count, so the graph stays valid until one of them changes (checked by the caller).
static int bvh_capture_graph(psm_bvh* b) {
    psm_ctx* c = b->ctx;
    bvh_drop_graph(b);
    int rc = sort_reserve(c, b->tri_count);  // nothing may be allocated while the stream is capturing
    if (rc != PSM_OK) return rc;
    if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); return PSM_ERR_HIP; }
    rc = launch_bvh_bounds(b);
    if (rc == PSM_OK) rc = launch_bvh_morton(b);
    if (rc == PSM_OK) rc = launch_sort(c, b->d_keys, b->d_idx, b->tri_count, b->d_small + SM_COUNT, 63);
    if (rc == PSM_OK) rc = launch_bvh_emit(b);
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(c->stream, &g);
    if (rc == PSM_OK && e == hipSuccess && g) e = hipGraphInstantiate(&b->build_graph, g, nullptr, nullptr, 0);
    if (g) (void)hipGraphDestroy(g);
    if (rc != PSM_OK || e != hipSuccess || !b->build_graph) {
        (void)hipGetLastError();
        b->build_graph = nullptr;
        return rc != PSM_OK ? rc : PSM_ERR_HIP;
    }
    b->graph_error_word = c->sort_error_word;  // what launch_sort left for sort_check (one-sweep: its timeout word)
    return PSM_OK;
}

int psm_bvh_build(psm_bvh* b, const double* opt) {
    if (!b) return PSM_ERR_INVALID;
    psm_ctx* c = b->ctx;
    // The graph belongs to (triangle count, sort algorithm, generation of the context's sort buffers).
    const int algo = sort_effective_algorithm(c);   // (a hybrid sort that overflowed has fallen back to the eight passes: another graph)
    if (b->graph_tris != b->tri_count || b->graph_algo != algo || b->graph_sort_gen != c->sort_gen) {
        bvh_drop_graph(b);
        b->plain_builds = 0;
    }
    // Plain launches: the first build of a configuration (a scene that is built once gains nothing from a capture, and
    // it allocates the sort's buffers), per-stage timing (events between the stages), or graphs switched off.
    if (!b->use_graph || c->timing == 1 || b->tri_count == 0 || b->plain_builds == 0) {
        int rc = bvh_build_plain(b, opt);
        b->plain_builds = (rc == PSM_OK) ? b->plain_builds + 1 : 0;
        b->graph_tris = b->tri_count, b->graph_algo = algo, b->graph_sort_gen = c->sort_gen;
        return rc;
    }
    (void)hipSetDevice(c->device);
    { int rc = bvh_upload_opt(b, opt); if (rc != PSM_OK) return rc; }
    if (!b->build_graph && bvh_capture_graph(b) != PSM_OK) {  // no graph on this runtime / stream: plain launches from now on
        b->use_graph = false;
        return bvh_build_plain(b, opt);
    }
    if (hipGraphLaunch(b->build_graph, c->stream) != hipSuccess) {
        (void)hipGetLastError();
        bvh_drop_graph(b);
        b->use_graph = false;
        return bvh_build_plain(b, opt);
    }
    c->sort_error_word = b->graph_error_word;
    b->topo_tris = b->tri_count;
    b->bounds_done = b->morton_done = b->sort_done = b->built = true;
    b->records_valid = false;
    return PSM_OK;
}

int psm_bvh_set_build_graph(psm_bvh* b, int enable) {
    if (!b) return PSM_ERR_INVALID;
    b->use_graph = enable != 0;
    if (!b->use_graph) bvh_drop_graph(b);
    return PSM_OK;
}

int psm_bvh_get_info(psm_bvh* b, psm_bvh_info* info) {
    if (!b || !info) return PSM_ERR_INVALID;
    psm_ctx* c = b->ctx;
    (void)hipSetDevice(c->device);
    uint32_t sm[SM_WORDS];
    PSM_HIP(c, hipMemcpyAsync(sm, b->d_small, sizeof(sm), hipMemcpyDeviceToHost, c->stream));
    PSM_HIP(c, hipStreamSynchronize(c->stream));
    int se = sort_check(c);
    if (se != PSM_OK) return se;
    info->triangle_count = b->tri_count;
    info->leaf_count = sm[SM_COUNT];
    info->root = (int32_t)sm[SM_ROOT];
    std::memcpy(info->transform, sm, 16 * sizeof(float));
    std::memcpy(info->bounds_min, sm + SM_BFLOAT, 4 * sizeof(float));
    std::memcpy(info->bounds_max, sm + SM_BFLOAT + 4, 4 * sizeof(float));
    return PSM_OK;
}

int psm_bvh_download(psm_bvh* b, int what, void* dst, size_t bytes) {
    if (!b || !dst) return PSM_ERR_INVALID;
    psm_ctx* c = b->ctx;
    (void)hipSetDevice(c->device);
    const void* src = nullptr;
    size_t elem = 0;
    switch (what) {
        case PSM_BVH_KEYS: src = b->d_keys; elem = 8; break;
        case PSM_BVH_INDICES: src = b->d_idx; elem = 4; break;
        case PSM_BVH_LEAF_BOX: src = b->d_leafbox; elem = 16; break;
        case PSM_BVH_LEAF_TRI: src = b->d_leaftri; elem = 4; break;
        case PSM_BVH_PAIR_BOX: src = b->d_pairbox; elem = 32; break;
        case PSM_BVH_LINK: src = b->d_link; elem = 8; break;
        case PSM_BVH_RANGE: src = b->d_range; elem = 8; break;
        case PSM_BVH_SORTED_TRI: src = b->d_sorted_tri; elem = 4; break;
        case PSM_BVH_POSITIONS: src = b->d_pos; elem = 36; break;
        case PSM_BVH_NORMALS: src = b->d_nrm; elem = 36; break;
        case PSM_BVH_MATERIALS: src = b->d_mats; elem = 4; break;
        case PSM_BVH_TEXCOORDS: src = b->d_tex; elem = 24; break;
        case PSM_BVH_NODE32: src = b->d_node32; elem = 32; break;
        default: return set_err(c, PSM_ERR_INVALID, "psm_bvh_download: unknown item");
    }
    if (bytes > b->cap * elem) return set_err(c, PSM_ERR_CAPACITY, "psm_bvh_download: too many bytes");
    if (what == PSM_BVH_PAIR_BOX || what == PSM_BVH_LINK || what == PSM_BVH_RANGE) {
        // the reference-shaped node records are not written by a build (traversal reads its own 32-byte record)
        if (!b->built) return set_err(c, PSM_ERR_STATE, "psm_bvh_download: node records before a build");
        int rc = launch_bvh_emit_records(b);
        if (rc != PSM_OK) return rc;
    }
    PSM_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    PSM_HIP(c, hipStreamSynchronize(c->stream));
    return sort_check(c);
}

// ---- Pipeline -----------------------------------------------------------------------------------
static void rt_free_grid(psm_rt* r) {
    for (int q = 0; q < 2; q++) { dev_free(r->qA[q]); dev_free(r->qB[q]); dev_free(r->qC[q]); dev_free(r->q_bases[q]); }
    dev_free(r->d_block);
    dev_free(r->hit0); dev_free(r->hitN); dev_free(r->pool);
    dev_free(r->t_coord); dev_free(r->t_sum); dev_free(r->t_flag);
}

int psm_rt_destroy(psm_rt* r) {
    if (!r) return PSM_ERR_INVALID;
    (void)hipSetDevice(r->ctx->device);
    (void)hipStreamSynchronize(r->ctx->stream);
    rt_free_grid(r);
    dev_free(r->presampled); dev_free(r->filtered); dev_free(r->d_lights); dev_free(r->d_mats); dev_free(r->d_cnt); dev_free(r->d_sky); dev_free(r->d_tex_table); dev_free(r->d_geoms);
    if (r->d_phase_mem) (void)hipFree(r->d_phase_mem);
    if (r->h_cnt) (void)hipHostFree(r->h_cnt);
    if (r->ev_cnt) (void)hipEventDestroy(r->ev_cnt);
    if (r->ev_fold) (void)hipEventDestroy(r->ev_fold);
    for (int i = 0; i < MAX_TEXTURES; i++) if (r->tex_host[i].texels) (void)hipFree(const_cast<uint32_t*>(r->tex_host[i].texels));
    delete r;
    return PSM_OK;
}

int psm_rt_create(psm_ctx* c, psm_rt** out) {
    if (!c || !out) return PSM_ERR_INVALID;
    (void)hipSetDevice(c->device);
    psm_rt* r = new (std::nothrow) psm_rt();
    if (!r) return PSM_ERR_INVALID;
    r->ctx = c;
    int rc = dev_alloc(c, &r->d_cnt, (size_t)8);
    if (rc == PSM_OK) rc = dev_alloc(c, &r->d_lights, (size_t)16);
    if (rc == PSM_OK) rc = dev_alloc(c, &r->d_tex_table, (size_t)MAX_TEXTURES);
    if (rc == PSM_OK) rc = dev_alloc(c, &r->d_geoms, (size_t)MAX_TRAV_OBJECTS);
    if (rc == PSM_OK) r->tex_dirty = true;
    if (rc != PSM_OK) { psm_rt_destroy(r); return rc; }
    (void)hipMemsetAsync(r->d_cnt, 0, 32, c->stream);
    // default sun, Pipeline.inl:93-98
    psm_light L[16];
    std::memset(L, 0, sizeof(L));
    for (int i = 0; i < 16; i++) {
        L[i].lightColor[0] = (255.f / 255.f) * 150.f;
        L[i].lightColor[1] = (250.f / 255.f) * 150.f;
        L[i].lightColor[2] = (244.f / 255.f) * 150.f;
        L[i].lightColor[3] = 40.0f;
        L[i].lightVector[0] = 0.3f; L[i].lightVector[1] = 1.0f; L[i].lightVector[2] = 0.1f; L[i].lightVector[3] = 400.0f;
    }
    if (hipMemcpyAsync(r->d_lights, L, sizeof(L), hipMemcpyHostToDevice, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess) { psm_rt_destroy(r); return PSM_ERR_HIP; }
    *out = r;
    return PSM_OK;
}

int psm_rt_resize_buffers(psm_rt* r, uint32_t w, uint32_t h) {
    if (!r || w == 0 || h == 0) return PSM_ERR_INVALID;
    psm_ctx* c = r->ctx;
    (void)hipSetDevice(c->device);
    PSM_HIP(c, hipStreamSynchronize(c->stream));
    rt_free_grid(r);
    r->w = w; r->h = h; r->y0 = 0; r->y1 = h; r->tile_mode = 0; r->tile_root = true;
    uint64_t wr = (uint64_t)w * h;
    uint64_t lim = std::min<uint64_t>(wr * 4, 4096ull * 4096ull);  // Pipeline.inl:187-189
    r->limit = (uint32_t)lim;
    size_t L = r->limit;
    size_t nb = (L + SHADE_BLOCK - 1) / SHADE_BLOCK;
    int rc = PSM_OK;
    auto A = [&](int x) { if (rc == PSM_OK) rc = x; };
    for (int q = 0; q < 2; q++) {  // one segment of QUEUE_SEG slots per shading workgroup (SHADE_BLOCK rays in, at most 4 x as many out)
        A(dev_alloc(c, &r->qA[q], nb * QUEUE_SEG)); A(dev_alloc(c, &r->qB[q], nb * QUEUE_SEG)); A(dev_alloc(c, &r->qC[q], nb * QUEUE_SEG));
        A(dev_alloc(c, &r->q_bases[q], nb + 2));
        r->q_nb[q] = 1;
    }
    A(dev_alloc(c, &r->d_block, nb + 2));
    A(dev_alloc(c, &r->hit0, L)); A(dev_alloc(c, &r->hitN, L));
    r->pool_cap = (uint32_t)std::max<size_t>(L / 2, 1024);  // hits buffer = L/2 in the reference, Pipeline.inl:193
    A(dev_alloc(c, &r->pool, (size_t)r->pool_cap));
    A(dev_alloc(c, &r->t_coord, (size_t)wr)); A(dev_alloc(c, &r->t_sum, (size_t)wr)); A(dev_alloc(c, &r->t_flag, (size_t)wr));
    if (rc != PSM_OK) return rc;
    PSM_HIP(c, hipMemsetAsync(r->t_sum, 0, wr * sizeof(float4), c->stream));
    PSM_HIP(c, hipMemsetAsync(r->t_flag, 0, wr * sizeof(int32_t), c->stream));
    PSM_HIP(c, hipMemsetAsync(r->t_coord, 0, wr * sizeof(float2), c->stream));
    PSM_HIP(c, hipMemsetAsync(r->d_cnt, 0, 32, c->stream));
    r->ray_count = 0; r->count_valid = true; r->cur = 0;
    if (r->dw == 0) return psm_rt_resize(r, w, h);
    return PSM_OK;
}

int psm_rt_resize(psm_rt* r, uint32_t dw, uint32_t dh) {
    if (!r || dw == 0 || dh == 0) return PSM_ERR_INVALID;
    psm_ctx* c = r->ctx;
    (void)hipSetDevice(c->device);
    PSM_HIP(c, hipStreamSynchronize(c->stream));
    dev_free(r->presampled); dev_free(r->filtered);
    r->dw = dw; r->dh = dh;
    int rc = dev_alloc(c, &r->presampled, (size_t)dw * dh);
    if (rc == PSM_OK) rc = dev_alloc(c, &r->filtered, (size_t)dw * dh);
    if (rc != PSM_OK) return rc;
    return psm_rt_clear_sampler(r);
}

int psm_rt_clear_sampler(psm_rt* r) {
    if (!r || !r->presampled) return PSM_ERR_INVALID;
    psm_ctx* c = r->ctx;
    (void)hipSetDevice(c->device);
    PSM_HIP(c, hipMemsetAsync(r->presampled, 0, (size_t)r->dw * r->dh * sizeof(float4), c->stream));
    PSM_HIP(c, hipMemsetAsync(r->filtered, 0, (size_t)r->dw * r->dh * sizeof(float4), c->stream));
    return PSM_OK;
}

int psm_rt_set_tile(psm_rt* r, uint32_t y0, uint32_t y1) {
    if (!r || y0 > y1 || y1 > r->h) return PSM_ERR_INVALID;
    r->y0 = y0; r->y1 = y1;
    r->tile_mode = 0;
    r->tile_root = true;
    return PSM_OK;
}
int psm_rt_set_tile_weighted(psm_rt* r, uint32_t rank, uint32_t world, const uint32_t* weights) {
    if (!r || world == 0 || rank >= world) return PSM_ERR_INVALID;
    BandMap m;
    if (!band_map_make(world, weights, &m))
        return set_err(r->ctx, PSM_ERR_INVALID, "psm_rt_set_tile_weighted: at most 64 ranks, and the weights must add up to 1..64");
    r->bands = m;
    r->tile_mode = 1; r->tile_rank = rank; r->tile_world = world;
    r->tile_root = rank == 0;  // tiles are gathered to rank 0 (psm_rt_unpack_texels_dev), which runs sample()
    return PSM_OK;
}
int psm_rt_set_tile_interleaved(psm_rt* r, uint32_t rank, uint32_t world) { return psm_rt_set_tile_weighted(r, rank, world, nullptr); }
int psm_rt_tile_texels(psm_rt* r, uint32_t* count) {
    if (!r || !count) return PSM_ERR_INVALID;
    *count = tile_texel_count(r);
    return PSM_OK;
}
int psm_rt_pack_texels_dev(psm_rt* r, float* d_dst) {
    if (!r || !d_dst || !r->t_sum) return PSM_ERR_INVALID;
    (void)hipSetDevice(r->ctx->device);
    return launch_rt_pack(r, r->ctx->stream, d_dst, 0, r->tile_mode ? &r->bands : nullptr, r->tile_mode ? r->tile_rank : r->y0, r->tile_mode ? r->tile_world : r->y1);
}
// the dealing a (rank, world) pair of the unpack calls refers to: rt's own when it is sharded over `world` ranks
// (psm_rt_set_tile_weighted on the gathering rank), the round-robin one otherwise
static bool dealing_for(const psm_rt* r, uint32_t world, BandMap* m) {
    if (r->tile_mode == 1 && r->tile_world == world) { *m = r->bands; return true; }
    return band_map_make(world, nullptr, m);
}
int psm_rt_unpack_texels_dev(psm_rt* r, int interleaved, uint32_t a, uint32_t b, const float* d_src) {
    if (!r || !d_src || !r->t_sum) return PSM_ERR_INVALID;
    if (interleaved ? (b == 0 || a >= b) : (a > b || b > r->h)) return PSM_ERR_INVALID;
    (void)hipSetDevice(r->ctx->device);
    BandMap m;
    if (interleaved && !dealing_for(r, b, &m)) return PSM_ERR_INVALID;
    return launch_rt_pack(r, r->ctx->stream, const_cast<float*>(d_src), 1, interleaved ? &m : nullptr, a, b);
}
int psm_rt_unpack_tiles_dev(psm_rt* r, uint32_t world, uint32_t skip_rank, const float* d_all, size_t stride_floats) {
    if (!r || !d_all || !r->t_sum || world == 0 || (stride_floats & 3u)) return PSM_ERR_INVALID;
    BandMap m;
    if (!dealing_for(r, world, &m)) return PSM_ERR_INVALID;
    if (stride_floats < (size_t)max_owned_texels(m, r->w, r->h) * 4) return set_err(r->ctx, PSM_ERR_CAPACITY, "psm_rt_unpack_tiles_dev: stride smaller than the largest tile");
    (void)hipSetDevice(r->ctx->device);
    return launch_rt_unpack_all(r, r->ctx->stream, d_all, m, skip_rank, stride_floats);
}
int psm_rt_ray_count_dev(psm_rt* r, int32_t* d_dst) {
    if (!r || !d_dst) return PSM_ERR_INVALID;
    psm_ctx* c = r->ctx;
    (void)hipSetDevice(c->device);
    PSM_HIP(c, hipMemcpyAsync(d_dst, r->d_cnt, 4, hipMemcpyDeviceToDevice, c->stream));
    return PSM_OK;
}
int psm_rt_set_ray_count(psm_rt* r, int32_t count) {
    if (!r || count < 0 || (uint32_t)count > r->limit) return PSM_ERR_INVALID;
    r->ray_count = (uint32_t)count;
    r->count_valid = true;
    return PSM_OK;
}

int psm_rt_set_lights(psm_rt* r, const psm_light* lights, uint32_t count) {
    if (!r || !lights || count == 0 || count > 16) return PSM_ERR_INVALID;
    psm_ctx* c = r->ctx;
    (void)hipSetDevice(c->device);
    PSM_HIP(c, hipMemcpyAsync(r->d_lights, lights, count * sizeof(psm_light), hipMemcpyHostToDevice, c->stream));
    PSM_HIP(c, hipStreamSynchronize(c->stream));
    r->light_count = count;
    return PSM_OK;
}

int psm_rt_set_sky(psm_rt* r, const float rgba[4]) {
    if (!r || !rgba) return PSM_ERR_INVALID;
    for (int k = 0; k < 4; k++) r->sky[k] = rgba[k];
    return PSM_OK;
}

int psm_rt_set_texture(psm_rt* r, uint32_t slot, const uint8_t* rgba8, uint32_t width, uint32_t height) {
    if (!r || slot == 0 || slot >= (uint32_t)MAX_TEXTURES) return PSM_ERR_INVALID;
    psm_ctx* c = r->ctx;
    (void)hipSetDevice(c->device);
    PSM_HIP(c, hipStreamSynchronize(c->stream));
    if (r->tex_host[slot].texels) (void)hipFree(const_cast<uint32_t*>(r->tex_host[slot].texels));
    r->tex_host[slot] = TexDesc{nullptr, 0, 0};
    r->tex_dirty = true;
    if (!rgba8 || width == 0 || height == 0) return PSM_OK;
    uint32_t* d = nullptr;
    int rc = dev_alloc(c, &d, (size_t)width * height);
    if (rc != PSM_OK) return rc;
    PSM_HIP(c, hipMemcpyAsync(d, rgba8, (size_t)width * height * 4, hipMemcpyHostToDevice, c->stream));
    PSM_HIP(c, hipStreamSynchronize(c->stream));
    r->tex_host[slot] = TexDesc{d, (int)width, (int)height};
    return PSM_OK;
}

int psm_rt_set_skybox(psm_rt* r, const uint8_t* rgba8, uint32_t width, uint32_t height) {
    if (!r) return PSM_ERR_INVALID;
    psm_ctx* c = r->ctx;
    (void)hipSetDevice(c->device);
    PSM_HIP(c, hipStreamSynchronize(c->stream));
    dev_free(r->d_sky);
    r->sky_w = r->sky_h = 0;
    if (!rgba8 || width == 0 || height == 0) return PSM_OK;
    int rc = dev_alloc(c, &r->d_sky, (size_t)width * height);
    if (rc != PSM_OK) return rc;
    PSM_HIP(c, hipMemcpyAsync(r->d_sky, rgba8, (size_t)width * height * 4, hipMemcpyHostToDevice, c->stream));
    PSM_HIP(c, hipStreamSynchronize(c->stream));
    r->sky_w = width; r->sky_h = height;
    return PSM_OK;
}

int psm_rt_set_materials(psm_rt* r, const psm_material* mats, uint32_t count, int32_t load_offset) {
    if (!r || (count && !mats)) return PSM_ERR_INVALID;
    psm_ctx* c = r->ctx;
    (void)hipSetDevice(c->device);
    PSM_HIP(c, hipStreamSynchronize(c->stream));
    dev_free(r->d_mats);
    int rc = dev_alloc(c, &r->d_mats, (size_t)std::max<uint32_t>(count, 1));
    if (rc != PSM_OK) return rc;
    if (count) {
        PSM_HIP(c, hipMemcpyAsync(r->d_mats, mats, count * sizeof(psm_material), hipMemcpyHostToDevice, c->stream));
        PSM_HIP(c, hipStreamSynchronize(c->stream));
    }
    r->mat_count = count;
    r->mat_offset = load_offset;
    // rt_shade builds only the lobe that survives the lobe pick as long as the other one's colour is certain to be 0 (shade.hip);
    // it is NaN instead -- and the reference queues the ray -- where the specular colour is 0 / 0 (a black full-metal material:
    // albedo 0, metallic 1, both after their fp16 round trip) or a colour overflows or is not a number to begin with. Without
    // textures a surface is its material's constants (surface.comp:81-161), so the materials decide: `ordinary` = diffuse rgb in
    // [0, 1] with one component of at least 1/1024, roughness and metallic in [0, 1]. The constants' alpha plays no part: without a
    // diffuse texture a hit's albedo is vec4(diffuse.xyz, 1) (fetchDiffuse, surface.comp:155-161), so an equal-distance chain of such
    // hits never composites (rayshading.comp:60-116 stops at alpha > 0.99999) -- ADVICE r04 feared a transparent constant here;
    // `+clearmetal` in tests/test_gpu_parity.py holds the case. (Residue: a shadow ray's weight is NaN where the hit point IS the
    // light's centre, a set of measure zero.)
    r->mats_ordinary = true;
    for (uint32_t i = 0; i < count; i++) {
        const psm_material& m = mats[i];
        bool ok = true;
        float top = 0.f;
        for (int k = 0; k < 3; k++) { ok = ok && m.diffuse[k] >= 0.f && m.diffuse[k] <= 1.f; top = m.diffuse[k] > top ? m.diffuse[k] : top; }
        ok = ok && top >= 1.0f / 1024.0f && m.specular[1] >= 0.f && m.specular[1] <= 1.f && m.specular[2] >= 0.f && m.specular[2] <= 1.f;
        r->mats_ordinary = r->mats_ordinary && ok;
    }
    return PSM_OK;
}

int psm_rt_camera(psm_rt* r, const float cam_inv[16], const float proj_inv[16], uint32_t time) {
    if (!r || !cam_inv || !proj_inv) return PSM_ERR_INVALID;
    if (!r->qA[0]) return set_err(r->ctx, PSM_ERR_STATE, "camera before resizeBuffers");
    (void)hipSetDevice(r->ctx->device);
    return launch_rt_camera(r, cam_inv, proj_inv, time);
}

int psm_rt_ray_count(psm_rt* r, int32_t* count) {
    if (!r || !count) return PSM_ERR_INVALID;
    psm_ctx* c = r->ctx;
    (void)hipSetDevice(c->device);
    if (!r->count_valid) {
        uint32_t v = 0;
        PSM_HIP(c, hipMemcpyAsync(&v, r->d_cnt, 4, hipMemcpyDeviceToHost, c->stream));
        PSM_HIP(c, hipStreamSynchronize(c->stream));
        r->ray_count = v;
        r->count_valid = true;
    }
    *count = (int32_t)r->ray_count;
    return PSM_OK;
}

int psm_rt_traverse(psm_rt* r, psm_bvh* b) {
    if (!r || !b) return PSM_ERR_INVALID;
    (void)hipSetDevice(r->ctx->device);
    if (!b->built) return set_err(r->ctx, PSM_ERR_STATE, "traverse before build");
    int32_t n;
    int rc = psm_rt_ray_count(r, &n);
    if (rc != PSM_OK) return rc;
    return launch_rt_traverse(r, b);
}


int psm_rt_set_camera_mode(psm_rt* r, int enable360) {
    if (!r) return PSM_ERR_INVALID;
    r->enable360 = enable360 ? 1 : 0;
    return PSM_OK;
}

int psm_rt_set_traverse_mode(psm_rt* r, int mode) {
    if (!r || mode < PSM_TRAVERSE_AUTO || mode > PSM_TRAVERSE_ADAPTIVE) return PSM_ERR_INVALID;
    r->trav_mode = mode;
    return PSM_OK;
}

int psm_rt_set_traverse_phases(psm_rt* r, const uint32_t* caps, uint32_t count, uint32_t min_rays) {
    if (!r || count > 7 || (count && !caps)) return PSM_ERR_INVALID;
    for (uint32_t k = 0; k < count; k++)
        if (caps[k] == 0) return PSM_ERR_INVALID;
    for (uint32_t k = 0; k < count; k++) r->phase_caps[k] = caps[k];
    r->phase_caps_n = (int)count;
    r->phase_min_rays = min_rays;
    r->trav_mode = count ? PSM_TRAVERSE_PHASED : PSM_TRAVERSE_WHOLE;
    return PSM_OK;
}

int psm_rt_set_traverse_adaptive(psm_rt* r, uint32_t min_live, uint32_t min_steps, uint32_t final_rays,
                                 uint32_t max_launches, uint32_t min_rays) {
    if (!r || min_live < 2 || min_live > 64 || max_launches < 2 || max_launches > 15) return PSM_ERR_INVALID;
    r->adapt_min_live = min_live;
    r->adapt_min_steps = min_steps;
    r->adapt_final_rays = final_rays;
    r->adapt_max_launches = max_launches;
    r->phase_min_rays = min_rays;
    return PSM_OK;
}

int psm_rt_set_traverse_solo(psm_rt* r, uint32_t solo_max) {
    if (!r || solo_max > 4u) return PSM_ERR_INVALID;
    r->solo_max = solo_max;
    return PSM_OK;
}


int psm_rt_reset_hits(psm_rt* r) {
    if (!r) return PSM_ERR_INVALID;
    r->trav_n = 0;
    return PSM_OK;
}

int psm_rt_shade(psm_rt* r, psm_bvh* b, uint32_t time) {
    if (!r || !b) return PSM_ERR_INVALID;
    (void)hipSetDevice(r->ctx->device);
    if (r->tex_dirty) {
        PSM_HIP(r->ctx, hipMemcpyAsync(r->d_tex_table, r->tex_host, sizeof(r->tex_host), hipMemcpyHostToDevice, r->ctx->stream));
        PSM_HIP(r->ctx, hipStreamSynchronize(r->ctx->stream));
        r->tex_dirty = false;
    }
    if (!r->d_mats) return set_err(r->ctx, PSM_ERR_STATE, "shade before set_materials");
    int32_t n;
    int rc = psm_rt_ray_count(r, &n);
    if (rc != PSM_OK) return rc;
    return launch_rt_shade(r, b, time);
}

int psm_rt_sample(psm_rt* r) {
    if (!r || !r->presampled || !r->t_sum) return PSM_ERR_INVALID;
    (void)hipSetDevice(r->ctx->device);
    return launch_rt_sample(r, r);
}

int psm_rt_snap(psm_rt* r, float* rgba, int raw) {
    if (!r || !rgba || !r->presampled) return PSM_ERR_INVALID;
    psm_ctx* c = r->ctx;
    (void)hipSetDevice(c->device);
    PSM_HIP(c, hipMemcpyAsync(rgba, raw ? r->presampled : r->filtered, (size_t)r->dw * r->dh * sizeof(float4),
                              hipMemcpyDeviceToHost, c->stream));
    PSM_HIP(c, hipStreamSynchronize(c->stream));
    return PSM_OK;
}

int psm_rt_get_texels_dev(psm_rt* r, uint32_t y0, uint32_t y1, float* d_dst) {
    if (!r || !d_dst || y0 > y1 || y1 > r->h) return PSM_ERR_INVALID;
    psm_ctx* c = r->ctx;
    (void)hipSetDevice(c->device);
    PSM_HIP(c, hipMemcpyAsync(d_dst, r->t_sum + (size_t)y0 * r->w, (size_t)(y1 - y0) * r->w * sizeof(float4),
                              hipMemcpyDeviceToDevice, c->stream));
    return PSM_OK;
}
int psm_rt_set_texels_dev(psm_rt* r, uint32_t y0, uint32_t y1, const float* d_src) {
    if (!r || !d_src || y0 > y1 || y1 > r->h) return PSM_ERR_INVALID;
    psm_ctx* c = r->ctx;
    (void)hipSetDevice(c->device);
    PSM_HIP(c, hipMemcpyAsync(r->t_sum + (size_t)y0 * r->w, d_src, (size_t)(y1 - y0) * r->w * sizeof(float4),
                              hipMemcpyDeviceToDevice, c->stream));
    return PSM_OK;
}

int psm_rt_download_rays(psm_rt* r, psm_ray* dst, uint32_t max_rays, uint32_t* count) {
    if (!r || !dst) return PSM_ERR_INVALID;
    psm_ctx* c = r->ctx;
    int32_t n;
    int rc = psm_rt_ray_count(r, &n);
    if (rc != PSM_OK) return rc;
    uint32_t m = std::min<uint32_t>((uint32_t)n, max_rays);
    std::vector<float4> A(m), B(m), C(m);
    if (m) {
        // the queue is segmented (one segment per shading workgroup): gather it into queue order on the device first
        float4* d_dense = nullptr;
        PSM_HIP(c, hipMalloc((void**)&d_dense, (size_t)3 * m * sizeof(float4)));
        rc = launch_rt_gather_queue(r, d_dense, m);
        if (rc == PSM_OK &&
            (hipMemcpyAsync(A.data(), d_dense, m * sizeof(float4), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
             hipMemcpyAsync(B.data(), d_dense + m, m * sizeof(float4), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
             hipMemcpyAsync(C.data(), d_dense + 2 * (size_t)m, m * sizeof(float4), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
             hipStreamSynchronize(c->stream) != hipSuccess))
            rc = set_err(c, PSM_ERR_HIP, "psm_rt_download_rays: copy");
        (void)hipFree(d_dense);
        if (rc != PSM_OK) return rc;
    }
    for (uint32_t i = 0; i < m; i++) {
        dst[i].origin[0] = A[i].x; dst[i].origin[1] = A[i].y; dst[i].origin[2] = A[i].z;
        dst[i].direct[0] = B[i].x; dst[i].direct[1] = B[i].y; dst[i].direct[2] = B[i].z;
        dst[i].color[0] = C[i].x; dst[i].color[1] = C[i].y; dst[i].color[2] = C[i].z;
        std::memcpy(&dst[i].texel, &A[i].w, 4);
        std::memcpy(&dst[i].bitfield, &B[i].w, 4);
        std::memcpy(&dst[i].pkey, &C[i].w, 4);
    }
    if (count) *count = (uint32_t)n;
    return PSM_OK;
}

int psm_rt_upload_rays(psm_rt* r, const psm_ray* src, uint32_t count) {
    if (!r || (count && !src)) return PSM_ERR_INVALID;
    psm_ctx* c = r->ctx;
    (void)hipSetDevice(c->device);
    if (!r->qA[0]) return set_err(c, PSM_ERR_STATE, "upload_rays before resizeBuffers");
    if (count > r->limit) return set_err(c, PSM_ERR_CAPACITY, "upload_rays: exceeds currentRayLimit");
    // a ray deposits its radiance into the texel it names (rt_shade: _collect): one outside the grid would write past the texel arrays
    for (uint32_t i = 0; i < count; i++)
        if (src[i].texel < 0 || (uint64_t)src[i].texel >= (uint64_t)r->w * r->h)
            return set_err(c, PSM_ERR_INVALID, "upload_rays: a ray's texel lies outside the ray grid");
    std::vector<float4> A(count), B(count), C(count);
    for (uint32_t i = 0; i < count; i++) {
        A[i].x = src[i].origin[0]; A[i].y = src[i].origin[1]; A[i].z = src[i].origin[2];
        B[i].x = src[i].direct[0]; B[i].y = src[i].direct[1]; B[i].z = src[i].direct[2];
        C[i].x = src[i].color[0]; C[i].y = src[i].color[1]; C[i].z = src[i].color[2];
        std::memcpy(&A[i].w, &src[i].texel, 4);
        std::memcpy(&B[i].w, &src[i].bitfield, 4);
        std::memcpy(&C[i].w, &src[i].pkey, 4);
    }
    if (count) {
        PSM_HIP(c, hipMemcpyAsync(r->qA[r->cur], A.data(), count * sizeof(float4), hipMemcpyHostToDevice, c->stream));
        PSM_HIP(c, hipMemcpyAsync(r->qB[r->cur], B.data(), count * sizeof(float4), hipMemcpyHostToDevice, c->stream));
        PSM_HIP(c, hipMemcpyAsync(r->qC[r->cur], C.data(), count * sizeof(float4), hipMemcpyHostToDevice, c->stream));
    }
    uint32_t cnt[3] = {count, 0, 0};
    PSM_HIP(c, hipMemcpyAsync(r->d_cnt, cnt, sizeof(cnt), hipMemcpyHostToDevice, c->stream));
    uint32_t bases[2] = {0, count};  // written densely: one segment
    PSM_HIP(c, hipMemcpyAsync(r->q_bases[r->cur], bases, sizeof(bases), hipMemcpyHostToDevice, c->stream));
    r->q_nb[r->cur] = 1;
    PSM_HIP(c, hipStreamSynchronize(c->stream));
    r->ray_count = count;
    r->count_valid = true;
    r->trav_n = 0;
    return PSM_OK;
}

int psm_rt_download_hits(psm_rt* r, psm_hit* hits, int32_t* counts, uint32_t max_rays) {
    if (!r || !hits || !counts) return PSM_ERR_INVALID;
    psm_ctx* c = r->ctx;
    (void)hipSetDevice(c->device);
    uint32_t m = std::min<uint32_t>(r->limit, max_rays);
    std::vector<float4> h0(m), pool(r->pool_cap);
    std::vector<uint32_t> hn(m);
    if (m) {
        PSM_HIP(c, hipMemcpyAsync(h0.data(), r->hit0, m * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
        PSM_HIP(c, hipMemcpyAsync(hn.data(), r->hitN, m * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        PSM_HIP(c, hipMemcpyAsync(pool.data(), r->pool, (size_t)r->pool_cap * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
        PSM_HIP(c, hipStreamSynchronize(c->stream));
    }
    for (uint32_t i = 0; i < m; i++) {
        uint32_t n = hn[i] & 15u, off = hn[i] >> 4;
        counts[i] = (int32_t)n;
        for (uint32_t k = 0; k < 8; k++) {
            psm_hit& o = hits[(size_t)i * 8 + k];
            float4 v = make_float4(0.f, 0.f, INF, 0.f);
            int tri = -1;
            if (k < n) {
                v = (k == 0) ? h0[i] : pool[off + k - 1];
                std::memcpy(&tri, &v.w, 4);
            }
            o.u = v.x; o.v = v.y; o.t = v.z; o.tri = tri;
        }
    }
    return PSM_OK;
}

int psm_rt_download_texels(psm_rt* r, float* sum_rgba, float* coord_xy, int32_t* flags) {
    if (!r || !r->t_sum) return PSM_ERR_INVALID;
    psm_ctx* c = r->ctx;
    (void)hipSetDevice(c->device);
    size_t wr = (size_t)r->w * r->h;
    if (sum_rgba) PSM_HIP(c, hipMemcpyAsync(sum_rgba, r->t_sum, wr * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    if (coord_xy) PSM_HIP(c, hipMemcpyAsync(coord_xy, r->t_coord, wr * sizeof(float2), hipMemcpyDeviceToHost, c->stream));
    if (flags) PSM_HIP(c, hipMemcpyAsync(flags, r->t_flag, wr * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    PSM_HIP(c, hipStreamSynchronize(c->stream));
    return PSM_OK;
}

// ---- statistics ---------------------------------------------------------------------------------
int psm_stats_enable(psm_ctx* c, int timing, int counting) {
    if (!c) return PSM_ERR_INVALID;
    c->timing = timing < 0 ? 0 : (timing > 2 ? 1 : timing);
    c->counting = counting != 0;
    return PSM_OK;
}
int psm_stats_reset(psm_ctx* c) {
    if (!c) return PSM_ERR_INVALID;
    (void)hipSetDevice(c->device);
    PSM_HIP(c, hipStreamSynchronize(c->stream));
    collect_timing(c);
    for (int i = 0; i < CAT_COUNT; i++) { c->cat_ms[i] = 0.f; c->cat_launches[i] = 0; }
    c->intervals.clear();
    c->rays_traced = 0;
    c->rounds = 0;
    PSM_HIP(c, hipMemsetAsync(c->d_counters, 0, sizeof(DevCounters), c->stream));
    return PSM_OK;
}
// Time origin for psm_stats_traverse_intervals: an event recorded now on `origin`'s stream (origin == c, or another
// context of the same device whose origin `c` shares, so that the launches of several lanes land on one time axis).
int psm_stats_reference(psm_ctx* c, psm_ctx* origin) {
    if (!c || !origin || c->device != origin->device) return PSM_ERR_INVALID;
    (void)hipSetDevice(c->device);
    if (origin == c) {
        if (!c->ref_event || c->ref_owner != c) {
            hipEvent_t e = nullptr;
            PSM_HIP(c, hipEventCreate(&e));
            c->ref_event = e; c->ref_owner = c;
        }
        PSM_HIP(c, hipEventRecord(c->ref_event, c->stream));
        PSM_HIP(c, hipEventSynchronize(c->ref_event));
    } else {
        if (!origin->ref_event || origin->ref_owner != origin) return set_err(c, PSM_ERR_STATE, "psm_stats_reference: the origin context has no reference of its own yet");
        c->ref_event = origin->ref_event; c->ref_owner = origin;
    }
    return PSM_OK;
}
// start / end (ms after the reference) of every traversal launch timed since the last reset; synchronises
int psm_stats_traverse_intervals(psm_ctx* c, float* start_end_ms, uint32_t cap_launches, uint32_t* count) {
    if (!c || !count) return PSM_ERR_INVALID;
    (void)hipSetDevice(c->device);
    PSM_HIP(c, hipStreamSynchronize(c->stream));
    collect_timing(c);
    const uint32_t n = (uint32_t)(c->intervals.size() / 2);
    *count = n;
    if (start_end_ms) for (uint32_t i = 0; i < n && i < cap_launches; i++) { start_end_ms[2 * i] = c->intervals[2 * i]; start_end_ms[2 * i + 1] = c->intervals[2 * i + 1]; }
    return PSM_OK;
}
int psm_stats_get(psm_ctx* c, psm_stats* out) {
    if (!c || !out) return PSM_ERR_INVALID;
    (void)hipSetDevice(c->device);
    DevCounters d;
    PSM_HIP(c, hipMemcpyAsync(&d, c->d_counters, sizeof(d), hipMemcpyDeviceToHost, c->stream));
    PSM_HIP(c, hipStreamSynchronize(c->stream));
    collect_timing(c);
    std::memset(out, 0, sizeof(*out));
    out->rays_traced = c->rays_traced;
    out->node_visits = d.node_visits;
    out->tri_tests = d.tri_tests;
    out->stack_drops = d.stack_drops;
    out->iter_caps = d.iter_caps;
    out->baked_drops = d.baked_drops;
    out->chain_pool_drops = d.chain_pool_drops;
    out->ray_limit_drops = d.ray_limit_drops;
    out->traverse_launches = c->cat_launches[CAT_TRAVERSE];
    out->traverse_ms = c->cat_ms[CAT_TRAVERSE];
    out->build_ms = c->cat_ms[CAT_BUILD];
    out->sort_ms = c->cat_ms[CAT_SORT];
    out->shade_ms = c->cat_ms[CAT_SHADE];
    out->camera_ms = c->cat_ms[CAT_CAMERA];
    out->sample_ms = c->cat_ms[CAT_SAMPLE];
    out->rounds = c->rounds;
    out->bounds_ms = c->cat_ms[CAT_BOUNDS];
    out->morton_ms = c->cat_ms[CAT_MORTON];
    out->emit_ms = c->cat_ms[CAT_EMIT];
    out->wave_clock_ticks = d.wave_clock_ticks;
    out->wave_real_ticks = d.wave_real_ticks;
    out->wave_steps = d.wave_steps;
    out->waves = d.waves;
    out->handover_launches = c->cat_launches[CAT_TRAVERSE_HANDOVER];
    out->handover_ms = c->cat_ms[CAT_TRAVERSE_HANDOVER];
    return PSM_OK;
}

}  // extern "C"
