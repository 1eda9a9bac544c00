// transport_shm.hip -- a host-staged transport for psm_dist (include/psm_hip.h "transport seam").
//
// RCCL refuses two ranks on one device, and the test boxes have one GPU. This transport lets several PROCESSES that
// share a GPU be real peers of one another: each exchange is staged through a POSIX shared-memory segment (device ->
// host copy into the rank's slot, a sequence counter published with release / acquire ordering, host -> device copy on
// the receiving side). It is synchronous on the host, bounded (every wait for a peer has a deadline and an abort flag,
// PSM_ERR_PEER afterwards) and slow -- it exists so that psm_dist_render_frames / psm_dist_render_batch run against
// peers that park in other rounds, own other band counts and fail (tests/test_gpu_dist.py); it is never picked
// silently (PSM_DIST_TRANSPORT=hoststaged asks for it), and a bench.py line that ran over it says "rehearsal": true at the top
// level and carries value = null: nothing a driver could take for a scaling measurement.
//
// Segment: Header | Ctl[world] | i32 slots [world][I32_SLOT] | tile slots [world][slot_bytes].
//   gather k:    rank r waits until root has consumed its slot of gather k-1 (g_done[r] >= k-1), copies its tile in,
//                publishes g_ready[r] = k; root waits for g_ready[r] >= k for every r, copies the slots to its device
//                buffer, publishes g_done[r] = k. Senders do not wait for the root (like a stream-ordered ncclGather).
//   allgather k: everyone waits until all ranks have read exchange k-1 (a_done >= k-1), writes its slot, publishes
//                a_ready = k, waits for all a_ready >= k, reads all slots, publishes a_done = k.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstring>
#include <new>
#include <thread>

#include "psm_internal.h"

namespace psm {

constexpr size_t I32_SLOT = 4096;  // ints per rank and exchange (2 x 64 lanes needed)
constexpr uint32_t SHM_MAGIC = 0x50534d44u;

struct ShmHeader {
    std::atomic<uint32_t> magic;
    uint32_t world;
    uint64_t slot_bytes;
    std::atomic<uint32_t> attached;
    std::atomic<uint32_t> aborted;  // a rank that timed out or went away: the others stop waiting
};
struct alignas(64) ShmCtl {
    std::atomic<uint64_t> g_ready, g_done, a_ready, a_done;
};

struct ShmTransport {
    int rank = 0, world = 1;
    size_t slot_bytes = 0, total = 0;
    uint32_t timeout_ms = 0;
    char* base = nullptr;
    ShmHeader* hdr = nullptr;
    ShmCtl* ctl = nullptr;
    int32_t* islots = nullptr;
    char* tslots = nullptr;
    uint64_t gseq = 0, aseq = 0;
    bool registered = false;
    std::string err;
    using clk = std::chrono::steady_clock;

    // wait until pred() holds; false after the deadline or when a peer has given up
    template <class P> bool wait(P pred, const char* what) {
        const clk::time_point dead = clk::now() + std::chrono::milliseconds(timeout_ms);
        for (uint32_t spins = 0;; spins++) {
            if (pred()) return true;
            if (hdr->aborted.load(std::memory_order_acquire)) { err = std::string(what) + ": a peer gave up"; return false; }
            if (clk::now() > dead) {
                hdr->aborted.store(1u, std::memory_order_release);
                err = std::string(what) + ": no answer from a peer within " + std::to_string(timeout_ms) + " ms";
                return false;
            }
            if (spins < 64) std::this_thread::yield();
            else std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
    }
};

static int shm_gather_f32(void* u, const float* d_send, float* d_recv, size_t count, int root, void* stream) {
    ShmTransport* t = (ShmTransport*)u;
    hipStream_t st = (hipStream_t)stream;
    const size_t bytes = count * sizeof(float);
    if (bytes > t->slot_bytes) { t->err = "gather: the tile is larger than the slot_bytes given to psm_dist_connect_hoststaged"; return PSM_ERR_CAPACITY; }
    const uint64_t k = ++t->gseq;
    ShmCtl& me = t->ctl[t->rank];
    if (!t->wait([&] { return me.g_done.load(std::memory_order_acquire) >= k - 1; }, "gather (slot free)")) return PSM_ERR_PEER;
    if (hipMemcpyAsync(t->tslots + (size_t)t->rank * t->slot_bytes, d_send, bytes, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) { t->err = "gather: device -> host copy failed"; t->hdr->aborted.store(1u); return PSM_ERR_HIP; }
    me.g_ready.store(k, std::memory_order_release);
    if (t->rank != root) return PSM_OK;
    for (int r = 0; r < t->world; r++) {
        ShmCtl& c = t->ctl[r];
        if (!t->wait([&] { return c.g_ready.load(std::memory_order_acquire) >= k; }, "gather (peer's tile)")) return PSM_ERR_PEER;
        if (hipMemcpyAsync((char*)d_recv + (size_t)r * bytes, t->tslots + (size_t)r * t->slot_bytes, bytes, hipMemcpyHostToDevice, st) != hipSuccess) {
            t->err = "gather: host -> device copy failed"; t->hdr->aborted.store(1u); return PSM_ERR_HIP;
        }
    }
    if (hipStreamSynchronize(st) != hipSuccess) { t->err = "gather: host -> device copy failed"; t->hdr->aborted.store(1u); return PSM_ERR_HIP; }
    for (int r = 0; r < t->world; r++) t->ctl[r].g_done.store(k, std::memory_order_release);
    return PSM_OK;
}

static int shm_allgather_i32(void* u, const int32_t* d_send, int32_t* d_recv, size_t n, void* stream) {
    ShmTransport* t = (ShmTransport*)u;
    hipStream_t st = (hipStream_t)stream;
    if (n > I32_SLOT) { t->err = "allgather: more ints than a slot holds"; return PSM_ERR_CAPACITY; }
    const uint64_t k = ++t->aseq;
    for (int r = 0; r < t->world; r++) {
        ShmCtl& c = t->ctl[r];
        if (!t->wait([&] { return c.a_done.load(std::memory_order_acquire) >= k - 1; }, "allgather (previous exchange read)")) return PSM_ERR_PEER;
    }
    if (hipMemcpyAsync(t->islots + (size_t)t->rank * I32_SLOT, d_send, n * sizeof(int32_t), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) { t->err = "allgather: device -> host copy failed"; t->hdr->aborted.store(1u); return PSM_ERR_HIP; }
    t->ctl[t->rank].a_ready.store(k, std::memory_order_release);
    for (int r = 0; r < t->world; r++) {
        ShmCtl& c = t->ctl[r];
        if (!t->wait([&] { return c.a_ready.load(std::memory_order_acquire) >= k; }, "allgather (peer's values)")) return PSM_ERR_PEER;
        if (hipMemcpyAsync(d_recv + (size_t)r * n, t->islots + (size_t)r * I32_SLOT, n * sizeof(int32_t), hipMemcpyHostToDevice, st) != hipSuccess) {
            t->err = "allgather: host -> device copy failed"; t->hdr->aborted.store(1u); return PSM_ERR_HIP;
        }
    }
    if (hipStreamSynchronize(st) != hipSuccess) { t->err = "allgather: host -> device copy failed"; t->hdr->aborted.store(1u); return PSM_ERR_HIP; }
    t->ctl[t->rank].a_done.store(k, std::memory_order_release);
    return PSM_OK;
}

static void shm_destroy(void* u) {
    ShmTransport* t = (ShmTransport*)u;
    if (t->base) {
        if (t->registered) (void)hipHostUnregister(t->base);
        munmap(t->base, t->total);
    }
    delete t;
}
static const char* shm_last_error(void* u) { return ((ShmTransport*)u)->err.c_str(); }

}  // namespace psm

using namespace psm;

extern "C" int psm_dist_connect_hoststaged(psm_dist* d, const char* shm_name, size_t slot_bytes, uint32_t timeout_ms) {
    if (!d || !shm_name || shm_name[0] != '/' || slot_bytes == 0 || timeout_ms == 0) return PSM_ERR_INVALID;
    if (d->connected) return set_err(d->ctx, PSM_ERR_STATE, "psm_dist_connect_hoststaged: the communicator already has a transport");
    (void)hipSetDevice(d->ctx->device);
    ShmTransport* t = new (std::nothrow) ShmTransport();
    if (!t) return PSM_ERR_INVALID;
    t->rank = d->rank; t->world = d->world; t->timeout_ms = timeout_ms;
    t->slot_bytes = (slot_bytes + 4095) & ~(size_t)4095;
    const size_t off_ctl = 4096, off_i32 = off_ctl + ((sizeof(ShmCtl) * (size_t)d->world + 4095) & ~(size_t)4095);
    const size_t off_tile = off_i32 + ((I32_SLOT * sizeof(int32_t) * (size_t)d->world + 4095) & ~(size_t)4095);
    t->total = off_tile + t->slot_bytes * (size_t)d->world;
    auto fail = [&](const std::string& why, int code) { d->ctx->err = "psm_dist_connect_hoststaged: " + why; shm_destroy(t); return code; };
    using clk = std::chrono::steady_clock;
    const clk::time_point dead = clk::now() + std::chrono::milliseconds(timeout_ms);
    int fd = -1;
    if (d->rank == 0) {
        // (callers name the segment for ONE run -- a uuid in the tests, a nonce from the side channel in dist.py --, so a file of
        // that name can only be the leftover of a run that died between create and attach: out of the way with it)
        (void)shm_unlink(shm_name);
        fd = shm_open(shm_name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0) return fail(std::string("shm_open(create ") + shm_name + "): " + strerror(errno), PSM_ERR_STATE);
        if (ftruncate(fd, (off_t)t->total) != 0) { close(fd); shm_unlink(shm_name); return fail("ftruncate: " + std::string(strerror(errno)), PSM_ERR_CAPACITY); }
    } else {
        for (;;) {  // rank 0 creates the segment; wait for it to appear at its full size
            fd = shm_open(shm_name, O_RDWR, 0600);
            struct stat sb;
            if (fd >= 0 && fstat(fd, &sb) == 0 && (size_t)sb.st_size >= t->total) break;
            if (fd >= 0) { close(fd); fd = -1; }
            if (clk::now() > dead) return fail(std::string("rank 0's segment ") + shm_name + " did not appear", PSM_ERR_PEER);
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
    }
    void* m = mmap(nullptr, t->total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { if (d->rank == 0) shm_unlink(shm_name); return fail("mmap: " + std::string(strerror(errno)), PSM_ERR_CAPACITY); }
    t->base = (char*)m;
    t->hdr = (ShmHeader*)t->base;
    t->ctl = (ShmCtl*)(t->base + off_ctl);
    t->islots = (int32_t*)(t->base + off_i32);
    t->tslots = t->base + off_tile;
    if (d->rank == 0) {  // a fresh segment is zero-filled: counters start at 0
        t->hdr->world = (uint32_t)d->world;
        t->hdr->slot_bytes = t->slot_bytes;
        t->hdr->magic.store(SHM_MAGIC, std::memory_order_release);
    } else {
        while (t->hdr->magic.load(std::memory_order_acquire) != SHM_MAGIC) {
            if (clk::now() > dead) return fail("rank 0 did not initialise the segment", PSM_ERR_PEER);
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
        if (t->hdr->world != (uint32_t)d->world || t->hdr->slot_bytes != t->slot_bytes)
            return fail("the segment was created for another world size / slot size", PSM_ERR_INVALID);
    }
    t->hdr->attached.fetch_add(1u, std::memory_order_acq_rel);
    while (t->hdr->attached.load(std::memory_order_acquire) < (uint32_t)d->world) {  // collective, like ncclCommInitRank
        if (clk::now() > dead || t->hdr->aborted.load()) {
            t->hdr->aborted.store(1u);
            if (d->rank == 0) shm_unlink(shm_name);
            return fail("not every rank attached in time", PSM_ERR_PEER);
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    if (d->rank == 0) shm_unlink(shm_name);  // everybody holds a mapping: the name can go
    // pinned staging makes the copies asynchronous-capable and faster; pageable memory works too, so a refusal is not an error
    t->registered = hipHostRegister(t->base, t->total, hipHostRegisterDefault) == hipSuccess;
    if (!t->registered) (void)hipGetLastError();
    psm_dist_transport tr = {t, shm_gather_f32, shm_allgather_i32, shm_destroy, shm_last_error, "host-staged"};
    int rc = psm_dist_connect_transport(d, &tr);
    if (rc != PSM_OK) shm_destroy(t);
    else d->comm_ranks = (int)t->hdr->attached.load(std::memory_order_acquire);   // the processes that attached to the segment
    return rc;
}
