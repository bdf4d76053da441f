// comm.hip - the exchange step of the row-partitioned engine: an RCCL all-gather of carry rows on a
// communicator handle.  The reference has no counterpart (NGCF.py is single-device).
//
// libngcf_hip.so does not link RCCL: the entry point binds to the librccl the process already has (the one
// torch.distributed loaded, so the handle and the code that uses it come from the same library) and only falls back to
// loading one by name.  ncclAllGather is asynchronous on the given stream like every other entry point.
#include "common.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {
using allgather_fn = ncclResult_t (*)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
using errstr_fn = const char *(*)(ncclResult_t);
using count_fn = ncclResult_t (*)(const ncclComm_t, int *);

struct Rccl {
    void *handle = nullptr;
    allgather_fn all_gather = nullptr;
    errstr_fn err_string = nullptr;
    count_fn comm_count = nullptr;
};

Rccl *rccl()
{
    static Rccl r = [] {
        Rccl x;
        for (const char *name : {"librccl.so.1", "librccl.so"}) {          // already in the process?
            x.handle = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
            if (x.handle) break;
        }
        if (!x.handle)
            for (const char *name : {"librccl.so.1", "librccl.so"}) {
                x.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (x.handle) break;
            }
        if (x.handle) {
            x.all_gather = (allgather_fn)dlsym(x.handle, "ncclAllGather");
            x.err_string = (errstr_fn)dlsym(x.handle, "ncclGetErrorString");
            x.comm_count = (count_fn)dlsym(x.handle, "ncclCommCount");
        }
        return x;
    }();
    return &r;
}
}  // namespace

extern "C" int ngcf_allgather_rows(void *nccl_comm, const float *send, float *recv, int64_t rows_per_rank, int d, void *stream)
{
    if (!nccl_comm || !send || !recv) return fail(NGCF_ERR_ARG, "allgather_rows: null argument");
    if (rows_per_rank < 0 || d <= 0) return fail(NGCF_ERR_ARG, "allgather_rows: bad sizes");
    Rccl *r = rccl();
    if (!r->handle || !r->all_gather) return fail(NGCF_ERR_HIP, "allgather_rows: librccl is not loadable (%s)", dlerror());
    if (rows_per_rank == 0) return NGCF_OK;
    const ncclResult_t rc = r->all_gather(send, recv, (size_t)rows_per_rank * (size_t)d, ncclFloat32, (ncclComm_t)nccl_comm,
                                          (hipStream_t)stream);
    if (rc != ncclSuccess) return fail(NGCF_ERR_HIP, "ncclAllGather failed: %s", r->err_string ? r->err_string(rc) : "?");
    return NGCF_OK;
}

// ranks of a communicator handle (host call; lets the caller size `recv`)
extern "C" int ngcf_comm_size(void *nccl_comm, int *n_ranks)
{
    if (!nccl_comm || !n_ranks) return fail(NGCF_ERR_ARG, "comm_size: null argument");
    Rccl *r = rccl();
    if (!r->handle || !r->comm_count) return fail(NGCF_ERR_HIP, "comm_size: librccl is not loadable");
    const ncclResult_t rc = r->comm_count((ncclComm_t)nccl_comm, n_ranks);
    if (rc != ncclSuccess) return fail(NGCF_ERR_HIP, "ncclCommCount failed: %s", r->err_string ? r->err_string(rc) : "?");
    return NGCF_OK;
}

// =============================================================================================
// ngcf_p2p - a CU-free exchange between the ranks of one node (r03; SURVEY.md 8e: "or roll a P2P hipMemcpyPeerAsync fan-out").
//
// Why: an RCCL collective is a persistent kernel on the same CUs as the compute; beside it the L2-swept SpMM (one workgroup per
// CU, whole LDS and register file, XCD-wide lock step) runs 3.4-4x slower (profiles/r02_c4_rank_lab.txt), so with RCCL the
// overlapped products had to stay on the 2x slower row-wise kernels.  Here the bytes move on the copy engines and the waiting is
// done by the HOST threads, so no foreign wave ever occupies a CU:
//   * every rank owns one exchange buffer (hipMalloc) whose IPC handle the peers open once (hipIpcOpenMemHandle): a rank's
//     producers write the rows the others need into ITS OWN buffer;
//   * publish(slot, seq): on the producer's stream, a system-scope release event (the rows leave this device's L2 for a reader
//     on another device) and then a host function that stores seq into a POSIX shared-memory word - "my data of step seq is there";
//   * pull(peer, slot, seq, ...): the consumer's host thread waits for that word (the hosts run ahead of their GPUs, so this is
//     normally already true), then enqueues hipMemcpyAsync(peer's buffer -> own memory) on a copy stream of its own: a
//     device-to-device copy across xGMI runs on an SDMA engine, and because it is the CONSUMER's own copy the runtime makes it
//     visible to the consumer's kernels that wait for it (join: an event per copy stream, CU-free stream dependencies);
//   * ack(peer, slot, seq) / wait_acks(slot, seq): the same in the other direction, so a producer overwrites a region of its
//     exchange buffer only after every peer has finished reading the previous contents (with two regions in turn the wait is
//     normally over before it starts).
// All waits are bounded (timeout -> NGCF_ERR_HIP with a message): a dead peer is an error, never a hang.
// =============================================================================================
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <string>

namespace {
constexpr int kP2pSlots = 64;           // sequence words per rank
constexpr int kP2pPayloads = 8192;      // host-function payload ring

struct P2pStore {
    uint64_t *dst;
    uint64_t val;
};

void p2p_store_cb(void *arg)
{
    const P2pStore *s = static_cast<const P2pStore *>(arg);
    __atomic_store_n(s->dst, s->val, __ATOMIC_RELEASE);
}

double now_ms()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}
}  // namespace

struct ngcf_p2p {
    int rank = 0, world = 1, device = 0;
    int64_t bytes = 0;
    char *local = nullptr;
    std::vector<char *> peer;            // peer[q]: rank q's exchange buffer in this process (peer[rank] == local)
    std::string shm_name;
    uint64_t *shm = nullptr;             // pub[world][slots] then ack[reader][owner][slots]
    size_t shm_bytes = 0;
    std::vector<hipStream_t> copy;       // one copy stream per source rank
    std::vector<hipEvent_t> copy_ev;
    std::vector<char> copy_used;
    hipEvent_t rel_ev[kP2pSlots] = {};
    hipEvent_t fence_ev = nullptr;
    hipStream_t pub_stream = nullptr;    // carries the host functions of publish()
    P2pStore payload[kP2pPayloads];
    int next_payload = 0;
    double blocked_ms = 0;               // how long this rank's HOST thread has sat in p2p_wait_word so far (ngcf_p2p_stats)
    int64_t waits = 0, waits_blocked = 0;
    uint64_t *pub(int r, int slot) { return shm + (size_t)r * kP2pSlots + slot; }
    uint64_t *ack(int reader, int owner, int slot) { return shm + (size_t)world * kP2pSlots + ((size_t)reader * world + owner) * kP2pSlots + slot; }
    P2pStore *store(uint64_t *dst, uint64_t val)
    {
        P2pStore *s = &payload[next_payload];
        next_payload = (next_payload + 1) % kP2pPayloads;
        s->dst = dst;
        s->val = val;
        return s;
    }
};

extern "C" void ngcf_p2p_destroy(ngcf_p2p_t *p)
{
    if (!p) return;
    for (hipStream_t s : p->copy)
        if (s) (void)hipStreamSynchronize(s);
    for (int q = 0; q < (int)p->peer.size(); ++q)
        if (q != p->rank && p->peer[q]) (void)hipIpcCloseMemHandle(p->peer[q]);
    for (hipEvent_t e : p->copy_ev)
        if (e) (void)hipEventDestroy(e);
    for (hipStream_t s : p->copy)
        if (s) (void)hipStreamDestroy(s);
    for (hipEvent_t e : p->rel_ev)
        if (e) (void)hipEventDestroy(e);
    if (p->fence_ev) (void)hipEventDestroy(p->fence_ev);
    if (p->pub_stream) {
        (void)hipStreamSynchronize(p->pub_stream);
        (void)hipStreamDestroy(p->pub_stream);
    }
    if (p->local) (void)hipFree(p->local);
    if (p->shm) munmap(p->shm, p->shm_bytes);
    if (!p->shm_name.empty()) shm_unlink(p->shm_name.c_str());     // every rank tries; the name disappears with the last mapping
    delete p;
}

extern "C" int ngcf_p2p_create(int rank, int world, int64_t bytes, const char *shm_name, ngcf_p2p_t **out)
{
    if (!out) return fail(NGCF_ERR_ARG, "p2p_create: null out");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world || bytes <= 0 || !shm_name || !*shm_name || world > 64)
        return fail(NGCF_ERR_ARG, "p2p_create: bad argument (rank %d of %d, %lld bytes)", rank, world, (long long)bytes);
    ngcf_p2p *p = new ngcf_p2p();
    p->rank = rank;
    p->world = world;
    p->bytes = bytes;
    p->peer.assign((size_t)world, nullptr);
    p->copy.assign((size_t)world, nullptr);
    p->copy_ev.assign((size_t)world, nullptr);
    p->copy_used.assign((size_t)world, 0);
    auto body = [&]() -> int {
        HIP_TRY(hipGetDevice(&p->device));
        HIP_TRY(hipMalloc(&p->local, (size_t)bytes));
        p->peer[(size_t)rank] = p->local;
        for (int q = 0; q < world; ++q) {
            HIP_TRY(hipStreamCreateWithFlags(&p->copy[(size_t)q], hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&p->copy_ev[(size_t)q], hipEventDisableTiming));
        }
        for (int s = 0; s < kP2pSlots; ++s) HIP_TRY(hipEventCreateWithFlags(&p->rel_ev[s], hipEventDisableTiming | hipEventReleaseToSystem));
        HIP_TRY(hipEventCreateWithFlags(&p->fence_ev, hipEventDisableTiming));
        HIP_TRY(hipStreamCreateWithFlags(&p->pub_stream, hipStreamNonBlocking));
        p->shm_name = shm_name[0] == '/' ? shm_name : std::string("/") + shm_name;
        const int fd = shm_open(p->shm_name.c_str(), O_CREAT | O_RDWR, 0600);
        if (fd < 0) return fail(NGCF_ERR_HIP, "p2p_create: shm_open(%s) failed: %s", p->shm_name.c_str(), strerror(errno));
        p->shm_bytes = ((size_t)world * kP2pSlots + (size_t)world * world * kP2pSlots) * sizeof(uint64_t);
        if (ftruncate(fd, (off_t)p->shm_bytes) != 0) {
            close(fd);
            return fail(NGCF_ERR_HIP, "p2p_create: ftruncate failed: %s", strerror(errno));
        }
        void *m = mmap(nullptr, p->shm_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (m == MAP_FAILED) return fail(NGCF_ERR_HIP, "p2p_create: mmap failed: %s", strerror(errno));
        p->shm = static_cast<uint64_t *>(m);
        return NGCF_OK;
    };
    const int rc = body();
    if (rc != NGCF_OK) {
        ngcf_p2p_destroy(p);
        return rc;
    }
    *out = p;
    return NGCF_OK;
}

extern "C" int ngcf_p2p_handle(ngcf_p2p_t *p, void *handle64)
{
    if (!p || !handle64) return fail(NGCF_ERR_ARG, "p2p_handle: null argument");
    hipIpcMemHandle_t h;
    HIP_TRY(hipIpcGetMemHandle(&h, p->local));
    static_assert(sizeof(h) == 64, "IPC handle size");
    memcpy(handle64, &h, sizeof(h));
    return NGCF_OK;
}

extern "C" int ngcf_p2p_connect(ngcf_p2p_t *p, const void *handles)
{
    if (!p || !handles) return fail(NGCF_ERR_ARG, "p2p_connect: null argument");
    for (int q = 0; q < p->world; ++q) {
        if (q == p->rank || p->peer[(size_t)q]) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, static_cast<const char *>(handles) + (size_t)q * 64, 64);
        void *ptr = nullptr;
        HIP_TRY(hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess));
        p->peer[(size_t)q] = static_cast<char *>(ptr);
    }
    return NGCF_OK;
}

extern "C" void *ngcf_p2p_local(ngcf_p2p_t *p) { return p ? p->local : nullptr; }
extern "C" int64_t ngcf_p2p_bytes(const ngcf_p2p_t *p) { return p ? p->bytes : -1; }

// everything enqueued on `stream` so far is complete and visible to the other devices when the peers read seq in the slot
extern "C" int ngcf_p2p_publish(ngcf_p2p_t *p, int slot, uint64_t seq, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!p || slot < 0 || slot >= kP2pSlots) return fail(NGCF_ERR_ARG, "p2p_publish: bad argument");
    // the host function runs on a stream of its own behind the event: the producer's stream is not held up by the callback
    HIP_TRY(hipEventRecord(p->rel_ev[slot], stream));
    HIP_TRY(hipStreamWaitEvent(p->pub_stream, p->rel_ev[slot], 0));
    HIP_TRY(hipLaunchHostFunc(p->pub_stream, p2p_store_cb, p->store(p->pub(p->rank, slot), seq)));
    return NGCF_OK;
}

static int p2p_wait_word(ngcf_p2p *p, const uint64_t *w, uint64_t seq, double timeout_ms, const char *what, int who, int slot)
{
    ++p->waits;
    if (__atomic_load_n(w, __ATOMIC_ACQUIRE) >= seq) return NGCF_OK;     // the usual case: the hosts run ahead of their GPUs
    ++p->waits_blocked;
    const double t0 = now_ms();
    for (int spins = 0;; ++spins) {
        if (__atomic_load_n(w, __ATOMIC_ACQUIRE) >= seq) {
            p->blocked_ms += now_ms() - t0;
            return NGCF_OK;
        }
        if (spins > 2000) sched_yield();
        if ((spins & 1023) == 1023 && now_ms() - t0 > timeout_ms)
            return fail(NGCF_ERR_HIP, "p2p: timed out after %.0f ms waiting for %s of rank %d (slot %d, step %llu, seen %llu)", timeout_ms, what,
                        who, slot, (unsigned long long)seq, (unsigned long long)__atomic_load_n(w, __ATOMIC_ACQUIRE));
    }
}

// dst[0:bytes] = rank `peer`'s exchange buffer [src_off : src_off + bytes], once that rank has published `seq` in `slot`
extern "C" int ngcf_p2p_pull(ngcf_p2p_t *p, int peer, int slot, uint64_t seq, int64_t src_off, void *dst, int64_t bytes, double timeout_ms)
{
    if (!p || peer < 0 || peer >= p->world || slot < 0 || slot >= kP2pSlots || !dst || bytes < 0 || src_off < 0 || src_off + bytes > p->bytes)
        return fail(NGCF_ERR_ARG, "p2p_pull: bad argument");
    if (!p->peer[(size_t)peer]) return fail(NGCF_ERR_ARG, "p2p_pull: rank %d is not connected", peer);
    const int rc = p2p_wait_word(p, p->pub(peer, slot), seq, timeout_ms, "the data", peer, slot);
    if (rc != NGCF_OK) return rc;
    if (bytes == 0) return NGCF_OK;
    HIP_TRY(hipMemcpyAsync(dst, p->peer[(size_t)peer] + src_off, (size_t)bytes, hipMemcpyDeviceToDevice, p->copy[(size_t)peer]));
    p->copy_used[(size_t)peer] = 1;
    return NGCF_OK;
}

// "I have read everything of step seq out of rank peer's slot" - stored when the copies enqueued so far from that rank are done
extern "C" int ngcf_p2p_ack(ngcf_p2p_t *p, int peer, int slot, uint64_t seq)
{
    if (!p || peer < 0 || peer >= p->world || slot < 0 || slot >= kP2pSlots) return fail(NGCF_ERR_ARG, "p2p_ack: bad argument");
    HIP_TRY(hipLaunchHostFunc(p->copy[(size_t)peer], p2p_store_cb, p->store(p->ack(p->rank, peer, slot), seq)));
    return NGCF_OK;
}

// host-blocks until every other rank has acknowledged step seq of this rank's slot (before its region is overwritten)
extern "C" int ngcf_p2p_wait_acks(ngcf_p2p_t *p, int slot, uint64_t seq, double timeout_ms)
{
    if (!p || slot < 0 || slot >= kP2pSlots) return fail(NGCF_ERR_ARG, "p2p_wait_acks: bad argument");
    if (seq == 0) return NGCF_OK;
    for (int q = 0; q < p->world; ++q) {
        if (q == p->rank) continue;
        const int rc = p2p_wait_word(p, p->ack(q, p->rank, slot), seq, timeout_ms, "the acknowledgement", q, slot);
        if (rc != NGCF_OK) return rc;
    }
    return NGCF_OK;
}

// how long the host thread has been held in waits so far (the exchange puts no kernel on a CU and the data moves on copy engines;
// what it costs is host time in these waits: zero when the peers' publications are already there)
extern "C" int ngcf_p2p_stats(ngcf_p2p_t *p, double *blocked_ms, int64_t *waits, int64_t *waits_blocked, int reset)
{
    if (!p) return fail(NGCF_ERR_ARG, "p2p_stats: null argument");
    if (blocked_ms) *blocked_ms = p->blocked_ms;
    if (waits) *waits = p->waits;
    if (waits_blocked) *waits_blocked = p->waits_blocked;
    if (reset) {
        p->blocked_ms = 0;
        p->waits = p->waits_blocked = 0;
    }
    return NGCF_OK;
}

// the copies enqueued from now on start only after everything enqueued on `stream` so far (the last readers of their destinations)
extern "C" int ngcf_p2p_fence(ngcf_p2p_t *p, void *stream)
{
    if (!p) return fail(NGCF_ERR_ARG, "p2p_fence: null argument");
    HIP_TRY(hipEventRecord(p->fence_ev, (hipStream_t)stream));
    for (int q = 0; q < p->world; ++q) HIP_TRY(hipStreamWaitEvent(p->copy[(size_t)q], p->fence_ev, 0));
    return NGCF_OK;
}

// `stream` continues only after every copy enqueued since the last join has landed
extern "C" int ngcf_p2p_join(ngcf_p2p_t *p, void *stream)
{
    if (!p) return fail(NGCF_ERR_ARG, "p2p_join: null argument");
    for (int q = 0; q < p->world; ++q) {
        if (!p->copy_used[(size_t)q]) continue;
        HIP_TRY(hipEventRecord(p->copy_ev[(size_t)q], p->copy[(size_t)q]));
        HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, p->copy_ev[(size_t)q], 0));
        p->copy_used[(size_t)q] = 0;
    }
    return NGCF_OK;
}

// out[r, :] = sum over q < n_slots of slots[q][r, :] in slot order (the owner's side of a reduce-scatter: fixed order, so every
// run - and every rank count's re-run - adds the partial sums the same way)
__global__ void sum_slots_kernel(const float *__restrict__ slots, int64_t slot_stride, int n_slots, int64_t n, float *__restrict__ out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        float4 s = *reinterpret_cast<const float4 *>(slots + i);
        for (int q = 1; q < n_slots; ++q) {
            const float4 x = *reinterpret_cast<const float4 *>(slots + q * slot_stride + i);
            s.x += x.x;
            s.y += x.y;
            s.z += x.z;
            s.w += x.w;
        }
        *reinterpret_cast<float4 *>(out + i) = s;
    }
}

extern "C" int ngcf_sum_slots_f32(const float *slots, int64_t slot_stride, int n_slots, int64_t n, float *out, void *stream)
{
    if (n == 0) return NGCF_OK;
    if (!slots || !out || n_slots < 1 || n < 0 || n % 4 != 0 || slot_stride % 4 != 0 || !aligned16(slots) || !aligned16(out))
        return fail(NGCF_ERR_ARG, "sum_slots: bad argument (counts are multiples of 4 floats, 16-byte aligned)");
    sum_slots_kernel<<<grid_for(n / 4, 256), 256, 0, (hipStream_t)stream>>>(slots, slot_stride, n_slots, n, out);
    LAUNCH_CHECK();
    return NGCF_OK;
}
