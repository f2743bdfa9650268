// gather.hip -- the one collective of a multi-GPU job: decimated audio + axcindicate flags of every rank to rank 0, where the
// reference's output / mixer threads live (output.cpp:899-961).  C ABI: mi_gather_* (include/mi_airband.h); Python twin:
// boondock-airband_amd/shard.py (torch.distributed).  Streams are independent devices (one demod thread per device in the
// reference, rtl_airband.cpp:1044-1078): nothing else crosses GPUs.
//
// RCCL point-to-point: rank r sends its [streams_r][nch][n] audio and [streams_r][nch][nbatches] flags, rank 0 receives them
// into the stream-major arrays of the whole job, all transfers of a step in one group on the gather's own stream (they run
// side by side over xGMI; rank-0 ingress is the bound).  The caller's stream is only touched by two event waits:
// the gather starts when the audio is complete and overlaps whatever the caller enqueues next.
// "Open batches only": the flags travel whole (1/8000 of the audio); of the audio only the (row, batch) blocks whose flag is
// not NO_SIGNAL -- what the reference's non-continuous outputs consume (output.cpp:518,568) -- compacted on the sender and
// scattered on rank 0 into zeroed arrays.  That mode needs the counts on the host: one stream synchronisation per step.
// RCCL is looked up with dlopen (librccl.so.1): the library has no link-time dependency on it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mi_airband.h"
#include "plan.hpp"

namespace mi {
std::string& last_error_ref();
}

namespace {

int gfail(int code, const std::string& what) {
    mi::last_error_ref() = what;
    return code;
}

struct Rccl {
    void* lib = nullptr;
    int (*get_unique_id)(void*) = nullptr;
    int (*comm_init_rank)(void**, int, mi_gather_id, int) = nullptr;
    int (*comm_destroy)(void*) = nullptr;
    int (*send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*group_start)() = nullptr;
    int (*group_end)() = nullptr;
    const char* (*error_string)(int) = nullptr;
    bool ok = false;
};
constexpr int kNcclChar = 0, kNcclFloat = 7;  // ncclDataType_t, rccl.h:459-466

Rccl& rccl() {
    static Rccl r = [] {
        Rccl x;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            x.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (x.lib)
                break;
        }
        if (!x.lib)
            return x;
        auto sym = [&](const char* n) { return dlsym(x.lib, n); };
        x.get_unique_id = reinterpret_cast<decltype(x.get_unique_id)>(sym("ncclGetUniqueId"));
        x.comm_init_rank = reinterpret_cast<decltype(x.comm_init_rank)>(sym("ncclCommInitRank"));
        x.comm_destroy = reinterpret_cast<decltype(x.comm_destroy)>(sym("ncclCommDestroy"));
        x.send = reinterpret_cast<decltype(x.send)>(sym("ncclSend"));
        x.recv = reinterpret_cast<decltype(x.recv)>(sym("ncclRecv"));
        x.group_start = reinterpret_cast<decltype(x.group_start)>(sym("ncclGroupStart"));
        x.group_end = reinterpret_cast<decltype(x.group_end)>(sym("ncclGroupEnd"));
        x.error_string = reinterpret_cast<decltype(x.error_string)>(sym("ncclGetErrorString"));
        x.ok = x.get_unique_id && x.comm_init_rank && x.comm_destroy && x.send && x.recv && x.group_start && x.group_end;
        return x;
    }();
    return r;
}

int nccl_fail(int rc, const char* where) {
    Rccl& r = rccl();
    return gfail(MI_ERR_HIP, std::string(where) + ": " + (r.error_string ? r.error_string(rc) : "RCCL error"));
}

#define NCCL_TRY(expr)                    \
    do {                                  \
        const int rc__ = (expr);          \
        if (rc__ != 0)                    \
            return nccl_fail(rc__, #expr); \
    } while (0)
#define G_HIP_TRY(expr)                                                                 \
    do {                                                                                \
        const hipError_t e__ = (expr);                                                  \
        if (e__ != hipSuccess)                                                          \
            return gfail(MI_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)); \
    } while (0)

// ---- the transport under the gather: point-to-point send / recv in groups, RCCL's semantics ----
// Two implementations: RCCL over xGMI (what a multi-GPU job runs), and a loopback between threads of ONE process on one GPU, which
// lets a single-GPU box drive every world > 1 branch of mi_gather_audio -- the grouped transfers, the two-phase open-only protocol,
// uneven and empty ranks -- with the same call sequence (tests/test_gather_c_abi.py).  The loopback is a test transport: a send
// blocks its host thread until the peer has posted the matching recv (RCCL would queue it), which the gather's call order allows.
struct Transport {
    virtual ~Transport() {}
    virtual int send(const void* buf, size_t count, int type, int peer, hipStream_t s) = 0;
    virtual int recv(void* buf, size_t count, int type, int peer, hipStream_t s) = 0;
    virtual int group_start() = 0;
    virtual int group_end() = 0;
};

struct RcclTransport : Transport {
    void* comm = nullptr;
    ~RcclTransport() override {
        if (comm && rccl().ok)
            (void)rccl().comm_destroy(comm);
    }
    int send(const void* buf, size_t count, int type, int peer, hipStream_t s) override {
        const int rc = rccl().send(buf, count, type, peer, comm, s);
        return rc ? nccl_fail(rc, "ncclSend") : MI_OK;
    }
    int recv(void* buf, size_t count, int type, int peer, hipStream_t s) override {
        const int rc = rccl().recv(buf, count, type, peer, comm, s);
        return rc ? nccl_fail(rc, "ncclRecv") : MI_OK;
    }
    int group_start() override {
        const int rc = rccl().group_start();
        return rc ? nccl_fail(rc, "ncclGroupStart") : MI_OK;
    }
    int group_end() override {
        const int rc = rccl().group_end();
        return rc ? nccl_fail(rc, "ncclGroupEnd") : MI_OK;
    }
};

constexpr char kLoopMagic[8] = {'M', 'I', 'L', 'O', 'O', 'P', '0', '1'};
struct LoopMsg {
    const void* src = nullptr;
    size_t bytes = 0;
    hipEvent_t ready = nullptr;  // recorded on the sender's stream: the data is there
    hipEvent_t done = nullptr;   // recorded on the receiver's stream: the copy has been made
    bool taken = false, failed = false;
};
struct LoopHub {  // one per loopback id: the mailboxes (src rank, dst rank) of a job whose ranks are threads
    std::mutex mu;
    std::condition_variable cv;
    std::map<std::pair<int, int>, std::deque<LoopMsg*>> box;
};
std::mutex g_hub_mu;
std::map<uint64_t, std::shared_ptr<LoopHub>> g_hubs;

struct LoopTransport : Transport {
    std::shared_ptr<LoopHub> hub;
    int rank = 0;
    static size_t bytes_of(size_t count, int type) { return count * (type == kNcclFloat ? sizeof(float) : 1); }
    int send(const void* buf, size_t count, int type, int peer, hipStream_t s) override {
        LoopMsg m;
        m.src = buf, m.bytes = bytes_of(count, type);
        if (hipEventCreateWithFlags(&m.ready, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&m.done, hipEventDisableTiming) != hipSuccess ||
            hipEventRecord(m.ready, s) != hipSuccess)
            return gfail(MI_ERR_HIP, "loopback send: event");
        std::unique_lock<std::mutex> lk(hub->mu);
        hub->box[{rank, peer}].push_back(&m);
        hub->cv.notify_all();
        const bool ok = hub->cv.wait_for(lk, std::chrono::seconds(60), [&] { return m.taken; });
        if (!ok) {  // nobody came for it: take it back
            auto& q = hub->box[{rank, peer}];
            for (auto it = q.begin(); it != q.end(); ++it)
                if (*it == &m) {
                    q.erase(it);
                    break;
                }
        }
        lk.unlock();
        // the buffer may be reused by whatever the sender enqueues next: behind the receiver's copy
        hipError_t e = (ok && !m.failed) ? hipStreamWaitEvent(s, m.done, 0) : hipSuccess;
        (void)hipEventDestroy(m.ready);
        if (ok)
            (void)hipEventSynchronize(m.done);  // (the event object dies with this frame)
        (void)hipEventDestroy(m.done);
        if (!ok)
            return gfail(MI_ERR_HIP, "loopback send: no matching recv within 60 s");
        if (m.failed || e != hipSuccess)
            return gfail(MI_ERR_HIP, "loopback send: the receiver's copy failed");
        return MI_OK;
    }
    int recv(void* buf, size_t count, int type, int peer, hipStream_t s) override {
        std::unique_lock<std::mutex> lk(hub->mu);
        auto& q = hub->box[{peer, rank}];
        if (!hub->cv.wait_for(lk, std::chrono::seconds(60), [&] { return !q.empty(); }))
            return gfail(MI_ERR_HIP, "loopback recv: no matching send within 60 s");
        LoopMsg* m = q.front();
        q.pop_front();
        hipError_t e = (m->bytes == bytes_of(count, type)) ? hipSuccess : hipErrorInvalidValue;  // RCCL would hang or corrupt: the test transport says so
        if (e == hipSuccess)
            e = hipStreamWaitEvent(s, m->ready, 0);
        if (e == hipSuccess && m->bytes)
            e = hipMemcpyAsync(buf, m->src, m->bytes, hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess)
            e = hipEventRecord(m->done, s);
        m->failed = e != hipSuccess;
        m->taken = true;
        hub->cv.notify_all();
        lk.unlock();
        if (e != hipSuccess)
            return gfail(MI_ERR_HIP, std::string("loopback recv: ") + (e == hipErrorInvalidValue ? "send / recv sizes differ" : hipGetErrorString(e)));
        return MI_OK;
    }
    int group_start() override { return MI_OK; }
    int group_end() override { return MI_OK; }
};

// every transfer of a step between group_start and group_end; an error in between still closes the group (RCCL keeps an open
// group per thread: leaving it open would swallow the next step's calls)
struct Group {
    Transport* t;
    bool open = false;
    explicit Group(Transport* tt) : t(tt) {}
    int start() {
        const int rc = t->group_start();
        open = rc == MI_OK;
        return rc;
    }
    int end() {
        open = false;
        return t->group_end();
    }
    ~Group() {
        if (open)
            (void)t->group_end();
    }
};
#define T_TRY(expr)             \
    do {                        \
        const int rc__ = (expr); \
        if (rc__ != MI_OK)      \
            return rc__;        \
    } while (0)

// blocks of WAVE_BATCH floats: dst block j = src block idx[j] (compaction on a sender), or dst block idx[j] = src block j
// (scatter on rank 0); 500 lanes x float4 per block
__global__ __launch_bounds__(256) void k_move_blocks(const float* __restrict__ src, float* __restrict__ dst, const int* __restrict__ idx, int nblocks, int scatter) {
    const int j = blockIdx.x;
    if (j >= nblocks)
        return;
    const size_t from = static_cast<size_t>(scatter ? j : idx[j]) * mi::kWaveBatch, to = static_cast<size_t>(scatter ? idx[j] : j) * mi::kWaveBatch;
    const float4* s = reinterpret_cast<const float4*>(src + from);
    float4* d = reinterpret_cast<float4*>(dst + to);
    for (int i = threadIdx.x; i < mi::kWaveBatch / 4; i += blockDim.x)
        d[i] = s[i];
}

}  // namespace

struct mi_gather {
    int rank = 0, world = 1, gpu = 0, nch = 0, max_batches = 0;
    std::vector<int> streams;      // per rank
    std::vector<size_t> row_lo;    // first row (stream * nch + channel) of each rank in the job-wide arrays
    size_t rows_local = 0, rows_total = 0;
    std::unique_ptr<Transport> tr;  // null at world 1
    hipStream_t side = nullptr;
    hipEvent_t ev_in = nullptr, ev_done = nullptr;
    // open-batches-only mode
    float* d_pack = nullptr;       // sender: compacted blocks; rank 0: landing area of the peers' blocks
    int* d_idx = nullptr;
    char* h_flags = nullptr;       // pinned: this rank's flags (sender) / everyone's flags (rank 0)
    std::vector<int> idx_host;
};

extern "C" {

int mi_gather_unique_id(mi_gather_id* id) {
    if (!id)
        return gfail(MI_ERR_INVALID, "NULL argument");
    Rccl& r = rccl();
    if (!r.ok)
        return gfail(MI_ERR_UNSUPPORTED, "RCCL (librccl.so.1) could not be loaded");
    NCCL_TRY(r.get_unique_id(id));
    return MI_OK;
}

int mi_gather_loopback_id(mi_gather_id* id, uint64_t job) {
    if (!id)
        return gfail(MI_ERR_INVALID, "NULL argument");
    std::memset(id->internal, 0, sizeof(id->internal));
    std::memcpy(id->internal, kLoopMagic, sizeof(kLoopMagic));
    std::memcpy(id->internal + 8, &job, sizeof(job));
    return MI_OK;
}

void mi_gather_destroy(mi_gather* g) {
    if (!g)
        return;
    (void)hipSetDevice(g->gpu);
    if (g->side)
        (void)hipStreamSynchronize(g->side);
    g->tr.reset();
    if (g->d_pack)
        (void)hipFree(g->d_pack);
    if (g->d_idx)
        (void)hipFree(g->d_idx);
    if (g->h_flags)
        (void)hipHostFree(g->h_flags);
    if (g->ev_in)
        (void)hipEventDestroy(g->ev_in);
    if (g->ev_done)
        (void)hipEventDestroy(g->ev_done);
    if (g->side)
        (void)hipStreamDestroy(g->side);
    delete g;
}

int mi_gather_create(const mi_gather_id* id, int rank, int world, int gpu, const int* streams_per_rank, int nch, int max_batches, mi_gather** out) {
    if (!out)
        return gfail(MI_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!streams_per_rank || world < 1 || rank < 0 || rank >= world || nch < 1 || max_batches < 1 || (world > 1 && !id))
        return gfail(MI_ERR_INVALID, "bad gather geometry");
    mi_gather* g = new (std::nothrow) mi_gather();
    if (!g)
        return gfail(MI_ERR_NOMEM, "host allocation failed");
    g->rank = rank, g->world = world, g->gpu = gpu, g->nch = nch, g->max_batches = max_batches;
    g->streams.assign(streams_per_rank, streams_per_rank + world);
    size_t lo = 0;
    for (int r = 0; r < world; ++r) {
        if (streams_per_rank[r] < 0) {
            delete g;
            return gfail(MI_ERR_INVALID, "negative stream count");
        }
        g->row_lo.push_back(lo);
        lo += static_cast<size_t>(streams_per_rank[r]) * nch;
    }
    g->rows_total = lo;
    g->rows_local = static_cast<size_t>(streams_per_rank[rank]) * nch;
    auto bail = [&](int code) {
        const std::string keep = mi::last_error_ref();
        mi_gather_destroy(g);
        mi::last_error_ref() = keep;
        return code;
    };
    hipError_t e = hipSetDevice(gpu);
    if (e == hipSuccess)
        e = hipStreamCreateWithFlags(&g->side, hipStreamNonBlocking);
    if (e == hipSuccess)
        e = hipEventCreateWithFlags(&g->ev_in, hipEventDisableTiming);
    if (e == hipSuccess)
        e = hipEventCreateWithFlags(&g->ev_done, hipEventDisableTiming);
    if (e != hipSuccess)
        return bail(gfail(e == hipErrorNoDevice || e == hipErrorInvalidDevice ? MI_ERR_NO_DEVICE : MI_ERR_HIP, std::string("mi_gather_create: ") + hipGetErrorString(e)));
    if (world > 1 && std::memcmp(id->internal, kLoopMagic, sizeof(kLoopMagic)) == 0) {
        uint64_t key = 0;
        std::memcpy(&key, id->internal + 8, sizeof(key));
        auto t = std::make_unique<LoopTransport>();
        t->rank = rank;
        {
            std::lock_guard<std::mutex> lk(g_hub_mu);
            std::shared_ptr<LoopHub>& hub = g_hubs[key];
            if (!hub)
                hub = std::make_shared<LoopHub>();
            t->hub = hub;
        }
        g->tr = std::move(t);
    } else if (world > 1) {
        Rccl& r = rccl();
        if (!r.ok)
            return bail(gfail(MI_ERR_UNSUPPORTED, "RCCL (librccl.so.1) could not be loaded"));
        auto t = std::make_unique<RcclTransport>();
        const int rc = r.comm_init_rank(&t->comm, world, *id, rank);
        if (rc != 0)
            return bail(nccl_fail(rc, "ncclCommInitRank"));
        g->tr = std::move(t);
    }
    *out = g;
    return MI_OK;
}

int mi_gather_audio(mi_gather* g, const float* d_waveout, const char* d_axc, int nbatches, int open_only, float* d_all_waveout, char* d_all_axc,
                    void* hip_stream) {
    if (!g || nbatches < 1 || nbatches > g->max_batches || (g->rows_local && (!d_waveout || !d_axc)) || (g->rank == 0 && (!d_all_waveout || !d_all_axc)))
        return gfail(MI_ERR_INVALID, "bad argument");
    G_HIP_TRY(hipSetDevice(g->gpu));
    Transport* const t = g->tr.get();
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    const size_t n = static_cast<size_t>(nbatches) * mi::kWaveBatch;
    const size_t nb = static_cast<size_t>(nbatches);
    // the gather's stream starts when the caller's stream has produced the audio (and, on rank 0, freed the destination)
    G_HIP_TRY(hipEventRecord(g->ev_in, s));
    G_HIP_TRY(hipStreamWaitEvent(g->side, g->ev_in, 0));
    hipStream_t q = g->side;
    if (!open_only) {
        if (g->rank == 0) {
            if (g->rows_local) {
                G_HIP_TRY(hipMemcpyAsync(d_all_waveout + g->row_lo[0] * n, d_waveout, g->rows_local * n * sizeof(float), hipMemcpyDeviceToDevice, q));
                G_HIP_TRY(hipMemcpyAsync(d_all_axc + g->row_lo[0] * nb, d_axc, g->rows_local * nb, hipMemcpyDeviceToDevice, q));
            }
            if (g->world > 1) {
                Group grp(t);
                T_TRY(grp.start());
                for (int p = 1; p < g->world; ++p) {
                    const size_t rows = static_cast<size_t>(g->streams[p]) * g->nch;
                    if (!rows)
                        continue;
                    T_TRY(t->recv(d_all_axc + g->row_lo[p] * nb, rows * nb, kNcclChar, p, q));
                    T_TRY(t->recv(d_all_waveout + g->row_lo[p] * n, rows * n, kNcclFloat, p, q));
                }
                T_TRY(grp.end());
            }
        } else if (g->rows_local) {
            Group grp(t);
            T_TRY(grp.start());
            T_TRY(t->send(d_axc, g->rows_local * nb, kNcclChar, 0, q));
            T_TRY(t->send(d_waveout, g->rows_local * n, kNcclFloat, 0, q));
            T_TRY(grp.end());
        }
        G_HIP_TRY(hipEventRecord(g->ev_done, q));
        return MI_OK;
    }
    // ---- open batches only ----
    const size_t max_blocks = (g->rank == 0 ? g->rows_total : g->rows_local) * static_cast<size_t>(g->max_batches);
    if (!g->d_pack && max_blocks) {  // all three or none: a half-allocated scratch must not reach the kernels below
        float* pack = nullptr;
        int* idx = nullptr;
        char* flags = nullptr;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&pack), max_blocks * mi::kWaveBatch * sizeof(float));
        if (e == hipSuccess)
            e = hipMalloc(reinterpret_cast<void**>(&idx), max_blocks * sizeof(int));
        if (e == hipSuccess)
            e = hipHostMalloc(reinterpret_cast<void**>(&flags), max_blocks, hipHostMallocDefault);
        if (e != hipSuccess) {
            if (pack)
                (void)hipFree(pack);
            if (idx)
                (void)hipFree(idx);
            return gfail(MI_ERR_NOMEM, std::string("mi_gather_audio: scratch of the open-batches mode: ") + hipGetErrorString(e));
        }
        g->d_pack = pack, g->d_idx = idx, g->h_flags = flags;
    }
    auto open_blocks = [&](const char* flags, size_t count, size_t base) {  // indices (relative to `base`) of the blocks that carry a signal
        for (size_t i = 0; i < count; ++i)
            if (flags[i] != MI_NO_SIGNAL)
                g->idx_host.push_back(static_cast<int>(base + i));
    };
    if (g->rank != 0) {
        if (!g->rows_local)
            return MI_OK;
        const size_t blocks = g->rows_local * nb;
        G_HIP_TRY(hipMemcpyAsync(g->h_flags, d_axc, blocks, hipMemcpyDeviceToHost, q));
        {
            Group grp(t);
            T_TRY(grp.start());
            T_TRY(t->send(d_axc, blocks, kNcclChar, 0, q));
            T_TRY(grp.end());
        }
        G_HIP_TRY(hipStreamSynchronize(q));
        g->idx_host.clear();
        open_blocks(g->h_flags, blocks, 0);
        const size_t k = g->idx_host.size();
        if (k) {
            G_HIP_TRY(hipMemcpyAsync(g->d_idx, g->idx_host.data(), k * sizeof(int), hipMemcpyHostToDevice, q));
            hipLaunchKernelGGL(k_move_blocks, dim3(static_cast<unsigned>(k)), dim3(256), 0, q, d_waveout, g->d_pack, g->d_idx, static_cast<int>(k), 0);
            G_HIP_TRY(hipGetLastError());
            Group grp(t);
            T_TRY(grp.start());
            T_TRY(t->send(g->d_pack, k * mi::kWaveBatch, kNcclFloat, 0, q));
            T_TRY(grp.end());
        }
        G_HIP_TRY(hipEventRecord(g->ev_done, q));
        return MI_OK;
    }
    // rank 0: everyone's flags first, then each peer's open blocks into the landing area, then scatter into zeroed arrays
    if (g->rows_local)
        G_HIP_TRY(hipMemcpyAsync(d_all_axc + g->row_lo[0] * nb, d_axc, g->rows_local * nb, hipMemcpyDeviceToDevice, q));
    if (g->world > 1) {
        Group grp(t);
        T_TRY(grp.start());
        for (int p = 1; p < g->world; ++p) {
            const size_t rows = static_cast<size_t>(g->streams[p]) * g->nch;
            if (rows)
                T_TRY(t->recv(d_all_axc + g->row_lo[p] * nb, rows * nb, kNcclChar, p, q));
        }
        T_TRY(grp.end());
    }
    G_HIP_TRY(hipMemcpyAsync(g->h_flags, d_all_axc, g->rows_total * nb, hipMemcpyDeviceToHost, q));
    G_HIP_TRY(hipMemsetAsync(d_all_waveout, 0, g->rows_total * n * sizeof(float), q));
    G_HIP_TRY(hipStreamSynchronize(q));
    g->idx_host.clear();
    std::vector<size_t> first(static_cast<size_t>(g->world) + 1, 0);  // landing-area position of each rank's first block
    for (int p = 0; p < g->world; ++p) {
        const size_t rows = static_cast<size_t>(g->streams[p]) * g->nch;
        open_blocks(g->h_flags + g->row_lo[p] * nb, rows * nb, g->row_lo[p] * nb);
        first[static_cast<size_t>(p) + 1] = g->idx_host.size();
    }
    const size_t k = g->idx_host.size();
    if (k) {
        G_HIP_TRY(hipMemcpyAsync(g->d_idx, g->idx_host.data(), k * sizeof(int), hipMemcpyHostToDevice, q));
        if (first[1]) {  // rank 0's own open blocks: compact them like a sender would, so that one scatter serves all
            // (their indices are relative to the job-wide array, whose rank-0 part starts at row_lo[0] = 0)
            hipLaunchKernelGGL(k_move_blocks, dim3(static_cast<unsigned>(first[1])), dim3(256), 0, q, d_waveout, g->d_pack, g->d_idx, static_cast<int>(first[1]), 0);
            G_HIP_TRY(hipGetLastError());
        }
        if (g->world > 1) {
            Group grp(t);
            T_TRY(grp.start());
            for (int p = 1; p < g->world; ++p) {
                const size_t cnt = first[static_cast<size_t>(p) + 1] - first[static_cast<size_t>(p)];
                if (cnt)
                    T_TRY(t->recv(g->d_pack + first[static_cast<size_t>(p)] * mi::kWaveBatch, cnt * mi::kWaveBatch, kNcclFloat, p, q));
            }
            T_TRY(grp.end());
        }
        hipLaunchKernelGGL(k_move_blocks, dim3(static_cast<unsigned>(k)), dim3(256), 0, q, g->d_pack, d_all_waveout, g->d_idx, static_cast<int>(k), 1);
        G_HIP_TRY(hipGetLastError());
    }
    G_HIP_TRY(hipEventRecord(g->ev_done, q));
    return MI_OK;
}

int mi_gather_stream_wait(mi_gather* g, void* hip_stream) {
    if (!g)
        return gfail(MI_ERR_INVALID, "NULL argument");
    G_HIP_TRY(hipSetDevice(g->gpu));
    G_HIP_TRY(hipStreamWaitEvent(static_cast<hipStream_t>(hip_stream), g->ev_done, 0));
    return MI_OK;
}

int mi_gather_sync(mi_gather* g) {
    if (!g)
        return gfail(MI_ERR_INVALID, "NULL argument");
    G_HIP_TRY(hipSetDevice(g->gpu));
    G_HIP_TRY(hipStreamSynchronize(g->side));
    return MI_OK;
}

}  // extern "C"
