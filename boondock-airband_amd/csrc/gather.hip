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

#include <cstring>
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
    void* comm = nullptr;
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

void mi_gather_destroy(mi_gather* g) {
    if (!g)
        return;
    (void)hipSetDevice(g->gpu);
    if (g->side)
        (void)hipStreamSynchronize(g->side);
    if (g->comm && rccl().ok)
        (void)rccl().comm_destroy(g->comm);
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
    if (world > 1) {
        Rccl& r = rccl();
        if (!r.ok)
            return bail(gfail(MI_ERR_UNSUPPORTED, "RCCL (librccl.so.1) could not be loaded"));
        const int rc = r.comm_init_rank(&g->comm, world, *id, rank);
        if (rc != 0)
            return bail(nccl_fail(rc, "ncclCommInitRank"));
    }
    *out = g;
    return MI_OK;
}

int mi_gather_audio(mi_gather* g, const float* d_waveout, const char* d_axc, int nbatches, int open_only, float* d_all_waveout, char* d_all_axc,
                    void* hip_stream) {
    if (!g || nbatches < 1 || nbatches > g->max_batches || (g->rows_local && (!d_waveout || !d_axc)) || (g->rank == 0 && (!d_all_waveout || !d_all_axc)))
        return gfail(MI_ERR_INVALID, "bad argument");
    G_HIP_TRY(hipSetDevice(g->gpu));
    Rccl& r = rccl();
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
                NCCL_TRY(r.group_start());
                for (int p = 1; p < g->world; ++p) {
                    const size_t rows = static_cast<size_t>(g->streams[p]) * g->nch;
                    if (!rows)
                        continue;
                    NCCL_TRY(r.recv(d_all_axc + g->row_lo[p] * nb, rows * nb, kNcclChar, p, g->comm, q));
                    NCCL_TRY(r.recv(d_all_waveout + g->row_lo[p] * n, rows * n, kNcclFloat, p, g->comm, q));
                }
                NCCL_TRY(r.group_end());
            }
        } else if (g->rows_local) {
            NCCL_TRY(r.group_start());
            NCCL_TRY(r.send(d_axc, g->rows_local * nb, kNcclChar, 0, g->comm, q));
            NCCL_TRY(r.send(d_waveout, g->rows_local * n, kNcclFloat, 0, g->comm, q));
            NCCL_TRY(r.group_end());
        }
        G_HIP_TRY(hipEventRecord(g->ev_done, q));
        return MI_OK;
    }
    // ---- open batches only ----
    const size_t max_blocks = (g->rank == 0 ? g->rows_total : g->rows_local) * static_cast<size_t>(g->max_batches);
    if (!g->d_pack && max_blocks) {
        G_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&g->d_pack), max_blocks * mi::kWaveBatch * sizeof(float)));
        G_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&g->d_idx), max_blocks * sizeof(int)));
        G_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&g->h_flags), max_blocks, hipHostMallocDefault));
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
        NCCL_TRY(r.group_start());
        NCCL_TRY(r.send(d_axc, blocks, kNcclChar, 0, g->comm, q));
        NCCL_TRY(r.group_end());
        G_HIP_TRY(hipStreamSynchronize(q));
        g->idx_host.clear();
        open_blocks(g->h_flags, blocks, 0);
        const size_t k = g->idx_host.size();
        if (k) {
            G_HIP_TRY(hipMemcpyAsync(g->d_idx, g->idx_host.data(), k * sizeof(int), hipMemcpyHostToDevice, q));
            hipLaunchKernelGGL(k_move_blocks, dim3(static_cast<unsigned>(k)), dim3(256), 0, q, d_waveout, g->d_pack, g->d_idx, static_cast<int>(k), 0);
            G_HIP_TRY(hipGetLastError());
            NCCL_TRY(r.group_start());
            NCCL_TRY(r.send(g->d_pack, k * mi::kWaveBatch, kNcclFloat, 0, g->comm, q));
            NCCL_TRY(r.group_end());
        }
        G_HIP_TRY(hipEventRecord(g->ev_done, q));
        return MI_OK;
    }
    // rank 0: everyone's flags first, then each peer's open blocks into the landing area, then scatter into zeroed arrays
    if (g->rows_local)
        G_HIP_TRY(hipMemcpyAsync(d_all_axc + g->row_lo[0] * nb, d_axc, g->rows_local * nb, hipMemcpyDeviceToDevice, q));
    if (g->world > 1) {
        NCCL_TRY(r.group_start());
        for (int p = 1; p < g->world; ++p) {
            const size_t rows = static_cast<size_t>(g->streams[p]) * g->nch;
            if (rows)
                NCCL_TRY(r.recv(d_all_axc + g->row_lo[p] * nb, rows * nb, kNcclChar, p, g->comm, q));
        }
        NCCL_TRY(r.group_end());
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
            NCCL_TRY(r.group_start());
            for (int p = 1; p < g->world; ++p) {
                const size_t cnt = first[static_cast<size_t>(p) + 1] - first[static_cast<size_t>(p)];
                if (cnt)
                    NCCL_TRY(r.recv(g->d_pack + first[static_cast<size_t>(p)] * mi::kWaveBatch, cnt * mi::kWaveBatch, kNcclFloat, p, g->comm, q));
            }
            NCCL_TRY(r.group_end());
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
