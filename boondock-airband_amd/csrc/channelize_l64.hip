// channelize_l64.hip -- ahead-of-time instance and launcher of the lane-resident stage 1 (l64_kernel.h).
//
// The kernel takes its pruning masks at compile time.  Built here: the full graph (any plan at N = 512 with a supported hop).
// A plan's own instance -- exactly its butterflies, straight-line -- is compiled by hipRTC when the handle is created
// (l64_jit.cpp) and launched through the module API; if that is not possible the full-graph instance below runs instead.
#include <hip/hip_runtime.h>

#include "kernels.hpp"
#include "l64_kernel.h"

namespace mi {
namespace {

struct FullMasks {
    static constexpr unsigned long long n[6] = {0x3ull, 0xfull, 0xffull, 0xffffull, 0xffffffffull, ~0ull};
};

template <int HOP>
__global__ __launch_bounds__(256, 2) void k_channelize_l64(const L64Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char l64_lds[];
    mi_l64::l64_body<HOP, FullMasks>(a, l64_lds);
}

}  // namespace

// windows of a wave that go through the exchange buffer per round (l64_kernel.h, same rule)
int l64_round_windows(int m6) {
    return m6 <= 8 ? 8 : (m6 <= 16 ? 4 : (m6 <= 32 ? 2 : 1));
}
int l64_zstride(int m6) {
    int zs = m6 * mi_l64::kZRow;
    while ((zs / 4) % 32 != 16)  // consecutive windows 16 write banks apart
        zs += 16;
    return zs;
}

bool l64_supported(int log2n, size_t hop_bytes, int bytes_per_sample) {
    const size_t hop = hop_bytes / (2 * static_cast<size_t>(bytes_per_sample));
    return log2n == 9 && (hop == 160 || hop == 128);
}

hipError_t launch_channelize_l64(const ChannelizeArgs& c, int sfmt, int nstreams, hipStream_t s) {
    const unsigned bps2 = sfmt == MI_SFMT_S16 ? 4u : (sfmt == MI_SFMT_F32 ? 8u : 2u);
    const unsigned hop = c.hop_bytes / bps2;
    if (hop != 160 && hop != 128)
        return hipErrorInvalidValue;
    const L64Jit* jit = c.l64_jit;  // the plan's own instance, if it could be compiled
    const int m6 = jit ? c.l64.m6 : 64;
    L64Args a{};
    a.iq = c.iq;
    a.stream_stride = c.stream_stride;
    a.valid_bytes = c.valid_bytes;
    a.nfft = c.nfft;
    a.plane_off = c.plane_off;
    a.mag = c.mag;
    a.cplx = reinterpret_cast<float*>(c.cplx);
    a.plane_stride = c.plane_stride;
    a.window = c.window;
    a.levels = c.levels;
    a.conv_scale = c.conv_scale;
    a.nch = c.nch;
    a.n_iq_rows = c.n_iq_rows;
    a.xmax = c.xmax;
    a.chan = jit ? c.l64_chan : c.l64_chan_full;
    a.nb_pad = c.l64.nb_pad;
    a.zstride = static_cast<unsigned>(l64_zstride(m6));
    const unsigned nsamp = static_cast<unsigned>(mi_l64::kTile - 1) * hop + mi_l64::kN;
    const unsigned padb = 4u * ((16u - 2u * hop) & 63u);
    a.span_bytes = (8u * nsamp + padb * ((nsamp + hop - 1) / hop) + 15u) & ~15u;
    a.ntiles = (c.nfft + mi_l64::kTile - 1) / mi_l64::kTile;
    a.sfmt = sfmt;
    a.linear_tiles = c.l64.linear_tiles;
    // (the exchange buffer of stages 7..9, 4 waves x round windows x zstride, lies over the span)
    const size_t lds = static_cast<size_t>(a.span_bytes) + 4 * mi_l64::kN + 1024 + 16 + static_cast<size_t>(c.nch) * mi_l64::kTile * 4 +
                       static_cast<size_t>(c.n_iq_rows) * mi_l64::kTile * 8;
    if (lds > 160 * 1024 || 4u * static_cast<size_t>(l64_round_windows(m6)) * a.zstride > a.span_bytes || static_cast<unsigned long long>(a.ntiles) * static_cast<unsigned>(nstreams) >= (1ull << 32))
        return hipErrorInvalidValue;
    a.nstreams = static_cast<unsigned>(nstreams);
    // persistent workgroups that draw runs of contiguous tiles from a ticket counter (l64_kernel.h): as many as the machine
    // holds at once when this launch has it to itself -- fewer are resident when other kernels of the pipeline run alongside,
    // the rest then find the tickets gone
    const unsigned long long ttotal = static_cast<unsigned long long>(a.ntiles) * a.nstreams;
    static const int cus = [] {  // (a property of the machine, asked once)
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess)
            (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    // a multiple of the workgroups a CU holds at once (3 when the instance was compiled for three waves per SIMD and its LDS
    // allows it, else 2): a remainder would queue behind the resident ones and leave CUs idle at the end
    const int resident = (jit && l64_jit_minwaves(jit) >= 3 && lds * 3 <= 160 * 1024) ? 3 : 2;
    const unsigned long long want = static_cast<unsigned long long>(cus) * (c.l64.wg_per_cu > 0 ? c.l64.wg_per_cu : 2 * resident);
    // runs of up to 8 tiles (output cache lines shared by neighbouring tiles stay in one workgroup), shorter when the launch is
    // small: at least ~8 runs per workgroup, so that the last ones to finish are not far behind
    if (!c.l64_tickets || !c.l64_ticket_seq)
        return hipErrorInvalidValue;
    unsigned run = static_cast<unsigned>(ttotal / (want * 8ull));
    run = run < 1u ? 1u : (run > 8u ? 8u : run);
    a.run_tiles = run;
    const unsigned long long nruns = (ttotal + run - 1) / run;
    const unsigned gx = static_cast<unsigned>(nruns < want ? nruns : want);
    a.ticket = c.l64_tickets + (*c.l64_ticket_seq)++ % kL64Tickets;
    {
        hipError_t e = hipMemsetAsync(a.ticket, 0, sizeof(unsigned), s);
        if (e != hipSuccess)
            return e;
    }
    if (jit)
        return l64_jit_launch(jit, a, gx, 1u, lds, s);
    auto kern = hop == 160 ? k_channelize_l64<160> : k_channelize_l64<128>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        if (e != hipSuccess)
            return e;
    }
    hipLaunchKernelGGL(kern, dim3(gx), dim3(256), lds, s, a);
    return hipGetLastError();
}

}  // namespace mi
