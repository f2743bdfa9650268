// channelize.hip -- stage 1 of the hot path on gfx950: sample conversion x window, sliding FFT, bin
// pick + magnitude.  Replaces rtl_airband.cpp:424-511 (and the NEON samplefft(), rtl_airband_neon.s:28-83,
// and FFTW3f / hello_fft) for a whole tile of consecutive windows per workgroup.
//
// One workgroup = TW consecutive windows of one stream.  The byte span the TW windows cover
// ((TW-1)*hop + fft_size samples; windows overlap 3.2x at fft 512) is read from HBM ONCE with 16-byte
// coalesced loads into LDS; every window then converts its samples out of LDS.  An FFT of N points is
// done by N/8 lanes, 8 points per lane, as radix-2 DIT stages taken three at a time in registers with
// one LDS exchange between passes.  The arithmetic of every butterfly is the documented FFT spec
// (DESIGN.md): t = w*b as {fma(-b.im, w.im, b.re*w.re), fma(b.im, w.re, b.re*w.im)}, a' = a + t,
// b' = a - t, so the result is bit-identical to the CPU oracle (zero signs aside).  Only the
// channel bins leave the chip: magnitude (and re/im for channels that need raw I/Q), staged per tile
// in LDS and written as contiguous rows of the [stream][channel][time] planes stage 2 reads.
//
// Compile with -ffp-contract=off: products and sums must round separately except where fma is spelled.
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace mi {
namespace {

template <int L>
struct FftGeom {
    static constexpr int N = 1 << L;
    static constexpr int TPF = N / 8;                      // lanes per FFT
    static constexpr int BLOCK = TPF < 256 ? 256 : TPF;    // threads per workgroup
    static constexpr int F = BLOCK / TPF;                  // FFTs in flight per workgroup
    static constexpr int NPASS = (L + 2) / 3;
    static constexpr int TW = L <= 10 ? 64 : (L == 11 ? 32 : (L == 12 ? 16 : 8));  // windows per tile
    static constexpr int XN = N + N / 8;                   // padded exchange length (float2)
};

// Lanes of one FFT exchange data through their slot of `xch`.  Up to N = 512 they all sit in one wave (64 lanes x 8 points;
// N = 256: two FFTs per wave), whose LDS operations execute in order: a wave-level fence orders the exchange, and the four
// waves of a workgroup never wait for each other inside the window loop.  Larger FFTs span waves and need the barrier.
template <int L>
__device__ __forceinline__ void fft_sync() {
    if constexpr ((1 << L) / 8 <= 64) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

__device__ __forceinline__ int xpad(int p) {
    return p + (p >> 3);
}

// generic butterfly (a, b) -> (a + w b, a - w b)
__device__ __forceinline__ void bfly(float2& a, float2& b, const float2 w) {
    const float tr = __builtin_fmaf(-b.y, w.y, b.x * w.x);
    const float ti = __builtin_fmaf(b.y, w.x, b.x * w.y);
    const float2 a0 = a;
    a.x = a0.x + tr;
    a.y = a0.y + ti;
    b.x = a0.x - tr;
    b.y = a0.y - ti;
}
// w = 1
__device__ __forceinline__ void bfly_one(float2& a, float2& b) {
    const float2 a0 = a, b0 = b;
    a.x = a0.x + b0.x;
    a.y = a0.y + b0.y;
    b.x = a0.x - b0.x;
    b.y = a0.y - b0.y;
}
// w = -j : t = (b.im, -b.re)
__device__ __forceinline__ void bfly_negj(float2& a, float2& b) {
    const float2 a0 = a, b0 = b;
    a.x = a0.x + b0.y;
    a.y = a0.y - b0.x;
    b.x = a0.x - b0.y;
    b.y = a0.y + b0.x;
}

// Pass K covers DIT stages 3K+1 .. min(3K+3, L).  A lane owns 8 elements that are closed under those
// stages: groups of G = 2^nst elements at stride S = 8^K.
template <int L, int K>
struct Pass {
    static constexpr int S = 1 << (3 * K);
    static constexpr int NST = (L - 3 * K) >= 3 ? 3 : (L - 3 * K);
    static constexpr int G = 1 << NST;
    static constexpr int GROUPS = 8 / G;
    static constexpr int NTW = GROUPS * (G - 1);  // distinct twiddles a lane needs in this pass (<= 7)

    __device__ static __forceinline__ int pos(int tau, int gi, int ri) {
        const int u = tau * GROUPS + gi;
        const int hi = u >> (3 * K);
        const int lo = u & (S - 1);
        return ((hi * G + ri) << (3 * K)) + lo;
    }
    // twiddle exponent of the butterfly (ri, ri + 2^a), stage 3K+1+a
    __device__ static __forceinline__ int tw_exp(int tau, int gi, int a, int m) {
        const int u = tau * GROUPS + gi;
        const int lo = u & (S - 1);
        return (m * S + lo) * ((1 << L) >> (3 * K + 1 + a));
    }
    __device__ static __forceinline__ void load_tw(int tau, const float2* __restrict__ tw, float2 (&r)[7]) {
#pragma unroll
        for (int gi = 0; gi < GROUPS; ++gi)
#pragma unroll
            for (int a = 0; a < NST; ++a)
#pragma unroll
                for (int m = 0; m < (1 << a); ++m)
                    r[gi * (G - 1) + ((1 << a) - 1) + m] = tw[tw_exp(tau, gi, a, m)];
    }
    __device__ static __forceinline__ void run(float2 (&x)[8], const float2 (&r)[7]) {
#pragma unroll
        for (int gi = 0; gi < GROUPS; ++gi)
#pragma unroll
            for (int a = 0; a < NST; ++a)
#pragma unroll
                for (int ri = 0; ri < G; ++ri)
                    if ((ri & (1 << a)) == 0) {
                        const int m = ri & ((1 << a) - 1);
                        bfly(x[gi * G + ri], x[gi * G + ri + (1 << a)], r[gi * (G - 1) + ((1 << a) - 1) + m]);
                    }
    }
};

// pass 0: S = 1, lo = 0, twiddles are 1, -j, W8 = tw[N/8], W8^3 = tw[3N/8]
__device__ __forceinline__ void pass0(float2 (&x)[8], const float2 w8, const float2 w83) {
    bfly_one(x[0], x[1]);
    bfly_one(x[2], x[3]);
    bfly_one(x[4], x[5]);
    bfly_one(x[6], x[7]);
    bfly_one(x[0], x[2]);
    bfly_negj(x[1], x[3]);
    bfly_one(x[4], x[6]);
    bfly_negj(x[5], x[7]);
    bfly_one(x[0], x[4]);
    bfly(x[1], x[5], w8);
    bfly_negj(x[2], x[6]);
    bfly(x[3], x[7], w83);
}

template <int L, int K>
__device__ __forceinline__ void later_passes(float2 (&x)[8], const float2 (&twr)[FftGeom<L>::NPASS][7], float2* __restrict__ xch, int tau) {
    if constexpr (K < FftGeom<L>::NPASS) {
        using P = Pass<L, K>;
#pragma unroll
        for (int gi = 0; gi < P::GROUPS; ++gi)
#pragma unroll
            for (int ri = 0; ri < P::G; ++ri)
                x[gi * P::G + ri] = xch[xpad(P::pos(tau, gi, ri))];
        P::run(x, twr[K]);
#pragma unroll
        for (int gi = 0; gi < P::GROUPS; ++gi)
#pragma unroll
            for (int ri = 0; ri < P::G; ++ri)
                xch[xpad(P::pos(tau, gi, ri))] = x[gi * P::G + ri];
        fft_sync<L>();
        later_passes<L, K + 1>(x, twr, xch, tau);
    }
}

template <int L, int K>
__device__ __forceinline__ void load_all_tw(int tau, const float2* __restrict__ tw, float2 (&twr)[FftGeom<L>::NPASS][7]) {
    if constexpr (K < FftGeom<L>::NPASS) {
        Pass<L, K>::load_tw(tau, tw, twr[K]);
        load_all_tw<L, K + 1>(tau, tw, twr);
    }
}

// u8 samples without the level table: (b - 127.5f) / 127.5f as a product with the reciprocal and one fma correction step.  The
// plan checks on the host, for all 256 values, that this reproduces the table the reference builds (rtl_airband.cpp:341-346)
// bit for bit before it selects this instantiation -- a lookup is two random LDS reads per sample in a kernel the LDS bounds.
constexpr int kSfmtU8Arith = 100;
__device__ __forceinline__ float level_u8(float n) {
    constexpr float d = 127.5f, rcp = 1.0f / 127.5f;
    const float q0 = n * rcp;
    return __builtin_fmaf(__builtin_fmaf(-q0, d, n), rcp, q0);
}

template <int SFMT>
__device__ __forceinline__ float2 fetch_sample(const unsigned char* __restrict__ span, int byte_off, const float* __restrict__ lut, float scale,
                                               float w) {
    float2 v;
    if constexpr (SFMT == kSfmtU8Arith) {
        const unsigned short b = *reinterpret_cast<const unsigned short*>(span + byte_off);
        v.x = level_u8(static_cast<float>(b & 0xff) - 127.5f) * w;
        v.y = level_u8(static_cast<float>(b >> 8) - 127.5f) * w;
    } else if constexpr (SFMT == MI_SFMT_U8 || SFMT == MI_SFMT_S8) {
        const unsigned short b = *reinterpret_cast<const unsigned short*>(span + byte_off);
        v.x = lut[b & 0xff] * w;  // rtl_airband.cpp:473-474
        v.y = lut[b >> 8] * w;
    } else if constexpr (SFMT == MI_SFMT_S16) {
        const short2 s = *reinterpret_cast<const short2*>(span + byte_off);
        v.x = scale * static_cast<float>(s.x) * w;  // rtl_airband.cpp:438-439
        v.y = scale * static_cast<float>(s.y) * w;
    } else {
        const float2 s = *reinterpret_cast<const float2*>(span + byte_off);
        v.x = scale * s.x * w;  // rtl_airband.cpp:456-457
        v.y = scale * s.y * w;
    }
    return v;
}

template <int L, int SFMT>
__global__ __launch_bounds__(FftGeom<L>::BLOCK) void k_channelize(const ChannelizeArgs a) {
    using Gm = FftGeom<L>;
    constexpr int N = Gm::N, TPF = Gm::TPF, F = Gm::F, TW = Gm::TW, XN = Gm::XN, BLOCK = Gm::BLOCK;
    constexpr int BPS2 = (SFMT == MI_SFMT_S16 ? 4 : (SFMT == MI_SFMT_F32 ? 8 : 2));  // bytes per complex sample

    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x;
    const int stream = blockIdx.y;
    const unsigned w0 = blockIdx.x * TW;
    const int nw = min(static_cast<unsigned>(TW), a.nfft - w0);

    const unsigned span_alloc = (static_cast<unsigned>(TW - 1) * a.hop_bytes + N * BPS2 + 16 + 15) & ~15u;
    unsigned char* span = lds;
    float* lut = reinterpret_cast<float*>(lds + span_alloc);
    float2* xch_all = reinterpret_cast<float2*>(lds + span_alloc + 1024);
    float* out_mag = reinterpret_cast<float*>(xch_all + F * XN);
    float2* out_iq = reinterpret_cast<float2*>(out_mag + a.nch * TW);

    // ---- HBM -> LDS: the byte span of this tile, each byte read once, 16 B per lane ----
    const unsigned char* gbase = a.iq + static_cast<size_t>(stream) * a.stream_stride;
    const long long b0 = static_cast<long long>(w0) * a.hop_bytes;
    const unsigned mis = static_cast<unsigned>(reinterpret_cast<uintptr_t>(gbase + b0) & 15);
    const unsigned need = static_cast<unsigned>(nw - 1) * a.hop_bytes + N * BPS2;
    const unsigned nchunks = (mis + need + 15) >> 4;
    for (unsigned c = tid; c < nchunks; c += BLOCK) {
        const long long off = b0 - mis + 16ll * c;
        uint4 v;
        if (off >= 0 && off + 16 <= static_cast<long long>(a.valid_bytes)) {
            v = *reinterpret_cast<const uint4*>(gbase + off);
        } else {
            unsigned char tmp[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const long long o = off + i;
                tmp[i] = (o >= 0 && o < static_cast<long long>(a.valid_bytes)) ? gbase[o] : 0;
            }
            v = *reinterpret_cast<uint4*>(tmp);
        }
        *reinterpret_cast<uint4*>(span + 16 * c) = v;
    }
    if constexpr (SFMT == MI_SFMT_U8 || SFMT == MI_SFMT_S8) {
        for (int i = tid; i < 256; i += BLOCK)
            lut[i] = a.levels[i];
    }

    // ---- per-lane constants: window coefficients of the 8 samples this lane converts, twiddles ----
    const int f = tid / TPF;
    const int tau = tid - f * TPF;
    float2* xch = xch_all + f * XN;
    const int nrev = (L > 3) ? static_cast<int>(__brev(static_cast<unsigned>(tau)) >> (32 - (L - 3))) : 0;
    int nidx[8];
    float wreg[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int rev3 = ((r & 1) << 2) | (r & 2) | ((r >> 2) & 1);
        nidx[r] = rev3 * TPF + nrev;  // natural sample index = bitrev_L(8 tau + r)
        wreg[r] = a.window[nidx[r]];
    }
    const float2* tw = reinterpret_cast<const float2*>(a.tw);
    const float2 w8 = tw[N / 8], w83 = tw[3 * N / 8];
    float2 twr[Gm::NPASS][7];
    load_all_tw<L, 1>(tau, tw, twr);
    __syncthreads();

    // ---- FFTs: F windows at a time ----
    for (int it = 0; it * F < nw; ++it) {
        const int wi = it * F + f;
        const bool active = wi < nw;
        float2 x[8];
        const int wbyte = static_cast<int>(mis) + (active ? wi : 0) * static_cast<int>(a.hop_bytes);
#pragma unroll
        for (int r = 0; r < 8; ++r)
            x[r] = fetch_sample<SFMT>(span, wbyte + nidx[r] * BPS2, lut, a.conv_scale, wreg[r]);
        pass0(x, w8, w83);
#pragma unroll
        for (int r = 0; r < 8; ++r)
            xch[xpad(tau * 8 + r)] = x[r];
        fft_sync<L>();
        later_passes<L, 1>(x, twr, xch, tau);
        // natural-order spectrum now sits in xch: pick the channel bins (rtl_airband.cpp:505-511)
        if (active) {
            for (int c = tau; c < a.nch; c += TPF) {
                const ChanParams& cp = a.cp[c];
                const uint32_t bin = a.st ? a.st[static_cast<size_t>(stream) * a.nch + c].afc_bin : cp.bin;  // dev->bins[j]
                const float2 v = xch[xpad(static_cast<int>(bin))];
                out_mag[c * TW + wi] = sqrtf(v.x * v.x + v.y * v.y);
                if (cp.iq_row >= 0)
                    out_iq[cp.iq_row * TW + wi] = v;
            }
            // AFC reads the spectrum of the last FFT before the batch trigger (rtl_airband.cpp:648-652, square() :186-192)
            if (a.afc_spec && w0 + static_cast<unsigned>(wi) == a.nfft - 1) {
                float* __restrict__ sq = a.afc_spec + static_cast<size_t>(stream) * N;
                for (int k = tau; k < N; k += TPF) {
                    const float2 v = xch[xpad(k)];
                    sq[k] = v.x * v.x + v.y * v.y;
                }
            }
        }
        fft_sync<L>();  // the slot is rewritten by the next window's pass 0
    }
    __syncthreads();  // the staged rows were written by every wave

    // ---- LDS -> HBM: contiguous rows of the planes ----
    for (int idx = tid; idx < a.nch * TW; idx += BLOCK) {
        const int c = idx / TW, i = idx - c * TW;
        if (i < nw)
            a.mag[(static_cast<size_t>(stream) * a.nch + c) * a.plane_stride + a.plane_off + w0 + i] = out_mag[idx];
    }
    if (a.xmax) {  // upper bound the time-parallel stage 2 starts its sandwich from
        for (int c = tid; c < a.nch; c += BLOCK) {
            float m = 0.0f;
            for (int i = 0; i < nw; ++i)
                m = fmaxf(m, out_mag[c * TW + i]);
            atomicMax(a.xmax + static_cast<size_t>(stream) * a.nch + c, __float_as_uint(m));
        }
    }
    for (int idx = tid; idx < a.n_iq_rows * TW; idx += BLOCK) {
        const int c = idx / TW, i = idx - c * TW;
        if (i < nw)
            a.cplx[(static_cast<size_t>(stream) * a.n_iq_rows + c) * a.plane_stride + a.plane_off + w0 + i] = out_iq[idx];
    }
}

// ---- N = 512 with the graph pruned to the picked bins (PrunePlan) ----
// A wave takes kGW windows at a time.  Pass 0 (stages 1-3) is the full one, a window per trip: the conversion x window of all
// 512 samples is needed whatever is picked.  It keeps only the residues R_3 of every block of 8.  Pass 1 (stages 4-6) then
// has 8 * m3 radix-8 work items per window instead of 64 -- (block of 64, residue in R_3) -- and pass 2 (stages 7-9) has m6:
// the items of the kGW windows are packed onto the lanes (class = lane & (classes - 1), so a lane's twiddles and output
// slots never change), each item runs the same Pass<9, K>::run butterflies on the same operands as the full kernel.
// Exchange buffers per window: b0[row ri][block hi * m3 + j] (row stride rs1), b1[row ri][j2] (rs2), b2[rank of the bin].
constexpr int kGW = 2;

template <int SFMT>
__global__ __launch_bounds__(256, 4) void k_channelize9p(const ChannelizeArgs a) {
    constexpr int L = 9, N = 512, TW = FftGeom<9>::TW, BLOCK = 256, NWAVE = 4;
    constexpr int BPS2 = (SFMT == MI_SFMT_S16 ? 4 : (SFMT == MI_SFMT_F32 ? 8 : 2));
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int stream = blockIdx.y;
    const unsigned w0 = blockIdx.x * TW;
    const int nw = min(static_cast<unsigned>(TW), a.nfft - w0);
    const PrunePlan& pp = a.prune;
    const int m3 = pp.m3, sh3 = pp.sh3, m6 = pp.m6, sh6 = pp.sh6, m9 = pp.m9, rs1 = pp.rs1, rs2 = pp.rs2;
    const int b0len = 8 * rs1, b2len = (m9 + 1) & ~1;  // float2 per window; b1 reuses b0's rows (rs2 <= rs1)
    const int b1len = b0len;
    const int wlen = kGW * (b0len + b2len);  // float2 per wave

    const unsigned span_alloc = (static_cast<unsigned>(TW - 1) * a.hop_bytes + N * BPS2 + 16 + 15) & ~15u;
    unsigned char* span = lds;
    float* lut = reinterpret_cast<float*>(lds + span_alloc);
    float2* xw = reinterpret_cast<float2*>(lds + span_alloc + 1024) + wave * wlen;
    float* out_mag = reinterpret_cast<float*>(reinterpret_cast<float2*>(lds + span_alloc + 1024) + NWAVE * wlen);
    float2* out_iq = reinterpret_cast<float2*>(out_mag + a.nch * TW);
    float2* b0 = xw;
    float2* b1 = b0;
    float2* b2 = b0 + kGW * b0len;

    // ---- HBM -> LDS: the byte span of this tile, each byte read once, 16 B per lane ----
    const unsigned char* gbase = a.iq + static_cast<size_t>(stream) * a.stream_stride;
    const long long byte0 = static_cast<long long>(w0) * a.hop_bytes;
    const unsigned mis = static_cast<unsigned>(reinterpret_cast<uintptr_t>(gbase + byte0) & 15);
    const unsigned need = static_cast<unsigned>(nw - 1) * a.hop_bytes + N * BPS2;
    const unsigned nchunks = (mis + need + 15) >> 4;
    for (unsigned c = tid; c < nchunks; c += BLOCK) {
        const long long off = byte0 - mis + 16ll * c;
        uint4 v;
        if (off >= 0 && off + 16 <= static_cast<long long>(a.valid_bytes)) {
            v = *reinterpret_cast<const uint4*>(gbase + off);
        } else {
            unsigned char tmp[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const long long o = off + i;
                tmp[i] = (o >= 0 && o < static_cast<long long>(a.valid_bytes)) ? gbase[o] : 0;
            }
            v = *reinterpret_cast<uint4*>(tmp);
        }
        *reinterpret_cast<uint4*>(span + 16 * c) = v;
    }
    if constexpr (SFMT == MI_SFMT_U8 || SFMT == MI_SFMT_S8) {
        for (int i = tid; i < 256; i += BLOCK)
            lut[i] = a.levels[i];
    }

    // ---- per-lane constants ----
    const int tau = lane;  // pass 0: lane = block of 8 at stage 3
    const int nrev = static_cast<int>(__brev(static_cast<unsigned>(tau)) >> (32 - (L - 3)));
    int nidx[8];
    float wreg[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int rev3 = ((r & 1) << 2) | (r & 2) | ((r >> 2) & 1);
        nidx[r] = rev3 * (N / 8) + nrev;  // natural sample index = bitrev_9(8 tau + r)
        wreg[r] = a.window[nidx[r]];
    }
    const float2* tw = reinterpret_cast<const float2*>(a.tw);
    const float2 w8 = tw[N / 8], w83 = tw[3 * N / 8];
    // the lane's item class in pass 1 / pass 2: 7 twiddles and 8 output slots each
    const int j1 = lane & ((1 << sh3) - 1), j2 = lane & ((1 << sh6) - 1);
    float2 t1[7], t2[7];
    unsigned o1[2], o2[2];  // 8 output slots each, a byte per slot (0xff = not needed)
    {
        const float* c1 = a.prune_t1 + static_cast<size_t>(j1) * kPruneClassWords;
        const float* c2 = a.prune_t2 + static_cast<size_t>(j2) * kPruneClassWords;
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            t1[k] = make_float2(c1[2 * k], c1[2 * k + 1]);
            t2[k] = make_float2(c2[2 * k], c2[2 * k + 1]);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            o1[k] = __float_as_uint(c1[14 + k]);
            o2[k] = __float_as_uint(c2[14 + k]);
        }
    }
    const int ipf1 = 8 << sh3;  // padded items of pass 1 per window (block hi = 0..7, class)
    const int hi1 = (lane & (ipf1 - 1)) >> sh3;
    __syncthreads();

    for (int wbase = wave * kGW; wbase < nw; wbase += NWAVE * kGW) {
        // ---- pass 0, one window per trip ----
#pragma unroll 1
        for (int g = 0; g < kGW; ++g) {
            const int wi = wbase + g;
            const int wbyte = static_cast<int>(mis) + (wi < nw ? wi : 0) * static_cast<int>(a.hop_bytes);
            float2 x[8];
#pragma unroll
            for (int r = 0; r < 8; ++r)
                x[r] = fetch_sample<SFMT>(span, wbyte + nidx[r] * BPS2, lut, a.conv_scale, wreg[r]);
            pass0(x, w8, w83);
            float2* d = b0 + g * b0len + (tau & 7) * rs1 + (tau >> 3) * m3;  // block tau = 8 hi + ri of pass 1
#pragma unroll
            for (int r = 0; r < 8; ++r)
                if (pp.rank3[r] >= 0)
                    d[pp.rank3[r]] = x[r];
        }
        fft_sync<L>();
        // ---- pass 1: kGW * ipf1 items ----
        for (int q = lane; q < kGW * ipf1; q += 64) {
            const int g = q >> (3 + sh3);
            if (j1 < m3) {
                const float2* src = b0 + g * b0len + hi1 * m3 + j1;
                float2 x[8];
#pragma unroll
                for (int ri = 0; ri < 8; ++ri)
                    x[ri] = src[ri * rs1];
                Pass<L, 1>::run(x, t1);
                fft_sync<L>();  // b1 lies over the rows just read
                float2* dst = b1 + g * b1len + hi1 * rs2;  // block hi1 of 64 = row of pass 2
#pragma unroll
                for (int ri = 0; ri < 8; ++ri) {
                    const unsigned o = (o1[ri >> 2] >> (8 * (ri & 3))) & 0xffu;
                    if (o != 0xffu)
                        dst[o] = x[ri];
                }
            }
        }
        fft_sync<L>();
        // ---- pass 2: kGW << sh6 items ----
        for (int q = lane; q < (kGW << sh6); q += 64) {
            const int g = q >> sh6;
            if (j2 < m6) {
                const float2* src = b1 + g * b1len + j2;
                float2 x[8];
#pragma unroll
                for (int ri = 0; ri < 8; ++ri)
                    x[ri] = src[ri * rs2];
                Pass<L, 2>::run(x, t2);
                float2* dst = b2 + g * b2len;
#pragma unroll
                for (int ri = 0; ri < 8; ++ri) {
                    const unsigned o = (o2[ri >> 2] >> (8 * (ri & 3))) & 0xffu;
                    if (o != 0xffu)
                        dst[o] = x[ri];
                }
            }
        }
        fft_sync<L>();
        // ---- the picked bins of the kGW windows (rtl_airband.cpp:505-511) ----
        for (int idx = lane; idx < kGW * a.nch; idx += 64) {
            const int g = idx / a.nch, c = idx - g * a.nch;
            const int wi = wbase + g;
            if (wi < nw) {
                const float2 v = b2[g * b2len + a.prune_rank[c]];
                out_mag[c * TW + wi] = sqrtf(v.x * v.x + v.y * v.y);
                const int iq_row = a.cp[c].iq_row;
                if (iq_row >= 0)
                    out_iq[iq_row * TW + wi] = v;
            }
        }
        fft_sync<L>();
    }
    __syncthreads();  // the staged rows were written by every wave

    // ---- LDS -> HBM: contiguous rows of the planes ----
    for (int idx = tid; idx < a.nch * TW; idx += BLOCK) {
        const int c = idx / TW, i = idx - c * TW;
        if (i < nw)
            a.mag[(static_cast<size_t>(stream) * a.nch + c) * a.plane_stride + a.plane_off + w0 + i] = out_mag[idx];
    }
    if (a.xmax) {
        for (int c = tid; c < a.nch; c += BLOCK) {
            float m = 0.0f;
            for (int i = 0; i < nw; ++i)
                m = fmaxf(m, out_mag[c * TW + i]);
            atomicMax(a.xmax + static_cast<size_t>(stream) * a.nch + c, __float_as_uint(m));
        }
    }
    for (int idx = tid; idx < a.n_iq_rows * TW; idx += BLOCK) {
        const int c = idx / TW, i = idx - c * TW;
        if (i < nw)
            a.cplx[(static_cast<size_t>(stream) * a.n_iq_rows + c) * a.plane_stride + a.plane_off + w0 + i] = out_iq[idx];
    }
}

template <int SFMT>
hipError_t launch_9p(const ChannelizeArgs& a, int nstreams, hipStream_t s) {
    using Gm = FftGeom<9>;
    constexpr int BPS2 = (SFMT == MI_SFMT_S16 ? 4 : (SFMT == MI_SFMT_F32 ? 8 : 2));
    const unsigned span_alloc = (static_cast<unsigned>(Gm::TW - 1) * a.hop_bytes + Gm::N * BPS2 + 16 + 15) & ~15u;
    const size_t wlen = static_cast<size_t>(kGW) * (8 * a.prune.rs1 + ((a.prune.m9 + 1) & ~1));
    const size_t lds = span_alloc + 1024 + 4 * wlen * 8 + static_cast<size_t>(a.nch) * Gm::TW * 4 + static_cast<size_t>(a.n_iq_rows) * Gm::TW * 8;
    if (lds > 160 * 1024)
        return hipErrorInvalidValue;
    auto kern = k_channelize9p<SFMT>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        if (e != hipSuccess)
            return e;
    }
    const dim3 grid((a.nfft + Gm::TW - 1) / Gm::TW, nstreams);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
    return hipGetLastError();
}

template <int L, int SFMT>
hipError_t launch_one(const ChannelizeArgs& a, int nstreams, hipStream_t s) {
    using Gm = FftGeom<L>;
    constexpr int BPS2 = (SFMT == MI_SFMT_S16 ? 4 : (SFMT == MI_SFMT_F32 ? 8 : 2));
    const unsigned span_alloc = (static_cast<unsigned>(Gm::TW - 1) * a.hop_bytes + Gm::N * BPS2 + 16 + 15) & ~15u;
    const size_t lds = span_alloc + 1024 + static_cast<size_t>(Gm::F) * Gm::XN * 8 + static_cast<size_t>(a.nch) * Gm::TW * 4 +
                       static_cast<size_t>(a.n_iq_rows) * Gm::TW * 8;
    if (lds > 160 * 1024)
        return hipErrorInvalidValue;
    auto kern = k_channelize<L, SFMT>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        if (e != hipSuccess)
            return e;
    }
    const dim3 grid((a.nfft + Gm::TW - 1) / Gm::TW, nstreams);
    hipLaunchKernelGGL(kern, grid, dim3(Gm::BLOCK), lds, s, a);
    return hipGetLastError();
}

template <int L>
hipError_t launch_fmt(const ChannelizeArgs& a, int sfmt, int nstreams, hipStream_t s) {
    switch (sfmt) {
        case MI_SFMT_U8: return a.conv_arith ? launch_one<L, kSfmtU8Arith>(a, nstreams, s) : launch_one<L, MI_SFMT_U8>(a, nstreams, s);
        case MI_SFMT_S8: return launch_one<L, MI_SFMT_S8>(a, nstreams, s);
        case MI_SFMT_S16: return launch_one<L, MI_SFMT_S16>(a, nstreams, s);
        case MI_SFMT_F32: return launch_one<L, MI_SFMT_F32>(a, nstreams, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace

hipError_t launch_channelize(const ChannelizeArgs& a, int log2n, int sfmt, int nstreams, hipStream_t s) {
    if (a.nfft == 0 || nstreams == 0)
        return hipSuccess;
    // the pruned graphs have no complete spectrum: AFC launches (they want one) take the full kernel
    if (log2n == 9 && a.l64.enabled && a.l64_chan && !a.afc_spec && !a.st)
        return launch_channelize_l64(a, sfmt, nstreams, s);
    if (log2n == 9 && a.prune.enabled && a.prune_t1 && a.prune_t2 && a.prune_rank && !a.afc_spec && !a.st) {
        switch (sfmt) {
            case MI_SFMT_U8: return a.conv_arith ? launch_9p<kSfmtU8Arith>(a, nstreams, s) : launch_9p<MI_SFMT_U8>(a, nstreams, s);
            case MI_SFMT_S8: return launch_9p<MI_SFMT_S8>(a, nstreams, s);
            case MI_SFMT_S16: return launch_9p<MI_SFMT_S16>(a, nstreams, s);
            case MI_SFMT_F32: return launch_9p<MI_SFMT_F32>(a, nstreams, s);
        }
        return hipErrorInvalidValue;
    }
    switch (log2n) {
        case 8: return launch_fmt<8>(a, sfmt, nstreams, s);
        case 9: return launch_fmt<9>(a, sfmt, nstreams, s);
        case 10: return launch_fmt<10>(a, sfmt, nstreams, s);
        case 11: return launch_fmt<11>(a, sfmt, nstreams, s);
        case 12: return launch_fmt<12>(a, sfmt, nstreams, s);
        case 13: return launch_fmt<13>(a, sfmt, nstreams, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace mi
