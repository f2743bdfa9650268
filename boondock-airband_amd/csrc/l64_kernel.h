// l64_kernel.h -- stage 1 at N = 512 with the first six DIT stages resident in the lanes.
//
// Replaces rtl_airband.cpp:424-511 (convert x window, FFT, bin pick + magnitude) and evaluates the very same radix-2 DIT
// graph as channelize.hip (any subset of the graph keeps every bit), mapped differently:
//
//   * A lane owns the 64 samples n = t + 8 m (t = lane & 7, m = 0..63) of one window; a wave works on 8 windows at a time
//     (g = lane >> 3).  Bit reversal puts sample m at local position rev6(m), and stages 1..6 of the 512-point graph then
//     never leave the lane: up to 192 butterflies in registers, no exchange, no barrier.
//   * Only the bins of the channel plan are wanted, so after stage s only the residues { bin mod 2^s } of every block are
//     live.  The masks M::n[s-1] (bit r: residue r is live after stage s) are compile-time constants: the kernel is
//     instantiated for the full graph ahead of time and compiled for a plan's own masks by hipRTC when the handle is
//     created (l64_jit.cpp) -- straight-line code with exactly the plan's butterflies, twiddles as instruction literals
//     (tw512.inc, checked against the plan's table on the host).
//   * Stages 7..9 combine the eight lanes of a window.  The live classes { bin mod 64 } go through a small LDS buffer,
//     then one lane per (window, channel) evaluates the 7 half-butterflies that lead to its bin -- a + w b with the sign
//     of the upper outputs folded into w (negating w negates t bit for bit, so a + (-t) is the graph's a - t) -- takes
//     the magnitude and drops it (and re / im for channels that need raw I/Q) into the tile's output rows.
//   * The tile's samples are converted ONCE per tile into a float span in LDS (windows overlap 3.2 x at hop 160): the
//     per-window work is one ds_read_b64 + one coefficient read + two multiplies per sample.  Rows of HOP samples are
//     padded by 16 words so that the eight windows of a wave start 16 banks apart (2 * 160 words = 0 mod 64 otherwise).
//
// Compile with -ffp-contract=off: products and sums must round separately except where fma is spelled.
#ifndef MI_L64_KERNEL_H
#define MI_L64_KERNEL_H

#include "l64_args.h"

namespace mi_l64 {

__device__ constexpr float kTw512[256][2] = {
#include "tw512.inc"
};

constexpr int kN = 512;
constexpr int kTile = 32;  // windows per workgroup: 4 waves x 8 windows
constexpr int kZRow = 80;  // bytes per (window, class) row of the exchange buffer: 8 lanes x (re, im), padded (b128 reads 20 banks apart)
constexpr int kSfmtS16 = 3, kSfmtF32 = 4;  // MI_SFMT_* (u8 = 1 and s8 = 2 go through the level table)

__host__ __device__ constexpr int rev6(int m) {
    return ((m & 1) << 5) | ((m & 2) << 3) | ((m & 4) << 1) | ((m & 8) >> 1) | ((m & 16) >> 3) | ((m & 32) >> 5);
}
__host__ __device__ constexpr int popc64(unsigned long long x) {
    int n = 0;
    for (; x; x &= x - 1)
        ++n;
    return n;
}

// one butterfly group of stage S: twiddle exponent RHO * 512 / 2^S, every block of the lane's 64 points.
// MODE bit 0: the lower outputs (residue RHO) are live, bit 1: the upper ones (residue RHO + 2^(S-1)).
template <int S, int RHO, int MODE>
__device__ __forceinline__ void bfly_group(float2 (&v)[64]) {
    constexpr int H = 1 << (S - 1), E = RHO * (kN >> S);
#pragma unroll
    for (int j = 0; j < (64 >> S); ++j) {
        const int ia = (j << S) + RHO, ib = ia + H;
        const float2 a = v[ia], b = v[ib];
        float2 t;
        if constexpr (E == 0) {
            t = b;  // w = 1
        } else if constexpr (E == kN / 4) {
            t = make_float2(b.y, -b.x);  // w = -j
        } else {
            t.x = __builtin_fmaf(-b.y, kTw512[E][1], b.x * kTw512[E][0]);
            t.y = __builtin_fmaf(b.y, kTw512[E][0], b.x * kTw512[E][1]);
        }
        if constexpr ((MODE & 1) != 0) {
            v[ia].x = a.x + t.x;
            v[ia].y = a.y + t.y;
        }
        if constexpr ((MODE & 2) != 0) {
            v[ib].x = a.x - t.x;
            v[ib].y = a.y - t.y;
        }
    }
}

template <class M, int S, int RHO>
__device__ __forceinline__ void stage_from(float2 (&v)[64]) {
    constexpr int H = 1 << (S - 1);
    if constexpr (RHO < H) {
        constexpr int MODE = static_cast<int>((M::n[S - 1] >> RHO) & 1ull) | (static_cast<int>((M::n[S - 1] >> (RHO + H)) & 1ull) << 1);
        if constexpr (MODE != 0)
            bfly_group<S, RHO, MODE>(v);
        stage_from<M, S, RHO + 1>(v);
    }
}

template <class M, int C>
__device__ __forceinline__ void put_classes(const float2 (&v)[64], unsigned char* __restrict__ zlane) {
    if constexpr (C < 64) {
        if constexpr (((M::n[5] >> C) & 1ull) != 0)
            *reinterpret_cast<float2*>(zlane + kZRow * popc64(M::n[5] & ((1ull << C) - 1ull))) = v[C];
        put_classes<M, C + 1>(v, zlane);
    }
}

// is local position q (after bit reversal) the input of any live butterfly?  With stage-1 pruning both inputs of a pair are
// always read, so every sample is needed; kept for clarity.
// a + w b, the half of a butterfly that leads to the wanted output
__device__ __forceinline__ float2 half_bfly(const float2 a, const float2 b, const float2 w) {
    const float tr = __builtin_fmaf(-b.y, w.y, b.x * w.x);
    const float ti = __builtin_fmaf(b.y, w.x, b.x * w.y);
    return make_float2(a.x + tr, a.y + ti);
}

// byte address of sample i of the tile's float span: rows of HOP samples, 16 words (mod 64) of padding between them
template <int HOP>
__device__ __forceinline__ unsigned span_addr(const unsigned i) {
    constexpr unsigned PADB = 4u * ((16u - 2u * HOP) & 63u);
    return 8u * i + PADB * (i / HOP);
}

// ---- the raw bytes of a tile: every byte of the capture is read once, two samples (4 / 8 / 16 bytes, aligned) per lane and
// trip, so that the converted samples of consecutive lanes are consecutive 16 bytes of the float span (no bank conflicts).
// kRawTrips covers a whole tile of u8 / s8 samples (2 * 2736 bytes over 256 lanes); wider formats take more passes. ----
constexpr int kRawTrips = 11;
typedef __attribute__((address_space(3))) volatile unsigned long long lds_u64;  // an LDS word read on its own (never merged)

struct TileGeom {
    const unsigned char* gbase;  // the stream's capture
    long long byte0;             // byte offset of the tile's first sample
    unsigned w0;                 // first window of the tile
    int nw;                      // windows in it
    unsigned nsamp;              // samples its windows span
    unsigned mis;                // misalignment of the first pair load (0 or one sample)
    unsigned npairs;
    int stream;
};

template <int HOP>
__device__ __forceinline__ TileGeom tile_geom(const L64Args& a, const int stream, const unsigned tile, const unsigned bps2) {
    TileGeom g;
    g.stream = stream;
    g.w0 = tile * kTile;
    g.nw = static_cast<int>(min(static_cast<unsigned>(kTile), a.nfft - g.w0));
    g.nsamp = static_cast<unsigned>(g.nw - 1) * HOP + kN;
    g.gbase = a.iq + static_cast<unsigned long long>(g.stream) * a.stream_stride;
    g.byte0 = static_cast<long long>(g.w0) * (static_cast<long long>(HOP) * bps2);
    const unsigned pairb = 2u * bps2;
    g.mis = static_cast<unsigned>(reinterpret_cast<unsigned long long>(g.gbase + g.byte0) & (pairb - 1));
    g.npairs = (g.mis + g.nsamp * bps2 + pairb - 1) / pairb;
    return g;
}

// one pair of samples (wd: its raw dwords) -> two converted samples at positions i0, i0 + 1 of the span
template <int HOP>
__device__ __forceinline__ void put_pair(const L64Args& a, unsigned char* span, const float* lut, const TileGeom& g, const unsigned c, const unsigned (&wd)[4],
                                         const int sfmt) {
    float2 s0, s1;
    if (sfmt != kSfmtS16 && sfmt != kSfmtF32) {
        s0 = make_float2(lut[wd[0] & 0xffu], lut[(wd[0] >> 8) & 0xffu]);  // rtl_airband.cpp:473-474
        s1 = make_float2(lut[(wd[0] >> 16) & 0xffu], lut[wd[0] >> 24]);
    } else if (sfmt == kSfmtS16) {
        const float scale = a.conv_scale;
        s0 = make_float2(scale * static_cast<float>(static_cast<short>(wd[0] & 0xffffu)), scale * static_cast<float>(static_cast<short>(wd[0] >> 16)));  // :438-439
        s1 = make_float2(scale * static_cast<float>(static_cast<short>(wd[1] & 0xffffu)), scale * static_cast<float>(static_cast<short>(wd[1] >> 16)));
    } else {
        const float scale = a.conv_scale;
        s0 = make_float2(scale * __uint_as_float(wd[0]), scale * __uint_as_float(wd[1]));  // :456-457
        s1 = make_float2(scale * __uint_as_float(wd[2]), scale * __uint_as_float(wd[3]));
    }
    const int shift = g.mis ? 1 : 0;  // the first pair starts one sample before the tile
    const int i0 = 2 * static_cast<int>(c) - shift;
    const unsigned ad0 = span_addr<HOP>(static_cast<unsigned>(i0 < 0 ? 0 : i0));
    if (shift == 0 && i0 + 1 < static_cast<int>(g.nsamp)) {
        // HOP is even: an aligned pair never straddles a row of the span
        *reinterpret_cast<float4*>(span + ad0) = make_float4(s0.x, s0.y, s1.x, s1.y);
    } else {
        if (i0 >= 0 && i0 < static_cast<int>(g.nsamp))
            *reinterpret_cast<float2*>(span + ad0) = s0;
        if (i0 + 1 < static_cast<int>(g.nsamp))
            *reinterpret_cast<float2*>(span + span_addr<HOP>(static_cast<unsigned>(i0 + 1))) = s1;
    }
}

// raw dwords of pair c of a tile; bytes outside the capture read as 0, like the exchange kernels
__device__ __forceinline__ void load_pair(const L64Args& a, const TileGeom& g, const unsigned c, const unsigned bps2, unsigned (&wd)[4]) {
    const unsigned pairb = 2u * bps2;
    const long long off = g.byte0 - g.mis + static_cast<long long>(pairb) * c;
    const long long valid = static_cast<long long>(a.valid_bytes);
    wd[0] = wd[1] = wd[2] = wd[3] = 0u;
    if (off >= 0 && off + pairb <= valid) {
        if (bps2 == 2u) {
            wd[0] = *reinterpret_cast<const unsigned*>(g.gbase + off);
        } else if (bps2 == 4u) {
            const uint2 q = *reinterpret_cast<const uint2*>(g.gbase + off);
            wd[0] = q.x, wd[1] = q.y;
        } else {
            const uint4 q = *reinterpret_cast<const uint4*>(g.gbase + off);
            wd[0] = q.x, wd[1] = q.y, wd[2] = q.z, wd[3] = q.w;
        }
    } else {
        for (unsigned i = 0; i < pairb; ++i) {
            const long long o = off + i;
            const unsigned byte = (o >= 0 && o < valid) ? g.gbase[o] : 0u;
            wd[i >> 2] |= byte << (8u * (i & 3u));
        }
    }
}

// A workgroup walks runs of `run_tiles` contiguous tiles of the launch (tile id = stream * ntiles + tile): consecutive tiles of a
// stream overlap in the capture and share cache lines in the output rows, which then meet in one CU's L1 / one XCD's L2.  The
// first run of a workgroup is its blockIdx, every further one is drawn from a ticket counter -- not a fixed share per
// workgroup: other kernels of the pipeline run alongside (segment passes, the core chains) and take LDS and wave slots, so only
// part of the grid is resident at a time, and with fixed shares the launch lasted as long as the workgroups that started last.
// The raw bytes of tile k+1 are requested before the FFTs of tile k start and converted after them: HBM latency hides under the
// arithmetic (byte formats: 11 dwords per lane stay in registers meanwhile).
template <int HOP, class M>
__device__ __forceinline__ void l64_body(const L64Args& a, unsigned char* lds) {
    const int tid0 = threadIdx.x;
    const int wave = tid0 >> 6, lane0 = tid0 & 63;
    const int sfmt = a.sfmt;
    const unsigned bps2 = sfmt == kSfmtS16 ? 4u : (sfmt == kSfmtF32 ? 8u : 2u);  // bytes per complex sample
    const bool bytes = bps2 == 2u;

    // this workgroup's first run of tiles [t_begin, t_end) of the launch's nstreams * ntiles (the host keeps that below 2^32)
    const unsigned ttotal = a.ntiles * a.nstreams;
    const unsigned nruns = (ttotal + a.run_tiles - 1u) / a.run_tiles;
    if (blockIdx.x >= nruns)
        return;
    unsigned t_begin = blockIdx.x * a.run_tiles;
    unsigned t_end = min(t_begin + a.run_tiles, ttotal);
    int nx_stream = static_cast<int>(t_begin / a.ntiles);          // (one division per run)
    unsigned nx_tile = t_begin - static_cast<unsigned>(nx_stream) * a.ntiles;

    unsigned char* const span = lds;
    float* const wtab = reinterpret_cast<float*>(lds + a.span_bytes);
    float* const lut = wtab + kN;
    constexpr int M6 = popc64(M::n[5]);
    constexpr int G = M6 <= 8 ? 8 : (M6 <= 16 ? 4 : (M6 <= 32 ? 2 : 1));  // windows of a wave per exchange round (host: l64_round_windows)
    // the exchange buffer of stages 7..9 lies over the span: by then every wave has its samples in registers (barrier below)
    unsigned char* const zbuf = lds + static_cast<unsigned>(wave) * (static_cast<unsigned>(G) * a.zstride);
    float* const out_mag = reinterpret_cast<float*>(lds + a.span_bytes + 4 * kN + 1024 + 16);
    float2* const out_iq = reinterpret_cast<float2*>(out_mag + a.nch * kTile);
    float2* const cplx = reinterpret_cast<float2*>(a.cplx);

    // ---- tables, per-lane constants of the combining step (lane -> channel is the same on every trip) ----
    for (int i = tid0; i < kN; i += 256)
        wtab[i] = a.window[i];
    if (bytes)
        lut[tid0] = a.levels[tid0];
    const int nbp = a.nb_pad;  // channels per window, padded to a power of two (8 .. 64)
    const int ch = lane0 & (nbp - 1);
    L64Chan cc = {0, -1, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (ch < a.nch)
        cc = a.chan[ch];
    const float2 w7 = make_float2(cc.w7x, cc.w7y), w8 = make_float2(cc.w8x, cc.w8y), w9 = make_float2(cc.w9x, cc.w9y);
    const int ntrip = (G * nbp + 63) / 64;

    // A tile is "fast" when it holds byte samples, is complete, starts on a pair boundary and lies inside the capture: its raw
    // dwords are prefetched into registers a tile ahead and converted with straight-line code.  Any other tile (the last
    // one of a stream, a capture that starts on an odd sample, the wider formats) is loaded when it is converted.
    unsigned pre[kRawTrips];
    auto is_fast = [&](const TileGeom& q) {
        return bytes && q.mis == 0u && q.nw == kTile && q.byte0 + 4ll * q.npairs <= static_cast<long long>(a.valid_bytes);
    };
    auto prefetch = [&](const TileGeom& q) {
        const unsigned* src = reinterpret_cast<const unsigned*>(q.gbase + q.byte0) + tid0;
#pragma unroll
        for (int k = 0; k < kRawTrips; ++k)
            pre[k] = (tid0 + 256u * k < q.npairs) ? src[256 * k] : 0u;
    };
    TileGeom geo = tile_geom<HOP>(a, nx_stream, nx_tile, bps2);
    bool fast = is_fast(geo);
    if (fast)
        prefetch(geo);
    __syncthreads();  // tables

    unsigned* const next_run = reinterpret_cast<unsigned*>(lut + 256);  // (16 bytes between the level table and the output rows)
    for (unsigned tt = t_begin; tt < t_end; ++tt) {
        const int nw = geo.nw;
        // the last tile of a run: draw the next one (the answer is read after the barrier below)
        if (tt + 1 == t_end && tid0 == 0)
            *next_run = gridDim.x + atomicAdd(a.ticket, 1u);
        // The lane's indices, opaque to the compiler once per tile: everything derived from them (11 store addresses of the
        // conversion, the sample addresses, the output indices) is then computed where it is used instead of being hoisted
        // out of the tile loop into ~60 registers that stay live across the register-resident FFT.
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63;
        const int g = lane >> 3, t = lane & 7;
        // ---- raw -> float span (the previous tile's FFTs are done with it: the barrier at the end of the loop body) ----
        if (fast) {
#pragma unroll
            for (int k = 0; k < kRawTrips; ++k) {
                const unsigned c = tid + 256u * k;  // pair c = samples 2c, 2c + 1: one row of the span (HOP is even)
                if (c < geo.npairs) {
                    const unsigned wd = pre[k];
                    *reinterpret_cast<float4*>(span + 16u * c + (4u * ((16u - 2u * HOP) & 63u)) * ((2u * c) / HOP)) =
                        make_float4(lut[wd & 0xffu], lut[(wd >> 8) & 0xffu], lut[(wd >> 16) & 0xffu], lut[wd >> 24]);  // rtl_airband.cpp:473-474
                }
            }
        } else {
#pragma unroll 1
            for (unsigned c = tid; c < geo.npairs; c += 256) {
                unsigned wd[4];
                load_pair(a, geo, c, bps2, wd);
                put_pair<HOP>(a, span, lut, geo, c, wd, sfmt);
            }
        }
        __syncthreads();
        // ---- request the next tile's bytes: they arrive while this tile's FFTs run ----
        const TileGeom cur = geo;
        bool more = tt + 1 < t_end;
        if (more) {
            if (++nx_tile == a.ntiles)
                nx_tile = 0, ++nx_stream;
        } else {
            const unsigned run = *reinterpret_cast<volatile unsigned*>(next_run);
            if (run < nruns) {  // the next run (one division per run)
                const unsigned nb = run * a.run_tiles;
                t_end = min(nb + a.run_tiles, ttotal);
                nx_stream = static_cast<int>(nb / a.ntiles);
                nx_tile = nb - static_cast<unsigned>(nx_stream) * a.ntiles;
                tt = nb - 1u;  // (the loop's increment makes it nb)
                more = true;
            }
        }
        if (more) {
            geo = tile_geom<HOP>(a, nx_stream, nx_tile, bps2);
            fast = is_fast(geo);
            if (fast)
                prefetch(geo);
        }

        const int wi = wave * 8 + g;          // window of this lane within the tile
        // (a tail tile may have fewer than 32 windows: lanes and whole waves past the tail redo window 0 of the tile and drop the
        // result -- no divergent control flow around the register-resident FFT)
        float2 v[64];
        {
            // ---- the lane's 64 samples x window coefficients into bit-reversed positions ----
            const unsigned wclamp = wi < nw ? static_cast<unsigned>(wi) : 0u;
            const unsigned char* ls = span + span_addr<HOP>(wclamp * HOP) + 8u * t;
            const float* lw = wtab + t;
            constexpr unsigned PADB = 4u * ((16u - 2u * HOP) & 63u);
            // Samples m and m + 32 land on the neighbouring positions rev6(m), rev6(m) + 1: stage 1 combines exactly those, so it
            // is taken as the pair arrives and only its live outputs stay in registers (32 instead of 64 points where the plan's
            // bins are all even or all odd).
            constexpr int MODE1 = static_cast<int>(M::n[0] & 3ull);
#pragma unroll
            for (int m = 0; m < 32; ++m) {
                // t + 8 m never crosses a row in the middle of the eight lanes: HOP is a multiple of 8
                // (volatile 8-byte reads: one ds_read_b64 each -- merged into ds_read2_b64 they would move half the bytes per clock)
                const unsigned long long r0 = *(const lds_u64*)(ls + 64u * m + PADB * ((8u * m) / HOP));
                const unsigned long long r1 = *(const lds_u64*)(ls + 64u * (m + 32) + PADB * ((8u * (m + 32)) / HOP));
                const float w0c = lw[8 * m], w1c = lw[8 * (m + 32)];
                const float2 x0 = make_float2(__uint_as_float(static_cast<unsigned>(r0)) * w0c, __uint_as_float(static_cast<unsigned>(r0 >> 32)) * w0c);  // :473-474
                const float2 x1 = make_float2(__uint_as_float(static_cast<unsigned>(r1)) * w1c, __uint_as_float(static_cast<unsigned>(r1 >> 32)) * w1c);
                const int p0 = rev6(m);  // even; sample m + 32 sits at p0 + 1
                if constexpr ((MODE1 & 1) != 0) {
                    v[p0].x = x0.x + x1.x;
                    v[p0].y = x0.y + x1.y;
                }
                if constexpr ((MODE1 & 2) != 0) {
                    v[p0 + 1].x = x0.x - x1.x;
                    v[p0 + 1].y = x0.y - x1.y;
                }
                // pin the sums here (an empty asm makes them opaque): otherwise the multiplies sink below the barrier to their first
                // use and all 64 raw samples + coefficients stay live across it
                if constexpr ((MODE1 & 1) != 0)
                    asm volatile("" : "+v"(v[p0].x), "+v"(v[p0].y));
                if constexpr ((MODE1 & 2) != 0)
                    asm volatile("" : "+v"(v[p0 + 1].x), "+v"(v[p0 + 1].y));
                if ((m & 7) == 7)
                    __builtin_amdgcn_sched_barrier(0);  // eight pairs in flight at a time: the loads of the next eight stay below
            }
        }
        __syncthreads();  // every wave has read its samples: the span may be overwritten by the exchange buffer
        {
            // ---- DIT stages 2..6 of the 512-point graph, in the lane (stage 1 was taken with the loads) ----
            stage_from<M, 2, 0>(v);
            stage_from<M, 3, 0>(v);
            stage_from<M, 4, 0>(v);
            stage_from<M, 5, 0>(v);
            stage_from<M, 6, 0>(v);
        }
        // ---- stages 7..9.  The live classes { bin mod 64 } of G windows of the wave at a time go through the exchange buffer
        // [window][class][t]; then one lane per (window, channel):
        //      X[bin] = ((Z0 + w7 Z4) + w8 (Z2 + w7 Z6)) + w9 ((Z1 + w7 Z5) + w8 (Z3 + w7 Z7)) ----
#pragma unroll 1
        for (int r = 0; r < 8 / G; ++r) {
            if ((g / G) == r)
                put_classes<M, 0>(v, zbuf + static_cast<unsigned>(g % G) * a.zstride + 8u * t);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            {
                for (int trip = 0; trip < ntrip; ++trip) {
                    const int gl = (lane + 64 * trip) / nbp;  // window of the round
                    const int wj = wave * 8 + r * G + gl;
                    if (ch < a.nch && gl < G && wj < nw) {
                        const float4* zp = reinterpret_cast<const float4*>(zbuf + static_cast<unsigned>(gl) * a.zstride + static_cast<unsigned>(cc.slot) * kZRow);
                        const float4 z01 = zp[0], z23 = zp[1], z45 = zp[2], z67 = zp[3];
                        const float2 a0 = half_bfly(make_float2(z01.x, z01.y), make_float2(z45.x, z45.y), w7);
                        const float2 a1 = half_bfly(make_float2(z01.z, z01.w), make_float2(z45.z, z45.w), w7);
                        const float2 a2 = half_bfly(make_float2(z23.x, z23.y), make_float2(z67.x, z67.y), w7);
                        const float2 a3 = half_bfly(make_float2(z23.z, z23.w), make_float2(z67.z, z67.w), w7);
                        const float2 b0 = half_bfly(a0, a2, w8);
                        const float2 b1 = half_bfly(a1, a3, w8);
                        const float2 x = half_bfly(b0, b1, w9);
                        out_mag[ch * kTile + wj] = sqrtf(x.x * x.x + x.y * x.y);  // rtl_airband.cpp:505-511
                        if (cc.iq_row >= 0)
                            out_iq[cc.iq_row * kTile + wj] = x;
                    }
                }
            }
            if (G < 8) {  // the next round overwrites the buffer
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
        __syncthreads();  // the staged rows were written by every wave, and every wave is done with the span

        // ---- LDS -> HBM: contiguous rows of the planes (the next trip writes the staging rows only after its own barrier) ----
        const unsigned long long row0 = static_cast<unsigned long long>(cur.stream) * a.nch;
        for (int idx = tid; idx < a.nch * kTile; idx += 256) {
            const int c = idx / kTile, i = idx - c * kTile;
            if (i < nw)
                a.mag[(row0 + c) * a.plane_stride + a.plane_off + cur.w0 + i] = out_mag[idx];
        }
        if (a.xmax) {  // upper bound the time-parallel stage 2 starts its sandwich from
            for (int c = tid; c < a.nch; c += 256) {
                float m = 0.0f;
                for (int i = 0; i < nw; ++i)
                    m = fmaxf(m, out_mag[c * kTile + i]);
                atomicMax(a.xmax + row0 + c, __float_as_uint(m));
            }
        }
        const unsigned long long zrow0 = static_cast<unsigned long long>(cur.stream) * a.n_iq_rows;
        for (int idx = tid; idx < a.n_iq_rows * kTile; idx += 256) {
            const int c = idx / kTile, i = idx - c * kTile;
            if (i < nw)
                cplx[(zrow0 + c) * a.plane_stride + a.plane_off + cur.w0 + i] = out_iq[idx];
        }
    }
}

}  // namespace mi_l64

#ifdef MI_L64_JIT
// run-time compilation for one plan: the masks and the hop arrive as macros
struct L64JitMasks {
    static constexpr unsigned long long n[6] = {L64_N1, L64_N2, L64_N3, L64_N4, L64_N5, L64_N6};
};
extern "C" __global__ __launch_bounds__(256, L64_MINWAVES) void l64_entry(const L64Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char l64_lds[];
    mi_l64::l64_body<L64_HOP, L64JitMasks>(a, l64_lds);
}
#endif

#endif
