// tp.hip -- time-parallel stage 2 for plain AM channels (no raw I/Q, no CTCSS, no notch), exact.
//
// The per-channel loop of demodulate() (rtl_airband.cpp:517-669) is a recurrence in time, so one stream
// with 8 channels exposes 8 serial chains.  This file cuts a long HBM-resident capture into time
// segments and still reproduces the serial result bit for bit.  Nothing here is approximate:
//
//  P1  k_tp_full    Squelch::pre_filter_.full_ (squelch.cpp:505) is a rounded EMA f' = fl(fl(f*0.99)+fl(x*b)),
//                   monotone in f.  Each lane runs it over its segment twice, from a lower bound (0) and an
//                   upper bound (1.0001 * max sample) of the true value, starting W1 steps early; once the
//                   two trajectories are bitwise equal they ARE the true value (sandwich).  Per 16-sample
//                   block it stores full at the block end, the block max of full, x[0] and min(x[1..15]);
//                   a block whose start was not yet coalesced is marked invalid.
//  A   k_tp_core    one wave per channel walks the blocks in order and keeps the exact
//                   (noise_floor_, moving_avg_cap_, pre_filter_.capped_, pre_filter_.full_) (squelch.cpp:477-514).
//                   Per block it first applies the noise-floor update, then one of
//                     MERGED     capped == full, capped < cap and the block max of full < cap: the cap never
//                                binds, so capped follows full: take P1's block-end value;
//                     SATURATED  after the block's first sample capped == cap and every other sample >= cap:
//                                capped stays cap (the reference's own shortcut, squelch.cpp:509-510);
//                     STEP       otherwise: the 16 samples are stepped one by one.
//                   It stores the exact core state at every segment boundary.
//      k_tp_core2   the same chain on three waves per channel (automatic squelch levels with a cap factor >= 1): wave 0 walks
//                   the noise-floor recurrence ahead under a hypothesis about its operand, wave 1 walks everything else and
//                   takes a value from wave 0 only where it can prove it is the true one, wave 2 fetches wave 0's operands.
//  B   k_tp_seg     one lane per (channel, segment of L = 512 .. 4096 steps): the complete state machine + AM AGC + audio,
//                   started TP_W steps early from the exact core state and a GUESSED state-machine/AGC state
//                   (idle CLOSED).  It records the state it had at its segment start (S), runs the segment
//                   writing audio, and records its end state (E).  Dead fields are canonicalised.  The first segments of a
//                   call start from the carried state -- or, when calls overlap (spec_head), warm up in the previous call's
//                   arrays like all others and are checked against the carried state by the scan.
//  C   k_tp_scan    per channel: segment k is accepted iff S_k equals E_{k-1} of an accepted predecessor
//                   (agcavgfast only where segment k reads it; otherwise it is passed through).  By
//                   induction from the true state at step 0 an accepted segment started from the true state,
//                   so its audio and E_k are the serial result.
//  D   k_tp_fix     a lane per unaccepted segment walks on from E_{k-1} (states only) through the following segments
//                   until its state meets their recorded S, leaving each member's true start state;
//      k_tp_redo    re-runs every member from that state, side by side (audio, record).
//      k_tp_settle  one wave per channel: scan again; while something is unaccepted, the same fix / redo by its own
//                   lanes and another scan; finally a serial re-run from the first unaccepted segment.  The
//                   result never depends on the speculation succeeding -- only the speed does.
//  E   k_tp_finish  applies the AM close-edge fades (rtl_airband.cpp:564-568) that were deferred as events,
//                   reduces axcindicate per WAVE_BATCH and writes the carried state for the next call.
//
// Compile with -ffp-contract=off.
#include <hip/hip_runtime.h>

#include "kernels.hpp"

#include <cstdlib>

namespace mi {
namespace {

enum : int { SQ_CLOSED = 0, SQ_OPENING = 1, SQ_CLOSING = 2, SQ_LSA = 3, SQ_OPEN = 4 };
constexpr int kOpenDelay = 197, kCloseDelay = 197, kLowSignalAbort = 88;
constexpr int kRecent = 1000, kFlap = 3;

// ---- the float recurrences, operation for operation (squelch.cpp:477-514) ----
__device__ __forceinline__ float ema99(const float f, const float x) {
    const float nfac = static_cast<float>(1.0 - static_cast<double>(0.99f));
    return f * 0.99f + x * nfac;
}
__device__ __forceinline__ float capped_step(const float c, const float x, const float cap) {
    const float e = ema99(c, x);
    const float m = (e < cap) ? e : cap;          // std::min(cap, e)
    return (c >= cap && x >= cap) ? cap : m;      // squelch.cpp:509-510 (branch-free: same values)
}
__device__ __forceinline__ float noise_floor_step(const float nf, const float c) {
    const float nfac = static_cast<float>(1.0 - static_cast<double>(0.97f));
    // std::min(capped, noise_floor); both are finite and > 0 here, so v_min_f32 returns the same bits as the ternary
    const float m = __builtin_fminf(c, nf);
    return nf * 0.97f + m * nfac + 1e-6f;
}
__device__ __forceinline__ float cap_of(const ChanParams& p, const float nf) {
    return p.using_manual_level ? p.manual_cap : p.cap_factor * nf;
}
__device__ __forceinline__ float level_of(const ChanParams& p, const float nf, const int recent) {
    if (p.using_manual_level)
        return p.manual_signal_level;
    return ((recent >= kFlap && p.flappy_signal_ratio < p.normal_signal_ratio) ? p.flappy_signal_ratio : p.normal_signal_ratio) * nf;
}

// =====================================================================================================
// P1: pre_filter_.full_ by sandwich, and the per-block aggregates
// =====================================================================================================
// The two bounding trajectories over samples [i0, i1) (multiples of 16) of one row.  A lane walks its own stretch of the row, so
// every load is a round trip of its own (a cache line per lane): 32 samples are requested per trip, a trip ahead of their
// use -- with four per trip and no prefetch the pass was bound by memory latency, 1024 round trips per lane (0.57 ms alone,
// 1.5 ms beside the other passes of a many-row call, on the front stream ahead of the next call's stage 1).
__device__ __forceinline__ void full_warmup(const float* __restrict__ x, uint32_t i0, const uint32_t i1, float& lo, float& hi) {
    auto step4 = [&](const float4 v) {
        lo = ema99(lo, v.x), hi = ema99(hi, v.x);
        lo = ema99(lo, v.y), hi = ema99(hi, v.y);
        lo = ema99(lo, v.z), hi = ema99(hi, v.z);
        lo = ema99(lo, v.w), hi = ema99(hi, v.w);
    };
    if (i0 + 32u <= i1) {
        float4 n[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            n[j] = *reinterpret_cast<const float4*>(x + i0 + 4 * j);
        for (; i0 + 32u <= i1; i0 += 32u) {
            float4 c[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                c[j] = n[j];
            if (i0 + 64u <= i1) {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    n[j] = *reinterpret_cast<const float4*>(x + i0 + 32u + 4 * j);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                step4(c[j]);
        }
    }
    for (; i0 < i1; i0 += 4)
        step4(*reinterpret_cast<const float4*>(x + i0));
}

__global__ __launch_bounds__(64) void k_tp_full(const TpArgs a) {
    const int lanes_per_row = (a.step1 - a.step0 + a.L - 1) / a.L;  // this chunk's lanes
    const int gid = blockIdx.x * 64 + threadIdx.x;
    if (gid >= a.nrows * lanes_per_row)
        return;
    const int r = gid / lanes_per_row, q = gid - r * lanes_per_row;
    const int row = a.rows[r];
    const float* __restrict__ x = a.mag + static_cast<size_t>(row) * a.plane_stride + kAgcExtra;
    const uint32_t t0 = a.step0 + static_cast<uint32_t>(q) * a.L;
    const uint32_t t1 = min(t0 + a.L, a.step1);
    const uint32_t tw = t0 > TP_W1 ? t0 - TP_W1 : 0;
    float lo, hi;
    if (tw == 0 && !a.prev_mag) {
        lo = hi = a.full0[r];  // the true value at the call start: exact from the first step
    } else {
        lo = 0.0f;
        // any earlier value of full_ is bounded by (a bound of) its value at the start of the call and the largest sample so far
        const float mx = fmaxf(__uint_as_float(a.xmax[row]), a.fullbound[r]);
        hi = mx * 1.0001f + 1e-30f;
        if (tw == 0 && t0 < TP_W1) {
            // The warm-up reaches back into the previous call: its last samples are still in its planes.  Nothing here
            // depends on the chain state of that call, so this kernel never waits for its core launches.
            const uint32_t nprev = TP_W1 - t0;  // a multiple of 16, like prev_n
            const float* __restrict__ xp = a.prev_mag + static_cast<size_t>(row) * a.plane_stride + kAgcExtra + (a.prev_n - nprev);
            full_warmup(xp, 0, nprev, lo, hi);
        }
    }
    full_warmup(x, tw, t0, lo, hi);  // (tw and t0 are multiples of 16)
    const size_t bbase = static_cast<size_t>(r) * a.nblk;
    float4 nb[4];  // the next block's samples, requested a block ahead
#pragma unroll
    for (int j = 0; j < 4; ++j)
        nb[j] = *reinterpret_cast<const float4*>(x + t0 + 4 * j);
    for (uint32_t i = t0; i < t1; i += 16) {
        const bool valid = (lo == hi);
        float fmax = 0.0f, xmin = 3.4e38f, x0 = 0.0f;
        float4 cb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            cb[j] = nb[j];
        if (i + 16 < t1) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                nb[j] = *reinterpret_cast<const float4*>(x + i + 16 + 4 * j);
        }
#pragma unroll
        for (int j = 0; j < 16; j += 4) {
            const float4 v = cb[j / 4];
            const float s[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                lo = ema99(lo, s[k]);
                hi = ema99(hi, s[k]);
                fmax = fmaxf(fmax, hi);
                if (j + k == 0)
                    x0 = s[k];
                else
                    xmin = fminf(xmin, s[k]);
            }
        }
        const size_t b = bbase + (i >> 4);
        a.blk_fe[b] = hi;
        a.blk_fm[b] = valid ? fmax : -1.0f;
        a.blk_x0[b] = x0;
        a.blk_xm[b] = xmin;
    }
}

// =====================================================================================================
// A: the exact core chain, one wave per channel
// =====================================================================================================
// wave-uniform lane read (v_readlane_b32: no LDS round trip)
__device__ __forceinline__ float rl(const float v, const int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ float uni(const float v) {
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}
// lane j <- lane j-1, lane 0 <- `first` (DPP wave_shr:1, all lanes take part: call it from uniform control flow only)
__device__ __forceinline__ float wave_shr1(const float v, const float first) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(first), __float_as_int(v), 0x138, 0xf, 0xf, false));
}

// ---- the noise-floor chain of a run of blocks, systolic over the lanes of one wave ----
// Lane l returns f_l(f_{l-1}(... f_first(nf))) where f_j(v) = v*0.97 + min(operand_j, v)*0.03 + 1e-6 (noise_floor_step) and
// `first` is the lowest active lane.  Call with EXEC = lanes >= first, the same `nf` on every lane and `passes` >= the
// number of lanes to settle.  The shifted source of lane `first` is invalid (lane 0: out of range; otherwise: disabled,
// and it holds nf anyway), so its DPP writes are dropped (bound_ctrl:0) and it keeps the pre-set values for input nf.
// The two s_nop cover the VALU-write -> DPP-read hazard (2 wait states), which the compiler cannot see inside asm.
#define NF_DPP " wave_shr:1 row_mask:0xf bank_mask:0xf\n"
#define NF_X4(P) P P P P
// Extra passes leave settled lanes unchanged, so the pass count is rounded up to the unrolled body (4 passes per asm block).
#define NF_PASS_MIN                             \
    "s_nop 1\n"                                 \
    "v_min_f32_dpp %1, %0, %4" NF_DPP           \
    "v_mul_f32_dpp %2, %0, %5" NF_DPP           \
    "v_mul_f32 %3, %6, %1\n"                    \
    "v_add_f32 %0, %2, %3\n"                    \
    "v_add_f32 %0, 0x358637bd, %0\n"
#define NF_PASS_SELF                            \
    "s_nop 1\n"                                 \
    "v_mul_f32_dpp %1, %0, %3" NF_DPP           \
    "v_mul_f32_dpp %2, %0, %4" NF_DPP           \
    "v_add_f32 %0, %1, %2\n"                    \
    "v_add_f32 %0, 0x358637bd, %0\n"
#define NF_PASS_SCALED                          \
    "s_nop 1\n"                                 \
    "v_mul_f32_dpp %1, %0, %4" NF_DPP           \
    "v_min_f32_dpp %1, %0, %1" NF_DPP           \
    "v_mul_f32_dpp %2, %0, %5" NF_DPP           \
    "v_mul_f32 %3, %6, %1\n"                    \
    "v_add_f32 %0, %2, %3\n"                    \
    "v_add_f32 %0, 0x358637bd, %0\n"
__device__ __forceinline__ float nf_chain_min(const float nf, const float operand, const int passes) {
    const float k97 = 0.97f, k03 = static_cast<float>(1.0 - static_cast<double>(0.97f));
    float v = nf, m, a = nf * k97, b;
    asm volatile("v_min_f32 %0, %1, %2" : "=v"(m) : "v"(operand), "v"(nf));
    int t = 0;  // a taken branch costs a lone wave about as much as four passes: sixteen per trip while they last
    for (; t + 16 <= passes; t += 16)
        asm volatile(NF_X4(NF_X4(NF_PASS_MIN)) : "+v"(v), "+v"(m), "+v"(a), "=&v"(b) : "v"(operand), "v"(k97), "v"(k03));
    for (; t < passes; t += 4)
        asm volatile(NF_X4(NF_PASS_MIN) : "+v"(v), "+v"(m), "+v"(a), "=&v"(b) : "v"(operand), "v"(k97), "v"(k03));
    return v;
}
// ... a full group in one piece: no loop, no branch
__device__ __forceinline__ float nf_chain_min64(const float nf, const float operand) {
    const float k97 = 0.97f, k03 = static_cast<float>(1.0 - static_cast<double>(0.97f));
    float v = nf, m, a = nf * k97, b;
    asm volatile("v_min_f32 %0, %1, %2" : "=v"(m) : "v"(operand), "v"(nf));
    asm volatile(NF_X4(NF_X4(NF_X4(NF_PASS_MIN))) : "+v"(v), "+v"(m), "+v"(a), "=&v"(b) : "v"(operand), "v"(k97), "v"(k03));
    return v;
}
// operand_j = the chain value itself (capped_ == cap >= noise floor): min() is the identity
__device__ __forceinline__ float nf_chain_self(const float nf, const int passes) {
    const float k97 = 0.97f, k03 = static_cast<float>(1.0 - static_cast<double>(0.97f));
    float v = nf, a = nf * k97, b = nf * k03;
    int t = 0;
    for (; t + 16 <= passes; t += 16)
        asm volatile(NF_X4(NF_X4(NF_PASS_SELF)) : "+v"(v), "+v"(a), "+v"(b) : "v"(k97), "v"(k03));
    for (; t < passes; t += 4)
        asm volatile(NF_X4(NF_PASS_SELF) : "+v"(v), "+v"(a), "+v"(b) : "v"(k97), "v"(k03));
    return v;
}
// ... a full group of the self chain in one piece
__device__ __forceinline__ float nf_chain_self64(const float nf) {
    const float k97 = 0.97f, k03 = static_cast<float>(1.0 - static_cast<double>(0.97f));
    float v = nf, a = nf * k97, b = nf * k03;
    asm volatile(NF_X4(NF_X4(NF_X4(NF_PASS_SELF))) : "+v"(v), "+v"(a), "+v"(b) : "v"(k97), "v"(k03));
    return v;
}
// operand_j = scale * chain value with scale < 1 (an SNR threshold below 0 dB)
__device__ __forceinline__ float nf_chain_scaled(const float nf, const float scale, const int passes) {
    const float k97 = 0.97f, k03 = static_cast<float>(1.0 - static_cast<double>(0.97f));
    float v = nf, a = nf * k97, m = nf * scale, b;
    asm volatile("v_min_f32 %0, %0, %1" : "+v"(m) : "v"(nf));
    int t = 0;
    for (; t + 16 <= passes; t += 16)
        asm volatile(NF_X4(NF_X4(NF_PASS_SCALED)) : "+v"(v), "+v"(m), "+v"(a), "=&v"(b) : "v"(scale), "v"(k97), "v"(k03));
    for (; t < passes; t += 4)
        asm volatile(NF_X4(NF_PASS_SCALED) : "+v"(v), "+v"(m), "+v"(a), "=&v"(b) : "v"(scale), "v"(k97), "v"(k03));
    return v;
}

// ---- the noise-floor chain of a full group, guess and verify (k_tp_core2, wave 0) ----
// In a quiet or a saturated stretch almost every step is the SELF step nf' = fl(fl(fl(nf*0.97) + fl(nf*0.03)) + 1e-6), and because 0.97f + (1 - 0.97f)
// is exactly 1 that is "the bit pattern of nf plus a constant" (67 in [0.125, 0.25): round(1e-6 / ulp)) except where the first product lies
// within 1/64 ulp of a tie -- then it is one less or one more.  The pattern of those exceptions is a slow rotation (the product's fraction
// moves by -0.01 per step and is kicked by +0.03 by every exception): none for ~70 steps, then one every third step for hundreds
// (tests/studies/nf_chain_study.py).  So the lanes GUESS: with the floor entering block kk known, lane j takes
//     g_j = nf_kk + h0 + h1 + h2 + h0 + ... (j - kk terms; on the bit patterns; h = the increments of the last three settled blocks)
// as the floor entering its block and computes the TRUE step o_j = F_j(g_j) from it with its own operand.  Lane kk's input is exact; if
// o_{j-1} == g_j for kk < j <= l, the inputs of lanes kk .. l are exact by induction, hence so are o_kk .. o_l -- including o_l, whose
// successor guessed wrong.  A round therefore settles the run up to and including the first lane whose result is not the next guess
// (a step below the floor, an exception the history did not predict), and the next round starts behind it.  Nothing is approximate: a
// value is used only after every step before it has been taken for real from a proven input.  About 3 rounds of ~30 instructions per
// 64 blocks on the bench signal instead of 64 passes of 6; a group whose rounds stay short is finished by the systolic passes.
struct NfGuess {
    uint32_t dh;  // per lane: the last regular increment (inc-1 .. inc+1 on the bit pattern) seen three blocks before, six, ... (0 = none)
};
__device__ __forceinline__ uint32_t wave_shr1_u32(const uint32_t v, const uint32_t first) {
    return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(first), static_cast<int>(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ uint32_t wave_shl1_u32(const uint32_t v, const uint32_t last) {
    return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(last), static_cast<int>(v), 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ uint32_t mul24(const uint32_t x, const uint32_t y) {
    return __umul24(x, y);
}
// Returns per lane the noise floor after the lane's block (all 64 lanes), leaves the floor after block 63 in nf.
__device__ __forceinline__ float nf_chain_guess64(float& nf, const float op, NfGuess& gs, const int lane, int& n_rounds) {
    uint32_t nfb = __builtin_amdgcn_readfirstlane(__float_as_uint(nf));
    const uint32_t inc = __builtin_amdgcn_readfirstlane(__float_as_uint(__uint_as_float(nfb) + 1e-6f)) - nfb;
    float vnf = nf;
    // a lane without a regular increment yet (or with one from another binade) expects the plain self step
    uint32_t dh = (gs.dh + 1u - inc <= 2u) ? gs.dh : inc;
    // the predictions of the first round: the increments of the previous group's last three blocks (lanes 60 .. 62 after the shift
    // at its end)
    uint32_t h0 = __builtin_amdgcn_readlane(dh, 60), h1 = __builtin_amdgcn_readlane(dh, 61), h2 = __builtin_amdgcn_readlane(dh, 62);
    uint32_t kk = 0;
    int rounds = 0;
    for (;;) {
        // lane kk + n: the sum of the n predicted increments h0, h1, h2, h0, ...
        const uint32_t n = static_cast<uint32_t>(lane) - kk;
        const uint32_t q = mul24(n, 43u) >> 7;  // n / 3 for n < 128 (lanes below kk: anything)
        const uint32_t r = n - mul24(q, 3u);
        const uint32_t part = mul24(min(r, 1u), h0) + mul24(r >> 1, h1);  // r == 0: 0, r == 1: h0, r == 2: h0 + h1
        const uint32_t g = nfb + mul24(q, h0 + h1 + h2) + part;
        const float o = noise_floor_step(__uint_as_float(g), op);
        const uint32_t ob = __float_as_uint(o);
        const uint32_t oprev = wave_shr1_u32(ob, 0u);
        // bit i: lane i + 1 exists, lies behind kk and its guess is not the true result of the lane before
        const unsigned long long bad = ((__ballot(oprev != g) >> 1) | (1ull << 63)) & (~0ull << kk);
        const uint32_t last = static_cast<uint32_t>(__ffsll(static_cast<long long>(bad)) - 1);  // lanes kk .. last had a proven input
        const bool settled = n <= last - kk;  // (unsigned: lanes below kk are far above)
        // lanes whose step is (by their guess) one below the floor: a run of them behind `last` is walked by systolic passes below
        const unsigned long long dips = __ballot(op < __uint_as_float(g));
        if (settled)
            vnf = o;
        const uint32_t d = ob - g;
        if (settled && d + 1u - inc <= 2u)
            dh = d;
        nfb = __builtin_amdgcn_readlane(ob, last);
        kk = last + 1u;
        ++rounds;
        if (kk >= 64u)
            break;
        // Steps below the floor come in runs (full_ stays under the floor for a few blocks) and each one is a misprediction of its
        // own: L of them cost L rounds, or L passes of 6 instructions.  (The guesses judge them from above -- a lane that is not
        // below the floor after all is walked correctly just the same.)  After eight rounds the rest of the group goes this way.
        const unsigned long long behind = ~(dips >> kk);
        uint32_t run = (behind == 0ull) ? 64u - kk : static_cast<uint32_t>(__ffsll(static_cast<long long>(behind)) - 1);
        if (__builtin_expect(rounds >= 8, 0))
            run = 64u - kk;
        if (run != 0u) {
            run = min(run, 64u - kk);
            float w = vnf;
            if (static_cast<uint32_t>(lane) >= kk)
                w = nf_chain_min(__uint_as_float(nfb), op, static_cast<int>(run));
            if (static_cast<uint32_t>(lane) - kk < run)  // lanes kk .. kk + run - 1
                vnf = w;
            kk += run;
            nfb = __builtin_amdgcn_readlane(__float_as_uint(vnf), kk - 1u);
            if (kk >= 64u)
                break;
            h0 = h1 = h2 = inc;  // (behind a dip the pattern starts afresh)
            continue;
        }
        // the increments of the last three settled blocks (a lane that took a step below the floor keeps what it had before)
        h0 = __builtin_amdgcn_readlane(dh, last + 62u), h1 = __builtin_amdgcn_readlane(dh, last + 63u), h2 = __builtin_amdgcn_readlane(dh, last);
    }
    n_rounds += rounds;
    // the next group's lane j is block 64 + j; 63 is a multiple of three, so lane j + 1 of this group is the same class
    gs.dh = wave_shl1_u32(dh, inc);
    nf = __uint_as_float(nfb);
    return vnf;
}

// ---- the same walk with the round written for a lone wave (round 3, tools/micro/lane_cost.hip, tools/round_cost.hip) ----
// What a round costs is not its instruction count but its dependent chain: 3.5 ns per dependent vector instruction, 8 ns for every value
// that goes from the vector unit through a scalar instruction and back, 10 ns per taken branch -- the round above is 140-170 ns, its
// bookkeeping around it as much again.  Here:
//   * groups of 63 blocks, so that a lane's class (block mod 3) never changes and the pattern vectors live across groups: P (the predicted
//     increments of the lanes before this one), P1 = P + the lane's own, pinc = the lane's own -- recomputed only when a class learns a new
//     increment or a run below the floor resets them;
//   * the guess is one add (base + P), the check is made by the PRODUCING lane (result - guess == pinc: no DPP, no shift of the ballot),
//     EXEC = the lanes behind kk (the compare returns 0 for the settled ones, the settled values are kept by a plain move), lane 62's pinc is
//     a value no step can produce, so the group end needs no extra bit;
//   * base of the next round (result of lane `last` - P of lane last + 1) is read with the same lane select as everything else: one
//     scalar round trip per round (compare -> find first -> readlane), 34 ns for a round that ends the group, 60 ns for one that ends early
//     and goes on inside the assembly block (a lone step below the floor; a lone exception, which is NOT learnt: it is gone three
//     blocks later, and learning and unlearning it cost two trips through the pattern code against one 60-ns round);
//   * the second exception of a group (the start or the end of a train: its class learns the increment) and a run below the floor
//     leave the assembly block for the C++ around it.
// Soundness is the argument of nf_chain_guess64: lane kk's input is exact, and result_j - guess_j == pinc_j means result_j == guess_{j+1}
// because guess_{j+1} - guess_j = P_{j+1} - P_j = pinc_j by construction (exact in 32 bits: increments < 2^22, 21 triples at most).
struct NfGuessL {           // per lane (loop-carried scalars end up in vector registers anyway)
    uint32_t P, P1, pinc;   // see above
    uint32_t key;           // the same on every lane: (nf bits >> 23) << 22 | the plain self step's increment in that binade; ~0: nothing valid
};
struct NfLaneConst {
    uint32_t q, a, b, cls;  // lane / 3, class >= 1, class == 2, class
};
__device__ __forceinline__ NfLaneConst nf_lane_const(const int lane) {
    NfLaneConst lc;
    lc.q = static_cast<uint32_t>(lane) / 3u;
    lc.cls = static_cast<uint32_t>(lane) - 3u * lc.q;
    lc.a = lc.cls >= 1u ? 1u : 0u;
    lc.b = lc.cls == 2u ? 1u : 0u;
    return lc;
}
__device__ __forceinline__ uint32_t sel_mask(const uint32_t a, const uint32_t b, const unsigned long long mask) {  // lane in mask ? b : a
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(mask));
    return r;
}
constexpr unsigned long long kGuessLast = 1ull << 62;
// the pattern vectors for the increments h0, h1, h2 of the three classes (wave-uniform)
__device__ __forceinline__ void nf_pattern(NfGuessL& gs, const NfLaneConst& lc, const uint32_t h0, const uint32_t h1, const uint32_t h2) {
    gs.P = mul24(lc.q, h0 + h1 + h2) + mul24(lc.a, h0) + mul24(lc.b, h1);
    const uint32_t own = sel_mask(sel_mask(h0, h1, 0x2492492492492492ull), h2, 0x4924924924924924ull);
    gs.P1 = gs.P + own;
    gs.pinc = sel_mask(own, 0xffffffffu, kGuessLast);  // (a step's increment is < 2^23)
}
// ... for the plain increment on every class
__device__ __forceinline__ void nf_pattern_flat(NfGuessL& gs, const int lane, const uint32_t inc) {
    gs.P = mul24(static_cast<uint32_t>(lane), inc);
    gs.P1 = gs.P + inc;
    gs.pinc = sel_mask(inc, 0xffffffffu, kGuessLast);
}
constexpr uint32_t kGuessGroup = 63;
// Returns per lane the noise floor after the lane's block (lanes 0 .. 62), leaves the floor after block 62 in nf.
__device__ __forceinline__ float nf_chain_guess63(float& nf, const float op, NfGuessL& gs, const NfLaneConst& lc, const int lane, int& n_rounds) {
    const float k97 = 0.97f, k03 = static_cast<float>(1.0 - static_cast<double>(0.97f));
    uint32_t nfb = __builtin_amdgcn_readfirstlane(__float_as_uint(nf));
    float vnf = nf;
    const uint32_t key = __builtin_amdgcn_readfirstlane(gs.key);
    uint32_t inc_ = key & 0x3fffffu;
    if (__builtin_expect((nfb >> 23) != (key >> 22), 0)) {  // another binade (or nothing valid yet): the increment and the pattern afresh
        const uint32_t inc = inc_ = __builtin_amdgcn_readfirstlane(__float_as_uint(__uint_as_float(nfb) + 1e-6f)) - nfb;
        if (inc - 1u >= (1u << 22) - 1u) {  // (nf below ~2e-6 or at the top of its binade: not worth a pattern) the passes
            vnf = nf_chain_min64(nf, op);
            nf = rl(vnf, static_cast<int>(kGuessGroup) - 1);
            gs.key = 0xffffffffu;
            n_rounds += 64;
            return vnf;
        }
        gs.key = (nfb >> 23) << 22 | inc;
        nf_pattern_flat(gs, lane, inc);
    }
    const uint32_t inc = __builtin_amdgcn_readfirstlane(inc_);  // (the compiler takes the merge of the two for divergent)
    uint32_t kk = 0, early = 0, odd = 0, base = nfb;            // (P of lane 0 is 0)
    for (;;) {
        // (temporaries in fixed registers: with more than one vector output the compiler takes every output of the block, the scalar
        //  ones included, for divergent and does the bookkeeping below in vector code)
        //  v120 guess  v121 result  v122 result - guess  v123 result - P1  v124, v125 products
        uint32_t last, dl, run, what;
        asm volatile(
            "s_lshl_b64 exec, -1, %[kk]\n"
            "1:\n"
            "v_add_u32 v120, %[base], %[P]\n"
            "v_min_f32 v125, %[op], v120\n"
            "v_mul_f32 v124, %[k97], v120\n"
            "v_mul_f32 v125, %[k03], v125\n"
            "v_add_f32 v124, v124, v125\n"
            "v_add_f32 v121, 0x358637bd, v124\n"
            "v_sub_u32 v122, v121, v120\n"
            "v_cmp_ne_u32 vcc, v122, %[pinc]\n"
            "v_mov_b32 %[vnf], v121\n"
            "v_sub_u32 v123, v121, %[P1]\n"
            "s_ff1_i32_b64 %[last], vcc\n"
            "s_add_i32 %[kk], %[last], 1\n"
            "s_lshl_b64 exec, -1, %[kk]\n"
            "s_cmp_eq_u32 %[last], 62\n"
            "s_mov_b32 %[what], 0\n"
            "v_readlane_b32 %[base], v123, %[last]\n"
            "v_readlane_b32 %[nfb], v121, %[last]\n"
            "s_cbranch_scc1 2f\n"                       // the group is done
            "v_readlane_b32 %[dl], v122, %[last]\n"
            "s_add_i32 %[early], %[early], 1\n"
            "s_sub_i32 s20, %[dl], %[incm1]\n"
            "s_cmp_le_u32 s20, 2\n"
            "s_cbranch_scc1 4f\n"                       // an exception the pattern did not have
            "v_cmp_lt_f32 vcc, %[op], v120\n"          // a step below the floor: is the block behind it one as well (judged by its guess)?
            "s_bitcmp1_b64 vcc, %[kk]\n"
            "s_cbranch_scc1 5f\n"                       // a run below the floor
            "6:\n"
            "s_cmp_ge_u32 %[early], 8\n"
            "s_cbranch_scc0 1b\n"                       // a lone step below the floor / a lone exception: the next round
            "v_cmp_lt_f32 vcc, %[op], v120\n"
            "5:\n"
            "s_lshr_b64 s[20:21], vcc, %[kk]\n"
            "s_not_b64 s[20:21], s[20:21]\n"
            "s_ff1_i32_b64 %[run], s[20:21]\n"         // (-1: every lane behind)
            "s_sub_i32 s20, 63, %[kk]\n"
            "s_min_u32 %[run], %[run], s20\n"
            "s_cmp_ge_u32 %[early], 8\n"
            "s_cselect_b32 %[run], s20, %[run]\n"      // after eight early ends the rest of the group
            "s_mov_b32 %[what], 2\n"
            "s_branch 2f\n"
            "4:\n"                                      // ... a lone one is passed like a lone step below the floor (the pattern stays as it
            "s_add_i32 %[odd], %[odd], 1\n"            //     is: learning and unlearning it costs more than the round), the second one of
            "s_cmp_ge_u32 %[odd], 2\n"                 //     a group is the start or the end of a train: the class of `last` learns it
            "s_cbranch_scc0 6b\n"
            "s_mov_b32 %[what], 1\n"
            "2:\n"
            "s_mov_b64 exec, -1\n"
            : [vnf] "+v"(vnf), [base] "+s"(base), [nfb] "+s"(nfb), [last] "=&s"(last), [dl] "=&s"(dl), [run] "=&s"(run), [what] "=&s"(what), [kk] "+s"(kk),
              [early] "+s"(early), [odd] "+s"(odd)
            : [P] "v"(gs.P), [P1] "v"(gs.P1), [pinc] "v"(gs.pinc), [op] "v"(op), [k97] "s"(k97), [k03] "s"(k03), [incm1] "s"(inc - 1u)
            : "vcc", "scc", "s20", "s21", "v120", "v121", "v122", "v123", "v124", "v125");
        if (what == 0u)
            break;
        if (what == 1u) {  // the class of `last` takes the increment seen
            const uint32_t c = __builtin_amdgcn_readlane(lc.cls, last);
            const uint32_t h0 = __builtin_amdgcn_readlane(gs.pinc, 0), h1 = __builtin_amdgcn_readlane(gs.pinc, 1), h2 = __builtin_amdgcn_readlane(gs.pinc, 2);
            nf_pattern(gs, lc, c == 0u ? dl : h0, c == 1u ? dl : h1, c == 2u ? dl : h2);
            base = nfb - __builtin_amdgcn_readlane(gs.P, kk);
            odd = 0;
            continue;
        }
        // Steps below the floor come in runs (full_ stays under the floor for a few blocks) and each one is a misprediction of its own:
        // the run behind `last` (judged by the guesses, i.e. from above -- a lane that is not below the floor after all is walked correctly
        // just the same) is walked by systolic passes, and after eight early ends the rest of the group goes that way.
        {
            float w = vnf;
            if (static_cast<uint32_t>(lane) >= kk)
                w = nf_chain_min(__uint_as_float(nfb), op, static_cast<int>(run));
            if (static_cast<uint32_t>(lane) - kk < run)  // lanes kk .. kk + run - 1
                vnf = w;
            kk += run;
            nfb = __builtin_amdgcn_readlane(__float_as_uint(vnf), kk - 1u);
            if (kk >= kGuessGroup)
                break;
            nf_pattern_flat(gs, lane, inc);  // (behind a run the pattern starts afresh)
            base = nfb - __builtin_amdgcn_readlane(gs.P, kk);
        }
    }
    n_rounds += static_cast<int>(early) + 1;
    nf = __uint_as_float(nfb);
    return vnf;
}

struct CoreGroup {  // what lane l holds for block g0 + l
    float fe, fm, x0, xm;
};
struct CoreSamples {  // the block's 16 raw samples: only read when a block of the group has to be stepped, and loaded then
    float4 s0, s1, s2, s3;
};

__device__ __forceinline__ CoreGroup core_load(const TpArgs& a, const float* __restrict__ x, const size_t bbase, const uint32_t g0, const int lane) {
    // Unconditional loads from a clamped block index: a value that is merged with a default at a join costs a register copy
    // per group and a wait at the join.  Lanes past the chunk read its last block; nobody looks at them (nb bounds every
    // use), fm = -1 only marks them for good measure.
    CoreGroup g;
    const uint32_t mine = g0 + lane;
    const uint32_t at = min(mine, a.blk1 - 1u);
    g.fe = a.blk_fe[bbase + at];
    const float fm = a.blk_fm[bbase + at];
    g.x0 = a.blk_x0[bbase + at];
    g.xm = a.blk_xm[bbase + at];
    g.fm = mine < a.blk1 ? fm : -1.0f;
    return g;
}
__device__ __forceinline__ CoreSamples core_samples(const TpArgs& a, const float* __restrict__ x, const uint32_t g0, const int lane) {
    CoreSamples g;
    const uint32_t at = min(g0 + lane, a.blk1 - 1u);
    const float4* __restrict__ sp = reinterpret_cast<const float4*>(x + static_cast<size_t>(at) * 16);
    g.s0 = sp[0], g.s1 = sp[1], g.s2 = sp[2], g.s3 = sp[3];
    return g;
}

__device__ __forceinline__ int trailing_ones_from(const unsigned long long okmask, const int kk) {
    const unsigned long long m = okmask >> kk;
    return (~m == 0ull) ? 64 - kk : __ffsll(static_cast<long long>(~m)) - 1;
}

// 16 samples of the bare EMA c' = c*0.99 + y (y = x*0.01 already formed): the value after the block and the largest value seen
__device__ __forceinline__ void ema_trial(const float (&ys)[16], const float c, float& cs, float& emax) {
    cs = c, emax = c;
#pragma unroll
    for (int j = 0; j < 16; j += 2) {
        const float c1 = cs * 0.99f + ys[j];
        cs = c1 * 0.99f + ys[j + 1];
        emax = fmaxf(fmaxf(emax, c1), cs);
    }
}

// ---- the core chain on three waves (k_tp_core2: wave 0 the passes, wave 1 everything else, wave 2 wave 0's operand loads) ----
// The noise-floor passes are 0.9 ms of the 2.0 ms a 64-s call spends in the chain; the rest is loads, verification, snapshots and
// the stepped blocks after each burst -- work that needs capped_, which only the walking wave knows.  But the passes do not need
// it: with the operand "full_ at the block's start" the recurrence noise_floor' = 0.97 nf + 0.03 min(operand, nf) + 1e-6 gives the
// true value in the merged regime (capped_ == full_), in a burst (both above the floor: min() is the floor) and in a decay unless
// full_ dips under the floor before capped_ has met it.  So wave 0 of the workgroup walks nothing but those passes, 64 blocks at a
// time, up to 12 groups ahead, and wave 1 does everything else exactly as the one-wave kernel does, taking the per-block values
// from a ring in LDS instead of computing them.  Every value taken is proven: a lane accepts its block only if min(true operand,
// nf) == min(wave 0's operand, nf) bit for bit, and after blocks wave 1 advanced by itself (single blocks, decays) it compares its
// own value with wave 0's before it takes another one.  On a disagreement wave 0 is sent back to that block with the true value
// (14-16 times per channel-minute of the gated test signal).  Every wait is bounded; if one runs out wave 1 walks the chain itself from there (`solo`).
#ifdef MI_CORE_PROF
#define CORE_PROF(...) __VA_ARGS__
static __device__ int g_prof_unsettled;  // (diagnostic build: blocks a lone fix lane took through the sample path; one lane at a time looks at it)
__device__ __forceinline__ unsigned long long prof_now() {
    return wall_clock64();  // 100 MHz
}
#else
#define CORE_PROF(...)
#endif
constexpr unsigned kNfRing = 2048;
constexpr unsigned kOpRing = 2048;
constexpr unsigned kFetchTrip = 512;
constexpr unsigned kDecRing = 256;  // entries of a decay record (a power of two)
constexpr unsigned kDecMax = 192;   // blocks a decay wave follows one decay  // blocks a fetch wave requests at a time (32 loads per lane in flight)
constexpr unsigned kShareSpin = 2u * 1000u * 1000u;
struct CoreShare {
    float nfring[kNfRing];  // noise floor after block b, at b % kNfRing
    // The block aggregates of k_tp_full, staged by the two fetch waves (waves 2 and 3) -- a lone wave that loads its own with two groups
    // in flight walks a group per 0.73 us whatever it does with it: the latency of the loads.
    float opring[kOpRing];  // full_ at the START of block b (= blk_fe of b - 1; the carried value for the chunk's first): wave 0's operand
    float fmring[kOpRing];  // blk_fm of block b (-1 past the end)
    float x0ring[kOpRing];  // blk_x0
    float xmring[kOpRing];  // blk_xm
    unsigned fetch_next[2];  // per fetch wave: first block of the trip it has not delivered yet (everything below both is there)
    // The decays after a burst (capped_ falling from the cap until it meets full_ again, ~110 blocks of 32 dependent operations), walked
    // ahead of wave 1 by the two decay waves (waves 4 and 5): per block the value entering it, the bare moving average after its 16
    // samples and the largest value on the way -- a function of the entering value and the samples alone.  Wave 1 takes an entry only
    // if its own capped_ IS the entering value, bit for bit, and judges with its own cap whether the cap binds.
    struct DecayRec {
        float cin[kDecRing], cout[kDecRing], emax[kDecRing];
        unsigned start;   // block of entry 0
        unsigned done;    // entries delivered
        unsigned active;  // the decay wave is still extending it
    } dec[2];
    unsigned w0_done;       // wave 0 has delivered every block below this one (since its last restart)
    unsigned w1_pos;        // block wave 1 is at: wave 0 stays within the ring's reach of it
    unsigned rb_seq, rb_ack, rb_blk;  // restart request of wave 1 / its acknowledgement
    float rb_nf;            // ... the noise floor entering block rb_blk
    unsigned quit;
};
// Both waves talk through LDS only, and the LDS operations of a wave execute in order: a flag written after the data is seen
// after the data, a datum read after the flag is read after it.  So the hand-offs need no fence -- a workgroup-scope fence would
// also wait for the wave's outstanding GLOBAL loads (the operands and blocks prefetched groups ahead), once per group -- only the
// compiler has to keep the order.  For the same reason every access goes through an LDS-typed pointer: through a generic one a
// volatile access is a flat_load followed by s_waitcnt vmcnt(0), which drains the prefetches just like the fence.
typedef __attribute__((address_space(3))) CoreShare LdsShare;
typedef __attribute__((address_space(3))) volatile unsigned lds_vu32;
typedef __attribute__((address_space(3))) volatile float lds_vf32;
__device__ __forceinline__ void share_order() {
    asm volatile("" ::: "memory");
}
// (wave-uniform by construction; readfirstlane lets the polling loops branch on scalars)
__device__ __forceinline__ unsigned share_peek(const __attribute__((address_space(3))) unsigned* p) {
    return __builtin_amdgcn_readfirstlane(*(const lds_vu32*)p);
}
__device__ __forceinline__ void share_post(__attribute__((address_space(3))) unsigned* p, const unsigned v) {
    *(lds_vu32*)p = v;
}
// wave 1: wait until wave 0 has answered the last restart request and delivered every block below `upto`
__device__ __forceinline__ bool share_wait(LdsShare* sh, const unsigned rb_seq_in, const uint32_t upto_in) {
    // (both wave-uniform; saying so keeps the polling loop on scalar branches)
    const unsigned rb_seq = __builtin_amdgcn_readfirstlane(rb_seq_in), upto = __builtin_amdgcn_readfirstlane(upto_in);
    for (unsigned spin = 0;; ++spin) {
        const unsigned ack = share_peek(&sh->rb_ack), done = share_peek(&sh->w0_done);
        if (ack == rb_seq && done >= upto)
            break;
        __builtin_amdgcn_s_sleep(1);
        if (spin > kShareSpin)
            return false;
    }
    share_order();
    return true;
}
// wave 1: send wave 0 back to block `blk` with the noise floor `nf` entering it
__device__ __forceinline__ bool share_rollback(LdsShare* sh, unsigned& rb_seq, const uint32_t blk, const float nf) {
    if (threadIdx.x == 64) {
        share_post(&sh->rb_blk, blk);
        *(lds_vf32*)&sh->rb_nf = nf;
    }
    share_order();
    ++rb_seq;
    if (threadIdx.x == 64)
        share_post(&sh->rb_seq, rb_seq);
    return share_wait(sh, rb_seq, blk);
}

template <bool kSplit>
__device__ __forceinline__ void core_walk(const TpArgs& a, LdsShare* sh, const int lane) {
    // The chain is the critical path of a call and shares its SIMD with waves of the wide passes of other chunks: ask the
    // issue arbiter to favour it.
    __builtin_amdgcn_s_setprio(3);
    const int r = blockIdx.x;
    const int row = a.rows[r];
    const ChanParams p = a.cp[row % a.nch];
    const float* __restrict__ x = a.mag + static_cast<size_t>(row) * a.plane_stride + kAgcExtra;
    const size_t bbase = static_cast<size_t>(r) * a.nblk;
    TpCore* __restrict__ core = a.core + static_cast<size_t>(r) * (a.nseg + 1);

    // every lane carries the same chain values (wave-uniform); lanes differ in the block they prefetched
    // the chain state comes from the previous chunk's core kernel (k_tp_prologue seeds it from the carried ChanState)
    float nf = a.core_carry[r].nf, cap = a.core_carry[r].cap, c = a.core_carry[r].c, full = a.core_carry[r].full;
    const uint32_t nblk = a.blk1;
    const uint32_t bps = a.L / 16;  // blocks per segment (a power of two: TP_L_MIN .. TP_L_MAX)
    const uint32_t bps_log = 31u - static_cast<uint32_t>(__builtin_clz(bps));

    int n_run = 0, n_single = 0, n_step = 0, n_fail = 0;  // diagnostics: blocks per path, failed hypotheses
    CORE_PROF(unsigned long long t_begin = prof_now(); unsigned long long t_wait = 0, t_stepping = 0, t_mark = 0; int n_waits = 0, n_from_rec = 0, n_reclook = 0, n_fast4 = 0, n_capb = 0, n_sysr = 0, n_sysb = 0, n_f1 = 0, n_gen = 0, n_lazy = 0; unsigned long long t_lazy = 0, t_recwait = 0, t_capb = 0, t_sysr = 0, t_f4 = 0, t_f1 = 0, t_gen = 0;)
    // kSplit: the noise-floor passes come from the chain wave (CoreShare).  `own`: this wave advanced the noise floor itself since it
    // last took a value from there (single blocks, decays): the next value taken is only good if the chain wave agrees on the one
    // before it.  `solo`: the chain wave was given up on (a wait ran out): from then on this wave walks the chain itself.
    bool own = false, solo = !kSplit;
    unsigned rb_seq = 0;
    uint32_t done_seen = a.blk0;  // blocks the chain wave is known to have delivered (forgotten when it is sent back)
    float fe_group = full;  // full_ at the start of the group: the chain wave's operand for the group's first block
    int n_rollback = 0;
    CoreGroup nxt, nxt2;
    if (!kSplit) {
        nxt = core_load(a, x, bbase, a.blk0, lane);
        nxt2 = core_load(a, x, bbase, a.blk0 + 64, lane);
    }
    uint32_t fetch_seen = a.blk0;  // kSplit: blocks whose aggregates the fetch waves are known to have delivered
    for (uint32_t g0 = a.blk0; g0 < nblk; g0 += 64) {
        CORE_PROF(const unsigned long long t_it0 = prof_now();)
        if (kSplit && !solo && !own && g0 + 256u <= nblk) {
            // ---- four groups in one regime at once.  Under the hypothesis that the regime persists every block's check depends on ring
            // values alone (the state entering a block is the noise floor the chain wave left for the block before, full_ at its
            // start, and capped_ == full_ or == the cap), so 256 blocks are judged by one pass of straight-line code and one branch:
            // the control flow around a group costs a lone wave as much as its arithmetic (tools/micro/branch_cost.hip).  Anything
            // else -- a block that fails, rings not filled that far yet -- is left to the group-by-group paths below.
            nf = uni(nf), cap = uni(cap), c = uni(c), full = uni(full);
            const bool merged = (c == full);
            if (merged || c == cap) {
                if (fetch_seen < g0 + 257u)
                    fetch_seen = min(share_peek(&sh->fetch_next[0]), share_peek(&sh->fetch_next[1]));
                if (done_seen < g0 + 256u)
                    done_seen = max(done_seen, share_peek(&sh->w0_done));
                if (fetch_seen >= g0 + 257u && done_seen >= g0 + 256u) {
                    share_order();
                    if (lane == 0)
                        share_post(&sh->w1_pos, g0);
                    bool ok = true;
                    float q_nfp[4], q_capp[4], q_ce[4], q_fe[4], vnf3 = 0.0f, fe3 = 0.0f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t b = g0 + 64u * static_cast<uint32_t>(j) + static_cast<uint32_t>(lane);
                        const bool head = j == 0 && lane == 0;  // the block this wave's own state enters
                        const float vnf = *(lds_vf32*)&sh->nfring[b & (kNfRing - 1u)];
                        const float nfp_r = *(lds_vf32*)&sh->nfring[(b - 1u) & (kNfRing - 1u)];
                        const float fep_r = *(lds_vf32*)&sh->opring[b & (kOpRing - 1u)];
                        const float fe = *(lds_vf32*)&sh->opring[(b + 1u) & (kOpRing - 1u)];
                        const float fm = *(lds_vf32*)&sh->fmring[b & (kOpRing - 1u)];
                        const float x0 = *(lds_vf32*)&sh->x0ring[b & (kOpRing - 1u)];
                        const float xm = *(lds_vf32*)&sh->xmring[b & (kOpRing - 1u)];
                        const float nf_prev = head ? nf : nfp_r;
                        const float cap_prev = head ? cap : cap_of(p, nf_prev);
                        const float capj = cap_of(p, vnf);
                        const float full_entry = head ? full : fep_r;
                        const float c_entry = merged ? full_entry : cap_prev;
                        const float opw = head ? fe_group : fep_r;
                        ok = ok && fm >= 0.0f && __builtin_fminf(c_entry, nf_prev) == __builtin_fminf(opw, nf_prev);
                        if (merged)
                            ok = ok && c_entry < capj && fm < capj;
                        else
                            ok = ok && capped_step(c_entry, x0, capj) == capj && xm >= capj;
                        q_nfp[j] = nf_prev, q_capp[j] = cap_prev, q_ce[j] = c_entry, q_fe[j] = full_entry;
                        if (j == 3)
                            vnf3 = vnf, fe3 = fe;
                    }
                    if (__ballot(ok) == ~0ull) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const uint32_t bj = g0 + 64u * static_cast<uint32_t>(j) + static_cast<uint32_t>(lane);
                            if ((bj & (bps - 1u)) == 0) {  // the state at a segment boundary
                                TpCore t;
                                t.nf = q_nfp[j], t.cap = q_capp[j], t.c = q_ce[j], t.full = q_fe[j];
                                core[bj >> bps_log] = t;
                            }
                        }
                        nf = rl(vnf3, 63);
                        cap = cap_of(p, nf);
                        full = rl(fe3, 63);
                        c = merged ? full : cap;
                        n_run += 256;
                        ++n_single;
                        CORE_PROF(++n_fast4; t_f4 += prof_now() - t_it0;)
                        fe_group = full;
                        g0 += 192u;
                        continue;
                    }
                }
            }
        }
        CoreGroup cur;
        float fe_prev;  // full_ at the start of lane's block, valid for lane > kk
        if (kSplit) {
            // the group's aggregates from the rings (entry b of opring is full_ at the start of block b, so blk_fe of b is entry b + 1)
            if (fetch_seen < g0 + 65u) {
                for (unsigned spin = 0;; ++spin) {
                    fetch_seen = min(share_peek(&sh->fetch_next[0]), share_peek(&sh->fetch_next[1]));
                    if (fetch_seen >= g0 + 65u || spin > 8u * kShareSpin)
                        break;
                    __builtin_amdgcn_s_sleep(1);
                }
                share_order();
            }
            const uint32_t slot = (g0 + lane) & (kOpRing - 1u), slot1 = (g0 + lane + 1u) & (kOpRing - 1u);
            fe_prev = *(lds_vf32*)&sh->opring[slot];
            cur.fe = *(lds_vf32*)&sh->opring[slot1];
            cur.fm = *(lds_vf32*)&sh->fmring[slot];
            cur.x0 = *(lds_vf32*)&sh->x0ring[slot];
            cur.xm = *(lds_vf32*)&sh->xmring[slot];
        } else {
            cur = nxt;
            nxt = nxt2;
            nxt2 = core_load(a, x, bbase, g0 + 128, lane);  // two groups in flight while this one is walked (past the end: the last block again)
            fe_prev = wave_shr1(cur.fe, 0.0f);
        }
        const int nb = static_cast<int>(min(64u, nblk - g0));
        const bool boundary = ((g0 + lane) & (bps - 1u)) == 0;  // (bps is a power of two)
        // A group of a quiet or a saturated stretch is settled from its aggregates alone.  The raw samples are requested when the
        // first block of the group has to be stepped -- or at its start when the group before ended in the middle of a decay.
        bool ys_ready = false, smp_ready = false;
        float yv[16];
        CoreSamples smp;
        if (c != full && c != cap) {
            smp = core_samples(a, x, g0, lane);
            smp_ready = true;
        }
        int kk = 0;
        bool skip_hyp = false;
        if (kSplit && nb == 64 && !solo && !own) {
            // ---- a whole group in one regime: the hypothesis path below with kk = 0 and all 64 lanes passing, as straight-line code
            // (that path is general and branchy: 0.55 us per group on a lone wave; this one is ~40 instructions).  Anything else --
            // a lane that fails, a group after blocks this wave stepped itself, a partial group -- is left to the general path.
            nf = uni(nf), cap = uni(cap), c = uni(c), full = uni(full);
            const bool merged = (c == full);
            if (merged || c == cap) {
                if (lane == 0)
                    share_post(&sh->w1_pos, g0);
                if (done_seen < g0 + 64u) {  // (the chain wave is usually several groups ahead: one look covers them)
                    CORE_PROF(t_mark = prof_now(); ++n_waits;)
                    for (unsigned spin = 0;; ++spin) {
                        done_seen = share_peek(&sh->w0_done);
                        if (done_seen >= g0 + 64u || spin > kShareSpin)
                            break;
                        __builtin_amdgcn_s_sleep(1);
                    }
                    share_order();
                    CORE_PROF(t_wait += prof_now() - t_mark;)
                }
                if (done_seen >= g0 + 64u) {
                    const float vnf = *(lds_vf32*)&sh->nfring[(g0 + lane) & (kNfRing - 1u)];
                    const float nf_prev = wave_shr1(vnf, nf);
                    const float cap_prev = (lane == 0) ? cap : cap_of(p, nf_prev);
                    const float capj = cap_of(p, vnf);
                    const float full_entry = (lane == 0) ? full : fe_prev;
                    const float c_entry = merged ? full_entry : cap_prev;
                    const float opw = (lane > 0) ? fe_prev : fe_group;
                    bool ok = cur.fm >= 0.0f && __builtin_fminf(c_entry, nf_prev) == __builtin_fminf(opw, nf_prev);
                    if (merged)
                        ok = ok && c_entry < capj && cur.fm < capj;
                    else
                        ok = ok && capped_step(c_entry, cur.x0, capj) == capj && cur.xm >= capj;
                    const unsigned long long okm = __ballot(ok);
                    if (okm == ~0ull) {
                        if (boundary) {
                            TpCore t;
                            t.nf = nf_prev, t.cap = cap_prev, t.c = c_entry, t.full = full_entry;
                            core[(g0 + lane) >> bps_log] = t;
                        }
                        nf = rl(vnf, 63);
                        cap = cap_of(p, nf);
                        full = rl(cur.fe, 63);
                        c = merged ? full : cap;
                        n_run += 64;
                        ++n_single;
                        fe_group = full;
                        CORE_PROF(++n_f1; t_f1 += prof_now() - t_it0;)
                        continue;
                    }
                    // The leading run of blocks that pass is what the general path below would accept from kk = 0 (the same checks on
                    // the same values), and the block behind it is one the hypothesis fails on: take the run here and send that block
                    // straight to the stepping code instead of judging the group a second time (64 regime changes per channel-minute).
                    const int nacc = trailing_ones_from(okm, 0);
                    if (nacc > 0) {
                        if (boundary && lane < nacc) {
                            TpCore t;
                            t.nf = nf_prev, t.cap = cap_prev, t.c = c_entry, t.full = full_entry;
                            core[(g0 + lane) >> bps_log] = t;
                        }
                        nf = rl(vnf, nacc - 1);
                        cap = cap_of(p, nf);
                        full = rl(cur.fe, nacc - 1);
                        c = merged ? full : cap;
                        kk = nacc;
                        n_run += nacc;
                        ++n_single;
                    }
                    ++n_fail;
                    skip_hyp = true;
                }
            }
        }
        while (kk < nb) {
            // (wave-uniform by construction; saying so lets the branches below be scalar ones)
            nf = uni(nf), cap = uni(cap), c = uni(c), full = uni(full);
            if (kSplit && lane == 0)
                share_post(&sh->w1_pos, g0 + kk);
            const bool judged = skip_hyp;  // (block kk has just failed the hypothesis in the whole-group path above)
            skip_hyp = false;
            if (!judged && (c == full || c == cap)) {
                // Hypothesis: the regime of the current block persists.  The noise-floor chain is walked serially
                // (the only true dependence), every block's precondition is then checked by its own lane.
                const bool merged = (c == full);
                // Systolic evaluation: lane j holds "noise floor after block j".  A lone wave issues one VALU instruction
                // every ~2.6 ns whether or not it depends on the previous one (tools/valu_latency.hip), so the chain is
                // priced in instructions per block, not in depth: each pass is 5 (4) VALU with the one-lane shift folded
                // into the consuming operations as a DPP modifier (nf_chain_*), no lane reads or writes.
                // After pass t lanes kk .. kk+t-1 are final; nb-kk passes settle the whole group.
                float vnf = nf;
                if (kSplit && !solo) {
                    // the chain wave has walked these blocks with min(full_ at the block's start, noise floor) as the operand
                    if (own) {  // does it agree on the value this wave computed for the block before?
                        const uint32_t at = g0 + kk;
                        if (at > a.blk0 && !(share_wait(sh, rb_seq, at) && uni(*(lds_vf32*)&sh->nfring[(at - 1u) & (kNfRing - 1u)]) == nf)) {
                            solo = !share_rollback(sh, rb_seq, at, nf) || solo;
                            done_seen = at;
                            ++n_rollback;
                        }
                        own = false;
                    }
                    CORE_PROF(t_mark = prof_now(); ++n_waits;)
                    if (!solo && !share_wait(sh, rb_seq, g0 + nb))
                        solo = true;
                    CORE_PROF(t_wait += prof_now() - t_mark;)
                    if (!solo && lane >= kk)
                        vnf = *(lds_vf32*)&sh->nfring[(g0 + lane) & (kNfRing - 1u)];
                }
                if (solo && lane >= kk) {  // lanes below kk keep the current state; lane kk never sees a valid shifted source
                    if (merged) {
                        const float cen = (lane == kk) ? c : fe_prev;  // capped_ (== full_) entering the lane's block
                        vnf = nf_chain_min(nf, cen, nb - kk);
                    } else if (p.using_manual_level) {
                        vnf = nf_chain_min(nf, p.manual_cap, nb - kk);  // capped_ == cap entering the block
                    } else if (p.cap_factor >= 1.0f) {
                        vnf = nf_chain_self(nf, nb - kk);  // min(cap_factor * nf, nf) == nf
                    } else {
                        vnf = nf_chain_scaled(nf, p.cap_factor, nb - kk);
                    }
                }
                const float nf_prev = wave_shr1(vnf, nf);  // lanes <= kk read the current state
                const float cap_prev = (lane == kk) ? cap : cap_of(p, nf_prev);
                const float capj = cap_of(p, vnf);
                const float full_entry = (lane == kk) ? full : fe_prev;
                const float c_entry = merged ? full_entry : cap_prev;
                bool ok = lane >= kk && lane < nb && cur.fm >= 0.0f;
                if (merged)
                    ok = ok && c_entry < capj && cur.fm < capj;  // MERGED: the cap never binds inside the block
                else
                    ok = ok && capped_step(c_entry, cur.x0, capj) == capj && cur.xm >= capj;  // SATURATED
                if (kSplit && !solo) {
                    // ... and the operand the chain wave used gives the same min() as the true one (it always does in the merged
                    // regime after an exact block, and in a burst as long as full_ and the cap are above the floor)
                    const float opw = (lane > 0) ? fe_prev : fe_group;
                    ok = ok && __builtin_fminf(c_entry, nf_prev) == __builtin_fminf(opw, nf_prev);
                }
                const int nacc = min(trailing_ones_from(__ballot(ok), kk), nb - kk);
                if (nacc > 0) {
                    if (boundary && lane >= kk && lane < kk + nacc) {
                        TpCore t;
                        t.nf = nf_prev, t.cap = cap_prev, t.c = c_entry, t.full = full_entry;
                        core[(g0 + lane) >> bps_log] = t;
                    }
                    const int last = kk + nacc - 1;
                    nf = rl(vnf, last);
                    cap = cap_of(p, nf);
                    full = rl(cur.fe, last);
                    c = merged ? full : cap;
                    kk += nacc;
                    n_run += nacc;
                    ++n_single;  // (diagnostic slot reused: number of accepted runs)
                    continue;
                }
                ++n_fail;
            }
            // one block, no hypothesis
            CORE_PROF(const unsigned long long t_step0 = prof_now();)
            own = true;
            const uint32_t blk = g0 + kk;
            if ((blk & (bps - 1u)) == 0 && lane == 0) {
                TpCore t;
                t.nf = nf, t.cap = cap, t.c = c, t.full = full;
                core[blk >> bps_log] = t;
            }
            float fe = rl(cur.fe, kk);
            const float fm = rl(cur.fm, kk), x0 = rl(cur.x0, kk), xm = rl(cur.xm, kk);
            // squelch.cpp:212-214: the noise floor moves on the first sample of each block, from capped_ of the previous sample
            nf = noise_floor_step(nf, c);
            cap = cap_of(p, nf);
            const bool valid = fm >= 0.0f;
            if (valid && c == full && c < cap && fm < cap) {  // MERGED
                c = fe;
                full = fe;
            } else if (valid && capped_step(c, x0, cap) == cap && xm >= cap) {  // SATURATED
                c = cap;
                full = fe;
            } else {  // STEP: the 16 samples one by one
                ++n_step;
                if (!smp_ready) {
                    CORE_PROF(const unsigned long long t_l0 = prof_now();)
                    smp = core_samples(a, x, g0, lane);
                    smp_ready = true;
                    CORE_PROF(asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); t_lazy += prof_now() - t_l0; ++n_lazy;)
                }
                if (valid) {  // full_ at the block end is already known exactly: only the capped_ chain is serial
                    // Everything that does not depend on c is done once per group, by all 64 lanes for their own blocks
                    // (y = x*b); a stepped block then costs 16 lane reads and the chain.
                    if (!ys_ready) {
                        const float nfac = static_cast<float>(1.0 - static_cast<double>(0.99f));
                        const float xv[16] = {smp.s0.x, smp.s0.y, smp.s0.z, smp.s0.w, smp.s1.x, smp.s1.y, smp.s1.z, smp.s1.w,
                                              smp.s2.x, smp.s2.y, smp.s2.z, smp.s2.w, smp.s3.x, smp.s3.y, smp.s3.z, smp.s3.w};
#pragma unroll
                        for (int j = 0; j < 16; ++j)
                            yv[j] = xv[j] * nfac;
                        ys_ready = true;
                    }
                    // Trial: the cap does not bind anywhere in the block (the decay after a burst).  Then capped_ is the
                    // bare EMA, two operations per sample; accepted iff every value stayed below the cap.  Every lane runs
                    // it on its own block from the common c; the stepped block's lane holds the answer (2 lane reads, not 16).
                    float cs_l, emax_l;
                    ema_trial(yv, c, cs_l, emax_l);
                    float cs = rl(cs_l, kk), emax = rl(emax_l, kk);
                    if (emax < cap) {
                        c = cs;
                        // Decay run: while capped_ has not met full_ again the following blocks have to be stepped too.
                        // Stay here for them (noise floor step, cap, trial) instead of going round the hypothesis logic;
                        // a block whose trial fails is left to the general path, which is exact for any block.
                        float fe_k = fe;
                        // The blocks after it, eight per round, systolic over the lanes at block granularity: lane j steps its
                        // own block (noise floor, cap, trial) from what lane j-1 left, so lane kk+t is final after t rounds of
                        // the same straight-line code -- no lane reads, no compare -> branch round trips between the blocks
                        // (there were five per block).  Lanes up to kk are switched off: lane kk+1 finds no valid shifted
                        // source and takes the current state instead.  Every lane judges its own block: it exists, its full_
                        // is valid, the cap does not bind in it (trial maximum below the block's cap), and capped_ has not met
                        // full_ at its start (the hypothesis path is cheaper from there).  The leading run of acceptable
                        // blocks is committed; a block that is not is left to the general path, which is exact for any block.
                        float nfL = nf, capL = cap, cL = c;  // state after the lane's block
                        // kSplit: the chain wave has the noise floor of these blocks already, and it is the true one wherever
                        // capped_ stays above the floor (min() takes the floor whichever operand it was given) -- which the
                        // lanes check as on the hypothesis path.  Then a round walks nothing but the EMA.  Taken only if the
                        // chain wave agrees on the value this wave just computed for block kk.
                        bool ring_ok = false;
                        float ringv = 0.0f;
                        if (kSplit && !solo && share_wait(sh, rb_seq, g0 + nb)) {
                            ringv = *(lds_vf32*)&sh->nfring[(g0 + lane) & (kNfRing - 1u)];
                            ring_ok = rl(ringv, kk) == nf;
                        }
                        while (kk + 1 < nb && c != fe_k) {
                            constexpr int kRound = 16;
                            bool ok = false;
                            float nf_in = nf, cap_in = cap, c_in = c;
                            if (kSplit && ring_ok) {
                                // ---- has a decay wave walked these blocks from exactly this capped_? ----
                                const uint32_t b1 = g0 + static_cast<uint32_t>(kk) + 1u;
                                int hit = -1;
                                uint32_t rstart = 0, rdone = 0;
                                CORE_PROF(const unsigned long long t_rw = prof_now();)
                                for (unsigned spin = 0;; ++spin) {
                                    bool coming = false;
#pragma unroll
                                    for (int d = 0; d < 2; ++d) {
                                        const uint32_t st = share_peek(&sh->dec[d].start), dn = share_peek(&sh->dec[d].done), ac = share_peek(&sh->dec[d].active);
                                        if (b1 - st < dn)
                                            hit = d, rstart = st, rdone = dn;
                                        else if (ac && b1 - st < kDecMax)
                                            coming = true;  // it is on its way here: cheaper to wait than to walk the same chain
                                    }
                                    if (hit >= 0 || !coming || spin > 20000u)
                                        break;
                                    __builtin_amdgcn_s_sleep(2);
                                }
                                CORE_PROF(t_recwait += prof_now() - t_rw; ++n_reclook;)
                                if (hit >= 0) {
                                    share_order();
                                    const uint32_t idx = g0 + static_cast<uint32_t>(lane) - rstart;
                                    const bool have = lane > kk && idx < rdone;
                                    const uint32_t slot = idx & (kDecRing - 1u);
                                    const float r_in = *(lds_vf32*)&sh->dec[hit].cin[slot], r_out = *(lds_vf32*)&sh->dec[hit].cout[slot],
                                                r_max = *(lds_vf32*)&sh->dec[hit].emax[slot];
                                    // the record is one chain; its link to this wave's state is the value entering block kk + 1
                                    // (the shift outside any lane-dependent choice: a lane switched off for it reads as `c` to its neighbour)
                                    const float r_out_below = wave_shr1(r_out, c);
                                    const float want_in = (lane == kk + 1) ? c : r_out_below;
                                    nf_in = wave_shr1(ringv, nf);
                                    cap_in = cap_of(p, nf_in);
                                    nfL = ringv, capL = cap_of(p, ringv);
                                    const bool okr = have && __float_as_uint(r_in) == __float_as_uint(want_in) && cur.fm >= 0.0f && r_max < capL && r_in != fe_prev &&
                                                     __builtin_fminf(r_in, nf_in) == __builtin_fminf(fe_prev, nf_in);
                                    const int nrec = min(trailing_ones_from(__ballot(okr), kk + 1), nb - 1 - kk);
                                    if (nrec > 0) {
                                        if (boundary && lane > kk && lane <= kk + nrec) {
                                            TpCore t;  // the state at the start of the lane's block
                                            t.nf = nf_in, t.cap = cap_in, t.c = r_in, t.full = fe_prev;
                                            core[(g0 + lane) >> bps_log] = t;
                                        }
                                        kk += nrec;
                                        nf = rl(nfL, kk);
                                        cap = rl(capL, kk);
                                        c = rl(r_out, kk);
                                        cL = r_out;
                                        fe_k = rl(cur.fe, kk);
                                        n_step += nrec;
                                        CORE_PROF(n_from_rec += nrec;)
                                        continue;
                                    }
                                }
                            }
                            CORE_PROF(++n_sysr; const unsigned long long t_s0 = prof_now();)
                            if (kSplit && ring_ok) {
                                float em1 = 0.0f;
                                if (lane > kk) {
#pragma unroll 1
                                    for (int t = 0; t < kRound; ++t) {
                                        c_in = wave_shr1(cL, c);
                                        float cs1;
                                        ema_trial(yv, c_in, cs1, em1);
                                        cL = cs1;
                                    }
                                }
                                nf_in = wave_shr1(ringv, nf);  // (lane kk+1: the chain wave's value for block kk, which is nf)
                                cap_in = cap_of(p, nf_in);
                                nfL = ringv, capL = cap_of(p, ringv);
                                ok = lane > kk && cur.fm >= 0.0f && em1 < capL && c_in != fe_prev &&
                                     __builtin_fminf(c_in, nf_in) == __builtin_fminf(fe_prev, nf_in);  // (lane > 0: the chain wave's operand is fe_prev)
                            } else if (lane > kk) {
#pragma unroll 1
                                for (int t = 0; t < kRound; ++t) {
                                    c_in = wave_shr1(cL, c);
                                    nf_in = wave_shr1(nfL, nf);
                                    cap_in = wave_shr1(capL, cap);
                                    const float nf1 = noise_floor_step(nf_in, c_in);
                                    const float cap1 = cap_of(p, nf1);
                                    float cs1, em1;
                                    ema_trial(yv, c_in, cs1, em1);
                                    ok = cur.fm >= 0.0f && em1 < cap1 && c_in != fe_prev;
                                    nfL = nf1, capL = cap1, cL = cs1;
                                }
                            }
                            const int nacc = min(min(trailing_ones_from(__ballot(ok), kk + 1), kRound), nb - 1 - kk);
                            CORE_PROF(t_sysr += prof_now() - t_s0; n_sysb += nacc;)
                            if (nacc == 0)
                                break;
                            if (boundary && lane > kk && lane <= kk + nacc) {
                                TpCore t;  // the state at the start of the lane's block
                                t.nf = nf_in, t.cap = cap_in, t.c = c_in, t.full = fe_prev;
                                core[(g0 + lane) >> bps_log] = t;
                            }
                            kk += nacc;
                            nf = rl(nfL, kk);
                            cap = rl(capL, kk);
                            c = rl(cL, kk);
                            fe_k = rl(cur.fe, kk);
                            n_step += nacc;
                            if (nacc < kRound)
                                break;
                        }
                        fe = fe_k;
                    } else {
                        CORE_PROF(++n_capb; const unsigned long long t_c0 = prof_now();)
                        // capped_step() with the shortcut "c >= cap && x >= cap" as "c >= t", t = cap where x >= cap, +inf elsewhere
                        const float xs[16] = {rl(smp.s0.x, kk), rl(smp.s0.y, kk), rl(smp.s0.z, kk), rl(smp.s0.w, kk), rl(smp.s1.x, kk), rl(smp.s1.y, kk),
                                              rl(smp.s1.z, kk), rl(smp.s1.w, kk), rl(smp.s2.x, kk), rl(smp.s2.y, kk), rl(smp.s2.z, kk), rl(smp.s2.w, kk),
                                              rl(smp.s3.x, kk), rl(smp.s3.y, kk), rl(smp.s3.z, kk), rl(smp.s3.w, kk)};
#pragma unroll
                        for (int j = 0; j < 16; ++j) {
                            const float t = (xs[j] >= cap) ? cap : __int_as_float(0x7f800000);
                            const float e = c * 0.99f + rl(yv[j], kk);
                            const float m = (e < cap) ? e : cap;
                            c = (c >= t) ? cap : m;
                        }
                        CORE_PROF(t_capb += prof_now() - t_c0;)
                    }
                    full = fe;
                } else {
                    const float xs[16] = {rl(smp.s0.x, kk), rl(smp.s0.y, kk), rl(smp.s0.z, kk), rl(smp.s0.w, kk), rl(smp.s1.x, kk), rl(smp.s1.y, kk),
                                          rl(smp.s1.z, kk), rl(smp.s1.w, kk), rl(smp.s2.x, kk), rl(smp.s2.y, kk), rl(smp.s2.z, kk), rl(smp.s2.w, kk),
                                          rl(smp.s3.x, kk), rl(smp.s3.y, kk), rl(smp.s3.z, kk), rl(smp.s3.w, kk)};
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        full = ema99(full, xs[j]);
                        c = capped_step(c, xs[j], cap);
                    }
                }
            }
            ++kk;
            CORE_PROF(t_stepping += prof_now() - t_step0;)
        }
        fe_group = rl(cur.fe, 63);
        CORE_PROF(++n_gen; t_gen += prof_now() - t_it0;)
    }
    CORE_PROF(if (lane == 0 && r < 8) printf("core w1 row %d: groups by path: four at once %d in %llu us, whole group %d in %llu us, general %d in %llu us (stepping included); raw samples asked for when a block had to be stepped: %d times, %llu us until they were there\n", r, n_fast4, t_f4 / 100, n_f1, t_f1 / 100, n_gen, t_gen / 100, n_lazy, t_lazy / 100);)
    CORE_PROF(if (lane == 0 && r < 8) printf("core w1 row %d: total %llu us  wait-w0 %llu us (%d waits)  stepping %llu us (record waits %llu us, %d looks)  run %d accepted-runs %d step %d (from records %d) fail %d rollback %d solo %d fast4 %d; cap-binding blocks %d in %llu us, systolic rounds %d (%d blocks) in %llu us\n", r,
                                           (prof_now() - t_begin) / 100, t_wait / 100, n_waits, t_stepping / 100, t_recwait / 100, n_reclook, n_run, n_single, n_step, n_from_rec, n_fail, n_rollback, (int)solo, n_fast4, n_capb, t_capb / 100, n_sysr, n_sysb, t_sysr / 100);)
    if (kSplit && lane == 0)
        share_post(&sh->quit, 1u);
    if (lane == 0) {
        TpCore t;
        t.nf = nf, t.cap = cap, t.c = c, t.full = full;
        core[(a.blk1 + bps - 1) / bps] = t;  // = the next chunk's first boundary, or the end of the call
        a.core_carry[r] = t;
        if (a.diag) {  // counters of the call: the first chunk starts them
            int* d = a.diag + a.nrows * 4 + r * 4;
            d[0] = (a.first_chunk ? 0 : d[0]) + n_run;
            d[1] = (a.first_chunk ? 0 : d[1]) + n_single;
            d[2] = (a.first_chunk ? 0 : d[2]) + n_step;
            d[3] = (a.first_chunk ? 0 : d[3]) + n_fail + 100000 * n_rollback + (kSplit && solo ? 50000000 : 0);
        }
    }
}

__global__ __launch_bounds__(64) void k_tp_core(const TpArgs a) {
    core_walk<false>(a, nullptr, threadIdx.x);
}

// ---- waves 4 and 5 of k_tp_core2: the decays, ahead of wave 1 ----
// A decay starts where the saturated regime ends: the first block whose SATURATED test fails behind one that passed, with
// capped_ == the cap entering it.  The decay waves find those blocks in what wave 0 has delivered (the same test as wave 1's, on
// the same rings), take turns, and follow each decay for up to kDecMax blocks with wave 1's own lane-systolic rounds, leaving
// (value in, value out, largest value) per block in their record.  Nothing here is trusted: see CoreShare::DecayRec.
__device__ __forceinline__ void decay_wave(const TpArgs& a, LdsShare* sh, const unsigned w, const int lane) {
    const int r = blockIdx.x;
    const int row = a.rows[r];
    const ChanParams p = a.cp[row % a.nch];
    const float* __restrict__ x = a.mag + static_cast<size_t>(row) * a.plane_stride + kAgcExtra;
    const uint32_t nblk = a.blk1;
    const float nf_carry = a.core_carry[r].nf;
    const float nfac = static_cast<float>(1.0 - static_cast<double>(0.99f));
    uint32_t sb = a.blk0;   // the group being scanned
    bool prev_sat = false;  // the block before it passed the SATURATED test
    unsigned edges_seen = 0, rb_seen = 0, idle = 0;
    uint32_t rec_end = 0;   // wave 1 has to be past this block before the record is overwritten
    CORE_PROF(unsigned long long t_dec = 0, t_hold = 0, t_m = 0; int n_dec = 0, n_blk = 0, n_met = 0;)
    for (;;) {
        const unsigned quit = share_peek(&sh->quit), w0_done = share_peek(&sh->w0_done), rs = share_peek(&sh->rb_seq), ack = share_peek(&sh->rb_ack),
                       w1_pos = share_peek(&sh->w1_pos);
        if (quit || sb >= nblk)
            break;
        if (rs != rb_seen) {  // wave 0 was sent back: what was scanned behind that block was scanned with another floor
            if (ack != rs) {
                __builtin_amdgcn_s_sleep(1);
                continue;
            }
            rb_seen = rs;
            const uint32_t rb = share_peek(&sh->rb_blk);
            if (rb < sb) {
                sb = a.blk0 + ((rb - a.blk0) & ~63u);
                prev_sat = false;
            }
            continue;
        }
        if (w1_pos >= sb + 64u) {  // wave 1 is past this group (it walks the decays it finds no record for itself)
            sb = a.blk0 + ((w1_pos - a.blk0) & ~63u);
            prev_sat = false;
            continue;
        }
        const uint32_t n = min(64u, nblk - sb);
        if (w0_done < sb + n) {
            __builtin_amdgcn_s_sleep(1);
            if (++idle > 8u * kShareSpin)
                break;
            continue;
        }
        idle = 0;
        share_order();
        const uint32_t b = sb + static_cast<uint32_t>(lane), slot = b & (kOpRing - 1u);
        const float vnf = *(lds_vf32*)&sh->nfring[b & (kNfRing - 1u)];
        const float nf_first = (sb > a.blk0) ? *(lds_vf32*)&sh->nfring[(sb - 1u) & (kNfRing - 1u)] : nf_carry;
        const float nf_prev = wave_shr1(vnf, nf_first);
        const float cap_prev = cap_of(p, nf_prev), capj = cap_of(p, vnf);
        const float x0 = *(lds_vf32*)&sh->x0ring[slot], xm = *(lds_vf32*)&sh->xmring[slot], fm = *(lds_vf32*)&sh->fmring[slot], fe_prev = *(lds_vf32*)&sh->opring[slot];
        const bool sat = static_cast<uint32_t>(lane) < n && fm >= 0.0f && capped_step(cap_prev, x0, capj) == capj && xm >= capj &&
                         __builtin_fminf(cap_prev, nf_prev) == __builtin_fminf(fe_prev, nf_prev);
        const unsigned long long m = __ballot(sat);
        unsigned long long edges = ~m & ((m << 1) | (prev_sat ? 1ull : 0ull)) & (n == 64u ? ~0ull : ((1ull << n) - 1ull));
        prev_sat = (m >> 63) != 0ull;
        while (edges) {
            const int e = __ffsll(static_cast<long long>(edges)) - 1;
            edges &= edges - 1ull;
            if ((edges_seen++ & 1u) != w)
                continue;
            // ---- follow the decay that starts at block sb + e from capped_ == the cap entering it ----
            __attribute__((address_space(3))) CoreShare::DecayRec* rec = &sh->dec[w];
            CORE_PROF(t_m = prof_now();)
            for (unsigned spin = 0; share_peek(&sh->w1_pos) < rec_end && !share_peek(&sh->quit) && spin < 8u * kShareSpin; ++spin)
                __builtin_amdgcn_s_sleep(2);  // wave 1 may still be reading the previous record
            CORE_PROF(t_hold += prof_now() - t_m; t_m = prof_now(); ++n_dec;)
            const uint32_t e0 = sb + static_cast<uint32_t>(e);
            if (lane == 0) {
                share_post(&rec->done, 0u);
                share_post(&rec->active, 1u);
            }
            share_order();
            if (lane == 0)
                share_post(&rec->start, e0);
            float c = rl(cap_prev, e);
            uint32_t count = 0;
            bool stop = false;
            for (uint32_t base_blk = e0; !stop && count < kDecMax && base_blk < nblk; base_blk += 64u) {
                const uint32_t nbk = min(min(64u, nblk - base_blk), kDecMax - count);
                const uint32_t at = min(base_blk + static_cast<uint32_t>(lane), nblk - 1u);
                const float4* __restrict__ sp = reinterpret_cast<const float4*>(x + static_cast<size_t>(at) * 16);
                const float4 s0 = sp[0], s1 = sp[1], s2 = sp[2], s3 = sp[3];
                const float yv[16] = {s0.x * nfac, s0.y * nfac, s0.z * nfac, s0.w * nfac, s1.x * nfac, s1.y * nfac, s1.z * nfac, s1.w * nfac,
                                      s2.x * nfac, s2.y * nfac, s2.z * nfac, s2.w * nfac, s3.x * nfac, s3.y * nfac, s3.z * nfac, s3.w * nfac};
                // full_ after the lane's block, if the fetch waves have it yet (merged again: the decay is over)
                const uint32_t fetched = min(share_peek(&sh->fetch_next[0]), share_peek(&sh->fetch_next[1]));
                const float fe_blk = *(lds_vf32*)&sh->opring[(base_blk + static_cast<uint32_t>(lane) + 1u) & (kOpRing - 1u)];
                const bool fe_known = base_blk + static_cast<uint32_t>(lane) + 1u < fetched;
                float cL = c;
                uint32_t base = 0;
                // The first blocks of a decay, one at a time while the cap still binds (capped_ starts AT the cap): the trial, and where
                // it fails the capped recurrence with the cap of the floor wave 0 has for the block -- what wave 1 does with them.
                // Their entries say "not a bare average" (largest value = +inf): wave 1 steps those blocks itself and finds the
                // chain again behind them.
                while (base_blk == e0 && base < nbk && base < 8u) {
                    const uint32_t blk = base_blk + base;
                    for (unsigned spin = 0; share_peek(&sh->w0_done) <= blk && spin < kShareSpin; ++spin)
                        __builtin_amdgcn_s_sleep(1);
                    const float capb = cap_of(p, uni(*(lds_vf32*)&sh->nfring[blk & (kNfRing - 1u)]));
                    float cs_l, em_l;
                    ema_trial(yv, c, cs_l, em_l);
                    const float cs = rl(cs_l, static_cast<int>(base)), em = rl(em_l, static_cast<int>(base));
                    if (em < capb)
                        break;  // the cap no longer binds: the rounds below take over from this block
                    const int l = static_cast<int>(base);
                    const float xs[16] = {rl(s0.x, l), rl(s0.y, l), rl(s0.z, l), rl(s0.w, l), rl(s1.x, l), rl(s1.y, l), rl(s1.z, l), rl(s1.w, l),
                                          rl(s2.x, l), rl(s2.y, l), rl(s2.z, l), rl(s2.w, l), rl(s3.x, l), rl(s3.y, l), rl(s3.z, l), rl(s3.w, l)};
                    float cc = c;
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const float t = (xs[j] >= capb) ? capb : __int_as_float(0x7f800000);
                        const float ev = cc * 0.99f + rl(yv[j], l);
                        const float mv = (ev < capb) ? ev : capb;
                        cc = (cc >= t) ? capb : mv;
                    }
                    if (lane == 0) {
                        const uint32_t idx = (blk - e0) & (kDecRing - 1u);
                        *(lds_vf32*)&rec->cin[idx] = c, *(lds_vf32*)&rec->cout[idx] = cc, *(lds_vf32*)&rec->emax[idx] = __int_as_float(0x7f800000);
                    }
                    (void)cs;
                    c = cc;
                    cL = c;
                    ++base;
                    share_order();
                    if (lane == 0)
                        share_post(&rec->done, blk - e0 + 1u);
                }
                for (; base < nbk && !stop; base += 16u) {
                    float c_in = c, em1 = 0.0f;
                    if (static_cast<uint32_t>(lane) >= base) {
#pragma unroll 1
                        for (int t = 0; t < 16; ++t) {
                            c_in = wave_shr1(cL, c);
                            float cs1;
                            ema_trial(yv, c_in, cs1, em1);
                            cL = cs1;
                        }
                    }
                    const uint32_t top = min(base + 16u, nbk);
                    const bool mine = static_cast<uint32_t>(lane) >= base && static_cast<uint32_t>(lane) < top;
                    if (mine) {
                        const uint32_t idx = (base_blk - e0 + static_cast<uint32_t>(lane)) & (kDecRing - 1u);
                        *(lds_vf32*)&rec->cin[idx] = c_in, *(lds_vf32*)&rec->cout[idx] = cL, *(lds_vf32*)&rec->emax[idx] = em1;
                    }
                    share_order();
                    if (lane == 0)
                        share_post(&rec->done, base_blk - e0 + top);
                    c = rl(cL, static_cast<int>(top) - 1);
                    // over when capped_ has met full_ again, or climbed back to (about) where it started: the next burst
                    const unsigned long long met = __ballot(mine && fe_known && cL == fe_blk);
                    stop = met != 0ull || share_peek(&sh->quit) != 0u || share_peek(&sh->rb_seq) != rb_seen;
                }
                count += nbk;
            }
            rec_end = e0 + count;
            CORE_PROF(t_dec += prof_now() - t_m; n_blk += count; n_met += stop ? 1 : 0;)
            if (lane == 0)
                share_post(&rec->active, 0u);
        }
        sb += 64u;
    }
    if (lane == 0)
        share_post(&sh->dec[w].active, 0u);
    CORE_PROF(if (lane == 0 && r < 2) printf("core decay wave %u row %d: %d decays, %d blocks, %d ended early; walking %llu us, held back %llu us; scanned to %u\n", w, r, n_dec, n_blk, n_met,
                                           t_dec / 100, t_hold / 100, sb);)
}

__global__ __launch_bounds__(512) void k_tp_core2(const TpArgs a) {
    __shared__ CoreShare sh_mem;
    LdsShare* const sh = (LdsShare*)&sh_mem;
    if (threadIdx.x == 0) {
        sh->w0_done = a.blk0, sh->w1_pos = a.blk0, sh->rb_seq = 0, sh->rb_ack = 0, sh->rb_blk = a.blk0, sh->rb_nf = 0.0f, sh->quit = 0;
        sh->fetch_next[0] = a.blk0, sh->fetch_next[1] = a.blk0;
        for (int d = 0; d < 2; ++d)
            sh->dec[d].start = 0xffffffffu, sh->dec[d].done = 0, sh->dec[d].active = 0;
    }
    __syncthreads();
    if (threadIdx.x >= 256) {
        // (waves go to the SIMDs of a CU in turn: waves 4 and 5 would share theirs with waves 0 and 1, which are the critical path
        //  and never sleep; waves 6 and 7 share with the fetch waves, which mostly do)
        if (a.core_decay && threadIdx.x >= 384) {
            __builtin_amdgcn_s_setprio(2);
            decay_wave(a, sh, threadIdx.x >= 448 ? 1u : 0u, threadIdx.x & 63);
        }
        return;
    }
    if (threadIdx.x >= 64 && threadIdx.x < 128) {
        core_walk<true>(a, sh, threadIdx.x - 64);
        return;
    }
    const int r = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const size_t bbase = static_cast<size_t>(r) * a.nblk;
    const uint32_t nblk = a.blk1;
    if (threadIdx.x >= 128 && threadIdx.x < 256) {
        // ---- waves 2 and 3: the aggregates of every block into the rings, kFetchTrip blocks per trip and wave in turn, as far ahead of
        // wave 1 as the rings reach.  Entry b of opring is full_ at the start of block b, so the entries run to b = blk1 inclusive.
        const unsigned w = (threadIdx.x >= 192) ? 1u : 0u;
        const float full0 = a.core_carry[r].full;
        uint32_t f = a.blk0 + w * kFetchTrip;
        unsigned idle = 0;
        constexpr int kPer = kFetchTrip / 64;
        for (;;) {
            const unsigned quit = share_peek(&sh->quit), w1_pos = share_peek(&sh->w1_pos);
            if (quit)
                break;
            if (f > nblk) {  // nothing left for this wave
                if (lane == 0)
                    share_post(&sh->fetch_next[w], 0xffffffffu);
                break;
            }
            if (f + kFetchTrip > w1_pos + (kOpRing - 64u)) {  // the ring is full (blocks behind wave 1 are free)
                __builtin_amdgcn_s_sleep(2);
                if (++idle > 8u * kShareSpin)
                    break;
                continue;
            }
            idle = 0;
            float vo[kPer], vm[kPer], v0[kPer], vx[kPer];
#pragma unroll
            for (int i = 0; i < kPer; ++i) {
                const uint32_t idx = f + 64u * i + lane;
                const uint32_t at = min(idx, nblk - 1u);
                vo[i] = a.blk_fe[bbase + (idx > 0u ? min(idx, nblk) - 1u : 0u)];
                vm[i] = a.blk_fm[bbase + at];
                v0[i] = a.blk_x0[bbase + at];
                vx[i] = a.blk_xm[bbase + at];
            }
#pragma unroll
            for (int i = 0; i < kPer; ++i) {
                const uint32_t idx = f + 64u * i + lane;
                const uint32_t slot = idx & (kOpRing - 1u);
                *(lds_vf32*)&sh->opring[slot] = (idx == a.blk0) ? full0 : vo[i];  // the chunk's first block: the carried value
                *(lds_vf32*)&sh->fmring[slot] = (idx < nblk) ? vm[i] : -1.0f;
                *(lds_vf32*)&sh->x0ring[slot] = v0[i];
                *(lds_vf32*)&sh->xmring[slot] = vx[i];
            }
            share_order();
            f += 2u * kFetchTrip;
            if (lane == 0)
                share_post(&sh->fetch_next[w], f > nblk ? 0xffffffffu : f);
        }
        return;
    }
    // ---- wave 0: the passes ----
    // A group is 64 x 6 issue slots of passes; everything else in the loop is kept off its path: the words the other waves write
    // are read a group ahead (stale by one group, which only makes the decisions below later or more cautious), the straight
    // path has one taken branch, and a full group runs its 64 passes as one block of code.
    __builtin_amdgcn_s_setprio(3);
    float nf = a.core_carry[r].nf;
    uint32_t blk = a.blk0;
    unsigned ack = 0, idle = 0;
    bool unmarked = false;  // the results of the last trip are in the ring, w0_done does not say so yet
    NfGuess gs = {0u};
    NfGuessL gl = {0u, 0u, 0u, 0xffffffffu};
    const NfLaneConst lc = nf_lane_const(lane);
    const uint32_t group = a.core_guess == 1 ? kGuessGroup : 64u;  // blocks per trip
    int n_rounds = 0;
    CORE_PROF(unsigned long long t_begin = prof_now(); unsigned long long t_idle = 0, t_mark = 0, t_finish = 0; int n_groups = 0, n_restarts = 0, n_trips = 0; bool idling = false; unsigned long long t_guess = 0;)
    // blocks wave 0 may be ahead of wave 1: at least two groups (wave 1 waits for the whole group it is in), at most the ring
    // (as far as the ring reaches: the decay waves find a decay when wave 0 passes its start, and the further ahead of wave 1 that is,
    //  the more of the decay's chain -- 11-17 us -- is walked before wave 1 needs it: 768 -> 1920 blocks, 1.24 -> 1.17-1.19 ms per
    //  64-s call; rings of 4096 blocks with more lead gain nothing: a restart then throws too much away)
    const uint32_t lead = max(128u, min(static_cast<uint32_t>(a.core_lead > 0 ? a.core_lead : 1920), kNfRing - 128u));
    unsigned quit = share_peek(&sh->quit), rs = share_peek(&sh->rb_seq), w1_pos = share_peek(&sh->w1_pos),
             op_done = min(share_peek(&sh->fetch_next[0]), share_peek(&sh->fetch_next[1]));
    for (;;) {
        const uint32_t n = min(group, nblk - blk);
        const bool go = !quit && rs == ack && blk < nblk && blk + 64u <= w1_pos + lead && op_done >= blk + n;
        if (__builtin_expect(!go, 0)) {
            if (quit)
                break;
            if (unmarked) {  // (nothing follows at once: the last trip's mark now)
                share_order();
                if (lane == 0)
                    share_post(&sh->w0_done, blk);
                unmarked = false;
            }
            CORE_PROF(if (!idling) { idling = true; t_mark = prof_now(); if (blk >= nblk && !t_finish) t_finish = t_mark; })
            if (rs != ack) {  // wave 1 disagrees from block rb_blk on: start again there with its noise floor
                share_order();
                blk = share_peek(&sh->rb_blk);
                nf = *(lds_vf32*)&sh->rb_nf;
                gs.dh = 0u, gl.key = 0xffffffffu;  // (the blocks before the restart point were not this chain's)
                CORE_PROF(++n_restarts;)
                ack = rs;
                if (lane == 0)
                    share_post(&sh->w0_done, blk);
                share_order();
                if (lane == 0)
                    share_post(&sh->rb_ack, ack);
            } else {  // done, far enough ahead, or no operands yet
                __builtin_amdgcn_s_sleep(1);
                if (++idle > 8u * kShareSpin)
                    break;  // (wave 1 never came: it gives up on its side as well)
            }
            quit = share_peek(&sh->quit), rs = share_peek(&sh->rb_seq), w1_pos = share_peek(&sh->w1_pos);
            op_done = min(share_peek(&sh->fetch_next[0]), share_peek(&sh->fetch_next[1]));
            continue;
        }
        idle = 0;
        CORE_PROF(if (idling) { idling = false; t_idle += prof_now() - t_mark; })
        float op = *(lds_vf32*)&sh->opring[(blk + lane) & (kOpRing - 1u)];
        // (the next trip's view of the other waves: in flight during the passes)
        const unsigned quit_n = *(const lds_vu32*)&sh->quit, rs_n = *(const lds_vu32*)&sh->rb_seq, w1_n = *(const lds_vu32*)&sh->w1_pos,
                       opd_n = *(const lds_vu32*)&sh->fetch_next[0], opd_n1 = *(const lds_vu32*)&sh->fetch_next[1];
        // The mark for the trip before this one: its results have long landed, and the wait for them is the wait for this trip's
        // operands, which the first round needs anyway (at the end of the trip it was a round trip to LDS of its own).
        if (unmarked) {
            share_order();
            if (lane == 0)
                share_post(&sh->w0_done, blk);
            unmarked = false;
        }
        float vnf;
        if (a.core_guess == 1 && n == kGuessGroup) {
            // What a trip costs beside its rounds is two round trips to LDS -- the operands in, the results out before the mark -- and
            // the bookkeeping above (0.4-0.5 us against 0.1-0.2 of rounds): up to four groups per trip, as far as the operands, the
            // chunk and the lead reach; each group's operands are asked for while the group before is walked, the results wait for
            // the mark once.
            const uint32_t reach = min(min(op_done, nblk), w1_pos + lead - 1u) - blk;  // (>= 63 here)
            // (eight per trip: no better; one per trip while the walking wave is right behind: no better either)
            const uint32_t ngr = reach >= 4u * kGuessGroup ? 4u : (reach >= 3u * kGuessGroup ? 3u : (reach >= 2u * kGuessGroup ? 2u : 1u));
            for (uint32_t h = 0;;) {
                const float op_next = *(lds_vf32*)&sh->opring[(blk + kGuessGroup + lane) & (kOpRing - 1u)];  // (unused after the last group)
                CORE_PROF(const unsigned long long tg0 = prof_now();)
                vnf = nf_chain_guess63(nf, op, gl, lc, lane, n_rounds);
                CORE_PROF(t_guess += prof_now() - tg0;)
                if (lane < static_cast<int>(kGuessGroup))
                    *(lds_vf32*)&sh->nfring[(blk + lane) & (kNfRing - 1u)] = vnf;
                blk += kGuessGroup;
                CORE_PROF(++n_groups;)
                if (++h >= ngr)
                    break;
                op = op_next;
            }
        } else {
            // If every operand of the group lies above anything the chain can reach in 64 steps from here (it grows by less than
            // 1e-6 (1 + 2^-23) + 2^-22 of itself per step while min() takes the chain value), min() is the identity throughout and
            // the pass needs one instruction less: the same operations in the same order, bit for bit.
            if (a.core_guess && n == 64u) {
                vnf = nf_chain_guess64(nf, op, gs, lane, n_rounds);
            } else {
                const bool above = n == 64u && __ballot(op >= nf * 1.0001f + 7e-5f) == ~0ull;
                vnf = above ? nf_chain_self64(nf) : ((n == 64u) ? nf_chain_min64(nf, op) : nf_chain_min(nf, op, static_cast<int>(n)));
                nf = rl(vnf, static_cast<int>(n) - 1);
                gs.dh = 0u, gl.key = 0xffffffffu;
            }
            if (lane < static_cast<int>(n))
                *(lds_vf32*)&sh->nfring[(blk + lane) & (kNfRing - 1u)] = vnf;
            blk += n;
            CORE_PROF(++n_groups;)
        }
        unmarked = true;
        CORE_PROF(++n_trips;)
        quit = __builtin_amdgcn_readfirstlane(quit_n), rs = __builtin_amdgcn_readfirstlane(rs_n);
        w1_pos = __builtin_amdgcn_readfirstlane(w1_n), op_done = min(__builtin_amdgcn_readfirstlane(opd_n), __builtin_amdgcn_readfirstlane(opd_n1));
    }
    CORE_PROF(if (lane == 0 && r < 8) printf("core w0 row %d: finished after %llu us  idle before that %llu us  groups %d rounds %d restarts %d  trips %d  in the rounds %llu us\n", r, (t_finish - t_begin) / 100, t_idle / 100, n_groups,
                                           n_rounds, n_restarts, n_trips, t_guess / 100);)
}

// Register budget of the lane kernels (k_tp_seg / k_tp_fix / k_tp_redo): waves per SIMD they are compiled for, 0 = whatever they take
#ifndef MI_LANE_EU
#define MI_LANE_EU 0
#endif
#if MI_LANE_EU > 0
#define MI_LANE_BOUNDS __launch_bounds__(64, MI_LANE_EU)
#else
#define MI_LANE_BOUNDS __launch_bounds__(64)
#endif
// =====================================================================================================
// B / D: the segment engine
// =====================================================================================================
struct TpLane {
    float nf, cap, c, full, level;
    int cur, next, delay, low, recent, closed;
    float agc;
    int d_open, d_flappy, uses_agc, open_mask, nev;
};

struct TpFsm {  // canonical comparable part
    int cur, next, delay, low, recent, closed;
};

__device__ __forceinline__ TpFsm canon(const TpLane& s) {
    TpFsm f;
    f.cur = s.cur;
    f.next = s.next;
    const bool delay_live = (s.next == SQ_OPENING && s.cur == SQ_OPENING) || (s.next == SQ_CLOSING && s.cur == SQ_CLOSING) ||
                            (s.next == SQ_LSA && (s.cur == SQ_LSA || s.cur == SQ_CLOSING));
    f.delay = delay_live ? s.delay : 0;
    // low_signal_count_ is reset on entering OPENING and never read in CLOSED / LOW_SIGNAL_ABORT
    f.low = (s.cur == SQ_OPENING || s.cur == SQ_CLOSING || s.cur == SQ_OPEN) ? s.low : 0;
    f.recent = s.recent;
    // closed_sample_count_ is rebuilt from 0 on every entry into CLOSED and read only at the end of OPENING
    f.closed = (s.cur == SQ_CLOSED || s.cur == SQ_OPENING) ? s.closed : 0;
    return f;
}
__device__ __forceinline__ bool same(const TpFsm& x, const TpFsm& y) {
    return x.cur == y.cur && x.next == y.next && x.delay == y.delay && x.low == y.low && x.recent == y.recent && x.closed == y.closed;
}

// one sample of the loop for a plain AM channel; returns waveout[j]
__device__ __forceinline__ float tp_step(TpLane& s, const ChanParams& p, const uint32_t i, const float x, const float aud,
                                         const float* __restrict__ magrow, const bool in_seg, const uint32_t batch0, int* __restrict__ ev_slot,
                                         const size_t ev_stride) {
    // ---- Squelch::update_current_state, squelch.cpp:363-460 (no post filter, no CTCSS) ----
    if (s.next == SQ_OPENING) {
        if (s.cur != SQ_OPENING) {
            s.delay = 0;
            s.low = 0;
            s.cur = SQ_OPENING;
        } else {
            s.delay++;
            if (s.delay >= kOpenDelay) {
                if (s.closed < kRecent) {
                    s.recent++;
                    if (s.recent >= kFlap && in_seg)
                        s.d_flappy++;
                    s.level = level_of(p, s.nf, s.recent);
                }
                s.next = (s.c >= s.level) ? SQ_OPEN : SQ_CLOSED;
            }
        }
    } else if (s.next == SQ_CLOSING) {
        if (s.cur != SQ_CLOSING) {
            s.delay = 0;
            s.cur = SQ_CLOSING;
        } else {
            s.delay++;
            if (s.delay >= kCloseDelay) {
                if (!(s.c >= s.level)) {
                    s.next = SQ_CLOSED;
                } else {
                    s.cur = SQ_OPEN;
                    s.next = SQ_OPEN;
                }
            }
        }
    } else if (s.next == SQ_LSA) {
        if (s.cur != SQ_LSA) {
            if (s.cur != SQ_CLOSING)
                s.delay = 0;
            s.cur = SQ_LSA;
        } else {
            s.delay++;
            if (s.delay >= kCloseDelay)
                s.next = SQ_CLOSED;
        }
    } else if (s.next == SQ_OPEN && s.cur != SQ_OPEN) {
        if (in_seg)
            s.d_open++;
        s.cur = SQ_OPEN;
    } else if (s.next == SQ_CLOSED && s.cur != SQ_CLOSED) {
        s.closed = 0;
        s.cur = SQ_CLOSED;
    } else if (s.next == SQ_CLOSED && s.cur == SQ_CLOSED) {
        if (s.closed < kRecent) {
            s.closed++;
        } else if (s.closed == kRecent) {
            if (s.recent != 0) {
                s.recent = 0;
                s.level = level_of(p, s.nf, 0);
            }
        }
    } else {
        s.cur = s.next;
    }
    // ---- process_raw_sample, squelch.cpp:203-245 ----
    if ((i & 15u) == 0) {
        s.nf = noise_floor_step(s.nf, s.c);
        s.cap = cap_of(p, s.nf);
        s.level = level_of(p, s.nf, s.recent);
    }
    s.full = ema99(s.full, x);
    s.c = capped_step(s.c, x, s.cap);
    if (s.cur == SQ_OPEN && !(s.c >= s.level))
        s.next = SQ_CLOSING;  // set_state(CLOSING) from OPEN
    if (s.cur == SQ_CLOSED && (s.c >= s.level))
        s.next = SQ_OPENING;  // set_state(OPENING) from CLOSED
    if (s.cur != SQ_CLOSED && s.cur != SQ_LSA) {
        if (x >= s.level) {
            s.low = 0;
        } else {
            s.low++;
            if (s.low >= kLowSignalAbort)
                s.next = (s.cur == SQ_OPENING) ? SQ_CLOSED : SQ_LSA;  // set_state(LOW_SIGNAL_ABORT), squelch.cpp:338-341
        }
    }
    // ---- AM edges, rtl_airband.cpp:554-569 ----
    if (s.cur != SQ_OPEN && s.next == SQ_OPEN) {  // first_open_sample: bootstrap agcavgfast
        static_assert(kAgcExtra % 25 == 0, "bootstrap batches");
#pragma unroll 1
        for (int k0 = 0; k0 < kAgcExtra; k0 += 25) {  // 25 independent loads in flight, then the serial average
            float w[25];
#pragma unroll
            for (int j = 0; j < 25; ++j)
                w[j] = magrow[i + k0 + j];
#pragma unroll
            for (int j = 0; j < 25; ++j)
                if (w[j] >= s.level)
                    s.agc = s.agc * 0.9f + w[j] * 0.1f;
        }
        if (in_seg)
            s.uses_agc = 1;
    } else if ((s.cur == SQ_CLOSING && s.next == SQ_CLOSED) || (s.cur != SQ_LSA && s.next == SQ_LSA)) {  // last_open_sample
        if (in_seg && s.nev < TP_MAXEV) {  // rare: straight to the segment's record (events are >= 197 steps apart)
            ev_slot[static_cast<size_t>(s.nev) * ev_stride] = static_cast<int>(i);
            s.nev++;
        }
    }
    // ---- audio, rtl_airband.cpp:574-641 ----
    float wout = 0.0f;
    if (s.cur == SQ_OPEN || s.cur == SQ_CLOSING) {
        if (in_seg)
            s.uses_agc = 1;
        if (x > s.level)
            s.agc = s.agc * 0.995f + x * 0.005f;
        wout = (aud - s.agc) / (s.agc * 1.5f);
        if (fabsf(wout) > 0.8f) {
            wout *= 0.85f;
            s.agc *= 1.15f;
        }
        wout *= p.ampfactor;
        if (wout != wout)
            wout = 0.0f;
        else if (wout > 1.0f)
            wout = 1.0f;
        else if (wout < -1.0f)
            wout = -1.0f;
        if (in_seg)
            s.open_mask |= 1 << (i / kWaveBatch - batch0);
    }
    return wout;
}

__device__ __forceinline__ void out_store4(const TpArgs& a, const int row, const uint32_t i, const float4 v) {
    // virtual waveout index of step i is AGC_EXTRA + i: [0, nsteps) emitted audio, the rest is the lookahead
    const uint32_t vi = kAgcExtra + i;
    if (vi < a.nsteps)
        *reinterpret_cast<float4*>(a.wmain + static_cast<size_t>(row) * a.wmain_stride + vi) = v;
    else
        *reinterpret_cast<float4*>(a.carry + static_cast<size_t>(row) * kAgcExtra + (vi - a.nsteps)) = v;
}

// four steps i .. i+3 (i a multiple of 4) from the chunk's squelch samples xc and audio samples ac
//
// Fast paths: in the steady and the waiting regimes of the state machine nothing but counters moves, which is
// proven for the whole chunk on a trial copy of the core state before anything is committed:
//   CLOSED/CLOSED    no sample reaches the squelch level          -> closed_sample_count_ (+ the flap reset), audio 0
//   OPEN/OPEN        capped_ stays >= level, low-signal count < 88 -> AGC + audio
//   OPENING/OPENING  delay_ does not run out, low count < 88       -> delay_ += 4, audio 0
//   CLOSING/CLOSING  delay_ does not run out, low count < 88       -> delay_ += 4, AGC + audio (still open)
//   LSA/LSA          delay_ does not run out                       -> delay_ += 4, audio 0
// Every other chunk (an edge, a delay running out, ...) takes tp_step() four times from the untouched state.
__device__ __forceinline__ void tp_chunk(TpLane& s, const ChanParams& p, const TpArgs& a, const int row, const float* __restrict__ magrow,
                                         const uint32_t i, const float4 xc, const float4 ac, const bool in_seg, const uint32_t batch0,
                                         int* __restrict__ ev_slot, const size_t ev_stride) {
    const float xv[4] = {xc.x, xc.y, xc.z, xc.w};
    const float av[4] = {ac.x, ac.y, ac.z, ac.w};
    float t_nf = s.nf, t_cap = s.cap, t_c = s.c, t_full = s.full, t_level = s.level;
    if ((i & 15u) == 0) {  // i is a multiple of 4: only the chunk's first step can start a 16-sample block
        t_nf = noise_floor_step(t_nf, t_c);
        t_cap = cap_of(p, t_nf);
        t_level = level_of(p, t_nf, s.recent);
    }
    bool all_below = true, all_above = true;
    int low = s.low, lowmax = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        t_full = ema99(t_full, xv[k]);
        t_c = capped_step(t_c, xv[k], t_cap);
        all_below = all_below && !(t_c >= t_level);
        all_above = all_above && (t_c >= t_level);
        low = (xv[k] >= t_level) ? 0 : low + 1;  // squelch.cpp:236-244
        lowmax = max(lowmax, low);
    }
    const bool same_state = s.cur == s.next;
    const bool low_ok = lowmax < kLowSignalAbort;
    const bool quiet = same_state && s.cur == SQ_CLOSED && all_below;
    const bool steady_open = same_state && s.cur == SQ_OPEN && all_above && low_ok;
    const bool wait_opening = same_state && s.cur == SQ_OPENING && s.delay + 4 < kOpenDelay && low_ok;
    const bool wait_closing = same_state && s.cur == SQ_CLOSING && s.delay + 4 < kCloseDelay && low_ok;
    const bool wait_lsa = same_state && s.cur == SQ_LSA && s.delay + 4 < kCloseDelay;
    float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
    if (quiet || steady_open || wait_opening || wait_closing || wait_lsa) {
        s.nf = t_nf, s.cap = t_cap, s.c = t_c, s.full = t_full, s.level = t_level;
        if (quiet) {
            // squelch.cpp:442-449 four times; the reset fires at a step that starts with the count at 1000.  It only
            // raises the level (normal >= flappy ratio), so the chunk stays quiet under the new level as well.
            if (s.closed + 3 >= kRecent && s.recent != 0) {
                s.recent = 0;
                s.level = level_of(p, s.nf, 0);
            }
            s.closed = min(s.closed + 4, kRecent);
        } else if (wait_lsa) {
            s.delay += 4;
        } else {
            s.low = low;
            if (!steady_open)
                s.delay += 4;
            if (!wait_opening) {  // OPEN or CLOSING: rtl_airband.cpp:574-641 with is_open() true
                float wv[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (xv[k] > t_level)
                        s.agc = s.agc * 0.995f + xv[k] * 0.005f;
                    // The AGC clip feedback (rtl_airband.cpp:580-584) puts the correctly rounded division on the serial
                    // chain.  Its outcome |waveout| > 0.8 is decided from a reciprocal estimate (relative error < 3e-7)
                    // unless the quotient is within 1 % of the threshold; the exact quotient is still the audio sample.
                    const float num = av[k] - s.agc, den = s.agc * 1.5f;
                    const float qa = fabsf(num) * __builtin_amdgcn_rcpf(den);
                    float wout = num / den;
                    bool clip;
                    if (qa < 0.79f)
                        clip = false;
                    else if (qa > 0.81f)
                        clip = true;
                    else
                        clip = fabsf(wout) > 0.8f;  // also taken when the estimate is not a number
                    if (clip) {
                        wout *= 0.85f;
                        s.agc *= 1.15f;
                    }
                    wout *= p.ampfactor;
                    if (wout != wout)
                        wout = 0.0f;
                    else if (wout > 1.0f)
                        wout = 1.0f;
                    else if (wout < -1.0f)
                        wout = -1.0f;
                    wv[k] = wout;
                    if (in_seg)
                        s.open_mask |= 1 << ((i + k) / kWaveBatch - batch0);
                }
                if (in_seg)
                    s.uses_agc = 1;
                w = make_float4(wv[0], wv[1], wv[2], wv[3]);
            }
        }
    } else {
        // general path (rare): one copy of the full step
        float wv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int k = 0; k < 4; ++k) {
            const float xk = (k == 0) ? xv[0] : (k == 1) ? xv[1] : (k == 2) ? xv[2] : xv[3];  // in registers already
            const float ak = (k == 0) ? av[0] : (k == 1) ? av[1] : (k == 2) ? av[2] : av[3];
            const float wk = tp_step(s, p, i + k, xk, ak, magrow, in_seg, batch0, ev_slot, ev_stride);
            wv[0] = (k == 0) ? wk : wv[0];
            wv[1] = (k == 1) ? wk : wv[1];
            wv[2] = (k == 2) ? wk : wv[2];
            wv[3] = (k == 3) ? wk : wv[3];
        }
        w = make_float4(wv[0], wv[1], wv[2], wv[3]);
    }
    if (in_seg)
        out_store4(a, row, i, w);
}

// ---- sixteen steps i .. i+15 (one squelch block) ----
// The wide lanes are issue bound like the core wave, so whole blocks are settled from the block aggregates of k_tp_full
// wherever the state machine cannot move:
//   capped_ over the block   MERGED (CLOSED only: just the maximum is known), SATURATED, or the bare-EMA trial (max and min)
//   low_signal_count_        stays 0 when no sample of the block is below the level, else counted sample by sample
//   CLOSED/CLOSED   max capped_ < level                                  -> closed_sample_count_ (+ the flap reset), audio 0
//   OPEN/OPEN       min capped_ >= level, low count < 88                  -> AGC + audio
//   OPENING/OPENING, CLOSING/CLOSING, LSA/LSA: delay_ does not run out    -> delay_ += 16 (+ AGC + audio while CLOSING)
// Any other block goes through tp_chunk() four times from the untouched state.
struct BlkAgg {
    float fe, fm, x0, xm;
};
__device__ __forceinline__ BlkAgg agg_load(const TpArgs& a, const size_t idx) {
    BlkAgg g;
    g.fe = a.blk_fe[idx], g.fm = a.blk_fm[idx], g.x0 = a.blk_x0[idx], g.xm = a.blk_xm[idx];
    return g;
}
struct BlkSamples {  // squelch samples (magrow[100 + i ..]) and audio samples (magrow[i ..]); named members: arrays end up in scratch
    float4 x0, x1, x2, x3, a0, a1, a2, a3;
};
__device__ __forceinline__ BlkSamples samples_load(const float* __restrict__ magrow, const uint32_t i) {
    BlkSamples q;
    const float4* __restrict__ xp = reinterpret_cast<const float4*>(magrow + kAgcExtra + i);
    const float4* __restrict__ ap = reinterpret_cast<const float4*>(magrow + i);
    q.x0 = xp[0], q.x1 = xp[1], q.x2 = xp[2], q.x3 = xp[3];
    q.a0 = ap[0], q.a1 = ap[1], q.a2 = ap[2], q.a3 = ap[3];
    return q;
}

// rtl_airband.cpp:574-641 with is_open() true, for one sample; returns waveout
template <bool kAudio>
__device__ __forceinline__ float agc_audio(TpLane& s, const ChanParams& p, const float x, const float aud, const float level) {
    if (x > level)
        s.agc = s.agc * 0.995f + x * 0.005f;
    // The AGC clip feedback (rtl_airband.cpp:580-584) puts the correctly rounded division on the serial chain.  Its
    // outcome |waveout| > 0.8 is decided by comparing |num| with 0.79 den and 0.81 den (each exact to 2e-7) unless the
    // quotient is within 1 % of the threshold; the exact quotient is still the audio sample (not formed at all while
    // warming up).  A den that is not a positive number fails both comparisons and takes the exact path.
    const float num = aud - s.agc, den = s.agc * 1.5f;
    const float an = fabsf(num);
    bool clip = an > den * 0.81f;
    if (!(an < den * 0.79f) && !clip)
        clip = fabsf(num / den) > 0.8f;
    if (clip)
        s.agc *= 1.15f;
    if (!kAudio)
        return 0.0f;
    float wout = num / den;
    if (clip)
        wout *= 0.85f;
    wout *= p.ampfactor;
    if (wout != wout)
        wout = 0.0f;
    else if (wout > 1.0f)
        wout = 1.0f;
    else if (wout < -1.0f)
        wout = -1.0f;
    return wout;
}

// agc_audio() over the 16 samples of a block without a branch per sample (a lone wave pays dearly for the scalar side of a
// divergent branch): the clip decisions come from the two comparisons alone and `amb` records whether any |num| fell between
// 0.79 den and 0.81 den.  Returns false in that case (nothing is committed; the caller takes the general path for the block).
template <bool kAudio>
__device__ __forceinline__ bool agc_block(float& agc_io, const ChanParams& p, const float (&xs)[16], const float (&as)[16], const float level,
                                          const bool all_above, float (&wv)[16]) {
    if (all_above) {
        // Every sample above the level.  If |num| < 1.18 agc at every sample the quotient |num| / (1.5 agc) stays below
        // 0.7867 (1 + 2e-7) < 0.79, so no sample clips and the average follows the plain recurrence: 7 instructions per
        // sample while warming up (only agcavgfast is needed), plus the quotient itself when the audio is emitted.
        // Anything else takes the general block below.
        float agc = agc_io, m = -1.0f;
        float nums[16], dens[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            agc = agc * 0.995f + xs[j] * 0.005f;
            const float num = as[j] - agc;
            m = fmaxf(m, fabsf(num) - agc * 1.18f);
            if (kAudio)
                nums[j] = num, dens[j] = agc * 1.5f;
        }
        if (m < 0.0f && agc > 0.0f && agc < 3.0e38f) {
            agc_io = agc;
            if (kAudio) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    float wout = (nums[j] / dens[j]) * p.ampfactor;
                    if (wout != wout)
                        wout = 0.0f;
                    else if (wout > 1.0f)
                        wout = 1.0f;
                    else if (wout < -1.0f)
                        wout = -1.0f;
                    wv[j] = wout;
                }
            }
            return true;
        }
    }
    float agc = agc_io, amb = -1.0f;
    float nums[16], dens[16], cfs[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const float upd = agc * 0.995f + xs[j] * 0.005f;
        agc = (xs[j] > level) ? upd : agc;
        const float num = as[j] - agc, den = agc * 1.5f;
        const float an = fabsf(num);
        const float lo = den * 0.79f, hi = den * 0.81f;
        amb = fmaxf(amb, fminf(an - lo, hi - an));  // >= 0 iff lo <= an <= hi
        const bool clip = an > hi;
        agc = clip ? agc * 1.15f : agc;
        nums[j] = num, dens[j] = den, cfs[j] = clip ? 0.85f : 1.0f;
    }
    if (!(amb < 0.0f && agc > 0.0f && agc < 3.0e38f))
        return false;
    agc_io = agc;
    if (kAudio) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float wout = (nums[j] / dens[j]) * cfs[j];  // x * 1.0f is x
            wout *= p.ampfactor;
            if (wout != wout)
                wout = 0.0f;
            else if (wout > 1.0f)
                wout = 1.0f;
            else if (wout < -1.0f)
                wout = -1.0f;
            wv[j] = wout;
        }
    }
    return true;
}

template <bool kInSeg>
__device__ __forceinline__ void tp_block(TpLane& s, const ChanParams& p, const TpArgs& a, const int row, const float* __restrict__ magrow,
                                         const uint32_t i, const BlkAgg g, BlkSamples q, const bool have, const uint32_t batch0,
                                         int* __restrict__ ev_slot, const size_t ev_stride) {
    const float nf1 = noise_floor_step(s.nf, s.c);
    const float cap1 = cap_of(p, nf1);
    const float level1 = level_of(p, nf1, s.recent);
    const bool valid = g.fm >= 0.0f;
    const bool same_state = s.cur == s.next;
    // ---- CLOSED and merged: nothing but the aggregates is needed ----
    if (same_state && s.cur == SQ_CLOSED && valid && s.c == s.full && s.c < cap1 && g.fm < cap1 && g.fm < level1) {
        s.nf = nf1, s.cap = cap1, s.level = level1, s.c = g.fe, s.full = g.fe;
        // squelch.cpp:442-449 sixteen times; the reset fires at a step that starts with the count at 1000.  It only
        // raises the level (normal >= flappy ratio), so the block stays quiet under the new level as well.
        if (s.closed + 15 >= kRecent && s.recent != 0) {
            s.recent = 0;
            s.level = level_of(p, s.nf, 0);
        }
        s.closed = min(s.closed + 16, kRecent);
        if (kInSeg) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                out_store4(a, row, i + 4 * k, make_float4(0.f, 0.f, 0.f, 0.f));
        }
        return;
    }
    if (!have)
        q = samples_load(magrow, i);
    const float xs[16] = {q.x0.x, q.x0.y, q.x0.z, q.x0.w, q.x1.x, q.x1.y, q.x1.z, q.x1.w,
                          q.x2.x, q.x2.y, q.x2.z, q.x2.w, q.x3.x, q.x3.y, q.x3.z, q.x3.w};
    bool settled = false;
    if (same_state && valid) {
        // ---- capped_ over the block ----
        float c_end = 0.f, cmax = 0.f, cmin = 0.f;
        bool c_ok = false;
        if (capped_step(s.c, g.x0, cap1) == cap1 && g.xm >= cap1) {  // SATURATED
            c_end = cmax = cmin = cap1;
            c_ok = true;
        } else {  // the cap does not bind anywhere: bare EMA
            const float nfac = static_cast<float>(1.0 - static_cast<double>(0.99f));
            float cs = s.c;
            cmax = cmin = 0.f;
            float emax = 0.f, emin = 3.4e38f;  // of capped_ after each sample (what the state machine compares)
#pragma unroll
            for (int j = 0; j < 16; j += 2) {
                const float c1 = cs * 0.99f + xs[j] * nfac;
                cs = c1 * 0.99f + xs[j + 1] * nfac;
                emax = fmaxf(fmaxf(emax, c1), cs);
                emin = fminf(fminf(emin, c1), cs);
            }
            c_end = cs, cmax = emax, cmin = emin;
            c_ok = emax < cap1 && !(s.c >= cap1 && xs[0] >= cap1);  // squelch.cpp:509-510 never takes the cap
        }
        // ---- low_signal_count_ over the block (squelch.cpp:236-244; not counted in CLOSED and LOW_SIGNAL_ABORT) ----
        int low = s.low, lowmax = 0;
        const bool counts = s.cur != SQ_CLOSED && s.cur != SQ_LSA;
        if (counts) {
            if (g.x0 >= level1 && g.xm >= level1) {
                low = 0;
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    low = (xs[j] >= level1) ? 0 : low + 1;
                    lowmax = max(lowmax, low);
                }
            }
        }
        const bool low_ok = lowmax < kLowSignalAbort;
        const bool quiet = s.cur == SQ_CLOSED && cmax < level1;
        const bool steady_open = s.cur == SQ_OPEN && cmin >= level1 && low_ok;
        const bool wait_opening = s.cur == SQ_OPENING && s.delay + 16 < kOpenDelay && low_ok;
        const bool wait_closing = s.cur == SQ_CLOSING && s.delay + 16 < kCloseDelay && low_ok;
        const bool wait_lsa = s.cur == SQ_LSA && s.delay + 16 < kCloseDelay;
        bool ok = c_ok && (quiet || steady_open || wait_opening || wait_closing || wait_lsa);
        float wv[16];
#pragma unroll
        for (int j = 0; j < 16; ++j)
            wv[j] = 0.f;
        float agc = s.agc;
        if (ok && (steady_open || wait_closing)) {  // still open: rtl_airband.cpp:574-641
            const float as[16] = {q.a0.x, q.a0.y, q.a0.z, q.a0.w, q.a1.x, q.a1.y, q.a1.z, q.a1.w,
                                  q.a2.x, q.a2.y, q.a2.z, q.a2.w, q.a3.x, q.a3.y, q.a3.z, q.a3.w};
            ok = agc_block<kInSeg>(agc, p, xs, as, level1, g.x0 > level1 && g.xm > level1, wv);
        }
        if (ok) {
            settled = true;
            s.nf = nf1, s.cap = cap1, s.level = level1, s.c = c_end, s.full = g.fe;
            if (quiet) {
                if (s.closed + 15 >= kRecent && s.recent != 0) {
                    s.recent = 0;
                    s.level = level_of(p, s.nf, 0);
                }
                s.closed = min(s.closed + 16, kRecent);
            } else if (wait_lsa) {
                s.delay += 16;
            } else {
                s.low = low;
                if (!steady_open)
                    s.delay += 16;
                if (!wait_opening) {
                    s.agc = agc;
                    if (kInSeg) {
                        s.open_mask |= 1 << (i / kWaveBatch - batch0);  // 2000 = 125 blocks: a block lies in one batch
                        s.uses_agc = 1;
                    }
                }
            }
            if (kInSeg) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    out_store4(a, row, i + 4 * k, make_float4(wv[4 * k], wv[4 * k + 1], wv[4 * k + 2], wv[4 * k + 3]));
            }
        }
    }
    if (!settled) {
        CORE_PROF(++g_prof_unsettled;)
#pragma unroll 1
        for (int k = 0; k < 4; ++k) {  // one copy of tp_chunk(); selects, not indexing, keep the samples in registers
            const float4 xk = (k == 0) ? q.x0 : (k == 1) ? q.x1 : (k == 2) ? q.x2 : q.x3;
            const float4 ak = (k == 0) ? q.a0 : (k == 1) ? q.a1 : (k == 2) ? q.a2 : q.a3;
            tp_chunk(s, p, a, row, magrow, i + 4 * k, xk, ak, kInSeg, batch0, ev_slot, ev_stride);
        }
    }
}

// run steps [i0, i1) (multiples of 16); writes audio when in_seg.  The next block's aggregates are always requested a
// block ahead; its samples only when the lane does not expect to settle it from the aggregates alone.
template <bool kInSeg>
__device__ __forceinline__ void tp_run(TpLane& s, const ChanParams& p, const TpArgs& a, const int r, const int row,
                                       const float* __restrict__ magrow, const uint32_t i0, const uint32_t i1, const uint32_t batch0,
                                       const size_t rec_idx) {
    if (i0 >= i1)
        return;
    int* __restrict__ ev_slot = a.rec + 19 * a.rec_stride + rec_idx;
    const size_t ev_stride = a.rec_stride;
    const size_t bbase = static_cast<size_t>(r) * a.nblk;
    BlkAgg gn = agg_load(a, bbase + (i0 >> 4));
    BlkSamples qn = samples_load(magrow, i0);
    bool have_n = true;
    for (uint32_t i = i0; i < i1; i += 16) {
        const BlkAgg g = gn;
        const BlkSamples q = qn;
        const bool have = have_n;
        if (i + 16 < i1) {
            gn = agg_load(a, bbase + ((i + 16) >> 4));
            // A quiet CLOSED block is settled from its aggregates alone (tp_block's first case) -- 16 bytes instead of 144.  The
            // next block's samples are requested only if it does not look like one from here (this block's aggregates, the state
            // before it); a block that turns out to need them after all loads them itself, once per change of regime.
            have_n = a.eager_samples || !(s.cur == SQ_CLOSED && s.next == SQ_CLOSED && s.c == s.full && g.fm >= 0.0f && g.fm < fminf(s.cap, s.level));
            if (have_n)
                qn = samples_load(magrow, i + 16);
        }
        tp_block<kInSeg>(s, p, a, row, magrow, i, g, q, have, batch0, ev_slot, ev_stride);
    }
}

__device__ __forceinline__ void rec_store(const TpArgs& a, const size_t idx, const TpFsm& S, const float s_agc, const TpLane& e) {
    int* __restrict__ rec = a.rec;
    const size_t st = a.rec_stride;
    rec[0 * st + idx] = S.cur;
    rec[1 * st + idx] = S.next;
    rec[2 * st + idx] = S.delay;
    rec[3 * st + idx] = S.low;
    rec[4 * st + idx] = S.recent;
    rec[5 * st + idx] = S.closed;
    rec[6 * st + idx] = __float_as_int(s_agc);
    const TpFsm E = canon(e);
    rec[7 * st + idx] = E.cur;
    rec[8 * st + idx] = E.next;
    rec[9 * st + idx] = E.delay;
    rec[10 * st + idx] = E.low;
    rec[11 * st + idx] = E.recent;
    rec[12 * st + idx] = E.closed;
    rec[13 * st + idx] = __float_as_int(e.agc);
    rec[14 * st + idx] = e.uses_agc;
    rec[15 * st + idx] = e.d_open;
    rec[16 * st + idx] = e.d_flappy;
    rec[17 * st + idx] = e.open_mask;
    rec[18 * st + idx] = e.nev;  // the event steps themselves were written when they happened
}

__device__ __forceinline__ void seg_reset(TpLane& s) {
    s.d_open = s.d_flappy = s.uses_agc = s.open_mask = s.nev = 0;
}

__device__ __forceinline__ void load_core(TpLane& s, const ChanParams& p, const TpCore& t) {
    s.nf = t.nf, s.cap = t.cap, s.c = t.c, s.full = t.full;
    s.level = level_of(p, s.nf, s.recent);
}

__global__ MI_LANE_BOUNDS void k_tp_seg(const TpArgs a) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;  // blockDim.x = lanes per wave (see launch_tp_seg)
    const int nsc = static_cast<int>(a.seg1 - a.seg0);  // segments of this chunk
    if (gid >= a.nrows * nsc)
        return;
    const int r = gid / nsc;
    const uint32_t k = a.seg0 + static_cast<uint32_t>(gid - r * nsc);
    const int row = a.rows[r];
    const ChanParams p = a.cp[row % a.nch];
    const float* __restrict__ magrow = a.mag + static_cast<size_t>(row) * a.plane_stride;
    const uint32_t s0 = k * a.L, s1 = min(s0 + a.L, a.nsteps);
    const uint32_t wsegs = TP_W / a.L;
    const uint32_t wk = k > wsegs ? k - wsegs : 0;  // boundary index where the warm-up starts
    TpLane s;
    seg_reset(s);
    const bool spec = a.spec_head != 0;
    if (wk == 0 && !spec) {  // from the true state at the start of the call
        const ChanState& cs = a.st[row];
        s.cur = cs.current_state, s.next = cs.next_state, s.delay = cs.delay, s.low = cs.low_signal_count;
        s.recent = static_cast<int>(cs.recent_open_count), s.closed = static_cast<int>(cs.closed_sample_count);
        s.agc = cs.agcavgfast;
    } else {  // guess: idle CLOSED; agcavgfast unknown (checked by the scan wherever it matters)
        s.cur = s.next = SQ_CLOSED;
        s.delay = s.low = s.recent = 0;
        s.closed = kRecent;
        // agcavgfast keeps its value while the squelch is closed, so what a transmission starts from is what the one before it left --
        // possibly seconds back, far beyond any warm-up.  The open-edge bootstrap (0.9^100) and the samples after it shrink whatever
        // the guess was wrong by, but only a guess within a fraction of a percent disappears under one ulp at once; 0.5 took ~3000
        // open samples (six 512-step segments per transmission for the fix chains).  The channel's last committed value is such a
        // guess whenever consecutive transmissions are about equally strong; it may be a call or two old, and it is only a guess.
        s.agc = a.agc_hint ? a.st[row].agcavgfast : 0.5f;
    }
    const size_t rec_idx = static_cast<size_t>(r) * a.nseg + k;
    // The warm-up is one run over this call's arrays, preceded -- when it reaches back over the start of the call and the head is
    // speculative -- by one over the previous call's (phase 0).  One copy of the block code serves both: the arrays are switched.
    TpArgs aw = a;
    const float* __restrict__ mrow = magrow;
    uint32_t w0 = wk * a.L, w1 = s0;
    int ph = 1;
    if (spec && k < wsegs) {
        const uint32_t wkp = (a.prev_n - (TP_W - k * a.L)) / a.L;  // the last boundary of the previous call that leaves TP_W steps
        load_core(s, p, a.prev_core[static_cast<size_t>(r) * (a.prev_nseg + 1) + wkp]);
        aw.blk_fe = const_cast<float*>(a.prev_blk_fe), aw.blk_fm = const_cast<float*>(a.prev_blk_fm);
        aw.blk_x0 = const_cast<float*>(a.prev_blk_x0), aw.blk_xm = const_cast<float*>(a.prev_blk_xm);
        aw.nblk = a.prev_nblk;
        mrow = a.prev_mag + static_cast<size_t>(row) * a.plane_stride;
        w0 = wkp * a.L, w1 = a.prev_n;
        ph = 0;
    } else {
        load_core(s, p, a.core[static_cast<size_t>(r) * (a.nseg + 1) + wk]);
    }
#pragma unroll 1
    for (; ph < 2; ++ph) {
        tp_run<false>(s, p, aw, r, row, mrow, w0, w1, 0, rec_idx);
        aw.blk_fe = a.blk_fe, aw.blk_fm = a.blk_fm, aw.blk_x0 = a.blk_x0, aw.blk_xm = a.blk_xm;
        aw.nblk = a.nblk;
        mrow = magrow;
        w0 = 0, w1 = s0;  // (only reached after phase 0: the rest of the warm-up, from the start of this call)
    }
    const TpFsm S = canon(s);
    const float s_agc = s.agc;
    seg_reset(s);
    tp_run<true>(s, p, a, r, row, magrow, s0, s1, s0 / kWaveBatch, static_cast<size_t>(r) * a.nseg + k);
    rec_store(a, static_cast<size_t>(r) * a.nseg + k, S, s_agc, s);
}

// =====================================================================================================
// C: scan -- accept segments whose recorded start state equals the predecessor's end state
// =====================================================================================================
__device__ __forceinline__ TpFsm rec_fsm(const int* __restrict__ rec, const size_t st, const size_t idx, const int base) {
    TpFsm f;
    f.cur = rec[(base + 0) * st + idx];
    f.next = rec[(base + 1) * st + idx];
    f.delay = rec[(base + 2) * st + idx];
    f.low = rec[(base + 3) * st + idx];
    f.recent = rec[(base + 4) * st + idx];
    f.closed = rec[(base + 5) * st + idx];
    return f;
}

// one wave per row; returns (wave-uniform) whether every segment of the chunk is accepted and the first that is not
__device__ __forceinline__ bool scan_row(const TpArgs& a, const int r, const int row, const int lane, const int diag_slot, const bool diag_assign,
                                         uint32_t& first_bad_out) {
    const int* __restrict__ rec = a.rec;
    const size_t st = a.rec_stride;
    const size_t base = static_cast<size_t>(r) * a.nseg;

    // the true state before step 0, canonical
    TpLane init;
    {
        const ChanState& cs = a.st[row];
        init.cur = cs.current_state, init.next = cs.next_state, init.delay = cs.delay, init.low = cs.low_signal_count;
        init.recent = static_cast<int>(cs.recent_open_count), init.closed = static_cast<int>(cs.closed_sample_count);
        init.agc = cs.agcavgfast;
    }
    TpFsm carryE = canon(init);
    float carryA = init.agc;
    bool all_ok = true;  // every segment so far accepted: carryE / carryA are the TRUE state
    uint32_t first_bad = a.seg1;
    int sum_open = 0, sum_flappy = 0, nbad = 0;

    for (uint32_t g0 = a.seg0; g0 < a.seg1; g0 += 64) {
        const uint32_t k = g0 + lane;
        const bool have = k < a.seg1;
        TpFsm S{}, E{};
        float s_agc = 0.f, e_agc = 0.f;
        int uses = 0, d_open = 0, d_flappy = 0;
        if (have) {
            S = rec_fsm(rec, st, base + k, 0);
            E = rec_fsm(rec, st, base + k, 7);
            s_agc = __int_as_float(rec[6 * st + base + k]);
            e_agc = __int_as_float(rec[13 * st + base + k]);
            uses = rec[14 * st + base + k];
            d_open = rec[15 * st + base + k];
            d_flappy = rec[16 * st + base + k];
        }
        // predecessor's end state
        TpFsm P;
        P.cur = __shfl_up(E.cur, 1), P.next = __shfl_up(E.next, 1), P.delay = __shfl_up(E.delay, 1);
        P.low = __shfl_up(E.low, 1), P.recent = __shfl_up(E.recent, 1), P.closed = __shfl_up(E.closed, 1);
        if (lane == 0)
            P = carryE;
        // agcavgfast before segment k = end value of the last earlier segment that touched it
        const unsigned long long umask = __ballot(have && uses);
        const unsigned long long below = umask & ((1ull << lane) - 1ull);
        const int src = below ? 63 - __clzll(below) : 0;
        const float a_from = __shfl(e_agc, src);
        const float A = below ? a_from : carryA;
        const bool ok = have && same(S, P) && (!uses || __float_as_int(s_agc) == __float_as_int(A));
        if (have) {
            // what a re-run of segment k has to start from (exact whenever every earlier segment was accepted)
            int* __restrict__ ts = a.tstart;
            const size_t ti = (base + k) * 8;
            ts[ti + 0] = P.cur, ts[ti + 1] = P.next, ts[ti + 2] = P.delay, ts[ti + 3] = P.low;
            ts[ti + 4] = P.recent, ts[ti + 5] = P.closed, ts[ti + 6] = __float_as_int(A);
            a.need[base + k] = ok ? 0 : 1;
        }
        const unsigned long long okmask = __ballot(ok);
        const unsigned long long havemask = __ballot(have);
        const unsigned long long bad = havemask & ~okmask;
        nbad += __popcll(bad);
        if (all_ok) {
            // counters only over the accepted prefix
            const int nacc = bad ? __ffsll(static_cast<long long>(bad)) - 1 : __popcll(havemask);
            int o = (lane < nacc) ? d_open : 0, f = (lane < nacc) ? d_flappy : 0;
            for (int off = 32; off > 0; off >>= 1) {
                o += __shfl_xor(o, off);
                f += __shfl_xor(f, off);
            }
            sum_open += o;
            sum_flappy += f;
            if (bad) {
                all_ok = false;
                first_bad = g0 + nacc;
            }
        }
        // carry to the next group
        const int lastl = __popcll(havemask) - 1;
        carryE.cur = __shfl(E.cur, lastl), carryE.next = __shfl(E.next, lastl), carryE.delay = __shfl(E.delay, lastl);
        carryE.low = __shfl(E.low, lastl), carryE.recent = __shfl(E.recent, lastl), carryE.closed = __shfl(E.closed, lastl);
        if (umask) {
            const int top = 63 - __clzll(umask);
            carryA = __shfl(e_agc, top);
        }
    }
    if (lane == 0) {
        TpFinal f;
        f.cur = carryE.cur, f.next = carryE.next, f.delay = carryE.delay, f.low = carryE.low, f.recent = carryE.recent, f.closed = carryE.closed;
        f.agc = carryA;
        f.all_ok = all_ok ? 1 : 0;
        f.first_bad = first_bad;
        f.d_open = sum_open;
        f.d_flappy = sum_flappy;
        a.fin[r] = f;
        if (a.diag) {  // unaccepted segments per scan round, summed over the chunks of the call
            int* d = a.diag + r * 4 + (diag_slot & 3);
            *d = (diag_assign ? 0 : *d) + nbad;
        }
    }
    first_bad_out = first_bad;
    return all_ok;
}

__global__ __launch_bounds__(64) void k_tp_scan(const TpArgs a) {
    const int r = blockIdx.x;
    uint32_t fb;
    (void)scan_row(a, r, a.rows[r], threadIdx.x, 0, a.first_chunk != 0, fb);
}

// =====================================================================================================
// D: re-run segments that were not accepted
// =====================================================================================================
__device__ __forceinline__ void lane_from_tstart(TpLane& s, const int* __restrict__ ts) {
    s.cur = ts[0], s.next = ts[1], s.delay = ts[2], s.low = ts[3], s.recent = ts[4], s.closed = ts[5];
    s.agc = __int_as_float(ts[6]);
}

// kStateOnly: walk the chain for its states alone (the cheap warm-up arithmetic, no audio): every member gets its true
// start state in tstart[] and need[] = 2, and k_tp_redo re-runs all of them side by side.  Otherwise each member is
// re-run in place, one after the other (the serial last resort).
template <bool kStateOnly>
__device__ __forceinline__ void rerun_chain(const TpArgs& a, const int r, const int row, const ChanParams& p, uint32_t k, const uint32_t max_chain,
                                            const bool stop_at_flagged, const bool to_the_end) {
    const float* __restrict__ magrow = a.mag + static_cast<size_t>(row) * a.plane_stride;
    const size_t base = static_cast<size_t>(r) * a.nseg;
    TpLane s;
    seg_reset(s);
    lane_from_tstart(s, a.tstart + (base + k) * 8);
    load_core(s, p, a.core[static_cast<size_t>(r) * (a.nseg + 1) + k]);
    for (uint32_t done = 0; k < a.seg1; ++k, ++done) {  // never beyond this chunk
        if (done > 0 && !to_the_end) {
            if (done >= max_chain)
                return;
            if (stop_at_flagged && a.need[base + k])
                return;  // that segment has its own lane
            // does the running state meet what segment k's lane recorded at its start?
            const TpFsm S = rec_fsm(a.rec, a.rec_stride, base + k, 0);
            const int uses = a.rec[14 * a.rec_stride + base + k];
            const int sagc = a.rec[6 * a.rec_stride + base + k];
            if (same(canon(s), S) && (!uses || sagc == __float_as_int(s.agc)))
                return;
        }
        const uint32_t s0 = k * a.L, s1 = min(s0 + a.L, a.nsteps);
        if (kStateOnly && done == 0 && a.redo_listed) {
            // The head of the chain starts from its true state here and now: run it for good (audio, record) instead of walking it
            // for its end state and leaving the re-run to k_tp_redo.  With a good guess of agcavgfast (k_tp_seg) a chain is this one
            // segment -- the one a transmission starts in -- and k_tp_redo finds its list empty.
            const TpFsm S = canon(s);
            const float s_agc = s.agc;
            seg_reset(s);
            CORE_PROF(const unsigned long long t_h = prof_now(); g_prof_unsettled = 0;)
            tp_run<true>(s, p, a, r, row, magrow, s0, s1, s0 / kWaveBatch, base + k);
            CORE_PROF(if (r == 0 && k < 200) printf("fix head r %d seg %u: %llu us, %d of 32 blocks through the sample path, start state %d end state %d\n", r, k, (prof_now() - t_h) / 100,
                                                   g_prof_unsettled, S.cur, s.cur);)
            rec_store(a, base + k, S, s_agc, s);
            a.need[base + k] = 0;
            seg_reset(s);
            continue;
        }
        if (kStateOnly) {
            int* __restrict__ ts = a.tstart + (base + k) * 8;
            ts[0] = s.cur, ts[1] = s.next, ts[2] = s.delay, ts[3] = s.low, ts[4] = s.recent, ts[5] = s.closed;
            ts[6] = __float_as_int(s.agc);
            a.need[base + k] = 2;
            if (a.redo_listed)
                a.redo[1 + atomicAdd(a.redo, 1)] = static_cast<int>(base + k);
            tp_run<false>(s, p, a, r, row, magrow, s0, s1, 0, base + k);
        } else {
            const TpFsm S = canon(s);
            const float s_agc = s.agc;
            seg_reset(s);
            tp_run<true>(s, p, a, r, row, magrow, s0, s1, s0 / kWaveBatch, base + k);
            rec_store(a, base + k, S, s_agc, s);
        }
    }
}

__global__ MI_LANE_BOUNDS void k_tp_fix(const TpArgs a) {
    const int gid = blockIdx.x * 64 + threadIdx.x;
    const int nsc = static_cast<int>(a.seg1 - a.seg0);
    if (gid >= a.nrows * nsc)
        return;
    const int r = gid / nsc;
    const uint32_t k = a.seg0 + static_cast<uint32_t>(gid - r * nsc);
    if (a.need[static_cast<size_t>(r) * a.nseg + k] != 1)
        return;
    const int row = a.rows[r];
    const ChanParams p = a.cp[row % a.nch];
    rerun_chain<true>(a, r, row, p, k, TP_MAXCHAIN, true, false);
}

// one segment from its true start state (tstart), audio and record included
__device__ __forceinline__ void redo_segment(const TpArgs& a, const int r, const int row, const ChanParams& p, const uint32_t k) {
    const size_t base = static_cast<size_t>(r) * a.nseg;
    const float* __restrict__ magrow = a.mag + static_cast<size_t>(row) * a.plane_stride;
    TpLane s;
    seg_reset(s);
    lane_from_tstart(s, a.tstart + (base + k) * 8);
    load_core(s, p, a.core[static_cast<size_t>(r) * (a.nseg + 1) + k]);
    const uint32_t s0 = k * a.L, s1 = min(s0 + a.L, a.nsteps);
    const TpFsm S = canon(s);
    const float s_agc = s.agc;
    seg_reset(s);
    tp_run<true>(s, p, a, r, row, magrow, s0, s1, s0 / kWaveBatch, base + k);
    rec_store(a, base + k, S, s_agc, s);
}

// every segment a chain of k_tp_fix passed through, from its true start state, side by side: one segment per wave (a lane that
// shares its wave with lanes at other points of their segments pays for their paths too), taken from the list the chains left
__global__ MI_LANE_BOUNDS void k_tp_redo(const TpArgs a) {
    const int count = a.redo[0];
    // blockDim.x segments per wave: 1 on plans of few rows (the re-runs are the critical path of the tail), 4 where hundreds of rows
    // leave thousands of them (throughput: the wave pays for every path its lanes take, but four at a time)
    for (int w = blockIdx.x * blockDim.x + threadIdx.x; w < count; w += gridDim.x * blockDim.x) {
        const int idx = a.redo[1 + w];
        const int r = idx / static_cast<int>(a.nseg);
        const uint32_t k = static_cast<uint32_t>(idx - r * static_cast<int>(a.nseg));
        const int row = a.rows[r];
        const ChanParams p = a.cp[row % a.nch];
        redo_segment(a, r, row, p, k);
    }
}

// Everything the first scan / fix / redo round left open, settled by the row's own wave without further launches (a launch
// costs tens of microseconds while the machine is busy with other passes, and the common case is "nothing left"): scan;
// while segments are unaccepted the lanes take one chain head each (states only), then one chain member each (audio), and
// scan again; after kSettleRounds the serial re-run from the first unaccepted segment.  What the lanes hand to each other
// goes through global memory, so every phase ends with an agent-scope fence (drains the stores, invalidates the L1).
constexpr int kSettleRounds = 6;
__global__ __launch_bounds__(64) void k_tp_settle(const TpArgs a) {
    const int r = blockIdx.x;
    const int row = a.rows[r];
    const int lane = threadIdx.x;
    const ChanParams p = a.cp[row % a.nch];
    const size_t base = static_cast<size_t>(r) * a.nseg;
    if (lane == 0 && a.first_chunk && a.diag)
        a.diag[r * 4 + 2] = a.diag[r * 4 + 3] = 0;
    uint32_t first_bad = a.seg1;
    bool ok = scan_row(a, r, row, lane, 1, a.first_chunk != 0, first_bad);
    for (int round = 0; !ok && round < kSettleRounds; ++round) {
        __threadfence();
        __syncthreads();
        for (uint32_t g0 = a.seg0; g0 < a.seg1; g0 += 64) {
            const uint32_t k = g0 + lane;
            if (k < a.seg1 && a.need[base + k] == 1)
                rerun_chain<true>(a, r, row, p, k, TP_MAXCHAIN, true, false);
        }
        __threadfence();
        __syncthreads();
        for (uint32_t g0 = a.seg0; g0 < a.seg1; g0 += 64) {
            const uint32_t k = g0 + lane;
            if (k < a.seg1 && a.need[base + k] == 2)
                redo_segment(a, r, row, p, k);
        }
        __threadfence();
        __syncthreads();
        ok = scan_row(a, r, row, lane, 2, false, first_bad);
    }
    if (!ok) {  // last resort: tstart of first_bad was written from an accepted predecessor, it is the true state
        __threadfence();
        __syncthreads();
        if (lane == 0)
            rerun_chain<false>(a, r, row, p, first_bad, 0xffffffffu, false, true);
        __threadfence();
        __syncthreads();
        (void)scan_row(a, r, row, lane, 3, false, first_bad);
    }
}

// =====================================================================================================
// E: deferred fades, axcindicate, carried state
// =====================================================================================================
// the deferred AM close-edge fades of one segment (rtl_airband.cpp:564-568), applied once all audio of the chunk is final
__device__ __forceinline__ void apply_fades(const TpArgs& a, const int row, const size_t idx) {
    const int nev = a.rec[18 * a.rec_stride + idx];
    float* __restrict__ wmain = a.wmain + static_cast<size_t>(row) * a.wmain_stride;
    float* __restrict__ carry = a.carry + static_cast<size_t>(row) * kAgcExtra;
    for (int e = 0; e < nev && e < TP_MAXEV; ++e) {
        const uint32_t i = static_cast<uint32_t>(a.rec[(19 + e) * a.rec_stride + idx]);
        // waveout[k] = waveout[k-1] * 0.94 for k = j-99 .. j-1, in virtual indices i+1 .. i+99
        float v = (i < a.nsteps) ? wmain[i] : carry[i - a.nsteps];
        for (int kk = 1; kk < kAgcExtra; ++kk) {
            v = v * 0.94f;
            const uint32_t vi = i + kk;
            if (vi < a.nsteps)
                wmain[vi] = v;
            else
                carry[vi - a.nsteps] = v;
        }
    }
}

__global__ __launch_bounds__(64) void k_tp_finish(const TpArgs a) {
    const int r = blockIdx.x;
    const int row = a.rows[r];
    const int lane = threadIdx.x;
    const size_t base = static_cast<size_t>(r) * a.nseg;
    for (uint32_t k = a.seg0 + lane; k < a.seg1; k += 64)  // (events are >= 197 steps apart: the fades of a row never overlap)
        apply_fades(a, row, base + k);
    // axcindicate per WAVE_BATCH from the segments' open masks
    int nopen = 0;
    for (uint32_t b = a.bat0 + lane; b < a.bat1; b += 64) {
        const uint32_t kmin = (b * kWaveBatch) / a.L, kmax = min((b * kWaveBatch + kWaveBatch - 1) / a.L, a.nseg - 1);
        bool open = false;
        for (uint32_t k = kmin; k <= kmax; ++k) {
            const int m = a.rec[17 * a.rec_stride + base + k];
            open |= ((m >> (b - (k * a.L) / kWaveBatch)) & 1) != 0;  // bit j of a segment's mask: its (j+1)-th batch
        }
        a.axc[static_cast<size_t>(row) * a.nbatches + b] = open ? MI_SIGNAL : MI_NO_SIGNAL;
        nopen += open ? 1 : 0;
    }
    for (int off = 32; off > 0; off >>= 1)
        nopen += __shfl_xor(nopen, off);
    // (the magnitude plane's last AGC_EXTRA samples, rtl_airband.cpp:643, are picked up where they are by the next call)
    if (lane != 0)
        return;
    const TpFinal f = a.fin[r];
    const TpCore t = a.core[static_cast<size_t>(r) * (a.nseg + 1) + (a.blk1 + a.L / 16 - 1) / (a.L / 16)];  // end of this chunk
    const ChanParams p = a.cp[row % a.nch];
    ChanState cs = a.st[row];
    cs.noise_floor = t.nf;
    cs.moving_avg_cap = t.cap;
    cs.pre_capped = t.c;
    cs.pre_full = t.full;
    cs.squelch_level_cache = 0.0f;  // a pure function of (recent_open_count_, noise_floor_): refilled on first use
    cs.current_state = f.cur;
    cs.next_state = f.next;
    cs.delay = f.delay;
    cs.low_signal_count = f.low;
    cs.recent_open_count = static_cast<uint32_t>(f.recent);
    cs.closed_sample_count = static_cast<uint32_t>(f.closed);
    cs.sample_count += a.step1 - a.step0;
    cs.buffer_head = static_cast<int32_t>((static_cast<uint32_t>(cs.buffer_head) + (a.step1 - a.step0)) % kSquelchRing);
    cs.buffer_tail = static_cast<int32_t>((static_cast<uint32_t>(cs.buffer_tail) + (a.step1 - a.step0)) % kSquelchRing);
    cs.open_count += static_cast<uint64_t>(f.d_open);
    cs.flappy_count += static_cast<uint64_t>(f.d_flappy);
    cs.agcavgfast = f.agc;
    cs.active_counter += static_cast<uint64_t>(nopen);
    a.st[row] = cs;
    if (a.stats) {
        mi_channel_stats s{};
        s.noise_level = cs.noise_floor;
        s.signal_level = cs.pre_full;
        s.squelch_level = level_of(p, cs.noise_floor, f.recent);
        s.agcavgfast = cs.agcavgfast;
        s.open_count = cs.open_count;
        s.flappy_count = cs.flappy_count;
        s.ctcss_count = s.no_ctcss_count = 0;
        s.active_counter = cs.active_counter;
        s.squelch_state = cs.current_state;
        s.signal_outside_filter = 0;
        a.stats[row] = s;
    }
}

// seed of the core chain from the carried ChanState (first call, after a serial call or a restored checkpoint)
__global__ void k_tp_prologue(const TpArgs a) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= a.nrows)
        return;
    const ChanState& cs = a.st[a.rows[gid]];
    TpCore t;
    t.nf = cs.noise_floor, t.cap = cs.moving_avg_cap, t.c = cs.pre_capped, t.full = cs.pre_full;
    a.core_carry[gid] = t;
}

// pre_filter_.full_ at the start of the call = the chain state before this call's first core launch (seeded just above, or
// left by the previous call's chain); k_tp_full reads it for its first lanes and as part of the sandwich's upper bound
__global__ void k_tp_full0(const TpArgs a) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid < a.nrows) {
        const float f = a.core_carry[gid].full;
        a.full0[gid] = f;
        a.fullbound[gid] = f;
    }
}

// When calls overlap the chain state of the previous call is not known yet: full_ at its end is bounded by the bound at its
// start and the largest sample it saw.
__global__ void k_tp_fullbound(const TpArgs a) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid < a.nrows)
        a.fullbound[gid] = fmaxf(a.fullbound[gid], __uint_as_float(a.xmax_prev[a.rows[gid]]));
}

// head of the emitted audio = lookahead of the previous call (output.cpp:948); runs on the caller's stream after the
// previous call's fades and before this call's
__global__ void k_tp_audio_head(const TpArgs a) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= a.nrows * kAgcExtra)
        return;
    const int r = gid / kAgcExtra, v = gid - r * kAgcExtra;
    const int row = a.rows[r];
    a.wmain[static_cast<size_t>(row) * a.wmain_stride + v] = a.carry_prev[static_cast<size_t>(row) * kAgcExtra + v];
}

// xmax[row] = max(xmax[row], largest of x[row][0 .. n)) on the bit patterns (the values are >= 0): what stage 1 leaves for k_tp_full when
// the planes come from somewhere else (mi_demod_process_planes)
__global__ __launch_bounds__(256) void k_row_max(const float* __restrict__ x, const size_t stride, const uint32_t n, unsigned* __restrict__ xmax) {
    const float* __restrict__ row = x + static_cast<size_t>(blockIdx.x) * stride;
    float m = 0.0f;
    for (uint32_t i = threadIdx.x; i < n; i += 256)
        m = fmaxf(m, row[i]);
    for (int o = 32; o > 0; o >>= 1)
        m = fmaxf(m, __shfl_down(m, o));
    if ((threadIdx.x & 63) == 0)
        atomicMax(&xmax[blockIdx.x], __float_as_uint(m));
}

}  // namespace

hipError_t launch_row_max(const float* x, size_t stride, uint32_t n, int rows, unsigned* xmax, hipStream_t s) {
    if (rows <= 0 || n == 0)
        return hipSuccess;
    hipLaunchKernelGGL(k_row_max, dim3(rows), dim3(256), 0, s, x, stride, n, xmax);
    return hipGetLastError();
}

#define TP_LAUNCH(kern, grid, block)                              \
    do {                                                          \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, s, a); \
        hipError_t e__ = hipGetLastError();                       \
        if (e__ != hipSuccess)                                    \
            return e__;                                           \
    } while (0)

hipError_t launch_tp_front(const TpArgs& a, hipStream_t s, bool seed_chain) {
    if (a.nrows == 0 || a.step1 <= a.step0)
        return hipSuccess;
    if (a.first_chunk && seed_chain) {
        TP_LAUNCH(k_tp_prologue, (a.nrows + 255) / 256, 256);
        TP_LAUNCH(k_tp_full0, (a.nrows + 255) / 256, 256);
    } else if (a.first_chunk) {
        TP_LAUNCH(k_tp_fullbound, (a.nrows + 255) / 256, 256);
    }
    const int lanes1 = a.nrows * static_cast<int>((a.step1 - a.step0 + a.L - 1) / a.L);
    TP_LAUNCH(k_tp_full, (lanes1 + 63) / 64, 64);
    return hipSuccess;
}

// on the caller's stream when a call starts: after the previous call's fades (they write the lookahead), before any
// segment pass of this call (its last segments write the next lookahead)
hipError_t launch_tp_audio_head(const TpArgs& a, hipStream_t s) {
    if (a.nrows == 0)
        return hipSuccess;
    TP_LAUNCH(k_tp_audio_head, (a.nrows * kAgcExtra + 255) / 256, 256);
    return hipSuccess;
}

hipError_t launch_tp_core(const TpArgs& a, hipStream_t s) {
    if (a.nrows == 0 || a.step1 <= a.step0)
        return hipSuccess;
    if (a.core_split)
        TP_LAUNCH(k_tp_core2, a.nrows, 512);
    else
        TP_LAUNCH(k_tp_core, a.nrows, 64);
    return hipSuccess;
}

#define TP_MARK(i)                                         \
    do {                                                   \
        if (marks) {                                       \
            hipError_t e__ = hipEventRecord(marks[i], s);  \
            if (e__ != hipSuccess)                         \
                return e__;                                \
        }                                                  \
    } while (0)

hipError_t launch_tp_seg(const TpArgs& a, hipStream_t s) {
    if (a.nrows == 0 || a.step1 <= a.step0)
        return hipSuccess;
    const int lanes = a.nrows * static_cast<int>(a.seg1 - a.seg0);
    // Lanes of one wave sit at different points of the capture, so a wave pays for every path one of its lanes takes
    // (a squelch edge costs ~9 us per block for the whole wave).  With a few rows there are far fewer lanes than the 1024
    // SIMDs x 64 of the machine and nothing else competes for them: spread them thin -- 4 per wave measured best at one stream
    // x 8 channels (2 waves per SIMD) -- and pack more only when the launch would exceed ~2 waves per SIMD.  With many rows the
    // pass runs beside the next call's stage 1, which wants three waves per SIMD of 167 registers: every wave of this pass
    // takes a third of a SIMD's register file out of its hands, so full waves (16 / 32 / 64 streams x 8 channels: +24 / +17 /
    // +13 % over 16 lanes per wave).  MI_OPT_TP_SEG_LANES overrides.
    int lpw = (a.seg_lpw >= 1 && a.seg_lpw <= 64) ? a.seg_lpw : 0;
    if (lpw == 0 && a.nrows > 32)
        lpw = 64;
    if (lpw == 0) {
        lpw = (lanes + 2047) / 2048;
        lpw = lpw < 4 ? 4 : (lpw > 64 ? 64 : lpw);
    }
    TP_LAUNCH(k_tp_seg, (lanes + lpw - 1) / lpw, lpw);
    return hipSuccess;
}

hipError_t launch_tp_rest(const TpArgs& a_in, hipStream_t s, hipEvent_t* marks) {
    TpArgs a = a_in;
    if (a.nrows == 0 || a.step1 <= a.step0)
        return hipSuccess;
    const int lanes = a.nrows * static_cast<int>(a.seg1 - a.seg0);
    TP_LAUNCH(k_tp_scan, a.nrows, 64);
    TP_MARK(0);
    {
        hipError_t e__ = hipMemsetAsync(a.redo, 0, sizeof(int), s);
        if (e__ != hipSuccess)
            return e__;
    }
    a.redo_listed = 1;
    TP_LAUNCH(k_tp_fix, (lanes + 63) / 64, 64);
    if (a.nrows <= 64)
        TP_LAUNCH(k_tp_redo, min(lanes, 2048), 1);
    else
        TP_LAUNCH(k_tp_redo, min((lanes + 3) / 4, 8192), 4);
    a.redo_listed = 0;  // (k_tp_settle's own rounds hand their members over through need[] alone)
    TP_MARK(1);
    TP_LAUNCH(k_tp_settle, a.nrows, 64);
    TP_LAUNCH(k_tp_finish, a.nrows, 64);
    TP_MARK(2);
    return hipSuccess;
}

}  // namespace mi
