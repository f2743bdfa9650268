// demod.hip -- stage 2 of the hot path on gfx950: the per-channel sample loop of demodulate()
// (rtl_airband.cpp:517-669) with Squelch (squelch.cpp), CTCSS (ctcss.cpp), NotchFilter/LowpassFilter
// (filters.cpp), the sincos LUT derotation (util.cpp:113-127), AM AGC and the NFM discriminators
// (rtl_airband.cpp:141-176).
//
// The loop is a recurrence in time (EMAs, a state machine, biquads, AGC with clip feedback): step i
// needs the state left by step i-1.  This kernel keeps one (stream, channel) per lane with its whole
// state in registers and walks the [stream][channel][time] planes stage 1 wrote; inputs are fetched a
// chunk ahead so the only serial chain is arithmetic.  Few lanes are packed per wave when there are
// few channels so that channels whose squelch states differ do not serialise each other's branches.
//
// Arithmetic is written operation for operation as the reference evaluates it in IEEE single precision
// (no contraction, correctly rounded / and sqrt): squelch decisions must be bit-exact.
// Compile with -ffp-contract=off.
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace mi {
namespace {

enum : int { SQ_CLOSED = 0, SQ_OPENING = 1, SQ_CLOSING = 2, SQ_LOW_SIGNAL_ABORT = 3, SQ_OPEN = 4 };  // squelch.h:104-110

constexpr int kOpenDelay = 197, kCloseDelay = 197, kLowSignalAbort = 88;  // squelch.cpp:49-51
constexpr uint32_t kRecentSampleSize = 1000, kFlapOpensThreshold = 3;    // squelch.cpp:62-63

// the handle row (stream * nch + channel) of the idx-th row of this launch
__device__ __forceinline__ int demod_row(const DemodArgs& a, const int idx) {
    return a.rows ? a.rows[idx] : idx;
}
__device__ __forceinline__ int demod_rows(const DemodArgs& a) {
    return a.rows ? a.nrows : a.nstreams * a.nch;
}

// ---- channel types ----
// A lone wave pays 10 / 24 cycles for a scalar branch (not taken / taken) and 21 / 35 for an exec-masked region (entered / skipped) --
// as much as 3 to 9 vector instructions (tools/micro/branch_cost.hip) -- and the per-channel loop is full of tests of the channel's
// configuration.  With one channel per wave the configuration is the same for the whole kernel, so the loop is compiled once per
// common type with those fields as constants (apply() overwrites the loaded ChanParams with what the dispatch has checked: every
// test of them folds away) and once with everything open.  A negative parameter leaves its field as loaded.
template <int kMod, int kRaw, int kIqo, int kLp, int kCtcss, int kNotch, int kManual>
struct ChanType {
    static __device__ __forceinline__ bool matches(const ChanParams& p) {
        return (kMod < 0 || p.modulation == kMod) && (kRaw < 0 || (p.needs_raw_iq != 0) == (kRaw != 0)) &&
               (kIqo < 0 || (p.has_iq_outputs != 0) == (kIqo != 0)) && (kLp < 0 || (p.lowpass_enabled != 0) == (kLp != 0)) &&
               (kCtcss < 0 || (p.ctcss_enabled != 0) == (kCtcss != 0)) && (kNotch < 0 || (p.notch_enabled != 0) == (kNotch != 0)) &&
               (kManual < 0 || (p.using_manual_level != 0) == (kManual != 0));
    }
    static __device__ __forceinline__ void apply(ChanParams& p) {
        if (kMod >= 0)
            p.modulation = kMod;
        if (kRaw >= 0)
            p.needs_raw_iq = kRaw;
        if (kIqo >= 0)
            p.has_iq_outputs = kIqo;
        if (kLp >= 0)
            p.lowpass_enabled = kLp;
        if (kCtcss >= 0)
            p.ctcss_enabled = kCtcss;
        if (kNotch >= 0)
            p.notch_enabled = kNotch;
        if (kManual >= 0)
            p.using_manual_level = kManual;
    }
    static constexpr int raw = kRaw;
    static constexpr bool am_plain = kMod == MI_MOD_AM && kRaw == 0 && kIqo == 0 && kLp == 0 && kCtcss == 0 && kNotch == 0 && kManual == 0;
};
using TyAny = ChanType<-1, -1, -1, -1, -1, -1, -1>;
using TyAmPlain = ChanType<MI_MOD_AM, 0, 0, 0, 0, 0, 0>;      // what tp.hip also takes: AM, nothing else
using TyNfmLp = ChanType<MI_MOD_NFM, 1, 0, 1, -1, -1, 0>;     // NFM with a low-pass filter (CTCSS / notch as configured)
using TyNfm = ChanType<MI_MOD_NFM, 1, 0, 0, -1, -1, 0>;       // NFM without

// ---- the audio wave's mailbox (k_demod_pw, NFM channels) ----
// The channel wave (FSM, filter) posts, in the order of the steps: the filtered I/Q of every step whose audio is processed
// (should_process_audio(): OPEN or CLOSING) as BLOCKs of 1..64 consecutive steps, the CTCSS resets (a transition to CLOSED),
// the ends of the WAVE_BATCHes, and QUIT.  The audio wave consumes them in that order, so what it computes is the serial loop's
// audio: it owns pr / pj / agcavgfast / prev_waveout, the notch filter, both CTCSS detector sets and their counters, the output
// gate, axcindicate and the audio / raw-I/Q stores of those steps.  Nothing flows back except the final state at QUIT.
constexpr unsigned kTokRing = 256;  // steps in flight (a power of two, >= 2 blocks)
constexpr unsigned kDescRing = 32;
// (plain AM channels: a step's token is (squelch level, wavein[j], wavein[j - AGC_EXTRA]) instead of the filtered I/Q, and the two AM
//  edges -- first_open_sample's bootstrap of agcavgfast and last_open_sample's fade, rtl_airband.cpp:554-569 -- are descriptors)
enum : unsigned { AUX_BLOCK = 1, AUX_RESET = 2, AUX_BATCH = 3, AUX_QUIT = 4, AUX_FIRST_OPEN = 5, AUX_LAST_OPEN = 6 };
struct AuxShare {
    float tk_re[kTokRing], tk_im[kTokRing], tk_ax[kTokRing];
    unsigned long long d_desc[kDescRing];  // low word: type | n << 8; high word: BLOCK: the first step's index, BATCH: the batch's index
    unsigned d_head;             // descriptors posted (channel wave)
    unsigned d_tail;             // descriptors consumed (audio wave)
    unsigned tk_tail;            // tokens consumed
    unsigned done;               // the audio wave has stored its state after QUIT
    unsigned fin[32];            // ... the part of ChanState it owns
};
typedef __attribute__((address_space(3))) AuxShare LdsAux;

struct Ctx {
    ChanState s;
    ChanParams p;  // by value: registers, not a global load per use
    float* ring;            // Squelch::buffer_
    const float* cc_fast;   // Goertzel coefficients
    const float* cc_slow;
    float* cq_fast;         // q1[kMaxTones], q2[kMaxTones]
    float* cq_slow;
    // One (stream, channel) per wave: every lane carries the same channel state and executes the same instructions (a
    // wave instruction costs the same for 1 or 64 active lanes), and the Goertzel banks are spread over the lanes --
    // lane d holds detector d of the fast and of the slow set (<= 52 each, ctcss.cpp:61-73).
    bool uni;
    int lane;
    float gf_c, gf_q1, gf_q2;  // fast set, detector `lane`
    float gs_c, gs_q1, gs_q2;  // slow set, detector `lane`
    // k_demod_pw, NFM channels: everything that hears the demodulated audio (discriminator, de-emphasis, CTCSS, gate, notch, the
    // output stores) belongs to the audio wave; this wave posts the filtered I/Q of the steps whose audio is processed (AuxShare)
    bool split;
    LdsAux* aux;
    unsigned tk_head, d_head;  // tokens / descriptors posted so far
    unsigned tk_tail_seen, d_tail_seen;  // ... and consumed, as last looked up (the rings are only looked at when these do not leave room)
    unsigned* timeouts;
};

__device__ __forceinline__ float lane_read(const float v, const int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

__device__ __forceinline__ float std_min(float a, float b) {  // std::min(a, b): b < a ? b : a
    return (b < a) ? b : a;
}

__device__ __forceinline__ bool flapping(const Ctx& c) {  // squelch.cpp:516-518
    return c.s.recent_open_count >= kFlapOpensThreshold;
}

__device__ __forceinline__ float squelch_level(Ctx& c) {  // squelch.cpp:164-177, cache and all
    if (c.p.using_manual_level)
        return c.p.manual_signal_level;
    if (c.s.squelch_level_cache == 0.0f) {
        if (flapping(c) && c.p.flappy_signal_ratio < c.p.normal_signal_ratio)
            c.s.squelch_level_cache = c.p.flappy_signal_ratio * c.s.noise_floor;
        else
            c.s.squelch_level_cache = c.p.normal_signal_ratio * c.s.noise_floor;
    }
    return c.s.squelch_level_cache;
}

__device__ __forceinline__ bool has_pre(Ctx& c) {  // squelch.cpp:462-464
    return c.s.pre_capped >= squelch_level(c);
}
__device__ __forceinline__ bool has_post(Ctx& c) {  // squelch.cpp:466-468
    return c.s.using_post_filter && c.s.post_capped >= c.ring[c.s.buffer_tail];
}
__device__ __forceinline__ bool has_signal(Ctx& c) {  // squelch.cpp:470-475
    if (c.s.using_post_filter)
        return has_pre(c) && has_post(c);
    return has_pre(c);
}

__device__ __forceinline__ void set_state(Ctx& c, int update) {  // squelch.cpp:297-361
    const int cur = c.s.current_state;
    if (cur == SQ_CLOSED && update == SQ_CLOSING)
        update = SQ_CLOSED;
    else if (cur == SQ_CLOSED && update == SQ_LOW_SIGNAL_ABORT)
        update = SQ_CLOSED;
    else if (cur == SQ_CLOSED && update == SQ_OPEN)
        update = SQ_OPENING;
    else if (cur == SQ_OPENING && update == SQ_LOW_SIGNAL_ABORT)
        update = SQ_CLOSED;
    else if (cur == SQ_LOW_SIGNAL_ABORT && update != SQ_LOW_SIGNAL_ABORT && update != SQ_CLOSED)
        update = SQ_CLOSED;
    else if (cur == SQ_OPEN && update == SQ_CLOSED)
        update = SQ_CLOSING;
    else if (cur == SQ_OPEN && update == SQ_OPENING)
        update = SQ_OPEN;
    c.s.next_state = update;
}

__device__ __forceinline__ void aux_post(Ctx& c, unsigned type, unsigned n, unsigned arg, float re, float im, float ax = 0.0f);
__device__ __forceinline__ void ctcss_reset(Ctx& c) {  // CTCSS::reset on both detectors, ctcss.cpp:165-172
    if (!c.p.ctcss_enabled)
        return;
    if (c.split) {  // the detectors live in the audio wave: the reset takes its place in the stream of its work
        aux_post(c, 2u /* AUX_RESET */, 0u, 0u, 0.0f, 0.0f);
        return;
    }
    if (c.uni) {
        c.gf_q1 = c.gf_q2 = c.gs_q1 = c.gs_q2 = 0.0f;
    } else {
        for (int d = 0; d < c.p.ctcss_fast_ndet; ++d)
            c.cq_fast[d] = c.cq_fast[kMaxTones + d] = 0.0f;
        for (int d = 0; d < c.p.ctcss_slow_ndet; ++d)
            c.cq_slow[d] = c.cq_slow[kMaxTones + d] = 0.0f;
    }
    c.s.cf_enough = c.s.cf_count = c.s.cf_has_tone = 0;
    c.s.cs_enough = c.s.cs_count = c.s.cs_has_tone = 0;
}

__device__ __forceinline__ void update_current_state(Ctx& c) {  // squelch.cpp:363-460
    ChanState& s = c.s;
    if (s.next_state == SQ_OPENING) {
        if (s.current_state != SQ_OPENING) {
            s.delay = 0;
            s.low_signal_count = 0;
            s.using_post_filter = 0;
            s.current_state = s.next_state;
        } else {
            s.delay++;
            if (s.delay >= kOpenDelay) {
                if (s.closed_sample_count < kRecentSampleSize) {
                    s.recent_open_count++;
                    if (flapping(c))
                        s.flappy_count++;
                    s.squelch_level_cache = 0.0f;
                }
                s.next_state = has_signal(c) ? SQ_OPEN : SQ_CLOSED;
            }
        }
    } else if (s.next_state == SQ_CLOSING) {
        if (s.current_state != SQ_CLOSING) {
            s.delay = 0;
            s.current_state = s.next_state;
        } else {
            s.delay++;
            if (s.delay >= kCloseDelay) {
                if (!has_signal(c)) {
                    s.next_state = SQ_CLOSED;
                } else {
                    s.current_state = SQ_OPEN;
                    s.next_state = SQ_OPEN;
                }
            }
        }
    } else if (s.next_state == SQ_LOW_SIGNAL_ABORT) {
        if (s.current_state != SQ_LOW_SIGNAL_ABORT) {
            if (s.current_state != SQ_CLOSING)
                s.delay = 0;
            s.current_state = s.next_state;
        } else {
            s.delay++;
            if (s.delay >= kCloseDelay)
                s.next_state = SQ_CLOSED;
        }
    } else if (s.next_state == SQ_OPEN && s.current_state != SQ_OPEN) {
        s.open_count++;
        s.current_state = s.next_state;
    } else if (s.next_state == SQ_CLOSED && s.current_state != SQ_CLOSED) {
        s.using_post_filter = 0;
        s.closed_sample_count = 0;
        s.current_state = s.next_state;
        ctcss_reset(c);
    } else if (s.next_state == SQ_CLOSED && s.current_state == SQ_CLOSED) {
        if (s.closed_sample_count < kRecentSampleSize) {
            s.closed_sample_count++;
        } else if (s.closed_sample_count == kRecentSampleSize) {
            s.recent_open_count = 0;
            s.squelch_level_cache = 0.0f;
        }
    } else {
        s.current_state = s.next_state;
    }
    s.buffer_tail = (s.buffer_tail + 1 == kSquelchRing) ? 0 : s.buffer_tail + 1;
    s.buffer_head = (s.buffer_head + 1 == kSquelchRing) ? 0 : s.buffer_head + 1;
}

__device__ __forceinline__ void update_avg(const float cap, float& full, float& capped, const float sample) {  // squelch.cpp:501-514
    const float decay_factor = 0.99f;
    const float new_factor = static_cast<float>(1.0 - static_cast<double>(0.99f));
    full = full * decay_factor + sample * new_factor;
    if (capped >= cap && sample >= cap)
        capped = cap;
    else
        capped = std_min(cap, capped * decay_factor + sample * new_factor);
}

__device__ __forceinline__ void process_raw(Ctx& c, const float sample) {  // squelch.cpp:195-246
    ChanState& s = c.s;
    update_current_state(c);
    s.sample_count++;
    if ((s.sample_count & 15u) == 0) {  // calculate_noise_floor, squelch.cpp:477-490
        const float decay_factor = 0.97f;
        const float new_factor = static_cast<float>(1.0 - static_cast<double>(0.97f));
        s.noise_floor = s.noise_floor * decay_factor + std_min(s.pre_capped, s.noise_floor) * new_factor + 1e-6f;
        s.moving_avg_cap = c.p.using_manual_level ? c.p.manual_cap : c.p.cap_factor * s.noise_floor;  // squelch.cpp:492-499
        s.squelch_level_cache = 0.0f;
    }
    update_avg(s.moving_avg_cap, s.pre_full, s.pre_capped, sample);
    if (c.p.lowpass_enabled)  // buffer_ is only ever read back through the post filter
        c.ring[s.buffer_head] = s.pre_capped * 0.9f;
    if (s.current_state == SQ_OPEN && !has_signal(c))
        set_state(c, SQ_CLOSING);
    if (s.current_state == SQ_CLOSED && has_signal(c))
        set_state(c, SQ_OPENING);
    if (s.current_state != SQ_CLOSED && s.current_state != SQ_LOW_SIGNAL_ABORT) {
        if (sample >= squelch_level(c)) {
            s.low_signal_count = 0;
        } else {
            s.low_signal_count++;
            if (s.low_signal_count >= kLowSignalAbort)
                set_state(c, SQ_LOW_SIGNAL_ABORT);
        }
    }
}

__device__ __forceinline__ bool should_filter(Ctx& c) {  // squelch.cpp:136-138
    return ((has_pre(c) || c.s.current_state != SQ_CLOSED) && c.s.current_state != SQ_LOW_SIGNAL_ABORT);
}

__device__ __forceinline__ void process_filtered(Ctx& c, const float sample) {  // squelch.cpp:248-276
    ChanState& s = c.s;
    if (!should_filter(c))
        return;
    if (s.current_state == SQ_OPENING) {
        if (s.delay < kSquelchRing)
            return;
        if (s.delay == kSquelchRing) {
            s.post_full = c.ring[s.buffer_tail];
            s.post_capped = c.ring[s.buffer_tail];
        }
    }
    s.using_post_filter = 1;
    update_avg(s.moving_avg_cap, s.post_full, s.post_capped, sample);
    if (s.post_capped < c.ring[s.buffer_tail])
        set_state(c, SQ_CLOSED);
}

// CTCSS::process_audio_sample on one detector set, ctcss.cpp:124-163 (+ ToneDetector, :44-55)
__device__ __forceinline__ void ctcss_process(const float* __restrict__ coeff, float* __restrict__ q, const int ndet, const int window, int& count,
                                              int& enough, int& has_tone, uint64_t& found, uint64_t& not_found, const float sample) {
    for (int d = 0; d < ndet; ++d) {
        const float q1 = q[d], q2 = q[kMaxTones + d];
        const float q0 = coeff[d] * q1 - q2 + sample;
        q[kMaxTones + d] = q1;
        q[d] = q0;
    }
    count++;
    if (count < window)
        return;
    enough = 1;
    float total = 0.0f, maxp = 0.0f, target = 0.0f;
    for (int d = 0; d < ndet; ++d) {
        const float q1 = q[d], q2 = q[kMaxTones + d];
        const float mag = q1 * q1 + q2 * q2 - q1 * q2 * coeff[d];
        total += mag;
        if (d == 0) {
            target = mag;
            maxp = mag;
        } else if (mag > maxp) {
            maxp = mag;
        }
        q[d] = 0.0f;
        q[kMaxTones + d] = 0.0f;
    }
    const float avg = total / static_cast<float>(ndet);
    if (target == maxp && target > avg) {
        has_tone = 1;
        found++;
    } else {
        has_tone = 0;
        not_found++;
    }
    count = 0;
}

// The same with one detector per lane: q0 = coeff*q1 - q2 + x in every lane at once; at the end of a window the powers
// are summed in detector order (the reference's `total += mag` order) by reading the lanes one after the other.
__device__ __forceinline__ void ctcss_process_lanes(const float coeff, float& q1, float& q2, const int ndet, const int window, int& count, int& enough,
                                                    int& has_tone, uint64_t& found, uint64_t& not_found, const float sample) {
    const float q0 = coeff * q1 - q2 + sample;
    q2 = q1;
    q1 = q0;
    count++;
    if (count < window)
        return;
    enough = 1;
    const float mag = q1 * q1 + q2 * q2 - q1 * q2 * coeff;
    float total = 0.0f, maxp = 0.0f, target = 0.0f;
    for (int d = 0; d < ndet; ++d) {
        const float m = lane_read(mag, d);
        total += m;
        if (d == 0) {
            target = m;
            maxp = m;
        } else if (m > maxp) {
            maxp = m;
        }
    }
    q1 = q2 = 0.0f;
    const float avg = total / static_cast<float>(ndet);
    if (target == maxp && target > avg) {
        has_tone = 1;
        found++;
    } else {
        has_tone = 0;
        not_found++;
    }
    count = 0;
}

__device__ __forceinline__ void process_audio(Ctx& c, const float sample) {  // squelch.cpp:278-295
    if (!c.p.ctcss_enabled)
        return;
    ChanState& s = c.s;
    if (c.uni) {
        if (s.current_state != SQ_CLOSED) {
            ctcss_process_lanes(c.gs_c, c.gs_q1, c.gs_q2, c.p.ctcss_slow_ndet, c.p.ctcss_slow_window, s.cs_count, s.cs_enough, s.cs_has_tone,
                                s.cs_found, s.cs_not_found, sample);
            if (!s.cs_enough)
                ctcss_process_lanes(c.gf_c, c.gf_q1, c.gf_q2, c.p.ctcss_fast_ndet, c.p.ctcss_fast_window, s.cf_count, s.cf_enough,
                                    s.cf_has_tone, s.cf_found, s.cf_not_found, sample);
        }
        return;
    }
    if (s.current_state != SQ_CLOSED) {
        ctcss_process(c.cc_slow, c.cq_slow, c.p.ctcss_slow_ndet, c.p.ctcss_slow_window, s.cs_count, s.cs_enough, s.cs_has_tone, s.cs_found,
                      s.cs_not_found, sample);
        if (!s.cs_enough)
            ctcss_process(c.cc_fast, c.cq_fast, c.p.ctcss_fast_ndet, c.p.ctcss_fast_window, s.cf_count, s.cf_enough, s.cf_has_tone, s.cf_found,
                          s.cf_not_found, sample);
    }
}

__device__ __forceinline__ bool is_open(const Ctx& c) {  // squelch.cpp:118-134
    if (c.s.current_state == SQ_OPEN || c.s.current_state == SQ_CLOSING) {
        if (c.p.ctcss_enabled)
            return c.s.cs_enough ? (c.s.cs_has_tone != 0) : (c.s.cf_has_tone != 0);
        return true;
    }
    return false;
}

__device__ __forceinline__ float fast_atan2(const float y, const float x) {  // rtl_airband.cpp:147-166
    const float pi4 = static_cast<float>(M_PI_4), pi34 = static_cast<float>(3 * M_PI_4);
    if (x == 0.0f && y == 0.0f)
        return 0;
    float yabs = y;
    if (yabs < 0.0f)
        yabs = -yabs;
    float angle;
    if (x >= 0.0f)
        angle = pi4 - pi4 * (x - yabs) / (x + yabs);
    else
        angle = pi34 - pi4 * (x + yabs) / (yabs - x);
    if (y < 0.0f)
        return -angle;
    return angle;
}

// ---------------------------------------------------------------------------------------------------------------------
// Steady-state blocks (one channel per wave).  While the squelch sits in CLOSED or in OPEN and no sample makes it want to
// leave, a step is the same straight-line arithmetic every time: up to 64 steps are then taken at once, step m in lane m.
// Whatever does not depend on the previous step (derotation, the FIR half of the biquads, magnitudes, discriminators,
// divisions, the output gate) is evaluated once across the lanes; every recurrence (the moving averages, the biquad
// feedback, AGC, de-emphasis) is a systolic chain: pass p shifts the chain registers one lane up (DPP wave_shr:1) and
// applies the step, so lane m holds the value after step m from pass m+1 on -- the same IEEE operations in the same order
// as the sample loop, on the same operands.  The first lane whose step would do anything else (pre/post filter crossing the
// squelch level, low-signal count reaching the abort, an AGC clip) ends the block: the steps before it (rounded down to a
// multiple of four, the I/O granule) are committed, and the sample loop takes over from the state they leave.
// lane j <- lane j-1; a lane whose source lane does not exist or is switched off keeps `keep`
__device__ __forceinline__ float shr1(const float v, const float keep) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(keep), __float_as_int(v), 0x138, 0xf, 0xf, false));
}

// n passes of a chain, sixteen per trip: a lane that has its final value keeps it under further passes, so rounding the
// count up is harmless.  (The compiler does not unroll loops around cross-lane operations by itself, and a taken branch
// costs a lone wave about as much as a whole pass.)
#define MI_PASSES4_(...) __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__
#define MI_PASSES(first, end, ...)                   \
    for (int p_ = (first); p_ < (end); p_ += 16) {   \
        MI_PASSES4_(MI_PASSES4_(__VA_ARGS__))        \
    }

typedef float v2f __attribute__((ext_vector_type(2)));

// Four passes of the pair of moving averages of Squelch::update_moving_avg (squelch.cpp:501-514), lane m = step m:
//   full   F[m] = F[m-1] * 0.99 + B[m]                                  (B = sample * (1 - 0.99))
//   capped C[m] = C[m-1] >= CAPX[m] ? CAP[m] : min(CAP[m], C[m-1] * 0.99 + B[m])   (CAPX = CAP where sample >= CAP, else +inf)
// T and P are only ever written through the shifted source, so lane 0, whose source does not exist, keeps what they were
// preset to: full_ * 0.99 and capped_ of the state the block starts from.  The order keeps two instructions between a
// write of F / C and its shifted read (the DPP hazard); the compare result goes through VCC, not an SGPR pair.
#define MI_EMA_DPP " wave_shr:1 row_mask:0xf bank_mask:0xf\n"
#define MI_EMA_PASS                               \
    "v_mul_f32_dpp %[T], %[F], %[K]" MI_EMA_DPP   \
    "v_add_f32 %[F], %[T], %[B]\n"                \
    "v_mov_b32_dpp %[P], %[C]" MI_EMA_DPP         \
    "v_mul_f32 %[V], %[K], %[P]\n"                \
    "v_cmp_ge_f32 vcc, %[P], %[CAPX]\n"           \
    "v_add_f32 %[V], %[V], %[B]\n"                \
    "v_min_f32 %[V], %[CAP], %[V]\n"              \
    "v_cndmask_b32 %[C], %[V], %[CAP], vcc\n"
#define MI_EMA_PASS4 MI_EMA_PASS MI_EMA_PASS MI_EMA_PASS MI_EMA_PASS
__device__ __forceinline__ void ema_passes16(float& F, float& C, float& T, float& P, const float B, const float CAP, const float CAPX) {
    const float K = 0.99f;
    float V;
    asm volatile("s_nop 1\n" MI_EMA_PASS4 MI_EMA_PASS4 MI_EMA_PASS4 MI_EMA_PASS4
                 : [F] "+v"(F), [C] "+v"(C), [T] "+v"(T), [P] "+v"(P), [V] "=&v"(V)
                 : [B] "v"(B), [CAP] "v"(CAP), [CAPX] "v"(CAPX), [K] "v"(K)
                 : "vcc");
}

// ... and of full_ alone: two instructions per step and the two wait states a DPP read of a fresh result needs
#define MI_FULL_PASS                              \
    "v_mul_f32_dpp %[T], %[F], %[K]" MI_EMA_DPP   \
    "v_add_f32 %[F], %[T], %[B]\n"                \
    "s_nop 1\n"
#define MI_FULL_PASS4 MI_FULL_PASS MI_FULL_PASS MI_FULL_PASS MI_FULL_PASS
__device__ __forceinline__ void full_passes16(float& F, float& T, const float B) {
    const float K = 0.99f;
    asm volatile("s_nop 1\n" MI_FULL_PASS4 MI_FULL_PASS4 MI_FULL_PASS4 MI_FULL_PASS4 : [F] "+v"(F), [T] "+v"(T) : [B] "v"(B), [K] "v"(K));
}

// Any chain of the form V[m] = V[m-1] * K + B[m], every product and sum rounded on its own (the AGC averages, the de-emphasis): n
// passes (rounded up to sixteen) of the same two instructions, lane m = step m, V[-1] = v_in.  Lane 0's shifted source does not
// exist, so it keeps reading the preset product v_in * K -- the very product its step forms.
__device__ __forceinline__ float chain_passes(const float v_in, const float B, const float K, const int n) {
    float V = 0.0f, T = v_in * K;
    for (int p_ = 0; p_ < n; p_ += 16)
        asm volatile("s_nop 1\n" MI_FULL_PASS4 MI_FULL_PASS4 MI_FULL_PASS4 MI_FULL_PASS4 : [F] "+v"(V), [T] "+v"(T) : [B] "v"(B), [K] "v"(K));
    return V;
}

// One noise-floor period [L, E) (at most 16 steps; lane m = step m) of the pre-filter averages (squelch.cpp:501-514) under the period's
// cap.  Fin / Cin: full_ / capped_ entering it (wave-uniform).  Two regimes need the recurrence of full_ only -- a third of the
// instructions of the pair:
//   MERGED     capped_ == full_ bit for bit on entry, below the cap, and full_ stays below it: the cap never binds and
//              `capped_ >= cap` never holds, so capped_ takes full_'s values (the same operations on the same operands);
//   SATURATED  every step leaves capped_ at the cap: either capped_ >= cap before it and the sample >= cap (squelch.cpp:509-510
//              assigns the cap), or the average it forms from them is >= cap (std::min returns the cap).  The first step of a
//              period usually goes the second way -- the cap has just risen with the noise floor.
// Both are verified for the whole period before anything is taken; otherwise the pair is walked as before.  Every lane runs
// every pass: lanes before L reproduce the values they have, whichever way they got them (all are the recurrence's values).
__device__ __forceinline__ void pre_period(float& F, float& C, float& T, float& Pc, const float b, const float x, const float cap, const float CAPv,
                                           float& CAPX, const int L, const int E, const int lane, const float Fin, const float Cin) {
    const bool mine = lane >= L;
    const bool in_period = mine && lane < E;
    CAPX = mine ? (x >= cap ? cap : __builtin_inff()) : CAPX;  // cap where sample >= cap, else +inf
    if (__float_as_uint(Cin) == __float_as_uint(Fin) && Cin < cap) {
        full_passes16(F, T, b);
        if (__ballot(in_period && !(F < cap)) == 0ull) {
            C = in_period ? F : C;
            return;
        }
    } else {
        const float Cp = lane == L ? Cin : cap;  // capped_ before the step, if every step before it left the cap
        const bool at_cap = (x >= cap && Cp >= cap) || (Cp * 0.99f + b >= cap);
        if (__ballot(in_period && !at_cap) == 0ull) {
            full_passes16(F, T, b);
            C = in_period ? cap : C;
            return;
        }
    }
    ema_passes16(F, C, T, Pc, b, CAPv, CAPX);  // a period is at most 16 steps; further passes change nothing
}

// The pre-filter averages and the noise floor (squelch.cpp:201-214, 477-514) over the steps [0, kmax) of a block, lane m = step m.
// In: the state entering the block (nf, cap, full, capd, sample_count_ sc) and the lane's sample x.  Out, per lane: full_ (F),
// capped_ (C), noise_floor_ (NFv) and moving_avg_cap_ (CAPv) after its step; nf / cap leave as they are after the block;
// zero_from = the first step that updates the noise floor (64: none).
// A period costs a lone wave more in bookkeeping than in arithmetic (~500 of ~800 cycles), so the two regimes of pre_period() are
// first tried for the block as a whole: full_ alone over all its steps, then the (at most five) noise-floor updates from the
// values capped_ would have at the period ends -- full_'s (MERGED) or the previous period's cap (SATURATED) -- and one check of
// every step's precondition under its own period's cap.  If it holds the values are the recurrence's; if not, nothing has been
// taken and the periods are walked one by one.
// kHaveF: full_ of the block's steps comes in F (the full_ wave of k_demod_pw walked it).
template <bool kHaveF>
__device__ __forceinline__ void pre_block(const ChanParams& P, float& nf, float& cap, const float full, const float capd, const uint32_t sc, const float x,
                                          const int kmax, const int lane, float& F, float& C, float& NFv, float& CAPv, int& zero_from) {
    const float k99 = 0.99f, n99 = static_cast<float>(1.0 - static_cast<double>(0.99f));
    const float k97 = 0.97f, n97 = static_cast<float>(1.0 - static_cast<double>(0.97f));
    const float b = x * n99;
    const int nb0 = __builtin_amdgcn_readfirstlane((15 - static_cast<int>(sc & 15u)) & 15);  // first step whose sample_count_ is a multiple of 16
    zero_from = nb0 < kmax ? nb0 : 64;
    const bool merged = __float_as_uint(capd) == __float_as_uint(full);
    if (merged || capd >= cap * 0.96875f) {  // (SATURATED: the cap has risen a little with the noise floor since capped_ was set to it)
        float Ft = kHaveF ? F : 0.0f, T = full * k99;  // T: what lane 0 keeps reading (its shifted source does not exist)
        if (!kHaveF) {
            for (int p_ = 0; p_ < kmax; p_ += 16)
                full_passes16(Ft, T, b);
        }
        float nf2 = nf, cap2 = cap, NF2 = nf, CAP2 = cap;
        for (int nb = nb0; nb < kmax; nb += 16) {  // calculate_noise_floor with capped_ as step nb-1 left it, squelch.cpp:477-490
            const float Cb = nb == 0 ? capd : (merged ? lane_read(Ft, nb - 1) : cap2);
            nf2 = nf2 * k97 + std_min(Cb, nf2) * n97 + 1e-6f;
            cap2 = P.using_manual_level ? P.manual_cap : P.cap_factor * nf2;
            const bool mine = lane >= nb;
            NF2 = mine ? nf2 : NF2;
            CAP2 = mine ? cap2 : CAP2;
        }
        bool bad;
        if (merged) {  // capped_ before and after the step stays below the step's cap: the cap never binds, capped_ >= cap never holds
            const float Fp = shr1(Ft, full);
            bad = !(Ft < CAP2 && Fp < CAP2);
        } else {  // the step leaves the cap: squelch.cpp:509-510 assigns it, or the average formed from capped_ and the sample reaches it
            const float Cp = shr1(CAP2, capd);
            bad = !((x >= CAP2 && Cp >= CAP2) || (Cp * k99 + b >= CAP2));
        }
        if (__ballot(bad && lane < kmax) == 0ull) {
            F = Ft, C = merged ? Ft : CAP2, NFv = NF2, CAPv = CAP2;
            nf = nf2, cap = cap2;
            return;
        }
    }
    // Every lane runs every pass with its own cap: the lanes before the current period have their final values and reproduce
    // them, the first lane of the period finds its predecessor's final value one lane down.
    F = 0.0f, C = 0.0f, NFv = nf, CAPv = cap;
    float T = full * k99, Pc = capd;  // what lane 0 keeps reading
    float CAPX = __builtin_inff();
    int L = 0, nb = nb0;
    while (L < kmax) {
        const float Fin = L ? lane_read(F, L - 1) : full, Cin = L ? lane_read(C, L - 1) : capd;
        if (L == nb) {  // calculate_noise_floor with the averages step L-1 left, squelch.cpp:477-490
            nf = nf * k97 + std_min(Cin, nf) * n97 + 1e-6f;
            cap = P.using_manual_level ? P.manual_cap : P.cap_factor * nf;
            nb += 16;
        }
        const int E = min(nb, kmax);
        const bool mine = lane >= L;
        NFv = mine ? nf : NFv;
        CAPv = mine ? cap : CAPv;
        pre_period(F, C, T, Pc, b, x, cap, CAPv, CAPX, L, E, lane, Fin, Cin);
        L = E;
    }
}

#ifdef MI_BLOCK_PROF
// (s_memtime is not ordered against the instructions around it and the compiler moves work across it: the scheduling barriers and the
//  wait for everything in flight pin a mark to its place -- at the price of the overlap they forbid, so the phases are upper bounds)
#define MI_PROF_MARK(k) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = __builtin_readcyclecounter(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); io.prof[k] += t_ - io.prof_t; io.prof_t = t_; } while (0)
#else
#define MI_PROF_MARK(k) do { } while (0)
#endif

// ---- the pre-filter wave (k_demod_pw) ----
// Of what a 64-step steady block costs, the pre-filter averages + noise floor of Squelch::process_raw_sample are the largest
// part (2 900 of 7 500 cycles on an AM channel, of 11 400 on NFM + low-pass + CTCSS): 64 dependent passes.  That recurrence
// (sample_count_, noise_floor_, moving_avg_cap_, pre_filter_.full_ / capped_: squelch.cpp:201-214, 477-514) takes the raw
// magnitudes and nothing else -- no squelch state, no filter, no audio.  So a second wave of the workgroup walks it over the
// whole call, 64 steps per trip, ahead of the channel wave, and leaves every step's values in a ring in LDS; the channel wave's
// steady blocks read their 64 lanes' values there instead of computing them (its sample loop keeps computing its own: the same
// numbers).  Nothing is speculative: the values are those of the serial recurrence.  The pre-filter wave reads a step's
// magnitude before the channel wave may overwrite it with the low-passed one (rtl_airband.cpp:548): the channel wave never
// passes the pre-filter wave.
constexpr unsigned kPreRing = 1024;  // steps
struct PreShare {
    float C[kPreRing], F[kPreRing], NF[kPreRing], CAP[kPreRing];  // after step i, at i % kPreRing
    unsigned h_done;  // the pre-filter wave has delivered every step below this one
    unsigned m_pos;   // the channel wave is at (or past) this step
    unsigned f_done;  // the full_ wave has delivered F of every step below this one
};
typedef __attribute__((address_space(3))) PreShare LdsPre;
typedef __attribute__((address_space(3))) volatile unsigned pre_vu32;
typedef __attribute__((address_space(3))) volatile float pre_vf32;
constexpr unsigned kPreSpin = 4u * 1000u * 1000u;
__device__ __forceinline__ unsigned pre_peek(const __attribute__((address_space(3))) unsigned* p) {
    return __builtin_amdgcn_readfirstlane(*(const pre_vu32*)p);
}
// channel wave: say where it is and wait until the pre-filter wave has delivered every step below `upto` (bounded)
__device__ __forceinline__ bool pre_wait(LdsPre* pre, const int lane, const uint32_t at, const uint32_t upto) {
    if (lane == 0)
        *(pre_vu32*)&pre->m_pos = at;
    for (unsigned spin = 0; pre_peek(&pre->h_done) < upto; ++spin) {
        __builtin_amdgcn_s_sleep(1);
        if (spin > kPreSpin) {
            return false;
        }
    }
    asm volatile("" ::: "memory");
    return true;
}

typedef __attribute__((address_space(3))) volatile unsigned aux_vu32;
typedef __attribute__((address_space(3))) volatile float aux_vf32;
typedef __attribute__((address_space(3))) volatile unsigned long long aux_vu64;
__device__ __forceinline__ unsigned aux_peek(const __attribute__((address_space(3))) unsigned* p) {
    return __builtin_amdgcn_readfirstlane(*(const aux_vu32*)p);
}
// channel wave: one descriptor, with n tokens (lane m < n holds step arg + m; n == 1 from the sample loop: wave-uniform values)
__device__ __forceinline__ void aux_post(Ctx& c, const unsigned type, const unsigned n, const unsigned arg, const float re, const float im, const float ax) {
    LdsAux* const x = c.aux;
    for (unsigned spin = 0;; ++spin) {  // room in both rings (the audio wave is the faster one: normally no wait, usually not even a look)
        if (c.tk_head + n - c.tk_tail_seen <= kTokRing && c.d_head + 1u - c.d_tail_seen <= kDescRing)
            break;
        c.tk_tail_seen = aux_peek(&x->tk_tail), c.d_tail_seen = aux_peek(&x->d_tail);
        if (c.tk_head + n - c.tk_tail_seen <= kTokRing && c.d_head + 1u - c.d_tail_seen <= kDescRing)
            break;
        __builtin_amdgcn_s_sleep(1);
        if (spin > 4u * kPreSpin) {  // (the audio wave is gone: counted, asserted 0 by the tests)
            if (c.lane == 0 && c.timeouts)
                atomicAdd(c.timeouts, 1u);
            break;
        }
    }
    if (static_cast<unsigned>(c.lane) < n) {
        const unsigned at = (c.tk_head + static_cast<unsigned>(c.lane)) & (kTokRing - 1u);
        *(aux_vf32*)&x->tk_re[at] = re;
        *(aux_vf32*)&x->tk_im[at] = im;
        if (c.p.modulation == MI_MOD_AM)
            *(aux_vf32*)&x->tk_ax[at] = ax;
    }
    if (c.lane == 0)
        *(aux_vu64*)&x->d_desc[c.d_head & (kDescRing - 1u)] = static_cast<unsigned long long>(type | (n << 8)) | (static_cast<unsigned long long>(arg) << 32);
    // a wave's LDS operations execute in order: the data before the mark (see pre_wave)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    c.tk_head += n;
    c.d_head += 1u;
    if (c.lane == 0)
        *(aux_vu32*)&x->d_head = c.d_head;
}

struct BlockIo {
#ifdef MI_BLOCK_PROF
    unsigned long long prof[8];
    unsigned long long prof_t;
    unsigned long long blocks, steps, after_ret;
#endif
    float* magrow;
    const float2* zrow;
    float* wmain;
    float* carry;
    float2* iqo;
    bool has_z, has_iqo;  // zrow / iqo are there (from the channel's type: constants in the typed instantiations)
    uint32_t n;
    // wavein of the 64 steps after the previous block, requested while that block ran (its first use is the block's first
    // operation; everything else a block loads is needed late enough to hide behind the pre-filter chain)
    float xpre;
    uint32_t xpre_i0;
};

// Returns the number of steps committed (a multiple of 4, possibly 0: then nothing was changed).  Call with current_state_
// == next_state_.  kmax in [4, 64], a multiple of 4, not past the end of the batch; the caller also keeps it short of the
// step that ends a CTCSS detector window and of the step whose delay_ count decides OPENING / CLOSING / LOW_SIGNAL_ABORT
// (and, while OPENING, the block does not straddle delay_ == buffer_size_, where the post filter starts).
//   CLOSED            averages + noise floor; ends when the pre-filter average reaches the level
//   OPENING           the same + low-signal count; raw-I/Q channels filter; post filter once delay_ >= buffer_size_
//   OPEN              + has_signal() must hold; audio
//   CLOSING           as OPEN without the has_signal() test (it is only asked when the delay runs out)
//   LOW_SIGNAL_ABORT  averages + noise floor only
// kSt: the state the block is in where the caller has branched on it (CLOSED, OPEN), else -1
template <bool kPre, int kSt>
__device__ __forceinline__ int steady_block(Ctx& c, const DemodArgs& a, BlockIo& io, const uint32_t i0_, const int kmax_, bool& batch_open, LdsPre* pre, bool& pre_on) {
    // wave-uniform by construction; say so, so that loop control stays on the scalar unit
    const uint32_t i0 = __builtin_amdgcn_readfirstlane(i0_);
    const int kmax = __builtin_amdgcn_readfirstlane(kmax_);
    ChanState& s = c.s;
    const ChanParams& P = c.p;
    const int lane = c.lane;
    const int st = kSt >= 0 ? kSt : s.current_state;
    const bool m_closed = st == SQ_CLOSED, m_open = st == SQ_OPEN, m_opening = st == SQ_OPENING, m_abort = st == SQ_LOW_SIGNAL_ABORT;
    const bool do_lsc = !m_closed && !m_abort;           // the low-signal count runs (squelch.cpp:233-245)
    const bool do_filter = do_lsc && io.has_z;  // should_filter_sample() holds on every step
    const bool lp = P.lowpass_enabled && io.has_z;
    const bool do_post = lp && do_lsc && (!m_opening || s.delay + 1 >= kSquelchRing);  // process_filtered_sample gets past its early return
    const bool post_init = do_post && m_opening && s.delay + 1 == kSquelchRing;         // ... and starts from buffer_[buffer_tail_]
    const bool do_audio = m_open || st == SQ_CLOSING;     // should_process_audio()
    const float k99 = 0.99f, n99 = static_cast<float>(1.0 - static_cast<double>(0.99f));
    const unsigned long long actmask = kmax >= 64 ? ~0ull : ((1ull << kmax) - 1ull);

#ifdef MI_BLOCK_PROF
    {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const unsigned long long t_ = __builtin_readcyclecounter();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (io.blocks)
            io.prof[7] += t_ - io.prof_t;  // since the previous block returned
        io.prof_t = t_;
    }
    io.blocks++;
#endif
    // ---- inputs of the block ----
    // wavein[j] was requested while the previous block ran (or is fetched now); its first use is the wait for it, and only
    // then are the other loads issued -- the next block's samples, and what this block needs after the pre-filter chain --
    // so that no wait in the chain has to cover a load that was issued a moment ago.
    float x = io.xpre;  // arrived before the previous block's stores were issued (see the end of this function)
    if (io.xpre_i0 != i0) {
        x = io.magrow[kAgcExtra + min(i0 + static_cast<uint32_t>(lane), io.n - 1u)];
        asm volatile("" : "+v"(x));  // the wait for this load stays inside the branch
    }
    __builtin_amdgcn_sched_barrier(0);
    {
        const uint32_t ni = i0 + static_cast<uint32_t>(kmax);
        const uint32_t idx = min(ni + static_cast<uint32_t>(lane), io.n - 1u);
        io.xpre = io.magrow[kAgcExtra + idx];
        io.xpre_i0 = ni;
    }
    const uint32_t li = min(i0 + static_cast<uint32_t>(lane), io.n - 1u);  // lanes past kmax read a valid address, nobody uses the value
    const float ax = io.magrow[li];  // wavein[j - AGC_EXTRA]
    // (an unconditional load: a value that is merged with a constant at a join is waited for at the join)
    const float2 z = (io.has_z ? io.zrow : reinterpret_cast<const float2*>(io.magrow))[io.has_z ? li : (li >> 1)];
    float rt = 0.0f;  // buffer_[buffer_tail_] as step m sees it
    if (do_post) {
        int t = s.buffer_tail + 1 + lane;
        t = t >= kSquelchRing ? t - kSquelchRing : t;
        rt = c.ring[t];
    }
    __builtin_amdgcn_sched_barrier(0);
    MI_PROF_MARK(6);

    // ---- Squelch::process_raw_sample: noise floor every 16th sample, pre-filter averages (squelch.cpp:195-246) ----
    float nf = s.noise_floor, cap = s.moving_avg_cap;
    float F = 0.0f, C = 0.0f, NFv = nf, CAPv = cap;
    int zero_from = 64;  // first step that cleared squelch_level_cache_
    bool from_ring = false;
    if constexpr (kPre) {
        // the pre-filter wave has walked these steps (and many more): take its values
        // (`pre` is an LDS address and may well be 0: never tested, the flag says whether the pre-filter wave is there)
        // (a wait that runs out is final: from then on this wave computes its own values and never looks at the ring again -- on a
        //  low-pass channel its blocks may since have overwritten magnitudes the late wave has yet to read, rtl_airband.cpp:548)
        if (pre_on && !pre_wait(pre, lane, i0, i0 + static_cast<uint32_t>(kmax))) {
            pre_on = false;
            if (lane == 0 && a.pre_timeouts)
                atomicAdd(a.pre_timeouts, 1u);
        }
        if (pre_on) {
            const unsigned at = (i0 + static_cast<uint32_t>(lane)) & (kPreRing - 1u);
            C = *(pre_vf32*)&pre->C[at], F = *(pre_vf32*)&pre->F[at], NFv = *(pre_vf32*)&pre->NF[at], CAPv = *(pre_vf32*)&pre->CAP[at];
            const int nb0 = __builtin_amdgcn_readfirstlane((15 - static_cast<int>(s.sample_count & 15u)) & 15);  // first step whose sample_count_ is a multiple of 16
            zero_from = nb0 < kmax ? nb0 : 64;
            from_ring = true;
        }
    }
    if (!from_ring) {
        int zf = 64;
        pre_block<false>(P, nf, cap, s.pre_full, s.pre_capped, s.sample_count, x, kmax, lane, F, C, NFv, CAPv, zf);
        zero_from = min(zero_from, zf);
    }
    MI_PROF_MARK(0);
    // squelch_level() as step m evaluates it (cache and all, squelch.cpp:164-177)
    uint32_t recent = s.recent_open_count;
    int ev_lane = 64;  // CLOSED: first step that finds closed_sample_count_ == recent_sample_size_ (squelch.cpp:442-449)
    if (m_closed) {
        const int togo = static_cast<int>(kRecentSampleSize) - static_cast<int>(s.closed_sample_count);
        ev_lane = togo < 0 ? 0 : (togo > 64 ? 64 : togo);
        zero_from = min(zero_from, ev_lane);
        if (lane >= ev_lane)
            recent = 0;
    }
    float level;
    if (P.using_manual_level) {
        level = P.manual_signal_level;
    } else {
        const float ratio = (recent >= kFlapOpensThreshold && P.flappy_signal_ratio < P.normal_signal_ratio) ? P.flappy_signal_ratio : P.normal_signal_ratio;
        level = (lane < zero_from && s.squelch_level_cache != 0.0f) ? s.squelch_level_cache : ratio * NFv;
    }
    // Does a step evaluate squelch_level() at all?  Everywhere except in LOW_SIGNAL_ABORT on a channel without raw I/Q, where
    // the sample loop never asks should_filter_sample(): there the cache stays cleared after a noise-floor update.
    const bool level_used = !(m_abort && !io.has_z);
    const bool has_pre = C >= level;

    bool fail = m_closed ? has_pre : (m_open ? !has_pre : false);
    int lsc = 0;  // low_signal_count_ after step m
    if (do_lsc) {
        const unsigned long long ge = __ballot(x >= level);
        const unsigned long long below = ge & ((2ull << lane) - 1ull);
        lsc = below ? lane - (63 - __builtin_clzll(below)) : s.low_signal_count + lane + 1;
        fail = fail || lsc >= kLowSignalAbort;
    }

    MI_PROF_MARK(1);
    // derotation, low-pass, magnitude (rtl_airband.cpp:532-552)
    float re = 0.0f, im = 0.0f, xf = x;
    float xr = 0.0f, xi = 0.0f;
    float PF = 0.0f, PC = 0.0f;
    if (do_filter) {
        const uint32_t phi = (s.dm_phi + static_cast<uint32_t>(lane) * P.dm_dphi) & 0xffffffu;
        const uint32_t idx = phi >> 16;  // sincosf_lut, util.cpp:113-127
        const float fract = static_cast<float>(phi & 0xffff) / 65536.0f;
        float v1 = a.sin_lut[idx], v2 = a.sin_lut[idx + 1];
        const float swf = v1 + (v2 - v1) * fract;
        v1 = a.cos_lut[idx];
        v2 = a.cos_lut[idx + 1];
        const float cwf = v1 + (v2 - v1) * fract;
        const float nswf = -swf;
        float re_tmp = z.x * cwf - z.y * nswf;
        float im_tmp = z.y * cwf + z.x * nswf;
        if (P.lowpass_enabled) {  // LowpassFilter::apply, filters.cpp:146-163
            xr = re_tmp / P.lowpass_gain;
            xi = im_tmp / P.lowpass_gain;
            const float xr1 = shr1(xr, s.lp_xr[2]), xi1 = shr1(xi, s.lp_xi[2]);
            const float xr2 = shr1(xr1, s.lp_xr[1]), xi2 = shr1(xi1, s.lp_xi[1]);
            // (re, im) pairs: the feedback is two packed multiplies and two packed adds per pass
            const v2f A = {(xr2 + xr) + (2.0f * xr1), (xi2 + xi) + (2.0f * xi1)};
            const v2f c0 = {P.lowpass_yc0, P.lowpass_yc0}, c1 = {P.lowpass_yc1, P.lowpass_yc1};
            v2f Y = {0.0f, 0.0f};
            v2f Y1 = {s.lp_yr[2], s.lp_yi[2]}, Y2 = {s.lp_yr[1], s.lp_yi[1]};
            MI_PASSES(0, kmax, {
                Y2.x = shr1(Y1.x, Y2.x);
                Y2.y = shr1(Y1.y, Y2.y);
                Y1.x = shr1(Y.x, Y1.x);
                Y1.y = shr1(Y.y, Y1.y);
                Y = (A + c0 * Y2) + c1 * Y1;
            })
            const float Yr = Y.x, Yi = Y.y;
            re_tmp = Yr;
            im_tmp = Yi;
        }
        re = re_tmp;
        im = im_tmp;
        xf = sqrtf(re * re + im * im);
        MI_PROF_MARK(2);
        if (do_post) {  // Squelch::process_filtered_sample, squelch.cpp:248-276
            const float b2 = xf * n99;
            const float capx = xf >= CAPv ? CAPv : __builtin_inff();
            const float rt0 = lane_read(rt, 0);
            float PT = (post_init ? rt0 : s.post_full) * k99, PCp = post_init ? rt0 : s.post_capped;
            for (int p_ = 0; p_ < kmax; p_ += 16)
                ema_passes16(PF, PC, PT, PCp, b2, CAPv, capx);
            if (m_open)
                fail = fail || !(PCp >= rt);  // has_post_filter_signal() in process_raw_sample of this step
            fail = fail || PC < rt;           // this step would ask for CLOSED
        }
    }

    MI_PROF_MARK(3);
    // audio (rtl_airband.cpp:571-609)
    float d = 0.0f;  // the sample handed to process_audio_sample
    float G = 0.0f;  // agcavgfast after step m
    const bool own_audio = do_audio && !c.split;  // (split: the audio wave's, from the filtered I/Q posted at the commit)
    if (own_audio) {
        if (P.modulation == MI_MOD_AM) {
            const bool upd = xf > level;
            const float bA = xf * 0.005f;
            if ((__ballot(!upd) & actmask) == 0ull) {  // every sample above the level: the plain average
                G = chain_passes(s.agcavgfast, bA, 0.995f, kmax);
            } else {
                float Gp = s.agcavgfast;
                MI_PASSES(0, kmax, {
                    Gp = shr1(G, Gp);
                    G = upd ? Gp * 0.995f + bA : Gp;
                })
            }
            d = (ax - G) / (G * 1.5f);
            fail = fail || fabsf(d) > 0.8f;  // the clip feeds back into the AGC: the sample loop takes that step
        } else {
            const float pr = shr1(re, s.pr), pj = shr1(im, s.pj);
            float w;
            if (!a.fm_quadri) {
                const float nbj = -pj;  // polar_disc_fast: multiply(ar, aj, br, -bj)
                const float cr = re * pr - im * nbj;
                const float cj = im * pr + re * nbj;
                w = static_cast<float>(static_cast<double>(fast_atan2(cj, cr)) * M_1_PI);
            } else {
                w = static_cast<float>(static_cast<double>((pr * im - re * pj) / (re * re + im * im + 1.0f)) * M_1_PI);
            }
            const float bN = w * 0.005f;
            G = chain_passes(s.agcavgfast, bN, 0.995f, kmax);
            const float e = (w - G) * P.one_minus_alpha;
            d = chain_passes(s.prev_waveout, e, P.alpha, kmax);  // (e + prev * alpha: the product, then the sum)
        }
    }

    MI_PROF_MARK(4);
    // The next block's samples have long arrived: take them out of the memory counter now, before the stores below enter it
    // (the counter is in order -- a wait for this load at the top of the next block would also wait for those stores).
    asm volatile("" : "+v"(io.xpre));
    const unsigned long long failm = __ballot(fail) & actmask;
    const int k = failm ? __builtin_ctzll(failm) : kmax;
    const int kc = k & ~3;
    if (kc == 0)
        return 0;
    const int last = kc - 1;
#ifdef MI_BLOCK_PROF
    io.steps += kc;
#endif

    // output gate (rtl_airband.cpp:612-641); is_open() cannot change inside the block (no detector window ends in it)
    const bool gate = own_audio && (!P.ctcss_enabled || (s.cs_enough ? (s.cs_has_tone != 0) : (s.cf_has_tone != 0)));
    float out = 0.0f;
    if (gate) {
        out = d;
        if (P.notch_enabled) {  // NotchFilter::apply, filters.cpp:50-64
            const float u1 = shr1(d, s.notch_x[2]);
            const float u2 = shr1(u1, s.notch_x[1]);
            const float B = P.notch_d0 * d - P.notch_d1 * u1 + P.notch_d0 * u2;
            float Y = 0.0f, Y1 = s.notch_y[2], Y2 = s.notch_y[1];
            MI_PASSES(0, kc, {
                Y2 = shr1(Y1, Y2);
                Y1 = shr1(Y, Y1);
                Y = B + P.notch_d1 * Y1 - P.notch_d2 * Y2;
            })
            out = Y;
            s.notch_x[0] = lane_read(d, last - 2);
            s.notch_x[1] = lane_read(d, last - 1);
            s.notch_x[2] = lane_read(d, last);
            s.notch_y[0] = lane_read(Y, last - 2);
            s.notch_y[1] = lane_read(Y, last - 1);
            s.notch_y[2] = lane_read(Y, last);
        }
        out *= P.ampfactor;
        if (out != out)
            out = 0.0f;
        else if (out > 1.0f)
            out = 1.0f;
        else if (out < -1.0f)
            out = -1.0f;
        batch_open = true;
    }

    // ---- commit the first kc steps ----
    if (lane < kc) {
        if (do_filter)
            io.magrow[kAgcExtra + i0 + lane] = xf;  // channel->wavein[j] is overwritten (rtl_airband.cpp:548)
        if (P.lowpass_enabled) {
            int h = s.buffer_head + 1 + lane;
            h = h >= kSquelchRing ? h - kSquelchRing : h;
            c.ring[h] = C * 0.9f;
        }
        if (!(do_audio && c.split)) {
            const uint32_t v = kAgcExtra + i0 + lane;
            float* dst = (v < io.n) ? io.wmain + v : io.carry + (v - io.n);
            *dst = out;
            if (io.has_iqo)
                io.iqo[i0 + lane] = gate ? make_float2(re, im) : make_float2(0.0f, 0.0f);
        }
    }
    if (do_audio && c.split) {
        if (P.modulation == MI_MOD_AM)
            aux_post(c, AUX_BLOCK, static_cast<unsigned>(kc), i0, level, xf, ax);
        else
            aux_post(c, AUX_BLOCK, static_cast<unsigned>(kc), i0, re, im);
    }
    if (own_audio && P.ctcss_enabled) {  // Squelch::process_audio_sample -> CTCSS::process_audio_sample (ctcss.cpp:124-135)
        const bool fast_too = !s.cs_enough;
        for (int m = 0; m < kc; m += 4) {  // kc is a multiple of 4
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float smp = lane_read(d, m + u);
                const float q0 = c.gs_c * c.gs_q1 - c.gs_q2 + smp;
                c.gs_q2 = c.gs_q1;
                c.gs_q1 = q0;
                if (fast_too) {
                    const float f0 = c.gf_c * c.gf_q1 - c.gf_q2 + smp;
                    c.gf_q2 = c.gf_q1;
                    c.gf_q1 = f0;
                }
            }
        }
        s.cs_count += kc;
        if (!s.cs_enough)
            s.cf_count += kc;
    }
    s.noise_floor = lane_read(NFv, last);
    s.moving_avg_cap = lane_read(CAPv, last);
    s.pre_full = lane_read(F, last);
    s.pre_capped = lane_read(C, last);
    if (P.using_manual_level || !level_used) {
        if (zero_from <= last)
            s.squelch_level_cache = 0.0f;
    } else {
        s.squelch_level_cache = lane_read(level, last);
    }
    s.sample_count += static_cast<uint32_t>(kc);
    s.buffer_head = s.buffer_head + kc >= kSquelchRing ? s.buffer_head + kc - kSquelchRing : s.buffer_head + kc;
    s.buffer_tail = s.buffer_tail + kc >= kSquelchRing ? s.buffer_tail + kc - kSquelchRing : s.buffer_tail + kc;
    if (m_closed) {
        if (ev_lane <= last)
            s.recent_open_count = 0;
        s.closed_sample_count = min(s.closed_sample_count + static_cast<uint32_t>(kc), kRecentSampleSize);
    } else if (!m_open) {
        s.delay += kc;
    }
    if (do_lsc)
        s.low_signal_count = __builtin_amdgcn_readlane(lsc, last);
    if (do_filter) {
        s.dm_phi = (s.dm_phi + static_cast<uint32_t>(kc) * P.dm_dphi) & 0xffffffu;
        if (P.lowpass_enabled) {
            s.lp_xr[0] = lane_read(xr, last - 2), s.lp_xi[0] = lane_read(xi, last - 2);
            s.lp_xr[1] = lane_read(xr, last - 1), s.lp_xi[1] = lane_read(xi, last - 1);
            s.lp_xr[2] = lane_read(xr, last), s.lp_xi[2] = lane_read(xi, last);
            s.lp_yr[0] = lane_read(re, last - 2), s.lp_yi[0] = lane_read(im, last - 2);
            s.lp_yr[1] = lane_read(re, last - 1), s.lp_yi[1] = lane_read(im, last - 1);
            s.lp_yr[2] = lane_read(re, last), s.lp_yi[2] = lane_read(im, last);
        }
        if (do_post) {
            s.post_full = lane_read(PF, last);
            s.post_capped = lane_read(PC, last);
            s.using_post_filter = 1;
        }
    }
    if (own_audio) {
        s.agcavgfast = lane_read(G, last);
        if (P.modulation != MI_MOD_AM) {
            s.pr = lane_read(re, last);
            s.pj = lane_read(im, last);
            s.prev_waveout = lane_read(d, last);
        }
    }
    MI_PROF_MARK(5);
    return kc;
}

// the end of a WAVE_BATCH: axcindicate and active_counter (rtl_airband.cpp:523,628,667-669) -- the audio wave's where it judges is_open()
#define MI_END_BATCH()                                                                                                 \
    do {                                                                                                               \
        if (c.split) {                                                                                                 \
            aux_post(c, AUX_BATCH, 0u, batch, 0.0f, 0.0f);                                                             \
        } else {                                                                                                       \
            a.axc[static_cast<size_t>(row) * a.axc_stride + batch] = batch_open ? MI_SIGNAL : MI_NO_SIGNAL;           \
            if (batch_open)                                                                                            \
                c.s.active_counter++;                                                                                  \
        }                                                                                                              \
        batch_open = false;                                                                                            \
        in_batch = 0;                                                                                                  \
        batch++;                                                                                                       \
    } while (0)


// The idle channel (k_demod_pw): CLOSED for recent_sample_size_ samples and more with no recent opening.  A step then moves nothing but
// the pre-filter averages and the noise floor -- the pre-filter wave's work -- the squelch ring and the counters: update_current_state
// clears the level cache at every step (squelch.cpp:442-449: closed_sample_count_ == recent_sample_size_), so the level of a step is
// normal_signal_ratio_ x its own noise floor, and the step stays CLOSED iff capped_ < level (has_signal() is has_pre_filter_signal():
// using_post_filter_ is off in CLOSED).  So a block is: two ring values per lane, one compare, the stores (zeros, the squelch ring of
// a low-pass channel), and the state after its last step read from the ring -- no sample is loaded at all.  Same decisions and the
// same state as steady_block<kPre, SQ_CLOSED> leaves, which takes over (with the sample loop) as soon as a step wants to open.
__device__ __forceinline__ void idle_streak(Ctx& c, const DemodArgs& a, BlockIo& bio, uint32_t& gi, const uint32_t ngroups, uint32_t& in_batch, uint32_t& batch,
                                            bool& batch_open, int& skip, bool& stale, LdsPre* pre, bool& pre_on, const int row) {
    ChanState& s = c.s;
    const ChanParams& P = c.p;
    const int lane = c.lane;
    // Inside the loop only what the next block needs is kept up (position, batch, the squelch ring's head); the rest of the state --
    // the averages, the noise floor, the level cache, sample_count_ -- is read from the ring once, when the run ends.
    uint32_t end = 0;           // steps [.., end) have been committed by this run (0: none)
    int head = s.buffer_head;   // (buffer_tail_ follows at the same distance: advanced by the same count at the end)
    const uint32_t i_first = gi * 4;
    const int phase0 = static_cast<int>((__builtin_amdgcn_readfirstlane(s.sample_count) + 1u) & 15u);
    for (;;) {
        int kmax = min(64, static_cast<int>(ngroups - gi) * 4);
        kmax = min(kmax, kWaveBatch - static_cast<int>(in_batch));  // the batch flag is written at a batch's last step
        const uint32_t i0 = gi * 4;
        const int phase = (phase0 + static_cast<int>(i0 - i_first)) & 15;
        if (phase && kmax == 64)
            kmax = 64 - phase;  // end on a multiple of 16 of sample_count_, like every steady block
        kmax &= ~3;
        if (kmax < 8)
            break;  // the sample loop takes this group
#ifdef MI_BLOCK_PROF
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long ti0 = __builtin_readcyclecounter();
        __builtin_amdgcn_sched_barrier(0);
#endif
        if (!pre_wait(pre, lane, i0, i0 + static_cast<uint32_t>(kmax))) {
            pre_on = false;  // (final: see steady_block)
            if (lane == 0 && a.pre_timeouts)
                atomicAdd(a.pre_timeouts, 1u);
            break;
        }
#ifdef MI_BLOCK_PROF
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long ti1 = __builtin_readcyclecounter();
        __builtin_amdgcn_sched_barrier(0);
        bio.prof[0] += ti1 - ti0;            // idle blocks: waiting for the pre-filter wave
        if (bio.prof_t)
            bio.prof[7] += ti0 - bio.prof_t;  // ... everything else of a block
        bio.prof_t = ti1;
        bio.blocks++;
#endif
        const unsigned at = (i0 + static_cast<uint32_t>(lane)) & (kPreRing - 1u);
        const float C = *(pre_vf32*)&pre->C[at], NFv = *(pre_vf32*)&pre->NF[at];
        const float level = P.using_manual_level ? P.manual_signal_level : P.normal_signal_ratio * NFv;
        const unsigned long long actmask = kmax >= 64 ? ~0ull : ((1ull << kmax) - 1ull);
        const unsigned long long failm = __ballot(C >= level) & actmask;
        const int k = failm ? static_cast<int>(__builtin_ctzll(failm)) : kmax;
        const int kc = k & ~3;
        if (kc == 0) {
            skip = 4;
            break;
        }
        if (lane < kc) {
            if (P.lowpass_enabled) {
                int h = head + 1 + lane;
                h = h >= kSquelchRing ? h - kSquelchRing : h;
                c.ring[h] = C * 0.9f;
            }
            const uint32_t v = kAgcExtra + i0 + static_cast<uint32_t>(lane);
            float* dst = (v < bio.n) ? bio.wmain + v : bio.carry + (v - bio.n);
            *dst = 0.0f;
            if (bio.has_iqo)
                bio.iqo[i0 + static_cast<uint32_t>(lane)] = make_float2(0.0f, 0.0f);
        }
        head = head + kc >= kSquelchRing ? head + kc - kSquelchRing : head + kc;
        end = i0 + static_cast<uint32_t>(kc);
        in_batch += static_cast<uint32_t>(kc);
        if (in_batch == kWaveBatch)
            MI_END_BATCH();
        gi += static_cast<uint32_t>(kc / 4);
        if (kc != kmax) {
            skip = 4;
            break;
        }
        if (gi >= ngroups)
            break;
    }
    if (end != 0) {  // the state after the last committed step
        const unsigned atl = (end - 1u) & (kPreRing - 1u);
        s.noise_floor = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(*(pre_vf32*)&pre->NF[atl])));
        s.moving_avg_cap = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(*(pre_vf32*)&pre->CAP[atl])));
        s.pre_full = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(*(pre_vf32*)&pre->F[atl])));
        s.pre_capped = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(*(pre_vf32*)&pre->C[atl])));
        s.squelch_level_cache = P.using_manual_level ? 0.0f : P.normal_signal_ratio * s.noise_floor;
        const uint32_t done = end - i_first;
        s.sample_count += done;
        s.buffer_head = head;
        s.buffer_tail = static_cast<int32_t>((static_cast<uint32_t>(s.buffer_tail) + done) % kSquelchRing);
        stale = true;  // the group fetched ahead is behind us now
    }
}

// The open plain AM channel with a pre-filter wave (k_demod_pw, k_demod_pw2): what is left for the channel's wave in OPEN is has_signal() --
// capped_ >= level, from the ring -- and the low-signal count (squelch.cpp:233-245), and then either handing (level, wavein[j],
// wavein[j - 100]) to the audio wave (kSplit) or the AM AGC and audio themselves (rtl_airband.cpp:574-585, 612-641).  As in idle_streak()
// nothing else of the channel is touched inside the loop.  Blocks start on a multiple of 16 of sample_count_ -- the first noise-floor
// update of the block is its first step, so the level cache is cleared there and every step's level is ratio x its own noise floor --
// else the general block takes one and aligns.  Same decisions, tokens / audio and state as steady_block<true, SQ_OPEN>.
template <bool kSplit>
__device__ __forceinline__ void open_streak_am(Ctx& c, const DemodArgs& a, BlockIo& bio, uint32_t& gi, const uint32_t ngroups, uint32_t& in_batch, uint32_t& batch,
                                               bool& batch_open, int& skip, bool& stale, LdsPre* pre, bool& pre_on, const int row) {
    ChanState& s = c.s;
    const ChanParams& P = c.p;
    const int lane = c.lane;
    const float ratio = (s.recent_open_count >= kFlapOpensThreshold && P.flappy_signal_ratio < P.normal_signal_ratio) ? P.flappy_signal_ratio : P.normal_signal_ratio;
    const float* __restrict__ xrow = bio.magrow + kAgcExtra;
    float xn = 0.0f, axn = 0.0f;  // the next block's samples, requested a block ahead
    uint32_t n_at = 0xffffffffu;
    for (;;) {
        int kmax = min(64, static_cast<int>(ngroups - gi) * 4);
        kmax = min(kmax, kWaveBatch - static_cast<int>(in_batch));  // the batch flag is written at a batch's last step
        kmax &= ~3;
        if (kmax < 8 || ((__builtin_amdgcn_readfirstlane(s.sample_count) + 1u) & 15u) != 0u)
            return;  // the general block (it aligns) or the sample loop
        const uint32_t i0 = gi * 4;
        const uint32_t li = min(i0 + static_cast<uint32_t>(lane), bio.n - 1u);
        float x = xn, ax = axn;
        if (n_at != i0)
            x = xrow[li], ax = bio.magrow[li];
        {
            const uint32_t ni = i0 + static_cast<uint32_t>(kmax);
            const uint32_t idx = min(ni + static_cast<uint32_t>(lane), bio.n - 1u);
            xn = xrow[idx], axn = bio.magrow[idx];
            n_at = ni;
        }
        if (!pre_wait(pre, lane, i0, i0 + static_cast<uint32_t>(kmax))) {
            pre_on = false;  // (final: see steady_block)
            if (lane == 0 && a.pre_timeouts)
                atomicAdd(a.pre_timeouts, 1u);
            return;
        }
        const unsigned at = (i0 + static_cast<uint32_t>(lane)) & (kPreRing - 1u);
        const float C = *(pre_vf32*)&pre->C[at], NFv = *(pre_vf32*)&pre->NF[at];
        const float level = ratio * NFv;
        asm volatile("" : "+v"(x), "+v"(ax));  // (the samples have arrived: out of the memory counter before any store below enters it)
        const unsigned long long actmask = kmax >= 64 ? ~0ull : ((1ull << kmax) - 1ull);
        const unsigned long long ge = __ballot(x >= level);
        const unsigned long long below = ge & ((2ull << lane) - 1ull);
        const int lsc = below ? lane - (63 - static_cast<int>(__builtin_clzll(below))) : s.low_signal_count + lane + 1;
        bool fail = !(C >= level) || lsc >= kLowSignalAbort;
        float G = 0.0f, out = 0.0f;
        if (!kSplit) {  // no audio wave (k_demod_pw2): AM AGC and audio here
            const bool upd = x > level;
            const float bA = x * 0.005f;
            if ((__ballot(!upd) & actmask) == 0ull) {  // every sample above the level: the plain average
                G = chain_passes(s.agcavgfast, bA, 0.995f, kmax);
            } else {
                float Gp = s.agcavgfast;
                MI_PASSES(0, kmax, {
                    Gp = shr1(G, Gp);
                    G = upd ? Gp * 0.995f + bA : Gp;
                })
            }
            const float d = (ax - G) / (G * 1.5f);
            fail = fail || fabsf(d) > 0.8f;  // the clip feeds back into the AGC: the sample loop takes that step
            out = d * P.ampfactor;
            if (out != out)
                out = 0.0f;
            else if (out > 1.0f)
                out = 1.0f;
            else if (out < -1.0f)
                out = -1.0f;
        }
        const unsigned long long failm = __ballot(fail) & actmask;
        const int k = failm ? static_cast<int>(__builtin_ctzll(failm)) : kmax;
        const int kc = k & ~3;
        if (kc == 0) {
            skip = 4;
            return;
        }
        const int last = kc - 1;
        if (kSplit) {
            aux_post(c, AUX_BLOCK, static_cast<unsigned>(kc), i0, level, x, ax);
        } else {
            if (lane < kc) {
                const uint32_t v = kAgcExtra + i0 + static_cast<uint32_t>(lane);
                float* dst = (v < bio.n) ? bio.wmain + v : bio.carry + (v - bio.n);
                *dst = out;
            }
            batch_open = true;
            s.agcavgfast = lane_read(G, last);
        }
        // the state after step `last`
        const unsigned atl = (i0 + static_cast<uint32_t>(last)) & (kPreRing - 1u);
        s.noise_floor = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(*(pre_vf32*)&pre->NF[atl])));
        s.moving_avg_cap = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(*(pre_vf32*)&pre->CAP[atl])));
        s.pre_full = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(*(pre_vf32*)&pre->F[atl])));
        s.pre_capped = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(*(pre_vf32*)&pre->C[atl])));
        s.squelch_level_cache = ratio * s.noise_floor;
        s.low_signal_count = __builtin_amdgcn_readlane(lsc, last);
        s.sample_count += static_cast<uint32_t>(kc);
        s.buffer_head = s.buffer_head + kc >= kSquelchRing ? s.buffer_head + kc - kSquelchRing : s.buffer_head + kc;
        s.buffer_tail = s.buffer_tail + kc >= kSquelchRing ? s.buffer_tail + kc - kSquelchRing : s.buffer_tail + kc;
        in_batch += static_cast<uint32_t>(kc);
        if (in_batch == kWaveBatch)
            MI_END_BATCH();
        gi += static_cast<uint32_t>(kc / 4);
        stale = true;  // the group fetched ahead is behind us now
        if (kc != kmax) {
            skip = 4;
            return;
        }
        if (gi >= ngroups)
            return;
    }
}

// The plain AM channel without helper waves (k_demod_uni: plans of hundreds of rows and more), idle or open: a lean run in the manner of
// idle_streak() with the pre-filter block (pre_block<false>) and, when open, the AM AGC and audio (rtl_airband.cpp:574-585, 612-641) in
// the loop.  The state a block hands to the next is a handful of locals; the channel's struct is written once, when the run ends.
// kOpen: the channel is OPEN (else CLOSED for recent_sample_size_ samples, no recent opening).  Blocks of an open run start on a
// multiple of 16 of sample_count_ (every step's level is then ratio x its own noise floor).  A step that wants anything else -- the
// squelch level crossed, the low-signal count at its limit, an AGC clip -- ends the run.
template <bool kOpen>
__device__ __forceinline__ void am_plain_run(Ctx& c, const DemodArgs& a, BlockIo& bio, uint32_t& gi, const uint32_t ngroups, uint32_t& in_batch, uint32_t& batch,
                                             bool& batch_open, int& skip, bool& stale, const int row) {
    ChanState& s = c.s;
    const ChanParams& P = c.p;
    const int lane = c.lane;
    const float ratio = (kOpen && s.recent_open_count >= kFlapOpensThreshold && P.flappy_signal_ratio < P.normal_signal_ratio) ? P.flappy_signal_ratio
                                                                                                                              : P.normal_signal_ratio;
    const float* __restrict__ xrow = bio.magrow + kAgcExtra;
    float nf = s.noise_floor, cap = s.moving_avg_cap, full = s.pre_full, capd = s.pre_capped, agc = s.agcavgfast;
    uint32_t sc = __builtin_amdgcn_readfirstlane(s.sample_count);
    int lowc = s.low_signal_count;
    uint32_t done = 0;
    float xn = 0.0f, axn = 0.0f;  // the next block's samples, requested a block ahead
    uint32_t n_at = 0xffffffffu;
    for (;;) {
        int kmax = min(64, static_cast<int>(ngroups - gi) * 4);
        kmax = min(kmax, kWaveBatch - static_cast<int>(in_batch));  // the batch flag is written at a batch's last step
        const int phase = static_cast<int>((sc + 1u) & 15u);
        if (kOpen) {
            if (phase != 0)
                break;  // (the general block aligns)
        } else if (phase && kmax == 64) {
            kmax = 64 - phase;
        }
        kmax &= ~3;
        if (kmax < 8)
            break;
        const uint32_t i0 = gi * 4;
        const uint32_t li = min(i0 + static_cast<uint32_t>(lane), bio.n - 1u);
        float x = xn, ax = axn;
        if (n_at != i0) {
            x = xrow[li];
            if (kOpen)
                ax = bio.magrow[li];
        }
        {
            const uint32_t ni = i0 + static_cast<uint32_t>(kmax);
            const uint32_t idx = min(ni + static_cast<uint32_t>(lane), bio.n - 1u);
            xn = xrow[idx];
            if (kOpen)
                axn = bio.magrow[idx];
            n_at = ni;
        }
        float F, C, NFv, CAPv;
        int zf;
        float nf_b = nf, cap_b = cap;
        pre_block<false>(P, nf_b, cap_b, full, capd, sc, x, kmax, lane, F, C, NFv, CAPv, zf);
        const float level = ratio * NFv;
        const unsigned long long actmask = kmax >= 64 ? ~0ull : ((1ull << kmax) - 1ull);
        bool fail;
        int lsc = 0;
        float G = 0.0f, out = 0.0f;
        if (kOpen) {
            const unsigned long long ge = __ballot(x >= level);
            const unsigned long long below = ge & ((2ull << lane) - 1ull);
            lsc = below ? lane - (63 - static_cast<int>(__builtin_clzll(below))) : lowc + lane + 1;
            fail = !(C >= level) || lsc >= kLowSignalAbort;
            const bool upd = x > level;
            const float bA = x * 0.005f;
            if ((__ballot(!upd) & actmask) == 0ull) {
                G = chain_passes(agc, bA, 0.995f, kmax);
            } else {
                float Gp = agc;
                MI_PASSES(0, kmax, {
                    Gp = shr1(G, Gp);
                    G = upd ? Gp * 0.995f + bA : Gp;
                })
            }
            const float d = (ax - G) / (G * 1.5f);
            fail = fail || fabsf(d) > 0.8f;  // the clip feeds back into the AGC: the sample loop takes that step
            out = d * P.ampfactor;
            if (out != out)
                out = 0.0f;
            else if (out > 1.0f)
                out = 1.0f;
            else if (out < -1.0f)
                out = -1.0f;
        } else {
            fail = C >= level;
        }
        asm volatile("" : "+v"(xn));  // (the next block's samples have arrived: out of the memory counter before the stores below enter it)
        const unsigned long long failm = __ballot(fail) & actmask;
        const int k = failm ? static_cast<int>(__builtin_ctzll(failm)) : kmax;
        const int kc = k & ~3;
#ifdef MI_LEAN_DEBUG
        if (kOpen && row == 0 && gi < 3000 && kc != kmax && lane == k)
            printf("  block i0 %u kmax %d fails at lane %d: C %g level %g lsc %d d-clip %d x %g\n", i0, kmax, k, C, level, lsc, (int)(fabsf((ax - G) / (G * 1.5f)) > 0.8f), x);
#endif
        if (kc == 0) {
            skip = 4;
            break;
        }
        const int last = kc - 1;
        if (lane < kc) {
            const uint32_t v = kAgcExtra + i0 + static_cast<uint32_t>(lane);
            float* dst = (v < bio.n) ? bio.wmain + v : bio.carry + (v - bio.n);
            *dst = out;
        }
        if (kOpen)
            batch_open = true;
        nf = lane_read(NFv, last), cap = lane_read(CAPv, last), full = lane_read(F, last), capd = lane_read(C, last);
        if (kOpen) {
            agc = lane_read(G, last);
            lowc = __builtin_amdgcn_readlane(lsc, last);
        }
        sc += static_cast<uint32_t>(kc);
        done += static_cast<uint32_t>(kc);
        in_batch += static_cast<uint32_t>(kc);
        if (in_batch == kWaveBatch)
            MI_END_BATCH();
        gi += static_cast<uint32_t>(kc / 4);
        if (kc != kmax) {
            skip = 4;
            break;
        }
        if (gi >= ngroups)
            break;
    }
#ifdef MI_LEAN_DEBUG
    if (kOpen && row == 0 && lane == 0 && gi < 3000)
        printf("lean open run: gi %u done %u sc %u in_batch %u skip %d\n", gi, done, sc, in_batch, skip);
#endif
    if (done != 0) {
        s.noise_floor = nf, s.moving_avg_cap = cap, s.pre_full = full, s.pre_capped = capd;
        s.squelch_level_cache = ratio * nf;
        s.sample_count = sc;
        s.buffer_head = static_cast<int32_t>((static_cast<uint32_t>(s.buffer_head) + done) % kSquelchRing);
        s.buffer_tail = static_cast<int32_t>((static_cast<uint32_t>(s.buffer_tail) + done) % kSquelchRing);
        if (kOpen) {
            s.agcavgfast = agc;
            s.low_signal_count = lowc;
        }
        stale = true;  // the group fetched ahead is behind us now
    }
}

// A run of steady blocks in one state (current_state_ == next_state_ == kSt, or whatever it is for kSt < 0): a block never changes
// the state, so while blocks commit in full nothing but their lengths has to be worked out between them -- the batch end, the
// 16-step phase of sample_count_, and in the waiting states the step whose delay_ decides.  Returns with gi at the first group no
// block took (possibly ngroups); `skip` says how many groups the sample loop should take before blocks are tried again.
// lean_next: a leaner run exists for this state once sample_count_ sits on a multiple of 16: hand back after the block that aligns it.
template <bool kPre, int kSt>
__device__ __forceinline__ void steady_streak(Ctx& c, const DemodArgs& a, BlockIo& bio, uint32_t& gi, const uint32_t ngroups, uint32_t& in_batch,
                                              uint32_t& batch, bool& batch_open, int& skip, bool& stale, LdsPre* pre, bool& pre_on, const int row,
                                              const bool lean_next = false, bool* again = nullptr) {
    const ChanParams& P = c.p;
    const int st = kSt >= 0 ? kSt : __builtin_amdgcn_readfirstlane(c.s.current_state);
    const bool lpz = P.lowpass_enabled && bio.has_z;
    for (;;) {
        int kmax = min(64, static_cast<int>(ngroups - gi) * 4);
        kmax = min(kmax, kWaveBatch - static_cast<int>(in_batch));  // the batch flag is written at a batch's last step
        // end on a multiple of 16 of sample_count_: the blocks after this one then see whole noise-floor periods
        const int phase = static_cast<int>((__builtin_amdgcn_readfirstlane(c.s.sample_count) + 1u) & 15u);
        if (phase && kmax == 64)
            kmax = 64 - phase;
        bool ok = true;
        if (st == SQ_OPENING || st == SQ_CLOSING || st == SQ_LOW_SIGNAL_ABORT) {
            const int delay = __builtin_amdgcn_readfirstlane(c.s.delay);
            // the step whose delay_ reaches open_delay_ / close_delay_ decides: the sample loop takes it
            kmax = min(kmax, (st == SQ_OPENING ? kOpenDelay : kCloseDelay) - 1 - delay);
            if (st == SQ_OPENING && lpz) {
                const int upf = __builtin_amdgcn_readfirstlane(c.s.using_post_filter);
                if (delay + 1 < kSquelchRing) {  // process_filtered_sample still returns early ...
                    kmax = min(kmax, kSquelchRing - 1 - delay);
                    ok = upf == 0;
                } else {  // ... or runs on every step (it starts from buffer_[buffer_tail_] at delay_ == buffer_size_)
                    ok = upf == (delay + 1 == kSquelchRing ? 0 : 1);
                }
            }
        }
        if (st == SQ_OPEN || st == SQ_CLOSING) {
            ok = ok && __builtin_amdgcn_readfirstlane(c.s.using_post_filter) == (lpz ? 1 : 0);
            if (P.ctcss_enabled && !c.split) {  // a detector window's last sample is taken by the sample loop
                kmax = min(kmax, P.ctcss_slow_window - 1 - __builtin_amdgcn_readfirstlane(c.s.cs_count));
                if (!c.s.cs_enough)
                    kmax = min(kmax, P.ctcss_fast_window - 1 - __builtin_amdgcn_readfirstlane(c.s.cf_count));
            }
        }
        if (!ok)
            kmax = 0;
        kmax &= ~3;
        if (kmax < 8)
            return;  // the sample loop takes this group
        const int kc = steady_block<kPre, kSt>(c, a, bio, gi * 4, kmax, batch_open, pre, pre_on);
        if (kc > 0) {
            in_batch += static_cast<uint32_t>(kc);
            if (in_batch == kWaveBatch)
                MI_END_BATCH();
            gi += static_cast<uint32_t>(kc / 4);
            stale = true;  // the group fetched ahead is behind us now
        }
#ifdef MI_BLOCK_PROF
        {   // split "between blocks": bookkeeping after the return vs loop head + eligibility
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            const unsigned long long t_ = __builtin_readcyclecounter();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            bio.after_ret += t_ - bio.prof_t;
            bio.prof_t = t_;
        }
#endif
        if (kc != kmax) {  // a step of the block wants something else: it is among the next four
            skip = 4;
            return;
        }
        if (gi >= ngroups)
            return;
        if (lean_next && ((__builtin_amdgcn_readfirstlane(c.s.sample_count) + 1u) & 15u) == 0u) {
            *again = true;  // (nothing failed: the caller looks at the state again instead of taking a group through the sample loop)
            return;
        }
    }
}

// kUni: one channel per wave -- all 64 lanes run it in lockstep on the same values.  It is its own instantiation so that
// the compiler's uniformity analysis sees a row that depends on blockIdx alone: the channel state then sits in scalar
// registers where it can, the state machine's integer work runs on the scalar unit and its branches are scalar branches.
template <bool kUni, bool kPre, class T>
__device__ __forceinline__ void demod_body(const DemodArgs& a, LdsPre* pre, LdsAux* aux) {
    bool pre_on = kPre;  // the pre-filter wave is there and delivering (k_demod_pw)
    const int rows = demod_rows(a);
    constexpr bool uni = kUni;
    const int ridx = kUni ? static_cast<int>(blockIdx.x) : static_cast<int>(blockIdx.x) * a.lanes_per_wave + static_cast<int>(threadIdx.x);
    if ((!kUni && static_cast<int>(threadIdx.x) >= a.lanes_per_wave) || ridx >= rows)
        return;
    const int row = demod_row(a, ridx);
    const int stream = row / a.nch, ch = row - stream * a.nch;
    Ctx c;
    c.uni = uni;
    c.lane = static_cast<int>(threadIdx.x);
    c.gf_c = c.gf_q1 = c.gf_q2 = c.gs_c = c.gs_q1 = c.gs_q2 = 0.0f;
    c.s = a.st[row];
    c.p = a.cp[ch];
    T::apply(c.p);
    const ChanParams& P = c.p;
    c.split = kPre && a.audio_wave && (P.modulation != MI_MOD_AM || TyAmPlain::matches(P));  // (the audio wave decides the same way)
    c.aux = aux;
    c.tk_head = c.d_head = 0;
    c.tk_tail_seen = c.d_tail_seen = 0;
    c.timeouts = a.pre_timeouts;
    c.ring = a.sq_ring + static_cast<size_t>(row) * kSquelchRing;
    c.cc_fast = c.cc_slow = nullptr;
    c.cq_fast = c.cq_slow = nullptr;
    if (P.ctcss_enabled) {
        c.cc_fast = a.ctcss_coeff + static_cast<size_t>(P.ctcss_row) * 2 * kMaxTones;
        c.cc_slow = c.cc_fast + kMaxTones;
        c.cq_fast = a.ctcss_q + (static_cast<size_t>(stream) * a.n_ctcss_rows + P.ctcss_row) * 4 * kMaxTones;
        c.cq_slow = c.cq_fast + 2 * kMaxTones;
        if (uni && !c.split) {  // detector `lane` of each set lives in this lane's registers for the whole call
            if (c.lane < P.ctcss_fast_ndet)
                c.gf_c = c.cc_fast[c.lane], c.gf_q1 = c.cq_fast[c.lane], c.gf_q2 = c.cq_fast[kMaxTones + c.lane];
            if (c.lane < P.ctcss_slow_ndet)
                c.gs_c = c.cc_slow[c.lane], c.gs_q1 = c.cq_slow[c.lane], c.gs_q2 = c.cq_slow[kMaxTones + c.lane];
        }
    }

    float* __restrict__ magrow = a.mag + static_cast<size_t>(row) * a.plane_stride;
    const float2* zrow = P.needs_raw_iq ? a.cplx + (static_cast<size_t>(stream) * a.n_iq_rows + P.iq_row) * a.plane_stride : nullptr;
    float* __restrict__ wmain = a.wmain + static_cast<size_t>(row) * a.wmain_stride;
    float* __restrict__ carry = a.carry + static_cast<size_t>(row) * kAgcExtra;
    const float* __restrict__ carry_in = (a.carry_in ? a.carry_in : a.carry) + static_cast<size_t>(row) * kAgcExtra;
    float2* __restrict__ iqo = (a.iq_out && P.has_iq_outputs) ? a.iq_out + static_cast<size_t>(row) * a.iq_out_stride : nullptr;
    const bool has_z = P.needs_raw_iq != 0, has_iqo = iqo != nullptr;
    const uint32_t n = a.nsteps;
    // virtual waveout: index v in [0, n) is emitted audio, [n, n+AGC_EXTRA) is the lookahead kept in `carry`
    auto W = [&](uint32_t v) -> float& { return v < n ? wmain[v] : carry[v - n]; };

    // the lookahead of the previous call is the head of this call's emitted audio (output.cpp:948)
    for (int v = 0; v < kAgcExtra; ++v)
        wmain[v] = carry_in[v];

    const bool am = P.modulation == MI_MOD_AM;
    const float ampfactor = P.ampfactor;
    bool batch_open = false;
    uint32_t in_batch = 0, batch = 0;
    // Memory traffic is batched four steps at a time: the inputs of steps i0+4 .. i0+7 are requested (16 B per row and
    // plane) while steps i0 .. i0+3 run, and the four outputs leave as one 16-B store.  A load's wait also waits for every
    // older store of the wave (vmcnt is in issue order), so a store per step would put the write latency on every step.
    // n and AGC_EXTRA are multiples of 4: a group never straddles the emitted audio / lookahead boundary.
    const float4* __restrict__ xg = reinterpret_cast<const float4*>(magrow + kAgcExtra);
    const float4* __restrict__ ag = reinterpret_cast<const float4*>(magrow);
    const float4* __restrict__ zg = reinterpret_cast<const float4*>(zrow);  // two complex samples per float4
    const uint32_t ngroups = n / 4;
    float4 nx = xg[0], na = ag[0];
    float4 nz0 = make_float4(0.f, 0.f, 0.f, 0.f), nz1 = nz0;
    if (has_z)
        nz0 = zg[0], nz1 = zg[1];

#ifdef MI_BLOCK_PROF
    BlockIo bio{{0, 0, 0, 0, 0, 0, 0, 0}, 0, 0, 0, 0, magrow, zrow, wmain, carry, iqo, has_z, has_iqo, n, 0.0f, 0xffffffffu};
    const unsigned long long prof_k0 = __builtin_readcyclecounter();
#else
    BlockIo bio{magrow, zrow, wmain, carry, iqo, has_z, has_iqo, n, 0.0f, 0xffffffffu};
#endif
    int skip = 0;  // groups to take one by one before the next steady block is tried
    // After a block the group fetched ahead is not the next one.  It is fetched again only if the sample loop really takes
    // the next group: a load into nx / na here would first wait for the previous load into the same registers, and that wait
    // (the memory counter is in order) also covers the stores the block has just issued.
    bool stale = false;
    {
        for (uint32_t gi = 0; gi < ngroups; ++gi) {
          if constexpr (kUni) {
              if (skip > 0) {
                  --skip;
              } else if (a.steady_blocks) {
                  // wave-uniform by construction (one channel per wave): say so, so that this stays on the scalar unit
                  const int st = __builtin_amdgcn_readfirstlane(c.s.current_state);
                  const int nxt = __builtin_amdgcn_readfirstlane(c.s.next_state);
                  if (st == nxt) {  // (a branch on the state here, and the blocks' own tests of it fold away)
                      if (!kPre && T::am_plain && st == SQ_CLOSED && __builtin_amdgcn_readfirstlane(c.s.closed_sample_count) == kRecentSampleSize &&
                          __builtin_amdgcn_readfirstlane(c.s.recent_open_count) == 0u)
                          am_plain_run<false>(c, a, bio, gi, ngroups, in_batch, batch, batch_open, skip, stale, row);
                      else if (!kPre && T::am_plain && st == SQ_OPEN && __builtin_amdgcn_readfirstlane(c.s.using_post_filter) == 0) {
                          if (((__builtin_amdgcn_readfirstlane(c.s.sample_count) + 1u) & 15u) != 0u) {  // (one general block aligns it)
                              bool again = false;
                              steady_streak<kPre, SQ_OPEN>(c, a, bio, gi, ngroups, in_batch, batch, batch_open, skip, stale, pre, pre_on, row, true, &again);
                              if (again && gi < ngroups) {
                                  --gi;  // the loop increment
                                  continue;
                              }
                          } else {
                              am_plain_run<true>(c, a, bio, gi, ngroups, in_batch, batch, batch_open, skip, stale, row);
                          }
                      } else if (kPre && st == SQ_OPEN && pre_on && TyAmPlain::matches(P) && __builtin_amdgcn_readfirstlane(c.s.using_post_filter) == 0) {
                          if (((__builtin_amdgcn_readfirstlane(c.s.sample_count) + 1u) & 15u) != 0u) {  // (one general block aligns it)
                              bool again = false;
                              steady_streak<kPre, SQ_OPEN>(c, a, bio, gi, ngroups, in_batch, batch, batch_open, skip, stale, pre, pre_on, row, true, &again);
                              if (again && gi < ngroups) {
                                  --gi;  // the loop increment
                                  continue;
                              }
                          } else if (c.split) {
                              open_streak_am<true>(c, a, bio, gi, ngroups, in_batch, batch, batch_open, skip, stale, pre, pre_on, row);
                          } else {
                              open_streak_am<false>(c, a, bio, gi, ngroups, in_batch, batch, batch_open, skip, stale, pre, pre_on, row);
                          }
                      } else if (kPre && st == SQ_CLOSED && pre_on && __builtin_amdgcn_readfirstlane(c.s.closed_sample_count) == kRecentSampleSize &&
                          __builtin_amdgcn_readfirstlane(c.s.recent_open_count) == 0u)
                          idle_streak(c, a, bio, gi, ngroups, in_batch, batch, batch_open, skip, stale, pre, pre_on, row);
                      else if (st == SQ_CLOSED)
                          steady_streak<kPre, SQ_CLOSED>(c, a, bio, gi, ngroups, in_batch, batch, batch_open, skip, stale, pre, pre_on, row);
                      else if (st == SQ_OPEN)
                          steady_streak<kPre, SQ_OPEN>(c, a, bio, gi, ngroups, in_batch, batch, batch_open, skip, stale, pre, pre_on, row);
                      else
                          steady_streak<kPre, -1>(c, a, bio, gi, ngroups, in_batch, batch, batch_open, skip, stale, pre, pre_on, row);
                      if (gi >= ngroups)
                          break;
                  }
              }
          }
          const uint32_t i0 = gi * 4;
          if constexpr (kPre) {
              if (pre_on) {
                  // A low-pass channel rewrites wavein[] with the filtered magnitude (rtl_airband.cpp:548): it must not pass the
                  // pre-filter wave, which reads the raw one.  Other channels only say where they are (the pre-filter wave stays
                  // within the ring's reach of that).
                  if (P.lowpass_enabled && has_z) {
                      if (!pre_wait(pre, c.lane, i0, i0 + 4u)) {
                          pre_on = false;  // (it never came: this wave computes everything itself from here on)
                          if (c.lane == 0 && a.pre_timeouts)
                              atomicAdd(a.pre_timeouts, 1u);
                      }
                  } else if (c.lane == 0) {
                      *(pre_vu32*)&pre->m_pos = i0;
                  }
              }
          }
          if (stale) {
              nx = xg[gi];
              na = ag[gi];
              if (has_z)
                  nz0 = zg[2 * gi], nz1 = zg[2 * gi + 1];
              stale = false;
          }
          const float4 cx = nx, ca = na, cz0 = nz0, cz1 = nz1;
          {
              const uint32_t gn = (gi + 1 < ngroups) ? gi + 1 : gi;
              nx = xg[gn];
              na = ag[gn];
              if (has_z)
                  nz0 = zg[2 * gn], nz1 = zg[2 * gn + 1];
          }
          const float gx[4] = {cx.x, cx.y, cx.z, cx.w}, ga[4] = {ca.x, ca.y, ca.z, ca.w};
          const float gre[4] = {cz0.x, cz0.z, cz1.x, cz1.z}, gim[4] = {cz0.y, cz0.w, cz1.y, cz1.w};
          float pend[4] = {0.f, 0.f, 0.f, 0.f};  // waveout of the group, stored together at its end
          int flushed = 0;                        // outputs [0, flushed) already went out one by one (a fade rewrote them)
          unsigned posted = 0;                    // bit m: step m went to the audio wave, which stores its outputs
          // Idle channel: CLOSED, and no sample of the group lifts pre_filter_.capped_ to the squelch level.  Then the four
          // steps only move the averages, the noise floor, the closed-sample counter and the squelch ring (squelch.cpp:
          // 442-449, 195-246 with every branch not taken); tried on copies, committed only if it held for all four.
          bool idle = false;
          if (c.s.current_state == SQ_CLOSED && c.s.next_state == SQ_CLOSED) {
              ChanState& s = c.s;
              float nf = s.noise_floor, cap = s.moving_avg_cap, full = s.pre_full, capd = s.pre_capped, cache = s.squelch_level_cache;
              uint32_t sc = s.sample_count, closed = s.closed_sample_count, recent = s.recent_open_count;
              int head = s.buffer_head, tail = s.buffer_tail;
              float ringv[4];
              int ringi[4];
              bool quiet = true;
#pragma unroll
              for (int m = 0; m < 4; ++m) {
                  if (closed < kRecentSampleSize) {
                      closed++;
                  } else if (closed == kRecentSampleSize) {
                      recent = 0;
                      cache = 0.0f;
                  }
                  tail = (tail + 1 == kSquelchRing) ? 0 : tail + 1;
                  head = (head + 1 == kSquelchRing) ? 0 : head + 1;
                  sc++;
                  if ((sc & 15u) == 0) {
                      const float new_factor = static_cast<float>(1.0 - static_cast<double>(0.97f));
                      nf = nf * 0.97f + std_min(capd, nf) * new_factor + 1e-6f;
                      cap = P.using_manual_level ? P.manual_cap : P.cap_factor * nf;
                      cache = 0.0f;
                  }
                  update_avg(cap, full, capd, gx[m]);
                  ringv[m] = capd * 0.9f;
                  ringi[m] = head;
                  float level;
                  if (P.using_manual_level) {
                      level = P.manual_signal_level;
                  } else {
                      if (cache == 0.0f)
                          cache = ((recent >= kFlapOpensThreshold && P.flappy_signal_ratio < P.normal_signal_ratio) ? P.flappy_signal_ratio
                                                                                                                    : P.normal_signal_ratio) * nf;
                      level = cache;
                  }
                  quiet = quiet && !(capd >= level);
              }
              if (quiet) {
                  idle = true;
                  s.noise_floor = nf, s.moving_avg_cap = cap, s.pre_full = full, s.pre_capped = capd, s.squelch_level_cache = cache;
                  s.sample_count = sc, s.closed_sample_count = closed, s.recent_open_count = recent;
                  s.buffer_head = head, s.buffer_tail = tail;
                  if (P.lowpass_enabled) {
#pragma unroll
                      for (int m = 0; m < 4; ++m)
                          c.ring[ringi[m]] = ringv[m];
                  }
                  if (has_iqo) {
                      float4* z = reinterpret_cast<float4*>(iqo + i0);
                      z[0] = make_float4(0.f, 0.f, 0.f, 0.f);
                      z[1] = make_float4(0.f, 0.f, 0.f, 0.f);
                  }
                  in_batch += 4;  // WAVE_BATCH is a multiple of 4: a batch ends at a group end
                  if (in_batch == kWaveBatch)
                      MI_END_BATCH();
              }
          }
          if (!idle) {
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            const uint32_t i = i0 + static_cast<uint32_t>(m);
            float x = gx[m];         // wavein[j]
            const float ax = ga[m];  // wavein[j - AGC_EXTRA]
            float re = gre[m], im = gim[m];  // iq_in[2*(j-AGC_EXTRA)], [+1]

            process_raw(c, x);  // rtl_airband.cpp:529

            if (has_z && should_filter(c)) {  // rtl_airband.cpp:532-552
                const uint32_t idx = c.s.dm_phi >> 16;  // sincosf_lut, util.cpp:113-127
                const float fract = static_cast<float>(c.s.dm_phi & 0xffff) / 65536.0f;
                float v1 = a.sin_lut[idx], v2 = a.sin_lut[idx + 1];
                const float swf = v1 + (v2 - v1) * fract;
                v1 = a.cos_lut[idx];
                v2 = a.cos_lut[idx + 1];
                const float cwf = v1 + (v2 - v1) * fract;
                const float nswf = -swf;  // multiply(real, imag, cwf, -swf), rtl_airband.cpp:141-144
                float re_tmp = re * cwf - im * nswf;
                float im_tmp = im * cwf + re * nswf;
                c.s.dm_phi = (c.s.dm_phi + P.dm_dphi) & 0xffffffu;
                if (P.lowpass_enabled) {  // LowpassFilter::apply, filters.cpp:146-163
                    ChanState& s = c.s;
                    s.lp_xr[0] = s.lp_xr[1], s.lp_xi[0] = s.lp_xi[1];
                    s.lp_xr[1] = s.lp_xr[2], s.lp_xi[1] = s.lp_xi[2];
                    s.lp_xr[2] = re_tmp / P.lowpass_gain;
                    s.lp_xi[2] = im_tmp / P.lowpass_gain;
                    s.lp_yr[0] = s.lp_yr[1], s.lp_yi[0] = s.lp_yi[1];
                    s.lp_yr[1] = s.lp_yr[2], s.lp_yi[1] = s.lp_yi[2];
                    s.lp_yr[2] = (s.lp_xr[0] + s.lp_xr[2]) + (2.0f * s.lp_xr[1]) + (P.lowpass_yc0 * s.lp_yr[0]) + (P.lowpass_yc1 * s.lp_yr[1]);
                    s.lp_yi[2] = (s.lp_xi[0] + s.lp_xi[2]) + (2.0f * s.lp_xi[1]) + (P.lowpass_yc0 * s.lp_yi[0]) + (P.lowpass_yc1 * s.lp_yi[1]);
                    re_tmp = s.lp_yr[2];
                    im_tmp = s.lp_yi[2];
                }
                re = re_tmp;
                im = im_tmp;
                x = sqrtf(re * re + im * im);
                magrow[kAgcExtra + i] = x;  // channel->wavein[j] is overwritten and read again AGC_EXTRA steps later
                if (P.lowpass_enabled)
                    process_filtered(c, x);
            }

            if (am && c.split) {  // the audio wave's: it owns agcavgfast and the outputs the fade rewrites
                if (c.s.current_state != SQ_OPEN && c.s.next_state == SQ_OPEN) {  // first_open_sample
                    aux_post(c, AUX_FIRST_OPEN, 1u, i, squelch_level(c), 0.0f);  // (one token: the level)
                } else if ((c.s.current_state == SQ_CLOSING && c.s.next_state == SQ_CLOSED) ||
                           (c.s.current_state != SQ_LOW_SIGNAL_ABORT && c.s.next_state == SQ_LOW_SIGNAL_ABORT)) {  // last_open_sample
                    // the fade rewrites the previous 99 outputs: those of the group's earlier steps that are this wave's (zeros) have to
                    // be in memory first
#pragma unroll
                    for (int k = 0; k < m; ++k)
                        if (!((posted >> k) & 1u))
                            W(kAgcExtra + i0 + k) = pend[k];
                    flushed = m;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    aux_post(c, AUX_LAST_OPEN, 0u, i, 0.0f, 0.0f);
                }
            } else if (am) {  // rtl_airband.cpp:554-569
                if (c.s.current_state != SQ_OPEN && c.s.next_state == SQ_OPEN) {  // first_open_sample
                    for (int kk = 0; kk < kAgcExtra; ++kk) {
                        const float w = magrow[i + kk];
                        if (w >= squelch_level(c))
                            c.s.agcavgfast = c.s.agcavgfast * 0.9f + w * 0.1f;
                    }
                } else if ((c.s.current_state == SQ_CLOSING && c.s.next_state == SQ_CLOSED) ||
                           (c.s.current_state != SQ_LOW_SIGNAL_ABORT && c.s.next_state == SQ_LOW_SIGNAL_ABORT)) {  // last_open_sample
                    // the fade rewrites the previous 99 outputs: the group's earlier ones have to be in memory first
#pragma unroll
                    for (int k = 0; k < m; ++k)
                        W(kAgcExtra + i0 + k) = pend[k];
                    flushed = m;
                    float v = W(i);
                    for (int kk = 1; kk < kAgcExtra; ++kk) {
                        v = v * 0.94f;
                        W(i + kk) = v;
                    }
                }
            }

            float wout = 0.0f;
            if (c.s.current_state == SQ_OPEN || c.s.current_state == SQ_CLOSING) {  // should_process_audio
                if (am && c.split) {
                    aux_post(c, AUX_BLOCK, 1u, i, squelch_level(c), x, ax);
                    posted |= 1u << m;
                } else if (am) {  // rtl_airband.cpp:575-585
                    if (x > squelch_level(c))
                        c.s.agcavgfast = c.s.agcavgfast * 0.995f + x * 0.005f;
                    wout = (ax - c.s.agcavgfast) / (c.s.agcavgfast * 1.5f);
                    if (fabsf(wout) > 0.8f) {
                        wout *= 0.85f;
                        c.s.agcavgfast *= 1.15f;
                    }
                } else if (c.split) {  // NFM: the audio wave's (rtl_airband.cpp:587-641 from the filtered I/Q)
                    aux_post(c, AUX_BLOCK, 1u, i, re, im);
                    posted |= 1u << m;
                } else {  // NFM, rtl_airband.cpp:587-604
                    if (!a.fm_quadri) {
                        const float nbj = -c.s.pj;  // polar_disc_fast: multiply(ar, aj, br, -bj)
                        const float cr = re * c.s.pr - im * nbj;
                        const float cj = im * c.s.pr + re * nbj;
                        wout = static_cast<float>(static_cast<double>(fast_atan2(cj, cr)) * M_1_PI);
                    } else {  // fm_quadri_demod
                        wout = static_cast<float>(static_cast<double>((c.s.pr * im - re * c.s.pj) / (re * re + im * im + 1.0f)) * M_1_PI);
                    }
                    c.s.pr = re;
                    c.s.pj = im;
                    c.s.agcavgfast = c.s.agcavgfast * 0.995f + wout * 0.005f;
                    wout -= c.s.agcavgfast;
                    wout = wout * P.one_minus_alpha + c.s.prev_waveout * P.alpha;
                    c.s.prev_waveout = wout;
                }
                if (!c.split)
                    process_audio(c, wout);  // rtl_airband.cpp:608
            }

            if (!c.split && is_open(c)) {  // rtl_airband.cpp:612-641
                if (P.notch_enabled) {  // NotchFilter::apply, filters.cpp:50-64
                    ChanState& s = c.s;
                    s.notch_x[0] = s.notch_x[1];
                    s.notch_x[1] = s.notch_x[2];
                    s.notch_x[2] = wout;
                    s.notch_y[0] = s.notch_y[1];
                    s.notch_y[1] = s.notch_y[2];
                    s.notch_y[2] = P.notch_d0 * s.notch_x[2] - P.notch_d1 * s.notch_x[1] + P.notch_d0 * s.notch_x[0] + P.notch_d1 * s.notch_y[1] -
                                   P.notch_d2 * s.notch_y[0];
                    wout = s.notch_y[2];
                }
                wout *= ampfactor;
                if (wout != wout)
                    wout = 0.0f;
                else if (wout > 1.0f)
                    wout = 1.0f;
                else if (wout < -1.0f)
                    wout = -1.0f;
                batch_open = true;
                if (has_iqo)
                    iqo[i] = make_float2(re, im);
            } else {
                wout = 0.0f;
                if (has_iqo && !((posted >> m) & 1u))
                    iqo[i] = make_float2(0.0f, 0.0f);
            }
            pend[m] = wout;

            if (++in_batch == kWaveBatch)  // rtl_airband.cpp:523,628,667-669
                MI_END_BATCH();
          }
          }
          if (flushed == 0 && posted == 0) {
              const uint32_t v = kAgcExtra + i0;
              float* dst = (v < n) ? wmain + v : carry + (v - n);
              *reinterpret_cast<float4*>(dst) = make_float4(pend[0], pend[1], pend[2], pend[3]);
          } else {
#pragma unroll
              for (int k = 0; k < 4; ++k)
                  if (k >= flushed && !((posted >> k) & 1u))
                      W(kAgcExtra + i0 + k) = pend[k];
          }
        }
    }

#ifdef MI_BLOCK_PROF
    if (kUni && threadIdx.x == 0)
        printf("blockprof row %d (mod %d lp %d ctcss %d notch %d): total %llu cyc, %llu blocks, %llu of %u steps in blocks; after-return %llu loads %llu between blocks %llu pre %llu level+lsc %llu filter %llu post %llu audio %llu commit %llu\n",
               row, (int)P.modulation, (int)P.lowpass_enabled, (int)P.ctcss_enabled, (int)P.notch_enabled, __builtin_readcyclecounter() - prof_k0, bio.blocks, bio.steps, n, bio.after_ret, bio.prof[6], bio.prof[7], bio.prof[0], bio.prof[1], bio.prof[2], bio.prof[3], bio.prof[4], bio.prof[5]);
#endif
    // plane carry: the last AGC_EXTRA (possibly low-pass-rewritten) magnitudes and raw bins move to the front,
    // the reference's memmove (rtl_airband.cpp:643-646)
    {
        float* __restrict__ hrow = a.mag_head + static_cast<size_t>(row) * a.plane_stride;
        for (int v = 0; v < kAgcExtra; ++v)
            hrow[v] = magrow[n + v];
    }
    if (has_z) {
        const size_t zoff = (static_cast<size_t>(stream) * a.n_iq_rows + P.iq_row) * a.plane_stride;
        const float2* zw = a.cplx + zoff;
        float2* zh = a.cplx_head + zoff;
        for (int v = 0; v < kAgcExtra; ++v)
            zh[v] = zw[n + v];
    }

    if (c.split) {
        // the audio wave finishes what it was sent, stores the detectors' state and hands back the part of ChanState it owns
        aux_post(c, AUX_QUIT, 0u, 0u, 0.0f, 0.0f);
        for (unsigned spin = 0; aux_peek(&aux->done) == 0u; ++spin) {
            __builtin_amdgcn_s_sleep(2);
            if (spin > 4u * kPreSpin) {
                if (c.lane == 0 && c.timeouts)
                    atomicAdd(c.timeouts, 1u);
                break;
            }
        }
        asm volatile("" ::: "memory");
        auto fw = [&](const int k) { return aux_peek(&aux->fin[k]); };
        auto ff = [&](const int k) { return __uint_as_float(fw(k)); };
        auto f64 = [&](const int k) { return static_cast<uint64_t>(fw(k)) | (static_cast<uint64_t>(fw(k + 1)) << 32); };
        ChanState& s = c.s;
        s.agcavgfast = ff(0), s.pr = ff(1), s.pj = ff(2), s.prev_waveout = ff(3);
        s.notch_x[0] = ff(4), s.notch_x[1] = ff(5), s.notch_x[2] = ff(6);
        s.notch_y[0] = ff(7), s.notch_y[1] = ff(8), s.notch_y[2] = ff(9);
        s.cf_enough = static_cast<int32_t>(fw(10)), s.cf_count = static_cast<int32_t>(fw(11)), s.cf_has_tone = static_cast<int32_t>(fw(12));
        s.cs_enough = static_cast<int32_t>(fw(13)), s.cs_count = static_cast<int32_t>(fw(14)), s.cs_has_tone = static_cast<int32_t>(fw(15));
        s.cf_found = f64(16), s.cf_not_found = f64(18), s.cs_found = f64(20), s.cs_not_found = f64(22);
        s.active_counter = f64(24);
    } else if (uni && P.ctcss_enabled) {
        if (c.lane < P.ctcss_fast_ndet)
            c.cq_fast[c.lane] = c.gf_q1, c.cq_fast[kMaxTones + c.lane] = c.gf_q2;
        if (c.lane < P.ctcss_slow_ndet)
            c.cq_slow[c.lane] = c.gs_q1, c.cq_slow[kMaxTones + c.lane] = c.gs_q2;
    }
    a.st[row] = c.s;
    if (a.stats) {  // what output.cpp:634-811 reads through the Squelch getters
        mi_channel_stats st;
        st.noise_level = c.s.noise_floor;
        st.signal_level = c.s.pre_full;
        float lvl;
        if (P.using_manual_level)
            lvl = P.manual_signal_level;
        else if (c.s.squelch_level_cache != 0.0f)
            lvl = c.s.squelch_level_cache;
        else
            lvl = ((flapping(c) && P.flappy_signal_ratio < P.normal_signal_ratio) ? P.flappy_signal_ratio : P.normal_signal_ratio) * c.s.noise_floor;
        st.squelch_level = lvl;
        st.agcavgfast = c.s.agcavgfast;
        st.open_count = c.s.open_count;
        st.flappy_count = c.s.flappy_count;
        st.ctcss_count = c.s.cs_found;
        st.no_ctcss_count = c.s.cs_not_found;
        st.active_counter = c.s.active_counter;
        st.squelch_state = c.s.current_state;
        const bool pre = c.s.pre_capped >= lvl;
        const bool post = c.s.using_post_filter && c.s.post_capped >= c.ring[c.s.buffer_tail];
        st.signal_outside_filter = (c.s.using_post_filter && pre && !post) ? 1 : 0;
        a.stats[row] = st;
    }
}

// the channel's type picks the instantiation (one channel per wave; packed lanes take the open one)
template <bool kUni, bool kPre>
__device__ __forceinline__ void demod_dispatch(const DemodArgs& a, LdsPre* pre, LdsAux* aux) {
    if constexpr (kUni) {
        const ChanParams& P0 = a.cp[demod_row(a, static_cast<int>(blockIdx.x)) % a.nch];
        if (TyAmPlain::matches(P0))
            demod_body<kUni, kPre, TyAmPlain>(a, pre, aux);
        else if (TyNfmLp::matches(P0))
            demod_body<kUni, kPre, TyNfmLp>(a, pre, aux);
        else if (TyNfm::matches(P0))
            demod_body<kUni, kPre, TyNfm>(a, pre, aux);
        else
            demod_body<kUni, kPre, TyAny>(a, pre, aux);
    } else {
        demod_body<kUni, kPre, TyAny>(a, pre, aux);
    }
}
// Register budgets (waves per SIMD the kernels are compiled for; diagnostic builds override them: make EXTRA=-DMI_UNI_EU=3).
//   k_demod_uni    one channel per wave, thousands of rows: 128 registers.  It spills (1.3 KB of scratch per lane), and still the
//                  rows are no slower (1.23 vs 1.30 ms per 2 s at 2 048 rows) -- and BASELINE configs[3] gains a fifth (146 -> 181 GS/s):
//                  the kernel shares every SIMD with stage 1 of the next call (167 registers a wave), and what the two can hold
//                  there together is what bounds the step.  96 registers and fewer: the spills take over
//                  (1.9 / 3.8 / 2.7 ms at 96 / 80 / 64).
//   k_demod_packed several channels per wave (more than MI_OPT_UNI_ROWS rows): 168, as measured in round 2.
//   k_demod_pw     four waves per channel, one channel per CU: a SIMD's whole register file each.
#ifndef MI_UNI_EU
#define MI_UNI_EU 4
#endif
#ifndef MI_PACKED_EU
#define MI_PACKED_EU 3
#endif
#ifndef MI_PW_EU
#define MI_PW_EU 1
#endif
#if MI_PW_EU > 0
#define MI_PW_BOUNDS __launch_bounds__(256, MI_PW_EU)
#else
#define MI_PW_BOUNDS __launch_bounds__(256)
#endif
__global__ __launch_bounds__(64, MI_UNI_EU) void k_demod_uni(const DemodArgs a) {
    demod_dispatch<true, false>(a, nullptr, nullptr);
}
__global__ __launch_bounds__(64, MI_PACKED_EU) void k_demod_packed(const DemodArgs a) {
    demod_dispatch<false, false>(a, nullptr, nullptr);
}

// The full_ wave of k_demod_pw: pre_filter_.full_ (squelch.cpp:505) of every step of the call -- a function of the raw magnitudes alone and
// the longest dependent chain of the pre-filter pair (two operations per step; the noise floor and, outside the decays, capped_ follow
// from it in a few operations per period).  Walked here on its own, 64 steps per trip, it sets the pace the pre-filter wave used to.
__device__ __forceinline__ void full_wave(const DemodArgs& a, LdsPre* pre, const int lane) {
    const int row = demod_row(a, static_cast<int>(blockIdx.x));
    const float* __restrict__ xrow = a.mag + static_cast<size_t>(row) * a.plane_stride + kAgcExtra;
    const uint32_t n = a.nsteps;
    const float k99 = 0.99f, n99 = static_cast<float>(1.0 - static_cast<double>(0.99f));
    float full = a.st[row].pre_full;
    float xn = xrow[min(static_cast<uint32_t>(lane), n - 1u)];  // the next block's samples, requested a block ahead
    uint32_t reach = 0;  // steps below this are within the ring's reach of the channel wave as last seen
#ifdef MI_BLOCK_PROF
    unsigned long long fw_t = __builtin_readcyclecounter(), fw_reach = 0, fw_load = 0, fw_chain = 0, fw_tail = 0, fw_n = 0;
#endif
    for (uint32_t i0 = 0; i0 < n; i0 += 64u) {
        const int kmax = static_cast<int>(min(64u, n - i0));
        if (i0 + 64u > reach) {
            for (unsigned idle = 0;; ++idle) {
                reach = pre_peek(&pre->m_pos) + (kPreRing - 64u);
                if (i0 + 64u <= reach)
                    break;
                __builtin_amdgcn_s_sleep(2);
                if (idle > 4u * kPreSpin)
                    return;  // (the channel wave is gone or stuck)
            }
        }
#ifdef MI_BLOCK_PROF
        unsigned long long fw_a = __builtin_readcyclecounter();
        fw_reach += fw_a - fw_t;
#endif
        float x = xn;
        asm volatile("" : "+v"(x));
#ifdef MI_BLOCK_PROF
        unsigned long long fw_b = __builtin_readcyclecounter();
        fw_load += fw_b - fw_a;
#endif
        xn = xrow[min(i0 + 64u + static_cast<uint32_t>(lane), n - 1u)];
        const float b = x * n99;
        float F = 0.0f, T = full * k99;  // T: what lane 0 keeps reading (its shifted source does not exist)
        for (int p_ = 0; p_ < kmax; p_ += 16)
            full_passes16(F, T, b);
#ifdef MI_BLOCK_PROF
        asm volatile("" : "+v"(F));
        unsigned long long fw_c = __builtin_readcyclecounter();
        fw_chain += fw_c - fw_b;
#endif
        if (lane < kmax)
            *(pre_vf32*)&pre->F[(i0 + static_cast<uint32_t>(lane)) & (kPreRing - 1u)] = F;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the ring values before the mark (see pre_wave)
        if (lane == 0)
            *(pre_vu32*)&pre->f_done = i0 + static_cast<uint32_t>(kmax);
        full = lane_read(F, kmax - 1);
#ifdef MI_BLOCK_PROF
        fw_t = __builtin_readcyclecounter();
        fw_tail += fw_t - fw_c;
        ++fw_n;
#endif
    }
#ifdef MI_BLOCK_PROF
    if (lane == 0 && (row == 0 || row == 5))
        printf("fullprof row %d: %llu blocks; per block: reach check %llu, sample wait %llu, chain %llu, ring + mark %llu\n", row, fw_n, fw_reach / fw_n, fw_load / fw_n, fw_chain / fw_n,
               fw_tail / fw_n);
#endif
}

// The pre-filter wave of k_demod_pw: the recurrence of steady_block()'s first phase, block after block over the whole call.
// kFullWave: full_ comes from the full_ wave's ring (k_demod_pw); else this wave walks it too (k_demod_pw2) and stays within the
// ring's reach of the channel wave itself.
template <bool kFullWave>
__device__ __forceinline__ void pre_wave(const DemodArgs& a, LdsPre* pre, const int lane) {
    const int row = demod_row(a, static_cast<int>(blockIdx.x));
    const ChanParams P = a.cp[row % a.nch];
    const ChanState& s0 = a.st[row];
    const float* __restrict__ xrow = a.mag + static_cast<size_t>(row) * a.plane_stride + kAgcExtra;
    const uint32_t n = a.nsteps;
    // the state the call starts from (wave-uniform)
    float nf = s0.noise_floor, cap = s0.moving_avg_cap, full = s0.pre_full, capd = s0.pre_capped;
    uint32_t sc = s0.sample_count;
    float xn = xrow[min(static_cast<uint32_t>(lane), n - 1u)];  // the next block's samples, requested a block ahead
    uint32_t reach = 0;  // (no full_ wave) steps below this are within the ring's reach of the channel wave as last seen
#ifdef MI_BLOCK_PROF
    unsigned long long pw_t0 = __builtin_readcyclecounter(), pw_wait = 0, pw_load = 0, pw_per = 0, pw_tail = 0, pw_fast = 0, pw_n = 0;
#endif
    for (uint32_t i0 = 0; i0 < n; i0 += 64u) {
        const int kmax = static_cast<int>(min(64u, n - i0));
        // (the full_ wave stays within the ring's reach of the channel wave, and this wave behind the full_ wave)
#ifdef MI_BLOCK_PROF
        unsigned long long pw_a = __builtin_readcyclecounter();
        pw_wait += pw_a - pw_t0;
#endif
        float x = xn;
        asm volatile("" : "+v"(x));
#ifdef MI_BLOCK_PROF
        unsigned long long pw_b = __builtin_readcyclecounter();
        pw_load += pw_b - pw_a;
#endif
        xn = xrow[min(i0 + 64u + static_cast<uint32_t>(lane), n - 1u)];
        if (kFullWave) {  // full_ of these steps from the full_ wave (usually blocks ahead)
            for (unsigned spin = 0; pre_peek(&pre->f_done) < i0 + static_cast<uint32_t>(kmax); ++spin) {
                __builtin_amdgcn_s_sleep(1);
                if (spin > 4u * kPreSpin)
                    return;
            }
        } else if (i0 + 64u > reach) {  // stay within the ring's reach of the channel wave (it posts its position before it waits for this wave)
            for (unsigned idle = 0;; ++idle) {
                reach = pre_peek(&pre->m_pos) + (kPreRing - 64u);
                if (i0 + 64u <= reach)
                    break;
                __builtin_amdgcn_s_sleep(2);
                if (idle > 4u * kPreSpin)
                    return;  // (the channel wave is gone or stuck: it computes its own values when its wait runs out)
            }
        }
        asm volatile("" ::: "memory");
#ifdef MI_BLOCK_PROF
        {
            const unsigned long long t_ = __builtin_readcyclecounter();
            pw_fast += t_ - pw_b;
            pw_b = t_;
        }
#endif
        float F = kFullWave ? *(pre_vf32*)&pre->F[(i0 + static_cast<uint32_t>(lane)) & (kPreRing - 1u)] : 0.0f;
        float C, NFv, CAPv;
        int zf;
        pre_block<kFullWave>(P, nf, cap, full, capd, sc, x, kmax, lane, F, C, NFv, CAPv, zf);
#ifdef MI_BLOCK_PROF
        unsigned long long pw_c = __builtin_readcyclecounter();
        pw_per += pw_c - pw_b;
#endif
        if (lane < kmax) {
            const unsigned at = (i0 + static_cast<uint32_t>(lane)) & (kPreRing - 1u);
            *(pre_vf32*)&pre->C[at] = C, *(pre_vf32*)&pre->NF[at] = NFv, *(pre_vf32*)&pre->CAP[at] = CAPv;
            if (!kFullWave)
                *(pre_vf32*)&pre->F[at] = F;
        }
        // The ring values before the mark.  A wave's LDS operations are issued and executed in order, so the compiler barrier is what
        // matters; the wait makes the order explicit at the price of the LDS counter only (a workgroup fence would also wait for
        // the global loads this wave has in flight for the next block: vmcnt(0) once per block).
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0)
            *(pre_vu32*)&pre->h_done = i0 + static_cast<uint32_t>(kmax);
        // the state the next block starts from
        full = lane_read(F, kmax - 1);
        capd = lane_read(C, kmax - 1);
        sc += static_cast<uint32_t>(kmax);
#ifdef MI_BLOCK_PROF
        pw_t0 = __builtin_readcyclecounter();
        pw_tail += pw_t0 - pw_c;
        ++pw_n;
#endif
    }
#ifdef MI_BLOCK_PROF
    if (lane == 0 && (row == 0 || row == 5))
        printf("preprof row %d: %llu blocks; per block: loop head %llu, sample wait %llu, wait for full_ %llu, floor + capped_ %llu, ring + state %llu\n", row, pw_n, pw_wait / pw_n,
               pw_load / pw_n, pw_fast / pw_n, pw_per / pw_n, pw_tail / pw_n);
#endif
}

// The audio wave of k_demod_pw (NFM channels): rtl_airband.cpp:587-641 and Squelch::process_audio_sample for the steps the channel
// wave posts, in its order.  A BLOCK of n steps is taken in units that end short of a CTCSS detector window's last sample -- within
// a unit is_open() cannot change -- as steady_block() does: step m in lane m, every recurrence a systolic chain; the sample that ends
// a window is a unit of its own and goes through ctcss_process_lanes() like a step of the sample loop.  The same IEEE operations in
// the same order on the same operands as there.
__device__ __forceinline__ void audio_wave(const DemodArgs& a, LdsAux* x, const int lane) {
    const int row = demod_row(a, static_cast<int>(blockIdx.x));
    const int stream = row / a.nch, ch = row - stream * a.nch;
    const ChanParams P = a.cp[ch];
    const bool am = P.modulation == MI_MOD_AM;
    if (!a.audio_wave || (am && !TyAmPlain::matches(P)))
        return;  // (the channel wave decides the same way and keeps everything)
    const float* __restrict__ magrow = a.mag + static_cast<size_t>(row) * a.plane_stride;
    const ChanState& s0 = a.st[row];
    float agc = s0.agcavgfast, pr = s0.pr, pj = s0.pj, pw = s0.prev_waveout;
    float nx0 = s0.notch_x[0], nx1 = s0.notch_x[1], nx2 = s0.notch_x[2];
    float ny0 = s0.notch_y[0], ny1 = s0.notch_y[1], ny2 = s0.notch_y[2];
    int cf_enough = s0.cf_enough, cf_count = s0.cf_count, cf_has_tone = s0.cf_has_tone;
    int cs_enough = s0.cs_enough, cs_count = s0.cs_count, cs_has_tone = s0.cs_has_tone;
    uint64_t cf_found = s0.cf_found, cf_not_found = s0.cf_not_found, cs_found = s0.cs_found, cs_not_found = s0.cs_not_found;
    uint64_t active = s0.active_counter;
    float gf_c = 0.0f, gf_q1 = 0.0f, gf_q2 = 0.0f, gs_c = 0.0f, gs_q1 = 0.0f, gs_q2 = 0.0f;
    float* cq_fast = nullptr;
    float* cq_slow = nullptr;
    if (P.ctcss_enabled) {
        const float* cc_fast = a.ctcss_coeff + static_cast<size_t>(P.ctcss_row) * 2 * kMaxTones;
        const float* cc_slow = cc_fast + kMaxTones;
        cq_fast = a.ctcss_q + (static_cast<size_t>(stream) * a.n_ctcss_rows + P.ctcss_row) * 4 * kMaxTones;
        cq_slow = cq_fast + 2 * kMaxTones;
        if (lane < P.ctcss_fast_ndet)
            gf_c = cc_fast[lane], gf_q1 = cq_fast[lane], gf_q2 = cq_fast[kMaxTones + lane];
        if (lane < P.ctcss_slow_ndet)
            gs_c = cc_slow[lane], gs_q1 = cq_slow[lane], gs_q2 = cq_slow[kMaxTones + lane];
    }
    float* __restrict__ wmain = a.wmain + static_cast<size_t>(row) * a.wmain_stride;
    float* __restrict__ carry = a.carry + static_cast<size_t>(row) * kAgcExtra;
    float2* __restrict__ iqo = (a.iq_out && P.has_iq_outputs) ? a.iq_out + static_cast<size_t>(row) * a.iq_out_stride : nullptr;
    const uint32_t nsteps = a.nsteps;
    bool batch_open = false;
    unsigned d_tail = 0, tk_pos = 0;

    for (;;) {
        // The mark and the descriptor it would announce are read in one round trip (in that order: LDS executes a wave's operations
        // in order, and the posting wave writes the descriptor, waits, then moves the mark -- a mark seen moved means the descriptor
        // read behind it is the new one; a mark not moved yet, and the descriptor read is thrown away).
        unsigned word = 0, arg = 0;
        for (unsigned spin = 0;; ++spin) {
            const unsigned hv = *(aux_vu32*)&x->d_head;
            const unsigned long long dv = *(aux_vu64*)&x->d_desc[d_tail & (kDescRing - 1u)];
            if (static_cast<unsigned>(__builtin_amdgcn_readfirstlane(hv)) != d_tail) {
                word = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(dv));
                arg = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(dv >> 32));
                break;
            }
            __builtin_amdgcn_s_sleep(2);
            if (spin > 32u * kPreSpin)
                return;  // (the channel wave is gone)
        }
        asm volatile("" ::: "memory");
        const unsigned type = word & 0xffu;
        if (type == AUX_BLOCK && am) {
            // AM AGC and audio (rtl_airband.cpp:574-585, 612-641; no CTCSS, no notch, no raw I/Q on these channels).  A clip --
            // |waveout| > 0.8 -- feeds back into agcavgfast: the steps before the first one are final, the clipping step is taken
            // by itself and the rest of the block follows as a unit of its own from the value it leaves.
            int remaining = static_cast<int>(word >> 8);
            unsigned off = 0;
            while (remaining > 0) {
                const int n = remaining;
                const unsigned at = (tk_pos + off + static_cast<unsigned>(lane)) & (kTokRing - 1u);
                const float level = *(aux_vf32*)&x->tk_re[at], xs = *(aux_vf32*)&x->tk_im[at], ax = *(aux_vf32*)&x->tk_ax[at];
                const bool upd = xs > level;
                const float bA = xs * 0.005f;
                float G = 0.0f;
                if (__ballot(lane < n && !upd) == 0ull) {  // every sample above the level: the plain average
                    G = chain_passes(agc, bA, 0.995f, n);
                } else {
                    float Gp = agc;
                    MI_PASSES(0, n, {
                        Gp = shr1(G, Gp);
                        G = upd ? Gp * 0.995f + bA : Gp;
                    })
                }
                const float d = (ax - G) / (G * 1.5f);
                const unsigned long long clipm = __ballot(lane < n && fabsf(d) > 0.8f);
                const int k = clipm ? static_cast<int>(__builtin_ctzll(clipm)) : n;  // steps before the first clip
                const int take = clipm ? k + 1 : n;
                float out = (lane == k) ? d * 0.85f : d;
                out *= P.ampfactor;
                if (out != out)
                    out = 0.0f;
                else if (out > 1.0f)
                    out = 1.0f;
                else if (out < -1.0f)
                    out = -1.0f;
                batch_open = true;
                if (lane < take) {
                    const uint32_t v = kAgcExtra + arg + off + static_cast<uint32_t>(lane);
                    float* dst = (v < nsteps) ? wmain + v : carry + (v - nsteps);
                    *dst = out;
                }
                agc = clipm ? lane_read(G, k) * 1.15f : lane_read(G, n - 1);
                off += static_cast<unsigned>(take);
                remaining -= take;
            }
            tk_pos += word >> 8;
            if (lane == 0)
                *(aux_vu32*)&x->tk_tail = tk_pos;
        } else if (type == AUX_FIRST_OPEN) {  // bootstrap of agcavgfast from the previous AGC_EXTRA magnitudes, rtl_airband.cpp:556-562
            const float level = *(aux_vf32*)&x->tk_re[tk_pos & (kTokRing - 1u)];
            tk_pos += 1u;
            if (lane == 0)
                *(aux_vu32*)&x->tk_tail = tk_pos;
            const float w0 = magrow[arg + static_cast<unsigned>(lane)];
            const float w1 = magrow[arg + 64u + static_cast<unsigned>(lane < kAgcExtra - 64 ? lane : 0)];
            for (int kk = 0; kk < 64; ++kk) {
                const float w = lane_read(w0, kk);
                agc = (w >= level) ? agc * 0.9f + w * 0.1f : agc;
            }
            for (int kk = 0; kk < kAgcExtra - 64; ++kk) {
                const float w = lane_read(w1, kk);
                agc = (w >= level) ? agc * 0.9f + w * 0.1f : agc;
            }
        } else if (type == AUX_LAST_OPEN) {  // the fade of the previous AGC_EXTRA - 1 outputs, rtl_airband.cpp:564-568
            auto W = [&](const uint32_t v) -> float* { return v < nsteps ? wmain + v : carry + (v - nsteps); };
            float v = *W(arg);
            for (int kk = 1; kk < kAgcExtra; ++kk) {
                v = v * 0.94f;
                if (lane == 0)
                    *W(arg + static_cast<uint32_t>(kk)) = v;
            }
        } else if (type == AUX_BLOCK) {
            int remaining = static_cast<int>(word >> 8);
            unsigned off = 0;
            while (remaining > 0) {
                int lim = 64;  // steps before the one that ends a detector window
                if (P.ctcss_enabled) {
                    lim = P.ctcss_slow_window - 1 - cs_count;
                    if (!cs_enough)
                        lim = min(lim, P.ctcss_fast_window - 1 - cf_count);
                }
                const bool ends_window = lim <= 0;
                const int n = ends_window ? 1 : min(remaining, lim);
                const unsigned at = (tk_pos + off + static_cast<unsigned>(lane)) & (kTokRing - 1u);
                const float re = *(aux_vf32*)&x->tk_re[at], im = *(aux_vf32*)&x->tk_im[at];
                // discriminator, DC block, de-emphasis (rtl_airband.cpp:587-604)
                const float prr = shr1(re, pr), pjj = shr1(im, pj);
                float w;
                if (!a.fm_quadri) {
                    const float nbj = -pjj;  // polar_disc_fast: multiply(ar, aj, br, -bj)
                    const float cr = re * prr - im * nbj;
                    const float cj = im * prr + re * nbj;
                    w = static_cast<float>(static_cast<double>(fast_atan2(cj, cr)) * M_1_PI);
                } else {
                    w = static_cast<float>(static_cast<double>((prr * im - re * pjj) / (re * re + im * im + 1.0f)) * M_1_PI);
                }
                const float bN = w * 0.005f;
                const float G = chain_passes(agc, bN, 0.995f, n);
                const float e = (w - G) * P.one_minus_alpha;
                const float D = chain_passes(pw, e, P.alpha, n);  // (e + prev * alpha: the product, then the sum)
                const float d = D;
                const int last = n - 1;
                // Squelch::process_audio_sample (squelch.cpp:278-295; the state is OPEN or CLOSING: never CLOSED)
                if (P.ctcss_enabled) {
                    if (ends_window) {
                        const float smp = lane_read(d, 0);
                        ctcss_process_lanes(gs_c, gs_q1, gs_q2, P.ctcss_slow_ndet, P.ctcss_slow_window, cs_count, cs_enough, cs_has_tone, cs_found, cs_not_found, smp);
                        if (!cs_enough)
                            ctcss_process_lanes(gf_c, gf_q1, gf_q2, P.ctcss_fast_ndet, P.ctcss_fast_window, cf_count, cf_enough, cf_has_tone, cf_found,
                                                cf_not_found, smp);
                    } else {
                        const bool fast_too = !cs_enough;
                        int m = 0;
                        for (; m + 4 <= n; m += 4) {
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const float smp = lane_read(d, m + u);
                                const float q0 = gs_c * gs_q1 - gs_q2 + smp;
                                gs_q2 = gs_q1;
                                gs_q1 = q0;
                                if (fast_too) {
                                    const float f0 = gf_c * gf_q1 - gf_q2 + smp;
                                    gf_q2 = gf_q1;
                                    gf_q1 = f0;
                                }
                            }
                        }
                        for (; m < n; ++m) {
                            const float smp = lane_read(d, m);
                            const float q0 = gs_c * gs_q1 - gs_q2 + smp;
                            gs_q2 = gs_q1;
                            gs_q1 = q0;
                            if (fast_too) {
                                const float f0 = gf_c * gf_q1 - gf_q2 + smp;
                                gf_q2 = gf_q1;
                                gf_q1 = f0;
                            }
                        }
                        cs_count += n;
                        if (fast_too)
                            cf_count += n;
                    }
                }
                // output gate (rtl_airband.cpp:612-641)
                const bool gate = !P.ctcss_enabled || (cs_enough ? (cs_has_tone != 0) : (cf_has_tone != 0));
                float out = 0.0f;
                if (gate) {
                    out = d;
                    if (P.notch_enabled) {  // NotchFilter::apply, filters.cpp:50-64
                        const float u1 = shr1(d, nx2);
                        const float u2 = shr1(u1, nx1);
                        const float B = P.notch_d0 * d - P.notch_d1 * u1 + P.notch_d0 * u2;
                        float Y = 0.0f, Y1 = ny2, Y2 = ny1;
                        MI_PASSES(0, n, {
                            Y2 = shr1(Y1, Y2);
                            Y1 = shr1(Y, Y1);
                            Y = B + P.notch_d1 * Y1 - P.notch_d2 * Y2;
                        })
                        out = Y;
                        // the last three inputs / outputs, of this unit and of what came before it
                        const float ox1 = nx1, ox2 = nx2, oy1 = ny1, oy2 = ny2;
                        nx0 = n >= 3 ? lane_read(d, max(last - 2, 0)) : (n == 2 ? ox2 : ox1);
                        nx1 = n >= 2 ? lane_read(d, max(last - 1, 0)) : ox2;
                        nx2 = lane_read(d, last);
                        ny0 = n >= 3 ? lane_read(Y, max(last - 2, 0)) : (n == 2 ? oy2 : oy1);
                        ny1 = n >= 2 ? lane_read(Y, max(last - 1, 0)) : oy2;
                        ny2 = lane_read(Y, last);
                    }
                    out *= P.ampfactor;
                    if (out != out)
                        out = 0.0f;
                    else if (out > 1.0f)
                        out = 1.0f;
                    else if (out < -1.0f)
                        out = -1.0f;
                    batch_open = true;
                }
                if (lane < n) {
                    const uint32_t i = arg + off + static_cast<uint32_t>(lane);
                    const uint32_t v = kAgcExtra + i;
                    float* dst = (v < nsteps) ? wmain + v : carry + (v - nsteps);
                    *dst = out;
                    if (iqo)
                        iqo[i] = gate ? make_float2(re, im) : make_float2(0.0f, 0.0f);
                }
                agc = lane_read(G, last);
                pr = lane_read(re, last);
                pj = lane_read(im, last);
                pw = lane_read(d, last);
                off += static_cast<unsigned>(n);
                remaining -= n;
            }
            tk_pos += word >> 8;
            if (lane == 0)
                *(aux_vu32*)&x->tk_tail = tk_pos;
        } else if (type == AUX_RESET) {  // ctcss_reset()
            gf_q1 = gf_q2 = gs_q1 = gs_q2 = 0.0f;
            cf_enough = cf_count = cf_has_tone = 0;
            cs_enough = cs_count = cs_has_tone = 0;
        } else if (type == AUX_BATCH) {
            if (lane == 0)
                a.axc[static_cast<size_t>(row) * a.axc_stride + arg] = batch_open ? MI_SIGNAL : MI_NO_SIGNAL;
            if (batch_open)
                active++;
            batch_open = false;
        } else {  // AUX_QUIT
            break;
        }
        ++d_tail;
        if (lane == 0)
            *(aux_vu32*)&x->d_tail = d_tail;
    }
    if (P.ctcss_enabled) {
        if (lane < P.ctcss_fast_ndet)
            cq_fast[lane] = gf_q1, cq_fast[kMaxTones + lane] = gf_q2;
        if (lane < P.ctcss_slow_ndet)
            cq_slow[lane] = gs_q1, cq_slow[kMaxTones + lane] = gs_q2;
    }
    if (lane == 0) {
        auto pf = [&](const int k, const float v) { *(aux_vu32*)&x->fin[k] = __float_as_uint(v); };
        auto pu = [&](const int k, const unsigned v) { *(aux_vu32*)&x->fin[k] = v; };
        auto p64 = [&](const int k, const uint64_t v) { pu(k, static_cast<unsigned>(v)), pu(k + 1, static_cast<unsigned>(v >> 32)); };
        pf(0, agc), pf(1, pr), pf(2, pj), pf(3, pw);
        pf(4, nx0), pf(5, nx1), pf(6, nx2), pf(7, ny0), pf(8, ny1), pf(9, ny2);
        pu(10, static_cast<unsigned>(cf_enough)), pu(11, static_cast<unsigned>(cf_count)), pu(12, static_cast<unsigned>(cf_has_tone));
        pu(13, static_cast<unsigned>(cs_enough)), pu(14, static_cast<unsigned>(cs_count)), pu(15, static_cast<unsigned>(cs_has_tone));
        p64(16, cf_found), p64(18, cf_not_found), p64(20, cs_found), p64(22, cs_not_found);
        p64(24, active);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0)
        *(aux_vu32*)&x->done = 1u;
}

// One channel per workgroup of four waves: the channel itself (as k_demod_uni), the full_ wave and the pre-filter wave ahead of it
// and, behind it, the audio wave.
struct PwShare {
    PreShare pre;
    AuxShare aux;
};
__global__ MI_PW_BOUNDS void k_demod_pw(const DemodArgs a) {
    __shared__ PwShare sh_mem;
    LdsPre* const pre = (LdsPre*)&sh_mem.pre;
    LdsAux* const aux = (LdsAux*)&sh_mem.aux;
    if (threadIdx.x == 0) {
        pre->h_done = 0, pre->m_pos = 0, pre->f_done = 0;
        aux->d_head = 0, aux->d_tail = 0, aux->tk_tail = 0, aux->done = 0;
    }
    __syncthreads();
    if (threadIdx.x >= 192) {
        full_wave(a, pre, static_cast<int>(threadIdx.x) - 192);
        return;
    }
    if (threadIdx.x >= 128) {
        audio_wave(a, aux, static_cast<int>(threadIdx.x) - 128);
        return;
    }
    if (threadIdx.x >= 64) {
        pre_wave<true>(a, pre, static_cast<int>(threadIdx.x) - 64);
        return;
    }
    demod_dispatch<true, true>(a, pre, aux);
}

// Two waves per channel: the channel itself, audio and all, and the pre-filter wave ahead of it (full_ included).  For plans of more
// rows than CUs, up to 1 024: two waves per row leave every SIMD room for stage 1 of the next call, which four waves with a register
// file each do not.
#ifndef MI_PW2_EU
#define MI_PW2_EU 4
#endif
__global__ __launch_bounds__(128, MI_PW2_EU) void k_demod_pw2(const DemodArgs a) {
    __shared__ PreShare sh_mem2;
    LdsPre* const pre = (LdsPre*)&sh_mem2;
    if (threadIdx.x == 0)
        pre->h_done = 0, pre->m_pos = 0, pre->f_done = 0;
    __syncthreads();
    if (threadIdx.x >= 64) {
        pre_wave<false>(a, pre, static_cast<int>(threadIdx.x) - 64);
        return;
    }
    demod_dispatch<true, true>(a, pre, nullptr);
}

__global__ void k_init_state(ChanState* st, float* carry, float* sq_ring, float* ctcss_q, const ChanParams* cp, int nstreams, int nch,
                             int n_ctcss_rows) {
    const int rows = nstreams * nch;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int gsz = gridDim.x * blockDim.x;
    for (int row = gid; row < rows; row += gsz) {
        const ChanParams& p = cp[row % nch];
        ChanState s;
        memset(&s, 0, sizeof(s));
        s.noise_floor = 5.0f;  // Squelch::Squelch, squelch.cpp:36-70
        s.moving_avg_cap = p.using_manual_level ? p.manual_cap : p.cap_factor * s.noise_floor;
        s.pre_full = s.pre_capped = 0.001f;
        s.post_full = s.post_capped = 0.001f;
        s.squelch_level_cache = 0.0f;
        s.next_state = s.current_state = SQ_CLOSED;
        s.sample_count = 0xffffffffu;  // sample_count_ = -1
        s.buffer_head = 0;
        s.buffer_tail = 1;
        s.agcavgfast = 0.5f;    // mk_freqlist, config.cpp:280
        s.prev_waveout = 0.5f;  // config.cpp:332
        s.afc_bin = p.bin;          // dev->bins[i] = dev->base_bins[i], config.cpp:669-670
        s.prev_axc = MI_NO_SIGNAL;  // config.cpp:322
        st[row] = s;
    }
    for (int i = gid; i < rows * kAgcExtra; i += gsz)
        carry[i] = 0.5f;  // channel->waveout[k] = 0.5, config.cpp:321
    for (int i = gid; i < rows * kSquelchRing; i += gsz)
        sq_ring[i] = 0.0f;
    for (size_t i = gid; i < static_cast<size_t>(nstreams) * n_ctcss_rows * 4 * kMaxTones; i += gsz)
        ctcss_q[i] = 0.0f;
}

__global__ void k_iqgen(const IqGenDerived* __restrict__ cfg, const int16_t* __restrict__ tab, uint32_t first_stream, size_t stream_stride,
                        uint64_t first, uint64_t count, unsigned char* __restrict__ out) {
    __shared__ IqGenDerived g;
    __shared__ int16_t stab[1024];
    for (unsigned i = threadIdx.x; i < sizeof(IqGenDerived) / 4; i += blockDim.x)
        reinterpret_cast<uint32_t*>(&g)[i] = reinterpret_cast<const uint32_t*>(cfg)[i];
    for (unsigned i = threadIdx.x; i < 1024; i += blockDim.x)
        stab[i] = tab[i];
    __syncthreads();
    const uint32_t stream = blockIdx.y;
    unsigned char* dst = out + static_cast<size_t>(stream) * stream_stride;
    // 8 complex samples (16 bytes) per lane per iteration
    const uint64_t ngroups = (count + 7) / 8;
    for (uint64_t grp = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; grp < ngroups; grp += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        unsigned char b[16];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            iq_sample(g, stab, first_stream + stream, first + grp * 8 + k, b + 2 * k);
        if (grp * 8 + 8 <= count) {
            *reinterpret_cast<uint4*>(dst + grp * 16) = *reinterpret_cast<uint4*>(b);
        } else {
            for (uint64_t k = 0; grp * 8 + k < count; ++k) {
                dst[grp * 16 + 2 * k] = b[2 * k];
                dst[grp * 16 + 2 * k + 1] = b[2 * k + 1];
            }
        }
    }
}

}  // namespace

hipError_t launch_demod(const DemodArgs& a, hipStream_t s) {
    const int rows = a.rows ? a.nrows : a.nstreams * a.nch;
    if (rows == 0 || a.nsteps == 0)
        return hipSuccess;
    const int blocks = (rows + a.lanes_per_wave - 1) / a.lanes_per_wave;
    if (a.lanes_per_wave == 1 && a.pre_wave == 2 && a.steady_blocks) {
        DemodArgs a2 = a;
        a2.audio_wave = 0;  // (the channel wave keeps the audio)
        hipLaunchKernelGGL(k_demod_pw2, dim3(blocks), dim3(128), 0, s, a2);
    } else if (a.lanes_per_wave == 1 && a.pre_wave && a.steady_blocks)
        hipLaunchKernelGGL(k_demod_pw, dim3(blocks), dim3(256), 0, s, a);
    else if (a.lanes_per_wave == 1)
        hipLaunchKernelGGL(k_demod_uni, dim3(blocks), dim3(64), 0, s, a);
    else
        hipLaunchKernelGGL(k_demod_packed, dim3(blocks), dim3(64), 0, s, a);
    return hipGetLastError();
}

namespace {
// AFC::check<STEP> (rtl_airband.cpp:193-219) on the squared spectrum
__device__ uint32_t afc_check(const float* __restrict__ sq, const uint32_t fft_size, const int step, const uint32_t base, const float base_value,
                              const uint32_t afc) {
    float threshold = 0.0f;
    uint32_t bin;
    for (bin = base;; bin += static_cast<uint32_t>(step)) {
        if (step < 0) {
            if (bin < 1u)
                break;
        } else if (bin + 1u >= fft_size) {
            break;
        }
        const float value = sq[bin + static_cast<uint32_t>(step)];
        if (value <= base_value)
            break;
        if (base == bin) {
            threshold = (value - base_value) / static_cast<float>(afc);
        } else {
            if ((value - base_value) < threshold)
                break;
            threshold = static_cast<float>(static_cast<double>(threshold) + static_cast<double>(threshold) / 10.0);  // `threshold += threshold / 10.0`
        }
    }
    return bin;
}

__global__ void k_afc(const AfcArgs a) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= a.nstreams * a.nch)
        return;
    const int stream = row / a.nch, ch = row - stream * a.nch;
    const ChanParams& P = a.cp[ch];
    if (P.afc == 0)
        return;
    ChanState& s = a.st[row];
    char* pax = a.axc + static_cast<size_t>(row) * a.axc_stride;
    char axc = *pax;
    const int prev = s.prev_axc;
    if (axc != MI_NO_SIGNAL && prev == MI_NO_SIGNAL) {
        const float* __restrict__ sq = a.spec + static_cast<size_t>(stream) * a.fft_size;
        const uint32_t base = P.bin;
        const float base_value = sq[base];
        uint32_t bin = afc_check(sq, static_cast<uint32_t>(a.fft_size), -1, base, base_value, P.afc);
        if (bin == base)
            bin = afc_check(sq, static_cast<uint32_t>(a.fft_size), 1, base, base_value, P.afc);
        if (s.afc_bin != bin) {
            s.afc_bin = bin;
            if (bin > base)
                axc = MI_AFC_UP;
            else if (bin < base)
                axc = MI_AFC_DOWN;
            *pax = axc;
        }
    } else if (axc == MI_NO_SIGNAL && prev != MI_NO_SIGNAL) {
        s.afc_bin = P.bin;
    }
    s.prev_axc = axc;
}
}  // namespace

namespace {
__global__ void k_move_head(float* __restrict__ dst, const float* __restrict__ src, const size_t plane_stride, const int rows, const int* __restrict__ row_list) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= rows * kAgcExtra)
        return;
    const int ri = gid / kAgcExtra, v = gid - ri * kAgcExtra;
    const int r = row_list ? row_list[ri] : ri;
    dst[static_cast<size_t>(r) * plane_stride + v] = src[static_cast<size_t>(r) * plane_stride + v];
}
}  // namespace

hipError_t launch_move_head(float* dst, const float* src, size_t plane_stride, int rows, hipStream_t s, const int* row_list) {
    if (rows == 0 || dst == src)
        return hipSuccess;
    hipLaunchKernelGGL(k_move_head, dim3((rows * kAgcExtra + 255) / 256), dim3(256), 0, s, dst, src, plane_stride, rows, row_list);
    return hipGetLastError();
}

hipError_t launch_afc(const AfcArgs& a, hipStream_t s) {
    const int rows = a.nstreams * a.nch;
    if (rows == 0)
        return hipSuccess;
    hipLaunchKernelGGL(k_afc, dim3((rows + 63) / 64), dim3(64), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_init_state(ChanState* st, float* carry, float* sq_ring, float* ctcss_q, const ChanParams* cp, int nstreams, int nch,
                             int n_ctcss_rows, hipStream_t s) {
    hipLaunchKernelGGL(k_init_state, dim3(64), dim3(256), 0, s, st, carry, sq_ring, ctcss_q, cp, nstreams, nch, n_ctcss_rows);
    return hipGetLastError();
}

hipError_t launch_iqgen(const IqGenDerived* d_cfg, const int16_t* d_tab, uint32_t first_stream, uint32_t nstreams, size_t stream_stride,
                        uint64_t first, uint64_t count, unsigned char* d_out, hipStream_t s) {
    if (count == 0 || nstreams == 0)
        return hipSuccess;
    const uint64_t ngroups = (count + 7) / 8;
    const unsigned bx = static_cast<unsigned>(ngroups / 256 + 1 > 4096 ? 4096 : ngroups / 256 + 1);
    hipLaunchKernelGGL(k_iqgen, dim3(bx, nstreams), dim3(256), 0, s, d_cfg, d_tab, first_stream, stream_stride, first, count, d_out);
    return hipGetLastError();
}

}  // namespace mi
