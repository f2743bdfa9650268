// kernels.hpp -- launch interfaces and device-resident structs shared by the .hip files and the C ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "iqgen.hpp"
#include "plan.hpp"

namespace mi {

// Complete DSP state of one (stream, channel): everything freq_t / channel_t / Squelch / filters carry
// from one batch to the next in the reference.  4- and 8-byte PODs only; it is the checkpoint blob too.
struct ChanState {
    // Squelch (squelch.h:117-158)
    float noise_floor;
    float moving_avg_cap;
    float pre_full, pre_capped;
    float post_full, post_capped;
    float squelch_level_cache;
    int32_t using_post_filter;
    int32_t next_state, current_state;
    int32_t delay;
    int32_t low_signal_count;
    uint32_t sample_count;  // only sample_count % 16 is ever observed (squelch.cpp:212)
    uint32_t recent_open_count, closed_sample_count;
    int32_t buffer_head, buffer_tail;
    uint64_t open_count, flappy_count;
    // CTCSS fast / slow (ctcss.h:86-97); detector q1/q2 live in the ctcss_q table
    int32_t cf_enough, cf_count, cf_has_tone;
    int32_t cs_enough, cs_count, cs_has_tone;
    uint64_t cf_found, cf_not_found, cs_found, cs_not_found;
    // freq_t / channel_t (boondock_airband.h:232-262)
    float agcavgfast;
    float pr, pj, prev_waveout;
    uint32_t dm_phi;
    uint32_t afc_bin;   // dev->bins[i]: the bin stage 1 picks for this channel (moved by AFC, rtl_airband.cpp:224-249)
    int32_t prev_axc;   // channel->axcindicate left by the previous batch (what `AFC afc(dev, i)` captures, :518)
    int32_t pad0;
    uint64_t active_counter;
    // NotchFilter x/y, LowpassFilter xv/yv (filters.h:33-61)
    float notch_x[3], notch_y[3];
    float lp_xr[3], lp_xi[3], lp_yr[3], lp_yi[3];
};

constexpr unsigned kL64Tickets = 32;  // launches of the lane-resident stage 1 that may be in flight at once, generously
struct ChannelizeArgs {
    const unsigned char* iq;  // stream s at iq + s*stream_stride
    size_t stream_stride;
    size_t valid_bytes;       // readable bytes per stream starting at its base
    uint32_t hop_bytes;
    uint32_t nfft;            // windows per stream in this launch
    float* mag;               // [nstreams*nch][plane_stride]
    float2* cplx;             // [nstreams*n_iq_rows][plane_stride]
    size_t plane_stride;
    uint32_t plane_off;       // plane index of window 0
    const float* window;
    const float* tw;
    const float* levels;
    int conv_arith;           // u8: the level table is reproduced by level_u8() (checked by the plan)
    float conv_scale;
    const ChanParams* cp;
    int nch, n_iq_rows;
    unsigned* xmax;  // [nstreams*nch] running max of the magnitudes written (bit pattern; values are >= 0), or null
    PrunePlan prune;          // stage-1 graph pruning (plan.hpp); prune.enabled selects the pruned instantiation
    const float* prune_t1;    // its item-class tables of pass 1 / pass 2 (kPruneClassWords floats per class)
    const float* prune_t2;
    const int* prune_rank;    // [nch] rank of each channel's bin among the distinct picked bins
    L64Plan l64;                   // lane-resident stage 1 (l64_kernel.h); l64.enabled selects it
    const L64Chan* l64_chan;       // [nch] for the plan's own instance
    const L64Chan* l64_chan_full;  // [nch] for the full-graph instance
    const struct L64Jit* l64_jit;  // the plan's own instance (l64_jit.cpp), or null: the full-graph instance runs
    unsigned* l64_tickets;         // [kL64Tickets] device counters, one per launch in turn (zeroed on the launch stream)
    unsigned* l64_ticket_seq;      // host: launches so far
    const ChanState* st;  // AFC handles: the bin of (stream, channel) is st[..].afc_bin; null: ChanParams::bin
    float* afc_spec;      // AFC handles: [nstreams][fft_size] re^2+im^2 of the LAST window of the launch (AFC::square), or null
};

struct DemodArgs {
    int nstreams, nch, n_iq_rows, n_ctcss_rows;
    const int* rows;    // the handle rows (stream * nch + channel) this launch takes, or null: all of them (mixed plans: the rows the
    int nrows;          // time-parallel path does not take)
    uint32_t nsteps;    // samples per channel in this launch (multiple of WAVE_BATCH)
    uint32_t nbatches;
    float* mag;         // [rows][plane_stride]; index AGC_EXTRA+i is the squelch sample of step i
    float2* cplx;
    float* mag_head;    // planes whose first AGC_EXTRA entries receive the carried samples at the end (the same planes, or
    float2* cplx_head;  // the set the next call's stage 1 is already filling)
    size_t plane_stride;
    float* wmain;       // emitted audio, [rows][wmain_stride], nsteps valid
    size_t wmain_stride;
    float* carry;       // [rows][AGC_EXTRA]: lookahead carried between calls
    const float* carry_in;  // ... the one the previous call left, where that is another buffer (null: `carry`)
    float2* iq_out;     // [rows][iq_out_stride] or null
    size_t iq_out_stride;
    char* axc;          // [rows][axc_stride], nbatches written
    uint32_t axc_stride;
    const ChanParams* cp;
    ChanState* st;      // [rows]
    const float* sin_lut;  // 257
    const float* cos_lut;  // 257
    float* sq_ring;        // [rows][kSquelchRing] (only read/written for low-pass channels)
    const float* ctcss_coeff;  // [n_ctcss_rows][2][kMaxTones]
    float* ctcss_q;            // [nstreams][n_ctcss_rows][2][2][kMaxTones]
    mi_channel_stats* stats;   // [rows]
    int fm_quadri;
    int lanes_per_wave;
    int steady_blocks;  // one channel per wave: take runs of steady CLOSED / OPEN steps 64 at a time (demod.hip)
    int pre_wave;       // ... with a second wave per channel that walks the pre-filter averages + noise floor ahead (k_demod_pw)
    int audio_wave;     // ... and, for NFM channels, a third that takes everything behind the filtered I/Q: audio, CTCSS, gate, stores
    unsigned* pre_timeouts;  // (device counter) waits of a channel wave for its pre-filter wave that ran out: expected 0
};

// AFC::finalize for one batch (rtl_airband.cpp:224-249), one thread per (stream, channel)
struct AfcArgs {
    int nstreams, nch, fft_size;
    const ChanParams* cp;
    ChanState* st;
    const float* spec;  // [nstreams][fft_size]
    char* axc;          // [rows][axc_stride]; entry 0 of each row is this batch
    uint32_t axc_stride;
};
hipError_t launch_afc(const AfcArgs& a, hipStream_t s);
// dst[row][0 .. AGC_EXTRA) = src[row][0 .. AGC_EXTRA) for every plane row (the reference's memmove, rtl_airband.cpp:643-646)
hipError_t launch_move_head(float* dst, const float* src, size_t plane_stride, int rows, hipStream_t s, const int* row_list = nullptr);  // (row_list: `rows` handle rows)

// ---- time-parallel stage 2 (tp.hip) ----
// Steps per segment (TpArgs::L): 512 .. 4096, a power of two, chosen per handle from its row count.  One stream x 8 channels
// needs short segments to find any parallelism at all (512: 2000 lanes per minute and channel, each paying TP_W steps of
// warm-up for 512 of work); hundreds of rows have parallelism to spare, and a 2048-step segment pays the same warm-up for
// four times the work.
constexpr uint32_t TP_L_MIN = 512, TP_L_MAX = 4096;
constexpr uint32_t TP_W = 4096;     // warm-up of the state machine / AGC before a segment (a multiple of every segment length)
constexpr uint32_t TP_W1 = 3072;    // warm-up of the full_ sandwich pass: 0.99^3072 = 4e-14 closes the sandwich over 10^6 of dynamic range (unclosed blocks are flagged invalid)
constexpr int TP_MAXEV = 24;        // close-edge fades per segment (they are >= 197 steps apart: at most 21 in 4096 steps)
constexpr uint32_t TP_MAXCHAIN = 32;
constexpr int TP_NREC = 19 + TP_MAXEV;

struct TpCore {
    float nf, cap, c, full;
};
struct TpFinal {
    int cur, next, delay, low, recent, closed;
    float agc;
    int all_ok;
    uint32_t first_bad;
    int d_open, d_flappy;
    int pad;
};

struct TpArgs {
    const int* rows;  // handle rows (stream*nch + ch) taking this path
    int nrows, nch;
    uint32_t nsteps, nbatches, nblk, nseg;  // totals of the call (array strides)
    uint32_t L;                             // steps per segment and per lane of the full_ sandwich pass (TP_L_MIN .. TP_L_MAX)
    // The call is processed in chunks (multiples of lcm(L, WAVE_BATCH) steps: 64000 at L = 512) so that the serial core
    // chain of chunk i+1 overlaps the parallel passes of chunk i on another HIP stream.  Absolute ranges:
    uint32_t step0, step1, seg0, seg1, blk0, blk1, bat0, bat1;
    int first_chunk, last_chunk;
    float* mag;
    size_t plane_stride;
    float* wmain;
    size_t wmain_stride;
    float* carry;             // [handle rows][AGC_EXTRA] audio lookahead this call leaves (one per scratch set)
    const float* carry_prev;  // ... and the one the previous call left: this call's first AGC_EXTRA emitted samples
    char* axc;
    const ChanParams* cp;
    ChanState* st;
    mi_channel_stats* stats;
    const unsigned* xmax;  // [handle rows] bit pattern of the largest magnitude stage 1 wrote so far in this call
    float *blk_fe, *blk_fm, *blk_x0, *blk_xm;  // [nrows][nblk]
    TpCore* core;                              // [nrows][nseg+1]
    TpCore* core_carry;                        // [nrows] chain state handed from one chunk's core kernel to the next
    float* full0;                              // [nrows] pre_filter_.full_ at the start of the call (only when the chain was seeded)
    float* fullbound;                          // [nrows] an upper bound of pre_filter_.full_ at the start of the call
    const float* prev_mag;                     // planes of the previous call (its last TP_W1 steps warm up this call's first lanes), or null
    uint32_t prev_n;                           // steps of the previous call
    const unsigned* xmax_prev;                 // xmax of the previous call
    // spec_head: the segment lanes whose warm-up reaches back over the start of the call (segments 0 .. TP_W / L - 1) begin it in
    // the previous call's arrays from a guessed state, like every other lane, instead of from the carried ChanState -- the segment
    // pass then needs nothing the previous call's tail writes, and the scan checks segment 0 against the carried state as it
    // checks any other segment against its predecessor.
    int spec_head;
    const float *prev_blk_fe, *prev_blk_fm, *prev_blk_x0, *prev_blk_xm;  // [nrows][prev_nblk]
    const TpCore* prev_core;                                              // [nrows][prev_nseg+1]
    uint32_t prev_nblk, prev_nseg;
    int* rec;                                  // [TP_NREC][rec_stride]
    size_t rec_stride;
    int* tstart;                               // [nrows*nseg][8]
    int* need;                                 // [nrows*nseg]
    int redo_listed;                           // k_tp_fix appends to `redo` (the first round; k_tp_settle's rounds do not)
    int* redo;                                 // [1 + nrows*nseg] work list of k_tp_redo: the count, then r * nseg + k of every segment to re-run
    TpFinal* fin;                              // [nrows]
    int* diag;                                 // [nrows][4] segments not accepted in scan 0..3, then [nrows][4] core-chain block counts
    int seg_lpw;                               // lanes per wave of k_tp_seg, 0 = auto (MI_OPT_TP_SEG_LANES)
    int eager_samples;                         // (diagnostic, MI_AIRBAND_TP_EAGER=1) segment lanes request every block's samples a block ahead
    int core_lead;                             // ... how many blocks its noise-floor wave may run ahead (0 = default)
    int core_split;                            // the noise-floor passes of the core chain on a wave of their own (k_tp_core2)
    int agc_hint;                              // segment lanes guess agcavgfast as the channel's last committed value (else 0.5)
    int core_decay;                            // ... the decays after bursts walked ahead by two more waves (decay_wave)
    int core_guess;                            // ... taken by guess-and-verify rounds instead of systolic passes (nf_chain_guess64)
};
inline uint32_t tp_chunk_unit(uint32_t L) {  // lcm(L, WAVE_BATCH = 2000) for L = 2^k >= 16: 2000 = 16 * 125
    return L * 125u;
}

// One chunk in three parts so the caller can put the serial part on its own stream:
hipError_t launch_tp_front(const TpArgs& a, hipStream_t s, bool seed_chain);       // (chain seed on the first chunk) + k_tp_full
hipError_t launch_tp_audio_head(const TpArgs& a, hipStream_t s);                   // emitted audio [0, AGC_EXTRA) = the previous lookahead
hipError_t launch_tp_core(const TpArgs& a, hipStream_t s);                         // k_tp_core
hipError_t launch_tp_seg(const TpArgs& a, hipStream_t s);                          // k_tp_seg (needs core(i) only)
hipError_t launch_tp_rest(const TpArgs& a, hipStream_t s, hipEvent_t* marks);      // k_tp_scan ... k_tp_finish (needs seg(i) and rest(i-1))
hipError_t launch_row_max(const float* x, size_t stride, uint32_t n, int rows, unsigned* xmax, hipStream_t s);  // xmax[row] |= max of x[row][0..n)
constexpr int TP_REST_MARKS = 3;  // marks (optional): after scan#0, after fix#0, after finish


hipError_t launch_channelize(const ChannelizeArgs& a, int log2n, int sfmt, int nstreams, hipStream_t s);
// channelize_l64.hip: the lane-resident N = 512 kernel (launch_channelize takes it when a.l64.enabled)
bool l64_supported(int log2n, size_t hop_bytes, int bytes_per_sample);
hipError_t launch_channelize_l64(const ChannelizeArgs& a, int sfmt, int nstreams, hipStream_t s);
int l64_round_windows(int m6);
int l64_zstride(int m6);
// l64_jit.cpp: the kernel compiled for one plan's masks by hipRTC (cached per (device, hop, masks) for the life of the process;
// null when hipRTC is missing or the compilation fails -- `why` then says so)
const L64Jit* l64_jit_get(int device, int hop, const uint64_t need[6], const char** why);
int l64_jit_minwaves(const L64Jit* j);
void l64_jit_set_cache_dir(const char* dir);            // null / "": no code objects on disk
void l64_jit_counts(int* compiled, int* from_disk);     // kernels compiled / loaded from the cache directory by this process
hipError_t l64_jit_launch(const L64Jit* j, const L64Args& a, unsigned gx, unsigned gy, size_t lds, hipStream_t s);
hipError_t launch_demod(const DemodArgs& a, hipStream_t s);
hipError_t launch_init_state(ChanState* st, float* carry, float* sq_ring, float* ctcss_q, const ChanParams* cp, int nstreams, int nch,
                             int n_ctcss_rows, hipStream_t s);
hipError_t launch_iqgen(const IqGenDerived* d_cfg, const int16_t* d_tab, uint32_t first_stream, uint32_t nstreams, size_t stream_stride,
                        uint64_t first, uint64_t count, unsigned char* d_out, hipStream_t s);

}  // namespace mi
