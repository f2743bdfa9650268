// mi_airband.cpp -- the C ABI of include/mi_airband.h over the HIP kernels.
// No CPU fallback exists: without a HIP device every compute entry point fails with MI_ERR_NO_DEVICE.
#include "../../include/mi_airband.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "kernels.hpp"
#include "plan.hpp"

namespace {

thread_local std::string g_err;

}  // namespace

namespace mi {
std::string& last_error_ref() {  // shared with mixer.hip
    return g_err;
}
}  // namespace mi

namespace {

int fail(int code, const std::string& what) {
    g_err = what;
    return code;
}

int hip_fail(hipError_t e, const char* where) {
    g_err = std::string(where) + ": " + hipGetErrorString(e);
    return (e == hipErrorNoDevice || e == hipErrorInvalidDevice) ? MI_ERR_NO_DEVICE : (e == hipErrorOutOfMemory ? MI_ERR_NOMEM : MI_ERR_HIP);
}

#define HIP_TRY(expr)                      \
    do {                                   \
        hipError_t e__ = (expr);           \
        if (e__ != hipSuccess)             \
            return hip_fail(e__, #expr);   \
    } while (0)
const char kFailedMsg[] = "an earlier call on this handle failed after its state had advanced: destroy it, or restore a checkpoint (mi_demod_set_state)";
// ... after the DSP state of the handle has advanced (enqueue() succeeded): the results of this call are lost and the state cannot be
// rolled back, so the handle is marked and refuses further calls (mi_demod_set_state with a checkpoint revives it)
#define HIP_TRY_F(expr)                                                              \
    do {                                                                             \
        hipError_t e__ = (expr);                                                     \
        if (e__ != hipSuccess) {                                                     \
            h->failed = true;                                                        \
            return hip_fail(e__, #expr);                                             \
        }                                                                            \
    } while (0)

template <class T>
hipError_t dalloc(T** p, size_t count) {
    *p = nullptr;
    if (count == 0)
        return hipSuccess;
    return hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T));
}

}  // namespace

struct mi_plan {
    mi::Plan plan;
};

struct mi_demod {
    mi::Plan plan;
    int gpu = 0;
    int nstreams = 0, nch = 0, rows = 0, max_batches = 0;
    bool first_call = true;  // waveend starts at 0: the first batch needs AGC_EXTRA more windows (config.cpp:808)
    size_t plane_stride = 0;
    hipStream_t own_stream = nullptr;
    // The time-parallel path keeps kSets sets of its per-call scratch (magnitude planes, block aggregates, core snapshots,
    // segment records, timing events) and cycles through them: stage 1, the aggregates, the core chain and the segment
    // passes of a call never touch what the tails of the two calls before it still read, so calls overlap (see enqueue()).
    // `cur` is the set of the last call; the serial path stays on it.
    int cur = 0;
    // a call writes the set of the call six back: the host may then keep four or five calls queued behind the one whose timings it
    // reads (mi_demod_kernel_time_prev, age 4 or 5).  The tail of a call ends 2.8 ms after its core chain starts and stage 1 + the
    // aggregates of a call need 0.85 ms before its own, so with the host reading at age 3 (four sets, rounds 2-3) the period could
    // not fall below (2.8 + 0.85) / 3 = 1.2 ms -- which is what the step took once the chain itself was down to 1.03
    static constexpr int kSets = 6;
    hipEvent_t ev[kSets][5] = {};  // per call: 0 begin, 1 stage 1 done, 2 call done, 3 serial k_demod begins (pipelined serial calls, mixed plans), 4 ... ends (mixed plans)
    static constexpr int kMaxChunks = 64, kEvPerChunk = 13, kSegStreams = 1;
    std::vector<hipEvent_t> chunk_ev[kSets];  // per chunk: stage1 begin/end, full end, core begin/end, seg begin/end, scan0/fix0/finish ends, rest begin
    hipStream_t aux_stream = nullptr;    // carries the serial core chain of the time-parallel path
    hipStream_t front_stream = nullptr;  // stage 1 + aggregates of the time-parallel path
    hipEvent_t ev_entry = nullptr;       // recorded on the caller's stream when a call starts
    hipEvent_t ev_head = nullptr;        // ... and after the audio head of the call has been written
    float* d_mag_set[kSets] = {};  // d_mag aliases d_mag_set[cur]
    float2* d_cplx_set[2] = {};    // pipelined serial calls alternate two plane sets (d_mag_set[0 / 1], d_cplx_set[0 / 1])
    int pset = 0;                  // ... the one holding the carried head
    bool serial_pipe = false;      // the last call was a pipelined serial call
    float* d_mag_last = nullptr;   // ... and these are the planes it worked on (mi_demod_read_planes)
    float2* d_cplx_last = nullptr;
    uint32_t head_off = 0;     // plane index where the AGC_EXTRA carried samples of every row live (0 after a serial call)
    bool steady_blocks = true;  // MI_OPT_STEADY_BLOCKS
    // Tuning switches of this handle (mi_demod_set_option; defaults from the MI_AIRBAND_* environment at mi_demod_create --
    // read per handle, the library keeps no process-wide state)
    int opt_tp = -1;          // MI_OPT_TIME_PARALLEL: -1 auto, 0 serial kernel, 1 time-parallel whenever eligible
    int opt_conv = -1;        // MI_OPT_U8_CONVERSION: -1 auto, 0 level table, 1 arithmetic
    bool opt_prune = true;    // MI_OPT_PRUNE_FFT
    int opt_uni_rows = 4096;  // MI_OPT_UNI_ROWS: up to this many rows keep one channel per wave in k_demod
    int opt_tp_chunks = 0;    // MI_OPT_TP_CHUNKS: 0 = measured default
    double opt_tp_ratio = 0;  // MI_OPT_TP_RATIO_PCT / 100: 0 = measured default
    int opt_tp_lpw = 0;       // MI_OPT_TP_SEG_LANES: lanes per wave of the segment pass, 0 = auto
    int opt_pre_wave = -1;       // MI_OPT_PRE_WAVE: serial kernel, one channel per wave: further waves per channel walk the squelch pre-filter ahead and the audio behind (k_demod_pw); -1 = up to 256 rows
    bool opt_audio_wave = true;  // MI_OPT_AUDIO_WAVE: ... and NFM channels a third wave for everything behind the filtered I/Q (audio, CTCSS, gate, stores)
    bool opt_spec_head = true;   // MI_OPT_SPEC_HEAD: overlapped calls start their first segments from a guessed state (see TpArgs)
    int opt_tp_eager = 0;        // (diagnostic, MI_AIRBAND_TP_EAGER)
    int opt_core_lead = 0;       // (diagnostic, MI_AIRBAND_CORE_LEAD) blocks the noise-floor wave may run ahead, 0 = default
    int opt_agc_hint = 1;        // (diagnostic, MI_AIRBAND_AGC_HINT=0) segment lanes start from agcavgfast = 0.5 instead of the channel's last value
    int opt_core_decay = 1;      // (diagnostic, MI_AIRBAND_CORE_DECAY=0) no decay waves: the walking wave steps every decay itself
    int opt_core_guess = 1;      // (diagnostic, MI_AIRBAND_CORE_GUESS) 0: the noise-floor wave walks systolic passes only; 2: the first guess-and-verify rounds (groups of 64)
    bool opt_core_split = true;  // MI_OPT_CORE_SPLIT: noise-floor passes of the core chain on their own wave (k_tp_core2)
    bool core_split_ok = false;  // ... the plan allows it: automatic squelch levels with a cap factor >= 1 on every channel
    bool opt_l64 = true;      // MI_OPT_LANE_FFT: the lane-resident stage 1 at N = 512 where the plan allows it
    int opt_l64_linear = 0;   // (diagnostic) tiles in blockIdx order instead of grouped per XCD
    bool opt_l64_jit = true;  // MI_OPT_LANE_FFT_JIT: compile the plan's own instance with hipRTC (else the full-graph instance)
    bool early_input = false;  // MI_OPT_EARLY_INPUT: the IQ of a call is valid when the call is made
    bool chain_live = false;   // d_core_carry holds the chain state at the end of the previous call (it was time-parallel)
    hipStream_t seg_stream[kSegStreams] = {nullptr};  // the speculative segment passes (need core(i) only)
    // MI_OPT_RESERVE_CUS: twins of the front and segment streams whose kernels keep off the last `reserve_cus` CUs (see enqueue)
    hipStream_t front_stream_m = nullptr;
    hipStream_t seg_stream_m[kSegStreams] = {nullptr};
    int opt_reserve_cus = -1;  // -1 auto: 32 for handles of up to 64 rows, none beyond; 0 none
    // MI_OPT_SPLIT_CUS: pipelined serial calls: stage 1 keeps off the last n CUs, k_demod runs on them alone; -1 auto (see enqueue)
    int opt_split_cus = -1;
    int split_state = 0;  // 0 undecided, 1 the two CU-masked streams exist, 2 none
    hipStream_t ps_front_m = nullptr, ps_demod_m = nullptr;
    hipEvent_t ev_ps_entry = nullptr, ev_ps_done = nullptr;
    int last_masked = -1;      // which side the previous time-parallel call used
    int masked_state = 0;      // 0 undecided, 1 the masked twins carry the time-parallel passes of this handle, 2 the plain streams do
    int tp_chunks[kSets] = {};
    mi::TpCore* d_core_carry = nullptr;
    float* d_full0 = nullptr;
    float* d_fullbound = nullptr;
    float* d_afc_spec = nullptr;  // [nstreams][fft_size] squared spectrum of the last window of a batch (AFC handles only)
    uint64_t set_seq[kSets] = {};  // call number that last used each event set (0 = never)
    uint64_t call_seq = 0;
    int set_path[kSets] = {};
    // device memory
    float* d_window = nullptr;
    float* d_tw = nullptr;
    float* d_prune_t1 = nullptr;  // stage-1 pruning tables (plan.prune)
    float* d_prune_t2 = nullptr;
    int* d_prune_rank = nullptr;
    L64Chan* d_l64_chan = nullptr;       // per-channel tables of the lane-resident stage 1 (plan.l64): the plan's own instance,
    L64Chan* d_l64_chan_full = nullptr;  // the full-graph instance
    unsigned* d_l64_tickets = nullptr;   // run counters of its launches (kernels.hpp, kL64Tickets)
    unsigned l64_ticket_seq = 0;
    const mi::L64Jit* l64_jit = nullptr; // the kernel compiled for this plan's masks (owned by the process-wide cache), or null
    bool l64_jit_tried = false;
    int last_stage1 = 0;  // MI_STAGE1_* of the last call
    float* d_levels = nullptr;
    float* d_sin = nullptr;
    float* d_cos = nullptr;
    mi::ChanParams* d_cp = nullptr;
    mi::ChanState* d_state = nullptr;
    float* d_mag = nullptr;
    float2* d_cplx = nullptr;
    float* d_carry = nullptr;        // the audio lookahead the last call left: aliases d_carry_set[.]
    float* d_carry_set[kSets] = {};  // time-parallel calls write the one of their scratch set (the next call's segment pass may run
                                     // before this call's tail has applied its fades and the next call has emitted the lookahead)
    float* d_ring = nullptr;
    float* d_ctcss_coeff = nullptr;
    float* d_ctcss_q = nullptr;
    mi_channel_stats* d_stats = nullptr;
    unsigned* d_pre_timeouts = nullptr;  // waits of a channel wave for its pre-filter wave that ran out (k_demod_pw): expected 0
    // staging for the host-buffer entries: three slots, so that one call can be uploaded and one downloaded while a third
    // computes (mi_demod_submit / mi_demod_wait; mi_demod_process uses slot 0 alone).  A time-parallel call has ~2.5 ms of
    // latency whatever its length (segment pass -> scan -> fix -> ...), so with two slots a 16-s call could not be fed faster
    // than one per (upload + latency) / 2.
    static constexpr int kSlots = 3;
    struct Slot {
        unsigned char* d_iq = nullptr;   // [nstreams][iq_stride]
        float* d_wout = nullptr;         // [rows][max steps + AGC_EXTRA]: emitted audio + lookahead, the host layout
        float2* d_iqout = nullptr;       // [rows][max steps]
        char* d_axc = nullptr;           // [rows][max batches]
        mi_channel_stats* d_stats = nullptr;  // [rows] snapshot of the statistics after this call
        unsigned char* h_in = nullptr;   // pinned: upload staging for sources that are not pinned themselves
        unsigned char* h_out = nullptr;  // pinned: audio / raw I/Q / flags / statistics on their way back
        hipEvent_t up_done = nullptr, done = nullptr;
        bool busy = false;
        // where the results of the call in flight go
        int nbatches = 0;
        float* waveout = nullptr;
        float* iq_out = nullptr;
        char* axc = nullptr;
        mi_channel_stats* stats = nullptr;
        bool wave_direct = false;  // waveout is page-locked: the audio is downloaded straight into it
    } slot[kSlots];
    size_t iq_stride = 0, h_out_bytes = 0;
    bool slots_ready[kSlots] = {};
    int slot_next = 0, slot_oldest = 0, in_flight = 0;
    bool failed = false;  // a call advanced the DSP state and then could not deliver its results: every further call is refused
    hipStream_t copy_stream = nullptr;  // uploads of submitted calls
    hipStream_t down_stream = nullptr;  // their downloads
    // time-parallel stage 2 (tp.hip): the plain AM channels of the plan.  A mixed plan (tp_mixed) sends those rows down the time-parallel
    // path and the others through the serial kernel in the same call (MI_OPT_MIXED_PLAN), on a stream of its own beside the chain.
    bool tp_eligible = false;
    bool tp_mixed = false;
    bool opt_mixed = true;
    int tp_rows = 0, ser_rows = 0;       // rows of either kind (tp_rows + ser_rows == rows)
    int* d_srows = nullptr;              // the serial kernel's rows of a mixed plan (d_rows: the time-parallel path's)
    hipStream_t ser_stream = nullptr;    // ... and its stream
    hipEvent_t ev_cplx_free[2] = {};     // the serial kernel of a mixed call has read complex plane set p
    bool cplx_busy[2] = {};
    bool ser_head_next = false;          // the serial kernel of the last (mixed) call left its rows' carried samples in the next plane set
    bool set_mixed[kSets] = {};
    int last_path = 0;  // 0 = serial kernel, 1 = time-parallel
    int* d_rows = nullptr;
    unsigned* d_xmax[kSets] = {};
    float *d_blk_fe[kSets] = {}, *d_blk_fm[kSets] = {}, *d_blk_x0[kSets] = {}, *d_blk_xm[kSets] = {};
    mi::TpCore* d_core[kSets] = {};
    int* d_rec[kSets] = {};  // per scratch set: the segment passes of the next call write theirs while this call's tail reads its own
    const float* prev_out_lo = nullptr;  // audio buffer of the previous call (its tail may still be writing it)
    const float* prev_out_hi = nullptr;
    // mi_demod_process_planes (test entry): stage 1 is replaced by a copy of caller-supplied planes, [row][inject_count]
    const float* inject_mag = nullptr;
    const float2* inject_cplx = nullptr;
    size_t inject_count = 0;
    const float* set_out_lo[kSets] = {};  // ... and of the time-parallel calls that used each scratch set last
    const float* set_out_hi[kSets] = {};
    int* d_tstart = nullptr;
    int* d_need = nullptr;
    int* d_redo = nullptr;  // [1 + rows*max_seg]: count, then the (row, segment) indices k_tp_fix leaves for k_tp_redo
    mi::TpFinal* d_fin = nullptr;
    int* d_diag = nullptr;
    uint32_t last_nseg[kSets] = {};
    size_t tp_max_blk = 0, tp_max_seg = 0;
    uint32_t tp_L = 512;  // steps per segment (kernels.hpp, TP_L_MIN .. TP_L_MAX), fixed when the handle is created
    int opt_tp_L = 0;     // MI_AIRBAND_TP_SEGMENT at create: 0 = by row count
};

namespace {

int n_fft_for(const mi_demod* h, int nbatches) {
    return nbatches * mi::kWaveBatch + (h->first_call ? mi::kAgcExtra : 0);
}

constexpr int kTpMinBatches = 8;  // below this the segments are too few to pay for the extra passes
// Above this many rows the serial kernel is the faster path by itself: its time is the latency of one row (≈ 20-30 ns per step since
// round 3) whatever the number of rows, while the wide passes of the time-parallel path grow with them -- 8 / 16 / 32 / 64 streams x 8
// AM channels, 8-s calls: 1.35 / 1.67 / 2.60 / 4.69 ms time-parallel against 2.4 / 2.5 / 2.5 / 3.9 ms for k_demod, which also leaves
// stage 1 of the next call more of every SIMD.  MI_OPT_TIME_PARALLEL = 1 still forces the time-parallel path.
constexpr int kTpAutoMaxRows = 256;
// A mixed plan's call is as long as the longer of its two halves, and the time-parallel half has a fixed latency of a millisecond and
// more whatever the call's length (segment pass, scan, fix ...), while the serial kernel takes 0.3 ms per second of signal for every
// row: the split pays from about six seconds per call on (8 streams x 32 mixed channels, 2 / 4 / 8 s per call: 1.36 / 1.83 / 2.24 ms
// split against 0.70 / 1.41 / 2.58 whole; 1 stream x 32 at fft 2048, 8 s: 1.85 against 2.51).
constexpr int kMixedMinBatches = 64;

// Defaults of a new handle's tuning switches from the caller's environment (A/B measurements, tests):
//   MI_AIRBAND_TP=0|1        serial kernel / time-parallel path whenever eligible
//   MI_AIRBAND_PRUNE=0       full FFT graph at N = 512 (the pruned one is bit-exact and faster where it applies)
//   MI_AIRBAND_CONV=lut|arith  u8 conversion through the level table / the arithmetic form the plan has checked against it
//   MI_AIRBAND_STEADY=0      serial stage 2 takes every step in the sample loop
//   MI_AIRBAND_TP_SEGMENT=512|1024|2048|4096  steps per segment of the time-parallel path (default: by row count; sizes the scratch,
//                            so it is read when the handle is created and has no mi_demod_set_option twin)
//   MI_AIRBAND_L64=0         no lane-resident stage 1 at N = 512 (the pruned / full exchange kernels instead)
//   MI_AIRBAND_UNI_ROWS=n, MI_AIRBAND_TP_CHUNKS=n, MI_AIRBAND_TP_RATIO=x, MI_AIRBAND_TP_LPW=n
void tuning_from_env(mi_demod* h) {
    // (MI_AIRBAND_DEBUG=1: every tuning variable found in the environment is named on stderr when a handle is created -- a variable
    //  exported for a test changes what a production handle does just as silently as it changes the test's)
    static const bool debug = [] {
        const char* e = std::getenv("MI_AIRBAND_DEBUG");
        return e && *e && std::atoi(e) != 0;
    }();
    auto get = [](const char* k) -> const char* {
        const char* e = std::getenv(k);
        if (e && *e && debug)
            std::fprintf(stderr, "mi_airband: %s=%s (from the environment)\n", k, e);
        return (e && *e) ? e : nullptr;
    };
    if (const char* e = get("MI_AIRBAND_TP"))
        h->opt_tp = std::atoi(e) != 0 ? 1 : 0;
    if (const char* e = get("MI_AIRBAND_CONV"))
        h->opt_conv = (e[0] == 'a' || e[0] == 'A') ? 1 : 0;
    if (const char* e = get("MI_AIRBAND_STEADY"))
        h->steady_blocks = std::atoi(e) != 0;
    if (const char* e = get("MI_AIRBAND_PRUNE"))
        h->opt_prune = std::atoi(e) != 0;
    if (const char* e = get("MI_AIRBAND_UNI_ROWS"))
        h->opt_uni_rows = std::max(1, std::atoi(e));
    if (const char* e = get("MI_AIRBAND_TP_CHUNKS"))
        h->opt_tp_chunks = std::max(1, std::atoi(e));
    if (const char* e = get("MI_AIRBAND_TP_RATIO"))
        h->opt_tp_ratio = std::max(0.25, std::atof(e));
    if (const char* e = get("MI_AIRBAND_TP_SEGMENT")) {
        const int v = std::atoi(e);
        h->opt_tp_L = (v == 512 || v == 1024 || v == 2048 || v == 4096) ? v : 0;
    }
    if (const char* e = get("MI_AIRBAND_CORE_SPLIT"))
        h->opt_core_split = std::atoi(e) != 0;
    if (const char* e = get("MI_AIRBAND_TP_EAGER"))
        h->opt_tp_eager = std::atoi(e) != 0;
    if (const char* e = get("MI_AIRBAND_CORE_LEAD"))
        h->opt_core_lead = std::max(0, std::atoi(e));
    if (const char* e = get("MI_AIRBAND_RESERVE_CUS"))
        h->opt_reserve_cus = std::max(-1, std::atoi(e));
    if (const char* e = get("MI_AIRBAND_SPLIT_CUS"))
        h->opt_split_cus = std::max(-1, std::atoi(e));
    if (const char* e = get("MI_AIRBAND_AGC_HINT"))
        h->opt_agc_hint = std::atoi(e) != 0;
    if (const char* e = get("MI_AIRBAND_CORE_DECAY"))
        h->opt_core_decay = std::atoi(e) != 0;
    if (const char* e = get("MI_AIRBAND_CORE_GUESS"))
        h->opt_core_guess = std::max(0, std::min(2, std::atoi(e)));
    if (const char* e = get("MI_AIRBAND_PRE_WAVE"))
        h->opt_pre_wave = std::atoi(e) < 0 ? -1 : std::min(2, std::atoi(e));
    if (const char* e = get("MI_AIRBAND_MIXED"))
        h->opt_mixed = std::atoi(e) != 0;
    if (const char* e = get("MI_AIRBAND_AUDIO_WAVE"))
        h->opt_audio_wave = std::atoi(e) != 0;
    if (const char* e = get("MI_AIRBAND_SPEC_HEAD"))
        h->opt_spec_head = std::atoi(e) != 0;
    if (const char* e = get("MI_AIRBAND_L64"))
        h->opt_l64 = std::atoi(e) != 0;
    if (const char* e = get("MI_AIRBAND_L64_JIT"))
        h->opt_l64_jit = std::atoi(e) != 0;
    if (const char* e = get("MI_AIRBAND_L64_WGS"))
        h->opt_l64_linear = std::max(0, std::atoi(e));
    if (const char* e = get("MI_AIRBAND_TP_LPW")) {
        const int v = std::atoi(e);
        h->opt_tp_lpw = (v >= 1 && v <= 64) ? v : 0;
    }
}

int slot_prepare(mi_demod* h, int k);  // (defined with the host-buffer entries below)

// The plan's own instance of the lane-resident stage 1 (hipRTC, or the code object an earlier start left on disk): 0.3-0.6 s
// when it has to be compiled, so it is asked for when the handle is created -- before any input thread fills a ring.
void stage1_compile(mi_demod* h) {
    if (h->l64_jit_tried || !h->opt_l64_jit || !h->opt_l64 || !h->plan.l64.enabled || !h->d_l64_chan)
        return;
    h->l64_jit_tried = true;
    const int hop = static_cast<int>(h->plan.hop_bytes / (2 * static_cast<size_t>(h->plan.bytes_per_sample)));
    h->l64_jit = mi::l64_jit_get(h->gpu, hop, h->plan.l64.need, nullptr);
}

int lanes_per_wave_for(const mi_demod* h) {
    // up to opt_uni_rows waves keep one channel each (the uniform instantiation of k_demod); beyond that pack lanes
    int lpw = (h->rows + h->opt_uni_rows - 1) / h->opt_uni_rows;
    return std::min(64, std::max(1, lpw));
}

// shared by both entry points; everything is enqueued on `s`
// the second plane set of the pipelined serial path, allocated the first time it is wanted
bool serial_sets_ready(mi_demod* h) {
    if (h->d_mag_set[1] && (h->d_cplx_set[1] || !h->d_cplx_set[0]))
        return true;
    const size_t rows = static_cast<size_t>(h->rows);
    if (!h->d_mag_set[1]) {
        float* m = nullptr;
        if (hipMalloc(reinterpret_cast<void**>(&m), rows * h->plane_stride * 4) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        (void)hipMemset(m, 0, rows * h->plane_stride * 4);
        h->d_mag_set[1] = m;
    }
    if (h->d_cplx_set[0] && !h->d_cplx_set[1]) {
        const size_t zn = static_cast<size_t>(h->nstreams) * h->plan.n_iq_rows * h->plane_stride;
        float2* z = nullptr;
        if (hipMalloc(reinterpret_cast<void**>(&z), zn * 8) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        (void)hipMemset(z, 0, zn * 8);
        h->d_cplx_set[1] = z;
    }
    return true;
}

int enqueue(mi_demod* h, const unsigned char* d_iq, size_t stream_stride, size_t valid_bytes, int nbatches, float* d_wmain, size_t wmain_stride,
            float2* d_iq_out, size_t iq_out_stride, char* d_axc, hipStream_t s, hipEvent_t iq_ready = nullptr) {
    // iq_ready (host-buffer entry, calls in flight): the IQ becomes valid when this event fires -- the streams that read it wait
    // for it and for nothing else, exactly as if the caller had vouched for the bytes (MI_OPT_EARLY_INPUT)
    const bool early_input = h->early_input || iq_ready != nullptr;
    const int nfft = n_fft_for(h, nbatches);
    mi::ChannelizeArgs ca{};
    ca.iq = d_iq;
    ca.stream_stride = stream_stride;
    ca.valid_bytes = valid_bytes;
    ca.hop_bytes = static_cast<uint32_t>(h->plan.hop_bytes);
    ca.nfft = static_cast<uint32_t>(nfft);
    ca.mag = h->d_mag;
    ca.cplx = h->d_cplx;
    ca.plane_stride = h->plane_stride;
    ca.plane_off = h->first_call ? 0 : mi::kAgcExtra;
    ca.window = h->d_window;
    ca.tw = h->d_tw;
    ca.prune = h->plan.prune;
    ca.prune.enabled = (ca.prune.enabled && h->opt_prune) ? 1 : 0;
    ca.prune_t1 = h->d_prune_t1;
    ca.prune_t2 = h->d_prune_t2;
    ca.prune_rank = h->d_prune_rank;
    ca.l64 = h->plan.l64;
    ca.l64.enabled = (ca.l64.enabled && h->opt_l64 && h->d_l64_chan) ? 1 : 0;
    ca.l64.linear_tiles = 0;
    ca.l64.wg_per_cu = h->opt_l64_linear;  // (MI_AIRBAND_L64_WGS: workgroups per CU of the persistent stage-1 launch, 0 = default)
    ca.l64_chan = h->d_l64_chan;
    ca.l64_tickets = h->d_l64_tickets;
    ca.l64_ticket_seq = &h->l64_ticket_seq;
    ca.l64_chan_full = h->d_l64_chan_full;
    if (ca.l64.enabled)  // (normally done by mi_demod_create; here only if the option was switched on afterwards)
        stage1_compile(h);
    ca.l64_jit = h->opt_l64_jit ? h->l64_jit : nullptr;
    // The prebuilt full-graph instance keeps all 64 points of a lane live and is slower than the exchange kernels: it runs
    // only when asked for (MI_OPT_LANE_FFT_JIT = 0, tests); without hipRTC the pruned / full exchange kernels take over.
    if (ca.l64.enabled && h->opt_l64_jit && !ca.l64_jit)
        ca.l64.enabled = 0;
    ca.levels = h->d_levels;
    {
        const bool pruned = ca.prune.enabled && h->plan.log2n == 9 && !h->plan.any_afc;
        const int cc = h->opt_conv;
        ca.conv_arith = (h->plan.conv_arith && (cc < 0 ? !pruned : cc == 1)) ? 1 : 0;
    }
    ca.conv_scale = h->plan.conv_scale;
    ca.cp = h->d_cp;
    ca.nch = h->nch;
    ca.n_iq_rows = h->plan.n_iq_rows;
    h->last_stage1 = (ca.l64.enabled && h->plan.log2n == 9 && !h->plan.any_afc) ? (ca.l64_jit ? MI_STAGE1_LANE_PLAN : MI_STAGE1_LANE_FULL)
                     : ((ca.prune.enabled && h->plan.log2n == 9 && !h->plan.any_afc) ? MI_STAGE1_EXCHANGE_PRUNED : MI_STAGE1_EXCHANGE_FULL);
    const int env = h->opt_tp;
    const bool use_tp = h->tp_eligible && env != 0 && (!h->tp_mixed || (h->opt_mixed && serial_sets_ready(h))) &&
                        (env == 1 || (nbatches >= (h->tp_mixed ? kMixedMinBatches : kTpMinBatches) && h->tp_rows <= kTpAutoMaxRows));
    ca.xmax = nullptr;  // (the time-parallel branch points it at its scratch set)

    mi::DemodArgs da{};
    da.nstreams = h->nstreams;
    da.nch = h->nch;
    da.rows = nullptr;
    da.nrows = h->rows;
    da.carry_in = nullptr;
    da.n_iq_rows = h->plan.n_iq_rows;
    da.n_ctcss_rows = h->plan.n_ctcss_rows;
    da.nsteps = static_cast<uint32_t>(nbatches) * mi::kWaveBatch;
    da.nbatches = static_cast<uint32_t>(nbatches);
    da.mag = h->d_mag;
    da.cplx = h->d_cplx;
    da.mag_head = h->d_mag;
    da.cplx_head = h->d_cplx;
    da.plane_stride = h->plane_stride;
    da.wmain = d_wmain;
    da.wmain_stride = wmain_stride;
    da.carry = h->d_carry;
    da.iq_out = d_iq_out;
    da.iq_out_stride = iq_out_stride;
    da.axc = d_axc;
    da.axc_stride = static_cast<uint32_t>(nbatches);
    da.cp = h->d_cp;
    da.st = h->d_state;
    da.sin_lut = h->d_sin;
    da.cos_lut = h->d_cos;
    da.sq_ring = h->d_ring;
    da.ctcss_coeff = h->d_ctcss_coeff;
    da.ctcss_q = h->d_ctcss_q;
    da.stats = h->d_stats;
    da.fm_quadri = h->plan.dev.fm_quadri;
    da.lanes_per_wave = lanes_per_wave_for(h);
    da.steady_blocks = h->steady_blocks ? 1 : 0;
    // the pre-filter wave pays where a call is bound by the latency of its rows (4 / 8 / 16 streams x 32 mixed channels: +44 / +37 /
    // +12 %); with a thousand rows and more the machine is full and a second wave per row only takes LDS and issue slots from
    // stage 1 (32 streams: +-0, 64 streams: -27 %)
    // (round 3: four waves per channel, each with a SIMD's register file to itself: one channel per CU, so up to 256 rows)
    // ... and two waves per channel (the channel with its audio, the pre-filter wave) up to 1 024 rows: 2 = k_demod_pw2
    da.pre_wave = h->opt_pre_wave < 0 ? (h->rows <= 256 ? 1 : (h->rows <= 1024 ? 2 : 0)) : std::min(2, h->opt_pre_wave);
    da.audio_wave = h->opt_audio_wave ? 1 : 0;
    da.pre_timeouts = h->d_pre_timeouts;

    // the serial kernels expect the carried AGC_EXTRA samples of every row at the front of the planes they work on
    auto head_in_place = [&]() -> int {
        if (h->head_off != 0) {
            HIP_TRY(mi::launch_move_head(h->d_mag, h->d_mag + h->head_off, h->plane_stride, h->rows, s));
            h->head_off = 0;
        }
        return MI_OK;
    };
    // stage 1 of the windows [f0, f0 + cc.nfft) of this call -- or, for mi_demod_process_planes, the caller's planes in their place
    auto stage1_launch = [&](const mi::ChannelizeArgs& cc, uint32_t f0, hipStream_t st) -> hipError_t {
        if (!h->inject_mag)
            return mi::launch_channelize(cc, h->plan.log2n, h->plan.dev.sfmt, h->nstreams, st);
        hipError_t e = hipMemcpy2DAsync(cc.mag + cc.plane_off, cc.plane_stride * 4, h->inject_mag + f0, h->inject_count * 4, static_cast<size_t>(cc.nfft) * 4,
                                        static_cast<size_t>(h->rows), hipMemcpyDeviceToDevice, st);
        const size_t zrows = static_cast<size_t>(h->nstreams) * h->plan.n_iq_rows;
        if (e == hipSuccess && zrows && h->inject_cplx)
            e = hipMemcpy2DAsync(cc.cplx + cc.plane_off, cc.plane_stride * 8, h->inject_cplx + f0, h->inject_count * 8, static_cast<size_t>(cc.nfft) * 8, zrows,
                                 hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess && cc.xmax)
            e = mi::launch_row_max(cc.mag + cc.plane_off, cc.plane_stride, cc.nfft, h->rows, cc.xmax, st);
        return e;
    };
    hipEvent_t* evc = h->ev[h->cur];  // (the time-parallel and the pipelined serial branch switch to the next set)
    bool pipelined_serial = false;
    if (use_tp) {
        // ---- time-parallel stage 2, pipelined over chunks of the call and across calls ----
        // The exact core chain (k_tp_core) is serial per channel and latency bound on 8 waves; everything else is wide.
        //   front stream : head carry, then per chunk stage 1 + k_tp_full           (needs the IQ; chunk 0's k_tp_full needs
        //                                                                             the chain state at the call start)
        //   aux stream   : k_tp_core(i) as soon as chunk i's aggregates exist         (one chain across chunks AND calls)
        //   seg streams  : k_tp_seg(i) as soon as core(i) is done and the previous call has left its final state
        //   caller's     : audio head, then scan / fix / redo / finish of chunk i after seg(i) and the chain of chunk i-1
        // With MI_OPT_EARLY_INPUT the front and aux streams do not wait for the caller's stream, i.e. for the segment
        // and fix passes of the previous call: consecutive calls overlap and the core chain runs back to back.
        const int q = (h->cur + 1) % mi_demod::kSets;  // the scratch set of this call
        float* const planes = h->d_mag_set[q];
        const bool overlap = early_input && h->chain_live && !h->first_call;
        const float* out_lo = d_wmain;
        const float* out_hi = d_wmain + static_cast<size_t>(h->rows - 1) * wmain_stride + da.nsteps;
        // segment passes may run under the previous call's tail only if they write a different audio buffer
        const bool seg_early = overlap && (out_hi <= h->prev_out_lo || out_lo >= h->prev_out_hi);
        const uint32_t n = da.nsteps;
        const uint32_t L = h->tp_L;
        const uint32_t chunk_unit = mi::tp_chunk_unit(L);
        const uint32_t units = n / chunk_unit;
        // Chunk sizes grow geometrically: a short first chunk gets the serial core chain going early (its stage 1 +
        // aggregates are all that precedes it), later chunks are long because every wide pass has a fixed latency per
        // launch.  MI_AIRBAND_TP_CHUNKS / MI_AIRBAND_TP_RATIO override the measured defaults.
        // An isolated call: 3 chunks growing by 1.5x (round 1; see below).  When calls overlap the chain is already running and stage 1 of this
        // call hides under the previous call: one chunk then -- every chunk boundary costs the chain a launch gap and the
        // tail passes on the caller's stream (scan / fix / redo / settle / finish) their fixed latencies once more, and with
        // two chunks those passes took as long per call as the chain itself (1 / 2 / 3 / 4 chunks over 20 steps: 2.21 / 2.40 /
        // 2.9 / 3.5 ms per step; over 5 steps, where the last call's drain weighs more, 1 and 2 are level).
        // (that is the few-rows case, where the per-channel chain is the critical path; with hundreds of rows the wide passes
        // are, and two chunks let stage 1 of the second run under the segment / fix passes of the first: 64 streams x 8
        // channels 168 vs 147 GS/s)
        // (isolated calls, round 2: with the chain on three waves an isolated call is a sum of fixed latencies -- stage 1, aggregates,
        // chain, segment pass, scan, fix, finish -- and every chunk adds the last four once more: 2 chunks, the second twice the
        // first, 3.7 instead of 4.1 ms per 64-s call and 2.6 instead of 3.3 per 16-s call; with many rows 2, 3 and 4 are level)
        int want = overlap ? (h->tp_rows <= 64 ? 1 : 2) : (h->tp_rows <= 64 ? 2 : 3);
        double ratio = overlap ? 1.0 : (h->tp_rows <= 64 ? 2.0 : 1.5);
        if (h->opt_tp_chunks > 0)
            want = h->opt_tp_chunks;
        if (h->opt_tp_ratio > 0)
            ratio = h->opt_tp_ratio;
        std::vector<uint32_t> bound{0};  // chunk i covers units [bound[i], bound[i+1])
        if (units > 0) {
            want = std::min<int>(want, static_cast<int>(units));
            double total = 0.0, w = 1.0;
            for (int i = 0; i < want; ++i, w *= ratio)
                total += w;
            double acc = 0.0;
            w = 1.0;
            for (int i = 0; i < want; ++i, w *= ratio) {
                acc += w;
                uint32_t bd = (i == want - 1) ? units : static_cast<uint32_t>(acc / total * units + 0.5);
                bd = std::max(bd, bound.back() + 1);
                bd = std::min(bd, units - static_cast<uint32_t>(want - 1 - i));
                bound.push_back(bd);
            }
        } else {
            bound.push_back(0);
        }
        const int C = static_cast<int>(bound.size()) - 1;
        if (C > mi_demod::kMaxChunks)
            return fail(MI_ERR_INVALID, "too many chunks");
        std::vector<hipEvent_t>& cev = h->chunk_ev[q];
        while (static_cast<int>(cev.size()) < C * mi_demod::kEvPerChunk) {
            hipEvent_t e = nullptr;
            HIP_TRY(hipEventCreate(&e));
            cev.push_back(e);
        }
        evc = h->ev[q];
        mi::TpArgs ta{};
        ta.rows = h->d_rows;
        ta.nrows = h->tp_rows;
        ta.nch = h->nch;
        ta.nsteps = n;
        ta.nbatches = da.nbatches;
        ta.nblk = n / 16;
        ta.L = L;
        ta.nseg = (n + L - 1) / L;
        ta.mag = planes;
        ta.plane_stride = h->plane_stride;
        ta.wmain = d_wmain;
        ta.wmain_stride = wmain_stride;
        ta.carry = h->d_carry_set[q];
        ta.carry_prev = h->d_carry;
        ta.axc = d_axc;
        ta.cp = h->d_cp;
        ta.st = h->d_state;
        ta.stats = h->d_stats;
        ta.xmax = h->d_xmax[q];
        ta.blk_fe = h->d_blk_fe[q];
        ta.blk_fm = h->d_blk_fm[q];
        ta.blk_x0 = h->d_blk_x0[q];
        ta.blk_xm = h->d_blk_xm[q];
        ta.core = h->d_core[q];
        ta.core_carry = h->d_core_carry;
        ta.full0 = h->d_full0;
        ta.fullbound = h->d_fullbound;
        ta.prev_mag = overlap ? h->d_mag : nullptr;  // (still the previous call's planes here)
        ta.prev_n = h->head_off;
        ta.xmax_prev = h->d_xmax[h->cur];
        ta.rec = h->d_rec[q];
        ta.rec_stride = static_cast<size_t>(h->rows) * h->tp_max_seg;
        ta.tstart = h->d_tstart;
        ta.need = h->d_need;
        ta.redo = h->d_redo;
        ta.fin = h->d_fin;
        ta.diag = h->d_diag;
        ta.seg_lpw = h->opt_tp_lpw;
        ta.core_split = (h->opt_core_split && h->core_split_ok) ? 1 : 0;
        ta.core_lead = h->opt_core_lead;
        ta.core_guess = h->opt_core_guess;
        ta.core_decay = h->opt_core_decay;
        ta.agc_hint = h->opt_agc_hint;
        ta.eager_samples = h->opt_tp_eager;
        // Speculative head: when this call's segment pass may run under the previous call's tail at all (seg_early) and that call
        // left what the warm-up needs (aggregates, core states at boundaries of the same segment length, TP_W steps of them),
        // no lane starts from the carried ChanState and no launch of the pass waits for the previous call.
        const bool spec_head = seg_early && h->opt_spec_head && h->head_off >= mi::TP_W && h->set_seq[h->cur] &&
                               h->set_path[h->cur] == 1;
        ta.spec_head = spec_head ? 1 : 0;
        ta.prev_blk_fe = h->d_blk_fe[h->cur], ta.prev_blk_fm = h->d_blk_fm[h->cur];
        ta.prev_blk_x0 = h->d_blk_x0[h->cur], ta.prev_blk_xm = h->d_blk_xm[h->cur];
        ta.prev_core = h->d_core[h->cur];
        ta.prev_nblk = h->head_off / 16;
        ta.prev_nseg = h->last_nseg[h->cur];
        h->last_nseg[q] = ta.nseg;
        auto chunk = [&](int i) {
            mi::TpArgs c = ta;
            c.step0 = bound[static_cast<size_t>(i)] * chunk_unit;
            c.step1 = (i == C - 1) ? n : bound[static_cast<size_t>(i) + 1] * chunk_unit;
            c.seg0 = c.step0 / L;
            c.seg1 = (c.step1 + L - 1) / L;
            c.blk0 = c.step0 / 16;
            c.blk1 = c.step1 / 16;
            c.bat0 = c.step0 / mi::kWaveBatch;
            c.bat1 = c.step1 / mi::kWaveBatch;
            c.first_chunk = i == 0;
            c.last_chunk = i == C - 1;
            return c;
        };
        const bool first_call = h->first_call;
        // A mixed plan: stage 1 leaves the raw bins of this call in complex plane set p, the serial kernel reads them there and leaves
        // its carried head in the other set for the next call (as the pipelined serial calls do).
        const int zp = (h->d_cplx && h->d_cplx == h->d_cplx_set[1]) ? 1 : 0, znp = zp ^ 1;
        if (h->tp_mixed) {
            ca.cplx = h->d_cplx_set[zp];
        }
        // The wide passes of a call (stage 1, aggregates, segment pass: thousands of waves that hold most of a SIMD's registers for a
        // millisecond) run beside the latency-bound kernels of its neighbours: the core chains, and the tail (scan / fix / redo /
        // settle / finish: a handful of lanes, 280-430 VGPRs a wave), which then wait for a wide wave to retire before they can
        // start at all -- 1.5 ms per call for 0.65 ms of work.  So on plans of few rows the wide passes keep off a few CUs
        // (hipExtStreamCreateWithCUMask), where the others always find room.  Such a stream is a blocking one (it synchronises
        // with the NULL stream): the twins are created at the handle's first time-parallel call and used by every call whose stream
        // is not the NULL stream.
        if (h->masked_state == 0) {
            const int want = h->opt_reserve_cus >= 0 ? h->opt_reserve_cus : (h->tp_rows <= 64 ? 32 : 0);
            h->masked_state = 2;
            hipDeviceProp_t prop{};
            if (want > 0 && hipGetDeviceProperties(&prop, h->gpu) == hipSuccess && prop.multiProcessorCount >= want + 32) {
                const int ncu = prop.multiProcessorCount, keep = ncu - want;
                std::vector<uint32_t> mask(static_cast<size_t>((ncu + 31) / 32), 0u);
                for (int i = 0; i < keep; ++i)
                    mask[static_cast<size_t>(i) / 32] |= 1u << (i % 32);
                hipError_t me = hipExtStreamCreateWithCUMask(&h->front_stream_m, static_cast<uint32_t>(mask.size()), mask.data());
                for (hipStream_t& ssm : h->seg_stream_m)
                    if (me == hipSuccess)
                        me = hipExtStreamCreateWithCUMask(&ssm, static_cast<uint32_t>(mask.size()), mask.data());
                if (me == hipSuccess)
                    h->masked_state = 1;
                else
                    (void)hipGetLastError();  // (no such streams here: the plain ones serve)
            }
        }
        // (a call on the NULL stream takes the plain streams whatever the handle decided; changing sides between calls is rare and
        //  costs a host wait: the passes of consecutive calls are ordered by their stream, not by events)
        const bool masked = h->masked_state == 1 && s != nullptr;
        if (h->last_masked >= 0 && h->last_masked != (masked ? 1 : 0)) {
            HIP_TRY(hipStreamSynchronize(h->last_masked ? h->front_stream_m : h->front_stream));
            for (int i = 0; i < mi_demod::kSegStreams; ++i)
                HIP_TRY(hipStreamSynchronize(h->last_masked ? h->seg_stream_m[i] : h->seg_stream[i]));
        }
        h->last_masked = masked ? 1 : 0;
        hipStream_t fs = masked ? h->front_stream_m : h->front_stream;
        auto stage1 = [&](const mi::TpArgs& c) -> hipError_t {  // the windows whose magnitudes are the chunk's squelch samples
            mi::ChannelizeArgs cc = ca;
            const uint32_t f0 = first_call ? (c.first_chunk ? 0u : c.step0 + mi::kAgcExtra) : c.step0;
            const uint32_t f1 = first_call ? c.step1 + mi::kAgcExtra : c.step1;
            cc.iq = ca.iq + static_cast<size_t>(f0) * ca.hop_bytes;
            cc.valid_bytes = ca.valid_bytes - static_cast<size_t>(f0) * ca.hop_bytes;
            cc.nfft = f1 - f0;
            cc.plane_off = ca.plane_off + f0;
            cc.mag = planes;
            cc.xmax = h->d_xmax[q];
            return stage1_launch(cc, f0, fs);
        };
        auto ev = [&](int i, int k) { return cev[static_cast<size_t>(i) * mi_demod::kEvPerChunk + k]; };
        HIP_TRY(hipEventRecord(h->ev_entry, s));
        HIP_TRY(hipEventRecord(evc[0], s));
        HIP_TRY(hipEventRecord(evc[1], s));
        HIP_TRY(mi::launch_tp_audio_head(ta, s));
        HIP_TRY(hipEventRecord(h->ev_head, s));  // the previous call is complete and its lookahead has been taken over
        if (iq_ready)
            HIP_TRY(hipStreamWaitEvent(fs, iq_ready, 0));
        if (!overlap)
            HIP_TRY(hipStreamWaitEvent(fs, h->ev_entry, 0));  // stage 1 honours the caller's stream order
        else if (h->set_seq[q])
            HIP_TRY(hipStreamWaitEvent(fs, h->ev[q][2], 0));  // the call that used this scratch set last (kSets back) has left it
        {
            // ... and the call after that one has read what its speculative head needed from that set (planes, aggregates, core
            // states): its segment pass is done (always long before; the wait costs nothing)
            const int qn = (q + 1) % mi_demod::kSets;
            if (h->set_seq[qn] && h->set_path[qn] == 1 && h->tp_chunks[qn] > 0)
                HIP_TRY(hipStreamWaitEvent(fs, h->chunk_ev[qn][static_cast<size_t>(h->tp_chunks[qn] - 1) * mi_demod::kEvPerChunk + 6], 0));
        }
        if (h->tp_mixed && h->cplx_busy[zp])  // (the serial kernel of the call before the previous one read this complex plane set)
            HIP_TRY(hipStreamWaitEvent(fs, h->ev_cplx_free[zp], 0));
        // the carried samples of the previous call (wherever they are) become the head of this call's planes -- of a mixed plan the
        // time-parallel rows' only where the serial kernel of the previous call has put its own rows' there itself
        if (h->tp_mixed && h->ser_head_next && planes == h->d_mag_set[(h->cur + 1) % mi_demod::kSets])
            HIP_TRY(mi::launch_move_head(planes, h->d_mag + h->head_off, h->plane_stride, h->tp_rows, fs, h->d_rows));
        else
            HIP_TRY(mi::launch_move_head(planes, h->d_mag + h->head_off, h->plane_stride, h->rows, fs));
        HIP_TRY(hipMemsetAsync(h->d_xmax[q], 0, static_cast<size_t>(h->rows) * sizeof(unsigned), fs));
        // Stage 1 + aggregates of every chunk first: nothing else feeds them (when calls overlap, k_tp_full warms its first
        // lanes up on the previous call's planes, so not even the chain state of that call).
        for (int i = 0; i < C; ++i) {
            const mi::TpArgs c = chunk(i);
            HIP_TRY(hipEventRecord(ev(i, 0), fs));
            HIP_TRY(stage1(c));
            HIP_TRY(hipEventRecord(ev(i, 1), fs));
            HIP_TRY(hipEventRecord(ev(i, 11), fs));
            HIP_TRY(mi::launch_tp_front(c, fs, /*seed_chain=*/!overlap));
            HIP_TRY(hipEventRecord(ev(i, 2), fs));
        }
        for (int i = 0; i < C; ++i) {
            const mi::TpArgs c = chunk(i);
            HIP_TRY(hipStreamWaitEvent(h->aux_stream, ev(i, 2), 0));
            HIP_TRY(hipEventRecord(ev(i, 3), h->aux_stream));
            HIP_TRY(mi::launch_tp_core(c, h->aux_stream));
            HIP_TRY(hipEventRecord(ev(i, 4), h->aux_stream));
            hipStream_t ss = masked ? h->seg_stream_m[i % mi_demod::kSegStreams] : h->seg_stream[i % mi_demod::kSegStreams];
            // A segment pass needs core(i).  It also has to wait for the previous call (ev_head) where it touches what
            // that call's tail still owns: the carried ChanState (the lanes of the first TP_W / L + 1 segments start
            // from it), the audio lookahead (written by the last segments) and the caller's audio buffer if it is the
            // one the previous call wrote.
            HIP_TRY(hipStreamWaitEvent(ss, ev(i, 4), 0));
            if (seg_early) {
                // The pass writes the caller's audio buffer: the last call that wrote the same memory has to be complete (its fades
                // rewrite audio).  Not the previous call (seg_early); with two buffers alternating the one before it, with three the
                // one before that -- then the pass has two tail periods of slack instead of one and the tails run back to back.
                for (int back = 1; back < mi_demod::kSets - 1; ++back) {
                    const int pq = (h->cur + mi_demod::kSets - back) % mi_demod::kSets;
                    if (!h->set_seq[pq])
                        break;
                    if (h->set_path[pq] != 1 || !(out_hi <= h->set_out_lo[pq] || out_lo >= h->set_out_hi[pq])) {
                        HIP_TRY(hipStreamWaitEvent(ss, h->ev[pq][2], 0));
                        break;
                    }
                }
            }
            // (events 5 -> 12 time the pass itself: they sit inside every wait of the segment stream; of a split first chunk
            // the body is timed, its few head segments are not)
            const uint32_t head_end = std::min<uint32_t>(c.seg1, mi::TP_W / L + 1);
            if (spec_head) {
                HIP_TRY(hipEventRecord(ev(i, 5), ss));
                HIP_TRY(mi::launch_tp_seg(c, ss));
                HIP_TRY(hipEventRecord(ev(i, 12), ss));
            } else if (!seg_early || c.last_chunk) {
                HIP_TRY(hipStreamWaitEvent(ss, h->ev_head, 0));
                HIP_TRY(hipEventRecord(ev(i, 5), ss));
                HIP_TRY(mi::launch_tp_seg(c, ss));
                HIP_TRY(hipEventRecord(ev(i, 12), ss));
            } else if (c.first_chunk && c.seg0 < head_end) {
                mi::TpArgs body = c, head = c;
                body.seg0 = head_end;
                head.seg1 = head_end;
                HIP_TRY(hipEventRecord(ev(i, 5), ss));
                if (body.seg0 < body.seg1)
                    HIP_TRY(mi::launch_tp_seg(body, ss));
                HIP_TRY(hipEventRecord(ev(i, 12), ss));
                HIP_TRY(hipStreamWaitEvent(ss, h->ev_head, 0));
                HIP_TRY(mi::launch_tp_seg(head, ss));
            } else {
                HIP_TRY(hipEventRecord(ev(i, 5), ss));
                HIP_TRY(mi::launch_tp_seg(c, ss));
                HIP_TRY(hipEventRecord(ev(i, 12), ss));
            }
            HIP_TRY(hipEventRecord(ev(i, 6), ss));
            HIP_TRY(hipStreamWaitEvent(s, ev(i, 6), 0));
            HIP_TRY(hipEventRecord(ev(i, 10), s));
            hipEvent_t marks[mi::TP_REST_MARKS] = {ev(i, 7), ev(i, 8), ev(i, 9)};
            HIP_TRY(mi::launch_tp_rest(c, s, marks));
        }
        h->set_mixed[q] = false;
        h->ser_head_next = false;
        if (h->tp_mixed && h->ser_rows > 0) {
            // ---- the rows the time-parallel path does not take: k_demod on its own stream, beside the chain ----
            // It needs stage 1 of the whole call (the last chunk's end on the front stream), the serial kernel of the previous call
            // (same stream) and, where calls do not overlap, the caller's stream order.
            hipStream_t zs = h->ser_stream;
            HIP_TRY(hipStreamWaitEvent(zs, ev(C - 1, 1), 0));
            if (!overlap)
                HIP_TRY(hipStreamWaitEvent(zs, h->ev_entry, 0));
            if (!seg_early)
                HIP_TRY(hipStreamWaitEvent(zs, h->ev_head, 0));  // (it writes the audio buffer the previous call wrote)
            mi::DemodArgs dm = da;
            dm.rows = h->d_srows;
            dm.nrows = h->ser_rows;
            dm.mag = planes;
            dm.cplx = h->d_cplx_set[zp];
            dm.mag_head = h->d_mag_set[(q + 1) % mi_demod::kSets];  // (free: the call that used it last is four calls back)
            dm.cplx_head = h->d_cplx_set[znp];
            dm.carry = h->d_carry_set[q];
            dm.carry_in = h->d_carry;
            dm.lanes_per_wave = std::min(64, std::max(1, (h->ser_rows + h->opt_uni_rows - 1) / h->opt_uni_rows));
            dm.pre_wave = (h->opt_pre_wave < 0 ? h->ser_rows <= 256 : h->opt_pre_wave != 0) ? 1 : 0;
            HIP_TRY(hipEventRecord(evc[3], zs));
            HIP_TRY(mi::launch_demod(dm, zs));
            HIP_TRY(hipEventRecord(evc[4], zs));
            HIP_TRY(hipEventRecord(h->ev_cplx_free[zp], zs));
            h->cplx_busy[zp] = true;
            HIP_TRY(hipStreamWaitEvent(s, evc[4], 0));
            h->d_cplx = h->d_cplx_set[znp];
            h->d_cplx_last = h->d_cplx_set[zp];
            h->pset = znp;
            h->set_mixed[q] = true;
            h->ser_head_next = true;
        }
        h->tp_chunks[q] = C;
        h->cur = q;
        h->d_carry = h->d_carry_set[q];
        h->d_mag = planes;
        h->head_off = n;  // (first call: the planes hold AGC_EXTRA + n samples, the last AGC_EXTRA start at n as well)
        h->chain_live = true;
        h->prev_out_lo = out_lo;
        h->prev_out_hi = out_hi;
        h->set_out_lo[q] = out_lo;
        h->set_out_hi[q] = out_hi;
    } else if (h->plan.any_afc) {
        if (iq_ready)
            HIP_TRY(hipStreamWaitEvent(s, iq_ready, 0));
        // AFC (rtl_airband.cpp:180-251): the bins stage 1 picks in batch b+1 depend on the squelch outcome of batch b, so
        // the batches are enqueued one at a time -- stage 1, channel loop, AFC::finalize -- with the bin table and the
        // previous indicator resident in ChanState: no host round trip inside the call.
        HIP_TRY(hipEventRecord(evc[0], s));
        size_t f0 = 0;  // first window of the batch, relative to the call
        for (int b = 0; b < nbatches; ++b) {
            const bool first = h->first_call && b == 0;
            const uint32_t nf = mi::kWaveBatch + (first ? mi::kAgcExtra : 0);
            mi::ChannelizeArgs cb = ca;
            cb.iq = ca.iq + f0 * ca.hop_bytes;
            cb.valid_bytes = ca.valid_bytes - f0 * ca.hop_bytes;
            cb.nfft = nf;
            cb.plane_off = first ? 0 : mi::kAgcExtra;
            cb.st = h->d_state;
            cb.afc_spec = h->d_afc_spec;
            HIP_TRY(mi::launch_channelize(cb, h->plan.log2n, h->plan.dev.sfmt, h->nstreams, s));
            if (b == 0)
                HIP_TRY(hipEventRecord(evc[1], s));
            mi::DemodArgs db = da;
            db.nsteps = mi::kWaveBatch;
            db.nbatches = 1;
            db.wmain = d_wmain + static_cast<size_t>(b) * mi::kWaveBatch;
            db.iq_out = d_iq_out ? d_iq_out + static_cast<size_t>(b) * mi::kWaveBatch : nullptr;
            db.axc = d_axc + b;
            HIP_TRY(mi::launch_demod(db, s));
            mi::AfcArgs aa{};
            aa.nstreams = h->nstreams;
            aa.nch = h->nch;
            aa.fft_size = h->plan.fft_size;
            aa.cp = h->d_cp;
            aa.st = h->d_state;
            aa.spec = h->d_afc_spec;
            aa.axc = d_axc + b;
            aa.axc_stride = static_cast<uint32_t>(nbatches);
            HIP_TRY(mi::launch_afc(aa, s));
            f0 += nf;
        }
    } else if (early_input && !h->first_call && (!h->tp_eligible || (h->head_off == 0 && (h->d_mag == h->d_mag_set[0] || h->d_mag == h->d_mag_set[1]))) &&
               serial_sets_ready(h)) {
        // (a handle whose plan the time-parallel path could take as well -- many rows, or MI_OPT_TIME_PARALLEL = 0 -- pipelines its serial
        //  calls like any other as long as its planes are where this branch keeps them: never after a time-parallel call)
        if (h->tp_eligible)
            h->pset = h->d_mag == h->d_mag_set[1] ? 1 : 0;
        // ---- serial stage 2 with consecutive calls overlapping (MI_OPT_EARLY_INPUT) ----
        // Two plane sets alternate.  Stage 1 of this call fills the body of set p on the front stream while the previous
        // call's k_demod, which reads the other set, still runs on the caller's stream (all that k_demod writes into set p
        // is the carried head, entries [0, AGC_EXTRA), which stage 1 does not touch).  k_demod of this call waits for its
        // stage 1 and leaves the head in the other set for the next call.
        const int q = (h->cur + 1) % mi_demod::kSets;  // event / timing set of this call
        const int before_prev = (h->cur + mi_demod::kSets - 1) % mi_demod::kSets;
        const int p = h->pset, np = p ^ 1;
        evc = h->ev[q];
        if (h->split_state == 0) {
            h->split_state = 2;
            hipDeviceProp_t prop{};
            // Two waves per row (k_demod_pw2) of 449 .. 512 rows fill 128 CUs two to a SIMD -- the kernel's pace alone -- and stage 1 of
            // that many streams takes as long on the other 128 as the kernel does: side by side on disjoint CUs 3.47 ms per 8-s call of
            // 64 x 8 AM channels, sharing every SIMD 3.8 (stage 1 3.4-3.5 ms beside the kernel's waves against 1.9 alone).  With fewer
            // rows the call is the kernel's latency either way and the split only takes CUs from stage 1 (tools/split_rows.sh); with
            // more the kernel needs more than 128 CUs (112 for 512 rows: 3.7 ms).
            const int want = h->opt_split_cus >= 0 ? h->opt_split_cus : ((da.pre_wave == 2 && h->rows > 448 && h->rows <= 512) ? 128 : 0);
            if (want > 0 && hipGetDeviceProperties(&prop, h->gpu) == hipSuccess && prop.multiProcessorCount >= want + 32) {
                const int ncu = prop.multiProcessorCount, keep = ncu - want;
                std::vector<uint32_t> m_front(static_cast<size_t>((ncu + 31) / 32), 0u), m_demod(m_front.size(), 0u);
                for (int i = 0; i < ncu; ++i)
                    (i < keep ? m_front : m_demod)[static_cast<size_t>(i) / 32] |= 1u << (i % 32);
                hipError_t me = hipExtStreamCreateWithCUMask(&h->ps_front_m, static_cast<uint32_t>(m_front.size()), m_front.data());
                if (me == hipSuccess)
                    me = hipExtStreamCreateWithCUMask(&h->ps_demod_m, static_cast<uint32_t>(m_demod.size()), m_demod.data());
                if (me == hipSuccess)
                    me = hipEventCreateWithFlags(&h->ev_ps_entry, hipEventDisableTiming);
                if (me == hipSuccess)
                    me = hipEventCreateWithFlags(&h->ev_ps_done, hipEventDisableTiming);
                if (me == hipSuccess)
                    h->split_state = 1;
                else
                    (void)hipGetLastError();
            }
        }
        const bool split = h->split_state == 1 && s != nullptr;
        hipStream_t fs = split ? h->ps_front_m : h->front_stream;
        if (iq_ready)
            HIP_TRY(hipStreamWaitEvent(fs, iq_ready, 0));
        if (h->serial_pipe && h->set_seq[before_prev])  // the call before the previous one read the body of set p
            HIP_TRY(hipStreamWaitEvent(fs, h->ev[before_prev][2], 0));
        else
            HIP_TRY(hipStreamWaitEvent(fs, h->ev[h->cur][2], 0));  // (first pipelined call: everything before it)
        ca.mag = h->d_mag_set[p];
        ca.cplx = h->d_cplx_set[p];
        HIP_TRY(hipEventRecord(evc[0], fs));
        HIP_TRY(stage1_launch(ca, 0, fs));
        HIP_TRY(hipEventRecord(evc[1], fs));
        da.mag = h->d_mag_set[p];
        da.cplx = h->d_cplx_set[p];
        da.mag_head = h->d_mag_set[np];
        da.cplx_head = h->d_cplx_set[np];
        if (split) {  // k_demod on the CUs stage 1 keeps off, in the caller's stream order all the same
            hipStream_t ds = h->ps_demod_m;
            HIP_TRY(hipEventRecord(h->ev_ps_entry, s));
            HIP_TRY(hipStreamWaitEvent(ds, h->ev_ps_entry, 0));
            HIP_TRY(hipStreamWaitEvent(ds, evc[1], 0));
            HIP_TRY(hipEventRecord(evc[3], ds));
            HIP_TRY(mi::launch_demod(da, ds));
            HIP_TRY(hipEventRecord(h->ev_ps_done, ds));
            HIP_TRY(hipStreamWaitEvent(s, h->ev_ps_done, 0));
        } else {
            HIP_TRY(hipStreamWaitEvent(s, evc[1], 0));
            HIP_TRY(hipEventRecord(evc[3], s));
            HIP_TRY(mi::launch_demod(da, s));
        }
        h->chain_live = false;
        h->cur = q;
        h->pset = np;
        h->d_mag = h->d_mag_set[np];
        h->d_cplx = h->d_cplx_set[np];
        h->d_mag_last = h->d_mag_set[p];
        h->d_cplx_last = h->d_cplx_set[p];
        h->serial_pipe = true;
        pipelined_serial = true;
    } else {
        if (iq_ready)
            HIP_TRY(hipStreamWaitEvent(s, iq_ready, 0));
        int rc = head_in_place();
        if (rc != MI_OK)
            return rc;
        h->chain_live = false;  // k_demod does not maintain the time-parallel chain state
        h->serial_pipe = false;
        HIP_TRY(hipEventRecord(evc[0], s));
        HIP_TRY(stage1_launch(ca, 0, s));
        HIP_TRY(hipEventRecord(evc[1], s));
        HIP_TRY(mi::launch_demod(da, s));
    }
    h->last_path = use_tp ? 1 : 0;
    if (!use_tp)
        h->ser_head_next = false;
    HIP_TRY(hipEventRecord(evc[2], s));
    h->set_seq[h->cur] = ++h->call_seq;
    h->set_path[h->cur] = pipelined_serial ? 2 : h->last_path;
    h->first_call = false;
    return MI_OK;
}

}  // namespace

extern "C" {

const char* mi_last_error(void) {
    return g_err.c_str();
}

int mi_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

void mi_demod_destroy(mi_demod* h) {
    if (!h)
        return;
    (void)hipSetDevice(h->gpu);
    (void)hipDeviceSynchronize();  // calls may still be in flight on the handle's own streams
    void* ptrs[] = {h->d_window, h->d_tw, h->d_prune_t1, h->d_prune_t2, h->d_prune_rank, h->d_l64_chan, h->d_l64_chan_full, h->d_l64_tickets, h->d_levels,      h->d_sin,     h->d_cos,   h->d_cp, h->d_state, h->d_cplx_set[0], h->d_cplx_set[1], h->d_carry_set[0], h->d_carry_set[1], h->d_carry_set[2], h->d_carry_set[3],
                    h->d_ring,   h->d_ctcss_coeff, h->d_ctcss_q, h->d_stats, h->d_pre_timeouts,
                    h->d_rows,   h->d_srows, h->d_tstart, h->d_need, h->d_redo, h->d_fin, h->d_diag, h->d_core_carry, h->d_full0, h->d_fullbound, h->d_afc_spec};
    for (void* p : ptrs)
        if (p)
            (void)hipFree(p);
    for (int q = 0; q < mi_demod::kSets; ++q) {
        void* sets[] = {h->d_mag_set[q], h->d_xmax[q], h->d_blk_fe[q], h->d_blk_fm[q], h->d_blk_x0[q], h->d_blk_xm[q], h->d_core[q], h->d_rec[q]};
        for (void* p : sets)
            if (p)
                (void)hipFree(p);
    }
    for (mi_demod::Slot& sl : h->slot) {
        void* dev[] = {sl.d_iq, sl.d_wout, sl.d_iqout, sl.d_axc, sl.d_stats};
        for (void* p : dev)
            if (p)
                (void)hipFree(p);
        if (sl.h_in)
            (void)hipHostFree(sl.h_in);
        if (sl.h_out)
            (void)hipHostFree(sl.h_out);
        if (sl.up_done)
            (void)hipEventDestroy(sl.up_done);
        if (sl.done)
            (void)hipEventDestroy(sl.done);
    }
    if (h->copy_stream)
        (void)hipStreamDestroy(h->copy_stream);
    if (h->down_stream)
        (void)hipStreamDestroy(h->down_stream);
    for (int q = 0; q < mi_demod::kSets; ++q) {
        for (hipEvent_t e : h->ev[q])
            if (e)
                (void)hipEventDestroy(e);
        for (hipEvent_t e : h->chunk_ev[q])
            if (e)
                (void)hipEventDestroy(e);
    }
    if (h->ev_entry)
        (void)hipEventDestroy(h->ev_entry);
    if (h->ev_head)
        (void)hipEventDestroy(h->ev_head);
    if (h->aux_stream)
        (void)hipStreamDestroy(h->aux_stream);
    if (h->ser_stream)
        (void)hipStreamDestroy(h->ser_stream);
    for (hipEvent_t e : h->ev_cplx_free)
        if (e)
            (void)hipEventDestroy(e);
    if (h->front_stream_m)
        (void)hipStreamDestroy(h->front_stream_m);
    if (h->ps_front_m)
        (void)hipStreamDestroy(h->ps_front_m);
    if (h->ps_demod_m)
        (void)hipStreamDestroy(h->ps_demod_m);
    if (h->ev_ps_entry)
        (void)hipEventDestroy(h->ev_ps_entry);
    if (h->ev_ps_done)
        (void)hipEventDestroy(h->ev_ps_done);
    for (hipStream_t ss : h->seg_stream_m)
        if (ss)
            (void)hipStreamDestroy(ss);
    if (h->front_stream)
        (void)hipStreamDestroy(h->front_stream);
    for (hipStream_t ss : h->seg_stream)
        if (ss)
            (void)hipStreamDestroy(ss);
    if (h->own_stream)
        (void)hipStreamDestroy(h->own_stream);
    delete h;
}

int mi_demod_create(const mi_device_cfg* dev, const mi_channel_cfg* chans, int nch, int nstreams, int max_batches, int gpu, mi_demod** out) {
    if (!out)
        return fail(MI_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!dev || !chans)
        return fail(MI_ERR_INVALID, "dev/chans is NULL");
    if (nstreams < 1 || max_batches < 1)
        return fail(MI_ERR_INVALID, "nstreams and max_batches must be >= 1");
    mi_demod* h = new (std::nothrow) mi_demod();
    if (!h)
        return fail(MI_ERR_NOMEM, "host allocation failed");
    tuning_from_env(h);
    const char* msg = "";
    int rc = mi::build_plan(*dev, chans, nch, h->plan, &msg);
    if (rc != MI_OK) {
        delete h;
        return fail(rc, msg);
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev < 1) {
        delete h;
        return fail(MI_ERR_NO_DEVICE, "no HIP device: the MI355X path has no CPU fallback");
    }
    if (gpu < 0 || gpu >= ndev) {
        delete h;
        return fail(MI_ERR_INVALID, "gpu index out of range");
    }
    h->gpu = gpu;
    h->nstreams = nstreams;
    h->nch = nch;
    h->rows = nstreams * nch;
    h->max_batches = max_batches;
    const mi::Plan& p = h->plan;
    const size_t max_steps = static_cast<size_t>(max_batches) * mi::kWaveBatch;
    h->plane_stride = (max_steps + 2 * mi::kAgcExtra + 3) & ~static_cast<size_t>(3);

    auto bail = [&](int code) {
        std::string keep = g_err;
        mi_demod_destroy(h);
        g_err = keep;
        return code;
    };
#define TRY_OR_BAIL(expr)                          \
    do {                                           \
        hipError_t e__ = (expr);                   \
        if (e__ != hipSuccess)                     \
            return bail(hip_fail(e__, #expr));     \
    } while (0)

    TRY_OR_BAIL(hipSetDevice(gpu));
    TRY_OR_BAIL(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    for (auto& evs : h->ev)
        for (hipEvent_t& ev : evs)
            TRY_OR_BAIL(hipEventCreate(&ev));
    TRY_OR_BAIL(hipEventCreate(&h->ev_entry));
    TRY_OR_BAIL(hipEventCreate(&h->ev_head));
    {
        // HIP multiplexes its streams onto a few hardware queues; two streams that share one run their kernels one after
        // the other.  The core chain must never queue behind a wide pass, so it gets the highest stream priority (its
        // own queue class), and the wide passes share as few other streams as the pipeline needs.
        int lo_prio = 0, hi_prio = 0;
        TRY_OR_BAIL(hipDeviceGetStreamPriorityRange(&lo_prio, &hi_prio));
        TRY_OR_BAIL(hipStreamCreateWithPriority(&h->aux_stream, hipStreamNonBlocking, hi_prio));
    }
    TRY_OR_BAIL(hipStreamCreateWithFlags(&h->front_stream, hipStreamNonBlocking));
    for (hipStream_t& ss : h->seg_stream)
        TRY_OR_BAIL(hipStreamCreateWithFlags(&ss, hipStreamNonBlocking));
    const size_t rows = static_cast<size_t>(h->rows);
    TRY_OR_BAIL(dalloc(&h->d_window, p.window.size()));
    TRY_OR_BAIL(dalloc(&h->d_tw, p.tw.size()));
    TRY_OR_BAIL(dalloc(&h->d_levels, 256));
    TRY_OR_BAIL(dalloc(&h->d_sin, 257));
    TRY_OR_BAIL(dalloc(&h->d_cos, 257));
    TRY_OR_BAIL(dalloc(&h->d_cp, static_cast<size_t>(nch)));
    TRY_OR_BAIL(dalloc(&h->d_state, rows));
    TRY_OR_BAIL(dalloc(&h->d_mag_set[0], rows * h->plane_stride));
    h->d_mag = h->d_mag_set[0];
    TRY_OR_BAIL(dalloc(&h->d_cplx, static_cast<size_t>(nstreams) * p.n_iq_rows * h->plane_stride));
    h->d_cplx_set[0] = h->d_cplx;
    TRY_OR_BAIL(dalloc(&h->d_carry_set[0], rows * mi::kAgcExtra));
    h->d_carry = h->d_carry_set[0];
    TRY_OR_BAIL(dalloc(&h->d_ring, rows * mi::kSquelchRing));
    TRY_OR_BAIL(dalloc(&h->d_ctcss_coeff, p.ctcss_coeff.size()));
    TRY_OR_BAIL(dalloc(&h->d_ctcss_q, static_cast<size_t>(nstreams) * p.n_ctcss_rows * 4 * mi::kMaxTones));
    TRY_OR_BAIL(dalloc(&h->d_stats, rows));
    TRY_OR_BAIL(dalloc(&h->d_pre_timeouts, 1));
    TRY_OR_BAIL(hipMemset(h->d_pre_timeouts, 0, sizeof(unsigned)));
    TRY_OR_BAIL(hipMemcpy(h->d_window, p.window.data(), p.window.size() * 4, hipMemcpyHostToDevice));
    TRY_OR_BAIL(hipMemcpy(h->d_tw, p.tw.data(), p.tw.size() * 4, hipMemcpyHostToDevice));
    if (p.prune.enabled) {
        TRY_OR_BAIL(dalloc(&h->d_prune_t1, p.prune_t1.size()));
        TRY_OR_BAIL(hipMemcpy(h->d_prune_t1, p.prune_t1.data(), p.prune_t1.size() * 4, hipMemcpyHostToDevice));
        TRY_OR_BAIL(dalloc(&h->d_prune_t2, p.prune_t2.size()));
        TRY_OR_BAIL(hipMemcpy(h->d_prune_t2, p.prune_t2.data(), p.prune_t2.size() * 4, hipMemcpyHostToDevice));
        TRY_OR_BAIL(dalloc(&h->d_prune_rank, p.prune_chan_rank.size()));
        TRY_OR_BAIL(hipMemcpy(h->d_prune_rank, p.prune_chan_rank.data(), p.prune_chan_rank.size() * 4, hipMemcpyHostToDevice));
    }
    if (p.l64.enabled) {
        TRY_OR_BAIL(dalloc(&h->d_l64_chan, p.l64_chan.size()));
        TRY_OR_BAIL(hipMemcpy(h->d_l64_chan, p.l64_chan.data(), p.l64_chan.size() * sizeof(L64Chan), hipMemcpyHostToDevice));
        TRY_OR_BAIL(dalloc(&h->d_l64_chan_full, p.l64_chan_full.size()));
        TRY_OR_BAIL(hipMemcpy(h->d_l64_chan_full, p.l64_chan_full.data(), p.l64_chan_full.size() * sizeof(L64Chan), hipMemcpyHostToDevice));
        TRY_OR_BAIL(dalloc(&h->d_l64_tickets, mi::kL64Tickets));
        TRY_OR_BAIL(hipMemset(h->d_l64_tickets, 0, mi::kL64Tickets * sizeof(unsigned)));
    }
    TRY_OR_BAIL(hipMemcpy(h->d_levels, p.levels.data(), 256 * 4, hipMemcpyHostToDevice));
    TRY_OR_BAIL(hipMemcpy(h->d_sin, p.sin_lut, 257 * 4, hipMemcpyHostToDevice));
    TRY_OR_BAIL(hipMemcpy(h->d_cos, p.cos_lut, 257 * 4, hipMemcpyHostToDevice));
    TRY_OR_BAIL(hipMemcpy(h->d_cp, p.cp.data(), p.cp.size() * sizeof(mi::ChanParams), hipMemcpyHostToDevice));
    if (!p.ctcss_coeff.empty())
        TRY_OR_BAIL(hipMemcpy(h->d_ctcss_coeff, p.ctcss_coeff.data(), p.ctcss_coeff.size() * 4, hipMemcpyHostToDevice));
    TRY_OR_BAIL(hipMemset(h->d_mag, 0, rows * h->plane_stride * 4));
    if (h->d_cplx)
        TRY_OR_BAIL(hipMemset(h->d_cplx, 0, static_cast<size_t>(nstreams) * p.n_iq_rows * h->plane_stride * 8));
    TRY_OR_BAIL(hipMemset(h->d_stats, 0, rows * sizeof(mi_channel_stats)));
    // which channels the time-parallel path can take: plain AM (no raw I/Q, CTCSS, notch); none where AFC moves the bins batch by batch
    std::vector<char> tp_ch(static_cast<size_t>(nch), 0);
    int tp_nch = 0;
    for (int c = 0; c < nch; ++c) {
        const mi::ChanParams& cp = p.cp[static_cast<size_t>(c)];
        tp_ch[static_cast<size_t>(c)] = !p.any_afc && cp.modulation == MI_MOD_AM && !cp.needs_raw_iq && !cp.ctcss_enabled && !cp.notch_enabled && cp.afc == 0;
        tp_nch += tp_ch[static_cast<size_t>(c)] ? 1 : 0;
    }
    h->tp_eligible = tp_nch > 0;
    h->tp_mixed = tp_nch > 0 && tp_nch < nch;
    h->tp_rows = nstreams * tp_nch;
    h->ser_rows = h->rows - h->tp_rows;
    h->core_split_ok = h->tp_eligible;
    for (int c = 0; c < nch; ++c)
        if (tp_ch[static_cast<size_t>(c)] && (p.cp[static_cast<size_t>(c)].using_manual_level || !(p.cp[static_cast<size_t>(c)].cap_factor >= 1.0f)))
            h->core_split_ok = false;  // (the chain wave's operand assumes cap >= noise floor in a burst)
    if (p.any_afc)
        TRY_OR_BAIL(dalloc(&h->d_afc_spec, static_cast<size_t>(nstreams) * p.fft_size));
    if (h->tp_eligible) {
        h->tp_max_blk = max_steps / 16;
        // segment length of the time-parallel path: short segments where rows are few (the parallelism has to come from time),
        // long ones where they are many (each lane pays TP_W steps of warm-up whatever its segment's length)
        if (h->opt_tp_L == 0)
            h->tp_L = h->tp_rows <= 32 ? 512u : 1024u;  // (2048 beyond 128 rows until the segment pass ran full waves: DESIGN 6)
        else
            h->tp_L = static_cast<uint32_t>(h->opt_tp_L);
        h->tp_max_seg = (max_steps + h->tp_L - 1) / h->tp_L;
        std::vector<int> tp_list, ser_list;  // handle rows (stream * nch + channel) of either kind, in row order
        for (size_t i = 0; i < rows; ++i)
            (tp_ch[i % static_cast<size_t>(nch)] ? tp_list : ser_list).push_back(static_cast<int>(i));
        TRY_OR_BAIL(dalloc(&h->d_rows, rows));
        TRY_OR_BAIL(hipMemcpy(h->d_rows, tp_list.data(), tp_list.size() * sizeof(int), hipMemcpyHostToDevice));
        if (!ser_list.empty()) {
            TRY_OR_BAIL(dalloc(&h->d_srows, ser_list.size()));
            TRY_OR_BAIL(hipMemcpy(h->d_srows, ser_list.data(), ser_list.size() * sizeof(int), hipMemcpyHostToDevice));
            {
                // HIP multiplexes its streams onto a few hardware queues and two streams that share one run their kernels one after the
                // other: on a stream of the default class the serial kernel (a millisecond and more) took turns with stage 1 of the next
                // call.  The lowest stream priority is a queue class of its own.
                int lo_prio = 0, hi_prio = 0;
                TRY_OR_BAIL(hipDeviceGetStreamPriorityRange(&lo_prio, &hi_prio));
                TRY_OR_BAIL(hipStreamCreateWithPriority(&h->ser_stream, hipStreamNonBlocking, lo_prio));
            }
            for (hipEvent_t& e : h->ev_cplx_free)
                TRY_OR_BAIL(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        for (int q = 1; q < mi_demod::kSets; ++q) {
            TRY_OR_BAIL(dalloc(&h->d_mag_set[q], rows * h->plane_stride));
            TRY_OR_BAIL(hipMemset(h->d_mag_set[q], 0, rows * h->plane_stride * 4));
            TRY_OR_BAIL(dalloc(&h->d_carry_set[q], rows * mi::kAgcExtra));
        }
        for (int q = 0; q < mi_demod::kSets; ++q) {
            TRY_OR_BAIL(dalloc(&h->d_xmax[q], rows));
            TRY_OR_BAIL(dalloc(&h->d_blk_fe[q], rows * h->tp_max_blk));
            TRY_OR_BAIL(dalloc(&h->d_blk_fm[q], rows * h->tp_max_blk));
            TRY_OR_BAIL(dalloc(&h->d_blk_x0[q], rows * h->tp_max_blk));
            TRY_OR_BAIL(dalloc(&h->d_blk_xm[q], rows * h->tp_max_blk));
            TRY_OR_BAIL(dalloc(&h->d_core[q], rows * (h->tp_max_seg + 1)));
        }
        for (int q = 0; q < mi_demod::kSets; ++q)
            TRY_OR_BAIL(dalloc(&h->d_rec[q], static_cast<size_t>(mi::TP_NREC) * rows * h->tp_max_seg));
        TRY_OR_BAIL(dalloc(&h->d_tstart, rows * h->tp_max_seg * 8));
        TRY_OR_BAIL(dalloc(&h->d_need, rows * h->tp_max_seg));
        TRY_OR_BAIL(dalloc(&h->d_redo, rows * h->tp_max_seg + 1));
        TRY_OR_BAIL(dalloc(&h->d_fin, rows));
        TRY_OR_BAIL(hipMemset(h->d_fin, 0, rows * sizeof(mi::TpFinal)));
        TRY_OR_BAIL(dalloc(&h->d_core_carry, rows));
        TRY_OR_BAIL(dalloc(&h->d_full0, rows));
        TRY_OR_BAIL(dalloc(&h->d_fullbound, rows));
        TRY_OR_BAIL(hipMemset(h->d_fullbound, 0, rows * sizeof(float)));
        TRY_OR_BAIL(dalloc(&h->d_diag, rows * 8));
        TRY_OR_BAIL(hipMemset(h->d_diag, 0, rows * 8 * sizeof(int)));
    }
    TRY_OR_BAIL(mi::launch_init_state(h->d_state, h->d_carry, h->d_ring, h->d_ctcss_q, h->d_cp, nstreams, nch, p.n_ctcss_rows, h->own_stream));
    TRY_OR_BAIL(hipStreamSynchronize(h->own_stream));
#undef TRY_OR_BAIL
    if (!p.any_afc)
        stage1_compile(h);  // (a failure leaves the exchange kernels: never an error)
    *out = h;
    return MI_OK;
}

int mi_demod_prepare(mi_demod* h, int host_slots) {
    if (!h)
        return fail(MI_ERR_INVALID, "NULL handle");
    if (host_slots < 0 || host_slots > mi_demod::kSlots)
        return fail(MI_ERR_INVALID, "host_slots out of range (0 .. 3)");
    HIP_TRY(hipSetDevice(h->gpu));
    stage1_compile(h);
    for (int k = 0; k < host_slots; ++k) {
        const int rc = slot_prepare(h, k);
        if (rc != MI_OK)
            return rc;
    }
    // The runtime gives a stream its hardware queue (and a copy engine its first transfer) when the stream is first used: a few
    // milliseconds each.  One small operation on every stream of the handle now, so that the first batch does not pay for them.
    unsigned* d_warm = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_warm), 256));
    std::vector<hipStream_t> streams = {h->own_stream, h->aux_stream, h->front_stream, h->copy_stream, h->down_stream};
    for (hipStream_t ss : h->seg_stream)
        streams.push_back(ss);
    hipError_t e = hipSuccess;
    for (hipStream_t st : streams)
        if (st && e == hipSuccess)
            e = hipMemsetAsync(d_warm, 0, 256, st);
    if (e == hipSuccess && host_slots > 0 && h->copy_stream && h->down_stream) {
        e = hipMemcpyAsync(d_warm, h->slot[0].h_in, 64, hipMemcpyHostToDevice, h->copy_stream);
        if (e == hipSuccess)
            e = hipStreamSynchronize(h->copy_stream);
        if (e == hipSuccess)
            e = hipMemcpyAsync(h->slot[0].h_out, d_warm, 64, hipMemcpyDeviceToHost, h->down_stream);
    }
    for (hipStream_t st : streams)
        if (st && e == hipSuccess)
            e = hipStreamSynchronize(st);
    (void)hipFree(d_warm);
    HIP_TRY(e);
    // ... and one rehearsal of the call itself (the first dispatch of a kernel on a queue sets up its scratch and kernel-argument
    // memory: 7 ms on the first batch otherwise).  The handle's state is saved before and restored after: a prepared handle is
    // bit for bit the handle mi_demod_create returned.
    std::vector<unsigned char> saved(mi_demod_state_size(h));
    int rc = mi_demod_get_state(h, saved.data(), saved.size());
    if (rc != MI_OK)
        return rc;
    std::vector<int> rehearsals = {1};
    if (h->tp_eligible && h->opt_tp != 0 && h->max_batches >= kTpMinBatches)
        rehearsals.push_back(kTpMinBatches);
    for (const int nb : rehearsals) {
        const size_t need = mi_demod_bytes_needed(h, nb);
        const size_t stride = (need + 255) & ~static_cast<size_t>(255);
        const size_t nsteps = static_cast<size_t>(nb) * mi::kWaveBatch, rows = static_cast<size_t>(h->rows);
        unsigned char* d_iq = nullptr;
        float *d_wo = nullptr, *d_iqo = nullptr;
        char* d_axc = nullptr;
        e = hipMalloc(reinterpret_cast<void**>(&d_iq), stride * h->nstreams);
        if (e == hipSuccess)
            e = hipMemset(d_iq, 0x80, stride * h->nstreams);
        if (e == hipSuccess)
            e = hipMalloc(reinterpret_cast<void**>(&d_wo), rows * nsteps * 4);
        if (e == hipSuccess)
            e = hipMalloc(reinterpret_cast<void**>(&d_iqo), rows * nsteps * 8);
        if (e == hipSuccess)
            e = hipMalloc(reinterpret_cast<void**>(&d_axc), rows * static_cast<size_t>(nb));
        if (e == hipSuccess) {
            rc = mi_demod_process_device(h, d_iq, stride, nb, d_wo, d_iqo, d_axc, h->own_stream);
            if (rc == MI_OK)
                e = hipDeviceSynchronize();
        }
        void* tmp[] = {d_iq, d_wo, d_iqo, d_axc};
        for (void* q : tmp)
            if (q)
                (void)hipFree(q);
        if (rc == MI_OK && e == hipSuccess)
            rc = mi_demod_set_state(h, saved.data(), saved.size());
        else if (rc == MI_OK)
            rc = fail(MI_ERR_HIP, hipGetErrorString(e));
        if (rc != MI_OK)
            return rc;
    }
    return MI_OK;
}

int mi_set_cache_dir(const char* dir) {
    mi::l64_jit_set_cache_dir(dir);
    return MI_OK;
}

int mi_jit_counts(int* compiled, int* from_cache) {
    mi::l64_jit_counts(compiled, from_cache);
    return MI_OK;
}

size_t mi_demod_hop_bytes(const mi_demod* h) {
    return h ? h->plan.hop_bytes : 0;
}

size_t mi_demod_bytes_needed(const mi_demod* h, int nbatches) {
    if (!h || nbatches < 1)
        return 0;
    const size_t nfft = static_cast<size_t>(n_fft_for(h, nbatches));
    return (nfft - 1) * h->plan.hop_bytes + 2 * static_cast<size_t>(h->plan.bytes_per_sample) * h->plan.fft_size;
}

size_t mi_demod_bytes_consumed(const mi_demod* h, int nbatches) {
    if (!h || nbatches < 1)
        return 0;
    return static_cast<size_t>(n_fft_for(h, nbatches)) * h->plan.hop_bytes;
}

int mi_demod_process_device(mi_demod* h, const void* d_iq, size_t stream_stride_bytes, int nbatches, float* d_waveout, float* d_iq_out,
                            char* d_axc, void* hip_stream) {
    if (!h || !d_iq || !d_waveout || !d_axc)
        return fail(MI_ERR_INVALID, "NULL argument");
    if (h->failed)
        return fail(MI_ERR_HIP, kFailedMsg);
    if (nbatches < 1 || nbatches > h->max_batches)
        return fail(MI_ERR_INVALID, "nbatches out of range for this handle");
    const size_t align = 2 * static_cast<size_t>(h->plan.bytes_per_sample);
    if (reinterpret_cast<uintptr_t>(d_iq) % align != 0 || stream_stride_bytes % align != 0)
        return fail(MI_ERR_INVALID, "IQ pointer and stride must be aligned to one complex sample");
    if (reinterpret_cast<uintptr_t>(d_waveout) % 16 != 0 || (d_iq_out && reinterpret_cast<uintptr_t>(d_iq_out) % 16 != 0))
        return fail(MI_ERR_INVALID, "d_waveout / d_iq_out must be 16-byte aligned (the kernels store four samples at a time)");
    const size_t need = mi_demod_bytes_needed(h, nbatches);
    if (h->nstreams > 1 && stream_stride_bytes < need)
        return fail(MI_ERR_INVALID, "stream stride shorter than the bytes one call reads");
    HIP_TRY(hipSetDevice(h->gpu));
    const size_t nsteps = static_cast<size_t>(nbatches) * mi::kWaveBatch;
    return enqueue(h, static_cast<const unsigned char*>(d_iq), stream_stride_bytes, need, nbatches, d_waveout, nsteps,
                   reinterpret_cast<float2*>(d_iq_out), nsteps, d_axc, static_cast<hipStream_t>(hip_stream));
}

namespace {

// Staging of slot k, allocated on first use and only committed when every piece exists (a failure leaves the slot unallocated
// and the handle usable: the next call tries again).
int slot_prepare(mi_demod* h, int k) {
    if (h->slots_ready[k])
        return MI_OK;
    mi_demod::Slot tmp;
    const size_t rows = static_cast<size_t>(h->rows);
    const size_t max_steps = static_cast<size_t>(h->max_batches) * mi::kWaveBatch;
    const size_t max_fft = max_steps + mi::kAgcExtra;
    const size_t stride = ((max_fft - 1) * h->plan.hop_bytes + 2 * static_cast<size_t>(h->plan.bytes_per_sample) * h->plan.fft_size + 255) & ~static_cast<size_t>(255);
    const size_t out_bytes = rows * (max_steps + mi::kAgcExtra) * 4 + rows * max_steps * 8 + rows * static_cast<size_t>(h->max_batches) + rows * sizeof(mi_channel_stats) + 64;
    hipError_t e = dalloc(&tmp.d_iq, stride * h->nstreams);
    if (e == hipSuccess)
        e = dalloc(&tmp.d_wout, rows * (max_steps + mi::kAgcExtra));
    if (e == hipSuccess)
        e = dalloc(&tmp.d_iqout, rows * max_steps);
    if (e == hipSuccess)
        e = dalloc(&tmp.d_axc, rows * static_cast<size_t>(h->max_batches));
    if (e == hipSuccess)
        e = dalloc(&tmp.d_stats, rows);
    if (e == hipSuccess)
        e = hipHostMalloc(reinterpret_cast<void**>(&tmp.h_in), stride * h->nstreams, hipHostMallocDefault);
    if (e == hipSuccess)
        e = hipHostMalloc(reinterpret_cast<void**>(&tmp.h_out), out_bytes, hipHostMallocDefault);
    if (e == hipSuccess)
        e = hipEventCreateWithFlags(&tmp.up_done, hipEventDisableTiming);
    if (e == hipSuccess)
        e = hipEventCreateWithFlags(&tmp.done, hipEventDisableTiming);
    if (e == hipSuccess && !h->copy_stream)
        e = hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking);
    if (e == hipSuccess && !h->down_stream)
        e = hipStreamCreateWithFlags(&h->down_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        void* dev[] = {tmp.d_iq, tmp.d_wout, tmp.d_iqout, tmp.d_axc, tmp.d_stats};
        for (void* p : dev)
            if (p)
                (void)hipFree(p);
        if (tmp.h_in)
            (void)hipHostFree(tmp.h_in);
        if (tmp.h_out)
            (void)hipHostFree(tmp.h_out);
        if (tmp.up_done)
            (void)hipEventDestroy(tmp.up_done);
        if (tmp.done)
            (void)hipEventDestroy(tmp.done);
        return hip_fail(e, "staging for the host-buffer entry");
    }
    h->iq_stride = stride;
    h->h_out_bytes = out_bytes;
    h->slot[k] = tmp;
    h->slots_ready[k] = true;
    return MI_OK;
}

// Is `p` host memory the GPU can read directly (hipHostMalloc / hipHostRegister / mi_host_alloc)?  Then the upload needs no
// staging copy: the copy engine reads the caller's ring itself.
bool is_pinned(const void* p) {
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return at.type == hipMemoryTypeHost;
}

// offsets of the pieces of a call's results inside a slot's pinned return buffer
struct OutLayout {
    size_t wave, iq, axc, stats;
};
OutLayout out_layout(const mi_demod* h, int nbatches) {
    const size_t rows = static_cast<size_t>(h->rows), n = static_cast<size_t>(nbatches) * mi::kWaveBatch;
    OutLayout o;
    o.wave = 0;
    o.iq = o.wave + rows * (n + mi::kAgcExtra) * 4;
    o.axc = o.iq + rows * n * 8;
    o.stats = (o.axc + rows * static_cast<size_t>(nbatches) + 15) & ~static_cast<size_t>(15);
    return o;
}

// upload + both stages + download of one call, all asynchronous; `pipelined`: upload and download on their own streams so
// that they overlap the compute of the neighbouring calls
int slot_launch(mi_demod* h, int k, const uint8_t* const* iq, int nbatches, bool want_iq, bool want_stats, bool pipelined) {
    mi_demod::Slot& sl = h->slot[k];
    const size_t rows = static_cast<size_t>(h->rows);
    const size_t nsteps = static_cast<size_t>(nbatches) * mi::kWaveBatch;
    const size_t need = mi_demod_bytes_needed(h, nbatches);
    hipStream_t s = h->own_stream;
    hipStream_t up = pipelined ? h->copy_stream : s;
    for (int i = 0; i < h->nstreams; ++i) {
        if (!iq[i])
            return fail(MI_ERR_INVALID, "NULL stream pointer");
        const unsigned char* src = iq[i];
        if (!is_pinned(src)) {
            std::memcpy(sl.h_in + static_cast<size_t>(i) * h->iq_stride, src, need);
            src = sl.h_in + static_cast<size_t>(i) * h->iq_stride;
        }
        HIP_TRY(hipMemcpyAsync(sl.d_iq + static_cast<size_t>(i) * h->iq_stride, src, need, hipMemcpyHostToDevice, up));
    }
    hipEvent_t ready = nullptr;
    if (pipelined) {
        HIP_TRY(hipEventRecord(sl.up_done, up));
        ready = sl.up_done;
    }
    // mi_demod_process: the staging copy is ordered by the stream, MI_OPT_EARLY_INPUT (valid when the call is made) does not hold.
    // mi_demod_submit: the upload has its own stream and event, which is all the front of the call waits for -- so consecutive
    // submitted calls overlap on the device like device-resident calls with the option set (their audio goes to the two
    // slots' buffers in turn, which is what lets the segment passes of one run under the tail of the other).
    const bool early = h->early_input;
    h->early_input = pipelined;
    const size_t wstride = nsteps + mi::kAgcExtra;  // the host layout: emitted audio followed by the lookahead (channel_t.waveout)
    int rc = enqueue(h, sl.d_iq, h->iq_stride, need, nbatches, sl.d_wout, wstride, want_iq ? sl.d_iqout : nullptr, nsteps, sl.d_axc, s, ready);
    h->early_input = early;
    if (rc != MI_OK)
        return rc;
    // the lookahead and the statistics belong to the handle and move on with the next call: snapshot them behind this one
    HIP_TRY_F(hipMemcpy2DAsync(sl.d_wout + nsteps, wstride * sizeof(float), h->d_carry, mi::kAgcExtra * sizeof(float), mi::kAgcExtra * sizeof(float), rows,
                             hipMemcpyDeviceToDevice, s));
    if (want_stats)
        HIP_TRY_F(hipMemcpyAsync(sl.d_stats, h->d_stats, rows * sizeof(mi_channel_stats), hipMemcpyDeviceToDevice, s));
    hipStream_t down = s;
    if (pipelined) {
        HIP_TRY_F(hipEventRecord(sl.done, s));
        HIP_TRY_F(hipStreamWaitEvent(h->down_stream, sl.done, 0));
        down = h->down_stream;
    }
    const OutLayout o = out_layout(h, nbatches);
    sl.wave_direct = is_pinned(sl.waveout);
    HIP_TRY_F(hipMemcpyAsync(sl.wave_direct ? reinterpret_cast<unsigned char*>(sl.waveout) : sl.h_out + o.wave, sl.d_wout, rows * wstride * sizeof(float),
                           hipMemcpyDeviceToHost, down));
    if (want_iq) {
        for (size_t r = 0; r < rows; ++r) {
            if (!h->plan.cp[r % h->nch].has_iq_outputs)
                continue;
            HIP_TRY_F(hipMemcpyAsync(sl.h_out + o.iq + r * nsteps * 8, sl.d_iqout + r * nsteps, nsteps * sizeof(float2), hipMemcpyDeviceToHost, down));
        }
    }
    HIP_TRY_F(hipMemcpyAsync(sl.h_out + o.axc, sl.d_axc, rows * static_cast<size_t>(nbatches), hipMemcpyDeviceToHost, down));
    if (want_stats)
        HIP_TRY_F(hipMemcpyAsync(sl.h_out + o.stats, sl.d_stats, rows * sizeof(mi_channel_stats), hipMemcpyDeviceToHost, down));
    HIP_TRY_F(hipEventRecord(sl.done, down));
    return MI_OK;
}

// wait for slot k's call and hand its results to the caller's arrays
int slot_collect(mi_demod* h, int k) {
    mi_demod::Slot& sl = h->slot[k];
    HIP_TRY(hipEventSynchronize(sl.done));
    const size_t rows = static_cast<size_t>(h->rows);
    const size_t nsteps = static_cast<size_t>(sl.nbatches) * mi::kWaveBatch;
    const OutLayout o = out_layout(h, sl.nbatches);
    if (!sl.wave_direct)
        std::memcpy(sl.waveout, sl.h_out + o.wave, rows * (nsteps + mi::kAgcExtra) * sizeof(float));
    if (sl.iq_out) {
        for (size_t r = 0; r < rows; ++r)
            if (h->plan.cp[r % h->nch].has_iq_outputs)  // rows of channels without iq outputs are untouched
                std::memcpy(sl.iq_out + r * nsteps * 2, sl.h_out + o.iq + r * nsteps * 8, nsteps * 8);
    }
    std::memcpy(sl.axc, sl.h_out + o.axc, rows * static_cast<size_t>(sl.nbatches));
    if (sl.stats)
        std::memcpy(sl.stats, sl.h_out + o.stats, rows * sizeof(mi_channel_stats));
    sl.busy = false;
    return MI_OK;
}

int check_host_call(mi_demod* h, const uint8_t* const* iq, int nbatches, float* waveout, char* axc) {
    if (!h || !iq || !waveout || !axc)
        return fail(MI_ERR_INVALID, "NULL argument");
    if (h->failed)
        return fail(MI_ERR_HIP, kFailedMsg);
    if (nbatches < 1 || nbatches > h->max_batches)
        return fail(MI_ERR_INVALID, "nbatches out of range for this handle");
    return MI_OK;
}

}  // namespace

void* mi_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}

void mi_host_free(void* p) {
    if (p)
        (void)hipHostFree(p);
}

int mi_demod_wait(mi_demod* h) {
    if (!h)
        return fail(MI_ERR_INVALID, "NULL handle");
    if (h->in_flight == 0)
        return fail(MI_ERR_INVALID, "no submitted call is in flight");
    HIP_TRY(hipSetDevice(h->gpu));
    const int k = h->slot_oldest;
    int rc = slot_collect(h, k);
    h->slot_oldest = (h->slot_oldest + 1) % mi_demod::kSlots;
    h->in_flight--;
    return rc;
}

int mi_demod_submit(mi_demod* h, const uint8_t* const* iq, int nbatches, float* waveout, float* iq_out, char* axc, mi_channel_stats* stats) {
    int rc = check_host_call(h, iq, nbatches, waveout, axc);
    if (rc != MI_OK)
        return rc;
    HIP_TRY(hipSetDevice(h->gpu));
    if (h->in_flight == mi_demod::kSlots) {  // every slot taken: the oldest call completes first (its outputs become valid here)
        rc = mi_demod_wait(h);
        if (rc != MI_OK)
            return rc;
    }
    const int k = h->slot_next;
    rc = slot_prepare(h, k);
    if (rc != MI_OK)
        return rc;
    mi_demod::Slot& sl = h->slot[k];
    sl.nbatches = nbatches;
    sl.waveout = waveout, sl.iq_out = iq_out, sl.axc = axc, sl.stats = stats;
    rc = slot_launch(h, k, iq, nbatches, iq_out != nullptr, stats != nullptr, /*pipelined=*/true);
    if (rc != MI_OK)
        return rc;
    sl.busy = true;
    if (h->in_flight == 0)
        h->slot_oldest = k;
    h->slot_next = (k + 1) % mi_demod::kSlots;
    h->in_flight++;
    return MI_OK;
}

int mi_demod_process(mi_demod* h, const uint8_t* const* iq, int nbatches, float* waveout, float* iq_out, char* axc, mi_channel_stats* stats) {
    int rc = check_host_call(h, iq, nbatches, waveout, axc);
    if (rc != MI_OK)
        return rc;
    HIP_TRY(hipSetDevice(h->gpu));
    while (h->in_flight > 0) {  // calls complete in order
        rc = mi_demod_wait(h);
        if (rc != MI_OK)
            return rc;
    }
    rc = slot_prepare(h, 0);
    if (rc != MI_OK)
        return rc;
    mi_demod::Slot& sl = h->slot[0];
    sl.nbatches = nbatches;
    sl.waveout = waveout, sl.iq_out = iq_out, sl.axc = axc, sl.stats = stats;
    // one call at a time: everything in order on the handle's own stream (the shortest path for the reference's cadence of one
    // WAVE_BATCH per call)
    rc = slot_launch(h, 0, iq, nbatches, iq_out != nullptr, stats != nullptr, /*pipelined=*/false);
    if (rc != MI_OK)
        return rc;
    return slot_collect(h, 0);
}

int mi_demod_process_planes(mi_demod* h, const float* mag, const float* cplx, int nbatches, float* waveout, float* iq_out, char* axc, mi_channel_stats* stats) {
    if (!h || !mag || !waveout || !axc)
        return fail(MI_ERR_INVALID, "NULL argument");
    if (nbatches < 1 || nbatches > h->max_batches)
        return fail(MI_ERR_INVALID, "nbatches out of range for this handle");
    if (h->plan.any_afc)
        return fail(MI_ERR_UNSUPPORTED, "AFC channels read the spectrum of stage 1: no plane entry for them");
    if (h->in_flight)
        return fail(MI_ERR_INVALID, "submitted calls are in flight");
    const size_t zrows = static_cast<size_t>(h->nstreams) * h->plan.n_iq_rows;
    if (zrows && !cplx)
        return fail(MI_ERR_INVALID, "this plan has raw-I/Q rows: cplx planes are needed");
    HIP_TRY(hipSetDevice(h->gpu));
    const size_t rows = static_cast<size_t>(h->rows), count = static_cast<size_t>(n_fft_for(h, nbatches));
    const size_t nsteps = static_cast<size_t>(nbatches) * mi::kWaveBatch, wlen = nsteps + mi::kAgcExtra;
    float *d_m = nullptr, *d_wo = nullptr;
    float2 *d_z = nullptr, *d_io = nullptr;
    char* d_ax = nullptr;
    hipError_t e = dalloc(&d_m, rows * count);
    if (e == hipSuccess && zrows)
        e = dalloc(&d_z, zrows * count);
    if (e == hipSuccess)
        e = dalloc(&d_wo, rows * wlen);
    if (e == hipSuccess && iq_out)
        e = dalloc(&d_io, rows * nsteps);
    if (e == hipSuccess)
        e = dalloc(&d_ax, rows * static_cast<size_t>(nbatches));
    if (e == hipSuccess)
        e = hipMemcpy(d_m, mag, rows * count * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess && zrows)
        e = hipMemcpy(d_z, cplx, zrows * count * 8, hipMemcpyHostToDevice);
    int rc = MI_OK;
    if (e == hipSuccess) {
        h->inject_mag = d_m, h->inject_cplx = d_z, h->inject_count = count;
        // (the IQ pointer is never read: stage 1 is the copy above; the audio goes to [row][wlen] like the host entry's)
        rc = enqueue(h, reinterpret_cast<const unsigned char*>(d_m), 0, 0, nbatches, d_wo, wlen, d_io, nsteps, d_ax, h->own_stream);
        h->inject_mag = nullptr, h->inject_cplx = nullptr, h->inject_count = 0;
        if (rc == MI_OK)
            e = hipDeviceSynchronize();
    }
    if (rc == MI_OK && e == hipSuccess)
        e = hipMemcpy2D(waveout, wlen * 4, d_wo, wlen * 4, nsteps * 4, rows, hipMemcpyDeviceToHost);
    if (rc == MI_OK && e == hipSuccess)  // the lookahead: what the handle carries to the next call
        e = hipMemcpy2D(waveout + nsteps, wlen * 4, h->d_carry, mi::kAgcExtra * 4, mi::kAgcExtra * 4, rows, hipMemcpyDeviceToHost);
    if (rc == MI_OK && e == hipSuccess && iq_out)
        e = hipMemcpy(iq_out, d_io, rows * nsteps * 8, hipMemcpyDeviceToHost);
    if (rc == MI_OK && e == hipSuccess)
        e = hipMemcpy(axc, d_ax, rows * static_cast<size_t>(nbatches), hipMemcpyDeviceToHost);
    if (rc == MI_OK && e == hipSuccess && stats)
        e = hipMemcpy(stats, h->d_stats, rows * sizeof(mi_channel_stats), hipMemcpyDeviceToHost);
    void* tmp[] = {d_m, d_z, d_wo, d_io, d_ax};
    for (void* q : tmp)
        if (q)
            (void)hipFree(q);
    if (rc != MI_OK)
        return rc;
    HIP_TRY(e);
    return MI_OK;
}

int mi_demod_get_stats(mi_demod* h, mi_channel_stats* stats) {
    if (!h || !stats)
        return fail(MI_ERR_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(h->gpu));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(stats, h->d_stats, static_cast<size_t>(h->rows) * sizeof(mi_channel_stats), hipMemcpyDeviceToHost));
    return MI_OK;
}

// ---- checkpoint: [header][ChanState rows][carry][ring][ctcss_q][mag head][cplx head] ----
namespace {
struct StateHeader {
    uint32_t magic, rows, nch, n_iq_rows, n_ctcss_rows, first_call, fft_log, pad;
};
size_t state_bytes(const mi_demod* h) {
    const size_t rows = static_cast<size_t>(h->rows);
    return sizeof(StateHeader) + rows * sizeof(mi::ChanState) + rows * mi::kAgcExtra * 4 + rows * mi::kSquelchRing * 4 +
           static_cast<size_t>(h->nstreams) * h->plan.n_ctcss_rows * 4 * mi::kMaxTones * 4 + rows * mi::kAgcExtra * 4 +
           static_cast<size_t>(h->nstreams) * h->plan.n_iq_rows * mi::kAgcExtra * 8;
}
}  // namespace

size_t mi_demod_state_size(const mi_demod* h) {
    return h ? state_bytes(h) : 0;
}

int mi_demod_get_state(mi_demod* h, void* buf, size_t len) {
    if (!h || !buf || len < state_bytes(h))
        return fail(MI_ERR_INVALID, "state buffer too small");
    HIP_TRY(hipSetDevice(h->gpu));
    HIP_TRY(hipDeviceSynchronize());
    auto* o = static_cast<unsigned char*>(buf);
    StateHeader hd{0x4d494142u, static_cast<uint32_t>(h->rows), static_cast<uint32_t>(h->nch), static_cast<uint32_t>(h->plan.n_iq_rows),
                   static_cast<uint32_t>(h->plan.n_ctcss_rows), h->first_call ? 1u : 0u, static_cast<uint32_t>(h->plan.log2n), 0};
    std::memcpy(o, &hd, sizeof(hd));
    o += sizeof(hd);
    const size_t rows = static_cast<size_t>(h->rows);
    auto pull = [&](const void* d, size_t bytes) -> hipError_t {
        if (bytes == 0)
            return hipSuccess;
        hipError_t e = hipMemcpy(o, d, bytes, hipMemcpyDeviceToHost);
        o += bytes;
        return e;
    };
    HIP_TRY(pull(h->d_state, rows * sizeof(mi::ChanState)));
    HIP_TRY(pull(h->d_carry, rows * mi::kAgcExtra * 4));
    HIP_TRY(pull(h->d_ring, rows * mi::kSquelchRing * 4));
    HIP_TRY(pull(h->d_ctcss_q, static_cast<size_t>(h->nstreams) * h->plan.n_ctcss_rows * 4 * mi::kMaxTones * 4));
    HIP_TRY(hipMemcpy2D(o, mi::kAgcExtra * 4, h->d_mag + h->head_off, h->plane_stride * 4, mi::kAgcExtra * 4, rows, hipMemcpyDeviceToHost));
    o += rows * mi::kAgcExtra * 4;
    const size_t zrows = static_cast<size_t>(h->nstreams) * h->plan.n_iq_rows;
    if (zrows)
        HIP_TRY(hipMemcpy2D(o, mi::kAgcExtra * 8, h->d_cplx, h->plane_stride * 8, mi::kAgcExtra * 8, zrows, hipMemcpyDeviceToHost));
    return MI_OK;
}

int mi_demod_set_state(mi_demod* h, const void* buf, size_t len) {
    if (!h || !buf || len < state_bytes(h))
        return fail(MI_ERR_INVALID, "state buffer too small");
    const auto* o = static_cast<const unsigned char*>(buf);
    StateHeader hd;
    std::memcpy(&hd, o, sizeof(hd));
    if (hd.magic != 0x4d494142u || hd.rows != static_cast<uint32_t>(h->rows) || hd.nch != static_cast<uint32_t>(h->nch) ||
        hd.n_iq_rows != static_cast<uint32_t>(h->plan.n_iq_rows) || hd.n_ctcss_rows != static_cast<uint32_t>(h->plan.n_ctcss_rows) ||
        hd.fft_log != static_cast<uint32_t>(h->plan.log2n))
        return fail(MI_ERR_INVALID, "state blob does not match this handle's configuration");
    o += sizeof(hd);
    HIP_TRY(hipSetDevice(h->gpu));
    HIP_TRY(hipDeviceSynchronize());
    const size_t rows = static_cast<size_t>(h->rows);
    auto push = [&](void* d, size_t bytes) -> hipError_t {
        if (bytes == 0)
            return hipSuccess;
        hipError_t e = hipMemcpy(d, o, bytes, hipMemcpyHostToDevice);
        o += bytes;
        return e;
    };
    HIP_TRY(push(h->d_state, rows * sizeof(mi::ChanState)));
    HIP_TRY(push(h->d_carry, rows * mi::kAgcExtra * 4));
    HIP_TRY(push(h->d_ring, rows * mi::kSquelchRing * 4));
    HIP_TRY(push(h->d_ctcss_q, static_cast<size_t>(h->nstreams) * h->plan.n_ctcss_rows * 4 * mi::kMaxTones * 4));
    HIP_TRY(hipMemcpy2D(h->d_mag, h->plane_stride * 4, o, mi::kAgcExtra * 4, mi::kAgcExtra * 4, rows, hipMemcpyHostToDevice));
    h->head_off = 0;
    h->ser_head_next = false;
    h->chain_live = false;  // the chain state of the time-parallel path is re-seeded from the restored ChanState
    o += rows * mi::kAgcExtra * 4;
    const size_t zrows = static_cast<size_t>(h->nstreams) * h->plan.n_iq_rows;
    if (zrows)
        HIP_TRY(hipMemcpy2D(h->d_cplx, h->plane_stride * 8, o, mi::kAgcExtra * 8, mi::kAgcExtra * 8, zrows, hipMemcpyHostToDevice));
    h->first_call = hd.first_call != 0;
    h->failed = false;
    return MI_OK;
}

int mi_demod_last_path(mi_demod* h, int* time_parallel, int* unverified_rows) {
    if (!h)
        return fail(MI_ERR_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(h->gpu));
    HIP_TRY(hipDeviceSynchronize());
    if (time_parallel)
        *time_parallel = h->last_path;
    if (unverified_rows) {
        *unverified_rows = 0;
        if (h->last_path == 1) {
            std::vector<mi::TpFinal> f(static_cast<size_t>(h->tp_rows));
            HIP_TRY(hipMemcpy(f.data(), h->d_fin, f.size() * sizeof(mi::TpFinal), hipMemcpyDeviceToHost));
            for (const mi::TpFinal& x : f)
                *unverified_rows += x.all_ok ? 0 : 1;
        }
    }
    return MI_OK;
}

int mi_demod_pre_wave_timeouts(mi_demod* h, unsigned* count) {
    if (!h || !count)
        return fail(MI_ERR_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(h->gpu));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(count, h->d_pre_timeouts, sizeof(unsigned), hipMemcpyDeviceToHost));
    return MI_OK;
}

int mi_demod_last_stage1(mi_demod* h, int* kind) {
    if (!h || !kind)
        return fail(MI_ERR_INVALID, "NULL argument");
    *kind = h->last_stage1;
    return MI_OK;
}

int mi_demod_tp_debug(mi_demod* h, int row, float* core4, int max_entries, int* diag4, int* nseg) {
    if (!h || row < 0 || row >= h->tp_rows || !h->tp_eligible || h->last_path != 1)
        return fail(MI_ERR_INVALID, "the last call did not take the time-parallel path");
    HIP_TRY(hipSetDevice(h->gpu));
    HIP_TRY(hipDeviceSynchronize());
    if (nseg)
        *nseg = static_cast<int>(h->last_nseg[h->cur]);
    if (core4) {
        const size_t n = std::min<size_t>(static_cast<size_t>(max_entries), h->last_nseg[h->cur] + 1);
        HIP_TRY(hipMemcpy(core4, h->d_core[h->cur] + static_cast<size_t>(row) * (h->last_nseg[h->cur] + 1), n * sizeof(mi::TpCore), hipMemcpyDeviceToHost));
    }
    if (diag4) {  // [0..3] scan rounds, [4..7] core-chain blocks: in accepted runs, single O(1), stepped, failed hypotheses
        HIP_TRY(hipMemcpy(diag4, h->d_diag + static_cast<size_t>(row) * 4, 4 * sizeof(int), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(diag4 + 4, h->d_diag + static_cast<size_t>(h->tp_rows) * 4 + static_cast<size_t>(row) * 4, 4 * sizeof(int), hipMemcpyDeviceToHost));
    }
    return MI_OK;
}

int mi_demod_read_planes(mi_demod* h, int stream, int ch, int first, int count, float* mag, float* iq) {
    if (!h || !mag || stream < 0 || stream >= h->nstreams || ch < 0 || ch >= h->nch || first < 0 || count < 0 ||
        static_cast<size_t>(first) + count > h->plane_stride)
        return fail(MI_ERR_INVALID, "bad plane range");
    HIP_TRY(hipSetDevice(h->gpu));
    HIP_TRY(hipDeviceSynchronize());
    const size_t row = static_cast<size_t>(stream) * h->nch + ch;
    const float* pm = h->serial_pipe ? h->d_mag_last : h->d_mag;  // (a pipelined serial call leaves d_mag on the set with the next head)
    const float2* pz = h->serial_pipe ? h->d_cplx_last : h->d_cplx;
    HIP_TRY(hipMemcpy(mag, pm + row * h->plane_stride + first, static_cast<size_t>(count) * 4, hipMemcpyDeviceToHost));
    const int iq_row = h->plan.cp[ch].iq_row;
    if (iq && iq_row >= 0) {
        const size_t zrow = static_cast<size_t>(stream) * h->plan.n_iq_rows + iq_row;
        HIP_TRY(hipMemcpy(iq, pz + zrow * h->plane_stride + first, static_cast<size_t>(count) * 8, hipMemcpyDeviceToHost));
    }
    return MI_OK;
}

// timing of the last call (age 0) or of the call before it (age 1, only while its event set has not been reused)
static int kernel_time_of(mi_demod* h, int age, int index, const char** name, float* ms_total, int* launches) {
    if (!h || index < 0 || age < 0 || age >= mi_demod::kSets)
        return fail(MI_ERR_INVALID, "bad argument");
    const int q = (h->cur + mi_demod::kSets - age) % mi_demod::kSets;
    if (!h->set_seq[q] || h->set_seq[q] + static_cast<uint64_t>(age) != h->set_seq[h->cur])
        return fail(MI_ERR_INVALID, "that call has not been timed (or its events were reused)");
    HIP_TRY(hipSetDevice(h->gpu));
    hipEvent_t* evq = h->ev[q];
    HIP_TRY(hipEventSynchronize(evq[2]));
    float t = 0.f;
    int n = 1;
    const char* nm = nullptr;
    if (h->set_path[q] == 0 || h->set_path[q] == 2) {
        if (index > 1)
            return fail(MI_ERR_INVALID, "kernel index out of range");
        nm = index == 0 ? "k_channelize" : "k_demod";
        if (h->set_path[q] == 2 && index == 1)  // pipelined serial call: k_demod starts at its own event on the caller's stream
            HIP_TRY(hipEventElapsedTime(&t, evq[3], evq[2]));
        else
            HIP_TRY(hipEventElapsedTime(&t, evq[index], evq[index + 1]));
    } else {
        // per chunk events: 0 stage1 begin, 1 stage1 end, 2 k_tp_full end (front stream), 3 core begin, 4 core end (aux stream),
        // 5 seg begin, 12 seg end, 6 all segment launches of the chunk done (segment stream), 10 scan#0 begin, 7 scan#0 end, 8 fix#0 + redo#0 end, 9 finish end (caller's stream), 11 k_tp_full begin
        static const char* const names[] = {"k_channelize", "k_tp_full", "k_tp_core", "k_tp_seg", "k_tp_scan#0", "k_tp_fix#0", "k_tp_rest"};
        static const int from[] = {0, 11, 3, 5, 10, 7, 8}, to[] = {1, 2, 4, 12, 7, 8, 9};
        if (index == 7 && h->set_mixed[q]) {  // a mixed plan: the serial kernel of the other rows, on its own stream
            HIP_TRY(hipEventElapsedTime(&t, evq[3], evq[4]));
            if (name)
                *name = "k_demod";
            if (ms_total)
                *ms_total = t;
            if (launches)
                *launches = 1;
            return MI_OK;
        }
        if (index > 6)
            return fail(MI_ERR_INVALID, "kernel index out of range");
        nm = names[index];
        n = h->tp_chunks[q];
        const std::vector<hipEvent_t>& cev = h->chunk_ev[q];
        for (int i = 0; i < n; ++i) {
            float d = 0.f;
            HIP_TRY(hipEventElapsedTime(&d, cev[static_cast<size_t>(i) * mi_demod::kEvPerChunk + from[index]],
                                        cev[static_cast<size_t>(i) * mi_demod::kEvPerChunk + to[index]]));
            t += d;
        }
    }
    if (name)
        *name = nm;
    if (ms_total)
        *ms_total = t;
    if (launches)
        *launches = n;
    return MI_OK;
}

int mi_demod_kernel_time(mi_demod* h, int index, const char** name, float* ms_total, int* launches) {
    return kernel_time_of(h, 0, index, name, ms_total, launches);
}

int mi_demod_kernel_time_prev(mi_demod* h, int age, int index, const char** name, float* ms_total, int* launches) {
    return kernel_time_of(h, age, index, name, ms_total, launches);
}

int mi_demod_event_ms(mi_demod* h, int ref_age, int age, int chunk, int event, float* ms) {
    if (!h || !ms || age < 0 || ref_age < age || ref_age >= mi_demod::kSets || event < 0 || event >= mi_demod::kEvPerChunk || chunk < 0)
        return fail(MI_ERR_INVALID, "bad argument");
    const int q = (h->cur + mi_demod::kSets - age) % mi_demod::kSets, qr = (h->cur + mi_demod::kSets - ref_age) % mi_demod::kSets;
    for (const int s : {q, qr})
        if (!h->set_seq[s] || h->set_path[s] != 1)
            return fail(MI_ERR_INVALID, "that call was not a time-parallel one (or its events were reused)");
    if (h->set_seq[q] + static_cast<uint64_t>(age) != h->set_seq[h->cur] || h->set_seq[qr] + static_cast<uint64_t>(ref_age) != h->set_seq[h->cur] ||
        chunk >= h->tp_chunks[q])
        return fail(MI_ERR_INVALID, "no such call or chunk");
    HIP_TRY(hipSetDevice(h->gpu));
    HIP_TRY(hipEventSynchronize(h->ev[q][2]));
    HIP_TRY(hipEventElapsedTime(ms, h->chunk_ev[qr][3], h->chunk_ev[q][static_cast<size_t>(chunk) * mi_demod::kEvPerChunk + event]));
    return MI_OK;
}

int mi_demod_last_kernel_ms(mi_demod* h, float* channelize_ms, float* demod_ms) {
    if (!h || !h->set_seq[h->cur])
        return fail(MI_ERR_INVALID, "no call has been timed yet");
    HIP_TRY(hipSetDevice(h->gpu));
    hipEvent_t* evq = h->ev[h->cur];
    HIP_TRY(hipEventSynchronize(evq[2]));
    float a = 0.f, b = 0.f;
    if (h->last_path == 0) {
        HIP_TRY(hipEventElapsedTime(&a, evq[0], evq[1]));
        HIP_TRY(hipEventElapsedTime(&b, evq[h->serial_pipe ? 3 : 1], evq[2]));
    } else {  // pipelined: stage 1 summed over the chunks, stage 2 = the rest of the call's wall time on the stream
        float total = 0.f;
        HIP_TRY(hipEventElapsedTime(&total, evq[0], evq[2]));
        int rc = mi_demod_kernel_time(h, 0, nullptr, &a, nullptr);
        if (rc != MI_OK)
            return rc;
        b = total - a;
    }
    if (channelize_ms)
        *channelize_ms = a;
    if (demod_ms)
        *demod_ms = b;
    return MI_OK;
}

int mi_demod_set_option(mi_demod* h, int option, int value) {
    if (!h)
        return fail(MI_ERR_INVALID, "NULL handle");
    switch (option) {
        case MI_OPT_EARLY_INPUT:
            h->early_input = value != 0;
            return MI_OK;
        case MI_OPT_STEADY_BLOCKS:
            h->steady_blocks = value != 0;
            return MI_OK;
        case MI_OPT_TIME_PARALLEL:
            h->opt_tp = value < 0 ? -1 : (value != 0 ? 1 : 0);
            return MI_OK;
        case MI_OPT_PRUNE_FFT:
            h->opt_prune = value != 0;
            return MI_OK;
        case MI_OPT_U8_CONVERSION:
            h->opt_conv = value < 0 ? -1 : (value != 0 ? 1 : 0);
            return MI_OK;
        case MI_OPT_UNI_ROWS:
            if (value < 1)
                return fail(MI_ERR_INVALID, "MI_OPT_UNI_ROWS must be >= 1");
            h->opt_uni_rows = value;
            return MI_OK;
        case MI_OPT_TP_CHUNKS:
            h->opt_tp_chunks = std::max(0, value);
            return MI_OK;
        case MI_OPT_TP_RATIO_PCT:
            h->opt_tp_ratio = value <= 0 ? 0.0 : std::max(0.25, value / 100.0);
            return MI_OK;
        case MI_OPT_TP_SEG_LANES:
            h->opt_tp_lpw = (value >= 1 && value <= 64) ? value : 0;
            return MI_OK;
        case MI_OPT_LANE_FFT:
            h->opt_l64 = value != 0;
            return MI_OK;
        case MI_OPT_CORE_SPLIT:
            h->opt_core_split = value != 0;
            return MI_OK;
        case MI_OPT_SPEC_HEAD:
            h->opt_spec_head = value != 0;
            return MI_OK;
        case MI_OPT_RESERVE_CUS:
            if (h->masked_state != 0)
                return fail(MI_ERR_INVALID, "MI_OPT_RESERVE_CUS is decided at the handle's first time-parallel call: set it before");
            h->opt_reserve_cus = value < 0 ? -1 : value;
            return MI_OK;
        case MI_OPT_PRE_WAVE:
            h->opt_pre_wave = value < 0 ? -1 : std::min(2, value);
            return MI_OK;
        case MI_OPT_LANE_FFT_JIT:
            h->opt_l64_jit = value != 0;
            return MI_OK;
        case MI_OPT_AUDIO_WAVE:
            h->opt_audio_wave = value != 0;
            return MI_OK;
        case MI_OPT_MIXED_PLAN:
            h->opt_mixed = value != 0;
            return MI_OK;
        case MI_OPT_SPLIT_CUS:
            if (h->split_state != 0)
                return fail(MI_ERR_INVALID, "MI_OPT_SPLIT_CUS is decided at the handle's first overlapped serial call: set it before");
            h->opt_split_cus = value < 0 ? -1 : value;
            return MI_OK;
        default:
            return fail(MI_ERR_INVALID, "unknown option");
    }
}

// ---------------- host-only plan views ----------------

int mi_plan_create(const mi_device_cfg* dev, const mi_channel_cfg* chans, int nch, mi_plan** out) {
    if (!out || !dev)
        return fail(MI_ERR_INVALID, "NULL argument");
    *out = nullptr;
    mi_plan* p = new (std::nothrow) mi_plan();
    if (!p)
        return fail(MI_ERR_NOMEM, "host allocation failed");
    const char* msg = "";
    int rc = mi::build_plan(*dev, chans, nch, p->plan, &msg);
    if (rc != MI_OK) {
        delete p;
        return fail(rc, msg);
    }
    *out = p;
    return MI_OK;
}

void mi_plan_destroy(mi_plan* p) {
    delete p;
}

int mi_plan_fft_size(const mi_plan* p) {
    return p ? p->plan.fft_size : 0;
}

int mi_plan_window(const mi_plan* p, float* out) {
    if (!p || !out)
        return fail(MI_ERR_INVALID, "NULL argument");
    std::memcpy(out, p->plan.window.data(), p->plan.window.size() * 4);
    return MI_OK;
}

int mi_plan_twiddles(const mi_plan* p, float* out) {
    if (!p || !out)
        return fail(MI_ERR_INVALID, "NULL argument");
    std::memcpy(out, p->plan.tw.data(), p->plan.tw.size() * 4);
    return MI_OK;
}

int mi_plan_levels(const mi_plan* p, float* out) {
    if (!p || !out)
        return fail(MI_ERR_INVALID, "NULL argument");
    std::memcpy(out, p->plan.levels.data(), 256 * 4);
    return MI_OK;
}

int mi_plan_sincos_lut(const mi_plan* p, float* sin_out, float* cos_out) {
    if (!p || !sin_out || !cos_out)
        return fail(MI_ERR_INVALID, "NULL argument");
    std::memcpy(sin_out, p->plan.sin_lut, 257 * 4);
    std::memcpy(cos_out, p->plan.cos_lut, 257 * 4);
    return MI_OK;
}

int mi_plan_channel(const mi_plan* p, int ch, mi_channel_derived* out) {
    if (!p || !out || ch < 0 || ch >= p->plan.nch)
        return fail(MI_ERR_INVALID, "bad channel index");
    const mi::ChanParams& c = p->plan.cp[ch];
    out->bin = c.bin;
    out->dm_dphi = c.dm_dphi;
    out->needs_raw_iq = c.needs_raw_iq;
    out->has_iq_outputs = c.has_iq_outputs;
    out->modulation = c.modulation;
    out->using_manual_level = c.using_manual_level;
    out->manual_signal_level = c.manual_signal_level;
    out->normal_signal_ratio = c.normal_signal_ratio;
    out->flappy_signal_ratio = c.flappy_signal_ratio;
    out->ampfactor = c.ampfactor;
    out->alpha = c.alpha;
    out->notch_enabled = c.notch_enabled;
    out->notch_d[0] = c.notch_d0;
    out->notch_d[1] = c.notch_d1;
    out->notch_d[2] = c.notch_d2;
    out->lowpass_enabled = c.lowpass_enabled;
    out->lowpass_gain = c.lowpass_gain;
    out->lowpass_ycoeffs[0] = c.lowpass_yc0;
    out->lowpass_ycoeffs[1] = c.lowpass_yc1;
    out->ctcss_enabled = c.ctcss_enabled;
    out->ctcss_fast_window = c.ctcss_fast_window;
    out->ctcss_slow_window = c.ctcss_slow_window;
    out->ctcss_fast_ndet = c.ctcss_fast_ndet;
    out->ctcss_slow_ndet = c.ctcss_slow_ndet;
    return MI_OK;
}

int mi_plan_ctcss_coeffs(const mi_plan* p, int ch, int slow, float* out) {
    if (!p || !out || ch < 0 || ch >= p->plan.nch)
        return fail(MI_ERR_INVALID, "bad channel index");
    const mi::ChanParams& c = p->plan.cp[ch];
    if (!c.ctcss_enabled)
        return fail(MI_ERR_INVALID, "channel has no ctcss");
    const float* base = p->plan.ctcss_coeff.data() + (static_cast<size_t>(c.ctcss_row) * 2 + (slow ? 1 : 0)) * mi::kMaxTones;
    std::memcpy(out, base, static_cast<size_t>(slow ? c.ctcss_slow_ndet : c.ctcss_fast_ndet) * 4);
    return MI_OK;
}

// ---------------- synthetic IQ ----------------

int mi_iqgen_host(const mi_iqgen_cfg* cfg, uint32_t stream_id, uint64_t first, uint64_t count, uint8_t* out) {
    if (!cfg || !out || cfg->ncarriers < 0 || cfg->ncarriers > 64 || cfg->sample_rate <= 0)
        return fail(MI_ERR_INVALID, "bad iqgen configuration");
    mi::IqGenDerived g;
    mi::iqgen_derive(*cfg, g);
    const int16_t* tab = mi::iqgen_sine_table();
    for (uint64_t i = 0; i < count; ++i)
        mi::iq_sample(g, tab, stream_id, first + i, out + 2 * i);
    return MI_OK;
}

int mi_iqgen_device(const mi_iqgen_cfg* cfg, uint32_t first_stream_id, uint32_t nstreams, size_t stream_stride_bytes, uint64_t first,
                    uint64_t count, void* d_out, void* hip_stream) {
    if (!cfg || !d_out || cfg->ncarriers < 0 || cfg->ncarriers > 64 || cfg->sample_rate <= 0)
        return fail(MI_ERR_INVALID, "bad iqgen configuration");
    if (reinterpret_cast<uintptr_t>(d_out) % 16 != 0 || stream_stride_bytes % 16 != 0)
        return fail(MI_ERR_INVALID, "iqgen output must be 16-byte aligned");
    mi::IqGenDerived g;
    mi::iqgen_derive(*cfg, g);
    mi::IqGenDerived* d_cfg = nullptr;
    int16_t* d_tab = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_cfg), sizeof(g)));
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_tab), 1024 * sizeof(int16_t));
    if (e != hipSuccess) {
        (void)hipFree(d_cfg);
        return hip_fail(e, "hipMalloc");
    }
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    int rc = MI_OK;
    if ((e = hipMemcpy(d_cfg, &g, sizeof(g), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(d_tab, mi::iqgen_sine_table(), 1024 * sizeof(int16_t), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = mi::launch_iqgen(d_cfg, d_tab, first_stream_id, nstreams, stream_stride_bytes, first, count, static_cast<unsigned char*>(d_out), s)) !=
            hipSuccess ||
        (e = hipStreamSynchronize(s)) != hipSuccess)
        rc = hip_fail(e, "mi_iqgen_device");
    (void)hipFree(d_cfg);
    (void)hipFree(d_tab);
    return rc;
}

}  // extern "C"

namespace mi {

void iqgen_derive(const mi_iqgen_cfg& cfg, IqGenDerived& out) {
    std::memset(&out, 0, sizeof(out));
    out.seed = cfg.seed;
    out.gate_samples = cfg.gate_samples;
    out.noise_q8_mul = cfg.noise_q8_mul;
    out.ncarriers = cfg.ncarriers;
    const double turn = 4294967296.0;
    for (int k = 0; k < cfg.ncarriers; ++k) {
        const mi_iqgen_carrier& c = cfg.carriers[k];
        const long long d = std::llround(static_cast<double>(c.offset_hz) / cfg.sample_rate * turn);
        out.c[k].dphi = static_cast<uint32_t>(static_cast<uint64_t>(d));
        out.c[k].dpsi_1k = static_cast<uint32_t>(std::llround(1000.0 / cfg.sample_rate * turn));
        out.c[k].dpsi_100 = static_cast<uint32_t>(std::llround(100.0 / cfg.sample_rate * turn));
        out.c[k].kind = c.kind;
        out.c[k].amp_q8 = c.amp_q8;
        out.c[k].gate_phase = c.gate_phase;
    }
}

const int16_t* iqgen_sine_table() {
    static int16_t tab[1024];
    static bool init = false;
    if (!init) {
        for (int i = 0; i < 1024; ++i)
            tab[i] = static_cast<int16_t>(std::lround(32767.0 * std::sin(2.0 * M_PI * i / 1024.0)));
        init = true;
    }
    return tab;
}

}  // namespace mi
