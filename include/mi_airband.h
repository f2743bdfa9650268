/*
 * mi_airband.h -- C ABI of the MI355X-native channelizer/demodulator (libmi_airband.so).
 *
 * This is the drop-in boundary for ONE path of Boondock-Airband: the body of demodulate()
 * (reference src/rtl_airband.cpp:308-694) -- sample conversion x window, sliding FFT, bin pick,
 * and the per-channel squelch / AM / NFM / CTCSS / filter loop -- between the reference's two
 * shared-memory contracts: the input_t byte ring upstream (src/input-common.h:39-57) and
 * channel_t.{waveout, iq_out, axcindicate} + device_t.waveavail downstream
 * (src/boondock_airband.h:243-297).  Plain pointers and sizes only; no C++ types, no exceptions,
 * never exits the process.  Every function returns 0 on success or a negative mi_status;
 * mi_last_error() gives the message (the reference's own convention for inputs is int 0/-1,
 * src/input-common.cpp:56-131; its VideoCore FFT seam returns -1/-2/-3, src/rtl_airband.cpp:318-332).
 *
 * Precedent seam in the reference: gpu_fft_prepare/gpu_fft_execute/gpu_fft_release
 * (src/hello_fft/gpu_fft.h:60-74) + samplefft() (src/boondock_airband.h:87-92).  This ABI replaces
 * the whole batch body, not only the FFT, so nothing but u8 IQ goes down and audio comes up.
 */
#ifndef MI_AIRBAND_H
#define MI_AIRBAND_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* build-time constants of the reference's NFM build (src/boondock_airband.h:64-75) */
#define MI_WAVE_RATE 16000
#define MI_WAVE_BATCH 2000
#define MI_AGC_EXTRA 100

typedef enum {
    MI_OK = 0,
    MI_ERR_INVALID = -1,     /* bad argument / unsupported configuration */
    MI_ERR_NO_DEVICE = -2,   /* HIP runtime or GPU missing: the product path never falls back to CPU */
    MI_ERR_NOMEM = -3,       /* host or device allocation failed */
    MI_ERR_HIP = -4,         /* a HIP call or kernel launch failed */
    MI_ERR_UNSUPPORTED = -5  /* valid in the reference, not built in this library */
} mi_status;

enum { MI_MOD_AM = 0, MI_MOD_NFM = 1 };                                        /* enum modulations, boondock_airband.h:202-208 */
enum { MI_SFMT_U8 = 1, MI_SFMT_S8 = 2, MI_SFMT_S16 = 3, MI_SFMT_F32 = 4 };     /* sample_format_t, input-common.h:31 */
enum { MI_NO_SIGNAL = ' ', MI_SIGNAL = '*', MI_AFC_UP = '<', MI_AFC_DOWN = '>' }; /* enum status, boondock_airband.h:101 */

/* The DSP-relevant subset of a `devices` entry (src/config.cpp:731-808). */
typedef struct mi_device_cfg {
    int sample_rate;  /* Hz, input_t.sample_rate */
    int centerfreq;   /* Hz, input_t.centerfreq */
    int fft_size_log; /* global fft_size = 1<<log, 8..13 (boondock_airband.h:80-82) */
    int sfmt;         /* MI_SFMT_*, input_t.sfmt */
    float fullscale;  /* input_t.fullscale (s16/f32 only) */
    int tau;          /* device `tau` in us; <0: the global default 200 us (rtl_airband.cpp:87) */
    int fm_quadri;    /* 0: FM_FAST_ATAN2; 1: FM_QUADRI_DEMOD (the -Q flag) */
} mi_device_cfg;

/* The DSP-relevant subset of a `channels` entry (src/config.cpp:312-729), multichannel mode. */
typedef struct mi_channel_cfg {
    int freq;                   /* Hz */
    int modulation;             /* MI_MOD_* */
    int squelch_threshold_dbfs; /* `squelch_threshold`: 0 = unset/auto, <0 manual level in dBFS */
    int has_snr_threshold;      /* `squelch_snr_threshold` present */
    float squelch_snr_db;       /*   its value; -1 keeps the default 9.54 dB */
    float notch_freq;           /* `notch` Hz, 0 = none */
    float notch_q;              /* `notch_q`, 0 = default 10 */
    float ctcss_freq;           /* `ctcss` Hz, 0 = none */
    int bandwidth;              /* `bandwidth` Hz: 0 = key absent; > 0 derotation + low-pass at bandwidth/2; < 0 = key present with the
                                 * value 0 or a negative one: the reference sets needs_raw_iq and builds no filter (config.cpp:595-622) */
    float ampfactor;            /* `ampfactor`, default 1 */
    int tau;                    /* channel `tau` us, <0 inherit the device's */
    int afc;                    /* `afc` 0..255 (rtl_airband.cpp:180-251): >0 lets the picked bin follow the carrier; such a
                                 * handle processes its batches one at a time on the device and reports MI_AFC_UP / MI_AFC_DOWN */
    int has_iq_outputs;         /* the channel has a rawfile output (config.cpp:162) */
} mi_channel_cfg;

/* What the reference's stats file / TUI / JSON status read through Squelch getters on freq_t
 * (src/output.cpp:634-811, rtl_airband.cpp:654-665), mirrored back once per call. */
typedef struct mi_channel_stats {
    float noise_level;   /* Squelch::noise_level() */
    float signal_level;  /* Squelch::signal_level() */
    float squelch_level; /* Squelch::squelch_level(), evaluated without touching its cache */
    float agcavgfast;    /* freq_t.agcavgfast */
    uint64_t open_count, flappy_count, ctcss_count, no_ctcss_count;
    uint64_t active_counter; /* freq_t.active_counter (batches with axcindicate != NO_SIGNAL) */
    int32_t squelch_state;   /* Squelch::State of current_state_ */
    int32_t signal_outside_filter;
} mi_channel_stats;

typedef struct mi_demod mi_demod; /* nstreams independent device streams x nch channels, state resident in HBM */

const char* mi_last_error(void);
/* number of visible GPUs (0 if none / no runtime) -- never initialises more than the HIP runtime */
int mi_device_count(void);

/* Replaces init_demod() + the per-thread setup at the top of demodulate() (rtl_airband.cpp:253-266,
 * 338-373): builds window, level LUTs, twiddles, per-channel derived parameters exactly as the
 * reference derives them on the host, allocates device memory for up to max_batches WAVE_BATCHes per
 * call, and resets every channel to the reference's initial state (config.cpp:271-287,319-334).
 * All nstreams streams share the channel plan (the reference's "several dongles, same config" case). */
int mi_demod_create(const mi_device_cfg* dev, const mi_channel_cfg* chans, int nch, int nstreams, int max_batches, int gpu,
                    mi_demod** out);
void mi_demod_destroy(mi_demod* h);

/* Everything slow happens before the first batch: like init_demod() (rtl_airband.cpp:1058-1082) this runs before the input
 * threads start, because the reference's ring holds only 0.5 s of u8 IQ (config.cpp:799-805; overflow: input-helpers.cpp:56-61).
 *   mi_demod_create   also obtains the stage-1 kernel of the handle's own channel plan (fft_size 512): compiled with hipRTC
 *                     (0.3-0.6 s) or, from the second process start on, loaded from the code-object cache (milliseconds).
 *   mi_demod_prepare  additionally allocates the page-locked staging of `host_slots` (0 .. 3) host-buffer calls in flight
 *                     (mi_demod_process needs 1, mi_demod_submit / _wait up to 3); without it a slot is allocated by the first
 *                     call that uses it.  The first mi_demod_process of a prepared handle does no allocation and no compilation.
 *   mi_set_cache_dir  where code objects are kept (process-wide; call before mi_demod_create).  NULL or "" switches the cache
 *                     off.  Default: $MI_AIRBAND_CACHE_DIR, else $XDG_CACHE_HOME/mi_airband, else ~/.cache/mi_airband.  A file
 *                     is keyed by GPU architecture, runtime version, plan and kernel source, written atomically, checked on load.
 *   mi_jit_counts     kernels this process compiled / loaded from the cache (diagnostic). */
int mi_demod_prepare(mi_demod* h, int host_slots);
int mi_set_cache_dir(const char* dir);
int mi_jit_counts(int* compiled, int* from_cache);

/* Ring accounting for the caller (rtl_airband.cpp:416-417, 691).  A call producing nbatches batches
 * runs n_fft windows, hop_bytes apart, the last one fft_size samples long:
 *   needed   = (n_fft-1)*hop_bytes + 2*bytes_per_sample*fft_size   contiguous bytes from the stream position
 *   consumed = n_fft*hop_bytes                                     how far input_t.bufs advances
 * n_fft = nbatches*WAVE_BATCH (+AGC_EXTRA on a handle's first call, as waveend starts at 0). */
size_t mi_demod_bytes_needed(const mi_demod* h, int nbatches);
size_t mi_demod_bytes_consumed(const mi_demod* h, int nbatches);
size_t mi_demod_hop_bytes(const mi_demod* h);

/* Host-buffer entry: the batch body of demodulate() for all streams of the handle.
 *   iq[s]      -> first byte of stream s at its current position (mi_demod_bytes_needed() readable bytes)
 *   waveout    [nstreams][nch][nbatches*WAVE_BATCH + AGC_EXTRA]: exactly channel_t.waveout after the
 *              reference's loop -- [0, nbatches*WAVE_BATCH) is what the output thread emits
 *              (output.cpp:945-950), the last AGC_EXTRA samples are the not-yet-final lookahead
 *   iq_out     [nstreams][nch][nbatches*WAVE_BATCH][2] or NULL; rows of channels without iq outputs are untouched
 *   axc        [nstreams][nch][nbatches] MI_NO_SIGNAL / MI_SIGNAL (/ MI_AFC_UP / MI_AFC_DOWN on afc channels) per batch
 *              (channel_t.axcindicate)
 *   stats      [nstreams][nch] or NULL
 * Synchronous: on return the outputs are complete (publish waveavail after this returns). */
int mi_demod_process(mi_demod* h, const uint8_t* const* iq, int nbatches, float* waveout, float* iq_out, char* axc,
                     mi_channel_stats* stats);

/* The same call split in two, so that a host can keep up to three calls in flight: mi_demod_submit() starts the upload of the IQ, both
 * stages and the download of the results and returns; mi_demod_wait() completes the OLDEST submitted call -- on return that
 * call's waveout / iq_out / axc / stats hold its results.  The upload of call k+1 (its own copy stream and device buffer)
 * then runs under the compute of call k, and the download of call k under the compute of call k+1 (on plans that take the
 * time-parallel stage 2, consecutive calls also overlap on the device as with MI_OPT_EARLY_INPUT); results are identical
 * to the same calls made one after the other.  At most three calls are in flight: a fourth mi_demod_submit() first completes
 * the oldest one.  The IQ bytes must stay valid until the call has been waited for only if they live in pinned memory (see
 * mi_host_alloc); any other source is copied into the handle's staging before mi_demod_submit() returns.
 * mi_demod_process() == mi_demod_submit() + mi_demod_wait() (after completing whatever was in flight). */
int mi_demod_submit(mi_demod* h, const uint8_t* const* iq, int nbatches, float* waveout, float* iq_out, char* axc,
                    mi_channel_stats* stats);
int mi_demod_wait(mi_demod* h);

/* Page-locked host memory the copy engine reads directly: an input_t ring (input-common.h:39-57, allocated in
 * config.cpp:804) placed here is uploaded without the staging memcpy, which is what bounds the host-buffer entries on long
 * calls (a single thread copies ~12 GB/s, the link moves ~50).  Pointers from hipHostMalloc / hipHostRegister are recognised
 * as well, also for `waveout` (the audio is then downloaded straight into it).  mi_host_alloc returns NULL on failure. */
void* mi_host_alloc(size_t bytes);
void mi_host_free(void* p);

/* Device-resident entry (capture already in HBM; used for bulk replay and by bench.py).
 *   d_iq            device pointer, stream s starts at d_iq + s*stream_stride_bytes
 *   d_waveout       device [nstreams][nch][nbatches*WAVE_BATCH] -- the emitted samples only; the
 *                   AGC_EXTRA lookahead stays in the handle; 16-byte aligned (so is d_iq_out)
 *   d_iq_out        device [nstreams][nch][nbatches*WAVE_BATCH][2] or NULL
 *   d_axc           device [nstreams][nch][nbatches]
 *   hip_stream      hipStream_t to enqueue on (NULL = default stream); asynchronous */
int mi_demod_process_device(mi_demod* h, const void* d_iq, size_t stream_stride_bytes, int nbatches, float* d_waveout,
                            float* d_iq_out, char* d_axc, void* hip_stream);

/* stats of the last completed call (synchronises the handle's stream) */
int mi_demod_get_stats(mi_demod* h, mi_channel_stats* stats /* [nstreams][nch] */);

/* Checkpoint / resume of the complete per-channel DSP state (the reference has none). */
size_t mi_demod_state_size(const mi_demod* h);
int mi_demod_get_state(mi_demod* h, void* buf, size_t len);
int mi_demod_set_state(mi_demod* h, const void* buf, size_t len);

/* Which stage-2 path the last call took: the serial per-channel kernel (0) or the time-parallel one (1,
 * plain AM channels and long enough calls; DESIGN.md "Time-parallel stage 2").  unverified_rows counts
 * channels whose segment chain was not fully accepted after the serial fallback -- always 0; it exists so
 * the tests can assert that.  Synchronises. */
int mi_demod_last_path(mi_demod* h, int* time_parallel, int* unverified_rows);

/* Which stage-1 kernel the last call ran (diagnostic; every one of them yields the same bits): the radix-8 exchange kernels
 * (full graph, or pruned to the picked bins at N = 512), or the lane-resident N = 512 kernel (prebuilt full graph, or compiled
 * for this plan's own FFT nodes by hipRTC -- MI_OPT_LANE_FFT / MI_OPT_LANE_FFT_JIT). */
enum { MI_STAGE1_EXCHANGE_FULL = 0, MI_STAGE1_EXCHANGE_PRUNED = 1, MI_STAGE1_LANE_FULL = 2, MI_STAGE1_LANE_PLAN = 3 };
int mi_demod_last_stage1(mi_demod* h, int* kind);
/* (diagnostic) how often, since the handle was created, a channel's wave of the serial kernel gave up waiting for the wave that walks
 * its squelch pre-filter ahead (k_demod_pw; it then computes everything itself, same results): expected 0 */
int mi_demod_pre_wave_timeouts(mi_demod* h, unsigned* count);

/* TEST ENTRY -- stage 2 alone over caller-supplied planes: the per-channel loop of demodulate() (rtl_airband.cpp:517-669 with Squelch,
 * CTCSS, filters) run by the same kernels and along the same paths as mi_demod_process (serial / time-parallel by nbatches),
 * with stage 1 (convert x window -> FFT -> bin pick) replaced by a copy.  It exists so that the reference's own vectors for this
 * part of the path (tests/golden/components_ref.npz: outputs of the reference's squelch.cpp / filters.cpp compiled unmodified) can
 * be fed to the HIP code directly; a host never needs it.
 *   mag   [nstreams*nch][count] the values channel_t.wavein would receive from the FFT, count = nbatches*WAVE_BATCH (+ AGC_EXTRA on the
 *         handle's first call, as waveend starts at 0): entry AGC_EXTRA + i of the handle's running sequence is the squelch's raw
 *         sample of step i (rtl_airband.cpp:529), entry i the AM audio sample (:580)
 *   cplx  [nstreams*n_iq_rows][count][2] fftout of the channels with needs_raw_iq, in channel order (NULL if the plan has none)
 *   waveout / iq_out / axc / stats  exactly as mi_demod_process.  Synchronous.  Not for AFC plans (MI_ERR_UNSUPPORTED). */
int mi_demod_process_planes(mi_demod* h, const float* mag, const float* cplx, int nbatches, float* waveout, float* iq_out, char* axc,
                            mi_channel_stats* stats);

/* Diagnostics of the time-parallel path after a call that took it (synchronises): the exact Squelch core
 * state {noise_floor_, moving_avg_cap_, pre_filter_.capped_, pre_filter_.full_} before each segment (512 .. 4096 steps, by row count)
 * (core4: [nseg+1][4]) and diag8[0..3] = segments not accepted in verification scans 0..3 (scan 3 runs after the
 * serial fallback and is always 0; diag4[2] != 0 means the fallback had to run). */
int mi_demod_tp_debug(mi_demod* h, int row, float* core4, int max_entries, int* diag8, int* nseg);

/* Diagnostic view of stage 1's output as stage 2 left it after the last call (synchronises): the
 * magnitude plane of (stream, ch), plane index 0 = the oldest carried sample; after a call of n steps
 * indices [0, AGC_EXTRA) hold the carry for the next call.  iq may be NULL; it is only filled for
 * channels that need raw I/Q. Used by the stage-1 parity tests (channel_t.wavein / iq_in). */
int mi_demod_read_planes(mi_demod* h, int stream, int ch, int first, int count, float* mag, float* iq);

/* Timing of the kernels of the last mi_demod_process_device() call on its stream, from HIP events
 * recorded around each launch (ms; synchronises). */
int mi_demod_last_kernel_ms(mi_demod* h, float* channelize_ms, float* demod_ms);

/* Per-kernel timing of the last call, from HIP events recorded on the launch streams around the launches.
 * index 0 is the channelize kernel; then "k_demod" (serial path) or the kernels of the time-parallel path
 * ("k_tp_full", "k_tp_core", "k_tp_seg", "k_tp_scan#0", "k_tp_fix#0", "k_tp_rest" = the remaining small launches).
 * The time-parallel path runs a long call in chunks, so a kernel is launched *launches times; *ms_total is the
 * sum over those launches (kernels of different chunks and calls overlap on several streams, so the sums exceed the wall time).
 * Returns MI_ERR_INVALID past the last index: iterate from 0 until it fails.  *name is a static string. */
int mi_demod_kernel_time(mi_demod* h, int index, const char** name, float* ms_total, int* launches);
/* The same for an earlier call: age 1 = the call before the last one, age 2 .. 5 = the ones before that (while the event set has
 * not been reused: six sets cycle; serial calls reuse the set of the call before them).  Lets a caller that keeps calls in flight
 * read the timings of call k after it has enqueued calls k+1 .. k+4, without draining the pipeline (reading them earlier makes the
 * host wait for the end of call k's tail and the front of the following calls start late: DESIGN.md section 6). */
int mi_demod_kernel_time_prev(mi_demod* h, int age, int index, const char** name, float* ms_total, int* launches);
/* (diagnostic) Where a launch of a time-parallel call sat in time: milliseconds from the start of the core chain of the call
 * `ref_age` calls back to event `event` of chunk `chunk` of the call `age` calls back (events per chunk: 0 / 1 stage 1 begin / end,
 * 11 / 2 k_tp_full begin / end, 3 / 4 core chain begin / end, 5 / 12 segment pass begin / end, 10 / 7 scan begin / end, 8 fix end,
 * 9 finish end).  tools/call_timeline.py prints the table. */
int mi_demod_event_ms(mi_demod* h, int ref_age, int age, int chunk, int event, float* ms);

/* Options.  MI_OPT_EARLY_INPUT (default 0): the caller guarantees that the IQ bytes handed to
 * mi_demod_process_device() are valid when the call is made (not merely in the order of `hip_stream`), e.g. a capture
 * already resident in HBM or a ring filled by a copy engine the caller has synchronised with.  The library may then read
 * them before the work queued earlier on `hip_stream` has finished, which lets stage 1 and the serial core chain of a call
 * overlap the segment / fix passes of the previous call (time-parallel path), or stage 1 overlap the channel loop of the
 * previous call (serial path; a second set of planes is allocated on first use).  Outputs still complete in stream order.
 * The audio buffer of a call must then also be free when the call is made (no reader of an earlier result still pending on
 * another stream): when it is not the buffer of the previous call, the segment passes may write it early.
 * MI_OPT_STEADY_BLOCKS (default 1): the serial stage 2 takes runs
 * of steps during which the squelch stays CLOSED or OPEN 64 at a time; 0 = every step in the sample loop.  Results are
 * bit-identical either way (audio, flags, statistics, checkpoint state); the switch exists for measurements and tests. */
enum {
    MI_OPT_EARLY_INPUT = 1,
    MI_OPT_STEADY_BLOCKS = 2,
    /* Tuning switches, all result-neutral (every combination is bit-identical; they exist for measurements and tests).
     * A handle takes its defaults from the caller's environment when it is created (MI_AIRBAND_TP, _PRUNE, _L64, _L64_JIT, _CORE_SPLIT, _CONV=lut|arith,
     * _STEADY, _UNI_ROWS, _TP_CHUNKS, _TP_RATIO, _TP_LPW); the library itself keeps no process-wide state. */
    MI_OPT_TIME_PARALLEL = 3, /* -1 auto (plain AM plans of up to 256 rows, calls of >= 8 batches), 0 serial kernel, 1 whenever eligible */
    MI_OPT_PRUNE_FFT = 4,     /* 1 (default): at N = 512 evaluate only the FFT nodes the picked bins need */
    MI_OPT_U8_CONVERSION = 5, /* -1 auto, 0 level table in LDS, 1 arithmetic (checked against the table by the plan) */
    MI_OPT_UNI_ROWS = 6,      /* rows (stream x channel) up to which the serial kernel keeps one channel per wave (4096) */
    MI_OPT_TP_CHUNKS = 7,     /* chunks a time-parallel call is cut into, 0 = default */
    MI_OPT_TP_RATIO_PCT = 8,  /* growth of consecutive chunks in percent (150 = 1.5x), 0 = default */
    MI_OPT_TP_SEG_LANES = 9,  /* lanes per wave of the segment pass, 0 = auto */
    MI_OPT_LANE_FFT = 10,     /* 1 (default): at N = 512 the first six FFT stages stay in the lanes (l64_kernel.h) where the plan allows */
    MI_OPT_CORE_SPLIT = 12,   /* 1 (default): the exact squelch core chain of the time-parallel path runs on three waves per channel -- one walks
                               * the noise-floor recurrence, one verifies, snapshots and steps, one fetches (tp.hip, k_tp_core2); 0: one wave */
    MI_OPT_SPEC_HEAD = 13,    /* 1 (default): when consecutive calls overlap (MI_OPT_EARLY_INPUT, alternating audio buffers) the first segments
                               * of a call warm up on the previous call's samples from a guessed state, as all others do, instead of waiting
                               * for the state that call's tail leaves; the scan checks them against it afterwards */
    MI_OPT_PRE_WAVE = 14,     /* serial stage 2 with one channel per wave: further waves of the channel's workgroup walk the squelch's pre-filter
                               * averages and noise floor over the call ahead of the channel's own wave (demod.hip, k_demod_pw: the full_ wave
                               * and the pre-filter wave).  -1 (default): up to 256 rows (streams x channels: one channel per CU) four waves per
                               * channel, up to 1 024 rows two (k_demod_pw2: the channel with its audio, and one wave for the pre-filter pair),
                               * beyond that none; 0 never, 1 four waves, 2 two waves */
    MI_OPT_RESERVE_CUS = 15,  /* time-parallel path: the wide passes of a call (stage 1, aggregates, segment pass) keep off this many CUs, which stay
                               * free for the core chains and the latency-bound tail kernels of the neighbouring calls (they need few waves but
                               * most of a SIMD's registers each, and otherwise wait for a wide wave to retire).  -1 (default): 32 on handles of
                               * up to 64 rows, 0 beyond; 0 never.  Takes effect only when the stream handed to mi_demod_process_device is not
                               * the NULL stream (the restricted streams are blocking ones: hipExtStreamCreateWithCUMask); set before the first
                               * time-parallel call.  The host-buffer entries use the handle's own stream and always qualify. */
    MI_OPT_AUDIO_WAVE = 16,   /* 1 (default): where MI_OPT_PRE_WAVE applies, an NFM channel gets a third wave that takes everything behind the
                               * filtered I/Q -- discriminator, DC block, de-emphasis, the CTCSS detector banks, output gate, notch filter,
                               * axcindicate and the audio / raw-I/Q stores -- from the channel's own wave (demod.hip, audio_wave) */
    MI_OPT_MIXED_PLAN = 17,   /* 1 (default): a plan that holds plain AM channels beside others (NFM, CTCSS, notch, low-pass, raw I/Q) sends the
                               * plain AM rows down the time-parallel path and the rest through the serial kernel, side by side in the same
                               * call (by itself in calls of >= 64 batches -- the time-parallel half has a fixed latency --, in any call under
                               * MI_OPT_TIME_PARALLEL = 1); 0: such a plan takes the serial kernel for every row */
    MI_OPT_SPLIT_CUS = 18,    /* Calls that take the serial kernel and overlap (MI_OPT_EARLY_INPUT): n > 0: k_demod runs alone on the last n CUs and
                               * stage 1 of the next call on the others (two CU-restricted streams, as under MI_OPT_RESERVE_CUS: only for calls whose
                               * stream is not the NULL stream); 0: the two share every CU; -1 (default): 128 for plans of 449 .. 512 rows -- two waves
                               * per row then fill 128 CUs two to a SIMD and stage 1 of as many streams is as long as the serial kernel: 64 streams x 8
                               * AM channels 344 -> 378 GS/s; with fewer rows the call is the serial kernel's latency either way.
                               * Set before the handle's first such call. */
    MI_OPT_LANE_FFT_JIT = 11  /* 1 (default): that kernel is compiled for the plan's own FFT nodes by hipRTC on first use (the code object is
                               * cached per (device, hop, masks) for the life of the process); 0, or hipRTC missing: the prebuilt full graph */
};
int mi_demod_set_option(mi_demod* h, int option, int value);

/* ---- host-only views of the derived plan (no GPU needed; used by the CPU test-suite) ---- */
typedef struct mi_plan mi_plan;
typedef struct mi_channel_derived {
    uint32_t bin;       /* dev->bins[i], config.cpp:669-670 */
    uint32_t dm_dphi;   /* channel_t.dm_dphi, config.cpp:682-713 */
    int32_t needs_raw_iq, has_iq_outputs, modulation;
    int32_t using_manual_level;
    float manual_signal_level, normal_signal_ratio, flappy_signal_ratio;
    float ampfactor, alpha;
    int32_t notch_enabled;
    float notch_d[3];
    int32_t lowpass_enabled;
    float lowpass_gain, lowpass_ycoeffs[2];
    int32_t ctcss_enabled, ctcss_fast_window, ctcss_slow_window, ctcss_fast_ndet, ctcss_slow_ndet;
} mi_channel_derived;

int mi_plan_create(const mi_device_cfg* dev, const mi_channel_cfg* chans, int nch, mi_plan** out);
void mi_plan_destroy(mi_plan* p);
int mi_plan_fft_size(const mi_plan* p);
int mi_plan_window(const mi_plan* p, float* out /* fft_size */);
int mi_plan_twiddles(const mi_plan* p, float* out /* fft_size/2 x {re,im} */);
int mi_plan_levels(const mi_plan* p, float* out /* 256, the LUT of the device's sample format */);
int mi_plan_sincos_lut(const mi_plan* p, float* sin_out /* 257 */, float* cos_out /* 257 */);
int mi_plan_channel(const mi_plan* p, int ch, mi_channel_derived* out);
int mi_plan_ctcss_coeffs(const mi_plan* p, int ch, int slow, float* out /* ndet */);

/* ---- synthetic IQ: integer-only, counter-based, identical on host and device ---- */
typedef struct mi_iqgen_carrier {
    int32_t offset_hz;   /* carrier offset from centre */
    int32_t kind;        /* 0 AM (1 kHz tone, 50 % depth); 1 NFM (1 kHz tone, 2.5 kHz dev); 2 NFM + 100 Hz CTCSS */
    int32_t amp_q8;      /* amplitude in 1/256 LSB (12 LSB = 3072) */
    int32_t gate_phase;  /* carrier is on while ((n / gate_samples) + gate_phase) is odd; gate_samples 0 = always on */
} mi_iqgen_carrier;

typedef struct mi_iqgen_cfg {
    int32_t sample_rate;
    uint64_t seed;
    int32_t noise_q8_mul;  /* noise scale: 111 gives sigma = 2.0 LSB */
    uint64_t gate_samples; /* on/off period in samples (sample_rate = 1 s) */
    int32_t ncarriers;
    mi_iqgen_carrier carriers[64];
} mi_iqgen_cfg;

/* u8 interleaved IQ for samples [first, first+count) of stream `stream_id` */
int mi_iqgen_host(const mi_iqgen_cfg* cfg, uint32_t stream_id, uint64_t first, uint64_t count, uint8_t* out);
int mi_iqgen_device(const mi_iqgen_cfg* cfg, uint32_t first_stream_id, uint32_t nstreams, size_t stream_stride_bytes, uint64_t first,
                    uint64_t count, void* d_out, void* hip_stream);

/* ---------------- mixer (src/mixer.cpp:56-98 connect, :114-140 put/mix, :157-261 thread) ----------------
 * One mixer_t: per WAVE_BATCH the output is zeroed, then every input that had a signal in that batch
 * (channel->axcindicate != NO_SIGNAL, output.cpp:564) is added as `sum[s] += in[s] * (ampfactor * ampl)` -- and
 * `* (ampfactor * ampr)` into the right channel when the mixer is stereo (some input has balance != 0,
 * mixer.cpp:82-83) -- in input order; an input whose multiplier is 0.0f is skipped (mixer.cpp:133-136); the mixer's
 * axcindicate is SIGNAL iff some input had one.  The reference mixes inputs in arrival order under jitter; this is the
 * jitter-free order (input index).  Inputs are rows of the audio the demod entry points write
 * ([row][row_stride] floats with row = stream * nch + channel, flags [row][axc_stride]), so on a multi-GPU job the
 * mixer runs on rank 0 over the gathered audio. */
typedef struct {
    int row;         /* stream * nch + channel of the audio buffer handed to mi_mixer_process_device */
    float ampfactor; /* mixer output's `ampfactor` (config.cpp:181), default 1 */
    float balance;   /* -1 .. 1 (config.cpp:182-186), default 0 */
} mi_mix_input;
typedef struct mi_mixer mi_mixer;

int mi_mixer_create(const mi_mix_input* inputs, int ninputs, int gpu, mi_mixer** out);
void mi_mixer_destroy(mi_mixer* m);
int mi_mixer_is_stereo(const mi_mixer* m);
/* d_waveout/d_axc: device buffers as written by mi_demod_process_device (nbatches batches); d_left / d_right:
 * [nbatches * WAVE_BATCH] (d_right may be NULL for a mono mixer, must not be for a stereo one); d_axc_out: [nbatches].
 * Enqueued on `hip_stream`, asynchronous. */
int mi_mixer_process_device(mi_mixer* m, const float* d_waveout, size_t row_stride, const char* d_axc, size_t axc_stride, int nbatches,
                            float* d_left, float* d_right, char* d_axc_out, void* hip_stream);

/* ---------------- multi-GPU: gather of audio + flags to rank 0 ----------------
 * One process per GPU; device streams are independent (one demod thread per device in the reference,
 * rtl_airband.cpp:1044-1078) and are partitioned stream-major over the ranks; nothing crosses GPUs except this gather, which
 * brings every rank's decimated audio and axcindicate flags to rank 0, where the reference's output / mixer threads run
 * (output.cpp:899-961).  RCCL point-to-point on the gather's own stream (librccl.so.1 is loaded on first use).
 *   mi_gather_unique_id   rank 0 makes the 128-byte id; the host program hands it to every rank (its launcher's channel)
 *   mi_gather_create      streams_per_rank[world]; max_batches bounds nbatches of a step
 *   mi_gather_audio       one step, every rank: d_waveout [streams_local][nch][nbatches*WAVE_BATCH], d_axc [streams_local][nch][nbatches]
 *                         as mi_demod_process_device wrote them on `hip_stream`; on rank 0 the job-wide arrays
 *                         d_all_waveout [streams_total][nch][nbatches*WAVE_BATCH], d_all_axc [streams_total][nch][nbatches] (ignored elsewhere).
 *                         Asynchronous: it starts when `hip_stream` reaches this point and runs on the gather's stream beside
 *                         whatever the caller enqueues next; do not overwrite its inputs / read its outputs before
 *                         mi_gather_stream_wait (makes a stream wait for it) or mi_gather_sync (the host waits).
 *                         open_only != 0: only the (row, batch) blocks whose flag is not MI_NO_SIGNAL travel -- what the reference's
 *                         non-continuous outputs consume (output.cpp:518,568) -- and rank 0 gets zeros for the others; this mode
 *                         synchronises the gather's stream with the host once per step (it needs the block counts). */
typedef struct { char internal[128]; } mi_gather_id;
typedef struct mi_gather mi_gather;
int mi_gather_unique_id(mi_gather_id* id);
/* TEST TRANSPORT: an id that makes mi_gather_create connect the ranks of job number `job` through an in-process loopback (one host
 * thread per rank, any GPU -- all on the same one is fine) instead of RCCL: device-to-device copies ordered by events, same
 * send / recv / group call sequence.  It lets a single-GPU machine execute every world > 1 branch of mi_gather_audio. */
int mi_gather_loopback_id(mi_gather_id* id, uint64_t job);
int mi_gather_create(const mi_gather_id* id, int rank, int world, int gpu, const int* streams_per_rank, int nch, int max_batches, mi_gather** out);
void mi_gather_destroy(mi_gather* g);
int mi_gather_audio(mi_gather* g, const float* d_waveout, const char* d_axc, int nbatches, int open_only, float* d_all_waveout, char* d_all_axc,
                    void* hip_stream);
int mi_gather_stream_wait(mi_gather* g, void* hip_stream);
int mi_gather_sync(mi_gather* g);

#ifdef __cplusplus
}
#endif
#endif /* MI_AIRBAND_H */
