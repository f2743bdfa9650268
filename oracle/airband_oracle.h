/*
 * airband_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, own words) of Boondock-Airband's demodulate() hot
 * path.  It is the checker the HIP path is compared against; nothing in the
 * shipped product (boondock-airband_amd/, include/) may include, link or call
 * it.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * Parity status (see DESIGN.md "Oracle"):
 *   - Squelch / CTCSS / NotchFilter / LowpassFilter restatements are PINNED:
 *     checked bit-for-bit against the reference's own squelch.cpp, ctcss.cpp,
 *     filters.cpp compiled unmodified into oracle/_ref (tests/test_oracle_vs_ref.py)
 *     and against committed fixtures generated from that build (tests/golden/).
 *   - The demodulate() glue (convert x window, bin pick, AM AGC, NFM
 *     discriminator, derotation, carry-over) is restated from
 *     src/rtl_airband.cpp:308-694 which cannot be compiled here (needs fftw3.h,
 *     lame, shout, libconfig++, generated config.h): PARITY UNPINNED by the
 *     reference for those lines; the reference's tests hold no vectors for them.
 *   - The FFT is FFTW3f in the reference (third-party, absent, unpinned version):
 *     restated as the DFT definition, implemented as a radix-2 DIT graph
 *     (ao_fft_forward); checked against a float64 numpy FFT.
 *
 * All file:line citations are relative to /root/reference/.
 */
#ifndef AIRBAND_ORACLE_H
#define AIRBAND_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/boondock_airband.h:64-75 (NFM build, the CMake default src/CMakeLists.txt:114) */
#define AO_WAVE_RATE 16000
#define AO_WAVE_BATCH 2000
#define AO_AGC_EXTRA 100
#define AO_WAVE_LEN (2 * AO_WAVE_BATCH + AO_AGC_EXTRA) /* 4100 */
#define AO_IQ_LEN (2 * 2 * AO_WAVE_BATCH + AO_AGC_EXTRA) /* 8100: unparenthesised macro, boondock_airband.h:247 */

enum { AO_MOD_AM = 0, AO_MOD_NFM = 1 };                                  /* boondock_airband.h:202-208 */
enum { AO_SFMT_U8 = 1, AO_SFMT_S8 = 2, AO_SFMT_S16 = 3, AO_SFMT_F32 = 4 }; /* input-common.h:31 */
enum { AO_SQ_CLOSED = 0, AO_SQ_OPENING, AO_SQ_CLOSING, AO_SQ_LOW_SIGNAL_ABORT, AO_SQ_OPEN }; /* squelch.h:104-110 */

#define AO_MAX_TONES 52

/* ---- CTCSS (src/ctcss.h, src/ctcss.cpp) ---- */
typedef struct {
    int enabled;
    float ctcss_freq;
    int window_size;
    uint64_t found_count, not_found_count;
    int ndet;
    float det_freq[AO_MAX_TONES];
    float det_coeff[AO_MAX_TONES];
    float det_mag[AO_MAX_TONES];
    int det_count[AO_MAX_TONES];
    float det_q1[AO_MAX_TONES], det_q2[AO_MAX_TONES];
    int enough_samples;
    int sample_count;
    int has_tone;
} ao_ctcss;

void ao_ctcss_init_disabled(ao_ctcss* c);
void ao_ctcss_init(ao_ctcss* c, float ctcss_freq, float sample_rate, int window_size);
void ao_ctcss_process(ao_ctcss* c, float sample);
void ao_ctcss_reset(ao_ctcss* c);
int ao_ctcss_has_tone(const ao_ctcss* c);

/* ---- Squelch (src/squelch.h, src/squelch.cpp) ---- */
typedef struct {
    float noise_floor;
    int using_manual_level;
    float manual_signal_level;
    float normal_signal_ratio, flappy_signal_ratio;
    float moving_avg_cap;
    float pre_full, pre_capped, post_full, post_capped;
    float squelch_level_cache;
    int using_post_filter;
    float pre_vs_post_factor;
    int open_delay, close_delay, low_signal_abort;
    int next_state, current_state;
    int delay;
    uint64_t open_count, sample_count, flappy_count;
    int low_signal_count;
    uint64_t recent_sample_size, flap_opens_threshold, recent_open_count, closed_sample_count;
    int buffer_size, buffer_head, buffer_tail;
    float buffer[102];
    ao_ctcss ctcss_fast, ctcss_slow;
} ao_squelch;

void ao_squelch_init(ao_squelch* s);
void ao_squelch_set_level_threshold(ao_squelch* s, float level);
void ao_squelch_set_snr_threshold(ao_squelch* s, float db);
void ao_squelch_set_ctcss(ao_squelch* s, float ctcss_freq, float sample_rate);
void ao_squelch_process_raw(ao_squelch* s, float sample);
void ao_squelch_process_filtered(ao_squelch* s, float sample);
void ao_squelch_process_audio(ao_squelch* s, float sample);
int ao_squelch_is_open(const ao_squelch* s);
int ao_squelch_should_filter(ao_squelch* s);
int ao_squelch_should_process_audio(const ao_squelch* s);
int ao_squelch_first_open_sample(const ao_squelch* s);
int ao_squelch_last_open_sample(const ao_squelch* s);
int ao_squelch_signal_outside_filter(ao_squelch* s);
float ao_squelch_level(ao_squelch* s); /* lazily cached like Squelch::squelch_level() */

/* ---- filters (src/filters.h, src/filters.cpp) ---- */
typedef struct {
    int enabled;
    float e, p, d[3], x[3], y[3];
} ao_notch;
typedef struct {
    int enabled;
    float ycoeffs[3], gain;
    float xv_re[3], xv_im[3], yv_re[3], yv_im[3];
} ao_lowpass;

void ao_notch_init_disabled(ao_notch* f);
void ao_notch_init(ao_notch* f, float notch_freq, float sample_freq, float q);
void ao_notch_apply(ao_notch* f, float* value);
void ao_lowpass_init_disabled(ao_lowpass* f);
void ao_lowpass_init(ao_lowpass* f, float freq, float sample_freq);
void ao_lowpass_apply(ao_lowpass* f, float* r, float* j);

/* ---- helpers from util.cpp / rtl_airband.cpp / config.cpp ---- */
void ao_sincos_lut_init(float* sin_lut257, float* cos_lut257);                               /* util.cpp:103-110 */
void ao_sincos_lut(const float* sin_lut, const float* cos_lut, uint32_t phi, float* s, float* c); /* util.cpp:113-127 */
float ao_dbfs_to_level(float dbfs, size_t fft_size);                                          /* util.cpp:169-176 */
void ao_window(float* w, size_t fft_size);                                                    /* rtl_airband.cpp:357-373 */
void ao_levels_u8(float* l256);                                                               /* rtl_airband.cpp:341-343 */
void ao_levels_s8(float* l256);                                                               /* rtl_airband.cpp:344-346 */
size_t ao_bin_for_freq(int freq, int centerfreq, int sample_rate, size_t fft_size);           /* config.cpp:669-670 */
uint32_t ao_dm_dphi(int freq, int centerfreq, int sample_rate);                               /* config.cpp:682-713 */
float ao_alpha_for_tau(int tau_us);                                                           /* rtl_airband.cpp:87, config.cpp:651,777-781 */
float ao_fast_atan2(float y, float x);                                                        /* rtl_airband.cpp:147-166 */
float ao_polar_disc_fast(float ar, float aj, float br, float bj);                             /* rtl_airband.cpp:168-172 */
float ao_fm_quadri_demod(float ar, float aj, float br, float bj);                             /* rtl_airband.cpp:174-176 */

/* ---- FFT: forward, unnormalised, e^{-j...} (what fftwf_plan_dft_1d(FFTW_FORWARD) computes,
 * rtl_airband.cpp:262-264,482).  Arithmetic is the documented radix-2 DIT graph. ---- */
typedef struct {
    int log2n;
    size_t n;
    float* tw_re; /* n/2 */
    float* tw_im;
    uint32_t* bitrev;
} ao_fft_plan;
int ao_fft_plan_init(ao_fft_plan* p, int log2n);
void ao_fft_plan_free(ao_fft_plan* p);
/* in/out: interleaved re,im; out-of-place */
void ao_fft_forward(const ao_fft_plan* p, const float* in, float* out);

/* ---- whole hot path ---- */
typedef struct {
    int sample_rate;  /* Hz */
    int centerfreq;   /* Hz */
    int fft_size_log; /* 8..13 */
    int sfmt;         /* AO_SFMT_* */
    float fullscale;  /* only s16/f32 */
    int tau;          /* device "tau" in us; <0 => global default 200us (rtl_airband.cpp:87) */
    int fm_quadri;    /* 0 = FM_FAST_ATAN2 (default), 1 = FM_QUADRI_DEMOD (-Q) */
} ao_device_cfg;

typedef struct {
    int freq;                   /* Hz */
    int modulation;             /* AO_MOD_* */
    int squelch_threshold_dbfs; /* 0 => not set / auto; <0 manual level (config.cpp:440-478) */
    int has_snr_threshold;      /* config key squelch_snr_threshold present */
    float squelch_snr_db;       /* -1 => keep default (config.cpp:479-518) */
    float notch_freq;           /* 0 => none */
    float notch_q;              /* 0 => default 10 */
    float ctcss_freq;           /* 0 => none */
    int bandwidth;              /* Hz; 0 => none (config.cpp:595-622) */
    float ampfactor;            /* default 1 */
    int tau;                    /* channel tau in us; <0 => inherit device alpha */
    int afc;                    /* 0..255 */
    int has_iq_outputs;         /* a rawfile output exists (config.cpp:162) */
} ao_channel_cfg;

typedef struct {
    /* channel_t (boondock_airband.h:243-270) */
    float wavein[AO_WAVE_LEN];
    float waveout[AO_WAVE_LEN];
    float iq_in[AO_IQ_LEN];
    float iq_out[AO_IQ_LEN];
    float pr, pj, prev_waveout, alpha;
    uint32_t dm_dphi, dm_phi;
    char axcindicate;
    unsigned char afc;
    int needs_raw_iq, has_iq_outputs;
    /* freq_t (boondock_airband.h:232-242) */
    int frequency;
    float agcavgfast, ampfactor;
    ao_squelch squelch;
    uint64_t active_counter;
    ao_notch notch;
    ao_lowpass lowpass;
    int modulation;
    size_t bin, base_bin;
} ao_channel;

typedef struct {
    ao_device_cfg cfg;
    size_t fft_size;
    size_t hop_bytes; /* "bps", rtl_airband.cpp:416 */
    int bytes_per_sample;
    int nch;
    ao_channel* ch;
    int waveend;
    ao_fft_plan plan;
    float* window;
    float levels_u8[256], levels_s8[256];
    float sin_lut[257], cos_lut[257];
    float *fftin, *fftout;
    /* optional per-sample trace of the stage-2 loop, channel-major, filled by ao_demod_push_window
     * when non-NULL: bit0 is_open, bit1 should_process_audio, bit2 should_filter, bits 4..6 state */
    uint8_t* trace;
    size_t trace_cap, trace_len; /* per channel */
} ao_demod;

ao_demod* ao_demod_create(const ao_device_cfg* dc, const ao_channel_cfg* cc, int nch);
void ao_demod_destroy(ao_demod* d);

/* One iteration of the reference's hot loop body for one device (rtl_airband.cpp:424-689):
 * convert+window the fft_size samples at `win`, FFT, bin pick, waveend++, and when
 * waveend >= WAVE_BATCH+AGC_EXTRA run the per-channel sample loop.  Returns 1 when a batch
 * completed (channel->waveout[0..WAVE_BATCH) etc. are then ready exactly as the output thread
 * would see them), else 0. */
int ao_demod_push_window(ao_demod* d, const unsigned char* win);

/* The output thread's per-batch duty on the contract (output.cpp:945-950):
 * memcpy(waveout, waveout + WAVE_BATCH, AGC_EXTRA*4). */
void ao_demod_output_carry(ao_demod* d);

/* Convenience driver: run over a linear capture, honouring the availability rule of
 * rtl_airband.cpp:417 (a window is processed only while >= hop_bytes + 2*bytes_per_sample*fft_size
 * bytes remain), playing the output thread after each batch.  Outputs are channel-major:
 * waveout[ch][b*WAVE_BATCH + i], iq_out[ch][2*(b*WAVE_BATCH+i)+{0,1}] (may be NULL), axc[ch][b].
 * Returns the number of complete batches (<= max_batches). */
int ao_demod_run(ao_demod* d, const unsigned char* iq, size_t nbytes, int max_batches, float* waveout, float* iq_out,
                 char* axc);

/* AFC::check<STEP> (rtl_airband.cpp:193-219): walk from `base` in direction step (-1 / +1) over the squared magnitudes of
 * the interleaved spectrum while they keep rising fast enough; returns the bin reached. */
size_t ao_afc_check(const float* fft_results, size_t fft_size, int step, size_t base, float base_value, unsigned char afc);

/* Stage-1 only (convert x window -> FFT -> bins): mag[ch][nfft], iq[ch][2*nfft] (iq may be NULL). */
void ao_stage1(ao_demod* d, const unsigned char* iq, size_t nfft, float* mag, float* iqout);

/* ---- mixer (src/mixer.cpp) ---- */
typedef struct {
    int row; /* row of the audio buffer: stream * nch + channel */
    float ampfactor, balance;
} ao_mix_input;
void ao_mix_waveforms(float* sum, const float* in, float mult, int size);
/* One mixer over `nbatches` batches of audio [row][row_stride] with per-batch indicators [row][axc_stride]; inputs are
 * mixed in index order.  left/right: [nbatches*WAVE_BATCH] (right may be NULL for a mono mixer); returns 1 if stereo. */
int ao_mixer_run(const ao_mix_input* inputs, int ninputs, const float* waveout, size_t row_stride, const char* axc, size_t axc_stride,
                 int nbatches, float* left, float* right, char* axc_out);

#ifdef __cplusplus
}
#endif
#endif
