/*
 * oracle_drivers.c -- TEST INFRASTRUCTURE ONLY.
 * Array drivers over the oracle's component restatements, with exactly the call protocol of
 * oracle/ref_driver.cpp so the two can be compared sample for sample.
 */
#include <math.h>
#include <string.h>

#include "airband_oracle.h"

typedef struct {
    float manual_level;
    int has_snr;
    float snr_db;
    float ctcss_freq;
    float ctcss_rate;
} ao_squelch_cfg;

typedef struct {
    uint64_t open_count, flappy_count, ctcss_count, no_ctcss_count;
    float noise_level, signal_level, squelch_level;
} ao_squelch_final;

void ao_squelch_run(const ao_squelch_cfg* cfg, const float* raw, const float* filt, const float* audio, size_t n, uint8_t* flags,
                    float* level, float* noise, float* signal, ao_squelch_final* fin) {
    ao_squelch sq;
    ao_squelch_init(&sq);
    if (cfg->manual_level > 0)
        ao_squelch_set_level_threshold(&sq, cfg->manual_level);
    if (cfg->has_snr)
        ao_squelch_set_snr_threshold(&sq, cfg->snr_db);
    if (cfg->ctcss_freq > 0)
        ao_squelch_set_ctcss(&sq, cfg->ctcss_freq, cfg->ctcss_rate);
    for (size_t i = 0; i < n; i++) {
        ao_squelch_process_raw(&sq, raw[i]);
        int f = ao_squelch_should_filter(&sq);
        if (f && filt)
            ao_squelch_process_filtered(&sq, filt[i]);
        int fo = ao_squelch_first_open_sample(&sq);
        int lo = ao_squelch_last_open_sample(&sq);
        float lvl = ao_squelch_level(&sq);
        int pa = ao_squelch_should_process_audio(&sq);
        if (pa && audio)
            ao_squelch_process_audio(&sq, audio[i]);
        int op = ao_squelch_is_open(&sq);
        int so = ao_squelch_signal_outside_filter(&sq);
        if (flags)
            flags[i] = (uint8_t)((op ? 1 : 0) | (pa ? 2 : 0) | (f ? 4 : 0) | (fo ? 8 : 0) | (lo ? 16 : 0) | (so ? 32 : 0));
        if (level)
            level[i] = lvl;
        if (noise)
            noise[i] = sq.noise_floor;
        if (signal)
            signal[i] = sq.pre_full;
    }
    if (fin) {
        fin->open_count = sq.open_count;
        fin->flappy_count = sq.flappy_count;
        fin->ctcss_count = sq.ctcss_slow.found_count;
        fin->no_ctcss_count = sq.ctcss_slow.not_found_count;
        fin->noise_level = sq.noise_floor;
        fin->signal_level = sq.pre_full;
        fin->squelch_level = ao_squelch_level(&sq);
    }
}

void ao_ctcss_run(float freq, float rate, int window, const float* x, size_t n, uint8_t* flags, uint64_t* found, uint64_t* not_found) {
    ao_ctcss c;
    ao_ctcss_init(&c, freq, rate, window);
    for (size_t i = 0; i < n; i++) {
        ao_ctcss_process(&c, x[i]);
        if (flags)
            flags[i] = (uint8_t)((ao_ctcss_has_tone(&c) ? 1 : 0) | (c.enough_samples ? 2 : 0));
    }
    *found = c.found_count;
    *not_found = c.not_found_count;
}

void ao_notch_run(float freq, float rate, float q, const float* x, size_t n, float* y) {
    ao_notch f;
    ao_notch_init(&f, freq, rate, q);
    for (size_t i = 0; i < n; i++) {
        float v = x[i];
        ao_notch_apply(&f, &v);
        y[i] = v;
    }
}

void ao_lowpass_run(float freq, float rate, const float* re, const float* im, size_t n, float* ore, float* oim) {
    ao_lowpass f;
    ao_lowpass_init(&f, freq, rate);
    for (size_t i = 0; i < n; i++) {
        float r = re[i], j = im[i];
        ao_lowpass_apply(&f, &r, &j);
        ore[i] = r;
        oim[i] = j;
    }
}

/* Tone::get_sample, generate_signal.cpp:32-35 (tones only) */
void ao_tone_run(int sample_rate, float freq, float ampl, size_t n, float* out) {
    size_t sample_count = 0;
    for (size_t i = 0; i < n; i++) {
        sample_count++;
        float value = 0.0f;
        value += (float)(ampl * sin(2 * M_PI * sample_count * freq / sample_rate));
        out[i] = value;
    }
}

int ao_ctcss_detector_count(float freq, float rate, int window) {
    ao_ctcss c;
    ao_ctcss_init(&c, freq, rate, window);
    return c.ndet;
}

size_t ao_sizeof_demod_channel(void) {
    return sizeof(ao_channel);
}

/* ---- views of derived parameters, laid out like mi_channel_derived (include/mi_airband.h) so the
 * product's host-side plan can be compared field by field ---- */
typedef struct {
    uint32_t bin, dm_dphi;
    int32_t needs_raw_iq, has_iq_outputs, modulation, using_manual_level;
    float manual_signal_level, normal_signal_ratio, flappy_signal_ratio, ampfactor, alpha;
    int32_t notch_enabled;
    float notch_d[3];
    int32_t lowpass_enabled;
    float lowpass_gain, lowpass_ycoeffs[2];
    int32_t ctcss_enabled, ctcss_fast_window, ctcss_slow_window, ctcss_fast_ndet, ctcss_slow_ndet;
} ao_channel_derived;

void ao_demod_channel_derived(const ao_demod* d, int i, ao_channel_derived* o) {
    const ao_channel* c = &d->ch[i];
    memset(o, 0, sizeof(*o));
    o->bin = (uint32_t)c->bin;
    o->dm_dphi = c->dm_dphi;
    o->needs_raw_iq = c->needs_raw_iq;
    o->has_iq_outputs = c->has_iq_outputs;
    o->modulation = c->modulation;
    o->using_manual_level = c->squelch.using_manual_level;
    o->manual_signal_level = c->squelch.manual_signal_level;
    o->normal_signal_ratio = c->squelch.normal_signal_ratio;
    o->flappy_signal_ratio = c->squelch.flappy_signal_ratio;
    o->ampfactor = c->ampfactor;
    o->alpha = c->alpha;
    o->notch_enabled = c->notch.enabled;
    memcpy(o->notch_d, c->notch.d, sizeof(o->notch_d));
    o->lowpass_enabled = c->lowpass.enabled;
    o->lowpass_gain = c->lowpass.gain;
    o->lowpass_ycoeffs[0] = c->lowpass.ycoeffs[0];
    o->lowpass_ycoeffs[1] = c->lowpass.ycoeffs[1];
    o->ctcss_enabled = c->squelch.ctcss_slow.enabled;
    o->ctcss_fast_window = c->squelch.ctcss_fast.window_size;
    o->ctcss_slow_window = c->squelch.ctcss_slow.window_size;
    o->ctcss_fast_ndet = c->squelch.ctcss_fast.ndet;
    o->ctcss_slow_ndet = c->squelch.ctcss_slow.ndet;
}

void ao_demod_ctcss_coeffs(const ao_demod* d, int i, int slow, float* out) {
    const ao_ctcss* c = slow ? &d->ch[i].squelch.ctcss_slow : &d->ch[i].squelch.ctcss_fast;
    memcpy(out, c->det_coeff, sizeof(float) * (size_t)c->ndet);
}

void ao_demod_tables(const ao_demod* d, float* window, float* tw_re_im, float* levels, float* sin_lut, float* cos_lut) {
    memcpy(window, d->window, sizeof(float) * d->fft_size);
    for (size_t k = 0; k < d->fft_size / 2; k++) {
        tw_re_im[2 * k] = d->plan.tw_re[k];
        tw_re_im[2 * k + 1] = d->plan.tw_im[k];
    }
    memcpy(levels, d->cfg.sfmt == AO_SFMT_S8 ? d->levels_s8 : d->levels_u8, sizeof(float) * 256);
    memcpy(sin_lut, d->sin_lut, sizeof(float) * 257);
    memcpy(cos_lut, d->cos_lut, sizeof(float) * 257);
}

/* forward FFT of one interleaved complex vector (for the DFT-definition check) */
void ao_fft_run(int log2n, const float* in, float* out) {
    ao_fft_plan p;
    if (ao_fft_plan_init(&p, log2n) != 0)
        return;
    ao_fft_forward(&p, in, out);
    ao_fft_plan_free(&p);
}

/* squelch state snapshot after feeding raw samples only (for the reference's behavioural unit tests) */
typedef struct {
    float noise_level, squelch_level;
    int32_t is_open, should_process_audio;
    uint64_t open_count, ctcss_count, no_ctcss_count;
} ao_squelch_probe;

/* exact (noise_floor_, moving_avg_cap_, pre_filter_.capped_, pre_filter_.full_) BEFORE raw sample k*stride,
 * k = 0 .. n/stride, for checking the time-parallel core chain of the HIP path (tp.hip) */
void ao_squelch_core_trace(const ao_squelch_cfg* cfg, const float* raw, size_t n, size_t stride, float* out4) {
    ao_squelch sq;
    ao_squelch_init(&sq);
    if (cfg->manual_level > 0)
        ao_squelch_set_level_threshold(&sq, cfg->manual_level);
    if (cfg->has_snr)
        ao_squelch_set_snr_threshold(&sq, cfg->snr_db);
    for (size_t i = 0; i <= n; i++) {
        if (i % stride == 0 || i == n) {
            size_t k = (i == n && i % stride != 0) ? i / stride + 1 : i / stride;
            out4[4 * k + 0] = sq.noise_floor;
            out4[4 * k + 1] = sq.moving_avg_cap;
            out4[4 * k + 2] = sq.pre_capped;
            out4[4 * k + 3] = sq.pre_full;
        }
        if (i < n)
            ao_squelch_process_raw(&sq, raw[i]);
    }
}

/* dev->bins[] as AFC left them (rtl_airband.cpp:224-249) */
void ao_demod_bins(const ao_demod* d, int32_t* bins, int32_t* base_bins) {
    for (int i = 0; i < d->nch; i++) {
        bins[i] = (int32_t)d->ch[i].bin;
        base_bins[i] = (int32_t)d->ch[i].base_bin;
    }
}
