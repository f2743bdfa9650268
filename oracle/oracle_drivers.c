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
