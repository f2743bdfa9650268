/*
 * ref_driver.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * Thin extern "C" driver (own code) over the REFERENCE's own classes, compiled from the sources
 * where they lie under /root/reference/src (squelch.cpp ctcss.cpp filters.cpp logging.cpp
 * generate_signal.cpp -- the same link set as the reference's unit-test binary,
 * src/CMakeLists.txt:378-391, minus gtest).  Output: oracle/_ref/libairband_ref.so (git-ignored).
 * It exists to pin oracle/airband_oracle.c against the real reference and to generate the
 * fixtures under tests/golden/.  No reference source is copied into this repository.
 *
 * rtl_airband.cpp / util.cpp / config.cpp / input-*.cpp are NOT buildable here (they need fftw3.h,
 * lame, shout, libconfig++ and the CMake-generated config.h) and are deliberately not part of this.
 */
#include <cstddef>
#include <cstdint>
#include <cstring>

#include "ctcss.h"
#include "filters.h"
#include "generate_signal.h"
#include "logging.h"
#include "squelch.h"

extern "C" {

struct ref_squelch_cfg {
    float manual_level;  /* >0: set_squelch_level_threshold(level) first */
    int has_snr;         /* then set_squelch_snr_threshold(snr_db) */
    float snr_db;
    float ctcss_freq;    /* >0: set_ctcss_freq(freq, ctcss_rate) */
    float ctcss_rate;
};

struct ref_squelch_final {
    uint64_t open_count, flappy_count, ctcss_count, no_ctcss_count;
    float noise_level, signal_level, squelch_level;
};

/* Per sample, in the order the demod loop uses them (rtl_airband.cpp:529-612):
 *   process_raw_sample(raw[i]);
 *   f = should_filter_sample(); if (f && filt) process_filtered_sample(filt[i]);
 *   fo = first_open_sample(); lo = last_open_sample(); lvl = squelch_level();
 *   pa = should_process_audio(); if (pa && audio) process_audio_sample(audio[i]);
 *   op = is_open();
 * flags[i] = op | pa<<1 | f<<2 | fo<<3 | lo<<4 | signal_outside_filter<<5 */
void ref_squelch_run(const ref_squelch_cfg* cfg, const float* raw, const float* filt, const float* audio, size_t n, uint8_t* flags,
                     float* level, float* noise, float* signal, ref_squelch_final* fin) {
    log_destination = NONE;
    Squelch sq;
    if (cfg->manual_level > 0)
        sq.set_squelch_level_threshold(cfg->manual_level);
    if (cfg->has_snr)
        sq.set_squelch_snr_threshold(cfg->snr_db);
    if (cfg->ctcss_freq > 0)
        sq.set_ctcss_freq(cfg->ctcss_freq, cfg->ctcss_rate);
    for (size_t i = 0; i < n; i++) {
        sq.process_raw_sample(raw[i]);
        bool f = sq.should_filter_sample();
        if (f && filt)
            sq.process_filtered_sample(filt[i]);
        bool fo = sq.first_open_sample();
        bool lo = sq.last_open_sample();
        float lvl = sq.squelch_level();
        bool pa = sq.should_process_audio();
        if (pa && audio)
            sq.process_audio_sample(audio[i]);
        bool op = sq.is_open();
        bool so = sq.signal_outside_filter();
        if (flags)
            flags[i] = (uint8_t)((op ? 1 : 0) | (pa ? 2 : 0) | (f ? 4 : 0) | (fo ? 8 : 0) | (lo ? 16 : 0) | (so ? 32 : 0));
        if (level)
            level[i] = lvl;
        if (noise)
            noise[i] = sq.noise_level();
        if (signal)
            signal[i] = sq.signal_level();
    }
    if (fin) {
        fin->open_count = sq.open_count();
        fin->flappy_count = sq.flappy_count();
        fin->ctcss_count = sq.ctcss_count();
        fin->no_ctcss_count = sq.no_ctcss_count();
        fin->noise_level = sq.noise_level();
        fin->signal_level = sq.signal_level();
        fin->squelch_level = sq.squelch_level();
    }
}

/* flags[i] = has_tone | enough_samples<<1 after process_audio_sample(x[i]) */
void ref_ctcss_run(float freq, float rate, int window, const float* x, size_t n, uint8_t* flags, uint64_t* found, uint64_t* not_found) {
    log_destination = NONE;
    CTCSS c(freq, rate, window);
    for (size_t i = 0; i < n; i++) {
        c.process_audio_sample(x[i]);
        if (flags)
            flags[i] = (uint8_t)((c.has_tone() ? 1 : 0) | (c.enough_samples() ? 2 : 0));
    }
    *found = c.found_count();
    *not_found = c.not_found_count();
}

void ref_notch_run(float freq, float rate, float q, const float* x, size_t n, float* y) {
    log_destination = NONE;
    NotchFilter f(freq, rate, q);
    for (size_t i = 0; i < n; i++) {
        float v = x[i];
        f.apply(v);
        y[i] = v;
    }
}

void ref_lowpass_run(float freq, float rate, const float* re, const float* im, size_t n, float* ore, float* oim) {
    log_destination = NONE;
    LowpassFilter f(freq, rate);
    for (size_t i = 0; i < n; i++) {
        float r = re[i], j = im[i];
        f.apply(r, j);
        ore[i] = r;
        oim[i] = j;
    }
}

int ref_filters_default_disabled(void) { /* what src/test_filters.cpp:33-41 asserts */
    NotchFilter n;
    LowpassFilter l;
    return (!n.enabled() && !l.enabled()) ? 1 : 0;
}

/* GenerateSignal with tones only (noise is random_device-seeded, generate_signal.cpp:41-46) */
void ref_tone_run(int sample_rate, float freq, float ampl, size_t n, float* out) {
    GenerateSignal g(sample_rate);
    g.add_tone(freq, ampl);
    for (size_t i = 0; i < n; i++)
        out[i] = g.get_sample();
}

int ref_standard_tones(float* out, int cap) {
    int n = (int)CTCSS::standard_tones.size();
    for (int i = 0; i < n && i < cap; i++)
        out[i] = CTCSS::standard_tones[i];
    return n;
}
}
