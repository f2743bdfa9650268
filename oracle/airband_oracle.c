/*
 * airband_oracle.c -- TEST INFRASTRUCTURE ONLY (see airband_oracle.h for the parity status).
 *
 * Plain-C restatement of the reference hot path.  Build: -O2 -fno-fast-math -ffp-contract=off
 * (the reference itself is built -ffast-math, src/CMakeLists.txt:18-21, so "the reference" is
 * compiler dependent; the oracle pins strict IEEE single precision evaluated left to right).
 * Citations: /root/reference/src/<file>:<line>.
 */
#define _GNU_SOURCE
#include "airband_oracle.h"

#include <complex.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* std::min(a,b) returns b<a ? b : a (libstdc++); keep that operand order */
static inline float std_minf(float a, float b) {
    return (b < a) ? b : a;
}

/* =========================== CTCSS (src/ctcss.cpp) =========================== */

/* ctcss.cpp:101-103 */
static const float ao_standard_tones[51] = {
    67.0f,  69.3f,  71.9f,  74.4f,  77.0f,  79.7f,  82.5f,  85.4f,  88.5f,  91.5f,  94.8f,  97.4f,  100.0f,
    103.5f, 107.2f, 110.9f, 114.8f, 118.8f, 123.0f, 127.3f, 131.8f, 136.5f, 141.3f, 146.2f, 150.0f, 151.4f,
    156.7f, 159.8f, 162.2f, 165.5f, 167.9f, 171.3f, 173.8f, 177.3f, 179.9f, 183.5f, 186.2f, 189.9f, 192.8f,
    196.6f, 199.5f, 203.5f, 206.5f, 210.7f, 218.1f, 225.7f, 229.1f, 233.6f, 241.8f, 250.3f, 254.1f};

/* ToneDetector ctor, ctcss.cpp:31-42.  omega is float, cos() on a float argument resolves to the
 * float overload in C++ (cosf), then 2.0*cosf is rounded to float. */
static float tone_coeff(float tone_freq, float sample_rate, int window_size) {
    int k = (int)(0.5 + (float)window_size * tone_freq / sample_rate);
    float omega = (float)((2.0 * M_PI * k) / window_size);
    return (float)(2.0 * cosf(omega));
}

/* ToneDetectorSet::add, ctcss.cpp:61-73: a tone whose coefficient equals an existing one is dropped */
static void ctcss_add(ao_ctcss* c, float tone_freq, float sample_rate) {
    float coeff = tone_coeff(tone_freq, sample_rate, c->window_size);
    for (int i = 0; i < c->ndet; i++)
        if (coeff == c->det_coeff[i])
            return;
    int i = c->ndet++;
    c->det_freq[i] = tone_freq;
    c->det_coeff[i] = coeff;
    c->det_mag[i] = 0.0f;
    c->det_count[i] = 0;
    c->det_q1[i] = c->det_q2[i] = 0.0f;
}

void ao_ctcss_init_disabled(ao_ctcss* c) { /* ctcss.h:74 */
    memset(c, 0, sizeof(*c));
}

void ao_ctcss_init(ao_ctcss* c, float ctcss_freq, float sample_rate, int window_size) { /* ctcss.cpp:105-122 */
    memset(c, 0, sizeof(*c));
    c->enabled = 1;
    c->ctcss_freq = ctcss_freq;
    c->window_size = window_size;
    ctcss_add(c, ctcss_freq, sample_rate);
    for (int t = 0; t < 51; t++) {
        if (fabsf(ctcss_freq - ao_standard_tones[t]) < 5)
            continue;
        ctcss_add(c, ao_standard_tones[t], sample_rate);
    }
    ao_ctcss_reset(c);
}

void ao_ctcss_reset(ao_ctcss* c) { /* ctcss.cpp:165-172, 57-59 */
    if (!c->enabled)
        return;
    for (int i = 0; i < c->ndet; i++) {
        c->det_count[i] = 0;
        c->det_q1[i] = c->det_q2[i] = 0.0f;
    }
    c->enough_samples = 0;
    c->sample_count = 0;
    c->has_tone = 0;
}

int ao_ctcss_has_tone(const ao_ctcss* c) { /* ctcss.h:84 */
    return !c->enabled || c->has_tone;
}

void ao_ctcss_process(ao_ctcss* c, float sample) { /* ctcss.cpp:124-163 */
    if (!c->enabled)
        return;
    for (int i = 0; i < c->ndet; i++) { /* ToneDetector::process_sample, ctcss.cpp:44-55 */
        float q0 = c->det_coeff[i] * c->det_q1[i] - c->det_q2[i] + sample;
        c->det_q2[i] = c->det_q1[i];
        c->det_q1[i] = q0;
        c->det_count[i]++;
        if (c->det_count[i] == c->window_size) {
            float q1 = c->det_q1[i], q2 = c->det_q2[i];
            c->det_mag[i] = q1 * q1 + q2 * q2 - q1 * q2 * c->det_coeff[i];
            c->det_count[i] = 0;
        }
    }
    c->sample_count++;
    if (c->sample_count < c->window_size)
        return;
    c->enough_samples = 1;
    /* sorted_powers (ctcss.cpp:87-99): only the largest power and the index-order mean are used */
    float total = 0.0f, maxp = c->det_mag[0];
    for (int i = 0; i < c->ndet; i++) {
        total += c->det_mag[i];
        if (c->det_mag[i] > maxp)
            maxp = c->det_mag[i];
    }
    float avg = total / (float)c->ndet; /* float / size_t -> float division */
    float target = c->det_mag[0];       /* the target tone is always detector 0 and its freq is unique */
    if (target == maxp && target > avg) {
        c->has_tone = 1;
        c->found_count++;
    } else {
        c->has_tone = 0;
        c->not_found_count++;
    }
    for (int i = 0; i < c->ndet; i++) { /* powers_.reset() */
        c->det_count[i] = 0;
        c->det_q1[i] = c->det_q2[i] = 0.0f;
    }
    c->sample_count = 0;
}

/* =========================== Squelch (src/squelch.cpp) =========================== */

static void sq_calc_cap(ao_squelch* s) { /* squelch.cpp:492-499 */
    if (s->using_manual_level)
        s->moving_avg_cap = 1.5f * s->manual_signal_level;
    else
        s->moving_avg_cap = 1.5f * s->normal_signal_ratio * s->noise_floor;
}

void ao_squelch_set_snr_threshold(ao_squelch* s, float db) { /* squelch.cpp:98-108 */
    s->using_manual_level = 0;
    s->normal_signal_ratio = (float)pow(10.0, db / 20.0);
    s->flappy_signal_ratio = s->normal_signal_ratio * 0.9f;
    sq_calc_cap(s);
}

void ao_squelch_set_level_threshold(ao_squelch* s, float level) { /* squelch.cpp:84-96 */
    if (level > 0) {
        s->using_manual_level = 1;
        s->manual_signal_level = level;
    } else {
        s->using_manual_level = 0;
    }
    sq_calc_cap(s);
}

void ao_squelch_set_ctcss(ao_squelch* s, float ctcss_freq, float sample_rate) { /* squelch.cpp:110-116 */
    ao_ctcss_init(&s->ctcss_fast, ctcss_freq, sample_rate, (int)(sample_rate * 0.05));
    ao_ctcss_init(&s->ctcss_slow, ctcss_freq, sample_rate, (int)(sample_rate * 0.4));
}

void ao_squelch_init(ao_squelch* s) { /* squelch.cpp:36-82 */
    memset(s, 0, sizeof(*s));
    s->noise_floor = 5.0f;
    ao_squelch_set_snr_threshold(s, 9.54f);
    s->manual_signal_level = -1.0f;
    s->pre_full = s->pre_capped = 0.001f;
    s->post_full = s->post_capped = 0.001f;
    s->squelch_level_cache = 0.0f;
    s->using_post_filter = 0;
    s->pre_vs_post_factor = 0.9f;
    s->open_delay = 197;
    s->close_delay = 197;
    s->low_signal_abort = 88;
    s->next_state = AO_SQ_CLOSED;
    s->current_state = AO_SQ_CLOSED;
    s->delay = 0;
    s->open_count = 0;
    s->sample_count = (uint64_t)-1;
    s->flappy_count = 0;
    s->low_signal_count = 0;
    s->recent_sample_size = 1000;
    s->flap_opens_threshold = 3;
    s->recent_open_count = 0;
    s->closed_sample_count = 0;
    s->buffer_size = 102;
    s->buffer_head = 0;
    s->buffer_tail = 1;
    ao_ctcss_init_disabled(&s->ctcss_fast);
    ao_ctcss_init_disabled(&s->ctcss_slow);
}

static int sq_flapping(const ao_squelch* s) { /* squelch.cpp:516-518 */
    return s->recent_open_count >= s->flap_opens_threshold;
}

float ao_squelch_level(ao_squelch* s) { /* squelch.cpp:164-177 */
    if (s->using_manual_level)
        return s->manual_signal_level;
    if (s->squelch_level_cache == 0.0f) {
        if (sq_flapping(s) && s->flappy_signal_ratio < s->normal_signal_ratio)
            s->squelch_level_cache = s->flappy_signal_ratio * s->noise_floor;
        else
            s->squelch_level_cache = s->normal_signal_ratio * s->noise_floor;
    }
    return s->squelch_level_cache;
}

static int sq_has_pre(ao_squelch* s) { /* squelch.cpp:462-464 */
    return s->pre_capped >= ao_squelch_level(s);
}
static int sq_has_post(ao_squelch* s) { /* squelch.cpp:466-468 */
    return s->using_post_filter && s->post_capped >= s->buffer[s->buffer_tail];
}
static int sq_has_signal(ao_squelch* s) { /* squelch.cpp:470-475 */
    if (s->using_post_filter)
        return sq_has_pre(s) && sq_has_post(s);
    return sq_has_pre(s);
}

int ao_squelch_is_open(const ao_squelch* s) { /* squelch.cpp:118-134 */
    if (s->current_state == AO_SQ_OPEN || s->current_state == AO_SQ_CLOSING) {
        if (s->ctcss_slow.enabled) {
            if (s->ctcss_slow.enough_samples)
                return ao_ctcss_has_tone(&s->ctcss_slow);
            return ao_ctcss_has_tone(&s->ctcss_fast);
        }
        return 1;
    }
    return 0;
}
int ao_squelch_should_filter(ao_squelch* s) { /* squelch.cpp:136-138 */
    return ((sq_has_pre(s) || s->current_state != AO_SQ_CLOSED) && s->current_state != AO_SQ_LOW_SIGNAL_ABORT);
}
int ao_squelch_should_process_audio(const ao_squelch* s) { /* squelch.cpp:140-142 */
    return (s->current_state == AO_SQ_OPEN || s->current_state == AO_SQ_CLOSING);
}
int ao_squelch_first_open_sample(const ao_squelch* s) { /* squelch.cpp:144-146 */
    return (s->current_state != AO_SQ_OPEN && s->next_state == AO_SQ_OPEN);
}
int ao_squelch_last_open_sample(const ao_squelch* s) { /* squelch.cpp:148-150 */
    return (s->current_state == AO_SQ_CLOSING && s->next_state == AO_SQ_CLOSED) ||
           (s->current_state != AO_SQ_LOW_SIGNAL_ABORT && s->next_state == AO_SQ_LOW_SIGNAL_ABORT);
}
int ao_squelch_signal_outside_filter(ao_squelch* s) { /* squelch.cpp:152-154 */
    return (s->using_post_filter && sq_has_pre(s) && !sq_has_post(s));
}

static void sq_set_state(ao_squelch* s, int update) { /* squelch.cpp:297-361 */
    int cur = s->current_state;
    if (cur == AO_SQ_CLOSED && update == AO_SQ_CLOSING)
        update = AO_SQ_CLOSED;
    else if (cur == AO_SQ_CLOSED && update == AO_SQ_LOW_SIGNAL_ABORT)
        update = AO_SQ_CLOSED;
    else if (cur == AO_SQ_CLOSED && update == AO_SQ_OPEN)
        update = AO_SQ_OPENING;
    else if (cur == AO_SQ_OPENING && update == AO_SQ_LOW_SIGNAL_ABORT)
        update = AO_SQ_CLOSED;
    else if (cur == AO_SQ_LOW_SIGNAL_ABORT && update != AO_SQ_LOW_SIGNAL_ABORT && update != AO_SQ_CLOSED)
        update = AO_SQ_CLOSED;
    else if (cur == AO_SQ_OPEN && update == AO_SQ_CLOSED)
        update = AO_SQ_CLOSING;
    else if (cur == AO_SQ_OPEN && update == AO_SQ_OPENING)
        update = AO_SQ_OPEN;
    s->next_state = update;
}

static void sq_update_current_state(ao_squelch* s) { /* squelch.cpp:363-460 */
    if (s->next_state == AO_SQ_OPENING) {
        if (s->current_state != AO_SQ_OPENING) {
            s->delay = 0;
            s->low_signal_count = 0;
            s->using_post_filter = 0;
            s->current_state = s->next_state;
        } else {
            s->delay++;
            if (s->delay >= s->open_delay) {
                if (s->closed_sample_count < s->recent_sample_size) {
                    s->recent_open_count++;
                    if (sq_flapping(s))
                        s->flappy_count++;
                    s->squelch_level_cache = 0.0f;
                }
                if (sq_has_signal(s))
                    s->next_state = AO_SQ_OPEN;
                else
                    s->next_state = AO_SQ_CLOSED;
            }
        }
    } else if (s->next_state == AO_SQ_CLOSING) {
        if (s->current_state != AO_SQ_CLOSING) {
            s->delay = 0;
            s->current_state = s->next_state;
        } else {
            s->delay++;
            if (s->delay >= s->close_delay) {
                if (!sq_has_signal(s)) {
                    s->next_state = AO_SQ_CLOSED;
                } else {
                    s->current_state = AO_SQ_OPEN;
                    s->next_state = AO_SQ_OPEN;
                }
            }
        }
    } else if (s->next_state == AO_SQ_LOW_SIGNAL_ABORT) {
        if (s->current_state != AO_SQ_LOW_SIGNAL_ABORT) {
            if (s->current_state != AO_SQ_CLOSING)
                s->delay = 0;
            s->current_state = s->next_state;
        } else {
            s->delay++;
            if (s->delay >= s->close_delay)
                s->next_state = AO_SQ_CLOSED;
        }
    } else if (s->next_state == AO_SQ_OPEN && s->current_state != AO_SQ_OPEN) {
        s->open_count++;
        s->current_state = s->next_state;
    } else if (s->next_state == AO_SQ_CLOSED && s->current_state != AO_SQ_CLOSED) {
        s->using_post_filter = 0;
        s->closed_sample_count = 0;
        s->current_state = s->next_state;
        ao_ctcss_reset(&s->ctcss_fast);
        ao_ctcss_reset(&s->ctcss_slow);
    } else if (s->next_state == AO_SQ_CLOSED && s->current_state == AO_SQ_CLOSED) {
        if (s->closed_sample_count < s->recent_sample_size) {
            s->closed_sample_count++;
        } else if (s->closed_sample_count == s->recent_sample_size) {
            s->recent_open_count = 0;
            s->squelch_level_cache = 0.0f;
        }
    } else {
        s->current_state = s->next_state;
    }
    s->buffer_tail = (s->buffer_tail + 1) % s->buffer_size;
    s->buffer_head = (s->buffer_head + 1) % s->buffer_size;
}

static void sq_calc_noise_floor(ao_squelch* s) { /* squelch.cpp:477-490 */
    const float decay_factor = 0.97f;
    const float new_factor = (float)(1.0 - (double)0.97f);
    s->noise_floor = s->noise_floor * decay_factor + std_minf(s->pre_capped, s->noise_floor) * new_factor + 1e-6f;
    sq_calc_cap(s);
    s->squelch_level_cache = 0.0f;
}

static void sq_update_avg(ao_squelch* s, float* full, float* capped, float sample) { /* squelch.cpp:501-514 */
    const float decay_factor = 0.99f;
    const float new_factor = (float)(1.0 - (double)0.99f);
    *full = *full * decay_factor + sample * new_factor;
    if (*capped >= s->moving_avg_cap && sample >= s->moving_avg_cap)
        *capped = s->moving_avg_cap;
    else
        *capped = std_minf(s->moving_avg_cap, *capped * decay_factor + sample * new_factor);
}

void ao_squelch_process_raw(ao_squelch* s, float sample) { /* squelch.cpp:195-246 */
    sq_update_current_state(s);
    s->sample_count++;
    if (s->sample_count % 16 == 0)
        sq_calc_noise_floor(s);
    sq_update_avg(s, &s->pre_full, &s->pre_capped, sample);
    s->buffer[s->buffer_head] = s->pre_capped * s->pre_vs_post_factor;
    if (s->current_state == AO_SQ_OPEN && !sq_has_signal(s))
        sq_set_state(s, AO_SQ_CLOSING);
    if (s->current_state == AO_SQ_CLOSED && sq_has_signal(s))
        sq_set_state(s, AO_SQ_OPENING);
    if (s->current_state != AO_SQ_CLOSED && s->current_state != AO_SQ_LOW_SIGNAL_ABORT) {
        if (sample >= ao_squelch_level(s)) {
            s->low_signal_count = 0;
        } else {
            s->low_signal_count++;
            if (s->low_signal_count >= s->low_signal_abort)
                sq_set_state(s, AO_SQ_LOW_SIGNAL_ABORT);
        }
    }
}

void ao_squelch_process_filtered(ao_squelch* s, float sample) { /* squelch.cpp:248-276 */
    if (!ao_squelch_should_filter(s))
        return;
    if (s->current_state == AO_SQ_OPENING) {
        if (s->delay < s->buffer_size)
            return;
        if (s->delay == s->buffer_size) {
            s->post_full = s->buffer[s->buffer_tail];
            s->post_capped = s->buffer[s->buffer_tail];
        }
    }
    s->using_post_filter = 1;
    sq_update_avg(s, &s->post_full, &s->post_capped, sample);
    if (s->post_capped < s->buffer[s->buffer_tail])
        sq_set_state(s, AO_SQ_CLOSED);
}

void ao_squelch_process_audio(ao_squelch* s, float sample) { /* squelch.cpp:278-295 */
    if (!s->ctcss_slow.enabled)
        return;
    if (s->current_state != AO_SQ_CLOSED) {
        ao_ctcss_process(&s->ctcss_slow, sample);
        if (!s->ctcss_slow.enough_samples)
            ao_ctcss_process(&s->ctcss_fast, sample);
    }
}

/* =========================== filters (src/filters.cpp) =========================== */

void ao_notch_init_disabled(ao_notch* f) {
    memset(f, 0, sizeof(*f));
}

void ao_notch_init(ao_notch* f, float notch_freq, float sample_freq, float q) { /* filters.cpp:30-48 */
    memset(f, 0, sizeof(*f));
    f->enabled = 1;
    if (notch_freq <= 0.0) {
        f->enabled = 0;
        return;
    }
    /* float arguments select the float overloads of tan/cos in the C++ reference */
    float wo = (float)(2 * M_PI * (notch_freq / sample_freq));
    f->e = 1 / (1 + tanf(wo / (q * 2)));
    f->p = cosf(wo);
    f->d[0] = f->e;
    f->d[1] = 2 * f->e * f->p;
    f->d[2] = (2 * f->e - 1);
}

void ao_notch_apply(ao_notch* f, float* value) { /* filters.cpp:50-64 */
    if (!f->enabled)
        return;
    f->x[0] = f->x[1];
    f->x[1] = f->x[2];
    f->x[2] = *value;
    f->y[0] = f->y[1];
    f->y[1] = f->y[2];
    f->y[2] = f->d[0] * f->x[2] - f->d[1] * f->x[1] + f->d[0] * f->x[0] + f->d[1] * f->y[1] - f->d[2] * f->y[0];
    *value = f->y[2];
}

void ao_lowpass_init_disabled(ao_lowpass* f) {
    memset(f, 0, sizeof(*f));
}

/* filters.cpp:101-144: blt / expand / multin / eval in complex<double> */
static double complex lp_blt(double complex pz) {
    return (2.0 + pz) / (2.0 - pz);
}
static void lp_multin(double complex w, int npz, double complex* coeffs) {
    double complex nw = -w;
    for (int i = npz; i >= 1; i--)
        coeffs[i] = (nw * coeffs[i]) + coeffs[i - 1];
    coeffs[0] = nw * coeffs[0];
}
static void lp_expand(double complex* pz, int npz, double complex* coeffs) {
    coeffs[0] = 1.0;
    for (int i = 0; i < npz; i++)
        coeffs[i + 1] = 0.0;
    for (int i = 0; i < npz; i++)
        lp_multin(pz[i], npz, coeffs);
}
static double complex lp_eval(double complex* coeffs, int npz, double complex z) {
    double complex sum = 0.0;
    for (int i = npz; i >= 0; i--)
        sum = (sum * z) + coeffs[i];
    return sum;
}

void ao_lowpass_init(ao_lowpass* f, float freq, float sample_freq) { /* filters.cpp:69-99 */
    memset(f, 0, sizeof(*f));
    f->enabled = 1;
    if (freq <= 0.0) {
        f->enabled = 0;
        return;
    }
    double raw_alpha = (double)freq / sample_freq;
    double warped_alpha = tan(M_PI * raw_alpha) / M_PI;
    double complex zeros[2] = {-1.0, -1.0};
    double complex poles[2];
    double complex bessel = CMPLX(-1.10160133059e+00, 6.36009824757e-01);
    double scale = M_PI * 2 * warped_alpha;
    poles[0] = lp_blt(CMPLX(scale * creal(bessel), scale * cimag(bessel)));
    poles[1] = lp_blt(CMPLX(scale * creal(bessel), scale * -cimag(bessel)));
    double complex topcoeffs[3], botcoeffs[3];
    lp_expand(zeros, 2, topcoeffs);
    lp_expand(poles, 2, botcoeffs);
    double complex g = lp_eval(topcoeffs, 2, 1.0) / lp_eval(botcoeffs, 2, 1.0);
    f->gain = (float)hypot(cimag(g), creal(g));
    for (int i = 0; i <= 2; i++)
        f->ycoeffs[i] = (float)(-(creal(botcoeffs[i]) / creal(botcoeffs[2])));
}

void ao_lowpass_apply(ao_lowpass* f, float* r, float* j) { /* filters.cpp:146-163; complex<float> ops are per component */
    if (!f->enabled)
        return;
    f->xv_re[0] = f->xv_re[1];
    f->xv_im[0] = f->xv_im[1];
    f->xv_re[1] = f->xv_re[2];
    f->xv_im[1] = f->xv_im[2];
    f->xv_re[2] = *r / f->gain;
    f->xv_im[2] = *j / f->gain;
    f->yv_re[0] = f->yv_re[1];
    f->yv_im[0] = f->yv_im[1];
    f->yv_re[1] = f->yv_re[2];
    f->yv_im[1] = f->yv_im[2];
    f->yv_re[2] = (f->xv_re[0] + f->xv_re[2]) + (2.0f * f->xv_re[1]) + (f->ycoeffs[0] * f->yv_re[0]) + (f->ycoeffs[1] * f->yv_re[1]);
    f->yv_im[2] = (f->xv_im[0] + f->xv_im[2]) + (2.0f * f->xv_im[1]) + (f->ycoeffs[0] * f->yv_im[0]) + (f->ycoeffs[1] * f->yv_im[1]);
    *r = f->yv_re[2];
    *j = f->yv_im[2];
}

/* =========================== helpers =========================== */

void ao_sincos_lut_init(float* sin_lut, float* cos_lut) { /* util.cpp:105-110 */
    for (uint32_t i = 0; i < 256; i++)
        sincosf((float)(2.0F * M_PI * (float)i / 256.0f), sin_lut + i, cos_lut + i);
    sin_lut[256] = sin_lut[0];
    cos_lut[256] = cos_lut[0];
}

void ao_sincos_lut(const float* sin_lut, const float* cos_lut, uint32_t phi, float* sine, float* cosine) { /* util.cpp:113-127 */
    uint32_t idx = phi >> 16;
    float fract = (float)(phi & 0xffff) / 65536.0f;
    float v1 = sin_lut[idx], v2 = sin_lut[idx + 1];
    *sine = v1 + (v2 - v1) * fract;
    v1 = cos_lut[idx];
    v2 = cos_lut[idx + 1];
    *cosine = v1 + (v2 - v1) * fract;
}

float ao_dbfs_to_level(float dbfs, size_t fft_size) { /* util.cpp:169-176 */
    float offset = 7.54f + 10.0f * log10f((float)(fft_size / 2)) - 2.38f;
    return (float)(pow(10.0, (dbfs - offset) / 20.0f) * (double)fft_size);
}

void ao_window(float* w, size_t fft_size) { /* rtl_airband.cpp:357-373 */
    const double a0 = 0.27105140069342f, a1 = 0.43329793923448f, a2 = 0.21812299954311f, a3 = 0.06592544638803f;
    const double a4 = 0.01081174209837f, a5 = 0.00077658482522f, a6 = 0.00001388721735f;
    for (size_t i = 0; i < fft_size; i++) {
        double x = a0 - (a1 * cos((2.0 * M_PI * i) / (fft_size - 1))) + (a2 * cos((4.0 * M_PI * i) / (fft_size - 1))) -
                   (a3 * cos((6.0 * M_PI * i) / (fft_size - 1))) + (a4 * cos((8.0 * M_PI * i) / (fft_size - 1))) -
                   (a5 * cos((10.0 * M_PI * i) / (fft_size - 1))) + (a6 * cos((12.0 * M_PI * i) / (fft_size - 1)));
        w[i] = (float)x;
    }
}

void ao_levels_u8(float* l) { /* rtl_airband.cpp:341-343 */
    for (int i = 0; i < 256; i++)
        l[i] = (i - 127.5f) / 127.5f;
}
void ao_levels_s8(float* l) { /* rtl_airband.cpp:344-346; index 128 (= -128) is never written there: stays 0 */
    memset(l, 0, 256 * sizeof(float));
    for (int16_t i = -127; i < 128; i++)
        l[(uint8_t)i] = i / 128.0f;
}

size_t ao_bin_for_freq(int freq, int centerfreq, int sample_rate, size_t fft_size) { /* config.cpp:669-670 */
    /* int + int - int, divided by (double)(int / size_t): note the INTEGER quotient sample_rate/fft_size */
    return (size_t)ceil((freq + sample_rate - centerfreq) / (double)((size_t)sample_rate / fft_size) - 1.0) % fft_size;
}

uint32_t ao_dm_dphi(int freq, int centerfreq, int sample_rate) { /* config.cpp:682-713 */
    double dm_dphi = (double)(freq - centerfreq);
    double decimation_factor = ((double)sample_rate / (double)AO_WAVE_RATE);
    double corr = (double)AO_WAVE_RATE / 2.0;
    corr *= (decimation_factor - round(decimation_factor));
    corr *= (double)(freq - centerfreq) / ((double)sample_rate / 2.0);
    dm_dphi -= corr;
    dm_dphi /= (double)AO_WAVE_RATE;
    dm_dphi -= trunc(dm_dphi);
    dm_dphi *= 256.0 * 65536.0;
    return (uint32_t)((int)dm_dphi);
}

float ao_alpha_for_tau(int tau_us) {
    if (tau_us < 0) /* global default, rtl_airband.cpp:87 */
        return (float)exp(-1.0f / (AO_WAVE_RATE * 2e-4));
    /* config.cpp:651,778 */
    return (tau_us == 0 ? 0.0f : (float)exp(-1.0f / (AO_WAVE_RATE * 1e-6 * tau_us)));
}

float ao_fast_atan2(float y, float x) { /* rtl_airband.cpp:147-166 */
    float yabs, angle;
    float pi4 = (float)M_PI_4, pi34 = (float)(3 * M_PI_4);
    if (x == 0.0f && y == 0.0f)
        return 0;
    yabs = y;
    if (yabs < 0.0f)
        yabs = -yabs;
    if (x >= 0.0f)
        angle = pi4 - pi4 * (x - yabs) / (x + yabs);
    else
        angle = pi34 - pi4 * (x + yabs) / (yabs - x);
    if (y < 0.0f)
        return -angle;
    return angle;
}

float ao_polar_disc_fast(float ar, float aj, float br, float bj) { /* rtl_airband.cpp:141-144,168-172 */
    float nbj = -bj;
    float cr = ar * br - aj * nbj;
    float cj = aj * br + ar * nbj;
    return (float)(ao_fast_atan2(cj, cr) * M_1_PI);
}

float ao_fm_quadri_demod(float ar, float aj, float br, float bj) { /* rtl_airband.cpp:174-176 */
    return (float)((br * aj - ar * bj) / (ar * ar + aj * aj + 1.0f) * M_1_PI);
}

/* =========================== FFT =========================== */
/*
 * Forward DFT X[k] = sum_n x[n] e^{-2 pi j k n / N}, unnormalised.
 * The arithmetic is fixed so the HIP kernel can reproduce it bit for bit:
 *   - twiddle table tw[k] = (float)cos(2 pi k/N), (float)(-sin(2 pi k/N)), k < N/2, computed in double,
 *     with tw[0] = (1,0) and tw[N/4] = (0,-1) forced exact;
 *   - bit-reversal permutation, then stages s = 1..log2 N of radix-2 DIT butterflies
 *     (a, b) -> (a + t, a - t), t = w*b with w = tw[j * N/2^s];
 *   - t = b for j == 0, t = (b.im, -b.re) for w == -j, otherwise
 *       t.re = fmaf(-b.im, w.im, b.re*w.re),  t.im = fmaf(b.im, w.re, b.re*w.im).
 */
int ao_fft_plan_init(ao_fft_plan* p, int log2n) {
    size_t n = (size_t)1 << log2n;
    p->log2n = log2n;
    p->n = n;
    p->tw_re = (float*)malloc(sizeof(float) * n / 2);
    p->tw_im = (float*)malloc(sizeof(float) * n / 2);
    p->bitrev = (uint32_t*)malloc(sizeof(uint32_t) * n);
    if (!p->tw_re || !p->tw_im || !p->bitrev)
        return -1;
    for (size_t k = 0; k < n / 2; k++) {
        double a = 2.0 * M_PI * (double)k / (double)n;
        p->tw_re[k] = (float)cos(a);
        p->tw_im[k] = (float)(-sin(a));
    }
    p->tw_re[0] = 1.0f;
    p->tw_im[0] = 0.0f;
    p->tw_re[n / 4] = 0.0f;
    p->tw_im[n / 4] = -1.0f;
    for (size_t i = 0; i < n; i++) {
        uint32_t r = 0;
        for (int b = 0; b < log2n; b++)
            if (i & ((size_t)1 << b))
                r |= 1u << (log2n - 1 - b);
        p->bitrev[i] = r;
    }
    return 0;
}

void ao_fft_plan_free(ao_fft_plan* p) {
    free(p->tw_re);
    free(p->tw_im);
    free(p->bitrev);
    memset(p, 0, sizeof(*p));
}

void ao_fft_forward(const ao_fft_plan* p, const float* in, float* out) {
    const size_t n = p->n;
    for (size_t i = 0; i < n; i++) {
        out[2 * i] = in[2 * p->bitrev[i]];
        out[2 * i + 1] = in[2 * p->bitrev[i] + 1];
    }
    for (int s = 1; s <= p->log2n; s++) {
        const size_t half = (size_t)1 << (s - 1);
        const size_t step = n >> s;
        for (size_t base = 0; base < n; base += 2 * half) {
            for (size_t j = 0; j < half; j++) {
                float* a = out + 2 * (base + j);
                float* b = out + 2 * (base + j + half);
                float tr, ti;
                const size_t e = j * step;
                if (e == 0) {
                    tr = b[0];
                    ti = b[1];
                } else if (e == n / 4) {
                    tr = b[1];
                    ti = -b[0];
                } else {
                    const float wr = p->tw_re[e], wi = p->tw_im[e];
                    tr = fmaf(-b[1], wi, b[0] * wr);
                    ti = fmaf(b[1], wr, b[0] * wi);
                }
                const float ar = a[0], ai = a[1];
                a[0] = ar + tr;
                a[1] = ai + ti;
                b[0] = ar - tr;
                b[1] = ai - ti;
            }
        }
    }
}

/* =========================== whole hot path =========================== */

ao_demod* ao_demod_create(const ao_device_cfg* dc, const ao_channel_cfg* cc, int nch) {
    ao_demod* d = (ao_demod*)calloc(1, sizeof(ao_demod));
    if (!d)
        return NULL;
    d->cfg = *dc;
    d->fft_size = (size_t)1 << dc->fft_size_log;
    d->bytes_per_sample = (dc->sfmt == AO_SFMT_S16) ? 2 : (dc->sfmt == AO_SFMT_F32 ? 4 : 1);
    /* rtl_airband.cpp:416 */
    d->hop_bytes = 2 * d->bytes_per_sample * (size_t)round((double)dc->sample_rate / (double)AO_WAVE_RATE);
    d->nch = nch;
    d->ch = (ao_channel*)calloc((size_t)nch, sizeof(ao_channel));
    d->window = (float*)malloc(sizeof(float) * d->fft_size);
    d->fftin = (float*)malloc(sizeof(float) * 2 * d->fft_size);
    d->fftout = (float*)malloc(sizeof(float) * 2 * d->fft_size);
    if (!d->ch || !d->window || !d->fftin || !d->fftout || ao_fft_plan_init(&d->plan, dc->fft_size_log) != 0) {
        ao_demod_destroy(d);
        return NULL;
    }
    ao_window(d->window, d->fft_size);
    ao_levels_u8(d->levels_u8);
    ao_levels_s8(d->levels_s8);
    ao_sincos_lut_init(d->sin_lut, d->cos_lut);
    float dev_alpha = ao_alpha_for_tau(dc->tau); /* config.cpp:777-781 */
    for (int i = 0; i < nch; i++) {
        ao_channel* c = &d->ch[i];
        const ao_channel_cfg* k = &cc[i];
        /* config.cpp:319-334 */
        for (int j = 0; j < AO_AGC_EXTRA; j++) {
            c->wavein[j] = 20;
            c->waveout[j] = 0.5;
        }
        c->axcindicate = ' ';
        c->pr = c->pj = 0;
        c->prev_waveout = 0.5;
        c->alpha = dev_alpha;
        c->afc = (unsigned char)k->afc;
        /* mk_freqlist, config.cpp:271-287 */
        c->frequency = k->freq;
        c->agcavgfast = 0.5f;
        c->ampfactor = 1.0f;
        ao_squelch_init(&c->squelch);
        c->active_counter = 0;
        c->modulation = k->modulation;
        ao_notch_init_disabled(&c->notch);
        ao_lowpass_init_disabled(&c->lowpass);
        if (k->has_iq_outputs) /* config.cpp:162 */
            c->needs_raw_iq = c->has_iq_outputs = 1;
        /* squelch_threshold, config.cpp:440-478 (set before squelch_snr_threshold) */
        if (k->squelch_threshold_dbfs < 0)
            ao_squelch_set_level_threshold(&c->squelch, ao_dbfs_to_level((float)k->squelch_threshold_dbfs, d->fft_size));
        /* squelch_snr_threshold, config.cpp:479-518 */
        if (k->has_snr_threshold && k->squelch_snr_db != -1.0f)
            ao_squelch_set_snr_threshold(&c->squelch, k->squelch_snr_db);
        /* notch, config.cpp:519-567 */
        if (k->notch_freq > 0) {
            float q = (k->notch_q > 0) ? k->notch_q : 10.0f;
            ao_notch_init(&c->notch, k->notch_freq, AO_WAVE_RATE, q);
        }
        /* ctcss, config.cpp:568-594 */
        if (k->ctcss_freq > 0)
            ao_squelch_set_ctcss(&c->squelch, k->ctcss_freq, AO_WAVE_RATE);
        /* bandwidth, config.cpp:595-622.  needs_raw_iq is set as soon as the key exists (:596); only a
         * positive value builds the low-pass.  Config value 0 = key absent; a key that is present with the
         * value 0 (the reference then `continue`s past ampfactor / tau / outputs, :609-611) or with a negative
         * value ("invalid, ignoring") is passed as bandwidth < 0: raw-I/Q path without a filter. */
        if (k->bandwidth != 0)
            c->needs_raw_iq = 1;
        if (k->bandwidth > 0)
            ao_lowpass_init(&c->lowpass, (float)k->bandwidth / 2, AO_WAVE_RATE);
        if (k->ampfactor >= 0) /* config.cpp:623-647 */
            c->ampfactor = k->ampfactor;
        if (k->tau >= 0) /* config.cpp:649-653 */
            c->alpha = ao_alpha_for_tau(k->tau);
        c->base_bin = c->bin = ao_bin_for_freq(k->freq, dc->centerfreq, dc->sample_rate, d->fft_size);
        if (c->modulation == AO_MOD_NFM) /* config.cpp:672-679 */
            c->needs_raw_iq = 1;
        if (c->needs_raw_iq) {
            c->dm_dphi = ao_dm_dphi(k->freq, dc->centerfreq, dc->sample_rate);
            c->dm_phi = 0;
        }
    }
    d->waveend = 0; /* config.cpp:808 */
    return d;
}

void ao_demod_destroy(ao_demod* d) {
    if (!d)
        return;
    ao_fft_plan_free(&d->plan);
    free(d->ch);
    free(d->window);
    free(d->fftin);
    free(d->fftout);
    free(d->trace);
    free(d);
}

/* rtl_airband.cpp:424-477 */
static void convert_window(ao_demod* d, const unsigned char* win) {
    const size_t n = d->fft_size;
    float* fftin = d->fftin;
    const float* window = d->window;
    if (d->cfg.sfmt == AO_SFMT_S16) {
        float const scale = 1.0f / d->cfg.fullscale;
        const short* buf2 = (const short*)win;
        for (size_t i = 0; i < n; i++, buf2 += 2) {
            fftin[2 * i] = scale * (float)buf2[0] * window[i];
            fftin[2 * i + 1] = scale * (float)buf2[1] * window[i];
        }
    } else if (d->cfg.sfmt == AO_SFMT_F32) {
        float const scale = 1.0f / d->cfg.fullscale;
        const float* buf2 = (const float*)win;
        for (size_t i = 0; i < n; i++, buf2 += 2) {
            fftin[2 * i] = scale * buf2[0] * window[i];
            fftin[2 * i + 1] = scale * buf2[1] * window[i];
        }
    } else {
        const float* levels = (d->cfg.sfmt == AO_SFMT_U8 ? d->levels_u8 : d->levels_s8);
        const unsigned char* buf2 = win;
        for (size_t i = 0; i < n; i++, buf2 += 2) {
            fftin[2 * i] = levels[buf2[0]] * window[i];
            fftin[2 * i + 1] = levels[buf2[1]] * window[i];
        }
    }
}

/* the per-channel sample loop, rtl_airband.cpp:517-669 */
/* class AFC, rtl_airband.cpp:180-251.  square(): :186-192; check<STEP>: :193-219; finalize: :224-249. */
static float afc_square(const float* fft_results, size_t index) {
    return fft_results[2 * index] * fft_results[2 * index] + fft_results[2 * index + 1] * fft_results[2 * index + 1];
}

size_t ao_afc_check(const float* fft_results, size_t fft_size, int step, size_t base, float base_value, unsigned char afc) {
    float threshold = 0;
    size_t bin;
    for (bin = base;; bin += (size_t)step) {
        if (step < 0) {
            if (bin < (size_t)-step)
                break;
        } else if ((size_t)(bin + (size_t)step) >= fft_size)
            break;
        const float value = afc_square(fft_results, (size_t)(bin + (size_t)step));
        if (value <= base_value)
            break;
        if (base == bin) {
            threshold = (value - base_value) / (float)afc;
        } else {
            if ((value - base_value) < threshold)
                break;
            threshold += threshold / 10.0; /* double arithmetic, stored back to float */
        }
    }
    return bin;
}

static void afc_finalize(ao_demod* d, ao_channel* channel, char prev_axcindicate) {
    if (channel->afc == 0)
        return;
    const char axcindicate = channel->axcindicate;
    if (axcindicate != ' ' && prev_axcindicate == ' ') {
        const size_t base = channel->base_bin;
        const float base_value = afc_square(d->fftout, base);
        size_t bin = ao_afc_check(d->fftout, d->fft_size, -1, base, base_value, channel->afc);
        if (bin == base)
            bin = ao_afc_check(d->fftout, d->fft_size, 1, base, base_value, channel->afc);
        if (channel->bin != bin) {
            channel->bin = bin;
            if (bin > base)
                channel->axcindicate = '<'; /* AFC_UP, boondock_airband.h:101 */
            else if (bin < base)
                channel->axcindicate = '>'; /* AFC_DOWN */
        }
    } else if (axcindicate == ' ' && prev_axcindicate != ' ')
        channel->bin = channel->base_bin;
}

static void channel_batch(ao_demod* d, int ci) {
    ao_channel* channel = &d->ch[ci];
    ao_squelch* sq = &channel->squelch;
    const char prev_axcindicate = channel->axcindicate; /* AFC afc(dev, i), rtl_airband.cpp:518 */
    channel->axcindicate = ' ';
    for (int j = AO_AGC_EXTRA; j < AO_WAVE_BATCH + AO_AGC_EXTRA; j++) {
        float* real = &channel->iq_in[2 * (j - AO_AGC_EXTRA)];
        float* imag = &channel->iq_in[2 * (j - AO_AGC_EXTRA) + 1];

        ao_squelch_process_raw(sq, channel->wavein[j]);

        if (ao_squelch_should_filter(sq) && channel->needs_raw_iq) {
            float swf, cwf, re_tmp, im_tmp;
            ao_sincos_lut(d->sin_lut, d->cos_lut, channel->dm_phi, &swf, &cwf);
            /* multiply(real, imag, cwf, -swf), rtl_airband.cpp:141-144 */
            float nswf = -swf;
            re_tmp = *real * cwf - *imag * nswf;
            im_tmp = *imag * cwf + *real * nswf;
            channel->dm_phi += channel->dm_dphi;
            channel->dm_phi &= 0xffffff;
            ao_lowpass_apply(&channel->lowpass, &re_tmp, &im_tmp);
            *real = re_tmp;
            *imag = im_tmp;
            channel->wavein[j] = sqrtf(*real * *real + *imag * *imag);
            if (channel->lowpass.enabled)
                ao_squelch_process_filtered(sq, channel->wavein[j]);
        }

        if (channel->modulation == AO_MOD_AM) {
            if (ao_squelch_first_open_sample(sq)) {
                for (int k = j - AO_AGC_EXTRA; k < j; k++) {
                    if (channel->wavein[k] >= ao_squelch_level(sq))
                        channel->agcavgfast = channel->agcavgfast * 0.9f + channel->wavein[k] * 0.1f;
                }
            } else if (ao_squelch_last_open_sample(sq)) {
                for (int k = j - AO_AGC_EXTRA + 1; k < j; k++)
                    channel->waveout[k] = channel->waveout[k - 1] * 0.94f;
            }
        }

        float* waveout = &channel->waveout[j];

        if (ao_squelch_should_process_audio(sq)) {
            if (channel->modulation == AO_MOD_AM) {
                if (channel->wavein[j] > ao_squelch_level(sq))
                    channel->agcavgfast = channel->agcavgfast * 0.995f + channel->wavein[j] * 0.005f;
                *waveout = (channel->wavein[j - AO_AGC_EXTRA] - channel->agcavgfast) / (channel->agcavgfast * 1.5f);
                if (fabsf(*waveout) > 0.8f) {
                    *waveout *= 0.85f;
                    channel->agcavgfast *= 1.15f;
                }
            } else if (channel->modulation == AO_MOD_NFM) {
                if (!d->cfg.fm_quadri)
                    *waveout = ao_polar_disc_fast(*real, *imag, channel->pr, channel->pj);
                else
                    *waveout = ao_fm_quadri_demod(*real, *imag, channel->pr, channel->pj);
                channel->pr = *real;
                channel->pj = *imag;
                channel->agcavgfast = channel->agcavgfast * 0.995f + *waveout * 0.005f;
                *waveout -= channel->agcavgfast;
                *waveout = *waveout * (1.0f - channel->alpha) + channel->prev_waveout * channel->alpha;
                channel->prev_waveout = *waveout;
            }
            ao_squelch_process_audio(sq, *waveout);
        }

        if (ao_squelch_is_open(sq)) {
            ao_notch_apply(&channel->notch, waveout);
            *waveout *= channel->ampfactor;
            if (isnan(*waveout))
                *waveout = 0.0;
            else if (*waveout > 1.0)
                *waveout = 1.0;
            else if (*waveout < -1.0)
                *waveout = -1.0;
            channel->axcindicate = '*';
            if (channel->has_iq_outputs) {
                channel->iq_out[2 * (j - AO_AGC_EXTRA)] = *real;
                channel->iq_out[2 * (j - AO_AGC_EXTRA) + 1] = *imag;
            }
        } else {
            *waveout = 0;
            if (channel->has_iq_outputs) {
                channel->iq_out[2 * (j - AO_AGC_EXTRA)] = 0;
                channel->iq_out[2 * (j - AO_AGC_EXTRA) + 1] = 0;
            }
        }

        if (d->trace && d->trace_len < d->trace_cap) {
            uint8_t t = (uint8_t)((ao_squelch_is_open(sq) ? 1 : 0) | (ao_squelch_should_process_audio(sq) ? 2 : 0) |
                                  (sq->current_state << 4));
            d->trace[(size_t)ci * d->trace_cap + d->trace_len + (size_t)(j - AO_AGC_EXTRA)] = t;
        }
    }
    memmove(channel->wavein, channel->wavein + AO_WAVE_BATCH, (size_t)(d->waveend - AO_WAVE_BATCH) * sizeof(float));
    if (channel->needs_raw_iq)
        memmove(channel->iq_in, channel->iq_in + 2 * AO_WAVE_BATCH, (size_t)(d->waveend - AO_WAVE_BATCH) * sizeof(float) * 2);
    afc_finalize(d, channel, prev_axcindicate); /* rtl_airband.cpp:648-652 */
    if (channel->axcindicate != ' ')
        channel->active_counter++;
}

int ao_demod_push_window(ao_demod* d, const unsigned char* win) {
    convert_window(d, win);
    ao_fft_forward(&d->plan, d->fftin, d->fftout);
    const float* fftout = d->fftout;
    for (int j = 0; j < d->nch; j++) { /* rtl_airband.cpp:505-511 */
        ao_channel* c = &d->ch[j];
        const float re = fftout[2 * c->bin], im = fftout[2 * c->bin + 1];
        c->wavein[d->waveend] = sqrtf(re * re + im * im);
        if (c->needs_raw_iq) {
            c->iq_in[2 * d->waveend] = re;
            c->iq_in[2 * d->waveend + 1] = im;
        }
    }
    d->waveend += 1; /* FFT_BATCH == 1 */
    if (d->waveend >= AO_WAVE_BATCH + AO_AGC_EXTRA) {
        for (int i = 0; i < d->nch; i++)
            channel_batch(d, i);
        if (d->trace)
            d->trace_len += AO_WAVE_BATCH;
        d->waveend -= AO_WAVE_BATCH;
        return 1;
    }
    return 0;
}

void ao_demod_output_carry(ao_demod* d) { /* output.cpp:948 */
    for (int i = 0; i < d->nch; i++)
        memcpy(d->ch[i].waveout, d->ch[i].waveout + AO_WAVE_BATCH, AO_AGC_EXTRA * 4);
}

int ao_demod_run(ao_demod* d, const unsigned char* iq, size_t nbytes, int max_batches, float* waveout, float* iq_out, char* axc) {
    size_t bufs = 0;
    int nb = 0;
    const size_t need = d->hop_bytes + d->fft_size * (size_t)d->bytes_per_sample * 2; /* rtl_airband.cpp:417 */
    while (nb < max_batches && nbytes - bufs >= need) {
        if (ao_demod_push_window(d, iq + bufs)) {
            for (int c = 0; c < d->nch; c++) {
                memcpy(waveout + ((size_t)c * max_batches + nb) * AO_WAVE_BATCH, d->ch[c].waveout, AO_WAVE_BATCH * sizeof(float));
                if (iq_out)
                    memcpy(iq_out + ((size_t)c * max_batches + nb) * 2 * AO_WAVE_BATCH, d->ch[c].iq_out,
                           2 * AO_WAVE_BATCH * sizeof(float));
                axc[(size_t)c * max_batches + nb] = d->ch[c].axcindicate;
            }
            ao_demod_output_carry(d);
            nb++;
        }
        bufs += d->hop_bytes; /* rtl_airband.cpp:691 */
    }
    return nb;
}

void ao_stage1(ao_demod* d, const unsigned char* iq, size_t nfft, float* mag, float* iqout) {
    for (size_t f = 0; f < nfft; f++) {
        convert_window(d, iq + f * d->hop_bytes);
        ao_fft_forward(&d->plan, d->fftin, d->fftout);
        for (int j = 0; j < d->nch; j++) {
            const float re = d->fftout[2 * d->ch[j].bin], im = d->fftout[2 * d->ch[j].bin + 1];
            mag[(size_t)j * nfft + f] = sqrtf(re * re + im * im);
            if (iqout) {
                iqout[((size_t)j * nfft + f) * 2] = re;
                iqout[((size_t)j * nfft + f) * 2 + 1] = im;
            }
        }
    }
}

/* ---- mixer (src/mixer.cpp): mixer_connect_input :56-98, mix_waveforms :133-140, mixer_thread :190-213 ---- */
void ao_mix_waveforms(float* sum, const float* in, float mult, int size) { /* mixer.cpp:133-140 */
    if (mult == 0.0f)
        return;
    for (int s = 0; s < size; s++)
        sum[s] += in[s] * mult;
}

int ao_mixer_run(const ao_mix_input* inputs, int ninputs, const float* waveout, size_t row_stride, const char* axc, size_t axc_stride,
                 int nbatches, float* left, float* right, char* axc_out) {
    int stereo = 0;
    for (int j = 0; j < ninputs; j++)
        if (inputs[j].balance != 0.0f)
            stereo = 1; /* mixer.cpp:82-83 */
    for (int b = 0; b < nbatches; b++) {
        float* l = left + (size_t)b * AO_WAVE_BATCH;
        float* r = right ? right + (size_t)b * AO_WAVE_BATCH : NULL;
        memset(l, 0, AO_WAVE_BATCH * sizeof(float)); /* mixer.cpp:194-199 */
        if (stereo && r)
            memset(r, 0, AO_WAVE_BATCH * sizeof(float));
        axc_out[b] = ' ';
        for (int j = 0; j < ninputs; j++) { /* inputs in index order: the jitter-free order of mixer.cpp:190-213 */
            const ao_mix_input* in = &inputs[j];
            const float ampl = fminf(1.0f, 1.0f - in->balance), ampr = fminf(1.0f, 1.0f + in->balance); /* mixer.cpp:80-81 */
            if (axc[(size_t)in->row * axc_stride + (size_t)b] == ' ') /* has_signal, output.cpp:564 */
                continue;
            const float* src = waveout + (size_t)in->row * row_stride + (size_t)b * AO_WAVE_BATCH;
            ao_mix_waveforms(l, src, in->ampfactor * ampl, AO_WAVE_BATCH);
            if (stereo && r)
                ao_mix_waveforms(r, src, in->ampfactor * ampr, AO_WAVE_BATCH);
            axc_out[b] = '*';
        }
    }
    return stereo;
}
