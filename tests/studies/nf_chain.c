/* nf_chain.c -- TEST INFRASTRUCTURE (study helper, compiled by tests/studies/nf_chain_study.py).
 *
 * Questions about the squelch core chain (noise_floor_, pre_filter_.capped_; /root/reference/src/squelch.cpp:195-214,
 * 477-514) that decide how tp.hip may cut it in time:
 *   1. does a trajectory started W blocks early from a bound of the true state meet the true one bit for bit?  
 *   2. how often does "noise floor after the next block == noise floor + a constant number of ulps" hold?       (k_tp_core2, wave 0)
 * Arithmetic: IEEE single, no contraction (-ffp-contract=off), the operations of the reference in its order.
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

static inline uint32_t fbits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
static inline float ema99(float f, float x) {
    const float nfac = (float)(1.0 - (double)0.99f);
    return f * 0.99f + x * nfac;
}
static inline float capped_step(float c, float x, float cap) {
    if (c >= cap && x >= cap)
        return cap;
    const float e = ema99(c, x);
    return e < cap ? e : cap;
}
static inline float nf_step(float nf, float c) {
    const float nfac = (float)(1.0 - (double)0.97f);
    const float m = c < nf ? c : nf;
    return nf * 0.97f + m * nfac + 1e-6f;
}

/* The exact chain over n samples (n % 16 == 0) from (nf, c, full); per block b: st[4b..] = nf, c, full ENTERING the block's first
 * sample (before the noise-floor update) and the noise floor after the update. */
void chain_exact(const float* x, size_t n, float cap_factor, float nf, float c, float full, float* st) {
    for (size_t i = 0; i < n; i++) {
        if ((i & 15) == 0) {
            float* o = st + 4 * (i >> 4);
            o[0] = nf, o[1] = c, o[2] = full;
            nf = nf_step(nf, c);
            o[3] = nf;
        }
        const float cap = cap_factor * nf;
        full = ema99(full, x[i]);
        c = capped_step(c, x[i], cap);
    }
}

/* Question 1.  For every segment start s = k*seg_blocks (k >= 1): start at block s - W from nf = truth * nf_scale (+ nf_add), c = the
 * true full_ there (a lane knows full_ exactly), walk forward, and report the first block >= s - W at which (nf, c) equal the truth
 * bit for bit -- or -1 if they have not met by the end of the segment.  st = chain_exact's output. */
void meet_blocks(const float* x, size_t n, float cap_factor, const float* st, size_t seg_blocks, size_t W, float nf_scale, float nf_add,
                 int32_t* met_after /* per segment: blocks after s - W, -1 = never within W + seg */) {
    const size_t nblk = n / 16;
    for (size_t k = 1; k * seg_blocks < nblk; k++) {
        const size_t s = k * seg_blocks;
        const size_t b0 = s > W ? s - W : 0;
        const size_t b1 = (s + seg_blocks < nblk) ? s + seg_blocks : nblk;
        float nf = st[4 * b0] * nf_scale + nf_add, full = st[4 * b0 + 2], c = full;
        int32_t met = -1;
        for (size_t b = b0; b < b1 && met < 0; b++) {
            if (fbits(nf) == fbits(st[4 * b]) && fbits(c) == fbits(st[4 * b + 1])) {
                met = (int32_t)(b - b0);
                break;
            }
            nf = nf_step(nf, c);
            const float cap = cap_factor * nf;
            for (int j = 0; j < 16; j++) {
                full = ema99(full, x[16 * b + j]);
                c = capped_step(c, x[16 * b + j], cap);
            }
        }
        met_after[k] = met;
    }
}

/* Question 2.  Wave 0 of k_tp_core2: lanes guess the noise floor entering their block as nf + inc*(lane - kk) on the bit pattern,
 * inc = bits(fl(nf + 1e-6)) - bits(nf); every lane takes its true step from its guess; the run of lanes whose result equals the next
 * lane's guess is accepted, plus the first lane whose result differs (its input was right, so its result is the truth).
 * Returns the number of rounds; hist[r] counts groups of 64 blocks that took r rounds (r capped at 64); ev_kind[0..2] counts why a
 * round ended early: 0 the step was not the self step (operand below the floor), 1 a self step off by some ulps, 2 group end. */
uint64_t guess_rounds(const float* st, size_t nblk, uint64_t* hist, uint64_t* ev_kind, int fallback_after, uint64_t* classic_blocks) {
    uint64_t rounds = 0;
    for (size_t g0 = 0; g0 < nblk; g0 += 64) {
        const size_t nb = (nblk - g0 < 64) ? nblk - g0 : 64;
        size_t kk = 0;
        int r = 0, shorts = 0;
        while (kk < nb) {
            const float nf = st[4 * (g0 + kk)];
            if (fallback_after > 0 && shorts >= fallback_after) {  /* the classical systolic pass walks the rest of the group */
                *classic_blocks += nb - kk;
                kk = nb;
                r += 1;
                break;
            }
            const uint32_t inc = fbits(nf + 1e-6f) - fbits(nf);
            size_t j = kk;
            for (;; j++) {
                uint32_t gb = fbits(nf) + inc * (uint32_t)(j - kk);
                float gv;
                memcpy(&gv, &gb, 4);
                /* (by induction gv == st[4*(g0+j)] here) */
                const float out = nf_step(gv, st[4 * (g0 + j) + 1]);
                if (j + 1 >= nb) {
                    ev_kind[2]++;
                    break;
                }
                if (fbits(out) != gb + inc) {
                    ev_kind[st[4 * (g0 + j) + 1] < gv ? 0 : 1]++;
                    break;
                }
            }
            shorts = (j - kk + 1 < 4) ? shorts + 1 : 0;
            kk = j + 1;
            r++;
        }
        rounds += (uint64_t)r;
        hist[r > 64 ? 64 : r]++;
    }
    return rounds;
}

/* increments of the noise floor over self steps (capped >= floor), in ulps relative to bits(fl(nf + 1e-6)) - bits(nf): hist[d + 3] */
void self_step_hist(const float* st, size_t nblk, uint64_t* hist) {
    for (size_t b = 0; b + 1 < nblk; b++) {
        const float nf = st[4 * b], c = st[4 * b + 1];
        if (c < nf)
            continue;
        const int inc = (int)(fbits(nf + 1e-6f) - fbits(nf));
        int d = (int)(fbits(st[4 * b + 3]) - fbits(nf)) - inc;
        d = d < -3 ? -3 : (d > 3 ? 3 : d);
        hist[d + 3]++;
    }
}

/* Wave 0 of k_tp_core2, second formulation: all 64 lanes hold a guess g_j of the floor entering block j, take the true step o_j =
 * F_j(g_j), and the group is done when o_{j-1} == g_j for every j (then every g_j, hence every o_j, is exact by induction).  Otherwise
 * the guesses are rebuilt as g_j = nf + sum_{i<j} (o_i - g_i) (a wave prefix sum) and the step repeated.  predictor: 0 = first guess
 * g_j = nf + j*inc; 1 = the previous group's observed increments shifted by `pshift` lanes (periodic pattern).  Returns the total
 * number of iterations; hist[k] = groups that needed k (capped at 16 = give up, classical pass). */
uint64_t refine_iterations(const float* st, size_t nblk, int predictor, int pshift, uint64_t* hist) {
    uint64_t total = 0;
    int32_t dprev[64];
    int have_prev = 0;
    for (size_t g0 = 0; g0 + 64 <= nblk; g0 += 64) {
        const uint32_t nfb = fbits(st[4 * g0]);
        const uint32_t inc = fbits(st[4 * g0] + 1e-6f) - nfb;
        uint32_t g[64], o[64];
        if (predictor == 1 && have_prev) {
            uint32_t acc = nfb;
            for (int j = 0; j < 64; j++) {
                g[j] = acc;
                acc += (uint32_t)dprev[(j + pshift) % 64 < 64 - pshift ? (j + pshift) : (j + pshift) % 64 % (pshift ? pshift * 21 : 1)];
            }
        } else {
            for (int j = 0; j < 64; j++)
                g[j] = nfb + inc * (uint32_t)j;
        }
        int it = 0;
        for (;;) {
            it++;
            int ok = 1;
            for (int j = 0; j < 64; j++) {
                float gv;
                memcpy(&gv, &g[j], 4);
                o[j] = fbits(nf_step(gv, st[4 * (g0 + j) + 1]));
            }
            for (int j = 1; j < 64; j++)
                if (o[j - 1] != g[j])
                    ok = 0;
            if (ok || it >= 16)
                break;
            uint32_t acc = nfb;
            for (int j = 0; j < 64; j++) {
                const uint32_t d = o[j] - g[j];
                g[j] = acc;
                acc += d;
            }
        }
        /* the exact increments of this group (for the predictor) */
        for (int j = 0; j < 64; j++)
            dprev[j] = (int32_t)(fbits(st[4 * (g0 + j) + 3]) - fbits(st[4 * (g0 + j)]));
        have_prev = 1;
        total += (uint64_t)it;
        hist[it]++;
    }
    return total;
}

/* Wave 0 of k_tp_core2 as built (nf_chain_guess64, tp.hip): rounds.  Position kk in the group is exact; lanes j >= kk guess the floor
 * entering their block as nf_kk + h0 + h1 + h2 + h0 + ... (j - kk terms), h = the increments of the last three settled blocks (an
 * increment that is not inc-1 .. inc+1 -- a step below the floor -- counts as inc); every lane takes the true step from its guess;
 * settled: the lanes up to and including the first whose result is not the next lane's guess.  hist[r] = groups that took r rounds
 * (cap 32); accepted_hist[n] = rounds that settled n blocks. */
uint64_t p3_rounds(const float* st, size_t nblk, uint64_t* hist, uint64_t* accepted_hist) {
    uint64_t total = 0;
    uint32_t e[3] = {0, 0, 0};
    int have = 0;
    for (size_t g0 = 0; g0 + 64 <= nblk; g0 += 64) {
        size_t kk = 0;
        int r = 0;
        while (kk < 64) {
            const float nf = st[4 * (g0 + kk)];
            const uint32_t nfb = fbits(nf), inc = fbits(nf + 1e-6f) - nfb;
            uint32_t ee[3];
            for (int q = 0; q < 3; q++)
                ee[q] = (have && e[q] + 1u - inc <= 2u) ? e[q] : inc;
            uint32_t g = nfb;
            size_t j = kk;
            for (;; j++) {
                float gv;
                memcpy(&gv, &g, 4);
                const uint32_t ob = fbits(nf_step(gv, st[4 * (g0 + j) + 1]));
                if (j + 1 >= 64 || ob != g + ee[(g0 + j) % 3])
                    break;
                g = ob;
            }
            /* blocks kk .. j are exact: refresh the history from the last three of them (older ones keep their slot) */
            for (size_t i = (j >= 2 ? j - 2 : 0); i <= j; i++)
                if (i >= kk || 1)
                    e[(g0 + i) % 3] = fbits(st[4 * (g0 + i) + 3]) - fbits(st[4 * (g0 + i)]);
            have = 1;
            accepted_hist[j - kk + 1]++;
            kk = j + 1;
            r++;
        }
        total += (uint64_t)r;
        hist[r > 32 ? 32 : r]++;
    }
    return total;
}

/* p3_rounds with the dip runs walked by systolic passes: when a round ends at a lane and the lanes behind it look like steps below
 * the floor by their (old) guesses (operand < guess), those L lanes are settled by L classical passes instead of L rounds.
 * out[0] = rounds, out[1] = classical passes, out[2] = dip runs. */
void p3_rounds_dips(const float* st, size_t nblk, uint64_t* out) {
    uint32_t e[3] = {0, 0, 0};
    int have = 0;
    out[0] = out[1] = out[2] = 0;
    for (size_t g0 = 0; g0 + 64 <= nblk; g0 += 64) {
        size_t kk = 0;
        while (kk < 64) {
            const float nf = st[4 * (g0 + kk)];
            const uint32_t nfb = fbits(nf), inc = fbits(nf + 1e-6f) - nfb;
            uint32_t ee[3], guess[65];
            for (int q = 0; q < 3; q++)
                ee[q] = (have && e[q] + 1u - inc <= 2u) ? e[q] : inc;
            guess[kk] = nfb;
            for (size_t j = kk; j < 64; j++)
                guess[j + 1] = guess[j] + ee[(g0 + j) % 3];
            size_t j = kk;
            uint32_t g = nfb;
            for (;; j++) {
                float gv;
                memcpy(&gv, &g, 4);
                const uint32_t ob = fbits(nf_step(gv, st[4 * (g0 + j) + 1]));
                if (j + 1 >= 64 || ob != guess[j + 1])
                    break;
                g = ob;
            }
            for (size_t i = (j >= 2 ? j - 2 : 0); i <= j; i++)
                e[(g0 + i) % 3] = fbits(st[4 * (g0 + i) + 3]) - fbits(st[4 * (g0 + i)]);
            have = 1;
            out[0]++;
            kk = j + 1;
            /* dip run behind the round's last lane, judged by the round's own guesses */
            size_t L = 0;
            while (kk + L < 64) {
                float gv;
                memcpy(&gv, &guess[kk + L], 4);
                if (!(st[4 * (g0 + kk + L) + 1] < gv))
                    break;
                L++;
            }
            if (L) {
                out[1] += L;
                out[2]++;
                for (size_t i = (kk + L >= 3 ? kk + L - 3 : 0); i < kk + L; i++)
                    e[(g0 + i) % 3] = fbits(st[4 * (g0 + i) + 3]) - fbits(st[4 * (g0 + i)]);
                kk += L;
            }
        }
    }
}
