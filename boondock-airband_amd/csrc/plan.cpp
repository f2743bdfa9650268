// plan.cpp -- host-side derivation of every constant the kernels use.  Each formula follows the
// reference's own expression, type by type, because the results feed float comparisons on the device
// (squelch decisions must be bit-exact).  Citations are /root/reference/src/<file>:<line>.
#include "plan.hpp"

#include <algorithm>
#include <cstring>

#include <cmath>
#include <complex>
#include <cstring>

namespace mi {
namespace {

// ToneDetector::ToneDetector, ctcss.cpp:31-42 (omega is a float; cos(float) is the float overload)
float goertzel_coeff(float tone_freq, float sample_rate, int window_size) {
    const int k = static_cast<int>(0.5 + window_size * tone_freq / sample_rate);
    const float omega = static_cast<float>((2.0 * M_PI * k) / window_size);
    return static_cast<float>(2.0 * std::cos(omega));
}

// CTCSS::CTCSS + ToneDetectorSet::add, ctcss.cpp:61-73,105-122: target tone first, then every standard
// tone at least 5 Hz away, dropping tones whose coefficient duplicates an earlier one.
int goertzel_bank(float ctcss_freq, float sample_rate, int window_size, float* coeffs) {
    static const float standard_tones[] = {67.0f,  69.3f,  71.9f,  74.4f,  77.0f,  79.7f,  82.5f,  85.4f,  88.5f,  91.5f,  94.8f,
                                           97.4f,  100.0f, 103.5f, 107.2f, 110.9f, 114.8f, 118.8f, 123.0f, 127.3f, 131.8f, 136.5f,
                                           141.3f, 146.2f, 150.0f, 151.4f, 156.7f, 159.8f, 162.2f, 165.5f, 167.9f, 171.3f, 173.8f,
                                           177.3f, 179.9f, 183.5f, 186.2f, 189.9f, 192.8f, 196.6f, 199.5f, 203.5f, 206.5f, 210.7f,
                                           218.1f, 225.7f, 229.1f, 233.6f, 241.8f, 250.3f, 254.1f};  // ctcss.cpp:101-103
    int n = 0;
    auto add = [&](float f) {
        const float c = goertzel_coeff(f, sample_rate, window_size);
        for (int i = 0; i < n; ++i)
            if (coeffs[i] == c)
                return;
        coeffs[n++] = c;
    };
    add(ctcss_freq);
    for (float t : standard_tones) {
        if (std::abs(ctcss_freq - t) < 5)
            continue;
        add(t);
    }
    return n;
}

// LowpassFilter::LowpassFilter, filters.cpp:69-144: 2nd-order Bessel through the bilinear transform
void bessel_lowpass(float freq, float sample_freq, float& gain, float& yc0, float& yc1) {
    using cd = std::complex<double>;
    const double raw_alpha = static_cast<double>(freq) / sample_freq;
    const double warped_alpha = std::tan(M_PI * raw_alpha) / M_PI;
    auto blt = [](cd pz) { return (2.0 + pz) / (2.0 - pz); };
    const cd bessel_pole(-1.10160133059e+00, 6.36009824757e-01);
    cd poles[2] = {blt(M_PI * 2 * warped_alpha * bessel_pole), blt(M_PI * 2 * warped_alpha * std::conj(bessel_pole))};
    cd zeros[2] = {-1.0, -1.0};
    auto expand = [](const cd* pz, cd* coeffs) {
        coeffs[0] = 1.0;
        coeffs[1] = coeffs[2] = 0.0;
        for (int i = 0; i < 2; ++i) {
            const cd nw = -pz[i];
            for (int k = 2; k >= 1; --k)
                coeffs[k] = (nw * coeffs[k]) + coeffs[k - 1];
            coeffs[0] = nw * coeffs[0];
        }
    };
    auto eval = [](const cd* coeffs, cd z) {
        cd sum(0.0);
        for (int i = 2; i >= 0; --i)
            sum = (sum * z) + coeffs[i];
        return sum;
    };
    cd top[3], bot[3];
    expand(zeros, top);
    expand(poles, bot);
    const cd g = eval(top, 1.0) / eval(bot, 1.0);
    gain = static_cast<float>(std::hypot(g.imag(), g.real()));
    yc0 = static_cast<float>(-(bot[0].real() / bot[2].real()));
    yc1 = static_cast<float>(-(bot[1].real() / bot[2].real()));
}

// dBFS_to_level, util.cpp:169-176
float dbfs_to_level(float dbfs, size_t fft_size) {
    const float offset = 7.54f + 10.0f * log10f(static_cast<float>(fft_size / 2)) - 2.38f;
    return static_cast<float>(std::pow(10.0, (dbfs - offset) / 20.0f) * static_cast<double>(fft_size));
}

// global alpha (rtl_airband.cpp:87) and the `tau` overrides (config.cpp:651,778)
float alpha_for_tau(int tau_us) {
    if (tau_us < 0)
        return static_cast<float>(std::exp(-1.0f / (kWaveRate * 2e-4)));
    return tau_us == 0 ? 0.0f : static_cast<float>(std::exp(-1.0f / (kWaveRate * 1e-6 * tau_us)));
}

}  // namespace

int build_plan(const mi_device_cfg& dev, const mi_channel_cfg* chans, int nch, Plan& p, const char** msg) {
    *msg = "";
    if (!chans || nch < 1) {
        *msg = "no channels configured";  // config.cpp:811-814
        return MI_ERR_INVALID;
    }
    if (dev.fft_size_log < 8 || dev.fft_size_log > 13) {
        *msg = "fft_size must be a power of two in 2^8..2^13";  // rtl_airband.cpp:808-822
        return MI_ERR_INVALID;
    }
    if (dev.sample_rate <= kWaveRate) {
        *msg = "sample_rate must be greater than WAVE_RATE";  // config.cpp:753-760,794
        return MI_ERR_INVALID;
    }
    if (dev.sfmt < MI_SFMT_U8 || dev.sfmt > MI_SFMT_F32) {
        *msg = "unknown sample format";
        return MI_ERR_INVALID;
    }
    if ((dev.sfmt == MI_SFMT_S16 || dev.sfmt == MI_SFMT_F32) && !(dev.fullscale > 0)) {
        *msg = "fullscale must be positive";  // config.cpp:791-792
        return MI_ERR_INVALID;
    }
    p.dev = dev;
    p.chans.assign(chans, chans + nch);
    p.nch = nch;
    p.log2n = dev.fft_size_log;
    p.fft_size = 1 << dev.fft_size_log;
    p.bytes_per_sample = dev.sfmt == MI_SFMT_S16 ? 2 : (dev.sfmt == MI_SFMT_F32 ? 4 : 1);
    p.hop_bytes = 2 * p.bytes_per_sample * static_cast<size_t>(std::round(static_cast<double>(dev.sample_rate) / kWaveRate));
    const size_t n = static_cast<size_t>(p.fft_size);

    // "blackman 7" window, rtl_airband.cpp:357-373: float-rounded constants widened to double
    p.window.resize(n);
    {
        const double a0 = 0.27105140069342f, a1 = 0.43329793923448f, a2 = 0.21812299954311f, a3 = 0.06592544638803f;
        const double a4 = 0.01081174209837f, a5 = 0.00077658482522f, a6 = 0.00001388721735f;
        for (size_t i = 0; i < n; ++i) {
            const double x = a0 - (a1 * cos((2.0 * M_PI * i) / (n - 1))) + (a2 * cos((4.0 * M_PI * i) / (n - 1))) -
                             (a3 * cos((6.0 * M_PI * i) / (n - 1))) + (a4 * cos((8.0 * M_PI * i) / (n - 1))) -
                             (a5 * cos((10.0 * M_PI * i) / (n - 1))) + (a6 * cos((12.0 * M_PI * i) / (n - 1)));
            p.window[i] = static_cast<float>(x);
        }
    }
    // FFT twiddles W_N^k = e^{-2 pi j k/N}, k < N/2, with the two exact points forced (FFT spec)
    p.tw.resize(n);
    for (size_t k = 0; k < n / 2; ++k) {
        const double a = 2.0 * M_PI * static_cast<double>(k) / static_cast<double>(n);
        p.tw[2 * k] = static_cast<float>(cos(a));
        p.tw[2 * k + 1] = static_cast<float>(-sin(a));
    }
    p.tw[0] = 1.0f;
    p.tw[1] = 0.0f;
    p.tw[2 * (n / 4)] = 0.0f;
    p.tw[2 * (n / 4) + 1] = -1.0f;

    // level LUTs, rtl_airband.cpp:341-346 (entry 128 of the s8 table is never written there: 0 here)
    p.levels.assign(256, 0.0f);
    if (dev.sfmt == MI_SFMT_S8) {
        for (int16_t i = -127; i < 128; ++i)
            p.levels[static_cast<uint8_t>(i)] = i / 128.0f;
    } else {
        for (int i = 0; i < 256; ++i)
            p.levels[i] = (i - 127.5f) / 127.5f;
    }
    p.conv_arith = false;
    if (dev.sfmt == MI_SFMT_U8) {  // the same operations as level_u8() in channelize.hip, IEEE fp32 on both sides
        volatile float d = 127.5f, rcp = 1.0f / 127.5f;
        bool same = true;
        for (int i = 0; i < 256 && same; ++i) {
            const float n = static_cast<float>(i) - 127.5f;
            volatile float q0 = n * rcp;
            const float q1 = std::fmaf(std::fmaf(-q0, d, n), rcp, q0);
            same = std::memcmp(&q1, &p.levels[static_cast<size_t>(i)], 4) == 0;
        }
        p.conv_arith = same;
    }
    p.conv_scale = (dev.sfmt == MI_SFMT_S16 || dev.sfmt == MI_SFMT_F32) ? 1.0f / dev.fullscale : 0.0f;  // rtl_airband.cpp:425,443

    // sincosf_lut_init, util.cpp:105-110
    for (uint32_t i = 0; i < 256; ++i)
        sincosf(static_cast<float>(2.0F * M_PI * static_cast<float>(i) / 256.0f), p.sin_lut + i, p.cos_lut + i);
    p.sin_lut[256] = p.sin_lut[0];
    p.cos_lut[256] = p.cos_lut[0];

    const float dev_alpha = alpha_for_tau(dev.tau);
    p.cp.assign(nch, ChanParams{});
    p.n_iq_rows = 0;
    p.n_ctcss_rows = 0;
    p.any_afc = false;
    p.ctcss_coeff.clear();
    for (int i = 0; i < nch; ++i) {
        const mi_channel_cfg& k = chans[i];
        ChanParams& c = p.cp[i];
        if (k.modulation != MI_MOD_AM && k.modulation != MI_MOD_NFM) {
            *msg = "unknown modulation";  // config.cpp:344-355
            return MI_ERR_INVALID;
        }
        if (k.afc < 0 || k.afc > 255) {
            *msg = "afc must be 0..255";  // unsigned char, config.cpp:355
            return MI_ERR_INVALID;
        }
        c.afc = static_cast<uint32_t>(k.afc);
        p.any_afc = p.any_afc || k.afc != 0;
        if (k.squelch_threshold_dbfs > 0) {
            *msg = "squelch_threshold must be less than or equal to 0";  // config.cpp:447-449
            return MI_ERR_INVALID;
        }
        if (k.has_snr_threshold && k.squelch_snr_db < 0 && k.squelch_snr_db != -1.0f) {
            *msg = "squelch_snr_threshold must be greater than or equal to 0";  // config.cpp:497-499
            return MI_ERR_INVALID;
        }
        if (k.ampfactor < 0) {
            *msg = "ampfactor must not be negative";  // config.cpp:628-631
            return MI_ERR_INVALID;
        }
        if (k.notch_q < 0) {
            *msg = "invalid value for notch_q";  // config.cpp:533-536
            return MI_ERR_INVALID;
        }
        c.modulation = k.modulation;
        c.ampfactor = k.ampfactor;
        c.alpha = k.tau >= 0 ? alpha_for_tau(k.tau) : dev_alpha;
        c.one_minus_alpha = 1.0f - c.alpha;
        c.has_iq_outputs = k.has_iq_outputs ? 1 : 0;
        c.needs_raw_iq = (k.has_iq_outputs || k.bandwidth != 0 || k.modulation == MI_MOD_NFM) ? 1 : 0;  // config.cpp:162,596,676 (< 0: key present without a usable value)

        // Squelch(): default snr 9.54 dB; then squelch_threshold, then squelch_snr_threshold (config.cpp:440-518)
        float ratio = static_cast<float>(std::pow(10.0, 9.54f / 20.0));  // squelch.cpp:38,100 (db is a float)
        c.using_manual_level = 0;
        c.manual_signal_level = -1.0f;
        if (k.squelch_threshold_dbfs < 0) {
            const float level = dbfs_to_level(static_cast<float>(k.squelch_threshold_dbfs), n);
            if (level > 0) {  // squelch.cpp:85
                c.using_manual_level = 1;
                c.manual_signal_level = level;
            }
        }
        if (k.has_snr_threshold && k.squelch_snr_db != -1.0f) {
            c.using_manual_level = 0;  // squelch.cpp:99
            ratio = static_cast<float>(std::pow(10.0, k.squelch_snr_db / 20.0));
        }
        c.normal_signal_ratio = ratio;
        c.flappy_signal_ratio = ratio * 0.9f;
        c.cap_factor = 1.5f * ratio;
        c.manual_cap = 1.5f * c.manual_signal_level;

        // NotchFilter, filters.cpp:30-48 (tan/cos on float arguments are the float overloads)
        c.notch_enabled = 0;
        if (k.notch_freq > 0) {
            const float q = k.notch_q > 0 ? k.notch_q : 10.0f;  // config.cpp:520,530-532
            const float wo = static_cast<float>(2 * M_PI * (k.notch_freq / static_cast<float>(kWaveRate)));
            const float e = 1 / (1 + std::tan(wo / (q * 2)));
            const float pp = std::cos(wo);
            c.notch_enabled = 1;
            c.notch_d0 = e;
            c.notch_d1 = 2 * e * pp;
            c.notch_d2 = (2 * e - 1);
        }
        c.lowpass_enabled = 0;
        if (k.bandwidth > 0) {  // config.cpp:595-622: LowpassFilter((float)bandwidth / 2, WAVE_RATE)
            c.lowpass_enabled = 1;
            bessel_lowpass(static_cast<float>(k.bandwidth) / 2, static_cast<float>(kWaveRate), c.lowpass_gain, c.lowpass_yc0, c.lowpass_yc1);
        }
        c.ctcss_enabled = 0;
        c.ctcss_row = -1;
        if (k.ctcss_freq > 0) {  // Squelch::set_ctcss_freq, squelch.cpp:110-116
            const float rate = static_cast<float>(kWaveRate);
            c.ctcss_enabled = 1;
            c.ctcss_fast_window = static_cast<int>(rate * 0.05);
            c.ctcss_slow_window = static_cast<int>(rate * 0.4);
            c.ctcss_row = p.n_ctcss_rows++;
            p.ctcss_coeff.resize(static_cast<size_t>(p.n_ctcss_rows) * 2 * kMaxTones, 0.0f);
            float* base = p.ctcss_coeff.data() + static_cast<size_t>(c.ctcss_row) * 2 * kMaxTones;
            c.ctcss_fast_ndet = goertzel_bank(k.ctcss_freq, rate, c.ctcss_fast_window, base);
            c.ctcss_slow_ndet = goertzel_bank(k.ctcss_freq, rate, c.ctcss_slow_window, base + kMaxTones);
        }
        // bin, config.cpp:669-670: note the INTEGER quotient sample_rate / fft_size
        c.bin = static_cast<uint32_t>(
            static_cast<size_t>(std::ceil((k.freq + dev.sample_rate - dev.centerfreq) / static_cast<double>(static_cast<size_t>(dev.sample_rate) / n) - 1.0)) % n);
        c.iq_row = -1;
        c.dm_dphi = 0;
        if (c.needs_raw_iq) {  // config.cpp:682-713
            c.iq_row = p.n_iq_rows++;
            double dm = static_cast<double>(k.freq - dev.centerfreq);
            const double decimation_factor = static_cast<double>(dev.sample_rate) / static_cast<double>(kWaveRate);
            double corr = static_cast<double>(kWaveRate) / 2.0;
            corr *= (decimation_factor - std::round(decimation_factor));
            corr *= static_cast<double>(k.freq - dev.centerfreq) / (static_cast<double>(dev.sample_rate) / 2.0);
            dm -= corr;
            dm /= static_cast<double>(kWaveRate);
            dm -= std::trunc(dm);
            dm *= 256.0 * 65536.0;
            c.dm_dphi = static_cast<uint32_t>(static_cast<int>(dm));
        }
    }
    // ---- stage-1 pruning (see PrunePlan) ----
    {
        PrunePlan& pr = p.prune;
        pr = PrunePlan{};
        p.prune_t1.clear();
        p.prune_t2.clear();
        p.prune_chan_rank.assign(static_cast<size_t>(nch), 0);
        const int L = p.log2n, N = p.fft_size;
        if (L == 9 && !p.any_afc) {
            auto residues = [&](int s) {
                std::vector<int> r;
                for (const ChanParams& c : p.cp)
                    r.push_back(static_cast<int>(c.bin) & ((1 << s) - 1));
                std::sort(r.begin(), r.end());
                r.erase(std::unique(r.begin(), r.end()), r.end());
                return r;
            };
            const std::vector<int> R3 = residues(3), R6 = residues(6), R9 = residues(9);
            auto rank_in = [](const std::vector<int>& r, int v) {
                const auto it = std::lower_bound(r.begin(), r.end(), v);
                return (it != r.end() && *it == v) ? static_cast<int>(it - r.begin()) : -1;
            };
            auto log2up = [](int m) {
                int sh = 0;
                while ((1 << sh) < m)
                    ++sh;
                return sh;
            };
            pr.m3 = static_cast<int>(R3.size()), pr.sh3 = log2up(pr.m3);
            pr.m6 = static_cast<int>(R6.size()), pr.sh6 = log2up(pr.m6);
            pr.m9 = static_cast<int>(R9.size());
            pr.rs1 = 8 * pr.m3 + 1;  // pass 1 reads a row with consecutive lanes; the odd stride spreads pass 0's writes
            pr.rs2 = pr.m6 + 1;
            for (int r = 0; r < 8; ++r)
                pr.rank3[r] = rank_in(R3, r);
            // class tables: twiddles exactly as Pass<9, K>::load_tw builds them for lo = the class's residue
            auto classes = [&](std::vector<float>& t, const std::vector<int>& Rin, int sh, int K, const std::vector<int>& Rout) {
                const int S = 1 << (3 * K);
                t.assign(static_cast<size_t>(1 << sh) * kPruneClassWords, 0.0f);
                for (size_t j = 0; j < Rin.size(); ++j) {
                    float* c = t.data() + j * kPruneClassWords;
                    const int lo = Rin[j];
                    for (int a = 0; a < 3; ++a)
                        for (int mm = 0; mm < (1 << a); ++mm) {
                            const int e = (mm * S + lo) * (N >> (3 * K + 1 + a));
                            c[2 * (((1 << a) - 1) + mm)] = p.tw[2 * static_cast<size_t>(e)];
                            c[2 * (((1 << a) - 1) + mm) + 1] = p.tw[2 * static_cast<size_t>(e) + 1];
                        }
                    uint32_t o[2] = {0, 0};  // a byte per output slot, 0xff = not needed
                    for (int ri = 0; ri < 8; ++ri)
                        o[ri >> 2] |= static_cast<uint32_t>(rank_in(Rout, ri * S + lo) & 0xff) << (8 * (ri & 3));
                    std::memcpy(c + 14, o, 8);
                }
            };
            classes(p.prune_t1, R3, pr.sh3, 1, R6);
            classes(p.prune_t2, R6, pr.sh6, 2, R9);
            for (int i = 0; i < nch; ++i)
                p.prune_chan_rank[static_cast<size_t>(i)] = rank_in(R9, static_cast<int>(p.cp[static_cast<size_t>(i)].bin));
            // it pays when the packed passes are well under the full ones: pass 1 shrinks to m3/8, pass 2 to m6/64
            pr.enabled = (pr.m3 * 8 + pr.m6 <= 64 && pr.m9 < 255) ? 1 : 0;
        }
    }
    // ---- lane-resident stage 1 (see L64Plan) ----
    {
        L64Plan& lp = p.l64;
        lp = L64Plan{};
        p.l64_chan.clear();
        p.l64_chan_full.clear();
        static const float kTwLit[256][2] = {
#include "tw512.inc"
        };
        bool lits_ok = p.fft_size == 512;
        for (size_t k = 0; lits_ok && k < 256; ++k)  // the kernel's literal twiddles must be the spec's table, bit for bit
            lits_ok = std::memcmp(&kTwLit[k][0], &p.tw[2 * k], 4) == 0 && std::memcmp(&kTwLit[k][1], &p.tw[2 * k + 1], 4) == 0;
        const size_t hop = p.hop_bytes / (2 * static_cast<size_t>(p.bytes_per_sample));
        if (lits_ok && !p.any_afc && nch <= 64 && (hop == 160 || hop == 128)) {
            for (int s = 1; s <= 6; ++s)
                for (const ChanParams& c : p.cp)
                    lp.need[s - 1] |= 1ull << (c.bin & ((1u << s) - 1u));
            lp.m6 = __builtin_popcountll(lp.need[5]);
            lp.nb_pad = 8;
            while (lp.nb_pad < nch)
                lp.nb_pad *= 2;
            auto tw_signed = [&](unsigned bin, int s, float& x, float& y) {  // stage s = 7, 8, 9: W_{2^s}^{bin mod 2^(s-1)}, negated for an upper output
                const unsigned half = 1u << (s - 1);
                const unsigned e = (bin & (half - 1u)) * (512u >> s);
                x = p.tw[2 * e], y = p.tw[2 * e + 1];
                if (bin & half)
                    x = -x, y = -y;
            };
            p.l64_chan.resize(static_cast<size_t>(nch));
            for (int i = 0; i < nch; ++i) {
                const ChanParams& c = p.cp[static_cast<size_t>(i)];
                L64Chan& o = p.l64_chan[static_cast<size_t>(i)];
                const unsigned cls = c.bin & 63u;
                o.slot = __builtin_popcountll(lp.need[5] & ((1ull << cls) - 1ull));
                o.iq_row = c.iq_row;
                tw_signed(c.bin, 7, o.w7x, o.w7y);
                tw_signed(c.bin, 8, o.w8x, o.w8y);
                tw_signed(c.bin, 9, o.w9x, o.w9y);
            }
            p.l64_chan_full = p.l64_chan;
            for (int i = 0; i < nch; ++i)
                p.l64_chan_full[static_cast<size_t>(i)].slot = static_cast<int>(p.cp[static_cast<size_t>(i)].bin & 63u);
            lp.enabled = 1;
        }
    }
    return MI_OK;
}

}  // namespace mi
