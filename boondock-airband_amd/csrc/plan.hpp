// plan.hpp -- host-side derived parameters of the hot path (everything the reference computes with
// libm in double before its demod thread starts).  Device code afterwards needs only + - * / sqrt,
// fma where the FFT spec says so, and compares.
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/mi_airband.h"
#include "l64_args.h"

namespace mi {

constexpr int kWaveRate = MI_WAVE_RATE;
constexpr int kWaveBatch = MI_WAVE_BATCH;
constexpr int kAgcExtra = MI_AGC_EXTRA;
constexpr int kMaxTones = 52;
constexpr int kSquelchRing = 102;  // Squelch::buffer_size_, squelch.cpp:67

// Per-channel constants, identical for every stream of a handle.  Plain 4-byte fields: the struct is
// copied verbatim into device memory.
struct ChanParams {
    uint32_t bin;
    uint32_t dm_dphi;
    int32_t modulation;
    int32_t needs_raw_iq;
    int32_t has_iq_outputs;
    int32_t iq_row;  // row of this channel in the complex-bin plane, -1 if !needs_raw_iq
    int32_t using_manual_level;
    float manual_signal_level;
    float normal_signal_ratio;
    float flappy_signal_ratio;
    float cap_factor;  // 1.5f * normal_signal_ratio (squelch.cpp:497 evaluates left to right)
    float manual_cap;  // 1.5f * manual_signal_level
    float ampfactor;
    float alpha;
    float one_minus_alpha;
    int32_t notch_enabled;
    float notch_d0, notch_d1, notch_d2;
    int32_t lowpass_enabled;
    float lowpass_gain, lowpass_yc0, lowpass_yc1;
    int32_t ctcss_enabled;
    int32_t ctcss_fast_window, ctcss_slow_window;
    int32_t ctcss_fast_ndet, ctcss_slow_ndet;
    int32_t ctcss_row;  // row in the detector coefficient/state tables, -1 if none
    uint32_t afc;       // channel_t.afc (0 = off .. 255), boondock_airband.h:258
};

// Pruning of the radix-2 DIT graph to the bins the channel plan picks (channelize.hip, N = 512).  After the radix-8 pass
// over stages 3K-2 .. 3K only the residues R_3K = { bin mod 8^K } of every block are needed, so the next pass has work for
// m_3K of every 8^K lanes only: its work items (block, residue) are packed densely onto the lanes, several windows per
// wave.  Every butterfly that is evaluated is the full graph's butterfly, so the picked bins keep their bits.
struct PrunePlan {
    int32_t enabled;      // 0: evaluate the full graph
    int32_t m3, sh3;      // |R_3| and log2 of the next power of two (items of pass 1 per window: 8 << sh3)
    int32_t m6, sh6;      // |R_6|                                    (items of pass 2 per window: 1 << sh6)
    int32_t m9;           // distinct picked bins
    int32_t rs1, rs2;     // row strides (float2) of the two exchange buffers
    int32_t rank3[8];     // rank of residue r in R_3, or -1 when pass 0 need not keep it
};
// Per work-item class (j = index into R_3 for pass 1, into R_6 for pass 2), 24 floats: 7 twiddles (x, y) in the order of
// Pass::run, 8 ints (as bits): where output ri of the radix-8 goes in the next buffer, or -1 when no picked bin needs it,
// 2 unused.
constexpr int kPruneClassWords = 24;

// The lane-resident stage 1 (l64_kernel.h, N = 512): which residues { bin mod 2^s } are live after stage s = 1..6
// (bit r of need[s-1]).  L64Chan (l64_args.h) tells the combining step where a channel's class sits and which twiddles
// lead to its bin.
struct L64Plan {
    int32_t enabled;
    int32_t m6;            // live classes mod 64
    int32_t nb_pad;        // channels per window padded to a power of two, 8 .. 64
    int32_t linear_tiles;  // (unused)
    int32_t wg_per_cu;     // (tuning, set by the caller) workgroups launched per CU, 0 = default
    uint64_t need[6];
};

struct Plan {
    mi_device_cfg dev{};
    std::vector<mi_channel_cfg> chans;
    int nch = 0;
    int log2n = 0;
    int fft_size = 0;
    int bytes_per_sample = 1;
    size_t hop_bytes = 0;  // "bps", rtl_airband.cpp:416
    int n_iq_rows = 0;     // channels with needs_raw_iq
    int n_ctcss_rows = 0;
    bool any_afc = false;  // some channel has afc > 0: bins follow the signal, batches are processed one at a time
    std::vector<float> window;        // fft_size
    std::vector<float> tw;            // fft_size/2 x {re, im}
    std::vector<float> levels;        // 256 (u8 or s8 LUT)
    bool conv_arith = false;          // u8: n * rcp with one fma correction equals the LUT for all 256 values
    float sin_lut[257], cos_lut[257];
    float conv_scale = 0.f;           // 1.0f / fullscale (s16, f32)
    std::vector<ChanParams> cp;       // nch
    std::vector<float> ctcss_coeff;   // n_ctcss_rows x 2 x kMaxTones  (fast, slow)
    PrunePlan prune{};                // stage-1 graph pruning (N = 512, no AFC, and only where it pays)
    std::vector<float> prune_t1;      // (1 << sh3) classes x kPruneClassWords
    std::vector<float> prune_t2;      // (1 << sh6) classes x kPruneClassWords
    std::vector<int32_t> prune_chan_rank;  // per channel: rank of its bin among the distinct picked bins
    L64Plan l64{};                    // lane-resident stage 1 (N = 512, no AFC, hop a supported multiple of 8, <= 64 channels)
    std::vector<L64Chan> l64_chan;       // nch: slots among the plan's live classes (the plan's own kernel instance)
    std::vector<L64Chan> l64_chan_full;  // nch: slot = bin mod 64 (the full-graph instance)
    float initial_noise_floor = 5.0f;
};

// returns MI_OK or a negative mi_status; msg receives the reason
int build_plan(const mi_device_cfg& dev, const mi_channel_cfg* chans, int nch, Plan& out, const char** msg);

}  // namespace mi
