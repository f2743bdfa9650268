// iqgen.hpp -- synthetic u8 IQ, SURVEY 8(d): integer-only and counter-based, so any sample of any
// stream can be produced independently (host loop or one GPU lane per sample) with identical bytes.
//
//   noise   : 64-bit mix of (seed, stream, n) split into four 16-bit uniforms per component, summed
//             (Irwin-Hall, sigma = 65536*sqrt(4/12)), scaled by noise_q8_mul/8192 -> Q8 LSBs
//   carrier : 32-bit phase = n * dphi (wraps), 1024-entry int16 sine table; AM 1 kHz / 50 % depth,
//             NFM 1 kHz tone with 2.5 kHz deviation (phase deviation beta*sin), optional 100 Hz CTCSS
//             at 375 Hz deviation; gated on/off every gate_samples
//   output  : clamp(round(127.5 + noise + sum carriers), 0, 255)
#pragma once
#include <cstdint>

#include "../../include/mi_airband.h"

#if defined(__HIPCC__)
#define MI_HD __host__ __device__
#else
#define MI_HD
#endif

namespace mi {

struct IqGenCarrierDerived {
    uint32_t dphi;      // carrier phase increment per sample (2^32 = one turn)
    uint32_t dpsi_1k;   // 1 kHz tone increment
    uint32_t dpsi_100;  // 100 Hz tone increment
    int32_t kind, amp_q8, gate_phase;
};

struct IqGenDerived {
    uint64_t seed;
    uint64_t gate_samples;
    int32_t noise_q8_mul;
    int32_t ncarriers;
    IqGenCarrierDerived c[64];
};

MI_HD inline uint64_t iq_mix64(uint64_t z) {  // splitmix64 finaliser
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

MI_HD inline int32_t iq_noise_q8(uint64_t h, int32_t mul) {
    const int32_t s = static_cast<int32_t>(h & 0xffff) + static_cast<int32_t>((h >> 16) & 0xffff) + static_cast<int32_t>((h >> 32) & 0xffff) +
                      static_cast<int32_t>((h >> 48) & 0xffff) - 131070;
    return static_cast<int32_t>((static_cast<int64_t>(s) * mul) >> 13);
}

// sin/cos from the int16 table (value 32767 = 1.0); phase is a 32-bit turn fraction
MI_HD inline int32_t iq_sin(const int16_t* tab, uint32_t phase) {
    return tab[phase >> 22];
}
MI_HD inline int32_t iq_cos(const int16_t* tab, uint32_t phase) {
    return tab[((phase + 0x40000000u) >> 22) & 1023];
}

// one complex sample -> two bytes
MI_HD inline void iq_sample(const IqGenDerived& g, const int16_t* tab, uint32_t stream, uint64_t n, uint8_t* out2) {
    const uint64_t key = g.seed ^ (static_cast<uint64_t>(stream) * 0x9E3779B97F4A7C15ull);
    const uint64_t hi = iq_mix64(key + 2 * n);
    const uint64_t hq = iq_mix64(key + 2 * n + 1);
    int64_t vi = 32640 + iq_noise_q8(hi, g.noise_q8_mul);  // 127.5 in Q8
    int64_t vq = 32640 + iq_noise_q8(hq, g.noise_q8_mul);
    const uint32_t n32 = static_cast<uint32_t>(n);
    for (int k = 0; k < g.ncarriers; ++k) {
        const IqGenCarrierDerived& c = g.c[k];
        if (g.gate_samples != 0 && (((n / g.gate_samples) + static_cast<uint64_t>(c.gate_phase)) & 1) == 0)
            continue;
        uint32_t phase = n32 * c.dphi;
        int64_t amp = c.amp_q8;  // Q8
        const int32_t tone = iq_sin(tab, n32 * c.dpsi_1k);
        if (c.kind == 0) {
            amp = (amp * (65534 + tone)) >> 16;  // 1 + 0.5 sin
        } else {
            // beta = 2.5 rad -> turns = 2.5/(2 pi) = 0.3979; in 2^32 units per unit sine (32767): 52154
            phase += static_cast<uint32_t>(static_cast<int64_t>(tone) * 52154);
            if (c.kind == 2)  // 375 Hz deviation at 100 Hz: beta = 3.75 rad -> 78231 per unit sine
                phase += static_cast<uint32_t>(static_cast<int64_t>(iq_sin(tab, n32 * c.dpsi_100)) * 78231);
        }
        vi += (amp * iq_cos(tab, phase)) >> 15;
        vq += (amp * iq_sin(tab, phase)) >> 15;
    }
    int64_t bi = (vi + 128) >> 8, bq = (vq + 128) >> 8;
    bi = bi < 0 ? 0 : (bi > 255 ? 255 : bi);
    bq = bq < 0 ? 0 : (bq > 255 ? 255 : bq);
    out2[0] = static_cast<uint8_t>(bi);
    out2[1] = static_cast<uint8_t>(bq);
}

// host: derive increments and the sine table (done once, on the host, also for the device path)
void iqgen_derive(const mi_iqgen_cfg& cfg, IqGenDerived& out);
const int16_t* iqgen_sine_table();  // 1024 entries

}  // namespace mi
