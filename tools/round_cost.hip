// Cost of one guess-and-verify round of the noise-floor walk (tp.hip, nf_chain_guess64) on a lone wave, and of its pieces.
// hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/round_cost.hip -o tools/round_cost && tools/round_cost
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ __forceinline__ float noise_floor_step(const float nf, const float c) {
    const float nfac = static_cast<float>(1.0 - static_cast<double>(0.97f));
    const float m = __builtin_fminf(c, nf);
    return nf * 0.97f + m * nfac + 1e-6f;
}
__device__ __forceinline__ uint32_t wave_shr1_u32(const uint32_t v, const uint32_t first) {
    return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(first), static_cast<int>(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ uint32_t wave_shl1_u32(const uint32_t v, const uint32_t last) {
    return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(last), static_cast<int>(v), 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ uint32_t mul24(const uint32_t x, const uint32_t y) { return __umul24(x, y); }

#include "round_variants.inc"

// operands: `dip_per_1024` of 1024 blocks lie below the floor (in runs of 1-3), the others far above it
__device__ __forceinline__ float operand_of(const int g, const int lane, const float nf, const int dip_per_1024) {
    uint32_t h = static_cast<uint32_t>(g) * 64u + static_cast<uint32_t>(lane);
    const uint32_t run = h >> 1;  // pairs of blocks share their draw: runs
    uint32_t x = run * 2654435761u;
    x ^= x >> 15, x *= 2246822519u, x ^= x >> 13;
    return (x & 1023u) < static_cast<uint32_t>(dip_per_1024) ? nf * (0.5f + (x >> 22) * (0.45f / 1024.f)) : 10.0f;
}

template <int K>
__device__ __forceinline__ float call_variant(float& nf, const float op, uint32_t& dh, NfGuess3& gs, const NfLanes3& lc, const int lane, int& rounds) {
    if constexpr (K == 3)
        return guess3(nf, op, gs, lc, lane, rounds);
    else if constexpr (K == 4)
        return guess4(nf, op, gs, lc, lane, rounds);
    else
        return variant<K>(nf, op, dh, lane, rounds);
}

template <int K, bool CHECK>
__global__ __launch_bounds__(64) void k(float* out, int* rounds_out, long long* ticks, int* wrong, int ngroups, int dips) {
    const int lane = threadIdx.x;
    float nf = 0.1801f;
    uint32_t dh = 0;
    NfGuess3 gs = {0u, 0u, 0u, 0u};
    const NfLanes3 lc = nf_lanes3(lane);
    int rounds = 0, nwrong = 0, dummy = 0;
    float acc = 0.f;
    const long long t0 = wall_clock64();
    for (int g = 0; g < ngroups; ++g) {
        const float op = operand_of(g, lane, nf, dips);
        float nf_ref = nf;
        uint32_t dh_ref = 0;
        const float v = call_variant<K>(nf, op, dh, gs, lc, lane, rounds);
        acc += v;
        if (CHECK) {
            const float r = variant<2>(nf_ref, op, dh_ref, lane, dummy);
            nwrong += __popcll(__ballot(__float_as_uint(r) != __float_as_uint(v))) + (__float_as_uint(nf_ref) != __float_as_uint(nf));
        }
        if (nf > 0.24f)
            nf = 0.1801f + g * 1e-9f;
    }
    const long long t1 = wall_clock64();
    out[lane] = acc;
    if (lane == 0)
        *rounds_out = rounds, *ticks = t1 - t0, *wrong = nwrong;
}

template <int K>
void run(const char* what, const int dips) {
    float* d;
    int *r, *w;
    long long* t;
    hipMalloc(&d, 256), hipMalloc(&r, 4), hipMalloc(&t, 8), hipMalloc(&w, 4);
    const int ng = 20000;
    int rounds, wrong;
    long long ticks;
    hipLaunchKernelGGL((k<K, true>), dim3(1), dim3(64), 0, 0, d, r, t, w, ng, dips);
    hipDeviceSynchronize();
    hipMemcpy(&wrong, w, 4, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL((k<K, false>), dim3(1), dim3(64), 0, 0, d, r, t, w, ng, dips);
    hipDeviceSynchronize();
    hipMemcpy(&rounds, r, 4, hipMemcpyDeviceToHost), hipMemcpy(&ticks, t, 8, hipMemcpyDeviceToHost);
    printf("%-44s dips %3d/1024: %6.1f ns per group, %5.2f rounds per group, %6.1f ns per round, values off the systolic walk: %d\n", what, dips,
           ticks * 10.0 / ng, (double)rounds / ng, ticks * 10.0 / rounds, wrong);
    hipFree(d), hipFree(r), hipFree(t), hipFree(w);
}

int main() {
    for (const int dips : {0, 10, 20, 40}) {
        run<0>("as built in round 2 (C++)", dips);
        run<1>("round 3 (cheaper bookkeeping)", dips);
        run<3>("lean round (prefix vector, no DPP)", dips);
        run<4>("lean round in assembly", dips);
        run<2>("systolic passes (64 x NF_PASS_MIN)", dips);
    }
    return 0;
}
