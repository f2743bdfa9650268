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

template <int K>
__global__ __launch_bounds__(64) void k(float* out, int* rounds_out, long long* ticks, int ngroups, float opv) {
    const int lane = threadIdx.x;
    float nf = 0.1801f;
    uint32_t dh = 0;
    int rounds = 0;
    float acc = 0.f;
    const long long t0 = wall_clock64();
    for (int g = 0; g < ngroups; ++g) {
        const float op = opv;  // all above the floor: self steps
        acc += variant<K>(nf, op, dh, lane, rounds);
        if (nf > 0.24f)
            nf = 0.1801f + g * 1e-9f;
    }
    const long long t1 = wall_clock64();
    out[lane] = acc;
    if (lane == 0)
        *rounds_out = rounds, *ticks = t1 - t0;
}

template <int K>
void run(const char* what) {
    float* d;
    int* r;
    long long* t;
    hipMalloc(&d, 256), hipMalloc(&r, 4), hipMalloc(&t, 8);
    const int ng = 20000;
    hipLaunchKernelGGL(k<K>, dim3(1), dim3(64), 0, 0, d, r, t, ng, 10.0f);
    hipDeviceSynchronize();
    int rounds;
    long long ticks;
    hipMemcpy(&rounds, r, 4, hipMemcpyDeviceToHost), hipMemcpy(&ticks, t, 8, hipMemcpyDeviceToHost);
    printf("%-60s %6.1f ns per group, %5.2f rounds per group, %6.1f ns per round\n", what, ticks * 10.0 / ng, (double)rounds / ng, ticks * 10.0 / rounds);
    hipFree(d), hipFree(r), hipFree(t);
}

int main() {
    run<0>("as built (C++)");
    run<1>("hand-ordered round");
    run<2>("systolic passes (64 x NF_PASS_MIN)");
    return 0;
}
