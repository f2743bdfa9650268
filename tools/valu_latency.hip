// Dependent-issue latency of the VALU operations the serial squelch chains are made of (one wave, one CU).
// hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/valu_latency.hip -o tools/valu_latency && tools/valu_latency
#include <hip/hip_runtime.h>
#include <cstdio>

#define N_IT 200000
#define REP 16

template <int K>
__global__ __launch_bounds__(64) void k(float* out, float a, float b, float c) {
    float x = a + threadIdx.x * 1e-9f;
    float y = b;
    for (int i = 0; i < N_IT; ++i) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            if (K == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(c));
            if (K == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(c));
            if (K == 2) asm volatile("v_min_f32 %0, %0, %1" : "+v"(x) : "v"(c));
            if (K == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double*>(&x)) : "v"(*reinterpret_cast<double*>(&y)));
            if (K == 4) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n" : "+v"(x));
            if (K == 5) { int s; asm volatile("v_readlane_b32 %0, %1, 3\n v_add_f32 %1, %0, %1" : "=&s"(s), "+v"(x)); }
            if (K == 6) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(c));
            if (K == 7) {  // the noise-floor step: min, mul, mul, add, add
                float m, t;
                asm volatile("v_min_f32 %1, %0, %3\n v_mul_f32 %2, 0x3f7851ec, %0\n v_mul_f32 %1, 0x3cf5c290, %1\n v_add_f32 %0, %2, %1\n v_add_f32 %0, 0x358637bd, %0"
                             : "+v"(x), "=&v"(m), "=&v"(t) : "v"(c));
            }
            if (K == 8) {  // independent pair: two chains interleaved (issue rate)
                asm volatile("v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %2" : "+v"(x), "+v"(y) : "v"(c));
            }
            if (K == 9) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(c));
        }
    }
    out[threadIdx.x] = x + y;
}

template <int K>
void run(const char* name, int ops) {
    float* d;
    hipMalloc(&d, 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    k<K><<<1, 64>>>(d, 1.0f, 2.0f, 1.0000001f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<K><<<1, 64>>>(d, 1.0f, 2.0f, 1.0000001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / (double(N_IT) * REP);
    printf("%-28s %7.2f ns per group (%d ops) = %6.2f ns/op\n", name, ns, ops, ns / ops);
    hipFree(d);
}

int main() {
    int clk = 0;
    hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("clock %d kHz\n", clk);
    run<0>("v_add_f32 dep", 1);
    run<1>("v_mul_f32 dep", 1);
    run<2>("v_min_f32 dep", 1);
    run<3>("v_pk_mul_f32 dep", 1);
    run<4>("v_mov_dpp wave_shr dep", 1);
    run<5>("readlane+add dep", 2);
    run<6>("v_fma_f32 dep", 1);
    run<7>("noise floor step (4 deep)", 4);
    run<8>("2 independent adds", 2);
    run<9>("v_cndmask dep", 1);
    return 0;
}
