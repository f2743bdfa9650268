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
#define WSHR " wave_shr:1 row_mask:0xf bank_mask:0xf\n"
            if (K == 10) {  // tp.hip NF_PASS_MIN: the systolic noise-floor pass
                float m = 0.f, t = 0.f, b2;
                asm volatile("s_nop 1\n v_min_f32_dpp %1, %0, %4" WSHR "v_mul_f32_dpp %2, %0, %5" WSHR "v_mul_f32 %3, %6, %1\n v_add_f32 %0, %2, %3\n v_add_f32 %0, 0x358637bd, %0"
                             : "+v"(x), "+v"(m), "+v"(t), "=&v"(b2) : "v"(c), "v"(a), "v"(b));
            }
            if (K == 11) {  // the same with one shifted copy and plain operations after it
                float m = 0.f, t, b2;
                asm volatile("s_nop 1\n v_mov_b32_dpp %1, %0" WSHR "v_min_f32 %3, %1, %4\n v_mul_f32 %2, %1, %5\n v_mul_f32 %3, %6, %3\n v_add_f32 %0, %2, %3\n v_add_f32 %0, 0x358637bd, %0"
                             : "+v"(x), "+v"(m), "=&v"(t), "=&v"(b2) : "v"(c), "v"(a), "v"(b));
            }
            if (K == 12) {  // demod.hip MI_EMA_PASS: two interleaved moving averages
                float T = 0.f, P = 0.f, V;
                asm volatile("v_mul_f32_dpp %2, %0, %5" WSHR "v_add_f32 %0, %2, %6\n v_mov_b32_dpp %3, %1" WSHR
                             "v_mul_f32 %4, %5, %3\n v_cmp_ge_f32 vcc, %3, %7\n v_add_f32 %4, %4, %6\n v_min_f32 %4, %7, %4\n v_cndmask_b32 %1, %4, %7, vcc"
                             : "+v"(x), "+v"(y), "+v"(T), "+v"(P), "=&v"(V) : "v"(a), "v"(b), "v"(c) : "vcc");
            }
            if (K == 13) {  // bare EMA pass: mul_dpp, add, nop
                float T = 0.f;
                asm volatile("v_mul_f32_dpp %1, %0, %2" WSHR "v_add_f32 %0, %1, %3\n s_nop 1" : "+v"(x), "+v"(T) : "v"(a), "v"(c));
            }
            if (K == 14) {  // plain dependent mul, add (the stepped EMA)
                asm volatile("v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %2" : "+v"(x) : "v"(a), "v"(c));
            }
        }
    }
    out[threadIdx.x] = x + y;
}

// the same dependent add with only the first `lanes` lanes switched on: does the SIMD skip idle quarters of the wave?
template <int LANES>
__global__ __launch_bounds__(64) void k_exec(float* out, float a, float c) {
    float x = a + threadIdx.x * 1e-9f;
    if (threadIdx.x < LANES) {
        for (int i = 0; i < N_IT; ++i) {
#pragma unroll
            for (int r = 0; r < REP; ++r)
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(c));
        }
    }
    out[threadIdx.x] = x;
}
// a taken scalar branch per dependent add
__global__ __launch_bounds__(64) void k_branch(float* out, float a, float c) {
    float x = a + threadIdx.x * 1e-9f;
    const unsigned long long t0 = __builtin_readcyclecounter();  // s_memtime: calibrates "cycles" against the event time
    for (int i = 0; i < N_IT * REP; ++i)
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(c));
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0)
        reinterpret_cast<unsigned long long*>(out + 32)[0] = t1 - t0;
}

template <typename F>
void run_fn(const char* name, int ops, F launch) {
    float* d;
    hipMalloc(&d, 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    launch(d);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch(d);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / (double(N_IT) * REP);
    unsigned long long ticks = 0;
    hipMemcpy(&ticks, d + 32, 8, hipMemcpyDeviceToHost);
    printf("%-28s %7.2f ns per group (%d ops) = %6.2f ns/op   [s_memtime ticks in kernel %llu over %.3f ms = %.3f per ns]\n", name, ns, ops, ns / ops, ticks, ms,
           ticks / (ms * 1e6));
    hipFree(d);
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
    run<10>("nf pass (dpp-fused, 5+nop)", 5);
    run<11>("nf pass (mov_dpp + plain, 6+nop)", 6);
    run<12>("ema pass (2 chains, 8)", 8);
    run<13>("bare ema pass (2+nop)", 2);
    run<14>("plain mul,add dep", 2);
    run_fn("v_add dep, 1 lane on", 1, [](float* d) { k_exec<1><<<1, 64>>>(d, 1.0f, 1.0000001f); });
    run_fn("v_add dep, 16 lanes on", 1, [](float* d) { k_exec<16><<<1, 64>>>(d, 1.0f, 1.0000001f); });
    run_fn("v_add dep, 32 lanes on", 1, [](float* d) { k_exec<32><<<1, 64>>>(d, 1.0f, 1.0000001f); });
    run_fn("v_add dep + taken branch", 1, [](float* d) { k_branch<<<1, 64>>>(d, 1.0f, 1.0000001f); });
    return 0;
}
