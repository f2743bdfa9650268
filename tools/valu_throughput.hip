// Issue throughput of the VALU / LDS operations stage 1 is made of, with 1, 2 and 4 waves per SIMD on every CU.
// hipcc -O3 --offload-arch=gfx950 tools/valu_throughput.hip -o tools/valu_throughput && tools/valu_throughput
// Prints shader cycles (s_memtime) per wave-instruction per SIMD: 2.0 would be a 32-lane datapath fully used.
#include <hip/hip_runtime.h>
#include <cstdio>

#define N_IT 4000
#define CH 8  // independent chains per lane

template <int K>
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* ticks, float a, float b) {
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 x[CH];
    __shared__ float2 sh[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x)
        sh[i] = make_float2(a, b);
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CH; ++c)
        x[c] = v2{a + c + threadIdx.x * 1e-9f, b - c};
    const v2 w = {a, b};
    const unsigned ld = (threadIdx.x * 8u) & 32767u;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < N_IT; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (K == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[c].x) : "v"(a), "v"(b));
            if (K == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x[c]) : "v"(w));
            if (K == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x[c]) : "v"(w));
            if (K == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[c]) : "v"(w));
            if (K == 4) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[c].x) : "v"(a));
            if (K == 5) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[c].x) : "v"(a));
            if (K == 6) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "+v"(x[c]) : "v"(w));
            if (K == 7) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(x[c].x));
            if (K == 8) asm volatile("v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(x[c].x) : "v"(2));
            if (K == 9) asm volatile("ds_read_b64 %0, %1 offset:%c2" : "=v"(x[c]) : "v"(ld), "i"(c * 512));
            if (K == 10) asm volatile("ds_read_b32 %0, %1 offset:%c2" : "=v"(x[c].x) : "v"(ld), "i"(c * 512));
            if (K == 11) asm volatile("ds_write_b64 %1, %0 offset:%c2" : : "v"(x[c]), "v"(ld), "i"(c * 512));
            if (K == 12) asm volatile("ds_read_u16 %0, %1 offset:%c2" : "=v"(x[c].x) : "v"(ld), "i"(c * 512));
            if (K == 13) asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(*reinterpret_cast<float4*>(&x[c & ~1])) : "v"(ld * 2u & 32767u), "i"(c * 1024));
            if (K == 14) asm volatile("v_sqrt_f32 %0, %0" : "+v"(x[c].x));
        }
        if (K >= 9 && K <= 13)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c)
        s += x[c].x + x[c].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + sh[threadIdx.x].x;
    if (threadIdx.x == 0)
        ticks[blockIdx.x] = t1 - t0;
}

template <int K>
void run(const char* name, int threads) {
    const int blocks = 256;
    float* d;
    unsigned long long* t;
    hipMalloc(&d, blocks * 1024 * 4);
    hipMalloc(&t, blocks * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    k<K><<<blocks, threads>>>(d, t, 1.0f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<K><<<blocks, threads>>>(d, t, 1.0f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256];
    hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
    double avg = 0;
    for (int i = 0; i < blocks; ++i)
        avg += h[i];
    avg /= blocks;
    const double wave_instr_per_simd = double(N_IT) * CH * (threads / 64) / 4.0;
    // s_memtime ticks at a fixed 100 MHz on gfx9 (REFCLK); wall time from events gives ns, the clock is unknown: report both
    printf("%-34s %2d waves/SIMD: %7.3f ms, %6.2f ns per wave-instr per SIMD (%5.2f cycles at 2.4 GHz); memtime ticks %.0f\n", name, threads / 256, ms,
           ms * 1e6 / wave_instr_per_simd, ms * 1e6 / wave_instr_per_simd * 2.4, avg);
    hipFree(d);
    hipFree(t);
}

#define RUN3(K, NAME) run<K>(NAME, 256), run<K>(NAME, 512), run<K>(NAME, 1024)
int main() {
    RUN3(0, "v_fma_f32");
    RUN3(1, "v_pk_fma_f32");
    RUN3(2, "v_pk_mul_f32");
    RUN3(3, "v_pk_add_f32");
    RUN3(4, "v_mul_f32");
    RUN3(5, "v_add_f32");
    RUN3(6, "v_pk_add_f32 op_sel+neg");
    RUN3(7, "v_cvt_f32_ubyte1");
    RUN3(8, "v_lshlrev_b32_sdwa");
    RUN3(9, "ds_read_b64");
    RUN3(10, "ds_read_b32");
    RUN3(11, "ds_write_b64");
    RUN3(12, "ds_read_u16");
    RUN3(13, "ds_read_b128");
    RUN3(14, "v_sqrt_f32");
    return 0;
}
