// Cost of control flow to a lone wave (one workgroup of 64 on an otherwise idle CU): per-iteration cycles (s_memtime) of
//   A  16 independent v_add                                    (the issue floor)
//   B  the same with 8 scalar branches that are NOT taken (fall through)
//   C  the same with 8 scalar branches TAKEN over 4 instructions each
//   D  8 s_and_saveexec + s_cbranch_execz regions, exec non-empty (fall through)
//   E  8 such regions with exec empty (branch taken)
//   F  8 taken branches over 64 instructions each (target in another cache line)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/branch_cost tools/micro/branch_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define ADD4 "v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n"
#define ADD16 ADD4 ADD4 ADD4 ADD4
#define ADD64 ADD16 ADD16 ADD16 ADD16
__global__ void k(float* out, unsigned long long* t, int iters, int zero) {
    float v = threadIdx.x, w = 1.0f;
    unsigned long long t0, t1;
    // A
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) asm volatile(ADD16 : "+v"(v) : "v"(w));
    t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) t[0] = t1 - t0;
    // B: s_cmp with a value that makes scc0 -> branch on scc1 not taken
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i)
        asm volatile(
            "s_cmp_lg_u32 %2, 0\n"
#define NT "s_cbranch_scc1 1f\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n 1:\n"
            NT NT NT NT NT NT NT NT
            : "+v"(v) : "v"(w), "s"(zero) : "scc");
    t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) t[1] = t1 - t0;
    // C: taken
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i)
        asm volatile(
            "s_cmp_eq_u32 %2, 0\n"
#define TK "s_cbranch_scc1 1f\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n 1:\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n"
            TK TK TK TK TK TK TK TK
            : "+v"(v) : "v"(w), "s"(zero) : "scc");
    t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) t[2] = t1 - t0;
    // D / E: saveexec regions
    for (int e = 0; e < 2; ++e) {
        const unsigned long long mask = e ? 0ull : ~0ull;
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; ++i)
            asm volatile(
#define SX "s_and_saveexec_b64 s[20:21], %2\n s_cbranch_execz 1f\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n 1:\n s_or_b64 exec, exec, s[20:21]\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n"
                SX SX SX SX SX SX SX SX
                : "+v"(v) : "v"(w), "s"(mask) : "s20", "s21", "scc");
        t1 = __builtin_readcyclecounter();
        if (threadIdx.x == 0) t[3 + e] = t1 - t0;
    }
    // F: taken over 64 instructions
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i)
        asm volatile(
            "s_cmp_eq_u32 %2, 0\n"
#define TF "s_cbranch_scc1 1f\n" ADD64 " 1:\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n"
            TF TF TF TF TF TF TF TF
            : "+v"(v) : "v"(w), "s"(zero) : "scc");
    t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) t[5] = t1 - t0;
    out[threadIdx.x] = v;
}
int main() {
    float* o; unsigned long long* t;
    hipMalloc(&o, 256); hipMalloc(&t, 64);
    const int iters = 20000;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, t, iters, 0); hipDeviceSynchronize(); }
    unsigned long long h[6]; hipMemcpy(h, t, 48, hipMemcpyDeviceToHost);
    const char* n[6] = {"A 16 v_add", "B 8 untaken scalar branches + 16 v_add", "C 8 taken scalar branches (skip 4) + 16 v_add", "D 8 saveexec regions entered (48 v_add)", "E 8 saveexec regions skipped (16 v_add)", "F 8 taken branches over 64 instr + 16 v_add"};
    for (int i = 0; i < 6; ++i) printf("%-48s %8.1f cycles per iteration\n", n[i], (double)h[i] / iters);
    return 0;
}
