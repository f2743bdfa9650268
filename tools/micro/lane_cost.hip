// Latency of the vector <-> scalar round trips a guess-and-verify round is made of, on a lone wave (ns per unit of a dependent chain):
//   A  v_add_u32 (dependent)
//   B  v_readlane (constant lane) -> v_add using the SGPR
//   C  v_cmp -> s_ff1 -> v_readlane (lane from the SGPR) -> v_add
//   D  C with an s_and_b64 between compare and find-first
//   E  v_readlane -> s_sub -> v_add
//   F  four dependent float operations (min, mul, add, add)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/lane_cost tools/micro/lane_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define X8(U) U U U U U U U U
__global__ void k(unsigned* out, long long* t, int iters) {
    unsigned v = threadIdx.x, w = threadIdx.x == 37 ? 1u : 0u;
    float f = 0.18f + threadIdx.x * 1e-6f, h = 10.0f, k97 = 0.97f, k03 = 0.03f, tmp;
    const unsigned long long all = ~0ull;
    long long t0, t1;
#define TIME(slot, ...)                                   \
    t0 = wall_clock64();                                  \
    for (int i = 0; i < iters; ++i) asm volatile(__VA_ARGS__); \
    t1 = wall_clock64();                                  \
    if (threadIdx.x == 0) t[slot] = t1 - t0;
    TIME(0, X8("v_add_u32 %0, %0, %1\n") : "+v"(v) : "v"(w))
    TIME(1, X8("v_readlane_b32 s20, %0, 5\n v_add_u32 %0, s20, %0\n") : "+v"(v) : "v"(w) : "s20")
    TIME(2, X8("v_cmp_ne_u32 vcc, 0, %1\n s_ff1_i32_b64 s20, vcc\n s_nop 3\n v_readlane_b32 s21, %0, s20\n v_add_u32 %0, s21, %0\n") : "+v"(v) : "v"(w) : "s20", "s21", "vcc")
    TIME(3, X8("v_cmp_ne_u32 vcc, 0, %1\n s_and_b64 vcc, vcc, %2\n s_ff1_i32_b64 s20, vcc\n s_nop 3\n v_readlane_b32 s21, %0, s20\n v_add_u32 %0, s21, %0\n") : "+v"(v) : "v"(w), "s"(all) : "s20", "s21", "vcc", "scc")
    TIME(4, X8("v_readlane_b32 s20, %0, 5\n s_sub_i32 s20, s20, 3\n v_add_u32 %0, s20, %0\n") : "+v"(v) : "v"(w) : "s20", "scc")
    TIME(5, X8("v_min_f32 %1, %2, %0\n v_mul_f32 %1, %4, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, 0x358637bd, %0\n") : "+v"(f), "=&v"(tmp) : "v"(h), "v"(k97), "v"(k03))
    // G: the compare made to depend on the chain value (as in the round): v_sub -> v_cmp -> s_ff1 -> readlane -> v_add
    TIME(6, X8("v_sub_u32 %1, %0, %0\n v_cmp_eq_u32 vcc, 0, %1\n s_ff1_i32_b64 s20, vcc\n s_nop 3\n v_readlane_b32 s21, %0, s20\n v_add_u32 %0, s21, %0\n") : "+v"(v), "+v"(w) : : "s20", "s21", "vcc")
    // R: the lean round (tools/round_variants.inc, K = 4) in a straight line, pieces of it taken out one at a time
    {
        const unsigned P = threadIdx.x * 67u, pinc = threadIdx.x == 63 ? ~0u : 67u, P1 = P + 67u;
        unsigned base = 0x3e386c22u, g, ob, d, tm, last, kk = 0, dl;
        float tt, m, vnf = 0.f;
        const float op = 10.0f;
#define R_HEAD "v_add_u32 %[g], %[base], %[P]\n v_min_f32 %[m], %[op], %[g]\n v_mul_f32 %[t], %[k97], %[g]\n v_mul_f32 %[m], %[k03], %[m]\n v_add_f32 %[t], %[t], %[m]\n v_add_f32 %[ob], 0x358637bd, %[t]\n v_sub_u32 %[d], %[ob], %[g]\n v_cmp_ne_u32 vcc, %[d], %[pinc]\n v_mov_b32 %[vnf], %[ob]\n v_sub_u32 %[tmp], %[ob], %[P1]\n s_ff1_i32_b64 %[last], vcc\n s_add_i32 %[kk], %[last], 1\n"
#define R_EXEC "s_lshl_b64 exec, -1, 0\n"
#define R_TAIL "s_cmp_eq_u32 %[last], 63\n s_nop 0\n v_readlane_b32 %[base], %[tmp], %[last]\n"
#define R_EARLY "v_readlane_b32 %[dl], %[d], %[last]\n v_cmp_lt_f32 vcc, %[op], %[g]\n s_sub_i32 s20, %[dl], 66\n s_cmp_le_u32 s20, 2\n s_cbranch_scc0 9f\n 9:\n s_bitcmp1_b64 vcc, %[kk]\n s_cbranch_scc1 8f\n 8:\n"
#define R_OPS : [vnf] "+v"(vnf), [ob] "=&v"(ob), [d] "=&v"(d), [g] "=&v"(g), [tmp] "=&v"(tm), [t] "=&v"(tt), [m] "=&v"(m), [base] "+s"(base), [last] "=&s"(last), [dl] "=&s"(dl), [kk] "+s"(kk) \
              : [P] "v"(P), [P1] "v"(P1), [pinc] "v"(pinc), [op] "v"(op), [k97] "s"(k97), [k03] "s"(k03) : "vcc", "scc", "s20"
        TIME(7, X8(R_HEAD R_EXEC R_TAIL) R_OPS)
        TIME(8, X8(R_HEAD "s_nop 0\n" R_TAIL) R_OPS)
        TIME(9, X8(R_HEAD R_EXEC R_TAIL R_EARLY) R_OPS)
        TIME(10, X8(R_HEAD "s_nop 0\n" R_TAIL R_EARLY) R_OPS)
        v += ob + d + __float_as_uint(vnf) + dl;
    }
    out[threadIdx.x] = v + __float_as_uint(f);
}
int main() {
    unsigned* o; long long* t;
    (void)hipMalloc(&o, 256); (void)hipMalloc(&t, 128);
    const int iters = 20000;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, t, iters); (void)hipDeviceSynchronize(); }
    long long h[11]; (void)hipMemcpy(h, t, 88, hipMemcpyDeviceToHost);
    const char* n[11] = {"A v_add_u32", "B v_readlane(const) -> v_add", "C v_cmp -> s_ff1 -> v_readlane(s) -> v_add", "D v_cmp -> s_and -> s_ff1 -> v_readlane(s) -> v_add",
                        "E v_readlane -> s_sub -> v_add", "F min, mul, add, add (float, dependent)", "G v_sub -> v_cmp -> s_ff1 -> v_readlane(s) -> v_add",
                        "R lean round: chain + exec write + next base", "R without the exec write", "R with the early-end tests (branches not taken)", "R early-end tests, no exec write"};
    for (int i = 0; i < 11; ++i) printf("%-56s %7.1f ns per unit\n", n[i], h[i] * 10.0 / (8.0 * iters));
    return 0;
}
