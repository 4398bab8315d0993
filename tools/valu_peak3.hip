// Microbenchmark: issue cost of the non-FMA instruction classes the blend kernels use (gfx950),
// 4 waves per SIMD, four independent chains per wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int MODE>
__global__ __launch_bounds__(64) void k(float *out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {
            REP16(asm volatile("v_min_f32 %0, %0, %4\n v_min_f32 %1, %1, %4\n v_min_f32 %2, %2, %4\n v_min_f32 %3, %3, %4"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4));)
        } else if (MODE == 1) {
            REP16(asm volatile("v_sub_f32 %0, %0, %4\n v_sub_f32 %1, %1, %4\n v_sub_f32 %2, %2, %4\n v_sub_f32 %3, %3, %4"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4));)
        } else if (MODE == 2) {
            REP16(asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4));)
        } else if (MODE == 3) {   // cndmask with a fixed vcc
            REP16(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4));)
        } else if (MODE == 4) {   // v_cmp to vcc only
            REP16(asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cmp_gt_f32 vcc, %1, %2\n v_cmp_gt_f32 vcc, %2, %3\n v_cmp_gt_f32 vcc, %3, %0"
                         :: "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc");)
        } else if (MODE == 5) {
            REP16(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (MODE == 6) {   // exp interleaved with 3 fma: does the transcendental unit overlap?
            REP16(asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5));)
        } else if (MODE == 7) {   // cmp(vcc) interleaved with 3 fma
            REP16(asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5) : "vcc");)
        } else if (MODE == 8) {
            REP16(asm volatile("v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4));)
        } else if (MODE == 9) {   // fma with an SGPR operand
            REP16(asm volatile("v_fma_f32 %0, %0, s20, %5\n v_fma_f32 %1, %1, s20, %5\n v_fma_f32 %2, %2, s21, %5\n v_fma_f32 %3, %3, s21, %5"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5) : "s20", "s21");)
        } else if (MODE == 10) {  // readlane
            REP16(asm volatile("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 5\n v_readlane_b32 s22, %2, 7\n v_readlane_b32 s23, %3, 9"
                         :: "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "s20", "s21", "s22", "s23");)
        } else if (MODE == 11) {  // salu chain beside nothing
            REP16(asm volatile("s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1"
                         ::: "s20", "s21", "s22", "s23", "scc");)
        } else if (MODE == 12) {  // fma + salu interleaved 1:1
            REP16(asm volatile("v_fma_f32 %0, %0, %4, %5\n s_add_u32 s20, s20, 1\n v_fma_f32 %1, %1, %4, %5\n s_add_u32 s21, s21, 1\n v_fma_f32 %2, %2, %4, %5\n s_add_u32 s22, s22, 1\n v_fma_f32 %3, %3, %4, %5\n s_add_u32 s23, s23, 1"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5) : "s20", "s21", "s22", "s23", "scc");)
        } else if (MODE == 13) {  // one dependent fma chain (latency with 4 waves)
            REP16(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5));)
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3;
}
template <int MODE> void run(const char *name, float *out, int wg) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 3000, blocks = 256 * 4 * wg;
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        (void)hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, iters); (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
    }
    (void)hipEventElapsedTime(&ms, e0, e1);
    double instr = (double)blocks * iters * 16 * (MODE == 12 ? 8 : 4);
    printf("%-34s waves/SIMD %d: %.2f cycles/instr/SIMD @2.4GHz\n", name, wg, 1024.0 * 2.4e9 * ms * 1e-3 / instr);
}
int main() {
    float *out; (void)hipMalloc(&out, 256 * 32 * 64 * 4 * 2);
    for (int wg = 1; wg <= 4; wg *= 2) {
        run<0>("v_min_f32", out, wg); run<1>("v_sub_f32", out, wg); run<2>("v_mul_f32", out, wg); run<3>("v_cndmask_b32 (vcc)", out, wg);
        run<4>("v_cmp_gt_f32 vcc", out, wg); run<5>("v_exp_f32", out, wg); run<6>("exp + 3 fma (per instr)", out, wg);
        run<7>("cmp + 3 fma (per instr)", out, wg); run<8>("v_and_b32", out, wg); run<9>("v_fma_f32 sgpr operand", out, wg);
        run<10>("v_readlane_b32", out, wg); run<11>("s_add_u32", out, wg); run<12>("fma+s_add 1:1 (per instr)", out, wg);
        run<13>("dependent fma chain", out, wg);
    }
    return 0;
}
