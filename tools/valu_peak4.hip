// Microbenchmark: v_cndmask / literal / compare variants (gfx950), 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int MODE>
__global__ __launch_bounds__(64) void k(float *out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5;
    asm volatile("s_mov_b64 vcc, 0x5555\n s_mov_b64 s[20:21], 0x3333" ::: "vcc", "s20", "s21");
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {   // cndmask, vcc set once
            REP16(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4));)
        } else if (MODE == 1) {   // e64 with an sgpr pair
            REP16(asm volatile("v_cndmask_b32 %0, %0, %4, s[20:21]\n v_cndmask_b32 %1, %1, %4, s[20:21]\n v_cndmask_b32 %2, %2, %4, s[20:21]\n v_cndmask_b32 %3, %3, %4, s[20:21]"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4));)
        } else if (MODE == 2) {   // cmp -> cndmask pairs, as compiled selects look
            REP16(asm volatile("v_cmp_gt_f32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %5, vcc\n v_cmp_gt_f32 vcc, %1, %4\n v_cndmask_b32 %1, %1, %5, vcc\n"
                               "v_cmp_gt_f32 vcc, %2, %4\n v_cndmask_b32 %2, %2, %5, vcc\n v_cmp_gt_f32 vcc, %3, %4\n v_cndmask_b32 %3, %3, %5, vcc"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5) : "vcc");)
        } else if (MODE == 3) {   // mul by literal
            REP16(asm volatile("v_mul_f32 %0, 0x3f7d70a4, %0\n v_mul_f32 %1, 0x3f7d70a4, %1\n v_mul_f32 %2, 0x3f7d70a4, %2\n v_mul_f32 %3, 0x3f7d70a4, %3"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (MODE == 4) {   // mul by inline constant
            REP16(asm volatile("v_mul_f32 %0, 0.5, %0\n v_mul_f32 %1, 0.5, %1\n v_mul_f32 %2, 0.5, %2\n v_mul_f32 %3, 0.5, %3"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (MODE == 5) {   // max
            REP16(asm volatile("v_max_f32 %0, %0, %4\n v_max_f32 %1, %1, %4\n v_max_f32 %2, %2, %4\n v_max_f32 %3, %3, %4"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4));)
        } else if (MODE == 6) {   // med3
            REP16(asm volatile("v_med3_f32 %0, %0, %4, %5\n v_med3_f32 %1, %1, %4, %5\n v_med3_f32 %2, %2, %4, %5\n v_med3_f32 %3, %3, %4, %5"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5));)
        } else if (MODE == 7) {   // cmp_class / cmp to vcc + s_and on the result (ballot style)
            REP16(asm volatile("v_cmp_gt_f32 vcc, %0, %1\n s_cmp_eq_u64 vcc, 0\n v_cmp_gt_f32 vcc, %1, %2\n s_cmp_eq_u64 vcc, 0\n v_cmp_gt_f32 vcc, %2, %3\n s_cmp_eq_u64 vcc, 0\n v_cmp_gt_f32 vcc, %3, %0\n s_cmp_eq_u64 vcc, 0"
                         :: "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc", "scc");)
        } else if (MODE == 8) {   // v_add_u32
            REP16(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4));)
        } else if (MODE == 9) {   // fma f32 x4 baseline
            REP16(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5));)
        } else if (MODE == 10) {  // v_mac/fmac (VOP2)
            REP16(asm volatile("v_fmac_f32 %0, %4, %5\n v_fmac_f32 %1, %4, %5\n v_fmac_f32 %2, %4, %5\n v_fmac_f32 %3, %4, %5"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5));)
        } else if (MODE == 11) {  // v_rcp
            REP16(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3;
}
template <int MODE> void run(const char *name, float *out, int per) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000, blocks = 256 * 4 * 4;
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        (void)hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, iters); (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
    }
    (void)hipEventElapsedTime(&ms, e0, e1);
    double instr = (double)blocks * iters * 16 * per;
    printf("%-34s %.2f cycles/instr/SIMD @2.4GHz\n", name, 1024.0 * 2.4e9 * ms * 1e-3 / instr);
}
int main() {
    float *out; (void)hipMalloc(&out, 256 * 32 * 64 * 4 * 2);
    run<9>("v_fma_f32", out, 4); run<10>("v_fmac_f32", out, 4); run<0>("v_cndmask vcc (fixed)", out, 4); run<1>("v_cndmask s[20:21] (fixed)", out, 4);
    run<2>("cmp+cndmask (per instr)", out, 8); run<3>("v_mul_f32 literal", out, 4); run<4>("v_mul_f32 inline const", out, 4);
    run<5>("v_max_f32", out, 4); run<6>("v_med3_f32", out, 4); run<7>("v_cmp + s_cmp (per instr)", out, 8); run<8>("v_add_u32", out, 4);
    run<11>("v_rcp_f32", out, 4);
    return 0;
}
