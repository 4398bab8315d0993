// Microbenchmark: issue cost of v_pk_fma_f32, DPP adds, v_cmp+v_cndmask at 8 waves/SIMD (gfx950).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float float2v __attribute__((ext_vector_type(2)));
#define REP16(x) x x x x x x x x x x x x x x x x
template <int MODE>
__global__ __launch_bounds__(64) void k(float *out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const float2v pb = {1.0001f, 1.0002f}, pc = {0.5f, 0.25f};
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {        // 8 pk_fma
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));)
        } else if (MODE == 1) { // dpp row_shr adds (independent chains)
            REP16(asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n"
                         "v_add_f32_dpp %2, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 row_shr:8 row_mask:0xf bank_mask:0xf"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (MODE == 2) { // row_bcast
            REP16(asm volatile("v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
                         "v_add_f32_dpp %2, %2, %2 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 row_bcast:31 row_mask:0xc bank_mask:0xf"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (MODE == 3) { // v_cmp to sgpr pair + cndmask
            REP16(asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc\n v_cmp_gt_f32 vcc, %1, %2\n v_cndmask_b32 %1, %1, %3, vcc"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) :: "vcc");)
        } else if (MODE == 4) { // plain fma x4
            REP16(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5));)
        } else if (MODE == 5) { // v_mov
            REP16(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (MODE == 6) { // v_rcp
            REP16(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (MODE == 7) { // v_cmp writing sgpr (e64) + s_and + branch-free
            REP16(asm volatile("v_cmp_gt_f32 s[20:21], %0, %1\n v_cmp_gt_f32 s[22:23], %1, %2\n v_cmp_gt_f32 s[24:25], %2, %3\n v_cmp_gt_f32 s[26:27], %3, %0"
                         :: "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "s20","s21","s22","s23","s24","s25","s26","s27");)
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + p0.x + p1.y + p2.x + p3.y;
}
template <int MODE> void run(const char *name, float *out) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 5000, blocks = 256 * 4 * 8;
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        (void)hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, iters); (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
    }
    (void)hipEventElapsedTime(&ms, e0, e1);
    double instr = (double)blocks * iters * 16 * 4;
    printf("%-28s %.2f cycles/instr/SIMD @2.4GHz\n", name, 1024.0 * 2.4e9 * ms * 1e-3 / instr);
}
int main() {
    float *out; (void)hipMalloc(&out, 256 * 32 * 64 * 4 * 2);
    run<4>("v_fma_f32", out); run<0>("v_pk_fma_f32", out); run<1>("v_add_f32_dpp row_shr", out); run<2>("v_add_f32_dpp row_bcast", out);
    run<3>("v_cmp(vcc)+v_cndmask avg", out); run<5>("v_mov_b32", out); run<6>("v_rcp_f32", out); run<7>("v_cmp_e64 -> sgpr", out);
    return 0;
}
