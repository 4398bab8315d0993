// Microbenchmark: VALU issue rate on gfx950 (wave-instructions per SIMD-cycle) at 1..8 waves/SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(64) void fma_kernel(float *out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float b = 1.0001f, c = 0.5f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            a0 = __builtin_fmaf(a0, b, c); a1 = __builtin_fmaf(a1, b, c); a2 = __builtin_fmaf(a2, b, c); a3 = __builtin_fmaf(a3, b, c);
            a4 = __builtin_fmaf(a4, b, c); a5 = __builtin_fmaf(a5, b, c); a6 = __builtin_fmaf(a6, b, c); a7 = __builtin_fmaf(a7, b, c);
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
__global__ __launch_bounds__(64) void exp_kernel(float *out, int iters) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + .1f, a2 = a0 + .2f, a3 = a0 + .3f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            a0 = __builtin_amdgcn_exp2f(a0) - 1.0f; a1 = __builtin_amdgcn_exp2f(a1) - 1.0f;
            a2 = __builtin_amdgcn_exp2f(a2) - 1.0f; a3 = __builtin_amdgcn_exp2f(a3) - 1.0f;
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3;
}
int main() {
    float *out; hipMalloc(&out, 256 * 32 * 64 * 4 * 2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int wps = 1; wps <= 8; wps *= 2) {
        const int blocks = 256 * 4 * wps;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0); hipLaunchKernelGGL(fma_kernel, dim3(blocks), dim3(64), 0, 0, out, iters); hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double instr = (double)blocks * iters * 16 * 8;
        printf("fma  waves/SIMD %d: %.3f ms, %.3f Twave-instr/s, %.2f cycles/instr/SIMD @2.4GHz\n", wps, ms, instr / ms / 1e9,
               1024.0 * 2.4e9 * ms * 1e-3 / instr);
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0); hipLaunchKernelGGL(exp_kernel, dim3(blocks), dim3(64), 0, 0, out, iters / 4); hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        hipEventElapsedTime(&ms, e0, e1);
        instr = (double)blocks * (iters / 4) * 16 * 4;   // exp count (plus as many subs)
        printf("exp+sub waves/SIMD %d: %.3f ms, %.2f cycles per (exp,sub) pair per SIMD\n", wps, ms, 1024.0 * 2.4e9 * ms * 1e-3 / instr);
    }
    return 0;
}
