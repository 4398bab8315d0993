// Checks the 8-value wave64 reduction built from v_permlane32_swap / v_permlane16_swap / DPP row_shr
// and times the swap instructions (gfx950).
#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ void swap32(float &a, float &b) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void swap16(float &a, float &b) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float row_sum15(float v) {
    v = dpp_add<0x111>(v); v = dpp_add<0x112>(v); v = dpp_add<0x114>(v); v = dpp_add<0x118>(v);
    return v;
}
__global__ void k(const float *in, float *out) {
    const int lane = threadIdx.x;
    float v[8];
    for (int i = 0; i < 8; i++) v[i] = in[i * 64 + lane];
    swap32(v[0], v[1]); float w0 = v[0] + v[1];
    swap32(v[2], v[3]); float w1 = v[2] + v[3];
    swap32(v[4], v[5]); float w2 = v[4] + v[5];
    swap32(v[6], v[7]); float w3 = v[6] + v[7];
    swap16(w0, w1); float x0 = w0 + w1;
    swap16(w2, w3); float x1 = w2 + w3;
    x0 = row_sum15(x0); x1 = row_sum15(x1);
    out[lane] = x0; out[64 + lane] = x1;
}
__global__ __launch_bounds__(64) void tk(float *out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) { swap32(a0, a1); swap16(a2, a3); swap32(a1, a2); swap16(a3, a0); }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3;
}
int main() {
    float h[8 * 64], *d, *o, ho[128];
    for (int i = 0; i < 8; i++) for (int l = 0; l < 64; l++) h[i * 64 + l] = (float)((i + 1) * 1000 + l);
    (void)hipMalloc(&d, sizeof h); (void)hipMalloc(&o, 256 * 32 * 64 * 4 * 2);
    (void)hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
    (void)hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost);
    double sums[8];
    for (int i = 0; i < 8; i++) { sums[i] = 0; for (int l = 0; l < 64; l++) sums[i] += h[i * 64 + l]; }
    for (int r = 0; r < 4; r++) {
        const int lane = 15 + 16 * r;
        for (int x = 0; x < 2; x++) {
            int which = -1;
            for (int i = 0; i < 8; i++) if (ho[x * 64 + lane] == (float)sums[i]) which = i;
            printf("lane %d x%d = %.0f -> v%d\n", lane, x, ho[x * 64 + lane], which);
        }
    }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 5000, blocks = 256 * 4 * 8; float ms;
    for (int rep = 0; rep < 2; rep++) { (void)hipEventRecord(e0); hipLaunchKernelGGL(tk, dim3(blocks), dim3(64), 0, 0, o, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); }
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("permlane swap: %.2f cycles/instr/SIMD @2.4GHz (incl. hazard nops)\n", 1024.0 * 2.4e9 * ms * 1e-3 / ((double)blocks * iters * 64));
    return 0;
}
