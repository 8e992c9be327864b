// Pure MFMA issue-rate probe for v_mfma_f32_16x16x4_f32: NACC independent accumulators per wave, WPS waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float seed) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = seed + threadIdx.x * 1e-3f, b = seed * 0.5f + threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        a += 1e-6f;
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks_per_cu, int iters) {
    int nblk = 256 * blocks_per_cu;
    float* out; hipMalloc(&out, nblk * 256 * sizeof(float));
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    probe<NACC><<<nblk, 256>>>(out, iters, 1.0f); hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(s); probe<NACC><<<nblk, 256>>>(out, iters, 1.0f + r); hipEventRecord(e); hipEventSynchronize(e);
        float ms; hipEventElapsedTime(&ms, s, e); if (ms < best) best = ms;
    }
    double flops = (double)nblk * 4 * iters * NACC * 2048.0;
    printf("NACC %2d waves/SIMD %d: %.3f ms  %.1f TFLOP/s\n", NACC, blocks_per_cu, best, flops / best / 1e9);
    hipFree(out);
}
int main() {
    for (int w = 1; w <= 4; ++w) { run<2>(w, 20000); run<3>(w, 14000); run<4>(w, 10000); run<6>(w, 7000); run<12>(w, 4000); }
    return 0;
}
