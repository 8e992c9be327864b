// MFMA issue-rate probe in the shape of the conv_fwd_pipe step: G row groups x TW column tiles of independent
// accumulators, KS = 4 k-steps with distinct operand registers, no memory traffic; WPS waves per SIMD (64-thread blocks).
// SPLIT = 1: even/odd k-steps go to two accumulator sets (twice the independent chains).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int G, int TW, int SPLIT>
__global__ __launch_bounds__(64, 2) void probe(float* out, int iters, float seed) {
    f32x4 acc[2][G][TW];
    float av[G][4], bv[4][TW];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int t = 0; t < TW; ++t) acc[h][g][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int s = 0; s < 4; ++s) av[g][s] = seed + threadIdx.x * 1e-3f + g + s * 0.1f;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int t = 0; t < TW; ++t) bv[s][t] = seed * 0.5f + threadIdx.x * 2e-3f + t + s * 0.3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int t = 0; t < TW; ++t) {
                    const int h = SPLIT ? (s & 1) : 0;
                    acc[h][g][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[s][t], av[g][s], acc[h][g][t], 0, 0, 0);
                }
        __builtin_amdgcn_sched_barrier(0);
    }
    float r = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int t = 0; t < TW; ++t) r += acc[h][g][t][0] + acc[h][g][t][1] + acc[h][g][t][2] + acc[h][g][t][3];
    out[blockIdx.x * 64 + threadIdx.x] = r;
}
template <int G, int TW, int SPLIT>
void run(int wps, int iters) {
    int nblk = 256 * 4 * wps;
    float* out; hipMalloc(&out, nblk * 64 * sizeof(float));
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    probe<G, TW, SPLIT><<<nblk, 64>>>(out, iters, 1.0f); hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(s); probe<G, TW, SPLIT><<<nblk, 64>>>(out, iters, 1.0f + r); hipEventRecord(e); hipEventSynchronize(e);
        float ms; hipEventElapsedTime(&ms, s, e); if (ms < best) best = ms;
    }
    double flops = (double)nblk * iters * 4.0 * G * TW * 2048.0;
    printf("G %d TW %d split %d waves/SIMD %d: %.3f ms  %.1f TFLOP/s\n", G, TW, SPLIT, wps, best, flops / best / 1e9);
    hipFree(out);
}
int main() {
    for (int w = 1; w <= 2; ++w) {
        run<1, 3, 0>(w, 40000); run<1, 3, 1>(w, 40000); run<2, 3, 0>(w, 20000); run<2, 3, 1>(w, 20000);
        run<3, 3, 0>(w, 14000); run<4, 3, 0>(w, 10000); run<1, 2, 0>(w, 40000); run<1, 2, 1>(w, 40000); run<4, 2, 0>(w, 10000);
    }
    return 0;
}
