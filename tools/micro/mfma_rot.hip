// MFMA issue-rate probe, round 4: v_mfma_f32_16x16x4_f32 as asm statements (program order = issue order, as in the conv
// kernels), NACC accumulators in rotation, NA / NB distinct A / B operand registers, W waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_rot.hip -o tools/micro/mfma_rot && tools/micro/mfma_rot
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC, int NA, int NB>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float seed) {
    f32x4 acc[NACC];
    float a[NA], b[NB];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NA; ++i) a[i] = seed + threadIdx.x * 1e-3f + i;
#pragma unroll
    for (int i = 0; i < NB; ++i) b[i] = seed * 0.5f + threadIdx.x * 2e-3f - i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i % NA]), "v"(b[i % NB]));
    }
    asm volatile("s_nop 15" ::: "memory");
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC, int NA, int NB>
void run(int w) {
    const int iters = 48000 / NACC;
    int nblk = 256 * w;
    float* out; (void)hipMalloc(&out, nblk * 256 * sizeof(float));
    hipEvent_t s, e; (void)hipEventCreate(&s); (void)hipEventCreate(&e);
    probe<NACC, NA, NB><<<nblk, 256>>>(out, iters, 1.0f); (void)hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 4; ++r) {
        (void)hipEventRecord(s); probe<NACC, NA, NB><<<nblk, 256>>>(out, iters, 1.0f + r); (void)hipEventRecord(e); (void)hipEventSynchronize(e);
        float ms; (void)hipEventElapsedTime(&ms, s, e); if (ms < best) best = ms;
    }
    double flops = (double)nblk * 4 * iters * NACC * 2048.0;
    printf("NACC %2d NA %d NB %2d waves/SIMD %d: %7.1f TFLOP/s\n", NACC, NA, NB, w, flops / best / 1e9);
    (void)hipFree(out);
}
template <int NACC> void sweep() { for (int w = 1; w <= 4; ++w) run<NACC, 4, 12>(w); }
int main() {
    sweep<1>(); sweep<2>(); sweep<3>(); sweep<4>(); sweep<5>(); sweep<6>(); sweep<7>(); sweep<8>(); sweep<9>(); sweep<10>();
    sweep<12>(); sweep<16>();
    for (int w = 1; w <= 3; ++w) { run<3, 1, 1>(w); run<3, 1, 3>(w); run<12, 1, 1>(w); run<4, 1, 1>(w); run<6, 2, 6>(w); }
    return 0;
}
