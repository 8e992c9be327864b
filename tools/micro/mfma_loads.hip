// Does vector-memory traffic into the register file slow the MFMA pipe down?  12 MFMAs (3 accumulators, the conv kernels'
// block) then L x global_load_dwordx4 into a ring of 2L register quads, W waves per SIMD; the loaded values are summed into a
// side accumulator once per ring turn (so the loads are real), never fed to the MFMAs.
//   MODE 0: every lane reads its own 16 B of a 1 KiB block per load (the weight-block pattern: 1 KiB contiguous per instruction)
//   MODE 1: gather pattern: lane (i, q) reads 16 B of row r(i) at column q (16 rows x 64 B per instruction), rows pseudo-random
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int L, int MODE>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ buf, size_t nrows, float* out, int iters, float seed) {
    f32x4 acc[3];
    float a[4], b[12];
    for (int i = 0; i < 3; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 4; ++i) a[i] = seed + threadIdx.x * 1e-3f + i;
    for (int i = 0; i < 12; ++i) b[i] = seed * 0.5f + threadIdx.x * 2e-3f - i;
    f32x4 ring[2 * (L > 0 ? L : 1)];
    for (int i = 0; i < 2 * (L > 0 ? L : 1); ++i) ring[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 side = {0.f, 0.f, 0.f, 0.f};
    const int lane = threadIdx.x & 63, i16 = lane & 15, q = lane >> 4;
    unsigned r = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
    const unsigned wave_id = (blockIdx.x * 4 + (threadIdx.x >> 6));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int i = 0; i < 12; ++i)
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i % 3]) : "v"(a[i / 3]), "v"(b[i]));
#pragma unroll
            for (int l = 0; l < L; ++l) {
                side += ring[half * L + l];                       // consume the value loaded two half-steps ago
                size_t off;
                if (MODE == 0) off = ((size_t)((wave_id * 7 + it * 3 + l) % (nrows / 16)) * 16 * 24 + (size_t)lane * 4);   // 1 KiB contiguous
                else { r = r * 1664525u + 1013904223u; const unsigned row = __shfl(r, i16, 64) % nrows; off = (size_t)row * 24 * 4 + q * 4 + (size_t)((it + l) % 6) * 16; }
                ring[half * L + l] = *(const f32x4*)(buf + off);
            }
        }
    }
    asm volatile("s_nop 15" ::: "memory");
    float s = side[0] + side[1] + side[2] + side[3];
    for (int i = 0; i < 3; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int L, int MODE>
void run(int w, const float* buf, size_t nrows) {
    const int iters = 2000;
    int nblk = 256 * w;
    float* out; (void)hipMalloc(&out, nblk * 256 * sizeof(float));
    hipEvent_t s, e; (void)hipEventCreate(&s); (void)hipEventCreate(&e);
    probe<L, MODE><<<nblk, 256>>>(buf, nrows, out, iters, 1.0f); (void)hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 3; ++r) {
        (void)hipEventRecord(s); probe<L, MODE><<<nblk, 256>>>(buf, nrows, out, iters, 1.0f + r); (void)hipEventRecord(e); (void)hipEventSynchronize(e);
        float ms; (void)hipEventElapsedTime(&ms, s, e); if (ms < best) best = ms;
    }
    double flops = (double)nblk * 4 * iters * 24 * 2048.0;
    printf("loads per 12 MFMAs %d mode %d waves/SIMD %d: %7.1f TFLOP/s  (%.1f GB/s per CU-load stream)\n", L, MODE, w, flops / best / 1e9,
           (double)nblk * 4 * iters * 2 * L * 1024.0 / best / 1e6 / 256);
    (void)hipFree(out);
}
int main() {
    const size_t nrows = 1200000;            // 1.2 M rows x 96 floats = 461 MB (MODE 1 gathers); MODE 0 walks the first part
    float* buf; (void)hipMalloc(&buf, nrows * 24 * 4 * sizeof(float));
    (void)hipMemset(buf, 0, nrows * 24 * 4 * sizeof(float));
    for (int w = 1; w <= 4; ++w) {
        run<0, 0>(w, buf, nrows); run<2, 0>(w, buf, nrows); run<4, 0>(w, buf, nrows); run<7, 0>(w, buf, nrows);
        run<2, 1>(w, buf, nrows); run<4, 1>(w, buf, nrows); run<7, 1>(w, buf, nrows);
    }
    return 0;
}
