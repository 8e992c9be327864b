// What does adding an offset's result into a SHARED LDS strip cost?  (round 5: the 256-row cooperative tile needs a flush that
// several waves can run against the same strip.)  12 waves per CU (3 workgroups x 4 waves, 53 KB of strip per workgroup), every
// wave repeats: B MFMAs (3 accumulators in rotation, the conv kernels' block), s_nop, flush of 4 row groups x 3 column tiles.
//   MODE 0: the shipped flush -- swapped operand roles, a lane holds four consecutive channels of ONE pair: ds_read_b128,
//           v_add, ds_write_b128 per (group, tile) into the wave's OWN 64-row part of the strip
//   MODE 1: fire-and-forget ds_add_f32 -- textbook roles, a lane holds channel i of pairs 4q .. 4q+3: four ds_add_f32 per
//           (group, tile) onto arbitrary rows of the 256-row strip (rows within a flush distinct)
//   MODE 2: as 1, swapped roles (lane: four consecutive channels of one pair): the bank pattern the shipped layout would give
// ROWS 0: consecutive output rows; 1: rows with random gaps (1..3); 2: random rows of the strip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int PITCH = 52;
template <int MODE, int B>
__global__ __launch_bounds__(256, 3) void probe(float* out, int iters, int rows_mode, float seed) {
    __shared__ float strip[256 * PITCH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, q = lane >> 4;
    for (int e = threadIdx.x; e < 256 * PITCH; e += 256) strip[e] = 0.f;
    __syncthreads();
    f32x4 acc[4][3];
    for (int g = 0; g < 4; ++g) for (int t = 0; t < 3; ++t) acc[g][t] = f32x4{seed, seed, seed, seed};
    float a[4], b[12];
    for (int s = 0; s < 4; ++s) a[s] = seed + threadIdx.x * 1e-3f + s;
    for (int s = 0; s < 12; ++s) b[s] = seed * 0.5f + threadIdx.x * 2e-3f - s;
    unsigned r = (blockIdx.x * 4 + wave) * 2654435761u + 12345u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll 1
        for (int blk = 0; blk < B; ++blk) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                asm volatile(
                    "v_mfma_f32_16x16x4_f32 %0, %7, %3, %0\n\tv_mfma_f32_16x16x4_f32 %1, %8, %3, %1\n\tv_mfma_f32_16x16x4_f32 %2, %9, %3, %2\n\t"
                    "v_mfma_f32_16x16x4_f32 %0, %10, %4, %0\n\tv_mfma_f32_16x16x4_f32 %1, %11, %4, %1\n\tv_mfma_f32_16x16x4_f32 %2, %12, %4, %2\n\t"
                    "v_mfma_f32_16x16x4_f32 %0, %13, %5, %0\n\tv_mfma_f32_16x16x4_f32 %1, %14, %5, %1\n\tv_mfma_f32_16x16x4_f32 %2, %15, %5, %2\n\t"
                    "v_mfma_f32_16x16x4_f32 %0, %16, %6, %0\n\tv_mfma_f32_16x16x4_f32 %1, %17, %6, %1\n\tv_mfma_f32_16x16x4_f32 %2, %18, %6, %2"
                    : "+v"(acc[g][0]), "+v"(acc[g][1]), "+v"(acc[g][2])
                    : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]),
                      "v"(b[6]), "v"(b[7]), "v"(b[8]), "v"(b[9]), "v"(b[10]), "v"(b[11]));
        }
        // the 64 output rows of this flush (wave-uniform start, per-slot rows)
        r = r * 1664525u + 1013904223u;
        const unsigned base = MODE == 0 ? (unsigned)wave * 64u : (r >> 8) % 192u;
        auto row_of = [&](int p) -> unsigned {       // output row of pair p (0..63) of the flush
            if (rows_mode == 0) return (base + p) & 255u;
            if (rows_mode == 1) return MODE == 0 ? (unsigned)wave * 64u + ((p * 37u + (r >> 12)) & 63u) : (base + p + ((p * 2654435761u + r) >> 30)) & 255u;
            return MODE == 0 ? (unsigned)wave * 64u + ((p * 29u + (r >> 12)) & 63u) : ((p * 167u + (r >> 10)) & 255u);
        };
        asm volatile("s_nop 15" ::: "memory");
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (MODE == 0) {
                float* rowp = strip + row_of(16 * g + i) * PITCH + 4 * q;
                f32x4 old[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) old[t] = *(const f32x4*)(rowp + 16 * t);
#pragma unroll
                for (int t = 0; t < 3; ++t) *(f32x4*)(rowp + 16 * t) = old[t] + acc[g][t];
            } else if (MODE == 1) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const unsigned addr = (row_of(16 * g + 4 * q + u) * PITCH + i) * 4u;
                    asm volatile("ds_add_f32 %0, %1\n\tds_add_f32 %0, %2 offset:64\n\tds_add_f32 %0, %3 offset:128"
                                 :: "v"(addr), "v"(acc[g][0][u]), "v"(acc[g][1][u]), "v"(acc[g][2][u]) : "memory");
                }
            } else {
                const unsigned addr0 = (row_of(16 * g + i) * PITCH + 4 * q) * 4u;
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        asm volatile("ds_add_f32 %0, %1" :: "v"(addr0 + (16 * t + u) * 4u), "v"(acc[g][t][u]) : "memory");
            }
#pragma unroll
            for (int t = 0; t < 3; ++t) acc[g][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    __syncthreads();
    float s = 0.f;
    for (int e = threadIdx.x; e < 256 * PITCH; e += 256) s += strip[e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE, int B>
void run(int rows_mode) {
    const int iters = 400, nblk = 256 * 3 * 4;
    float* out; (void)hipMalloc(&out, nblk * 256 * sizeof(float));
    hipEvent_t s, e; (void)hipEventCreate(&s); (void)hipEventCreate(&e);
    probe<MODE, B><<<nblk, 256>>>(out, iters, rows_mode, 1.0f); (void)hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 3; ++r) {
        (void)hipEventRecord(s); probe<MODE, B><<<nblk, 256>>>(out, iters, rows_mode, 1.0f + r); (void)hipEventRecord(e); (void)hipEventSynchronize(e);
        float ms; (void)hipEventElapsedTime(&ms, s, e); if (ms < best) best = ms;
    }
    // per CU: 12 resident waves, nblk * 4 waves in all -> (nblk * 4 / (256 * 12)) rounds of `iters` flushes per wave slot
    const double flushes_per_cu = (double)nblk * 4 * iters / 256.0;
    const double ns_per_flush_cu = best * 1e6 / flushes_per_cu;
    const double flops = (double)nblk * 4 * iters * B * 48 * 2048.0;
    printf("mode %d rows %d MFMA blocks %2d: %8.3f ms  %7.1f ns of CU time per flush (%.0f cycles at 2.4 GHz)  %7.1f TFLOP/s\n", MODE, rows_mode, B, best,
           ns_per_flush_cu, ns_per_flush_cu * 2.4, flops / best / 1e9);
    (void)hipFree(out);
}
int main() {
    for (int rows = 0; rows < 3; ++rows) {
        run<0, 0>(rows); run<1, 0>(rows); run<2, 0>(rows);
        run<0, 1>(rows); run<1, 1>(rows); run<2, 1>(rows);
        run<0, 6>(rows); run<1, 6>(rows); run<2, 6>(rows);
    }
    return 0;
}
