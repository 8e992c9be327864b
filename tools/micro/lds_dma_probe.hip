// Review item 1(d), round 5: do global_load_lds_dwordx4 pieces ride the same ~46 B/clk per-CU return path as loads into
// registers (tools/micro/mfma_loads.hip)?  Per half-iteration a wave issues 12 MFMAs (the conv kernels' block) and L one-KiB
// loads of a contiguous weight-block-like stream:
//   MODE 0: global_load_dwordx4 into a register ring (as mfma_loads.hip MODE 0)
//   MODE 1: global_load_lds_dwordx4 into the wave's LDS ring (2L KiB per wave), never read
//   MODE 2: as 1, and every landed piece is read back with one ds_read_b128 per lane (what an MFMA operand fetch would do)
// 3 waves per SIMD (the conv kernel's residency).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int L, int MODE>
__global__ __launch_bounds__(256, 3) void probe(const float* __restrict__ buf, size_t nblocks1k, float* out, int iters, float seed) {
    extern __shared__ float lds[];                         // 4 waves x 2L KiB
    f32x4 acc[3];
    float a[4], b[12];
    for (int i = 0; i < 3; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 4; ++i) a[i] = seed + threadIdx.x * 1e-3f + i;
    for (int i = 0; i < 12; ++i) b[i] = seed * 0.5f + threadIdx.x * 2e-3f - i;
    constexpr int R = 2 * (L > 0 ? L : 1);
    f32x4 ring[R];
    for (int i = 0; i < R; ++i) ring[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 side = {0.f, 0.f, 0.f, 0.f};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* myring = lds + wave * R * 256;                  // R pieces of 256 floats
    const unsigned wave_id = blockIdx.x * 4 + wave;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int i = 0; i < 12; ++i)
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i % 3]) : "v"(a[i / 3]), "v"(b[i]));
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const size_t off = (size_t)((wave_id * 7 + it * 3 + l) % nblocks1k) * 256 + (size_t)lane * 4;
                if (MODE == 0) {
                    side += ring[half * L + l];
                    ring[half * L + l] = *(const f32x4*)(buf + off);
                } else {
                    if (MODE == 2) {                       // the piece that landed two half-steps ago (all DMAs issued before the
                                                           // previous round's are complete: in-order queue)
                        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * L - 1) : "memory");
                        side += *(const f32x4*)(myring + (half * L + l) * 256 + lane * 4);
                    }
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(buf + off),
                                                     (__attribute__((address_space(3))) void*)(myring + (half * L + l) * 256), 16, 0, 0);
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15" ::: "memory");
    float s = side[0] + side[1] + side[2] + side[3];
    if (MODE == 1) s += myring[lane];
    for (int i = 0; i < 3; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int L, int MODE>
void run(const float* buf, size_t nblocks1k) {
    const int iters = 2000, w = 3;
    int nblk = 256 * w * 4;
    float* out; (void)hipMalloc(&out, nblk * 256 * sizeof(float));
    const size_t shmem = 4 * 2 * (L > 0 ? L : 1) * 1024;
    hipEvent_t s, e; (void)hipEventCreate(&s); (void)hipEventCreate(&e);
    probe<L, MODE><<<nblk, 256, shmem>>>(buf, nblocks1k, out, iters, 1.0f); (void)hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 3; ++r) {
        (void)hipEventRecord(s); probe<L, MODE><<<nblk, 256, shmem>>>(buf, nblocks1k, out, iters, 1.0f + r); (void)hipEventRecord(e); (void)hipEventSynchronize(e);
        float ms; (void)hipEventElapsedTime(&ms, s, e); if (ms < best) best = ms;
    }
    double flops = (double)nblk * 4 * iters * 24 * 2048.0;
    printf("loads per 12 MFMAs %d mode %d (%s): %7.1f TFLOP/s  %.1f B/clk per CU at 2.4 GHz\n", L, MODE,
           MODE == 0 ? "registers" : MODE == 1 ? "LDS-DMA, unread" : "LDS-DMA + ds_read_b128", flops / best / 1e9,
           (double)nblk * 4 * iters * 2 * L * 1024.0 / (best * 1e-3) / 256 / 2.4e9);
    (void)hipFree(out);
}
int main() {
    const size_t nblocks1k = 972;                          // a 96 -> 96 layer's packed image: 972 KiB, L2-resident
    float* buf; (void)hipMalloc(&buf, nblocks1k * 1024 + 4096);
    (void)hipMemset(buf, 0, nblocks1k * 1024 + 4096);
    run<0, 0>(buf, nblocks1k);
    run<3, 0>(buf, nblocks1k); run<3, 1>(buf, nblocks1k); run<3, 2>(buf, nblocks1k);
    run<7, 0>(buf, nblocks1k); run<7, 1>(buf, nblocks1k); run<7, 2>(buf, nblocks1k);
    return 0;
}
