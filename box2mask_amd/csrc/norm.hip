// BatchNorm statistics / apply / backward, small elementwise ops and segment pooling.
// All HBM-bound: float4 accesses, one pass per tensor, deterministic two-stage column reductions.
#include "b2m_common.h"

#define RED_MAX_BLOCKS 1280   // 256 CUs x 5 resident blocks of bn_bwd_reduce: one full round, no tail (4096: 12 % slower)

// Column reduction skeleton.  256 threads; thread -> (float4 column group cg, row slot rs).
// F(row, cg) returns two float4 contributions (a, b); the block writes double partial sums
// partial[blk*2c + col] (a) and partial[blk*2c + c + col] (b).
template <class F>
__device__ __forceinline__ void column_reduce(int64_t n, int c, double* __restrict__ partial, F f) {
    extern __shared__ float red[];             // [nslots][c4][8]
    const int c4 = c >> 2;
    const int nslots = 256 / c4;
    const int cg = threadIdx.x % c4, rs = threadIdx.x / c4;
    const int64_t rows_per_blk = (n + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_blk;
    int64_t r1 = r0 + rows_per_blk;
    if (r1 > n) r1 = n;
    f32x4 sa = {0, 0, 0, 0}, sb = {0, 0, 0, 0};
    if (rs < nslots) {
        int64_t r = r0 + rs;
        for (; r + 3 * nslots < r1; r += 4 * nslots) {  // four independent rows per iteration (memory-level parallelism)
            f32x4 a0, b0, a1, b1, a2, b2, a3, b3;
            f(r, cg, a0, b0);
            f(r + nslots, cg, a1, b1);
            f(r + 2 * nslots, cg, a2, b2);
            f(r + 3 * nslots, cg, a3, b3);
            sa += a0; sb += b0; sa += a1; sb += b1; sa += a2; sb += b2; sa += a3; sb += b3;
        }
        for (; r < r1; r += nslots) {
            f32x4 a, b;
            f(r, cg, a, b);
            sa += a; sb += b;
        }
        float* p = red + ((size_t)rs * c4 + cg) * 8;
        *(f32x4*)p = sa; *(f32x4*)(p + 4) = sb;
    }
    __syncthreads();
    if (rs == 0) {
        double da[4] = {0, 0, 0, 0}, db[4] = {0, 0, 0, 0};
        for (int s = 0; s < nslots; ++s) {
            const float* p = red + ((size_t)s * c4 + cg) * 8;
#pragma unroll
            for (int u = 0; u < 4; ++u) { da[u] += (double)p[u]; db[u] += (double)p[4 + u]; }
        }
        double* o = partial + (size_t)blockIdx.x * 2 * c;
#pragma unroll
        for (int u = 0; u < 4; ++u) { o[cg * 4 + u] = da[u]; o[c + cg * 4 + u] = db[u]; }
    }
}
// The same with the loads of four rows issued BEFORE any of them is used: `load(row, cg, in)` fills NIN float4 registers,
// `f(in, a, b)` turns them into the two contributions.  In column_reduce the functor loads and computes, and with
// wave-uniform branches inside it hipcc waited for every row's data before issuing the next row's loads: two or three
// 16-byte loads in flight per thread, 3.5 TB/s; staged, eight to twelve.
template <int NIN, class L, class F>
__device__ __forceinline__ void column_reduce_staged(int64_t n, int c, double* __restrict__ partial, L load, F f) {
    extern __shared__ float red[];             // [nslots][c4][8]
    const int c4 = c >> 2;
    const int nslots = 256 / c4;
    const int cg = threadIdx.x % c4, rs = threadIdx.x / c4;
    const int64_t rows_per_blk = (n + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_blk;
    int64_t r1 = r0 + rows_per_blk;
    if (r1 > n) r1 = n;
    f32x4 sa = {0, 0, 0, 0}, sb = {0, 0, 0, 0};
    if (rs < nslots) {
        int64_t r = r0 + rs;
        for (; r + 3 * nslots < r1; r += 4 * nslots) {
            f32x4 in[4][NIN];
#pragma unroll
            for (int u = 0; u < 4; ++u) load(r + (int64_t)u * nslots, cg, in[u]);
            __builtin_amdgcn_sched_barrier(0);         // (left alone, the scheduler sinks each load to its first use)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                f32x4 a, b;
                f(in[u], a, b);
                sa += a; sb += b;
            }
        }
        for (; r < r1; r += nslots) {
            f32x4 in[NIN], a, b;
            load(r, cg, in);
            f(in, a, b);
            sa += a; sb += b;
        }
        float* p = red + ((size_t)rs * c4 + cg) * 8;
        *(f32x4*)p = sa; *(f32x4*)(p + 4) = sb;
    }
    __syncthreads();
    if (rs == 0) {
        double da[4] = {0, 0, 0, 0}, db[4] = {0, 0, 0, 0};
        for (int s = 0; s < nslots; ++s) {
            const float* p = red + ((size_t)s * c4 + cg) * 8;
#pragma unroll
            for (int u = 0; u < 4; ++u) { da[u] += (double)p[u]; db[u] += (double)p[4 + u]; }
        }
        double* o = partial + (size_t)blockIdx.x * 2 * c;
#pragma unroll
        for (int u = 0; u < 4; ++u) { o[cg * 4 + u] = da[u]; o[c + cg * 4 + u] = db[u]; }
    }
}
// one wave per output column: lanes sum strided partials, then a fixed-order shuffle tree (deterministic)
__global__ __launch_bounds__(64) void reduce_final_kernel(const double* __restrict__ partial, int nblk, int c2,
                                                          double* __restrict__ out, float* __restrict__ out_lo,
                                                          float* __restrict__ out_hi, float f32_scale = 1.f) {
    const int j = blockIdx.x;
    double s = 0;
    for (int b = threadIdx.x; b < nblk; b += 64) s += partial[(size_t)b * c2 + j];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d, 64);
    if (threadIdx.x == 0) {
        out[j] = s;
        const int c = c2 >> 1;          // fp32 copies of the two halves go to two separate buffers
        // (f32_scale: half-precision training un-scales the parameter gradients here; the fp64 sums stay as they are)
        if (j < c) { if (out_lo) out_lo[j] = (float)s * f32_scale; }
        else if (out_hi) out_hi[j - c] = (float)s * f32_scale;
    }
}
static int reduce_blocks(int64_t n) {
    int64_t b = (n + 255) / 256;
    if (b < 1) b = 1;
    if (b > RED_MAX_BLOCKS) b = RED_MAX_BLOCKS;
    return (int)b;
}

// ------------------------------------------------------------------ BN forward
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, int64_t ldx, int64_t n, int c,
                                                       double* __restrict__ partial) {
    column_reduce_staged<1>(n, c, partial,
        [&](int64_t r, int cg, f32x4 (&in)[1]) { in[0] = *(const f32x4*)(x + r * ldx + cg * 4); },
        [&](const f32x4 (&in)[1], f32x4& a, f32x4& b) { a = in[0]; b = in[0] * in[0]; });
}
extern "C" int b2m_bn_stats(const float* x, int64_t ldx, int64_t n, int32_t c, double* partial, double* stats,
                            void* stream) {
    hipStream_t st = (hipStream_t)stream;
    B2M_CHECK_ARG(x && partial && stats && c > 0 && c % 4 == 0 && c <= 1024 && ldx % 4 == 0 && ldx >= c,
                  "c and ldx must be multiples of 4, c <= 1024");
    B2M_CHECK_ARG(((uintptr_t)x % 16) == 0, "x must be 16-byte aligned");
    const int nblk = reduce_blocks(n);
    const int c4 = c / 4, nslots = 256 / c4;
    bn_stats_kernel<<<nblk, 256, (size_t)nslots * c4 * 8 * sizeof(float), st>>>(x, ldx, n, c, partial);
    reduce_final_kernel<<<2 * c, 64, 0, st>>>(partial, nblk, 2 * c, stats, nullptr, nullptr);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}

__global__ void bn_finalize_kernel(const double* __restrict__ stats, double count_host, const double* __restrict__ count_dev,
                                   int c, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float eps, float momentum,
                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                   float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ scale,
                                   float* __restrict__ shift) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= c) return;
    const double count = count_dev ? *count_dev : count_host;      // SyncBN: the all-reduced row count stays on the device
    double m, var;
    if (stats) {
        m = stats[j] / count;
        var = stats[c + j] / count - m * m;
        if (var < 0) var = 0;
        if (running_mean) {
            double unb = count > 1 ? var * count / (count - 1) : var;
            running_mean[j] = (float)((1.0 - momentum) * (double)running_mean[j] + momentum * m);
            running_var[j] = (float)((1.0 - momentum) * (double)running_var[j] + momentum * unb);
        }
    } else {                      // evaluation mode: running statistics
        m = running_mean[j];
        var = running_var[j];
    }
    double is = 1.0 / sqrt(var + (double)eps);
    double g = gamma ? (double)gamma[j] : 1.0, b = beta ? (double)beta[j] : 0.0;
    if (mean) mean[j] = (float)m;
    if (invstd) invstd[j] = (float)is;
    scale[j] = (float)(g * is);
    shift[j] = (float)(b - m * g * is);
}
// final reduction of the partial column sums fused with the finalize math: one wave per channel
__global__ __launch_bounds__(64) void bn_final_finalize_kernel(const double* __restrict__ partial, int nblk, double count,
                                                               int c, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float eps, float momentum,
                                                               float* __restrict__ running_mean,
                                                               float* __restrict__ running_var, float* __restrict__ mean,
                                                               float* __restrict__ invstd, float* __restrict__ scale,
                                                               float* __restrict__ shift, double* __restrict__ stats) {
    const int j = blockIdx.x;
    double s1 = 0, s2 = 0;
    for (int b = threadIdx.x; b < nblk; b += 64) {
        s1 += partial[(size_t)b * 2 * c + j];
        s2 += partial[(size_t)b * 2 * c + c + j];
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { s1 += __shfl_down(s1, d, 64); s2 += __shfl_down(s2, d, 64); }
    if (threadIdx.x != 0) return;
    if (stats) { stats[j] = s1; stats[c + j] = s2; }
    const double m = s1 / count;
    double var = s2 / count - m * m;
    if (var < 0) var = 0;
    if (running_mean) {
        const double unb = count > 1 ? var * count / (count - 1) : var;
        running_mean[j] = (float)((1.0 - momentum) * (double)running_mean[j] + momentum * m);
        running_var[j] = (float)((1.0 - momentum) * (double)running_var[j] + momentum * unb);
    }
    const double is = 1.0 / sqrt(var + (double)eps);
    const double g = gamma ? (double)gamma[j] : 1.0, b = beta ? (double)beta[j] : 0.0;
    if (mean) mean[j] = (float)m;
    if (invstd) invstd[j] = (float)is;
    scale[j] = (float)(g * is);
    shift[j] = (float)(b - m * g * is);
}
extern "C" int b2m_bn_stats_finalize(const float* x, int64_t ldx, int64_t n, int32_t c, double* partial, double* stats,
                                     const float* gamma, const float* beta, float eps, float momentum,
                                     float* running_mean, float* running_var, float* mean, float* invstd, float* scale,
                                     float* shift, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    B2M_CHECK_ARG(x && partial && scale && shift && c > 0 && c % 4 == 0 && c <= 1024 && ldx % 4 == 0 && ldx >= c,
                  "c and ldx must be multiples of 4, c <= 1024");
    B2M_CHECK_ARG(((uintptr_t)x % 16) == 0 && n >= 1, "x must be 16-byte aligned, n >= 1");
    const int nblk = reduce_blocks(n);
    const int c4 = c / 4, nslots = 256 / c4;
    bn_stats_kernel<<<nblk, 256, (size_t)nslots * c4 * 8 * sizeof(float), st>>>(x, ldx, n, c, partial);
    bn_final_finalize_kernel<<<c, 64, 0, st>>>(partial, nblk, (double)n, c, gamma, beta, eps, momentum, running_mean,
                                               running_var, mean, invstd, scale, shift, stats);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}

// Statistics from the per-tile column sums a convolution left behind (b2m_conv_fwd_stats): tile_stats [ntiles][2c]
// fp64 -> partial [nblk][2c] fp64, the layout reduce_final_kernel / bn_final_finalize_kernel read.  Fixed order.
__global__ __launch_bounds__(256) void bn_tilestats_kernel(const double* __restrict__ ts, int64_t ntiles, int c2,
                                                           double* __restrict__ partial) {
    const int64_t per = (ntiles + gridDim.x - 1) / gridDim.x;
    const int64_t t0 = (int64_t)blockIdx.x * per;
    int64_t t1 = t0 + per;
    if (t1 > ntiles) t1 = ntiles;
    for (int j = threadIdx.x; j < c2; j += 256) {
        double s = 0;
        for (int64_t t = t0; t < t1; ++t) s += ts[t * c2 + j];
        partial[(size_t)blockIdx.x * c2 + j] = s;
    }
}
static int tilestats_blocks(int64_t ntiles) {
    int64_t b = (ntiles + 15) / 16;
    if (b < 1) b = 1;
    if (b > RED_MAX_BLOCKS) b = RED_MAX_BLOCKS;
    return (int)b;
}
extern "C" int b2m_bn_tilestats(const double* tile_stats, int64_t ntiles, int32_t c, double* partial, double* stats,
                                void* stream) {
    hipStream_t st = (hipStream_t)stream;
    B2M_CHECK_ARG(tile_stats && partial && stats && ntiles >= 1 && c > 0 && c <= 1024, "bad arguments");
    const int nblk = tilestats_blocks(ntiles);
    bn_tilestats_kernel<<<nblk, 256, 0, st>>>(tile_stats, ntiles, 2 * c, partial);
    reduce_final_kernel<<<2 * c, 64, 0, st>>>(partial, nblk, 2 * c, stats, nullptr, nullptr);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}
// The same in ONE launch for maps of up to a few thousand tiles (every trunk layer below level 0: 85 launches of the forward
// pass sit on its critical path, each worth its launch gap): a workgroup owns 8 channels -- thread (tile lane tl, column
// cl) adds up column sum (cl < 8) or sum of squares (cl >= 8) of channel ch0 + cl % 8 over the tiles tl, tl + 16, ... (two
// 64-byte segments of a tile's row per tile lane), the 16 tile lanes meet in LDS in fixed order, and the channel's finalize
// math runs in the same workgroup.  Deterministic (another summation grouping than the two-launch form: last bits differ).
__global__ __launch_bounds__(256) void bn_tilestats_finalize_one_kernel(const double* __restrict__ ts, int64_t ntiles, double count,
                                                                        int c, const float* __restrict__ gamma,
                                                                        const float* __restrict__ beta, float eps, float momentum,
                                                                        float* __restrict__ running_mean,
                                                                        float* __restrict__ running_var, float* __restrict__ mean,
                                                                        float* __restrict__ invstd, float* __restrict__ scale,
                                                                        float* __restrict__ shift, double* __restrict__ stats) {
    __shared__ double red[16][16];
    const int tl = threadIdx.x >> 4, cl = threadIdx.x & 15;
    const int ch = blockIdx.x * 8 + (cl & 7);
    const int col = (cl < 8 ? 0 : c) + ch;
    double s[4] = {0., 0., 0., 0.};
    if (ch < c) {
        const double* p = ts + col;
        const int64_t c2 = 2 * (int64_t)c;
        int64_t t = tl;
        for (; t + 48 < ntiles; t += 64) {          // four tiles' loads in flight per thread
#pragma unroll
            for (int u = 0; u < 4; ++u) s[u] += p[(t + 16 * u) * c2];
        }
        for (; t < ntiles; t += 16) s[0] += p[t * c2];
    }
    red[tl][cl] = (s[0] + s[1]) + (s[2] + s[3]);
    __syncthreads();
    if (tl != 0 || cl >= 8 || ch >= c) return;
    double s1 = 0, s2 = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s1 += red[r][cl]; s2 += red[r][cl + 8]; }
    const int j = ch;
    if (stats) { stats[j] = s1; stats[c + j] = s2; }
    const double m = s1 / count;
    double var = s2 / count - m * m;
    if (var < 0) var = 0;
    if (running_mean) {
        const double unb = count > 1 ? var * count / (count - 1) : var;
        running_mean[j] = (float)((1.0 - momentum) * (double)running_mean[j] + momentum * m);
        running_var[j] = (float)((1.0 - momentum) * (double)running_var[j] + momentum * unb);
    }
    const double is = 1.0 / sqrt(var + (double)eps);
    const double g = gamma ? (double)gamma[j] : 1.0, b = beta ? (double)beta[j] : 0.0;
    if (mean) mean[j] = (float)m;
    if (invstd) invstd[j] = (float)is;
    scale[j] = (float)(g * is);
    shift[j] = (float)(b - m * g * is);
}
extern "C" int b2m_bn_tilestats_finalize(const double* tile_stats, int64_t ntiles, int64_t n, int32_t c, double* partial,
                                         double* stats, const float* gamma, const float* beta, float eps, float momentum,
                                         float* running_mean, float* running_var, float* mean, float* invstd, float* scale,
                                         float* shift, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    B2M_CHECK_ARG(tile_stats && partial && scale && shift && ntiles >= 1 && n >= 1 && c > 0 && c <= 1024, "bad arguments");
    // (round 6: 8192 -> 2048 tiles.  A workgroup of the one-launch form owns 8 channels over ALL tiles: 4 workgroups for a 32-channel
    // layer, 26 us on the 4.5 k tiles of level 1 where the two-launch reduction takes 10.)
    if (ntiles <= b2m_env_int("B2M_BN_TS_ONE", 2048)) {
        bn_tilestats_finalize_one_kernel<<<(c + 7) / 8, 256, 0, st>>>(tile_stats, ntiles, (double)n, c, gamma, beta, eps, momentum,
                                                                      running_mean, running_var, mean, invstd, scale, shift, stats);
        B2M_LAUNCH_CHECK();
        return B2M_OK;
    }
    const int nblk = tilestats_blocks(ntiles);
    bn_tilestats_kernel<<<nblk, 256, 0, st>>>(tile_stats, ntiles, 2 * c, partial);
    bn_final_finalize_kernel<<<c, 64, 0, st>>>(partial, nblk, (double)n, c, gamma, beta, eps, momentum, running_mean,
                                               running_var, mean, invstd, scale, shift, stats);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}

extern "C" int b2m_bn_finalize(const double* stats, double count, const double* count_dev, int32_t c, const float* gamma, const float* beta,
                               float eps, float momentum, float* running_mean, float* running_var, float* mean,
                               float* invstd, float* scale, float* shift, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    B2M_CHECK_ARG(c > 0 && scale && shift, "bad arguments");
    B2M_CHECK_ARG(stats || (running_mean && running_var), "eval mode needs running statistics");
    B2M_CHECK_ARG(!stats || count_dev || count >= 1, "count must be >= 1 (or given on the device)");
    bn_finalize_kernel<<<(c + 255) / 256, 256, 0, st>>>(stats, count, count_dev, c, gamma, beta, eps, momentum, running_mean,
                                                        running_var, mean, invstd, scale, shift);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, int64_t ldx, int64_t n, int c4,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       const float* __restrict__ res, int64_t ldr, int relu,
                                                       float* __restrict__ y, int64_t ldy) {
    // thread -> (row slot, float4 column group); rows are walked with a constant stride: no per-element division
    const int nslots = 256 / c4;
    const int cg = threadIdx.x % c4, rs = threadIdx.x / c4;
    if (rs >= nslots) return;
    const f32x4 s = *(const f32x4*)(scale + cg * 4), b = *(const f32x4*)(shift + cg * 4);
    for (int64_t r = (int64_t)blockIdx.x * nslots + rs; r < n; r += (int64_t)gridDim.x * nslots) {
        f32x4 v = *(const f32x4*)(x + r * ldx + cg * 4);
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = __builtin_fmaf(v[u], s[u], b[u]);    // the backward recomputes this sign
        if (res) v += *(const f32x4*)(res + r * ldr + cg * 4);
        if (relu) {
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = v[u] > 0.f ? v[u] : 0.f;
        }
        *(f32x4*)(y + r * ldy + cg * 4) = v;
    }
}
// grid for the row-chunk elementwise kernels: blocks of (256/c4) rows, enough blocks to fill the chip
static unsigned row_grid(int64_t n, int c4) {
    const int nslots = 256 / c4;
    int64_t g = (n + nslots - 1) / nslots;
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    return (unsigned)g;
}
static unsigned ew_grid(int64_t total) {
    int64_t g = (total + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    return (unsigned)g;
}
extern "C" int b2m_bn_apply(const float* x, int64_t ldx, int64_t n, int32_t c, const float* scale, const float* shift,
                            const float* residual, int64_t ldr, int32_t relu, float* y, int64_t ldy, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    B2M_CHECK_ARG(x && y && scale && shift && c > 0 && c % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 &&
                      (!residual || ldr % 4 == 0),
                  "c and leading dimensions must be multiples of 4");
    if (n == 0) return B2M_OK;
    B2M_CHECK_ARG(c <= 1024, "c <= 1024");
    bn_apply_kernel<<<row_grid(n, c / 4), 256, 0, st>>>(x, ldx, n, c / 4, scale, shift, residual, ldr, relu, y, ldy);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}

// ------------------------------------------------------------------ BN backward
template <bool RELU, bool HASY>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dy, int64_t lddy,
                                                            const float* __restrict__ y, int64_t ldy,
                                                            const float* __restrict__ x, int64_t ldx, int64_t n, int c,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ invstd,
                                                            const float* __restrict__ mscale,
                                                            const float* __restrict__ mshift,
                                                            double* __restrict__ partial) {
    // the per-column constants of this thread's column group, loaded once
    const int c4 = c >> 2, mycg = (threadIdx.x % c4) * 4;
    const f32x4 m = *(const f32x4*)(mean + mycg), is = *(const f32x4*)(invstd + mycg);
    f32x4 ms = {0.f, 0.f, 0.f, 0.f}, mb = {0.f, 0.f, 0.f, 0.f};
    if (RELU && !HASY) { ms = *(const f32x4*)(mscale + mycg); mb = *(const f32x4*)(mshift + mycg); }
    constexpr int NIN = HASY ? 3 : 2;
    column_reduce_staged<NIN>(n, c, partial,
        [&](int64_t r, int cg, f32x4 (&in)[NIN]) {
            in[0] = *(const f32x4*)(dy + r * lddy + cg * 4);
            in[1] = *(const f32x4*)(x + r * ldx + cg * 4);
            if constexpr (HASY) in[2] = *(const f32x4*)(y + r * ldy + cg * 4);
        },
        [&](const f32x4 (&in)[NIN], f32x4& a, f32x4& b) {
            f32x4 g = in[0];
            const f32x4 xx = in[1];
            if constexpr (RELU) {
                f32x4 yy;
                if constexpr (HASY) yy = in[2];
                else {      // no residual: the sign of the forward's fmaf(x, scale, shift), one tensor read less
#pragma unroll
                    for (int u = 0; u < 4; ++u) yy[u] = __builtin_fmaf(xx[u], ms[u], mb[u]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) g[u] = yy[u] > 0.f ? g[u] : 0.f;
            }
            a = g; b = g * ((xx - m) * is);
        });
}
extern "C" int b2m_bn_bwd_reduce(const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* x,
                                 int64_t ldx, int64_t n, int32_t c, const float* mean, const float* invstd,
                                 int32_t relu, const float* mask_scale, const float* mask_shift, double* partial,
                                 double* sums, float* dbeta_f32, float* dgamma_f32, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    B2M_CHECK_ARG(dy && x && mean && invstd && partial && sums && (!relu || y || (mask_scale && mask_shift)),
                  "NULL argument (relu needs y, or mask_scale and mask_shift)");
    B2M_CHECK_ARG(c > 0 && c % 4 == 0 && c <= 1024 && lddy % 4 == 0 && ldx % 4 == 0 && (!relu || ldy % 4 == 0),
                  "c and leading dimensions must be multiples of 4");
    const int nblk = reduce_blocks(n);
    const int c4 = c / 4, nslots = 256 / c4;
    const size_t lds = (size_t)nslots * c4 * 8 * sizeof(float);
    if (!relu) bn_bwd_reduce_kernel<false, false><<<nblk, 256, lds, st>>>(dy, lddy, nullptr, 0, x, ldx, n, c, mean, invstd, nullptr, nullptr, partial);
    else if (y) bn_bwd_reduce_kernel<true, true><<<nblk, 256, lds, st>>>(dy, lddy, y, ldy, x, ldx, n, c, mean, invstd, nullptr, nullptr, partial);
    else bn_bwd_reduce_kernel<true, false><<<nblk, 256, lds, st>>>(dy, lddy, nullptr, 0, x, ldx, n, c, mean, invstd, mask_scale, mask_shift, partial);
    reduce_final_kernel<<<2 * c, 64, 0, st>>>(partial, nblk, 2 * c, sums, dbeta_f32, dgamma_f32);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, int64_t lddy,
                                                           const float* __restrict__ y, int64_t ldy,
                                                           const float* __restrict__ x, int64_t ldx, int64_t n, int c,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma,
                                                           const double* __restrict__ sums, double count_host,
                                                           const double* __restrict__ count_dev, int relu,
                                                           const float* __restrict__ mscale,
                                                           const float* __restrict__ mshift,
                                                           float* __restrict__ dx, int64_t lddx,
                                                           float* __restrict__ dres, int64_t lddres) {
    const int c4 = c >> 2;
    const float inv_n = (float)(1.0 / (count_dev ? *count_dev : count_host));
    const int nslots = 256 / c4;
    const int cg = threadIdx.x % c4, rs = threadIdx.x / c4;
    if (rs >= nslots) return;
    const f32x4 m = *(const f32x4*)(mean + cg * 4), is = *(const f32x4*)(invstd + cg * 4);
    f32x4 msc = {0.f, 0.f, 0.f, 0.f}, msh = {0.f, 0.f, 0.f, 0.f};
    if (mscale) { msc = *(const f32x4*)(mscale + cg * 4); msh = *(const f32x4*)(mshift + cg * 4); }
    f32x4 sg, sgx, ga;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        sg[u] = (float)sums[cg * 4 + u] * inv_n; sgx[u] = (float)sums[c + cg * 4 + u] * inv_n;
        ga[u] = (gamma ? gamma[cg * 4 + u] : 1.f) * is[u];
    }
    for (int64_t r = (int64_t)blockIdx.x * nslots + rs; r < n; r += (int64_t)gridDim.x * nslots) {
        f32x4 g = *(const f32x4*)(dy + r * lddy + cg * 4);
        const f32x4 xx = *(const f32x4*)(x + r * ldx + cg * 4);
        if (relu) {
            f32x4 yy;
            if (y) yy = *(const f32x4*)(y + r * ldy + cg * 4);
            else {
#pragma unroll
                for (int u = 0; u < 4; ++u) yy[u] = __builtin_fmaf(xx[u], msc[u], msh[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) g[u] = yy[u] > 0.f ? g[u] : 0.f;
        }
        f32x4 out;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float xh = (xx[u] - m[u]) * is[u];
            out[u] = ga[u] * (g[u] - sg[u] - xh * sgx[u]);
        }
        *(f32x4*)(dx + r * lddx + cg * 4) = out;
        if (dres) *(f32x4*)(dres + r * lddres + cg * 4) = g;
    }
}
extern "C" int b2m_bn_bwd_apply(const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* x,
                                int64_t ldx, int64_t n, int32_t c, const float* mean, const float* invstd,
                                const float* gamma, const double* sums, double count, const double* count_dev,
                                int32_t relu, const float* mask_scale, const float* mask_shift, float* dx, int64_t lddx, float* dres,
                                int64_t lddres, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    B2M_CHECK_ARG(dy && x && mean && invstd && sums && dx && (!relu || y || (mask_scale && mask_shift)),
                  "NULL argument (relu needs y, or mask_scale and mask_shift)");
    B2M_CHECK_ARG(c > 0 && c % 4 == 0 && lddy % 4 == 0 && ldx % 4 == 0 && lddx % 4 == 0 && (!relu || ldy % 4 == 0) &&
                      (!dres || lddres % 4 == 0) && (count_dev || count >= 1),
                  "c and leading dimensions must be multiples of 4");
    if (n == 0) return B2M_OK;
    B2M_CHECK_ARG(c <= 1024, "c <= 1024");
    bn_bwd_apply_kernel<<<row_grid(n, c / 4), 256, 0, st>>>(dy, lddy, y, ldy, x, ldx, n, c, mean, invstd, gamma, sums,
                                                             count, count_dev, relu, y ? nullptr : mask_scale,
                                                             y ? nullptr : mask_shift, dx, lddx, dres, lddres);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}

// ------------------------------------------------------------------ half-precision training: BatchNorm with binary16 I/O
// (round 6; /root/reference/configs/arkitscenes.txt is the workload BASELINE names for it.)  The activations and their gradients live
// in HBM as IEEE half -- every pass below moves half the bytes of its fp32 twin --; statistics, the per-column constants and
// all arithmetic stay fp32 / fp64 exactly as above, one rounding to half on the way out.  The ReLU mask is always the stored
// output's sign (y > 0 on the half value the next layer saw), never recomputed from x: a pre-activation that rounds to +0 must
// be "off" in both directions.  Pitches in ELEMENTS, multiples of 4; 8-byte aligned rows.
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 ldh4(const _Float16* p) {
    const h16x4 v = *(const h16x4*)p;
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
__device__ __forceinline__ void sth4(_Float16* p, const f32x4 v) {
    h16x4 h;
    h[0] = (_Float16)v[0]; h[1] = (_Float16)v[1]; h[2] = (_Float16)v[2]; h[3] = (_Float16)v[3];
    *(h16x4*)p = h;
}
__global__ __launch_bounds__(256) void bn_stats_h_kernel(const _Float16* __restrict__ x, int64_t ldx, int64_t n, int c,
                                                         double* __restrict__ partial) {
    column_reduce_staged<1>(n, c, partial,
        [&](int64_t r, int cg, f32x4 (&in)[1]) { in[0] = ldh4(x + r * ldx + cg * 4); },
        [&](const f32x4 (&in)[1], f32x4& a, f32x4& b) { a = in[0]; b = in[0] * in[0]; });
}
extern "C" int b2m_bn_stats_h(const void* x, int64_t ldx, int64_t n, int32_t c, double* partial, double* stats, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    B2M_CHECK_ARG(x && partial && stats && c > 0 && c % 4 == 0 && c <= 1024 && ldx % 4 == 0 && ldx >= c && n >= 1,
                  "c and ldx must be multiples of 4, c <= 1024");
    B2M_CHECK_ARG(((uintptr_t)x % 8) == 0, "x must be 8-byte aligned");
    const int nblk = reduce_blocks(n);
    const int c4 = c / 4, nslots = 256 / c4;
    bn_stats_h_kernel<<<nblk, 256, (size_t)nslots * c4 * 8 * sizeof(float), st>>>((const _Float16*)x, ldx, n, c, partial);
    reduce_final_kernel<<<2 * c, 64, 0, st>>>(partial, nblk, 2 * c, stats, nullptr, nullptr);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}
// statistics + finalize of a half tensor in two launches (b2m_bn_stats_finalize's half twin)
extern "C" int b2m_bn_stats_finalize_h(const void* x, int64_t ldx, int64_t n, int32_t c, double* partial, const float* gamma,
                                       const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                                       float* mean, float* invstd, float* scale, float* shift, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    B2M_CHECK_ARG(x && partial && scale && shift && c > 0 && c % 4 == 0 && c <= 1024 && ldx % 4 == 0 && ldx >= c && n >= 1,
                  "c and ldx must be multiples of 4, c <= 1024");
    B2M_CHECK_ARG(((uintptr_t)x % 8) == 0, "x must be 8-byte aligned");
    const int nblk = reduce_blocks(n);
    const int c4 = c / 4, nslots = 256 / c4;
    bn_stats_h_kernel<<<nblk, 256, (size_t)nslots * c4 * 8 * sizeof(float), st>>>((const _Float16*)x, ldx, n, c, partial);
    bn_final_finalize_kernel<<<c, 64, 0, st>>>(partial, nblk, (double)n, c, gamma, beta, eps, momentum, running_mean, running_var,
                                               mean, invstd, scale, shift, nullptr);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}
__global__ __launch_bounds__(256) void bn_apply_h_kernel(const _Float16* __restrict__ x, int64_t ldx, int64_t n, int c4,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         const _Float16* __restrict__ res, int64_t ldr, int relu,
                                                         _Float16* __restrict__ y, int64_t ldy) {
    const int nslots = 256 / c4;
    const int cg = threadIdx.x % c4, rs = threadIdx.x / c4;
    if (rs >= nslots) return;
    const f32x4 s = *(const f32x4*)(scale + cg * 4), b = *(const f32x4*)(shift + cg * 4);
    for (int64_t r = (int64_t)blockIdx.x * nslots + rs; r < n; r += (int64_t)gridDim.x * nslots) {
        f32x4 v = ldh4(x + r * ldx + cg * 4);
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = __builtin_fmaf(v[u], s[u], b[u]);
        if (res) v += ldh4(res + r * ldr + cg * 4);
        if (relu) {
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = v[u] > 0.f ? v[u] : 0.f;
        }
        sth4(y + r * ldy + cg * 4, v);
    }
}
extern "C" int b2m_bn_apply_h(const void* x, int64_t ldx, int64_t n, int32_t c, const float* scale, const float* shift,
                              const void* residual, int64_t ldr, int32_t relu, void* y, int64_t ldy, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    B2M_CHECK_ARG(x && y && scale && shift && c > 0 && c % 4 == 0 && c <= 1024 && ldx % 4 == 0 && ldy % 4 == 0 &&
                      (!residual || ldr % 4 == 0), "c and leading dimensions must be multiples of 4, c <= 1024");
    B2M_CHECK_ARG(((uintptr_t)x % 8) == 0 && ((uintptr_t)y % 8) == 0 && ((uintptr_t)residual % 8) == 0, "8-byte aligned rows");
    if (n == 0) return B2M_OK;
    bn_apply_h_kernel<<<row_grid(n, c / 4), 256, 0, st>>>((const _Float16*)x, ldx, n, c / 4, scale, shift,
                                                          (const _Float16*)residual, ldr, relu, (_Float16*)y, ldy);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}
// HASY = false (ReLU without a fused residual): the mask is the sign of the forward's fmaf(x, scale, shift), recomputed from the x
// the kernel reads anyway -- one tensor read less, and the forward need not keep y (as the fp32 kernels do since round 2; a value that
// underflowed to zero in binary16, |v| < 3e-8, counts as positive here: the derivative of the unrounded function).
template <bool RELU, bool HASY>
__global__ __launch_bounds__(256) void bn_bwd_reduce_h_kernel(const _Float16* __restrict__ dy, int64_t lddy,
                                                              const _Float16* __restrict__ y, int64_t ldy,
                                                              const _Float16* __restrict__ x, int64_t ldx, int64_t n, int c,
                                                              const float* __restrict__ mean, const float* __restrict__ invstd,
                                                              const float* __restrict__ mscale, const float* __restrict__ mshift,
                                                              double* __restrict__ partial) {
    const int c4 = c >> 2, mycg = (threadIdx.x % c4) * 4;
    const f32x4 m = *(const f32x4*)(mean + mycg), is = *(const f32x4*)(invstd + mycg);
    f32x4 ms = {0.f, 0.f, 0.f, 0.f}, mb = {0.f, 0.f, 0.f, 0.f};
    if (RELU && !HASY) { ms = *(const f32x4*)(mscale + mycg); mb = *(const f32x4*)(mshift + mycg); }
    constexpr int NIN = (RELU && HASY) ? 3 : 2;
    column_reduce_staged<NIN>(n, c, partial,
        [&](int64_t r, int cg, f32x4 (&in)[NIN]) {
            in[0] = ldh4(dy + r * lddy + cg * 4);
            in[1] = ldh4(x + r * ldx + cg * 4);
            if constexpr (RELU && HASY) in[2] = ldh4(y + r * ldy + cg * 4);
        },
        [&](const f32x4 (&in)[NIN], f32x4& a, f32x4& b) {
            f32x4 g = in[0];
            if constexpr (RELU) {
                f32x4 yy;
                if constexpr (HASY) yy = in[NIN - 1];
                else {
#pragma unroll
                    for (int u = 0; u < 4; ++u) yy[u] = __builtin_fmaf(in[1][u], ms[u], mb[u]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) g[u] = yy[u] > 0.f ? g[u] : 0.f;
            }
            a = g; b = g * ((in[1] - m) * is);
        });
}
extern "C" int b2m_bn_bwd_reduce_h(const void* dy, int64_t lddy, const void* y, int64_t ldy, const void* x, int64_t ldx,
                                   int64_t n, int32_t c, const float* mean, const float* invstd, int32_t relu,
                                   const float* mask_scale, const float* mask_shift, double* partial,
                                   double* sums, float* dbeta_f32, float* dgamma_f32, float param_grad_scale, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    B2M_CHECK_ARG(dy && x && mean && invstd && partial && sums && (!relu || y || (mask_scale && mask_shift)),
                  "NULL argument (relu needs y, or mask_scale and mask_shift)");
    B2M_CHECK_ARG(c > 0 && c % 4 == 0 && c <= 1024 && lddy % 4 == 0 && ldx % 4 == 0 && (!relu || !y || ldy % 4 == 0) && n >= 1,
                  "c and leading dimensions must be multiples of 4");
    B2M_CHECK_ARG(((uintptr_t)dy % 8) == 0 && ((uintptr_t)x % 8) == 0 && ((uintptr_t)y % 8) == 0, "8-byte aligned rows");
    const int nblk = reduce_blocks(n);
    const int c4 = c / 4, nslots = 256 / c4;
    const size_t lds = (size_t)nslots * c4 * 8 * sizeof(float);
    if (relu && y) bn_bwd_reduce_h_kernel<true, true><<<nblk, 256, lds, st>>>((const _Float16*)dy, lddy, (const _Float16*)y, ldy, (const _Float16*)x, ldx, n, c, mean, invstd, nullptr, nullptr, partial);
    else if (relu) bn_bwd_reduce_h_kernel<true, false><<<nblk, 256, lds, st>>>((const _Float16*)dy, lddy, nullptr, 0, (const _Float16*)x, ldx, n, c, mean, invstd, mask_scale, mask_shift, partial);
    else bn_bwd_reduce_h_kernel<false, false><<<nblk, 256, lds, st>>>((const _Float16*)dy, lddy, nullptr, 0, (const _Float16*)x, ldx, n, c, mean, invstd, nullptr, nullptr, partial);
    reduce_final_kernel<<<2 * c, 64, 0, st>>>(partial, nblk, 2 * c, sums, dbeta_f32, dgamma_f32, param_grad_scale);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}
__global__ __launch_bounds__(256) void bn_bwd_apply_h_kernel(const _Float16* __restrict__ dy, int64_t lddy,
                                                             const _Float16* __restrict__ y, int64_t ldy,
                                                             const _Float16* __restrict__ x, int64_t ldx, int64_t n, int c,
                                                             const float* __restrict__ mean, const float* __restrict__ invstd,
                                                             const float* __restrict__ gamma, const double* __restrict__ sums,
                                                             double count_host, const double* __restrict__ count_dev, int relu,
                                                             const float* __restrict__ mscale, const float* __restrict__ mshift,
                                                             _Float16* __restrict__ dx, int64_t lddx,
                                                             _Float16* __restrict__ dres, int64_t lddres) {
    const int c4 = c >> 2;
    const float inv_n = (float)(1.0 / (count_dev ? *count_dev : count_host));
    const int nslots = 256 / c4;
    const int cg = threadIdx.x % c4, rs = threadIdx.x / c4;
    if (rs >= nslots) return;
    const f32x4 m = *(const f32x4*)(mean + cg * 4), is = *(const f32x4*)(invstd + cg * 4);
    f32x4 sg, sgx, ga;
    f32x4 msc = {0.f, 0.f, 0.f, 0.f}, msh = {0.f, 0.f, 0.f, 0.f};
    if (relu && !y) { msc = *(const f32x4*)(mscale + cg * 4); msh = *(const f32x4*)(mshift + cg * 4); }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        sg[u] = (float)sums[cg * 4 + u] * inv_n; sgx[u] = (float)sums[c + cg * 4 + u] * inv_n;
        ga[u] = (gamma ? gamma[cg * 4 + u] : 1.f) * is[u];
    }
    for (int64_t r = (int64_t)blockIdx.x * nslots + rs; r < n; r += (int64_t)gridDim.x * nslots) {
        f32x4 g = ldh4(dy + r * lddy + cg * 4);
        const f32x4 xx = ldh4(x + r * ldx + cg * 4);
        if (relu) {
            f32x4 yy;
            if (y) yy = ldh4(y + r * ldy + cg * 4);
            else {
#pragma unroll
                for (int u = 0; u < 4; ++u) yy[u] = __builtin_fmaf(xx[u], msc[u], msh[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) g[u] = yy[u] > 0.f ? g[u] : 0.f;
        }
        f32x4 out;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float xh = (xx[u] - m[u]) * is[u];
            out[u] = ga[u] * (g[u] - sg[u] - xh * sgx[u]);
        }
        sth4(dx + r * lddx + cg * 4, out);
        if (dres) sth4(dres + r * lddres + cg * 4, g);
    }
}
extern "C" int b2m_bn_bwd_apply_h(const void* dy, int64_t lddy, const void* y, int64_t ldy, const void* x, int64_t ldx, int64_t n,
                                  int32_t c, const float* mean, const float* invstd, const float* gamma, const double* sums,
                                  double count, const double* count_dev, int32_t relu, const float* mask_scale,
                                  const float* mask_shift, void* dx, int64_t lddx, void* dres, int64_t lddres, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    B2M_CHECK_ARG(dy && x && mean && invstd && sums && dx && (!relu || y || (mask_scale && mask_shift)),
                  "NULL argument (relu needs y, or mask_scale and mask_shift)");
    B2M_CHECK_ARG(c > 0 && c % 4 == 0 && c <= 1024 && lddy % 4 == 0 && ldx % 4 == 0 && lddx % 4 == 0 && (!relu || !y || ldy % 4 == 0) &&
                      (!dres || lddres % 4 == 0) && (count_dev || count >= 1), "c and leading dimensions must be multiples of 4");
    B2M_CHECK_ARG(((uintptr_t)dy % 8) == 0 && ((uintptr_t)x % 8) == 0 && ((uintptr_t)y % 8) == 0 && ((uintptr_t)dx % 8) == 0 &&
                      ((uintptr_t)dres % 8) == 0, "8-byte aligned rows");
    if (n == 0) return B2M_OK;
    bn_bwd_apply_h_kernel<<<row_grid(n, c / 4), 256, 0, st>>>((const _Float16*)dy, lddy, (const _Float16*)y, ldy, (const _Float16*)x, ldx,
                                                              n, c, mean, invstd, gamma, sums, count, count_dev, relu,
                                                              mask_scale, mask_shift, (_Float16*)dx, lddx, (_Float16*)dres, lddres);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}

// ------------------------------------------------------------------ two BatchNorms that meet in one add
// The last step of a BasicBlock with a shortcut convolution: y = relu(BN_a(conv2 out) + BN_b(1x1 shortcut out))
// (/root/reference/models/resnet.py:73-82).  As two BatchNorm launches the shortcut's normalised tensor is written and
// read back (2T of traffic), and backward writes the residual gradient g = dy * (y > 0) for the shortcut's BatchNorm to
// read twice (3T).  Both normalisations depend on nothing but their own input, and both backward reductions need only
// g, so: one apply over (x_a, x_b), one reduction for (sum g, sum g*xhat_a, sum g*xhat_b), one apply for (dx_a, dx_b) --
// and under SyncBN ONE packed all-reduce per direction for the pair instead of two.
__global__ __launch_bounds__(256) void bn_apply2_kernel(const float* __restrict__ xa, int64_t lda, const float* __restrict__ xb,
                                                        int64_t ldb, int64_t n, int c4, const float* __restrict__ sa,
                                                        const float* __restrict__ ba, const float* __restrict__ sb,
                                                        const float* __restrict__ bb, int relu, float* __restrict__ y,
                                                        int64_t ldy) {
    const int nslots = 256 / c4;
    const int cg = threadIdx.x % c4, rs = threadIdx.x / c4;
    if (rs >= nslots) return;
    const f32x4 s1 = *(const f32x4*)(sa + cg * 4), b1 = *(const f32x4*)(ba + cg * 4);
    const f32x4 s2 = *(const f32x4*)(sb + cg * 4), b2 = *(const f32x4*)(bb + cg * 4);
    for (int64_t r = (int64_t)blockIdx.x * nslots + rs; r < n; r += (int64_t)gridDim.x * nslots) {
        f32x4 v = *(const f32x4*)(xa + r * lda + cg * 4);
        f32x4 w = *(const f32x4*)(xb + r * ldb + cg * 4);
#pragma unroll
        for (int u = 0; u < 4; ++u) {        // exactly the two roundings of the separate launches: fma, fma, add
            v[u] = __builtin_fmaf(v[u], s1[u], b1[u]);
            w[u] = __builtin_fmaf(w[u], s2[u], b2[u]);
            v[u] = v[u] + w[u];
            if (relu) v[u] = v[u] > 0.f ? v[u] : 0.f;
        }
        *(f32x4*)(y + r * ldy + cg * 4) = v;
    }
}
extern "C" int b2m_bn_apply2(const float* xa, int64_t lda, const float* xb, int64_t ldb, int64_t n, int32_t c,
                             const float* scale_a, const float* shift_a, const float* scale_b, const float* shift_b,
                             int32_t relu, float* y, int64_t ldy, void* stream) {
    B2M_CHECK_ARG(xa && xb && y && scale_a && shift_a && scale_b && shift_b && c > 0 && c % 4 == 0 && c <= 1024 &&
                      lda % 4 == 0 && ldb % 4 == 0 && ldy % 4 == 0, "c and leading dimensions must be multiples of 4, c <= 1024");
    if (n == 0) return B2M_OK;
    bn_apply2_kernel<<<row_grid(n, c / 4), 256, 0, (hipStream_t)stream>>>(xa, lda, xb, ldb, n, c / 4, scale_a, shift_a, scale_b,
                                                                          shift_b, relu, y, ldy);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}
// backward reduction of the pair: partial[blk][3c] = (sum g, sum g*xhat_a, sum g*xhat_b), g = dy * (y > 0 if relu).
// Same thread -> (column group, row slot) mapping, row ranges and summation order as column_reduce_staged, so the first two
// sums are bit for bit those of bn_bwd_reduce_kernel on (dy, y, x_a).
__global__ __launch_bounds__(256) void bn_bwd_reduce2_kernel(const float* __restrict__ dy, int64_t lddy,
                                                             const float* __restrict__ y, int64_t ldy,
                                                             const float* __restrict__ xa, int64_t lda,
                                                             const float* __restrict__ xb, int64_t ldb, int64_t n, int c,
                                                             const float* __restrict__ mean_a, const float* __restrict__ invstd_a,
                                                             const float* __restrict__ mean_b, const float* __restrict__ invstd_b,
                                                             int relu, double* __restrict__ partial) {
    extern __shared__ float red[];             // [nslots][c4][12]
    const int c4 = c >> 2;
    const int nslots = 256 / c4;
    const int cg = threadIdx.x % c4, rs = threadIdx.x / c4;
    const int64_t rows_per_blk = (n + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_blk;
    int64_t r1 = r0 + rows_per_blk;
    if (r1 > n) r1 = n;
    f32x4 s0 = {0, 0, 0, 0}, s1 = {0, 0, 0, 0}, s2 = {0, 0, 0, 0};
    if (rs < nslots) {
        const f32x4 ma = *(const f32x4*)(mean_a + cg * 4), ia = *(const f32x4*)(invstd_a + cg * 4);
        const f32x4 mb = *(const f32x4*)(mean_b + cg * 4), ib = *(const f32x4*)(invstd_b + cg * 4);
        auto term = [&](const f32x4& g0, const f32x4& yy, const f32x4& va, const f32x4& vb) {
            f32x4 g = g0;
            if (relu) {
#pragma unroll
                for (int u = 0; u < 4; ++u) g[u] = yy[u] > 0.f ? g[u] : 0.f;
            }
            s0 += g; s1 += g * ((va - ma) * ia); s2 += g * ((vb - mb) * ib);
        };
        int64_t r = r0 + rs;
        for (; r + 3 * nslots < r1; r += 4 * nslots) {
            f32x4 in[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t rr = r + (int64_t)u * nslots;
                in[u][0] = *(const f32x4*)(dy + rr * lddy + cg * 4);
                in[u][1] = *(const f32x4*)(y + rr * ldy + cg * 4);
                in[u][2] = *(const f32x4*)(xa + rr * lda + cg * 4);
                in[u][3] = *(const f32x4*)(xb + rr * ldb + cg * 4);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; ++u) term(in[u][0], in[u][1], in[u][2], in[u][3]);
        }
        for (; r < r1; r += nslots)
            term(*(const f32x4*)(dy + r * lddy + cg * 4), *(const f32x4*)(y + r * ldy + cg * 4),
                 *(const f32x4*)(xa + r * lda + cg * 4), *(const f32x4*)(xb + r * ldb + cg * 4));
        float* p = red + ((size_t)rs * c4 + cg) * 12;
        *(f32x4*)p = s0; *(f32x4*)(p + 4) = s1; *(f32x4*)(p + 8) = s2;
    }
    __syncthreads();
    if (rs == 0) {
        double d[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int s_ = 0; s_ < nslots; ++s_) {
            const float* p = red + ((size_t)s_ * c4 + cg) * 12;
#pragma unroll
            for (int u = 0; u < 12; ++u) d[u] += (double)p[u];
        }
        double* o = partial + (size_t)blockIdx.x * 3 * c;
#pragma unroll
        for (int u = 0; u < 4; ++u) { o[cg * 4 + u] = d[u]; o[c + cg * 4 + u] = d[4 + u]; o[2 * c + cg * 4 + u] = d[8 + u]; }
    }
}
extern "C" int b2m_bn_bwd_reduce2(const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* xa, int64_t lda,
                                  const float* xb, int64_t ldb, int64_t n, int32_t c, const float* mean_a,
                                  const float* invstd_a, const float* mean_b, const float* invstd_b, int32_t relu,
                                  double* partial, double* sums, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    B2M_CHECK_ARG(dy && y && xa && xb && mean_a && invstd_a && mean_b && invstd_b && partial && sums, "NULL argument");
    B2M_CHECK_ARG(c > 0 && c % 4 == 0 && c <= 1024 && lddy % 4 == 0 && ldy % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0,
                  "c and leading dimensions must be multiples of 4");
    const int nblk = reduce_blocks(n);
    const int c4 = c / 4, nslots = 256 / c4;
    bn_bwd_reduce2_kernel<<<nblk, 256, (size_t)nslots * c4 * 12 * sizeof(float), st>>>(dy, lddy, y, ldy, xa, lda, xb, ldb, n, c,
        mean_a, invstd_a, mean_b, invstd_b, relu, partial);
    reduce_final_kernel<<<3 * c, 64, 0, st>>>(partial, nblk, 3 * c, sums, nullptr, nullptr);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}
__global__ __launch_bounds__(256) void bn_bwd_apply2_kernel(const float* __restrict__ dy, int64_t lddy,
                                                            const float* __restrict__ y, int64_t ldy,
                                                            const float* __restrict__ xa, int64_t lda,
                                                            const float* __restrict__ xb, int64_t ldb, int64_t n, int c,
                                                            const float* __restrict__ mean_a, const float* __restrict__ invstd_a,
                                                            const float* __restrict__ gamma_a, const float* __restrict__ mean_b,
                                                            const float* __restrict__ invstd_b, const float* __restrict__ gamma_b,
                                                            const double* __restrict__ sums, double count_host,
                                                            const double* __restrict__ count_dev, int relu,
                                                            float* __restrict__ dxa, int64_t lddxa, float* __restrict__ dxb,
                                                            int64_t lddxb, float* __restrict__ dbeta_a, float* __restrict__ dgamma_a,
                                                            float* __restrict__ dbeta_b, float* __restrict__ dgamma_b,
                                                            const double* __restrict__ psums) {
    const int c4 = c >> 2;
    const float inv_n = (float)(1.0 / (count_dev ? *count_dev : count_host));
    const int nslots = 256 / c4;
    const int cg = threadIdx.x % c4, rs = threadIdx.x / c4;
    if (rs >= nslots) return;
    const f32x4 ma = *(const f32x4*)(mean_a + cg * 4), ia = *(const f32x4*)(invstd_a + cg * 4);
    const f32x4 mb = *(const f32x4*)(mean_b + cg * 4), ib = *(const f32x4*)(invstd_b + cg * 4);
    f32x4 sg, sga, sgb, ga, gb;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int j = cg * 4 + u;
        sg[u] = (float)sums[j] * inv_n; sga[u] = (float)sums[c + j] * inv_n; sgb[u] = (float)sums[2 * c + j] * inv_n;
        ga[u] = (gamma_a ? gamma_a[j] : 1.f) * ia[u]; gb[u] = (gamma_b ? gamma_b[j] : 1.f) * ib[u];
        if (blockIdx.x == 0 && rs == 0) {        // the parameter gradients: fp32 copies of THIS RANK's sums (the gradient
            // all-reduce averages them over the ranks like every other parameter gradient; torch SyncBatchNorm does the same)
            if (dbeta_a) dbeta_a[j] = (float)psums[j];
            if (dbeta_b) dbeta_b[j] = (float)psums[j];
            if (dgamma_a) dgamma_a[j] = (float)psums[c + j];
            if (dgamma_b) dgamma_b[j] = (float)psums[2 * c + j];
        }
    }
    for (int64_t r = (int64_t)blockIdx.x * nslots + rs; r < n; r += (int64_t)gridDim.x * nslots) {
        f32x4 g = *(const f32x4*)(dy + r * lddy + cg * 4);
        const f32x4 va = *(const f32x4*)(xa + r * lda + cg * 4), vb = *(const f32x4*)(xb + r * ldb + cg * 4);
        if (relu) {
            const f32x4 yy = *(const f32x4*)(y + r * ldy + cg * 4);
#pragma unroll
            for (int u = 0; u < 4; ++u) g[u] = yy[u] > 0.f ? g[u] : 0.f;
        }
        f32x4 oa, ob;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            oa[u] = ga[u] * (g[u] - sg[u] - ((va[u] - ma[u]) * ia[u]) * sga[u]);
            ob[u] = gb[u] * (g[u] - sg[u] - ((vb[u] - mb[u]) * ib[u]) * sgb[u]);
        }
        *(f32x4*)(dxa + r * lddxa + cg * 4) = oa;
        *(f32x4*)(dxb + r * lddxb + cg * 4) = ob;
    }
}
extern "C" int b2m_bn_bwd_apply2(const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* xa, int64_t lda,
                                 const float* xb, int64_t ldb, int64_t n, int32_t c, const float* mean_a,
                                 const float* invstd_a, const float* gamma_a, const float* mean_b, const float* invstd_b,
                                 const float* gamma_b, const double* sums, double count, const double* count_dev,
                                 int32_t relu, float* dxa, int64_t lddxa, float* dxb, int64_t lddxb, float* dbeta_a,
                                 float* dgamma_a, float* dbeta_b, float* dgamma_b, const double* local_sums, void* stream) {
    B2M_CHECK_ARG(dy && y && xa && xb && mean_a && invstd_a && mean_b && invstd_b && sums && dxa && dxb, "NULL argument");
    B2M_CHECK_ARG(c > 0 && c % 4 == 0 && c <= 1024 && lddy % 4 == 0 && ldy % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 &&
                      lddxa % 4 == 0 && lddxb % 4 == 0 && (count_dev || count >= 1),
                  "c and leading dimensions must be multiples of 4");
    if (n == 0) return B2M_OK;
    bn_bwd_apply2_kernel<<<row_grid(n, c / 4), 256, 0, (hipStream_t)stream>>>(dy, lddy, y, ldy, xa, lda, xb, ldb, n, c, mean_a,
        invstd_a, gamma_a, mean_b, invstd_b, gamma_b, sums, count, count_dev, relu, dxa, lddxa, dxb, lddxb, dbeta_a, dgamma_a,
        dbeta_b, dgamma_b, local_sums ? local_sums : sums);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}

// ------------------------------------------------------------------ BatchNorm of small maps in ONE launch
// The deep U-Net levels (a few hundred to a few thousand rows) and the heads' segment rows: the tensors live in L2, and the
// five launches per layer (statistics, their final sum + finalize, apply; backward: reduce, final sum, apply) were latency,
// not bandwidth -- 38 of the 89 BatchNorm layers of a ScanNet step have <= 4 k rows.  Here one workgroup owns FOUR
// channels (one 16-byte column group) for ALL rows: pass 1 sums in fp64 per thread (thread = row slot, fixed order), the
// block combines in a fixed tree, the finalize math runs in the block, pass 2 re-reads the rows (L2 hits) and writes.
// Deterministic; the statistics are fp64 sums of the fp32 inputs exactly like the two-stage kernels (other grouping).
#define BN_SMALL_THREADS 256
// block-wide sum of NV doubles per thread -> every thread gets the totals (fixed order: lanes by shuffle, waves in order)
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* sh /* [4][NV] */) {
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        double t = v[u];
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) t += __shfl_down(t, d, 64);
        v[u] = t;
    }
    const int wave = threadIdx.x >> 6;
    __syncthreads();                                    // (sh may still be read from a previous use)
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int u = 0; u < NV; ++u) sh[wave * NV + u] = v[u];
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NV; ++u) v[u] = (sh[u] + sh[NV + u]) + (sh[2 * NV + u] + sh[3 * NV + u]);
}
// PHASE 0: the whole layer.  SyncBN (the statistics of all ranks meet between the two halves: b2m_bn_small_fwd_stats ->
// all-reduce of xchg[2c + 1] -> b2m_bn_small_fwd_apply): PHASE 1 = pass 1 only, this rank's column sums and row count to
// xchg; PHASE 2 = finalize from the all-reduced xchg + pass 2.
template <int PHASE>
__global__ __launch_bounds__(BN_SMALL_THREADS) void bn_small_fwd_kernel(
        const float* __restrict__ x, int64_t ldx, int n, int c, const float* __restrict__ gamma, const float* __restrict__ beta,
        float eps, float momentum, float* __restrict__ running_mean, float* __restrict__ running_var,
        float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ scale, float* __restrict__ shift,
        const float* __restrict__ res, int64_t ldr, int relu, float* __restrict__ y, int64_t ldy, double* __restrict__ xchg) {
    __shared__ double sh[4 * 8];
    __shared__ float sc[8];
    const int col = blockIdx.x * 4;
    double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if constexpr (PHASE != 2) {
        for (int r = threadIdx.x; r < n; r += BN_SMALL_THREADS) {
            const f32x4 v = *(const f32x4*)(x + (int64_t)r * ldx + col);
#pragma unroll
            for (int u = 0; u < 4; ++u) { const double d = (double)v[u]; s[u] += d; s[4 + u] = fma(d, d, s[4 + u]); }
        }
        block_sum<8>(s, sh);
    }
    if constexpr (PHASE == 1) {
        if (threadIdx.x < 4) { xchg[col + threadIdx.x] = s[threadIdx.x]; xchg[c + col + threadIdx.x] = s[4 + threadIdx.x]; }
        if (blockIdx.x == 0 && threadIdx.x == 0) xchg[2 * c] = (double)n;
        return;
    }
    if (threadIdx.x < 4) {
        const int j = col + threadIdx.x, u = threadIdx.x;
        double count = (double)n;
        if constexpr (PHASE == 2) { s[u] = xchg[j]; s[4 + u] = xchg[c + j]; count = xchg[2 * c]; }
        const double m = s[u] / count;
        double var = s[4 + u] / count - m * m;
        if (var < 0) var = 0;
        if (running_mean) {
            const double unb = count > 1 ? var * count / (count - 1) : var;
            running_mean[j] = (float)((1.0 - momentum) * (double)running_mean[j] + momentum * m);
            running_var[j] = (float)((1.0 - momentum) * (double)running_var[j] + momentum * unb);
        }
        const double is = 1.0 / sqrt(var + (double)eps);
        const double g = gamma ? (double)gamma[j] : 1.0, b = beta ? (double)beta[j] : 0.0;
        mean[j] = (float)m; invstd[j] = (float)is;
        const float fs = (float)(g * is), fb = (float)(b - m * g * is);
        scale[j] = fs; shift[j] = fb;
        sc[u] = fs; sc[4 + u] = fb;
    }
    __syncthreads();
    f32x4 a, b;
#pragma unroll
    for (int u = 0; u < 4; ++u) { a[u] = sc[u]; b[u] = sc[4 + u]; }
    for (int r = threadIdx.x; r < n; r += BN_SMALL_THREADS) {
        f32x4 v = *(const f32x4*)(x + (int64_t)r * ldx + col);
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = __builtin_fmaf(v[u], a[u], b[u]);       // (the backward recomputes this sign)
        if (res) v += *(const f32x4*)(res + (int64_t)r * ldr + col);
        if (relu) {
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = v[u] > 0.f ? v[u] : 0.f;
        }
        *(f32x4*)(y + (int64_t)r * ldy + col) = v;
    }
}
extern "C" int b2m_bn_small_fwd(const float* x, int64_t ldx, int64_t n, int32_t c, const float* gamma, const float* beta,
                                float eps, float momentum, float* running_mean, float* running_var, float* mean,
                                float* invstd, float* scale, float* shift, const float* residual, int64_t ldr, int32_t relu,
                                float* y, int64_t ldy, void* stream) {
    B2M_CHECK_ARG(x && y && mean && invstd && scale && shift && n >= 1 && n <= B2M_BN_SMALL_MAX_ROWS, "bad arguments / too many rows");
    B2M_CHECK_ARG(c > 0 && c % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && (!residual || ldr % 4 == 0) && ldx >= c && ldy >= c,
                  "c and leading dimensions must be multiples of 4");
    B2M_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)residual % 16) == 0, "16-byte alignment");
    B2M_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "running statistics: both or none");
    bn_small_fwd_kernel<0><<<c / 4, BN_SMALL_THREADS, 0, (hipStream_t)stream>>>(x, ldx, (int)n, c, gamma, beta, eps, momentum,
        running_mean, running_var, mean, invstd, scale, shift, residual, ldr, relu, y, ldy, nullptr);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}
extern "C" int b2m_bn_small_fwd_stats(const float* x, int64_t ldx, int64_t n, int32_t c, double* xchg, void* stream) {
    B2M_CHECK_ARG(x && xchg && n >= 1 && n <= B2M_BN_SMALL_MAX_ROWS, "bad arguments / too many rows");
    B2M_CHECK_ARG(c > 0 && c % 4 == 0 && ldx % 4 == 0 && ldx >= c && ((uintptr_t)x % 16) == 0, "c, ldx multiples of 4; 16-byte alignment");
    bn_small_fwd_kernel<1><<<c / 4, BN_SMALL_THREADS, 0, (hipStream_t)stream>>>(x, ldx, (int)n, c, nullptr, nullptr, 0.f, 0.f, nullptr,
        nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, 0, xchg);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}
extern "C" int b2m_bn_small_fwd_apply(const double* xchg, const float* x, int64_t ldx, int64_t n, int32_t c, const float* gamma,
                                      const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                                      float* mean, float* invstd, float* scale, float* shift, const float* residual, int64_t ldr,
                                      int32_t relu, float* y, int64_t ldy, void* stream) {
    B2M_CHECK_ARG(xchg && x && y && mean && invstd && scale && shift && n >= 1 && n <= B2M_BN_SMALL_MAX_ROWS, "bad arguments / too many rows");
    B2M_CHECK_ARG(c > 0 && c % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && (!residual || ldr % 4 == 0) && ldx >= c && ldy >= c,
                  "c and leading dimensions must be multiples of 4");
    B2M_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)residual % 16) == 0, "16-byte alignment");
    B2M_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "running statistics: both or none");
    bn_small_fwd_kernel<2><<<c / 4, BN_SMALL_THREADS, 0, (hipStream_t)stream>>>(x, ldx, (int)n, c, gamma, beta, eps, momentum,
        running_mean, running_var, mean, invstd, scale, shift, residual, ldr, relu, y, ldy, (double*)xchg);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}

// PHASE as in bn_small_fwd_kernel: 1 = reduction only (this rank's sums to xchg[2c], dbeta / dgamma from them), 2 = apply from
// the all-reduced xchg[2c] with the global row count *count_dev.
template <bool RELU, bool HASY, int PHASE = 0>
__global__ __launch_bounds__(BN_SMALL_THREADS) void bn_small_bwd_kernel(
        const float* __restrict__ dy, int64_t lddy, const float* __restrict__ y, int64_t ldy, const float* __restrict__ x,
        int64_t ldx, int n, int c, const float* __restrict__ mean, const float* __restrict__ invstd,
        const float* __restrict__ gamma, const float* __restrict__ mscale, const float* __restrict__ mshift,
        float* __restrict__ dbeta, float* __restrict__ dgamma, float* __restrict__ dx, int64_t lddx,
        float* __restrict__ dres, int64_t lddres, double* __restrict__ xchg = nullptr, const double* __restrict__ count_dev = nullptr) {
    __shared__ double sh[4 * 8];
    const int col = blockIdx.x * 4;
    const f32x4 m = *(const f32x4*)(mean + col), is = *(const f32x4*)(invstd + col);
    f32x4 ms = {0.f, 0.f, 0.f, 0.f}, mb = {0.f, 0.f, 0.f, 0.f};
    if (RELU && !HASY) { ms = *(const f32x4*)(mscale + col); mb = *(const f32x4*)(mshift + col); }
    auto masked = [&](int r, f32x4& g, f32x4& xx) {
        g = *(const f32x4*)(dy + (int64_t)r * lddy + col);
        xx = *(const f32x4*)(x + (int64_t)r * ldx + col);
        if constexpr (RELU) {
            f32x4 yy;
            if constexpr (HASY) yy = *(const f32x4*)(y + (int64_t)r * ldy + col);
            else {
#pragma unroll
                for (int u = 0; u < 4; ++u) yy[u] = __builtin_fmaf(xx[u], ms[u], mb[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) g[u] = yy[u] > 0.f ? g[u] : 0.f;
        }
    };
    double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if constexpr (PHASE != 2) {
        for (int r = threadIdx.x; r < n; r += BN_SMALL_THREADS) {
            f32x4 g, xx;
            masked(r, g, xx);
#pragma unroll
            for (int u = 0; u < 4; ++u) { s[u] += (double)g[u]; s[4 + u] += (double)(g[u] * ((xx[u] - m[u]) * is[u])); }
        }
        block_sum<8>(s, sh);
        if (threadIdx.x < 4) {           // (the parameter gradients: THIS rank's sums; the gradient all-reduce averages them)
            if (dbeta) dbeta[col + threadIdx.x] = (float)s[threadIdx.x];
            if (dgamma) dgamma[col + threadIdx.x] = (float)s[4 + threadIdx.x];
        }
    }
    if constexpr (PHASE == 1) {
        if (threadIdx.x < 4) { xchg[col + threadIdx.x] = s[threadIdx.x]; xchg[c + col + threadIdx.x] = s[4 + threadIdx.x]; }
        return;
    }
    if constexpr (PHASE == 2) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { s[u] = xchg[col + u]; s[4 + u] = xchg[c + col + u]; }
    }
    const float inv_n = (float)(1.0 / (PHASE == 2 ? *count_dev : (double)n));
    f32x4 sg, sgx, ga;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        sg[u] = (float)s[u] * inv_n; sgx[u] = (float)s[4 + u] * inv_n;
        ga[u] = (gamma ? gamma[col + u] : 1.f) * is[u];
    }
    for (int r = threadIdx.x; r < n; r += BN_SMALL_THREADS) {
        f32x4 g, xx, out;
        masked(r, g, xx);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float xh = (xx[u] - m[u]) * is[u];
            out[u] = ga[u] * (g[u] - sg[u] - xh * sgx[u]);
        }
        *(f32x4*)(dx + (int64_t)r * lddx + col) = out;
        if (dres) *(f32x4*)(dres + (int64_t)r * lddres + col) = g;
    }
}
extern "C" int b2m_bn_small_bwd(const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* x, int64_t ldx,
                                int64_t n, int32_t c, const float* mean, const float* invstd, const float* gamma, int32_t relu,
                                const float* mask_scale, const float* mask_shift, float* dbeta_f32, float* dgamma_f32,
                                float* dx, int64_t lddx, float* dres, int64_t lddres, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    B2M_CHECK_ARG(dy && x && mean && invstd && dx && (!relu || y || (mask_scale && mask_shift)) && n >= 1 &&
                      n <= B2M_BN_SMALL_MAX_ROWS, "NULL argument / too many rows");
    B2M_CHECK_ARG(c > 0 && c % 4 == 0 && lddy % 4 == 0 && ldx % 4 == 0 && lddx % 4 == 0 && (!(relu && y) || ldy % 4 == 0) &&
                      (!dres || lddres % 4 == 0), "c and leading dimensions must be multiples of 4");
    B2M_CHECK_ARG(((uintptr_t)dy % 16) == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)dx % 16) == 0 &&
                      ((uintptr_t)y % 16) == 0 && ((uintptr_t)dres % 16) == 0, "16-byte alignment");
    const dim3 grid(c / 4);
    if (!relu) bn_small_bwd_kernel<false, false><<<grid, BN_SMALL_THREADS, 0, st>>>(dy, lddy, nullptr, 0, x, ldx, (int)n, c, mean, invstd, gamma, nullptr, nullptr, dbeta_f32, dgamma_f32, dx, lddx, dres, lddres);
    else if (y) bn_small_bwd_kernel<true, true><<<grid, BN_SMALL_THREADS, 0, st>>>(dy, lddy, y, ldy, x, ldx, (int)n, c, mean, invstd, gamma, nullptr, nullptr, dbeta_f32, dgamma_f32, dx, lddx, dres, lddres);
    else bn_small_bwd_kernel<true, false><<<grid, BN_SMALL_THREADS, 0, st>>>(dy, lddy, nullptr, 0, x, ldx, (int)n, c, mean, invstd, gamma, mask_scale, mask_shift, dbeta_f32, dgamma_f32, dx, lddx, dres, lddres);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}
// SyncBN halves of b2m_bn_small_bwd: phase 1 leaves this rank's (sum g, sum g * xhat) in xchg[2c] (and dbeta / dgamma from
// them), the caller all-reduces xchg, phase 2 applies with the global row count *count_dev.
extern "C" int b2m_bn_small_bwd_phase(int32_t phase, const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* x,
                                      int64_t ldx, int64_t n, int32_t c, const float* mean, const float* invstd, const float* gamma,
                                      int32_t relu, const float* mask_scale, const float* mask_shift, float* dbeta_f32,
                                      float* dgamma_f32, float* dx, int64_t lddx, float* dres, int64_t lddres, double* xchg,
                                      const double* count_dev, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    B2M_CHECK_ARG((phase == 1 || phase == 2) && xchg && (phase == 1 || (count_dev && dx)), "phase is 1 or 2; xchg / count_dev / dx");
    B2M_CHECK_ARG(dy && x && mean && invstd && (!relu || y || (mask_scale && mask_shift)) && n >= 1 &&
                      n <= B2M_BN_SMALL_MAX_ROWS, "NULL argument / too many rows");
    B2M_CHECK_ARG(c > 0 && c % 4 == 0 && lddy % 4 == 0 && ldx % 4 == 0 && (phase == 1 || lddx % 4 == 0) && (!(relu && y) || ldy % 4 == 0) &&
                      (!dres || lddres % 4 == 0), "c and leading dimensions must be multiples of 4");
    B2M_CHECK_ARG(((uintptr_t)dy % 16) == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)dx % 16) == 0 &&
                      ((uintptr_t)y % 16) == 0 && ((uintptr_t)dres % 16) == 0, "16-byte alignment");
    const dim3 grid(c / 4);
#define B2M_SMALL_BWD(R, H, P) bn_small_bwd_kernel<R, H, P><<<grid, BN_SMALL_THREADS, 0, st>>>(dy, lddy, (H) ? y : nullptr, (H) ? ldy : 0, x, ldx, \
        (int)n, c, mean, invstd, gamma, (R) && !(H) ? mask_scale : nullptr, (R) && !(H) ? mask_shift : nullptr, dbeta_f32, dgamma_f32, dx, lddx,   \
        dres, lddres, xchg, count_dev)
    if (phase == 1) {
        if (!relu) B2M_SMALL_BWD(false, false, 1); else if (y) B2M_SMALL_BWD(true, true, 1); else B2M_SMALL_BWD(true, false, 1);
    } else {
        if (!relu) B2M_SMALL_BWD(false, false, 2); else if (y) B2M_SMALL_BWD(true, true, 2); else B2M_SMALL_BWD(true, false, 2);
    }
#undef B2M_SMALL_BWD
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}

// ------------------------------------------------------------------ elementwise
__global__ void relu_fwd_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ y) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        float v = x[e]; y[e] = v > 0.f ? v : 0.f;
    }
}
__global__ void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, int64_t n,
                                float* __restrict__ dx) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
        dx[e] = y[e] > 0.f ? dy[e] : 0.f;
}
__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n, float* __restrict__ o) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
        o[e] = a[e] + b[e];
}
extern "C" int b2m_relu_fwd(const float* x, int64_t n_elem, float* y, void* stream) {
    if (n_elem == 0) return B2M_OK;
    relu_fwd_kernel<<<ew_grid(n_elem), 256, 0, (hipStream_t)stream>>>(x, n_elem, y);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}
extern "C" int b2m_relu_bwd(const float* dy, const float* y, int64_t n_elem, float* dx, void* stream) {
    if (n_elem == 0) return B2M_OK;
    relu_bwd_kernel<<<ew_grid(n_elem), 256, 0, (hipStream_t)stream>>>(dy, y, n_elem, dx);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}
extern "C" int b2m_add(const float* a, const float* b, int64_t n_elem, float* out, void* stream) {
    if (n_elem == 0) return B2M_OK;
    add_kernel<<<ew_grid(n_elem), 256, 0, (hipStream_t)stream>>>(a, b, n_elem, out);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}

// ------------------------------------------------------------------ segment pooling
__global__ void pool_count_kernel(const int64_t* __restrict__ ids, int64_t n, int32_t* __restrict__ counts) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicAdd(&counts[ids[i]], 1);
}
// Rows arrive in Morton order, so consecutive rows mostly belong to the same segment: a block walks a run of
// POOL_RUN rows, thread = channel, and flushes its running sum with one atomic per channel only when the segment
// changes (a few atomics per run instead of one per element: 0.48 -> 0.2 ms on a 1.2 M x 96 batch).
#define POOL_RUN 64
__global__ __launch_bounds__(256) void pool_sum_kernel(const float* __restrict__ x, int64_t ldx, int64_t n, int c,
                                                       const int64_t* __restrict__ ids, float* __restrict__ out,
                                                       int32_t* __restrict__ counts) {
    const int64_t r0 = (int64_t)blockIdx.x * POOL_RUN;
    int64_t r1 = r0 + POOL_RUN;
    if (r1 > n) r1 = n;
    for (int col = threadIdx.x; col < c; col += blockDim.x) {       // one pass unless c > 256
        int64_t cur = ids[r0];
        float acc = 0.f;
        int len = 0;                                                 // rows of the current segment (channel 0 counts)
        int64_t r = r0;
        for (; r + 3 < r1; r += 4) {                                 // four rows' loads in flight, then in row order
            int64_t id4[4]; float v4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { id4[u] = ids[r + u]; v4[u] = x[(r + u) * ldx + col]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (id4[u] != cur) {
                    atomicAdd(&out[cur * c + col], acc);
                    if (col == 0) atomicAdd(&counts[cur], len);
                    acc = 0.f; len = 0; cur = id4[u];
                }
                acc += v4[u];
                ++len;
            }
        }
        for (; r < r1; ++r) {
            const int64_t id = ids[r];
            if (id != cur) {
                atomicAdd(&out[cur * c + col], acc);
                if (col == 0) atomicAdd(&counts[cur], len);
                acc = 0.f; len = 0; cur = id;
            }
            acc += x[r * ldx + col];
            ++len;
        }
        atomicAdd(&out[cur * c + col], acc);
        if (col == 0) atomicAdd(&counts[cur], len);
    }
}
// Deterministic segment mean (B2M_DETERMINISTIC=1): rows arrive grouped by segment through `order` (stable sort of
// the ids, done by the caller), one workgroup per segment, thread = channel, four interleaved partial sums combined
// in a fixed association -- no atomics, the same bits on every run.
__global__ __launch_bounds__(256) void pool_mean_sorted_kernel(const float* __restrict__ x, int64_t ldx, int c,
                                                               const int64_t* __restrict__ order,
                                                               const int64_t* __restrict__ seg_start,
                                                               float* __restrict__ out, int32_t* __restrict__ counts) {
    const int64_t s = blockIdx.x;
    const int64_t r0 = seg_start[s], r1 = seg_start[s + 1];
    if (threadIdx.x == 0) counts[s] = (int32_t)(r1 - r0);
    for (int col = threadIdx.x; col < c; col += blockDim.x) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int64_t r = r0;
        for (; r + 4 <= r1; r += 4) {
            const int64_t i0 = order[r], i1 = order[r + 1], i2 = order[r + 2], i3 = order[r + 3];
            a0 += x[i0 * ldx + col]; a1 += x[i1 * ldx + col]; a2 += x[i2 * ldx + col]; a3 += x[i3 * ldx + col];
        }
        for (; r < r1; ++r) a0 += x[order[r] * ldx + col];
        const float sum = (a0 + a1) + (a2 + a3);
        out[s * c + col] = r1 > r0 ? sum / (float)(r1 - r0) : 0.f;
    }
}
extern "C" int b2m_segment_mean_sorted(const float* x, int64_t ldx, int64_t n, int32_t c, const int64_t* order,
                                       const int64_t* seg_start, int64_t n_seg, float* out, int32_t* counts,
                                       void* stream) {
    B2M_CHECK_ARG(x && order && seg_start && out && counts && c > 0 && ldx >= c && n >= 0 && n_seg >= 0, "bad arguments");
    if (n_seg == 0) return B2M_OK;
    B2M_CHECK_ARG(n_seg < (1ll << 31), "too many segments");
    pool_mean_sorted_kernel<<<(unsigned)n_seg, (unsigned)(c >= 256 ? 256 : (c + 63) / 64 * 64), 0, (hipStream_t)stream>>>(
        x, ldx, c, order, seg_start, out, counts);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}
__global__ void pool_div_kernel(float* __restrict__ out, const int32_t* __restrict__ counts, int64_t n_seg, int c) {
    const int64_t total = n_seg * c;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int cnt = counts[e / c];
        out[e] = cnt > 0 ? out[e] / (float)cnt : 0.f;
    }
}
// max: pack (order-preserving float bits, ~row) into 64 bits; atomicMax picks the largest value and,
// among equal values, the lowest row -> deterministic argmax.
__device__ __forceinline__ uint32_t f2ord(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o) {
    uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
    return __uint_as_float(u);
}
__global__ __launch_bounds__(256) void pool_max_kernel(const float* __restrict__ x, int64_t ldx, int64_t n, int c,
                                                       const int64_t* __restrict__ ids,
                                                       unsigned long long* __restrict__ packed) {
    const int64_t total = n * c;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / c; const int col = (int)(e - r * c);
        unsigned long long p = ((unsigned long long)f2ord(x[r * ldx + col]) << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)r);
        atomicMax(&packed[ids[r] * c + col], p);
    }
}
__global__ void pool_max_decode_kernel(const unsigned long long* __restrict__ packed, int64_t total,
                                       float* __restrict__ out, int32_t* __restrict__ argmax) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        unsigned long long p = packed[e];
        if (p == 0ull) { out[e] = 0.f; argmax[e] = -1; }
        else { out[e] = ord2f((uint32_t)(p >> 32)); argmax[e] = (int32_t)(0xFFFFFFFFu - (uint32_t)p); }
    }
}
extern "C" int b2m_segment_pool_fwd(const float* x, int64_t ldx, int64_t n, int32_t c, const int64_t* ids,
                                    int64_t n_seg, int32_t mode, float* out, int32_t* counts, int32_t* argmax,
                                    uint64_t* scratch, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    B2M_CHECK_ARG(x && ids && out && counts && c > 0 && ldx >= c && n_seg >= 0, "bad arguments");
    B2M_CHECK_ARG(mode == 0 || (mode == 1 && argmax && scratch), "mode 1 (max) needs argmax and scratch");
    if (n_seg == 0) return B2M_OK;
    B2M_HIP(hipMemsetAsync(counts, 0, n_seg * sizeof(int32_t), st));
    if (mode == 0) {            // the sum kernel counts the rows per segment as it goes
        B2M_HIP(hipMemsetAsync(out, 0, (size_t)n_seg * c * sizeof(float), st));
        if (n > 0) pool_sum_kernel<<<(unsigned)cdiv64(n, POOL_RUN), (unsigned)(c >= 256 ? 256 : (c + 63) / 64 * 64), 0, st>>>(x, ldx, n, c, ids, out, counts);
        pool_div_kernel<<<ew_grid(n_seg * c), 256, 0, st>>>(out, counts, n_seg, c);
    } else {
        if (n > 0) pool_count_kernel<<<(unsigned)cdiv64(n, 256), 256, 0, st>>>(ids, n, counts);
        B2M_HIP(hipMemsetAsync(scratch, 0, (size_t)n_seg * c * sizeof(uint64_t), st));
        if (n > 0) pool_max_kernel<<<ew_grid(n * c), 256, 0, st>>>(x, ldx, n, c, ids, (unsigned long long*)scratch);
        pool_max_decode_kernel<<<ew_grid(n_seg * c), 256, 0, st>>>((const unsigned long long*)scratch, n_seg * c, out,
                                                                   argmax);
    }
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}
__global__ __launch_bounds__(256) void pool_bwd_kernel(const float* __restrict__ dout, int64_t n, int c,
                                                       const int64_t* __restrict__ ids, int mode,
                                                       const int32_t* __restrict__ counts,
                                                       const int32_t* __restrict__ argmax, float* __restrict__ dx,
                                                       int64_t lddx) {
    const int64_t total = n * c;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / c; const int col = (int)(e - r * c);
        const int64_t s = ids[r];
        float g = dout[s * c + col];
        if (mode == 0) g = g / (float)counts[s];
        else g = argmax[s * c + col] == (int32_t)r ? g : 0.f;
        dx[r * lddx + col] = g;
    }
}
// the same with 16-byte accesses: thread -> (row slot, float4 column group), rows walked with a constant stride (no
// per-element division); c % 4 == 0, dx rows and dout 16-byte aligned
__global__ __launch_bounds__(256) void pool_bwd_vec_kernel(const float* __restrict__ dout, int64_t n, int c4,
                                                           const int64_t* __restrict__ ids, int mode,
                                                           const int32_t* __restrict__ counts,
                                                           const int32_t* __restrict__ argmax, float* __restrict__ dx,
                                                           int64_t lddx) {
    const int nslots = 256 / c4;
    const int cg = threadIdx.x % c4, rs = threadIdx.x / c4;
    if (rs >= nslots) return;
    const int c = c4 * 4;
    for (int64_t r = (int64_t)blockIdx.x * nslots + rs; r < n; r += (int64_t)gridDim.x * nslots) {
        const int64_t s = ids[r];
        f32x4 g = *(const f32x4*)(dout + s * c + cg * 4);
        if (mode == 0) {
            const float cnt = (float)counts[s];
#pragma unroll
            for (int u = 0; u < 4; ++u) g[u] = g[u] / cnt;              // (a division, as the scalar kernel: the same bits)
        } else {
            const i32x4 a = *(const i32x4*)(argmax + s * c + cg * 4);
#pragma unroll
            for (int u = 0; u < 4; ++u) g[u] = a[u] == (int32_t)r ? g[u] : 0.f;
        }
        *(f32x4*)(dx + r * lddx + cg * 4) = g;
    }
}
extern "C" int b2m_segment_pool_bwd(const float* dout, int64_t n, int32_t c, const int64_t* ids, int64_t n_seg,
                                    int32_t mode, const int32_t* counts, const int32_t* argmax, float* dx,
                                    int64_t lddx, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    B2M_CHECK_ARG(dout && ids && dx && c > 0 && lddx >= c, "bad arguments");
    B2M_CHECK_ARG((mode == 0 && counts) || (mode == 1 && argmax), "mode 0 needs counts, mode 1 needs argmax");
    if (n == 0) return B2M_OK;
    if (c % 4 == 0 && c <= 1024 && lddx % 4 == 0 && ((uintptr_t)dout % 16) == 0 && ((uintptr_t)dx % 16) == 0 &&
        (mode == 0 || ((uintptr_t)argmax % 16) == 0))
        pool_bwd_vec_kernel<<<row_grid(n, c / 4), 256, 0, st>>>(dout, n, c / 4, ids, mode, counts, argmax, dx, lddx);
    else
        pool_bwd_kernel<<<ew_grid(n * c), 256, 0, st>>>(dout, n, c, ids, mode, counts, argmax, dx, lddx);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}
