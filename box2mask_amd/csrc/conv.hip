// Sparse convolution on the tile rulebook: forward / data-gradient (one kernel) and weight gradient.
// fp32 in, fp32 accumulate on v_mfma_f32_16x16x4_f32 (exact fp32, bitwise an fmaf chain).
//
// Forward work decomposition: an ITEM is (tile of 64 output rows, strip of 32 output channels)
// and belongs to ONE wave; the four waves of a workgroup take four consecutive items so that their gathers
// share L1.  A wave walks the tile's active kernel offsets; for every offset the valid (in,out) pairs are
// already compacted (rulebook), so the MFMA row groups are dense: up to 4 groups of 16 pairs.  A comes
// straight from global memory (each lane owns one gathered row, 4 consecutive channels), B (weights)
// straight from L2 into registers, the per-offset result accumulates in registers over all input-channel
// chunks and is then added into the wave-private 64x32 output strip in LDS with plain read-modify-write
// (LDS float atomics measured ~200 cycles per wave instruction: 5 TFLOP/s; never use them here).
// No workgroup barrier anywhere; latency is covered by occupancy (8 KiB LDS, <=128 VGPRs: 4 waves/SIMD)
// and by issuing two chunks of loads before the first MFMA block.
#include "b2m_common.h"
#include <stdlib.h>

struct ConvArgs {
    const float* x1; int64_t ldx1; int c1;
    const float* x2; int64_t ldx2; int c2;
    const float* w; int64_t ldw; int K;
    const float* bias;
    const int32_t* rb_in; const uint8_t* rb_out; const int32_t* rb_cnt;
    int64_t n_out, ntiles;
    float* y; int64_t ldy; int cout; int accumulate; int nstrips; int vec_store; int a_scalar;
};

// loads of padded / out-of-range operands are redirected here (pointer select, no select on the loaded value)
__device__ float g_zeros[64];

static_assert(B2M_TILE == 64, "conv kernels assume 64-row tiles (4 row groups of 16)");
#define NG 4     // row groups per tile

__device__ __forceinline__ int cs_index(int row, int col) { return row * 32 + (col ^ ((row & 1) << 4)); }

template <int KC, bool IDENT, bool ASCALAR, bool PAIR>
__global__ __launch_bounds__(256, PAIR ? 3 : 4) void conv_fwd_kernel(ConvArgs a) {
    constexpr int KS = KC / 4;                // k-steps per chunk == floats per lane per gathered row
    __shared__ float smem[4 * B2M_TILE * 32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int64_t item = (int64_t)blockIdx.x * 4 + wave;
    const int64_t tile = item / a.nstrips;
    const int strip = (int)(item % a.nstrips);
    if (tile >= a.ntiles) return;             // whole wave leaves; no barriers below
    const int col0 = strip * 32;
    const int cin = a.c1 + a.c2;
    const int nchunk = (cin + KC - 1) / KC;
    const int64_t ldr = a.ntiles * B2M_TILE;
    const int64_t row0 = tile * B2M_TILE;
    float* Cs = smem + wave * (B2M_TILE * 32);

    // ---- init the strip: 0 | Y (accumulate) | + bias
    for (int e = lane; e < B2M_TILE * 8; e += 64) {
        const int row = e >> 3, c4 = (e & 7) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const int64_t grow = row0 + row;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int col = col0 + c4 + u;
            if (col < a.cout) {
                float t = a.bias ? a.bias[col] : 0.f;
                if (a.accumulate && grow < a.n_out) t += a.y[grow * a.ldy + col];
                v[u] = t;
            }
        }
        *(f32x4*)&Cs[cs_index(row, c4)] = v;
    }

    // ---- active offsets of this tile (K <= 128): lane k holds the pair count of offset k / k+64
    int cnt0 = 0, cnt1 = 0;
    if (IDENT) {
        int64_t rem = a.n_out - row0;
        if (lane == 0) cnt0 = rem < B2M_TILE ? (int)rem : B2M_TILE;
    } else {
        if (lane < a.K) cnt0 = a.rb_cnt[(int64_t)lane * a.ntiles + tile];
        if (lane + 64 < a.K) cnt1 = a.rb_cnt[(int64_t)(lane + 64) * a.ntiles + tile];
    }
    uint64_t m0 = __ballot(cnt0 > 0), m1 = __ballot(cnt1 > 0);

    // one chunk of operands: B = weights of (offset, chunk), A = gathered rows of the 4 row groups.
    // All loads are unconditional (pointer select to a zero buffer), so their number is static.
    auto load_chunk = [&](float (&av)[NG][KS], float (&bv)[KS][2], const int (&idx)[NG], int k, int c) {
        const int cb = c * KC;
        const bool cv = c < nchunk;            // wave-uniform: the second chunk of the last pair may not exist
        const float* src; int64_t ld; int cl, climit;
        if (cb < a.c1) { src = a.x1; ld = a.ldx1; cl = cb + KS * q; climit = a.c1; }
        else { src = a.x2; ld = a.ldx2; cl = cb - a.c1 + KS * q; climit = a.c2; }
        const bool vc = cv && cl < climit;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int ci = cb + KS * q + s;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int col = col0 + 16 * t + i;
                const float* p = a.w + ((int64_t)k * cin + ci) * a.ldw + col;
                bv[s][t] = *((cv && ci < cin && col < a.cout) ? p : g_zeros);
            }
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int r = idx[g];
            const float* p = src + (int64_t)r * ld + cl;
            if constexpr (ASCALAR) {           // odd channel counts (head gradients): per-element
#pragma unroll
                for (int s = 0; s < KS; ++s) av[g][s] = *((r >= 0 && cv && cl + s < climit) ? p + s : g_zeros);
            } else if constexpr (KS == 4) {
                const f32x4 v = *(const f32x4*)((r >= 0 && vc) ? p : g_zeros);
                av[g][0] = v[0]; av[g][1] = v[1]; av[g][2] = v[2]; av[g][3] = v[3];
            } else {
                const f32x2 v = *(const f32x2*)((r >= 0 && vc) ? p : g_zeros);
                av[g][0] = v[0]; av[g][1] = v[1];
            }
        }
    };

    for (;;) {
        int k;
        if (m0) { k = __builtin_ctzll(m0); m0 &= m0 - 1; }
        else if (m1) { k = 64 + __builtin_ctzll(m1); m1 &= m1 - 1; }
        else break;
        const int n = k < 64 ? __builtin_amdgcn_readlane(cnt0, k) : __builtin_amdgcn_readlane(cnt1, k - 64);
        const int G = (n + 15) >> 4;           // 1..4 dense row groups
        // pair lists: lane (i,q) gathers input row idx[g] and later flushes the 4 output rows packed in out[g]
        int idx[NG]; uint32_t out[NG];
        const int64_t base = (int64_t)k * ldr + row0;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (IDENT) {
                int64_t r = row0 + 16 * g + i;
                idx[g] = r < a.n_out ? (int)r : -1;
                const int p = 16 * g + 4 * q;
                out[g] = (uint32_t)p | ((uint32_t)(p + 1) << 8) | ((uint32_t)(p + 2) << 16) | ((uint32_t)(p + 3) << 24);
            } else {                           // slots beyond the pair count hold -1 / 0 by construction
                idx[g] = a.rb_in[base + 16 * g + i];
                out[g] = *(const uint32_t*)(a.rb_out + base + 16 * g + 4 * q);
            }
        }
        f32x4 acc[NG][2];
#pragma unroll
        for (int g = 0; g < NG; ++g) { acc[g][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[g][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }

        if constexpr (PAIR) {
            // two chunks of loads in flight before the first MFMA block (<=168 VGPRs, 3 waves per SIMD)
            for (int c = 0; c < nchunk; c += 2) {
                float av0[NG][KS], bv0[KS][2], av1[NG][KS], bv1[KS][2];
                load_chunk(av0, bv0, idx, k, c);
                load_chunk(av1, bv1, idx, k, c + 1);
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    if (g < G) {                        // wave-uniform
#pragma unroll
                        for (int s = 0; s < KS; ++s) {
                            acc[g][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[g][s], bv0[s][0], acc[g][0], 0, 0, 0);
                            acc[g][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[g][s], bv0[s][1], acc[g][1], 0, 0, 0);
                        }
                    }
                }
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    if (g < G) {
#pragma unroll
                        for (int s = 0; s < KS; ++s) {
                            acc[g][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[g][s], bv1[s][0], acc[g][0], 0, 0, 0);
                            acc[g][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[g][s], bv1[s][1], acc[g][1], 0, 0, 0);
                        }
                    }
                }
            }
        } else {
            // one chunk at a time (<=128 VGPRs, 4 waves per SIMD)
            for (int c = 0; c < nchunk; ++c) {
                float av0[NG][KS], bv0[KS][2];
                load_chunk(av0, bv0, idx, k, c);
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    if (g < G) {
#pragma unroll
                        for (int s = 0; s < KS; ++s) {
                            acc[g][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[g][s], bv0[s][0], acc[g][0], 0, 0, 0);
                            acc[g][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[g][s], bv0[s][1], acc[g][1], 0, 0, 0);
                        }
                    }
                }
            }
        }
        // ---- add the offset's result into the strip.  D[row = 4q + r][col = i]; the pairs of one offset have
        // distinct output rows, and padded pairs (A == 0 -> acc == 0) are skipped, so no two lanes of an
        // instruction touch the same address: plain read-modify-write is race free inside the wave.
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g < G) {
                const uint32_t o4 = out[g];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (16 * g + 4 * q + r < n) {
                        const int row = (o4 >> (8 * r)) & 255;
                        Cs[cs_index(row, i)] += acc[g][0][r];
                        Cs[cs_index(row, 16 + i)] += acc[g][1][r];
                    }
                }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");

    // ---- write the strip
    for (int e = lane; e < B2M_TILE * 8; e += 64) {
        const int row = e >> 3, c4 = (e & 7) * 4;
        const int64_t grow = row0 + row;
        if (grow >= a.n_out) continue;
        const f32x4 v = *(const f32x4*)&Cs[cs_index(row, c4)];
        const int col = col0 + c4;
        float* dst = a.y + grow * a.ldy + col;
        if (a.vec_store && col + 3 < a.cout) {
            *(f32x4*)dst = v;
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) if (col + u < a.cout) dst[u] = v[u];
        }
    }
}

extern "C" int b2m_conv_fwd(const float* x1, int64_t ldx1, int32_t c1, const float* x2, int64_t ldx2, int32_t c2,
                            const float* w, int64_t ldw, int32_t K, const float* bias, const int32_t* rb_in,
                            const uint8_t* rb_out, const int32_t* rb_cnt, int64_t n_out, float* y, int64_t ldy,
                            int32_t cout, int32_t accumulate, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    B2M_CHECK_ARG(x1 && w && y && c1 > 0 && c2 >= 0 && cout > 0 && K >= 1 && K <= 128, "bad pointers/sizes (K<=128)");
    B2M_CHECK_ARG((rb_in == nullptr) == (rb_out == nullptr) && (rb_in == nullptr) == (rb_cnt == nullptr),
                  "rulebook pointers must be all set or all NULL");
    B2M_CHECK_ARG(rb_in != nullptr || K == 1, "identity rulebook needs K == 1");
    B2M_CHECK_ARG(c2 == 0 || x2 != nullptr, "x2 is NULL");
    B2M_CHECK_ARG(ldw >= cout && ldy >= cout && ldx1 >= c1 && (c2 == 0 || ldx2 >= c2), "leading dimension too small");
    const int cin = c1 + c2;
    const int KC = cin >= 16 ? 16 : 8;
    const int KS = KC / 4;
    B2M_CHECK_ARG(c2 == 0 || c1 % KC == 0, "with two sources c1 must be a multiple of 16");
    const bool aligned = c1 % KS == 0 && c2 % KS == 0 && ldx1 % KS == 0 && (c2 == 0 || ldx2 % KS == 0) &&
                         ((uintptr_t)x1 % (4 * KS)) == 0 && ((uintptr_t)x2 % (4 * KS)) == 0;
    if (n_out == 0) return B2M_OK;
    ConvArgs a;
    a.x1 = x1; a.ldx1 = ldx1; a.c1 = c1; a.x2 = x2; a.ldx2 = ldx2; a.c2 = c2;
    a.w = w; a.ldw = ldw; a.K = K; a.bias = bias;
    a.rb_in = rb_in; a.rb_out = rb_out; a.rb_cnt = rb_cnt;
    a.n_out = n_out; a.ntiles = cdiv64(n_out, B2M_TILE);
    a.y = y; a.ldy = ldy; a.cout = cout; a.accumulate = accumulate;
    a.nstrips = (cout + 31) / 32;
    a.vec_store = (ldy % 4 == 0 && ((uintptr_t)y % 16) == 0) ? 1 : 0;
    a.a_scalar = aligned ? 0 : 1;
    const int64_t items = a.ntiles * a.nstrips;
    const unsigned grid = (unsigned)cdiv64(items, 4);
    const bool ident = rb_in == nullptr;
    const int variant = (KC == 16 ? 4 : 0) | (ident ? 2 : 0) | (a.a_scalar ? 1 : 0);
    static int pair = -1;                       // B2M_CONV_PAIR=0/1 selects the chunk pipelining variant
    if (pair < 0) { const char* e = getenv("B2M_CONV_PAIR"); pair = e ? (atoi(e) != 0) : 0; }
#define B2M_CONV_CASE(V, KCV, ID, AS)                                             \
    case V:                                                                       \
        if (pair) conv_fwd_kernel<KCV, ID, AS, true><<<grid, 256, 0, st>>>(a);    \
        else conv_fwd_kernel<KCV, ID, AS, false><<<grid, 256, 0, st>>>(a);        \
        break;
    switch (variant) {
        B2M_CONV_CASE(0, 8, false, false) B2M_CONV_CASE(1, 8, false, true) B2M_CONV_CASE(2, 8, true, false)
        B2M_CONV_CASE(3, 8, true, true) B2M_CONV_CASE(4, 16, false, false) B2M_CONV_CASE(5, 16, false, true)
        B2M_CONV_CASE(6, 16, true, false) B2M_CONV_CASE(7, 16, true, true)
    }
#undef B2M_CONV_CASE
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}

// ------------------------------------------------------------------ weight transpose (+ mirror)
__global__ void weight_transpose_kernel(const float* __restrict__ w, int64_t ldw, int K, int cin, int cout,
                                        float* __restrict__ wt, int64_t ldwt, int mirror) {
    // 32x32 tiles through LDS so both the read (along co) and the write (along ci) are coalesced
    __shared__ float t[32][33];
    const int kk = blockIdx.z, src = mirror ? K - 1 - kk : kk;
    const int ci0 = blockIdx.y * 32, co0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 256 threads: 8 rows per pass
    for (int r = ty; r < 32; r += 8) {
        int ci = ci0 + r, co = co0 + tx;
        t[r][tx] = (ci < cin && co < cout) ? w[((int64_t)src * cin + ci) * ldw + co] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        int co = co0 + r, ci = ci0 + tx;
        if (co < cout && ci < cin) wt[((int64_t)kk * cout + co) * ldwt + ci] = t[tx][r];
    }
}
extern "C" int b2m_weight_transpose(const float* w, int64_t ldw, int32_t K, int32_t cin, int32_t cout, float* wt,
                                    int64_t ldwt, int32_t mirror, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    B2M_CHECK_ARG(w && wt && K >= 1 && cin > 0 && cout > 0 && ldw >= cout && ldwt >= cin, "bad arguments");
    dim3 grid((cout + 31) / 32, (cin + 31) / 32, K);
    weight_transpose_kernel<<<grid, 256, 0, st>>>(w, ldw, K, cin, cout, wt, ldwt, mirror);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}

// ------------------------------------------------------------------ weight gradient
// One wave owns a (16*MI x 16*NJ) block of dW[k] and reduces over the pairs of a chunk of tiles;
// the pair dimension is the MFMA k dimension, so compaction costs nothing.  No LDS, no barrier:
// latency is hidden by occupancy (about 100 VGPRs -> 4-5 waves per SIMD).
struct WgradArgs {
    const float* x; int64_t ldx; int cin;
    const float* dy; int64_t lddy; int cout;
    const int32_t* rb_in; const uint8_t* rb_out; const int32_t* rb_cnt;
    int64_t n_out, ntiles; int K;
    float* dw; int64_t lddw, dw_kstride;
    int tiles_per_chunk, nmb, nnb;
};

template <int MI, int NJ>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int k = blockIdx.x;
    const int blk = blockIdx.z * 4 + wave;
    if (blk >= a.nmb * a.nnb) return;
    const int ci0 = (blk / a.nnb) * 16 * MI, co0 = (blk % a.nnb) * 16 * NJ;
    const bool identity = a.rb_in == nullptr;
    const int64_t ldr = a.ntiles * B2M_TILE;
    f32x4 acc[MI][NJ];
#pragma unroll
    for (int m = 0; m < MI; ++m)
#pragma unroll
        for (int n = 0; n < NJ; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int64_t t0 = (int64_t)blockIdx.y * a.tiles_per_chunk;
    int64_t t1 = t0 + a.tiles_per_chunk;
    if (t1 > a.ntiles) t1 = a.ntiles;
    for (int64_t tile = t0; tile < t1; ++tile) {
        const int64_t row0 = tile * B2M_TILE;
        int n;
        if (identity) { int64_t rem = a.n_out - row0; n = rem < B2M_TILE ? (int)rem : B2M_TILE; }
        else n = __builtin_amdgcn_readfirstlane(a.rb_cnt[(int64_t)k * a.ntiles + tile]);
        const int G = (n + 15) >> 4;
        for (int g = 0; g < G; ++g) {
            int rin[4]; uint32_t o4;
            if (identity) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    int64_t r = row0 + 16 * g + 4 * q + s;
                    rin[s] = r < a.n_out ? (int)r : -1;
                }
                const int p = 16 * g + 4 * q;
                o4 = (uint32_t)p | ((uint32_t)(p + 1) << 8) | ((uint32_t)(p + 2) << 16) | ((uint32_t)(p + 3) << 24);
            } else {
                const int64_t base = (int64_t)k * ldr + row0 + 16 * g + 4 * q;
                i32x4 v = *(const i32x4*)(a.rb_in + base);
                rin[0] = v[0]; rin[1] = v[1]; rin[2] = v[2]; rin[3] = v[3];
                o4 = *(const uint32_t*)(a.rb_out + base);
            }
            float av[4][MI], bv[4][NJ];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int r = rin[s];
                const int64_t ro = row0 + ((o4 >> (8 * s)) & 255);
#pragma unroll
                for (int m = 0; m < MI; ++m) {
                    const int ci = ci0 + 16 * m + i;
                    av[s][m] = (r >= 0 && ci < a.cin) ? a.x[(int64_t)r * a.ldx + ci] : 0.f;
                }
#pragma unroll
                for (int nn = 0; nn < NJ; ++nn) {
                    const int co = co0 + 16 * nn + i;
                    bv[s][nn] = (r >= 0 && co < a.cout) ? a.dy[ro * a.lddy + co] : 0.f;
                }
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int m = 0; m < MI; ++m)
#pragma unroll
                    for (int nn = 0; nn < NJ; ++nn)
                        acc[m][nn] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s][m], bv[s][nn], acc[m][nn], 0, 0, 0);
        }
    }
    // D[row = 4q + r (ci), col = i (co)]
#pragma unroll
    for (int m = 0; m < MI; ++m)
#pragma unroll
        for (int nn = 0; nn < NJ; ++nn)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = ci0 + 16 * m + 4 * q + r, co = co0 + 16 * nn + i;
                if (ci < a.cin && co < a.cout) {
                    const float v = acc[m][nn][r];
                    if (v != 0.f) atomicAdd(&a.dw[(int64_t)k * a.dw_kstride + (int64_t)ci * a.lddw + co], v);
                }
            }
}

template <int MI>
static void launch_wgrad_nj(int NJ, dim3 grid, hipStream_t st, const WgradArgs& a) {
    switch (NJ) {
        case 1: conv_wgrad_kernel<MI, 1><<<grid, 256, 0, st>>>(a); break;
        case 2: conv_wgrad_kernel<MI, 2><<<grid, 256, 0, st>>>(a); break;
        case 3: conv_wgrad_kernel<MI, 3><<<grid, 256, 0, st>>>(a); break;
        default: conv_wgrad_kernel<MI, 4><<<grid, 256, 0, st>>>(a); break;
    }
}
static int pick_blk(int c) {      // 16-column sub-tiles per wave block
    if (c % 64 == 0) return 4;
    if (c % 48 == 0) return 3;
    if (c % 32 == 0) return 2;
    if (c <= 16) return 1;
    if (c <= 32) return 2;
    if (c <= 48) return 3;
    return 4;
}

extern "C" int b2m_conv_wgrad(const float* x, int64_t ldx, int32_t cin, const float* dy, int64_t lddy, int32_t cout,
                              const int32_t* rb_in, const uint8_t* rb_out, const int32_t* rb_cnt, int64_t n_out,
                              int32_t K, float* dw, int64_t lddw, int64_t dw_kstride, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    B2M_CHECK_ARG(x && dy && dw && cin > 0 && cout > 0 && K >= 1 && K <= 65535, "bad pointers/sizes");
    B2M_CHECK_ARG((rb_in == nullptr) == (rb_out == nullptr) && (rb_in == nullptr) == (rb_cnt == nullptr),
                  "rulebook pointers must be all set or all NULL");
    B2M_CHECK_ARG(rb_in != nullptr || K == 1, "identity rulebook needs K == 1");
    B2M_CHECK_ARG(ldx >= cin && lddy >= cout && lddw >= cout && dw_kstride >= (int64_t)cin * lddw,
                  "leading dimension too small");
    if (n_out == 0) return B2M_OK;
    WgradArgs a;
    a.x = x; a.ldx = ldx; a.cin = cin; a.dy = dy; a.lddy = lddy; a.cout = cout;
    a.rb_in = rb_in; a.rb_out = rb_out; a.rb_cnt = rb_cnt;
    a.n_out = n_out; a.ntiles = cdiv64(n_out, B2M_TILE); a.K = K; a.dw = dw; a.lddw = lddw; a.dw_kstride = dw_kstride;
    const int MI = pick_blk(cin), NJ = pick_blk(cout);
    a.nmb = (cin + 16 * MI - 1) / (16 * MI);
    a.nnb = (cout + 16 * NJ - 1) / (16 * NJ);
    // chunking: enough waves to fill the chip (>= ~8k), at most 65535 chunks, at least 1 tile per chunk
    const int64_t blocks_per_chunk = (int64_t)K * a.nmb * a.nnb;
    int64_t want_chunks = cdiv64(16384, blocks_per_chunk);
    if (want_chunks < 1) want_chunks = 1;
    int64_t tpc = cdiv64(a.ntiles, want_chunks);
    if (tpc < 1) tpc = 1;
    if (tpc > 64) tpc = 64;
    if (cdiv64(a.ntiles, tpc) > 65535) tpc = cdiv64(a.ntiles, 65535);
    a.tiles_per_chunk = (int)tpc;
    dim3 grid((unsigned)K, (unsigned)cdiv64(a.ntiles, tpc), (unsigned)((a.nmb * a.nnb + 3) / 4));
    switch (MI) {
        case 1: launch_wgrad_nj<1>(NJ, grid, st, a); break;
        case 2: launch_wgrad_nj<2>(NJ, grid, st, a); break;
        case 3: launch_wgrad_nj<3>(NJ, grid, st, a); break;
        default: launch_wgrad_nj<4>(NJ, grid, st, a); break;
    }
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}
