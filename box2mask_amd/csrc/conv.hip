// Sparse convolution on the tile rulebook: forward / data-gradient (one kernel) and weight gradient.
// fp32 in, fp32 accumulate on v_mfma_f32_16x16x4_f32 (exact fp32, bitwise an fmaf chain).
//
// Forward work decomposition: an ITEM is (tile of 128 output rows, strip of 32 output
// channels) and belongs to ONE wave; the four waves of a workgroup take four consecutive items so that
// their gathers share L1.  A wave walks the tile's active kernel offsets; for every offset the valid
// (in,out) pairs are already compacted (rulebook), so MFMA row groups are dense: 16 pairs x 16 input
// channels per step.  A comes straight from global memory (each lane owns one gathered row),
// B (weights) straight from L2 into registers, and the 128x32 output strip accumulates in a wave-private
// LDS region with ds_add_f32 -- no workgroup barrier anywhere.
#include "b2m_common.h"

struct ConvArgs {
    const float* x1; int64_t ldx1; int c1;
    const float* x2; int64_t ldx2; int c2;
    const float* w; int64_t ldw; int K;
    const float* bias;
    const int32_t* rb_in; const uint8_t* rb_out; const int32_t* rb_cnt;
    int64_t n_out, ntiles;
    float* y; int64_t ldy; int cout; int accumulate; int nstrips; int vec_store; int a_scalar;
};

__device__ __forceinline__ int cs_index(int row, int col) { return row * 32 + (col ^ ((row & 1) << 4)); }

template <int KC>
__global__ __launch_bounds__(256) void conv_fwd_kernel(ConvArgs a) {
    constexpr int KS = KC / 4;                // k-steps per chunk == floats per lane per gathered row
    extern __shared__ float smem[];
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
    const bool identity = a.rb_in == nullptr;
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
    if (identity) {
        int64_t rem = a.n_out - row0;
        if (lane == 0) cnt0 = rem < B2M_TILE ? (int)rem : B2M_TILE;
    } else {
        if (lane < a.K) cnt0 = a.rb_cnt[(int64_t)lane * a.ntiles + tile];
        if (lane + 64 < a.K) cnt1 = a.rb_cnt[(int64_t)(lane + 64) * a.ntiles + tile];
    }
    uint64_t m0 = __ballot(cnt0 > 0), m1 = __ballot(cnt1 > 0);
    auto next_active = [&]() -> int {
        int k;
        if (m0) { k = __builtin_ctzll(m0); m0 &= m0 - 1; }
        else if (m1) { k = 64 + __builtin_ctzll(m1); m1 &= m1 - 1; }
        else k = -1;
        return k;
    };
    auto get_cnt = [&](int k) -> int {
        return k < 64 ? __builtin_amdgcn_readlane(cnt0, k) : __builtin_amdgcn_readlane(cnt1, k - 64);
    };
    auto load_idx = [&](int k, int n, int (&idx)[8], uint32_t (&out)[8]) {
        const int G = (n + 15) >> 4;
        const int64_t base = (int64_t)k * ldr + row0;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            idx[g] = -1; out[g] = 0;
            if (g < G) {
                if (identity) {
                    int64_t r = row0 + 16 * g + i;
                    idx[g] = r < a.n_out ? (int)r : -1;
                    const int p = 16 * g + 4 * q;
                    out[g] = (uint32_t)p | ((uint32_t)(p + 1) << 8) | ((uint32_t)(p + 2) << 16) | ((uint32_t)(p + 3) << 24);
                } else {
                    idx[g] = a.rb_in[base + 16 * g + i];
                    out[g] = *(const uint32_t*)(a.rb_out + base + 16 * g + 4 * q);
                }
            }
        }
    };
    auto load_a = [&](float (&av)[8][KS], const int (&idx)[8], int n, int c) {
        const int G = (n + 15) >> 4;
        const int cb = c * KC;
        const float* src; int64_t ld; int cl, climit;
        if (cb < a.c1) { src = a.x1; ld = a.ldx1; cl = cb + KS * q; climit = a.c1; }
        else { src = a.x2; ld = a.ldx2; cl = cb - a.c1 + KS * q; climit = a.c2; }
        const bool vc = cl < climit;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
#pragma unroll
            for (int s = 0; s < KS; ++s) av[g][s] = 0.f;
            if (g < G) {
                const int r = idx[g];
                if (r >= 0 && a.a_scalar) {                      // odd channel counts (head gradients): per-element
                    const float* p = src + (int64_t)r * ld + cl;
#pragma unroll
                    for (int s = 0; s < KS; ++s) if (cl + s < climit) av[g][s] = p[s];
                } else if (r >= 0 && vc) {
                    const float* p = src + (int64_t)r * ld + cl;
                    if constexpr (KS == 4) {
                        f32x4 v = *(const f32x4*)p;
                        av[g][0] = v[0]; av[g][1] = v[1]; av[g][2] = v[2]; av[g][3] = v[3];
                    } else {
                        f32x2 v = *(const f32x2*)p;
                        av[g][0] = v[0]; av[g][1] = v[1];
                    }
                }
            }
        }
    };
    auto load_b = [&](float (&bv)[KS][2], int k, int c) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int ci = c * KC + KS * q + s;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int col = col0 + 16 * t + i;
                bv[s][t] = (ci < cin && col < a.cout) ? a.w[((int64_t)k * cin + ci) * a.ldw + col] : 0.f;
            }
        }
    };
    auto compute = [&](const float (&av)[8][KS], const float (&bv)[KS][2], const uint32_t (&out)[8], int n) {
        const int G = (n + 15) >> 4;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (g < G) {
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g][s], bv[s][0], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g][s], bv[s][1], acc1, 0, 0, 0);
                }
                const uint32_t o4 = out[g];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (16 * g + 4 * q + r < n) {
                        const int row = (o4 >> (8 * r)) & 255;
                        atomicAdd(&Cs[cs_index(row, i)], acc0[r]);
                        atomicAdd(&Cs[cs_index(row, 16 + i)], acc1[r]);
                    }
                }
            }
        }
    };

    int k = next_active();
    if (k >= 0) {
        int n = get_cnt(k);
        int idxc[8], idxn[8];
        uint32_t outc[8], outn[8];
        load_idx(k, n, idxc, outc);
        int k2 = next_active();
        int n2 = k2 >= 0 ? get_cnt(k2) : 0;
#pragma unroll
        for (int g = 0; g < 8; ++g) { idxn[g] = -1; outn[g] = 0; }
        if (k2 >= 0) load_idx(k2, n2, idxn, outn);
        float ac[8][KS], bc[KS][2];
        load_a(ac, idxc, n, 0);
        load_b(bc, k, 0);
        int c = 0;
        for (;;) {
            float an[8][KS], bn[KS][2];
            const bool same = c + 1 < nchunk;
            const int nk = same ? k : k2;
            if (nk >= 0) {
                if (same) { load_a(an, idxc, n, c + 1); load_b(bn, k, c + 1); }
                else { load_a(an, idxn, n2, 0); load_b(bn, k2, 0); }
            }
            compute(ac, bc, outc, n);
            if (nk < 0) break;
            if (!same) {
                k = k2; n = n2;
#pragma unroll
                for (int g = 0; g < 8; ++g) { idxc[g] = idxn[g]; outc[g] = outn[g]; }
                k2 = next_active();
                n2 = k2 >= 0 ? get_cnt(k2) : 0;
                if (k2 >= 0) load_idx(k2, n2, idxn, outn);
                c = 0;
            } else {
                ++c;
            }
#pragma unroll
            for (int g = 0; g < 8; ++g)
#pragma unroll
                for (int s = 0; s < KS; ++s) ac[g][s] = an[g][s];
#pragma unroll
            for (int s = 0; s < KS; ++s) { bc[s][0] = bn[s][0]; bc[s][1] = bn[s][1]; }
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
    const size_t lds = 4 * B2M_TILE * 32 * sizeof(float);     // 64 KiB: two workgroups per CU
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv_fwd_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)conv_fwd_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    if (KC == 16) conv_fwd_kernel<16><<<grid, 256, lds, st>>>(a);
    else conv_fwd_kernel<8><<<grid, 256, lds, st>>>(a);
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
