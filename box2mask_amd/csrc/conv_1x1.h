// 1x1 convolution (identity map, K = 1): Y = [X1 | X2] W (+ bias) (+ Y) as a streaming GEMM.  Included by conv.hip.
//
// The general kernels treat a 1x1 layer as a one-offset sparse convolution: every (tile, 32-column strip) wave reads the
// tile's input rows again, adds its result into an LDS strip and copies the strip out -- 0.43 ms for the 1.2 M x 128 -> 96
// layer of the benchmark (2.5 TB/s of algorithmic bytes).  Without a pair list none of that is needed: the 16 pairs of a
// row group ARE rows 16g .. 16g+15 of the tile, so the accumulators can stay in registers from the first input channel
// to the last and go straight to Y.  One wave per (tile of 64 rows, column group of SPW strips): the input rows are read
// once per column group (cout <= 96: once), with the operands of the next 16-channel chunk in flight under the MFMAs of
// the current one; the operands are swapped (weights as A, rows as B) so that lane (q, i) ends up with 4 consecutive
// columns of row i and stores 16 bytes.
//
// Same packed weight image as the other kernels (K = 1: strips of 32 columns, blocks [strip][chunk] of 64 lanes x 8
// floats stored [u][lane][4]) and the same arguments.
template <int SPW>
__global__ __launch_bounds__(64, SPW == 3 ? 2 : 3) void conv_1x1_kernel(ConvArgs a) {
    constexpr int TW = 2, KS = 4;
    constexpr int NB = TW * SPW;              // 16-column blocks per wave
    constexpr int LW = 64 * TW * KS;          // floats per packed weight block
    const int lane = threadIdx.x & 63;
    const int i = lane & 15, q = lane >> 4;
    const int ncg = a.nstrips / SPW;          // column groups per tile
    const int64_t tile = blockIdx.x / ncg;
    const int cgi = (int)(blockIdx.x % ncg);
    const int strip0 = cgi * SPW;
    const int col0 = strip0 * 16 * TW;
    const int nch1 = a.c1 >> 4, NC = (a.c1 + a.c2) >> 4;
    const int64_t row0 = tile * B2M_TILE;

    // byte offsets of this lane's four rows (row group g, row i); rows past the end read row n_out - 1 (never stored)
    uint32_t r1[NG], r2[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        int64_t r = row0 + 16 * g + i;
        if (r >= a.n_out) r = a.n_out - 1;
        r1[g] = (uint32_t)r * ((uint32_t)a.ldx1 * 4u) + (uint32_t)q * 16u;
        r2[g] = (uint32_t)r * ((uint32_t)a.ldx2 * 4u) + (uint32_t)q * 16u;
    }
    const uint32_t wlo = (uint32_t)lane * 16u;

    f32x4 av[2][NG];
    f32x4 bw[2][SPW][TW];
    auto load_chunk = [&](int j, int c) {
        const bool first = c < nch1;                                        // wave-uniform source select
        const char* src = (const char*)(first ? a.x1 + (c << 4) : a.x2 + ((c - nch1) << 4));
#pragma unroll
        for (int g = 0; g < NG; ++g) av[j][g] = *(const f32x4*)(src + (first ? r1[g] : r2[g]));
#pragma unroll
        for (int sp = 0; sp < SPW; ++sp) {
            const char* wsrc = (const char*)a.wp + (size_t)((strip0 + sp) * NC + c) * (size_t)(LW * 4);
#pragma unroll
            for (int u = 0; u < TW; ++u) bw[j][sp][u] = *(const f32x4*)(wsrc + (wlo + 1024u * u));
        }
    };

    f32x4 acc[NG][NB];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[g][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto mfma_chunk = [&](int j) {
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int sp = 0; sp < SPW; ++sp)
#pragma unroll
                    for (int t = 0; t < TW; ++t) {
                        // float f = TW*s + t of the block: piece u = f / 4, element f % 4
                        const float w = bw[j][sp][(TW * s + t) >> 2][(TW * s + t) & 3];
                        acc[g][sp * TW + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, av[j][g][s], acc[g][sp * TW + t], 0, 0, 0);
                    }
    };
    // two chunks per round so that the two register buffers keep their names (no copies between rounds)
    load_chunk(0, 0);
    int c = 0;
    for (; c + 1 < NC; c += 2) {
        load_chunk(1, c + 1);
        mfma_chunk(0);
        load_chunk(0, c + 2 < NC ? c + 2 : c);           // (past the end: a harmless reload, keeps the loop uniform)
        mfma_chunk(1);
    }
    if (c < NC) mfma_chunk(0);

    // ---- Y: lane (q, i) holds columns col0 + 16 b + 4 q .. + 3 of row 16 g + i
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int64_t grow = row0 + 16 * g + i;
        if (grow >= a.n_out) continue;
        // accumulate: the row's old values of ALL column blocks are requested before the first store (round 5: a load behind a
        // store to the same array cannot be moved in front of it by the compiler -- NB dependent round trips per row group)
        f32x4 old[NB];
        if (a.accumulate && a.vec_store) {
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int col = col0 + 16 * b + 4 * q;
                old[b] = col + 3 < a.cout ? *(const f32x4*)(a.y + grow * a.ldy + col) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int col = col0 + 16 * b + 4 * q;
            if (col >= a.cout) continue;
            f32x4 v = acc[g][b];
            float* dst = a.y + grow * a.ldy + col;
            if (a.vec_store && col + 3 < a.cout) {
                if (a.bias) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] += a.bias[col + u];
                }
                if (a.accumulate) v += old[b];
                if (a.ep_scale) v = conv_epilogue(a, v, grow, col);
                *(f32x4*)dst = v;
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (col + u < a.cout) {
                        float t = v[u];
                        if (a.bias) t += a.bias[col + u];
                        if (a.accumulate) t += dst[u];
                        dst[u] = t;
                    }
            }
        }
    }
}
