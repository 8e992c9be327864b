// conv_flow2_kernel: the un-split forward / data-gradient kernel of round 4.  Same software pipeline as
// conv_fwd_flow_kernel (conv_fwd_flow.h: D steps (offset, chunk) in flight across offset boundaries, every MFMA block an
// asm statement with tied accumulators, a static number of loads per step), with two changes to what a wave owns:
//
//  * NT = 2: an item is TWO tiles of 64 output rows (consecutive positions of the dispatch order) x one strip of 16*TW
//    output channels.  The pair lists of the two tiles for one kernel offset are concatenated before they are cut into
//    MFMA row groups of 16: ceil((n0 + n1) / 16) groups instead of ceil(n0 / 16) + ceil(n1 / 16).  On the benchmark's maps
//    the useful share of the executed MFMAs rises from 0.86 / 0.81 / 0.80 (levels 0 / 1 / 2) to 0.92 / 0.89 / 0.89
//    (tools/fill_stats.py), every weight piece a wave loads feeds up to 8 row groups instead of 4, and the per-offset
//    work (list fetch, flush set-up) is paid once for two tiles.  The rulebook layout (64-row tiles, uint8 rows) and with
//    it the weight-gradient kernel are unchanged; the wave concatenates the lists with one cross-lane permute per tile and
//    row group.  Cost: the two 64-row half-strips are 2 x 13 KiB (TW = 3) or 2 x 9 KiB (TW = 2) of LDS and 8 row groups
//    of accumulators: 6 / 8 waves per CU.
//  * PERS = 1: a persistent grid.  Every wave slot of the chip holds one wave for the whole launch; a wave draws its
//    next items from the ticket counter of its XCD's run of the dispatch order (heavy tiles first at the end of a run,
//    b2m_rulebook_balance) with one atomic per PERS items (single items near the end of the run) and, when its own run is
//    exhausted, from the other XCDs' runs.  No workgroup launch between items, and the
//    eight runs no longer have to carry exactly equal work.  The counters reset themselves: the last wave to leave
//    (a ninth counter) zeroes them for the next launch on the same stream (one counter block per stream, conv.hip).
#pragma once

template <int D, int TW, int NT, int PERS, int EXP = 0, int WPG = 1>
__global__ __launch_bounds__(64 * WPG, NT == 2 ? 2 : ((TW == 2 && D == 2) ? 4 : 3)) void conv_flow2_kernel(ConvArgs a) {
    constexpr int KS = 4;                     // k-steps per 16-channel chunk == floats per lane per gathered row
    constexpr int SW = 16 * TW;               // output channels per strip
    constexpr int LW = 64 * TW * KS;          // floats per packed weight block
    constexpr int PITCH = SW + 4;             // strip row pitch in floats
    constexpr int ROWS = B2M_TILE * NT;       // output rows per item
    constexpr int NGT = NG * NT;              // row groups per offset, at most
    constexpr uint32_t PADWORD = (uint32_t)ROWS << 24;     // pair slot without a pair: gathers row 0, flushes nowhere
    // WPG > 1 (plain grid only): the WPG waves of a workgroup are WPG consecutive items -- the strips of ONE tile -- so that
    // they run on one CU and their gathers of the same rows meet in its L1
    __shared__ float CsAll[WPG * ROWS * PITCH];
    const int wave = WPG == 1 ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* const Cs = CsAll + wave * (ROWS * PITCH);
    const int lane = threadIdx.x & 63;
    const int i = lane & 15, q = lane >> 4;
    const int nch1 = a.c1 >> 4, NC = (a.c1 + a.c2) >> 4;      // chunks of the first source / in all; NC % D == 0
    const int64_t ldr = a.ntiles * B2M_TILE;
    const int nt = (int)a.ntiles;
    const int nstrips = a.nstrips;
    unsigned* const tick = a.tickets;

    // run of dispatch-order positions XCD x works on: the balanced boundaries of the rulebook, else equal eighths
    auto run_bounds = [&](int x, int& s0, int& s1) {
        if (a.xcd_start) { s0 = a.xcd_start[x]; s1 = a.xcd_start[x + 1]; }
        else {
            const int per = (nt + 7) >> 3;
            s0 = x * per < nt ? x * per : nt;
            s1 = s0 + per < nt ? s0 + per : nt;
        }
    };
    // (wave-uniform) a batch of tickets of run r: [j, jend).  Same-address device-scope atomics complete at about one per
    // 300 ns (measured: one draw per item from 4096 waves doubled the run time of the 32 -> 32 layers), so a draw takes
    // `batch` consecutive items while the run is long and single items near its end.
    const int wpr = (int)(gridDim.x >> 3) + 1;          // waves per run
    int jend = 0;
    auto draw = [&](int r, int nit_r, int jlast) -> int {
        const unsigned b = (PERS > 1 && nit_r - jlast > 8 * wpr) ? (unsigned)PERS : 1u;
        unsigned t = 0;
        if (lane == 0) t = atomicAdd(tick + r, b);
        const int j0 = (int)__builtin_amdgcn_readfirstlane(t);
        jend = j0 + (int)b < nit_r ? j0 + (int)b : nit_r;
        return j0;
    };
    int run = blockIdx.x & 7, tries = 0, s0, s1;
    run_bounds(run, s0, s1);
    int nit = ((s1 - s0 + NT - 1) / NT) * nstrips;
    int j = PERS ? draw(run, nit, 0) : (int)(blockIdx.x >> 3) * WPG + wave;

    const uint32_t wlo = (uint32_t)lane * 16u;         // packed block layout [u][lane][4 floats]: pack_pos()
    const uint32_t q16 = (uint32_t)q * 16u;
    const uint32_t ld1 = (uint32_t)a.ldx1 * 4u, ld2 = (uint32_t)a.ldx2 * 4u;
    const uint32_t wkstride = (uint32_t)nstrips * (uint32_t)NC;

#ifdef B2M_STAMPS
    unsigned long long st_a, st_b, st_c, st_pro = 0, st_loop = 0, st_flush = 0, st_life = 0, st_noff = 0, st_items = 0, st_epi = 0,
                       st_grp = 0, st_adv = 0, st_draw = 0;
#endif
    for (;;) {
        if (j >= nit) {
            if (!PERS) return;
            if (++tries == 8) break;          // all eight runs are exhausted
            run = (run + 1) & 7;
            run_bounds(run, s0, s1);
            nit = ((s1 - s0 + NT - 1) / NT) * nstrips;
            j = draw(run, nit, 0);
            continue;
        }

#ifdef B2M_STAMPS
        unsigned long long st_begin;
        B2M_STAMP(st_begin);
#endif
        const int sup = j / nstrips, strip = j - sup * nstrips;
        int tile[NT];
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            const int pos = s0 + sup * NT + u;
            tile[u] = -1;
            if (pos < s1) tile[u] = a.tile_order ? a.tile_order[pos] : pos;
        }
        const int col0 = strip * SW;
        auto tile_of_row = [&](int row) -> int { return NT == 1 ? tile[0] : (row < B2M_TILE ? tile[0] : tile[NT - 1]); };

        // ---- init the strip: 0 | Y (accumulate) | + bias
        for (int e = lane; e < ROWS * (SW / 4); e += 64) {
            const int row = e / (SW / 4), c4 = (e % (SW / 4)) * 4;
            const int t = tile_of_row(row);
            const int64_t grow = (int64_t)t * B2M_TILE + (row & (B2M_TILE - 1));
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int col = col0 + c4 + u;
                if (col < a.cout) {
                    float tt = a.bias ? a.bias[col] : 0.f;
                    if (a.accumulate && t >= 0 && grow < a.n_out) tt += a.y[grow * a.ldy + col];
                    v[u] = tt;
                }
            }
            *(f32x4*)&Cs[row * PITCH + c4] = v;
        }

        // ---- active offsets (K <= 64): lane k holds the pair counts of offset k in the item's tiles
        int cnt[NT], ctot = 0;
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            cnt[u] = 0;
            if (lane < a.K && tile[u] >= 0) cnt[u] = a.rb_cnt[(int64_t)lane * a.ntiles + tile[u]];
            ctot += cnt[u];
        }
        const uint64_t m0 = __ballot(ctot > 0);
        auto next_active = [&](int k) -> int {    // first active offset after k, or -1 (scalar)
            const int kk = k + 1;
            if (kk < 64) {
                const uint64_t r = m0 >> kk;
                if (r) return kk + __builtin_ctzll(r);
            }
            return -1;
        };
        auto groups_of = [&](int k) -> int { return (__builtin_amdgcn_readlane(ctot, k) + 15) >> 4; };

        int kC = next_active(-1);
        if (kC >= 0) {
            // pair lists of an offset: slot `lane` of every tile of the item
            auto list_load = [&](int k, int (&ri)[NT], int (&ro)[NT]) {
#pragma unroll
                for (int u = 0; u < NT; ++u) {
                    const int tb = tile[u] < 0 ? 0 : tile[u];              // (an absent second tile: valid memory, never selected)
                    const int64_t base = (int64_t)k * ldr + (int64_t)tb * B2M_TILE + lane;
                    ri[u] = a.rb_in[base];
                    ro[u] = a.rb_out[base];
                }
            };
            // word of pair 16g + i of the concatenated list: input row | output row inside the item << 24
            auto list_words = [&](int k, const int (&ri)[NT], const int (&ro)[NT], uint32_t (&w)[NGT]) {
                if constexpr (NT == 1) {
                    const uint32_t word = ri[0] < 0 ? PADWORD : ((uint32_t)ri[0] | ((uint32_t)ro[0] << 24));
#pragma unroll
                    for (int g = 0; g < NGT; ++g) w[g] = (uint32_t)__builtin_amdgcn_ds_bpermute((16 * g + i) << 2, (int)word);
                } else {
                    const int n0 = __builtin_amdgcn_readlane(cnt[0], k), n1 = __builtin_amdgcn_readlane(cnt[1], k);
                    const uint32_t word0 = ((uint32_t)ri[0] & 0xFFFFFFu) | ((uint32_t)ro[0] << 24);
                    const uint32_t word1 = ((uint32_t)ri[1] & 0xFFFFFFu) | (((uint32_t)ro[1] + (uint32_t)B2M_TILE) << 24);
#pragma unroll
                    for (int g = 0; g < NGT; ++g) {
                        const int p = 16 * g + i;                           // position in the concatenated list
                        const uint32_t wb = (uint32_t)__builtin_amdgcn_ds_bpermute(((p - n0) & 63) << 2, (int)word1);
                        uint32_t wv = (uint32_t)(p - n0) < (uint32_t)n1 ? wb : PADWORD;
                        if (g < NG) {                                       // (p < 64: may still belong to the first tile)
                            const uint32_t wa = (uint32_t)__builtin_amdgcn_ds_bpermute(p << 2, (int)word0);
                            wv = p < n0 ? wa : wv;
                        }
                        w[g] = wv;
                    }
                }
            };

            const uint32_t wstrip = (uint32_t)strip * (uint32_t)NC;
            float av[D][NGT][KS], bv[D][KS][TW];
            auto src_of = [&](int c, uint32_t& ld4) -> const char* {
                const bool first = c < nch1;                                        // wave-uniform source select
                ld4 = first ? ld1 : ld2;
                return (const char*)(first ? a.x1 + (c << 4) : a.x2 + ((c - nch1) << 4));
            };
            auto gather = [&](int jb, int g, const char* src, uint32_t ld4, uint32_t word) {
                const uint32_t off = __umul24(word & 0xFFFFFFu, ld4) + q16;
                const f32x4 v = (EXP & 4) ? __builtin_nontemporal_load((const f32x4*)(src + off)) : *(const f32x4*)(src + off);
                av[jb][g][0] = v[0]; av[jb][g][1] = v[1]; av[jb][g][2] = v[2]; av[jb][g][3] = v[3];
            };
            auto weights = [&](int jb, int k, int c) {
                const uint32_t blk = (uint32_t)k * wkstride + wstrip + (uint32_t)c;     // wave-uniform
                const char* wsrc = (const char*)a.wp + (size_t)blk * (size_t)(LW * 4);
                float wv[TW * KS];
#pragma unroll
                for (int u = 0; u < TW; ++u) {
                    const f32x4 w4 = (EXP & 2) ? __builtin_nontemporal_load((const f32x4*)(wsrc + (wlo + 1024u * u)))
                                               : *(const f32x4*)(wsrc + (wlo + 1024u * u));
                    wv[4 * u] = w4[0]; wv[4 * u + 1] = w4[1]; wv[4 * u + 2] = w4[2]; wv[4 * u + 3] = w4[3];
                }
#pragma unroll
                for (int s = 0; s < KS; ++s)
#pragma unroll
                    for (int t = 0; t < TW; ++t) bv[jb][s][t] = wv[TW * s + t];
            };

            // ---- prologue: lists of the first three offsets, operands of the first D steps
            uint32_t wC[NGT], wN[NGT];
            int rawi[NT], rawo[NT];
            int kN = next_active(kC);
            int kNc = kN < 0 ? kC : kN;
            int kNN = kN < 0 ? -1 : next_active(kN);
            {
                int r0i[NT], r0o[NT], r1i[NT], r1o[NT];
                list_load(kC, r0i, r0o);
                list_load(kNc, r1i, r1o);
                list_words(kC, r0i, r0o, wC);
                list_words(kNc, r1i, r1o, wN);
            }
            int GC = groups_of(kC);
#pragma unroll
            for (int jb = 0; jb < D; ++jb) {
                uint32_t ld4;
                const char* src = src_of(jb, ld4);
#pragma unroll
                for (int g = 0; g < NGT; ++g) gather(jb, g, src, ld4, wC[g]);
                weights(jb, kC, jb);
            }
            int kL = kNN < 0 ? kNc : kNN;                 // offset whose raw list is in flight
            list_load(kL, rawi, rawo);

            f32x4 acc[NGT][TW];
#pragma unroll
            for (int g = 0; g < NGT; ++g)
#pragma unroll
                for (int t = 0; t < TW; ++t) acc[g][t] = f32x4{0.f, 0.f, 0.f, 0.f};

            // the 4*TW MFMAs of (row group g, chunk step in buffer jb) as ONE asm statement, accumulators tied (conv_fwd_flow.h)
            auto mfma_group = [&](int jb, int g) {
                if constexpr (TW == 3) {
                    asm volatile(
                        "v_mfma_f32_16x16x4_f32 %0, %7, %3, %0\n\tv_mfma_f32_16x16x4_f32 %1, %8, %3, %1\n\tv_mfma_f32_16x16x4_f32 %2, %9, %3, %2\n\t"
                        "v_mfma_f32_16x16x4_f32 %0, %10, %4, %0\n\tv_mfma_f32_16x16x4_f32 %1, %11, %4, %1\n\tv_mfma_f32_16x16x4_f32 %2, %12, %4, %2\n\t"
                        "v_mfma_f32_16x16x4_f32 %0, %13, %5, %0\n\tv_mfma_f32_16x16x4_f32 %1, %14, %5, %1\n\tv_mfma_f32_16x16x4_f32 %2, %15, %5, %2\n\t"
                        "v_mfma_f32_16x16x4_f32 %0, %16, %6, %0\n\tv_mfma_f32_16x16x4_f32 %1, %17, %6, %1\n\tv_mfma_f32_16x16x4_f32 %2, %18, %6, %2"
                        : "+v"(acc[g][0]), "+v"(acc[g][1]), "+v"(acc[g][2])
                        : "v"(av[jb][g][0]), "v"(av[jb][g][1]), "v"(av[jb][g][2]), "v"(av[jb][g][3]),
                          "v"(bv[jb][0][0]), "v"(bv[jb][0][1]), "v"(bv[jb][0][2]), "v"(bv[jb][1][0]), "v"(bv[jb][1][1]), "v"(bv[jb][1][2]),
                          "v"(bv[jb][2][0]), "v"(bv[jb][2][1]), "v"(bv[jb][2][2]), "v"(bv[jb][3][0]), "v"(bv[jb][3][1]), "v"(bv[jb][3][2])
                        : "memory");
                } else {
                    asm volatile(
                        "v_mfma_f32_16x16x4_f32 %0, %6, %2, %0\n\tv_mfma_f32_16x16x4_f32 %1, %7, %2, %1\n\t"
                        "v_mfma_f32_16x16x4_f32 %0, %8, %3, %0\n\tv_mfma_f32_16x16x4_f32 %1, %9, %3, %1\n\t"
                        "v_mfma_f32_16x16x4_f32 %0, %10, %4, %0\n\tv_mfma_f32_16x16x4_f32 %1, %11, %4, %1\n\t"
                        "v_mfma_f32_16x16x4_f32 %0, %12, %5, %0\n\tv_mfma_f32_16x16x4_f32 %1, %13, %5, %1"
                        : "+v"(acc[g][0]), "+v"(acc[g][1])
                        : "v"(av[jb][g][0]), "v"(av[jb][g][1]), "v"(av[jb][g][2]), "v"(av[jb][g][3]),
                          "v"(bv[jb][0][0]), "v"(bv[jb][0][1]), "v"(bv[jb][1][0]), "v"(bv[jb][1][1]),
                          "v"(bv[jb][2][0]), "v"(bv[jb][2][1]), "v"(bv[jb][3][0]), "v"(bv[jb][3][1])
                        : "memory");
                }
            };

#ifdef B2M_STAMPS
            B2M_STAMP(st_a);
            st_pro += st_a - st_begin;
#endif
            for (;;) {
#ifdef B2M_STAMPS
                B2M_STAMP(st_a);
                st_noff += 1; st_grp += (unsigned long long)(GC * NC);
#endif
                int c0 = 0;
                do {                          // (bottom-tested: NC >= D.  With a zero-trip path hipcc cannot count the loads behind the
                                              // pair-list fetch and drains the whole queue -- s_waitcnt vmcnt(0) -- at every offset)
                    // the D prefetches of this round target one offset: the current one, or -- in its last round -- the next
                    const bool wrap = c0 + D >= NC;
                    const int kT = wrap ? kNc : kC;
                    const int cT = wrap ? 0 : c0 + D;
                    uint32_t wT[NGT];
#pragma unroll
                    for (int g = 0; g < NGT; ++g) wT[g] = wrap ? wN[g] : wC[g];
#pragma unroll
                    for (int jb = 0; jb < D; ++jb) {
                        uint32_t ld4;
                        const char* src = src_of(cT + jb, ld4);
#pragma unroll
                        for (int g = 0; g < NGT; ++g) {
                            if (g < GC) {                                        // wave-uniform
                                if constexpr (EXP & 1) __builtin_amdgcn_s_setprio(2);
                                mfma_group(jb, g);
                                if constexpr (EXP & 1) __builtin_amdgcn_s_setprio(0);
                            }
                            gather(jb, g, src, ld4, wT[g]);
                        }
                        weights(jb, kT, cT + jb);
                    }
                    c0 += D;
                } while (c0 < NC);
#ifdef B2M_STAMPS
                B2M_STAMP(st_b);
                st_loop += st_b - st_a;
#endif
                // ---- add the offset's result into the strip: lane (i,q) holds channels 16t + 4q .. +3 of pair 16g + i
                asm volatile("s_nop 15" ::: "memory");        // MFMA result -> VALU read: >= 11 wait states (8-pass MFMA)
#pragma unroll
                for (int g = 0; g < NGT; ++g) {
                    if (g < GC) {
                        const uint32_t orow = wC[g] >> 24;
                        if (orow < (uint32_t)ROWS) {
                            float* rowp = Cs + orow * PITCH + 4 * q;
                            f32x4 old[TW];
#pragma unroll
                            for (int t = 0; t < TW; ++t) old[t] = *(const f32x4*)(rowp + 16 * t);
#pragma unroll
                            for (int t = 0; t < TW; ++t) *(f32x4*)(rowp + 16 * t) = old[t] + acc[g][t];
                        }
#pragma unroll
                        for (int t = 0; t < TW; ++t) acc[g][t] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
#ifdef B2M_STAMPS
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                B2M_STAMP(st_c);
                st_flush += st_c - st_b;
#endif
                if (kN < 0) break;
                // ---- advance: next offset becomes current; the list fetched an offset ago becomes next; fetch one more
                kC = kN;
#pragma unroll
                for (int g = 0; g < NGT; ++g) wC[g] = wN[g];
                kN = kNN; kNc = kN < 0 ? kC : kN;
                GC = groups_of(kC);
                list_words(kL, rawi, rawo, wN);
                kNN = kN < 0 ? -1 : next_active(kN);
                kL = kNN < 0 ? kNc : kNN;
                list_load(kL, rawi, rawo);
#ifdef B2M_STAMPS
                B2M_STAMP(st_a);
                st_adv += st_a - st_c;
#endif
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        }

#ifdef B2M_STAMPS
        B2M_STAMP(st_b);
#endif
        // ---- BatchNorm column sums of the finished half-strips, then write them out (16-byte aligned rows: coalesced)
        if (a.stats) {
#pragma unroll
            for (int u = 0; u < NT; ++u)
                if (tile[u] >= 0) strip_column_sums<SW>(a, Cs + u * B2M_TILE * PITCH, tile[u], (int64_t)tile[u] * B2M_TILE, col0, lane);
        }
        for (int e = lane; e < ROWS * (SW / 4); e += 64) {
            const int row = e / (SW / 4), c4 = (e % (SW / 4)) * 4;
            const int t = tile_of_row(row);
            const int64_t grow = (int64_t)t * B2M_TILE + (row & (B2M_TILE - 1));
            if (t < 0 || grow >= a.n_out) continue;
            const f32x4 v = *(const f32x4*)&Cs[row * PITCH + c4];
            const int col = col0 + c4;
            float* dst = a.y + grow * a.ldy + col;
            if (a.vec_store && col + 3 < a.cout) {
                *(f32x4*)dst = v;
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) if (col + u < a.cout) dst[u] = v[u];
            }
        }
#ifdef B2M_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        B2M_STAMP(st_c);
        st_epi += st_c - st_b; st_life += st_c - st_begin; st_items += 1;
        if (!PERS) break;
#else
        if (!PERS) return;
#endif
        if (++j >= jend) j = draw(run, nit, j);
#ifdef B2M_STAMPS
        B2M_STAMP(st_a);
        st_draw += st_a - st_c;
#endif
    }
#ifdef B2M_STAMPS
    if (lane == 0) {
        atomicAdd(&g_stamps[0], st_pro); atomicAdd(&g_stamps[1], st_loop); atomicAdd(&g_stamps[2], st_flush);
        atomicAdd(&g_stamps[3], st_life); atomicAdd(&g_stamps[4], st_noff); atomicAdd(&g_stamps[5], st_items);
        atomicAdd(&g_stamps[6], st_epi); atomicAdd(&g_stamps[7], st_grp); atomicAdd(&g_stamps[8], st_adv);
        atomicAdd(&g_stamps[9], st_draw);
    }
    if (!PERS) return;
#endif
    // ---- the last wave to leave resets the counters for the next launch on this stream
    if (lane == 0) {
        const unsigned done = atomicAdd(tick + 8, 1u);
        if (done == gridDim.x - 1) {
#pragma unroll
            for (int r = 0; r < 9; ++r) atomicExch(tick + r, 0u);
        }
    }
}
