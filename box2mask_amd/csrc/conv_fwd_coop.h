// conv_fwd_coop_kernel (round 5): the flow kernel's pipeline on a workgroup-cooperative 256-row tile.  Included by conv.hip.
//
// What it is for.  conv_fwd_flow_kernel cuts the pairs of a (64-row tile, offset) visit into MFMA row groups of 16: on the
// benchmark's maps 0.878 (level 0) ... 0.80 (levels 2-3) of the executed MFMAs are useful, a third of the visits have ONE row
// group (their 4 + TW loads per 12 * TW MFMAs sit on the CU's vector-memory return path), and every visit pays the list fetch,
// the flush and the advance.  Here a workgroup of four waves owns FOUR consecutive tiles (256 output rows) x one strip of
// 16 * TW output channels.  The pair lists of the four tiles are read as ONE list per offset (the rulebook is unchanged: lane
// L of a unit computes which tile's slot holds pair 64 r + L of the concatenation), cut into UNITS of <= 64 pairs = <= 4 row
// groups; the supertile's units are dealt to the four waves, largest first, snake order.  tools/fill_stats.py: useful share
// 0.958 / 0.942 / 0.940 / 0.938 on levels 0..3, 26 ... 38 % fewer visits, one-group units 30 -> 14 % (level 0).
//
// A unit is exactly a visit of the flow kernel (same D = 2 pipeline across unit boundaries, same hand-issued EXEC-masked
// loads, same counted waits, same MFMA blocks), but its 64 output rows lie anywhere in the 256-row strip the four waves
// SHARE (4 x 13 KB: the LDS per wave is unchanged, three workgroups = twelve waves per CU as before).  Two waves may add
// into the same row at the same time, so the flush is the flow kernel's 16-byte read-modify-write under a workgroup-local
// spin lock in LDS (one ds_cmpst by one lane; a wave flushes once per ~20 k cycles and holds the lock for ~0.5 k: measured
// contention is a few per cent of the flushes).  LDS float atomics are NOT an option: tools/micro/lds_flush.hip measures
// ds_add_f32 at ~190 cycles per wave instruction (9.3 k cycles per flush against 240 for the read-modify-write).
//
// fp32 only, un-split maps, K <= 32 offsets, input rows < 2^23 (pair word = input row | output row << 23, 9 bits: 0..255 and
// the padding mark 256).  Everything else (bias, accumulate, two sources, BatchNorm column sums per 64-row tile, inference
// epilogue) as in conv_fwd_flow_kernel.  Results differ from the flow kernel's in the summation order over offsets only.
#pragma once

// DBG (diagnostics, tools/bench_conv.py with B2M_COOP_DBG; wrong results possible): 1 = flush without the lock
template <int TW, int DBG = 0, int DEAL = 0>
__global__ __launch_bounds__(256, TW == 2 ? 4 : 3) void conv_fwd_coop_kernel(ConvArgs a) {
    constexpr int D = 2;
    constexpr int KS = 4;
    constexpr int SW = 16 * TW;
    constexpr int LW = 64 * TW * KS;
    constexpr int PITCH = SW + 4;
    constexpr int ROWS = 4 * B2M_TILE;
    constexpr int STRIP = B2M_TILE * PITCH;
    __shared__ float smem[ROWS * PITCH];
    __shared__ uint8_t utab[4][128];
    __shared__ unsigned lockw;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 15, q = lane >> 4;
    const int64_t wg = a.xcd_start ? wg_index_balanced(a.xcd_start, a.wg_per_tile) : wg_index(a.nwg, a.xcd_per);
    if (wg < 0) return;
    int64_t stile = wg / a.nstrips;                        // supertile: tiles 4 * stile .. + 3
    const int strip = (int)(wg % a.nstrips);
    const int64_t nst = (a.ntiles + 3) >> 2;
    if (stile >= nst) return;
    B2M_CLOCK_BEGIN();
#ifdef B2M_STAMPS
    unsigned long long cs_begin, cs_t0, cs_t1, cs_t2, cs_wait = 0, cs_hold = 0, cs_fail = 0, cs_units = 0, cs_start, cs_done, cs_end;
    B2M_STAMP(cs_begin);
#endif
    if (a.tile_order) stile = a.tile_order[stile];
    const int64_t tile0 = stile * 4;
    const int col0 = strip * SW;
    float* Cq = smem + wave * STRIP;                       // this wave's quarter: initialised and written out by it
    const int64_t ldr = a.ntiles * B2M_TILE;
    const int64_t row0 = tile0 * B2M_TILE;
    const int64_t qrow0 = row0 + (int64_t)wave * B2M_TILE;

    // ---- init the quarter: 0 | Y (accumulate) | + bias
    for (int e = lane; e < B2M_TILE * (SW / 4); e += 64) {
        const int row = e / (SW / 4), c4 = (e % (SW / 4)) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const int64_t grow = qrow0 + row;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int col = col0 + c4 + u;
            if (col < a.cout) {
                float t = a.bias ? a.bias[col] : 0.f;
                if (a.accumulate && grow < a.n_out) t += a.y[grow * a.ldy + col];
                v[u] = t;
            }
        }
        *(f32x4*)&Cq[row * PITCH + c4] = v;
    }
    if (threadIdx.x == 0) lockw = 0u;

    // ---- units: lane k holds the pair counts of offset k in the four tiles
    int cnt0 = 0, cnt1 = 0, cnt2 = 0, cnt3 = 0;
    if (lane < a.K) {
        const int32_t* cp = a.rb_cnt + (int64_t)lane * a.ntiles + tile0;
        cnt0 = cp[0];
        if (tile0 + 1 < a.ntiles) cnt1 = cp[1];
        if (tile0 + 2 < a.ntiles) cnt2 = cp[2];
        if (tile0 + 3 < a.ntiles) cnt3 = cp[3];
    }
    const int ntot = cnt0 + cnt1 + cnt2 + cnt3;            // <= 256
    // The units of the supertile, largest first: whole rounds of 64 pairs (four row groups), then the offsets' last partial
    // rounds with 3, 2, 1 groups.  DEAL 0: snake order over the table (the four waves work on neighbouring entries -- the same
    // offset's rounds -- at the same time and share its weight blocks in L1); DEAL 1: every wave takes a contiguous quarter of
    // the four-group units (different offsets at any time), the partial rounds in snake order.
    int nuw;                                               // units of this wave
    int myunits = 0;                                       // lane i: this wave's i-th unit = offset | round << 5
    {
        const int full = ntot >> 6, gl = ((ntot & 63) + 15) >> 4;      // whole rounds, row groups of the partial one
        const int c4 = full + (gl == 4);                               // four-group units of this offset
        // exclusive scan of c4 (0..4) over the lanes from three ballots
        const uint64_t b0 = __ballot(c4 & 1), b1 = __ballot(c4 & 2), b2 = __ballot(c4 & 4);
        const int p4 = prefix_popc(b0) + 2 * prefix_popc(b1) + 4 * prefix_popc(b2);
        const int t4 = __builtin_popcountll(b0) + 2 * __builtin_popcountll(b1) + 4 * __builtin_popcountll(b2);
        const uint64_t m3 = __ballot(gl == 3), m2 = __ballot(gl == 2), m1 = __ballot(gl == 1);
        const int t3 = t4 + __builtin_popcountll(m3), t2 = t3 + __builtin_popcountll(m2);
        const int NU = t2 + __builtin_popcountll(m1);
        uint8_t* tab = utab[wave];
        for (int r = 0; r < c4; ++r) tab[p4 + r] = (uint8_t)(lane | (r << 5));
        if (gl == 3) tab[t4 + prefix_popc(m3)] = (uint8_t)(lane | (full << 5));
        if (gl == 2) tab[t3 + prefix_popc(m2)] = (uint8_t)(lane | (full << 5));
        if (gl == 1) tab[t2 + prefix_popc(m1)] = (uint8_t)(lane | (full << 5));
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        if constexpr (DEAL == 0) {
            auto pos_of = [&](int j) -> int { return 8 * (j >> 1) + ((j & 1) ? 7 - wave : wave); };
            nuw = __builtin_popcountll(__ballot(lane < 32 && pos_of(lane) < NU));
            if (lane < nuw) myunits = tab[pos_of(lane)];
        } else {
            const int q4 = t4 >> 2;                                   // four-group units per wave, contiguous
            const int rest = NU - 4 * q4;                             // the others: snake order
            auto pos_of = [&](int j) -> int { return 4 * q4 + 8 * (j >> 1) + ((j & 1) ? 7 - wave : wave); };
            const int nr = __builtin_popcountll(__ballot(lane < 32 && pos_of(lane) < NU));
            nuw = q4 + nr;
            (void)rest;
            if (lane < q4) myunits = tab[wave * q4 + lane];
            else if (lane < nuw) myunits = tab[pos_of(lane - q4)];
        }
    }
    __syncthreads();                                       // every quarter initialised, the lock word cleared
#ifdef B2M_STAMPS
    B2M_STAMP(cs_start);
#endif

    // unit i of this wave: offset k, pairs [p0, pe) of the offset's concatenated list, G row groups (all scalar)
    auto unit_info = [&](int i, int& k, int& p0, int& pe, int& G) {
        const int e = __builtin_amdgcn_readlane(myunits, i);
        k = e & 31;
        const int n = __builtin_amdgcn_readlane(ntot, k);
        p0 = (e >> 5) * 64;
        pe = p0 + 64 < n ? p0 + 64 : n;
        G = (pe - p0 + 15) >> 4;
    };

    if (nuw > 0) {
        // pair list of a unit: lane L fetches pair p = p0 + L of the four tiles' concatenated lists -- tile t where
        // P_t <= p < P_t+1, slot p - P_t -- and keeps `meta` = 64 t (the tile's first row inside the strip) or 256 (no such pair)
        auto list_addr = [&](int k, int p0, int pe, uint32_t& slot, int& meta) {
            const int P1 = __builtin_amdgcn_readlane(cnt0, k), P2 = P1 + __builtin_amdgcn_readlane(cnt1, k),
                      P3 = P2 + __builtin_amdgcn_readlane(cnt2, k);
            const int p = p0 + lane;
            const int t = (p >= P1) + (p >= P2) + (p >= P3);
            const int sl = p - (t == 0 ? 0 : t == 1 ? P1 : t == 2 ? P2 : P3);
            const bool valid = p < pe;
            slot = valid ? (uint32_t)(64 * t + sl) : 0u;
            meta = valid ? 64 * t : ROWS;
        };
        auto list_load = [&](int k, int p0, int pe, int& r_in, int& r_out, int& meta) {
            uint32_t slot;
            list_addr(k, p0, pe, slot, meta);
            const int64_t base = (int64_t)k * ldr + row0 + slot;
            r_in = a.rb_in[base];
            r_out = a.rb_out[base];
        };
        auto list_load_hl = [&](int k, int p0, int pe, int& r_in, int& r_out, int& meta) {
            uint32_t slot;
            list_addr(k, p0, pe, slot, meta);
            const int64_t base = (int64_t)k * ldr + row0;                  // wave-uniform
            const int32_t* pin = a.rb_in + base;
            const uint8_t* pout = a.rb_out + base;
            asm volatile("global_load_dword %0, %1, %2" : "+v"(r_in) : "v"(slot * 4u), "s"(pin) : "memory");
            asm volatile("global_load_ubyte %0, %1, %2" : "+v"(r_out) : "v"(slot), "s"(pout) : "memory");
        };
        auto list_words = [&](int r_in, int r_out, int meta, uint32_t (&w)[NG]) {
            const uint32_t word = meta >= ROWS ? ((uint32_t)ROWS << 23) : ((uint32_t)r_in | ((uint32_t)(meta + r_out) << 23));
#pragma unroll
            for (int g = 0; g < NG; ++g) w[g] = (uint32_t)__builtin_amdgcn_ds_bpermute((16 * g + i) << 2, (int)word);
        };

        const int nch1 = a.c1 >> 4, NC = (a.c1 + a.c2) >> 4;
        const uint32_t wlo = (uint32_t)lane * 16u;
        const uint32_t q16 = (uint32_t)q * 16u;
        const uint32_t ld1 = (uint32_t)a.ldx1 * 4u, ld2 = (uint32_t)a.ldx2 * 4u;
        const uint32_t wstrip = (uint32_t)strip * (uint32_t)NC;
        const uint32_t wkstride = (uint32_t)a.nstrips * (uint32_t)NC;
        f32x4 av[D][NG], bw[D][TW];
#pragma unroll
        for (int j = 0; j < D; ++j) {
#pragma unroll
            for (int g = 0; g < NG; ++g) av[j][g] = f32x4{};
#pragma unroll
            for (int u = 0; u < TW; ++u) bw[j][u] = f32x4{};
        }
#define B2M_BV(j, s, t) bw[j][(TW * (s) + (t)) >> 2][(TW * (s) + (t)) & 3]
        auto src_of = [&](int c, uint32_t& ld4) -> const char* {
            const bool first = c < nch1;
            ld4 = first ? ld1 : ld2;
            return (first ? (const char*)a.x1 + c * 64 : (const char*)a.x2 + (c - nch1) * 64);
        };
        auto gather = [&](int j, int g, const char* src, uint32_t ld4, uint32_t word, bool present) {
            const uint32_t off = __umul24(word & 0x7FFFFFu, ld4) + q16;
            const uint64_t em = present ? ~0ull : 1ull;
            asm volatile("s_mov_b64 exec, %3\n\tglobal_load_dwordx4 %0, %1, %2\n\ts_mov_b64 exec, -1"
                         : "+v"(av[j][g]) : "v"(off), "s"(src), "s"(em) : "memory");
        };
        auto weights = [&](int j, int k, int c) {
            const uint32_t blk = (uint32_t)k * wkstride + wstrip + (uint32_t)c;
            const char* wsrc = (const char*)a.wp + (size_t)blk * (size_t)(LW * 4);
#pragma unroll
            for (int u = 0; u < TW; ++u) {
                if (u == 0) asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(bw[j][u]) : "v"(wlo), "s"(wsrc) : "memory");
                else if (u == 1) asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "+v"(bw[j][u]) : "v"(wlo), "s"(wsrc) : "memory");
                else asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048" : "+v"(bw[j][u]) : "v"(wlo), "s"(wsrc) : "memory");
            }
        };

        // ---- prologue: lists of the first three units, operands of the first D steps
        uint32_t wC[NG], wN[NG];
        int rawi = 0, rawo = 0, rawm;
        int jC = 0, jN = nuw > 1 ? 1 : -1, jNN = nuw > 2 ? 2 : -1;
        int kC, GC, kN, GN;
        {
            int p0, pe, r0i, r0o, r0m, r1i, r1o, r1m;
            unit_info(0, kC, p0, pe, GC);
            list_load(kC, p0, pe, r0i, r0o, r0m);
            unit_info(jN < 0 ? 0 : jN, kN, p0, pe, GN);
            list_load(kN, p0, pe, r1i, r1o, r1m);
            list_words(r0i, r0o, r0m, wC);
            list_words(r1i, r1o, r1m, wN);
        }
#pragma unroll
        for (int j = 0; j < D; ++j) {
            uint32_t ld4;
            const char* src = src_of(j, ld4);
#pragma unroll
            for (int g = 0; g < NG; ++g) gather(j, g, src, ld4, wC[g], g < GC);
            weights(j, kC, j);
        }
        {
            int k, p0, pe, G;
            unit_info(jNN < 0 ? (jN < 0 ? jC : jN) : jNN, k, p0, pe, G);
            list_load_hl(k, p0, pe, rawi, rawo, rawm);
        }

        f32x4 acc[NG][TW];
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int t = 0; t < TW; ++t) acc[g][t] = f32x4{0.f, 0.f, 0.f, 0.f};

        auto mfma_group = [&](int j, int g) {
            if constexpr (TW == 3) {
                asm volatile(
                    "v_mfma_f32_16x16x4_f32 %0, %7, %3, %0\n\tv_mfma_f32_16x16x4_f32 %1, %8, %3, %1\n\tv_mfma_f32_16x16x4_f32 %2, %9, %3, %2\n\t"
                    "v_mfma_f32_16x16x4_f32 %0, %10, %4, %0\n\tv_mfma_f32_16x16x4_f32 %1, %11, %4, %1\n\tv_mfma_f32_16x16x4_f32 %2, %12, %4, %2\n\t"
                    "v_mfma_f32_16x16x4_f32 %0, %13, %5, %0\n\tv_mfma_f32_16x16x4_f32 %1, %14, %5, %1\n\tv_mfma_f32_16x16x4_f32 %2, %15, %5, %2\n\t"
                    "v_mfma_f32_16x16x4_f32 %0, %16, %6, %0\n\tv_mfma_f32_16x16x4_f32 %1, %17, %6, %1\n\tv_mfma_f32_16x16x4_f32 %2, %18, %6, %2"
                    : "+v"(acc[g][0]), "+v"(acc[g][1]), "+v"(acc[g][2])
                    : "v"(av[j][g][0]), "v"(av[j][g][1]), "v"(av[j][g][2]), "v"(av[j][g][3]),
                      "v"(B2M_BV(j, 0, 0)), "v"(B2M_BV(j, 0, 1)), "v"(B2M_BV(j, 0, 2)), "v"(B2M_BV(j, 1, 0)), "v"(B2M_BV(j, 1, 1)), "v"(B2M_BV(j, 1, 2)),
                      "v"(B2M_BV(j, 2, 0)), "v"(B2M_BV(j, 2, 1)), "v"(B2M_BV(j, 2, 2)), "v"(B2M_BV(j, 3, 0)), "v"(B2M_BV(j, 3, 1)), "v"(B2M_BV(j, 3, 2))
                    : "memory");
            } else {
                asm volatile(
                    "v_mfma_f32_16x16x4_f32 %0, %6, %2, %0\n\tv_mfma_f32_16x16x4_f32 %1, %7, %2, %1\n\t"
                    "v_mfma_f32_16x16x4_f32 %0, %8, %3, %0\n\tv_mfma_f32_16x16x4_f32 %1, %9, %3, %1\n\t"
                    "v_mfma_f32_16x16x4_f32 %0, %10, %4, %0\n\tv_mfma_f32_16x16x4_f32 %1, %11, %4, %1\n\t"
                    "v_mfma_f32_16x16x4_f32 %0, %12, %5, %0\n\tv_mfma_f32_16x16x4_f32 %1, %13, %5, %1"
                    : "+v"(acc[g][0]), "+v"(acc[g][1])
                    : "v"(av[j][g][0]), "v"(av[j][g][1]), "v"(av[j][g][2]), "v"(av[j][g][3]),
                      "v"(B2M_BV(j, 0, 0)), "v"(B2M_BV(j, 0, 1)), "v"(B2M_BV(j, 1, 0)), "v"(B2M_BV(j, 1, 1)),
                      "v"(B2M_BV(j, 2, 0)), "v"(B2M_BV(j, 2, 1)), "v"(B2M_BV(j, 3, 0)), "v"(B2M_BV(j, 3, 1))
                    : "memory");
            }
        };

        for (;;) {
            int c0 = 0;
            do {
                const bool wrap = c0 + D >= NC;
                const int kT = wrap ? kN : kC;
                const int cT = wrap ? 0 : c0 + D;
                uint32_t wT[NG];
#pragma unroll
                for (int g = 0; g < NG; ++g) wT[g] = wrap ? wN[g] : wC[g];
                const int GT = wrap ? GN : GC;
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    uint32_t ld4;
                    const char* src = src_of(cT + j, ld4);
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        if (g < GC) {
                            asm volatile("s_waitcnt vmcnt(%0)" :: "n"((D - 1) * (NG + TW) + g) : "memory");
                            mfma_group(j, g);
                        }
                        gather(j, g, src, ld4, wT[g], g < GT);
                    }
                    weights(j, kT, cT + j);
                }
                c0 += D;
            } while (c0 < NC);
            // ---- add the unit's result into the shared strip, under the workgroup's lock
            asm volatile("s_nop 15" ::: "memory");
#ifdef B2M_STAMPS
            B2M_STAMP(cs_t0);
#endif
            if constexpr (!(DBG & 1)) {
                for (;;) {
                    unsigned old = 1u;
                    if (lane == 0) {
                        unsigned expect = 0u;
                        __hip_atomic_compare_exchange_strong(&lockw, &expect, 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        old = expect;
                    }
                    if (__builtin_amdgcn_readfirstlane(old) == 0u) break;
#ifdef B2M_STAMPS
                    cs_fail += 1;
#endif
                    __builtin_amdgcn_s_sleep(1);
                }
            }
#ifdef B2M_STAMPS
            B2M_STAMP(cs_t1);
#endif
            asm volatile("" ::: "memory");
            // TW = 3 (registers to spare at three waves per SIMD): two row groups' reads in flight at a time -- the lock is held
            // for two LDS round trips, not four
            if constexpr (TW == 2) {
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    if (g < GC) {
                        const uint32_t orow = wC[g] >> 23;
                        if (orow < (uint32_t)ROWS) {
                            float* rowp = smem + orow * PITCH + 4 * q;
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
            } else
#pragma unroll
            for (int h = 0; h < NG / 2; ++h) {
                if (2 * h < GC) {
                    const uint32_t orA = wC[2 * h] >> 23, orB = wC[2 * h + 1] >> 23;
                    const bool vA = orA < (uint32_t)ROWS, vB = 2 * h + 1 < GC && orB < (uint32_t)ROWS;
                    float* rowA = smem + orA * PITCH + 4 * q;
                    float* rowB = smem + orB * PITCH + 4 * q;
                    f32x4 oldA[TW], oldB[TW];
                    if (vA) {
#pragma unroll
                        for (int t = 0; t < TW; ++t) oldA[t] = *(const f32x4*)(rowA + 16 * t);
                    }
                    if (vB) {
#pragma unroll
                        for (int t = 0; t < TW; ++t) oldB[t] = *(const f32x4*)(rowB + 16 * t);
                    }
                    if (vA) {
#pragma unroll
                        for (int t = 0; t < TW; ++t) *(f32x4*)(rowA + 16 * t) = oldA[t] + acc[2 * h][t];
                    }
                    if (vB) {
#pragma unroll
                        for (int t = 0; t < TW; ++t) *(f32x4*)(rowB + 16 * t) = oldB[t] + acc[2 * h + 1][t];
                    }
#pragma unroll
                    for (int t = 0; t < TW; ++t) { acc[2 * h][t] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[2 * h + 1][t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
                }
            }
            asm volatile("" ::: "memory");
            if (!(DBG & 1) && lane == 0) __hip_atomic_store(&lockw, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#ifdef B2M_STAMPS
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            B2M_STAMP(cs_t2);
            cs_wait += cs_t1 - cs_t0; cs_hold += cs_t2 - cs_t1; cs_units += 1;
#endif
            if (jN < 0) break;
            // ---- advance
            jC = jN; kC = kN; GC = GN;
#pragma unroll
            for (int g = 0; g < NG; ++g) wC[g] = wN[g];
            jN = jNN;
            if (jN >= 0) { int p0, pe; unit_info(jN, kN, p0, pe, GN); }
            {
                int li, lo;
                asm volatile("s_waitcnt vmcnt(%4)\n\tv_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(li), "=&v"(lo) : "v"(rawi), "v"(rawo),
                             "n"(D * (NG + TW)) : "memory");
                list_words(li, lo, rawm, wN);
            }
            jNN = (jN >= 0 && jN + 1 < nuw) ? jN + 1 : -1;
            {
                int k, p0, pe, G;
                unit_info(jNN < 0 ? (jN < 0 ? jC : jN) : jNN, k, p0, pe, G);
                list_load_hl(k, p0, pe, rawi, rawo, rawm);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
#ifdef B2M_STAMPS
    B2M_STAMP(cs_done);
#endif
    __syncthreads();                                       // every unit of the supertile is in the strip
#ifdef B2M_STAMPS
    B2M_STAMP(cs_end);
    if (lane == 0) {
        atomicAdd(&g_stamps[0], cs_wait); atomicAdd(&g_stamps[1], cs_hold); atomicAdd(&g_stamps[2], cs_fail);
        atomicAdd(&g_stamps[3], cs_end - cs_begin); atomicAdd(&g_stamps[4], cs_units); atomicAdd(&g_stamps[5], 1ull);
        atomicAdd(&g_stamps[6], cs_start - cs_begin); atomicAdd(&g_stamps[7], cs_end - cs_done);
    }
#endif

    // ---- column sums and write-out of this wave's quarter (= tile tile0 + wave)
    if (tile0 + wave >= a.ntiles) return;
    if (a.stats) strip_column_sums<SW>(a, Cq, tile0 + wave, qrow0, col0, lane);
    for (int e = lane; e < B2M_TILE * (SW / 4); e += 64) {
        const int row = e / (SW / 4), c4 = (e % (SW / 4)) * 4;
        const int64_t grow = qrow0 + row;
        if (grow >= a.n_out) continue;
        f32x4 v = *(const f32x4*)&Cq[row * PITCH + c4];
        const int col = col0 + c4;
        float* dst = a.y + grow * a.ldy + col;
        if (a.vec_store && col + 3 < a.cout) {
            if (a.ep_scale) v = conv_epilogue(a, v, grow, col);
            *(f32x4*)dst = v;
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) if (col + u < a.cout) dst[u] = v[u];
        }
    }
    B2M_CLOCK_END(0);
}
#undef B2M_BV
