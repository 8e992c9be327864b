// conv_fwd_pipe_kernel: the never-draining form of the forward / data-gradient kernel for the shapes that carry the
// FLOPs (real rulebook, 16-channel chunks, 32-bit addressable operands, un-split maps).  Included by conv.hip.
//
// Same work decomposition as conv_fwd_kernel -- one wave owns (tile of 64 output rows, strip of 16*TW output
// channels), walks the tile's active offsets, accumulates an offset over all input-channel chunks in registers and
// adds the result into its private LDS strip -- but the walk is ONE flat sequence of steps (offset, chunk) that is
// software-pipelined D steps deep ACROSS offset boundaries:
//
//   step s:   [MFMAs of step s from register buffer s % D]   [loads of step s + D into the same buffer]
//
// so while the last chunks of an offset multiply, the first chunks of the next offset are already in flight, and the
// pair lists are fetched two offsets ahead.  In conv_fwd_kernel every offset started with an exposed list fetch
// (~2 k cycles) and an exposed first operand fetch and ended with a drained pipeline (profiles/r01_conv_stall_analysis.md:
// a wave owned the MFMA pipe a quarter of its life).  Every load is unconditional and the loop bodies are unrolled over
// the D buffers, so the number of loads in flight is static and the waits are counted; past the last offset the
// prefetches re-read the last offset (valid addresses, never used).  The deeper pipeline needs ~200 VGPRs: 2 waves
// per SIMD, each of them MFMA-dense.
//
// Operand roles are swapped against conv_fwd_kernel: the weights are the MFMA "A" operand and the gathered rows the
// "B" operand (the register images of both are identical for 16x16x4, so the packed weight image is unchanged).  The
// result tile is then D[channel][pair]: a lane holds FOUR CONSECUTIVE CHANNELS of ONE pair, and the add into the LDS
// strip is one 16-byte read-modify-write per (row group, 16-column tile) instead of four 4-byte ones; the lane needs
// only the output row of its own pair.  Pair lists: lane L of the wave loads slot L of the offset (input row + output
// row, packed into one word: rows < 2^24), and the four words a lane needs (pairs i, 16+i, 32+i, 48+i) come from a
// cross-lane permute -- 2 loads + 4 ds_bpermute per offset instead of 8 loads.
#pragma once
#include <type_traits>

// DBG (diagnostic builds of tools/pipe_breakdown.py only, results are WRONG): 1 = no strip flush, 2 = no gathers inside the
// loop, 4 = no weight loads inside the loop -- each removes one component so that its cost shows in the launch time
// WPB = waves per workgroup.  The waves never communicate, but a workgroup's LDS and wave slots are only released when
// its LAST wave ends, and the items of a workgroup differ in work (active offsets, row groups): with 4 waves per
// workgroup only 1.6 of the 2 wave slots per SIMD were occupied on average (PMC, profiles/r02_pipe_analysis.md).
// One wave per workgroup frees every slot the moment its item is done.
template <int D, int TW, int DBG = 0, int WPB = 1, bool SKIPG = false>
__global__ __launch_bounds__(64 * WPB, 2) void conv_fwd_pipe_kernel(ConvArgs a) {
    constexpr int KS = 4;                     // k-steps per 16-channel chunk == floats per lane per gathered row
    constexpr int SW = 16 * TW;               // output channels per strip
    constexpr int LW = 64 * TW * KS;          // floats per packed weight block
    constexpr int PITCH = SW + 4;             // strip row pitch in floats: 16-byte multiples that do not alias banks
    constexpr int WROWS = B2M_TILE + 1;       // 64 rows + the spare row padded pairs are steered to
    __shared__ float smem[WPB * WROWS * PITCH];
    const int lane = threadIdx.x & 63;
    const int wave = WPB == 1 ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 15, q = lane >> 4;
    const int64_t wg = wg_index(a.nwg, a.xcd_per);
    if (wg < 0) return;
    const int64_t item = wg * WPB + wave;
    const int64_t tile = item / a.nstrips;
    const int strip = (int)(item % a.nstrips);
    if (tile >= a.ntiles) return;             // whole wave leaves; there is no barrier in this kernel
    const int col0 = strip * SW;
    const int nch1 = a.c1 >> 4, NC = (a.c1 + a.c2) >> 4;      // chunks of the first source / in all (NC % D == 0)
    const int64_t ldr = a.ntiles * B2M_TILE;
    const int64_t row0 = tile * B2M_TILE;
    float* Cs = smem + wave * (WROWS * PITCH);
#ifdef B2M_STAMPS
    unsigned long long st_begin, st_a = 0, st_b = 0, st_loop0 = 0, st_end, st_mfma = 0, st_issue = 0, st_flush = 0, st_adv = 0, st_steps = 0, st_offs = 0;
    B2M_STAMP(st_begin);
#endif

    // ---- init the strip: 0 | Y (accumulate) | + bias
    for (int e = lane; e < B2M_TILE * (SW / 4); e += 64) {
        const int row = e / (SW / 4), c4 = (e % (SW / 4)) * 4;
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
        *(f32x4*)&Cs[row * PITCH + c4] = v;
    }

    // ---- active offsets (K <= 128): lane k holds the pair count of offset k / k + 64
    int cnt0 = 0, cnt1 = 0;
    if (lane < a.K) cnt0 = a.rb_cnt[(int64_t)lane * a.ntiles + tile];
    if (lane + 64 < a.K) cnt1 = a.rb_cnt[(int64_t)(lane + 64) * a.ntiles + tile];
    const uint64_t m0 = __ballot(cnt0 > 0), m1 = __ballot(cnt1 > 0);
    auto next_active = [&](int k) -> int {    // first active offset after k, or -1 (scalar)
        int kk = k + 1;
        if (kk < 64) {
            const uint64_t r = m0 >> kk;
            if (r) return kk + __builtin_ctzll(r);
            kk = 64;
        }
        if (kk < 128) {
            const uint64_t r = m1 >> (kk - 64);
            if (r) return kk + __builtin_ctzll(r);
        }
        return -1;
    };
    auto groups_of = [&](int k) -> int {
        const int n = k < 64 ? __builtin_amdgcn_readlane(cnt0, k) : __builtin_amdgcn_readlane(cnt1, k - 64);
        return (n + 15) >> 4;
    };

    int kC = next_active(-1);
    if (kC >= 0) {
        // pair list of an offset: slot `lane` -> word = input row | output row << 24 (padded slot: row 0 -> spare row)
        auto list_load = [&](int k, int& r_in, int& r_out) {
            const int64_t base = (int64_t)k * ldr + row0 + lane;
            r_in = a.rb_in[base];
            r_out = a.rb_out[base];
        };
        auto list_words = [&](int r_in, int r_out, uint32_t (&w)[NG]) {
            const uint32_t word = r_in < 0 ? ((uint32_t)B2M_TILE << 24) : ((uint32_t)r_in | ((uint32_t)r_out << 24));
#pragma unroll
            for (int g = 0; g < NG; ++g) w[g] = (uint32_t)__builtin_amdgcn_ds_bpermute((16 * g + i) << 2, (int)word);
        };

        const uint32_t wlo = (uint32_t)lane * 16u;         // block layout [u][lane][4]: pack_pos()
        const uint32_t q16 = (uint32_t)q * 16u;
        const uint32_t ld1 = (uint32_t)a.ldx1 * 4u, ld2 = (uint32_t)a.ldx2 * 4u;
        float av[D][NG][KS], bv[D][KS][TW];
        // all loads of step (offset k, chunk c) into register buffer j
        auto issue = [&](int j, int k, int c, const uint32_t (&w)[NG], bool in_loop, int gt) {
            const char* wsrc = (const char*)(a.wp + (((int64_t)k * a.nstrips + strip) * NC + c) * LW);
            float wv[TW * KS];
            if (!((DBG & 4) && in_loop)) {
#pragma unroll
                for (int u = 0; u < TW * KS / 4; ++u) {
                    const f32x4 w4 = *(const f32x4*)(wsrc + (wlo + 1024 * u));
                    wv[4 * u] = w4[0]; wv[4 * u + 1] = w4[1]; wv[4 * u + 2] = w4[2]; wv[4 * u + 3] = w4[3];
                }
#pragma unroll
                for (int s = 0; s < KS; ++s)
#pragma unroll
                    for (int t = 0; t < TW; ++t) bv[j][s][t] = wv[TW * s + t];
            }
            if ((DBG & 2) && in_loop) return;
            const bool first = c < nch1;                                    // wave-uniform source select
            const char* src = (const char*)(first ? a.x1 + (c << 4) : a.x2 + ((c - nch1) << 4));
            const uint32_t ld4 = first ? ld1 : ld2;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (!SKIPG || g < gt) {         // SKIPG: no gather for row groups the target offset does not have (wave-uniform)
                    const uint32_t off = __umul24(w[g] & 0xFFFFFFu, ld4) + q16;
                    const f32x4 v = *(const f32x4*)(src + off);
                    av[j][g][0] = v[0]; av[j][g][1] = v[1]; av[j][g][2] = v[2]; av[j][g][3] = v[3];
                }
            }
        };

        // ---- prologue: lists of the first three offsets, operands of the first D steps
        uint32_t wC[NG], wN[NG];
        int rawi, rawo;
        int kN = next_active(kC);
        int kNc = kN < 0 ? kC : kN;
        int kNN = kN < 0 ? -1 : next_active(kN);
        {
            int r0i, r0o, r1i, r1o;
            list_load(kC, r0i, r0o);
            list_load(kNc, r1i, r1o);
            list_load(kNN < 0 ? kNc : kNN, rawi, rawo);
            list_words(r0i, r0o, wC);
            list_words(r1i, r1o, wN);
        }
        int GC = groups_of(kC), GN = groups_of(kNc);
#pragma unroll
        for (int j = 0; j < D; ++j) issue(j, kC, j, wC, false, GC);

        f32x4 acc[NG][TW];
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int t = 0; t < TW; ++t) acc[g][t] = f32x4{0.f, 0.f, 0.f, 0.f};

        auto mfma_step = [&](auto gc, int j) {
            constexpr int G = decltype(gc)::value;
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int g = 0; g < G; ++g)
#pragma unroll
                    for (int t = 0; t < TW; ++t)
                        acc[g][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[j][s][t], av[j][g][s], acc[g][t], 0, 0, 0);
        };

#ifdef B2M_STAMPS
        B2M_STAMP(st_loop0);
#endif
        for (;;) {
            for (int c0 = 0; c0 < NC; c0 += D) {
                // the D prefetches of this round target one offset: the current one, or -- in its last round -- the next
                const bool wrap = c0 + D >= NC;
                const int kT = wrap ? kNc : kC;
                const int cT = wrap ? c0 + D - NC : c0 + D;
                const int GT = wrap ? GN : GC;
                uint32_t wT[NG];
#pragma unroll
                for (int g = 0; g < NG; ++g) wT[g] = wrap ? wN[g] : wC[g];
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    // one straight-line block of GC*KS*TW MFMAs per row-group count (wave-uniform switch); k-step
                    // outermost, so consecutive MFMAs go to GC*TW independent accumulators
#ifdef B2M_STAMPS
                    B2M_STAMP(st_a);
#endif
                    switch (GC) {
                        case 1: mfma_step(std::integral_constant<int, 1>{}, j); break;
                        case 2: mfma_step(std::integral_constant<int, 2>{}, j); break;
                        case 3: mfma_step(std::integral_constant<int, 3>{}, j); break;
                        default: mfma_step(std::integral_constant<int, 4>{}, j); break;
                    }
                    __builtin_amdgcn_sched_barrier(0);
#ifdef B2M_STAMPS
                    B2M_STAMP(st_b); st_mfma += st_b - st_a; st_steps += 1;
#endif
                    issue(j, kT, cT + j, wT, true, GT);
                    __builtin_amdgcn_sched_barrier(0);
#ifdef B2M_STAMPS
                    B2M_STAMP(st_a); st_issue += st_a - st_b;
#endif
                }
            }
#ifdef B2M_STAMPS
            B2M_STAMP(st_a);
#endif
            // ---- add the offset's result into the strip: lane (i,q) holds channels 16t + 4q .. +3 of pair 16g + i
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g < GC && !(DBG & 1)) {
                    float* rowp = Cs + (wC[g] >> 24) * PITCH + 4 * q;
                    f32x4 old[TW];
#pragma unroll
                    for (int t = 0; t < TW; ++t) old[t] = *(const f32x4*)(rowp + 16 * t);
#pragma unroll
                    for (int t = 0; t < TW; ++t) {
                        *(f32x4*)(rowp + 16 * t) = old[t] + acc[g][t];
                        acc[g][t] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
            }
#ifdef B2M_STAMPS
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            B2M_STAMP(st_b); st_flush += st_b - st_a; st_offs += 1;
#endif
            if (kN < 0) break;
            // ---- advance: next offset becomes current; the list fetched an offset ago becomes next; fetch one more
            kC = kN; GC = GN;
#pragma unroll
            for (int g = 0; g < NG; ++g) wC[g] = wN[g];
            kN = kNN; kNc = kN < 0 ? kC : kN;
            list_words(rawi, rawo, wN);
            GN = groups_of(kNc);
            kNN = kN < 0 ? -1 : next_active(kN);
            list_load(kNN < 0 ? kNc : kNN, rawi, rawo);
#ifdef B2M_STAMPS
            B2M_STAMP(st_a); st_adv += st_a - st_b;
#endif
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
#ifdef B2M_STAMPS
    unsigned long long st_w0;
    B2M_STAMP(st_w0);
#endif

    // ---- write the strip (rows of the strip are 16-byte aligned: coalesced vector stores)
    for (int e = lane; e < B2M_TILE * (SW / 4); e += 64) {
        const int row = e / (SW / 4), c4 = (e % (SW / 4)) * 4;
        const int64_t grow = row0 + row;
        if (grow >= a.n_out) continue;
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
    B2M_STAMP(st_end);
    if (lane == 0 && kC >= 0) {
        atomicAdd(&g_stamps[0], st_mfma); atomicAdd(&g_stamps[1], st_issue); atomicAdd(&g_stamps[2], st_flush);
        atomicAdd(&g_stamps[3], st_end - st_begin); atomicAdd(&g_stamps[4], st_offs); atomicAdd(&g_stamps[5], 1ull);
        atomicAdd(&g_stamps[6], st_end - st_w0); atomicAdd(&g_stamps[7], st_steps);
        atomicAdd(&g_stamps[8], st_loop0 - st_begin); atomicAdd(&g_stamps[9], st_adv);
    }
#endif
}
