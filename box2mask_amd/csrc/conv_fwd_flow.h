// conv_fwd_flow_kernel: forward / data-gradient kernel for the shapes that carry the FLOPs (real rulebook, 16-channel
// chunks, 32-bit addressable operands, un-split maps).  Included by conv.hip.
//
// Work decomposition as in conv_fwd_kernel -- one wave owns (tile of 64 output rows, strip of 16*TW output channels),
// walks the tile's active offsets, accumulates an offset over all input-channel chunks in registers and adds the
// result into its private LDS strip -- with three changes that the s_memtime stamps of round 2 asked for
// (profiles/r02_conv_analysis.md: a wave of the first pipelined kernel spent 32 % of its life in its own MFMAs, the
// rest in load issue, waits, flushes and accumulator copies, and two waves per SIMD cannot fill the pipe from that):
//
//  * the walk over (offset, chunk) steps is one flat software pipeline, D steps deep, ACROSS offset boundaries: while
//    the last chunks of an offset multiply, the first chunks of the next one are in flight; pair lists are fetched two
//    offsets ahead;
//  * the row-group count G of an offset is a template parameter of the code that handles the whole offset (one
//    wave-uniform switch per offset, not per step), so the accumulators are locals of that code: the MFMAs update them
//    in place (the per-step switch made hipcc keep two copies of the accumulator file and move between them: 196
//    VGPRs).  <= 168 VGPRs -> three waves per SIMD;
//  * every step issues the same number of vector loads (4 gathers + TW weight pieces; row groups an offset does not
//    have gather row 0, an L1 hit), so the compiler's counted waits are exact: vmcnt((D-1)*(4+TW)) in front of an MFMA
//    block.  With gathers skipped for absent row groups the counts were conservative and every MFMA block also waited
//    for most of the step issued after it.
//
// Operand roles are swapped against conv_fwd_kernel: the weights are the MFMA "A" operand and the gathered rows the
// "B" operand (the register images of both are identical for 16x16x4, so the packed weight image is unchanged).  The
// result tile is then D[channel][pair]: a lane holds FOUR CONSECUTIVE CHANNELS of ONE pair, and the add into the LDS
// strip is one 16-byte read-modify-write per (row group, 16-column tile); the lane needs only the output row of its own
// pair.  Pair lists: lane L of the wave loads slot L of the offset (input row + output row, packed into one word: rows
// < 2^24), and the four words a lane needs (pairs i, 16+i, 32+i, 48+i) come from a cross-lane permute.
#pragma once
#include <type_traits>

// Column sums of a finished output strip (64 rows x SW columns in LDS, row pitch SW + 4): lane c adds up column c over
// the tile's valid rows and stores sum / sum of squares to stats[tile][0 / 1][col0 + c] -- what bn_stats_kernel would
// read the whole of Y again for.
// R16 (the F16 kernels in half-precision training): the sums are those of the values as STORED -- each fp32 accumulator rounded to
// binary16 first -- so that they are the statistics of the tensor the BatchNorm behind the layer reads.
template <int SW, bool R16 = false>
__device__ __forceinline__ void strip_column_sums(const ConvArgs& a, const float* strip, int64_t tile, int64_t row0, int col0, int lane) {
    constexpr int PITCH = SW + 4;
    const int64_t rem = a.n_out - row0;
    const int rows = rem < B2M_TILE ? (int)rem : B2M_TILE;
    if (lane < SW && col0 + lane < a.cout) {
        // fp64: the variance is a difference of these sums (8-row deep levels!).  Four independent chains (rows r, r+1,
        // r+2, r+3 of every group of four): the dependent fp64 add latency would otherwise be the whole cost
        double s[4] = {0., 0., 0., 0.}, s2[4] = {0., 0., 0., 0.};
        int r = 0;
        for (; r + 3 < rows; r += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float f = strip[(r + u) * PITCH + lane];
                const double v = R16 ? (double)(float)(_Float16)f : (double)f;
                s[u] += v; s2[u] = fma(v, v, s2[u]);
            }
        }
        for (; r < rows; ++r) {
            const float f = strip[r * PITCH + lane];
            const double v = R16 ? (double)(float)(_Float16)f : (double)f;
            s[0] += v; s2[0] = fma(v, v, s2[0]);
        }
        double* o = a.stats + tile * 2 * a.cout + col0 + lane;
        o[0] = (s[0] + s[1]) + (s[2] + s[3]); o[a.cout] = (s2[0] + s2[1]) + (s2[2] + s2[3]);
    }
}
#define tile_column_sums(a, strip, tile, row0, col0, lane) strip_column_sums<SW, (F16 != 0)>(a, strip, tile, row0, col0, lane)

// Half output (F16 kernels): four consecutive columns of output row `grow` -- the inference epilogue in fp32 (scale / shift,
// residual read as half, ReLU), one rounding to half, one 8-byte store.  ConvArgs::y / ep_res hold half data here.
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_half4(const ConvArgs& a, f32x4 v, int64_t grow, int col) {
    if (a.ep_scale) {
        const f32x4 s = *(const f32x4*)(a.ep_scale + col), b = *(const f32x4*)(a.ep_shift + col);
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = __builtin_fmaf(v[u], s[u], b[u]);
    }
    if (a.ep_res) {
        const f16x4 r = *(const f16x4*)((const _Float16*)a.ep_res + grow * a.ld_res + col);
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] += (float)r[u];
    }
    if (a.ep_relu) {
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = v[u] > 0.f ? v[u] : 0.f;
    }
    f16x4 h;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        // saturate instead of overflowing to infinity (|x| > 65504: an un-normalised activation; inf - inf = NaN three layers on)
        const float c = __builtin_fminf(__builtin_fmaxf(v[u], -65504.f), 65504.f);
        h[u] = (_Float16)c;
    }
    *(f16x4*)((_Float16*)a.y + grow * a.ldy + col) = h;
}

// DBG (diagnostic builds of tools/pipe_breakdown.py only, results are WRONG): 1 = no strip flush, 2 = no gathers inside
// the loop, 4 = no weight loads inside the loop, 8 = no MFMAs (everything else: the floor a faster multiply would leave) -- each removes one component so that its cost shows in the launch time
// WPB = 4: split maps (deep U-Net levels).  The four waves of a workgroup are four slices of ONE (tile, strip): the
// tile's active offsets are dealt round-robin to nslice / ncs slices and the input-channel chunks to ncs parts; the
// waves add their strips up in LDS behind one barrier and wave 0 writes (plain stores for exactly 4 slices, else fp32
// atomics into the pre-zeroed Y).
// HL = 1 (round 4): the operand loads of the hot loop are issued by hand -- asm statements with the waits counted here, not
// by hipcc -- so that the gather of an ABSENT row group can run with one active lane.  Every step still issues NG + TW
// vector loads (a static count: the waits stay exact), but a 16-byte-per-lane load occupies the CU's vector-memory return
// path for ~22 cycles whatever it reads (tools/micro/mfma_loads.hip: 46 B/clk per CU), and with 4 + TW full-width loads
// per step the one-group steps -- a third of all steps on the benchmark's maps -- were bound by that path, not by their
// 12 * TW MFMAs.  A conditional (branchy) narrow load in C++ loses hipcc's counted waits (vmcnt(0) in front of every MFMA
// block); the EXEC mask around ONE load instruction keeps the instruction count and costs two scalar moves.
// F16 = 1 (round 4, inference only: b2m_conv_fwd_h): activations, residual and output are IEEE half in HBM, the packed weights
// half, the accumulators and the LDS strip fp32.  A step is still one 64-byte piece of every gathered row and TW one-KiB
// weight pieces -- now 32 input channels -- and ONE v_mfma_f32_16x16x32_f16 per (row group, column tile) instead of four
// v_mfma_f32_16x16x4_f32: the same pipeline, waits and masks, a sixteenth of the MFMA cycles per channel.  No bias, no
// accumulate, no BatchNorm column sums; at most 4 slices (the combine of more is an fp32 atomic into Y).
// F16 = 2: 16 input channels per step (8-byte loads, v_mfma_f32_16x16x16_f16) for the layers whose 32-channel chunks cannot be
// dealt to the D steps of a round (32 input channels: one chunk per offset).
// UP = 1 (round 5): TRANSPOSED k2s2 maps in scatter form (b2m_conv_up).  Every fine row has exactly one (parent, offset) pair, so
// tiled over the fine (output) rows an offset of a tile has ~8 pairs: one half-empty row group per visit, 30 ... 45 TFLOP/s.
// Here the wave walks the map's DOWN rulebook -- tiled over the COARSE rows: up to 64 pairs per (tile, offset) --
// with the roles of its two row numbers exchanged: the gathered rows are the tile's own coarse rows (row0 + rb_out), and
// the result of pair j is stored straight to fine row rb_in[j] of Y (each fine row is written exactly once per launch: no
// LDS strip, no flush, no write-out; accumulate = a plain read-modify-write of the gradient already there: the row has one
// writer).
template <int D, int TW, int DBG = 0, int WPB = 1, int HL = 0, int F16 = 0, int UP = 0>
__global__ __launch_bounds__(64 * WPB, TW == 4 ? 2 : (TW == 2 && D == 2 && !UP) ? 4 : 3) void conv_fwd_flow_kernel(ConvArgs a) {
    static_assert(!F16 || HL, "the half variant exists with hand-issued loads only");
    static_assert(TW <= 3 || F16, "64-column strips: the half variants only");
    static_assert(!UP || (HL && WPB == 1 && !F16), "the scatter form: fp32, un-split, hand-issued loads");
    constexpr int KS = 4;                     // k-steps per 16-channel chunk == floats per lane per gathered row
    constexpr int ESZ = F16 ? 2 : 4;          // bytes per activation element
    constexpr int CSH = F16 == 1 ? 5 : 4;     // log2(input channels per step): 64 bytes of a row (F16 = 2: 32 bytes, see below)
    constexpr int RB = F16 == 2 ? 32 : 64;    // bytes of a row per step
    constexpr int SW = 16 * TW;               // output channels per strip
    constexpr int LW = 64 * TW * KS;          // floats per packed weight block
    constexpr int PITCH = SW + 4;             // strip row pitch in floats: 16-byte multiples that do not alias banks
    constexpr int STRIP = B2M_TILE * PITCH;   // 13 KiB per wave (TW = 3): twelve waves per CU
    __shared__ float smem[UP ? 4 : WPB * STRIP];
    const int lane = threadIdx.x & 63;
    const int wave = WPB == 1 ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 15, q = lane >> 4;
    const int64_t wg = a.xcd_start ? wg_index_balanced(a.xcd_start, a.wg_per_tile) : wg_index(a.nwg, a.xcd_per);
    if (wg < 0) return;
    const int nch1 = a.c1 >> CSH, NC = (a.c1 + a.c2) >> CSH;  // chunks of the first source / in all
    int64_t item = wg;
    int slice = 0, nks = 1, cb = 0, ce = NC;                  // offset slice of nks, chunk range [cb, ce): (ce - cb) % D == 0
    bool lead = true;
    if constexpr (WPB > 1) {
        const int64_t witem = wg * WPB + wave;
        const int sl = (int)(witem % a.nslice), cslice = sl % a.ncs;
        item = witem / a.nslice;
        slice = sl / a.ncs; nks = a.nslice / a.ncs;
        cb = NC / a.ncs * cslice; ce = cb + NC / a.ncs;
        lead = sl == 0;
    }
    // (32-bit: a launch has < 2^31 items -- the 64-bit division is ~200 scalar instructions at the head of every wave)
    int64_t tile = (int64_t)((uint32_t)item / (uint32_t)a.nstrips);
    const int strip = (int)((uint32_t)item % (uint32_t)a.nstrips);
    if (tile >= a.ntiles) return;             // the whole workgroup leaves (its waves share the item)
    B2M_CLOCK_BEGIN();
    B2M_RES_BEGIN();
#ifdef B2M_STAMPS
    unsigned long long fs_begin, fs_init = 0, fs_pro = 0, fs_t0 = 0, fs_t1 = 0, fs_t2 = 0, fs_loop = 0, fs_flush = 0, fs_adv = 0, fs_vis = 0, fs_walk = 0, fs_stat = 0;
    if constexpr (WPB == 1 && !F16 && HL == 1 && DBG == 0) { B2M_STAMP(fs_begin); }
#endif
    if (a.tile_order) tile = a.tile_order[tile];
    const int col0 = strip * SW;
    float* Cs = smem + wave * STRIP;
    const int64_t ldr = a.ntiles * B2M_TILE;
    const int64_t row0 = tile * B2M_TILE;

    // the pair counts of the tile's offsets: issued first, their round trip runs beside the strip init
    int cnt0 = 0, cnt1 = 0;
    if (lane < a.K) cnt0 = a.rb_cnt[(int64_t)lane * a.ntiles + tile];
    if (lane + 64 < a.K) cnt1 = a.rb_cnt[(int64_t)(lane + 64) * a.ntiles + tile];
    // ---- init the strip: 0 | Y (accumulate) | + bias
    // accumulate (a data gradient added onto the gradient already there): ALL of the tile's rows of Y are requested before the first
    // one is stored -- round 5; the general loop below waits for its four scalar loads in every one of its 8 / 12 rounds, twelve
    // dependent memory round trips at the head of the wave
    constexpr int NIT = B2M_TILE * (SW / 4) / 64;
    const bool vec_acc = !UP && !F16 && a.accumulate && a.nslice == 1 && a.vec_store && !a.bias && col0 + SW <= a.cout;
    if (vec_acc) {
        f32x4 pre[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int e = lane + 64 * it;
            const int row = e / (SW / 4), c4 = (e % (SW / 4)) * 4;
            const int64_t grow = row0 + row;
            pre[it] = grow < a.n_out ? *(const f32x4*)(a.y + grow * a.ldy + col0 + c4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int e = lane + 64 * it;
            *(f32x4*)&Cs[(e / (SW / 4)) * PITCH + (e % (SW / 4)) * 4] = pre[it];
        }
    }
    for (int e = lane; !UP && !vec_acc && e < B2M_TILE * (SW / 4); e += 64) {
        const int row = e / (SW / 4), c4 = (e % (SW / 4)) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const int64_t grow = row0 + row;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int col = col0 + c4 + u;
            if (col < a.cout) {
                float t = 0.f;
                if constexpr (!F16) {
                    t = (a.bias && lead) ? a.bias[col] : 0.f;
                    if (a.accumulate && a.nslice == 1 && grow < a.n_out) t += a.y[grow * a.ldy + col];
                }
                v[u] = t;
            }
        }
        *(f32x4*)&Cs[row * PITCH + c4] = v;
    }

#ifdef B2M_STAMPS
            if constexpr (WPB == 1 && !F16 && HL == 1 && DBG == 0) {
    B2M_STAMP(fs_init);
            }
#endif
    // ---- active offsets (K <= 128): lane k holds the pair count of offset k / k + 64 (loaded above the strip init)
    uint64_t m0 = __ballot(cnt0 > 0), m1 = __ballot(cnt1 > 0);
    if constexpr (WPB > 1) {                  // keep the active offsets whose rank among the active ones is ours
        const int r0 = prefix_popc(m0), r1 = __builtin_popcountll(m0) + prefix_popc(m1);
        m0 = __ballot(cnt0 > 0 && r0 % nks == slice);
        m1 = __ballot(cnt1 > 0 && r1 % nks == slice);
    }
    // Round 5: FULL visits first, chained in registers.  A visit with all 64 pairs (a third of the visits on the benchmark's
    // level-0 maps, a fifth on level 1: tools/fill_stats_units.py) maps pair j to output row j, so consecutive full visits add
    // into the SAME accumulator tiles: walked first, they need one flush at the end of their chain instead of one each -- half
    // of the flush work of a level-0 tile (a flush is 2 * G * TW 16-byte LDS accesses on the wave's critical path).  An offset
    // is carried as k | 128 once the walk has left the full ones (un-split maps only; ConvArgs::chain).
    uint64_t f0 = 0, f1 = 0;
    if constexpr (WPB == 1) {
        if (a.chain) { f0 = __ballot(cnt0 == B2M_TILE); f1 = __ballot(cnt1 == B2M_TILE); m0 &= ~f0; m1 &= ~f1; }
    }
    auto next_in = [&](uint64_t a0, uint64_t a1, int k) -> int {    // first set offset after k, or -1 (scalar)
        int kk = k + 1;
        if (kk < 64) {
            const uint64_t r = a0 >> kk;
            if (r) return kk + __builtin_ctzll(r);
            kk = 64;
        }
        if (kk < 128) {
            const uint64_t r = a1 >> (kk - 64);
            if (r) return kk + __builtin_ctzll(r);
        }
        return -1;
    };
    auto next_active = [&](int e) -> int {    // offset after e (-1: the first) in the walk: the full ones, then the rest | 128
        int k = e < 0 ? -1 : (e & 127);
        if constexpr (WPB == 1) {
            if (e < 128) {
                const int r = next_in(f0, f1, k);
                if (r >= 0) return r;
                k = -1;
            }
        }
        const int r = next_in(m0, m1, k);
        return r < 0 ? -1 : (r | 128);
    };
    auto groups_of = [&](int e) -> int {
        const int k = e & 127;
        const int n = k < 64 ? __builtin_amdgcn_readlane(cnt0, k) : __builtin_amdgcn_readlane(cnt1, k - 64);
        return (n + 15) >> 4;
    };

    // (DBG & 32, diagnostic: the walk runs TWICE per wave -- what a launch costs per wave outside the walk shows as
    // 2 * T(plain) - T(this); results are wrong by the second pass's sums)
#pragma unroll 1
    for (int rep = 0; rep < ((DBG & 32) ? 2 : 1); ++rep) {
    int kC = next_active(-1);
    if (kC >= 0) {
        // pair list of an offset: slot `lane` -> word = input row | output row << 24 (padded slot: row 0, output row 64)
        auto list_load = [&](int e, int& r_in, int& r_out) {
            const int64_t base = (int64_t)(e & 127) * ldr + row0 + lane;
            r_in = a.rb_in[base];
            r_out = a.rb_out[base];
        };
        // the same fetch issued by hand (HL): consumed one offset later behind `s_waitcnt vmcnt(D * (NG + TW))` -- at least that
        // many operand loads are younger -- instead of hipcc's vmcnt(0), which cannot see the hand-issued loads and would drain
        // all of them at every offset advance
        const uint32_t lane4 = (uint32_t)lane * 4u;
        auto list_load_hl = [&](int e, int& r_in, int& r_out) {
            const int64_t base = (int64_t)(e & 127) * ldr + row0;          // wave-uniform
            const int32_t* pin = a.rb_in + base;
            const uint8_t* pout = a.rb_out + base;
            // (destinations as IN/OUT operands, like the operand loads: the registers stay allocated up to the statement
            // that waits for them, whatever hipcc moves in between)
            asm volatile("global_load_dword %0, %1, %2" : "+v"(r_in) : "v"(lane4), "s"(pin) : "memory");
            asm volatile("global_load_ubyte %0, %1, %2" : "+v"(r_out) : "v"((uint32_t)lane), "s"(pout) : "memory");
        };
        auto list_words = [&](int r_in, int r_out, uint32_t (&w)[NG]) {
            // (UP: a padded slot gathers the tile's first row and is marked by the fine row 0xFFFFFF: nothing is stored for it)
            const uint32_t word = r_in < 0 ? (UP ? 0x00FFFFFFu : ((uint32_t)B2M_TILE << 24)) : ((uint32_t)r_in | ((uint32_t)r_out << 24));
#pragma unroll
            for (int g = 0; g < NG; ++g) w[g] = (uint32_t)__builtin_amdgcn_ds_bpermute((16 * g + i) << 2, (int)word);
        };

        const uint32_t wlo = (uint32_t)lane * (F16 == 2 ? 8u : 16u);         // packed block layout [u][lane][4 floats]: pack_pos()
        const uint32_t q16 = (uint32_t)q * (F16 == 2 ? 8u : 16u);       // the lane's bytes inside the row piece
        const uint32_t ld1 = (uint32_t)a.ldx1 * (uint32_t)ESZ, ld2 = (uint32_t)a.ldx2 * (uint32_t)ESZ;     // row pitch in bytes
        // (F16, 32-column strips on an image packed for 64-column strips: the image's strip s >> 1, its pieces 2 (s & 1), 2 (s & 1) + 1)
        const bool wide = F16 && TW == 2 && a.img_wide;
        const uint32_t wstrip = (uint32_t)(wide ? strip >> 1 : strip) * (uint32_t)NC;
        const uint32_t wkstride = (uint32_t)(wide ? a.nstrips >> 1 : a.nstrips) * (uint32_t)NC;
        const uint32_t wsub = wide ? (uint32_t)(strip & 1) * (F16 == 2 ? 1024u : 2048u) : 0u;
        // operand registers: buffer j holds row group g's four k-steps (one 16-byte gather) and the TW weight pieces; MFMA operand
        // (k-step s, column tile t) of the weights is float TW * s + t of the buffer's TW pieces.  With hand-issued loads every
        // load statement takes its destination as an IN/OUT operand: the previous content stays alive, in that very register, up
        // to the load -- hipcc, which does not know that a skipped group's load may still be in flight, would otherwise hand the
        // "dead" register to the next address computation (tests/test_isa.py traces this).
        using opv = std::conditional_t<F16 == 2, f32x2, f32x4>;           // one load's registers
        opv av[D][NG], bw[D][TW];
        if constexpr (HL) {
#pragma unroll
            for (int j = 0; j < D; ++j) {
#pragma unroll
                for (int g = 0; g < NG; ++g) av[j][g] = opv{};
#pragma unroll
                for (int u = 0; u < TW; ++u) bw[j][u] = opv{};
            }
        }
#define B2M_BV(j, s, t) bw[j][(TW * (s) + (t)) >> 2][(TW * (s) + (t)) & 3]
        // loads of step (offset k, chunk c) into register buffer j: NG gathers + TW weight pieces, always
        auto src_of = [&](int c, uint32_t& ld4) -> const char* {
            const bool first = c < nch1;                                        // wave-uniform source select
            ld4 = first ? ld1 : ld2;
            return (first ? (const char*)a.x1 + c * RB : (const char*)a.x2 + (c - nch1) * RB);
        };
        // `present` (wave-uniform): the step that will consume this buffer has row group g
        auto gather = [&](int j, int g, const char* src, uint32_t ld4, uint32_t word, bool present) {
            const uint32_t off = __umul24(UP ? (uint32_t)row0 + (word >> 24) : (word & 0xFFFFFFu), ld4) + q16;
            if constexpr (HL) {
                const uint64_t em = present ? ~0ull : 1ull;                 // absent group: one lane fetches, the rest keep stale registers
                if constexpr (F16 == 2)
                    asm volatile("s_mov_b64 exec, %3\n\tglobal_load_dwordx2 %0, %1, %2\n\ts_mov_b64 exec, -1"
                                 : "+v"(av[j][g]) : "v"(off), "s"(src), "s"(em) : "memory");
                else
                    asm volatile("s_mov_b64 exec, %3\n\tglobal_load_dwordx4 %0, %1, %2\n\ts_mov_b64 exec, -1"
                                 : "+v"(av[j][g]) : "v"(off), "s"(src), "s"(em) : "memory");
            } else {
                av[j][g] = *(const opv*)(src + off);
            }
        };
        auto weights = [&](int j, int e, int c) {
            const uint32_t blk = (uint32_t)(e & 127) * wkstride + wstrip + (uint32_t)c;     // wave-uniform
            size_t woff = (size_t)blk * ((size_t)(F16 == 2 ? LW * 2 : LW * 4) << (wide ? 1 : 0)) + wsub;
            if constexpr (F16 == 2) {
                // Wave-uniform, and told so: in the 16-channel-chunk variants hipcc computes this offset on the vector ALU and then
                // hands the asm statements below a VGPR pair for their "s" operand -- which `-S` prints without complaint and only
                // the assembler (`-c`) rejects ("invalid operand for instruction").
                woff = ((size_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(woff >> 32)) << 32) |
                       (size_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)woff);
            }
            const char* wsrc = (const char*)a.wp + woff;
#pragma unroll
            for (int u = 0; u < TW; ++u) {
                if constexpr (F16 == 2) {
                    if (u == 0) asm volatile("global_load_dwordx2 %0, %1, %2" : "+v"(bw[j][u]) : "v"(wlo), "s"(wsrc) : "memory");
                    else if (u == 1) asm volatile("global_load_dwordx2 %0, %1, %2 offset:512" : "+v"(bw[j][u]) : "v"(wlo), "s"(wsrc) : "memory");
                    else if (u == 2) asm volatile("global_load_dwordx2 %0, %1, %2 offset:1024" : "+v"(bw[j][u]) : "v"(wlo), "s"(wsrc) : "memory");
                    else asm volatile("global_load_dwordx2 %0, %1, %2 offset:1536" : "+v"(bw[j][u]) : "v"(wlo), "s"(wsrc) : "memory");
                } else if constexpr (HL) {
                    if (u == 0) asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(bw[j][u]) : "v"(wlo), "s"(wsrc) : "memory");
                    else if (u == 1) asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "+v"(bw[j][u]) : "v"(wlo), "s"(wsrc) : "memory");
                    else if (u == 2) asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048" : "+v"(bw[j][u]) : "v"(wlo), "s"(wsrc) : "memory");
                    else asm volatile("global_load_dwordx4 %0, %1, %2 offset:3072" : "+v"(bw[j][u]) : "v"(wlo), "s"(wsrc) : "memory");
                } else {
                    bw[j][u] = *(const opv*)(wsrc + (wlo + 1024u * u));
                }
            }
        };

        // ---- prologue: lists of the first three offsets, operands of the first D steps
        uint32_t wC[NG], wN[NG];
        int rawi = 0, rawo = 0;
        int kN = next_active(kC);
        int kNc = kN < 0 ? kC : kN;
        int kNN = kN < 0 ? -1 : next_active(kN);
        {
            int r0i, r0o, r1i, r1o;
            list_load(kC, r0i, r0o);
            list_load(kNc, r1i, r1o);
            list_words(r0i, r0o, wC);
            list_words(r1i, r1o, wN);
        }
        int GC = groups_of(kC), GN = groups_of(kNc);
#pragma unroll
        for (int j = 0; j < D; ++j) {
            uint32_t ld4;
            const char* src = src_of(cb + j, ld4);
#pragma unroll
            for (int g = 0; g < NG; ++g) gather(j, g, src, ld4, wC[g], g < GC);
            weights(j, kC, cb + j);
        }
        if constexpr (HL) list_load_hl(kNN < 0 ? kNc : kNN, rawi, rawo);
        else list_load(kNN < 0 ? kNc : kNN, rawi, rawo);  // behind the step loads, as in the steady state

        f32x4 acc[NG][TW];
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int t = 0; t < TW; ++t) acc[g][t] = f32x4{0.f, 0.f, 0.f, 0.f};

        // The 4*TW MFMAs of (row group g, chunk step in buffer j) as ONE asm statement with the accumulators tied
        // (in place).  Written with the builtin, hipcc un-tied the accumulators around the wave-uniform branches on the
        // group count (two copies of the accumulator file, v_mov chains, 196 VGPRs) and hoisted MFMA blocks over the loads
        // that refill their operand buffer.  k-step outermost: an accumulator is reused every TW MFMAs (>= 64 cycles
        // apart; the dependent latency is 40).  The hardware interlocks MFMA -> MFMA on the same accumulator; the one
        // software-visible hazard, a non-MFMA read of a fresh result, is covered by the s_nop in front of the flush.
        auto mfma_group = [&](int j, int g) {
            if constexpr (F16 == 2) {
                if constexpr (TW == 4) {
                    asm volatile("v_mfma_f32_16x16x16_f16 %0, %5, %4, %0\n\tv_mfma_f32_16x16x16_f16 %1, %6, %4, %1\n\t"
                                 "v_mfma_f32_16x16x16_f16 %2, %7, %4, %2\n\tv_mfma_f32_16x16x16_f16 %3, %8, %4, %3"
                                 : "+v"(acc[g][0]), "+v"(acc[g][1]), "+v"(acc[g][2]), "+v"(acc[g][3])
                                 : "v"(av[j][g]), "v"(bw[j][0]), "v"(bw[j][1]), "v"(bw[j][2]), "v"(bw[j][3]) : "memory");
                } else if constexpr (TW == 3) {
                    asm volatile("v_mfma_f32_16x16x16_f16 %0, %4, %3, %0\n\tv_mfma_f32_16x16x16_f16 %1, %5, %3, %1\n\t"
                                 "v_mfma_f32_16x16x16_f16 %2, %6, %3, %2"
                                 : "+v"(acc[g][0]), "+v"(acc[g][1]), "+v"(acc[g][2])
                                 : "v"(av[j][g]), "v"(bw[j][0]), "v"(bw[j][1]), "v"(bw[j][2]) : "memory");
                } else {
                    asm volatile("v_mfma_f32_16x16x16_f16 %0, %3, %2, %0\n\tv_mfma_f32_16x16x16_f16 %1, %4, %2, %1"
                                 : "+v"(acc[g][0]), "+v"(acc[g][1])
                                 : "v"(av[j][g]), "v"(bw[j][0]), "v"(bw[j][1]) : "memory");
                }
            } else if constexpr (F16) {
                // weights = A (lane (i, q): input channels 8q .. 8q + 7 of output channel 16t + i), gathered rows = B (lane (i, q):
                // the same eight channels of pair i): D[channel][pair] as in the fp32 form
                if constexpr (TW == 4) {          // (round 6: 64-column strips for the layers whose output channels come in 64s)
                    asm volatile("v_mfma_f32_16x16x32_f16 %0, %5, %4, %0\n\tv_mfma_f32_16x16x32_f16 %1, %6, %4, %1\n\t"
                                 "v_mfma_f32_16x16x32_f16 %2, %7, %4, %2\n\tv_mfma_f32_16x16x32_f16 %3, %8, %4, %3"
                                 : "+v"(acc[g][0]), "+v"(acc[g][1]), "+v"(acc[g][2]), "+v"(acc[g][3])
                                 : "v"(av[j][g]), "v"(bw[j][0]), "v"(bw[j][1]), "v"(bw[j][2]), "v"(bw[j][3]) : "memory");
                } else if constexpr (TW == 3) {
                    asm volatile("v_mfma_f32_16x16x32_f16 %0, %4, %3, %0\n\tv_mfma_f32_16x16x32_f16 %1, %5, %3, %1\n\t"
                                 "v_mfma_f32_16x16x32_f16 %2, %6, %3, %2"
                                 : "+v"(acc[g][0]), "+v"(acc[g][1]), "+v"(acc[g][2])
                                 : "v"(av[j][g]), "v"(bw[j][0]), "v"(bw[j][1]), "v"(bw[j][2]) : "memory");
                } else {
                    asm volatile("v_mfma_f32_16x16x32_f16 %0, %3, %2, %0\n\tv_mfma_f32_16x16x32_f16 %1, %4, %2, %1"
                                 : "+v"(acc[g][0]), "+v"(acc[g][1])
                                 : "v"(av[j][g]), "v"(bw[j][0]), "v"(bw[j][1]) : "memory");
                }
            } else if constexpr (TW == 3) {
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
#ifdef B2M_STAMPS
            if constexpr (WPB == 1 && !F16 && HL == 1 && DBG == 0) {
            B2M_STAMP(fs_t0);
            if (fs_vis == 0 && fs_loop == 0 && fs_pro == 0) fs_pro = fs_t0;
            }
#endif
            int c0 = cb;
            do {                              // (bottom-tested, ce - cb >= D: with a zero-trip path hipcc cannot count the loads behind the
                                              // pair-list fetch and drains the whole queue -- s_waitcnt vmcnt(0) -- at every offset advance)
                // the D prefetches of this round target one offset: the current one, or -- in its last round -- the next
                const bool wrap = c0 + D >= ce;
                const int kT = wrap ? kNc : kC;
                const int cT = wrap ? cb : c0 + D;
                uint32_t wT[NG];
#pragma unroll
                for (int g = 0; g < NG; ++g) wT[g] = wrap ? wN[g] : wC[g];
                const int GT = wrap ? GN : GC;                              // row groups of the offset the prefetches target
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    // step in buffer j: the MFMAs of row group g, then at once the gather that refills the group's
                    // operand registers for step + D (the matrix pipe still works off the queued MFMAs meanwhile);
                    // absent row groups skip their MFMAs only -- every step issues NG + TW loads
                    uint32_t ld4;
                    const char* src = src_of(cT + j, ld4);
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        if constexpr (!(DBG & 8)) {
                            if (g < GC) {                                   // wave-uniform
                                // hand-issued loads: the operands of this step are complete when all but the loads issued
                                // since -- the other D - 1 buffers' NG + TW, and this step's g refills -- have landed (the
                                // compiler-tracked pair-list loads in between only make this wait a little early)
                                if constexpr (HL) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((D - 1) * (NG + TW) + g) : "memory");
                                mfma_group(j, g);
                            }
                        }
                        if constexpr (!(DBG & 2)) gather(j, g, src, ld4, wT[g], g < GT);
                    }
                    if constexpr (!(DBG & 4)) weights(j, kT, cT + j);
                }
                c0 += D;
            } while (c0 < ce);
#ifdef B2M_STAMPS
            if constexpr (WPB == 1 && !F16 && HL == 1 && DBG == 0) {
            B2M_STAMP(fs_t1); fs_loop += fs_t1 - fs_t0;
            }
#endif
            // ---- add the offset's result into the strip: lane (i,q) holds channels 16t + 4q .. +3 of pair 16g + i;
            // padded pairs (output row 64) take no part.  Not between two full visits: the next one adds into the same tiles
            const bool chained = WPB == 1 && kC < 128 && kN >= 0 && kN < 128;
            if (!chained) {
            asm volatile("s_nop 15" ::: "memory");        // MFMA result -> VALU read: >= 11 wait states (8-pass MFMA)
            // scatter form with 32-column strips (registers to spare), accumulate: the old values of ALL row groups are requested
            // before the first store -- one memory round trip per visit instead of one per row group
            constexpr bool UPB = UP && TW == 2;
            f32x4 oldall[UPB ? NG : 1][TW];
            if constexpr (UPB) {
                if (a.accumulate) {
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        const uint32_t frow = wC[g] & 0xFFFFFFu;
                        if (g < GC && frow != 0xFFFFFFu) {
#pragma unroll
                            for (int t = 0; t < TW; ++t)
                                if (col0 + 16 * t + 4 * q + 3 < a.cout)
                                    oldall[g][t] = *(const f32x4*)(a.y + (int64_t)frow * a.ldy + col0 + 4 * q + 16 * t);
                        }
                    }
                }
            }
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g < GC) {
                    if constexpr (UP) {
                        const uint32_t frow = wC[g] & 0xFFFFFFu;            // the fine row this pair's result belongs to
                        if (frow != 0xFFFFFFu) {
                            float* dst0 = a.y + (int64_t)frow * a.ldy + col0 + 4 * q;
                            // accumulate: the row has ONE writer in this launch -- a plain read-modify-write (fp32 atomics, one per
                            // element, ran the 32-channel level-0 data gradient at 5 TFLOP/s: 38 M atomics per launch)
                            f32x4 old[TW];
                            if (a.accumulate) {
#pragma unroll
                                for (int t = 0; t < TW; ++t) {
                                    if constexpr (UPB) old[t] = oldall[g][t];
                                    else if (col0 + 16 * t + 4 * q + 3 < a.cout) old[t] = *(const f32x4*)(dst0 + 16 * t);
                                }
                            }
#pragma unroll
                            for (int t = 0; t < TW; ++t) {
                                const int col = col0 + 16 * t + 4 * q;
                                if (col + 3 < a.cout) {
                                    f32x4 v = acc[g][t];
                                    if (a.bias) v += *(const f32x4*)(a.bias + col);
                                    if (a.accumulate) v += old[t];
                                    else if (a.ep_scale) v = conv_epilogue(a, v, (int64_t)frow, col);
                                    *(f32x4*)(dst0 + 16 * t) = v;
                                }
                            }
                        }
                    } else if constexpr (!(DBG & 1)) {
                        const uint32_t orow = wC[g] >> 24;
                        if (orow < B2M_TILE) {
                            float* rowp = Cs + orow * PITCH + 4 * q;
                            f32x4 old[TW];
#pragma unroll
                            for (int t = 0; t < TW; ++t) old[t] = *(const f32x4*)(rowp + 16 * t);
#pragma unroll
                            for (int t = 0; t < TW; ++t) *(f32x4*)(rowp + 16 * t) = old[t] + acc[g][t];
                        }
                    }
#pragma unroll
                    for (int t = 0; t < TW; ++t) acc[g][t] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            }
#ifdef B2M_STAMPS
            if constexpr (WPB == 1 && !F16 && HL == 1 && DBG == 0) {
            B2M_STAMP(fs_t2); fs_flush += fs_t2 - fs_t1; fs_vis += 1;
            }
#endif
            if (kN < 0) break;
            // ---- advance: next offset becomes current; the list fetched an offset ago becomes next; fetch one more
            kC = kN;
#pragma unroll
            for (int g = 0; g < NG; ++g) wC[g] = wN[g];
            kN = kNN; kNc = kN < 0 ? kC : kN;
            GC = GN; GN = groups_of(kNc);
            if constexpr (HL) {
                // the statement that waits for the list is the only reader of the two loaded registers (as in/out operands of a
                // bare wait hipcc may copy them IN FRONT of it: conv_wgrad_flow_kernel, round 4)
                int li, lo;
                asm volatile("s_waitcnt vmcnt(%4)\n\tv_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(li), "=&v"(lo) : "v"(rawi), "v"(rawo),
                             "n"(D * (NG + TW)) : "memory");
                list_words(li, lo, wN);
            } else list_words(rawi, rawo, wN);
            kNN = kN < 0 ? -1 : next_active(kN);
            if constexpr (HL) list_load_hl(kNN < 0 ? kNc : kNN, rawi, rawo);
            else list_load(kNN < 0 ? kNc : kNN, rawi, rawo);
#ifdef B2M_STAMPS
            if constexpr (WPB == 1 && !F16 && HL == 1 && DBG == 0) {
            { unsigned long long t3; B2M_STAMP(t3); fs_adv += t3 - fs_t2; }
            }
#endif
        }
        // hand-issued loads: the last round's prefetches are still in flight and the compiler, which cannot see them, is about
        // to reuse their destination registers (addresses of the write-out!): drain them first
        if constexpr (HL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
    }

#ifdef B2M_STAMPS
            if constexpr (WPB == 1 && !F16 && HL == 1 && DBG == 0) {
    B2M_STAMP(fs_walk);
            }
#endif
    if constexpr (UP) return;                 // (everything is in Y already)
    // ---- write the strip (rows of the strip are 16-byte aligned: coalesced vector stores)
    if constexpr (WPB > 1) {
        __syncthreads();
        if (wave != 0) { B2M_RES_END((int64_t)blockIdx.x * WPB + wave, 0); return; }
        const bool plain = a.nslice == WPB && !a.accumulate;
        // (round 6) accumulate with exactly one workgroup per (tile, strip): this wave is the only writer of the strip's rows, so
        // the sum onto the tensor already there is a plain 16-byte read-modify-write, not 4 atomics per lane
        const bool rmw = a.nslice == WPB && a.accumulate && !F16;
        for (int e = lane; e < B2M_TILE * (SW / 4); e += 64) {
            const int row = e / (SW / 4), c4 = (e % (SW / 4)) * 4;
            const int64_t grow = row0 + row;
            if (grow >= a.n_out) continue;
            f32x4 v = *(const f32x4*)&smem[row * PITCH + c4];
#pragma unroll
            for (int w = 1; w < WPB; ++w) v += *(const f32x4*)&smem[w * STRIP + row * PITCH + c4];
            if (a.stats) *(f32x4*)&smem[row * PITCH + c4] = v;  // keep the combined strip for the column sums below
            const int col = col0 + c4;
            if constexpr (F16) {
                if (col + 3 < a.cout) store_half4(a, v, grow, col);
                continue;
            }
            float* dst = a.y + grow * a.ldy + col;
            if (plain && a.vec_store && col + 3 < a.cout) {
                if (a.ep_scale) v = conv_epilogue(a, v, grow, col);
                *(f32x4*)dst = v;
            } else if (rmw && a.vec_store && col + 3 < a.cout) {
                *(f32x4*)dst = *(const f32x4*)dst + v;
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (col + u < a.cout) {
                        if (plain) dst[u] = v[u];
                        else if (v[u] != 0.f) atomicAdd(dst + u, v[u]);
                    }
                }
            }
        }
        if (a.stats) {
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            tile_column_sums(a, smem, tile, row0, col0, lane);
        }
        B2M_RES_END((int64_t)blockIdx.x * WPB, 0);
        return;
    }
    if (a.stats) tile_column_sums(a, Cs, tile, row0, col0, lane);
#ifdef B2M_STAMPS
            if constexpr (WPB == 1 && !F16 && HL == 1 && DBG == 0) {
    B2M_STAMP(fs_stat);
            }
#endif
    for (int e = lane; e < B2M_TILE * (SW / 4); e += 64) {
        const int row = e / (SW / 4), c4 = (e % (SW / 4)) * 4;
        const int64_t grow = row0 + row;
        if (grow >= a.n_out) continue;
        f32x4 v = *(const f32x4*)&Cs[row * PITCH + c4];
        const int col = col0 + c4;
        if constexpr (F16) {
            if (col + 3 < a.cout) store_half4(a, v, grow, col);
            continue;
        }
        float* dst = a.y + grow * a.ldy + col;
        if (a.vec_store && col + 3 < a.cout) {
            if (a.ep_scale) v = conv_epilogue(a, v, grow, col);
            *(f32x4*)dst = v;
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) if (col + u < a.cout) dst[u] = v[u];
        }
    }
#ifdef B2M_STAMPS
    if constexpr (WPB == 1 && !F16 && HL == 1 && DBG == 0) {
        unsigned long long fs_end; B2M_STAMP(fs_end);
        if (lane == 0) {
            atomicAdd(&g_stamps[0], fs_pro - fs_begin); atomicAdd(&g_stamps[1], fs_loop); atomicAdd(&g_stamps[2], fs_flush);
            atomicAdd(&g_stamps[3], fs_adv); atomicAdd(&g_stamps[4], fs_end - fs_walk); atomicAdd(&g_stamps[5], fs_end - fs_begin);
            atomicAdd(&g_stamps[6], fs_vis); atomicAdd(&g_stamps[7], 1ull); atomicAdd(&g_stamps[8], fs_stat - fs_walk);
            atomicAdd(&g_stamps[9], fs_init - fs_begin);
        }
    }
#endif
    B2M_CLOCK_END(0);
    B2M_RES_END((int64_t)blockIdx.x * WPB + wave, 0);
}
#undef tile_column_sums
#undef B2M_BV
