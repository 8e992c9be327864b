// Sparse convolution on the tile rulebook: forward / data-gradient (one kernel) and weight gradient.
// fp32 in, fp32 accumulate on v_mfma_f32_16x16x4_f32 (exact fp32, bitwise an fmaf chain).
//
// Forward work decomposition: an ITEM is (tile of 64 output rows, strip of 32 or 48 output channels)
// and belongs to ONE wave; the four waves of a workgroup take four consecutive items so that their gathers
// share L1.  A wave walks the tile's active kernel offsets; for every offset the valid (in,out) pairs are
// already compacted (rulebook), so the MFMA row groups are dense: up to 4 groups of 16 pairs.  A comes
// straight from global memory (each lane owns one gathered row, 4 consecutive channels), B (weights) from the
// packed fragment-order image, the per-offset result accumulates in registers over all input-channel chunks and is
// then added into the wave-private output strip in LDS with plain, batched read-modify-write (LDS float atomics
// measured ~200 cycles per wave instruction: 5 TFLOP/s; never use them here).  No workgroup barrier in the main
// path; latency is covered by occupancy (8-12 KiB LDS, <=128 / <=168 VGPRs: 4 / 3 waves per SIMD) and by one or two
// chunks of loads in flight.  Small maps are split over slices (offsets, for tiny maps also channel chunks) whose
// 4 waves per workgroup combine in LDS behind ONE barrier at the end.
#include "b2m_common.h"
#include <stdlib.h>
#include <type_traits>

struct ConvArgs {
    const float* x1; int64_t ldx1; int c1;
    const float* x2; int64_t ldx2; int c2;
    const float* wp; int K;
    const float* bias;
    const int32_t* rb_in; const uint8_t* rb_out; const int32_t* rb_cnt;
    int64_t n_out, ntiles;
    float* y; int64_t ldy; int cout; int accumulate; int nstrips; int vec_store;
    int nslice;      // >1: the tile's active offsets are dealt to nslice waves which add their strips atomically
    int chain;       // conv_fwd_flow_kernel, un-split maps: full visits first, chained in registers (conv_fwd_flow.h)
    int fast32;      // rows < 2^24, pitches < 2^22 floats, tensors < 4 GiB: 24-bit multiply + 32-bit byte offsets
    const float* zeros;   // address of g_zeros passed as data (a select of addresses, not a branch around the load)
    int ncs;         // chunk slices: a slice is (offset slice, part ncs of the input-channel chunks); nslice % ncs == 0
    int wg_combine;  // nslice % 4 == 0: the 4 waves of a workgroup are 4 slices of one item and add up in LDS first
    int64_t nwg;     // workgroups of work; the grid is padded, see xcd_order()
    int64_t xcd_per; // > 0: XCD-aware order in chunks of this many workgroups, see wg_index()
    const int32_t* xcd_start;  // != NULL: work-balanced XCD runs, first tile of XCD 0..7 and ntiles (wg_index_balanced)
    int64_t wg_per_tile;       //          workgroups per tile in that order
    const int32_t* tile_order; //          tile worked on at position j of that order (heavy tiles of a run's end first)
    double* stats;   // != NULL: per-tile column sums of the finished output, [tile][2][cout] (sum, sum of squares): the
                     // BatchNorm statistics of the following layer without another pass over Y (conv_fwd_flow_kernel only)
    // Inference epilogue (b2m_conv_fwd_affine): the strip is written as  [relu]( fmaf(Y, ep_scale[col], ep_shift[col]) [+ ep_res] )
    // -- the eval-mode BatchNorm (+ residual) (+ ReLU) that follows every trunk convolution, in exactly b2m_bn_apply's
    // arithmetic, without its launch and without the round trip of Y through HBM.  16-byte column groups only.
    const float* ep_scale; const float* ep_shift; const float* ep_res; int64_t ld_res; int ep_relu;
    int img_wide;    // F16 kernels with 32-column strips reading an image packed for 64-column strips: strip s is the half s & 1 of
                     // the image's strip s >> 1 (the strip width of the IMAGE follows the channel count, that of the LAUNCH the map's size)
};

// loads of out-of-range operands are redirected here (pointer select, no select on the loaded value)
__device__ __attribute__((aligned(16))) float g_zeros[64];

// the inference epilogue on four consecutive columns of output row `grow` (ConvArgs::ep_scale != NULL; col + 3 < cout)
__device__ __forceinline__ f32x4 conv_epilogue(const ConvArgs& a, f32x4 v, int64_t grow, int col) {
    const f32x4 s = *(const f32x4*)(a.ep_scale + col), b = *(const f32x4*)(a.ep_shift + col);
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = __builtin_fmaf(v[u], s[u], b[u]);
    if (a.ep_res) v += *(const f32x4*)(a.ep_res + grow * a.ld_res + col);
    if (a.ep_relu) {
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = v[u] > 0.f ? v[u] : 0.f;
    }
    return v;
}

// Diagnostic build only (-DB2M_STAMPS, tools/stamps.py): s_memtime stamps per phase of the offset walk, summed
// over all waves.  Not compiled into the shipped library.
#ifdef B2M_STAMPS
__device__ unsigned long long g_stamps[12];
#define B2M_STAMP(v)                                 \
    do {                                             \
        __builtin_amdgcn_sched_barrier(0);           \
        v = __builtin_amdgcn_s_memtime();            \
        __builtin_amdgcn_sched_barrier(0);           \
    } while (0)
extern "C" int b2m_debug_stamps(unsigned long long* out8, int reset) {   // out8: 12 counters
    if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_stamps), sizeof(g_stamps)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#else
#define B2M_STAMP(v) do { } while (0)
#endif
// Diagnostic build only (-DB2M_CLOCKS, tools/clocks.py): the shader clock a kernel's waves see -- delta s_memtime (shader
// cycles) over delta s_memrealtime (100 MHz) from the first to the last instruction of every wave, summed per kernel.
// Two stamps per wave: the timing of the kernel is not disturbed.  [0..2] forward / data gradient, [3..5] weight gradient:
// cycles, 10 ns ticks, waves.
#ifdef B2M_CLOCKS
__device__ unsigned long long g_clocks[8];
extern "C" int b2m_debug_clocks(unsigned long long* out8, int reset) {
    if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_clocks), sizeof(g_clocks)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_clocks), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#define B2M_CLOCK_BEGIN() const unsigned long long ck_c0 = __builtin_amdgcn_s_memtime(), ck_r0 = __builtin_amdgcn_s_memrealtime()
#define B2M_CLOCK_END(slot)                                                                                             \
    do {                                                                                                                \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                     \
        const unsigned long long ck_c1 = __builtin_amdgcn_s_memtime(), ck_r1 = __builtin_amdgcn_s_memrealtime();        \
        if ((threadIdx.x & 63) == 0) {                                                                                  \
            atomicAdd(&g_clocks[slot], ck_c1 - ck_c0); atomicAdd(&g_clocks[slot + 1], ck_r1 - ck_r0);                   \
            atomicAdd(&g_clocks[slot + 2], 1ull);                                                                       \
        }                                                                                                               \
    } while (0)
#else
#define B2M_CLOCK_BEGIN() do { } while (0)
#define B2M_CLOCK_END(slot) do { } while (0)
#endif

// Diagnostic build only (-DB2M_RESIDENCY, tools/residency.py): when every wave of a conv_fwd_flow launch begins and ends
// (s_memrealtime, 100 MHz) -- one slot per wave, no atomics; the host turns the intervals into waves resident over time.
#ifdef B2M_RESIDENCY
#define B2M_RES_CAP (1 << 18)
__device__ unsigned long long g_res[3 * B2M_RES_CAP];
extern "C" int b2m_debug_residency(unsigned long long* host_out, int reset) {      // host_out: 3 * B2M_RES_CAP words (or NULL)
    if (host_out && hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_res), sizeof(g_res)) != hipSuccess) return -1;
    if (reset) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_res)) != hipSuccess || hipMemset(p, 0, sizeof(g_res)) != hipSuccess) return -1;
    }
    return B2M_RES_CAP;
}
// (fp32 gather form only: in the half and scatter variants two more live scalars push the pair-list pointers of the
// hand-issued loads out of the scalar file, which their "s" operands do not allow -- the -DB2M_CLOCKS build has that problem)
#define B2M_RES_BEGIN() unsigned long long rs_t0 = 0; if constexpr (!F16 && !UP && DBG == 0) rs_t0 = __builtin_amdgcn_s_memrealtime()
#define B2M_RES_END(slot_index, meta)                                                                       \
    if constexpr (!F16 && !UP && DBG == 0) {                                                                \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                         \
        const unsigned long long rs_t1 = __builtin_amdgcn_s_memrealtime();                                  \
        const unsigned long long rs_i = (unsigned long long)(slot_index);                                   \
        if ((threadIdx.x & 63) == 0 && rs_i < B2M_RES_CAP) {                                                \
            g_res[3 * rs_i] = rs_t0; g_res[3 * rs_i + 1] = rs_t1; g_res[3 * rs_i + 2] = (unsigned long long)(meta); \
        }                                                                                                   \
    }
#else
#define B2M_RES_BEGIN() do { } while (0)
#define B2M_RES_END(slot_index, meta) do { } while (0)
#endif

// The shader clock the chip holds under fp32-MFMA load (bench.py: before and after the bracketed roofline passes, so that a
// fraction of the 2.4 GHz peak can be read against the clock the lease actually ran at).  Three waves per SIMD run `iters`
// blocks of 12 v_mfma_f32_16x16x4_f32 (the convolution kernels' block); every wave stamps s_memtime (shader cycles) and
// s_memrealtime (100 MHz) around its loop: out[0] += cycles, out[1] += ticks, out[2] += waves.
__global__ __launch_bounds__(256) void clock_probe_kernel(unsigned long long* out, int iters, float seed) {
    f32x4 acc[3];
    float av[4], bv[12];
    for (int i = 0; i < 3; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 4; ++i) av[i] = seed + threadIdx.x * 1e-3f + i;
    for (int i = 0; i < 12; ++i) bv[i] = seed * 0.5f + threadIdx.x * 2e-3f - i;
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_sched_barrier(0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 12; ++i)
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i % 3]) : "v"(av[i / 3]), "v"(bv[i]));
    }
    asm volatile("s_nop 15" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_sched_barrier(0);
    float sink = 0.f;
    for (int i = 0; i < 3; ++i) sink += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&out[0], c1 - c0); atomicAdd(&out[1], r1 - r0); atomicAdd(&out[2], 1ull);
        if (sink == 12345.678f) out[3] = 1ull;                 // (keeps the MFMAs alive)
    }
}
extern "C" int b2m_clock_probe(unsigned long long* out4, int32_t iters, void* stream) {
    B2M_CHECK_ARG(out4 && iters >= 1 && iters <= (1 << 20), "bad arguments");
    hipStream_t st = (hipStream_t)stream;
    B2M_HIP(hipMemsetAsync(out4, 0, 4 * sizeof(unsigned long long), st));
    clock_probe_kernel<<<768, 256, 0, st>>>(out4, iters, 1.0f);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}

// Workgroups are dealt to the 8 XCDs round-robin by their linear id, and every XCD has its own L2.  Consecutive
// tiles are neighbours in space (Morton row order) and gather largely the same input rows, so an XCD should work on
// CONTIGUOUS runs of the work -- with the plain order the same rows were fetched into up to 8 L2s (PMC: L2-miss
// traffic 3.6x the algorithmic bytes of conv_fwd).  The work is cut into chunks of `chunk` workgroups and XCD x takes
// chunks x, x+8, x+16, ...: hardware workgroup b = 8*j + x does work item ((j / chunk) * 8 + x) * chunk + j % chunk.
// Default: one chunk per XCD (contiguous eighths).  Work per tile varies over a scene, so the eighths differ in total
// work (up to 8 % on 4 scenes, 5 % on the 8 scenes of the benchmark); finer chunks (32 tiles) even that out
// but measured 0-2 % SLOWER in the training step -- L2 locality is worth more than the balance
// (profiles/r02_conv_analysis.md).
struct XcdOrder { int64_t chunk; unsigned grid; };
static inline XcdOrder xcd_order(int64_t nwg, int64_t chunk_pref) {
    XcdOrder o;
    if (chunk_pref <= 0 || nwg <= 0) { o.chunk = 0; o.grid = (unsigned)nwg; return o; }
    int64_t chunk = chunk_pref;
    const int64_t eighth = cdiv64(nwg, 8);
    if (chunk > eighth) chunk = eighth;                    // small launches: one chunk per XCD (the round-1 order)
    const int64_t nchunks = cdiv64(nwg, chunk);
    o.chunk = chunk;
    o.grid = (unsigned)(cdiv64(nchunks, 8) * chunk * 8);
    return o;
}
__device__ __forceinline__ int64_t wg_index(int64_t nwg, int64_t chunk) {
    if (chunk <= 0) return blockIdx.x;
    const int64_t x = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int64_t v = ((j / chunk) * 8 + x) * chunk + (j % chunk);
    return v < nwg ? v : -1;
}

// Work-balanced form: XCD x owns the tiles [start[x], start[x+1]) that b2m_rulebook_balance chose (equal WORK, not
// equal tile counts), `per_tile` workgroups each; hardware workgroup 8*j + x is the j-th of XCD x.  The grid covers the
// longest run the balance kernel allows (B2M_XCD_CAP tiles); workgroups past the end of their XCD's run leave.
#define B2M_XCD_CAP(ntiles) (((ntiles) * 5 + 31) / 32)
__device__ __forceinline__ int64_t wg_index_balanced(const int32_t* __restrict__ start, int64_t per_tile) {
    const int x = blockIdx.x & 7;
    const int64_t j = blockIdx.x >> 3;
    const int64_t v = (int64_t)start[x] * per_tile + j;
    return v < (int64_t)start[x + 1] * per_tile ? v : -1;
}

static_assert(B2M_TILE == 64, "conv kernels assume 64-row tiles (4 row groups of 16)");
#define NG 4     // row groups per tile

// LDS strip addressing.  A flush instruction touches 4 rows x 16 columns: with a row stride of 32 floats all rows
// alias to the same banks, so odd rows are XOR-swizzled by 16 columns; a stride of 48 floats alternates by itself.
template <int TW>
__device__ __forceinline__ int cs_index(int row, int col) {
    return TW == 2 ? row * 32 + (col ^ ((row & 1) << 4)) : row * (16 * TW) + col;
}

// Packed weight image (b2m_weight_pack): blocks of 64 lanes x TW*KS floats, ordered [k][strip][chunk]; lane
// (q,i) of a block owns B[chunk*KC + KS*q + s][strip*SW + 16*t + i] as its float f = TW*s + t (SW = 16*TW columns per
// strip, TW = 3 when cout is a multiple of 48, else 2).  Inside a block the floats are stored [f / 4][lane][f % 4]
// when TW*KS is a multiple of 4 -- the u-th 16-byte load of a wave reads ONE contiguous KiB (8 cache lines; stored
// lane-major, every load instruction of a 48-byte-per-lane block touched all 24 lines of the block) -- and lane-major
// [lane][f] otherwise (KC = 8 with 48-column strips: 6 floats per lane).  Zero padded: no predicates.
__host__ __device__ __forceinline__ int pack_pos(int lane, int f, int FPL) {
    return (FPL % 4 == 0) ? (f >> 2) * 256 + lane * 4 + (f & 3) : lane * FPL + f;
}
// inverse: position inside a block -> (lane, f)
__host__ __device__ __forceinline__ void pack_unpos(int pos, int FPL, int& lane, int& f) {
    if (FPL % 4 == 0) { lane = (pos & 255) >> 2; f = (pos >> 8) * 4 + (pos & 3); }
    else { lane = pos / FPL; f = pos % FPL; }
}
template <int KC, bool IDENT, bool ASCALAR, int NPF, int TW>
__global__ __launch_bounds__(256, (NPF <= 2 && TW == 2) ? 4 : 3) void conv_fwd_kernel(ConvArgs a) {
    constexpr int KS = KC / 4;                // k-steps per chunk == floats per lane per gathered row
    constexpr int SW = 16 * TW;               // output channels per strip
    constexpr int LW = 64 * TW * KS;          // floats per packed weight block
    __shared__ float smem[4 * (B2M_TILE + 1) * SW];   // per wave: 64 rows + one spare row for padded pairs
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int64_t wg = wg_index(a.nwg, a.xcd_per);
    if (wg < 0) return;
    const int64_t witem = wg * 4 + wave;
    const int sl = (int)(witem % a.nslice);
    const int slice = sl / a.ncs, cslice = sl % a.ncs;       // offsets are dealt to nslice/ncs slices, chunks to ncs
    const int nkslice = a.nslice / a.ncs;
    const int64_t item = witem / a.nslice;
    const int64_t tile = item / a.nstrips;
    const int strip = (int)(item % a.nstrips);
    if (tile >= a.ntiles) return;             // whole wave leaves; no barriers below
    const int col0 = strip * SW;
    const int cin = a.c1 + a.c2;
    const int nchunk = (cin + KC - 1) / KC;
    const int nch1 = (a.c1 + KC - 1) / KC;    // chunks served by the first source (c1 % KC == 0 when c2 > 0)
    const int64_t ldr = a.ntiles * B2M_TILE;
    const int64_t row0 = tile * B2M_TILE;
    float* Cs = smem + wave * ((B2M_TILE + 1) * SW);
#ifdef B2M_STAMPS
    unsigned long long st_begin, st0, st1, st2, st3, st_idx = 0, st_loop = 0, st_epi = 0, st_noff = 0, st_groups = 0;
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
                float t = (a.bias && sl == 0) ? a.bias[col] : 0.f;
                if (a.accumulate && a.nslice == 1 && grow < a.n_out) t += a.y[grow * a.ldy + col];
                v[u] = t;
            }
        }
        *(f32x4*)&Cs[cs_index<TW>(row, c4)] = v;
    }

#ifdef B2M_STAMPS
    unsigned long long st_a, st_b;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    B2M_STAMP(st_a);
#endif
    // ---- active offsets of this tile (K <= 128): lane k holds the pair count of offset k / k+64
    int cnt0 = 0, cnt1 = 0;
    if (IDENT) {
        int64_t rem = a.n_out - row0;
        if (lane == 0) cnt0 = rem < B2M_TILE ? (int)rem : B2M_TILE;
    } else {
        if (lane < a.K) cnt0 = a.rb_cnt[(int64_t)lane * a.ntiles + tile];
        if (lane + 64 < a.K) cnt1 = a.rb_cnt[(int64_t)(lane + 64) * a.ntiles + tile];
    }

#ifdef B2M_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    B2M_STAMP(st_b);
#endif
    // Walk the kernel offsets with purely scalar control flow: the pair count of offset k is read from lane k
    // (k is wave-uniform), empty offsets are skipped, and with split-K the active offsets are dealt round-robin
    // to the slices by a running phase counter.
    int phase = 0;
    for (int k = 0; k < a.K; ++k) {
        const int n = k < 64 ? __builtin_amdgcn_readlane(cnt0, k) : __builtin_amdgcn_readlane(cnt1, k - 64);
        if (n == 0) continue;
        const bool mine = phase == slice;
        phase = phase + 1 == nkslice ? 0 : phase + 1;
        if (!mine) continue;
        const int G = (n + 15) >> 4;           // 1..4 dense row groups
        B2M_STAMP(st0);
        // pair lists: lane (i,q) gathers input row idx[g] and later flushes the 4 output rows packed in out[g].
        // Rows of padded pairs (idx < 0) are clamped to row 0: an MFMA output row depends only on its own A row,
        // and the flush below skips padded pairs, so whatever they compute is never used.
        int idx[NG]; uint32_t out[NG];
        {
            const int64_t base = (int64_t)k * ldr + row0;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (IDENT) {
                    int64_t r = row0 + 16 * g + i;
                    idx[g] = r < a.n_out ? (int)r : 0;
                    const int p = 16 * g + 4 * q;
                    out[g] = (uint32_t)p | ((uint32_t)(p + 1) << 8) | ((uint32_t)(p + 2) << 16) | ((uint32_t)(p + 3) << 24);
                } else {                       // slots beyond the pair count hold -1 / 0 by construction
                    const int r = a.rb_in[base + 16 * g + i];
                    idx[g] = r < 0 ? 0 : r;
                    out[g] = *(const uint32_t*)(a.rb_out + base + 16 * g + 4 * q);
                }
            }
        }
#ifdef B2M_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        B2M_STAMP(st1);
#endif
        f32x4 acc[NG][TW];
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int t = 0; t < TW; ++t) acc[g][t] = f32x4{0.f, 0.f, 0.f, 0.f};

        // packed weights of (k, strip): wave-uniform base (scalar registers) + a 32-bit lane offset
        const char* wbase = (const char*)(a.wp + ((int64_t)k * a.nstrips + strip) * nchunk * LW);
        constexpr bool WLIN = (TW * KS) % 4 == 0;     // [u][lane][4] block layout, see pack_pos()
        const uint32_t wlo = WLIN ? (uint32_t)lane * 16u : (uint32_t)lane * (TW * KS * 4);
        constexpr uint32_t WU = WLIN ? 1024u : 16u;    // byte distance between a lane's consecutive 16-byte pieces

        // one source tensor: chunks [c_lo, c_hi) of the concatenated input channels.  NPF chunks of loads are
        // issued back to back before the first MFMA block (memory-level parallelism per wave); the loads are
        // unconditional -- a chunk index past the end is clamped and only its MFMAs are skipped -- so that the
        // number of loads in flight is static and the compiler emits counted waits.
        auto run_source = [&](const float* src, int64_t ld, int csrc, int c_base, int c_lo, int c_hi) {
            const float* pa[NG];
            uint32_t bo[NG];               // fast32: byte offset of lane's first channel in its gathered row
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                pa[g] = src + (int64_t)idx[g] * ld + KS * q;
                bo[g] = __umul24((uint32_t)idx[g], (uint32_t)ld * 4u) + (uint32_t)(KS * q * 4);
            }
            for (int c0 = c_lo; c0 < c_hi; c0 += NPF) {
                float bv[NPF][KS][TW], av[NPF][NG][KS];
#pragma unroll
                for (int j = 0; j < NPF; ++j) {
                    const int c = (c0 + j < c_hi) ? c0 + j : c_hi - 1;
                    const int cb = (c - c_base) * KC;      // channel offset inside this source
                    {
                        // TW*KS contiguous floats per lane, ordered [s][t]
                        float wv[TW * KS];
#pragma unroll
                        for (int u = 0; u < TW * KS / 4; ++u) {
                            const f32x4 w4 = *(const f32x4*)(wbase + (size_t)c * (LW * 4) + (wlo + WU * u));
                            wv[4 * u] = w4[0]; wv[4 * u + 1] = w4[1]; wv[4 * u + 2] = w4[2]; wv[4 * u + 3] = w4[3];
                        }
                        if constexpr ((TW * KS) % 4 != 0) {         // TW == 3, KS == 2: 6 floats = 4 + 2
                            const f32x2 w2 = *(const f32x2*)(wbase + (size_t)c * (LW * 4) + (wlo + 16 * (TW * KS / 4)));
                            wv[4 * (TW * KS / 4)] = w2[0]; wv[4 * (TW * KS / 4) + 1] = w2[1];
                        }
#pragma unroll
                        for (int s = 0; s < KS; ++s)
#pragma unroll
                            for (int t = 0; t < TW; ++t) bv[j][s][t] = wv[TW * s + t];
                    }
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        // fast32: wave-uniform base (source + chunk, scalar) + 32-bit per-lane offset -> no vector
                        // address arithmetic per chunk
                        const float* p = (!ASCALAR && a.fast32) ? (const float*)((const char*)(src + cb) + bo[g]) : pa[g] + cb;
                        if constexpr (ASCALAR) {   // odd channel counts (head gradients): per-element, predicated
#pragma unroll
                            for (int s = 0; s < KS; ++s) av[j][g][s] = *((cb + KS * q + s < csrc) ? p + s : a.zeros);
                        } else if constexpr (KS == 4) {
                            const f32x4 v = *(const f32x4*)p;
                            av[j][g][0] = v[0]; av[j][g][1] = v[1]; av[j][g][2] = v[2]; av[j][g][3] = v[3];
                        } else {
                            const f32x2 v = *(const f32x2*)p;
                            av[j][g][0] = v[0]; av[j][g][1] = v[1];
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < NPF; ++j) {
                    if (c0 + j < c_hi) {                    // wave-uniform
                        // k-step outermost: consecutive MFMAs go to G*TW different accumulators.  The pipe needs
                        // >= ~8 independent accumulators in flight to run at peak (tools/micro/mfma_peak.hip: 118
                        // TFLOP/s with 4, 150 with 12); iterating s innermost left only TW = 2..3.
#pragma unroll
                        for (int s = 0; s < KS; ++s) {
#pragma unroll
                            for (int g = 0; g < NG; ++g) {
                                if (g < G) {                // wave-uniform
#pragma unroll
                                    for (int t = 0; t < TW; ++t)
                                        acc[g][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][g][s], bv[j][s][t], acc[g][t], 0, 0, 0);
                                }
                            }
                        }
                    }
                }
            }
        };
        // (tiny maps also split the input-channel chunks: part cslice of ncs of every source)
        run_source(a.x1, a.ldx1, a.c1, 0, nch1 * cslice / a.ncs, nch1 * (cslice + 1) / a.ncs);
        if (a.c2 > 0) run_source(a.x2, a.ldx2, a.c2, nch1, nch1 + (nchunk - nch1) * cslice / a.ncs, nch1 + (nchunk - nch1) * (cslice + 1) / a.ncs);
        B2M_STAMP(st2);

        // ---- add the offset's result into the strip.  D[row = 4q + r][col = i]; the pairs of one offset have
        // distinct output rows, so the read-modify-writes of an offset never alias and can be batched: all reads
        // of FG row groups, then the adds, then the writes (one LDS round trip per batch instead of one per
        // element -- unbatched, the dependent ds_read -> add -> ds_write chains were 15 % of a wave's life).
        // Padded pairs are steered to the spare row B2M_TILE of the strip, so there is no divergence.
        constexpr int FG = 2;
#pragma unroll
        for (int g0 = 0; g0 < NG; g0 += FG) {
            if (g0 < G) {
                int ad[FG][4];
                float old[FG][4][TW];
#pragma unroll
                for (int gg = 0; gg < FG; ++gg) {
                    const int g = g0 + gg;
                    const uint32_t o4 = out[g];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = (16 * g + 4 * q + r < n) ? (int)((o4 >> (8 * r)) & 255) : B2M_TILE;
                        ad[gg][r] = row;
#pragma unroll
                        for (int t = 0; t < TW; ++t) old[gg][r][t] = Cs[cs_index<TW>(row, 16 * t + i)];
                    }
                }
#pragma unroll
                for (int gg = 0; gg < FG; ++gg)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int t = 0; t < TW; ++t)
                            Cs[cs_index<TW>(ad[gg][r], 16 * t + i)] = old[gg][r][t] + acc[g0 + gg][t][r];
            }
        }
#ifdef B2M_STAMPS
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        B2M_STAMP(st3);
        st_idx += st1 - st0; st_loop += st2 - st1; st_epi += st3 - st2; st_noff += 1; st_groups += G;
#endif
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
#ifdef B2M_STAMPS
    unsigned long long st_mid;
    B2M_STAMP(st_mid);
#endif

    // ---- write the strip
    if (a.wg_combine) {
        // split-K on small maps: the workgroup's 4 waves hold 4 slices of the same (tile, strip).  Wave 0 adds the four
        // LDS strips and writes once -- a quarter of the atomics, none at all when there are exactly 4 slices.
        __syncthreads();
        if (wave != 0) return;
        const bool plain = a.nslice == 4 && !a.accumulate;
        for (int e = lane; e < B2M_TILE * (SW / 4); e += 64) {
            const int row = e / (SW / 4), c4 = (e % (SW / 4)) * 4;
            const int64_t grow = row0 + row;
            if (grow >= a.n_out) continue;
            const int ci = cs_index<TW>(row, c4);
            f32x4 v = *(const f32x4*)&smem[ci];
#pragma unroll
            for (int w = 1; w < 4; ++w) v += *(const f32x4*)&smem[w * ((B2M_TILE + 1) * SW) + ci];
            const int col = col0 + c4;
            float* dst = a.y + grow * a.ldy + col;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (col + u < a.cout) {
                    if (plain) dst[u] = v[u];
                    else if (v[u] != 0.f) atomicAdd(dst + u, v[u]);
                }
            }
        }
        return;
    }
    for (int e = lane; e < B2M_TILE * (SW / 4); e += 64) {
        const int row = e / (SW / 4), c4 = (e % (SW / 4)) * 4;
        const int64_t grow = row0 + row;
        if (grow >= a.n_out) continue;
        const f32x4 v = *(const f32x4*)&Cs[cs_index<TW>(row, c4)];
        const int col = col0 + c4;
        float* dst = a.y + grow * a.ldy + col;
        if (a.nslice > 1) {                      // small maps: slices combine with fp32 atomics (Y pre-zeroed)
#pragma unroll
            for (int u = 0; u < 4; ++u) if (col + u < a.cout && v[u] != 0.f) atomicAdd(dst + u, v[u]);
        } else if (a.vec_store && col + 3 < a.cout) {
            *(f32x4*)dst = v;
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) if (col + u < a.cout) dst[u] = v[u];
        }
    }
#ifdef B2M_STAMPS
    unsigned long long st_end;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    B2M_STAMP(st_end);
    if (lane == 0) {
        atomicAdd(&g_stamps[0], st_idx); atomicAdd(&g_stamps[1], st_loop); atomicAdd(&g_stamps[2], st_epi);
        atomicAdd(&g_stamps[3], st_end - st_begin); atomicAdd(&g_stamps[4], st_noff); atomicAdd(&g_stamps[5], 1ull);
        atomicAdd(&g_stamps[6], st_end - st_mid); atomicAdd(&g_stamps[7], st_groups);
        atomicAdd(&g_stamps[8], st_a - st_begin); atomicAdd(&g_stamps[9], st_b - st_a);
    }
#endif
}

// conv_stem_kernel: the network's first layer -- 5x5x5, 6 (padded 8) -> 32 channels on the level-0 map: 125 offsets with
// ~15 pairs each per tile, 4 MFMAs of work per (tile, offset).  In conv_fwd_kernel every offset is a chain of two
// dependent memory round trips (pair list, then the gathered rows) in front of those 4 MFMAs and an LDS update behind
// them: 1.12 ms at 12 TFLOP/s, all of it latency, and as the first kernel of a step nothing runs beside it.  Here the walk
// over the tile's ACTIVE offsets is software-pipelined by hand: while offset k multiplies and updates the strip, the rows of
// the next active offset are in flight and the pair list of the one after is being fetched.
// One wave per tile (cout <= 32: one strip), 4 tiles per workgroup; single source, 8-channel chunk, un-split maps.
template <bool EP>      // EP: the inference epilogue (a variant of its own: at 128 VGPRs the extra pointers spilled in the training kernel)
__global__ __launch_bounds__(256, 4) void conv_stem_kernel(ConvArgs a) {
    constexpr int TW = 2, SW = 32;
    __shared__ float smem[4 * (B2M_TILE + 1) * SW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int64_t wg = wg_index(a.nwg, a.xcd_per);
    if (wg < 0) return;
    const int64_t tile = wg * 4 + wave;
    if (tile >= a.ntiles) return;
    const int64_t ldr = a.ntiles * B2M_TILE, row0 = tile * B2M_TILE;
    float* Cs = smem + wave * ((B2M_TILE + 1) * SW);
    for (int e = lane; e < B2M_TILE * (SW / 4); e += 64) {
        const int row = e / (SW / 4), c4 = (e % (SW / 4)) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const int64_t grow = row0 + row;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int col = c4 + u;
            if (col < a.cout) {
                float t = a.bias ? a.bias[col] : 0.f;
                if (a.accumulate && grow < a.n_out) t += a.y[grow * a.ldy + col];
                v[u] = t;
            }
        }
        *(f32x4*)&Cs[cs_index<TW>(row, c4)] = v;
    }
    int cnt0 = 0, cnt1 = 0;
    if (lane < a.K) cnt0 = a.rb_cnt[(int64_t)lane * a.ntiles + tile];
    if (lane + 64 < a.K) cnt1 = a.rb_cnt[(int64_t)(lane + 64) * a.ntiles + tile];
    const uint64_t m0 = __ballot(cnt0 > 0), m1 = __ballot(cnt1 > 0);
    auto next_active = [&](int k) -> int {
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
    auto count_of = [&](int k) { return k < 64 ? __builtin_amdgcn_readlane(cnt0, k) : __builtin_amdgcn_readlane(cnt1, k - 64); };
    const uint32_t ld4 = (uint32_t)a.ldx1 * 4u;
    // operands of offset k: lane (i, q) gathers channels 2q, 2q + 1 of its row per group; one 16-byte weight piece
    auto load_ops = [&](int k, const int (&idx)[NG], f32x2 (&av)[NG], f32x4& wv) {
        // (all four row groups, always: with the loads of absent groups skipped the number of loads in flight is no longer
        // static, the counted waits turn into waits for everything, and the kernel runs as fast as without any pipeline --
        // measured: 16.4 instead of 25.1 TFLOP/s)
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const uint32_t r = idx[g] < 0 ? 0u : (uint32_t)idx[g];          // padded slots read row 0, their result is never used
            av[g] = *(const f32x2*)((const char*)a.x1 + (__umul24(r, ld4) + 8u * (uint32_t)q));
        }
        wv = *(const f32x4*)((const char*)a.wp + (size_t)k * 1024 + (size_t)lane * 16);   // block (k, strip 0, chunk 0): [lane][s][t]
    };
    // Three active offsets in the pipe: offset t multiplies while the rows of t + 1 have been in flight for one offset and those of
    // t + 2 are issued now; the input-row list of t + 3 goes out with them, its output-row words behind the strip update.  With a
    // single offset of look-ahead one memory latency per offset was the whole run time (0.72 ms; the layer's MFMAs are 0.15 ms).
    // The three operand sets take turns under a period-3 unroll: NO register of an in-flight load is ever moved (a v_mov of a
    // load destination needs vmcnt(0): a first version that rotated the sets by copying ran exactly as fast as no pipeline).
    auto load_idx = [&](int k, int (&idx)[NG]) {
        const int64_t base = (int64_t)k * ldr + row0;
#pragma unroll
        for (int g = 0; g < NG; ++g) idx[g] = a.rb_in[base + 16 * g + i];
    };
    auto load_out = [&](int k, uint32_t (&out)[NG]) {
        const int64_t base = (int64_t)k * ldr + row0;
#pragma unroll
        for (int g = 0; g < NG; ++g) out[g] = *(const uint32_t*)(a.rb_out + base + 16 * g + 4 * q);
    };
    int k0 = next_active(-1);
    if (k0 >= 0) {
        int idx[NG];
        uint32_t out[3][NG];
        f32x2 av[3][NG];
        f32x4 w[3];
        // (a drained queue position repeats the last valid offset: harmless loads, never used)
        int k1 = next_active(k0);
        int k2 = k1 < 0 ? -1 : next_active(k1);
        int k3 = k2 < 0 ? -1 : next_active(k2);
        auto valid = [&](int k, int fallback) { return k < 0 ? fallback : k; };
        {
            const int k1c = valid(k1, k0), k2c = valid(k2, k1c);
            int idx0[NG], idx1[NG];
            load_idx(k0, idx0); load_idx(k1c, idx1); load_idx(k2c, idx);
            load_out(k0, out[0]); load_out(k1c, out[1]); load_out(k2c, out[2]);
            load_ops(k0, idx0, av[0], w[0]);
            load_ops(k1c, idx1, av[1], w[1]);
        }
        bool more = true;
        while (more) {
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                if (k0 < 0) { more = false; break; }            // (wave-uniform)
                const int n = count_of(k0);
                const int G = (n + 15) >> 4;
                const int k2c = valid(k2, valid(k1, k0)), k3c = valid(k3, k2c);
                load_ops(k2c, idx, av[(u + 2) % 3], w[(u + 2) % 3]);       // idx = input rows of offset t + 2 (fetched one offset ago)
                load_idx(k3c, idx);
                f32x4 acc[NG][TW];
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    if (g < G) {                                    // wave-uniform
#pragma unroll
                        for (int t = 0; t < TW; ++t) {
                            f32x4 c = {0.f, 0.f, 0.f, 0.f};
                            c = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][g][0], w[u][t], c, 0, 0, 0);          // w = [s][t]: s = 0
                            acc[g][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][g][1], w[u][2 + t], c, 0, 0, 0);
                        }
                    }
                }
                // add into the strip: D[row = 4q + r][col = i]; the pairs of an offset have distinct output rows; padded pairs go
                // to the spare row B2M_TILE
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    if (g < G) {
                        int ad[4];
                        float old[4][TW];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            ad[r] = (16 * g + 4 * q + r < n) ? (int)((out[u][g] >> (8 * r)) & 255) : B2M_TILE;
#pragma unroll
                            for (int t = 0; t < TW; ++t) old[r][t] = Cs[cs_index<TW>(ad[r], 16 * t + i)];
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r)
#pragma unroll
                            for (int t = 0; t < TW; ++t) Cs[cs_index<TW>(ad[r], 16 * t + i)] = old[r][t] + acc[g][t][r];
                    }
                }
                load_out(k3c, out[u]);                          // (this set's words have just been used)
                k0 = k1; k1 = k2; k2 = k3;
                k3 = k2 < 0 ? -1 : next_active(k2);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
    if (a.stats && lane < a.cout) {
        // column sums of the finished strip for the BatchNorm that follows (fp64, four independent chains), as
        // strip_column_sums does for the flow kernel: bn0 then needs no pass over the 1.2 M x 32 output
        const int64_t rem = a.n_out - row0;
        const int rows = rem < B2M_TILE ? (int)rem : B2M_TILE;
        double s1[4] = {0., 0., 0., 0.}, s2[4] = {0., 0., 0., 0.};
        int r = 0;
        for (; r + 3 < rows; r += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double v = (double)Cs[cs_index<TW>(r + u, lane)];
                s1[u] += v; s2[u] = fma(v, v, s2[u]);
            }
        }
        for (; r < rows; ++r) {
            const double v = (double)Cs[cs_index<TW>(r, lane)];
            s1[0] += v; s2[0] = fma(v, v, s2[0]);
        }
        double* o = a.stats + tile * 2 * a.cout + lane;
        o[0] = (s1[0] + s1[1]) + (s1[2] + s1[3]); o[a.cout] = (s2[0] + s2[1]) + (s2[2] + s2[3]);
    }
    for (int e = lane; e < B2M_TILE * (SW / 4); e += 64) {
        const int row = e / (SW / 4), c4 = (e % (SW / 4)) * 4;
        const int64_t grow = row0 + row;
        if (grow >= a.n_out) continue;
        f32x4 v = *(const f32x4*)&Cs[cs_index<TW>(row, c4)];
        float* dst = a.y + grow * a.ldy + c4;
        if (a.vec_store && c4 + 3 < a.cout) {
            if constexpr (EP) v = conv_epilogue(a, v, grow, c4);
            *(f32x4*)dst = v;
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) if (c4 + u < a.cout) dst[u] = v[u];
        }
    }
}

#include "conv_fwd_flow.h"
#include "conv_1x1.h"

static int env_flag(const char* name, int dflt);
static inline int conv_kc(int cin) { return cin >= 16 ? 16 : 8; }
// strip width in 16-column tiles: 48-column strips for the 96-channel spatial layers (A fragments reused 3x:
// +7 % in the A/B of tools/bench_conv.py), else 32 (1x1 layers measured faster with 32)
static inline int conv_tw(int cout, int K) { return (cout % 48 == 0 && K > 1 && env_flag("B2M_CONV_TW3", 1)) ? 3 : 2; }

extern "C" int64_t b2m_weight_pack_size(int32_t K, int32_t cin, int32_t cout) {
    const int KC = conv_kc(cin);
    const int TW = conv_tw(cout, K);
    const int64_t nchunk = (cin + KC - 1) / KC, nstrip = (cout + 16 * TW - 1) / (16 * TW);
    return (int64_t)K * nstrip * nchunk * (64 * TW * (KC / 4));
}

// logical B[k][ci][co]:  transpose == 0:  w[k][ci][co]          (CI = rows, CO = cols)
//                        transpose == 1:  w[src(k)][sb + co][ci] (CI = cols, CO = sc), src(k) = mirror ? K-1-k : k
__global__ void weight_pack_kernel(const float* __restrict__ w, int64_t ldw, int K, int rows, int cols, int transpose,
                                   int mirror, int sb, int CI, int CO, int KC, int TW, float* __restrict__ wp) {
    const int KS = KC / 4, LW = 64 * TW * KS, SW = 16 * TW;
    const int nchunk = (CI + KC - 1) / KC, nstrip = (CO + SW - 1) / SW;
    const int64_t total = (int64_t)K * nstrip * nchunk * LW;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int lane, f;
        pack_unpos((int)(e % LW), TW * KS, lane, f);
        const int64_t blk = e / LW;
        const int chunk = (int)(blk % nchunk); const int64_t b2 = blk / nchunk;
        const int strip = (int)(b2 % nstrip); const int k = (int)(b2 / nstrip);
        const int s = f / TW, t = f % TW, q = lane >> 4, i = lane & 15;
        const int ci = chunk * KC + KS * q + s, co = strip * SW + 16 * t + i;
        float v = 0.f;
        if (ci < CI && co < CO) {
            if (!transpose) v = w[((int64_t)k * rows + ci) * ldw + co];
            else v = w[((int64_t)(mirror ? K - 1 - k : k) * rows + sb + co) * ldw + ci];
        }
        wp[e] = v;
    }
}
extern "C" int b2m_weight_pack(const float* w, int64_t ldw, int32_t K, int32_t cin, int32_t cout, int32_t transpose,
                               int32_t mirror, int32_t slice_begin, int32_t slice_count, float* wp, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    B2M_CHECK_ARG(w && wp && K >= 1 && cin > 0 && cout > 0 && ldw >= cout, "bad arguments");
    int CI, CO;
    if (!transpose) { CI = cin; CO = cout; }
    else {
        B2M_CHECK_ARG(slice_begin >= 0 && slice_count > 0 && slice_begin + slice_count <= cin, "bad channel slice");
        CI = cout; CO = slice_count;
    }
    const int64_t total = b2m_weight_pack_size(K, CI, CO);
    int64_t grid = (total + 255) / 256;
    if (grid > 65536) grid = 65536;
    weight_pack_kernel<<<(unsigned)grid, 256, 0, st>>>(w, ldw, K, cin, cout, transpose, mirror, slice_begin, CI, CO,
                                                       conv_kc(CI), conv_tw(CO, K), wp);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}

// ---- all layers of a network in one launch: a plan of descriptors (built on the host once, kept on the device)
struct PackDesc {
    const float* w; float* wp;
    int64_t ldw, first_block, total;
    int32_t K, rows, cols, transpose, mirror, sb, CI, CO, KC, TW;
};
extern "C" int32_t b2m_weight_pack_plan_size(void) { return (int32_t)sizeof(PackDesc); }
extern "C" int64_t b2m_weight_pack_plan(int32_t n, const int64_t* w, const int64_t* wp, const int64_t* ldw,
                                        const int32_t* K, const int32_t* cin, const int32_t* cout,
                                        const int32_t* transpose, const int32_t* mirror, const int32_t* slice_begin,
                                        const int32_t* slice_count, void* plan_host) {
    B2M_CHECK_ARG(n >= 0 && (n == 0 || (w && wp && ldw && K && cin && cout && transpose && mirror && slice_begin &&
                                        slice_count && plan_host)), "bad arguments");
    PackDesc* d = (PackDesc*)plan_host;
    int64_t blocks = 0;
    for (int i = 0; i < n; ++i) {
        B2M_CHECK_ARG(w[i] && wp[i] && K[i] >= 1 && cin[i] > 0 && cout[i] > 0 && ldw[i] >= cout[i], "bad layer");
        PackDesc e;
        e.w = (const float*)(uintptr_t)w[i]; e.wp = (float*)(uintptr_t)wp[i]; e.ldw = ldw[i];
        e.K = K[i]; e.rows = cin[i]; e.cols = cout[i]; e.transpose = transpose[i]; e.mirror = mirror[i]; e.sb = slice_begin[i];
        if (!transpose[i]) { e.CI = cin[i]; e.CO = cout[i]; }
        else {
            B2M_CHECK_ARG(slice_begin[i] >= 0 && slice_count[i] > 0 && slice_begin[i] + slice_count[i] <= cin[i], "bad channel slice");
            e.CI = cout[i]; e.CO = slice_count[i];
        }
        e.KC = conv_kc(e.CI); e.TW = conv_tw(e.CO, e.K);
        e.total = b2m_weight_pack_size(e.K, e.CI, e.CO);
        e.first_block = blocks;
        blocks += e.total / (64 * e.TW * (e.KC / 4));          // one workgroup per packed block
        d[i] = e;
    }
    return blocks;
}
// one workgroup per packed block (64 lanes x TW*KS floats = a KC x 16*TW tile of the logical B[k]): the source tile is
// read in whole row segments (64..192 B contiguous) into LDS and written out in fragment order.  Reading it in
// fragment order straight from memory touched a different 64-byte segment with every 4-byte load (1.4 TB/s).
#define PACK_PER_WG 8
// one packed block; KC / TW as compile-time constants (0: read them from the descriptor) -- every index below is a
// division by them, and with runtime divisors the divisions were most of the kernel (1.5 TB/s)
template <int KCc, int TWc>
__device__ __forceinline__ void pack_one_block(const PackDesc& d, int64_t pb, float* tile) {
    const int KC = KCc ? KCc : d.KC, TW = TWc ? TWc : d.TW;
    const int KS = KC / 4, SW = 16 * TW, LW = 64 * TW * KS;
    const int nchunk = (d.CI + KC - 1) / KC, nstrip = (d.CO + SW - 1) / SW;
    const int chunk = (int)(pb % nchunk); const int64_t b2 = pb / nchunk;
    const int strip = (int)(b2 % nstrip); const int k = (int)(b2 / nstrip);
    const int ci0 = chunk * KC, co0 = strip * SW;
    __syncthreads();                                                 // the previous tile has been consumed
    // whole 16-channel chunks of 16-byte aligned rows: 16-byte loads and stores (one each per thread and block)
    const bool vec = KCc == 16 && (d.ldw & 3) == 0 && ((uintptr_t)d.w & 15) == 0 && (d.sb & 3) == 0;
    if (vec) {
        const int j = threadIdx.x;
        if (j < KC * SW / 4) {
            if (!d.transpose) {                                      // rows of B are rows of w: SW contiguous floats
                const int ci_l = j / (SW / 4), co_l = (j % (SW / 4)) * 4;
                const float* src = d.w + ((int64_t)k * d.rows + ci0 + ci_l) * d.ldw + co0 + co_l;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (ci0 + ci_l < d.CI) {
                    if (co0 + co_l + 3 < d.CO) v = *(const f32x4*)src;
                    else for (int u = 0; u < 4; ++u) if (co0 + co_l + u < d.CO) v[u] = src[u];
                }
                *(f32x4*)&tile[ci_l * SW + co_l] = v;
            } else {                                                 // rows of B are columns of w: KC contiguous floats
                const int co_l = j / (KC / 4), ci_l = (j % (KC / 4)) * 4;
                const float* src = d.w + ((int64_t)(d.mirror ? d.K - 1 - k : k) * d.rows + d.sb + co0 + co_l) * d.ldw + ci0 + ci_l;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (co0 + co_l < d.CO) {
                    if (ci0 + ci_l + 3 < d.CI) v = *(const f32x4*)src;
                    else for (int u = 0; u < 4; ++u) if (ci0 + ci_l + u < d.CI) v[u] = src[u];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) tile[(ci_l + u) * SW + co_l] = v[u];
            }
        }
    } else {
        for (int j = threadIdx.x; j < KC * SW; j += 256) {
            int ci_l, co_l;
            if (!d.transpose) { ci_l = j / SW; co_l = j % SW; }
            else { co_l = j / KC; ci_l = j % KC; }
            float v = 0.f;
            if (ci0 + ci_l < d.CI && co0 + co_l < d.CO) {
                if (!d.transpose) v = d.w[((int64_t)k * d.rows + ci0 + ci_l) * d.ldw + co0 + co_l];
                else v = d.w[((int64_t)(d.mirror ? d.K - 1 - k : k) * d.rows + d.sb + co0 + co_l) * d.ldw + ci0 + ci_l];
            }
            tile[ci_l * SW + co_l] = v;
        }
    }
    __syncthreads();
    float* out = d.wp + pb * LW;
    if (KCc == 16) {                                                 // layout [piece u][lane][4]: 4 consecutive floats = one (u, lane)
        const int e4 = threadIdx.x;
        if (e4 < LW / 4) {
            const int lane = e4 & 63, u = e4 >> 6, q = lane >> 4, i = lane & 15;
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = 4 * u + r, s_ = f / TW, t = f % TW;
                v[r] = tile[(KS * q + s_) * SW + 16 * t + i];
            }
            *(f32x4*)&out[4 * e4] = v;
        }
    } else {
        for (int e = threadIdx.x; e < LW; e += 256) {
            int lane, f;
            pack_unpos(e, TW * KS, lane, f);
            const int s_ = f / TW, t = f % TW, q = lane >> 4, i = lane & 15;
            out[e] = tile[(KS * q + s_) * SW + 16 * t + i];
        }
    }
}
__global__ __launch_bounds__(256) void weight_pack_batch_kernel(const PackDesc* __restrict__ plan, int n) {
    __shared__ __attribute__((aligned(16))) float tile[16 * 48];
    const int64_t blk0 = (int64_t)blockIdx.x * PACK_PER_WG;
    int lo = 0, hi = n - 1;                              // last descriptor with first_block <= blk0
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (plan[mid].first_block <= blk0) lo = mid; else hi = mid - 1;
    }
    for (int it = 0; it < PACK_PER_WG; ++it) {
        const int64_t blk = blk0 + it;
        while (lo + 1 < n && plan[lo + 1].first_block <= blk) ++lo;      // crossing into the next layer is rare
        const PackDesc d = plan[lo];
        const int64_t pb = blk - d.first_block;
        if (pb * (64 * d.TW * (d.KC / 4)) >= d.total) return;            // past the last layer
        if (d.KC == 16 && d.TW == 3) pack_one_block<16, 3>(d, pb, tile);  // (workgroup-uniform)
        else if (d.KC == 16 && d.TW == 2) pack_one_block<16, 2>(d, pb, tile);
        else pack_one_block<0, 0>(d, pb, tile);
    }
}
extern "C" int b2m_weight_pack_run(const void* plan_dev, int32_t n, int64_t total_blocks, void* stream) {
    B2M_CHECK_ARG(n >= 0 && total_blocks >= 0 && total_blocks < (1ll << 31) && (n == 0 || plan_dev), "bad arguments");
    if (n == 0 || total_blocks == 0) return B2M_OK;
    weight_pack_batch_kernel<<<(unsigned)cdiv64(total_blocks, PACK_PER_WG), 256, 0, (hipStream_t)stream>>>((const PackDesc*)plan_dev, n);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}

struct ConvEpilogue { const float* scale; const float* shift; const float* res; int64_t ld_res; int relu; };
static int conv_fwd_impl(const float* x1, int64_t ldx1, int32_t c1, const float* x2, int64_t ldx2, int32_t c2,
                         int64_t n_in, const float* wp, int32_t K, const float* bias, const int32_t* rb_in,
                         const uint8_t* rb_out, const int32_t* rb_cnt, int64_t n_out, float* y, int64_t ldy,
                         int32_t cout, int32_t accumulate, double* tile_stats, int32_t* wrote_stats, void* stream,
                         const ConvEpilogue* ep = nullptr, int32_t* fused = nullptr) {
    hipStream_t st = (hipStream_t)stream;
    if (wrote_stats) *wrote_stats = 0;
    if (fused) *fused = 0;
    B2M_CHECK_ARG(x1 && wp && y && c1 > 0 && c2 >= 0 && cout > 0 && K >= 1 && K <= 128, "bad pointers/sizes (K<=128)");
    B2M_CHECK_ARG((rb_in == nullptr) == (rb_out == nullptr) && (rb_in == nullptr) == (rb_cnt == nullptr),
                  "rulebook pointers must be all set or all NULL");
    B2M_CHECK_ARG(rb_in != nullptr || K == 1, "identity rulebook needs K == 1");
    B2M_CHECK_ARG(c2 == 0 || x2 != nullptr, "x2 is NULL");
    B2M_CHECK_ARG(n_in >= 1, "n_in (rows of x1/x2) must be >= 1");
    B2M_CHECK_ARG(ldy >= cout && ldx1 >= c1 && (c2 == 0 || ldx2 >= c2), "leading dimension too small");
    B2M_CHECK_ARG(((uintptr_t)wp % 16) == 0, "packed weights must be 16-byte aligned");
    const int cin = c1 + c2;
    const int KC = conv_kc(cin);
    const int KS = KC / 4;
    B2M_CHECK_ARG(c2 == 0 || c1 % KC == 0, "with two sources c1 must be a multiple of 16");
    // fast path: every chunk is complete and every gathered segment is aligned; otherwise per-element loads
    const bool fast = c1 % KC == 0 && c2 % KC == 0 && ldx1 % KS == 0 && (c2 == 0 || ldx2 % KS == 0) &&
                      ((uintptr_t)x1 % (4 * KS)) == 0 && ((uintptr_t)x2 % (4 * KS)) == 0;
    if (n_out == 0) return B2M_OK;
    ConvArgs a{};
    a.x1 = x1; a.ldx1 = ldx1; a.c1 = c1; a.x2 = x2; a.ldx2 = ldx2; a.c2 = c2;
    a.wp = wp; a.K = K; a.bias = bias;
    a.rb_in = rb_in; a.rb_out = rb_out; a.rb_cnt = rb_cnt;
    a.n_out = n_out; a.ntiles = cdiv64(n_out, B2M_TILE);
    a.y = y; a.ldy = ldy; a.cout = cout; a.accumulate = accumulate;
    a.stats = nullptr; a.chain = 0;
    a.ep_scale = a.ep_shift = a.ep_res = nullptr; a.ld_res = 0; a.ep_relu = 0;
    a.xcd_start = nullptr; a.wg_per_tile = 0; a.tile_order = nullptr;
    const int TW = conv_tw(cout, K);
    a.nstrips = (cout + 16 * TW - 1) / (16 * TW);
    a.vec_store = (ldy % 4 == 0 && ((uintptr_t)y % 16) == 0) ? 1 : 0;
    // the inference epilogue needs whole 16-byte column groups everywhere it touches
    const bool ep_ok = ep && ep->scale && ep->shift && a.vec_store && cout % 4 == 0 && !accumulate && !bias &&
                       ((uintptr_t)ep->scale % 16) == 0 && ((uintptr_t)ep->shift % 16) == 0 &&
                       (!ep->res || (ep->ld_res % 4 == 0 && ep->ld_res >= cout && ((uintptr_t)ep->res % 16) == 0));
    auto use_epilogue = [&]() {
        a.ep_scale = ep->scale; a.ep_shift = ep->shift; a.ep_res = ep->res; a.ld_res = ep->ld_res; a.ep_relu = ep->relu;
        if (fused) *fused = 1;
    };
    // Small maps (deep U-Net levels: a few hundred rows, 256 channels) have too few (tile, strip) items to
    // fill 1024 SIMDs and each item walks K*cin/16 dependent steps: split the offsets over up to 16 waves.
    const int64_t items0 = a.ntiles * a.nstrips;
    int nslice = 1;
    // (more, smaller slices were measured too: 2-6x as many waves lose 0-60 % to the atomic combine)
    // 6144 rather than one wave per SIMD slot (4096): a map with ~4.3k items ran 1.05 rounds of waves at 50 TFLOP/s;
    // as 4 in-LDS-combined slices it runs at 61
    // B2M_DETERMINISTIC=1: never split (the split-K combine of more than 4 slices adds with fp32 atomics)
    // (round 5: layers with one or two 16-channel chunks -- the 32-channel blocks of level 1, 4.5 k items at the benchmark's
    // size -- have so little work per (offset, slice) that the split costs more than the second round of waves: 52 -> 56
    // TFLOP/s un-split; they split only below 4096 items)
    const int64_t target = env_flag("B2M_DETERMINISTIC", 0) ? 0 : env_flag("B2M_CONV_TARGET", cin <= 32 ? 4096 : 6144);
    if (items0 < target && K > 1) {
        nslice = (int)cdiv64(target, items0);
        if (nslice > 16) nslice = 16;
        if (nslice > K) nslice = K;
        if (K >= 4 && nslice > 1) {                    // whole workgroups per item: in-LDS combine of 4 slices
            nslice = (nslice + 3) / 4 * 4;
            if (nslice > K) nslice = K / 4 * 4;
            if (nslice > 16) nslice = 16;
            // At most 4 slices (one workgroup per item: plain stores, no zero-fill, no atomics, and the epilogue can
            // take the BatchNorm column sums) unless the map is tiny: measured equal or faster than 8..16 slices from
            // 8 tiles up (the atomic combine and the memset eat what the extra waves gain), slower below.
            const int cap = env_flag("B2M_CONV_MAXSLICE", a.ntiles >= 8 ? 4 : 16);
            if (nslice > cap) nslice = cap;
        }
    }
    // tiny maps (a few tiles): also split the input-channel chunks, up to 4 ways, so that a wave's dependent chain of
    // chunk steps gets short; the slices still combine in LDS / with atomics
    int ncs = 1;
    const int nchunk_h = (cin + KC - 1) / KC;
    while (nslice > 1 && ncs < 4 && items0 * nslice * ncs < env_flag("B2M_CONV_CHUNK_ITEMS", 2048) && nchunk_h / (ncs * 2) >= 2 &&
           env_flag("B2M_CONV_CHUNKSPLIT", 1)) ncs *= 2;
    nslice *= ncs;
    // (round 6) medium maps that run un-split -- a few rounds of long-lived waves, tools/residency.py: 14-18 % of such a launch is
    // its drain -- as TWO slices per item (workgroups of two waves that combine in LDS): half the wave lifetime, half the drain.
    // B2M_CONV_SPLIT2 = largest item count that takes this form (0: off)
    // -- only where the hand-issued flow kernel runs (the one kernel with a two-wave instantiation)
    const bool fast32_ok = n_in < (1 << 24) && ldx1 < (1 << 22) && ldx2 < (1 << 22) && n_in * ldx1 * 4 < (1ll << 32) &&
                           n_in * ldx2 * 4 < (1ll << 32) && env_flag("B2M_CONV_FAST32", 1);
    bool split2 = false;
    if (nslice == 1 && K >= 8 && rb_in != nullptr && fast && KC == 16 && fast32_ok && (cin / 16) % 2 == 0 && cin >= 32 &&
        env_flag("B2M_CONV_PIPE", 2) && env_flag("B2M_CONV_HANDLOADS", 1) && !env_flag("B2M_PIPE_DBG", 0) &&
        !env_flag("B2M_DETERMINISTIC", 0) && items0 >= target && items0 < env_flag("B2M_CONV_SPLIT2", 0)) {
        nslice = 2;
        split2 = true;
    }
    a.ncs = ncs;
    a.nslice = nslice;
    a.wg_combine = ((nslice > 1 && nslice % 4 == 0 && env_flag("B2M_CONV_WGCOMBINE", 1)) || split2) ? 1 : 0;
    a.fast32 = (n_in < (1 << 24) && ldx1 < (1 << 22) && ldx2 < (1 << 22) && n_in * ldx1 * 4 < (1ll << 32) &&
                n_in * ldx2 * 4 < (1ll << 32) && env_flag("B2M_CONV_FAST32", 1)) ? 1 : 0;
    static const float* zeros_addr = nullptr;
    if (!zeros_addr) B2M_HIP(hipGetSymbolAddress((void**)&zeros_addr, HIP_SYMBOL(g_zeros)));
    a.zeros = zeros_addr;
    if (nslice > 1 && !accumulate && !(a.wg_combine && (nslice == 4 || split2)))
        B2M_HIP(hipMemset2DAsync(y, (size_t)ldy * sizeof(float), 0, (size_t)cout * sizeof(float), (size_t)n_out, st));
    const int64_t items = items0 * nslice;
    a.nwg = cdiv64(items, 4);
    // XCD-aware order: contiguous eighths; a workgroup holds 4 (tile, strip, slice) items
    const int64_t xcd_tiles = env_flag("B2M_XCD", 1) ? (1 << 30) : 0;        // (finer chunks measured equal or slower: one run per XCD)
    const XcdOrder xo = xcd_order(a.nwg, xcd_tiles * a.nstrips * nslice / 4);
    a.xcd_per = xo.chunk;
    const unsigned grid = xo.grid;
    const bool ident = rb_in == nullptr;
    // 1x1 layers with whole 16-channel chunks: the streaming-GEMM kernel (conv_1x1.h), accumulators in registers
    if (ident && fast && KC == 16 && a.fast32 && n_in >= n_out && env_flag("B2M_CONV_1X1", 1)) {
        int spw = a.nstrips % 3 == 0 ? 3 : a.nstrips % 2 == 0 ? 2 : 1;
        if (a.ntiles * (a.nstrips / spw) < 2048) spw = 1;       // few rows (the heads on segments): one wave per strip
        const int64_t g1 = a.ntiles * (a.nstrips / spw);
        B2M_CHECK_ARG(g1 < (1ll << 31), "too many workgroups");
        if (ep_ok) use_epilogue();
        if (spw == 3) conv_1x1_kernel<3><<<(unsigned)g1, 64, 0, st>>>(a);
        else if (spw == 2) conv_1x1_kernel<2><<<(unsigned)g1, 64, 0, st>>>(a);
        else conv_1x1_kernel<1><<<(unsigned)g1, 64, 0, st>>>(a);
        B2M_LAUNCH_CHECK();
        return B2M_OK;
    }
    // Real rulebook, whole 16-channel chunks, 32-bit addressable: the flat-pipeline kernel (conv_fwd_flow.h), two steps
    // in flight (B2M_CONV_PIPE=0: off) -- un-split maps with one wave per workgroup, split maps with the four
    // waves of a workgroup as four slices that combine in LDS.
    {
        const int nc = cin / 16;
        // (round 5: the three-steps-in-flight form of the fp32 kernel -- B2M_CONV_PIPE=3, measured +-0.3 % in rounds 2 and 4 -- is gone)
        const int depth = env_flag("B2M_CONV_PIPE", 2) ? 2 : 0;
        const bool split_ok = nslice == 1 || (a.wg_combine && env_flag("B2M_CONV_FLOW_SPLIT", 1));
        if (depth >= 2 && !ident && fast && KC == 16 && split_ok && a.fast32 && nc % ncs == 0 && (nc / ncs) % depth == 0 &&
            nc / ncs >= depth) {
            const int wpb = nslice == 1 ? 1 : split2 ? 2 : 4;
            B2M_CHECK_ARG(items0 < (1ll << 31), "too many (tile, strip) items");      // (32-bit index arithmetic in the kernel)
            a.chain = (wpb == 1 && env_flag("B2M_CONV_CHAIN", 1)) ? 1 : 0;
            // the workgroup that writes a (tile, strip) sees its final values: un-split maps, or exactly 4 slices
            // combined in LDS and stored plainly
            if (tile_stats && (nslice == 1 || ((nslice == 4 || split2) && !accumulate))) {
                a.stats = tile_stats;
                if (wrote_stats) *wrote_stats = 1;
            }
            // (the workgroup that writes a (tile, strip) holds its final values: un-split, or exactly 4 slices combined in LDS)
            if (ep_ok && (nslice == 1 || nslice == 4 || split2)) use_epilogue();
            a.nwg = cdiv64(items, wpb);
            XcdOrder fo = xcd_order(a.nwg, xcd_tiles * a.nstrips * nslice / wpb);
            a.xcd_per = fo.chunk;
            // XCD runs of equal work (the tail of rb_cnt, b2m_rulebook_balance) instead of equal tile counts
            if (a.ntiles >= B2M_BALANCE_MIN_TILES && xcd_tiles > 0 && (a.nstrips * nslice) % wpb == 0 && env_flag("B2M_XCD_BALANCE", 1)) {
                a.xcd_start = rb_cnt + (int64_t)K * a.ntiles;
                a.tile_order = env_flag("B2M_XCD_ORDER", 1) ? a.xcd_start + 16 + a.ntiles : nullptr;
                a.wg_per_tile = a.nstrips * nslice / wpb;
                fo.grid = (unsigned)(8 * B2M_XCD_CAP(a.ntiles) * a.wg_per_tile);
            }
            const int dbg = env_flag("B2M_PIPE_DBG", 0);      // diagnostic builds, wrong results: tools/pipe_breakdown.py
            const int hl = env_flag("B2M_CONV_HANDLOADS", 1);     // hand-issued operand loads, absent row groups masked (conv_fwd_flow.h)
            if (hl && depth == 2 && dbg == 32 && wpb == 1 && TW == 3) {       // diagnostic: the walk twice per wave
                conv_fwd_flow_kernel<2, 3, 32, 1, 1><<<fo.grid, 64, 0, st>>>(a);
                B2M_LAUNCH_CHECK();
                return B2M_OK;
            }
            if (hl && depth == 2 && dbg == 8 && wpb == 1 && TW == 3) {        // diagnostic: every load, flush and list step, no MFMA
                conv_fwd_flow_kernel<2, 3, 8, 1, 1><<<fo.grid, 64, 0, st>>>(a);    // (hand-issued loads cannot be optimised away)
                B2M_LAUNCH_CHECK();
                return B2M_OK;
            }
            if (hl && depth == 2 && !dbg) {
                if (wpb == 2) {
                    if (TW == 3) conv_fwd_flow_kernel<2, 3, 0, 2, 1><<<fo.grid, 128, 0, st>>>(a);
                    else conv_fwd_flow_kernel<2, 2, 0, 2, 1><<<fo.grid, 128, 0, st>>>(a);
                } else if (wpb == 4) {
                    if (TW == 3) conv_fwd_flow_kernel<2, 3, 0, 4, 1><<<fo.grid, 256, 0, st>>>(a);
                    else conv_fwd_flow_kernel<2, 2, 0, 4, 1><<<fo.grid, 256, 0, st>>>(a);
                } else {
                    if (TW == 3) conv_fwd_flow_kernel<2, 3, 0, 1, 1><<<fo.grid, 64, 0, st>>>(a);
                    else conv_fwd_flow_kernel<2, 2, 0, 1, 1><<<fo.grid, 64, 0, st>>>(a);
                }
                B2M_LAUNCH_CHECK();
                return B2M_OK;
            }
            if (wpb == 4) {
                if (TW == 3) conv_fwd_flow_kernel<2, 3, 0, 4><<<fo.grid, 256, 0, st>>>(a);
                else conv_fwd_flow_kernel<2, 2, 0, 4><<<fo.grid, 256, 0, st>>>(a);
            } else if (dbg && TW == 3) {          // diagnostic builds of tools/pipe_breakdown.py (wrong results)
                switch (dbg) {
                    case 1: conv_fwd_flow_kernel<2, 3, 1><<<fo.grid, 64, 0, st>>>(a); break;
                    case 2: conv_fwd_flow_kernel<2, 3, 2><<<fo.grid, 64, 0, st>>>(a); break;
                    case 4: conv_fwd_flow_kernel<2, 3, 4><<<fo.grid, 64, 0, st>>>(a); break;
                    case 8: conv_fwd_flow_kernel<2, 3, 8><<<fo.grid, 64, 0, st>>>(a); break;
                    default: conv_fwd_flow_kernel<2, 3, 6><<<fo.grid, 64, 0, st>>>(a); break;
                }
            } else {
                if (TW == 3) conv_fwd_flow_kernel<2, 3><<<fo.grid, 64, 0, st>>>(a);
                else conv_fwd_flow_kernel<2, 2><<<fo.grid, 64, 0, st>>>(a);
            }
            B2M_LAUNCH_CHECK();
            return B2M_OK;
        }
    }
    // the first layer's shape (one 8-channel chunk, one 32-column strip, un-split): the hand-pipelined walk over the offsets
    if (KC == 8 && !ident && fast && cin == 8 && c2 == 0 && nslice == 1 && a.nstrips == 1 && TW == 2 && a.fast32 &&
        env_flag("B2M_CONV_STEM", 1)) {
        if (tile_stats) {                      // the workgroup that writes a tile sees its final values (un-split)
            a.stats = tile_stats;
            if (wrote_stats) *wrote_stats = 1;
        }
        if (ep_ok) {
            use_epilogue();
            conv_stem_kernel<true><<<grid, 256, 0, st>>>(a);
        } else conv_stem_kernel<false><<<grid, 256, 0, st>>>(a);
        B2M_LAUNCH_CHECK();
        return B2M_OK;
    }
    const int variant = (KC == 16 ? 4 : 0) | (ident ? 2 : 0) | (fast ? 0 : 1);
    // chunks of loads in flight per wave: 2 pays on the 48-column-strip layers with >= 4 chunks (+7 % in the A/B of
    // tools/bench_conv.py), 1 elsewhere (narrow / 1x1 layers lose occupancy with 2)
    const int npf = (TW == 3 && cin >= 64) ? 2 : 1;
#define B2M_CONV_LAUNCH(KCV, ID, AS, NPFV)                                                                 \
    do {                                                                                                   \
        if (TW == 3) conv_fwd_kernel<KCV, ID, AS, NPFV, 3><<<grid, 256, 0, st>>>(a);                       \
        else conv_fwd_kernel<KCV, ID, AS, NPFV, 2><<<grid, 256, 0, st>>>(a);                               \
    } while (0)
#define B2M_CONV_CASE(V, KCV, ID, AS)                                                          \
    case V:                                                                                    \
        if (npf >= 2) B2M_CONV_LAUNCH(KCV, ID, AS, 2);                                         \
        else B2M_CONV_LAUNCH(KCV, ID, AS, 1);                                                  \
        break;
    switch (variant) {
        B2M_CONV_CASE(0, 8, false, false) B2M_CONV_CASE(1, 8, false, true) B2M_CONV_CASE(2, 8, true, false)
        B2M_CONV_CASE(3, 8, true, true) B2M_CONV_CASE(4, 16, false, false) B2M_CONV_CASE(5, 16, false, true)
        B2M_CONV_CASE(6, 16, true, false) B2M_CONV_CASE(7, 16, true, true)
    }
#undef B2M_CONV_CASE
#undef B2M_CONV_LAUNCH
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}

extern "C" int b2m_conv_fwd(const float* x1, int64_t ldx1, int32_t c1, const float* x2, int64_t ldx2, int32_t c2,
                            int64_t n_in, const float* wp, int32_t K, const float* bias, const int32_t* rb_in,
                            const uint8_t* rb_out, const int32_t* rb_cnt, int64_t n_out, float* y, int64_t ldy,
                            int32_t cout, int32_t accumulate, void* stream) {
    return conv_fwd_impl(x1, ldx1, c1, x2, ldx2, c2, n_in, wp, K, bias, rb_in, rb_out, rb_cnt, n_out, y, ldy, cout, accumulate,
                         nullptr, nullptr, stream);
}
extern "C" int b2m_conv_fwd_affine(const float* x1, int64_t ldx1, int32_t c1, const float* x2, int64_t ldx2, int32_t c2,
                                   int64_t n_in, const float* wp, int32_t K, const int32_t* rb_in, const uint8_t* rb_out,
                                   const int32_t* rb_cnt, int64_t n_out, float* y, int64_t ldy, int32_t cout,
                                   const float* scale, const float* shift, const float* res, int64_t ld_res, int32_t relu,
                                   int32_t* fused, void* stream) {
    B2M_CHECK_ARG(scale && shift && fused, "scale / shift / fused are NULL");
    const ConvEpilogue ep = {scale, shift, res, ld_res, relu};
    return conv_fwd_impl(x1, ldx1, c1, x2, ldx2, c2, n_in, wp, K, nullptr, rb_in, rb_out, rb_cnt, n_out, y, ldy, cout, 0, nullptr,
                         nullptr, stream, &ep, fused);
}
// Transposed k2s2 convolution (and the data gradient of the strided one) in scatter form: conv_fwd_flow_kernel<.., UP>
// over the map's DOWN rulebook (tiled over the n_coarse input rows; rb_in = fine row, rb_out = coarse row inside the tile).
// *ran = 0: the shape is not one the kernel takes (odd channel counts, a handful of tiles) -- nothing was launched and the
// caller runs b2m_conv_fwd on the UP rulebook instead.
extern "C" int b2m_conv_up(const float* x1, int64_t ldx1, int32_t c1, const float* x2, int64_t ldx2, int32_t c2, int64_t n_coarse,
                           const float* wp, int32_t K, const float* bias, const int32_t* rb_in, const uint8_t* rb_out,
                           const int32_t* rb_cnt, float* y, int64_t ldy, int32_t cout, int64_t n_fine, int32_t accumulate,
                           const float* scale, const float* shift, const float* res, int64_t ld_res, int32_t relu,
                           int32_t* ran, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    B2M_CHECK_ARG(ran, "ran is NULL");
    *ran = 0;
    B2M_CHECK_ARG(x1 && wp && y && rb_in && rb_out && rb_cnt && c1 > 0 && c2 >= 0 && cout > 0 && K >= 1 && K <= 128, "bad pointers/sizes");
    B2M_CHECK_ARG(c2 == 0 || x2 != nullptr, "x2 is NULL");
    B2M_CHECK_ARG(ldy >= cout && ldx1 >= c1 && (c2 == 0 || ldx2 >= c2), "leading dimension too small");
    B2M_CHECK_ARG((scale == nullptr) == (shift == nullptr), "scale and shift come together");
    if (n_coarse <= 0 || n_fine <= 0) { *ran = 1; return B2M_OK; }
    const int cin = c1 + c2;
    const int TW = conv_tw(cout, K);
    ConvArgs a{};
    a.x1 = x1; a.ldx1 = ldx1; a.c1 = c1; a.x2 = x2; a.ldx2 = ldx2; a.c2 = c2;
    a.wp = wp; a.K = K; a.bias = bias;
    a.rb_in = rb_in; a.rb_out = rb_out; a.rb_cnt = rb_cnt;
    a.n_out = n_coarse; a.ntiles = cdiv64(n_coarse, B2M_TILE);
    a.y = y; a.ldy = ldy; a.cout = cout; a.accumulate = accumulate;
    a.stats = nullptr; a.chain = 0;
    a.ep_scale = scale; a.ep_shift = shift; a.ep_res = res; a.ld_res = ld_res; a.ep_relu = relu;
    a.xcd_start = nullptr; a.wg_per_tile = 0; a.tile_order = nullptr;
    a.nstrips = (cout + 16 * TW - 1) / (16 * TW);
    a.vec_store = 1; a.nslice = 1; a.ncs = 1; a.wg_combine = 0; a.zeros = nullptr;
    const int nc = cin / 16;
    const int64_t n_max = n_coarse > n_fine ? n_coarse : n_fine;
    const bool ok = c1 % 16 == 0 && c2 % 16 == 0 && nc >= 2 && nc % 2 == 0 && cout % 4 == 0 && ldy % 4 == 0 && ((uintptr_t)y % 16) == 0 &&
                    ldx1 % 4 == 0 && (c2 == 0 || ldx2 % 4 == 0) && ((uintptr_t)x1 % 16) == 0 && ((uintptr_t)x2 % 16) == 0 &&
                    ((uintptr_t)wp % 16) == 0 && (!bias || ((uintptr_t)bias % 16) == 0) &&
                    (!scale || (!accumulate && !bias && ((uintptr_t)scale % 16) == 0 && ((uintptr_t)shift % 16) == 0 &&
                                (!res || (ld_res % 4 == 0 && ld_res >= cout && ((uintptr_t)res % 16) == 0)))) &&
                    n_max < (1 << 24) && ldx1 < (1 << 22) && ldx2 < (1 << 22) && n_coarse * ldx1 * 4 < (1ll << 32) &&
                    n_coarse * ldx2 * 4 < (1ll << 32) &&
                    a.ntiles * a.nstrips >= env_flag("B2M_CONV_UP_MIN_ITEMS", 450) && !env_flag("B2M_PIPE_DBG", 0) &&
                    env_flag("B2M_CONV_UP", 1);
    if (!ok) return B2M_OK;
    a.fast32 = 1;
    a.nwg = a.ntiles * a.nstrips;
    B2M_CHECK_ARG(a.nwg < (1ll << 31), "too many (tile, strip) items");
    const int64_t xcd_tiles = env_flag("B2M_XCD", 1) ? (1 << 30) : 0;
    XcdOrder fo = xcd_order(a.nwg, xcd_tiles * a.nstrips);
    a.xcd_per = fo.chunk;
    if (a.ntiles >= B2M_BALANCE_MIN_TILES && xcd_tiles > 0 && env_flag("B2M_XCD_BALANCE", 1)) {
        a.xcd_start = rb_cnt + (int64_t)K * a.ntiles;
        a.tile_order = env_flag("B2M_XCD_ORDER", 1) ? a.xcd_start + 16 + a.ntiles : nullptr;
        a.wg_per_tile = a.nstrips;
        fo.grid = (unsigned)(8 * B2M_XCD_CAP(a.ntiles) * a.wg_per_tile);
    }
    if (TW == 3) conv_fwd_flow_kernel<2, 3, 0, 1, 1, 0, 1><<<fo.grid, 64, 0, st>>>(a);
    else conv_fwd_flow_kernel<2, 2, 0, 1, 1, 0, 1><<<fo.grid, 64, 0, st>>>(a);
    B2M_LAUNCH_CHECK();
    *ran = 1;
    return B2M_OK;
}
// ------------------------------------------------------------------ half activations (inference)
// Packed HALF weight image of b2m_conv_fwd_h: blocks [k][strip][chunk] of TW pieces, piece t = [lane][E halfs] with E = 8
// (chunks of CK = 32 input channels, v_mfma_f32_16x16x32_f16) or 4 (CK = 16, v_mfma_f32_16x16x16_f16): lane (i, q) of piece t
// holds W[k][chunk * CK + E * q + j][strip * 16 * TW + 16 * t + i], j = 0 .. E - 1 -- the MFMA's A operand as it sits in the
// registers.  Forward weights only (no transpose / mirror: there is no backward pass in half).
static inline int conv_h_ck(int c1, int c2) {
    // 32-channel chunks when they can be dealt to the steps of a round (2 or 3 in flight), else 16-channel chunks
    if (c1 % 32 == 0 && c2 % 32 == 0) { const int nc = (c1 + c2) / 32; if (nc % 2 == 0 || nc % 3 == 0) return 32; }
    return 16;
}
// strip width of the half IMAGES in 16-column tiles: 64-column strips where the output channels come in 64s (round 6: every strip
// gathers the layer's input rows again -- a 128-channel layer in 32-column strips gathered them four times: level 1 128->128 180 ->
// 259 TFLOP/s, level 2 156 -> 211, level 3 256->256 153 -> 205, level 1 64->64 142 -> 160).  The 64-column kernel holds two waves per
// SIMD, not three or four, and halves the number of work items: on maps of a few dozen tiles it loses (level 4 256->256 111 -> 93),
// so b2m_conv_fwd_h launches the 32-column kernel there -- on the same image (ConvArgs::img_wide).  B2M_CONV_TW4_H=0: as fp32.
static inline int conv_tw_h(int cout, int K) {
    if (K > 1 && cout % 64 == 0 && env_flag("B2M_CONV_TW4_H", 1)) return 4;
    return conv_tw(cout, K);
}
extern "C" int64_t b2m_weight_pack_h_size(int32_t K, int32_t c1, int32_t c2, int32_t cout) {      // in halfs
    const int TW = conv_tw_h(cout, K);
    return (int64_t)K * ((cout + 16 * TW - 1) / (16 * TW)) * ((c1 + c2) / conv_h_ck(c1, c2)) * (64 * TW * (conv_h_ck(c1, c2) / 4));
}
// (source element of logical weight (k, ci, co): w[kk * sk + ci * sci + co * sco], kk = mirror ? K - 1 - k : k -- the plain image
// has sk = cin * ldw, sci = ldw, sco = 1; the image of the TRANSPOSED weights, the data gradient's operand, exchanges sci and sco)
__global__ void weight_pack_h_kernel(const float* __restrict__ w, int64_t sk, int64_t sci, int64_t sco, int mirror, int K, int cin,
                                     int cout, int CK, int TW, _Float16* __restrict__ wp, int64_t total) {
    const int E = CK / 4, SW = 16 * TW, BLK = 64 * TW * E;
    const int nchunk = cin / CK, nstrip = (cout + SW - 1) / SW;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int pos = (int)(e % BLK);
        const int t = pos / (64 * E), lane = (pos / E) % 64, j = pos % E;
        const int64_t blk = e / BLK;
        const int chunk = (int)(blk % nchunk); const int64_t b2 = blk / nchunk;
        const int strip = (int)(b2 % nstrip), k = (int)(b2 / nstrip);
        const int ci = chunk * CK + E * (lane >> 4) + j, co = strip * SW + 16 * t + (lane & 15);
        const int kk = mirror ? K - 1 - k : k;
        wp[e] = (co < cout) ? (_Float16)w[(int64_t)kk * sk + (int64_t)ci * sci + (int64_t)co * sco] : (_Float16)0.f;
    }
}
extern "C" int b2m_weight_pack_h(const float* w, int64_t ldw, int32_t K, int32_t c1, int32_t c2, int32_t cout, void* wp, void* stream) {
    B2M_CHECK_ARG(w && wp && K >= 1 && c1 > 0 && c2 >= 0 && cout > 0 && ldw >= cout, "bad arguments");
    B2M_CHECK_ARG(c1 % 16 == 0 && c2 % 16 == 0, "input channels of both sources must be multiples of 16");
    const int64_t total = b2m_weight_pack_h_size(K, c1, c2, cout);
    int64_t grid = (total + 255) / 256;
    if (grid > 65536) grid = 65536;
    weight_pack_h_kernel<<<(unsigned)grid, 256, 0, (hipStream_t)stream>>>(w, (int64_t)(c1 + c2) * ldw, ldw, 1, 0, K, c1 + c2, cout,
                                                                         conv_h_ck(c1, c2), conv_tw_h(cout, K), (_Float16*)wp, total);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}
// Half image of the TRANSPOSED weights for the data gradient through b2m_conv_fwd_h (half-precision training): the layer's fp32
// weights W (K, cin, cout, contiguous) -> the image of W'[k] = W[mirror ? K-1-k : k][s0 : s0 + sc, :]^T, i.e. of a layer with
// cout input channels and sc output channels (b2m_weight_pack_h_size(K, cout, 0, sc) halfs).
extern "C" int b2m_weight_pack_h_t(const float* w, int32_t K, int32_t cin, int32_t cout, int32_t mirror, int32_t s0, int32_t sc,
                                   void* wp, void* stream) {
    B2M_CHECK_ARG(w && wp && K >= 1 && cin > 0 && cout > 0 && s0 >= 0 && sc > 0 && s0 + sc <= cin, "bad arguments");
    B2M_CHECK_ARG(cout % 16 == 0, "the layer's output channels (the data gradient's input channels) must be a multiple of 16");
    const int64_t total = b2m_weight_pack_h_size(K, cout, 0, sc);
    int64_t grid = (total + 255) / 256;
    if (grid > 65536) grid = 65536;
    weight_pack_h_kernel<<<(unsigned)grid, 256, 0, (hipStream_t)stream>>>(w + (int64_t)s0 * cout, (int64_t)cin * cout, 1, cout, mirror ? 1 : 0,
                                                                         K, cout, sc, conv_h_ck(cout, 0), conv_tw_h(sc, K), (_Float16*)wp, total);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}
// ---- every half image of a training step in ONE launch (half-precision training: a forward image and one or two transposed
// images per layer, ~90 per step): a table of descriptors built on the host once (b2m_weight_pack_h_plan), kept on the device.
struct PackHDesc {
    const float* w; _Float16* wp;
    int64_t sk, sci, sco, total, first_block;
    int32_t mirror, K, cin, cout, CK, TW;
};
#define PACKH_PER_WG 2048          // elements of an image per workgroup
extern "C" int32_t b2m_weight_pack_h_plan_size(void) { return (int32_t)sizeof(PackHDesc); }
extern "C" int64_t b2m_weight_pack_h_plan(int32_t n, const int64_t* w, const int64_t* wp, const int32_t* K, const int32_t* cin,
                                          const int32_t* cout, const int32_t* c1, const int32_t* transposed, const int32_t* mirror,
                                          const int32_t* s0, const int32_t* sc, void* plan_host) {
    B2M_CHECK_ARG(n >= 0 && (n == 0 || (w && wp && K && cin && cout && c1 && transposed && mirror && s0 && sc && plan_host)), "bad arguments");
    PackHDesc* d = (PackHDesc*)plan_host;
    int64_t blocks = 0;
    for (int i = 0; i < n; ++i) {
        B2M_CHECK_ARG(w[i] && wp[i] && K[i] >= 1 && cin[i] > 0 && cout[i] > 0, "bad layer");
        PackHDesc e;
        e.wp = (_Float16*)(uintptr_t)wp[i]; e.K = K[i]; e.sk = (int64_t)cin[i] * cout[i];
        if (!transposed[i]) {          // b2m_weight_pack_h(w, cout, K, c1, cin - c1, cout)
            const int c2 = cin[i] - c1[i];
            B2M_CHECK_ARG(c1[i] > 0 && c2 >= 0 && c1[i] % 16 == 0 && c2 % 16 == 0, "input channels of both sources must be multiples of 16");
            e.w = (const float*)(uintptr_t)w[i]; e.sci = cout[i]; e.sco = 1; e.mirror = 0;
            e.cin = cin[i]; e.cout = cout[i]; e.CK = conv_h_ck(c1[i], c2); e.TW = conv_tw_h(cout[i], K[i]);
            e.total = b2m_weight_pack_h_size(K[i], c1[i], c2, cout[i]);
        } else {                       // b2m_weight_pack_h_t(w, K, cin, cout, mirror, s0, sc)
            B2M_CHECK_ARG(s0[i] >= 0 && sc[i] > 0 && s0[i] + sc[i] <= cin[i] && cout[i] % 16 == 0, "bad channel slice");
            e.w = (const float*)(uintptr_t)w[i] + (int64_t)s0[i] * cout[i]; e.sci = 1; e.sco = cout[i]; e.mirror = mirror[i] ? 1 : 0;
            e.cin = cout[i]; e.cout = sc[i]; e.CK = conv_h_ck(cout[i], 0); e.TW = conv_tw_h(sc[i], K[i]);
            e.total = b2m_weight_pack_h_size(K[i], cout[i], 0, sc[i]);
        }
        e.first_block = blocks;
        blocks += cdiv64(e.total, PACKH_PER_WG);
        d[i] = e;
    }
    return blocks;
}
__global__ __launch_bounds__(256) void weight_pack_h_batch_kernel(const PackHDesc* __restrict__ plan, int n) {
    // the image of this workgroup: the last descriptor whose first block is not behind it
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (plan[mid].first_block <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const PackHDesc d = plan[lo];
    const int E = d.CK / 4, SW = 16 * d.TW, BLK = 64 * d.TW * E;
    const int nchunk = d.cin / d.CK, nstrip = (d.cout + SW - 1) / SW;
    const int64_t e0 = ((int64_t)blockIdx.x - d.first_block) * PACKH_PER_WG;
    for (int u = threadIdx.x; u < PACKH_PER_WG; u += 256) {
        const int64_t e = e0 + u;
        if (e >= d.total) break;
        const int pos = (int)(e % BLK);
        const int t = pos / (64 * E), lane = (pos / E) % 64, j = pos % E;
        const int64_t blk = e / BLK;
        const int chunk = (int)(blk % nchunk); const int64_t b2 = blk / nchunk;
        const int strip = (int)(b2 % nstrip), k = (int)(b2 / nstrip);
        const int ci = chunk * d.CK + E * (lane >> 4) + j, co = strip * SW + 16 * t + (lane & 15);
        const int kk = d.mirror ? d.K - 1 - k : k;
        d.wp[e] = (co < d.cout) ? (_Float16)d.w[(int64_t)kk * d.sk + (int64_t)ci * d.sci + (int64_t)co * d.sco] : (_Float16)0.f;
    }
}
extern "C" int b2m_weight_pack_h_run(const void* plan_dev, int32_t n, int64_t total_blocks, void* stream) {
    B2M_CHECK_ARG(n >= 0 && total_blocks >= 0 && total_blocks < (1ll << 31) && (n == 0 || plan_dev), "bad arguments");
    if (n == 0 || total_blocks == 0) return B2M_OK;
    weight_pack_h_batch_kernel<<<(unsigned)total_blocks, 256, 0, (hipStream_t)stream>>>((const PackHDesc*)plan_dev, n);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}
// Y(half) = [relu]( conv(X1 | X2)(half) * scale + shift [+ res(half)] ) through conv_fwd_flow_kernel<.., F16>: real rulebooks
// only (a 1x1 layer comes with the identity rulebook of its map: b2m_rulebook of K = 1), input channels of both sources in
// multiples of 16, output channels in multiples of 16, 16-byte aligned rows.  scale / shift may be NULL (plain convolution).
static int conv_fwd_h_impl(const void* x1, int64_t ldx1, int32_t c1, const void* x2, int64_t ldx2, int32_t c2, int64_t n_in,
                           const void* wp, int32_t K, const int32_t* rb_in, const uint8_t* rb_out, const int32_t* rb_cnt,
                           int64_t n_out, void* y, int64_t ldy, int32_t cout, const float* scale, const float* shift,
                           const void* res, int64_t ld_res, int32_t relu, double* tile_stats, void* stream);
extern "C" int b2m_conv_fwd_h(const void* x1, int64_t ldx1, int32_t c1, const void* x2, int64_t ldx2, int32_t c2, int64_t n_in,
                              const void* wp, int32_t K, const int32_t* rb_in, const uint8_t* rb_out, const int32_t* rb_cnt,
                              int64_t n_out, void* y, int64_t ldy, int32_t cout, const float* scale, const float* shift,
                              const void* res, int64_t ld_res, int32_t relu, void* stream) {
    return conv_fwd_h_impl(x1, ldx1, c1, x2, ldx2, c2, n_in, wp, K, rb_in, rb_out, rb_cnt, n_out, y, ldy, cout, scale, shift, res, ld_res,
                           relu, nullptr, stream);
}
// The plain half convolution that also leaves the per-tile column sums of its output AS STORED (each value rounded to binary16
// first): tile_stats[tile][0 / 1][cout] = sum / sum of squares over the tile's 64 rows, fp64 -- what the training-mode BatchNorm
// behind the layer would otherwise read the whole output for (b2m_bn_tilestats / b2m_bn_tilestats_finalize take them).
extern "C" int b2m_conv_fwd_h_stats(const void* x1, int64_t ldx1, int32_t c1, const void* x2, int64_t ldx2, int32_t c2, int64_t n_in,
                                    const void* wp, int32_t K, const int32_t* rb_in, const uint8_t* rb_out, const int32_t* rb_cnt,
                                    int64_t n_out, void* y, int64_t ldy, int32_t cout, double* tile_stats, void* stream) {
    B2M_CHECK_ARG(tile_stats != nullptr, "tile_stats is NULL");
    return conv_fwd_h_impl(x1, ldx1, c1, x2, ldx2, c2, n_in, wp, K, rb_in, rb_out, rb_cnt, n_out, y, ldy, cout, nullptr, nullptr, nullptr, 0,
                           0, tile_stats, stream);
}
static int conv_fwd_h_impl(const void* x1, int64_t ldx1, int32_t c1, const void* x2, int64_t ldx2, int32_t c2, int64_t n_in,
                           const void* wp, int32_t K, const int32_t* rb_in, const uint8_t* rb_out, const int32_t* rb_cnt,
                           int64_t n_out, void* y, int64_t ldy, int32_t cout, const float* scale, const float* shift,
                           const void* res, int64_t ld_res, int32_t relu, double* tile_stats, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    B2M_CHECK_ARG(x1 && wp && y && rb_in && rb_out && rb_cnt && c1 > 0 && c2 >= 0 && cout > 0 && K >= 1 && K <= 128, "bad pointers/sizes (K<=128)");
    B2M_CHECK_ARG(c2 == 0 || x2 != nullptr, "x2 is NULL");
    B2M_CHECK_ARG((scale == nullptr) == (shift == nullptr), "scale and shift: both or none");
    B2M_CHECK_ARG(c1 % 16 == 0 && c2 % 16 == 0 && cout % 16 == 0, "channels must be multiples of 16");
    B2M_CHECK_ARG(ldx1 % 8 == 0 && ldx1 >= c1 && (c2 == 0 || (ldx2 % 8 == 0 && ldx2 >= c2)) && ldy % 4 == 0 && ldy >= cout &&
                  (!res || (ld_res % 4 == 0 && ld_res >= cout)), "row pitches: 16-byte multiples for the inputs, 8 for output / residual");
    B2M_CHECK_ARG(((uintptr_t)x1 % 16) == 0 && ((uintptr_t)x2 % 16) == 0 && ((uintptr_t)wp % 16) == 0 && ((uintptr_t)y % 8) == 0 &&
                  ((uintptr_t)res % 8) == 0 && ((uintptr_t)scale % 16) == 0 && ((uintptr_t)shift % 16) == 0, "alignment");
    B2M_CHECK_ARG(n_in >= 1 && n_in < (1 << 24) && ldx1 < (1 << 22) && ldx2 < (1 << 22) && n_in * ldx1 * 2 < (1ll << 32) &&
                  n_in * ldx2 * 2 < (1ll << 32), "inputs must be 32-bit addressable (rows < 2^24, tensors < 4 GiB)");
    if (n_out == 0) return B2M_OK;
    ConvArgs a{};
    a.x1 = (const float*)x1; a.ldx1 = ldx1; a.c1 = c1; a.x2 = (const float*)x2; a.ldx2 = ldx2; a.c2 = c2;
    a.wp = (const float*)wp; a.K = K; a.bias = nullptr;
    a.rb_in = rb_in; a.rb_out = rb_out; a.rb_cnt = rb_cnt;
    a.n_out = n_out; a.ntiles = cdiv64(n_out, B2M_TILE);
    a.y = (float*)y; a.ldy = ldy; a.cout = cout; a.accumulate = 0; a.stats = tile_stats; a.vec_store = 1; a.fast32 = 1;
    a.ep_scale = scale; a.ep_shift = shift; a.ep_res = (const float*)res; a.ld_res = ld_res; a.ep_relu = relu;
    a.xcd_start = nullptr; a.wg_per_tile = 0; a.tile_order = nullptr; a.zeros = nullptr;
    // the image is packed for 64-column strips wherever the output channels come in 64s; a map of fewer than 256 tiles runs its
    // 32-column kernel on that image (two waves per SIMD and half the work items cost more there than the second gather)
    const int TW_img = conv_tw_h(cout, K);
    const int TW = (TW_img == 4 && a.ntiles < env_flag("B2M_CONV_TW4_H_MIN_TILES", 256)) ? 2 : TW_img;
    a.img_wide = (TW_img == 4 && TW == 2) ? 1 : 0;
    a.nstrips = (cout + 16 * TW - 1) / (16 * TW);
    const int CK = conv_h_ck(c1, c2), nc = (c1 + c2) / CK;
    int depth = nc % 2 == 0 ? 2 : 3;
    B2M_CHECK_ARG(CK == 32 || nc % 2 == 0, "input channels: multiples of 32 (or an even number of 16-channel chunks)");
    // small maps: the active offsets of an item dealt to the 4 waves of a workgroup (combined in LDS, plain stores); never more
    const int64_t items0 = a.ntiles * a.nstrips;
    const int nslice = (items0 < env_flag("B2M_CONV_TARGET", 6144) && K >= 4) ? 4 : 1;
    a.nslice = nslice; a.ncs = 1; a.wg_combine = nslice == 4;
    const int wpb = nslice == 1 ? 1 : 4;
    const int64_t items = items0 * nslice;
    a.nwg = cdiv64(items, wpb);
    const int64_t xcd_tiles = env_flag("B2M_XCD", 1) ? (1 << 30) : 0;
    XcdOrder fo = xcd_order(a.nwg, xcd_tiles * a.nstrips * nslice / wpb);
    a.xcd_per = fo.chunk;
    if (a.ntiles >= B2M_BALANCE_MIN_TILES && xcd_tiles > 0 && (a.nstrips * nslice) % wpb == 0 && K > 1 && env_flag("B2M_XCD_BALANCE", 1)) {
        a.xcd_start = rb_cnt + (int64_t)K * a.ntiles;
        a.tile_order = env_flag("B2M_XCD_ORDER", 1) ? a.xcd_start + 16 + a.ntiles : nullptr;
        a.wg_per_tile = a.nstrips * nslice / wpb;
        fo.grid = (unsigned)(8 * B2M_XCD_CAP(a.ntiles) * a.wg_per_tile);
    }
#define B2M_FLOW_H(D_, TW_, WPB_, F_) conv_fwd_flow_kernel<D_, TW_, 0, WPB_, 1, F_><<<fo.grid, 64 * WPB_, 0, st>>>(a)
    if (CK == 16) {
        if (wpb == 4) { if (TW == 4) B2M_FLOW_H(2, 4, 4, 2); else if (TW == 3) B2M_FLOW_H(2, 3, 4, 2); else B2M_FLOW_H(2, 2, 4, 2); }
        else { if (TW == 4) B2M_FLOW_H(2, 4, 1, 2); else if (TW == 3) B2M_FLOW_H(2, 3, 1, 2); else B2M_FLOW_H(2, 2, 1, 2); }
    } else if (depth == 2) {
        if (wpb == 4) { if (TW == 4) B2M_FLOW_H(2, 4, 4, 1); else if (TW == 3) B2M_FLOW_H(2, 3, 4, 1); else B2M_FLOW_H(2, 2, 4, 1); }
        else { if (TW == 4) B2M_FLOW_H(2, 4, 1, 1); else if (TW == 3) B2M_FLOW_H(2, 3, 1, 1); else B2M_FLOW_H(2, 2, 1, 1); }
    } else {
        if (wpb == 4) { if (TW == 4) B2M_FLOW_H(3, 4, 4, 1); else if (TW == 3) B2M_FLOW_H(3, 3, 4, 1); else B2M_FLOW_H(3, 2, 4, 1); }
        else { if (TW == 4) B2M_FLOW_H(3, 4, 1, 1); else if (TW == 3) B2M_FLOW_H(3, 3, 1, 1); else B2M_FLOW_H(3, 2, 1, 1); }
    }
#undef B2M_FLOW_H
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}
extern "C" int b2m_conv_fwd_stats(const float* x1, int64_t ldx1, int32_t c1, const float* x2, int64_t ldx2, int32_t c2,
                                  int64_t n_in, const float* wp, int32_t K, const float* bias, const int32_t* rb_in,
                                  const uint8_t* rb_out, const int32_t* rb_cnt, int64_t n_out, float* y, int64_t ldy,
                                  int32_t cout, int32_t accumulate, double* tile_stats, int32_t* wrote_stats, void* stream) {
    B2M_CHECK_ARG(tile_stats && wrote_stats, "tile_stats / wrote_stats are NULL");
    return conv_fwd_impl(x1, ldx1, c1, x2, ldx2, c2, n_in, wp, K, bias, rb_in, rb_out, rb_cnt, n_out, y, ldy, cout, accumulate,
                         tile_stats, wrote_stats, stream);
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
    const float* zeros;      // address of g_zeros passed as data: a select of ADDRESSES, not a branch around the load
    int fast32;              // complete blocks and 24/32-bit addressable tensors: cheap address arithmetic
    int64_t nwg, xcd_per;    // XCD-aware workgroup order (wg_index); work item = (k fastest, block group, tile chunk)
    int nz;                  // block groups of 4 (ci,co) blocks
    int pipe;                // software-pipelined kernel (real rulebook, fast32)
    float* partial;          // deterministic mode: per tile-chunk partial dW (dense [chunk][K][cin][cout]), plain stores
    const int32_t* xcd_start; // != NULL: work-balanced XCD runs of tiles (b2m_rulebook_balance); a run is cut into tile
                              // chunks of tiles_per_chunk from ITS first tile
    int kpack;                // offsets per workgroup: 1, or -- layers with only 1 or 2 (ci, co) blocks -- 4 or 2: the waves
                              // a single block would leave idle take the neighbouring offsets of the same tile chunk
    int kgroups;              // ceil(K / kpack)
    int handloads;            // conv_wgrad_flow_kernel<.., HL = 1> (B2M_WGRAD_HANDLOADS)
    int swap;                 // b2m_conv_wgrad_tr: x is indexed by the tile's own rows (row0 + rb_out), dy by rb_in
    int half;                 // b2m_conv_wgrad_h: x and dy are IEEE binary16 (pitches in elements), converted on load; fp32 MFMA, fp32 dW
    float out_scale;          // ... and the block is multiplied by this on its way into dW (1 / loss scale)
    int trh;                  // ... on the f16 MFMA through the transposing LDS read (conv_wgrad_trh_kernel)
};
// work item of a wave -> (offset k, block blk, tile range [t0, t1)); false: nothing to do.  A workgroup is (offset group,
// block group, tile chunk); its 4 waves are 4 blocks of one offset (kpack = 1) or 4 / kpack blocks of kpack consecutive
// offsets.  `chunk` numbers the tile chunks of the plain order (it addresses the deterministic mode's partial buffer,
// which never uses the balanced order).
__device__ __forceinline__ bool wgrad_item(const WgradArgs& a, int wave, int& k, int& blk, int64_t& chunk, int64_t& t0,
                                           int64_t& t1) {
    int64_t j;
    const int32_t* run = nullptr;
    if (a.xcd_start) {
        run = a.xcd_start + (blockIdx.x & 7);
        j = blockIdx.x >> 3;
    } else {
        j = wg_index(a.nwg, a.xcd_per);
        if (j < 0) return false;
    }
    const int kg = (int)(j % a.kgroups);
    const int64_t rest = j / a.kgroups;
    const int zg = (int)(rest % a.nz);
    chunk = rest / a.nz;
    if (a.kpack == 1) { k = kg; blk = zg * 4 + wave; }
    else { const int nb = 4 / a.kpack; k = kg * a.kpack + wave / nb; blk = wave % nb; }
    if (k >= a.K || blk >= a.nmb * a.nnb) return false;
    if (run) {
        const int64_t s1 = run[1];
        t0 = run[0] + chunk * a.tiles_per_chunk;
        if (t0 >= s1) return false;
        t1 = t0 + a.tiles_per_chunk < s1 ? t0 + a.tiles_per_chunk : s1;
    } else {
        t0 = chunk * a.tiles_per_chunk;
        t1 = t0 + a.tiles_per_chunk < a.ntiles ? t0 + a.tiles_per_chunk : a.ntiles;
    }
    return true;
}

// H = 1 (b2m_conv_wgrad_h, half-precision training): the operands are binary16 in memory -- half the gathered bytes --, every
// element is converted as it is loaded, and the block is accumulated by the same fp32 MFMAs (the f16 MFMA wants four PAIRS per
// lane where memory holds four CHANNELS per pair: an LDS transpose per slot, not built).
template <int MI, int NJ, int H = 0>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {
    constexpr int ESZ = H ? 2 : 4;            // bytes per operand element
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    int k, blk;
    int64_t chunk, t0, t1;
    if (!wgrad_item(a, wave, k, blk, chunk, t0, t1)) return;
    const int ci0 = (blk / a.nnb) * 16 * MI, co0 = (blk % a.nnb) * 16 * NJ;
    const bool identity = a.rb_in == nullptr;
    const int64_t ldr = a.ntiles * B2M_TILE;
    f32x4 acc[MI][NJ];
#pragma unroll
    for (int m = 0; m < MI; ++m)
#pragma unroll
        for (int n = 0; n < NJ; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    // a group SLOT = 16 pairs of one (tile, offset): slot = tile*4 + g
    const int64_t kbase = (int64_t)k * ldr;
    auto load_slot = [&](int64_t slot, int (&rin)[4], uint32_t& o4) {
        if (identity) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                int64_t r = slot * 16 + 4 * q + s;
                rin[s] = r < a.n_out ? (int)r : -1;
            }
            const int p = (int)(slot & 3) * 16 + 4 * q;
            o4 = (uint32_t)p | ((uint32_t)(p + 1) << 8) | ((uint32_t)(p + 2) << 16) | ((uint32_t)(p + 3) << 24);
        } else {
            const int64_t base = kbase + slot * 16 + 4 * q;
            i32x4 v = *(const i32x4*)(a.rb_in + base);
            rin[0] = v[0]; rin[1] = v[1]; rin[2] = v[2]; rin[3] = v[3];
            o4 = *(const uint32_t*)(a.rb_out + base);
        }
    };
    const uint32_t ldx4 = (uint32_t)a.ldx * (uint32_t)ESZ, lddy4 = (uint32_t)a.lddy * (uint32_t)ESZ;
    const uint32_t cxb = (uint32_t)(ci0 + i) * (uint32_t)ESZ, cyb = (uint32_t)(co0 + i) * (uint32_t)ESZ;
    auto process = [&](int64_t slot, const int (&rin)[4], uint32_t o4) {
        if (__ballot(rin[0] >= 0) == 0) return;        // wave-uniform; also makes the list wait explicit before the gathers
        const int64_t row0 = (slot >> 2) * B2M_TILE;
        float av[4][MI], bv[4][NJ];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int r = rin[s];
            const int64_t ro = row0 + ((o4 >> (8 * s)) & 255);
            // (b2m_conv_wgrad_tr: x lives on the tile's rows, dy on the rows the pair list names)
            const int64_t rx = a.swap ? ro : (int64_t)(r < 0 ? 0 : r), ry = a.swap ? (int64_t)(r < 0 ? 0 : r) : ro;
            if (a.fast32) {
                // complete blocks, tensors below 2^31 bytes: ONE full-rate 24-bit multiply-add per gathered row gives
                // the byte offset, the sub-tile offsets (64 B apart) become immediates of the loads
                const uint32_t bx = __umul24((uint32_t)rx, ldx4) + cxb;
                const uint32_t by = __umul24((uint32_t)ry, lddy4) + cyb;
                const char* px = (const char*)a.x + bx;
                const char* py = (const char*)a.dy + by;
                if (r >= 0) {
                    if constexpr (H) {
#pragma unroll
                        for (int m = 0; m < MI; ++m) av[s][m] = (float)*(const _Float16*)(px + 32 * m);
#pragma unroll
                        for (int nn = 0; nn < NJ; ++nn) bv[s][nn] = (float)*(const _Float16*)(py + 32 * nn);
                    } else {
#pragma unroll
                        for (int m = 0; m < MI; ++m) av[s][m] = *(const float*)(px + 64 * m);
#pragma unroll
                        for (int nn = 0; nn < NJ; ++nn) bv[s][nn] = *(const float*)(py + 64 * nn);
                    }
                } else {
#pragma unroll
                    for (int m = 0; m < MI; ++m) av[s][m] = 0.f;
#pragma unroll
                    for (int nn = 0; nn < NJ; ++nn) bv[s][nn] = 0.f;
                }
            } else {
#pragma unroll
                for (int m = 0; m < MI; ++m) {
                    const int ci = ci0 + 16 * m + i;
                    if constexpr (H) av[s][m] = (r >= 0 && ci < a.cin) ? (float)((const _Float16*)a.x)[rx * a.ldx + ci] : 0.f;
                    else av[s][m] = (r >= 0 && ci < a.cin) ? a.x[rx * a.ldx + ci] : 0.f;
                }
#pragma unroll
                for (int nn = 0; nn < NJ; ++nn) {
                    const int co = co0 + 16 * nn + i;
                    if constexpr (H) bv[s][nn] = (r >= 0 && co < a.cout) ? (float)((const _Float16*)a.dy)[ry * a.lddy + co] : 0.f;
                    else bv[s][nn] = (r >= 0 && co < a.cout) ? a.dy[ry * a.lddy + co] : 0.f;
                }
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int m = 0; m < MI; ++m)
#pragma unroll
                for (int nn = 0; nn < NJ; ++nn)
                    acc[m][nn] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s][m], bv[s][nn], acc[m][nn], 0, 0, 0);
    };
    // nested walk: only the non-empty groups of every tile (list, then data).  A flat slot walk with the next list
    // prefetched and a vectorised channel map were both measured slower (tools/bench_conv.py A/B, round 1).
    for (int64_t tile = t0; tile < t1; ++tile) {
        int n;
        if (identity) { int64_t rem = a.n_out - tile * B2M_TILE; n = rem < B2M_TILE ? (int)rem : B2M_TILE; }
        else n = __builtin_amdgcn_readfirstlane(a.rb_cnt[(int64_t)k * a.ntiles + tile]);
        const int G = (n + 15) >> 4;
        for (int g = 0; g < G; ++g) {
            int rin[4]; uint32_t o4;
            load_slot(tile * NG + g, rin, o4);
            process(tile * NG + g, rin, o4);
        }
    }
    // D[row = 4q + r (A's lane index), col = i (B's lane index)]
#pragma unroll
    for (int m = 0; m < MI; ++m)
#pragma unroll
        for (int nn = 0; nn < NJ; ++nn)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = ci0 + 16 * m + 4 * q + r, co = co0 + 16 * nn + i;
                if (ci < a.cin && co < a.cout) {
                    float v = acc[m][nn][r];
                    if constexpr (H) v *= a.out_scale;
                    if (a.partial) a.partial[(((int64_t)chunk * a.K + k) * a.cin + ci) * a.cout + co] = v;
                    else if (v != 0.f) atomicAdd(&a.dw[(int64_t)k * a.dw_kstride + (int64_t)ci * a.lddw + co], v);
                }
            }
}
// deterministic mode: dW += sum over the tile chunks of the partial blocks, in chunk order
__global__ void wgrad_reduce_kernel(const float* __restrict__ partial, int nchunks, int K, int cin, int cout,
                                    float* __restrict__ dw, int64_t lddw, int64_t dw_kstride) {
    const int64_t per = (int64_t)K * cin * cout;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < per; e += (int64_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int ch = 0; ch < nchunks; ++ch) s += partial[(int64_t)ch * per + e];
        const int co = (int)(e % cout); const int64_t e1 = e / cout;
        const int ci = (int)(e1 % cin); const int k = (int)(e1 / cin);
        dw[(int64_t)k * dw_kstride + (int64_t)ci * lddw + co] += s;
    }
}

// Hand-issued loads of conv_wgrad_flow_kernel<MI, NJ, HL != 0>: the two asm statements of a k-step, composed per block shape
// (MI, NJ in 2..4) from named operands.  WG_GATHER: the MI + NJ operand loads inside ONE EXEC window (destinations are in/out
// operands, see the kernel); WG_SELECT: the counted wait and the moves (0 for a missing pair) into the MFMA operands.
#define WG_LA2 "global_load_dword %[a0], %[bx], %[px]\n\tglobal_load_dword %[a1], %[bx], %[px] offset:64\n\t"
#define WG_LA3 WG_LA2 "global_load_dword %[a2], %[bx], %[px] offset:128\n\t"
#define WG_LA4 WG_LA3 "global_load_dword %[a3], %[bx], %[px] offset:192\n\t"
#define WG_LB2 "global_load_dword %[b0], %[by], %[py]\n\tglobal_load_dword %[b1], %[by], %[py] offset:64\n\t"
#define WG_LB3 WG_LB2 "global_load_dword %[b2], %[by], %[py] offset:128\n\t"
#define WG_LB4 WG_LB3 "global_load_dword %[b3], %[by], %[py] offset:192\n\t"
#define WG_OA2 [a0] "+v"(av[s][0]), [a1] "+v"(av[s][1])
#define WG_OA3 WG_OA2, [a2] "+v"(av[s][2])
#define WG_OA4 WG_OA3, [a3] "+v"(av[s][3])
#define WG_OB2 [b0] "+v"(bv[s][0]), [b1] "+v"(bv[s][1])
#define WG_OB3 WG_OB2, [b2] "+v"(bv[s][2])
#define WG_OB4 WG_OB3, [b3] "+v"(bv[s][3])
#define WG_GATHER(MI_, NJ_)                                                                                        \
    asm volatile("s_mov_b64 exec, %[em]\n\t" WG_LA##MI_ WG_LB##NJ_ "s_mov_b64 exec, -1"                            \
                 : WG_OA##MI_, WG_OB##NJ_                                                                          \
                 : [bx] "v"(bx), [by] "v"(by), [px] "s"(a.x), [py] "s"(a.dy), [em] "s"(em) : "memory");
#define WG_SA2 "v_cndmask_b32_e64 %[za0], 0, %[a0], %[pm]\n\tv_cndmask_b32_e64 %[za1], 0, %[a1], %[pm]\n\t"
#define WG_SA3 WG_SA2 "v_cndmask_b32_e64 %[za2], 0, %[a2], %[pm]\n\t"
#define WG_SA4 WG_SA3 "v_cndmask_b32_e64 %[za3], 0, %[a3], %[pm]\n\t"
#define WG_SB2 "v_cndmask_b32_e64 %[zb0], 0, %[b0], %[pm]\n\tv_cndmask_b32_e64 %[zb1], 0, %[b1], %[pm]"
#define WG_SB3 WG_SB2 "\n\tv_cndmask_b32_e64 %[zb2], 0, %[b2], %[pm]"
#define WG_SB4 WG_SB3 "\n\tv_cndmask_b32_e64 %[zb3], 0, %[b3], %[pm]"
#define WG_ZA2 [za0] "=&v"(az[0]), [za1] "=&v"(az[1])
#define WG_ZA3 WG_ZA2, [za2] "=&v"(az[2])
#define WG_ZA4 WG_ZA3, [za3] "=&v"(az[3])
#define WG_ZB2 [zb0] "=&v"(bz[0]), [zb1] "=&v"(bz[1])
#define WG_ZB3 WG_ZB2, [zb2] "=&v"(bz[2])
#define WG_ZB4 WG_ZB3, [zb3] "=&v"(bz[3])
#define WG_IA2 [a0] "v"(av[s][0]), [a1] "v"(av[s][1])
#define WG_IA3 WG_IA2, [a2] "v"(av[s][2])
#define WG_IA4 WG_IA3, [a3] "v"(av[s][3])
#define WG_IB2 [b0] "v"(bv[s][0]), [b1] "v"(bv[s][1])
#define WG_IB3 WG_IB2, [b2] "v"(bv[s][2])
#define WG_IB4 WG_IB3, [b3] "v"(bv[s][3])
#define WG_SELECT(MI_, NJ_)                                                                                        \
    asm volatile("s_waitcnt vmcnt(%[cnt])\n\t" WG_SA##MI_ WG_SB##NJ_                                               \
                 : WG_ZA##MI_, WG_ZB##NJ_                                                                          \
                 : WG_IA##MI_, WG_IB##NJ_, [pm] "s"(pm), [cnt] "n"(3 * (MI + NJ) + 2) : "memory");
#define WG_DISPATCH(STMT)                                                                                          \
    if constexpr (MI == 2 && NJ == 2) { STMT(2, 2) } else if constexpr (MI == 2 && NJ == 3) { STMT(2, 3) }         \
    else if constexpr (MI == 2 && NJ == 4) { STMT(2, 4) } else if constexpr (MI == 3 && NJ == 2) { STMT(3, 2) }    \
    else if constexpr (MI == 3 && NJ == 3) { STMT(3, 3) } else if constexpr (MI == 3 && NJ == 4) { STMT(3, 4) }    \
    else if constexpr (MI == 4 && NJ == 2) { STMT(4, 2) } else if constexpr (MI == 4 && NJ == 3) { STMT(4, 3) }    \
    else { STMT(4, 4) }

// conv_wgrad_flow_kernel: the common case (real rulebook, complete blocks, 32-bit addressable operands).
// conv_wgrad_kernel above handles a slot as  list -> wait -> gathers -> wait -> 36..64 MFMAs.  Here the walk over the
// non-empty 16-pair slots of the chunk is flat and pipelined: the pair list is fetched two slots ahead, the operands one
// slot ahead.  Differences to the first pipelined form (round 1):
//  * a k-step (one MFMA per block element) covers the four CONSECUTIVE pairs 4s .. 4s+3 of the slot (lane quarter q
//    supplies pair 4s + q; the words come from a cross-lane permute of the list), so a slot with n pairs needs only
//    ceil(n/4) k-steps: the MFMAs of the empty ones are skipped (before: always 4; useful share of the executed
//    MFMAs 0.8 -> 0.97);
//  * every MFMA is an asm statement with the accumulator tied (in place).  With the builtin, hipcc kept the
//    accumulators in AGPRs and copied all of them to VGPRs and back around every slot (108 moves per 36 MFMAs);
//  * each slot issues the same number of loads (k-steps without pairs gather row 0: L1 hits), so the counted waits
//    in front of the MFMAs are exact; the loads that refill a k-step's registers follow its MFMAs directly.
// HL = 1 (round 4; real rulebooks, blocks of 2..4 x 2..4 sub-tiles): the operand and pair-list loads are issued by hand, as in
// conv_fwd_flow_kernel -- a k-step's MI + NJ loads sit in ONE EXEC window that holds the lanes whose pair exists (a slot's
// missing pairs and empty k-steps fetch nothing: 14 % of the loads on the benchmark's maps), the waits are counted here: the
// operands of k-step s are complete when all but the 3 (MI + NJ) refills and the 2 list loads issued since have landed.
// hipcc knows nothing of a load in flight, so (tests/test_isa.py follows the registers through the assembly):
//  * every register such a load writes is READ only by the asm statement that waits for it (as the in/out operand of a bare
//    wait hipcc copied it to a fresh register IN FRONT of the wait: whole blocks of dW wrong on some runs);
//  * the load takes its destination as an IN/OUT operand, so the register is never free (as a plain output it is dead from
//    its last use on -- and that use, the wait of a k-step without pairs, may have been skipped with the load still in
//    flight: the next address computation landed in it, a GPU memory fault).
// SWP = 1 (round 5, b2m_conv_wgrad_tr): x lives on the tile's rows, dy on the rows the pair list names -- a variant of its own: as a
// run-time switch the select cost the 2 x 4 blocks a resident wave (81 VGPRs)
template <int MI, int NJ, int HL = 0, int SWP = 0>
__global__ __launch_bounds__(256) void conv_wgrad_flow_kernel(WgradArgs a) {
    static_assert(!HL || (MI >= 2 && NJ >= 2), "hand-issued loads: blocks of 2..4 x 2..4 sub-tiles");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 15, q = lane >> 4;
    int k, blk;
    int64_t chunk, t0, t1;
    if (!wgrad_item(a, wave, k, blk, chunk, t0, t1)) return;
    B2M_CLOCK_BEGIN();
    const int ci0 = (blk / a.nnb) * 16 * MI, co0 = (blk % a.nnb) * 16 * NJ;
    const int64_t ldr = a.ntiles * B2M_TILE;
    const int nt = (int)(t1 - t0);                       // <= 64 tiles: lane t holds the pair count of tile t0 + t
    // identity map (1x1 layers): pair j of a tile is (row j, row j); never with hand-issued loads
    const bool ident = HL ? false : a.rb_in == nullptr;
    int cnt = 0;
    if (lane < nt) {
        if (ident) { const int64_t rem = a.n_out - (t0 + lane) * B2M_TILE; cnt = rem < B2M_TILE ? (int)rem : B2M_TILE; }
        else cnt = a.rb_cnt[(int64_t)k * a.ntiles + t0 + lane];
    }
    const uint64_t live = __ballot(cnt > 0);
    if (live == 0) return;

    // flat walk: position = (tile index ti, group g); advance() returns false past the end
    auto pairs_of = [&](int ti, int g) { const int n = __builtin_amdgcn_readlane(cnt, ti) - 16 * g; return n > 16 ? 16 : n; };
    auto advance = [&](int& ti, int& g) {
        if (16 * (g + 1) < __builtin_amdgcn_readlane(cnt, ti)) { ++g; return true; }
        const uint64_t rest_mask = ti >= 63 ? 0ull : (live >> (ti + 1));
        if (rest_mask == 0) return false;
        ti = ti + 1 + __builtin_ctzll(rest_mask); g = 0;
        return true;
    };
    const int64_t kbase = (int64_t)k * ldr + t0 * B2M_TILE;
    // list of a slot: lane (i, .) loads entry i; word = input row | row inside the tile << 24, bit 31 = no pair
    auto load_list = [&](int ti, int g, int& r_in, int& r_out) {
        if (ident) {                                     // (wave-uniform)
            const int64_t row = (t0 + ti) * B2M_TILE + 16 * g + i;
            r_in = row < a.n_out ? (int)row : -1;
            r_out = 16 * g + i;
            return;
        }
        const int64_t base = kbase + (int64_t)ti * B2M_TILE + 16 * g + i;
        r_in = a.rb_in[base];
        r_out = a.rb_out[base];
    };
    auto load_list_hl = [&](int ti, int g, int& r_in, int& r_out) {      // (hand-issued)
        const int64_t base = kbase + (int64_t)ti * B2M_TILE + 16 * g;      // wave-uniform
        const int32_t* pin = a.rb_in + base;
        const uint8_t* pout = a.rb_out + base;
        // (destinations as IN/OUT operands, like the operand loads: the registers stay allocated up to the statement that
        // waits for them, whatever hipcc moves in between)
        asm volatile("global_load_dword %0, %1, %2" : "+v"(r_in) : "v"((uint32_t)i * 4u), "s"(pin) : "memory");
        asm volatile("global_load_ubyte %0, %1, %2" : "+v"(r_out) : "v"((uint32_t)i), "s"(pout) : "memory");
    };
    // the word of pair 4s + q for k-step s
    auto words = [&](int r_in, int r_out, uint32_t (&w)[4]) {
        const uint32_t word = r_in < 0 ? 0x80000000u : ((uint32_t)r_in | ((uint32_t)r_out << 24));
#pragma unroll
        for (int s = 0; s < 4; ++s) w[s] = (uint32_t)__builtin_amdgcn_ds_bpermute((4 * s + q) << 2, (int)word);
    };
    const uint32_t ldx4 = (uint32_t)a.ldx * 4u, lddy4 = (uint32_t)a.lddy * 4u;
    const uint32_t cxb = (uint32_t)(ci0 + i) * 4u, cyb = (uint32_t)(co0 + i) * 4u;
    float av[4][MI], bv[4][NJ];
    if constexpr (HL != 0) {
        // hand-issued loads take their destination as an IN/OUT operand: the previous content stays alive, in that register, up to
        // the load.  (As a plain output the register is dead to hipcc from its last use on -- and the last use, the wait of a
        // k-step without pairs, may have been skipped with the load still in flight: the next address computation landed there.)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int m = 0; m < MI; ++m) av[s][m] = 0.f;
#pragma unroll
            for (int n = 0; n < NJ; ++n) bv[s][n] = 0.f;
        }
    }
    // operands of k-step s of the slot in tile ti (MI + NJ loads, always)
    auto gather = [&](int s, int ti, uint32_t word) {
        const uint32_t row0 = (uint32_t)((t0 + ti) * B2M_TILE);
        const uint32_t rlist = word & 0xFFFFFFu, rtile = row0 + ((word >> 24) & 63u);
        const uint32_t bx = __umul24(SWP ? rtile : rlist, ldx4) + cxb;
        const uint32_t by = __umul24(SWP ? rlist : rtile, lddy4) + cyb;
        const char* px = (const char*)a.x + bx;
        const char* py = (const char*)a.dy + by;
        if constexpr (HL) {
            // lanes whose pair exists (bit 31 of the word clear); at least lane 0, so that the loads are always issued
            const uint64_t em = __ballot((int)word >= 0) | 1ull;
            WG_DISPATCH(WG_GATHER)
        } else {
#pragma unroll
            for (int m = 0; m < MI; ++m) av[s][m] = *(const float*)(px + 64 * m);
#pragma unroll
            for (int nn = 0; nn < NJ; ++nn) bv[s][nn] = *(const float*)(py + 64 * nn);
        }
    };
    f32x4 acc[MI][NJ];
#pragma unroll
    for (int m = 0; m < MI; ++m)
#pragma unroll
        for (int n = 0; n < NJ; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    // prologue: list + operands of slot 0, list of slot 1, list load of slot 2
    int tiC = __builtin_ctzll(live), gC = 0;
    uint32_t wC[4], wN[4];
    int rawi = 0, rawo = 0;
    int tiN = tiC, gN = gC;
    bool hasN = advance(tiN, gN);
    {
        int r0i, r0o, r1i, r1o;
        load_list(tiC, gC, r0i, r0o);
        load_list(hasN ? tiN : tiC, hasN ? gN : gC, r1i, r1o);
        words(r0i, r0o, wC);
        words(r1i, r1o, wN);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) gather(s, tiC, wC[s]);
    int tiNN = tiN, gNN = gN;
    bool hasNN = hasN && advance(tiNN, gNN);
    if (HL) load_list_hl(hasNN ? tiNN : tiN, hasNN ? gNN : gN, rawi, rawo);
    else load_list(hasNN ? tiNN : tiN, hasNN ? gNN : gN, rawi, rawo);
    if (!hasN) {
#pragma unroll
        for (int s = 0; s < 4; ++s) wN[s] = 0x80000000u;       // no next slot: the refills gather row 0 and are never used
    }
    int nkC = (pairs_of(tiC, gC) + 3) >> 2;

    for (;;) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s < nkC) {                                       // wave-uniform
                float bz[NJ], az[MI];
                if constexpr (HL) {
                    // ONE statement waits for the k-step's operands and moves them (0 for a missing pair: the masked lanes'
                    // registers are stale, possibly NaN) into the MFMA operands.  The loaded registers are plain INPUTS of
                    // it: as in/out operands of a bare wait, hipcc copied them to fresh registers IN FRONT of the wait.
                    const uint64_t pm = __ballot((int)wC[s] >= 0);
                    WG_DISPATCH(WG_SELECT)
                } else {
#pragma unroll
                    for (int nn = 0; nn < NJ; ++nn) bz[nn] = (int)wC[s] >= 0 ? bv[s][nn] : 0.f;      // no pair: contributes 0
#pragma unroll
                    for (int m = 0; m < MI; ++m) az[m] = av[s][m];
                }
#pragma unroll
                for (int m = 0; m < MI; ++m)
#pragma unroll
                    for (int nn = 0; nn < NJ; ++nn)
#ifdef B2M_WGRAD_NOMFMA      // diagnostic build (wrong results): what the loop costs without its MFMAs -- the operands stay consumed
                        asm volatile("" : "+v"(acc[m][nn]) : "v"(az[m]), "v"(bz[nn]));
#else
                    {
                        // hipcc-tracked loads: the operands were just written by VALU instructions (the select of bz, the
                        // conversions of the half form) and an MFMA that reads a VGPR fewer than 2 wait states after a VALU wrote
                        // it reads the OLD value -- hipcc pads that for its own MFMAs, not for an asm statement.  (Seen in round
                        // 6 on the half form's 2 x 2 blocks: the first MFMA of a k-step took bz[0] unmasked.  The hand-issued
                        // form's select is an asm statement, behind which hipcc places one state of its own; tests/test_isa.py
                        // checks the distance on every variant.)
                        if (!HL && m == 0 && nn == 0)
                            asm volatile("s_nop 1\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[m][nn]) : "v"(az[m]), "v"(bz[nn]));
                        else
                            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[m][nn]) : "v"(az[m]), "v"(bz[nn]));
                    }
#endif
            }
            asm volatile("" ::: "memory");                       // the refill stays behind the MFMAs that read the registers
            gather(s, tiN, wN[s]);
        }
        if (!hasN) break;
        // next slot becomes current; its successor's list has arrived; fetch one more
        tiC = tiN; gC = gN;
        nkC = (pairs_of(tiC, gC) + 3) >> 2;
#pragma unroll
        for (int s = 0; s < 4; ++s) wC[s] = wN[s];
        hasN = hasNN; tiN = tiNN; gN = gNN;
        // (hand-issued list loads: one slot old, the 4 k-steps' refills are younger)
        if constexpr (HL) {   // (the loaded registers are inputs of the statement that waits for them, see above)
            int li, lo;
            asm volatile("s_waitcnt vmcnt(%4)\n\tv_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(li), "=&v"(lo) : "v"(rawi), "v"(rawo),
                         "n"(4 * (MI + NJ)) : "memory");
            words(li, lo, wN);
        } else words(rawi, rawo, wN);
        if (!hasN) {
#pragma unroll
            for (int s = 0; s < 4; ++s) wN[s] = 0x80000000u;
        }
        hasNN = hasN && advance(tiNN, gNN);
        if (HL) load_list_hl(hasNN ? tiNN : tiN, hasNN ? gNN : gN, rawi, rawo);
        else load_list(hasNN ? tiNN : tiN, hasNN ? gNN : gN, rawi, rawo);
    }
    // hand-issued loads: the last refills are still in flight and hipcc is about to reuse their registers
    if constexpr (HL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_nop 15" ::: "memory");          // MFMA result -> VALU / memory read: >= 12 wait states
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
    B2M_CLOCK_END(3);
}
// The same walk for IEEE binary16 operands (round 6, b2m_conv_wgrad_h: half-precision training) -- a kernel of its own, not a
// variant of the template above: sharing the body through a forced-inline function changed hipcc's schedule of the fp32
// kernels with 1 x 2 / 2 x 2 blocks and their weight gradients came out wrong (tests/test_gpu_conv_regimes.py caught it).
// hipcc-tracked loads only: a load brings the 16 bits of its element into a 32-bit register, the conversion to fp32 sits in front
// of the k-step's MFMAs, one slot after the load; the slots and the fp32 MFMAs are those of the fp32 kernel, the gathered bytes half.
template <int MI, int NJ, int SWP>
__global__ __launch_bounds__(256) void conv_wgrad_flow_h_kernel(WgradArgs a) {
    constexpr int HL = 0;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 15, q = lane >> 4;
    int k, blk;
    int64_t chunk, t0, t1;
    if (!wgrad_item(a, wave, k, blk, chunk, t0, t1)) return;
    B2M_CLOCK_BEGIN();
    const int ci0 = (blk / a.nnb) * 16 * MI, co0 = (blk % a.nnb) * 16 * NJ;
    const int64_t ldr = a.ntiles * B2M_TILE;
    const int nt = (int)(t1 - t0);                       // <= 64 tiles: lane t holds the pair count of tile t0 + t
    // identity map (1x1 layers): pair j of a tile is (row j, row j); never with hand-issued loads
    const bool ident = HL ? false : a.rb_in == nullptr;
    int cnt = 0;
    if (lane < nt) {
        if (ident) { const int64_t rem = a.n_out - (t0 + lane) * B2M_TILE; cnt = rem < B2M_TILE ? (int)rem : B2M_TILE; }
        else cnt = a.rb_cnt[(int64_t)k * a.ntiles + t0 + lane];
    }
    const uint64_t live = __ballot(cnt > 0);
    if (live == 0) return;

    // flat walk: position = (tile index ti, group g); advance() returns false past the end
    auto pairs_of = [&](int ti, int g) { const int n = __builtin_amdgcn_readlane(cnt, ti) - 16 * g; return n > 16 ? 16 : n; };
    auto advance = [&](int& ti, int& g) {
        if (16 * (g + 1) < __builtin_amdgcn_readlane(cnt, ti)) { ++g; return true; }
        const uint64_t rest_mask = ti >= 63 ? 0ull : (live >> (ti + 1));
        if (rest_mask == 0) return false;
        ti = ti + 1 + __builtin_ctzll(rest_mask); g = 0;
        return true;
    };
    const int64_t kbase = (int64_t)k * ldr + t0 * B2M_TILE;
    // list of a slot: lane (i, .) loads entry i; word = input row | row inside the tile << 24, bit 31 = no pair
    auto load_list = [&](int ti, int g, int& r_in, int& r_out) {
        if (ident) {                                     // (wave-uniform)
            const int64_t row = (t0 + ti) * B2M_TILE + 16 * g + i;
            r_in = row < a.n_out ? (int)row : -1;
            r_out = 16 * g + i;
            return;
        }
        const int64_t base = kbase + (int64_t)ti * B2M_TILE + 16 * g + i;
        r_in = a.rb_in[base];
        r_out = a.rb_out[base];
    };
    auto load_list_hl = [&](int ti, int g, int& r_in, int& r_out) {      // (hand-issued)
        const int64_t base = kbase + (int64_t)ti * B2M_TILE + 16 * g;      // wave-uniform
        const int32_t* pin = a.rb_in + base;
        const uint8_t* pout = a.rb_out + base;
        // (destinations as IN/OUT operands, like the operand loads: the registers stay allocated up to the statement that
        // waits for them, whatever hipcc moves in between)
        asm volatile("global_load_dword %0, %1, %2" : "+v"(r_in) : "v"((uint32_t)i * 4u), "s"(pin) : "memory");
        asm volatile("global_load_ubyte %0, %1, %2" : "+v"(r_out) : "v"((uint32_t)i), "s"(pout) : "memory");
    };
    // the word of pair 4s + q for k-step s
    auto words = [&](int r_in, int r_out, uint32_t (&w)[4]) {
        const uint32_t word = r_in < 0 ? 0x80000000u : ((uint32_t)r_in | ((uint32_t)r_out << 24));
#pragma unroll
        for (int s = 0; s < 4; ++s) w[s] = (uint32_t)__builtin_amdgcn_ds_bpermute((4 * s + q) << 2, (int)word);
    };
    const uint32_t ldx4 = (uint32_t)a.ldx * 2u, lddy4 = (uint32_t)a.lddy * 2u;
    const uint32_t cxb = (uint32_t)(ci0 + i) * 2u, cyb = (uint32_t)(co0 + i) * 2u;
    uint32_t av[4][MI], bv[4][NJ];          // (the 16 bits of a loaded element, zero-extended)
    // operands of k-step s of the slot in tile ti (MI + NJ loads, always)
    auto gather = [&](int s, int ti, uint32_t word) {
        const uint32_t row0 = (uint32_t)((t0 + ti) * B2M_TILE);
        const uint32_t rlist = word & 0xFFFFFFu, rtile = row0 + ((word >> 24) & 63u);
        const uint32_t bx = __umul24(SWP ? rtile : rlist, ldx4) + cxb;
        const uint32_t by = __umul24(SWP ? rlist : rtile, lddy4) + cyb;
        const char* px = (const char*)a.x + bx;
        const char* py = (const char*)a.dy + by;
        {
#pragma unroll
            for (int m = 0; m < MI; ++m) av[s][m] = (uint32_t)*(const unsigned short*)(px + 32 * m);
#pragma unroll
            for (int nn = 0; nn < NJ; ++nn) bv[s][nn] = (uint32_t)*(const unsigned short*)(py + 32 * nn);
        }
    };
    f32x4 acc[MI][NJ];
#pragma unroll
    for (int m = 0; m < MI; ++m)
#pragma unroll
        for (int n = 0; n < NJ; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    // prologue: list + operands of slot 0, list of slot 1, list load of slot 2
    int tiC = __builtin_ctzll(live), gC = 0;
    uint32_t wC[4], wN[4];
    int rawi = 0, rawo = 0;
    int tiN = tiC, gN = gC;
    bool hasN = advance(tiN, gN);
    {
        int r0i, r0o, r1i, r1o;
        load_list(tiC, gC, r0i, r0o);
        load_list(hasN ? tiN : tiC, hasN ? gN : gC, r1i, r1o);
        words(r0i, r0o, wC);
        words(r1i, r1o, wN);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) gather(s, tiC, wC[s]);
    int tiNN = tiN, gNN = gN;
    bool hasNN = hasN && advance(tiNN, gNN);
    if (HL) load_list_hl(hasNN ? tiNN : tiN, hasNN ? gNN : gN, rawi, rawo);
    else load_list(hasNN ? tiNN : tiN, hasNN ? gNN : gN, rawi, rawo);
    if (!hasN) {
#pragma unroll
        for (int s = 0; s < 4; ++s) wN[s] = 0x80000000u;       // no next slot: the refills gather row 0 and are never used
    }
    int nkC = (pairs_of(tiC, gC) + 3) >> 2;

    for (;;) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s < nkC) {                                       // wave-uniform
                float bz[NJ], az[MI];
                {
#pragma unroll
                    for (int nn = 0; nn < NJ; ++nn)
                        bz[nn] = (int)wC[s] >= 0 ? (float)__builtin_bit_cast(_Float16, (unsigned short)bv[s][nn]) : 0.f;
#pragma unroll
                    for (int m = 0; m < MI; ++m) az[m] = (float)__builtin_bit_cast(_Float16, (unsigned short)av[s][m]);
                }
#pragma unroll
                for (int m = 0; m < MI; ++m)
#pragma unroll
                    for (int nn = 0; nn < NJ; ++nn)
#ifdef B2M_WGRAD_NOMFMA      // diagnostic build (wrong results): what the loop costs without its MFMAs -- the operands stay consumed
                        asm volatile("" : "+v"(acc[m][nn]) : "v"(az[m]), "v"(bz[nn]));
#else
                    {
                        // (VALU write -> MFMA operand: 2 wait states, see the fp32 kernel)
                        if (!HL && m == 0 && nn == 0)
                            asm volatile("s_nop 1\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[m][nn]) : "v"(az[m]), "v"(bz[nn]));
                        else
                            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[m][nn]) : "v"(az[m]), "v"(bz[nn]));
                    }
#endif
            }
            asm volatile("" ::: "memory");                       // the refill stays behind the MFMAs that read the registers
            gather(s, tiN, wN[s]);
        }
        if (!hasN) break;
        // next slot becomes current; its successor's list has arrived; fetch one more
        tiC = tiN; gC = gN;
        nkC = (pairs_of(tiC, gC) + 3) >> 2;
#pragma unroll
        for (int s = 0; s < 4; ++s) wC[s] = wN[s];
        hasN = hasNN; tiN = tiNN; gN = gNN;
        // (hand-issued list loads: one slot old, the 4 k-steps' refills are younger)
        if constexpr (HL) {   // (the loaded registers are inputs of the statement that waits for them, see above)
            int li, lo;
            asm volatile("s_waitcnt vmcnt(%4)\n\tv_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(li), "=&v"(lo) : "v"(rawi), "v"(rawo),
                         "n"(4 * (MI + NJ)) : "memory");
            words(li, lo, wN);
        } else words(rawi, rawo, wN);
        if (!hasN) {
#pragma unroll
            for (int s = 0; s < 4; ++s) wN[s] = 0x80000000u;
        }
        hasNN = hasN && advance(tiNN, gNN);
        if (HL) load_list_hl(hasNN ? tiNN : tiN, hasNN ? gNN : gN, rawi, rawo);
        else load_list(hasNN ? tiNN : tiN, hasNN ? gNN : gN, rawi, rawo);
    }
    // hand-issued loads: the last refills are still in flight and hipcc is about to reuse their registers
    if constexpr (HL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_nop 15" ::: "memory");          // MFMA result -> VALU / memory read: >= 12 wait states
#pragma unroll
    for (int m = 0; m < MI; ++m)
#pragma unroll
        for (int nn = 0; nn < NJ; ++nn)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = ci0 + 16 * m + 4 * q + r, co = co0 + 16 * nn + i;
                if (ci < a.cin && co < a.cout) {
                    const float v = acc[m][nn][r] * a.out_scale;          // (1 / loss scale)
                    if (v != 0.f) atomicAdd(&a.dw[(int64_t)k * a.dw_kstride + (int64_t)ci * a.lddw + co], v);
                }
            }
    B2M_CLOCK_END(3);
}

// Half operands on the f16 MFMA (round 6, the default of b2m_conv_wgrad_h).  dW[ci][co] = sum over pairs of x[pair][ci] * dy[pair][co]:
// the MFMA's reduction index is the PAIR, and memory holds a pair's channels side by side -- v_mfma_f32_16x16x16_f16 wants lane
// (i, g) to hold channel i of the four pairs 4g .. 4g+3.  The kernels above gather element by element (a 2-byte load per lane and
// sub-tile, 4 (MI + NJ) loads per 16-pair slot) and multiply on the fp32 MFMA (4 MI NJ per slot): 69 % of the fp32 MFMA peak on the
// 96 -> 96 layers, and that peak is the bound.  Here a wave
//   * gathers the 16 rows of a slot with 16-byte loads (a lane = 8 channels of one pair: 2 MI / 64 + 2 NJ / 64 loads per slot),
//   * writes them row-major into its own LDS image ([16 pairs][channels of the block], two buffers; no other wave reads it: no
//     barrier, LDS operations of a wave complete in order),
//   * reads the operands back with ds_read_b64_tr_b16 -- gfx950's transposing read: a group of 16 lanes takes a 4-row x 16-column
//     block of 16-bit elements and lane i receives column i of the 4 rows, exactly the MFMA's A / B layout -- MI + NJ reads,
//   * and issues MI NJ v_mfma_f32_16x16x16_f16 per slot (fp32 accumulators; the products of two halves are exact in fp32, so
//     the sums differ from the kernels above by their order only).
// Missing pairs are zero rows (the load's ADDRESS is switched to g_zeros); a slot always costs its MI NJ MFMAs (a quarter of one
// fp32 k-step each).  Walk, work items, XCD order and the atomics into dW are those of conv_wgrad_flow_kernel; loads are tracked by
// hipcc, the MFMAs are builtins (hipcc pads their hazards).  Row pitch of the LDS image: 32 B (MI = 1), 96 B (MI = 2, 3), 160 B
// (MI = 4): the eight rows a 32-lane half reads start in eight different groups of 8 banks.
typedef __fp16 b2m_h4v __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int M> struct TrhPitch { static constexpr int value = M == 1 ? 32 : M <= 3 ? 96 : 160; };
template <int MI, int NJ, int SWP>
__global__ __launch_bounds__(256) void conv_wgrad_trh_kernel(WgradArgs a) {
    constexpr int SX = TrhPitch<MI>::value, SY = TrhPitch<NJ>::value;      // bytes per image row
    constexpr int CPX = 2 * MI, CPY = 2 * NJ;                                // 16-byte chunks per row
    constexpr int NX = (16 * CPX + 63) / 64, NY = (16 * CPY + 63) / 64;      // loads per slot and lane
    constexpr int BUF = 16 * (SX + SY);                                      // one buffer: x image, dy image
    __shared__ __attribute__((aligned(16))) unsigned char lds_all[4 * 2 * BUF];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 15, q = lane >> 4;
    int k, blk;
    int64_t chunk, t0, t1;
    if (!wgrad_item(a, wave, k, blk, chunk, t0, t1)) return;
    const int ci0 = (blk / a.nnb) * 16 * MI, co0 = (blk % a.nnb) * 16 * NJ;
    const int64_t ldr = a.ntiles * B2M_TILE;
    const int nt = (int)(t1 - t0);
    int cnt = 0;
    if (lane < nt) cnt = a.rb_cnt[(int64_t)k * a.ntiles + t0 + lane];
    const uint64_t live = __ballot(cnt > 0);
    if (live == 0) return;
    unsigned char* const lds = lds_all + wave * 2 * BUF;

    auto pairs_of = [&](int ti, int g) { const int n = __builtin_amdgcn_readlane(cnt, ti) - 16 * g; return n > 16 ? 16 : n; };
    auto advance = [&](int& ti, int& g) {
        if (16 * (g + 1) < __builtin_amdgcn_readlane(cnt, ti)) { ++g; return true; }
        const uint64_t rest_mask = ti >= 63 ? 0ull : (live >> (ti + 1));
        if (rest_mask == 0) return false;
        ti = ti + 1 + __builtin_ctzll(rest_mask); g = 0;
        return true;
    };
    const int64_t kbase = (int64_t)k * ldr + t0 * B2M_TILE;
    // entry i of the slot: input row | row inside the tile << 24, bit 31 = no pair
    auto load_word = [&](int ti, int g) -> uint32_t {
        const int64_t base = kbase + (int64_t)ti * B2M_TILE + 16 * g + i;
        const int r_in = a.rb_in[base];
        const uint32_t r_out = a.rb_out[base];
        return (r_in < 0 || i >= pairs_of(ti, g)) ? 0x80000000u : ((uint32_t)r_in | (r_out << 24));
    };
    // chunk e = lane + 64 j of an image: pair e / CP, 16-byte chunk e % CP of its row
    int px[NX], py[NY];
    uint32_t cxb[NX], cyb[NY], lx[NX], ly[NY];
    bool actx[NX], acty[NY];
#pragma unroll
    for (int j = 0; j < NX; ++j) {
        const int e = lane + 64 * j;
        actx[j] = e < 16 * CPX;
        px[j] = actx[j] ? e / CPX : 0;
        cxb[j] = (uint32_t)(ci0 * 2 + 16 * (e % CPX));
        lx[j] = (uint32_t)(px[j] * SX + 16 * (e % CPX));
    }
#pragma unroll
    for (int j = 0; j < NY; ++j) {
        const int e = lane + 64 * j;
        acty[j] = e < 16 * CPY;
        py[j] = acty[j] ? e / CPY : 0;
        cyb[j] = (uint32_t)(co0 * 2 + 16 * (e % CPY));
        ly[j] = (uint32_t)(16 * SX + py[j] * SY + 16 * (e % CPY));
    }
    const uint32_t ldxb = (uint32_t)a.ldx * 2u, lddyb = (uint32_t)a.lddy * 2u;
    const char* const zeros = (const char*)a.zeros;
    u32x4 rx[NX], ry[NY];
    auto gather = [&](int ti, uint32_t word) {
        const uint32_t row0 = (uint32_t)((t0 + ti) * B2M_TILE);
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const uint32_t w = (uint32_t)__builtin_amdgcn_ds_bpermute(px[j] << 2, (int)word);
            const uint32_t rlist = w & 0xFFFFFFu, rtile = row0 + ((w >> 24) & 63u);
            const char* p = (const char*)a.x + (__umul24(SWP ? rtile : rlist, ldxb) + cxb[j]);
            rx[j] = *(const u32x4*)(((int)w >= 0 && actx[j]) ? p : zeros);
        }
#pragma unroll
        for (int j = 0; j < NY; ++j) {
            const uint32_t w = (uint32_t)__builtin_amdgcn_ds_bpermute(py[j] << 2, (int)word);
            const uint32_t rlist = w & 0xFFFFFFu, rtile = row0 + ((w >> 24) & 63u);
            const char* p = (const char*)a.dy + (__umul24(SWP ? rlist : rtile, lddyb) + cyb[j]);
            ry[j] = *(const u32x4*)(((int)w >= 0 && acty[j]) ? p : zeros);
        }
    };
    auto store = [&](int buf) {
        unsigned char* b = lds + buf * BUF;
#pragma unroll
        for (int j = 0; j < NX; ++j)
            if (actx[j]) *(u32x4*)(b + lx[j]) = rx[j];
#pragma unroll
        for (int j = 0; j < NY; ++j)
            if (acty[j]) *(u32x4*)(b + ly[j]) = ry[j];
    };
    // transposing read: lane 4q' + p' of a 16-lane group addresses row 4 group + q', columns 4p' .. 4p'+3 of the block
    const uint32_t trx = (uint32_t)((4 * q + (i >> 2)) * SX + 8 * (i & 3));
    const uint32_t try_ = (uint32_t)(16 * SX + (4 * q + (i >> 2)) * SY + 8 * (i & 3));
    f32x4 acc[MI][NJ];
#pragma unroll
    for (int m = 0; m < MI; ++m)
#pragma unroll
        for (int n = 0; n < NJ; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    typedef __attribute__((address_space(3))) b2m_h4v* lds_h4;
    auto compute = [&](int buf) {
        unsigned char* b = lds + buf * BUF;
        f16x4 av[MI], bv[NJ];
#pragma unroll
        for (int m = 0; m < MI; ++m) av[m] = __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h4)(b + trx + 32 * m)));
#pragma unroll
        for (int nn = 0; nn < NJ; ++nn) bv[nn] = __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h4)(b + try_ + 32 * nn)));
#pragma unroll
        for (int m = 0; m < MI; ++m)
#pragma unroll
            for (int nn = 0; nn < NJ; ++nn) acc[m][nn] = __builtin_amdgcn_mfma_f32_16x16x16f16(av[m], bv[nn], acc[m][nn], 0, 0, 0);
    };

    // prologue: slot C staged in buffer 0, slot N's rows in flight, slot NN's list entry in flight
    int tiC = __builtin_ctzll(live), gC = 0;
    int tiN = tiC, gN = gC;
    bool hasN = advance(tiN, gN);
    int tiNN = tiN, gNN = gN;
    bool hasNN = hasN && advance(tiNN, gNN);
    {
        const uint32_t wC = load_word(tiC, gC);
        const uint32_t wN = hasN ? load_word(tiN, gN) : 0x80000000u;
        gather(tiC, wC);
        store(0);
        gather(tiN, wN);
    }
    uint32_t wNN = hasNN ? load_word(tiNN, gNN) : 0x80000000u;
    int buf = 0;
    for (;;) {
        compute(buf);
        if (!hasN) break;
        store(buf ^ 1);                     // slot N's rows have arrived (hipcc's wait); the buffer was read a slot ago
        buf ^= 1;
        hasN = hasNN; tiN = tiNN; gN = gNN;
        if (hasN) {
            gather(tiN, wNN);
            hasNN = advance(tiNN, gNN);
            if (hasNN) wNN = load_word(tiNN, gNN);
        }
    }
#pragma unroll
    for (int m = 0; m < MI; ++m)
#pragma unroll
        for (int nn = 0; nn < NJ; ++nn)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = ci0 + 16 * m + 4 * q + r, co = co0 + 16 * nn + i;
                const float v = acc[m][nn][r] * a.out_scale;
                if (v != 0.f) atomicAdd(&a.dw[(int64_t)k * a.dw_kstride + (int64_t)ci * a.lddw + co], v);
            }
}

// Weight gradient of an identity map (1x1 layer) with FEW output channels -- the last layer of every head (96 -> 3 / 1 /
// 20 on the batch's segments): dW[ci][co] = sum_r x[r][ci] * dy[r][co] is a skinny reduction, not a GEMM.  The MFMA
// kernels pad the output channels to 16 and, with a row pitch below 16 floats, fell back to scalar addressing: 87-128 us
// per launch at 0.02-0.3 TFLOP/s (round 2).  Here a workgroup takes a chunk of rows, stages their dy in LDS, and thread
// (part, ci) keeps the CO sums of its input channel in registers over its share of the rows (x is read coalesced, the dy
// values are LDS broadcasts); one atomic per dW element and workgroup at the end.
#define WGN_ROWS 512
template <int CO>
__global__ __launch_bounds__(256) void wgrad_narrow_kernel(const float* __restrict__ x, int64_t ldx, int cin,
                                                           const float* __restrict__ dy, int64_t lddy, int cout, int64_t n,
                                                           float* __restrict__ dw, int64_t lddw) {
    __shared__ __attribute__((aligned(16))) float sdy[WGN_ROWS * CO];      // later reused for the partial sums of the row shares
    const int64_t r0 = (int64_t)blockIdx.x * WGN_ROWS;
    const int rows = (int)((n - r0) < WGN_ROWS ? (n - r0) : WGN_ROWS);
    for (int e = threadIdx.x; e < rows * CO; e += 256) {
        const int r = e / CO, co = e % CO;
        sdy[e] = co < cout ? dy[(r0 + r) * lddy + co] : 0.f;
    }
    __syncthreads();
    const int cw = cin < 256 ? cin : 256;                       // input channels per round
    const int nparts = 256 / cw;                                // row shares per input channel
    for (int ci0 = 0; ci0 < cin; ci0 += cw) {
        const int ci = ci0 + (int)threadIdx.x % cw, part = (int)threadIdx.x / cw;
        const bool active = ci < cin && part < nparts;
        float acc[CO];
#pragma unroll
        for (int co = 0; co < CO; ++co) acc[co] = 0.f;
        if (active) {
            const float* xp = x + r0 * ldx + ci;
            int r = part;
            for (; r + 3 * nparts < rows; r += 4 * nparts) {         // four rows' loads in flight
                float xv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) xv[u] = xp[(int64_t)(r + u * nparts) * ldx];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const f32x4* d4 = (const f32x4*)&sdy[(r + u * nparts) * CO];
#pragma unroll
                    for (int c4 = 0; c4 < CO / 4; ++c4) {
                        const f32x4 d = d4[c4];
#pragma unroll
                        for (int v = 0; v < 4; ++v) acc[4 * c4 + v] = __builtin_fmaf(xv[u], d[v], acc[4 * c4 + v]);
                    }
                }
            }
            for (; r < rows; r += nparts) {
                const float xv = xp[(int64_t)r * ldx];
#pragma unroll
                for (int co = 0; co < CO; ++co) acc[co] = __builtin_fmaf(xv, sdy[r * CO + co], acc[co]);
            }
        }
        // the row shares of one input channel meet in LDS; ONE atomic per dW element and workgroup
        __syncthreads();
        float* red = sdy;                                       // [part][cw][CO] <= 256 * CO floats <= WGN_ROWS * CO
        if (active) {
#pragma unroll
            for (int co = 0; co < CO; ++co) red[(part * cw + (ci - ci0)) * CO + co] = acc[co];
        }
        __syncthreads();
        for (int e = threadIdx.x; e < cw * CO; e += 256) {
            const int cl = e / CO, co = e % CO;
            if (ci0 + cl < cin && co < cout) {
                float s = 0.f;
                for (int p = 0; p < nparts; ++p) s += red[(p * cw + cl) * CO + co];
                if (s != 0.f) atomicAdd(&dw[(int64_t)(ci0 + cl) * lddw + co], s);
            }
        }
        if (ci0 + cw < cin) {                                   // (more than 256 input channels: dy is staged again)
            __syncthreads();
            for (int e = threadIdx.x; e < rows * CO; e += 256) {
                const int r = e / CO, co = e % CO;
                sdy[e] = co < cout ? dy[(r0 + r) * lddy + co] : 0.f;
            }
            __syncthreads();
        }
    }
}

template <int MI>
static void launch_wgrad_nj(int NJ, dim3 grid, hipStream_t st, const WgradArgs& a) {
    if (a.trh) {                     // half operands, complete blocks, 16-byte aligned rows: f16 MFMA
        if (a.swap) {
            switch (NJ) {
                case 1: conv_wgrad_trh_kernel<MI, 1, 1><<<grid, 256, 0, st>>>(a); break;
                case 2: conv_wgrad_trh_kernel<MI, 2, 1><<<grid, 256, 0, st>>>(a); break;
                case 3: conv_wgrad_trh_kernel<MI, 3, 1><<<grid, 256, 0, st>>>(a); break;
                default: conv_wgrad_trh_kernel<MI, 4, 1><<<grid, 256, 0, st>>>(a); break;
            }
        } else {
            switch (NJ) {
                case 1: conv_wgrad_trh_kernel<MI, 1, 0><<<grid, 256, 0, st>>>(a); break;
                case 2: conv_wgrad_trh_kernel<MI, 2, 0><<<grid, 256, 0, st>>>(a); break;
                case 3: conv_wgrad_trh_kernel<MI, 3, 0><<<grid, 256, 0, st>>>(a); break;
                default: conv_wgrad_trh_kernel<MI, 4, 0><<<grid, 256, 0, st>>>(a); break;
            }
        }
        return;
    }
    if (a.pipe && a.half) {          // half operands: the flat-pipeline kernel with hipcc-tracked loads, both row-role forms
        if (a.swap) {
            switch (NJ) {
                case 1: conv_wgrad_flow_h_kernel<MI, 1, 1><<<grid, 256, 0, st>>>(a); break;
                case 2: conv_wgrad_flow_h_kernel<MI, 2, 1><<<grid, 256, 0, st>>>(a); break;
                case 3: conv_wgrad_flow_h_kernel<MI, 3, 1><<<grid, 256, 0, st>>>(a); break;
                default: conv_wgrad_flow_h_kernel<MI, 4, 1><<<grid, 256, 0, st>>>(a); break;
            }
        } else {
            switch (NJ) {
                case 1: conv_wgrad_flow_h_kernel<MI, 1, 0><<<grid, 256, 0, st>>>(a); break;
                case 2: conv_wgrad_flow_h_kernel<MI, 2, 0><<<grid, 256, 0, st>>>(a); break;
                case 3: conv_wgrad_flow_h_kernel<MI, 3, 0><<<grid, 256, 0, st>>>(a); break;
                default: conv_wgrad_flow_h_kernel<MI, 4, 0><<<grid, 256, 0, st>>>(a); break;
            }
        }
        return;
    }
    if (a.pipe) {
        // hand-issued loads (B2M_WGRAD_HANDLOADS: 0 never, 1 the 48 x 48 and 64 x 64 blocks only, 2 every block of 2..4 x 2..4 sub-tiles)
        if constexpr (MI >= 2) {
            // (real rulebooks only: an identity map has every pair of a tile but the last one's -- nothing to mask, and the
            // A/B says so: 128 -> 96 on 1.2 M rows 94.0 against 95.3 TFLOP/s)
            if (a.rb_in && NJ >= 2 && (a.handloads >= 2 || (a.handloads == 1 && MI == NJ && MI >= 3))) {
                const size_t xl = (size_t)env_flag("B2M_WGRAD_LDS", 0);      // diagnostic: extra LDS per workgroup caps the resident waves
                if (a.swap) {
                    switch (NJ) {
                        case 2: conv_wgrad_flow_kernel<MI, 2, 1, 1><<<grid, 256, xl, st>>>(a); break;
                        case 3: conv_wgrad_flow_kernel<MI, 3, 1, 1><<<grid, 256, xl, st>>>(a); break;
                        default: conv_wgrad_flow_kernel<MI, 4, 1, 1><<<grid, 256, xl, st>>>(a); break;
                    }
                    return;
                }
                switch (NJ) {
                    case 2: conv_wgrad_flow_kernel<MI, 2, 1><<<grid, 256, xl, st>>>(a); break;
                    case 3: conv_wgrad_flow_kernel<MI, 3, 1><<<grid, 256, xl, st>>>(a); break;
                    default: conv_wgrad_flow_kernel<MI, 4, 1><<<grid, 256, xl, st>>>(a); break;
                }
                return;
            }
        }
        if (!a.swap) {          // (exchanged row roles exist in the hand-issued form and in the plain kernel below)
            switch (NJ) {
                case 1: conv_wgrad_flow_kernel<MI, 1><<<grid, 256, 0, st>>>(a); break;
                case 2: conv_wgrad_flow_kernel<MI, 2><<<grid, 256, 0, st>>>(a); break;
                case 3: conv_wgrad_flow_kernel<MI, 3><<<grid, 256, 0, st>>>(a); break;
                default: conv_wgrad_flow_kernel<MI, 4><<<grid, 256, 0, st>>>(a); break;
            }
            return;
        }
    }
    if (a.half) {
        switch (NJ) {
            case 1: conv_wgrad_kernel<MI, 1, 1><<<grid, 256, 0, st>>>(a); break;
            case 2: conv_wgrad_kernel<MI, 2, 1><<<grid, 256, 0, st>>>(a); break;
            case 3: conv_wgrad_kernel<MI, 3, 1><<<grid, 256, 0, st>>>(a); break;
            default: conv_wgrad_kernel<MI, 4, 1><<<grid, 256, 0, st>>>(a); break;
        }
        return;
    }
    switch (NJ) {
        case 1: conv_wgrad_kernel<MI, 1><<<grid, 256, 0, st>>>(a); break;
        case 2: conv_wgrad_kernel<MI, 2><<<grid, 256, 0, st>>>(a); break;
        case 3: conv_wgrad_kernel<MI, 3><<<grid, 256, 0, st>>>(a); break;
        default: conv_wgrad_kernel<MI, 4><<<grid, 256, 0, st>>>(a); break;
    }
}
static void launch_wgrad(int MI, int NJ, dim3 grid, hipStream_t st, const WgradArgs& a) {
    switch (MI) {
        case 1: launch_wgrad_nj<1>(NJ, grid, st, a); break;
        case 2: launch_wgrad_nj<2>(NJ, grid, st, a); break;
        case 3: launch_wgrad_nj<3>(NJ, grid, st, a); break;
        default: launch_wgrad_nj<4>(NJ, grid, st, a); break;
    }
}
// tuning switches (A/B inside one process: tools/bench_conv.py); read once per process (b2m_reload_env re-reads them)
static int env_flag(const char* name, int dflt) { return b2m_env_int(name, dflt); }
static int pick_blk(int c) {      // 16-column sub-tiles per wave block
    // 64 channels: four 32x32 blocks keep all 4 waves of a workgroup busy and fit the pipelined kernel at full
    // occupancy (+27..35 % over one 64x64 block)
    if (c == 64) return 2;
    if (c % 64 == 0) return 4;
    if (c % 48 == 0) return 3;
    if (c % 32 == 0) return 2;
    if (c <= 16) return 1;
    if (c <= 32) return 2;
    if (c <= 48) return 3;
    return 4;
}

// tile chunks of the deterministic path: few, so that the partial buffer stays small
#define B2M_WGRAD_DET_CHUNKS 32
extern "C" int64_t b2m_conv_wgrad_workspace(int32_t K, int32_t cin, int32_t cout) {
    return (int64_t)B2M_WGRAD_DET_CHUNKS * K * cin * cout;
}
static int conv_wgrad_impl(const float* x, int64_t ldx, int32_t cin, int64_t n_in, const float* dy, int64_t lddy,
                           int32_t cout, const int32_t* rb_in, const uint8_t* rb_out, const int32_t* rb_cnt,
                           int64_t n_out, int32_t K, float* dw, int64_t lddw, int64_t dw_kstride, float* workspace,
                           void* stream, int tr, int half = 0, float out_scale = 1.f);
extern "C" int b2m_conv_wgrad(const float* x, int64_t ldx, int32_t cin, int64_t n_in, const float* dy, int64_t lddy,
                              int32_t cout, const int32_t* rb_in, const uint8_t* rb_out, const int32_t* rb_cnt,
                              int64_t n_out, int32_t K, float* dw, int64_t lddw, int64_t dw_kstride, float* workspace,
                              void* stream) {
    return conv_wgrad_impl(x, ldx, cin, n_in, dy, lddy, cout, rb_in, rb_out, rb_cnt, n_out, K, dw, lddw, dw_kstride, workspace, stream, 0);
}
// The same reduction with the roles of the rulebook's two row numbers EXCHANGED: dW[k][ci][co] += sum x[tile*TILE + rb_out][ci] *
// dy[rb_in][co] -- x has n_out rows (the rulebook's tiles), dy n_in rows.  What it is for: the weight gradient of a TRANSPOSED
// k2s2 map over the map's DOWN rulebook (x = the layer's coarse input rows, dy = the gradient of its fine output rows): tiled over
// the coarse rows an offset of a tile has up to 64 pairs, tiled over the fine rows ~8 -- half-empty 16-pair slots, two k-steps each
// (35 TFLOP/s on the benchmark's level-1 / level-2 maps).
extern "C" int b2m_conv_wgrad_tr(const float* x, int64_t ldx, int32_t cin, int64_t n_in, const float* dy, int64_t lddy,
                                 int32_t cout, const int32_t* rb_in, const uint8_t* rb_out, const int32_t* rb_cnt,
                                 int64_t n_out, int32_t K, float* dw, int64_t lddw, int64_t dw_kstride, float* workspace,
                                 void* stream) {
    return conv_wgrad_impl(x, ldx, cin, n_in, dy, lddy, cout, rb_in, rb_out, rb_cnt, n_out, K, dw, lddw, dw_kstride, workspace, stream, 1);
}
static int conv_wgrad_impl(const float* x, int64_t ldx, int32_t cin, int64_t n_in, const float* dy, int64_t lddy,
                           int32_t cout, const int32_t* rb_in, const uint8_t* rb_out, const int32_t* rb_cnt,
                           int64_t n_out, int32_t K, float* dw, int64_t lddw, int64_t dw_kstride, float* workspace,
                           void* stream, int tr, int half, float out_scale) {
    hipStream_t st = (hipStream_t)stream;
    B2M_CHECK_ARG(x && dy && dw && cin > 0 && cout > 0 && K >= 1 && K <= 65535 && n_in >= 0, "bad pointers/sizes");
    B2M_CHECK_ARG((rb_in == nullptr) == (rb_out == nullptr) && (rb_in == nullptr) == (rb_cnt == nullptr),
                  "rulebook pointers must be all set or all NULL");
    B2M_CHECK_ARG(rb_in != nullptr || K == 1, "identity rulebook needs K == 1");
    B2M_CHECK_ARG(!tr || rb_in != nullptr, "b2m_conv_wgrad_tr needs a rulebook");
    B2M_CHECK_ARG(ldx >= cin && lddy >= cout && lddw >= cout && dw_kstride >= (int64_t)cin * lddw,
                  "leading dimension too small");
    if (n_out == 0) return B2M_OK;
    WgradArgs a{};
    a.swap = tr;
    a.half = half;
    a.out_scale = out_scale;
    const int esz = half ? 2 : 4;
    a.x = x; a.ldx = ldx; a.cin = cin; a.dy = dy; a.lddy = lddy; a.cout = cout;
    a.rb_in = rb_in; a.rb_out = rb_out; a.rb_cnt = rb_cnt;
    a.n_out = n_out; a.ntiles = cdiv64(n_out, B2M_TILE); a.K = K; a.dw = dw; a.lddw = lddw; a.dw_kstride = dw_kstride;
    static const float* zeros_addr = nullptr;
    if (!zeros_addr) B2M_HIP(hipGetSymbolAddress((void**)&zeros_addr, HIP_SYMBOL(g_zeros)));
    a.zeros = zeros_addr;
    // 1x1 layer with few output channels (the heads' last layers): a plain reduction, see wgrad_narrow_kernel
    if (!half && !tr && rb_in == nullptr && cout <= 32 && cout % 16 != 0 && cin <= 1024 && !workspace && n_in >= n_out && env_flag("B2M_WGRAD_NARROW", 1)) {
        const unsigned g = (unsigned)cdiv64(n_out, WGN_ROWS);
        if (cout <= 4) wgrad_narrow_kernel<4><<<g, 256, 0, st>>>(x, ldx, cin, dy, lddy, cout, n_out, dw, lddw);
        else if (cout <= 8) wgrad_narrow_kernel<8><<<g, 256, 0, st>>>(x, ldx, cin, dy, lddy, cout, n_out, dw, lddw);
        else if (cout <= 16) wgrad_narrow_kernel<16><<<g, 256, 0, st>>>(x, ldx, cin, dy, lddy, cout, n_out, dw, lddw);
        else wgrad_narrow_kernel<32><<<g, 256, 0, st>>>(x, ldx, cin, dy, lddy, cout, n_out, dw, lddw);
        B2M_LAUNCH_CHECK();
        return B2M_OK;
    }
    const int MI = pick_blk(cin), NJ = pick_blk(cout);
    a.nmb = (cin + 16 * MI - 1) / (16 * MI);
    a.nnb = (cout + 16 * NJ - 1) / (16 * NJ);
    // chunking: enough waves to fill the chip several times over, but a wave should own at least 4-8 tiles: every wave
    // ends with 16*MI x 16*NJ atomics into dW, and on the small deep-level maps those outweighed the MFMA work (A/B per
    // layer: +30..50 % there with the floor).  Large maps (>= 2048 tiles: levels 0 and 1 of the benchmark) are cut twice
    // as fine, into chunks of at most 32 tiles: a wave's life is its chunk's pairs of one offset, and with 64-tile chunks
    // the launch ended on a long tail of them (level-0 96->96: 96 -> 100 TFLOP/s, level-1: 85 -> 91, 64->64: 69 -> 76;
    // the same cut on the 1 k-tile level-2 maps LOSES 20 %: there the atomics of twice as many waves weigh more).
    const int64_t blocks_per_chunk = (int64_t)K * a.nmb * a.nnb;
    const bool large = a.ntiles >= 2048;
    int64_t want_chunks = cdiv64(large ? 32768 : 16384, blocks_per_chunk);
    if (want_chunks < 1) want_chunks = 1;
    int64_t tpc = cdiv64(a.ntiles, want_chunks);
    // (floor 8 from 32 tiles up: level-3 128->128 58 -> 66, level-4 256->256 58 -> 67 TFLOP/s; the 11-tile level-5 maps lose
    // a quarter with it and keep 4)
    // (identity maps of a few hundred tiles -- the heads' 96 -> 96 layers on ~10 k segments: 8-tile chunks left 20 workgroups
    // on 256 CUs, 40 us per launch; one block's atomics per 2 tiles are nothing against that)
    const int min_tiles = env_flag("B2M_WGRAD_MIN_TILES", (rb_in == nullptr && a.ntiles <= 1024) ? 2 : a.ntiles >= 32 ? 8 : 4);
    if (tpc < min_tiles) tpc = min_tiles;
    { const int mx = large ? 32 : 64; if (tpc > mx) tpc = mx; }
    // (half operands on the f16 MFMA, conv_wgrad_trh_kernel: a slot costs a quarter of its fp32 MFMA time, so a wave's fixed costs
    // -- the MI NJ x 256 atomics at its end above all -- weigh four times as much: chunks of 16 tiles wherever that still leaves
    // 500 workgroups.  Level 1 128->128 124 -> 209 TFLOP/s, level 1 96->96 121 -> 170, level 2 128->128 115 -> 166, level 3
    // 256->256 125 -> 158; with fewer workgroups (level 3 128->128: 270) 95 -> 71, and the transposed maps lose: both keep 8.)
    if (half && !tr && rb_in != nullptr && !workspace && tpc < 16 && b2m_env_int("B2M_WGRAD_MIN_TILES", -1) < 0) {
        const int nblk_ = a.nmb * a.nnb;
        const int kp_ = (nblk_ <= 2 && K >= 4 && env_flag("B2M_WGRAD_KPACK", 1)) ? 4 / nblk_ : 1;
        const int64_t wgs16 = (int64_t)((K + kp_ - 1) / kp_) * ((nblk_ + 3) / 4) * cdiv64(a.ntiles, 16);
        if (wgs16 >= 500 && env_flag("B2M_WGRAD_TRH", 1)) tpc = 16;
    }
    // Deterministic mode (workspace given): at most B2M_WGRAD_DET_CHUNKS tile chunks, every chunk stores its partial
    // blocks plainly and a second kernel adds them up in chunk order -- no atomics, the same bits on every run.
    a.partial = workspace;
    if (workspace) tpc = cdiv64(a.ntiles, B2M_WGRAD_DET_CHUNKS);
    a.tiles_per_chunk = (int)tpc;
    a.nz = (a.nmb * a.nnb + 3) / 4;
    // layers with one or two blocks (32->32, the 6->32 stem, 32->96): 4 or 2 offsets per workgroup instead of idle waves
    const int nblocks = a.nmb * a.nnb;
    a.kpack = (nblocks <= 2 && K >= 4 && env_flag("B2M_WGRAD_KPACK", 1)) ? 4 / nblocks : 1;
    a.kgroups = (K + a.kpack - 1) / a.kpack;
    a.nwg = (int64_t)a.kgroups * cdiv64(a.ntiles, tpc) * a.nz;
    B2M_CHECK_ARG(a.nwg < (1ll << 31) - 8, "too many workgroups");
    // work item = (k fastest, block group, tile chunk): a chunk of the XCD order = all offsets and blocks of
    // tile chunks of one eighth (contiguous)
    const XcdOrder xo = xcd_order(a.nwg, env_flag("B2M_XCD", 1) ? (int64_t)(1 << 20) * a.kgroups * a.nz : 0);
    a.xcd_per = xo.chunk;
    dim3 grid(xo.grid);
    // XCD runs of equal work (the tail of rb_cnt, b2m_rulebook_balance), each cut into tile chunks from its own start
    a.xcd_start = nullptr;
    if (rb_cnt && !workspace && a.ntiles >= B2M_BALANCE_MIN_TILES && env_flag("B2M_XCD", 1) && env_flag("B2M_XCD_BALANCE", 1)) {
        a.xcd_start = rb_cnt + (int64_t)K * a.ntiles;
        grid = dim3((unsigned)(8 * a.kgroups * a.nz * cdiv64(B2M_XCD_CAP(a.ntiles), tpc)));
    }
    // 24-bit multiply operands and 32-bit byte offsets: rows < 2^24, row pitch < 2^22 floats, tensors < 4 GiB
    // (blocks may overhang cin/cout as long as the row PITCH covers them: the extra columns only feed dW rows /
    // columns that are never written)
    // (b2m_conv_wgrad_tr: x has the n_out rows of the tiles, dy the n_in rows the pair lists name)
    const int64_t nx = tr ? n_out : n_in, ny = tr ? n_in : n_out;
    a.fast32 = (ldx >= (int64_t)a.nmb * 16 * MI && lddy >= (int64_t)a.nnb * 16 * NJ && n_out < (1 << 24) && n_in < (1 << 24) &&
                ldx < (1 << 22) && lddy < (1 << 22) && ny * lddy * esz < (1ll << 32) && nx * ldx * esz < (1ll << 32) &&
                env_flag("B2M_WGRAD_FAST32", 1)) ? 1 : 0;
    // The flat-pipeline kernel for real rulebooks and 32-bit addressable operands.  Its MFMAs are asm statements the
    // compiler's hazard recogniser cannot see: a block with a single accumulator (MI = NJ = 1: consecutive MFMAs on the
    // same registers) stays on the plain kernel, where the builtin lets hipcc place whatever the dependence needs.
    a.pipe = (a.fast32 && (rb_in != nullptr || (n_in >= n_out && env_flag("B2M_WGRAD_PIPE_IDENT", 1))) && MI * NJ >= 2 &&
              !workspace && env_flag("B2M_WGRAD_PIPE", 1)) ? 1 : 0;
    // (diagnostics, tools/debug_wgrad_h22.py: half operands, blocks of at most four sub-tiles on the plain kernel)
    if (half && MI * NJ <= 4 && env_flag("B2M_WGRAD_H_PIPE_SMALL", 1) == 0) a.pipe = 0;
    a.handloads = env_flag("B2M_WGRAD_HANDLOADS", 2);
    // half operands: the f16-MFMA kernel wherever its 16-byte row chunks exist (complete blocks, aligned rows, a real rulebook)
    a.trh = (half && a.fast32 && rb_in != nullptr && !workspace && cin % (16 * MI) == 0 && cout % (16 * NJ) == 0 &&
             ((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0 && ldx % 8 == 0 && lddy % 8 == 0 && env_flag("B2M_WGRAD_TRH", 1)) ? 1 : 0;
    launch_wgrad(MI, NJ, grid, st, a);
    if (workspace) {
        const int nchunks = (int)cdiv64(a.ntiles, tpc);
        const int64_t per = (int64_t)K * cin * cout;
        wgrad_reduce_kernel<<<(unsigned)(cdiv64(per, 256) > 4096 ? 4096 : cdiv64(per, 256)), 256, 0, st>>>(
            workspace, nchunks, K, cin, cout, dw, lddw, dw_kstride);
    }
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}
// Half-precision training: the same reduction with x and dy stored as IEEE binary16 (pitches in elements), dW accumulated in
// fp32 with atomics (no deterministic form).  tr != 0: the rulebook's row roles exchanged, as b2m_conv_wgrad_tr.
extern "C" int b2m_conv_wgrad_h(const void* x, int64_t ldx, int32_t cin, int64_t n_in, const void* dy, int64_t lddy, int32_t cout,
                                const int32_t* rb_in, const uint8_t* rb_out, const int32_t* rb_cnt, int64_t n_out, int32_t K,
                                float* dw, int64_t lddw, int64_t dw_kstride, int32_t tr, float out_scale, void* stream) {
    B2M_CHECK_ARG(((uintptr_t)x % 2) == 0 && ((uintptr_t)dy % 2) == 0, "x / dy must be 2-byte aligned");
    return conv_wgrad_impl((const float*)x, ldx, cin, n_in, (const float*)dy, lddy, cout, rb_in, rb_out, rb_cnt, n_out, K, dw, lddw,
                           dw_kstride, nullptr, stream, tr ? 1 : 0, 1, out_scale);
}
