// Coordinate hashing, strided coordinate generation, kernel maps and the tile rulebook.
// All integer work, HBM/L2-latency bound; results are bit-exact against oracle/sparse_ref.py.
#include "b2m_common.h"
#include <cstdlib>
#include <stdarg.h>
#include <string.h>

static thread_local char g_err[512] = "";
void b2m_set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
extern "C" const char* b2m_last_error(void) { return g_err; }

// Tuning switches (B2M_* environment variables): read ONCE per process and kept in a small table -- a training step makes
// ~300 convolution calls that each consult ~25 switches, and getenv() walks the whole environment every time (round 2:
// ~4 ms of host time per step).  b2m_reload_env() drops the table (tests and A/B tools that flip a switch in-process).
#include <mutex>
namespace {
struct EnvEntry { const char* name; int value; };
EnvEntry g_env[96];
int g_env_n = 0;
std::mutex g_env_mu;
}
int b2m_env_int(const char* name, int dflt) {
    std::lock_guard<std::mutex> lock(g_env_mu);
    for (int i = 0; i < g_env_n; ++i)
        if (g_env[i].name == name || strcmp(g_env[i].name, name) == 0) return g_env[i].value == INT32_MIN ? dflt : g_env[i].value;
    const char* e = getenv(name);
    const int v = e ? atoi(e) : INT32_MIN;          // INT32_MIN = unset (the caller's default applies, it may differ per call)
    if (g_env_n < (int)(sizeof(g_env) / sizeof(g_env[0]))) { g_env[g_env_n].name = name; g_env[g_env_n].value = v; ++g_env_n; }
    return v == INT32_MIN ? dflt : v;
}
extern "C" int b2m_reload_env(void) {
    std::lock_guard<std::mutex> lock(g_env_mu);
    g_env_n = 0;
    return 0;
}
extern "C" int b2m_version(void) { return 1; }
extern "C" int b2m_device_ok(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return 0;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) != hipSuccess) return 0;
    return strncmp(p.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}

// ------------------------------------------------------------------ exclusive scan (int32)
// 1024 items per block of 256 threads; three launches; deterministic.
#define SCAN_ITEMS 1024
__device__ __forceinline__ int wave_incl_scan(int v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(v, d, 64);
        if (lane_id() >= d) v += t;
    }
    return v;
}
// returns exclusive prefix of v within the 256-thread block, *total = block sum
__device__ __forceinline__ int block_excl_scan(int v, int* total) {
    __shared__ int wsum[4];
    int incl = wave_incl_scan(v);
    int w = threadIdx.x >> 6;
    if (lane_id() == 63) wsum[w] = incl;
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { int s = wsum[i]; if (i < w) off += s; tot += s; }
    __syncthreads();
    *total = tot;
    return off + incl - v;
}
__global__ void scan_reduce_kernel(const int* __restrict__ in, int64_t n, int* __restrict__ bsum) {
    int64_t base = (int64_t)blockIdx.x * SCAN_ITEMS + threadIdx.x * 4;
    int s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) if (base + i < n) s += in[base + i];
    int tot; block_excl_scan(s, &tot);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}
__global__ void scan_bsum_kernel(int* __restrict__ bsum, int64_t nb, int* __restrict__ total) {
    int carry = 0;
    for (int64_t b0 = 0; b0 < nb; b0 += 256) {
        int64_t i = b0 + threadIdx.x;
        int v = i < nb ? bsum[i] : 0;
        int tot; int ex = block_excl_scan(v, &tot);
        if (i < nb) bsum[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) *total = carry;
}
__global__ void scan_apply_kernel(const int* __restrict__ in, int64_t n, const int* __restrict__ bsum,
                                  int* __restrict__ out) {
    int64_t base = (int64_t)blockIdx.x * SCAN_ITEMS + threadIdx.x * 4;
    int v[4]; int s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = base + i < n ? in[base + i] : 0; s += v[i]; }
    int tot; int ex = block_excl_scan(s, &tot) + bsum[blockIdx.x];
#pragma unroll
    for (int i = 0; i < 4; ++i) { if (base + i < n) out[base + i] = ex; ex += v[i]; }
}
// out may alias in.  bsum: int32[cdiv(n,1024)+1]; the last element receives the total.
int b2m_scan_excl(const int* in, int64_t n, int* out, int* bsum, hipStream_t st) {
    int64_t nb = cdiv64(n, SCAN_ITEMS);
    if (nb == 0) { return hipMemsetAsync(bsum, 0, sizeof(int), st) == hipSuccess ? 0 : B2M_ERR_HIP; }
    scan_reduce_kernel<<<dim3((unsigned)nb), 256, 0, st>>>(in, n, bsum);
    scan_bsum_kernel<<<1, 256, 0, st>>>(bsum, nb, bsum + nb);
    scan_apply_kernel<<<dim3((unsigned)nb), 256, 0, st>>>(in, n, bsum, out);
    return 0;
}

// ------------------------------------------------------------------ hash build
__global__ void hash_insert_kernel(const int32_t* __restrict__ coords, int64_t n, uint64_t* __restrict__ keys,
                                   int32_t* __restrict__ vals, int64_t mask, int32_t shift_mask,
                                   int32_t* __restrict__ slot_of) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    i32x4 c = *(const i32x4*)(coords + i * 4);
    // shift_mask = ~(2*ts-1) for strided insertion, ~0 for plain insertion
    uint64_t key = b2m_pack(c.x, c.y & shift_mask, c.z & shift_mask, c.w & shift_mask);
    int64_t s = (int64_t)(b2m_hash(key) & (uint64_t)mask);
    for (;;) {
        unsigned long long prev = atomicCAS((unsigned long long*)&keys[s], (unsigned long long)B2M_EMPTY_KEY,
                                            (unsigned long long)key);
        if (prev == B2M_EMPTY_KEY || prev == key) break;
        s = (s + 1) & mask;
    }
    atomicMin(&vals[s], (int32_t)i);
    if (slot_of) slot_of[i] = (int32_t)s;
}
__global__ void count_dups_kernel(const int32_t* __restrict__ coords, int64_t n, const uint64_t* __restrict__ keys,
                                  const int32_t* __restrict__ vals, int64_t mask, int32_t* __restrict__ dup) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    i32x4 c = *(const i32x4*)(coords + i * 4);
    int64_t s = b2m_find(keys, mask, b2m_pack(c.x, c.y, c.z, c.w));
    if (s < 0 || vals[s] != (int32_t)i) atomicAdd(dup, 1);
}

static bool is_pow2(int64_t v) { return v > 0 && (v & (v - 1)) == 0; }

// Two buffers filled with a 32-bit pattern each in ONE launch (the empty markers of a hash table's keys and values, the two
// k2s2 tables): the runtime's memset is a launch per buffer, and a batch's 8 coordinate maps + 7 strided maps paid 30 of them.
// Both byte counts are multiples of 4; 16-byte stores where the address allows.
__global__ void fill_pair_kernel(uint32_t* __restrict__ a, int64_t na, uint32_t pa, uint32_t* __restrict__ b, int64_t nb, uint32_t pb) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int pass = 0; pass < 2; ++pass) {
        uint32_t* p = pass ? b : a;
        const int64_t n = pass ? nb : na;
        const uint32_t v = pass ? pb : pa;
        if (!p || n <= 0) continue;
        if (((uintptr_t)p & 15) == 0) {
            const int64_t n4 = n / 4;
            const uint4 v4 = make_uint4(v, v, v, v);
            for (int64_t i = t; i < n4; i += stride) ((uint4*)p)[i] = v4;
            for (int64_t i = n4 * 4 + t; i < n; i += stride) p[i] = v;
        } else {
            for (int64_t i = t; i < n; i += stride) p[i] = v;
        }
    }
}
static int fill_pair(void* a, int64_t bytes_a, uint32_t pa, void* b, int64_t bytes_b, uint32_t pb, hipStream_t st) {
    const int64_t words = (bytes_a > bytes_b ? bytes_a : bytes_b) / 16 + 1;
    int64_t nb = cdiv64(words, 256);
    if (nb > 2048) nb = 2048;
    fill_pair_kernel<<<(unsigned)nb, 256, 0, st>>>((uint32_t*)a, bytes_a / 4, pa, (uint32_t*)b, bytes_b / 4, pb);
    return hipGetLastError() == hipSuccess ? 0 : B2M_ERR_HIP;
}

extern "C" int b2m_coords_build(const int32_t* coords, int64_t n, uint64_t* keys, int32_t* vals, int64_t cap,
                                int32_t* dup_count, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    B2M_CHECK_ARG(n >= 0 && n < (1ll << 31), "n out of range");
    B2M_CHECK_ARG(is_pow2(cap) && cap >= 2 * n && cap < (1ll << 31), "cap must be a power of two >= 2n");
    if (fill_pair(keys, cap * (int64_t)sizeof(uint64_t), 0xFFFFFFFFu, vals, cap * (int64_t)sizeof(int32_t), 0x7F7F7F7Fu, st)) return B2M_ERR_HIP;
    if (dup_count) B2M_HIP(hipMemsetAsync(dup_count, 0, sizeof(int32_t), st));
    if (n == 0) return B2M_OK;
    unsigned nb = (unsigned)cdiv64(n, 256);
    hash_insert_kernel<<<nb, 256, 0, st>>>(coords, n, keys, vals, cap - 1, ~0, nullptr);
    B2M_LAUNCH_CHECK();
    if (dup_count) {
        count_dups_kernel<<<nb, 256, 0, st>>>(coords, n, keys, vals, cap - 1, dup_count);
        B2M_LAUNCH_CHECK();
    }
    return B2M_OK;
}

// ------------------------------------------------------------------ strided coordinates
__global__ void stride_flag_kernel(const int32_t* __restrict__ vals, const int32_t* __restrict__ slot_of, int64_t n,
                                   int32_t* __restrict__ flag) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = vals[slot_of[i]] == (int32_t)i ? 1 : 0;
}
// first rows write the coarse coordinate and publish their coarse row id into the table.
// In-place update of vals is safe: a slot holds the first fine row f before and pos[f] <= f after;
// every other row i of that slot has i > f, so neither value can equal i.
__global__ void stride_emit_kernel(const int32_t* __restrict__ coords, int64_t n, int32_t sm,
                                   const int32_t* __restrict__ slot_of, const int32_t* __restrict__ pos,
                                   int32_t* vals, int32_t* __restrict__ coords_out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int s = slot_of[i];
    if (vals[s] != (int32_t)i) return;       // not the first occurrence
    int r = pos[i];
    i32x4 c = *(const i32x4*)(coords + i * 4);
    i32x4 o; o.x = c.x; o.y = c.y & sm; o.z = c.z & sm; o.w = c.w & sm;
    *(i32x4*)(coords_out + (int64_t)r * 4) = o;
    vals[s] = r;
}
__global__ void stride_parent_kernel(const int32_t* __restrict__ coords, int64_t n, int32_t ts,
                                     const int32_t* __restrict__ slot_of, const int32_t* __restrict__ newvals,
                                     int32_t* __restrict__ parent, int32_t* __restrict__ koff) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    parent[i] = newvals[slot_of[i]];
    i32x4 c = *(const i32x4*)(coords + i * 4);
    int ox = (c.y & ts) ? 1 : 0, oy = (c.z & ts) ? 1 : 0, oz = (c.w & ts) ? 1 : 0;
    koff[i] = ox + 2 * oy + 4 * oz;
}
extern "C" int b2m_coords_stride(const int32_t* coords, int64_t n, int32_t ts, int32_t* coords_out, int32_t* parent,
                                 int32_t* koff, uint64_t* keys, int32_t* vals, int64_t cap, int32_t* scratch,
                                 int64_t* n_out_host, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    B2M_CHECK_ARG(n >= 0 && n < (1ll << 31), "n out of range");
    B2M_CHECK_ARG(ts > 0 && (ts & (ts - 1)) == 0, "ts must be a power of two");
    B2M_CHECK_ARG(is_pow2(cap) && cap >= 2 * n && cap < (1ll << 31), "cap must be a power of two >= 2n");
    B2M_CHECK_ARG(n_out_host != nullptr, "n_out_host is NULL");
    *n_out_host = 0;
    if (fill_pair(keys, cap * (int64_t)sizeof(uint64_t), 0xFFFFFFFFu, vals, cap * (int64_t)sizeof(int32_t), 0x7F7F7F7Fu, st)) return B2M_ERR_HIP;
    if (n == 0) return B2M_OK;
    int32_t* slot_of = scratch;            // [n]
    int32_t* pos = scratch + n;            // [n]  first-occurrence flag, then its exclusive scan in place
    int32_t* bsum = scratch + 2 * n;       // [cdiv(n,1024)+1]
    int64_t nbs = cdiv64(n, SCAN_ITEMS);
    const int32_t sm = ~(2 * ts - 1);
    unsigned nb = (unsigned)cdiv64(n, 256);
    hash_insert_kernel<<<nb, 256, 0, st>>>(coords, n, keys, vals, cap - 1, sm, slot_of);
    stride_flag_kernel<<<nb, 256, 0, st>>>(vals, slot_of, n, pos);
    int rc = b2m_scan_excl(pos, n, pos, bsum, st);
    if (rc) return rc;
    stride_emit_kernel<<<nb, 256, 0, st>>>(coords, n, sm, slot_of, pos, vals, coords_out);
    stride_parent_kernel<<<nb, 256, 0, st>>>(coords, n, ts, slot_of, vals, parent, koff);
    B2M_LAUNCH_CHECK();
    int32_t total = 0;
    B2M_HIP(hipMemcpyAsync(&total, bsum + nbs, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    B2M_HIP(hipStreamSynchronize(st));
    *n_out_host = total;
    return B2M_OK;
}

// ------------------------------------------------------------------ stride-1 kernel map (neighbour table)
// Optional occupancy bitmap (level 0): bit ((b*Z + z)*Y + y)*X + x is set for every voxel, so the ksize probes of
// one (dy,dz) line are ksize consecutive bits of one or two words.  Three out of four probes of the 5x5x5 stem
// kernel miss; with the bitmap a miss costs a bit test in a few MB of L2-resident words instead of a walk through
// the 48 MB hash table (PMC: 2.8 GB of L2-miss traffic per launch without it).
struct OccDims { int32_t X, Y, Z; };
__global__ void occupancy_kernel(const int32_t* __restrict__ coords, int64_t n, OccDims d,
                                 unsigned long long* __restrict__ bits) {
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= n) return;
    const i32x4 c = *(const i32x4*)(coords + o * 4);
    const int64_t b = (((int64_t)c.x * d.Z + c.w) * d.Y + c.z) * d.X + c.y;
    atomicOr(&bits[b >> 6], 1ull << (b & 63));
}
extern "C" int b2m_occupancy(const int32_t* coords, int64_t n, int32_t batches, int32_t dim_x, int32_t dim_y,
                             int32_t dim_z, uint64_t* bits, int64_t words, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    B2M_CHECK_ARG(coords && bits && n >= 0 && batches > 0 && dim_x > 0 && dim_y > 0 && dim_z > 0, "bad arguments");
    B2M_CHECK_ARG(words >= cdiv64((int64_t)batches * dim_x * dim_y * dim_z, 64) + 1, "bitmap too small (needs one spare word)");
    B2M_HIP(hipMemsetAsync(bits, 0, (size_t)words * sizeof(uint64_t), st));
    if (n > 0) occupancy_kernel<<<(unsigned)cdiv64(n, 256), 256, 0, st>>>(coords, n, OccDims{dim_x, dim_y, dim_z},
                                                                         (unsigned long long*)bits);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}

template <bool OCC>
__global__ void kernel_map_kernel(const int32_t* __restrict__ coords, int64_t n, int32_t ksize, int32_t ts,
                                  const uint64_t* __restrict__ keys, const int32_t* __restrict__ vals, int64_t mask,
                                  const uint64_t* __restrict__ occ, OccDims d, int32_t* __restrict__ nbr, int64_t ld) {
    int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= n) return;
    const int h = ksize / 2;
    // blockIdx.y selects one (dy,dz) line of the kernel: ksize independent probes per thread
    const int dy = ((int)blockIdx.y % ksize - h) * ts, dz = ((int)blockIdx.y / ksize - h) * ts;
    i32x4 c = *(const i32x4*)(coords + o * 4);
    const int y = c.z + dy, z = c.w + dz;
    bool yz = (unsigned)y < 65536u && (unsigned)z < 65536u;
    uint32_t line = ~0u;                       // bit kx: the voxel (x - h + kx, y, z) may exist
    if (OCC) {
        yz = yz && y < d.Y && z < d.Z;
        if (yz) {
            // bits x-h .. x+h of the line; positions left of 0 / right of X-1 are masked (they belong to other lines)
            const int64_t b0 = (((int64_t)c.x * d.Z + z) * d.Y + y) * d.X + (c.y - h);
            const int64_t bb = b0 < 0 ? 0 : b0;
            const int sh = (int)(bb & 63);
            uint64_t w = occ[bb >> 6] >> sh;
            if (sh) w |= occ[(bb >> 6) + 1] << (64 - sh);
            if (b0 < 0) w <<= (int)(-b0);
            line = (uint32_t)w;
            for (int kx = 0; kx < ksize; ++kx) {
                const int x = c.y - h + kx;
                if (x < 0 || x >= d.X) line &= ~(1u << kx);
            }
        }
    }
    for (int kx = 0; kx < ksize; ++kx) {
        const int x = c.y + (kx - h) * ts;
        int r = -1;
        if (yz && (unsigned)x < 65536u && ((line >> kx) & 1u)) {
            int64_t s = b2m_find(keys, mask, b2m_pack(c.x, x, y, z));
            if (s >= 0) r = vals[s];
        }
        nbr[(int64_t)((int)blockIdx.y * ksize + kx) * ld + o] = r;
    }
}
extern "C" int b2m_kernel_map(const int32_t* coords, int64_t n, int32_t ksize, int32_t ts, const uint64_t* keys,
                              const int32_t* vals, int64_t cap, const uint64_t* occ, int32_t dim_x, int32_t dim_y,
                              int32_t dim_z, int32_t* nbr, int64_t ld, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    B2M_CHECK_ARG(ksize == 1 || ksize == 3 || ksize == 5 || ksize == 7, "ksize must be odd (1,3,5,7)");
    B2M_CHECK_ARG(ld >= n && is_pow2(cap), "ld < n or cap not pow2");
    B2M_CHECK_ARG(occ == nullptr || (ts == 1 && dim_x > 0 && dim_y > 0 && dim_z > 0), "the occupancy bitmap is for stride 1");
    if (n == 0) return B2M_OK;
    const dim3 grid((unsigned)cdiv64(n, 256), (unsigned)(ksize * ksize));
    const OccDims d{dim_x, dim_y, dim_z};
    if (occ) kernel_map_kernel<true><<<grid, 256, 0, st>>>(coords, n, ksize, ts, keys, vals, cap - 1, occ, d, nbr, ld);
    else kernel_map_kernel<false><<<grid, 256, 0, st>>>(coords, n, ksize, ts, keys, vals, cap - 1, occ, d, nbr, ld);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}

// ------------------------------------------------------------------ k2s2 tables
__global__ void stride_tables_kernel(const int32_t* __restrict__ parent, const int32_t* __restrict__ koff, int64_t n,
                                     int32_t* __restrict__ child, int64_t ld_c, int32_t* __restrict__ up, int64_t ld_f) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int p = parent[i], k = koff[i];
    if (child) child[(int64_t)k * ld_c + p] = (int32_t)i;
    if (up) up[(int64_t)k * ld_f + i] = p;
}
extern "C" int b2m_stride_tables(const int32_t* parent, const int32_t* koff, int64_t n_fine, int64_t n_coarse,
                                 int32_t* child, int64_t ld_c, int32_t* up, int64_t ld_f, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    B2M_CHECK_ARG((!child || ld_c >= n_coarse) && (!up || ld_f >= n_fine), "leading dimension too small");
    if (child || up)
        if (fill_pair(child, child ? 8 * ld_c * (int64_t)sizeof(int32_t) : 0, 0xFFFFFFFFu, up, up ? 8 * ld_f * (int64_t)sizeof(int32_t) : 0, 0xFFFFFFFFu, st))
            return B2M_ERR_HIP;
    if (n_fine == 0) return B2M_OK;
    stride_tables_kernel<<<(unsigned)cdiv64(n_fine, 256), 256, 0, st>>>(parent, koff, n_fine, child, ld_c, up, ld_f);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}

// ------------------------------------------------------------------ XCD work boundaries of a rulebook
// The convolution kernels give every XCD one CONTIGUOUS run of tiles (its L2 then holds the rows neighbouring tiles
// gather in common).  Work per tile varies over a scene (surface tiles have 9..15 active offsets, interior ones 27): with
// equal tile counts the eight runs differ by 5..10 % in work and the launch ends when the heaviest XCD does.  The
// boundaries are therefore set by WORK: tile cost = sum over the active offsets of (3 x row groups of 16 pairs + 1), the
// MFMA steps plus the per-offset overhead of conv_fwd_flow_kernel; the weight-gradient kernel's k-steps follow the same
// pair counts.  Tail of rb_cnt (b2m.h): [K*ntiles + 0..8] = first tile of XCD 0..7 and ntiles; [K*ntiles + 16 + t] =
// cost of tile t (scratch).  No run is longer than ceil(1.25 * ntiles / 8) tiles (B2M_XCD_CAP): the launch grids are
// sized for that without a host read of the boundaries.
__global__ __launch_bounds__(256) void rulebook_cost_kernel(int32_t* __restrict__ rb_cnt, int32_t K, int64_t ntiles) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntiles) return;
    int cost = 0;
    for (int k = 0; k < K; ++k) {
        const int c = rb_cnt[(int64_t)k * ntiles + t];
        if (c > 0) cost += 3 * ((c + 15) >> 4) + 1;
    }
    rb_cnt[(int64_t)K * ntiles + 16 + t] = cost;
}
// (every block of rulebook_order_kernel runs this redundantly: no second launch, no grid-wide barrier)
__device__ __forceinline__ void rulebook_runs(const int32_t* __restrict__ cost, int64_t ntiles, long long* part, int* start) {
    const int tid = threadIdx.x;
    const int64_t per = (ntiles + 1023) / 1024;
    const int64_t lo = tid * per < ntiles ? tid * per : ntiles;
    const int64_t hi = lo + per < ntiles ? lo + per : ntiles;
    long long s = 0;
    for (int64_t t = lo; t < hi; ++t) s += cost[t];
    part[tid] = s;
    if (tid < 9) start[tid] = tid == 8 ? (int)ntiles : (int)(ntiles * tid / 8);     // (no work at all: equal runs)
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {                   // inclusive scan of the 1024 chunk sums
        const long long v = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    const long long total = part[1023];
    long long run = part[tid] - s;                         // work in front of this thread's tiles
    if (total > 0) {
        for (int64_t t = lo; t < hi; ++t) {
            const long long before = run;
            run += cost[t];
            // XCD x begins behind the tile that carries the prefix over x/8 of the total
#pragma unroll
            for (int x = 1; x < 8; ++x) {
                const long long target = (total * x + 7) / 8;
                if (before < target && run >= target) start[x] = (int)(t + 1);
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        const int cap = (int)((ntiles * 5 + 31) / 32);     // B2M_XCD_CAP: 1.25 x an eighth, rounded up
        for (int x = 1; x < 8; ++x) {
            if (start[x] < start[x - 1]) start[x] = start[x - 1];
            if (start[x] > start[x - 1] + cap) start[x] = start[x - 1] + cap;
        }
        for (int x = 7; x >= 1; --x)
            if (start[x] < start[x + 1] - cap) start[x] = start[x + 1] - cap;
    }
    __syncthreads();
}
// Dispatch order of the tiles inside each XCD run: [K*ntiles + 16 + ntiles + j] = tile worked on at position j.  A
// conv_fwd_flow wave owns a tile for all of its offsets, and tiles differ 10x in cost (interior tiles: 27 offsets x 4 row
// groups; surface tiles a third of that): in plain row order the launch ends with a few waves still inside heavy tiles
// they took last -- up to 0.3 ms of a 2.3 ms launch with most SIMDs idle.  The last `window` positions of every run are
// therefore ordered by cost class, heaviest first (8 classes of equal width below the run's largest cost,
// stable: row order inside a class, so neighbouring tiles of a class still run together); the positions before the
// window keep the row order and with it all of the L2 locality.
#define ORDER_MAX_CLASSES 8
__global__ __launch_bounds__(1024) void rulebook_order_kernel(int32_t* __restrict__ rb_cnt, int32_t K, int64_t ntiles,
                                                              int window, int ncls) {
    __shared__ int cnts[ORDER_MAX_CLASSES][1024];
    __shared__ long long part[1024];
    __shared__ int red[1024];
    __shared__ int base[ORDER_MAX_CLASSES + 1];
    __shared__ int start[9];
    int32_t* out = rb_cnt + (int64_t)K * ntiles;
    const int32_t* cost = out + 16;
    int32_t* order = rb_cnt + (int64_t)K * ntiles + 16 + ntiles;
    const int tid = threadIdx.x;
    rulebook_runs(cost, ntiles, part, start);
    if (blockIdx.x == 0 && tid < 16) out[tid] = tid < 9 ? start[tid] : 0;
    const int s0 = start[blockIdx.x], s1 = start[blockIdx.x + 1];
    const int w0 = (window <= 0 || s1 - window < s0) ? s0 : s1 - window;
    for (int t = s0 + tid; t < w0; t += 1024) order[t] = t;
    const int n = s1 - w0;
    if (n <= 0) return;
    const int per = (n + 1023) / 1024;
    const int lo = w0 + (tid * per < n ? tid * per : n);
    const int hi = lo + per < s1 ? lo + per : s1;
    int mx = 0;
    for (int t = lo; t < hi; ++t) mx = cost[t] > mx ? cost[t] : mx;
    red[tid] = mx;
    __syncthreads();
    for (int d = 512; d > 0; d >>= 1) {
        if (tid < d && red[tid + d] > red[tid]) red[tid] = red[tid + d];
        __syncthreads();
    }
    const int cmax = red[0];
    auto cls_of = [&](int c) { const int q = (int)((long long)c * ncls / (cmax + 1)); return ncls - 1 - q; };   // 0 = heaviest
    int mine[ORDER_MAX_CLASSES];
#pragma unroll
    for (int c = 0; c < ORDER_MAX_CLASSES; ++c) mine[c] = 0;
    for (int t = lo; t < hi; ++t) {
        const int q = cls_of(cost[t]);
#pragma unroll
        for (int c = 0; c < ORDER_MAX_CLASSES; ++c) mine[c] += (c == q);
    }
#pragma unroll
    for (int c = 0; c < ORDER_MAX_CLASSES; ++c) cnts[c][tid] = mine[c];
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {                   // inclusive scans over the threads, all classes at once
        int v[ORDER_MAX_CLASSES];
#pragma unroll
        for (int c = 0; c < ORDER_MAX_CLASSES; ++c) v[c] = tid >= d ? cnts[c][tid - d] : 0;
        __syncthreads();
#pragma unroll
        for (int c = 0; c < ORDER_MAX_CLASSES; ++c) cnts[c][tid] += v[c];
        __syncthreads();
    }
    if (tid == 0) {
        base[0] = 0;
        for (int c = 0; c < ORDER_MAX_CLASSES; ++c) base[c + 1] = base[c] + cnts[c][1023];
    }
    __syncthreads();
    int pos[ORDER_MAX_CLASSES];
#pragma unroll
    for (int c = 0; c < ORDER_MAX_CLASSES; ++c) pos[c] = w0 + base[c] + cnts[c][tid] - mine[c];
    for (int t = lo; t < hi; ++t) {
        const int q = cls_of(cost[t]);
        int p = 0;
#pragma unroll
        for (int c = 0; c < ORDER_MAX_CLASSES; ++c)
            if (c == q) { p = pos[c]; pos[c] += 1; }
        order[p] = t;
    }
}
static int coords_env(const char* name, int dflt) { return b2m_env_int(name, dflt); }
extern "C" int64_t b2m_rulebook_cnt_size(int32_t K, int64_t n_out) {
    const int64_t ntiles = cdiv64(n_out, B2M_TILE);
    return (int64_t)K * ntiles + 16 + 2 * ntiles;
}
extern "C" int b2m_rulebook_balance(int32_t* rb_cnt, int32_t K, int64_t n_out, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    B2M_CHECK_ARG(rb_cnt && K >= 1, "bad arguments");
    const int64_t ntiles = cdiv64(n_out, B2M_TILE);
    if (ntiles < B2M_BALANCE_MIN_TILES) return B2M_OK;     // (the convolutions use the tail from that many tiles on)
    rulebook_cost_kernel<<<(unsigned)cdiv64(ntiles, 256), 256, 0, st>>>(rb_cnt, K, ntiles);
    int ncls = 8;                      // cost classes and window: 4 .. 16 classes, 256 .. 2048 positions measured equal
    if (ncls < 1) ncls = 1;
    if (ncls > ORDER_MAX_CLASSES) ncls = ORDER_MAX_CLASSES;
    rulebook_order_kernel<<<8, 1024, 0, st>>>(rb_cnt, K, ntiles, 768, ncls);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}

// ------------------------------------------------------------------ tile rulebook
// one wave per tile of B2M_TILE (= 64) output rows: lane = row; per offset one ballot compacts the valid
// pairs in row order
static_assert(B2M_TILE == 64, "the rulebook kernel maps one lane to one tile row");
__global__ __launch_bounds__(256) void rulebook_kernel(const int32_t* __restrict__ nbr, int64_t ld, int32_t K,
                                                       int64_t n_out, int64_t ntiles, int32_t* __restrict__ rb_in,
                                                       uint8_t* __restrict__ rb_out, int32_t* __restrict__ rb_cnt,
                                                       int32_t* __restrict__ pair_total) {
    const int64_t t = blockIdx.x;
    const int k = blockIdx.y * 4 + (threadIdx.x >> 6);      // one wave per (tile, offset)
    if (k >= K) return;
    const int lane = lane_id();
    const int64_t ldr = ntiles * B2M_TILE;
    const int64_t o = t * B2M_TILE + lane;
    const int v = o < n_out ? nbr[(int64_t)k * ld + o] : -1;
    const uint64_t b = __ballot(v >= 0);
    const int c = __popcll(b);
    const int64_t base = (int64_t)k * ldr + t * B2M_TILE;
    if (v >= 0) { const int p = prefix_popc(b); rb_in[base + p] = v; rb_out[base + p] = (uint8_t)lane; }
    // the c valid pairs land in slots 0..c-1; lanes c..63 write the padding of slots c..63, so every slot of the
    // rulebook is written exactly once and no memset of the (up to 750 MB) arrays is needed
    if (lane >= c) { rb_in[base + lane] = -1; rb_out[base + lane] = 0; }
    if (lane == 0) {
        rb_cnt[(int64_t)k * ntiles + t] = c;
        if (pair_total && c) atomicAdd(&pair_total[k], c);
    }
}
extern "C" int b2m_rulebook(const int32_t* nbr, int64_t ld, int32_t K, int64_t n_out, int32_t* rb_in, uint8_t* rb_out,
                            int32_t* rb_cnt, int32_t* pair_total, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    B2M_CHECK_ARG(K >= 1 && K <= 343 && ld >= n_out, "bad K or ld");
    int64_t ntiles = cdiv64(n_out, B2M_TILE);
    if (pair_total) B2M_HIP(hipMemsetAsync(pair_total, 0, K * sizeof(int32_t), st));
    if (ntiles == 0) return B2M_OK;
    rulebook_kernel<<<dim3((unsigned)ntiles, (unsigned)((K + 3) / 4)), 256, 0, st>>>(nbr, ld, K, n_out, ntiles, rb_in,
                                                                                   rb_out, rb_cnt, pair_total);
    B2M_LAUNCH_CHECK();
    return b2m_rulebook_balance(rb_cnt, K, n_out, stream);
}

// ------------------------------------------------------------------ stride-1 kernel map straight into the rulebook
// kernel_map_kernel + rulebook_kernel in one pass, without the K x N neighbour table in between (600 MB written
// and read again for the 5x5x5 stem).  One wave per (tile, (dy,dz) line): lane = output row; the ksize offsets of
// the line are probed one after the other and compacted with a ballot each.  Same probes, same order, same
// rulebook bits as the two-step path (tests/test_gpu_coords.py).
template <bool OCC>
__global__ __launch_bounds__(256) void map_rulebook_kernel(const int32_t* __restrict__ coords, int64_t n, int32_t ksize,
                                                           int32_t ts, const uint64_t* __restrict__ keys,
                                                           const int32_t* __restrict__ vals, int64_t mask,
                                                           const uint64_t* __restrict__ occ, OccDims d, int64_t ntiles,
                                                           int32_t* __restrict__ rb_in, uint8_t* __restrict__ rb_out,
                                                           int32_t* __restrict__ rb_cnt) {
    const int64_t t = blockIdx.x;
    const int line = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (line >= ksize * ksize) return;
    const int lane = lane_id();
    const int64_t o = t * B2M_TILE + lane;
    const int h = ksize / 2;
    const int dy = (line % ksize - h) * ts, dz = (line / ksize - h) * ts;
    i32x4 c = {0, 0, 0, 0};
    if (o < n) c = *(const i32x4*)(coords + o * 4);
    const int y = c.z + dy, z = c.w + dz;
    bool yz = o < n && (unsigned)y < 65536u && (unsigned)z < 65536u;
    uint32_t bitsx = ~0u;
    if (OCC) {
        yz = yz && y < d.Y && z < d.Z;
        if (yz) {
            const int64_t b0 = (((int64_t)c.x * d.Z + z) * d.Y + y) * d.X + (c.y - h);
            const int64_t bb = b0 < 0 ? 0 : b0;
            const int sh = (int)(bb & 63);
            uint64_t w = occ[bb >> 6] >> sh;
            if (sh) w |= occ[(bb >> 6) + 1] << (64 - sh);
            if (b0 < 0) w <<= (int)(-b0);
            bitsx = (uint32_t)w;
            for (int kx = 0; kx < ksize; ++kx) {
                const int x = c.y - h + kx;
                if (x < 0 || x >= d.X) bitsx &= ~(1u << kx);
            }
        }
    }
    const int64_t ldr = ntiles * B2M_TILE;
    for (int kx = 0; kx < ksize; ++kx) {
        const int x = c.y + (kx - h) * ts;
        int v = -1;
        if (yz && (unsigned)x < 65536u && ((bitsx >> kx) & 1u)) {
            const int64_t s = b2m_find(keys, mask, b2m_pack(c.x, x, y, z));
            if (s >= 0) v = vals[s];
        }
        const int k = line * ksize + kx;
        const uint64_t b = __ballot(v >= 0);
        const int cnt = __popcll(b);
        const int64_t base = (int64_t)k * ldr + t * B2M_TILE;
        if (v >= 0) { const int p = prefix_popc(b); rb_in[base + p] = v; rb_out[base + p] = (uint8_t)lane; }
        if (lane >= cnt) { rb_in[base + lane] = -1; rb_out[base + lane] = 0; }
        if (lane == 0) rb_cnt[(int64_t)k * ntiles + t] = cnt;
    }
}
extern "C" int b2m_kernel_map_rulebook(const int32_t* coords, int64_t n, int32_t ksize, int32_t ts, const uint64_t* keys,
                                       const int32_t* vals, int64_t cap, const uint64_t* occ, int32_t dim_x,
                                       int32_t dim_y, int32_t dim_z, int32_t* rb_in, uint8_t* rb_out, int32_t* rb_cnt,
                                       void* stream) {
    hipStream_t st = (hipStream_t)stream;
    B2M_CHECK_ARG(ksize == 1 || ksize == 3 || ksize == 5 || ksize == 7, "ksize must be odd (1,3,5,7)");
    B2M_CHECK_ARG(coords && keys && vals && rb_in && rb_out && rb_cnt && is_pow2(cap), "bad arguments");
    B2M_CHECK_ARG(occ == nullptr || (ts == 1 && dim_x > 0 && dim_y > 0 && dim_z > 0), "the occupancy bitmap is for stride 1");
    const int64_t ntiles = cdiv64(n, B2M_TILE);
    if (ntiles == 0) return B2M_OK;
    const dim3 grid((unsigned)ntiles, (unsigned)((ksize * ksize + 3) / 4));
    const OccDims d{dim_x, dim_y, dim_z};
    if (occ) map_rulebook_kernel<true><<<grid, 256, 0, st>>>(coords, n, ksize, ts, keys, vals, cap - 1, occ, d, ntiles, rb_in, rb_out, rb_cnt);
    else map_rulebook_kernel<false><<<grid, 256, 0, st>>>(coords, n, ksize, ts, keys, vals, cap - 1, occ, d, ntiles, rb_in, rb_out, rb_cnt);
    B2M_LAUNCH_CHECK();
    return b2m_rulebook_balance(rb_cnt, ksize * ksize * ksize, n, stream);
}

// ------------------------------------------------------------------ Morton keys (spatial row order)
__device__ __forceinline__ uint64_t spread3(uint32_t v) {      // 16 bits -> every third bit
    uint64_t x = v & 0xFFFFull;
    x = (x | (x << 32)) & 0x00FF00000000FFFFull;
    x = (x | (x << 16)) & 0x00FF0000FF0000FFull;
    x = (x | (x << 8)) & 0xF00F00F00F00F00Full;
    x = (x | (x << 4)) & 0x30C30C30C30C30C3ull;
    x = (x | (x << 2)) & 0x9249249249249249ull;
    return x;
}
__global__ void morton_keys_kernel(const int32_t* __restrict__ coords, int64_t n, int64_t* __restrict__ keys) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    i32x4 c = *(const i32x4*)(coords + i * 4);
    // batch index in the top 15 bits (keys stay non-negative as int64), 3 x 16 interleaved coordinate bits below
    uint64_t k = ((uint64_t)(uint32_t)c.x << 48) | spread3((uint32_t)c.y) | (spread3((uint32_t)c.z) << 1) |
                 (spread3((uint32_t)c.w) << 2);
    keys[i] = (int64_t)k;
}
// Hilbert keys: the same role as the Morton keys, a curve without their jumps (a 64-row tile of a surface is a more compact patch:
// on the benchmark's maps the useful share of the executed MFMAs rises 0.864 -> 0.878 on level 0, 0.813 -> 0.831 on level 1,
// and a tile has 5 % fewer active offsets).  Skilling's transform ("Programming the Hilbert curve", 2004) of the three
// `bits`-bit coordinates into the transposed index, whose bits are then interleaved exactly like a Morton key's.
__global__ void hilbert_keys_kernel(const int32_t* __restrict__ coords, int64_t n, int bits, int64_t* __restrict__ keys) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    i32x4 c = *(const i32x4*)(coords + i * 4);
    uint32_t X[3] = {(uint32_t)c.y, (uint32_t)c.z, (uint32_t)c.w};
    const uint32_t M = 1u << (bits - 1);
    for (uint32_t Q = M; Q > 1; Q >>= 1) {               // inverse undo of the excess work
        const uint32_t P = Q - 1;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (X[a] & Q) X[0] ^= P;
            else { const uint32_t t = (X[0] ^ X[a]) & P; X[0] ^= t; X[a] ^= t; }
        }
    }
    X[1] ^= X[0]; X[2] ^= X[1];                           // Gray encode
    uint32_t t = 0;
    for (uint32_t Q = M; Q > 1; Q >>= 1)
        if (X[2] & Q) t ^= Q - 1;
    X[0] ^= t; X[1] ^= t; X[2] ^= t;
    const uint64_t k = ((uint64_t)(uint32_t)c.x << 48) | (spread3(X[0]) << 2) | (spread3(X[1]) << 1) | spread3(X[2]);
    keys[i] = (int64_t)k;
}
// ------------------------------------------------------------------ radix argsort of 64-bit keys (Morton row order)
// LSD radix sort of (key, row) pairs, 8 bits per pass, stable, only over the digits a caller-supplied bit mask says can
// differ (Morton keys of a batch: 3 x bitlength(max coordinate) low bits + the batch bits at 48: 4-5 passes instead of 8).
// Per pass: (1) one 256-bin histogram per block of RS_TILE keys (LDS atomics), stored digit-major; (2) one workgroup per digit
// scans that digit's block counts and leaves the digit's total; (3) every block scans the 256 totals, ranks its keys again --
// in input order, wave by wave: lanes with equal digits find each other with eight ballots, the lowest of them owns the
// digit's running counter in LDS -- and scatters key and row.  The last pass also writes the permutation and its inverse
// as int64 (torch index tensors), so nothing of the sort is left to torch (round 3: torch.argsort = rocprim onesweep).
#define RS_THREADS 256
#define RS_ITEMS 16
#define RS_TILE (RS_THREADS * RS_ITEMS)
__global__ __launch_bounds__(RS_THREADS) void rs_hist_kernel(const uint64_t* __restrict__ keys, int64_t n, int shift, int nblk,
                                                             int32_t* __restrict__ hist) {
    __shared__ int h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RS_TILE;
#pragma unroll 4
    for (int it = 0; it < RS_ITEMS; ++it) {
        const int64_t i = base + it * RS_THREADS + threadIdx.x;
        if (i < n) atomicAdd(&h[(int)((keys[i] >> shift) & 255)], 1);
    }
    __syncthreads();
    hist[(int64_t)threadIdx.x * nblk + blockIdx.x] = h[threadIdx.x];
}
// per digit d (one workgroup each): exclusive scan of the digit's nblk block counts in place, the digit's total to tot[d]
// (round 4, first form: ONE workgroup scanned all 256 x nblk counts -- 109 us per pass on a 1.2 M-key batch)
__global__ __launch_bounds__(256) void rs_scan_kernel(int32_t* __restrict__ hist, int nblk, int32_t* __restrict__ tot) {
    __shared__ int part[256];
    const int tid = threadIdx.x;
    int32_t* a = hist + (int64_t)blockIdx.x * nblk;
    const int per = (nblk + 255) / 256;
    const int lo = tid * per < nblk ? tid * per : nblk, hi = lo + per < nblk ? lo + per : nblk;
    int s = 0;
    for (int i = lo; i < hi; ++i) s += a[i];
    part[tid] = s;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const int v = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - s;
    for (int i = lo; i < hi; ++i) { const int v = a[i]; a[i] = run; run += v; }
    if (tid == 255) tot[blockIdx.x] = part[255];
}
__global__ __launch_bounds__(RS_THREADS) void rs_scatter_kernel(const uint64_t* __restrict__ keys, const int32_t* __restrict__ vals,
                                                                int64_t n, int shift, int nblk, const int32_t* __restrict__ hist,
                                                                const int32_t* __restrict__ tot,
                                                                uint64_t* __restrict__ keys_out, int32_t* __restrict__ vals_out,
                                                                int64_t* __restrict__ perm64, int64_t* __restrict__ inv64) {
    __shared__ int base[256];                       // next output position of every digit for this block
    __shared__ int wcnt[RS_THREADS / 64][256];      // keys of the round with that digit, per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // first position of digit `tid` = keys with smaller digits (scan of the 256 digit totals, here) + keys with this digit in
    // earlier blocks (rs_scan_kernel)
    const int mine = tot[tid];
    base[tid] = mine;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const int v = tid >= d ? base[tid - d] : 0;
        __syncthreads();
        base[tid] += v;
        __syncthreads();
    }
    const int first = base[tid] - mine + hist[(int64_t)tid * nblk + blockIdx.x];
    __syncthreads();
    base[tid] = first;
    const int64_t tile0 = (int64_t)blockIdx.x * RS_TILE;
    for (int it = 0; it < RS_ITEMS; ++it) {
#pragma unroll
        for (int w = 0; w < RS_THREADS / 64; ++w) wcnt[w][tid] = 0;
        __syncthreads();
        const int64_t i = tile0 + it * RS_THREADS + tid;
        const bool live = i < n;
        const uint64_t k = live ? keys[i] : 0;
        const int d = (int)((k >> shift) & 255);
        // lanes of this wave with the same digit
        uint64_t same = __ballot(live);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const uint64_t m = __ballot((d >> b) & 1);
            same &= ((d >> b) & 1) ? m : ~m;
        }
        const int below = __builtin_popcountll(same & ((1ull << lane) - 1));
        if (live && below == 0) wcnt[wave][d] = __builtin_popcountll(same);
        __syncthreads();
        if (live) {
            int pos = base[d] + below;
            for (int w = 0; w < wave; ++w) pos += wcnt[w][d];
            const int v = vals ? vals[i] : (int)i;
            keys_out[pos] = k; vals_out[pos] = v;
            if (perm64) { perm64[pos] = v; inv64[v] = pos; }
        }
        __syncthreads();
        int add = 0;
#pragma unroll
        for (int w = 0; w < RS_THREADS / 64; ++w) add += wcnt[w][tid];
        base[tid] += add;
        // (the zeroing of wcnt at the top of the next round is behind the barrier above)
    }
}
extern "C" int64_t b2m_radix_argsort_scratch(int64_t n) {
    const int64_t nblk = cdiv64(n > 0 ? n : 1, RS_TILE);
    return 2 * n * 8 + 2 * n * 4 + 256 * nblk * 4 + 256 * 4 + 64;          // two key buffers, two row buffers, the histograms, digit totals
}
extern "C" int b2m_radix_argsort(const uint64_t* keys, int64_t n, uint64_t bit_mask, int64_t* perm, int64_t* inv_perm,
                                 void* scratch, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    B2M_CHECK_ARG(n >= 0 && n < (1ll << 31) && (n == 0 || (keys && perm && inv_perm && scratch)), "bad arguments");
    if (n == 0) return B2M_OK;
    const int nblk = (int)cdiv64(n, RS_TILE);
    uint64_t* kbuf[2] = {(uint64_t*)scratch, (uint64_t*)scratch + n};
    int32_t* vbuf[2] = {(int32_t*)((uint64_t*)scratch + 2 * n), (int32_t*)((uint64_t*)scratch + 2 * n) + n};
    int32_t* hist = vbuf[1] + n;
    int32_t* tot = hist + (int64_t)256 * nblk;
    int shifts[8], np = 0;
    for (int sft = 0; sft < 64; sft += 8)
        if ((bit_mask >> sft) & 255ull) shifts[np++] = sft;
    if (np == 0) shifts[np++] = 0;                   // (all keys equal: one pass writes the identity permutation)
    const uint64_t* kin = keys;
    const int32_t* vin = nullptr;
    for (int p = 0; p < np; ++p) {
        const bool last = p == np - 1;
        rs_hist_kernel<<<nblk, RS_THREADS, 0, st>>>(kin, n, shifts[p], nblk, hist);
        rs_scan_kernel<<<256, 256, 0, st>>>(hist, nblk, tot);
        rs_scatter_kernel<<<nblk, RS_THREADS, 0, st>>>(kin, vin, n, shifts[p], nblk, hist, tot, kbuf[p & 1], vbuf[p & 1],
                                                       last ? perm : nullptr, last ? inv_perm : nullptr);
        kin = kbuf[p & 1]; vin = vbuf[p & 1];
    }
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}

extern "C" int b2m_hilbert_keys(const int32_t* coords, int64_t n, int32_t bits, int64_t* keys, void* stream) {
    B2M_CHECK_ARG(n >= 0 && (n == 0 || (coords && keys)) && bits >= 1 && bits <= 16, "bad arguments (1 <= bits <= 16)");
    if (n == 0) return B2M_OK;
    hilbert_keys_kernel<<<(unsigned)cdiv64(n, 256), 256, 0, (hipStream_t)stream>>>(coords, n, bits, keys);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}
extern "C" int b2m_morton_keys(const int32_t* coords, int64_t n, int64_t* keys, void* stream) {
    B2M_CHECK_ARG(n >= 0 && (n == 0 || (coords && keys)), "bad arguments");
    if (n == 0) return B2M_OK;
    morton_keys_kernel<<<(unsigned)cdiv64(n, 256), 256, 0, (hipStream_t)stream>>>(coords, n, keys);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}
