// Scene preparation on the device: the voxelisation block of the reference's dataset class
// (/root/reference/models/dataloader.py:61-123) -- np.round + np.unique(axis=0, return_inverse=True), the
// nearest-point association (sklearn ball tree, k = 1), np.unique of the segment ids, per-segment centroids.
//
// All of it is HBM/latency-bound integer work on the coordinate hash already used for the kernel maps:
//   keys   : (x,y,z) = rint((pos - shift) / voxel_size) in fp64 exactly as numpy evaluates it, packed x<<42|y<<21|z
//   unique : insert into the open-addressing table, collect first inserters, bitonic-sort the unique keys
//            (lexicographic order == numeric order of the packed key), rank = position in the sorted list
//   nearest: every point visits the <= 27 voxel centres within sqrt(0.75) voxels of itself; the winner per voxel is
//            found with two passes of 64-bit atomicMin (distance bits, then lowest point index at that distance).
//            The distance is the ball tree's reduced distance, summed x,y,z in fp64 without contraction.
#include "b2m_common.h"
#include "../../include/b2m_prepare.h"
#pragma clang fp contract(off)

// order-preserving map double -> uint64 (for atomicMin over signed doubles)
__device__ __forceinline__ unsigned long long f64_ordered(double v) {
    unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double f64_unordered(unsigned long long u) {
    u = (u >> 63) ? (u & 0x7FFFFFFFFFFFFFFFull) : ~u;
    return __longlong_as_double((long long)u);
}

// ------------------------------------------------------------------ shift = min(0, min(positions))
__global__ void vox_shift_init_kernel(unsigned long long* acc) { *acc = f64_ordered(0.0); }
__global__ void vox_shift_kernel(const double* __restrict__ pos, int64_t n3, unsigned long long* __restrict__ acc) {
    double m = 0.0;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n3; e += (int64_t)gridDim.x * blockDim.x) {
        const double v = pos[e];
        m = v < m ? v : m;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double t = __shfl_xor(m, o);
        m = t < m ? t : m;
    }
    if (lane_id() == 0 && m < 0.0) atomicMin(acc, f64_ordered(m));
}
__global__ void vox_shift_final_kernel(const unsigned long long* acc, double* shift) { *shift = f64_unordered(*acc); }

extern "C" int b2m_vox_shift(const double* pos, int64_t n_pts, double* shift, uint64_t* scratch, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    B2M_CHECK_ARG(pos && shift && scratch && n_pts >= 0, "bad arguments");
    vox_shift_init_kernel<<<1, 1, 0, st>>>((unsigned long long*)scratch);
    if (n_pts > 0) {
        int64_t nb = cdiv64(n_pts * 3, 256 * 8);
        if (nb > 2048) nb = 2048;
        vox_shift_kernel<<<(unsigned)nb, 256, 0, st>>>(pos, n_pts * 3, (unsigned long long*)scratch);
    }
    vox_shift_final_kernel<<<1, 1, 0, st>>>((const unsigned long long*)scratch, shift);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}

// ------------------------------------------------------------------ voxel keys
#define VOX_BITS 21
#define VOX_LIM (1 << VOX_BITS)
__device__ __forceinline__ uint64_t vox_pack(int64_t x, int64_t y, int64_t z) {
    return ((uint64_t)x << (2 * VOX_BITS)) | ((uint64_t)y << VOX_BITS) | (uint64_t)z;
}

__global__ void vox_keys_kernel(const double* __restrict__ pos, int64_t n, const double* __restrict__ shift,
                                double voxel_size, uint64_t* __restrict__ keys, int32_t* __restrict__ bad) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const double m = *shift;
    // (positions - min(0, min)) / voxel_size, then np.round (round half to even): dataloader.py:63-67
    const double x = rint((pos[3 * p] - m) / voxel_size);
    const double y = rint((pos[3 * p + 1] - m) / voxel_size);
    const double z = rint((pos[3 * p + 2] - m) / voxel_size);
    if (!(x >= 0.0 && x < (double)VOX_LIM && y >= 0.0 && y < (double)VOX_LIM && z >= 0.0 && z < (double)VOX_LIM)) {
        atomicAdd(bad, 1);
        keys[p] = 0;
        return;
    }
    keys[p] = vox_pack((int64_t)x, (int64_t)y, (int64_t)z);
}

extern "C" int b2m_vox_keys(const double* pos, int64_t n_pts, const double* shift, double voxel_size, uint64_t* keys,
                            int32_t* bad_count, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    B2M_CHECK_ARG(pos && shift && keys && bad_count && n_pts >= 0 && voxel_size > 0.0, "bad arguments");
    B2M_HIP(hipMemsetAsync(bad_count, 0, sizeof(int32_t), st));
    if (n_pts == 0) return B2M_OK;
    vox_keys_kernel<<<(unsigned)cdiv64(n_pts, 256), 256, 0, st>>>(pos, n_pts, shift, voxel_size, keys, bad_count);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}

// ------------------------------------------------------------------ unique with inverse on 64-bit keys
// slot of `key` in the table, inserting it if absent; the first inserter publishes the key in the (unordered) unique list.
// A plain read in front of every compare-and-swap: once a key is in the table, later arrivals never touch the atomic unit.
__device__ __forceinline__ int64_t unique_insert_one(uint64_t key, uint64_t* __restrict__ tkeys, int64_t mask,
                                                     uint64_t* __restrict__ ukeys, int32_t* __restrict__ n_unique) {
    int64_t s = (int64_t)(b2m_hash(key) & (uint64_t)mask);
    for (;;) {
        uint64_t cur = __hip_atomic_load(&tkeys[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == B2M_EMPTY_KEY) {
            cur = atomicCAS((unsigned long long*)&tkeys[s], (unsigned long long)B2M_EMPTY_KEY, (unsigned long long)key);
            if (cur == B2M_EMPTY_KEY) {
                ukeys[atomicAdd(n_unique, 1)] = key;
                return s;
            }
        }
        if (cur == key) return s;
        s = (s + 1) & mask;
    }
}
// Inputs with few distinct keys (1.2 M ground-truth instance ids holding a few dozen values: utils/eval_metric.py:316-330)
// made every thread of the grid fight for the same handful of slots (round 2: 7.7 ms per scene).  Now a wave first
// elects, per distinct key it holds, ONE lane that probes / inserts, and hands the slot to the others by a cross-lane
// read: one atomic per (wave, key) at most.  A wave whose keys are mostly distinct (voxel keys) sees that in the first
// election and takes the per-lane path at once.
__global__ void unique_insert_kernel(const uint64_t* __restrict__ in, int64_t n, uint64_t* __restrict__ tkeys,
                                     int64_t mask, int32_t* __restrict__ slot_of, uint64_t* __restrict__ ukeys,
                                     int32_t* __restrict__ n_unique) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = lane_id();
    bool pending = i < n;
    const uint64_t key = pending ? in[i] : 0;
    int64_t myslot = -1;
    for (int iter = 0; iter < 16; ++iter) {
        const uint64_t act = __ballot(pending);
        if (act == 0) break;
        const int leader = __builtin_ctzll(act);
        const uint64_t lk = (uint64_t)__shfl((long long)key, leader, 64);
        const bool same = pending && key == lk;
        const uint64_t grp = __ballot(same);
        if (iter == 0 && __builtin_popcountll(grp) * 8 < __builtin_popcountll(act)) break;     // mostly distinct keys
        long long s = 0;
        if (lane == leader) s = unique_insert_one(key, tkeys, mask, ukeys, n_unique);
        s = __shfl(s, leader, 64);
        if (same) { myslot = s; pending = false; }
    }
    if (pending) myslot = unique_insert_one(key, tkeys, mask, ukeys, n_unique);
    if (i < n) slot_of[i] = (int32_t)myslot;
}

static bool pow2(int64_t v) { return v > 0 && (v & (v - 1)) == 0; }

static int unique_insert_launch(const uint64_t* in, int64_t n, uint64_t* tkeys, int64_t cap, int32_t* slot_of, uint64_t* ukeys,
                                int32_t* n_unique, hipStream_t st) {
    B2M_CHECK_ARG(in && tkeys && slot_of && ukeys && n_unique && n >= 0 && n < (1ll << 31), "bad arguments");
    B2M_CHECK_ARG(pow2(cap) && cap >= 2 * n && cap < (1ll << 31), "cap must be a power of two >= 2n");
    B2M_HIP(hipMemsetAsync(tkeys, 0xFF, cap * sizeof(uint64_t), st));
    B2M_HIP(hipMemsetAsync(n_unique, 0, sizeof(int32_t), st));
    if (n > 0) {
        unique_insert_kernel<<<(unsigned)cdiv64(n, 256), 256, 0, st>>>(in, n, tkeys, cap - 1, slot_of, ukeys, n_unique);
        B2M_LAUNCH_CHECK();
    }
    return B2M_OK;
}
extern "C" int64_t b2m_unique_insert(const uint64_t* in, int64_t n, uint64_t* tkeys, int64_t cap, int32_t* slot_of,
                                     uint64_t* ukeys, int32_t* n_unique, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const int rc = unique_insert_launch(in, n, tkeys, cap, slot_of, ukeys, n_unique, st);
    if (rc != B2M_OK) return rc;
    int32_t h = 0;
    B2M_HIP(hipMemcpyAsync(&h, n_unique, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    B2M_HIP(hipStreamSynchronize(st));
    return (int64_t)h;
}
// the same without the host read: the count stays in *n_unique on the device (prepare.voxelize_scenes reads the counts of all
// scenes of a batch at once)
extern "C" int b2m_unique_insert_async(const uint64_t* in, int64_t n, uint64_t* tkeys, int64_t cap, int32_t* slot_of,
                                       uint64_t* ukeys, int32_t* n_unique, void* stream) {
    return unique_insert_launch(in, n, tkeys, cap, slot_of, ukeys, n_unique, (hipStream_t)stream);
}

// ---- bitonic sort of uint64 keys, ascending; n is a power of two (caller pads with 0xFF..FF)
#define SORT_CHUNK 4096
#define SORT_THREADS 512
__device__ __forceinline__ void cmpx(uint64_t& a, uint64_t& b, bool up) {
    if ((a > b) == up) { const uint64_t t = a; a = b; b = t; }
}
// phase 0: every chunk fully sorted for all k <= chunk (direction alternates with the global index)
// phase 1: the steps j = chunk/2 .. 1 of stage k (k > chunk)
__global__ __launch_bounds__(SORT_THREADS) void sort_local_kernel(uint64_t* __restrict__ a, int64_t n, int chunk, int64_t kstage) {
    __shared__ uint64_t s[SORT_CHUNK];
    const int64_t base = (int64_t)blockIdx.x * chunk;
    for (int e = threadIdx.x; e < chunk; e += SORT_THREADS) s[e] = a[base + e];
    __syncthreads();
    const int64_t k0 = kstage ? kstage : 2;
    const int64_t k1 = kstage ? kstage : chunk;
    for (int64_t k = k0; k <= k1; k <<= 1) {
        const int jstart = (int)((k >> 1) < (chunk >> 1) ? (k >> 1) : (chunk >> 1));
        for (int j = jstart; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < (chunk >> 1); t += SORT_THREADS) {
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));      // index with bit j clear
                const bool up = (((base + lo) & k) == 0);
                uint64_t x = s[lo], y = s[lo | j];
                cmpx(x, y, up);
                s[lo] = x; s[lo | j] = y;
            }
            __syncthreads();
        }
    }
    for (int e = threadIdx.x; e < chunk; e += SORT_THREADS) a[base + e] = s[e];
}
__global__ void sort_global_kernel(uint64_t* __restrict__ a, int64_t n, int64_t k, int64_t j) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (n >> 1)) return;
    const int64_t lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
    const bool up = ((lo & k) == 0);
    uint64_t x = a[lo], y = a[lo | j];
    if ((x > y) == up) { a[lo] = y; a[lo | j] = x; }
}

extern "C" int b2m_sort_u64(uint64_t* keys, int64_t n_pad, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    B2M_CHECK_ARG(keys && pow2(n_pad) && n_pad >= 2 && n_pad < (1ll << 31), "n_pad must be a power of two >= 2");
    const int chunk = n_pad < SORT_CHUNK ? (int)n_pad : SORT_CHUNK;
    const unsigned nblk = (unsigned)(n_pad / chunk);
    sort_local_kernel<<<nblk, SORT_THREADS, 0, st>>>(keys, n_pad, chunk, 0);
    for (int64_t k = (int64_t)chunk << 1; k <= n_pad; k <<= 1) {
        for (int64_t j = k >> 1; j >= chunk; j >>= 1)
            sort_global_kernel<<<(unsigned)cdiv64(n_pad >> 1, 256), 256, 0, st>>>(keys, n_pad, k, j);
        sort_local_kernel<<<nblk, SORT_THREADS, 0, st>>>(keys, n_pad, chunk, k);
    }
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}

__global__ void unique_rank_kernel(const uint64_t* __restrict__ sorted, int64_t nu, const uint64_t* __restrict__ tkeys,
                                   int32_t* __restrict__ tvals, int64_t mask) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nu) return;
    const int64_t s = b2m_find(tkeys, mask, sorted[r]);
    if (s >= 0) tvals[s] = (int32_t)r;
}
__global__ void unique_inverse_kernel(const int32_t* __restrict__ slot_of, int64_t n, const int32_t* __restrict__ tvals,
                                      int64_t* __restrict__ inverse) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) inverse[i] = tvals[slot_of[i]];
}

extern "C" int b2m_unique_rank(const uint64_t* sorted, int64_t n_unique, const uint64_t* tkeys, int32_t* tvals,
                               int64_t cap, const int32_t* slot_of, int64_t n, int64_t* inverse, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    B2M_CHECK_ARG(sorted && tkeys && tvals && slot_of && inverse && pow2(cap) && n >= 0 && n_unique >= 0, "bad arguments");
    if (n_unique > 0) unique_rank_kernel<<<(unsigned)cdiv64(n_unique, 256), 256, 0, st>>>(sorted, n_unique, tkeys, tvals, cap - 1);
    if (n > 0) unique_inverse_kernel<<<(unsigned)cdiv64(n, 256), 256, 0, st>>>(slot_of, n, tvals, inverse);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}

// ------------------------------------------------------------------ sorted voxel keys -> [b,x,y,z] rows
__global__ void vox_decode_kernel(const uint64_t* __restrict__ sorted, int64_t n, int32_t batch, int32_t* __restrict__ coords) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const uint64_t k = sorted[v];
    i32x4 c;
    c.x = batch;
    c.y = (int32_t)(k >> (2 * VOX_BITS));
    c.z = (int32_t)((k >> VOX_BITS) & (VOX_LIM - 1));
    c.w = (int32_t)(k & (VOX_LIM - 1));
    *(i32x4*)(coords + 4 * v) = c;
}
extern "C" int b2m_vox_decode(const uint64_t* sorted, int64_t n_vox, int32_t batch, int32_t* coords, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    B2M_CHECK_ARG(sorted && coords && n_vox >= 0, "bad arguments");
    if (n_vox > 0) vox_decode_kernel<<<(unsigned)cdiv64(n_vox, 256), 256, 0, st>>>(sorted, n_vox, batch, coords);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}

// ------------------------------------------------------------------ nearest scene point of every voxel centre
template <bool SECOND>
__global__ void vox_nearest_kernel(const double* __restrict__ pos, int64_t n, const double* __restrict__ shift,
                                   double voxel_size, const uint64_t* __restrict__ tkeys,
                                   const int32_t* __restrict__ tvals, int64_t mask,
                                   unsigned long long* __restrict__ best, int32_t* __restrict__ point2vox) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const double m = *shift;
    const double px = (pos[3 * p] - m) / voxel_size, py = (pos[3 * p + 1] - m) / voxel_size,
                 pz = (pos[3 * p + 2] - m) / voxel_size;
    const int64_t cx = (int64_t)rint(px), cy = (int64_t)rint(py), cz = (int64_t)rint(pz);
    for (int dx = -1; dx <= 1; ++dx) {
        const int64_t x = cx + dx;
        if (x < 0 || x >= VOX_LIM) continue;
        const double ex = (double)x - px;
        const double d0 = ex * ex;
        for (int dy = -1; dy <= 1; ++dy) {
            const int64_t y = cy + dy;
            if (y < 0 || y >= VOX_LIM) continue;
            const double ey = (double)y - py;
            const double d1 = d0 + ey * ey;
            for (int dz = -1; dz <= 1; ++dz) {
                const int64_t z = cz + dz;
                if (z < 0 || z >= VOX_LIM) continue;
                const double ez = (double)z - pz;
                const double d = d1 + ez * ez;       // ball-tree reduced distance, summed x, y, z
                // every voxel owns a point within 0.75 (squared) of its centre, so nothing farther can win
                if (d > 0.75000001) continue;
                const int64_t s = b2m_find(tkeys, mask, vox_pack(x, y, z));
                if (s < 0) continue;
                const int32_t v = tvals[s];
                const unsigned long long bits = (unsigned long long)__double_as_longlong(d);   // d >= 0: bit order == value order
                if (!SECOND) atomicMin(&best[v], bits);
                else if (best[v] == bits) atomicMin(&point2vox[v], (int32_t)p);
            }
        }
    }
}

extern "C" int b2m_vox_nearest(const double* pos, int64_t n_pts, const double* shift, double voxel_size,
                               const uint64_t* tkeys, const int32_t* tvals, int64_t cap, int64_t n_vox,
                               uint64_t* best, int32_t* point2vox, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    B2M_CHECK_ARG(pos && shift && tkeys && tvals && best && point2vox && pow2(cap) && n_pts >= 0 && n_vox >= 0 &&
                  n_pts < (1ll << 31), "bad arguments");
    B2M_HIP(hipMemsetAsync(best, 0xFF, (size_t)n_vox * sizeof(uint64_t), st));
    B2M_HIP(hipMemsetAsync(point2vox, 0x7F, (size_t)n_vox * sizeof(int32_t), st));
    if (n_pts == 0) return B2M_OK;
    const unsigned nb = (unsigned)cdiv64(n_pts, 256);
    vox_nearest_kernel<false><<<nb, 256, 0, st>>>(pos, n_pts, shift, voxel_size, tkeys, tvals, cap - 1,
                                                  (unsigned long long*)best, point2vox);
    vox_nearest_kernel<true><<<nb, 256, 0, st>>>(pos, n_pts, shift, voxel_size, tkeys, tvals, cap - 1,
                                                 (unsigned long long*)best, point2vox);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}

// ------------------------------------------------------------------ features / segment id of the associated point
__global__ void vox_gather_kernel(const int32_t* __restrict__ point2vox, int64_t n_vox, const double* __restrict__ colors,
                                  const double* __restrict__ normals, const int64_t* __restrict__ segments,
                                  float* __restrict__ feats, int32_t nfeat, int64_t* __restrict__ vox_segments) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_vox) return;
    const int64_t p = point2vox[v];
#pragma unroll
    for (int c = 0; c < 3; ++c) feats[v * nfeat + c] = (float)colors[3 * p + c];
    if (normals) {
#pragma unroll
        for (int c = 0; c < 3; ++c) feats[v * nfeat + 3 + c] = (float)normals[3 * p + c];
    }
    if (segments) vox_segments[v] = segments[p];
}
extern "C" int b2m_vox_gather(const int32_t* point2vox, int64_t n_vox, const double* colors, const double* normals,
                              const int64_t* segments, float* feats, int64_t* vox_segments, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    B2M_CHECK_ARG(point2vox && colors && feats && n_vox >= 0 && ((segments == nullptr) == (vox_segments == nullptr)),
                  "bad arguments");
    if (n_vox > 0)
        vox_gather_kernel<<<(unsigned)cdiv64(n_vox, 256), 256, 0, st>>>(point2vox, n_vox, colors, normals, segments, feats,
                                                                        normals ? 6 : 3, vox_segments);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}

// ------------------------------------------------------------------ per-segment centroid (dataloader.py:113-117)
// exact integer sums of the voxel indices (order independent), then one fp64 evaluation per segment
__global__ void seg_accum_kernel(const int32_t* __restrict__ coords, const int64_t* __restrict__ seg2vox, int64_t n_vox,
                                 unsigned long long* __restrict__ sums, int32_t* __restrict__ counts) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_vox) return;
    const int64_t s = seg2vox[v];
    const i32x4 c = *(const i32x4*)(coords + 4 * v);
    atomicAdd(&sums[3 * s], (unsigned long long)c.y);
    atomicAdd(&sums[3 * s + 1], (unsigned long long)c.z);
    atomicAdd(&sums[3 * s + 2], (unsigned long long)c.w);
    atomicAdd(&counts[s], 1);
}
__global__ void seg_centroid_kernel(const unsigned long long* __restrict__ sums, const int32_t* __restrict__ counts,
                                    int64_t n_seg, double voxel_size, const double* __restrict__ shift,
                                    double* __restrict__ out) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seg) return;
    const double n = (double)counts[s], m = *shift;
#pragma unroll
    for (int d = 0; d < 3; ++d) out[3 * s + d] = ((double)sums[3 * s + d] / n) * voxel_size + m;
}
extern "C" int b2m_seg_centroid(const int32_t* coords, const int64_t* seg2vox, int64_t n_vox, int64_t n_seg,
                                double voxel_size, const double* shift, uint64_t* sums, int32_t* counts, double* out,
                                void* stream) {
    hipStream_t st = (hipStream_t)stream;
    B2M_CHECK_ARG(coords && seg2vox && shift && sums && counts && out && n_vox >= 0 && n_seg >= 0, "bad arguments");
    B2M_HIP(hipMemsetAsync(sums, 0, (size_t)n_seg * 3 * sizeof(uint64_t), st));
    B2M_HIP(hipMemsetAsync(counts, 0, (size_t)n_seg * sizeof(int32_t), st));
    if (n_vox > 0)
        seg_accum_kernel<<<(unsigned)cdiv64(n_vox, 256), 256, 0, st>>>(coords, seg2vox, n_vox, (unsigned long long*)sums, counts);
    if (n_seg > 0)
        seg_centroid_kernel<<<(unsigned)cdiv64(n_seg, 256), 256, 0, st>>>((const unsigned long long*)sums, counts, n_seg,
                                                                          voxel_size, shift, out);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}

// ------------------------------------------------------------------ box supervision: points -> boxes -> segments
// approx_association of /root/reference/models/dataloader.py:203-314 (segment branch, the ScanNet configuration):
// the reference builds a boxes x points occupancy matrix and walks points and segments in Python.
#define BOX_MAX 1024
__global__ void box_membership_kernel(const double* __restrict__ pos, int64_t n, const double* __restrict__ bb_min,
                                      const double* __restrict__ bb_max, const float* __restrict__ bb_volume,
                                      int32_t nb, int32_t* __restrict__ count, int32_t* __restrict__ first_bb,
                                      int32_t* __restrict__ smallest_bb) {
    __shared__ double lo[BOX_MAX * 3], hi[BOX_MAX * 3];
    __shared__ float vol[BOX_MAX];
    for (int e = threadIdx.x; e < nb * 3; e += blockDim.x) { lo[e] = bb_min[e]; hi[e] = bb_max[e]; }
    for (int e = threadIdx.x; e < nb; e += blockDim.x) vol[e] = bb_volume[e];
    __syncthreads();
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const double x = pos[3 * p], y = pos[3 * p + 1], z = pos[3 * p + 2];
    int c = 0, first = -1, small = -1;
    float sv = 0.f;
    for (int b = 0; b < nb; ++b) {
        // is_within_bb_np (utils/util.py:91-92): closed intervals on every axis
        const bool in = x >= lo[3 * b] && y >= lo[3 * b + 1] && z >= lo[3 * b + 2] && x <= hi[3 * b] &&
                        y <= hi[3 * b + 1] && z <= hi[3 * b + 2];
        if (in) {
            if (c == 0) first = b;
            if (c == 0 || vol[b] < sv) { small = b; sv = vol[b]; }      // np.argmin: first of the smallest
            ++c;
        }
    }
    count[p] = c; first_bb[p] = first; smallest_bb[p] = small;
}

extern "C" int b2m_box_membership(const double* pos, int64_t n_pts, const double* bb_min, const double* bb_max,
                                  const float* bb_volume, int32_t n_boxes, int32_t* count, int32_t* first_bb,
                                  int32_t* smallest_bb, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    B2M_CHECK_ARG(pos && count && first_bb && smallest_bb && n_pts >= 0, "bad arguments");
    B2M_CHECK_ARG(n_boxes >= 0 && n_boxes <= BOX_MAX && (n_boxes == 0 || (bb_min && bb_max && bb_volume)),
                  "0 <= n_boxes <= 1024");
    if (n_pts > 0)
        box_membership_kernel<<<(unsigned)cdiv64(n_pts, 256), 256, 0, st>>>(pos, n_pts, bb_min, bb_max, bb_volume, n_boxes,
                                                                            count, first_bb, smallest_bb);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}

// per segment: lexicographic minimum of (number of boxes, point index) over its points
__global__ void seg_vote_kernel(const int64_t* __restrict__ segments, int64_t n, const uint64_t* __restrict__ tkeys,
                                const int32_t* __restrict__ tvals, int64_t mask, const int32_t* __restrict__ count,
                                unsigned long long* __restrict__ best, int32_t* __restrict__ seg_of_point) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int64_t s = b2m_find(tkeys, mask, (uint64_t)segments[p]);
    const int32_t r = s < 0 ? -1 : tvals[s];        // segments that lost all their voxels are not in the table
    seg_of_point[p] = r;
    if (r >= 0) atomicMin(&best[r], ((unsigned long long)(uint32_t)count[p] << 32) | (unsigned long long)(uint32_t)p);
}
__global__ void seg_assign_kernel(const unsigned long long* __restrict__ best, int64_t n_seg,
                                  const int32_t* __restrict__ first_bb, const int32_t* __restrict__ smallest_bb,
                                  const int64_t* __restrict__ instance_ids, int32_t heuristic,
                                  int64_t* __restrict__ inst_per_seg) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seg) return;
    const unsigned long long b = best[s];
    int64_t inst = -2;                               // unknown (dataloader.py:279)
    if (b != ~0ull) {
        const uint32_t c = (uint32_t)(b >> 32), p = (uint32_t)b;
        if (c == 0) inst = -1;                       // a point outside every box: background (:292-294)
        else if (c == 1) inst = instance_ids[first_bb[p]];           // (:283-290)
        else if (heuristic) inst = instance_ids[smallest_bb[p]];     // smallest box of the least-covered point (:298-309)
    }
    inst_per_seg[s] = inst;
}
__global__ void seg_broadcast_kernel(const int32_t* __restrict__ seg_of_point, int64_t n,
                                     const int64_t* __restrict__ inst_per_seg, int64_t* __restrict__ inst_per_point) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int32_t r = seg_of_point[p];
    inst_per_point[p] = r < 0 ? -2 : inst_per_seg[r];
}

extern "C" int b2m_seg_box_vote(const int64_t* segments, int64_t n_pts, const uint64_t* tkeys, const int32_t* tvals,
                                int64_t cap, int64_t n_seg, const int32_t* count, const int32_t* first_bb,
                                const int32_t* smallest_bb, const int64_t* instance_ids, int32_t smallest_bb_heuristic,
                                uint64_t* best, int32_t* seg_of_point, int64_t* inst_per_seg, int64_t* inst_per_point,
                                void* stream) {
    hipStream_t st = (hipStream_t)stream;
    B2M_CHECK_ARG(segments && tkeys && tvals && count && first_bb && smallest_bb && best && seg_of_point &&
                  inst_per_seg && inst_per_point && pow2(cap) && n_pts >= 0 && n_pts < (1ll << 32) && n_seg >= 0,
                  "bad arguments");
    B2M_HIP(hipMemsetAsync(best, 0xFF, (size_t)n_seg * sizeof(uint64_t), st));
    if (n_pts > 0)
        seg_vote_kernel<<<(unsigned)cdiv64(n_pts, 256), 256, 0, st>>>(segments, n_pts, tkeys, tvals, cap - 1, count,
                                                                      (unsigned long long*)best, seg_of_point);
    if (n_seg > 0)
        seg_assign_kernel<<<(unsigned)cdiv64(n_seg, 256), 256, 0, st>>>((const unsigned long long*)best, n_seg, first_bb,
                                                                        smallest_bb, instance_ids, smallest_bb_heuristic,
                                                                        inst_per_seg);
    if (n_pts > 0)
        seg_broadcast_kernel<<<(unsigned)cdiv64(n_pts, 256), 256, 0, st>>>(seg_of_point, n_pts, inst_per_seg, inst_per_point);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}

// ------------------------------------------------------------------ the other association branches
// Oriented boxes (ARKitScenes.approx_association, dataloader.py:545-557): point p is in box b iff
// -h_b <= R_b (p - c_b) <= h_b on every axis (closed).  fp64, products summed x, y, z without contraction.
#define OBB_CHUNK 256
__global__ void obb_membership_kernel(const double* __restrict__ pos, int64_t n, const double* __restrict__ centers,
                                      const double* __restrict__ rot, const double* __restrict__ half, int32_t nb,
                                      int32_t* __restrict__ count, int32_t* __restrict__ first_bb) {
#pragma clang fp contract(off)
    __shared__ double sc[OBB_CHUNK * 3], sr[OBB_CHUNK * 9], sh[OBB_CHUNK * 3];
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double x = 0, y = 0, z = 0;
    if (p < n) { x = pos[3 * p]; y = pos[3 * p + 1]; z = pos[3 * p + 2]; }
    int c = 0, first = -1;
    for (int b0 = 0; b0 < nb; b0 += OBB_CHUNK) {
        const int m = nb - b0 < OBB_CHUNK ? nb - b0 : OBB_CHUNK;
        __syncthreads();
        for (int e = threadIdx.x; e < m * 3; e += blockDim.x) { sc[e] = centers[(int64_t)b0 * 3 + e]; sh[e] = half[(int64_t)b0 * 3 + e]; }
        for (int e = threadIdx.x; e < m * 9; e += blockDim.x) sr[e] = rot[(int64_t)b0 * 9 + e];
        __syncthreads();
        for (int b = 0; b < m; ++b) {
            const double dx = x - sc[3 * b], dy = y - sc[3 * b + 1], dz = z - sc[3 * b + 2];
            bool in = true;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const double v = (sr[9 * b + 3 * j] * dx + sr[9 * b + 3 * j + 1] * dy) + sr[9 * b + 3 * j + 2] * dz;
                in = in && v >= -sh[3 * b + j] && v <= sh[3 * b + j];
            }
            if (in) {
                if (c == 0) first = b0 + b;
                ++c;
            }
        }
    }
    if (p < n) { count[p] = c; first_bb[p] = first; }
}
extern "C" int b2m_obb_membership(const double* pos, int64_t n_pts, const double* centers, const double* rotations,
                                  const double* half_sizes, int32_t n_boxes, int32_t* count, int32_t* first_bb,
                                  void* stream) {
    B2M_CHECK_ARG(pos && count && first_bb && n_pts >= 0 && n_boxes >= 0, "bad arguments");
    B2M_CHECK_ARG(n_boxes == 0 || (centers && rotations && half_sizes), "NULL boxes");
    if (n_pts > 0)
        obb_membership_kernel<<<(unsigned)cdiv64(n_pts, 256), 256, 0, (hipStream_t)stream>>>(pos, n_pts, centers, rotations,
                                                                                             half_sizes, n_boxes, count, first_bb);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}

// rank of every point's segment among the voxel-level segments (-1: the segment has no voxel)
__global__ void seg_rank_kernel(const int64_t* __restrict__ segments, int64_t n, const uint64_t* __restrict__ tkeys,
                                const int32_t* __restrict__ tvals, int64_t mask, int32_t* __restrict__ seg_of_point) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int64_t s = b2m_find(tkeys, mask, (uint64_t)segments[p]);
    seg_of_point[p] = s < 0 ? -1 : tvals[s];
}
extern "C" int b2m_seg_rank(const int64_t* segments, int64_t n_pts, const uint64_t* tkeys, const int32_t* tvals,
                            int64_t cap, int32_t* seg_of_point, void* stream) {
    B2M_CHECK_ARG(segments && tkeys && tvals && seg_of_point && pow2(cap) && n_pts >= 0, "bad arguments");
    if (n_pts > 0)
        seg_rank_kernel<<<(unsigned)cdiv64(n_pts, 256), 256, 0, (hipStream_t)stream>>>(segments, n_pts, tkeys, tvals, cap - 1,
                                                                                       seg_of_point);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}

// Majority vote per segment (scipy.stats.mode over the points of a segment, dataloader.py:262-270, 913-921):
// the most frequent class, the LOWEST class index on ties (classes are numbered in ascending order of their value, which is
// scipy's tie rule).  hist: int32[n_seg * n_class] scratch.
__global__ void seg_class_hist_kernel(const int32_t* __restrict__ seg_of_point, const int32_t* __restrict__ cls, int64_t n,
                                      int32_t n_class, int32_t* __restrict__ hist) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int32_t r = seg_of_point[p];
    if (r >= 0) atomicAdd(&hist[(int64_t)r * n_class + cls[p]], 1);
}
__global__ void seg_mode_pick_kernel(const int32_t* __restrict__ hist, int64_t n_seg, int32_t n_class,
                                     int32_t* __restrict__ mode_cls) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seg) return;
    int best = 0, bc = -1;
    for (int c = 0; c < n_class; ++c) {
        const int v = hist[s * n_class + c];
        if (v > bc) { bc = v; best = c; }
    }
    mode_cls[s] = best;
}
extern "C" int b2m_seg_mode(const int32_t* seg_of_point, const int32_t* cls, int64_t n_pts, int64_t n_seg, int32_t n_class,
                            int32_t* hist, int32_t* mode_cls, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    B2M_CHECK_ARG(seg_of_point && cls && hist && mode_cls && n_pts >= 0 && n_seg >= 0 && n_class >= 1, "bad arguments");
    if (n_seg == 0) return B2M_OK;
    B2M_HIP(hipMemsetAsync(hist, 0, (size_t)n_seg * n_class * sizeof(int32_t), st));
    if (n_pts > 0)
        seg_class_hist_kernel<<<(unsigned)cdiv64(n_pts, 256), 256, 0, st>>>(seg_of_point, cls, n_pts, n_class, hist);
    seg_mode_pick_kernel<<<(unsigned)cdiv64(n_seg, 256), 256, 0, st>>>(hist, n_seg, n_class, mode_cls);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}
