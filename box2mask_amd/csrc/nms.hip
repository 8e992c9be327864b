// Box votes -> instance masks: non-maximum clustering, heat-map projection, mask NMS, label histogram.
// Integer / comparison work; results must be bit-exact against models/iou_nms.py, so floating-point
// contraction is disabled for this file and every IoU is evaluated in torch's operation order.
#include "b2m_common.h"
#pragma clang fp contract(off)

__device__ __forceinline__ float fmax_t(float a, float b) { return a > b ? a : b; }
__device__ __forceinline__ float fmin_t(float a, float b) { return a < b ? a : b; }
__device__ __forceinline__ float clamp0(float x) { return x < 0.f ? 0.f : x; }

// IoU of two axis-aligned boxes given as [min3,max3]; order of operations of torch_IOUs / set_IOUs
// (/root/reference/models/iou_nms.py:4-22,26-45): prod over 3 sides as (s0*s1)*s2,
// union = ((area_a + area_b) - inter) + 1e-6f, result = inter / union (IEEE division).
__device__ __forceinline__ float box_iou(const float* a, const float* b) {
    const float as0 = a[3] - a[0], as1 = a[4] - a[1], as2 = a[5] - a[2];
    const float bs0 = b[3] - b[0], bs1 = b[4] - b[1], bs2 = b[5] - b[2];
    const float i0 = clamp0(fmin_t(a[3], b[3]) - fmax_t(a[0], b[0]));
    const float i1 = clamp0(fmin_t(a[4], b[4]) - fmax_t(a[1], b[1]));
    const float i2 = clamp0(fmin_t(a[5], b[5]) - fmax_t(a[2], b[2]));
    const float inter = (i0 * i1) * i2;
    const float aa = (as0 * as1) * as2, ba = (bs0 * bs1) * bs2;
    const float uni = ((aa + ba) - inter) + 0.000001f;
    return inter / uni;
}

__global__ void set_ious_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n,
                                float* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x[6], y[6];
#pragma unroll
    for (int u = 0; u < 6; ++u) { x[u] = a[i * 6 + u]; y[u] = b[i * 6 + u]; }
    out[i] = box_iou(x, y);
}
extern "C" int b2m_set_ious(const float* a, const float* b, int64_t n, float* out, void* stream) {
    if (n == 0) return B2M_OK;
    B2M_CHECK_ARG(a && b && out && n > 0, "bad arguments");
    set_ious_kernel<<<(unsigned)cdiv64(n, 256), 256, 0, (hipStream_t)stream>>>(a, b, n, out);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}

// ------------------------------------------------------------------ non-maximum clustering
#define NMC_THREADS 1024
#define NMC_LDS_SORT 4096
#define NMC_MAX_N 262144

__device__ __forceinline__ uint32_t f2ord_u(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

template <class P>
__device__ __forceinline__ void bitonic_sort(P keys, int npad) {
    for (int k = 2; k <= npad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < (npad >> 1); t += NMC_THREADS) {
                const int i = ((t / j) * 2 * j) + (t % j), l = i + j;
                const uint64_t a = keys[i], b = keys[l];
                const bool up = (i & k) == 0;
                if ((a > b) == up) { keys[i] = b; keys[l] = a; }
            }
            __syncthreads();
        }
    }
}

// one workgroup clusters the n boxes of one scene
__device__ __forceinline__ void nmc_scene(const float* __restrict__ boxes, int n, float th, int max_k,
                                          int32_t* __restrict__ reps, int32_t* __restrict__ assign,
                                          float* __restrict__ heat, int32_t* __restrict__ k_out,
                                          uint64_t* __restrict__ order, int npad) {
    __shared__ uint64_t skeys[NMC_LDS_SORT];
    __shared__ uint32_t alive[NMC_MAX_N / 32];
    __shared__ int s_first;
    __shared__ float s_box[6];
    const int tid = threadIdx.x;
    const bool in_lds = npad <= NMC_LDS_SORT;

    // ---- visiting order: descending score, ties by ascending row
    for (int j = tid; j < npad; j += NMC_THREADS) {
        uint64_t key = ~0ull;
        if (j < n) key = ((uint64_t)f2ord_u(-boxes[(int64_t)j * 7]) << 32) | (uint32_t)j;
        if (in_lds) skeys[j] = key; else order[j] = key;
    }
    for (int w = tid; w < (n + 31) / 32; w += NMC_THREADS) {
        const int rem = n - w * 32;
        alive[w] = rem >= 32 ? 0xFFFFFFFFu : ((1u << rem) - 1u);
    }
    if (tid == 0) s_first = 0x7FFFFFFF;
    __syncthreads();
    if (in_lds) {
        bitonic_sort(skeys, npad);
        for (int j = tid; j < npad; j += NMC_THREADS) order[j] = skeys[j];
    } else {
        bitonic_sort(order, npad);
    }
    __syncthreads();
    auto ord = [&](int p) -> int { return (int)(uint32_t)(in_lds ? skeys[p] : order[p]); };
    auto is_alive = [&](int j) -> bool { return (alive[j >> 5] >> (j & 31)) & 1u; };

    int kc = 0, p0 = 0;
    for (;;) {
        // ---- next representative: first alive box in visiting order
        int pf;
        for (;;) {
            const int p = p0 + tid;
            if (p < n && is_alive(ord(p))) atomicMin(&s_first, p);
            __syncthreads();
            pf = s_first;
            __syncthreads();                   // every wave has read s_first before any wave publishes the next window
            if (pf != 0x7FFFFFFF || p0 + NMC_THREADS >= n) break;
            p0 += NMC_THREADS;
        }
        if (pf == 0x7FFFFFFF) break;
        const int r = ord(pf);
        if (tid < 6) s_box[tid] = boxes[(int64_t)r * 7 + 1 + tid];
        __syncthreads();                       // s_box visible; everyone has read s_first
        if (tid == 0) { s_first = 0x7FFFFFFF; reps[kc] = r; }
        float rb[6];
#pragma unroll
        for (int u = 0; u < 6; ++u) rb[u] = s_box[u];
        for (int j = tid; j < n; j += NMC_THREADS) {
            float bj[6];
#pragma unroll
            for (int u = 0; u < 6; ++u) bj[u] = boxes[(int64_t)j * 7 + 1 + u];
            float iou = box_iou(rb, bj);
            if (j == r) iou = 1.0f;            // iou_nms.py:90
            if (kc < max_k) heat[(int64_t)kc * n + j] = iou;
            if (is_alive(j) && !(iou <= th)) { // iou_nms.py:97-100 (`<=` keeps the box in `remaining`)
                assign[j] = kc;
                atomicAnd(&alive[j >> 5], ~(1u << (j & 31)));
            }
        }
        ++kc;
        p0 = pf + 1;
        __syncthreads();
    }
    if (tid == 0) *k_out = kc;
}
__global__ __launch_bounds__(NMC_THREADS) void nmc_kernel(const float* __restrict__ boxes, int n, float th, int max_k,
                                                          int32_t* __restrict__ reps, int32_t* __restrict__ assign,
                                                          float* __restrict__ heat, int32_t* __restrict__ k_out,
                                                          uint64_t* __restrict__ order, int npad) {
    nmc_scene(boxes, n, th, max_k, reps, assign, heat, k_out, order, npad);
}
// all scenes of a batch in one launch: workgroup s clusters scene s (descriptor row s = box_off, n, npad, max_k,
// heat_off, order_off; boxes / reps / assign are the per-scene arrays back to back)
#define NMC_DESC 6
__global__ __launch_bounds__(NMC_THREADS) void nmc_batch_kernel(const float* __restrict__ boxes, const int64_t* __restrict__ desc,
                                                                float th, int32_t* __restrict__ reps,
                                                                int32_t* __restrict__ assign, float* __restrict__ heat,
                                                                int32_t* __restrict__ k_out, uint64_t* __restrict__ order) {
    const int64_t* d = desc + (int64_t)blockIdx.x * NMC_DESC;
    const int64_t box_off = d[0];
    const int n = (int)d[1], npad = (int)d[2], max_k = (int)d[3];
    if (n == 0) {
        if (threadIdx.x == 0) k_out[blockIdx.x] = 0;
        return;
    }
    nmc_scene(boxes + box_off * 7, n, th, max_k, reps + box_off, assign + box_off, heat + d[4], k_out + blockIdx.x,
              order + d[5], npad);
}
extern "C" int b2m_nmc_batch(const float* boxes, const int64_t* desc, int32_t n_scenes, int32_t max_n, float cluster_th,
                             int32_t* reps, int32_t* assign, float* heat, int32_t* k_out, uint64_t* order, void* stream) {
    B2M_CHECK_ARG(n_scenes >= 0 && max_n >= 0 && max_n <= NMC_MAX_N, "every scene must have <= 262144 boxes");
    B2M_CHECK_ARG(cluster_th > 0.f && cluster_th < 1.f, "cluster_th must be in (0,1)");   // iou_nms.py:71
    if (n_scenes == 0) return B2M_OK;
    B2M_CHECK_ARG(desc && k_out && (max_n == 0 || (boxes && reps && assign && order)), "NULL argument");
    nmc_batch_kernel<<<(unsigned)n_scenes, NMC_THREADS, 0, (hipStream_t)stream>>>(boxes, desc, cluster_th, reps, assign, heat,
                                                                                  k_out, order);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}

extern "C" int b2m_nmc(const float* boxes, int32_t n, float cluster_th, int32_t max_k, int32_t* reps, int32_t* assign,
                       float* heat, int32_t* k_out, uint64_t* order, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    B2M_CHECK_ARG(boxes && reps && assign && k_out && order && (heat || max_k == 0), "NULL argument");
    B2M_CHECK_ARG(n >= 0 && n <= NMC_MAX_N, "n must be <= 262144");
    B2M_CHECK_ARG(cluster_th > 0.f && cluster_th < 1.f, "cluster_th must be in (0,1)");   // iou_nms.py:71
    if (n == 0) { B2M_HIP(hipMemsetAsync(k_out, 0, sizeof(int32_t), st)); return B2M_OK; }
    int npad = 1;
    while (npad < n) npad <<= 1;
    if (npad < 2) npad = 2;
    nmc_kernel<<<1, NMC_THREADS, 0, st>>>(boxes, n, cluster_th, max_k, reps, assign, heat, k_out, order, npad);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}

// ------------------------------------------------------------------ heat-maps -> voxel bit masks
__global__ __launch_bounds__(256) void mask_project_kernel(const float* __restrict__ heat, int n_fg,
                                                           const int32_t* __restrict__ sel,
                                                           const int32_t* __restrict__ fg_slot,
                                                           const int64_t* __restrict__ seg2vox, int64_t n_vox, float th,
                                                           uint64_t* __restrict__ bits, int64_t words) {
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int r = blockIdx.y;
    if (w >= words) return;
    const int64_t v = w * 64 + lane_id();
    bool bit = false;
    if (v < n_vox) {
        const int slot = fg_slot[seg2vox[v]];
        const float val = slot >= 0 ? heat[(int64_t)sel[r] * n_fg + slot] : 0.f;
        bit = val > th;
    }
    const uint64_t m = __ballot(bit);
    if (lane_id() == 0) bits[(int64_t)r * words + w] = m;
}
extern "C" int b2m_mask_project(const float* heat, int32_t n_fg, const int32_t* sel, int32_t ksel,
                                const int32_t* fg_slot, const int64_t* seg2vox, int64_t n_vox, float mask_bin_th,
                                uint64_t* bits, int64_t words, void* stream) {
    B2M_CHECK_ARG(ksel >= 0 && words == cdiv64(n_vox, 64), "bad sizes");
    if (ksel == 0 || n_vox == 0) return B2M_OK;        // no cluster passed the score filter: nothing to project
    B2M_CHECK_ARG(heat && sel && fg_slot && seg2vox && bits && n_fg > 0, "bad arguments");
    dim3 grid((unsigned)cdiv64(words, 4), (unsigned)ksel);
    mask_project_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(heat, n_fg, sel, fg_slot, seg2vox, n_vox, mask_bin_th,
                                                               bits, words);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}

// ------------------------------------------------------------------ mask NMS
__global__ __launch_bounds__(256) void mask_inter_kernel(const uint64_t* __restrict__ bits, int k, int64_t words,
                                                         int32_t* __restrict__ inter) {
    const int i = blockIdx.y, j = blockIdx.x;
    if (j < i) return;
    __shared__ int wsum[4];
    int s = 0;
    const uint64_t* a = bits + (int64_t)i * words;
    const uint64_t* b = bits + (int64_t)j * words;
    for (int64_t w = threadIdx.x; w < words; w += 256) s += __popcll(a[w] & b[w]);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d, 64);
    if (lane_id() == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int t = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        inter[(int64_t)i * k + j] = t;
        inter[(int64_t)j * k + i] = t;
    }
}
__global__ __launch_bounds__(256) void mask_greedy_kernel(const int32_t* __restrict__ inter, int k, float th,
                                                          int32_t* __restrict__ keep, int32_t* __restrict__ n_keep) {
    // keep[] doubles as the alive flag: 1 = still remaining / kept, 0 = suppressed
    for (int j = threadIdx.x; j < k; j += 256) keep[j] = 1;
    __syncthreads();
    int nk = 0;
    for (int i = 0; i < k; ++i) {
        if (keep[i]) {                                           // uniform: all threads read the same word
            ++nk;
            const int ci = inter[(int64_t)i * k + i];
            for (int j = i + 1 + threadIdx.x; j < k; j += 256) {
                if (!keep[j]) continue;
                const int in = inter[(int64_t)i * k + j];
                const int un = ci + inter[(int64_t)j * k + j] - in;
                const float iou = (float)in / (float)un;         // int64 true-division -> float32 in torch
                if (!(iou <= th)) keep[j] = 0;                   // iou_nms.py:138 keeps `ious <= th`
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_keep = nk;
}
extern "C" int b2m_mask_nms(const uint64_t* bits, int32_t k, int64_t words, float th, int32_t* inter, int32_t* keep,
                            int32_t* n_keep, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    B2M_CHECK_ARG(bits && inter && keep && n_keep && k >= 0 && k <= 65535 && words >= 0, "bad arguments");
    if (k == 0) { B2M_HIP(hipMemsetAsync(n_keep, 0, sizeof(int32_t), st)); return B2M_OK; }
    mask_inter_kernel<<<dim3((unsigned)k, (unsigned)k), 256, 0, st>>>(bits, k, words, inter);
    mask_greedy_kernel<<<1, 256, 0, st>>>(inter, k, th, keep, n_keep);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}

// ------------------------------------------------------------------ per-instance label histogram
__global__ __launch_bounds__(256) void label_hist_kernel(const uint64_t* __restrict__ bits, int64_t words,
                                                         const int32_t* __restrict__ rows,
                                                         const int32_t* __restrict__ sem, int64_t n_vox, int n_class,
                                                         int32_t* __restrict__ labels) {
    __shared__ int hist[256];
    const int r = blockIdx.x;
    const int64_t row = rows ? rows[r] : r;
    hist[threadIdx.x] = 0;
    __syncthreads();
    for (int64_t w = threadIdx.x; w < words; w += 256) {
        uint64_t m = bits[row * words + w];
        while (m) {
            const int b = __builtin_ctzll(m);
            m &= m - 1;
            const int64_t v = w * 64 + b;
            if (v < n_vox) {
                const int c = sem[v];
                if (c >= 0 && c < n_class) atomicAdd(&hist[c], 1);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int best = 0, bc = hist[0];
        for (int c = 1; c < n_class; ++c) if (hist[c] > bc) { bc = hist[c]; best = c; }   // first maximum (np.argmax)
        labels[r] = best;
    }
}
extern "C" int b2m_label_hist(const uint64_t* bits, int64_t words, const int32_t* rows, int32_t k, const int32_t* sem,
                              int64_t n_vox, int32_t n_class, int32_t* labels, void* stream) {
    B2M_CHECK_ARG(k >= 0 && n_class >= 1 && n_class <= 256, "bad sizes (n_class <= 256)");
    if (k == 0) return B2M_OK;
    B2M_CHECK_ARG(bits && sem && labels, "NULL argument");
    label_hist_kernel<<<(unsigned)k, 256, 0, (hipStream_t)stream>>>(bits, words, rows, sem, n_vox, n_class, labels);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}

// ------------------------------------------------------------------ full label histogram of every mask row
// hist[r][c] = |{v in mask r : label[v] == c}|: the prediction x ground-truth-instance intersection counts of
// assign_instances_for_scan (/root/reference/utils/eval_metric.py:316-330), all pairs in one pass over the bits.
#define HIST_MAX 2048
__global__ __launch_bounds__(256) void mask_hist_kernel(const uint64_t* __restrict__ bits, int64_t words,
                                                        const int32_t* __restrict__ label, int64_t n, int n_class,
                                                        int64_t words_per_block, int32_t* __restrict__ hist) {
    __shared__ int h[HIST_MAX];
    const int r = blockIdx.x;
    for (int c = threadIdx.x; c < n_class; c += 256) h[c] = 0;
    __syncthreads();
    const int64_t w0 = (int64_t)blockIdx.y * words_per_block;
    int64_t w1 = w0 + words_per_block;
    if (w1 > words) w1 = words;
    for (int64_t w = w0 + threadIdx.x; w < w1; w += 256) {
        uint64_t m = bits[(int64_t)r * words + w];
        while (m) {
            const int b = __builtin_ctzll(m);
            m &= m - 1;
            const int64_t v = w * 64 + b;
            if (v < n) {
                const int c = label[v];
                if (c >= 0 && c < n_class) atomicAdd(&h[c], 1);
            }
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < n_class; c += 256)
        if (h[c]) atomicAdd(&hist[(int64_t)r * n_class + c], h[c]);
}
extern "C" int b2m_mask_hist(const uint64_t* bits, int64_t words, int32_t k, const int32_t* label, int64_t n,
                             int32_t n_class, int32_t* hist, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    B2M_CHECK_ARG(k >= 0 && k <= 65535 && n_class >= 1 && n_class <= HIST_MAX && n >= 0, "bad sizes (n_class <= 2048)");
    if (k == 0) return B2M_OK;
    B2M_CHECK_ARG(bits && label && hist, "NULL argument");
    B2M_HIP(hipMemsetAsync(hist, 0, (size_t)k * n_class * sizeof(int32_t), st));
    if (words == 0) return B2M_OK;
    const int64_t wpb = 2048;
    mask_hist_kernel<<<dim3((unsigned)k, (unsigned)cdiv64(words, wpb)), 256, 0, st>>>(bits, words, label, n, n_class, wpb, hist);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}

// ------------------------------------------------------------------ bit rows -> byte masks through an index
__global__ void mask_gather_kernel(const uint64_t* __restrict__ bits, int64_t words, const int32_t* __restrict__ rows,
                                   const int64_t* __restrict__ index, int64_t n_pts, uint8_t* __restrict__ out) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    if (p >= n_pts) return;
    const int64_t row = rows ? rows[r] : r;
    const int64_t v = index ? index[p] : p;
    out[(int64_t)r * n_pts + p] = (uint8_t)((bits[row * words + (v >> 6)] >> (v & 63)) & 1ull);
}
extern "C" int b2m_mask_gather(const uint64_t* bits, int64_t words, const int32_t* rows, int32_t k,
                               const int64_t* index, int64_t n_pts, uint8_t* out, void* stream) {
    B2M_CHECK_ARG(k >= 0 && k <= 65535 && n_pts >= 0, "bad sizes");
    if (k == 0 || n_pts == 0) return B2M_OK;            // empty selection: nothing to write (out may be NULL)
    B2M_CHECK_ARG(bits && out, "NULL argument");
    mask_gather_kernel<<<dim3((unsigned)cdiv64(n_pts, 256), (unsigned)k), 256, 0, (hipStream_t)stream>>>(
        bits, words, rows, index, n_pts, out);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}

// ------------------------------------------------------------------ bool rows -> bit rows
__global__ __launch_bounds__(256) void mask_pack_kernel(const uint8_t* __restrict__ masks, int64_t n,
                                                        uint64_t* __restrict__ bits, int64_t words) {
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int r = blockIdx.y;
    if (w >= words) return;
    const int64_t v = w * 64 + lane_id();
    const bool bit = v < n && masks[(int64_t)r * n + v] != 0;
    const uint64_t m = __ballot(bit);
    if (lane_id() == 0) bits[(int64_t)r * words + w] = m;
}
extern "C" int b2m_mask_pack(const uint8_t* masks, int32_t k, int64_t n, uint64_t* bits, int64_t words, void* stream) {
    B2M_CHECK_ARG(k >= 0 && k <= 65535 && words == cdiv64(n, 64), "bad sizes");
    if (k == 0 || n == 0) return B2M_OK;
    B2M_CHECK_ARG(masks && bits, "NULL argument");
    mask_pack_kernel<<<dim3((unsigned)cdiv64(words, 4), (unsigned)k), 256, 0, (hipStream_t)stream>>>(masks, n, bits, words);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}

// ------------------------------------------------------------------ the mask stages for ALL scenes of a batch
// detection2mask walks the scenes of a batch (/root/reference/models/detection_net.py:390-477); per scene the kernels above
// are a few microseconds of work behind a launch each (8 scenes: 32 launches, label_hist 0.42 ms at 16 GB/s).  Here one
// launch per stage covers every (scene, instance) row: a table of B2M_MASK_DESC int64 fields per scene (include/b2m.h) holds
// the scene's pointers and sizes, a block finds its scene from the running row counts in the table.
struct MaskScene {
    const float* heat; int64_t n_fg; const int32_t* sel; int64_t ksel; const int32_t* fg_slot; const int64_t* seg2vox;
    int64_t n_vox; uint64_t* bits; int64_t words; int32_t* inter; int32_t* keep; const int32_t* rows; int64_t kk;
    const int32_t* sem; int32_t* labels; const int64_t* index; int64_t n_pts; uint8_t* out; int64_t row0_sel; int64_t row0_kept;
};
static_assert(sizeof(MaskScene) == B2M_MASK_DESC * 8, "MaskScene must match B2M_MASK_DESC");
// scene of global row r (rows of scene s: [row0, row0 + count)), or -1
__device__ __forceinline__ int scene_of_row(const MaskScene* sc, int n_scenes, int64_t r, bool kept, int64_t& local) {
    for (int s = 0; s < n_scenes; ++s) {
        const int64_t r0 = kept ? sc[s].row0_kept : sc[s].row0_sel, cnt = kept ? sc[s].kk : sc[s].ksel;
        if (r >= r0 && r < r0 + cnt) { local = r - r0; return s; }
    }
    return -1;
}
__global__ __launch_bounds__(256) void mask_project_batch_kernel(const MaskScene* __restrict__ sc, int n_scenes, float th) {
    int64_t r;
    const int s = scene_of_row(sc, n_scenes, blockIdx.y, false, r);
    if (s < 0) return;
    const MaskScene d = sc[s];
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= d.words) return;
    const int64_t v = w * 64 + lane_id();
    bool bit = false;
    if (v < d.n_vox) {
        const int slot = d.fg_slot[d.seg2vox[v]];
        const float val = slot >= 0 ? d.heat[(int64_t)d.sel[r] * d.n_fg + slot] : 0.f;
        bit = val > th;
    }
    const uint64_t m = __ballot(bit);
    if (lane_id() == 0) d.bits[r * d.words + w] = m;
}
__global__ __launch_bounds__(256) void mask_inter_batch_kernel(const MaskScene* __restrict__ sc) {
    const MaskScene d = sc[blockIdx.z];
    const int i = blockIdx.y, j = blockIdx.x, k = (int)d.ksel;
    if (i >= k || j >= k || j < i || d.keep == nullptr) return;
    __shared__ int wsum[4];
    int s = 0;
    const uint64_t* a = d.bits + (int64_t)i * d.words;
    const uint64_t* b = d.bits + (int64_t)j * d.words;
    for (int64_t w = threadIdx.x; w < d.words; w += 256) s += __popcll(a[w] & b[w]);
#pragma unroll
    for (int e = 32; e > 0; e >>= 1) s += __shfl_down(s, e, 64);
    if (lane_id() == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int t = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        d.inter[(int64_t)i * k + j] = t;
        d.inter[(int64_t)j * k + i] = t;
    }
}
__global__ __launch_bounds__(256) void mask_greedy_batch_kernel(const MaskScene* __restrict__ sc, float th) {
    const MaskScene d = sc[blockIdx.x];
    const int k = (int)d.ksel;
    if (d.keep == nullptr || k == 0) return;
    int32_t* keep = d.keep;
    const int32_t* inter = d.inter;
    for (int j = threadIdx.x; j < k; j += 256) keep[j] = 1;
    __syncthreads();
    for (int i = 0; i < k; ++i) {
        if (keep[i]) {
            const int ci = inter[(int64_t)i * k + i];
            for (int j = i + 1 + threadIdx.x; j < k; j += 256) {
                if (!keep[j]) continue;
                const int in = inter[(int64_t)i * k + j];
                const int un = ci + inter[(int64_t)j * k + j] - in;
                const float iou = (float)in / (float)un;
                if (!(iou <= th)) keep[j] = 0;
            }
        }
        __syncthreads();
    }
}
// label histogram of a mask row.  A wave reads 64 consecutive words of the bit row at once (512 contiguous bytes) and then
// works only on the non-zero ones, all 64 lanes on the 64 voxels of one word (a coalesced read of their labels): a thread
// per word walked its set bits one by one behind a 4-byte gather each (round 2: 0.42 ms per batch at 16 GB/s).
__global__ __launch_bounds__(256) void label_hist_batch_kernel(const MaskScene* __restrict__ sc, int n_scenes, int n_class) {
    __shared__ int hist[256];
    int64_t r;
    const int s = scene_of_row(sc, n_scenes, blockIdx.x, true, r);
    if (s < 0) return;
    const MaskScene d = sc[s];
    const int64_t row = d.rows ? d.rows[r] : r;
    hist[threadIdx.x] = 0;
    __syncthreads();
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const uint64_t* brow = d.bits + row * d.words;
    for (int64_t w0 = (int64_t)wave * 64; w0 < d.words; w0 += 256) {
        const uint64_t mine = w0 + lane < d.words ? brow[w0 + lane] : 0ull;
        uint64_t nz = __ballot(mine != 0ull);
        while (nz) {
            const int l = __builtin_ctzll(nz);
            nz &= nz - 1;
            const uint64_t m = (uint64_t)__shfl((long long)mine, l, 64);
            const int64_t v = (w0 + l) * 64 + lane;
            if (((m >> lane) & 1ull) && v < d.n_vox) {
                const int c = d.sem[v];
                if (c >= 0 && c < n_class) atomicAdd(&hist[c], 1);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int best = 0, bc = hist[0];
        for (int c = 1; c < n_class; ++c) if (hist[c] > bc) { bc = hist[c]; best = c; }   // first maximum (np.argmax)
        d.labels[r] = best;
    }
}
// bit rows -> byte masks of ALL kept rows of a scene: a thread owns four consecutive output points, reads their voxel
// indices ONCE (the per-row kernel re-read the 8-byte index for every instance: 99 x 9.6 MB on the benchmark batch) and
// writes one 4-byte word per row
__global__ __launch_bounds__(256) void mask_gather_batch_kernel(const MaskScene* __restrict__ sc) {
    const MaskScene d = sc[blockIdx.y];
    const int64_t p0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (p0 >= d.n_pts || d.kk == 0) return;
    int64_t v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t p = p0 + u < d.n_pts ? p0 + u : d.n_pts - 1;
        v[u] = d.index ? d.index[p] : p;
    }
    const bool whole = p0 + 3 < d.n_pts && ((d.n_pts & 3) == 0);     // the row pitch keeps 4-byte alignment
    for (int64_t r = 0; r < d.kk; ++r) {
        const uint64_t* brow = d.bits + (d.rows ? (int64_t)d.rows[r] : r) * d.words;
        uint32_t w = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) w |= (uint32_t)((brow[v[u] >> 6] >> (v[u] & 63)) & 1ull) << (8 * u);
        uint8_t* o = d.out + r * d.n_pts + p0;
        if (whole) *(uint32_t*)o = w;
        else {
#pragma unroll
            for (int u = 0; u < 4; ++u) if (p0 + u < d.n_pts) o[u] = (uint8_t)(w >> (8 * u));
        }
    }
}
extern "C" int b2m_mask_project_batch(const int64_t* desc, int32_t n_scenes, int64_t total_sel, int64_t max_words,
                                      float mask_bin_th, void* stream) {
    B2M_CHECK_ARG(n_scenes >= 0 && total_sel >= 0 && total_sel <= 65535 && max_words >= 0, "bad sizes (at most 65535 rows per batch)");
    if (n_scenes == 0 || total_sel == 0 || max_words == 0) return B2M_OK;
    B2M_CHECK_ARG(desc, "NULL argument");
    mask_project_batch_kernel<<<dim3((unsigned)cdiv64(max_words, 4), (unsigned)total_sel), 256, 0, (hipStream_t)stream>>>(
        (const MaskScene*)desc, n_scenes, mask_bin_th);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}
extern "C" int b2m_mask_nms_batch(const int64_t* desc, int32_t n_scenes, int32_t max_k, float th, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    B2M_CHECK_ARG(n_scenes >= 0 && n_scenes <= 65535 && max_k >= 0 && max_k <= 65535, "bad sizes");
    if (n_scenes == 0 || max_k == 0) return B2M_OK;
    B2M_CHECK_ARG(desc, "NULL argument");
    mask_inter_batch_kernel<<<dim3((unsigned)max_k, (unsigned)max_k, (unsigned)n_scenes), 256, 0, st>>>((const MaskScene*)desc);
    mask_greedy_batch_kernel<<<(unsigned)n_scenes, 256, 0, st>>>((const MaskScene*)desc, th);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}
extern "C" int b2m_label_hist_batch(const int64_t* desc, int32_t n_scenes, int64_t total_kept, int32_t n_class, void* stream) {
    B2M_CHECK_ARG(n_scenes >= 0 && total_kept >= 0 && n_class >= 1 && n_class <= 256, "bad sizes (n_class <= 256)");
    if (n_scenes == 0 || total_kept == 0) return B2M_OK;
    B2M_CHECK_ARG(desc, "NULL argument");
    label_hist_batch_kernel<<<(unsigned)total_kept, 256, 0, (hipStream_t)stream>>>((const MaskScene*)desc, n_scenes, n_class);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}
extern "C" int b2m_mask_gather_batch(const int64_t* desc, int32_t n_scenes, int64_t total_kept, int64_t max_pts, void* stream) {
    B2M_CHECK_ARG(n_scenes >= 0 && n_scenes <= 65535 && total_kept >= 0 && max_pts >= 0, "bad sizes");
    if (n_scenes == 0 || total_kept == 0 || max_pts == 0) return B2M_OK;
    B2M_CHECK_ARG(desc, "NULL argument");
    mask_gather_batch_kernel<<<dim3((unsigned)cdiv64(max_pts, 1024), (unsigned)n_scenes), 256, 0, (hipStream_t)stream>>>(
        (const MaskScene*)desc);
    B2M_LAUNCH_CHECK();
    return B2M_OK;
}
