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

// ------------------------------------------------------------------ detection losses, forward + gradients in one launch
// The box and semantics terms of Model.compute_loss_detection in the ScanNet configuration
// (/root/reference/models/model.py:62-88 L1 offset / bounds, :133-176 IoU-target score loss, :194-210 semantics): torch
// evaluates them as ~75 elementwise launches forward and as many backward, all on the critical path between the forward and
// the backward pass of the network.  Here one thread owns one prediction row: the L1 terms, the box IoU against the ground
// truth (set_IOUs' arithmetic), BCE-with-logits against it, the Pearson sums of the logging correlation, cross entropy over
// the class logits and its argmax -- and, in the same pass, the gradient of the weighted total with respect to every head
// output (each is an elementwise function of the row once the normalisers F = foreground rows and n_valid are known).
// Sums in fp64 (block tree, then one atomic per block and value).  out[16] doubles:
//   0 sum |d off|   1 sum |d bounds|   2 sum BCE   3 sum iou   4 sum logit   5 sum iou^2   6 sum logit^2   7 sum iou*logit
//   8 sum CE        9 correct class   10 (unused)
struct LossArgs {
    const float* off; int64_t ld_off; const float* bnd; int64_t ld_bnd; const float* sc; int64_t ld_sc;
    const float* sem; int64_t ld_sem; int C;
    const float* gt_off; const float* gt_bnd; const float* loc;            // (S, 3) dense
    const uint8_t* fg; const int64_t* gt_sem; int64_t S;
    float w_off, w_bnd, w_sc, w_sem, min_bb; double F; const double* n_valid;
    float* d_off; float* d_bnd; float* d_sc; float* d_sem; int64_t* argmax; double* out;
};
__global__ __launch_bounds__(256) void detection_loss_kernel(LossArgs a) {
    __shared__ double red[4][10];
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double v[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (s < a.S) {
        const bool fg = a.fg == nullptr || a.fg[s] != 0;
        const float inv_f = (float)(1.0 / a.F);
        float po[3], pb[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) { po[j] = a.off[s * a.ld_off + j]; pb[j] = a.bnd[s * a.ld_bnd + j]; }
        if (fg) {
            float l_off = 0.f, l_bnd = 0.f, pbox[6], gbox[6];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float go = a.gt_off[s * 3 + j], gb = a.gt_bnd[s * 3 + j], lc = a.loc[s * 3 + j];
                const float d0 = po[j] - go, d1 = pb[j] - gb;
                l_off += fabsf(d0); l_bnd += fabsf(d1);
                a.d_off[s * 3 + j] = d0 > 0.f ? a.w_off * inv_f : (d0 < 0.f ? -a.w_off * inv_f : 0.f);
                a.d_bnd[s * 3 + j] = d1 > 0.f ? a.w_bnd * inv_f : (d1 < 0.f ? -a.w_bnd * inv_f : 0.f);
                const float pc = po[j] + lc, gc = go + lc;                  // voted / true centre (model.py:147-150)
                const float pbc = pb[j] < a.min_bb ? a.min_bb : pb[j];      // torch.clamp(pred_bounds, min=min_bb_size)
                pbox[j] = pc - pbc; pbox[3 + j] = pc + pbc;
                gbox[j] = gc - gb; gbox[3 + j] = gc + gb;
            }
            v[0] = l_off; v[1] = l_bnd;
            if (a.sc) {
                const float iou = box_iou(gbox, pbox);                      // set_IOUs(gt_bbs, pred_bbs)
                const double x = (double)a.sc[s * a.ld_sc], y = (double)iou;
                v[2] = (x > 0 ? x : 0) - x * y + log1p(exp(-fabs(x)));      // BCEWithLogitsLoss
                v[3] = y; v[4] = x; v[5] = y * y; v[6] = x * x; v[7] = x * y;
                a.d_sc[s] = (float)((double)a.w_sc * (1.0 / (1.0 + exp(-x)) - y) / a.F);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 3; ++j) { a.d_off[s * 3 + j] = 0.f; a.d_bnd[s * 3 + j] = 0.f; }
            if (a.sc) a.d_sc[s] = 0.f;
        }
        if (a.sem) {
            const float* z = a.sem + s * a.ld_sem;
            const int64_t t = a.gt_sem[s];
            float mx = z[0]; int am = 0;
            for (int c = 1; c < a.C; ++c) if (z[c] > mx) { mx = z[c]; am = c; }        // first maximum, like torch.argmax
            double den = 0;
            for (int c = 0; c < a.C; ++c) den += exp((double)z[c] - (double)mx);
            a.argmax[s] = am;
            v[9] = (t == am) ? 1.0 : 0.0;
            const bool valid = t >= 0 && t < a.C;                                       // ignore_index = -100
            const double inv_n = 1.0 / *a.n_valid;
            if (valid) v[8] = log(den) - ((double)z[t] - (double)mx);
            for (int c = 0; c < a.C; ++c) {
                const double p = exp((double)z[c] - (double)mx) / den;
                a.d_sem[s * a.C + c] = valid ? (float)((double)a.w_sem * (p - (c == t ? 1.0 : 0.0)) * inv_n) : 0.f;
            }
        }
    }
    // block sum in a fixed tree, one atomic per value and block
#pragma unroll
    for (int u = 0; u < 10; ++u) {
        double t = v[u];
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) t += __shfl_down(t, d, 64);
        v[u] = t;
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int u = 0; u < 10; ++u) red[threadIdx.x >> 6][u] = v[u];
    }
    __syncthreads();
    if (threadIdx.x < 10) {
        const double t = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        if (t != 0.0) atomicAdd(a.out + threadIdx.x, t);
    }
}
// res[0] total, 1 offset_loss, 2 bounds_loss, 3 bb_score_loss, 4 bb_target_scores, 5 bb_scores_correlation, 6 semantics_loss,
// 7 semantics_acc
__global__ void detection_loss_final_kernel(const double* __restrict__ o, double F, const double* __restrict__ n_valid, int64_t S,
                                            float w_off, float w_bnd, float w_sc, float w_sem, int has_sc, int has_sem,
                                            float* __restrict__ res) {
    const double l_off = o[0] / F, l_bnd = o[1] / F;
    double total = (double)w_off * l_off + (double)w_bnd * l_bnd;
    res[1] = (float)l_off; res[2] = (float)l_bnd;
    res[3] = res[4] = res[5] = res[6] = res[7] = 0.f;
    if (has_sc) {
        const double bce = o[2] / F;
        total += (double)w_sc * bce;
        res[3] = (float)bce; res[4] = (float)(o[3] / F);
        const double ca = o[5] - o[3] * o[3] / F, cb = o[6] - o[4] * o[4] / F, cab = o[7] - o[3] * o[4] / F;
        // Pearson r of (IoU, score logit), a logging value (/root/reference/models/model.py:88).  The centred sums can come out a
        // rounding error below zero for (nearly) constant inputs: clamped before the root (a negative product gave NaN -> +-inf
        // through the floor below).  Zero variance -- all IoUs 0 at a fresh initialisation -- reads 0 here, as model._pearsonr
        // does; scipy.stats.pearsonr returns NaN with a warning for such input
        const double va = ca > 0.0 ? ca : 0.0, vb = cb > 0.0 ? cb : 0.0;
        const double den = sqrt(va * vb);
        res[5] = den > 1e-300 ? (float)(cab / den) : 0.f;
    }
    if (has_sem) {
        const double ce = o[8] / *n_valid;
        total += (double)w_sem * ce;
        res[6] = (float)ce; res[7] = (float)(o[9] / (double)S);
    }
    res[0] = (float)total;
}
extern "C" int b2m_detection_loss(const float* off, int64_t ld_off, const float* bnd, int64_t ld_bnd, const float* sc, int64_t ld_sc,
                                  const float* sem, int64_t ld_sem, int32_t n_class, const float* gt_off, const float* gt_bnd,
                                  const float* loc, const uint8_t* fg, const int64_t* gt_sem, int64_t S, double n_fg,
                                  const double* n_valid, float w_off, float w_bnd, float w_sc, float w_sem, float min_bb_size,
                                  float* d_off, float* d_bnd, float* d_sc, float* d_sem, int64_t* argmax, double* sums,
                                  float* result, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    B2M_CHECK_ARG(off && bnd && gt_off && gt_bnd && loc && d_off && d_bnd && sums && result && S >= 1 && n_fg >= 1, "bad arguments");
    B2M_CHECK_ARG(ld_off >= 3 && ld_bnd >= 3 && (!sc || (d_sc && ld_sc >= 1)), "leading dimensions / score gradient");
    B2M_CHECK_ARG(!sem || (d_sem && argmax && gt_sem && n_valid && n_class >= 1 && ld_sem >= n_class), "semantics arguments");
    LossArgs a{};
    a.off = off; a.ld_off = ld_off; a.bnd = bnd; a.ld_bnd = ld_bnd; a.sc = sc; a.ld_sc = ld_sc; a.sem = sem; a.ld_sem = ld_sem;
    a.C = n_class; a.gt_off = gt_off; a.gt_bnd = gt_bnd; a.loc = loc; a.fg = fg; a.gt_sem = gt_sem; a.S = S;
    a.w_off = w_off; a.w_bnd = w_bnd; a.w_sc = w_sc; a.w_sem = w_sem; a.min_bb = min_bb_size; a.F = n_fg; a.n_valid = n_valid;
    a.d_off = d_off; a.d_bnd = d_bnd; a.d_sc = d_sc; a.d_sem = d_sem; a.argmax = argmax; a.out = sums;
    B2M_HIP(hipMemsetAsync(sums, 0, 16 * sizeof(double), st));
    detection_loss_kernel<<<(unsigned)cdiv64(S, 256), 256, 0, st>>>(a);
    detection_loss_final_kernel<<<1, 1, 0, st>>>(sums, n_fg, n_valid, S, w_off, w_bnd, w_sc, w_sem, sc ? 1 : 0, sem ? 1 : 0, result);
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
// Round 5: a wave owns one 64-voxel word of a scene for ALL of the scene's selected rows -- the voxel's segment and foreground slot
// (seg2vox, fg_slot: 12 bytes per voxel) are read once instead of once per row (PMC: 4.8 x the algorithmic bytes before); lane r
// collects row r's word and the wave writes up to 64 rows' words with one instruction.
__global__ __launch_bounds__(256) void mask_project_batch_kernel(const MaskScene* __restrict__ sc, int n_scenes, float th) {
    if ((int)blockIdx.y >= n_scenes) return;
    const MaskScene d = sc[blockIdx.y];
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= d.words || d.ksel == 0) return;
    const int lane = lane_id();
    const int64_t v = w * 64 + lane;
    const int slot = v < d.n_vox ? d.fg_slot[d.seg2vox[v]] : -1;
    for (int64_t r0 = 0; r0 < d.ksel; r0 += 64) {
        const int nr = d.ksel - r0 < 64 ? (int)(d.ksel - r0) : 64;
        uint64_t mine = 0;
        for (int r = 0; r < nr; ++r) {
            const float val = slot >= 0 ? d.heat[(int64_t)d.sel[r0 + r] * d.n_fg + slot] : 0.f;
            const uint64_t m = __ballot(val > th);
            if (lane == r) mine = m;
        }
        if (lane < nr) d.bits[(r0 + lane) * d.words + w] = mine;
    }
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
    mask_project_batch_kernel<<<dim3((unsigned)cdiv64(max_words, 4), (unsigned)n_scenes), 256, 0, (hipStream_t)stream>>>(
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
// ---- the same through a voxel-major image of the kept rows (round 5).  mask_gather_batch_kernel looks every (row, point) up in
// the row's bit image: kk x n_pts eight-byte loads for kk x n_pts bytes of output (99 x 2 M on the benchmark batch: 0.19 ms at 13 % of
// the HBM rate the output alone would need).  Transposed once -- word c of voxel v holds the bits of kept rows 64c .. 64c + 63 at v --
// a point needs ONE load per 64 rows, and the kernel is what it looks like: 8 bytes of index and kk bytes of mask per point.
// tbits of a scene: n_vox x nw words, nw = ceil(kk / 64).
__global__ __launch_bounds__(256) void mask_transpose_batch_kernel(const MaskScene* __restrict__ sc, const int64_t* __restrict__ tptr) {
    const MaskScene d = sc[blockIdx.y];
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);       // a wave owns the 64 voxels of bit word w
    if (w >= d.words || d.kk == 0) return;
    const int lane = lane_id();
    uint64_t* tb = (uint64_t*)tptr[blockIdx.y];
    const int nw = (int)((d.kk + 63) >> 6);
    const int64_t v = w * 64 + lane;
    for (int c = 0; c < nw; ++c) {
        const int64_t r = (int64_t)c * 64 + lane;                          // lane r: word w of kept row r
        const uint64_t mine = r < d.kk ? d.bits[(d.rows ? (int64_t)d.rows[r] : r) * d.words + w] : 0ull;
        uint64_t out = 0;                                                  // lane j: bit r = bit j of lane r's word
#pragma unroll 8
        for (int j = 0; j < 64; ++j) {
            const uint64_t m = __ballot((mine >> j) & 1ull);
            if (lane == j) out = m;
        }
        if (v < d.n_vox) tb[v * nw + c] = out;
    }
}
// a thread owns 16 consecutive output points: their voxels' words once per 64 rows, one 16-byte store per row
__global__ __launch_bounds__(256) void mask_gather_t_batch_kernel(const MaskScene* __restrict__ sc, const int64_t* __restrict__ tptr) {
    const MaskScene d = sc[blockIdx.y];
    const int64_t p0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 16;
    if (p0 >= d.n_pts || d.kk == 0) return;
    // (pointers read from the table are generic to hipcc -- flat_load / flat_store; these three are device memory)
    typedef __attribute__((address_space(1))) const uint64_t* gp_u64;
    typedef __attribute__((address_space(1))) const int64_t* gp_i64;
    typedef __attribute__((address_space(1))) uint8_t* gp_u8;
    const gp_u64 tb = (gp_u64)(uintptr_t)tptr[blockIdx.y];
    const gp_i64 index = (gp_i64)(uintptr_t)d.index;
    const gp_u8 outp = (gp_u8)(uintptr_t)d.out;
    const int nw = (int)((d.kk + 63) >> 6);
    int64_t v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int64_t p = p0 + u < d.n_pts ? p0 + u : d.n_pts - 1;
        v[u] = d.index ? index[p] : p;
    }
    // (a row starts wherever r * n_pts falls: the 16-byte store is an UNALIGNED one -- global_store_dwordx4 takes any address in the
    // default access mode, and hipcc emits it for a 16-byte memcpy to a byte pointer)
    const bool whole = p0 + 15 < d.n_pts;
    for (int c = 0; c < nw; ++c) {
        uint64_t t[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) t[u] = tb[v[u] * nw + c];
        const int nr = d.kk - 64 * c < 64 ? (int)(d.kk - 64 * c) : 64;
        for (int r = 0; r < nr; ++r) {
            uint32_t q[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint32_t x = 0;
#pragma unroll
                for (int u = 0; u < 4; ++u) x |= (uint32_t)((t[4 * g + u] >> r) & 1ull) << (8 * u);
                q[g] = x;
            }
            gp_u8 o = outp + ((int64_t)c * 64 + r) * d.n_pts + p0;
            if (whole) {
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                typedef u32x4 u32x4_unaligned __attribute__((aligned(1)));
                *(__attribute__((address_space(1))) u32x4_unaligned*)o = u32x4{q[0], q[1], q[2], q[3]};
            }
            else {
#pragma unroll
                for (int u = 0; u < 16; ++u) if (p0 + u < d.n_pts) o[u] = (uint8_t)(q[u >> 2] >> (8 * (u & 3)));
            }
        }
    }
}
extern "C" int b2m_mask_gather_batch_t(const int64_t* desc, int32_t n_scenes, int64_t total_kept, int64_t max_pts, int64_t max_words,
                                       const int64_t* tbits, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    B2M_CHECK_ARG(n_scenes >= 0 && n_scenes <= 65535 && total_kept >= 0 && max_pts >= 0 && max_words >= 0, "bad sizes");
    if (n_scenes == 0 || total_kept == 0 || max_pts == 0 || max_words == 0) return B2M_OK;
    B2M_CHECK_ARG(desc && tbits, "NULL argument");
    mask_transpose_batch_kernel<<<dim3((unsigned)cdiv64(max_words, 4), (unsigned)n_scenes), 256, 0, st>>>((const MaskScene*)desc, tbits);
    mask_gather_t_batch_kernel<<<dim3((unsigned)cdiv64(max_pts, 4096), (unsigned)n_scenes), 256, 0, st>>>((const MaskScene*)desc, tbits);
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
