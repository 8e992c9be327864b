/*
 * b2m.h — C ABI of the MI355X (gfx950) hot path of Box2Mask.
 *
 * The reference reaches this path through a *Python* surface (MinkowskiEngine 0.5.4 operators and
 * models/iou_nms.py); the functions below are what a binding for that surface calls.  Each entry
 * cites the reference interface it replaces (paths relative to /root/reference; [ME] = behaviour
 * of the un-vendored MinkowskiEngine 0.5.4 dependency, docs/installation.md:6,42).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in `_host`
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it; no function
 *     synchronises unless documented ("syncs")
 *   - no allocation inside: the caller owns every buffer, including scratch
 *   - return value: 0 = ok, negative = error (b2m_last_error() gives a thread-local message)
 *   - coordinates: int32 rows [b,x,y,z], all fields in [0, 65535]; features: fp32 row-major with
 *     an explicit leading dimension (in floats)
 *   - B2M_TILE = 64 output rows is the unit of the "tile rulebook" 
 */
#ifndef B2M_H
#define B2M_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define B2M_TILE 64
#define B2M_OK 0
#define B2M_ERR_ARG (-1)
#define B2M_ERR_HIP (-2)
#define B2M_ERR_UNSUPPORTED (-3)

const char* b2m_last_error(void);
int b2m_version(void);
/* 1 if the library was built for gfx950 and a device is present */
int b2m_device_ok(void);
/* The B2M_* tuning switches are read from the environment once per process; this makes the next call re-read them
 * (tests and A/B tools that change a switch in-process).  Always returns 0. */
int b2m_reload_env(void);

/* ---------------------------------------------------------------- coordinate maps (integer) */

/* Hash table of a coordinate set.  Replaces [ME] CoordinateMap insert reached from
 * ME.SparseTensor(feats, coords) — models/model.py:43, models/detection_net.py:499,503.
 * keys[cap] (uint64) and vals[cap] (int32), cap a power of two >= 2*n.  After the call
 * vals[slot(key)] = smallest row index carrying that key (row i for unique coordinates).
 * dup_count (1 int32, may be NULL) receives the number of duplicate rows. */
int b2m_coords_build(const int32_t* coords, int64_t n, uint64_t* keys, int32_t* vals, int64_t cap,
                     int32_t* dup_count, void* stream);

/* Morton (Z-order) key per coordinate row: batch index in bits 48.., x/y/z bits interleaved below.  Sorting the rows
 * of a batch by this key before building the maps makes 64-row tiles spatially compact (denser rulebook groups,
 * fewer active offsets per tile, better L2 locality of the gathers); row order is otherwise free ([ME] gives no
 * order guarantee beyond input order, which the host restores at the boundary).  Requires b < 32768. */
int b2m_morton_keys(const int32_t* coords, int64_t n, int64_t* keys, void* stream);
/* Hilbert-curve key per coordinate row, same layout (batch index in bits 48.., 3 x `bits` index bits below; every coordinate
 * < 2^bits, bits <= 16): the default row order -- a curve without the Z-order's jumps gives more compact tiles (denser
 * rulebook groups, fewer active offsets per tile). */
int b2m_hilbert_keys(const int32_t* coords, int64_t n, int32_t bits, int64_t* keys, void* stream);

/* Stable argsort of n 64-bit keys (LSD radix sort, 8 bits per pass, only over the bytes in which `bit_mask` has a bit: the bits
 * that can differ between keys; ~0 = all eight): perm[j] = index of the j-th smallest key, inv_perm[perm[j]] = j (int64, ready
 * to index torch tensors).  scratch: b2m_radix_argsort_scratch(n) bytes.  Orders the rows of a batch by Morton key
 * (CoordinateManager(reorder=True)); replaces torch.argsort (rocprim) on the product path. */
int64_t b2m_radix_argsort_scratch(int64_t n);
int b2m_radix_argsort(const uint64_t* keys, int64_t n, uint64_t bit_mask, int64_t* perm, int64_t* inv_perm, void* scratch,
                      void* stream);

/* Strided (kernel 2, stride 2) coordinate generation: out = floor(c / 2ts) * 2ts, unique, rows in
 * order of first occurrence.  Replaces [ME] stride() reached from the seven k=2,s=2 convolutions,
 * models/detection_net.py:42,48,54,61,68,74,81.
 *   coords_out[n*4]   coarse coordinates (first *n_out_host rows valid)
 *   parent[n]         coarse row of every fine row
 *   koff[n]           kernel offset index of the fine row inside its parent, ox + 2*oy + 4*oz
 *   keys/vals[cap]    hash table of the coarse level (key -> coarse row), cap >= 2*n pow2
 *   scratch           int32[2*n + n/1024 + 2]
 * Syncs the stream once to return the coarse row count in *n_out_host. */
int b2m_coords_stride(const int32_t* coords, int64_t n, int32_t ts,
                      int32_t* coords_out, int32_t* parent, int32_t* koff,
                      uint64_t* keys, int32_t* vals, int64_t cap,
                      int32_t* scratch, int64_t* n_out_host, void* stream);

/* Stride-1 kernel map as a neighbour table: nbr[k*ld + o] = row of coords[o] + offset_k, or -1.
 * offsets: odd ksize centred, x fastest (k = (dx+h) + ks*(dy+h) + ks^2*(dz+h)), times ts.
 * Replaces [ME] kernel_map() for kernel_size 3 / 5, stride 1 — models/resnet.py:61-65,
 * models/detection_net.py:37.  ld >= n.  occ/dim_*: see b2m_occupancy (NULL/0 = probe every offset). */
int b2m_kernel_map(const int32_t* coords, int64_t n, int32_t ksize, int32_t ts,
                   const uint64_t* keys, const int32_t* vals, int64_t cap,
                   const uint64_t* occ, int32_t dim_x, int32_t dim_y, int32_t dim_z,
                   int32_t* nbr, int64_t ld, void* stream);

/* b2m_kernel_map followed by b2m_rulebook in one pass, without the K x n neighbour table: the tile rulebook
 * (rb_in int32[K*ntiles*64], rb_out uint8[same], rb_cnt int32[b2m_rulebook_cnt_size], ntiles = ceil(n/64)) of the stride-1
 * kernel map of an odd cubic kernel.  Bit-identical to the two-step path. */
int b2m_kernel_map_rulebook(const int32_t* coords, int64_t n, int32_t ksize, int32_t ts,
                            const uint64_t* keys, const int32_t* vals, int64_t cap,
                            const uint64_t* occ, int32_t dim_x, int32_t dim_y, int32_t dim_z,
                            int32_t* rb_in, uint8_t* rb_out, int32_t* rb_cnt, void* stream);

/* Occupancy bitmap of a stride-1 coordinate set: bit ((b*dim_z + z)*dim_y + y)*dim_x + x.  Optional accelerator of
 * b2m_kernel_map (occ != NULL, ts == 1): offsets whose bit is clear are answered without touching the hash table.
 * All coordinates must lie in [0,dim_*) and b in [0,batches); words >= ceil(batches*dim_x*dim_y*dim_z / 64) + 1. */
int b2m_occupancy(const int32_t* coords, int64_t n, int32_t batches, int32_t dim_x, int32_t dim_y, int32_t dim_z,
                  uint64_t* bits, int64_t words, void* stream);

/* k2s2 maps from (parent, koff):  child[k*ld_c + o] = fine row (table over coarse rows, used by the
 * strided convolution) and up[k*ld_f + i] = parent[i] iff koff[i]==k (table over fine rows, used by
 * the transposed convolution, models/detection_net.py:88-133).  Either table may be NULL. */
int b2m_stride_tables(const int32_t* parent, const int32_t* koff, int64_t n_fine, int64_t n_coarse,
                      int32_t* child, int64_t ld_c, int32_t* up, int64_t ld_f, void* stream);

/* Tile rulebook: per (offset k, tile t of B2M_TILE output rows) the valid pairs compacted in row order.
 *   rb_in [k*ldr + t*TILE + j]  input row of pair j      (-1 beyond the count)
 *   rb_out[k*ldr + t*TILE + j]  output row - t*TILE      (0 beyond the count)
 *   rb_cnt[k*ntiles + t]       number of pairs
 *   pair_total[K]              pairs per offset (may be NULL)
 * ldr = ntiles*TILE, ntiles = ceil(n_out/TILE).
 * rb_cnt has b2m_rulebook_cnt_size(K, n_out) = K*ntiles + 16 + 2*ntiles entries: behind the counts comes the tail
 * b2m_rulebook_balance writes (both builders call it themselves):
 *   rb_cnt[K*ntiles + x], x = 0..8   first tile of the run of tiles XCD x works on (x = 8: ntiles).  The runs carry equal
 *                                    WORK (3 per row group of 16 pairs + 1 per active offset, summed over a tile), not
 *                                    equal tile counts, and none is longer than ceil(1.25 * ntiles / 8) tiles;
 *   rb_cnt[K*ntiles + 16 + t]        that cost of tile t;
 *   rb_cnt[K*ntiles + 16 + ntiles + j]  the tile worked on at position j: inside every run the last B2M_XCD_WINDOW
 *                                    (768) positions are ordered by cost class, heaviest first (B2M_XCD_CLASSES = 8
 *                                    classes of equal width below the largest cost of the window, row order inside
 *                                    a class), the positions before keep the row order.  b2m_conv_fwd walks the tiles in
 *                                    this order (B2M_XCD_ORDER=0: row order): a launch then ends on light tiles.
 * The tail is written for rulebooks with >= 64 tiles, and b2m_conv_fwd / b2m_conv_wgrad read it from that size on
 * (B2M_XCD_BALANCE=0: equal tile counts, the tail is not read).  A caller that fills rb_in/rb_out/rb_cnt itself calls
 * b2m_rulebook_balance once (two launches: the costs, then boundaries + order by eight workgroups). */
int64_t b2m_rulebook_cnt_size(int32_t K, int64_t n_out);
int b2m_rulebook_balance(int32_t* rb_cnt, int32_t K, int64_t n_out, void* stream);
int b2m_rulebook(const int32_t* nbr, int64_t ld, int32_t K, int64_t n_out,
                 int32_t* rb_in, uint8_t* rb_out, int32_t* rb_cnt, int32_t* pair_total, void* stream);

/* Detection losses of the ScanNet configuration, values AND gradients in one pass over the prediction rows
 * (/root/reference/models/model.py:62-88, 133-176, 194-210; replaces ~150 elementwise torch launches per step):
 *   offset_loss = mean_fg sum_j |off - gt_off|, bounds_loss likewise, bb_score_loss = BCEWithLogits(score, IoU(gt box, predicted
 *   box with bounds clamped at min_bb_size)) over the foreground rows, semantics_loss = CrossEntropy(sem, gt_sem; ignore < 0 or
 *   >= n_class) over the n_valid rows, total = sum of weight * loss.
 * off / bnd / sc: (S,3) / (S,3) / (S,1) head outputs with row pitches ld_*; sem (S, n_class) or NULL; sc NULL = no score term.
 * fg: uint8 (S) foreground flags or NULL (all rows), n_fg = their count; n_valid: device scalar = rows with a valid class.
 * d_*: gradient of `total` w.r.t. the head outputs, dense (S,3) / (S,3) / (S) / (S, n_class); argmax (S) = predicted class.
 * sums: double[16] scratch; result: float[8] = total, offset_loss, bounds_loss, bb_score_loss, bb_target_scores (mean IoU),
 * bb_scores_correlation (Pearson of IoU and score logit), semantics_loss, semantics_acc. */
int b2m_detection_loss(const float* off, int64_t ld_off, const float* bnd, int64_t ld_bnd, const float* sc, int64_t ld_sc,
                       const float* sem, int64_t ld_sem, int32_t n_class, const float* gt_off, const float* gt_bnd,
                       const float* loc, const uint8_t* fg, const int64_t* gt_sem, int64_t S, double n_fg,
                       const double* n_valid, float w_off, float w_bnd, float w_sc, float w_sem, float min_bb_size,
                       float* d_off, float* d_bnd, float* d_sc, float* d_sem, int64_t* argmax, double* sums,
                       float* result, void* stream);

/* ---------------------------------------------------------------- sparse convolution (fp32, MFMA) */

/* Packed weight image read by b2m_conv_fwd.  The logical operand is B[k][ci][co]:
 *   transpose == 0 : B = w                                   (ci < cin, co < cout)      forward weights
 *   transpose == 1 : B[k][ci][co] = w[src(k)][slice_begin + co][ci], src(k) = mirror ? K-1-k : k
 *                    (ci < cout, co < slice_count)            weights of the data gradient w.r.t. the
 *                    input channels [slice_begin, slice_begin + slice_count) (mirror: stride-1 odd kernels)
 * w is [K][cin][ldw].  Layout: blocks of 64 lanes x TW*KS floats ordered [k][strip of 16*TW co][chunk of KC ci],
 * KC = 16 if the operand has >= 16 input channels else 8, KS = KC/4, TW = 3 if the operand's output channel count
 * is a multiple of 48 and K > 1, else 2; lane (q = lane/16, i = lane%16) holds B[chunk*KC + KS*q + s][strip*16*TW + 16*t + i]
 * at float TW*s + t; zero padded.  Size: b2m_weight_pack_size (the image is opaque to callers). */
int64_t b2m_weight_pack_size(int32_t K, int32_t cin, int32_t cout);
int b2m_weight_pack(const float* w, int64_t ldw, int32_t K, int32_t cin, int32_t cout, int32_t transpose,
                    int32_t mirror, int32_t slice_begin, int32_t slice_count, float* wp, void* stream);

/* The packed images of many layers in one launch.  b2m_weight_pack_plan fills plan_host (n descriptors of
 * b2m_weight_pack_plan_size() bytes; host memory) from per-layer arguments with the meaning of b2m_weight_pack
 * (w / wp: device addresses as int64) and returns the number of workgroups of the launch; the caller copies the plan to the
 * device once and calls b2m_weight_pack_run whenever the weights changed. */
int32_t b2m_weight_pack_plan_size(void);
int64_t b2m_weight_pack_plan(int32_t n, const int64_t* w, const int64_t* wp, const int64_t* ldw, const int32_t* K,
                             const int32_t* cin, const int32_t* cout, const int32_t* transpose,
                             const int32_t* mirror, const int32_t* slice_begin, const int32_t* slice_count,
                             void* plan_host);
int b2m_weight_pack_run(const void* plan_dev, int32_t n, int64_t total_blocks, void* stream);

/* Y[o, 0:cout] (+)= sum_k [X1|X2][in_k(o), :] @ B[k]  (+ bias)
 * Replaces [ME] ConvolutionForward / ConvolutionTransposeForward (resnet.py:61-65,
 * detection_net.py:37-135) and, with an identity rulebook (rb_in == NULL, K == 1), the 1x1
 * `mm` fast path (resnet.py:151-158, detection_net.py:172-193).  Two sources implement
 * ME.cat (detection_net.py:286-336) without materialising the concatenation: input channel
 * c < c1 comes from x1, the rest from x2 (c2 may be 0, x2 NULL; c1 % 16 == 0 when c2 > 0); both have n_in rows.
 *   wp    packed B for (K, c1+c2, cout), see b2m_weight_pack
 *   bias  [cout] or NULL
 *   accumulate != 0: add to the existing Y instead of overwriting
 * The same entry computes the data gradient when given the transposed/mirrored image.
 * Maps with fewer than 4096 (tile, 32-channel strip) items split the kernel offsets over several waves
 * that combine with fp32 atomics (sum order then varies in the last bits). */
int b2m_conv_fwd(const float* x1, int64_t ldx1, int32_t c1, const float* x2, int64_t ldx2, int32_t c2,
                 int64_t n_in, const float* wp, int32_t K, const float* bias,
                 const int32_t* rb_in, const uint8_t* rb_out, const int32_t* rb_cnt,
                 int64_t n_out, float* y, int64_t ldy, int32_t cout, int32_t accumulate, void* stream);

/* b2m_conv_fwd that also leaves the per-tile column sums of the finished output: tile_stats[t][0][c] = sum over the
 * rows of tile t (64 output rows) of Y[row, c], tile_stats[t][1][c] the sum of squares ([ceil(n_out/64)][2][cout]
 * doubles, accumulated in fp64) -- the statistics pass of the MinkowskiBatchNorm that follows every trunk convolution (resnet.py:61-66,
 * detection_net.py:37-135) without reading Y again; consumed by b2m_bn_tilestats(_finalize).  *wrote_stats (host) = 1
 * if the sums were written (real rulebook, whole 16-channel chunks, un-split map or exactly 4 in-LDS-combined
 * slices), 0 if this shape takes a kernel that cannot (the caller then runs b2m_bn_stats as before). */
int b2m_conv_fwd_stats(const float* x1, int64_t ldx1, int32_t c1, const float* x2, int64_t ldx2, int32_t c2,
                       int64_t n_in, const float* wp, int32_t K, const float* bias,
                       const int32_t* rb_in, const uint8_t* rb_out, const int32_t* rb_cnt,
                       int64_t n_out, float* y, int64_t ldy, int32_t cout, int32_t accumulate,
                       double* tile_stats, int32_t* wrote_stats, void* stream);

/* Inference form of a trunk layer: convolution + the eval-mode BatchNorm that follows it (+ residual) (+ ReLU) in one launch
 * (/root/reference/models/resnet.py:70-83, detection_net.py:234-337 under model.eval(), evaluation.py:70-98):
 *   Y[o, c] = [relu]( fmaf(sum_k [x1|x2][in_k(o)] B[k] [., c], scale[c], shift[c]) [+ res[o, c]] )
 * with scale / shift from b2m_bn_finalize on the running statistics -- the arithmetic of b2m_conv_fwd followed by
 * b2m_bn_apply, bit for bit, without the second launch and without the round trip of Y through HBM (the strip is
 * transformed as it is written out).  *fused (host) = 1 if the epilogue was applied; 0 if this shape's kernel cannot
 * (more than 4 split-K slices, odd column counts, the general fallback kernel): Y then holds the plain convolution
 * and the caller runs b2m_bn_apply.  No bias, no accumulate; cout % 4 == 0 and 16-byte aligned rows for fused = 1. */
int b2m_conv_fwd_affine(const float* x1, int64_t ldx1, int32_t c1, const float* x2, int64_t ldx2, int32_t c2,
                        int64_t n_in, const float* wp, int32_t K,
                        const int32_t* rb_in, const uint8_t* rb_out, const int32_t* rb_cnt,
                        int64_t n_out, float* y, int64_t ldy, int32_t cout,
                        const float* scale, const float* shift, const float* res, int64_t ld_res, int32_t relu,
                        int32_t* fused, void* stream);

/* Half-precision inference form of a trunk layer (BASELINE.json configs[4] "fp16 features on CDNA4"; the reference itself is
 * fp32-only, so this is a build extension with no reference line to replace -- the layer it computes is the one of
 * b2m_conv_fwd_affine):  Y(half) = [relu]( fmaf(conv([x1|x2](half), B(half)), scale, shift) [+ res(half)] ),  fp32 accumulation
 * (v_mfma_f32_16x16x32_f16 / _16x16x16_f16), fp32 epilogue, one rounding to half on the way out.  x1 / x2 / res / y are IEEE
 * binary16 with row pitches in ELEMENTS (inputs: multiples of 8, 16-byte aligned; y / res: multiples of 4); wp is the image
 * b2m_weight_pack_h makes of the layer's (K, c1 + c2, cout) fp32 weights (b2m_weight_pack_h_size halfs).  Real rulebooks
 * only -- a 1x1 layer passes the identity rulebook of its map (b2m_rulebook of the table 0 .. n-1, K = 1).  c1, c2, cout
 * multiples of 16 (c1 + c2 a multiple of 32, or an even number of 16-channel chunks); rows < 2^24, inputs < 4 GiB.
 * scale / shift NULL: plain convolution.  Small maps are split over the 4 waves of a workgroup, never further. */
int64_t b2m_weight_pack_h_size(int32_t K, int32_t c1, int32_t c2, int32_t cout);
int b2m_weight_pack_h(const float* w, int64_t ldw, int32_t K, int32_t c1, int32_t c2, int32_t cout, void* wp, void* stream);
int b2m_conv_fwd_h(const void* x1, int64_t ldx1, int32_t c1, const void* x2, int64_t ldx2, int32_t c2, int64_t n_in,
                   const void* wp, int32_t K, const int32_t* rb_in, const uint8_t* rb_out, const int32_t* rb_cnt,
                   int64_t n_out, void* y, int64_t ldy, int32_t cout, const float* scale, const float* shift,
                   const void* res, int64_t ld_res, int32_t relu, void* stream);
/* ... the plain half convolution (no epilogue) that also leaves the per-tile column sums of its output as stored -- every value
 * rounded to binary16 first --: tile_stats[ceil(n_out / 64)][2][cout] doubles (sum, sum of squares per tile of 64 rows), the input
 * of b2m_bn_tilestats / b2m_bn_tilestats_finalize.  Half-precision training: the training-mode BatchNorm behind the layer
 * (resnet.py:61-66) then needs no pass over the layer's output for its statistics. */
int b2m_conv_fwd_h_stats(const void* x1, int64_t ldx1, int32_t c1, const void* x2, int64_t ldx2, int32_t c2, int64_t n_in,
                         const void* wp, int32_t K, const int32_t* rb_in, const uint8_t* rb_out, const int32_t* rb_cnt,
                         int64_t n_out, void* y, int64_t ldy, int32_t cout, double* tile_stats, void* stream);

/* Transposed k2s2 convolution / data gradient of the strided one, in SCATTER form: Y[fine] (+)= W[koff(fine)] . X[parent(fine)]
 * walked over the map's DOWN rulebook (b2m_stride_tables -> b2m_rulebook: tiled over the n_coarse INPUT rows, rb_in = fine row,
 * rb_out = coarse row inside its tile), up to 64 pairs per (tile, offset) instead of the ~8 the fine-row tiling gives.
 * Replaces [ME] MinkowskiConvolutionTranspose forward (detection_net.py:96-129) and the data gradient of the strided
 * MinkowskiConvolution (detection_net.py:52-91) where b2m_conv_fwd over the UP rulebook served before.  Every fine row has
 * exactly one pair: plain 16-byte stores, or (accumulate != 0) a 16-byte read-modify-write of Y (one writer per row).  scale / shift (+ res)
 * (+ relu): the inference epilogue of b2m_conv_fwd_affine (accumulate == 0, bias == NULL).  wp: the packed image b2m_conv_fwd
 * takes.  *ran = 1 if the kernel ran, 0 if the shape is not one it takes (the caller then uses b2m_conv_fwd): whole 16-channel
 * input chunks in even number, cout % 4 == 0, 16-byte aligned operands, < 2^24 rows, at least B2M_CONV_UP_MIN_ITEMS (450)
 * (tile, strip) items. */
int b2m_conv_up(const float* x1, int64_t ldx1, int32_t c1, const float* x2, int64_t ldx2, int32_t c2, int64_t n_coarse,
                const float* wp, int32_t K, const float* bias, const int32_t* rb_in, const uint8_t* rb_out,
                const int32_t* rb_cnt, float* y, int64_t ldy, int32_t cout, int64_t n_fine, int32_t accumulate,
                const float* scale, const float* shift, const float* res, int64_t ld_res, int32_t relu,
                int32_t* ran, void* stream);

/* dW[k][ci][co] += sum over pairs (i,o) of offset k:  X[i, ci] * dY[o, co]     (fp32 atomics)
 * Replaces [ME] ConvolutionBackward (weight part).  x: n_in rows indexed by rb_in (ldx, cin columns used),
 * dy: rows indexed by tile*TILE+rb_out.  dw element (k,ci,co) lives at dw[k*dw_kstride + ci*lddw + co]
 * (so a channel sub-block of a wider weight tensor can be targeted); the caller zeroes it. */
/* workspace: NULL = the tile chunks add their dW blocks with fp32 atomics (fast, order-dependent last bits);
 * non-NULL (b2m_conv_wgrad_workspace(K,cin,cout) floats) = deterministic two-stage combine: every chunk stores its
 * partial blocks, a second kernel adds them in chunk order. */
int64_t b2m_conv_wgrad_workspace(int32_t K, int32_t cin, int32_t cout);
int b2m_conv_wgrad(const float* x, int64_t ldx, int32_t cin, int64_t n_in, const float* dy, int64_t lddy, int32_t cout,
                   const int32_t* rb_in, const uint8_t* rb_out, const int32_t* rb_cnt,
                   int64_t n_out, int32_t K, float* dw, int64_t lddw, int64_t dw_kstride, float* workspace, void* stream);
/* The same reduction with the roles of the rulebook's two row numbers EXCHANGED:
 *   dW[k][ci][co] += sum over pairs of offset k:  X[tile*TILE + rb_out, ci] * dY[rb_in, co]
 * x has n_out rows (the rows the rulebook is tiled over), dy n_in rows.  Weight gradient of a TRANSPOSED k2s2 map over the map's
 * DOWN rulebook (x = the layer's coarse input rows, dy = the gradient of its fine output rows): up to 64 pairs per (tile, offset)
 * where the UP rulebook has ~8 in half-empty 16-pair slots.  Replaces [ME] ConvolutionTransposeBackward (weight part),
 * /root/reference/models/detection_net.py:96-133 (the decoder's MinkowskiConvolutionTranspose layers). */
int b2m_conv_wgrad_tr(const float* x, int64_t ldx, int32_t cin, int64_t n_in, const float* dy, int64_t lddy, int32_t cout,
                   const int32_t* rb_in, const uint8_t* rb_out, const int32_t* rb_cnt,
                   int64_t n_out, int32_t K, float* dw, int64_t lddw, int64_t dw_kstride, float* workspace, void* stream);

/* ---------------------------------------------------------------- half-precision TRAINING (round 6; BASELINE configs[4])
 * Activations and their gradients as IEEE binary16 in HBM between the layers of the trunk; fp32 master weights, fp32 / fp64
 * statistics and accumulation.  Forward and data gradient of a layer are b2m_conv_fwd_h (above: scale / shift NULL, the data
 * gradient with the image b2m_weight_pack_h makes of the transposed, mirrored weights); these are the remaining pieces.
 * Pointers to half data are `void*`, pitches in ELEMENTS (multiples of 4, rows 8-byte aligned).
 *   b2m_conv_wgrad_h      dW[k][ci][co] += out_scale * sum_pairs X[i, ci] * dY[o, co] with X, dY half, dW fp32 (atomics; out_scale =
 *                         1 / loss scale).  tr != 0: the rulebook's row roles exchanged (b2m_conv_wgrad_tr).  Replaces [ME]
 *                         ConvolutionBackward (weight part).  With complete 16-channel blocks, 16-byte aligned rows (pitches in
 *                         multiples of 8 elements) and a real rulebook the products run on the f16 MFMA (operands transposed by
 *                         ds_read_b64_tr_b16; the product of two halves is exact in fp32, the sums differ by their order);
 *                         otherwise the operands are converted on load and multiplied on the fp32 MFMA.
 *   b2m_bn_stats_h        column sums / sums of squares of a half tensor (fp64; partial: 2*c*4096 doubles; stats: 2*c).
 *   b2m_bn_apply_h        y = [relu](fmaf(x, scale, shift) [+ res]), half in / out.
 *   b2m_bn_bwd_reduce_h   sums[0:c] = sum g, sums[c:2c] = sum g * xhat with g = dy * (y > 0) (relu); dbeta / dgamma = the sums as
 *                         fp32 times param_grad_scale (1 / loss scale).
 *   b2m_bn_bwd_apply_h    dx = gamma * invstd * (g - mean(g) - xhat * mean(g * xhat)), dres = g; half out.
 * The BatchNorm arithmetic is that of b2m_bn_stats / _apply / _bwd_reduce / _bwd_apply (resnet.py:63,66,73-82); the ReLU mask
 * is always the sign of the stored half output y. */
int b2m_conv_wgrad_h(const void* x, int64_t ldx, int32_t cin, int64_t n_in, const void* dy, int64_t lddy, int32_t cout,
                     const int32_t* rb_in, const uint8_t* rb_out, const int32_t* rb_cnt, int64_t n_out, int32_t K,
                     float* dw, int64_t lddw, int64_t dw_kstride, int32_t tr, float out_scale, void* stream);
/* half image of W'[k] = W[mirror ? K-1-k : k][s0 : s0+sc, :]^T, W (K, cin, cout) contiguous fp32: the operand with which
 * b2m_conv_fwd_h computes the gradient w.r.t. input channels [s0, s0+sc) (b2m_weight_pack_h_size(K, cout, 0, sc) halfs). */
int b2m_weight_pack_h_t(const float* w, int32_t K, int32_t cin, int32_t cout, int32_t mirror, int32_t s0, int32_t sc, void* wp,
                        void* stream);
/* Every half image of a training step in one launch (half_train.py: ~90 images per step).  b2m_weight_pack_h_plan fills plan_host
 * (n descriptors of b2m_weight_pack_h_plan_size() bytes, host memory): image i is b2m_weight_pack_h(w, cout, K, c1, cin - c1, cout)
 * when transposed[i] == 0 and b2m_weight_pack_h_t(w, K, cin, cout, mirror, s0, sc) otherwise, of the contiguous (K, cin, cout)
 * weights at address w[i] into wp[i]; returns the number of workgroups b2m_weight_pack_h_run launches (< 0: error).  The caller
 * copies the table to the device once and runs it whenever the weights changed. */
int32_t b2m_weight_pack_h_plan_size(void);
int64_t b2m_weight_pack_h_plan(int32_t n, const int64_t* w, const int64_t* wp, const int32_t* K, const int32_t* cin, const int32_t* cout,
                               const int32_t* c1, const int32_t* transposed, const int32_t* mirror, const int32_t* s0, const int32_t* sc,
                               void* plan_host);
int b2m_weight_pack_h_run(const void* plan_dev, int32_t n, int64_t total_blocks, void* stream);
/* b2m_bn_stats_h + the finalize of b2m_bn_stats_finalize in two launches (local statistics, count = n). */
int b2m_bn_stats_finalize_h(const void* x, int64_t ldx, int64_t n, int32_t c, double* partial, const float* gamma, const float* beta,
                            float eps, float momentum, float* running_mean, float* running_var, float* mean, float* invstd,
                            float* scale, float* shift, void* stream);
int b2m_bn_stats_h(const void* x, int64_t ldx, int64_t n, int32_t c, double* partial, double* stats, void* stream);
int b2m_bn_apply_h(const void* x, int64_t ldx, int64_t n, int32_t c, const float* scale, const float* shift,
                   const void* residual, int64_t ldr, int32_t relu, void* y, int64_t ldy, void* stream);
/* (relu != 0 with y == NULL: the mask is the sign of fmaf(x, mask_scale, mask_shift) -- the forward's scale / shift of a layer WITHOUT
 * a fused residual -- recomputed from the x that is read anyway; the forward then need not keep y.  As b2m_bn_bwd_reduce / _apply.) */
int b2m_bn_bwd_reduce_h(const void* dy, int64_t lddy, const void* y, int64_t ldy, const void* x, int64_t ldx, int64_t n,
                        int32_t c, const float* mean, const float* invstd, int32_t relu, const float* mask_scale,
                        const float* mask_shift, double* partial, double* sums,
                        float* dbeta_f32, float* dgamma_f32, float param_grad_scale, void* stream);
int b2m_bn_bwd_apply_h(const void* dy, int64_t lddy, const void* y, int64_t ldy, const void* x, int64_t ldx, int64_t n,
                       int32_t c, const float* mean, const float* invstd, const float* gamma, const double* sums,
                       double count, const double* count_dev, int32_t relu, const float* mask_scale, const float* mask_shift,
                       void* dx, int64_t lddx, void* dres, int64_t lddres, void* stream);

/* ---------------------------------------------------------------- SyncBN statistics exchange inside one node (opt-in)
 * Device-side all-reduce (SUM) of n <= b2m_xchg_max_doubles() doubles between <= b2m_xchg_max_ranks() ranks whose MAILBOXES
 * (b2m_xchg_size() bytes of device memory each) are mapped into one another through HIP IPC: one launch of one workgroup
 * writes this rank's values into slot [rank] of every mailbox, raises its flag there, waits (bounded: B2M_XCHG_TIMEOUT_S seconds of wall time,
 * default 120; then *err = 1 and the result is NaN) for every flag of its own mailbox and adds the slots in rank order.  Replaces, for the 2c + 1 ... 3c doubles of a SyncBN layer
 * (/root/reference/models/model.py:25), the collective library's all-reduce.  Set-up per rank: b2m_xchg_alloc -> own mailbox + a
 * 64-byte IPC handle; the handles travel over the process group; b2m_xchg_open maps a peer's mailbox; `peers_dev` is a device
 * array of `world` mailbox addresses in rank order (own mailbox at [rank]).  `epoch` counts the exchanges of the group from 1
 * (every rank passes the same value); out may alias vals.  Exercised with two processes on one GPU; across GPUs the mailboxes
 * need peer-visible memory (fine-grained, which b2m_xchg_alloc asks for first). */
int64_t b2m_xchg_size(void);
int32_t b2m_xchg_max_doubles(void);
int32_t b2m_xchg_max_ranks(void);
int b2m_xchg_alloc(void** buf, void* handle64);
int b2m_xchg_open(const void* handle64, void** ptr);
int b2m_xchg_close(void* ptr);
int b2m_xchg_free(void* buf);
/* 1 if `buf` (a mailbox of this process from b2m_xchg_alloc) is a fine-grained allocation -- what ranks on DIFFERENT devices need:
 * a running kernel of a peer device must see the writes without a kernel boundary.  The plain allocation the call falls back to
 * is only coherent between processes that share one device. */
int32_t b2m_xchg_is_finegrained(const void* buf);
int b2m_xchg_allreduce(const double* vals, int32_t n, const void* const* peers_dev, int32_t rank, int32_t world,
                       uint64_t epoch, double* out, int32_t* err, void* stream);

/* Measurement aid (bench.py `roofline.clock_mhz`; no counterpart in the reference): the shader clock the device holds
 * under fp32-MFMA load.  768 workgroups x 4 waves run `iters` blocks of 12 v_mfma_f32_16x16x4_f32; every wave stamps
 * s_memtime (shader cycles) and s_memrealtime (100 MHz ticks) around its loop.  out4 (device, zeroed by the call):
 * [0] sum of cycles, [1] sum of ticks, [2] waves; clock = out4[0] / out4[1] * 100 MHz. */
int b2m_clock_probe(unsigned long long* out4, int32_t iters, void* stream);

/* ---------------------------------------------------------------- batch norm / elementwise (fp32, HBM-bound) */

/* Column sums for BatchNorm: stats[0:c] = sum x, stats[c:2c] = sum x^2 (double), deterministic
 * two-stage reduction.  partial: double[2*c*nblk_max] scratch with nblk_max = 4096.
 * Replaces the reduction inside torch.nn.BatchNorm1d wrapped by ME.MinkowskiBatchNorm
 * (resnet.py:63,66; detection_net.py:40-135). */
int b2m_bn_stats(const float* x, int64_t ldx, int64_t n, int32_t c, double* partial, double* stats, void* stream);

/* b2m_bn_stats followed by b2m_bn_finalize for the single-process case (count = n): the final reduction and the
 * finalize math share one launch.  stats may be NULL. */
int b2m_bn_stats_finalize(const float* x, int64_t ldx, int64_t n, int32_t c, double* partial, double* stats,
                          const float* gamma, const float* beta, float eps, float momentum,
                          float* running_mean, float* running_var, float* mean, float* invstd,
                          float* scale, float* shift, void* stream);

/* BatchNorm statistics from the per-tile column sums of b2m_conv_fwd_stats instead of a pass over x:
 * b2m_bn_tilestats = b2m_bn_stats, b2m_bn_tilestats_finalize = b2m_bn_stats_finalize with (tile_stats, ntiles) in
 * place of (x, ldx); n = number of rows the sums cover.  partial: double[2*c*1280] scratch.  Maps of up to 8192 tiles
 * (B2M_BN_TS_ONE) take ONE launch for the reduction and the finalize math (fixed summation order, other grouping). */
int b2m_bn_tilestats(const double* tile_stats, int64_t ntiles, int32_t c, double* partial, double* stats, void* stream);
int b2m_bn_tilestats_finalize(const double* tile_stats, int64_t ntiles, int64_t n, int32_t c, double* partial,
                              double* stats, const float* gamma, const float* beta, float eps, float momentum,
                              float* running_mean, float* running_var, float* mean, float* invstd, float* scale,
                              float* shift, void* stream);

/* From (possibly all-reduced) sums: scale/shift for the apply kernel, saved mean/invstd, and the
 * running-statistics update (momentum, unbiased variance), all on device.
 * count = number of rows the sums cover.  Under SyncBN the global count travels with the sums through the
 * all-reduce and never visits the host: pass it as count_dev (device pointer to one double; overrides count). */
int b2m_bn_finalize(const double* stats, double count, const double* count_dev, int32_t c, const float* gamma, const float* beta,
                    float eps, float momentum, float* running_mean, float* running_var,
                    float* mean, float* invstd, float* scale, float* shift, void* stream);

/* y = x*scale + shift (+ residual) (ReLU if relu) */
int b2m_bn_apply(const float* x, int64_t ldx, int64_t n, int32_t c, const float* scale, const float* shift,
                 const float* residual, int64_t ldr, int32_t relu, float* y, int64_t ldy, void* stream);

/* Backward reduction: g = dy * (relu ? y>0 : 1);  sums[0:c] = sum g (= dbeta), sums[c:2c] = sum g*xhat (= dgamma);
 * dbeta_f32 / dgamma_f32 (c floats each, may be NULL) receive the two halves rounded to fp32, in two separate
 * buffers so that the caller can hand them to autograd as parameter gradients without a copy.
 * With relu and no fused residual, y may be NULL when mask_scale / mask_shift (the scale / shift of the forward's
 * b2m_bn_apply) are given: the mask is then fmaf(x, scale, shift) > 0, bit for bit the forward's decision, and the
 * kernel reads one tensor less. */
int b2m_bn_bwd_reduce(const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* x, int64_t ldx,
                      int64_t n, int32_t c, const float* mean, const float* invstd, int32_t relu,
                      const float* mask_scale, const float* mask_shift, double* partial, double* sums,
                      float* dbeta_f32, float* dgamma_f32, void* stream);

/* dx = gamma*invstd*(g - sum_g/count - xhat*sum_gxhat/count); optionally dres = g.  y / mask_* as in
 * b2m_bn_bwd_reduce. */
int b2m_bn_bwd_apply(const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* x, int64_t ldx,
                     int64_t n, int32_t c, const float* mean, const float* invstd, const float* gamma,
                     const double* sums, double count, const double* count_dev, int32_t relu, const float* mask_scale,
                     const float* mask_shift, float* dx, int64_t lddx, float* dres, int64_t lddres, void* stream);

/* Two BatchNorms that meet in one add -- the end of a BasicBlock with a shortcut convolution:
 *   y = relu?(BN_a(x_a) + BN_b(x_b))      (/root/reference/models/resnet.py:73-82: norm2 + downsample.1 + add + ReLU)
 * b2m_bn_apply2: both affine maps, the add and the ReLU in one pass (the shortcut's normalised tensor is never stored);
 * b2m_bn_bwd_reduce2: sums[0:c] = sum g, sums[c:2c] = sum g*xhat_a, sums[2c:3c] = sum g*xhat_b with g = dy * (y > 0 if relu)
 *   (partial: double[3*c*1280] scratch) -- under SyncBN ONE all-reduce of 3c doubles for the pair;
 * b2m_bn_bwd_apply2: dx_a, dx_b (from `sums`: under SyncBN the sums over all ranks) and the four parameter gradients
 * (fp32 copies of `local_sums`, this rank's own sums; NULL: of `sums`; any gradient may be NULL). */
int b2m_bn_apply2(const float* xa, int64_t lda, const float* xb, int64_t ldb, int64_t n, int32_t c,
                  const float* scale_a, const float* shift_a, const float* scale_b, const float* shift_b,
                  int32_t relu, float* y, int64_t ldy, void* stream);
int b2m_bn_bwd_reduce2(const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* xa, int64_t lda,
                       const float* xb, int64_t ldb, int64_t n, int32_t c, const float* mean_a, const float* invstd_a,
                       const float* mean_b, const float* invstd_b, int32_t relu, double* partial, double* sums,
                       void* stream);
int b2m_bn_bwd_apply2(const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* xa, int64_t lda,
                      const float* xb, int64_t ldb, int64_t n, int32_t c, const float* mean_a, const float* invstd_a,
                      const float* gamma_a, const float* mean_b, const float* invstd_b, const float* gamma_b,
                      const double* sums, double count, const double* count_dev, int32_t relu, float* dxa, int64_t lddxa,
                      float* dxb, int64_t lddxb, float* dbeta_a, float* dgamma_a, float* dbeta_b, float* dgamma_b,
                      const double* local_sums, void* stream);

/* Training-mode BatchNorm of a SMALL map (n <= B2M_BN_SMALL_MAX_ROWS rows: the deep U-Net levels, the heads' segment
 * rows) in ONE launch each way: b2m_bn_small_fwd = statistics (fp64) + finalize (mean / invstd / scale / shift, running
 * statistics) + apply (+residual)(+ReLU); b2m_bn_small_bwd = b2m_bn_bwd_reduce + b2m_bn_bwd_apply (dbeta / dgamma may be
 * NULL).  Same arguments and meaning as the multi-launch entries; not for SyncBN (the statistics never leave the kernel). */
#define B2M_BN_SMALL_MAX_ROWS 16384
int b2m_bn_small_fwd(const float* x, int64_t ldx, int64_t n, int32_t c, const float* gamma, const float* beta,
                     float eps, float momentum, float* running_mean, float* running_var, float* mean, float* invstd,
                     float* scale, float* shift, const float* residual, int64_t ldr, int32_t relu, float* y, int64_t ldy,
                     void* stream);
int b2m_bn_small_bwd(const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* x, int64_t ldx,
                     int64_t n, int32_t c, const float* mean, const float* invstd, const float* gamma, int32_t relu,
                     const float* mask_scale, const float* mask_shift, float* dbeta_f32, float* dgamma_f32,
                     float* dx, int64_t lddx, float* dres, int64_t lddres, void* stream);
/* SyncBN forms of the two (the statistics of all ranks meet between two launches instead of inside one):
 *   b2m_bn_small_fwd_stats  -> xchg[0:c] = sum x, xchg[c:2c] = sum x^2, xchg[2c] = n of THIS rank (fp64);
 *   the caller all-reduces xchg[2c + 1];
 *   b2m_bn_small_fwd_apply  = finalize from xchg (global count xchg[2c]) + apply (+residual)(+ReLU).
 *   b2m_bn_small_bwd_phase(1, ...) -> xchg[0:c] = sum g, xchg[c:2c] = sum g * xhat of this rank, dbeta / dgamma from them
 *   (the gradient all-reduce averages parameter gradients over the ranks); all-reduce of xchg[2c];
 *   b2m_bn_small_bwd_phase(2, ...) = dx (and dres) from xchg with the global row count *count_dev.
 * Replaces torch.nn.SyncBatchNorm's exchanges for the small maps (/root/reference/models/model.py:25). */
int b2m_bn_small_fwd_stats(const float* x, int64_t ldx, int64_t n, int32_t c, double* xchg, void* stream);
int b2m_bn_small_fwd_apply(const double* xchg, const float* x, int64_t ldx, int64_t n, int32_t c, const float* gamma,
                           const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                           float* mean, float* invstd, float* scale, float* shift, const float* residual, int64_t ldr,
                           int32_t relu, float* y, int64_t ldy, void* stream);
int b2m_bn_small_bwd_phase(int32_t phase, const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* x,
                           int64_t ldx, int64_t n, int32_t c, const float* mean, const float* invstd, const float* gamma,
                           int32_t relu, const float* mask_scale, const float* mask_shift, float* dbeta_f32,
                           float* dgamma_f32, float* dx, int64_t lddx, float* dres, int64_t lddres, double* xchg,
                           const double* count_dev, void* stream);

/* out = relu(a) (b == NULL) or a + b, optional relu; grad helper: dx = dy * (y > 0). */
int b2m_relu_fwd(const float* x, int64_t n_elem, float* y, void* stream);
int b2m_relu_bwd(const float* dy, const float* y, int64_t n_elem, float* dx, void* stream);
int b2m_add(const float* a, const float* b, int64_t n_elem, float* out, void* stream);

/* ---------------------------------------------------------------- segment pooling */

/* out[s,:] = mean (mode 0) or max (mode 1) of the x rows with ids[row]==s; output row s <-> pooling id s.
 * Replaces the "overwrite batch column, rebuild SparseTensor, global pool" sequence of
 * models/detection_net.py:345-352 with a direct segmented reduction (no second hash build).
 * mode 0 sums with fp32 atomics (order-dependent in the last bits) and divides by the exact integer
 * count; mode 1 is deterministic (64-bit packed atomicMax, lowest row wins ties).
 *   counts[n_seg] int32 out; argmax[n_seg*c] int32 out (mode 1 only); scratch uint64[n_seg*c] (mode 1 only) */
int b2m_segment_pool_fwd(const float* x, int64_t ldx, int64_t n, int32_t c, const int64_t* ids, int64_t n_seg,
                         int32_t mode, float* out, int32_t* counts, int32_t* argmax, uint64_t* scratch,
                         void* stream);

/* Deterministic segment mean: `order` = row indices grouped by segment (stable sort of the ids), seg_start[s] ..
 * seg_start[s+1] = the rows of segment s inside `order`.  One workgroup per segment, fixed summation order, no
 * atomics.  counts[s] = rows of segment s.  (Same result as mode 0 of b2m_segment_pool_fwd up to summation order.) */
int b2m_segment_mean_sorted(const float* x, int64_t ldx, int64_t n, int32_t c, const int64_t* order,
                            const int64_t* seg_start, int64_t n_seg, float* out, int32_t* counts, void* stream);
int b2m_segment_pool_bwd(const float* dout, int64_t n, int32_t c, const int64_t* ids, int64_t n_seg,
                         int32_t mode, const int32_t* counts, const int32_t* argmax,
                         float* dx, int64_t lddx, void* stream);

/* ---------------------------------------------------------------- box votes -> instance masks */

/* Greedy non-maximum clustering of n boxes [score,min3,max3] (fp32, one scene).
 * Replaces NMS_clustering(boxes, cluster_th, get_heatmaps=True), models/iou_nms.py:68-105
 * (IoU formula of torch_IOUs, iou_nms.py:26-45, evaluated in the same fp32 operation order).
 * Score ties are broken by the lower row index (the reference's unstable argsort leaves them
 * undefined).  Outputs: reps[<=n] representative row per cluster, assign[n] cluster index of every
 * box, heat[max_k*n] row-major IoU heat-maps (row r = IoU(box reps[r], all boxes), heat[r, reps[r]]=1),
 * *k_out = number of clusters (device int32).  If the cluster count exceeds max_k the extra
 * heat-map rows are not written (k_out still holds the true count).
 * order: uint64[npow2(n)] scratch (npow2 = next power of two); on return its first n entries hold
 * the box rows in visiting order (descending score) in their low 32 bits.  n <= 262144. */
int b2m_nmc(const float* boxes, int32_t n, float cluster_th, int32_t max_k,
            int32_t* reps, int32_t* assign, float* heat, int32_t* k_out, uint64_t* order, void* stream);

/* The same for all scenes of a batch in ONE launch (one workgroup per scene; the per-scene loop of
 * SelectionNet.detection2mask, models/detection_net.py:390-425).  boxes / reps / assign hold the scenes back to back;
 * desc (device, int64[n_scenes][6]) = {first box row, n, npow2(n) (>= 2), max_k, first heat element, first order
 * element} per scene; k_out[n_scenes].  max_n = the largest n (<= 262144).  Scene s gets exactly what b2m_nmc returns
 * for its boxes. */
int b2m_nmc_batch(const float* boxes, const int64_t* desc, int32_t n_scenes, int32_t max_n, float cluster_th,
                  int32_t* reps, int32_t* assign, float* heat, int32_t* k_out, uint64_t* order, void* stream);

/* Heat-map rows -> voxel bit masks.  For selected cluster rows sel[0:ksel] of heat (k x n_fg):
 * value(v) = fg_slot[seg2vox[v]] >= 0 ? heat[sel[r], fg_slot[seg2vox[v]]] : 0;
 * bit v of bits[r*words + v/64] = value(v) > mask_bin_th.  Replaces models/detection_net.py:436-446
 * (zero-padded background, projection through seg2vox, threshold).  words = ceil(n_vox/64). */
int b2m_mask_project(const float* heat, int32_t n_fg, const int32_t* sel, int32_t ksel,
                     const int32_t* fg_slot, const int64_t* seg2vox, int64_t n_vox, float mask_bin_th,
                     uint64_t* bits, int64_t words, void* stream);

/* Greedy mask NMS in the given (score-descending) order; IoU = |a&b| / |a|b| from popcounts,
 * evaluated as float32(inter)/float32(union) like torch's int64 true-division.
 * Replaces mask_NMS(sorted_masks, th), models/iou_nms.py:130-144 (masks_iou 109-121).
 * inter: int32[k*k] scratch, keep[k] int32 flags out, *n_keep device int32. */
int b2m_mask_nms(const uint64_t* bits, int32_t k, int64_t words, float th,
                 int32_t* inter, int32_t* keep, int32_t* n_keep, void* stream);

/* labels[r] = argmax_c |{v in mask rows[r] : sem[v]==c}| (lowest c on ties; 0 for an empty mask).
 * Replaces the bincount/argmax loop of models/detection_net.py:461-466.  0 <= sem < n_class <= 256.
 * rows: int32[k] mask row per output (NULL = identity). */
int b2m_label_hist(const uint64_t* bits, int64_t words, const int32_t* rows, int32_t k, const int32_t* sem,
                   int64_t n_vox, int32_t n_class, int32_t* labels, void* stream);

/* hist[r*n_class + c] = |{v < n in bit row r : label[v] == c}| for every mask row: all prediction x ground-truth
 * intersection counts of assign_instances_for_scan, utils/eval_metric.py:316-330 (label = dense ground-truth
 * instance index per point), in one pass.  n_class <= 2048; labels outside [0, n_class) are skipped. */
int b2m_mask_hist(const uint64_t* bits, int64_t words, int32_t k, const int32_t* label, int64_t n, int32_t n_class,
                  int32_t* hist, void* stream);

/* Gather bits through an index (vox2point) into a byte mask: out[r*n_pts + p] = bit index[p] of row rows[r]
 * (rows NULL = identity, index NULL = identity).
 * Replaces pred_masks[:, vox2point], models/detection_net.py:469-471. */
int b2m_mask_gather(const uint64_t* bits, int64_t words, const int32_t* rows, int32_t k,
                    const int64_t* index, int64_t n_pts, uint8_t* out, void* stream);

/* Pack k boolean (byte) mask rows of n elements into bit rows: bit v of bits[r*words + v/64].
 * Input adaptor for b2m_mask_nms when the caller holds torch bool masks (mask_NMS, iou_nms.py:130). */
int b2m_mask_pack(const uint8_t* masks, int32_t k, int64_t n, uint64_t* bits, int64_t words, void* stream);

/* Row-wise IoU of two (n,6) [min,max] box sets: set_IOUs, models/iou_nms.py:4-22. */
int b2m_set_ious(const float* a, const float* b, int64_t n, float* out, void* stream);

/* ---- the mask stages of detection2mask for ALL scenes of a batch: one launch per stage
 * (/root/reference/models/detection_net.py:390-477 walks the scenes one by one).  `desc`: device array of n_scenes records of
 * B2M_MASK_DESC int64 fields (pointers stored as integers):
 *    0 heat (float*, K x n_fg)       1 n_fg            2 sel (int32*, clusters that passed the score filter)   3 ksel
 *    4 fg_slot (int32*, per segment) 5 seg2vox (int64*) 6 n_vox            7 bits (uint64*, ksel x words, written by project)
 *    8 words = ceil(n_vox / 64)      9 inter (int32*, ksel x ksel scratch)  10 keep (int32*, ksel flags; NULL: no mask NMS)
 *   11 rows (int32*, surviving rows of bits, NULL = 0..kk-1)   12 kk       13 sem (int32*, label per voxel)
 *   14 labels (int32*, kk out)      15 index (int64*, voxel of every output point; NULL = identity)     16 n_pts
 *   17 out (uint8*, kk x n_pts)     18 first global row of this scene among the `sel` rows of the batch  19 ... among the kept rows
 * Each entry reads only the fields of its stage, so the table may be completed between the stages (sel after the score
 * filter, rows / kk after the keep flags were read).  Results are those of b2m_mask_project / b2m_mask_nms / b2m_label_hist /
 * b2m_mask_gather scene by scene. */
#define B2M_MASK_DESC 20
int b2m_mask_project_batch(const int64_t* desc, int32_t n_scenes, int64_t total_sel, int64_t max_words, float mask_bin_th,
                           void* stream);
int b2m_mask_nms_batch(const int64_t* desc, int32_t n_scenes, int32_t max_k, float mask_nms_th, void* stream);
int b2m_label_hist_batch(const int64_t* desc, int32_t n_scenes, int64_t total_kept, int32_t n_class, void* stream);
int b2m_mask_gather_batch(const int64_t* desc, int32_t n_scenes, int64_t total_kept, int64_t max_pts, void* stream);
/* b2m_mask_gather_batch through a voxel-major image of the kept rows, built here first: `tbits` = device array of n_scenes
 * pointers, scene s -> n_vox x ceil(kk / 64) uint64 words of scratch (word c of voxel v: the bits of kept rows 64c .. 64c + 63 at
 * v).  One 8-byte look-up per point and 64 rows instead of one per point and row; the same bytes in `out`.  max_words = the
 * largest `words` of the table. */
int b2m_mask_gather_batch_t(const int64_t* desc, int32_t n_scenes, int64_t total_kept, int64_t max_pts, int64_t max_words,
                            const int64_t* tbits, void* stream);

#ifdef __cplusplus
}
#endif
#endif
