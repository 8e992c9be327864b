/* C ABI of the scene-preparation path (voxelisation + collate on the device).
 *
 * Replaces, for one scene, the voxelisation block of the reference's dataset class,
 * /root/reference/models/dataloader.py:61-123 (np.round + np.unique(axis=0, return_inverse=True), the sklearn
 * ball-tree nearest-point association, np.unique over segment ids, the per-segment centroid loop); the batch is
 * then assembled as collate_fn does (dataloader.py:946-995, utils/util.py:123-130) by the host mirror
 * box2mask_amd/prepare.py.  Same conventions as include/b2m.h: device pointers, caller-owned buffers and scratch,
 * HIP stream as void*, 0 or a negative B2M_ERR_* code (b2m_last_error()).  Same shared library.
 */
#ifndef B2M_PREPARE_H
#define B2M_PREPARE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* *shift = min(0, min over all 3*n_pts coordinates)  -- dataloader.py:63.  scratch: 1 uint64. */
int b2m_vox_shift(const double* pos, int64_t n_pts, double* shift, uint64_t* scratch, void* stream);

/* keys[p] = x<<42 | y<<21 | z with (x,y,z) = rint((pos[p] - *shift) / voxel_size), evaluated in fp64 exactly as
 * numpy does (subtract, divide, round half to even) -- dataloader.py:63-67.  Numeric order of the keys is the
 * lexicographic (x,y,z) order np.unique(axis=0) sorts by.  *bad_count = points outside [0, 2^21) per axis. */
int b2m_vox_keys(const double* pos, int64_t n_pts, const double* shift, double voxel_size, uint64_t* keys,
                 int32_t* bad_count, void* stream);

/* First half of np.unique(return_inverse=True) on 64-bit keys (voxel keys, segment ids): inserts every key into
 * the open-addressing table tkeys[cap] (cap = power of two >= 2n; the value 2^64-1 is reserved), records the slot
 * of every input element and appends each distinct key once to ukeys (unordered).  Synchronises the stream and
 * returns the number of distinct keys (or a negative error code). */
int64_t b2m_unique_insert(const uint64_t* in, int64_t n, uint64_t* tkeys, int64_t cap, int32_t* slot_of,
                          uint64_t* ukeys, int32_t* n_unique, void* stream);
/* The same without the host read: no synchronisation, the count stays in *n_unique on the device (a caller that prepares several
 * scenes reads all counts at once: box2mask_amd.prepare.voxelize_scenes). */
int b2m_unique_insert_async(const uint64_t* in, int64_t n, uint64_t* tkeys, int64_t cap, int32_t* slot_of,
                          uint64_t* ukeys, int32_t* n_unique, void* stream);

/* In-place ascending bitonic sort of n_pad keys (power of two; pad with 2^64-1). */
int b2m_sort_u64(uint64_t* keys, int64_t n_pad, void* stream);

/* Second half: tvals[slot of sorted[r]] = r for r < n_unique, then inverse[i] = tvals[slot_of[i]]
 * (the `return_inverse` array: vox2point of dataloader.py:68, seg2vox of :108). */
int b2m_unique_rank(const uint64_t* sorted, int64_t n_unique, const uint64_t* tkeys, int32_t* tvals, int64_t cap,
                    const int32_t* slot_of, int64_t n, int64_t* inverse, void* stream);

/* coords[v] = [batch, x, y, z] (int32) of the sorted voxel keys: the rows ME.utils.batched_coordinates makes of
 * ret['vox_coords'] (dataloader.py:68, 966). */
int b2m_vox_decode(const uint64_t* sorted, int64_t n_vox, int32_t batch, int32_t* coords, void* stream);

/* point2vox[v] = index of the scene point nearest to voxel centre v (Euclidean, in voxel units), the result of
 * NearestNeighbors(n_neighbors=1, algorithm='ball_tree').fit(input_coords).kneighbors(vox_coords),
 * dataloader.py:75-77.  Exact: the distance is the ball tree's reduced distance (sum of squares over x, y, z in
 * fp64, no contraction); among points at exactly the same distance the lowest index wins (the reference's choice
 * there depends on the tree traversal).  tkeys/tvals: the voxel table after b2m_unique_rank.
 * best: uint64[n_vox] scratch. */
int b2m_vox_nearest(const double* pos, int64_t n_pts, const double* shift, double voxel_size, const uint64_t* tkeys,
                    const int32_t* tvals, int64_t cap, int64_t n_vox, uint64_t* best, int32_t* point2vox,
                    void* stream);

/* feats[v] = float32([colors | normals][point2vox[v]]) (normals may be NULL: 3 features),
 * vox_segments[v] = segments[point2vox[v]]  -- dataloader.py:82-91 with the .float() of collate_fn (:967). */
int b2m_vox_gather(const int32_t* point2vox, int64_t n_vox, const double* colors, const double* normals,
                   const int64_t* segments, float* feats, int64_t* vox_segments, void* stream);

/* out[s] = mean over the voxels v of segment s (seg2vox[v] == s) of (coords[v] * voxel_size + *shift): the
 * segment_middle loop of dataloader.py:110-117.  Integer sums (exact, order independent), one fp64 evaluation per
 * segment; agrees with numpy's running fp64 mean to a few ulp.  sums: uint64[3*n_seg], counts: int32[n_seg]. */
int b2m_seg_centroid(const int32_t* coords, const int64_t* seg2vox, int64_t n_vox, int64_t n_seg, double voxel_size,
                     const double* shift, uint64_t* sums, int32_t* counts, double* out, void* stream);

/* ---- box supervision: approx_association, dataloader.py:203-314, segment branch ---- */

/* Per scene point: count[p] = number of boxes [bb_min, bb_max] (closed, (n_boxes,3) fp64 each) containing it,
 * first_bb[p] = lowest such box index (or -1), smallest_bb[p] = the one of smallest bb_volume among them (first on
 * ties, or -1).  Replaces the boxes x points occupancy matrix and the per-point argwhere list of :235-240.
 * n_boxes <= 1024. */
int b2m_box_membership(const double* pos, int64_t n_pts, const double* bb_min, const double* bb_max,
                       const float* bb_volume, int32_t n_boxes, int32_t* count, int32_t* first_bb,
                       int32_t* smallest_bb, void* stream);

/* Segment vote of :274-309.  For every voxel-level segment (rank r in the segment table tkeys/tvals of
 * b2m_unique_rank) the point with the lexicographically smallest (count, index) decides:
 * count 0 -> -1 (background); 1 -> instance_ids[first_bb]; >1 -> instance_ids[smallest_bb] when
 * smallest_bb_heuristic, else -2 (unknown).  inst_per_point[p] = the verdict of p's segment, -2 for points whose
 * segment has no voxel.  best: uint64[n_seg] scratch, seg_of_point: int32[n_pts] scratch/out. */
int b2m_seg_box_vote(const int64_t* segments, int64_t n_pts, const uint64_t* tkeys, const int32_t* tvals, int64_t cap,
                     int64_t n_seg, const int32_t* count, const int32_t* first_bb, const int32_t* smallest_bb,
                     const int64_t* instance_ids, int32_t smallest_bb_heuristic, uint64_t* best,
                     int32_t* seg_of_point, int64_t* inst_per_seg, int64_t* inst_per_point, void* stream);

/* ---- the other association branches ---- */

/* Oriented boxes of ARKitScenes.approx_association (dataloader.py:545-557): count[p] = number of boxes b with
 * -half[b] <= R_b (pos[p] - centers[b]) <= half[b] on every axis (closed; R_b row-major 3x3, everything fp64),
 * first_bb[p] = the lowest such b or -1. */
int b2m_obb_membership(const double* pos, int64_t n_pts, const double* centers, const double* rotations,
                       const double* half_sizes, int32_t n_boxes, int32_t* count, int32_t* first_bb, void* stream);

/* seg_of_point[p] = rank of point p's segment among the voxel-level segments (table of b2m_unique_rank), -1 when the
 * segment has no voxel: the `seg_id == scene['segments']` masks of dataloader.py:265, 281, 866, 914 in one pass. */
int b2m_seg_rank(const int64_t* segments, int64_t n_pts, const uint64_t* tkeys, const int32_t* tvals, int64_t cap,
                 int32_t* seg_of_point, void* stream);

/* mode_cls[s] = most frequent class among the points of segment s, lowest class index on ties: scipy.stats.mode of
 * dataloader.py:267 (majority_vote) and :916-917 (S3DIS) when the classes are numbered in ascending order of their
 * value.  cls[p] in [0, n_class); hist: int32[n_seg * n_class] scratch. */
int b2m_seg_mode(const int32_t* seg_of_point, const int32_t* cls, int64_t n_pts, int64_t n_seg, int32_t n_class,
                 int32_t* hist, int32_t* mode_cls, void* stream);

#ifdef __cplusplus
}
#endif
#endif
