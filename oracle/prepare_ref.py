"""CPU restatement of the reference's scene preparation -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
(box2mask_amd/prepare.py) never does.

Follows /root/reference/models/dataloader.py:61-123 (voxelisation block of ScanNet.__getitem__) and :946-995
(collate_fn) with /root/reference/utils/util.py:123-130 (to_unique).  The nearest-point association uses
scikit-learn's ball tree exactly as the reference does (sklearn is the reference's own dependency for this step and
is part of this image); `nearest_bruteforce` is an independent fp64 check of it for small inputs.

Pinned: tests/golden/prepare.npz holds inputs and outputs of the REAL reference code (tools/gen_golden.py drives
ScanNet.__getitem__ and collate_fn of /root/reference with a synthetic scene behind stand-ins for the modules that
only load data); tests/test_oracle_prepare.py checks this restatement against it bit for bit.
"""
import numpy as np

try:                                    # the reference's dependency for dataloader.py:75
    from sklearn.neighbors import NearestNeighbors
except ImportError:                     # pragma: no cover
    NearestNeighbors = None


def voxelize_scene(positions, colors, normals, segments, voxel_size, use_normals=True):
    """dataloader.py:61-123 for do_segment_pooling=True.  All arrays numpy (float64 / int64)."""
    positions = np.asarray(positions)
    shift = min(0, np.min(positions))                                   # :63
    input_coords = (positions - shift) / voxel_size                     # :63-65
    vox = np.round(input_coords)                                        # :67
    vox_coords, vox2point = np.unique(vox, axis=0, return_inverse=True)  # :68
    vox2point = vox2point.reshape(-1)
    nbrs = NearestNeighbors(n_neighbors=1, algorithm='ball_tree').fit(input_coords)   # :75
    point2vox = nbrs.kneighbors(vox_coords, return_distance=False).reshape(-1)        # :76-77
    feats = np.concatenate([colors, normals], 1) if use_normals else np.asarray(colors)   # :82-88
    vox_segments = np.asarray(segments)[point2vox]                      # :90
    vox_features = feats[point2vox]                                     # :91
    vox_world = vox_coords * voxel_size + shift                         # :94
    uniq, seg2vox = np.unique(vox_segments, return_inverse=True)        # :108
    seg2vox = seg2vox.reshape(-1)
    seg2point = seg2vox[vox2point]                                      # :109
    middle = np.zeros((uniq.shape[0], 3))
    for i, seg in enumerate(uniq):                                      # :113-115
        middle[i] = np.mean(vox_world[seg == vox_segments], axis=0)
    return {'vox_coords': vox_coords, 'vox2point': vox2point, 'point2vox': point2vox, 'vox_segments': vox_segments,
            'vox_features': vox_features, 'vox_world_coords': vox_world, 'seg2vox': seg2vox, 'seg2point': seg2point,
            'input_location': middle, 'pred2point': seg2point, 'unique_vox_segments': uniq}


def nearest_bruteforce(input_coords, vox_coords, chunk=256):
    """argmin_p sum_j (vox[j] - p[j])^2 in fp64, summed x,y,z; lowest p on exact ties."""
    input_coords = np.asarray(input_coords, np.float64)
    out = np.empty(len(vox_coords), np.int64)
    for a in range(0, len(vox_coords), chunk):
        q = np.asarray(vox_coords[a:a + chunk], np.float64)
        d = (q[:, None, 0] - input_coords[None, :, 0]) ** 2
        d = d + (q[:, None, 1] - input_coords[None, :, 1]) ** 2
        d = d + (q[:, None, 2] - input_coords[None, :, 2]) ** 2
        out[a:a + chunk] = np.argmin(d, 1)
    return out


def to_unique(segments):
    """utils/util.py:123-130."""
    segs = [np.array(s, copy=True) for s in segments]
    for i in range(1, len(segs)):
        segs[i] += np.max(segs[i - 1]) + 1
    cat = np.concatenate(segs, 0)
    _, pooling_ids = np.unique(cat, return_inverse=True)
    return pooling_ids.reshape(-1).astype(np.int64)


def collate(items):
    """collate_fn.__call__ in 'test' mode (dataloader.py:954-984): numpy arrays with the dtypes of the torch
    tensors the reference makes."""
    coords = []
    for b, it in enumerate(items):                                      # ME.utils.batched_coordinates, :966
        c = np.floor(it['vox_coords']).astype(np.int32)
        coords.append(np.concatenate([np.full((len(c), 1), b, np.int32), c], 1))
    return {
        'vox_coords': np.concatenate(coords, 0),
        'vox_features': np.concatenate([it['vox_features'] for it in items], 0).astype(np.float32),       # :967
        'batch_ids': np.concatenate([np.full(len(it['input_location']), b, np.int64)                      # :969-974
                                     for b, it in enumerate(items)], 0),
        'input_location': np.concatenate([it['input_location'] for it in items], 0).astype(np.float32),   # :980
        'pooling_ids': to_unique([it['vox_segments'] for it in items]),                                   # :981
    }


def approx_association(positions, segments, labels, unique_segs, smallest_bb_heuristic=True):
    """dataloader.py:203-314, segment branch (point_association = majority_vote = False, no dropout / noise).
    Vectorised restatement of the reference's Python loops; pinned by tests/golden/prepare.npz.
    Returns (inst_per_point (P,), inst_per_seg (S,)) with -1 background, -2 unknown."""
    semantics = np.asarray(labels['per_instance_semantics'])
    scene_fg = (semantics > 2) & (semantics != 22)                                   # :207-208
    centers = np.asarray(labels['per_instance_bb_centers'])[scene_fg]                # :219
    bounds = np.asarray(labels['per_instance_bb_bounds'])[scene_fg] + 0.005          # :221
    min_corner, max_corner = centers - bounds, centers + bounds
    instance_ids = np.asarray(labels['unique_instances'])[scene_fg]                  # :224
    occ = (np.all(positions[None] >= min_corner[:, None], axis=-1) &
           np.all(positions[None] <= max_corner[:, None], axis=-1))                  # :235 (B, P)
    num = occ.sum(axis=0)                                                            # :239
    bb_volume = np.prod(2 * bounds, axis=1)                                          # :240
    segments = np.asarray(segments)
    inst_per_point = np.full(len(positions), -2, np.int64)                           # :279
    inst_per_seg = np.full(len(unique_segs), -2, np.int64)
    for i, seg_id in enumerate(unique_segs):                                         # :281-296
        idx = np.nonzero(seg_id == segments)[0]
        n_on = num[idx]
        m = n_on.min()
        if m == 1:
            p = idx[np.nonzero(n_on == 1)[0][0]]
            inst = instance_ids[np.nonzero(occ[:, p])[0][0]]
        elif m == 0:
            inst = -1
        elif smallest_bb_heuristic:                                                  # :298-309
            p = idx[n_on.argmin()]
            box_ids = np.nonzero(occ[:, p])[0]
            inst = instance_ids[box_ids[np.argmin(bb_volume[box_ids])]]
        else:
            continue
        inst_per_seg[i] = inst
        inst_per_point[idx] = inst
    return inst_per_point, inst_per_seg


def bbs_supervision(item, labels, inst_per_seg):
    """dataloader.py:165-200 for do_segment_pooling=True: per-segment training targets."""
    instances = inst_per_seg
    seg_inst = np.asarray(labels['seg2inst'])[item['unique_vox_segments']]           # :176
    gt_full_sem = np.asarray(labels['per_instance_semantics'])[seg_inst]             # :177
    gt_unlabeled = gt_full_sem == 0
    fg = instances > -1                                                              # :180
    gt_bb_bounds = np.zeros((len(fg), 3))
    gt_bb_bounds[fg] = np.asarray(labels['per_instance_bb_bounds'])[instances[fg]]   # :184-185
    gt_bb_centers = np.zeros((len(fg), 3))
    gt_bb_centers[fg] = np.asarray(labels['per_instance_bb_centers'])[instances[fg]]
    gt_bb_offsets = gt_bb_centers - (item['input_location'] * fg[:, None] + 0)       # :190
    gt_semantics = np.zeros(len(fg), dtype=np.int64)
    gt_semantics[fg] = np.asarray(labels['per_instance_semantics'])[instances[fg]]   # :196
    gt_semantics[instances == -1] = 2                                                # :198
    gt_semantics[gt_unlabeled] = 0                                                   # :200
    return {'fg_instances': fg, 'gt_bb_bounds': gt_bb_bounds, 'gt_bb_offsets': gt_bb_offsets,
            'gt_semantics': gt_semantics}


# ------------------------------------------------------------------ the other association branches
def _mode(values):
    """scipy.stats.mode(values, None)[0][0] of the reference's SciPy (< 1.11): the most frequent value, the smallest
    one on ties."""
    u, c = np.unique(values, return_counts=True)
    return u[np.argmax(c)]


def _scannet_boxes(labels):
    semantics = np.asarray(labels['per_instance_semantics'])
    scene_fg = (semantics > 2) & (semantics != 22)                                   # dataloader.py:207-208
    centers = np.asarray(labels['per_instance_bb_centers'])[scene_fg]
    bounds = np.asarray(labels['per_instance_bb_bounds'])[scene_fg] + 0.005
    return centers - bounds, centers + bounds, np.asarray(labels['unique_instances'])[scene_fg], np.prod(2 * bounds, axis=1)


def _occupancy(positions, min_corner, max_corner):
    return (np.all(positions[None] >= min_corner[:, None], axis=-1) &
            np.all(positions[None] <= max_corner[:, None], axis=-1))                 # utils/util.py:91-92, (B, P)


def approx_association_points(positions, segments, labels, unique_segs, smallest_bb_heuristic, majority_vote):
    """dataloader.py:241-272: the point_association (majority_vote=False -> (inst_per_point, None)) and the
    majority_vote branch of ScanNet.approx_association (no dropout / noise)."""
    min_corner, max_corner, instance_ids, bb_volume = _scannet_boxes(labels)
    occ = _occupancy(positions, min_corner, max_corner)
    num = occ.sum(axis=0)
    inst_per_point = np.full(len(positions), -1, np.int64)                           # :244
    for i in range(len(positions)):
        if num[i] == 1:
            inst_per_point[i] = instance_ids[np.nonzero(occ[:, i])[0][0]]            # :246-248
        elif num[i] > 1:
            if not smallest_bb_heuristic:
                inst_per_point[i] = -2                                               # :250-251
            else:
                box_ids = np.nonzero(occ[:, i])[0]
                inst_per_point[i] = instance_ids[box_ids[np.argmin(bb_volume[box_ids])]]   # :253-256
    if not majority_vote:
        return inst_per_point, None                                                  # :258-259
    segments = np.asarray(segments)
    pooled = np.full(len(positions), -2, np.int64)                                   # :263-264
    per_seg = np.full(len(unique_segs), -2, np.int64)
    for i, seg_id in enumerate(unique_segs):                                         # :265-270
        m = seg_id == segments
        per_seg[i] = _mode(inst_per_point[m])
        pooled[m] = per_seg[i]
    return pooled, per_seg


def arkit_association(positions, segments, labels, unique_segs, point_association):
    """ARKitScenes.approx_association, dataloader.py:539-621: oriented boxes, point or segment association."""
    instance_ids = np.asarray(labels['unique_instances'])
    centers = np.asarray(labels['per_instance_bb_centers'])
    bounds = np.asarray(labels['per_instance_bb_bounds']) + 0.05                     # :547
    rotations = np.asarray(labels['per_instance_bb_rotations'])
    occ = np.zeros([rotations.shape[0], positions.shape[0]], dtype=bool)
    for i in range(rotations.shape[0]):                                              # :553-557
        pc = positions - centers[i]
        rot = np.reshape(rotations[i], [3, 3])
        q = (rot @ pc.T).T
        occ[i] = np.all(q >= -bounds[i, :], axis=-1) & np.all(q <= bounds[i, :], axis=-1)
    num = occ.sum(axis=0)
    if point_association:                                                            # :568-578
        inst_per_point = np.full(len(positions), -1, np.int64)
        one = num == 1
        inst_per_point[one] = instance_ids[np.argmax(occ[:, one], axis=0)]
        inst_per_point[num > 1] = -2
        return inst_per_point, None
    segments = np.asarray(segments)
    pooled = np.full(len(positions), -2, np.int64)                                   # :599-600
    per_seg = np.full(len(unique_segs), -2, np.int64)
    for i, seg_id in enumerate(unique_segs):                                         # :601-616
        idx = np.nonzero(seg_id == segments)[0]
        n_on = num[idx]
        m = n_on.min()
        if m == 1:
            p = idx[np.nonzero(n_on == 1)[0][0]]
            per_seg[i] = instance_ids[np.nonzero(occ[:, p])[0][0]]
            pooled[idx] = per_seg[i]
        elif m == 0:
            per_seg[i] = -1
            pooled[idx] = -1
    return pooled, per_seg


def s3dis_association(positions, segments, labels, unique_segs, point_association, ignore_wall_ceiling_floor):
    """S3DIS.approx_association, dataloader.py:805-927."""
    semantics = np.asarray(labels['per_instance_semantics'])
    scene_fg = (semantics > 2) if ignore_wall_ceiling_floor else (semantics >= 0)    # dataprocessing/s3dis.py:79-82
    inst = np.full(len(positions), -1, np.int64)
    sem = np.full(len(positions), -1, np.int64)
    for part, sel in enumerate((scene_fg, ~scene_fg)):                               # :813-852, :871-900
        ids = np.asarray(labels['unique_instances'])[sel]
        sids = semantics[sel]
        centers = np.asarray(labels['per_instance_bb_centers'])[sel]
        bounds = np.asarray(labels['per_instance_bb_bounds'])[sel] + 0.0001
        occ = _occupancy(positions, centers - bounds, centers + bounds)
        num = occ.sum(axis=0)
        open_ = (inst == -1) if part == 1 else np.ones(len(positions), bool)
        for b in range(len(ids)):
            m = occ[b] & (num == 1) & open_
            inst[m] = ids[b]
            sem[m] = sids[b]
        inst[(num > 1) & open_] = -2
        sem[(num > 1) & open_] = -100
    inst[inst == -1] = -2                                                            # :901-902
    sem[sem == -1] = -100
    if point_association:
        return inst, sem
    segments = np.asarray(segments)
    pooled = np.full(len(positions), -1, np.int64)                                   # :857
    per_seg = np.full(len(unique_segs), -2, np.int64)
    sem_seg = np.full(len(unique_segs), -100, np.int64)
    for i, seg_id in enumerate(unique_segs):                                         # :913-921
        m = seg_id == segments
        per_seg[i] = _mode(inst[m])
        sem_seg[i] = _mode(sem[m])
        pooled[m] = per_seg[i]
    return pooled, sem, per_seg, sem_seg
