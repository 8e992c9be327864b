"""Scene preparation on the device: voxelisation, nearest-point association, segment ranks / centroids, collate.

Host mirror of the voxelisation block of ``ScanNet.__getitem__`` (/root/reference/models/dataloader.py:61-123) and of
``collate_fn`` (:946-995, with ``to_unique`` of /root/reference/utils/util.py:123-130) over the C ABI of
include/b2m_prepare.h.  Same dictionary keys as the reference; values are torch tensors ON THE DEVICE (the
reference returns numpy arrays from CPU workers), so a batch goes from raw scene points to ``Model.compute_loss``
without crossing PCIe again.  There is no CPU fallback.

Deviations: ``vox_coords`` of an item is the int32 ``[b,x,y,z]`` row array the reference only makes in
collate; among scene points at exactly the same distance from a voxel centre the lowest index is associated (the
reference's ball tree picks by traversal order); segment centroids agree to a few fp64 ulp (integer sums).
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from ._lib import ptr


def _pow2(n: int) -> int:
    return 1 << max(1, (int(n) - 1).bit_length())


def _dev(x, dtype, device):
    t = torch.as_tensor(np.ascontiguousarray(x) if isinstance(x, np.ndarray) else x)
    return t.to(device=device, dtype=dtype).contiguous()


def _unique_begin(keys: torch.Tensor):
    """First half of np.unique(keys, return_inverse=True) for non-negative int64 keys on the device: the hash insert, WITHOUT the
    host read of the number of distinct keys (it stays in `n_unique` on the device)."""
    dev = keys.device
    n = keys.shape[0]
    cap = _pow2(max(2 * n, 16))
    tkeys = torch.empty(cap, dtype=torch.int64, device=dev)
    tvals = torch.empty(cap, dtype=torch.int32, device=dev)
    slot_of = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    ukeys = torch.full((_pow2(max(n, 2)),), -1, dtype=torch.int64, device=dev)       # 2^64-1 = padding, sorts last
    n_unique = torch.empty(1, dtype=torch.int32, device=dev)
    _lib.call('b2m_unique_insert_async', ptr(keys), n, ptr(tkeys), cap, ptr(slot_of), ptr(ukeys), ptr(n_unique))
    return dict(n=n, cap=cap, tkeys=tkeys, tvals=tvals, slot_of=slot_of, ukeys=ukeys, n_unique=n_unique)


def _unique_finish(u: dict, nu: int):
    """Second half, once the count is on the host: sort of the distinct keys, ranks, inverse.
    Returns (sorted unique keys (padded buffer), n_unique, inverse int64, table keys, table vals, cap)."""
    _lib.call('b2m_sort_u64', ptr(u['ukeys']), _pow2(max(nu, 2)))
    inverse = torch.empty(u['n'], dtype=torch.int64, device=u['ukeys'].device)
    _lib.call('b2m_unique_rank', ptr(u['ukeys']), nu, ptr(u['tkeys']), ptr(u['tvals']), u['cap'], ptr(u['slot_of']), u['n'],
              ptr(inverse))
    return u['ukeys'], int(nu), inverse, u['tkeys'], u['tvals'], u['cap']


def _unique_inverse(keys: torch.Tensor):
    """np.unique(keys, return_inverse=True) for non-negative int64 keys on the device (one host read: the count)."""
    u = _unique_begin(keys)
    return _unique_finish(u, int(u['n_unique'].item()))


def voxelize_scene(scene: dict, voxel_size: float, use_normals_input: bool = True, device=None,
                   do_segment_pooling: bool = True) -> dict:
    """The voxelisation block of the dataset item -- ScanNet.__getitem__ (dataloader.py:61-123), identical in
    ARKitScenes.__getitem__ (:394-455, 4 cm voxels in configs/arkitscenes.txt) and S3DIS.__getitem__ (:671-730).
    With do_segment_pooling=False the predictions live on the voxels: input_location = vox_world_coords,
    pred2point = vox2point (:98-105).

    scene: {'positions' (P,3), 'colors' (P,3), 'normals' (P,3), 'segments' (P,)} numpy or torch (float64 / int64,
    as dataprocessing/scannet.py:412 provides them).  Returns the reference's item keys."""
    return voxelize_scenes([scene], voxel_size, use_normals_input, device, do_segment_pooling)[0]


def voxelize_scenes(scenes, voxel_size: float, use_normals_input: bool = True, device=None,
                    do_segment_pooling: bool = True) -> list:
    """voxelize_scene for all scenes of a batch with TWO host reads per batch (round 5; one scene used to cost four): every
    scene's kernels of a stage are queued before the stage's small results -- the voxel counts with the range / segment-id
    checks, then the segment counts -- come back in one copy.  Same kernels, same order per scene: the same bits."""
    _lib.require_gpu()
    dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    st = []
    # ---- stage A: keys and the voxel hash of every scene
    for scene in scenes:
        pos = _dev(scene['positions'], torch.float64, dev)
        assert pos.dim() == 2 and pos.shape[1] == 3, 'positions must be (P,3)'
        P = pos.shape[0]
        if P == 0:
            raise ValueError('voxelize_scene: empty scene')
        colors = _dev(scene['colors'], torch.float64, dev)
        normals = _dev(scene['normals'], torch.float64, dev) if use_normals_input else None
        segments = _dev(scene['segments'], torch.int64, dev).reshape(-1)
        assert colors.shape == (P, 3) and segments.shape == (P,) and (normals is None or normals.shape == (P, 3))
        shift = torch.empty(1, dtype=torch.float64, device=dev)
        scratch = torch.empty(1, dtype=torch.int64, device=dev)
        _lib.call('b2m_vox_shift', ptr(pos), P, ptr(shift), ptr(scratch))
        keys = torch.empty(P, dtype=torch.int64, device=dev)
        bad = torch.empty(1, dtype=torch.int32, device=dev)
        _lib.call('b2m_vox_keys', ptr(pos), P, ptr(shift), float(voxel_size), ptr(keys), ptr(bad))
        u = _unique_begin(keys)
        st.append(dict(scene=scene, pos=pos, P=P, colors=colors, normals=normals, segments=segments, shift=shift, bad=bad, u=u,
                       seg_min=segments.min()))
    # one read: voxel counts, out-of-range points, smallest segment id of every scene
    head = torch.stack([torch.stack([s['u']['n_unique'][0].long(), s['bad'][0].long(), s['seg_min']]) for s in st]).cpu().tolist()
    for s, (nu, nbad, smin) in zip(st, head):
        if nbad:
            raise ValueError('voxelize_scene: %d points fall outside 2^21 voxels per axis (or are NaN)' % nbad)
        if smin < 0:
            raise ValueError('voxelize_scene: negative segment ids')
        s['N'] = int(nu)
    # ---- stage B: voxel rows, nearest points, features; the segment hash
    for s in st:
        P, pos, shift = s['P'], s['pos'], s['shift']
        ukeys, N, vox2point, tkeys, tvals, cap = _unique_finish(s['u'], s['N'])
        coords = torch.empty((N, 4), dtype=torch.int32, device=dev)
        _lib.call('b2m_vox_decode', ptr(ukeys), N, 0, ptr(coords))
        best = torch.empty(N, dtype=torch.int64, device=dev)
        point2vox = torch.empty(N, dtype=torch.int32, device=dev)
        _lib.call('b2m_vox_nearest', ptr(pos), P, ptr(shift), float(voxel_size), ptr(tkeys), ptr(tvals), cap, N,
                  ptr(best), ptr(point2vox))
        feats = torch.empty((N, 6 if use_normals_input else 3), dtype=torch.float32, device=dev)
        vox_segments = torch.empty(N, dtype=torch.int64, device=dev)
        _lib.call('b2m_vox_gather', ptr(point2vox), N, ptr(s['colors']), ptr(s['normals']), ptr(s['segments']), ptr(feats),
                  ptr(vox_segments))
        s.update(vox2point=vox2point, coords=coords, point2vox=point2vox, feats=feats, vox_segments=vox_segments,
                 useg=_unique_begin(vox_segments))          # segments (dataloader.py:106-120)
    seg_counts = torch.cat([s['useg']['n_unique'] for s in st]).cpu().tolist()          # the second (and last) read
    # ---- stage C: segment ranks and centroids
    items = []
    for s, S in zip(st, seg_counts):
        N, coords, shift, vox2point = s['N'], s['coords'], s['shift'], s['vox2point']
        useg, S, seg2vox, seg_tkeys, seg_tvals, seg_cap = _unique_finish(s['useg'], int(S))
        sums = torch.empty(3 * S, dtype=torch.int64, device=dev)
        counts = torch.empty(S, dtype=torch.int32, device=dev)
        middle = torch.empty((S, 3), dtype=torch.float64, device=dev)
        _lib.call('b2m_seg_centroid', ptr(coords), ptr(seg2vox), N, S, float(voxel_size), ptr(shift), ptr(sums),
                  ptr(counts), ptr(middle))
        seg2point = seg2vox[vox2point]
        item = {
            'scene': s['scene'], 'vox_coords': coords, 'vox2point': vox2point, 'point2vox': s['point2vox'].long(),
            'vox_segments': s['vox_segments'], 'vox_features': s['feats'], 'seg2vox': seg2vox, 'seg2point': seg2point,
            'pred2point': seg2point, 'input_location': middle, 'unique_vox_segments': useg[:S],
            'voxel_shift': shift, 'voxel_size': float(voxel_size), 'do_segment_pooling': bool(do_segment_pooling),
            '_device_scene': {'positions': s['pos'], 'segments': s['segments']}, '_segment_table': (seg_tkeys, seg_tvals, seg_cap),
        }
        if not do_segment_pooling:
            # :98-105: no segment keys; unique_vox_segments stays (the box supervision recomputes it, :131-132)
            for k in ('seg2vox', 'seg2point'):
                del item[k]
            item['input_location'] = vox_world_coords(item)
            item['pred2point'] = vox2point
        items.append(item)
    return items


def vox_world_coords(item: dict) -> torch.Tensor:
    """ret['vox_world_coords'] of dataloader.py:94: voxel centres in the scene's world frame (fp64)."""
    return item['vox_coords'][:, 1:].double() * item['voxel_size'] + item['voxel_shift']


def _np(x):
    return x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x)


def _scene_name(item):
    scene = item['scene']
    return scene['name'] if isinstance(scene, dict) else str(scene)


def _axis_box_membership(pos, min_corner, max_corner, bb_volume):
    """count / first box / smallest box per point for axis-aligned boxes (b2m_box_membership)."""
    dev, P, B = pos.device, pos.shape[0], len(min_corner)
    d_min, d_max = _dev(min_corner.reshape(-1, 3), torch.float64, dev), _dev(max_corner.reshape(-1, 3), torch.float64, dev)
    d_vol = _dev(bb_volume.reshape(-1), torch.float32, dev)
    count = torch.empty(P, dtype=torch.int32, device=dev)
    first_bb, smallest_bb = torch.empty_like(count), torch.empty_like(count)
    _lib.call('b2m_box_membership', ptr(pos), P, ptr(d_min) if B else None, ptr(d_max) if B else None,
              ptr(d_vol) if B else None, B, ptr(count), ptr(first_bb), ptr(smallest_bb))
    return count, first_bb, smallest_bb


def _seg_rank(item):
    """Rank of every scene point's segment among the voxel-level segments, -1 for segments without a voxel."""
    segments = item['_device_scene']['segments']
    tkeys, tvals, cap = item['_segment_table']
    out = torch.empty(segments.shape[0], dtype=torch.int32, device=segments.device)
    _lib.call('b2m_seg_rank', ptr(segments), segments.shape[0], ptr(tkeys), ptr(tvals), cap, ptr(out))
    return out


def _seg_mode(seg_of_point, values, n_seg):
    """scipy.stats.mode of `values` over the points of every segment (smallest value on ties): (n_seg,) int64."""
    dev = values.device
    uniq = torch.unique(values)                                 # ascending: class index order = value order
    cls = torch.searchsorted(uniq, values).int().contiguous()
    C = int(uniq.shape[0])
    hist = torch.empty(max(n_seg * C, 1), dtype=torch.int32, device=dev)
    mode_cls = torch.empty(max(n_seg, 1), dtype=torch.int32, device=dev)
    _lib.call('b2m_seg_mode', ptr(seg_of_point), ptr(cls), values.shape[0], n_seg, C, ptr(hist), ptr(mode_cls))
    return uniq[mode_cls[:n_seg].long()]


def _segment_vote(item, count, first_bb, smallest_bb, d_ids, heuristic):
    """The segment branch of approx_association (dataloader.py:274-309 / :596-618): b2m_seg_box_vote."""
    segments = item['_device_scene']['segments']
    dev, P = segments.device, segments.shape[0]
    tkeys, tvals, cap = item['_segment_table']
    S = item['unique_vox_segments'].shape[0]
    best = torch.empty(S, dtype=torch.int64, device=dev)
    seg_of_point = torch.empty(P, dtype=torch.int32, device=dev)
    inst_per_seg = torch.empty(S, dtype=torch.int64, device=dev)
    inst_per_point = torch.empty(P, dtype=torch.int64, device=dev)
    _lib.call('b2m_seg_box_vote', ptr(segments), P, ptr(tkeys), ptr(tvals), cap, S, ptr(count), ptr(first_bb),
              ptr(smallest_bb), ptr(d_ids) if d_ids.numel() else None, 1 if heuristic else 0,
              ptr(best), ptr(seg_of_point), ptr(inst_per_seg), ptr(inst_per_point))
    return inst_per_point, inst_per_seg


def _point_votes(count, first_bb, multi, d_ids):
    """Per-point instance (dataloader.py:244-258): one box -> its id, several -> `multi` (a tensor or -2), none -> -1."""
    one = d_ids[first_bb.clamp_min(0).long()] if d_ids.numel() else torch.zeros_like(count, dtype=torch.int64)
    many = multi if torch.is_tensor(multi) else torch.full_like(one, int(multi))
    return torch.where(count == 1, one, torch.where(count > 1, many, torch.full_like(one, -1)))


def approx_association(item: dict, labels: dict, cfg, dataset: str = 'scannet'):
    """``approx_association`` of the three dataset classes on the device.

    scannet (dataloader.py:203-314): axis-aligned foreground boxes (+0.005), optional dropout_boxes / noisy_boxes;
      point_association -> (inst_per_point, None); majority_vote -> scipy.stats.mode per segment; else the segment vote
      with the optional smallest_bb_heuristic.
    arkitscenes (:539-621): oriented boxes (+0.05, R (p - c) within the half sizes); point_association or segment vote.
    s3dis (:805-927): foreground boxes first, background boxes for the points no foreground box holds (+0.0001);
      returns (inst_per_point, sem_per_point) for point_association, else
      (inst_per_point_pooled, sem_per_point, inst_per_seg, sem_per_seg) by majority vote."""
    pos = item['_device_scene']['positions']
    dev = pos.device
    P = pos.shape[0]
    name = _scene_name(item)
    S = item['unique_vox_segments'].shape[0]
    if dataset == 'scannet':
        semantics = np.array(_np(labels['per_instance_semantics']))
        scene_fg = (semantics > 2) & (semantics != 22)
        if getattr(cfg, 'dropout_boxes', None):
            rng = np.random.default_rng(seed=abs(int(name, 36)))
            scene_fg[scene_fg] = rng.binomial(1, 1 - cfg.dropout_boxes, scene_fg.sum()) != 0
        centers = _np(labels['per_instance_bb_centers'])[scene_fg]
        bounds = _np(labels['per_instance_bb_bounds'])[scene_fg] + 0.005
        min_corner, max_corner = centers - bounds, centers + bounds
        instance_ids = _np(labels['unique_instances'])[scene_fg]
        if getattr(cfg, 'noisy_boxes', None):
            rng = np.random.default_rng(seed=abs(int(name, 36)))
            # in place, like the reference: the float32 corner arrays absorb the float64 noise with a rounding
            min_corner += rng.normal(loc=0, scale=cfg.noisy_boxes / 2, size=min_corner.shape)
            max_corner += rng.normal(loc=0, scale=cfg.noisy_boxes / 2, size=max_corner.shape)
            item['noisy_bbs'] = min_corner, max_corner
        bb_volume = np.prod(2 * bounds, axis=1)
        d_ids = _dev(instance_ids.reshape(-1), torch.int64, dev)
        count, first_bb, smallest_bb = _axis_box_membership(pos, min_corner, max_corner, bb_volume)
        heuristic = bool(getattr(cfg, 'smallest_bb_heuristic', False))
        if getattr(cfg, 'point_association', False) or getattr(cfg, 'majority_vote', False):
            multi = d_ids[smallest_bb.clamp_min(0).long()] if (heuristic and d_ids.numel()) else -2
            inst_per_point = _point_votes(count, first_bb, multi, d_ids)
            if getattr(cfg, 'point_association', False):
                return inst_per_point, None
            rank = _seg_rank(item)                                                  # :262-270
            inst_per_seg = _seg_mode(rank, inst_per_point, S)
            pooled = torch.where(rank >= 0, inst_per_seg[rank.clamp_min(0).long()], torch.full_like(inst_per_point, -2))
            return pooled, inst_per_seg
        return _segment_vote(item, count, first_bb, smallest_bb, d_ids, heuristic)
    if dataset == 'arkitscenes':
        instance_ids = _np(labels['unique_instances'])
        centers = _np(labels['per_instance_bb_centers'])
        bounds = _np(labels['per_instance_bb_bounds']) + 0.05
        rot = _np(labels['per_instance_bb_rotations']).reshape(-1, 9)
        B = rot.shape[0]
        d_ids = _dev(instance_ids.reshape(-1), torch.int64, dev)
        count = torch.empty(P, dtype=torch.int32, device=dev)
        first_bb = torch.empty_like(count)
        d_c, d_r = _dev(centers.reshape(-1, 3), torch.float64, dev), _dev(rot, torch.float64, dev)
        d_h = _dev(bounds.reshape(-1, 3), torch.float64, dev)           # (kept alive until the launch has been issued)
        _lib.call('b2m_obb_membership', ptr(pos), P, ptr(d_c) if B else None, ptr(d_r) if B else None,
                  ptr(d_h) if B else None, B, ptr(count), ptr(first_bb))
        if getattr(cfg, 'point_association', False):
            return _point_votes(count, first_bb, -2, d_ids), None
        return _segment_vote(item, count, first_bb, first_bb, d_ids, False)
    if dataset == 's3dis':
        semantics = _np(labels['per_instance_semantics'])
        scene_fg = (semantics > 2) if getattr(cfg, 'ignore_wall_ceiling_floor', False) else (semantics >= 0)
        inst = torch.full((P,), -1, dtype=torch.int64, device=dev)
        sem = torch.full((P,), -1, dtype=torch.int64, device=dev)
        for part, sel in enumerate((scene_fg, ~scene_fg)):                          # foreground first, then background
            ids = _dev(_np(labels['unique_instances'])[sel].reshape(-1), torch.int64, dev)
            sids = _dev(semantics[sel].reshape(-1), torch.int64, dev)
            centers = _np(labels['per_instance_bb_centers'])[sel]
            bounds = _np(labels['per_instance_bb_bounds'])[sel] + 0.0001
            count, first_bb, _ = _axis_box_membership(pos, centers - bounds, centers + bounds,
                                                      np.zeros(len(centers), np.float32))
            open_ = (inst == -1) if part == 1 else torch.ones_like(inst, dtype=torch.bool)     # :893 undecided_mask
            one, many = (count == 1) & open_, (count > 1) & open_
            if ids.numel():
                inst = torch.where(one, ids[first_bb.clamp_min(0).long()], inst)
                sem = torch.where(one, sids[first_bb.clamp_min(0).long()], sem)
            inst = torch.where(many, torch.full_like(inst, -2), inst)
            sem = torch.where(many, torch.full_like(sem, -100), sem)
        inst = torch.where(inst == -1, torch.full_like(inst, -2), inst)             # :901-902
        sem = torch.where(sem == -1, torch.full_like(sem, -100), sem)
        if getattr(cfg, 'point_association', False):
            return inst, sem
        rank = _seg_rank(item)                                                      # :913-921
        inst_per_seg, sem_per_seg = _seg_mode(rank, inst, S), _seg_mode(rank, sem, S)
        pooled = torch.where(rank >= 0, inst_per_seg[rank.clamp_min(0).long()], torch.full_like(inst, -1))
        return pooled, sem, inst_per_seg, sem_per_seg
    raise ValueError('unknown dataset %r' % (dataset,))


def box_supervision(item: dict, labels: dict, cfg, dataset: str = 'scannet') -> dict:
    """``bbs_supervision`` + ``approx_association`` of the dataset classes (ScanNet dataloader.py:165-314, ARKitScenes
    :493-621, S3DIS :760-927): weak box labels -> per-segment (or, without segment pooling, per-voxel) instance, box
    and semantic targets.  Adds the reference's keys to ``item`` and returns it.

    labels: 'per_instance_semantics' (I,), 'per_instance_bb_centers' (I,3) f32, 'per_instance_bb_bounds' (I,3) f32,
    'unique_instances' (I,), 'seg2inst' (max segment id + 1,) -- dataprocessing/scannet.py:432-436 -- plus
    'per_instance_bb_rotations' (I,9) for ARKitScenes and the per-point 'semantics' without segment pooling."""
    pos = item['_device_scene']['positions']
    dev = pos.device
    pooling = item.get('do_segment_pooling', True)
    assoc = approx_association(item, labels, cfg, dataset)
    per_sem = _dev(_np(labels['per_instance_semantics']), torch.int64, dev)
    per_bounds = _dev(_np(labels['per_instance_bb_bounds']), torch.float64, dev)
    per_centers = _dev(_np(labels['per_instance_bb_centers']), torch.float64, dev)
    if dataset == 's3dis':
        if not pooling:
            # the reference reads sem_per_seg, which this branch never defines (dataloader.py:768 / :773): NameError there
            raise NotImplementedError('S3DIS.bbs_supervision needs do_segment_pooling (the reference fails without it)')
        inst_per_point, sem_per_point, instances, sem_per_seg = assoc
        item['pseudo_inst'] = inst_per_point, instances
        fg = (sem_per_seg > 2) if getattr(cfg, 'ignore_wall_ceiling_floor', False) else (sem_per_seg >= 0)
        bg = ~fg & (instances != -2)
        safe = instances.clamp_min(0)
        fgc = fg[:, None].to(torch.float64)
        item['fg_instances'] = fg
        item['gt_bb_bounds'] = per_bounds[safe] * fgc
        item['gt_bb_offsets'] = per_centers[safe] * fgc - item['input_location'] * fgc
        item['gt_semantics'] = torch.where(fg | bg, per_sem[safe], torch.full_like(safe, -100))
        item['gt_per_vox_semantics'] = sem_per_point[item['point2vox']]
        item['labels'] = labels
        return item
    inst_per_point, inst_per_seg = assoc
    item['pseudo_inst'] = inst_per_point, inst_per_seg
    if not pooling:
        instances = inst_per_point[item['point2vox']]                               # :170-171
        gt_full_sem = _dev(_np(labels['semantics']), torch.int64, dev)[item['point2vox']]
    else:
        if inst_per_seg is None:
            raise RuntimeError('point_association and segment pooling are incompatible (dataloader.py:173-174)')
        instances = inst_per_seg
        seg2inst = _dev(_np(labels['seg2inst']), torch.int64, dev)
        gt_full_sem = per_sem[seg2inst[item['unique_vox_segments']]] if dataset == 'scannet' else None
    fg = instances > -1
    safe = instances.clamp_min(0)
    fgc = fg[:, None].to(torch.float64)
    item['fg_instances'] = fg
    item['gt_bb_bounds'] = per_bounds[safe] * fgc
    item['gt_bb_offsets'] = per_centers[safe] * fgc - item['input_location'] * fgc
    sem = torch.where(fg, per_sem[safe], torch.zeros_like(safe))
    sem = torch.where(instances == -1, torch.full_like(sem, 2), sem)
    if dataset == 'scannet':                                                        # :200; ARKitScenes keeps it (:536)
        sem = torch.where(gt_full_sem == 0, torch.zeros_like(sem), sem)
    item['gt_semantics'] = sem
    item['labels'] = labels
    return item


def mask_supervision(item: dict, labels: dict, cfg, dataset: str = 'scannet') -> dict:
    """Full (mask) supervision of the dataset classes (ScanNet dataloader.py:138-161, ARKitScenes :472-491, S3DIS
    :737-758): ground-truth instance / box / semantic targets per segment, or per voxel without segment pooling."""
    dev = item['_device_scene']['positions'].device
    pooling = item.get('do_segment_pooling', True)
    seg2inst = _dev(_np(labels['seg2inst']), torch.int64, dev)
    item['vox_instances'] = seg2inst[item['vox_segments']]
    if not pooling:
        p2v = item['point2vox']
        item['gt_semantics'] = _dev(_np(labels['semantics']), torch.int64, dev)[p2v]
        item['gt_bb_bounds'] = _dev(_np(labels['bb_bounds']), torch.float64, dev)[p2v]
        centers = _dev(_np(labels['bb_centers']), torch.float64, dev)[p2v]
        item['instance_ids'] = item['vox_instances']
    else:
        seg_inst = seg2inst[item['unique_vox_segments']]
        item['gt_bb_bounds'] = _dev(_np(labels['per_instance_bb_bounds']), torch.float64, dev)[seg_inst]
        item['gt_semantics'] = _dev(_np(labels['per_instance_semantics']), torch.int64, dev)[seg_inst]
        centers = _dev(_np(labels['per_instance_bb_centers']), torch.float64, dev)[seg_inst]
        if dataset == 's3dis':
            item['gt_per_vox_semantics'] = _dev(_np(labels['semantics']), torch.int64, dev)[item['point2vox']]
        item['instance_ids'] = seg_inst
    item['gt_bb_offsets'] = centers - item['input_location']
    sem = item['gt_semantics']
    if dataset == 'scannet':
        item['fg_instances'] = (sem > 2) & (sem != 22)
    elif dataset == 'arkitscenes':
        item['fg_instances'] = sem > 2
    else:
        item['fg_instances'] = (sem > 2) if getattr(cfg, 'ignore_wall_ceiling_floor', False) else (sem >= 0)
    item['labels'] = labels
    return item


_GT_KEYS = (('gt_bb_bounds', torch.float32), ('gt_bb_offsets', torch.float32), ('gt_semantics', torch.int64),
            ('fg_instances', torch.bool), ('gt_per_vox_semantics', torch.int64))


def collate(items, mode: str = 'train') -> dict:
    """collate_fn.__call__ (dataloader.py:954-995) for do_segment_pooling=True: every key becomes a list over the
    scenes, then the tensors the model consumes are concatenated.  ``pooling_ids`` = per-scene segment rank plus the
    number of segments of the earlier scenes, which is what to_unique (util.py:123-130) computes."""
    ret = {}
    for it in items:
        for k, v in it.items():
            ret.setdefault(k, []).append(v)
    coords = []
    for b, c in enumerate(ret['vox_coords']):
        c = c.clone()
        c[:, 0] = b                                            # ME.utils.batched_coordinates (:966)
        coords.append(c)
    dev = coords[0].device
    ret['vox_coords'] = torch.cat(coords, 0)
    ret['vox_features'] = torch.cat(ret['vox_features'], 0).float()
    n_seg = [int(loc.shape[0]) for loc in ret['input_location']]
    ret['batch_ids'] = torch.cat([torch.full((n,), b, dtype=torch.int64, device=dev) for b, n in enumerate(n_seg)], 0)
    ret['input_location'] = torch.cat(ret['input_location'], 0).float()
    offs = np.concatenate([[0], np.cumsum(n_seg)[:-1]])
    ret['pooling_ids'] = torch.cat([s + int(o) for s, o in zip(ret['seg2vox'], offs)], 0)
    if mode == 'test':
        return ret
    for k, dt in _GT_KEYS:
        if k in ret:
            ret[k] = torch.cat([torch.as_tensor(v).to(dev) for v in ret[k]], 0).to(dt)
    return ret
