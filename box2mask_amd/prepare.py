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


def _unique_inverse(keys: torch.Tensor):
    """np.unique(keys, return_inverse=True) for non-negative int64 keys on the device.
    Returns (sorted unique keys (padded buffer), n_unique, inverse int64, table keys, table vals, cap)."""
    dev = keys.device
    n = keys.shape[0]
    cap = _pow2(max(2 * n, 16))
    tkeys = torch.empty(cap, dtype=torch.int64, device=dev)
    tvals = torch.empty(cap, dtype=torch.int32, device=dev)
    slot_of = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    ukeys = torch.full((_pow2(max(n, 2)),), -1, dtype=torch.int64, device=dev)       # 2^64-1 = padding, sorts last
    n_unique = torch.empty(1, dtype=torch.int32, device=dev)
    lib = _lib.load()
    nu = lib.b2m_unique_insert(ptr(keys), n, ptr(tkeys), cap, ptr(slot_of), ptr(ukeys), ptr(n_unique), _lib.stream())
    if nu < 0:
        raise _lib.B2MError('b2m_unique_insert failed (%d): %s' % (nu, lib.b2m_last_error().decode()))
    _lib.call('b2m_sort_u64', ptr(ukeys), _pow2(max(nu, 2)))
    inverse = torch.empty(n, dtype=torch.int64, device=dev)
    _lib.call('b2m_unique_rank', ptr(ukeys), nu, ptr(tkeys), ptr(tvals), cap, ptr(slot_of), n, ptr(inverse))
    return ukeys, int(nu), inverse, tkeys, tvals, cap


def voxelize_scene(scene: dict, voxel_size: float, use_normals_input: bool = True, device=None) -> dict:
    """The voxelisation block of the dataset item (dataloader.py:61-123, do_segment_pooling=True).

    scene: {'positions' (P,3), 'colors' (P,3), 'normals' (P,3), 'segments' (P,)} numpy or torch (float64 / int64,
    as dataprocessing/scannet.py:412 provides them).  Returns the reference's item keys."""
    _lib.require_gpu()
    dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
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
    ukeys, N, vox2point, tkeys, tvals, cap = _unique_inverse(keys)          # synchronises
    if int(bad.item()):
        raise ValueError('voxelize_scene: %d points fall outside 2^21 voxels per axis (or are NaN)' % int(bad.item()))
    coords = torch.empty((N, 4), dtype=torch.int32, device=dev)
    _lib.call('b2m_vox_decode', ptr(ukeys), N, 0, ptr(coords))
    best = torch.empty(N, dtype=torch.int64, device=dev)
    point2vox = torch.empty(N, dtype=torch.int32, device=dev)
    _lib.call('b2m_vox_nearest', ptr(pos), P, ptr(shift), float(voxel_size), ptr(tkeys), ptr(tvals), cap, N,
              ptr(best), ptr(point2vox))
    feats = torch.empty((N, 6 if use_normals_input else 3), dtype=torch.float32, device=dev)
    vox_segments = torch.empty(N, dtype=torch.int64, device=dev)
    _lib.call('b2m_vox_gather', ptr(point2vox), N, ptr(colors), ptr(normals), ptr(segments), ptr(feats),
              ptr(vox_segments))

    # ---- segments (dataloader.py:106-120)
    if int(segments.min().item()) < 0:
        raise ValueError('voxelize_scene: negative segment ids')
    useg, S, seg2vox, seg_tkeys, seg_tvals, seg_cap = _unique_inverse(vox_segments)
    sums = torch.empty(3 * S, dtype=torch.int64, device=dev)
    counts = torch.empty(S, dtype=torch.int32, device=dev)
    middle = torch.empty((S, 3), dtype=torch.float64, device=dev)
    _lib.call('b2m_seg_centroid', ptr(coords), ptr(seg2vox), N, S, float(voxel_size), ptr(shift), ptr(sums),
              ptr(counts), ptr(middle))
    seg2point = seg2vox[vox2point]
    return {
        'scene': scene, 'vox_coords': coords, 'vox2point': vox2point, 'point2vox': point2vox.long(),
        'vox_segments': vox_segments, 'vox_features': feats, 'seg2vox': seg2vox, 'seg2point': seg2point,
        'pred2point': seg2point, 'input_location': middle, 'unique_vox_segments': useg[:S],
        'voxel_shift': shift, 'voxel_size': float(voxel_size),
        '_device_scene': {'positions': pos, 'segments': segments}, '_segment_table': (seg_tkeys, seg_tvals, seg_cap),
    }


def vox_world_coords(item: dict) -> torch.Tensor:
    """ret['vox_world_coords'] of dataloader.py:94: voxel centres in the scene's world frame (fp64)."""
    return item['vox_coords'][:, 1:].double() * item['voxel_size'] + item['voxel_shift']


def box_supervision(item: dict, labels: dict, cfg) -> dict:
    """``bbs_supervision`` + ``approx_association`` of the dataset class (dataloader.py:165-314) for the ScanNet
    configuration (do_segment_pooling, no point_association / majority_vote): weak box labels -> per-segment
    instance, box and semantic targets.  Adds the reference's keys to ``item`` and returns it.

    labels: 'per_instance_semantics' (I,), 'per_instance_bb_centers' (I,3) f32, 'per_instance_bb_bounds' (I,3) f32,
    'unique_instances' (I,), 'seg2inst' (max segment id + 1,) -- dataprocessing/scannet.py:432-436."""
    if getattr(cfg, 'point_association', False) or getattr(cfg, 'majority_vote', False):
        raise NotImplementedError('only the segment association of configs/scannet.txt is on the device')
    pos, segments = item['_device_scene']['positions'], item['_device_scene']['segments']
    dev = pos.device
    P = pos.shape[0]
    scene = item['scene']
    name = scene['name'] if isinstance(scene, dict) else str(scene)
    # ---- boxes of the foreground instances (:206-233); a handful of rows, prepared on the host like the reference
    semantics = np.asarray(labels['per_instance_semantics'])
    scene_fg = (semantics > 2) & (semantics != 22)
    if getattr(cfg, 'dropout_boxes', None):
        rng = np.random.default_rng(seed=abs(int(name, 36)))
        scene_fg[scene_fg] = rng.binomial(1, 1 - cfg.dropout_boxes, scene_fg.sum()) != 0
    centers = np.asarray(labels['per_instance_bb_centers'])[scene_fg]
    bounds = np.asarray(labels['per_instance_bb_bounds'])[scene_fg] + 0.005
    min_corner, max_corner = centers - bounds, centers + bounds
    instance_ids = np.asarray(labels['unique_instances'])[scene_fg]
    if getattr(cfg, 'noisy_boxes', None):
        rng = np.random.default_rng(seed=abs(int(name, 36)))
        # in place, like the reference: the float32 corner arrays absorb the float64 noise with a rounding
        min_corner += rng.normal(loc=0, scale=cfg.noisy_boxes / 2, size=min_corner.shape)
        max_corner += rng.normal(loc=0, scale=cfg.noisy_boxes / 2, size=max_corner.shape)
        item['noisy_bbs'] = min_corner, max_corner
    bb_volume = np.prod(2 * bounds, axis=1)
    B = len(instance_ids)
    d_min, d_max = _dev(min_corner.reshape(-1, 3), torch.float64, dev), _dev(max_corner.reshape(-1, 3), torch.float64, dev)
    d_vol = _dev(bb_volume.reshape(-1), torch.float32, dev)
    d_ids = _dev(instance_ids.reshape(-1), torch.int64, dev)
    count = torch.empty(P, dtype=torch.int32, device=dev)
    first_bb, smallest_bb = torch.empty_like(count), torch.empty_like(count)
    _lib.call('b2m_box_membership', ptr(pos), P, ptr(d_min) if B else None, ptr(d_max) if B else None,
              ptr(d_vol) if B else None, B, ptr(count), ptr(first_bb), ptr(smallest_bb))
    tkeys, tvals, cap = item['_segment_table']
    S = item['input_location'].shape[0]
    best = torch.empty(S, dtype=torch.int64, device=dev)
    seg_of_point = torch.empty(P, dtype=torch.int32, device=dev)
    inst_per_seg = torch.empty(S, dtype=torch.int64, device=dev)
    inst_per_point = torch.empty(P, dtype=torch.int64, device=dev)
    _lib.call('b2m_seg_box_vote', ptr(segments), P, ptr(tkeys), ptr(tvals), cap, S, ptr(count), ptr(first_bb),
              ptr(smallest_bb), ptr(d_ids) if B else None, 1 if getattr(cfg, 'smallest_bb_heuristic', False) else 0,
              ptr(best), ptr(seg_of_point), ptr(inst_per_seg), ptr(inst_per_point))
    item['pseudo_inst'] = inst_per_point, inst_per_seg
    # ---- per-segment targets (:176-200): gathers over a few thousand rows
    instances = inst_per_seg
    seg2inst = _dev(np.asarray(labels['seg2inst']), torch.int64, dev)
    per_sem = _dev(semantics, torch.int64, dev)
    gt_full_sem = per_sem[seg2inst[item['unique_vox_segments']]]
    fg = instances > -1
    safe = instances.clamp_min(0)
    per_bounds = _dev(np.asarray(labels['per_instance_bb_bounds']), torch.float64, dev)
    per_centers = _dev(np.asarray(labels['per_instance_bb_centers']), torch.float64, dev)
    fgc = fg[:, None].to(torch.float64)
    item['fg_instances'] = fg
    item['gt_bb_bounds'] = per_bounds[safe] * fgc
    item['gt_bb_offsets'] = per_centers[safe] * fgc - item['input_location'] * fgc
    sem = torch.where(fg, per_sem[safe], torch.zeros_like(safe))
    sem = torch.where(instances == -1, torch.full_like(sem, 2), sem)
    item['gt_semantics'] = torch.where(gt_full_sem == 0, torch.zeros_like(sem), sem)
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
