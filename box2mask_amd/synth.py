"""Synthetic ScanNet-shaped scenes and batches (no dataset is available offline).

Follows the generator specified in SURVEY.md §8(d): an axis-aligned room (floor + wall
strips + 10-25 furniture boxes) whose surfaces are sampled at ``pts_per_m2`` points/m^2 and
voxelised exactly like the reference data loader (``/root/reference/models/dataloader.py:61-68``:
shift non-negative, divide by voxel size, ``np.round``, ``np.unique(axis=0)``).  The batch
dictionary produced by :func:`collate` has the layout of the reference ``collate_fn``
(``models/dataloader.py:946-995``), which is the input contract of the hot path.

Everything here is numpy on the host; it is *input generation*, not part of the measured path.
"""
from __future__ import annotations

import numpy as np
import torch

# /root/reference/dataprocessing/scannet.py:114-118,135-136 (class tables restated as constants)
SCANNET_SEMANTIC_VALID_CLASS_IDS = np.array(
    [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16, 24, 28, 33, 34, 36, 39])
SCANNET_INSTANCE_VALID_CLASS_IDS = np.array(
    [3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16, 24, 28, 33, 34, 36, 39])


def scannet_tables():
    """(semantic_valid_class_ids float tensor, semantic_id2idx long LUT, instance_id2idx, is_foreground)."""
    valid = torch.Tensor(SCANNET_SEMANTIC_VALID_CLASS_IDS)
    id2idx = torch.zeros(41).fill_(-100).long()
    id2idx[SCANNET_SEMANTIC_VALID_CLASS_IDS] = torch.arange(len(SCANNET_SEMANTIC_VALID_CLASS_IDS)).long()
    inst2idx = torch.zeros(41).fill_(-100).long()
    inst2idx[SCANNET_INSTANCE_VALID_CLASS_IDS] = torch.arange(len(SCANNET_INSTANCE_VALID_CLASS_IDS)).long()

    def is_foreground(sem):
        return (sem > 2) & (sem != 22)

    return valid, id2idx, inst2idx, is_foreground


def _sample_rect(rng, origin, eu, ev, normal, density, surf_id, cell):
    """Uniform samples on the parallelogram origin + a*eu + b*ev, a,b in [0,1)."""
    lu, lv = np.linalg.norm(eu), np.linalg.norm(ev)
    n = max(int(lu * lv * density), 1)
    a = rng.random(n)
    b = rng.random(n)
    pts = origin[None] + a[:, None] * eu[None] + b[:, None] * ev[None]
    nrm = np.broadcast_to(np.asarray(normal, dtype=np.float64), pts.shape)
    # segment = 0.25 m surface cell on this surface
    cu = np.floor(a * lu / cell).astype(np.int64)
    cv = np.floor(b * lv / cell).astype(np.int64)
    seg_key = (np.int64(surf_id) << 20) | (cu << 10) | cv
    return pts, nrm, seg_key


def make_scene(seed: int, target_voxels: int = 150_000, voxel_size: float = 0.02,
               pts_per_m2: float = 2.0e4, cell: float = 0.25, noise_sigma: float = 0.0,
               name: str | None = None, points_only: bool = False):
    """One synthetic indoor scene in the per-item format the reference Dataset returns.

    ``target_voxels`` scales the room so that the voxel count lands within about +-10 % of it
    (wall height is solved for after floor + furniture are placed; room size scales with sqrt).
    """
    rng = np.random.default_rng(1000 + seed)
    scale = np.sqrt(target_voxels / 150_000.0)
    L = rng.uniform(3.5, 6.0) * scale
    W = rng.uniform(3.0, 4.5) * scale
    n_box = int(rng.integers(10, 26))
    vox_per_m2 = 1.0 / (voxel_size * voxel_size)

    parts = []   # (pts, normals, seg_key, instance, semantic)
    surf = 0
    # floor (semantic id 2)
    p, n, s = _sample_rect(rng, np.array([0., 0., 0.]), np.array([L, 0, 0.]), np.array([0, W, 0.]),
                           (0, 0, 1), pts_per_m2, surf, cell)
    parts.append((p, n, s, -1, 2)); surf += 1
    area = L * W
    # furniture boxes (5 faces: top + 4 sides), semantic = random instance class
    boxes = []
    target_area = target_voxels / vox_per_m2
    for b in range(n_box):
        if b >= 3 and area > 0.85 * target_area - 0.3 * 2 * (L + W):   # leave room for >= 0.3 m of wall
            break
        sx, sy = rng.uniform(0.3, 1.2, 2) * min(scale, 1.0)
        sz = rng.uniform(0.3, 1.0) * min(scale, 1.0)
        sx, sy = min(sx, 0.8 * L), min(sy, 0.8 * W)
        cx = rng.uniform(sx / 2, L - sx / 2)
        cy = rng.uniform(sy / 2, W - sy / 2)
        x0, y0, x1, y1 = cx - sx / 2, cy - sy / 2, cx + sx / 2, cy + sy / 2
        sem = int(rng.choice(SCANNET_INSTANCE_VALID_CLASS_IDS))
        faces = [
            (np.array([x0, y0, sz]), np.array([sx, 0, 0.]), np.array([0, sy, 0.]), (0, 0, 1)),
            (np.array([x0, y0, 0.]), np.array([sx, 0, 0.]), np.array([0, 0, sz]), (0, -1, 0)),
            (np.array([x0, y1, 0.]), np.array([sx, 0, 0.]), np.array([0, 0, sz]), (0, 1, 0)),
            (np.array([x0, y0, 0.]), np.array([0, sy, 0.]), np.array([0, 0, sz]), (-1, 0, 0)),
            (np.array([x1, y0, 0.]), np.array([0, sy, 0.]), np.array([0, 0, sz]), (1, 0, 0)),
        ]
        for (o, eu, ev, nn) in faces:
            p, n, s = _sample_rect(rng, o, eu, ev, nn, pts_per_m2, surf, cell)
            parts.append((p, n, s, b, sem)); surf += 1
            area += np.linalg.norm(eu) * np.linalg.norm(ev)
        boxes.append((np.array([cx, cy, sz / 2]), np.array([sx / 2, sy / 2, sz / 2]), sem))
    # wall strips: height solved so that total surface area ~ target_voxels / vox_per_m2
    remaining = 1.02 * target_area - area
    H = float(np.clip(remaining / (2 * (L + W)), 0.3, 3.0))
    walls = [
        (np.array([0., 0., 0.]), np.array([L, 0, 0.]), np.array([0, 0, H]), (0, 1, 0)),
        (np.array([0., W, 0.]), np.array([L, 0, 0.]), np.array([0, 0, H]), (0, -1, 0)),
        (np.array([0., 0., 0.]), np.array([0, W, 0.]), np.array([0, 0, H]), (1, 0, 0)),
        (np.array([L, 0., 0.]), np.array([0, W, 0.]), np.array([0, 0, H]), (-1, 0, 0)),
    ]
    for (o, eu, ev, nn) in walls:
        p, n, s = _sample_rect(rng, o, eu, ev, nn, pts_per_m2, surf, cell)
        parts.append((p, n, s, -1, 1)); surf += 1

    positions = np.concatenate([q[0] for q in parts], 0)
    if noise_sigma > 0:
        positions = positions + rng.normal(0.0, noise_sigma, positions.shape)
    normals = np.concatenate([q[1] for q in parts], 0).astype(np.float64)
    seg_key = np.concatenate([q[2] for q in parts], 0)
    pt_inst = np.concatenate([np.full(len(q[0]), q[3], np.int64) for q in parts], 0)
    pt_sem = np.concatenate([np.full(len(q[0]), q[4], np.int64) for q in parts], 0)
    colors = rng.normal(0.0, 1.0, (len(positions), 3))
    _, segments = np.unique(seg_key, return_inverse=True)   # dense per-scene segment ids
    if points_only:     # the raw scene, as dataprocessing/scannet.py hands it to the dataset class (float64)
        segments = segments.reshape(-1).astype(np.int64)
        # labels in the layout of dataprocessing/scannet.py:432-436: furniture boxes are instances 0..B-1, floor
        # and walls follow as instances of their own (semantics 2 / 1)
        B = len(boxes)
        inst = np.where(pt_inst >= 0, pt_inst, B + (pt_sem == 1))
        seg2inst = np.zeros(int(segments.max()) + 1, np.int32)
        seg2inst[segments] = inst
        labels = {
            'unique_instances': np.arange(B + 2),
            'per_instance_semantics': np.array([b[2] for b in boxes] + [2, 1], np.int32),
            'per_instance_bb_centers': np.array([b[0] for b in boxes] + [[L / 2, W / 2, 0.0], [L / 2, W / 2, H / 2]],
                                                np.float32),
            'per_instance_bb_bounds': np.array([b[1] for b in boxes] + [[L / 2, W / 2, 0.01], [L / 2, W / 2, H / 2]],
                                               np.float32),
            'seg2inst': seg2inst,
        }
        return {'name': name or ('synth%04d' % seed), 'positions': positions, 'colors': colors,
                'normals': normals, 'segments': segments, 'labels': labels}

    # ---- voxelisation, as /root/reference/models/dataloader.py:61-68 ----
    input_coords = positions - min(0, np.min(positions))
    input_coords = input_coords / voxel_size
    vox_coords_f = np.round(input_coords)
    vox_coords, first_idx, vox2point = np.unique(vox_coords_f, axis=0, return_index=True, return_inverse=True)
    vox2point = vox2point.reshape(-1)
    # the reference associates each voxel with its nearest scene point (ball tree); any point
    # inside the voxel serves the same purpose for synthetic data: use the first one.
    point2vox = first_idx
    feats = np.concatenate([colors, normals], 1)[point2vox].astype(np.float32)
    vox_segments = segments[point2vox]
    vox_world = vox_coords * voxel_size + min(0, np.min(positions))

    # ---- per-segment quantities (dataloader.py:106-123) ----
    unique_segs, seg2vox = np.unique(vox_segments, return_inverse=True)
    S = len(unique_segs)
    cnt = np.bincount(seg2vox, minlength=S).astype(np.float64)
    input_location = np.stack([np.bincount(seg2vox, weights=vox_world[:, d], minlength=S) / cnt
                               for d in range(3)], 1)
    seg_inst = np.full(S, -1, np.int64)
    seg_sem = np.zeros(S, np.int64)
    seg_inst[seg2vox] = pt_inst[point2vox]
    seg_sem[seg2vox] = pt_sem[point2vox]
    fg = seg_inst > -1
    gt_bounds = np.zeros((S, 3))
    gt_centers = np.zeros((S, 3))
    if len(boxes):
        bc = np.stack([b[0] for b in boxes]); bb = np.stack([b[1] for b in boxes])
        gt_bounds[fg] = bb[seg_inst[fg]]
        gt_centers[fg] = bc[seg_inst[fg]]
    gt_offsets = gt_centers - input_location * fg[:, None]
    return {
        'scene': {'name': name or ('synth%04d' % seed)},
        'vox_coords': vox_coords,                      # (N,3) float (np.unique order = lexicographic)
        'vox_features': feats,                         # (N,6)
        'vox_segments': vox_segments,                  # (N,)
        'vox2point': vox2point, 'seg2vox': seg2vox,
        'input_location': input_location,              # (S,3)
        'gt_bb_bounds': gt_bounds, 'gt_bb_offsets': gt_offsets,
        'gt_semantics': seg_sem, 'fg_instances': fg,
        'labels': {'boxes': boxes},
    }


def batched_coordinates(coords_list):
    """[ME-mem] ``ME.utils.batched_coordinates``: list of (N_i,3) -> (sum N_i,4) int32 [b,x,y,z]
    (floats are floored).  Call site: /root/reference/models/dataloader.py:966."""
    out = []
    for b, c in enumerate(coords_list):
        c = np.floor(np.asarray(c)).astype(np.int32)
        out.append(np.concatenate([np.full((len(c), 1), b, np.int32), c], 1))
    return torch.from_numpy(np.concatenate(out, 0))


def to_unique(segments):
    """Dense pooling ids across scenes; restates /root/reference/utils/util.py:123-130."""
    segs = [np.array(s, copy=True) for s in segments]
    for i in range(1, len(segs)):
        segs[i] += np.max(segs[i - 1]) + 1
    cat = np.concatenate(segs, 0)
    _, pooling_ids = np.unique(cat, return_inverse=True)
    return torch.from_numpy(pooling_ids.reshape(-1)).long()


def collate(items, mode: str = 'train'):
    """Batch dict with the layout of the reference ``collate_fn`` (dataloader.py:954-995)."""
    ret = {}
    for it in items:
        for k, v in it.items():
            ret.setdefault(k, []).append(v)
    ret['vox_coords'] = batched_coordinates(ret['vox_coords'])
    ret['vox_features'] = torch.from_numpy(np.concatenate(ret['vox_features'], 0)).float()
    bids = [np.full(len(loc), b, np.int64) for b, loc in enumerate(ret['input_location'])]
    ret['batch_ids'] = torch.from_numpy(np.concatenate(bids, 0)).long()
    ret['input_location'] = torch.from_numpy(np.concatenate(ret['input_location'], 0)).float()
    ret['pooling_ids'] = to_unique(ret['vox_segments'])
    if mode == 'test':
        return ret
    ret['gt_bb_bounds'] = torch.from_numpy(np.concatenate(ret['gt_bb_bounds'], 0)).float()
    ret['gt_bb_offsets'] = torch.from_numpy(np.concatenate(ret['gt_bb_offsets'], 0)).float()
    ret['gt_semantics'] = torch.from_numpy(np.concatenate(ret['gt_semantics'], 0)).long()
    ret['fg_instances'] = torch.from_numpy(np.concatenate(ret['fg_instances'], 0)).bool()
    return ret


def make_batch(batch_size: int, seed0: int = 0, target_voxels: int = 150_000, mode: str = 'train', **kw):
    return collate([make_scene(seed0 + s, target_voxels=target_voxels, **kw) for s in range(batch_size)], mode)


def gt_instance_ids(batch, b: int) -> np.ndarray:
    """Per-point ground-truth ids ``label*1000 + instance`` of scene b of a synthetic batch, in the layout of ScanNet's
    gt_instance_data_txt files (what utils/eval_metric.py reads): furniture segments become instances (one per
    distinct box), floor / walls keep their semantic label with instance 0."""
    m = (batch['batch_ids'].cpu() == b).numpy()
    seg_sem = batch['gt_semantics'].cpu()[m].numpy()
    centre = (batch['input_location'].cpu()[m] + batch['gt_bb_offsets'].cpu()[m]).numpy().round(3)
    _, inst = np.unique(np.concatenate([centre, seg_sem[:, None]], 1), axis=0, return_inverse=True)
    seg_gt = np.where(batch['fg_instances'].cpu()[m].numpy(), seg_sem * 1000 + inst.reshape(-1) + 1, seg_sem * 1000)
    return seg_gt[np.asarray(batch['seg2vox'][b])][np.asarray(batch['vox2point'][b])].astype(np.int64)


def make_votes(seed: int, n_obj: int = 30, n_seg: int = 1500, sigma: float = 0.025):
    """Synthetic box votes for the clustering benchmark: ``n_seg`` segments vote for
    ``n_obj`` objects with Gaussian noise on offsets/bounds; score logits ~ N(0,2).
    Returns boxes (n_seg,7) fp32 ``[score, min3, max3]`` with tie-free scores."""
    rng = np.random.default_rng(2000 + seed)
    centers = rng.uniform([0, 0, 0.2], [6, 4.5, 1.2], (n_obj, 3))
    bounds = rng.uniform(0.15, 0.6, (n_obj, 3))
    which = rng.integers(0, n_obj, n_seg)
    c = centers[which] + rng.normal(0, sigma, (n_seg, 3))
    b = np.maximum(bounds[which] + rng.normal(0, sigma, (n_seg, 3)), 0.04)
    logits = rng.normal(0, 2.0, n_seg)
    score = 1.0 / (1.0 + np.exp(-logits))
    boxes = np.concatenate([score[:, None], c - b, c + b], 1).astype(np.float32)
    # enforce tie-free scores (reference argsort is unstable: iou_nms.py:78)
    _, idx = np.unique(boxes[:, 0], return_index=True)
    boxes = boxes[np.sort(idx)]
    return boxes
