"""Box-format and bookkeeping helpers of the votes -> masks path.

Behavioural contract (names, argument order, result layout) = /root/reference/utils/util.py:46-70 (box
corners), :94-98 (h/m/s split), :123-130 (dense pooling ids); the bodies are written from that contract and
pinned by tests/golden/iou_nms.npz (`bbs_out`, `uniq_out`).
"""
from __future__ import annotations

import numpy as np
import torch


def _corners(centers, half_extent, xp):
    # [min corner | max corner], one row per box
    return xp.cat((centers - half_extent, centers + half_extent), 1) if xp is torch else \
        np.concatenate((centers - half_extent, centers + half_extent), 1)


def to_bbs_min_max(locations, offsets, bounds, scores=None, use_torch=True):
    """Votes -> axis-aligned boxes: the voted centre is `locations + offsets`, `bounds` are half extents.
    Result rows are [score, min xyz, max xyz] when a (n,1) score column is given, else [min xyz, max xyz]
    (the score comes FIRST; the reference's comment at util.py:45 says otherwise, its code and every
    consumer -- iou_nms.py:79 sorts by column 0 -- say this).  numpy inputs with use_torch=False give a
    float64 array like the reference's np.zeros-based construction."""
    if use_torch:
        centers = offsets + locations
        boxes = _corners(centers, bounds, torch).to(torch.float32)
        return boxes if scores is None else torch.cat((scores, boxes), 1)
    centers = np.asarray(offsets) + np.asarray(locations)
    boxes = _corners(centers, np.asarray(bounds), np).astype(np.float64)
    return boxes if scores is None else np.concatenate((scores, boxes), 1)


def to_bbs_min_max_(centers, bounds, device):
    """Same corner layout from centres that are already absolute (loss code: model.py:98-100,147-150)."""
    return _corners(centers, bounds, torch).to(device=device, dtype=torch.float32)


def convertSecs(sec):
    """Seconds -> (hours, minutes, seconds) as ints; hours are not wrapped.  The three quotients are the float
    expressions checkpoint file names were written with (training.py:229), so names round-trip."""
    return tuple(int(v) for v in (sec / 3600, (sec / 60) % 60, sec % 60))


def to_unique(segments):
    """Per-scene segment id arrays -> one LongTensor of dense pooling ids over the batch: ids of scene i are
    shifted past the largest id of scenes < i, then ranked (np.unique order).  The caller's arrays are not
    modified."""
    shifted, base = [], 0
    for seg in segments:
        seg = np.asarray(seg)
        s = seg + base
        shifted.append(s)
        base = int(s.max()) + 1 if s.size else base
    flat = np.concatenate(shifted, 0) if shifted else np.zeros(0, np.int64)
    _, rank = np.unique(flat, return_inverse=True)
    return torch.from_numpy(np.ascontiguousarray(rank.reshape(-1))).long()
