"""Box-format helpers restating /root/reference/utils/util.py:46-70,94-98,123-130 (device-agnostic
torch glue around the hot path; no heavy compute)."""
from __future__ import annotations

import copy

import numpy as np
import torch


def to_bbs_min_max(locations, offsets, bounds, scores=None, use_torch=True):
    """centre = location + offset; box = [centre - bounds, centre + bounds]; the score column is
    PREPENDED -> (n,7) [score, min3, max3]  (util.py:46-64; the comment at util.py:45 says otherwise)."""
    centers = offsets + locations
    if use_torch:
        bbs = torch.zeros((centers.shape[0], 6), device=centers.device, dtype=centers.dtype)
        bbs[:, :3] = centers - bounds
        bbs[:, 3:] = centers + bounds
        if scores is not None:
            bbs = torch.cat((scores, bbs), axis=1)
    else:
        bbs = np.zeros((centers.shape[0], 6))
        bbs[:, :3] = centers - bounds
        bbs[:, 3:] = centers + bounds
        if scores is not None:
            bbs = np.concatenate((scores, bbs), axis=1)
    return bbs


def to_bbs_min_max_(centers, bounds, device):
    bounding_boxes = torch.zeros((bounds.shape[0], 6), device=device)
    bounding_boxes[:, :3] = centers - bounds
    bounding_boxes[:, 3:] = centers + bounds
    return bounding_boxes


def convertSecs(sec):
    seconds = int(sec % 60)
    minutes = int((sec / 60) % 60)
    hours = int((sec / (60 * 60)))
    return hours, minutes, seconds


def to_unique(segments):
    """Dense pooling ids over a batch (util.py:123-130)."""
    unique_segments = copy.deepcopy(segments)
    for i in range(1, len(unique_segments)):
        unique_segments[i] += np.max(unique_segments[i - 1]) + 1
    unique_segments = np.concatenate(unique_segments, 0)
    _, pooling_ids = np.unique(unique_segments, return_inverse=True)
    return torch.from_numpy(pooling_ids.reshape(-1)).long()
