"""ORACLE (test infrastructure; never imported by the product).

numpy restatement of /root/reference/models/iou_nms.py (set_IOUs 4-22, torch_IOUs 26-45,
NMS_clustering 68-105, masks_iou 109-121, mask_NMS 130-144, semIOU 146-157), of the box helpers
/root/reference/utils/util.py:46-70 and of the per-scene body of SelectionNet.detection2mask
(/root/reference/models/detection_net.py:369-488).

PINNED: tests/test_oracle_nms.py checks every function bit-for-bit against golden vectors produced by
the real reference code in the build container (tools/gen_golden.py -> tests/golden/*.npz).

All arithmetic is explicit float32 in torch's evaluation order: prod over the three sides as
(s0*s1)*s2, union = ((a + b) - inter) + float32(1e-6), IEEE division; thresholds are compared as
float32 (torch compares a float32 tensor with a Python scalar in float32).
Score ties are visited in ascending row order (the reference's unstable argsort leaves them undefined).
"""
from __future__ import annotations

import numpy as np

F = np.float32
EPS = F(0.000001)


def box_ious(box, boxes):
    """torch_IOUs(box (6,), boxes (n,6)) -> (n,) float32."""
    box = np.asarray(box, F)
    boxes = np.asarray(boxes, F)
    bs = box[3:] - box[:3]
    ss = boxes[:, 3:] - boxes[:, :3]
    imin = np.maximum(box[:3][None], boxes[:, :3])
    imax = np.minimum(box[3:][None], boxes[:, 3:])
    d = imax - imin
    isl = np.where(d < 0, F(0), d).astype(F)
    inter = (isl[:, 0] * isl[:, 1]) * isl[:, 2]
    ba = (bs[0] * bs[1]) * bs[2]
    sa = (ss[:, 0] * ss[:, 1]) * ss[:, 2]
    union = ((ba + sa) - inter) + EPS
    with np.errstate(divide='ignore', invalid='ignore'):
        return (inter / union).astype(F)


def set_ious(a, b):
    """set_IOUs(boxes_a (n,6), boxes (n,6)) -> (n,) float32."""
    a = np.asarray(a, F)
    b = np.asarray(b, F)
    sa = a[:, 3:] - a[:, :3]
    sb = b[:, 3:] - b[:, :3]
    assert (sa >= 0).all() and (sb >= 0).all()
    d = np.minimum(a[:, 3:], b[:, 3:]) - np.maximum(a[:, :3], b[:, :3])
    isl = np.where(d < 0, F(0), d).astype(F)
    inter = (isl[:, 0] * isl[:, 1]) * isl[:, 2]
    aa = (sa[:, 0] * sa[:, 1]) * sa[:, 2]
    ab = (sb[:, 0] * sb[:, 1]) * sb[:, 2]
    union = ((aa + ab) - inter) + EPS
    return (inter / union).astype(F)


def to_bbs_min_max(locations, offsets, bounds, scores=None):
    """util.py:46-64: [score | centre-bounds | centre+bounds], score column first."""
    loc, off, bnd = (np.asarray(v, F) for v in (locations, offsets, bounds))
    c = off + loc
    bbs = np.concatenate([c - bnd, c + bnd], 1).astype(F)
    if scores is not None:
        bbs = np.concatenate([np.asarray(scores, F).reshape(-1, 1), bbs], 1)
    return bbs


def nms_clustering(boxes, cluster_th):
    """NMS_clustering(boxes (n,7), th, get_heatmaps=True) -> (reps int64 (K,), clusters list, heat (K,n) f32)."""
    boxes = np.asarray(boxes, F)
    assert boxes.ndim == 2 and boxes.shape[1] == 7 and 0 < cluster_th < 1
    th = F(cluster_th)
    remaining = np.argsort(-boxes[:, 0], kind='stable')
    b6 = boxes[:, 1:]
    reps, clusters, heats = [], [], []
    while len(remaining) > 0:
        r = remaining[0]
        heat = box_ious(b6[r], b6)
        heat[r] = F(1)
        ious = heat[remaining]
        keep = ious <= th
        reps.append(r)
        clusters.append(remaining[~keep])
        heats.append(heat)
        remaining = remaining[keep]
    return np.asarray(reps, np.int64), clusters, np.stack(heats, 0)


def masks_iou(mask, masks):
    inter = (mask[None] & masks).sum(1).astype(np.int64)
    union = (mask[None] | masks).sum(1).astype(np.int64)
    with np.errstate(divide='ignore', invalid='ignore'):
        return inter.astype(F) / union.astype(F)          # torch int64 true-division -> float32


def mask_nms(sorted_masks, th):
    """mask_NMS(sorted_masks (K,N) bool, th) -> (kept int64, suppressed list)."""
    th = F(th)
    remaining = np.arange(len(sorted_masks))
    kept, suppressed = [], []
    while len(remaining) > 0:
        m = sorted_masks[remaining]
        ious = masks_iou(m[0], m)
        ious[0] = F(1)
        keep = ious <= th
        kept.append(remaining[0])
        suppressed.append((remaining[0], remaining[~keep]))
        remaining = remaining[keep]
    return np.asarray(kept, np.int64), suppressed


def sem_iou(pred, gt):
    """semIOU (iou_nms.py:146-157) -> float64 array over the sorted labels present."""
    pred = np.asarray(pred)
    gt = np.asarray(gt)
    v = gt > -100
    gt, pred = gt[v], pred[v]
    out = []
    for l in np.unique(np.concatenate([gt, pred])):
        i = np.int64(((pred == l) & (gt == l)).sum())
        u = np.int64(((pred == l) | (gt == l)).sum())
        out.append(float(F(i) / (F(u) + EPS)))
    return np.array(out)


def detection2mask_scene(scene_bbs_all, scene_sem, is_foreground, seg2vox, vox2point, eval_ths, mode='eval',
                         sem_vox=None):
    """One scene of detection2mask (detection_net.py:390-477) for the segment-pooling / per-segment
    semantics flow.  scene_bbs_all: (S,7) boxes of all segments of the scene, scene_sem: (S,) raw class
    ids.  Returns dict(conf, label_id, mask) (+ intermediate results for kernel-level tests)."""
    cluster_th, score_th, mask_bin_th, mask_nms_th = eval_ths
    scene_bbs_all = np.asarray(scene_bbs_all, F)
    fg = np.asarray(is_foreground(scene_sem)).astype(bool)
    bbs = scene_bbs_all[fg]
    reps, clusters, heat = nms_clustering(bbs, cluster_th)
    scores = bbs[reps][:, 0]
    sel = scores > F(score_th)
    heat_s, scores_s, reps_s = heat[sel], scores[sel], reps[sel]
    w_bg = np.zeros((len(heat_s), len(fg)), F)
    w_bg[:, fg] = heat_s
    heat_vox = w_bg[:, seg2vox]
    sem_v = np.asarray(scene_sem)[seg2vox] if sem_vox is None else np.asarray(sem_vox)
    masks = heat_vox > F(mask_bin_th)
    kept, _ = mask_nms(masks, mask_nms_th)
    masks_k = masks[kept]
    labels = np.zeros(len(masks_k), np.int32)
    for i, m in enumerate(masks_k):
        labels[i] = np.argmax(np.bincount(sem_v[m]))
    out_masks = masks_k[:, vox2point] if mode == 'eval' else masks_k
    return {'conf': scores_s[kept], 'label_id': labels, 'mask': out_masks,
            'reps': reps, 'heat': heat, 'sel': np.nonzero(sel)[0], 'kept': kept, 'vox_masks': masks}
