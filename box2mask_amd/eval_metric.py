"""ScanNet instance-segmentation AP of predicted masks.

Mirrors the entry points of /root/reference/utils/eval_metric.py that ``Evaluater.scannet_eval`` drives
(models/evaluation.py:318-321): ``assign_instances_for_scan`` (:281-345), ``evaluate_matches`` (:102-260),
``compute_averages`` (:262-278), ``compute_eval`` (:450-474).  The expensive part of the reference -- one
``count_nonzero(gt_ids == id & pred_mask)`` over all scene points per (prediction, ground-truth instance) pair -- is
a single pass of ``b2m_mask_hist`` over the bit-packed masks; the matching itself walks a few hundred records per
scene and stays on the host.

Matches are kept as arrays per (scene, class) instead of the reference's nested dictionaries:
``gt_id, gt_vert`` (ground-truth instances of the class, ascending id), ``conf, vert, void`` (kept predictions of the
class in prediction order), ``inter`` (predictions x ground truths), ``uid`` (scene-wide prediction number).
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from ._lib import ptr

CLASS_LABELS = ['cabinet', 'bed', 'chair', 'sofa', 'table', 'door', 'window', 'bookshelf', 'picture', 'counter',
                'desk', 'curtain', 'refrigerator', 'shower curtain', 'toilet', 'sink', 'bathtub', 'otherfurniture']
VALID_CLASS_IDS = np.array([3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16, 24, 28, 33, 34, 36, 39])
OVERLAPS = np.append(np.arange(0.5, 0.95, 0.05), 0.25)         # eval_metric.py:17
MIN_REGION_SIZE = 100                                           # :19


def intersections(pred_masks, gt_ids):
    """(unique ground-truth ids (G,), their vertex counts (G,), inter (K,G) int64) for K masks over n points."""
    _lib.require_gpu()
    dev = torch.device('cuda', torch.cuda.current_device())
    masks = torch.as_tensor(pred_masks).to(dev)
    if masks.dtype != torch.bool:
        masks = masks != 0                                      # np.not_equal(pred_mask, 0), :306
    masks = masks.to(torch.uint8).contiguous()
    k, n = masks.shape
    ids = torch.as_tensor(np.ascontiguousarray(gt_ids) if isinstance(gt_ids, np.ndarray) else gt_ids).to(dev).long()
    assert ids.shape == (n,), 'mask length %d vs %d ground-truth ids' % (n, ids.shape[0])
    from .prepare import _unique_inverse
    if int(ids.min().item()) < 0:
        raise ValueError('negative ground-truth ids')
    ukeys, g, dense, _, _, _ = _unique_inverse(ids)              # np.unique(gt_ids) with the inverse, on the device
    uniq = ukeys[:g]
    if g > 2048:
        raise ValueError('more than 2048 ground-truth ids in one scene')
    words = (n + 63) // 64
    bits = torch.empty((k + 1, max(words, 1)), dtype=torch.int64, device=dev)
    allp = torch.ones((1, n), dtype=torch.uint8, device=dev)    # row k: every point -> vertex count of each id
    both = torch.cat([masks, allp], 0)
    _lib.call('b2m_mask_pack', ptr(both), k + 1, n, ptr(bits), words)
    hist = torch.empty((k + 1, g), dtype=torch.int32, device=dev)
    dense32 = dense.int().contiguous()
    _lib.call('b2m_mask_hist', ptr(bits), words, k + 1, ptr(dense32), n, g, ptr(hist))
    h = hist.cpu().numpy().astype(np.int64)
    return uniq.cpu().numpy(), h[k], h[:k]


def assign_instances_for_scan(scene_name, pred_info, gt_ids):
    """pred_info: {'conf' (K,), 'label_id' (K,), 'mask' (K,n)} as ``Model.pred2mask(..., 'eval')`` returns;
    gt_ids: (n,) int64 ``label*1000 + instance`` (the reference reads them from gt_instance_data_txt)."""
    label_id = np.asarray(torch.as_tensor(pred_info['label_id']).cpu()).astype(np.int64)
    conf = np.asarray(torch.as_tensor(pred_info['conf']).cpu())
    uniq, gt_vert, inter = intersections(pred_info['mask'], gt_ids)
    return assign_from_counts(scene_name, label_id, conf, uniq, gt_vert, inter)


def assign_from_counts(scene_name, label_id, conf, uniq, gt_vert, inter):
    """Host half of assign_instances_for_scan: the per-class records from the intersection counts."""
    pred_vert = inter.sum(1)
    void = inter[:, ~np.isin(uniq // 1000, VALID_CLASS_IDS)].sum(1)          # bool_void, :297
    valid = np.isin(label_id, VALID_CLASS_IDS) & (pred_vert >= MIN_REGION_SIZE)   # :301-309
    uid = np.cumsum(valid) - 1                                               # num_pred_instances numbering
    out = {}
    for cls in VALID_CLASS_IDS:
        gsel = np.nonzero((uniq != 0) & (uniq // 1000 == cls))[0]            # get_instances, :82-97
        psel = np.nonzero(valid & (label_id == cls))[0]
        out[int(cls)] = {'gt_id': uniq[gsel], 'gt_vert': gt_vert[gsel], 'conf': conf[psel], 'vert': pred_vert[psel],
                         'void': void[psel], 'inter': inter[np.ix_(psel, gsel)], 'uid': uid[psel]}
    return {'scene': scene_name, 'n_pred': int(valid.sum()), 'classes': out}


def _average_precision(y_true, y_score, hard_false_negatives):
    """Area under the precision/recall curve over the distinct score thresholds, eval_metric.py:205-247."""
    order = np.argsort(y_score)
    score, true = y_score[order], y_true[order]
    below = np.cumsum(true)
    _, first = np.unique(score, return_index=True)
    n_true = below[-1] if len(below) else 0
    below = np.append(below, 0)                        # index -1 -> nothing below the lowest threshold
    precision, recall = np.zeros(len(first) + 1), np.zeros(len(first) + 1)
    for j, i in enumerate(first):
        tp = n_true - below[i - 1]
        fp = len(score) - i - tp
        fn = below[i - 1] + hard_false_negatives
        precision[j] = float(tp) / (tp + fp)
        recall[j] = float(tp) / (tp + fn)
    precision[-1], recall[-1] = 1.0, 0.0
    r = np.append(np.append(recall[0], recall), 0.0)
    steps = np.convolve(r, [-0.5, 0, 0.5], 'valid')
    return float(np.dot(precision, steps)), {'p': precision, 'r': recall, 'rstep': steps}


def evaluate_matches(matches):
    """matches: {scene: assign_instances_for_scan(...)}.  Returns (ap (1, classes, overlaps), pr_curves)."""
    ap = np.zeros((1, len(CLASS_LABELS), len(OVERLAPS)))
    curves = {}
    for oi, th in enumerate(OVERLAPS):
        curves[th] = {}
        taken = {m: np.zeros(matches[m]['n_pred'], bool) for m in matches}       # pred_visited, :117-124
        for li, cls in enumerate(VALID_CLASS_IDS):
            y_true, y_score = [], []
            hard_fn, has_gt, has_pred = 0, False, False
            for m in matches:
                c = matches[m]['classes'][int(cls)]
                inter, vert, conf, uid = c['inter'], c['vert'], c['conf'], c['uid']
                union = c['gt_vert'][None, :] + vert[:, None] - inter
                iou = inter / np.maximum(union, 1)
                counted = (c['gt_id'] >= 1000) & (c['gt_vert'] >= MIN_REGION_SIZE)   # :133-134
                has_gt |= bool(counted.any())
                has_pred |= len(conf) > 0
                # greedy assignment in ground-truth order, predictions in prediction order (:143-172)
                for g in np.nonzero(counted)[0]:
                    best = None
                    for p in np.nonzero(inter[:, g] > 0)[0]:
                        if taken[m][uid[p]] or not iou[p, g] > th:
                            continue
                        if best is None:
                            best = conf[p]
                            taken[m][uid[p]] = True
                        else:               # a second prediction on a matched ground truth is a false positive
                            y_true.append(0); y_score.append(min(best, conf[p]))
                            best = max(best, conf[p])
                    if best is None:
                        hard_fn += 1
                    else:
                        y_true.append(1); y_score.append(best)
                # predictions that reach no ground truth of their class at this overlap (:178-199)
                for p in range(len(conf)):
                    hit = inter[p] > 0
                    if (iou[p][hit] > th).any():
                        continue
                    ignore = c['void'][p] + inter[p][hit & (c['gt_id'] < 1000)].sum() + \
                        inter[p][hit & (c['gt_vert'] < MIN_REGION_SIZE)].sum()
                    if float(ignore) / vert[p] <= th:
                        y_true.append(0); y_score.append(conf[p])
            if has_gt and has_pred:
                ap[0, li, oi], curves[th][CLASS_LABELS[li]] = _average_precision(
                    np.asarray(y_true, np.float64), np.asarray(y_score, np.float64), hard_fn)
            elif has_gt:
                ap[0, li, oi] = 0.0
            else:
                ap[0, li, oi] = float('nan')
    return ap, curves


def compute_averages(aps):
    """eval_metric.py:262-278."""
    o50, o25 = np.isclose(OVERLAPS, 0.5), np.isclose(OVERLAPS, 0.25)
    rest = ~o25
    avg = {'all_ap': np.nanmean(aps[0][:, rest]), 'all_ap_50%': np.nanmean(aps[0][:, o50]),
           'all_ap_25%': np.nanmean(aps[0][:, o25]), 'classes': {}}
    for li, name in enumerate(CLASS_LABELS):
        avg['classes'][name] = {'ap': np.average(aps[0, li, rest]), 'ap50%': np.average(aps[0, li, o50]),
                                'ap25%': np.average(aps[0, li, o25])}
    return avg


def compute_eval(results, gt_ids_by_scene):
    """results: {scene: pred_info}; gt_ids_by_scene: {scene: (n,) ids} (the reference reads
    data/scannet/gt_instance_data_txt/<scene>.txt, eval_metric.py:451-465).  Returns (averages, pr_curves)."""
    matches = {name: assign_instances_for_scan(name, results[name], gt_ids_by_scene[name]) for name in results}
    aps, curves = evaluate_matches(matches)
    return compute_averages(aps), curves
