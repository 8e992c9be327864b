"""CPU check of the AP evaluation's device half -- TEST INFRASTRUCTURE ONLY (tests/ and bench.py's cpu_baseline leg).

`intersections` restates the counting of /root/reference/utils/eval_metric.py:281-330 (one count_nonzero over all
points per prediction x ground-truth pair).  The matching / AP arithmetic itself (eval_metric.py:102-278) is pinned
through tests/golden/eval_metric.npz, which holds AP tables computed by the real reference functions
(tools/gen_golden.py eval)."""
import numpy as np


def intersections(pred_masks, gt_ids):
    masks = np.not_equal(np.asarray(pred_masks), 0)                      # :306
    gt_ids = np.asarray(gt_ids)
    uniq = np.unique(gt_ids)
    vert = np.array([(gt_ids == u).sum() for u in uniq], np.int64)       # Instance.get_instance_verts, :55
    inter = np.zeros((len(masks), len(uniq)), np.int64)
    for k, m in enumerate(masks):
        for g, u in enumerate(uniq):
            inter[k, g] = np.count_nonzero(np.logical_and(gt_ids == u, m))   # :322
    return uniq, vert, inter
