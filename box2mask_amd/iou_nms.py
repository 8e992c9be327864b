"""Box-vote clustering and mask NMS with the function surface of /root/reference/models/iou_nms.py,
backed by the HIP kernels of csrc/nms.hip (b2m_nmc, b2m_mask_project, b2m_mask_nms, b2m_label_hist,
b2m_mask_gather, b2m_set_ious).  Inputs may live on the CPU (as in the reference, which runs this
stage on `.cpu()` tensors) or on the GPU; compute always happens on the GPU and results are returned
on the device of the input.  There is no CPU fallback.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib

_call = _lib.call


def _dev():
    _lib.require_gpu()
    return torch.device('cuda', torch.cuda.current_device())


def set_IOUs(boxes_a, boxes, check=True):
    """Row-wise IoU of two (n,6) [min,max] box sets (iou_nms.py:4-22)."""
    assert boxes_a.shape[1] == 6 and boxes.shape[1] == 6
    dev = _dev()
    a = boxes_a.detach().to(dev, torch.float32).contiguous()
    b = boxes.detach().to(dev, torch.float32).contiguous()
    if check:
        # iou_nms.py:9 asserts non-negative side lengths (one blocking host sync, as in the reference);
        # the training step passes check=False because its boxes are non-negative by construction
        assert bool(torch.all(a[:, 3:] - a[:, :3] >= 0)) and bool(torch.all(b[:, 3:] - b[:, :3] >= 0))
    out = torch.empty(a.shape[0], dtype=torch.float32, device=dev)
    _call('b2m_set_ious', a.data_ptr(), b.data_ptr(), a.shape[0], out.data_ptr())
    return out.to(boxes.device)


def torch_IOUs(box, boxes):
    """One (6,) box against (n,6) boxes (iou_nms.py:26-45)."""
    assert box.shape[0] == 6 and boxes.shape[1] == 6
    return set_IOUs(box.reshape(1, 6).expand(boxes.shape[0], 6), boxes)


class NMCResult:
    """Device-side result of the clustering kernel (kept on the GPU for detection2mask)."""
    __slots__ = ('k', 'reps', 'assign', 'heat', 'order', 'n')


def nmc_device(boxes_dev: torch.Tensor, cluster_th: float, max_k: int | None = None) -> NMCResult:
    n = boxes_dev.shape[0]
    dev = boxes_dev.device
    if n == 0:
        # the reference crashes on an empty scene (torch.stack([]), iou_nms.py:103)
        raise ValueError('NMS_clustering needs at least one box (the reference raises on n == 0 as well)')
    assert boxes_dev.shape[1] == 7 and boxes_dev.dim() == 2
    assert 0 < cluster_th < 1
    npad = 2
    while npad < n:
        npad <<= 1
    res = NMCResult()
    res.n = n
    res.reps = torch.empty(n, dtype=torch.int32, device=dev)
    res.assign = torch.empty(n, dtype=torch.int32, device=dev)
    res.order = torch.empty(npad, dtype=torch.int64, device=dev)
    kout = torch.zeros(1, dtype=torch.int32, device=dev)
    mk = min(n, 256) if max_k is None else max_k
    while True:
        heat = torch.empty((mk, n), dtype=torch.float32, device=dev)
        _call('b2m_nmc', boxes_dev.data_ptr(), n, float(cluster_th), mk, res.reps.data_ptr(), res.assign.data_ptr(),
              heat.data_ptr(), kout.data_ptr(), res.order.data_ptr())
        k = int(kout.item())
        if k <= mk:
            break
        mk = k                      # rare: more clusters than rows reserved -> rerun with the exact size
    res.k, res.heat = k, heat[:k]
    return res


def nmc_device_batch(boxes_list, cluster_th: float):
    """Cluster the boxes of every scene of a batch in ONE launch (one workgroup per scene) and read the cluster counts
    back once.  `boxes_list`: per scene an (n_s, 7) fp32 device tensor (n_s may be 0).  Returns a list of NMCResult
    (None for an empty scene), each identical to nmc_device() on that scene alone."""
    assert 0 < cluster_th < 1
    S = len(boxes_list)
    if S == 0:
        return []
    dev = boxes_list[0].device
    ns = [int(b.shape[0]) for b in boxes_list]
    desc = np.zeros((S, 6), np.int64)
    box_off = heat_off = order_off = 0
    for s, n in enumerate(ns):
        npad = 2
        while npad < n:
            npad <<= 1
        mk = min(n, 256)
        desc[s] = (box_off, n, npad, mk, heat_off, order_off)
        box_off += n; heat_off += mk * n; order_off += npad
    boxes = torch.cat([b.reshape(-1, 7) for b in boxes_list], 0).contiguous() if box_off else \
        torch.zeros((1, 7), dtype=torch.float32, device=dev)
    reps = torch.empty(max(box_off, 1), dtype=torch.int32, device=dev)
    assign = torch.empty(max(box_off, 1), dtype=torch.int32, device=dev)
    heat = torch.empty(max(heat_off, 1), dtype=torch.float32, device=dev)
    order = torch.empty(max(order_off, 1), dtype=torch.int64, device=dev)
    kout = torch.zeros(S, dtype=torch.int32, device=dev)
    desc_dev = torch.from_numpy(desc).to(dev)
    _call('b2m_nmc_batch', boxes.data_ptr(), desc_dev.data_ptr(), S, max(ns), float(cluster_th), reps.data_ptr(),
          assign.data_ptr(), heat.data_ptr(), kout.data_ptr(), order.data_ptr())
    ks = kout.cpu().tolist()                                    # the one host read of the stage
    out = []
    for s, n in enumerate(ns):
        if n == 0:
            out.append(None)
            continue
        bo, _, npad, mk, ho, oo = (int(v) for v in desc[s])
        if ks[s] > mk:                                          # rare: more clusters than heat rows reserved
            out.append(nmc_device(boxes[bo:bo + n], cluster_th, max_k=ks[s]))
            continue
        r = NMCResult()
        r.n, r.k = n, ks[s]
        r.reps, r.assign = reps[bo:bo + n], assign[bo:bo + n]
        r.order = order[oo:oo + npad]
        r.heat = heat[ho:ho + mk * n].reshape(mk, n)[:ks[s]]
        out.append(r)
    return out


def NMS_clustering(boxes, cluster_th=0.5, get_heatmaps=True):
    """Greedy non-maximum clustering (iou_nms.py:68-105).  Returns
    (representatives int64 (K,), clusters list[K] of index tensors, heatmaps (K,n) fp32)."""
    assert boxes.shape[1] == 7 and len(boxes.shape) == 2
    assert cluster_th > 0 and cluster_th < 1
    side = boxes[:, 4:] - boxes[:, 1:4]
    if not bool(torch.all(torch.min(side, axis=1)[0] > 0)):
        print('Warning: Invalid boxes found.')           # iou_nms.py:73-76 (warn only)
    dev = _dev()
    b = boxes.detach().to(dev, torch.float32).contiguous()
    r = nmc_device(b, cluster_th)
    reps = r.reps[:r.k].long()
    order = (r.order[:r.n] & 0xFFFFFFFF).long()
    assign_sorted = r.assign.long()[order]
    clusters = [order[assign_sorted == c].to(boxes.device) for c in range(r.k)]
    if get_heatmaps:
        return reps.to(boxes.device), clusters, r.heat.to(boxes.device)
    return reps.to(boxes.device), clusters


def pack_masks(masks_dev: torch.Tensor):
    """(K,N) bool on the GPU -> (K, ceil(N/64)) int64 bit rows (bit v of word v//64)."""
    K, N = masks_dev.shape
    words = (N + 63) // 64
    bits = torch.empty((K, max(words, 1)), dtype=torch.int64, device=masks_dev.device)
    if K and N:
        _call('b2m_mask_pack', masks_dev.data_ptr(), K, N, bits.data_ptr(), words)
    return bits, words


def mask_nms_device(bits: torch.Tensor, k: int, words: int, th: float):
    dev = bits.device
    inter = torch.empty(max(k * k, 1), dtype=torch.int32, device=dev)
    keep = torch.empty(max(k, 1), dtype=torch.int32, device=dev)
    nkeep = torch.zeros(1, dtype=torch.int32, device=dev)
    _call('b2m_mask_nms', bits.data_ptr(), k, words, float(th), inter.data_ptr(), keep.data_ptr(), nkeep.data_ptr())
    return keep[:k], inter


def masks_iou(mask, masks, allow_empty=False):
    """IoU of one bool mask against (K,N) bool masks (iou_nms.py:109-121)."""
    dev = _dev()
    m = torch.cat([mask.reshape(1, -1), masks], 0).to(dev).bool().contiguous()
    if not allow_empty:
        assert bool(torch.all(torch.sum(m, axis=1) > 0))
    bits, words = pack_masks(m)
    k = m.shape[0]
    _, inter = mask_nms_device(bits, k, words, 0.5)
    inter = inter[:k * k].reshape(k, k).long()
    cnt = torch.diagonal(inter)
    i = inter[0, 1:]
    u = cnt[0] + cnt[1:] - i
    if not allow_empty:
        return (i / u).to(masks.device)
    ret = torch.zeros_like(u).float()
    ret[u > 0] = i[u > 0] / u[u > 0]
    return ret.to(masks.device)


def mask_NMS(sorted_masks, cluster_th=0.5, allow_empty=False):
    """Greedy NMS over boolean masks in the given order (iou_nms.py:130-144).
    Returns (kept indices int64, suppressed list of (kept, suppressed-by-it) pairs)."""
    dev = _dev()
    m = sorted_masks.to(dev).bool().contiguous()
    k = m.shape[0]
    if not allow_empty:
        assert bool(torch.all(torch.sum(m, axis=1) > 0))
    bits, words = pack_masks(m)
    keep, inter = mask_nms_device(bits, k, words, cluster_th)
    kept = torch.nonzero(keep).reshape(-1)
    # suppression bookkeeping (second return value of the reference, unused by detection2mask)
    inter = inter[:k * k].reshape(k, k).cpu().long()
    cnt = torch.diagonal(inter)
    alive = torch.ones(k, dtype=torch.bool)
    suppressed = []
    for i in kept.cpu().tolist():
        un = cnt[i] + cnt - inter[i]
        iou = inter[i].float() / un.float()
        iou[i] = 1
        hit = alive & ~(iou <= cluster_th)
        hit[:i] = False
        suppressed.append((torch.tensor(i), torch.nonzero(hit).reshape(-1)))
        alive &= ~hit
    return kept.to(sorted_masks.device), suppressed


def semIOU(pred_label, gt_label):
    """Per-label IoU over valid (> -100) rows (iou_nms.py:146-157); logging only.  One device
    pass (bincount of label pairs) and a single host transfer instead of an `.item()` per label."""
    valid = gt_label > -100
    gt = gt_label[valid]
    pr = pred_label[valid]
    if gt.numel() == 0:
        return np.array([])
    allv = torch.cat((gt, pr))
    lo = int(allv.min())
    L = int(allv.max()) - lo + 1
    conf = torch.bincount((gt - lo) * L + (pr - lo), minlength=L * L).reshape(L, L)
    inter = torch.diagonal(conf)
    union = conf.sum(0) + conf.sum(1) - inter
    present = union > 0
    iou = (inter[present] / (union[present] + 1e-6)).float()
    return iou.cpu().numpy().astype(np.float64)
