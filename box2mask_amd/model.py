"""Model wrapper: the drop-in boundary.  Same constructor, methods, loss-dict keys and
state-dict layout as /root/reference/models/model.py:14-288, so models/training.py and
models/evaluation.py can drive it unchanged.  Losses are the reference formulas evaluated with torch
on the GPU; the forward/backward of the network goes through the HIP library.
"""
from __future__ import annotations

import os
from glob import glob

import torch

from . import _lib
from . import iou_nms
from . import nn as ME
from .detection_net import SelectionNet
from .iou_nms import semIOU
from .util import to_bbs_min_max_


def dist_world() -> int:
    import torch.distributed as dist
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def _pearsonr(a: torch.Tensor, b: torch.Tensor):
    """Pearson correlation on the device (scipy.stats.pearsonr at model.py:170,191 is logging only;
    computing it here removes two blocking D2H copies per step).  Returns a 0-dim tensor (`.item()` works)."""
    a = a.double() - a.double().mean()
    b = b.double() - b.double().mean()
    return (a * b).sum() / torch.sqrt((a * a).sum() * (b * b).sum()).clamp_min(1e-300)


def _paired_box_iou(a: torch.Tensor, b: torch.Tensor, floor: float = 1e-6):
    """IoU of box a[i] with box b[i]; rows are [min xyz, max xyz].  The IoU term of the training loss
    (model.py:111-126): volumes from the corner differences, intersection clamped at zero, union floored at 1e-6."""
    def volume(lo, hi):
        side = hi - lo
        return side[..., 0] * side[..., 1] * side[..., 2]
    lo, hi = torch.maximum(a[..., :3], b[..., :3]), torch.minimum(a[..., 3:], b[..., 3:])
    side = (hi - lo).clamp(min=0)
    inter = side[..., 0] * side[..., 1] * side[..., 2]
    union = volume(a[..., :3], a[..., 3:]) + volume(b[..., :3], b[..., 3:]) - inter
    return inter / torch.clamp(union, min=floor)


class Model:
    def __init__(self, cfg, semantic_valid_class_ids, semantic_id2idx, instance_id2idx, is_foreground, device='cuda'):
        _lib.require_gpu()                       # fails loudly without the HIP extension / a GPU
        self.cfg = cfg
        self.device = device
        self.semantic_valid_class_ids = semantic_valid_class_ids
        self.semantic_id2idx = semantic_id2idx
        self.instance_id2idx = instance_id2idx
        self.is_foreground = is_foreground
        self.detection_model = SelectionNet(cfg, device, semantic_valid_class_ids, is_foreground,
                                            out_channels=[96, 96, 6]).to(device)
        # cfg.half_inference (a build extension, BASELINE configs[4] "fp16 features on CDNA4"; absent in the reference's configs):
        # inference passes -- model.eval() + no gradients -- run the trunk on half activations (SelectionNet.half_trunk)
        self.detection_model.half_trunk = 'inference' if getattr(cfg, 'half_inference', False) else False
        self._dp = None
        # every parameter gradient is a view into one flat buffer (one memset per step, in-place all-reduce)
        from .grad_arena import GradArena
        self._arena = GradArena(list(self.detection_model.parameters()))
        self.detection_model._grad_arena = self._arena
        if cfg.multigpu:
            # model.py:24-25: DDP + SyncBN.  Here: packed SyncBN statistics and a flat-bucket RCCL
            # gradient all-reduce (box2mask_amd/parallel.py) instead of torch DDP's hook machinery.
            from .parallel import GradAllReduce
            ME.MinkowskiSyncBatchNorm.convert_sync_batchnorm(self.detection_model)
            self._dp = GradAllReduce(list(self.detection_model.parameters()), arena=self._arena,
                                     buffers=list(self.detection_model.buffers()))
            self._dp.broadcast_parameters()
            # B2M_SYNCBN_IPC=1 (opt-in, one node): the SyncBN statistics go through mailboxes mapped between the ranks instead of
            # through the collective library -- one one-workgroup launch per exchange (parallel.IpcExchange)
            from . import parallel, functional as F_
            if parallel.syncbn_ipc_enabled() and dist_world() > 1 and F_.ipc_exchange is None:
                F_.ipc_exchange = parallel.IpcExchange(device=torch.device(device) if device is not None else None)
        self.BCEWithLogitsLoss = torch.nn.BCEWithLogitsLoss().to(device)
        self.semantics_loss = torch.nn.CrossEntropyLoss(ignore_index=-100).to(device)
        self._id2idx_dev = None
        self._pf_stream = None
        self._prefetched = None

    def compute_loss(self, batch, epoch):
        losses_dict, pred = self.compute_loss_detection(batch, epoch)
        return losses_dict

    # ---- next batch's sparse tensor ahead of time (optional; the reference's loop does not need to call it)
    @staticmethod
    def _batch_key(batch):
        c, f = batch['vox_coords'], batch['vox_features']
        return (id(c), c.data_ptr(), tuple(c.shape), id(f), f.data_ptr())

    def prefetch(self, batch, ready=None, loss_rows=True):
        """Build the ME.SparseTensor of `batch` -- Morton order, coordinate hash, the 7 strided coordinate maps, the 16
        kernel maps and their rulebooks -- NOW, on a second stream, and hand it to the next `compute_loss(batch, ...)`.
        Called right after `optimizer.step()` for the batch the data loader already holds, the ~2 ms of map kernels and
        the dozen host reads of their row counts run beside the current step's backward pass instead of in front of the
        next step's first convolution (the host reads wait for the side stream only).  `vox_coords` / `vox_features`
        may be (pinned) host tensors: the copy runs on the side stream as well.  Never required: a batch that was not
        prefetched is built inside compute_loss as before; results are identical.

        `ready` says when the tensors of `batch` are valid on the device: None = after everything enqueued on the current
        stream so far (always safe, but the side stream then starts only when the current step has finished -- only the
        host reads move); a torch.cuda.Event recorded after they were written; True = they are complete already (host
        tensors, or device tensors made before the current step was enqueued) -- the maps are then built WHILE the
        current step runs.
        Inference (evaluation.py:70-98): the same call one scene ahead of Model.get_prediction -- `loss_rows=False` skips the
        foreground row list only the loss terms read."""
        if self._pf_stream is None:
            self._pf_stream = torch.cuda.Stream(device=self.device)
        side = self._pf_stream
        if ready is None:
            side.wait_stream(torch.cuda.current_stream())
        elif ready is not True:
            side.wait_event(ready)
        with torch.cuda.stream(side):
            sin = ME.SparseTensor(batch['vox_features'], batch['vox_coords'], device=self.device)
            if sin.manager is not None:
                sin.manager.prefetch(8, same=[(0, 5)] + [(l, 3) for l in range(8)], strided=True)
            fg_rows = None                                     # the loss terms' foreground row list (a host read too)
            if loss_rows and 'fg_instances' in batch and (self.cfg.loss_on_fg_instances or self.cfg.bb_supervision):
                fg_rows = torch.nonzero(batch['fg_instances'].to(self.device)).reshape(-1)
        self._prefetched = (self._batch_key(batch), sin, fg_rows)

    def _take_prefetched(self, batch):
        pf = self._prefetched
        self._prefetched = None
        if pf is None or pf[0] != self._batch_key(batch):
            return None, None
        sin, fg_rows = pf[1], pf[2]
        main = torch.cuda.current_stream()
        main.wait_stream(self._pf_stream)
        # built on the side stream, used (and later freed) while kernels of THIS stream read it
        sin.F.record_stream(main)
        for t in sin.manager.tensors():
            t.record_stream(main)
        if fg_rows is not None:
            fg_rows.record_stream(main)
        return sin, fg_rows

    def _sem_lut(self):
        if self._id2idx_dev is None:
            self._id2idx_dev = self.semantic_id2idx.to(self.device)
        return self._id2idx_dev

    def compute_loss_detection(self, batch, epoch):
        """model.py:38-225: same loss terms, keys and weights."""
        device = self.device
        cfg = self.cfg
        on_fg = cfg.loss_on_fg_instances or cfg.bb_supervision
        # The reference selects the foreground rows with a boolean mask in every loss term (model.py:65-66 ...): each such
        # indexing is a host read of the row count.  Here the row list is made ONCE, BEFORE the forward pass is enqueued (the
        # host still has nothing to wait for), and every term gathers with it: same rows, same order, no host read while
        # the device works through the network -- the backward pass is enqueued behind the forward without a stall.
        sin, fg_rows = self._take_prefetched(batch)            # Model.prefetch(batch) ran: both were made ahead of time
        fg = batch['fg_instances'].to(device) if 'fg_instances' in batch else None
        if fg is not None and on_fg and fg_rows is None:
            fg_rows = torch.nonzero(fg).reshape(-1)
        if sin is None:
            sin = ME.SparseTensor(batch['vox_features'], batch['vox_coords'], device=device)
        n_seg = batch['input_location'].shape[0] if cfg.do_segment_pooling else None
        pred = self.detection_model(sin, batch['pooling_ids'].to(device), n_seg)
        for mlp_head, sparse_tensor in pred.items():
            pred[mlp_head] = sparse_tensor.F
        losses_dict = {'optimization_loss': 0}

        fused = self._fused_losses(batch, pred, epoch, fg, fg_rows, on_fg)
        if fused is not None:
            return fused, pred

        if cfg.mlp_offsets in cfg.network_heads:                     # model.py:62-73
            gt_offsets, pred_offsets = batch['gt_bb_offsets'].to(device), pred[cfg.mlp_offsets]
            if on_fg:
                pred_offsets, gt_offsets = pred_offsets[fg_rows], gt_offsets[fg_rows]
            offset_loss_per_pred = torch.sum(torch.abs(pred_offsets - gt_offsets), axis=1)
            offset_loss = torch.mean(offset_loss_per_pred)
            losses_dict['optimization_loss'] += cfg.loss_weight_bb_offsets * offset_loss
            losses_dict['offset_loss'] = offset_loss.detach()

        if cfg.mlp_bounds in cfg.network_heads:                      # model.py:76-88
            gt_bounds, pred_bounds = batch['gt_bb_bounds'].to(device), pred[cfg.mlp_bounds]
            if on_fg:
                pred_bounds, gt_bounds = pred_bounds[fg_rows], gt_bounds[fg_rows]
            bounds_loss = torch.mean(torch.sum(torch.abs(pred_bounds - gt_bounds), axis=1))
            losses_dict['optimization_loss'] += cfg.loss_weight_bb_bounds * bounds_loss
            losses_dict['bounds_loss'] = bounds_loss.detach()

        if cfg.use_bb_iou_loss:                                      # model.py:91-129
            pred_bounds, pred_offsets = pred[cfg.mlp_bounds], pred[cfg.mlp_offsets]
            loc = batch['input_location'].to(device)
            gt_offsets, gt_bounds = batch['gt_bb_offsets'].to(device), batch['gt_bb_bounds'].to(device)
            if on_fg:
                pred_bounds, pred_offsets = pred_bounds[fg_rows], pred_offsets[fg_rows]
                gt_bounds, gt_offsets, loc = gt_bounds[fg_rows], gt_offsets[fg_rows], loc[fg_rows]
            pred_bounds = torch.clamp(pred_bounds, min=cfg.min_bb_size)
            ious = _paired_box_iou(to_bbs_min_max_(pred_offsets + loc, pred_bounds, device),
                                   to_bbs_min_max_(gt_offsets + loc, gt_bounds, device))
            iou_loss = torch.mean(1.0 - ious)
            losses_dict['optimization_loss'] += cfg.loss_weight_bb_iou * iou_loss
            losses_dict['iou_loss'] = iou_loss.detach()

        if cfg.mlp_bb_scores in cfg.network_heads:                   # model.py:133-176
            loss_weight_bb_scores = cfg.loss_weight_bb_scores
            if epoch < cfg.mlp_bb_scores_start_epoch:
                loss_weight_bb_scores = 0
            pred_scores = pred[cfg.mlp_bb_scores].reshape(-1)
            pred_bounds, pred_offsets = pred[cfg.mlp_bounds], pred[cfg.mlp_offsets]
            loc = batch['input_location'].to(device)
            gt_offsets, gt_bounds = batch['gt_bb_offsets'].to(device), batch['gt_bb_bounds'].to(device)
            if on_fg:
                pred_scores, pred_bounds, pred_offsets = pred_scores[fg_rows], pred_bounds[fg_rows], pred_offsets[fg_rows]
                loc, gt_offsets, gt_bounds = loc[fg_rows], gt_offsets[fg_rows], gt_bounds[fg_rows]
            gt_bbs = to_bbs_min_max_(gt_offsets + loc, gt_bounds, device)
            pred_bounds = torch.clamp(pred_bounds, min=cfg.min_bb_size)
            pred_bbs = to_bbs_min_max_(pred_offsets + loc, pred_bounds, device)
            ious = iou_nms.set_IOUs(gt_bbs, pred_bbs, check=False).detach()   # sides >= 0 by construction
            score_loss = self.BCEWithLogitsLoss(pred_scores, ious)
            losses_dict['bb_scores_correlation'] = _pearsonr(ious, pred_scores.detach())
            losses_dict['optimization_loss'] += loss_weight_bb_scores * score_loss
            losses_dict['bb_score_loss'] = score_loss.detach()
            losses_dict['bb_target_scores'] = torch.mean(ious)

        if cfg.mlp_center_scores in cfg.network_heads and epoch >= cfg.mlp_center_scores_start_epoch:   # 179-192
            pred_scores = pred[cfg.mlp_center_scores].reshape(-1)
            gt_scores = offset_loss_per_pred.detach()
            if cfg.loss_on_fg_instances:
                pred_scores = pred_scores[fg_rows]
            score_loss = torch.mean(torch.abs(pred_scores - gt_scores))
            losses_dict['optimization_loss'] += cfg.loss_weight_center_scores * score_loss
            losses_dict['center_score_loss'] = score_loss.detach()
            losses_dict['center_scores_correlation'] = _pearsonr(gt_scores, pred_scores.detach())

        if cfg.mlp_semantics in cfg.network_heads:                   # model.py:194-210
            pred_semantics = pred[cfg.mlp_semantics]
            gt_semantics = self._sem_lut()[batch['gt_semantics'].to(device)]
            semantics_loss = self.semantics_loss(pred_semantics, gt_semantics)
            pred_semantics_int = torch.argmax(pred_semantics, 1)
            semantics_acc = torch.sum(pred_semantics_int == gt_semantics) / len(gt_semantics)
            losses_dict['optimization_loss'] += cfg.loss_weight_semantics * semantics_loss
            losses_dict['semantics_loss'] = semantics_loss.detach()
            losses_dict['semantics_acc'] = semantics_acc.detach()
            losses_dict['semantics_mIoU'] = _LazyMean(pred_semantics_int, gt_semantics)

        if cfg.mlp_per_vox_semantics in cfg.network_heads:           # model.py:212-223
            pred_semantics = pred[cfg.mlp_per_vox_semantics]
            gt_semantics = self._sem_lut()[batch['gt_per_vox_semantics'].to(device)]
            per_vox_semantics_loss = self.semantics_loss(pred_semantics, gt_semantics)
            pred_semantics_int = torch.argmax(pred_semantics, 1)
            per_vox_semantics_acc = torch.sum(pred_semantics_int == gt_semantics) / len(gt_semantics)
            losses_dict['optimization_loss'] += cfg.loss_weight_per_vox_semantics * per_vox_semantics_loss
            losses_dict['per_vox_semantics_loss'] = per_vox_semantics_loss.detach()
            losses_dict['per_vox_semantics_acc'] = per_vox_semantics_acc.detach()
        return losses_dict, pred

    def _fused_losses(self, batch, pred, epoch, fg, fg_rows, on_fg):
        """The ScanNet loss configuration (offsets + bounds [+ IoU-target scores] [+ semantics] on the foreground segments)
        as ONE launch for values and gradients (functional.detection_loss) instead of ~150 elementwise torch launches between
        the forward and the backward pass; None for every other configuration (IoU loss, centre scores, per-voxel semantics:
        the term-by-term code below).  Same keys, same values (tests/test_gpu_net.py::test_losses_match_reference_golden)."""
        from . import functional as F_
        cfg, device = self.cfg, self.device
        heads = cfg.network_heads
        if not F_.fused_loss_enabled() or cfg.use_bb_iou_loss or cfg.mlp_center_scores in heads or \
                cfg.mlp_per_vox_semantics in heads or cfg.mlp_offsets not in heads or cfg.mlp_bounds not in heads:
            return None
        if on_fg and (fg is None or fg_rows is None):
            return None
        S = pred[cfg.mlp_offsets].shape[0]
        n_fg = int(fg_rows.shape[0]) if on_fg else S
        if S == 0 or n_fg == 0:
            return None
        f32 = lambda t: t.to(device=device, dtype=torch.float32).contiguous()
        has_sc, has_sem = cfg.mlp_bb_scores in heads, cfg.mlp_semantics in heads
        w_sc = 0.0
        if has_sc:
            w_sc = float(cfg.loss_weight_bb_scores) if epoch >= cfg.mlp_bb_scores_start_epoch else 0.0
        gt_sem = n_valid = None
        if has_sem:
            gt_sem = self._sem_lut()[batch['gt_semantics'].to(device)].contiguous()
            n_valid = (gt_sem >= 0).sum(dtype=torch.float64).reshape(1)
        res, argmax = F_.detection_loss(
            pred[cfg.mlp_offsets], pred[cfg.mlp_bounds], pred[cfg.mlp_bb_scores] if has_sc else None,
            pred[cfg.mlp_semantics] if has_sem else None, f32(batch['gt_bb_offsets']), f32(batch['gt_bb_bounds']),
            f32(batch['input_location']), fg.to(torch.uint8).contiguous() if on_fg else None, gt_sem, n_fg, n_valid,
            (float(cfg.loss_weight_bb_offsets), float(cfg.loss_weight_bb_bounds), w_sc, float(cfg.loss_weight_semantics)),
            float(cfg.min_bb_size))
        d = res.detach()
        out = {'optimization_loss': res[0], 'offset_loss': d[1], 'bounds_loss': d[2]}
        if has_sc:
            out['bb_scores_correlation'] = d[5]
            out['bb_score_loss'] = d[3]
            out['bb_target_scores'] = d[4]
        if has_sem:
            out['semantics_loss'] = d[6]
            out['semantics_acc'] = d[7]
            out['semantics_mIoU'] = _LazyMean(argmax, gt_sem)
        return out

    def get_prediction(self, batch, with_grad=False, to_cpu=True, min_size=True, get_all=False):
        # (an evaluation loop that called Model.prefetch(batch) one scene ahead: the maps are there already)
        # (one made for ANOTHER batch -- the training loop's next step, with a validation pass in between -- stays where it is)
        mine = self._prefetched is not None and self._prefetched[0] == self._batch_key(batch)
        sin, _ = self._take_prefetched(batch) if mine else (None, None)
        return self.detection_model.get_prediction(batch, with_grad=with_grad, to_cpu=to_cpu, min_size=min_size, sin=sin)

    def pred2mask(self, batch, pred, mode):
        return self.detection_model.detection2mask(batch, pred, self.cfg, mode, True, *self.cfg.eval_ths)

    def parameters(self):
        return self.detection_model.parameters()

    def to(self, device):
        self.detection_model = self.detection_model.to(device)
        return self

    def eval(self):
        self.detection_model.eval()

    def train(self):
        self.detection_model.train()

    def load_state_dict(self, state_dict, strict=True):
        return self.detection_model.load_state_dict(state_dict, strict)

    def state_dict(self):
        return self.detection_model.state_dict()

    def load_checkpoint(self, checkpoint=None, closest_to=None):
        """Restore a checkpoint written by training.py:216-224 (`checkpoint_{h}h:{m}m:{s}s_{seconds}.tar` under
        cfg.checkpoint_path).  Without a name the newest one is taken -- or, with `closest_to` (hours), the one
        whose training time is nearest.  Returns (epoch, training_time, name, iteration_num); (0, 0) when the
        directory is empty, as model.py:264-288 does."""
        root = self.cfg.checkpoint_path
        if checkpoint is not None:
            path = '%s%s.tar' % (root, checkpoint)
        else:
            stamped = []                                  # (training seconds, path)
            for f in glob(root + '/*'):
                stem = os.path.splitext(os.path.basename(f))[0]
                try:
                    stamped.append((float(stem.rsplit('_', 1)[-1]), f))
                except ValueError:
                    raise ValueError('unexpected file in the checkpoint directory: %s' % f)
            if not stamped:
                print('No checkpoints found at {}'.format(root))
                return 0, 0
            stamped.sort(key=lambda e: e[0])
            if closest_to:
                want = closest_to * 3600.0
                path = min(stamped, key=lambda e: abs(e[0] - want))[1]
            else:
                path = stamped[-1][1]
        print('Loaded checkpoint from: {}'.format(path))
        state = torch.load(path, map_location=self.device)
        self.load_state_dict(state['model_state_dict'])
        name = os.path.basename(path)[:-len('.tar')]
        return state['epoch'], state['training_time'], name, state['iteration_num']


class _LazyMean:
    """semantics_mIoU is a logging value (model.py:205,210); evaluating it lazily keeps the blocking
    host transfer out of the training step unless the caller actually asks for `.item()`."""

    def __init__(self, pred, gt):
        self._p, self._g, self._v = pred.detach(), gt, None

    def item(self):
        if self._v is None:
            v = semIOU(self._p, self._g)
            self._v = float(v.mean()) if v.size else float('nan')
        return self._v

    def __float__(self):
        return self.item()
