"""SelectionNet: the 8-level sparse residual U-Net, segment pooling, MLP heads and the
votes -> instance-mask post-processing, with the class surface and parameter names of
/root/reference/models/detection_net.py (network_initialization 34-230, forward 234-364,
detection2mask 369-488, get_prediction 493-521).  All sparse arithmetic runs in the HIP library.
"""
from __future__ import annotations

import os

import numpy as np
import torch
from torch import nn

from . import _lib
from . import iou_nms
from . import functional as F_
from . import nn as ME
from .resnet import BasicBlock, ResNetBase
from .util import to_bbs_min_max

_call = _lib.call


class SelectionNet(ResNetBase):
    BLOCK = BasicBlock
    DILATIONS = (1, 1, 1, 1, 1, 1, 1, 1)
    PLANES = (32, 64, 128, 256, 256, 128, 96, 96)
    added_PLANES = (256, 256, 256, 256, 256, 256)
    INIT_DIM = 32
    OUT_TENSOR_STRIDE = 1

    def __init__(self, cfg, device, semantic_valid_class_ids, is_foreground, out_channels=[96, 96, 3], D=3,
                 mlp_head=False):
        self.device = device
        self.cfg = cfg
        self.mlp_head = mlp_head
        self.LAYERS = (cfg.layers,) * 8
        self.added_LAYERS = (cfg.layers,) * 6
        self.semantic_valid_class_ids = semantic_valid_class_ids
        self.is_foreground = is_foreground
        ResNetBase.__init__(self, cfg.in_channels, out_channels, D)

    def network_initialization(self, in_channels, out_channels=[96, 96, 3], D=None, mlp_head=False):
        P, A = self.PLANES, self.added_PLANES
        self.inplanes = self.INIT_DIM
        self.conv0p1s1 = ME.MinkowskiConvolution(in_channels, self.inplanes, kernel_size=5, dimension=D)
        self.bn0 = ME.MinkowskiBatchNorm(self.inplanes)

        def down(name_conv, name_bn, name_block, planes, layers):
            setattr(self, name_conv, ME.MinkowskiConvolution(self.inplanes, self.inplanes, kernel_size=2, stride=2,
                                                             dimension=D))
            setattr(self, name_bn, ME.MinkowskiBatchNorm(self.inplanes))
            setattr(self, name_block, self._make_layer(self.BLOCK, planes, layers))

        down('conv1p1s2', 'bn1', 'block1', P[0], self.LAYERS[0])
        down('conv2p2s2', 'bn2', 'block2', P[1], self.LAYERS[1])
        down('conv3p4s2', 'bn3', 'block3', P[2], self.LAYERS[2])
        down('conv4p8s2', 'bn4', 'block4', P[3], self.LAYERS[3])
        down('added_conv1p16s2', 'added_bn1', 'added_block1', A[0], self.added_LAYERS[0])
        down('added_conv2p32s2', 'added_bn2', 'added_block2', A[1], self.added_LAYERS[1])
        down('added_conv3p64s2', 'added_bn3', 'added_block3', A[2], self.added_LAYERS[2])

        def up(name_conv, name_bn, name_block, planes, skip_planes, layers):
            setattr(self, name_conv, ME.MinkowskiConvolutionTranspose(self.inplanes, planes, kernel_size=2, stride=2,
                                                                      dimension=D))
            setattr(self, name_bn, ME.MinkowskiBatchNorm(planes))
            self.inplanes = planes + skip_planes
            setattr(self, name_block, self._make_layer(self.BLOCK, planes, layers))

        E = self.BLOCK.expansion
        up('added_convtr4p128s2', 'added_bntr4', 'added_block4', A[3], A[1] * E, self.added_LAYERS[3])
        up('added_convtr5p64s2', 'added_bntr5', 'added_block5', A[4], A[0] * E, self.added_LAYERS[4])
        up('added_convtr6p32s2', 'added_bntr6', 'added_block6', A[5], P[3] * E, self.added_LAYERS[5])
        up('convtr4p16s2', 'bntr4', 'block5', P[4], P[2] * E, self.LAYERS[4])
        up('convtr5p8s2', 'bntr5', 'block6', P[5], P[1] * E, self.LAYERS[5])
        up('convtr6p4s2', 'bntr6', 'block7', P[6], P[0] * E, self.LAYERS[6])
        up('convtr7p2s2', 'bntr7', 'block8', P[7], self.INIT_DIM, self.LAYERS[7])

        if self.cfg.load_unused_head:           # detection_net.py:142-164 (old checkpoints only)
            self.final0 = ME.MinkowskiConvolution(P[7] * E, out_channels[0], kernel_size=1, bias=True, dimension=D)
            self.final0_bn = ME.MinkowskiBatchNorm(out_channels[0])
            self.final1 = ME.MinkowskiConvolution(out_channels[0], out_channels[1], kernel_size=1, bias=True,
                                                  dimension=D)
            self.final1_bn = ME.MinkowskiBatchNorm(out_channels[1])
            self.final2 = ME.MinkowskiConvolution(out_channels[1], out_channels[2], kernel_size=1, bias=True,
                                                  dimension=D)
        self.relu = ME.MinkowskiReLU()

        def mlp_head(output_dim):               # detection_net.py:170-194: conv1x1-ReLU-BN x2, conv1x1
            return nn.Sequential(
                ME.MinkowskiConvolution(P[7] * E, out_channels[0], kernel_size=1, bias=True, dimension=D),
                ME.MinkowskiReLU(),
                ME.MinkowskiBatchNorm(out_channels[0]),
                ME.MinkowskiConvolution(out_channels[0], out_channels[1], kernel_size=1, bias=True, dimension=D),
                ME.MinkowskiReLU(),
                ME.MinkowskiBatchNorm(out_channels[1]),
                ME.MinkowskiConvolution(out_channels[1], output_dim, kernel_size=1, bias=True, dimension=D))

        cfg = self.cfg
        self.network_heads = {}
        self.requires_voxel_outputs = False
        for network_head in cfg.network_heads:
            if network_head == cfg.mlp_offsets:
                self.mlp_offsets = mlp_head(3)
                self.network_heads[network_head] = self.mlp_offsets
            if network_head == cfg.mlp_bounds:
                self.mlp_bounds = mlp_head(3)
                self.network_heads[network_head] = self.mlp_bounds
            if network_head == cfg.mlp_bb_scores:
                self.mlp_score = mlp_head(1)    # the reference builds it twice (208-209); only RNG-stream parity
                self.mlp_score = mlp_head(1)
                self.network_heads[network_head] = self.mlp_score
            if network_head == cfg.mlp_center_scores:
                self.mlp_center_score = mlp_head(1)
                self.network_heads[network_head] = self.mlp_center_score
            if network_head == cfg.mlp_semantics:
                self.mlp_semantics = mlp_head(len(self.semantic_valid_class_ids))
                self.network_heads[network_head] = self.mlp_semantics
            if network_head == cfg.mlp_per_vox_semantics:
                self.mlp_per_vox_semantics = mlp_head(len(self.semantic_valid_class_ids))
                self.network_heads[network_head] = self.mlp_per_vox_semantics
                self.requires_voxel_outputs = True

    # conv -> BN -> ReLU with the BN+ReLU epilogue fused into one launch
    @staticmethod
    def _cbr(conv, bn, x, skip=False):
        """conv -> BN -> ReLU.  skip: x has a second consumer later in the network (the decoder's ME.cat); returns
        (result, x') with x' the alias that consumer must take (ME.MinkowskiConvolution.forward(passthrough=True))."""
        if bn.fusable():                    # inference: one launch (ME.MinkowskiConvolution.forward(fuse=...))
            out = conv(x, fuse=(bn, None, True))
            return (out, x) if skip else out
        if skip:
            out, x = conv(x, passthrough=True)
        else:
            out = conv(x)
        out = out.new(bn.apply_bn(out.F, relu=True, count_key=ME.count_key_of(out), defer_counter=True))
        return (out, x) if skip else out

    def forward(self, x, pooling_ids=None, n_segments=None):
        """x: SparseTensor at tensor stride 1 -> dict head-name -> tensor holder with `.F`
        (detection_net.py:234-364)."""
        cbr = self._cbr
        tr = getattr(self, '_trace', None)          # optional dict: named intermediates for parity debugging
        F_.join_side_streams()                      # (a backward pass that died half-way leaves its side stream un-joined)
        # one launch repacks every layer's weight images for this pass (an inference pass keeps the previous ones if no
        # training pass ran in between)
        F_.packed_weights.begin_pass(inference=not (self.training or torch.is_grad_enabled()))
        F_.zero_slab.new_pass()                     # the tiny maps' zero-filled outputs: a fresh chunk per pass
        ME.defer_counters(True)                     # every BatchNorm's batch counter in ONE foreach add at the end of the pass
        arena = getattr(self, '_grad_arena', None)
        if arena is not None and torch.is_grad_enabled():
            arena.begin_pass()                      # one memset: every parameter gradient of this pass starts at zero
        if self.training or torch.is_grad_enabled():
            F_.note_training_pass()                 # an optimizer step may follow: what inference caches (half weight images, eval-mode
                                                    # BatchNorm affine maps) goes stale

        def T(name, t):
            if tr is not None:
                tr[name] = t.F
            return t

        # all 8 coordinate maps and every kernel map of the U-Net now, before the first convolution is enqueued (their row
        # counts are host reads: see CoordinateManager.prefetch)
        mgr = getattr(x, 'manager', None)
        if mgr is not None:
            mgr.prefetch(8, same=[(0, 5)] + [(l, 3) for l in range(8)], strided=True)
        out_p1 = T('out_p1', cbr(self.conv0p1s1, self.bn0, x))
        # Half trunk (inference only; BASELINE configs[4] "fp16 features on CDNA4"): behind the 6-channel stem every trunk
        # activation lives in HBM as IEEE half and every convolution (+ BatchNorm + residual + ReLU) is one
        # conv_fwd_flow_kernel<.., F16> launch -- half operands, f16 MFMA, fp32 accumulation and epilogue; the pooled
        # features and the heads are fp32 again.
        # half_trunk: True (every pass must be an inference pass) | 'inference' (cfg.half_inference: inference passes run in half,
        # training passes in fp32) | False
        ht = getattr(self, 'half_trunk', False)
        half = bool(ht) and (ht != 'inference' or self.bn0.fusable())
        # Half-precision TRAINING (half_train.py; `half_training = True` or cfg.half_training): the same region -- behind the stem,
        # in front of the pooling -- in binary16 with gradients, loss-scaled; fp32 master weights.  Training passes only.
        half_train = (bool(getattr(self, 'half_training', False) or getattr(self.cfg, 'half_training', False)) and self.training and
                      torch.is_grad_enabled() and not half)
        if half_train:
            from . import half_train as HT
            HT.loss_scale[0] = float(getattr(self.cfg, 'half_loss_scale', 1024.0))
            HT.images.begin_pass()                  # every half weight image of the step in one launch, beside the stem
            out_p1 = out_p1.new(HT.to_half(out_p1.F))
        if half:
            if not self.bn0.fusable():
                raise RuntimeError('half_trunk is an inference mode: model.eval() and torch.no_grad() (and B2M_CONV_AFFINE=1)')
            out_p1 = out_p1.new(out_p1.F.half())
        # every encoder output feeds the next strided convolution AND the decoder's ME.cat: the decoder takes the alias the
        # strided convolution hands back (skip=True), so the two gradients are summed inside that convolution's
        # data-gradient kernel
        down, out_p1 = cbr(self.conv1p1s2, self.bn1, out_p1, skip=True)
        out_b1p2 = T('block1', self.block1(T('down1', down)))
        down, out_b1p2 = cbr(self.conv2p2s2, self.bn2, out_b1p2, skip=True)
        out_b2p4 = T('block2', self.block2(T('down2', down)))
        down, out_b2p4 = cbr(self.conv3p4s2, self.bn3, out_b2p4, skip=True)
        out_b3p8 = T('block3', self.block3(T('down3', down)))
        if half_train:
            # Half training keeps the levels below tensor stride 8 in fp32: a few thousand rows at most -- no bytes to save, and the
            # fp32 forms of the small maps are the fused ones (one-launch BatchNorm, paired block ends, tile statistics).  (Not for
            # accuracy: those levels' train-mode BatchNorms over a few dozen rows amplify the rounding that ENTERS them 15 x
            # whichever precision they run in, tools/debug_half_train.py.)  The gradient leaves the loss-scaled half region at
            # `to_half` and re-enters it at `to_float`, as at the region's ends.
            down = cbr(self.conv4p8s2, self.bn4, out_b3p8.new(HT.to_float(out_b3p8.F)))
        else:
            down, out_b3p8 = cbr(self.conv4p8s2, self.bn4, out_b3p8, skip=True)
        out_b4p16 = T('block4', self.block4(T('down4', down)))
        down, out_b4p16 = cbr(self.added_conv1p16s2, self.added_bn1, out_b4p16, skip=True)
        out_added_b1p32 = T('added_block1', self.added_block1(T('down5', down)))
        down, out_added_b1p32 = cbr(self.added_conv2p32s2, self.added_bn2, out_added_b1p32, skip=True)
        out_added_b2p64 = T('added_block2', self.added_block2(T('down6', down)))
        down, out_added_b2p64 = cbr(self.added_conv3p64s2, self.added_bn3, out_added_b2p64, skip=True)
        out = T('added_block3', self.added_block3(T('down7', down)))

        out = T('added_block4', self.added_block4(ME.cat(T('up6', cbr(self.added_convtr4p128s2, self.added_bntr4, out)), out_added_b2p64)))
        out = T('added_block5', self.added_block5(ME.cat(T('up5', cbr(self.added_convtr5p64s2, self.added_bntr5, out)), out_added_b1p32)))
        out = T('added_block6', self.added_block6(ME.cat(T('up4', cbr(self.added_convtr6p32s2, self.added_bntr6, out)), out_b4p16)))
        up3 = cbr(self.convtr4p16s2, self.bntr4, out)
        if half_train:
            up3 = up3.new(HT.to_half(up3.F))        # (back into the half region at tensor stride 8)
        out = T('block5', self.block5(ME.cat(T('up3', up3), out_b3p8)))
        out = T('block6', self.block6(ME.cat(T('up2', cbr(self.convtr5p8s2, self.bntr5, out)), out_b2p4)))
        out = T('block7', self.block7(ME.cat(T('up1', cbr(self.convtr6p4s2, self.bntr6, out)), out_b1p2)))
        out = T('block8', self.block8(ME.cat(T('up0', cbr(self.convtr7p2s2, self.bntr7, out)), out_p1)))

        if half:
            out = out.new(out.F.float())
        if half_train:
            out = out.new(HT.to_float(out.F))
        outputs = {}
        perm = x.manager.perm if getattr(x, 'manager', None) is not None else None
        if self.requires_voxel_outputs:
            outputs['vox_feats'] = out
        if self.cfg.do_segment_pooling:
            assert pooling_ids is not None
            if perm is not None:                      # pooling ids arrive in input order, rows are in spatial order
                pooling_ids = pooling_ids.to(perm.device)[perm]
            mode = 'max' if self.cfg.max_pool_segments_detection_net else 'avg'
            if n_segments is None:
                n_segments = int(pooling_ids.max().item()) + 1
            out = ME.PooledTensor(F_.segment_pool(out.F, pooling_ids, n_segments, mode))
        # The heads that read the pooled features are independent of one another and layer-for-layer alike (conv1x1-ReLU-BN x2,
        # conv1x1): they run in lockstep, so that under SyncBN the BatchNorms at equal depth share one statistics exchange per
        # direction (ME.batch_norm_group)
        shared = [h for h in self.cfg.network_heads if not (self.requires_voxel_outputs and 'per_vox' in h)]
        seqs = [self.network_heads[h] for h in shared]
        lockstep = self.training and F_._sync_group() is not None        # (one process: head by head, the reference's order)
        if lockstep and len(seqs) > 1 and len({len(sq) for sq in seqs}) == 1:
            ts = [out] * len(seqs)
            for stage in range(len(seqs[0])):
                layers = [sq[stage] for sq in seqs]
                if all(isinstance(l, ME.MinkowskiBatchNorm) for l in layers):
                    ts = ME.batch_norm_group(layers, ts)
                else:
                    ts = [l(t) for l, t in zip(layers, ts)]
            outputs.update(zip(shared, ts))
        for network_head in self.cfg.network_heads:
            if network_head not in outputs:
                src = outputs['vox_feats'] if (self.requires_voxel_outputs and 'per_vox' in network_head) else out
                outputs[network_head] = self.network_heads[network_head](src)
            if self.cfg.mlp_bounds_relu and network_head == self.cfg.mlp_bounds:
                outputs[network_head] = self.relu(outputs[network_head])
        if perm is not None:
            # per-voxel results leave in the row order of the input coordinates (detection_net.py:347 relies on it)
            for name, t in outputs.items():
                if t.F.shape[0] == perm.shape[0] and not isinstance(t, ME.PooledTensor):
                    outputs[name] = ME.PooledTensor(t.features_in_input_order())
        ME.flush_batch_counters()
        ME.defer_counters(False)
        return outputs

    # ------------------------------------------------------------------ votes -> instance masks
    def detection2mask(self, batch, pred, cfg, mode, score_filtering=True, cluster_th=0.3, score_th=0.3,
                       mask_bin_th=0.3, mask_nms_th=0.3):
        """Predictions -> {scene name: {'conf','label_id','mask'}} (detection_net.py:369-488).
        Box construction and the sigmoid run with torch on the device `pred` lives on (CPU in the
        reference's evaluation flow); clustering, mask projection, mask NMS, the label histogram and
        the voxel->point projection are HIP kernels; only scalars and the final results cross PCIe."""
        _lib.require_gpu()
        dev = torch.device('cuda', torch.cuda.current_device())
        pdev = pred[cfg.mlp_offsets].device
        pred_bbs = to_bbs_min_max(batch['input_location'].to(pdev), pred[cfg.mlp_offsets], pred[cfg.mlp_bounds],
                                  torch.nn.Sigmoid()(pred[cfg.mlp_bb_scores]))
        if cfg.mlp_per_vox_semantics in cfg.network_heads:
            pred_semantics = torch.argmax(pred[cfg.mlp_per_vox_semantics], 1)
        else:
            pred_semantics = torch.argmax(pred[cfg.mlp_semantics], 1)
            pred_semantics = self.semantic_valid_class_ids.to(pdev)[pred_semantics].long()
        n_class = int(max(int(self.semantic_valid_class_ids.max()) + 1, 1))
        batch_ids = batch['batch_ids'].to(pdev)
        # The reference walks the scenes one by one (detection_net.py:390-477).  Here every stage runs for ALL scenes
        # before its (small) results are read back ONCE: clusters of the whole batch in one launch, then the score
        # filter, the mask NMS flags and the labels -- four host reads per batch instead of about ten per scene.
        # ---------- stage 0: per-scene inputs
        sc = []
        vox_start = 0
        for scene_idx, scene in enumerate(batch['scene']):
            scene_mask = batch_ids == scene_idx
            if cfg.do_segment_pooling:
                seg2vox = torch.as_tensor(batch['seg2vox'][scene_idx]).long()
            else:
                # predictions already live on the voxels (detection_net.py:436-445 projects only `if cfg.do_segment_pooling`):
                # the voxel of vote j IS j.  (The reference's branch then indexes per-voxel arrays with masks that have one
                # column per FOREGROUND vote, :463 / :470, and fails on the first background voxel; here background votes are
                # zero-padded exactly as the pooled branch does, which is the same result whenever the reference has one.)
                seg2vox = torch.arange(int(scene_mask.sum()), dtype=torch.long)
            n_vox = seg2vox.shape[0]
            s2v = seg2vox.to(dev)
            if not self.requires_voxel_outputs:
                scene_pred_semantics = pred_semantics[scene_mask]
                scene_pred_fg = self.is_foreground(scene_pred_semantics)
                sem_vox = scene_pred_semantics.to(dev)[s2v]
            else:
                # S3DIS flow (detection_net.py:398-415): per-voxel semantics, majority vote per segment
                sem_vox_all = pred_semantics[vox_start:vox_start + n_vox] if pred_semantics.shape[0] != n_vox \
                    else pred_semantics
                sem_vox = sem_vox_all.to(dev)
                n_seg = int(s2v.max()) + 1
                votes = torch.bincount(s2v * n_class + sem_vox, minlength=n_seg * n_class).reshape(n_seg, n_class)
                scene_pred_fg = self.is_foreground(torch.argmax(votes, 1)).to(pdev)
            vox_start += n_vox
            scene_pred_bbs = pred_bbs[scene_mask][scene_pred_fg]
            boxes = scene_pred_bbs.detach().to(dev, torch.float32).contiguous()
            if boxes.shape[0] == 0:
                # the reference crashes on a scene without foreground votes (torch.stack([]), iou_nms.py:103)
                raise ValueError('NMS_clustering needs at least one box (the reference raises on n == 0 as well)')
            fg_dev = scene_pred_fg.to(dev)
            fg_slot = torch.where(fg_dev, torch.cumsum(fg_dev.int(), 0) - 1, torch.full_like(fg_dev.int(), -1)).int()
            sc.append(dict(name=scene['name'], idx=scene_idx, s2v=s2v, n_vox=n_vox, sem32=sem_vox.int().contiguous(),
                           fg=scene_pred_fg, fg_dev=fg_dev, fg_slot=fg_slot, bbs=scene_pred_bbs, boxes=boxes,
                           n_fg=boxes.shape[0], words=(n_vox + 63) // 64))
        if not sc:
            return {}
        # ---------- stage 1: instance clusters (iou_nms.py:68-105), all scenes in one launch
        rs = iou_nms.nmc_device_batch([d['boxes'] for d in sc], cluster_th)
        # ---------- stage 2: score filter (detection_net.py:427-432) on the host from one read of the representatives' scores
        rep_scores = torch.cat([d['boxes'][r.reps[:r.k].long(), 0] for d, r in zip(sc, rs)]).cpu().numpy()
        reps_host = torch.cat([r.reps[:r.k] for r in rs]).cpu().numpy().astype(np.int64)
        o = 0
        for d, r in zip(sc, rs):
            d['reps'] = reps_host[o:o + r.k]
            scores = rep_scores[o:o + r.k]
            o += r.k
            sel = np.nonzero(scores > np.float32(score_th))[0] if score_filtering else np.arange(r.k)
            d['sel'] = sel.astype(np.int64)
        # ---------- heat-maps -> voxel masks (436-446): zero-padded background, seg2vox projection; duplicate removal
        # (448): mask NMS, skipped for per-voxel predictions (449-451).  One launch per stage for ALL scenes: a table of
        # B2M_MASK_DESC fields per scene (include/b2m.h) carries the scene's pointers and sizes.
        S = len(sc)
        desc = np.zeros((S, 20), np.int64)
        ksels = [int(d['sel'].shape[0]) for d in sc]
        sel_all = torch.from_numpy(np.concatenate([d['sel'] for d in sc]).astype(np.int32)).to(dev) if sum(ksels) else \
            torch.zeros(1, dtype=torch.int32, device=dev)
        bits_all = torch.empty(max(sum(k * max(d['words'], 1) for k, d in zip(ksels, sc)), 1), dtype=torch.int64, device=dev)
        inter_all = torch.empty(max(sum(k * k for k in ksels), 1), dtype=torch.int32, device=dev)
        keep_all = torch.empty(max(sum(ksels), 1), dtype=torch.int32, device=dev)
        do_nms = not self.requires_voxel_outputs
        keep_host = np.zeros(0, np.int32)
        row0 = boff = ioff = 0
        for s_, (d, r, ksel) in enumerate(zip(sc, rs, ksels)):
            d['bits_off'] = boff
            desc[s_, 0:9] = (r.heat.data_ptr(), d['n_fg'], sel_all.data_ptr() + 4 * row0, ksel, d['fg_slot'].data_ptr(),
                             d['s2v'].data_ptr(), d['n_vox'], bits_all.data_ptr() + 8 * boff, d['words'])
            desc[s_, 9] = inter_all.data_ptr() + 4 * ioff
            desc[s_, 10] = keep_all.data_ptr() + 4 * row0 if (do_nms and ksel > 0) else 0
            desc[s_, 18] = row0
            row0 += ksel; boff += ksel * max(d['words'], 1); ioff += ksel * ksel
        total_sel = row0
        desc_dev = torch.from_numpy(desc).to(dev)
        _call('b2m_mask_project_batch', desc_dev.data_ptr(), S, total_sel, max(d['words'] for d in sc), float(mask_bin_th))
        if do_nms and total_sel:
            _call('b2m_mask_nms_batch', desc_dev.data_ptr(), S, max(ksels), float(mask_nms_th))
            keep_host = keep_all[:total_sel].cpu().numpy()           # the one host read of the stage
        o = 0
        for d, ksel in zip(sc, ksels):
            d['kept'] = np.nonzero(keep_host[o:o + ksel])[0].astype(np.int64) if do_nms else np.arange(ksel, dtype=np.int64)
            o += ksel
        # ---------- label per instance: argmax of the label histogram inside the mask (461-466)
        kks = [int(d['kept'].shape[0]) for d in sc]
        total_kept = sum(kks)
        kept_all = torch.from_numpy(np.concatenate([d['kept'] for d in sc]).astype(np.int32)).to(dev) if total_kept else \
            torch.zeros(1, dtype=torch.int32, device=dev)
        labels_all = torch.zeros(max(total_kept, 1), dtype=torch.int32, device=dev)
        outs, v2ps = [], []
        row0 = 0
        for s_, (d, kk) in enumerate(zip(sc, kks)):
            if mode == 'eval':
                v2p = torch.as_tensor(batch['vox2point'][d['idx']]).long().to(dev)
                n_pts, idx_ptr = int(v2p.shape[0]), v2p.data_ptr()
                v2ps.append(v2p)
            else:
                n_pts, idx_ptr = d['n_vox'], 0
            out = torch.empty((kk, n_pts), dtype=torch.uint8, device=dev)
            outs.append(out)
            desc[s_, 11:18] = (kept_all.data_ptr() + 4 * row0, kk, d['sem32'].data_ptr(), labels_all.data_ptr() + 4 * row0,
                               idx_ptr, n_pts, out.data_ptr())
            desc[s_, 19] = row0
            row0 += kk
        # (the scenes' voxel-major scratch images of b2m_mask_gather_batch_t ride behind the table: one upload)
        nws = [(kk + 63) // 64 for kk in kks]
        tb_all = torch.empty(max(sum(d['n_vox'] * nw for d, nw in zip(sc, nws)), 1), dtype=torch.int64, device=dev)
        tb_ptrs = np.zeros(S, np.int64)
        toff = 0
        for s_, (d, nw) in enumerate(zip(sc, nws)):
            tb_ptrs[s_] = tb_all.data_ptr() + 8 * toff
            toff += d['n_vox'] * nw
        desc_dev = torch.from_numpy(np.concatenate([desc.reshape(-1), tb_ptrs])).to(dev)
        _call('b2m_label_hist_batch', desc_dev.data_ptr(), S, total_kept, n_class)
        if os.environ.get('B2M_MASK_GATHER_T', '1') == '1':
            _call('b2m_mask_gather_batch_t', desc_dev.data_ptr(), S, total_kept, max(int(o_.shape[1]) for o_ in outs),
                  max(d['words'] for d in sc), desc_dev.data_ptr() + 8 * S * 20)
        else:
            _call('b2m_mask_gather_batch', desc_dev.data_ptr(), S, total_kept, max(int(o_.shape[1]) for o_ in outs))
        labels_host = labels_all[:total_kept].cpu().numpy().astype('int32')
        # bytes every implementation of the four stages moves (for bench.py's roofline of the leg; nothing on the path reads it)
        self._d2m_bytes = {
            'b2m_mask_project_batch': float(sum(12 * d['n_vox'] + 4 * k * d['n_fg'] + 8 * k * d['words'] for d, k in zip(sc, ksels))),
            'b2m_mask_nms_batch': float(sum(8 * k * d['words'] + 4 * k * k for d, k in zip(sc, ksels))) if do_nms else 0.0,
            'b2m_label_hist_batch': float(sum(8 * kk * d['words'] + 4 * d['n_vox'] for d, kk in zip(sc, kks))),
            'b2m_mask_gather_batch': float(sum(8 * kk * d['words'] + 8 * o_.shape[1] + kk * o_.shape[1] for d, kk, o_ in zip(sc, kks, outs))),
        }
        self._d2m_bytes['b2m_mask_gather_batch_t'] = self._d2m_bytes['b2m_mask_gather_batch']      # (the same operands, every one once)
        results = {}
        o = 0
        for d, r, out in zip(sc, rs, outs):
            kk = d['kept'].shape[0]
            instance_labels = labels_host[o:o + kk]
            o += kk
            final_rows = d['sel'][d['kept']]             # cluster row of every surviving instance
            rep_rows = torch.from_numpy(d['reps'][final_rows]).to(pdev)
            bb_scores = d['bbs'][rep_rows, 0]
            if mode == 'eval':
                results[d['name']] = {'conf': bb_scores, 'label_id': instance_labels, 'mask': out.bool().to(pdev)}
            else:
                fr = torch.from_numpy(final_rows).to(dev)
                heat_w_bg = torch.zeros((kk, d['fg_dev'].shape[0]), device=dev)
                heat_w_bg[:, d['fg_dev']] = r.heat[fr]
                results[d['name']] = {
                    'conf': bb_scores, 'label_id': instance_labels, 'mask': out.bool().to(pdev),
                    'cluster_representatives': rep_rows,
                    'cluster_heatmaps': heat_w_bg[:, d['s2v']].to(pdev),
                    'bbs': d['bbs'][rep_rows],
                    'pred_fg': d['fg'],
                }
        return results

    # ------------------------------------------------------------------ prediction
    def get_prediction(self, batch, with_grad=True, to_cpu=False, to_numpy=False, min_size=True, sin=None):
        """detection_net.py:493-517.  `sin`: the batch's sparse tensor if the caller holds it already (Model.prefetch built it
        on a second stream, one scene ahead of the evaluation loop) -- same maps, same outputs."""
        n_seg = batch['input_location'].shape[0] if self.cfg.do_segment_pooling else None
        if not with_grad:
            with torch.no_grad():
                if sin is None:
                    sin = ME.SparseTensor(batch['vox_features'], batch['vox_coords'], device=self.device)
                pred = self(sin, batch['pooling_ids'].to(self.device), n_seg)
        else:
            if sin is None:
                sin = ME.SparseTensor(batch['vox_features'], batch['vox_coords'], device=self.device)
            pred = self(sin, batch['pooling_ids'].to(self.device), n_seg)
        for mlp_head, sparse_tensor in pred.items():
            pred[mlp_head] = sparse_tensor.F if not to_cpu else sparse_tensor.F.cpu()
        if min_size:
            self.to_min_size(pred)
        if to_numpy:
            for mlp_head, tensor in pred.items():
                pred[mlp_head] = tensor.numpy()
        return pred

    def to_min_size(self, pred):
        if self.cfg.mlp_bounds in pred.keys() and self.cfg.min_bb_size is not None:
            pred[self.cfg.mlp_bounds] = torch.clamp(pred[self.cfg.mlp_bounds], min=self.cfg.min_bb_size)
