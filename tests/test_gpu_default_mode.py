"""The mode bench.py times -- weight gradients on the side stream, in-place pass-through gradients, atomic weight-gradient
and split-K combines, BatchNorm statistics from the convolution epilogue, Model.prefetch beside the step -- against the
oracle at NETWORK level (reference schedule: /root/reference/models/detection_net.py:234-364, resnet.py:70-83).

No test in this file sets B2M_DETERMINISTIC: a missing stream join, a record_stream slip or an in-place accumulate onto a
tensor another consumer still reads shows here as a wrong gradient.
"""
import os

import numpy as np
import pytest
import torch

from box2mask_amd import synth
from box2mask_amd.config import scannet_config

pytestmark = pytest.mark.gpu

HEADS = ['mlp_offsets', 'mlp_bounds', 'mlp_bb_scores', 'mlp_semantics']


def _rel(a, b):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)


def _default_env(monkeypatch):
    for k in ('B2M_DETERMINISTIC', 'B2M_WGRAD_STREAM', 'B2M_CONV_PASSTHROUGH', 'B2M_CONV_STATS', 'B2M_CONV_PIPE',
              'B2M_CONV_FLOW_SPLIT', 'B2M_WGRAD_PIPE', 'B2M_XCD', 'B2M_CONV_1X1', 'B2M_CONV_TARGET'):
        monkeypatch.delenv(k, raising=False)
    from box2mask_amd import _lib
    if hasattr(_lib, 'reload_env'):
        _lib.reload_env()


from _parity import MaskRecorder as _MaskRecorder, row_rel_err  # noqa: E402


def _oracle_grads(sd, batch, gws, cfg, training, dtype, hier=None):
    from oracle import unet_ref
    p = {k: (v.to(dtype).clone().requires_grad_(True) if v.is_floating_point() and 'running' not in k
             else (v.to(dtype) if v.is_floating_point() else v)) for k, v in sd.items()}
    out = unet_ref.forward(p, batch['vox_coords'].numpy(), batch['vox_features'].to(dtype), batch['pooling_ids'], cfg,
                           training=training, n_segments=batch['input_location'].shape[0], hier=hier)
    sum((out[h] * gws[h].to(dtype)).sum() for h in HEADS).backward()
    return p, out


W = {'mlp_offsets': 3, 'mlp_bounds': 3, 'mlp_bb_scores': 1, 'mlp_semantics': 20}


def test_default_mode_gradients_match_oracle_affine_batchnorm(monkeypatch):
    """BatchNorm in eval mode is an affine map, so with the ReLU decisions shared (_MaskRecorder) the network is one
    linear map on both sides: EVERY parameter gradient of the 8-level network must agree with the fp64 CPU oracle to 1e-3
    of its maximum (observed: a few 1e-5).  Three different batches in one process (the caching allocator hands the
    previous step's blocks out again while the side stream may still be reading them), the next batch's maps prefetched
    on a third stream beside each backward pass, no synchronisation between backward and the reads of the gradients."""
    _default_env(monkeypatch)
    from box2mask_amd import functional as F_
    from box2mask_amd.model import Model
    from box2mask_amd import nn as ME
    from oracle import sparse_ref
    assert F_.wgrad_on_side_stream() and F_.conv_passthrough() and not F_.deterministic()
    cfg = scannet_config()
    torch.manual_seed(11)
    model = Model(cfg, *synth.scannet_tables())
    net = model.detection_model
    # scenes large enough that levels 0-1 run un-split, levels 2.. the split maps, and the weight gradients several chunks
    batches = [synth.make_batch(5, seed0=300 + 10 * r, target_voxels=(4000, 6000, 2500)[r], pts_per_m2=8000.0) for r in range(3)]
    # running statistics := statistics of batch 0 (momentum 1), affine parameters away from (1, 0): a normalising,
    # non-trivial affine BatchNorm
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, ME.MinkowskiBatchNorm):
                m.bn.momentum = 1.0
                m.bn.weight.uniform_(0.6, 1.4)
                m.bn.bias.uniform_(-0.3, 0.3)
        net.train()
        net(ME.SparseTensor(batches[0]['vox_features'], batches[0]['vox_coords']), batches[0]['pooling_ids'].cuda(),
            batches[0]['input_location'].shape[0])
    net.eval()
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    worst_all = 0.0
    for r, batch in enumerate(batches):
        S_ = batch['input_location'].shape[0]
        torch.manual_seed(20 + r)
        gws = {h: torch.randn(S_, W[h]) for h in HEADS}
        for p in net.parameters():
            p.grad = None
        with monkeypatch.context() as mp:
            rec = _MaskRecorder(mp)
            sin = ME.SparseTensor(batch['vox_features'], batch['vox_coords'])
            assert sin.manager.perm is not None                  # the benchmarked row order
            out = net(sin, batch['pooling_ids'].cuda(), S_)
        loss = sum((out[h].F * gws[h].cuda()).sum() for h in HEADS)
        model.prefetch(batches[(r + 1) % 3], ready=True)         # third stream: the next batch's maps beside the backward pass
        loss.backward()
        # NO synchronize here: the gradients are read on the current stream, which backward's callback joined
        grads = {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
        model._prefetched = None
        hier = sparse_ref.Hierarchy(batch['vox_coords'].numpy())
        with monkeypatch.context() as mp:
            it = rec.replay(sin.manager, hier, S_, mp)
            p64, o64 = _oracle_grads(sd, batch, gws, cfg, False, torch.float64, hier)
            assert next(it, None) is None                        # every recorded decision was consumed, in order
        print('batch %d: %d replayed ReLUs, largest disagreeing fraction %.2e, largest |x|/rms of a disagreeing element %.2e'
              % ((r, len(rec.checks)) + rec.summary()))
        for h in HEADS:
            assert _rel(out[h].F, o64[h]) < 1e-3, (r, h, _rel(out[h].F, o64[h]))
            assert row_rel_err(out[h].F, o64[h]) < 1e-2, (r, h, row_rel_err(out[h].F, o64[h]))
        rows = sorted(((_rel(grads[n], p64[n].grad), n) for n in grads), reverse=True)
        assert len(rows) == len(list(net.parameters())) > 250    # (eval-mode BatchNorm has parameter gradients too)
        print('batch %d (%d voxels): worst gradient errors' % (r, batch['vox_coords'].shape[0]), rows[:3])
        worst_all = max(worst_all, rows[0][0])
        assert rows[0][0] < 1e-3, rows[:5]
    print('worst relative gradient error over 3 batches: %.3e' % worst_all)


def test_default_mode_train_batchnorm_many_scenes(monkeypatch):
    """Train-mode BatchNorm, default mode, 32 small scenes: the deepest level keeps >= 32 rows, so the batch statistics
    are well conditioned.  With the ReLU decisions shared, every parameter gradient against the fp64 oracle <= 1e-3
    (observed 8e-5)."""
    _default_env(monkeypatch)
    from box2mask_amd.detection_net import SelectionNet
    from box2mask_amd import nn as ME
    from oracle import sparse_ref
    cfg = scannet_config()
    valid, _, _, is_fg = synth.scannet_tables()
    torch.manual_seed(2)
    net = SelectionNet(cfg, 'cuda', valid, is_fg, out_channels=[96, 96, 6]).cuda().train()
    batch = synth.make_batch(32, seed0=500, target_voxels=1100, pts_per_m2=6000.0)
    S_ = batch['input_location'].shape[0]
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    torch.manual_seed(1)
    gws = {h: torch.randn(S_, W[h]) for h in HEADS}
    with monkeypatch.context() as mp:
        rec = _MaskRecorder(mp)
        sin = ME.SparseTensor(batch['vox_features'], batch['vox_coords'])
        assert sin.manager.n(7) >= 32
        out = net(sin, batch['pooling_ids'].cuda(), S_)
    sum((out[h].F * gws[h].cuda()).sum() for h in HEADS).backward()
    hier = sparse_ref.Hierarchy(batch['vox_coords'].numpy())
    # ONE fp64 oracle pass, with the device's ReLU decisions replayed (and checked against the oracle's own: tests/_parity.py).
    # (Rounds 3-4 ran a second, unshared pass only to print the per-cent-level figures it gives; tools/debug_default_mode.py still does.)
    with monkeypatch.context() as mp:
        rec.replay(sin.manager, hier, S_, mp)
        pm, om = _oracle_grads(sd, batch, gws, cfg, True, torch.float64, hier)
    print('%d replayed ReLUs, largest disagreeing fraction %.2e, largest |x|/rms of a disagreeing element %.2e'
          % ((len(rec.checks),) + rec.summary()))
    for h in HEADS:
        assert _rel(out[h].F, om[h]) < 1e-3, (h, _rel(out[h].F, om[h]))
    rows = sorted(((_rel(prm.grad, pm[name].grad), name) for name, prm in net.named_parameters()), reverse=True)
    e_m = sorted(r[0] for r in rows)
    q = lambda v, f: v[min(int(f * len(v)), len(v) - 1)]
    print('gradient error vs fp64 (shared ReLU decisions): median %.3e, p90 %.3e, max %.3e' % (q(e_m, .5), q(e_m, .9), e_m[-1]))
    for r in rows[:5]:
        print('   worst: %.3e %s' % r)
    assert e_m[-1] < 1e-3, rows[:5]


def test_default_and_deterministic_mode_on_a_block_chain(monkeypatch):
    """conv-BN-ReLU-conv-BN (+1x1 shortcut-BN) add ReLU twice, the strided convolution, a block on the next level, the
    transposed convolution back, ME.cat with the skip and a block with a two-source shortcut -- train-mode BatchNorm over
    tens of thousands of rows (no tiny levels).  The default mode (side stream, in-place pass-through accumulation, epilogue
    statistics, split maps, paired BatchNorm) and the ordered single-stream mode are BOTH held against the fp64 oracle with
    the ReLU decisions of the respective device run: output, input gradient and every parameter gradient <= 1e-4 (observed
    2e-6).  The two modes' outputs agree to 1e-4 directly; their GRADIENTS are not compared with each other: the two
    forwards differ by 5e-7, a handful of the 2.9 M units of a layer land on the other side of zero, and one such unit moves
    a weight gradient that is a sum over N pairs by about 1/sqrt(N) of its size -- measured 4e-2 on the 8 x 64 x 32
    transposed-convolution kernel between two CORRECT runs (tools/debug_chain.py)."""
    from box2mask_amd import nn as ME, _lib
    from box2mask_amd.resnet import BasicBlock
    from box2mask_amd import functional as F_
    from oracle import sparse_ref as S
    from torch import nn
    b = synth.make_batch(3, seed0=70, target_voxels=20000, pts_per_m2=8000.0)

    class Chain(nn.Module):
        def __init__(self):
            super().__init__()
            short = nn.Sequential(ME.MinkowskiConvolution(32, 64, kernel_size=1, dimension=3), ME.MinkowskiBatchNorm(64))
            self.b0 = BasicBlock(32, 64, downsample=short, dimension=3)
            self.b1 = BasicBlock(64, 64, dimension=3)
            self.down = ME.MinkowskiConvolution(64, 64, kernel_size=2, stride=2, dimension=3)
            self.bn = ME.MinkowskiBatchNorm(64)
            self.b2 = BasicBlock(64, 64, dimension=3)
            self.up = ME.MinkowskiConvolutionTranspose(64, 32, kernel_size=2, stride=2, dimension=3)
            self.bnu = ME.MinkowskiBatchNorm(32)
            short2 = nn.Sequential(ME.MinkowskiConvolution(96, 48, kernel_size=1, dimension=3), ME.MinkowskiBatchNorm(48))
            self.b3 = BasicBlock(96, 48, downsample=short2, dimension=3)

        def forward(self, x):
            e = self.b1(self.b0(x))
            d, e = self.down(e, passthrough=True)
            d = d.new(self.bn.apply_bn(d.F, relu=True, count_key=ME.count_key_of(d)))
            d = self.b2(d)
            u = self.up(d)
            u = u.new(self.bnu.apply_bn(u.F, relu=True, count_key=ME.count_key_of(u)))
            y = self.b3(ME.cat(u, e))
            ME.flush_batch_counters()
            return y

    def oracle(sd, x, gy, hier):
        """The chain on oracle/sparse_ref.py in fp64 (resnet.py:70-83 per block), rows in input order."""
        p = {k: (v.double().clone().requires_grad_(True) if v.is_floating_point() and 'running' not in k else v)
             for k, v in sd.items()}
        bn = lambda name, t: S.batch_norm(t, p[name + '.bn.weight'], p[name + '.bn.bias'], None, None, True)

        def block(name, t, nbr, short):
            out = torch.relu(bn(name + '.norm1', S.conv_nbr(t, p[name + '.conv1.kernel'], nbr)))
            out = bn(name + '.norm2', S.conv_nbr(out, p[name + '.conv2.kernel'], nbr))
            res = bn(name + '.downsample.1', S.conv_nbr(t, p[name + '.downsample.0.kernel'], None)) if short else t
            return torch.relu(out + res)
        x = x.double().clone().requires_grad_(True)
        e = block('b1', block('b0', x, hier.k3(0), True), hier.k3(0), False)
        d = torch.relu(bn('bn', S.conv_nbr(e, p['down.kernel'], hier.down(0))))
        d = block('b2', d, hier.k3(1), False)
        u = torch.relu(bn('bnu', S.conv_nbr(d, p['up.kernel'], hier.up(0))))
        y = block('b3', torch.cat([u, e], 1), hier.k3(0), True)
        (y * gy.double()).sum().backward()
        return y.detach(), x.grad, {k: v.grad for k, v in p.items() if torch.is_tensor(v) and v.requires_grad}

    hier = S.Hierarchy(b['vox_coords'].numpy(), n_levels=2)
    torch.manual_seed(10)
    feats = torch.randn(b['vox_coords'].shape[0], 32)
    gy = torch.randn(b['vox_coords'].shape[0], 48)
    outputs = {}
    for mode in ('default', 'default again', 'deterministic'):
        _default_env(monkeypatch)
        if mode == 'deterministic':
            monkeypatch.setenv('B2M_DETERMINISTIC', '1')
        else:
            assert F_.wgrad_on_side_stream()
        torch.manual_seed(9)
        net = Chain().cuda().train()
        sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
        with monkeypatch.context() as mp:
            rec = _MaskRecorder(mp)
            sin = ME.SparseTensor(feats, b['vox_coords'])          # (rows permuted to Morton order inside)
            sin.F.requires_grad_(True)
            F_.packed_weights.begin_pass()
            y = net(sin).F
        perm, inv = sin.manager.perm, sin.manager.inv_perm
        (y * gy.cuda()[perm]).sum().backward()
        # no synchronize: read on the current stream
        y_in, dx_in = y.detach()[inv].cpu(), sin.F.grad[inv].cpu()      # back to input row order
        g = {n: q.grad.cpu() for n, q in net.named_parameters()}
        if mode == 'default again':          # (the second default run: against the first one, not a third oracle pass)
            assert _rel(y_in, outputs['default']) < 1e-4
            outputs[mode] = y_in
            continue
        with monkeypatch.context() as mp:
            it = rec.replay(sin.manager, hier, -1, mp)
            oy, odx, og = oracle(sd, feats, gy, hier)
            assert next(it, None) is None
        worst = max((_rel(g[n], og[n]), n) for n in g)
        print('%-14s vs fp64 oracle: output %.2e  input gradient %.2e  worst parameter gradient %.2e %s'
              % (mode, _rel(y_in, oy), _rel(dx_in, odx), worst[0], worst[1]))
        assert _rel(y_in, oy) < 1e-4 and _rel(dx_in, odx) < 1e-4 and worst[0] < 1e-4, (mode, worst)
        outputs[mode] = y_in
    monkeypatch.delenv('B2M_DETERMINISTIC')
    assert _rel(outputs['default'], outputs['deterministic']) < 1e-4
    assert _rel(outputs['default again'], outputs['deterministic']) < 1e-4


def test_full_size_scene_forward_matches_oracle(monkeypatch):
    """ONE 150 k-voxel scene (the metric's size), whole network, train-mode BatchNorm, default mode: every head and the
    per-voxel trunk features against oracle/unet_ref.py within the north_star tolerance (1e-3 of the tensor's maximum).
    The composition no per-layer test sees: epilogue statistics -> tile-statistics BatchNorm -> pass-through aliases ->
    split maps exactly as they occur on a full scene."""
    _default_env(monkeypatch)
    from box2mask_amd.detection_net import SelectionNet
    from box2mask_amd import nn as ME
    from oracle import unet_ref
    cfg = scannet_config()
    valid, _, _, is_fg = synth.scannet_tables()
    torch.manual_seed(0)
    net = SelectionNet(cfg, 'cuda', valid, is_fg, out_channels=[96, 96, 6]).cuda().train()
    batch = synth.make_batch(1, seed0=0, target_voxels=150_000)
    n = batch['vox_coords'].shape[0]
    assert 135_000 <= n <= 165_000
    S_ = batch['input_location'].shape[0]
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    net._trace = {}
    with torch.no_grad():
        sin = ME.SparseTensor(batch['vox_features'], batch['vox_coords'])
        out = net(sin, batch['pooling_ids'].cuda(), S_)
        trunk = net._trace['block8'][sin.manager.inv_perm] if sin.manager.inv_perm is not None else net._trace['block8']
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    with torch.no_grad():
        ref = unet_ref.forward(sd, batch['vox_coords'].numpy(), batch['vox_features'], batch['pooling_ids'], cfg,
                               training=True, n_segments=S_, return_trunk=True)
    errs = {h: _rel(out[h].F, ref[h]) for h in HEADS}
    errs['vox_feats'] = _rel(trunk, ref['_trunk'])
    print('full-size scene (%d voxels, %d rows at level 7): rel errors %s' % (n, sin.manager.n(7), errs))
    assert max(errs.values()) < 1e-3, errs


@pytest.mark.gpu
def test_weight_images_packed_on_the_side_stream_follow_the_optimizer(monkeypatch):
    """A training pass packs the images of its first layers on the main stream and the rest -- the late forward images, every
    data-gradient image -- on the side stream, beside the first layers (functional._PackedWeights.begin_pass); the first user of
    such an image waits for it.  Four optimizer steps: after every pass EVERY registered image must be, bit for bit, the image of
    the weights as the optimizer left them before that pass (an image that missed an update, or was packed from weights the
    optimizer was still writing, differs), and the groups are where the design says: the late forward images were waited for
    during the forward pass, the data-gradient images are still guarded when backward begins.  (A trajectory comparison cannot
    do this job in the default mode: with a handful of rows at the deepest level the atomics' rounding noise alone moves the
    second loss of two identical runs by 4e-4.)"""
    from box2mask_amd import functional as F_
    from box2mask_amd.model import Model
    _default_env(monkeypatch)
    batch = synth.make_batch(4, seed0=21, target_voxels=20000, pts_per_m2=8000.0)
    torch.manual_seed(11)
    model = Model(scannet_config(), *synth.scannet_tables())
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=3e-3, fused=True)
    pw = F_.packed_weights
    for step in range(4):
        opt.zero_grad()
        ld = model.compute_loss(batch, 150)
        guarded = sorted(pw.pending)
        ld['optimization_loss'].backward()
        torch.cuda.synchronize()
        if step >= 1:                       # (the first pass registers the images one by one; from the second on the plan exists)
            assert guarded == [2], guarded
            sizes = [sum(1 for g in pw.group.values() if g == j) for j in range(3)]
            assert min(sizes) > 0, sizes    # every group is populated: early / late forward images, data-gradient images
        bad = []
        for key, e in pw.entries.items():
            w = e[0]()
            if w is None:
                continue
            K, cin, cout, transpose, mirror, sb, sc = e[1]
            w3 = w.detach() if w.dim() == 3 else w.detach().unsqueeze(0)
            fresh = F_.weight_pack(w3, transpose, mirror, sb, sc)
            if not torch.equal(fresh, e[2]):
                bad.append((key[1:], tuple(w.shape)))
        assert not bad, (step, len(bad), bad[:4])
        opt.step()
    assert len(pw.entries) > 150
