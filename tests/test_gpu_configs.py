"""BASELINE.json configs 3-5 as parity / property cases (they are not bench lines):
S3DIS-shaped (13 classes, per-voxel semantics head, large rooms) and ARKit-shaped (28 classes, 4 cm voxels,
mixed scene sizes, fp16 input features)."""
import numpy as np
import pytest
import torch

from box2mask_amd import synth
from box2mask_amd.config import scannet_config

pytestmark = pytest.mark.gpu


def _s3dis_tables():
    valid = torch.Tensor(np.arange(13))
    id2idx = torch.arange(13).long()
    return valid, id2idx, id2idx.clone(), (lambda s: s > 2)


def _s3dis_cfg(**kw):
    return scannet_config(network_heads=['mlp_offsets', 'mlp_bounds', 'mlp_bb_scores', 'mlp_per_vox_semantics'],
                          eval_ths=[0.5, 0.03, 0.3, 0.6], loss_weight_bb_scores=3.0, batch_size=4, **kw)


def _rel(a, b):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-9)


def test_s3dis_config_forward_matches_oracle():
    """Per-voxel head (N_vox x 13) + segment heads on the same trunk (detection_net.py:342-359)."""
    from box2mask_amd.detection_net import SelectionNet
    from box2mask_amd import nn as ME
    from oracle import unet_ref
    cfg = _s3dis_cfg()
    valid, _, _, is_fg = _s3dis_tables()
    torch.manual_seed(3)
    net = SelectionNet(cfg, 'cuda', valid, is_fg, out_channels=[96, 96, 6]).cuda().train()
    batch = synth.make_batch(8, seed0=40, target_voxels=1500, pts_per_m2=6000.0)
    S_ = batch['input_location'].shape[0]
    p_cpu = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    out = net(ME.SparseTensor(batch['vox_features'], batch['vox_coords']), batch['pooling_ids'].cuda(), S_)
    ref = unet_ref.forward(p_cpu, batch['vox_coords'].numpy(), batch['vox_features'], batch['pooling_ids'], cfg,
                           training=True, n_segments=S_)
    assert out['mlp_per_vox_semantics'].F.shape == (batch['vox_coords'].shape[0], 13)
    assert out['mlp_offsets'].F.shape == (S_, 3)
    for h in cfg.network_heads:
        assert _rel(out[h].F, ref[h]) < 1e-3, h
    assert _rel(out['vox_feats'].F, ref['vox_feats']) < 1e-3


def test_s3dis_sized_room_train_step_and_masks():
    """Two ~400k-voxel rooms (S3DIS rooms are 0.25-1 M voxels after the reference's 0.25 point sampling):
    size-independent properties of a full step and of the votes->masks path."""
    from box2mask_amd.model import Model
    cfg = _s3dis_cfg()
    torch.manual_seed(0)
    model = Model(cfg, *_s3dis_tables())
    items = [synth.make_scene(s, target_voxels=400_000) for s in (0, 1)]
    batch = synth.collate(items)
    n_vox = batch['vox_coords'].shape[0]
    assert n_vox > 700_000
    rng = np.random.default_rng(0)
    batch['gt_semantics'] = batch['gt_semantics'] % 13
    batch['gt_per_vox_semantics'] = torch.from_numpy(rng.integers(0, 13, n_vox))
    model.train()
    losses = model.compute_loss(batch, 150)
    losses['optimization_loss'].backward()
    for k, v in losses.items():
        assert np.isfinite(v.item() if hasattr(v, 'item') else float(v)), k
    assert 'per_vox_semantics_loss' in losses
    g = [p.grad for p in model.parameters()]
    assert all(x is not None and torch.isfinite(x).all() for x in g)
    # coordinate hierarchy of a big room: every level strictly smaller, deepest levels non-trivial
    from box2mask_amd.sparse import CoordinateManager
    m = CoordinateManager(batch['vox_coords']); m.ensure_level(7)
    ns = [m.n(l) for l in range(8)]
    assert all(a > b for a, b in zip(ns, ns[1:])) and ns[7] >= 2
    # inference + masks for one room (the reference evaluates S3DIS with batch size 1)
    one = synth.collate(items[:1], mode='test')
    model.eval()
    pred = model.get_prediction(one)
    assert pred['mlp_per_vox_semantics'].shape == (one['vox_coords'].shape[0], 13)
    res = model.pred2mask(one, pred, 'eval')
    r = res[one['scene'][0]['name']]
    assert r['mask'].dtype == torch.bool and r['mask'].shape[1] == len(one['vox2point'][0])
    assert r['mask'].shape[0] == len(r['conf']) == len(r['label_id'])
    if r['mask'].shape[0]:
        assert bool(r['mask'].any(1).all())              # every instance has at least its representative segment
        assert (np.asarray(r['label_id']) >= 0).all() and (np.asarray(r['label_id']) < 13).all()


def _arkit_tables():
    ids = np.array([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 15, 16, 18, 19, 20, 21, 22, 23, 24, 25, 28, 33, 34, 36, 39])
    valid = torch.Tensor(ids)
    id2idx = torch.zeros(41).fill_(-100).long(); id2idx[ids] = torch.arange(len(ids)).long()
    return valid, id2idx, id2idx.clone(), (lambda s: s > 2)


def test_arkit_config_fp16_features_mixed_scenes():
    """configs/arkitscenes.txt: 4 cm voxels, batch 4, 28 classes; features arrive as fp16."""
    from box2mask_amd.model import Model
    cfg = scannet_config(eval_ths=[0.5, 0.05, 0.4, 0.6], loss_weight_bb_scores=3.0, loss_weight_semantics=0.3,
                         voxel_size=0.04, batch_size=4)
    torch.manual_seed(1)
    model = Model(cfg, *_arkit_tables())
    sizes = [8000, 30000, 15000, 50000]                   # mixed-scale scenes
    items = [synth.make_scene(10 + i, target_voxels=tv, voxel_size=0.04, pts_per_m2=5000.0) for i, tv in enumerate(sizes)]
    batch = synth.collate(items)
    valid_ids = _arkit_tables()[0].long().numpy()
    batch['gt_semantics'] = torch.from_numpy(valid_ids[batch['gt_semantics'].numpy() % len(valid_ids)])
    assert model.detection_model.mlp_semantics[6].kernel.shape == (96, 28)
    half = dict(batch); half['vox_features'] = batch['vox_features'].half()
    full = dict(batch); full['vox_features'] = half['vox_features'].float()
    model.eval()
    p16 = model.get_prediction(half); p32 = model.get_prediction(full)
    for h in p16:
        # fp16 in, fp32 compute; equal up to the summation order of the fp32 atomics (segment mean, split-K)
        assert p16[h].dtype == torch.float32 and torch.allclose(p16[h], p32[h], rtol=1e-4, atol=1e-5), h
    model.train()
    losses = model.compute_loss(half, 150)
    losses['optimization_loss'].backward()
    assert np.isfinite(losses['optimization_loss'].item())
    res = model.pred2mask(batch, model.get_prediction(batch), 'eval')
    assert set(res) == {s['name'] for s in batch['scene']}
