"""Inference path (/root/reference/models/evaluation.py:70-98, detection_net.py:493-517): under model.eval() and
torch.no_grad() every trunk convolution applies the eval-mode BatchNorm (+ residual) (+ ReLU) that follows it on the way
out of its kernel (b2m_conv_fwd_affine, functional.conv_affine).

  1. the fused launch against convolution + b2m_bn_apply on the same tensors, BIT FOR BIT, for every kernel that carries
     the epilogue (flow kernel un-split / 4 slices, 1x1 kernel with 1 / 2 / 3 strips per wave, the 5x5x5 stem) and for the
     shapes that must fall back (more than 4 slices);
  2. the whole network: fused against unfused (bit for bit in deterministic mode, to rounding in the default mode) and
     against the CPU oracle (<= 1e-3, north_star tolerance).
"""
import zlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-9)


@pytest.fixture(scope='module')
def maps():
    from test_gpu_ops import _scene
    from box2mask_amd.sparse import CoordinateManager
    b = _scene()
    m = CoordinateManager(b['vox_coords'])
    m.ensure_level(2)
    return m


CASES = [
    # (kind, level, (c1, c2), cout, regime)
    ('k3', 0, (96, 0), 96, 'unsplit'), ('k3', 0, (96, 32), 96, 'unsplit'), ('k3', 0, (32, 0), 32, 'unsplit'),
    ('k3', 1, (64, 0), 128, 'split4'), ('k3', 1, (256, 128), 256, 'split4'), ('k3', 0, (96, 0), 96, 'many_slices'),
    ('k5', 0, (6, 0), 32, 'unsplit'), ('down', 0, (32, 0), 32, 'unsplit'), ('up', 0, (96, 0), 96, 'unsplit'),
    ('1x1', 0, (128, 0), 96, 'unsplit'), ('1x1', 0, (96, 32), 128, 'unsplit'), ('1x1', 0, (64, 0), 32, 'unsplit'),
]
ENV = {'unsplit': {'B2M_CONV_TARGET': '0'}, 'split4': {'B2M_CONV_TARGET': '800', 'B2M_CONV_CHUNKSPLIT': '0'},
       'many_slices': {'B2M_CONV_TARGET': '100000', 'B2M_CONV_MAXSLICE': '16'}}


@pytest.mark.parametrize('res,relu', [(False, True), (True, True), (False, False)])
@pytest.mark.parametrize('kind,level,cins,cout,regime', CASES)
def test_conv_affine_equals_conv_then_bn_apply(maps, monkeypatch, kind, level, cins, cout, regime, res, relu):
    from box2mask_amd import functional as F_, _lib
    for k, v in ENV[regime].items():
        monkeypatch.setenv(k, v)
    m = maps
    c1, c2 = cins
    if kind == 'k3':
        rb = m.rulebook_same(level, 3); K = 27; n_in = n_out = m.n(level)
    elif kind == 'k5':
        rb = m.rulebook_same(level, 5); K = 125; n_in = n_out = m.n(level)
    elif kind == 'down':
        rb = m.rulebook_down(level); K = 8; n_in, n_out = m.n(level), m.n(level + 1)
    elif kind == 'up':
        rb = m.rulebook_up(level); K = 8; n_in, n_out = m.n(level + 1), m.n(level)
    else:
        rb = None; K = 1; n_in = n_out = m.n(level)
    torch.manual_seed(zlib.crc32(repr((kind, level, cins, cout, regime)).encode()) % 1000)
    x1 = torch.randn(n_in, c1, device='cuda')
    x2 = torch.randn(n_in, c2, device='cuda') if c2 else None
    w = torch.randn(K, c1 + c2, cout, device='cuda') * 0.05 if K > 1 else torch.randn(c1 + c2, cout, device='cuda') * 0.05
    scale = torch.rand(cout, device='cuda') + 0.5
    shift = torch.randn(cout, device='cuda')
    r = torch.randn(n_out, cout, device='cuda') if res else None
    calls = []
    _lib.set_hook(lambda name, a, meta=None: calls.append(name))
    try:
        y = F_.conv_affine(x1, x2, w, rb, n_out, scale, shift, r, relu)
    finally:
        _lib.set_hook(None)
    fused = 'b2m_bn_apply' not in calls
    assert fused == (regime != 'many_slices'), calls          # the epilogue runs wherever the kernel can carry it
    # reference: the training-mode layering -- plain convolution, then b2m_bn_apply
    y0 = F_.sparse_conv(x1, x2, w, None, rb, rb, kind in ('k3', 'k5'), n_out)
    y1 = torch.empty_like(y0)
    F_._call('b2m_bn_apply', y0.data_ptr(), y0.stride(0), n_out, cout, scale.data_ptr(), shift.data_ptr(), F_._ptr(r),
             r.stride(0) if r is not None else 0, 1 if relu else 0, y1.data_ptr(), y1.stride(0))
    torch.cuda.synchronize()
    if regime == 'many_slices':
        assert _rel(y, y1) < 1e-5              # (atomic split-K combine: the order of the additions varies from run to run)
    else:
        assert torch.equal(y, y1), 'fused epilogue differs from conv + bn_apply: %.3e' % _rel(y, y1)
    # and against plain torch arithmetic
    t = y0 * scale + shift
    if res:
        t = t + r
    if relu:
        t = torch.relu(t)
    assert _rel(y, t) < 1e-5


def _model_and_batch(n_vox=9000, bs=3, seed=5):
    from box2mask_amd import synth
    from box2mask_amd.config import scannet_config
    from box2mask_amd.model import Model
    torch.manual_seed(seed)
    cfg = scannet_config()
    model = Model(cfg, *synth.scannet_tables(), device='cuda:0')
    batch = synth.make_batch(bs, seed0=seed, target_voxels=n_vox, pts_per_m2=6000.0)
    # running statistics and affine parameters away from their initial values
    with torch.no_grad():
        for mod in model.detection_model.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.running_mean.normal_(0, 0.2); mod.running_var.uniform_(0.5, 1.5)
                mod.weight.uniform_(0.7, 1.3); mod.bias.normal_(0, 0.1)
    model.eval()
    return model, batch, cfg


def _names(calls):
    out = {}
    for c in calls:
        out[c] = out.get(c, 0) + 1
    return out


def test_inference_fused_equals_unfused_and_oracle(monkeypatch):
    from box2mask_amd import _lib
    from oracle import unet_ref
    model, batch, cfg = _model_and_batch()
    p_cpu = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    res = {}
    for mode, env in (('det_fused', {'B2M_DETERMINISTIC': '1', 'B2M_CONV_AFFINE': '1'}),
                      ('det_unfused', {'B2M_DETERMINISTIC': '1', 'B2M_CONV_AFFINE': '0'}),
                      ('fused', {'B2M_CONV_AFFINE': '1'}), ('unfused', {'B2M_CONV_AFFINE': '0'})):
        for k in ('B2M_DETERMINISTIC', 'B2M_CONV_AFFINE'):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        calls = []
        _lib.set_hook(lambda name, a, meta=None: calls.append(name))
        try:
            pred = model.get_prediction(batch, with_grad=False, to_cpu=True, min_size=False)
        finally:
            _lib.set_hook(None)
        res[mode] = (pred, _names(calls))
    # the fused pass has no BatchNorm launch of its own left in the trunk: what remains are the heads' ReLU -> BatchNorm
    # layers (2 per head) and the few layers whose tiny maps are combined with atomics
    nf, nu = res['det_fused'][1], res['det_unfused'][1]
    assert nf.get('b2m_conv_fwd_affine', 0) >= 60 and nu.get('b2m_conv_fwd_affine', 0) == 0
    assert nf.get('b2m_bn_apply', 0) + nf.get('b2m_bn_apply2', 0) <= 2 * len(cfg.network_heads)
    assert nu.get('b2m_bn_apply', 0) + nu.get('b2m_bn_apply2', 0) >= 60
    for h in cfg.network_heads:
        assert torch.equal(res['det_fused'][0][h], res['det_unfused'][0][h]), h
        assert _rel(res['fused'][0][h], res['unfused'][0][h]) < 1e-5, h
        assert _rel(res['fused'][0][h], res['det_fused'][0][h]) < 1e-5, h
    ref = unet_ref.forward(p_cpu, batch['vox_coords'].numpy(), batch['vox_features'], batch['pooling_ids'], cfg,
                           training=False, n_segments=batch['input_location'].shape[0])
    for h in cfg.network_heads:
        e = _rel(res['fused'][0][h], ref[h])
        assert e < 1e-3, (h, e)


def test_next_scene_prefetched_gives_the_same_outputs(monkeypatch):
    """An evaluation loop one scene ahead: Model.prefetch(scene) builds the scene's sparse tensor and every map on a second stream,
    Model.get_prediction(scene) takes them -- the same outputs, bit for bit, as the pass that builds its maps itself, and a
    prefetch made for ANOTHER scene is not taken (evaluation.py:70-98 with a data loader that runs ahead)."""
    monkeypatch.setenv('B2M_DETERMINISTIC', '1')
    model, batch, cfg = _model_and_batch()
    other = _model_and_batch(n_vox=5000, bs=1, seed=9)[1]
    plain = model.get_prediction(batch, with_grad=False, to_cpu=True, min_size=True)
    model.prefetch(batch, ready=True, loss_rows=False)
    assert model._prefetched is not None
    ahead = model.get_prediction(batch, with_grad=False, to_cpu=True, min_size=True)
    assert model._prefetched is None                       # consumed
    model.prefetch(other, ready=True, loss_rows=False)     # a scene the next call is NOT about
    again = model.get_prediction(batch, with_grad=False, to_cpu=True, min_size=True)
    assert model._prefetched is not None                   # ... and it stays for the call it was made for
    model.get_prediction(other, with_grad=False, to_cpu=True, min_size=True)
    assert model._prefetched is None
    for h in cfg.network_heads:
        assert torch.equal(plain[h], ahead[h]), h
        assert torch.equal(plain[h], again[h]), h


def test_eval_affine_cache_follows_the_parameters():
    """The cached (scale, shift) of a layer is rebuilt after load_state_dict / an in-place change of the statistics."""
    from box2mask_amd import nn as ME
    bn = ME.MinkowskiBatchNorm(32).cuda().eval()
    with torch.no_grad():
        s0, b0 = bn.eval_affine()
        assert bn.eval_affine()[0] is s0                          # cached
        bn.bn.running_var.fill_(4.0)
        s1, _ = bn.eval_affine()
        assert s1 is not s0 and torch.allclose(s1, torch.full_like(s1, 1.0 / np.sqrt(4.0 + bn.bn.eps)))
        sd = {k: v.clone() for k, v in bn.state_dict().items()}
        sd['bn.weight'] = torch.full_like(sd['bn.weight'], 2.0)
        bn.load_state_dict(sd)
        s2, _ = bn.eval_affine()
        assert torch.allclose(s2, 2.0 * s1)


def test_eval_between_training_steps_sees_the_new_statistics(monkeypatch):
    """train -> eval -> train -> eval: the second inference pass must run on the running statistics and parameters as the
    training passes in between left them.  This package's BatchNorm kernels update the running statistics through raw
    pointers and a fused optimizer steps the parameters without bumping version counters -- the cached eval-mode affine maps
    (MinkowskiBatchNorm.eval_affine) are keyed on functional's training epoch as well.  Reference: the unfused inference path
    (B2M_CONV_AFFINE=0), which reads the running statistics afresh in every pass."""
    model, batch, cfg = _model_and_batch(n_vox=4000, bs=2)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)

    def train_steps(k):
        model.train()
        for _ in range(k):
            opt.zero_grad()
            model.compute_loss(batch, 150)['optimization_loss'].backward()
            opt.step()
        model.eval()

    def predict(fused):
        monkeypatch.setenv('B2M_CONV_AFFINE', '1' if fused else '0')
        return model.get_prediction(batch, with_grad=False, to_cpu=True, min_size=False)
    train_steps(1)
    first = predict(True)                       # builds the caches
    train_steps(3)
    fused, unfused = predict(True), predict(False)
    for h in cfg.network_heads:
        assert torch.isfinite(unfused[h]).all()
        assert _rel(fused[h], unfused[h]) < 1e-5, (h, _rel(fused[h], unfused[h]))
    assert max(_rel(fused[h], first[h]) for h in cfg.network_heads) > 1e-3      # the state did move


def test_eval_after_an_optimizer_step_without_a_training_pass_in_between(monkeypatch):
    """backward, EVAL pass, optimizer.step(), eval pass: the step belongs to a backward pass from before the first eval pass, so
    no training-mode pass separates the two inference passes -- the fused optimizer bumps no version counter, and the second
    pass must still run on the stepped weights (every torch optimizer step advances functional's training epoch).  The same
    for parameters written through `.data` by GradAllReduce.broadcast_parameters (here: its single-rank equivalent, a copy
    through `.data` followed by the invalidation it performs).  Reference: a fresh model loaded with the final state."""
    from box2mask_amd import functional as F_
    model, batch, cfg = _model_and_batch(n_vox=4000, bs=2)
    opt = torch.optim.Adam(model.parameters(), lr=5e-3, fused=True)
    predict = lambda m: m.get_prediction(batch, with_grad=False, to_cpu=True, min_size=False)
    model.train()
    opt.zero_grad()
    model.compute_loss(batch, 150)['optimization_loss'].backward()
    model.eval()
    before = predict(model)                      # caches built from the un-stepped weights
    opt.step()
    after = predict(model)
    fresh, _, _ = _model_and_batch(n_vox=4000, bs=2)
    fresh.load_state_dict({k: v.clone() for k, v in model.state_dict().items()})
    fresh.eval()
    want = predict(fresh)
    for h in cfg.network_heads:
        assert _rel(after[h], want[h]) < 1e-5, (h, _rel(after[h], want[h]))
    assert max(_rel(after[h], before[h]) for h in cfg.network_heads) > 1e-4          # the step did move the outputs
    # parameters replaced through .data (what a broadcast from another rank does), then the invalidation broadcast_parameters ends with
    torch.manual_seed(5)
    with torch.no_grad():
        for p in model.parameters():
            p.data.copy_(p.data + 0.05 * torch.randn_like(p.data) * p.data.abs().mean())
    F_.note_training_pass()
    moved = predict(model)
    fresh.load_state_dict({k: v.clone() for k, v in model.state_dict().items()})
    want = predict(fresh)
    for h in cfg.network_heads:
        assert _rel(moved[h], want[h]) < 1e-5, (h, _rel(moved[h], want[h]))


def test_inference_passes_follow_load_state_dict(monkeypatch):
    """Two inference passes with a load_state_dict in between (no training pass: the packed weight images of the first pass
    are kept by default) -- the second one runs on the loaded weights: the copies bump the parameters' version counters and
    the images are rebuilt at lookup.  Checked against a fresh model that never saw the old weights."""
    model, batch, cfg = _model_and_batch(n_vox=4000, bs=2)
    first = model.get_prediction(batch, with_grad=False, to_cpu=True, min_size=False)
    again = model.get_prediction(batch, with_grad=False, to_cpu=True, min_size=False)
    for h in cfg.network_heads:
        assert _rel(again[h], first[h]) < 1e-5
    torch.manual_seed(77)
    sd = {k: (v + 0.05 * torch.randn_like(v) if v.is_floating_point() and 'running_var' not in k and v.dim() > 0 else v.clone())
          for k, v in model.state_dict().items()}
    model.load_state_dict(sd)
    loaded = model.get_prediction(batch, with_grad=False, to_cpu=True, min_size=False)
    fresh, _, _ = _model_and_batch(n_vox=4000, bs=2)
    fresh.load_state_dict(sd)
    fresh.eval()
    ref = fresh.get_prediction(batch, with_grad=False, to_cpu=True, min_size=False)
    for h in cfg.network_heads:
        assert _rel(loaded[h], ref[h]) < 1e-5, (h, _rel(loaded[h], ref[h]))
    assert max(_rel(loaded[h], first[h]) for h in cfg.network_heads) > 1e-3


def test_validation_losses_between_training_steps_like_the_reference_trainer(monkeypatch):
    """/root/reference/models/training.py:64-66, 263-284: `model.train(); compute_loss; backward; step` with, every so often,
    `model.eval(); with torch.no_grad(): compute_loss(val_batch)`.  The validation passes run the fused inference layers; their
    losses must be those of the unfused layering (B2M_CONV_AFFINE=0) on the model's CURRENT state at every validation."""
    model, batch, cfg = _model_and_batch(n_vox=4000, bs=2)
    val_batch = _model_and_batch(n_vox=3000, bs=2, seed=9)[1]
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)

    def val(fused):
        monkeypatch.setenv('B2M_CONV_AFFINE', '1' if fused else '0')
        model.eval()
        with torch.no_grad():
            return {k: float(v) for k, v in model.compute_loss(val_batch, 150).items() if hasattr(v, 'item') or isinstance(v, float)}
    seen = []
    for it in range(3):
        model.train()
        opt.zero_grad()
        model.compute_loss(batch, 150)['optimization_loss'].backward()
        opt.step()
        a, b = val(True), val(False)
        for k in a:
            assert np.isfinite(a[k]) and abs(a[k] - b[k]) <= 1e-4 * max(1.0, abs(b[k])), (it, k, a[k], b[k])
        seen.append(a['optimization_loss'])
    assert len(set(round(v, 6) for v in seen)) == 3           # the state moved between the validations
