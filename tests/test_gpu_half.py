"""Half-precision inference trunk (BASELINE.json configs[4]: "ARKitScenes config ... fp16 features on CDNA4"; a build extension,
the reference is fp32-only): activations in HBM as IEEE half, f16 MFMA with fp32 accumulation, fp32 BatchNorm epilogue
(b2m_conv_fwd_h, conv_fwd_flow_kernel<.., F16>).

  1. every layer kind against the fp32 kernel on the SAME half-rounded operands (what is left is the summation order and the
     one rounding of the result to half: <= 1e-3 of the tensor's maximum, and 2^-10 relative element by element);
  2. the whole network: half trunk against the fp32 inference path and the CPU oracle (tolerance stated in the test: the
     half trunk rounds every activation of ~40 layers to 11 bits).
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
    # (kind, level, (c1, c2), cout): 32-channel chunks 2 / 3 steps in flight, 16-channel chunks; 48- and 32-column strips;
    # un-split maps and 4 slices per workgroup (level 2)
    ('k3', 0, (96, 0), 96), ('k3', 0, (96, 32), 96), ('k3', 0, (32, 0), 32), ('k3', 0, (64, 0), 64), ('k3', 1, (128, 0), 128),
    ('k3', 1, (64, 0), 128), ('k3', 2, (256, 0), 256), ('k3', 2, (256, 128), 256), ('k3', 2, (96, 0), 96), ('k3', 2, (32, 0), 96),
    ('down', 0, (32, 0), 32), ('down', 1, (96, 0), 96), ('up', 0, (96, 0), 96), ('up', 1, (256, 0), 128),
    ('1x1', 0, (128, 0), 96), ('1x1', 0, (32, 0), 64), ('1x1', 2, (96, 32), 128), ('1x1', 1, (96, 0), 32),
    ('k3', 0, (160, 0), 32),      # 5 chunks of 32: falls to 16-channel chunks
]


# every layer shape with the trunk's usual epilogue (BatchNorm + ReLU); the other epilogues -- residual, no ReLU, no affine map: code
# after the walk, the same for every shape -- on one case per kernel form (48- / 32-column strips, split map, k2s2 both ways, 1x1,
# 16-channel chunks).  Round 5: was 19 x 4.
_EPILOGUES = [(True, True, True), (False, False, True), (False, False, False)]
_EPILOGUE_CASES = [0, 1, 4, 6, 10, 12, 14, 18]


def _half_matrix():
    out = [pytest.param(*c, False, True, True) for c in CASES]
    out += [pytest.param(*CASES[i], *e) for i in _EPILOGUE_CASES for e in _EPILOGUES]
    return out


@pytest.mark.parametrize('kind,level,cins,cout,res,relu,affine', _half_matrix())
def test_half_layer_equals_fp32_kernel_on_half_rounded_operands(maps, monkeypatch, kind, level, cins, cout, res, relu, affine):
    from box2mask_amd import functional as F_
    m = maps
    c1, c2 = cins
    if kind == 'k3':
        rb = m.rulebook_same(level, 3); K = 27; n_in = n_out = m.n(level); rb32 = rb
    elif kind == 'down':
        rb = rb32 = m.rulebook_down(level); K = 8; n_in, n_out = m.n(level), m.n(level + 1)
    elif kind == 'up':
        rb = rb32 = m.rulebook_up(level); K = 8; n_in, n_out = m.n(level + 1), m.n(level)
    else:
        rb = m.rulebook_identity(level); rb32 = None; K = 1; n_in = n_out = m.n(level)
    torch.manual_seed(zlib.crc32(repr((kind, level, cins, cout)).encode()) % 1000)
    x1 = torch.randn(n_in, c1, device='cuda').half()
    x2 = torch.randn(n_in, c2, device='cuda').half() if c2 else None
    w = torch.randn(K, c1 + c2, cout, device='cuda') * (2.0 / ((c1 + c2) * min(K, 10)) ** 0.5)
    w = w.half().float().contiguous()                          # weights exactly representable in half
    if K == 1:
        w = w[0].contiguous()
    scale = (torch.rand(cout, device='cuda') + 0.5) if affine else None
    shift = torch.randn(cout, device='cuda') if affine else None
    r = torch.randn(n_out, cout, device='cuda').half() if res else None
    y = F_.conv_affine_h(x1, x2, w, rb, n_out, scale, shift, r, relu)
    assert y.dtype == torch.float16 and y.shape == (n_out, cout)
    # reference: the fp32 kernels on the same numbers
    one = torch.ones(cout, device='cuda'); zero = torch.zeros(cout, device='cuda')
    ref = F_.conv_affine(x1.float(), x2.float() if c2 else None, w, rb32, n_out, scale if affine else one, shift if affine else zero,
                         r.float() if res else None, relu)
    torch.cuda.synchronize()
    assert torch.isfinite(y.float()).all()
    e = _rel(y.float(), ref)
    assert e < 1e-3, 'half layer differs from the fp32 kernel on the same operands: %.3e' % e
    # element by element: one rounding to half (2^-11 relative, 2^-10 allowed) + the summation order of the fp32 accumulation
    d = (y.float() - ref).abs()
    bound = ref.abs() * 2.0 ** -10 + 3e-5 * float(ref.abs().max())
    assert bool((d <= bound).all()), float((d / bound).max())


def _model_and_batch(n_vox=9000, bs=3, seed=5):
    from test_gpu_inference import _model_and_batch as mb
    return mb(n_vox, bs, seed)


def test_half_trunk_network_against_fp32_inference_and_oracle():
    from box2mask_amd import _lib
    from oracle import unet_ref
    model, batch, cfg = _model_and_batch()
    p_cpu = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    p32 = model.get_prediction(batch, with_grad=False, to_cpu=True, min_size=False)
    calls = []
    model.detection_model.half_trunk = True
    _lib.set_hook(lambda name, a, meta=None: calls.append(name))
    try:
        p16 = model.get_prediction(batch, with_grad=False, to_cpu=True, min_size=False)
    finally:
        _lib.set_hook(None)
        model.detection_model.half_trunk = False
    n_h = sum(1 for c in calls if c == 'b2m_conv_fwd_h')
    n_a = sum(1 for c in calls if c == 'b2m_conv_fwd_affine')
    assert n_h >= 75 and n_a == 1, (n_h, n_a)                 # every trunk layer but the 6-channel stem runs in half
    ref = unet_ref.forward(p_cpu, batch['vox_coords'].numpy(), batch['vox_features'], batch['pooling_ids'], cfg,
                           training=False, n_segments=batch['input_location'].shape[0])
    worst = 0.0
    for h in cfg.network_heads:
        assert p16[h].dtype == torch.float32 and torch.isfinite(p16[h]).all()
        e32, eo = _rel(p16[h], p32[h]), _rel(p16[h], ref[h])
        worst = max(worst, e32, eo)
        # ~40 layers, each rounding its activations to half (2^-11 relative): observed 2e-3 .. 6e-3 of the head's maximum
        assert e32 < 2e-2 and eo < 2e-2, (h, e32, eo)
    print('half trunk vs fp32 inference / oracle: worst rel-to-max %.2e' % worst)
    # ... and the fp32 path is untouched by the switch
    p32b = model.get_prediction(batch, with_grad=False, to_cpu=True, min_size=False)
    for h in cfg.network_heads:
        assert _rel(p32b[h], p32[h]) < 1e-5


def test_half_trunk_refuses_training_mode():
    model, batch, cfg = _model_and_batch(n_vox=3000, bs=1)
    model.detection_model.half_trunk = True
    try:
        model.detection_model.train()
        with pytest.raises(RuntimeError):
            model.get_prediction(batch, with_grad=True, to_cpu=True, min_size=False)
    finally:
        model.detection_model.half_trunk = False
        model.eval()


def test_half_weight_image_follows_the_parameters():
    from box2mask_amd import functional as F_
    w = torch.nn.Parameter(torch.randn(27, 32, 32, device='cuda'))
    a = F_.weight_pack_h(w, 32, 0)
    assert F_.weight_pack_h(w, 32, 0) is a
    with torch.no_grad():
        w.mul_(2.0)
    b = F_.weight_pack_h(w, 32, 0)
    torch.cuda.synchronize()
    assert b is not a and torch.equal(b.float(), 2.0 * a.float())


def test_half_trunk_on_the_arkit_and_s3dis_configurations():
    """BASELINE configs[4] as written: 4 cm voxels, batch 4, 28 classes, features arriving as fp16, mixed scene sizes -- and the
    S3DIS configuration, whose per-voxel head reads the trunk's (converted) output on every voxel: half trunk against the
    fp32 inference path of the same model, instance masks from both."""
    from test_gpu_configs import _arkit_tables, _s3dis_cfg, _s3dis_tables
    from box2mask_amd import synth
    from box2mask_amd.config import scannet_config
    from box2mask_amd.model import Model
    # ---- ARKit-shaped
    cfg = scannet_config(eval_ths=[0.5, 0.05, 0.4, 0.6], loss_weight_bb_scores=3.0, loss_weight_semantics=0.3, voxel_size=0.04, batch_size=4)
    torch.manual_seed(1)
    model = Model(cfg, *_arkit_tables())
    items = [synth.make_scene(10 + i, target_voxels=tv, voxel_size=0.04, pts_per_m2=5000.0) for i, tv in enumerate([8000, 30000, 15000, 50000])]
    batch = synth.collate(items)
    batch['vox_features'] = batch['vox_features'].half()               # "fp16 features"
    model.eval()
    p32 = model.get_prediction(batch)
    model.detection_model.half_trunk = True
    try:
        p16 = model.get_prediction(batch)
        res = model.pred2mask(batch, p16, 'eval')
    finally:
        model.detection_model.half_trunk = False
    assert set(res) == {s['name'] for s in batch['scene']}
    for h in p32:
        assert p16[h].dtype == torch.float32 and p16[h].shape == p32[h].shape
        assert _rel(p16[h], p32[h]) < 2e-2, (h, _rel(p16[h], p32[h]))
    # ---- S3DIS-shaped: per-voxel semantics head
    cfg = _s3dis_cfg()
    torch.manual_seed(0)
    model = Model(cfg, *_s3dis_tables())
    one = synth.collate([synth.make_scene(0, target_voxels=60_000)], mode='test')
    model.eval()
    p32 = model.get_prediction(one)
    model.detection_model.half_trunk = True
    try:
        p16 = model.get_prediction(one)
    finally:
        model.detection_model.half_trunk = False
    assert p16['mlp_per_vox_semantics'].shape == (one['vox_coords'].shape[0], 13)
    for h in p32:
        assert _rel(p16[h], p32[h]) < 2e-2, (h, _rel(p16[h], p32[h]))
    # the per-voxel class decisions: equal wherever the fp32 margin between the two best classes is not a rounding matter
    a, b = p16['mlp_per_vox_semantics'], p32['mlp_per_vox_semantics']
    top2 = b.topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 2e-2 * float(b.abs().max())
    assert bool((a.argmax(1)[clear] == b.argmax(1)[clear]).all()) and float(clear.float().mean()) > 0.5


def test_cfg_half_inference_switches_only_the_inference_passes():
    """cfg.half_inference: get_prediction runs the half trunk, a training step of the same model runs fp32 as before."""
    from box2mask_amd import _lib, synth
    from box2mask_amd.config import scannet_config
    from box2mask_amd.model import Model
    torch.manual_seed(2)
    model = Model(scannet_config(half_inference=True), *synth.scannet_tables(), device='cuda:0')
    batch = synth.make_batch(2, seed0=9, target_voxels=4000, pts_per_m2=6000.0)
    calls = []
    _lib.set_hook(lambda name, a, meta=None: calls.append(name))
    try:
        model.train()
        losses = model.compute_loss(batch, 150)
        losses['optimization_loss'].backward()
        n_train_h = calls.count('b2m_conv_fwd_h')
        model.eval()
        pred = model.get_prediction(batch)
    finally:
        _lib.set_hook(None)
    assert n_train_h == 0 and calls.count('b2m_conv_fwd_h') >= 75
    assert np.isfinite(losses['optimization_loss'].item()) and all(torch.isfinite(v).all() for v in pred.values())
    # a validation pass after further training sees the UPDATED weights and running statistics (neither a fused optimizer nor this
    # package's BatchNorm kernels bump torch's version counters: every training pass advances functional's training epoch)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
    model.train()
    for _ in range(3):
        opt.zero_grad()
        model.compute_loss(batch, 150)['optimization_loss'].backward()
        opt.step()
    model.eval()
    pred2 = model.get_prediction(batch)
    model.detection_model.half_trunk = False
    pred2_f32 = model.get_prediction(batch)
    for h in pred:
        assert torch.isfinite(pred2_f32[h]).all()
        assert _rel(pred2[h], pred2_f32[h]) < 2e-2, h                  # the half pass follows the fp32 pass of the NEW state
    assert max(_rel(pred2[h], pred[h]) for h in pred) > 1e-2             # ... which differs from the old one


def test_half_layer_isolated_voxels_wide_pitches_and_empty_map():
    """Rows without any neighbour (an offset's pair list empty in most tiles), operands that are column windows of wider
    tensors (row pitch > channels), an output written into a window -- and a map without rows."""
    from box2mask_amd import functional as F_
    from box2mask_amd.sparse import CoordinateManager
    c = np.array([[0, 10 * i, 7 * (i % 5), 3 * (i % 7)] for i in range(300)], np.int32)
    c = np.unique(c, axis=0)
    m = CoordinateManager(torch.from_numpy(c))
    rb = m.rulebook_same(0, 3)
    n = len(c)
    torch.manual_seed(4)
    wide = torch.randn(n, 160, device='cuda').half()
    x1, x2 = wide[:, :96], wide[:, 96:128]                       # pitch 160 halfs, 16-byte aligned windows
    w = (torch.randn(27, 128, 64, device='cuda') * 0.1).half().float()
    res_wide = torch.randn(n, 72, device='cuda').half()
    res = res_wide[:, 8:]                                        # pitch 72, window at an 16-byte offset
    scale = torch.rand(64, device='cuda') + 0.5; shift = torch.randn(64, device='cuda')
    y = F_.conv_affine_h(x1, x2, w, rb, n, scale, shift, res, True)
    ref = F_.conv_affine(x1.float().contiguous(), x2.float().contiguous(), w, rb, n, scale, shift, res.float().contiguous(), True)
    torch.cuda.synchronize()
    assert _rel(y.float(), ref) < 1e-3
    # isolated rows see only the centre offset: y = relu(x W[13] * scale + shift + res)
    t = torch.relu((torch.cat([x1, x2], 1).float() @ w[13]) * scale + shift + res.float())
    iso = torch.from_numpy(np.array([i for i in range(n) if i % 7 == 0])).cuda()
    assert _rel(y.float()[iso], t[iso]) < 2e-3
    # an empty map
    e = F_.conv_affine_h(torch.empty(0, 32, device='cuda', dtype=torch.float16), None, w[:, :32, :32].contiguous(),
                         CoordinateManager(torch.zeros((0, 4), dtype=torch.int32)).rulebook_same(0, 3), 0)
    assert e.shape == (0, 32) and e.dtype == torch.float16


def test_half_output_saturates_instead_of_overflowing(maps):
    from box2mask_amd import functional as F_
    m = maps
    rb = m.rulebook_same(2, 3); n = m.n(2)
    x = torch.full((n, 32), 200.0, device='cuda').half()
    w = torch.full((27, 32, 32), 8.0, device='cuda')
    y = F_.conv_affine_h(x, None, w, rb, n)                      # 200 * 8 * 32 * (>= 1 neighbour) = 51 200 per neighbour
    torch.cuda.synchronize()
    assert torch.isfinite(y.float()).all() and float(y.float().max()) == 65504.0
    y = F_.conv_affine_h(x, None, -w, rb, n)
    assert torch.isfinite(y.float()).all() and float(y.float().min()) == -65504.0
