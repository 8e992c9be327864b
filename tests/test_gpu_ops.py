"""GPU parity of the individual operators (conv fwd / dgrad / wgrad, BN, pooling) against the oracle."""
import zlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 1e-3        # north_star: conv features within 1e-3 fp32


def _close(a, b, what, tol=TOL):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(float(b.abs().max()), 1e-6)
    err = float((a - b).abs().max()) / scale
    assert err < tol, '%s: max rel-to-max error %.3e' % (what, err)
    return err


def _scene(n_target=6000, seed=0, bs=2):
    from box2mask_amd import synth
    return synth.make_batch(bs, seed0=seed, target_voxels=n_target, pts_per_m2=6000.0)


@pytest.fixture(scope='module')
def maps():
    from box2mask_amd.sparse import CoordinateManager
    from oracle import sparse_ref as S
    b = _scene()
    coords = b['vox_coords'].numpy()
    m = CoordinateManager(b['vox_coords'])
    h = S.Hierarchy(coords, n_levels=3)
    m.ensure_level(2)
    return m, h


CONV_CASES = [
    # (kind, level, cin (c1,c2), cout, bias)
    ('k3', 0, (32, 0), 32, False), ('k3', 0, (96, 0), 96, False), ('k3', 0, (96, 32), 96, False),
    ('k3', 1, (64, 0), 128, False), ('k3', 1, (256, 128), 256, False), ('k5', 0, (6, 0), 32, False),
    ('down', 0, (32, 0), 32, False), ('down', 1, (64, 0), 64, False), ('up', 0, (96, 0), 96, False),
    ('up', 1, (128, 0), 96, False), ('1x1', 0, (128, 0), 96, False), ('1x1', 0, (96, 32), 96, False),
    ('1x1', 1, (96, 0), 3, True), ('1x1', 1, (96, 0), 1, True), ('1x1', 1, (96, 0), 20, True),
    ('1x1', 1, (96, 0), 13, True),
]


@pytest.mark.parametrize('kind,level,cins,cout,bias', CONV_CASES)
def test_conv_forward_backward(maps, kind, level, cins, cout, bias):
    _conv_case(maps, kind, level, cins, cout, bias)


@pytest.mark.parametrize('kind,level,cins,cout,bias', [CONV_CASES[1], CONV_CASES[2], CONV_CASES[4], CONV_CASES[5],
                                                      CONV_CASES[8], CONV_CASES[12]])
def test_conv_64bit_addressing(maps, monkeypatch, kind, level, cins, cout, bias):
    """Operands beyond 4 GiB / 2^24 rows take 64-bit address arithmetic inside the kernels; the switch is decided
    per launch, so force it here on small inputs (the C library reads the variables at every call)."""
    monkeypatch.setenv('B2M_CONV_FAST32', '0')
    monkeypatch.setenv('B2M_WGRAD_FAST32', '0')
    _conv_case(maps, kind, level, cins, cout, bias)


def _conv_case(maps, kind, level, cins, cout, bias):
    from box2mask_amd import functional as F_
    from oracle import sparse_ref as S
    m, h = maps
    c1, c2 = cins
    cin = c1 + c2
    torch.manual_seed(zlib.crc32(repr((kind, level, cin, cout)).encode()) % 1000)     # stable across processes
    if kind == 'k3':
        rb_f = rb_b = m.rulebook_same(level, 3); nbr = h.k3(level); K = 27; mirror = True; n_in = n_out = h.n(level)
    elif kind == 'k5':
        rb_f = rb_b = m.rulebook_same(level, 5); nbr = h.k_first(); K = 125; mirror = True; n_in = n_out = h.n(level)
    elif kind == 'down':
        rb_f, rb_b = m.rulebook_down(level), m.rulebook_up(level); nbr = h.down(level); K = 8; mirror = False
        n_in, n_out = h.n(level), h.n(level + 1)
    elif kind == 'up':
        rb_f, rb_b = m.rulebook_up(level), m.rulebook_down(level); nbr = h.up(level); K = 8; mirror = False
        n_in, n_out = h.n(level + 1), h.n(level)
    else:
        rb_f = rb_b = None; nbr = None; K = 1; mirror = False; n_in = n_out = h.n(level)
    x = torch.randn(n_in, cin)
    w = torch.randn(K, cin, cout) / (cin * min(K, 10)) ** 0.5 if K > 1 else torch.randn(cin, cout) / cin ** 0.5
    b = torch.randn(1, cout) if bias else None
    gy = torch.randn(n_out, cout)
    # oracle
    xo, wo = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    bo = b.clone().requires_grad_(True) if bias else None
    yo = S.conv_nbr(xo, wo, nbr, bo)
    yo.backward(gy)
    # product
    xg = x.cuda().requires_grad_(True); wg = w.cuda().requires_grad_(True)
    bg = b.cuda().requires_grad_(True) if bias else None
    if c2:
        x1, x2 = xg[:, :c1].contiguous(), xg[:, c1:].contiguous()
    else:
        x1, x2 = xg, None
    yg = F_.sparse_conv(x1, x2, wg, bg, rb_f, rb_b, mirror, n_out)
    yg.backward(gy.cuda())
    torch.cuda.synchronize()
    _close(yg, yo, 'forward')
    _close(xg.grad, xo.grad, 'dgrad')
    _close(wg.grad, wo.grad, 'wgrad')
    if bias:
        _close(bg.grad, bo.grad, 'bias grad')


def test_conv_accumulate_and_empty_tiles(maps):
    """Rows without any neighbour (isolated voxels) and accumulate=1."""
    from box2mask_amd import functional as F_
    from box2mask_amd.sparse import CoordinateManager
    from oracle import sparse_ref as S
    c = np.array([[0, 10 * i, 7 * (i % 5), 3 * (i % 7)] for i in range(300)], np.int32)
    c = np.unique(c, axis=0)
    m = CoordinateManager(torch.from_numpy(c))
    rb = m.rulebook_same(0, 3)
    x = torch.randn(len(c), 32); w = torch.randn(27, 32, 64) * 0.1
    y0 = torch.randn(len(c), 64)
    ref = S.conv_nbr(x, w, S.kernel_map_same(c, 3, 1)) + y0
    out = y0.cuda().clone()
    F_.conv_raw(x.cuda(), None, F_.weight_pack(w.cuda()), 27, None, rb, len(c), 64, out=out, accumulate=True)
    _close(out, ref, 'accumulate')


@pytest.mark.parametrize('c,relu,res', [(32, True, False), (96, True, True), (96, False, False), (256, True, False)])
def test_batch_norm(c, relu, res):
    from box2mask_amd import functional as F_
    torch.manual_seed(c)
    n = 5000
    x = torch.randn(n, c) * 2 + 0.5
    r = torch.randn(n, c) if res else None
    gy = torch.randn(n, c)
    bn = torch.nn.BatchNorm1d(c)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.5, 0.5)
    xo = x.clone().requires_grad_(True); ro = r.clone().requires_grad_(True) if res else None
    yo = bn(xo)
    if res: yo = yo + ro
    if relu: yo = torch.relu(yo)
    yo.backward(gy)
    g = torch.nn.BatchNorm1d(c).cuda()
    with torch.no_grad():
        g.weight.copy_(bn.weight); g.bias.copy_(bn.bias)
    xg = x.cuda().requires_grad_(True); rg = r.cuda().requires_grad_(True) if res else None
    yg = F_.batch_norm(xg, g.weight, g.bias, g.running_mean, g.running_var, True, 0.1, 1e-5, rg, relu, False)
    yg.backward(gy.cuda())
    torch.cuda.synchronize()
    _close(yg, yo, 'bn fwd', 1e-5); _close(xg.grad, xo.grad, 'bn dx', 1e-4)
    _close(g.weight.grad, bn.weight.grad, 'dgamma', 1e-4); _close(g.bias.grad, bn.bias.grad, 'dbeta', 1e-4)
    _close(g.running_mean, bn.running_mean, 'running_mean', 1e-5); _close(g.running_var, bn.running_var, 'running_var', 1e-5)
    if res: _close(rg.grad, ro.grad, 'dres', 1e-5)
    # eval mode
    bn.eval()
    ye = bn(x)
    yge = F_.batch_norm(x.cuda(), g.weight, g.bias, g.running_mean, g.running_var, False, 0.1, 1e-5, None, False, False)
    _close(yge, ye, 'bn eval', 1e-5)


@pytest.mark.parametrize('mode', ['avg', 'max'])
def test_segment_pool(mode):
    from box2mask_amd import functional as F_
    from oracle import sparse_ref as S
    torch.manual_seed(3)
    n, c, s = 7000, 96, 211
    ids = torch.randint(0, s, (n,)); ids[:s] = torch.arange(s)
    x = torch.randn(n, c); gy = torch.randn(s, c)
    xo = x.clone().requires_grad_(True)
    yo = S.segment_pool(xo, ids, s, mode); yo.backward(gy)
    xg = x.cuda().requires_grad_(True)
    yg = F_.segment_pool(xg, ids.cuda(), s, mode); yg.backward(gy.cuda())
    _close(yg, yo, 'pool fwd', 1e-5); _close(xg.grad, xo.grad, 'pool bwd', 1e-5)


def test_relu_and_set_ious():
    from box2mask_amd import functional as F_
    x = torch.randn(1000, 7)
    xg = x.cuda().requires_grad_(True)
    y = F_.relu(xg); y.backward(torch.ones_like(y))
    assert torch.equal(y.cpu(), torch.relu(x)) and torch.equal(xg.grad.cpu(), (x > 0).float())
