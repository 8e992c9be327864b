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
    from _parity import row_rel_err, ROW_FACTOR
    rerr = row_rel_err(a, b)              # rows of small magnitude count with their own scale (tests/_parity.py)
    assert rerr < ROW_FACTOR * tol, '%s: max per-row relative error %.3e' % (what, rerr)
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


@pytest.mark.parametrize('n', [5000, 700, 9])          # two-stage kernels | one-launch small-map kernels (functional.bn_small_rows)
@pytest.mark.parametrize('c,relu,res', [(32, True, False), (96, True, True), (96, False, False), (256, True, False)])
def test_batch_norm(c, relu, res, n):
    from box2mask_amd import functional as F_
    assert (n <= F_.bn_small_rows()) == (n < 5000)
    torch.manual_seed(c)
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
    # eval-mode gradients, affine parameters included (torch.nn.BatchNorm1d has them)
    bn.zero_grad(); g.zero_grad()
    xo2 = x.clone().requires_grad_(True); xg2 = x.cuda().requires_grad_(True)
    yo2 = torch.relu(bn(xo2)) if relu else bn(xo2)
    yo2.backward(gy)
    F_.batch_norm(xg2, g.weight, g.bias, g.running_mean, g.running_var, False, 0.1, 1e-5, None, relu, False).backward(gy.cuda())
    _close(xg2.grad, xo2.grad, 'bn eval dx', 1e-5)
    _close(g.weight.grad, bn.weight.grad, 'bn eval dgamma', 1e-4); _close(g.bias.grad, bn.bias.grad, 'bn eval dbeta', 1e-4)


def test_batch_norm_small_and_two_stage_kernels_agree(monkeypatch):
    """The one-launch BatchNorm of small maps against the two-stage kernels on the same tensors (B2M_BN_SMALL_ROWS=0),
    with residual + ReLU, at the row counts of levels 4-7 and of the heads."""
    from box2mask_amd import functional as F_
    for n, c in ((3214, 256), (674, 256), (109, 256), (2, 256), (9752, 96), (4096, 32)):
        outs = []
        for small in (1, 0):
            monkeypatch.setenv('B2M_BN_SMALL_ROWS', '16384' if small else '0')
            torch.manual_seed(n)
            x = (torch.randn(n, c, device='cuda') * 1.7 + 0.3).requires_grad_(True)
            r = torch.randn(n, c, device='cuda').requires_grad_(True)
            gam = (torch.rand(c, device='cuda') + 0.5).requires_grad_(True); bet = torch.randn(c, device='cuda').requires_grad_(True)
            rm, rv = torch.zeros(c, device='cuda'), torch.ones(c, device='cuda')
            y = F_.batch_norm(x, gam, bet, rm, rv, True, 0.1, 1e-5, r, True, False)
            y.backward(torch.randn(n, c, device='cuda', generator=torch.Generator('cuda').manual_seed(1)))
            outs.append((y.detach(), x.grad, r.grad, gam.grad, bet.grad, rm, rv))
        for a, b, what in zip(outs[0], outs[1], ('y', 'dx', 'dres', 'dgamma', 'dbeta', 'running_mean', 'running_var')):
            if n == 2 and what in ('dx', 'dgamma'):
                continue          # two rows: x_hat = +-1 exactly, dx is a difference of equal numbers (pure rounding noise)
            _close(a, b, 'n=%d %s' % (n, what), 2e-5)


@pytest.mark.parametrize('n,c,res,relu', [(3214, 256, True, True), (109, 256, False, True), (9752, 96, False, False), (2, 32, True, True)])
def test_batch_norm_small_syncbn_halves_equal_one_launch(n, c, res, relu):
    """SyncBN on a small map runs the one-launch kernels cut in two around the statistics exchange (b2m_bn_small_fwd_stats ->
    all-reduce -> b2m_bn_small_fwd_apply; b2m_bn_small_bwd_phase 1 -> all-reduce -> 2).  With one rank the exchange is the
    identity, so both halves together must give the bits of the one-launch kernel."""
    from box2mask_amd import functional as F_
    call, ptr = F_._call, F_._ptr
    torch.manual_seed(n + c)
    x = torch.randn(n, c, device='cuda') * 1.3 + 0.2
    r = torch.randn(n, c, device='cuda') if res else None
    gam = torch.rand(c, device='cuda') + 0.5; bet = torch.randn(c, device='cuda')
    dy = torch.randn(n, c, device='cuda')
    f32 = lambda: torch.empty(c, device='cuda')

    def forward(halves):
        rm, rv = torch.zeros(c, device='cuda'), torch.ones(c, device='cuda')
        mean, inv, sc, sh = f32(), f32(), f32(), f32()
        y = torch.empty_like(x)
        tail = (ptr(gam), ptr(bet), 1e-5, 0.1, rm.data_ptr(), rv.data_ptr(), mean.data_ptr(), inv.data_ptr(), sc.data_ptr(),
                sh.data_ptr(), ptr(r), r.stride(0) if r is not None else 0, 1 if relu else 0, y.data_ptr(), y.stride(0))
        xchg = None
        if halves:
            xchg = torch.empty(2 * c + 1, dtype=torch.float64, device='cuda')
            call('b2m_bn_small_fwd_stats', x.data_ptr(), x.stride(0), n, c, xchg.data_ptr())
            call('b2m_bn_small_fwd_apply', xchg.data_ptr(), x.data_ptr(), x.stride(0), n, c, *tail)
        else:
            call('b2m_bn_small_fwd', x.data_ptr(), x.stride(0), n, c, *tail)
        return y, rm, rv, mean, inv, sc, sh, xchg
    a, b = forward(True), forward(False)
    for u, v, what in zip(a[:7], b[:7], ('y', 'running_mean', 'running_var', 'mean', 'invstd', 'scale', 'shift')):
        assert torch.equal(u, v), what
    assert float(a[7][2 * c]) == n
    y, mean, inv, sc, sh = b[0], b[3], b[4], b[5], b[6]
    use_y = relu and res                   # (without a residual the mask is recomputed from x)

    def backward(halves):
        dx, dres = torch.empty_like(x), (torch.empty_like(x) if res else None)
        dbeta, dgamma = f32(), f32()
        head = (dy.data_ptr(), dy.stride(0), y.data_ptr() if use_y else None, y.stride(0) if use_y else 0, x.data_ptr(), x.stride(0), n, c,
                mean.data_ptr(), inv.data_ptr(), ptr(gam), 1 if relu else 0, None if use_y or not relu else sc.data_ptr(),
                None if use_y or not relu else sh.data_ptr())
        if halves:
            xchg = torch.empty(2 * c, dtype=torch.float64, device='cuda')
            cnt = torch.full((1,), float(n), dtype=torch.float64, device='cuda')
            call('b2m_bn_small_bwd_phase', 1, *head, dbeta.data_ptr(), dgamma.data_ptr(), None, 0, None, 0, xchg.data_ptr(), None)
            call('b2m_bn_small_bwd_phase', 2, *head, None, None, dx.data_ptr(), dx.stride(0), ptr(dres),
                 dres.stride(0) if dres is not None else 0, xchg.data_ptr(), cnt.data_ptr())
        else:
            call('b2m_bn_small_bwd', *head, dbeta.data_ptr(), dgamma.data_ptr(), dx.data_ptr(), dx.stride(0), ptr(dres),
                 dres.stride(0) if dres is not None else 0)
        return dx, dres, dbeta, dgamma
    ga, gb = backward(True), backward(False)
    for u, v, what in zip(ga, gb, ('dx', 'dres', 'dbeta', 'dgamma')):
        assert (u is None and v is None) or torch.equal(u, v), what


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


@pytest.mark.parametrize('n,c', [(20000, 96), (5000, 256), (300, 64)])
def test_batch_norm_pair_equals_two_batch_norms(n, c, monkeypatch):
    """relu(BN_a(x_a) + BN_b(x_b)) as one paired operator (functional._BatchNormPair: the end of a BasicBlock with a shortcut
    convolution, resnet.py:73-82) against the two separate BatchNorm operators: the forward bit for bit (same fma / fma /
    add), the gradients to rounding; train and eval mode."""
    from box2mask_amd import functional as F_
    monkeypatch.setenv('B2M_BN_SMALL_ROWS', '0')
    gen = torch.Generator('cuda').manual_seed(n + c)
    rnd = lambda *s: torch.randn(*s, device='cuda', generator=gen)
    xa0, xb0, gy = rnd(n, c) * 1.5 + 0.2, rnd(n, c) * 0.7 - 0.4, rnd(n, c)
    for training in (True, False):
        res = []
        for paired in (True, False):
            xa, xb = xa0.clone().requires_grad_(True), xb0.clone().requires_grad_(True)
            prm = [(torch.rand(c, device='cuda', generator=gen) + 0.5).requires_grad_(True) for _ in range(2)] + \
                  [rnd(c).requires_grad_(True) for _ in range(2)]
            gen.manual_seed(7)          # (same parameters for both forms)
            with torch.no_grad():
                for t in prm:
                    t.copy_(torch.rand(c, device='cuda', generator=gen) + 0.5)
            ga, gb, ba, bb = prm
            rma, rva, rmb, rvb = rnd(c) * 0.1, torch.rand(c, device='cuda', generator=gen) + 0.5, rnd(c) * 0.1, \
                torch.rand(c, device='cuda', generator=gen) + 0.5
            gen.manual_seed(n + c + 1)
            if paired:
                y = F_.batch_norm_pair(xa, (ga, ba, rma, rva, 0.1, 1e-5), xb, (gb, bb, rmb, rvb, 0.1, 1e-5), training, True, False)
            else:
                r = F_.batch_norm(xb, gb, bb, rmb, rvb, training, 0.1, 1e-5, None, False, False)
                y = F_.batch_norm(xa, ga, ba, rma, rva, training, 0.1, 1e-5, r, True, False)
            y.backward(gy)
            res.append((y.detach(), xa.grad, xb.grad, ga.grad, ba.grad, gb.grad, bb.grad, rma, rva, rmb, rvb))
        assert torch.equal(res[0][0], res[1][0]), 'paired forward differs from the two launches'
        for a, b, what in zip(res[0][1:], res[1][1:], ('dxa', 'dxb', 'dgamma_a', 'dbeta_a', 'dgamma_b', 'dbeta_b', 'rm_a', 'rv_a', 'rm_b', 'rv_b')):
            _close(a, b, '%s training=%s' % (what, training), 2e-6)


@pytest.mark.parametrize('cin,cout,n', [(96, 3, 9752), (96, 1, 9752), (96, 20, 9752), (256, 13, 700), (320, 28, 1500), (32, 6, 5)])
def test_narrow_weight_gradient_of_a_1x1_layer(cin, cout, n, monkeypatch):
    """wgrad_narrow_kernel (1x1 layers with few output channels: the heads' last layers) against x^T dy, accumulating onto
    a non-zero dW, incl. more than 256 input channels (two rounds) and fewer rows than a workgroup's chunk; and against the
    MFMA kernels it replaces (B2M_WGRAD_NARROW=0)."""
    from box2mask_amd import functional as F_
    torch.manual_seed(cin * 100 + cout)
    x = torch.randn(n, cin, device='cuda'); dy = torch.randn(n, cout, device='cuda')
    dw0 = torch.randn(1, cin, cout, device='cuda')
    ref = dw0[0].double() + x.double().t() @ dy.double()
    outs = []
    for flag in ('1', '0'):
        monkeypatch.setenv('B2M_WGRAD_NARROW', flag)
        dw = dw0.clone()
        F_.wgrad_raw(x, dy, None, 1, dw, 0)
        _close(dw[0], ref, 'narrow wgrad %dx%d (B2M_WGRAD_NARROW=%s)' % (cin, cout, flag), 1e-5)
        outs.append(dw)
    _close(outs[0], outs[1], 'narrow vs MFMA kernel', 1e-5)
