"""Parity of the sparse convolution in EVERY dispatch regime of csrc/conv.hip, at the sizes the benchmark runs, and
against an independent dense reference.

  1. forced regimes on the 12 k-row maps of test_gpu_ops: the un-split vector-store path that carries levels 0/1 in
     the benchmark (B2M_CONV_TARGET=0), the atomic split-K combine, no chunk slices, the plain weight-gradient
     kernel, 64-tile chunks in the pipelined weight-gradient kernel (the 64-lane `live` walk incl. lane 63), the
     non-XCD workgroup order and XCD runs of equal tile counts (the default: runs of equal work);
  2. whole layers at benchmark size -- one 150 k-voxel scene (k3 96->96, 128(96|32)->96, k5 6->32, 1x1 128->96)
     and the 1.2 M-row batch of BASELINE configs[1] (k3 96->96) -- forward, data gradient, weight gradient
     against oracle/sparse_ref.conv_nbr_explicit;
  3. dense equivalence on RANDOM occupancy without the oracle: stride-1 maps == F.conv3d on the zero-filled
     grid read at the active sites, k2s2 == F.conv3d(stride=2), the transposed k2s2 == F.conv_transpose3d masked
     to the existing fine sites; gradients from torch autograd of the dense op
     (/root/reference/models/resnet.py:61-65, detection_net.py:37-135).
"""
import zlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

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


# ------------------------------------------------------------------ 1. forced dispatch regimes
REGIMES = {
    'unsplit': {'B2M_CONV_TARGET': '0'},
    'unsplit_compiler_tracked_loads': {'B2M_CONV_TARGET': '0', 'B2M_CONV_HANDLOADS': '0'},
    'split_compiler_tracked_loads': {'B2M_CONV_HANDLOADS': '0'},
    'unsplit_32_column_strips': {'B2M_CONV_TARGET': '0', 'B2M_CONV_TW3': '0'},
    'atomic_combine': {'B2M_CONV_WGCOMBINE': '0'},
    'no_chunk_slices': {'B2M_CONV_CHUNKSPLIT': '0'},
    'many_slices': {'B2M_CONV_TARGET': '100000', 'B2M_CONV_MAXSLICE': '16'},
    # (round 6) un-split maps as two slices per item, workgroups of two waves (the form B2M_CONV_SPLIT2 gives the medium maps)
    'two_slices': {'B2M_CONV_TARGET': '0', 'B2M_CONV_SPLIT2': '100000000'},
    'wgrad_plain': {'B2M_WGRAD_PIPE': '0'},
    'wgrad_compiler_tracked_loads': {'B2M_WGRAD_HANDLOADS': '0'},
    'wgrad_hand_issued_loads_square_blocks_only': {'B2M_WGRAD_HANDLOADS': '1'},
    'wgrad_of_transposed_maps_over_the_up_rulebook': {'B2M_WGRAD_UP': '0'},     # (default: b2m_conv_wgrad_tr over the DOWN rulebook)
    'wgrad_64_tile_chunks': {'B2M_WGRAD_MIN_TILES': '64'},
    # the ordered two-stage combine: the plain kernel with the row roles exchanged on the transposed maps (b2m_conv_wgrad_tr)
    'deterministic': {'B2M_DETERMINISTIC': '1'},
    'deterministic_up_rulebook': {'B2M_DETERMINISTIC': '1', 'B2M_WGRAD_UP': '0'},
    'wgrad_one_offset_per_workgroup': {'B2M_WGRAD_KPACK': '0'},
    'no_xcd_order': {'B2M_XCD': '0'},
    'xcd_equal_tile_counts': {'B2M_XCD_BALANCE': '0'},
    'unsplit_64bit': {'B2M_CONV_TARGET': '0', 'B2M_CONV_FAST32': '0', 'B2M_WGRAD_FAST32': '0'},
}


@pytest.fixture(scope='module')
def maps():
    from test_gpu_ops import _scene
    from box2mask_amd.sparse import CoordinateManager
    from oracle import sparse_ref as S
    b = _scene()
    m = CoordinateManager(b['vox_coords'])
    h = S.Hierarchy(b['vox_coords'].numpy(), n_levels=3)
    m.ensure_level(2)
    return m, h


# CONV_CASES of test_gpu_ops a regime is run on -- the cases whose launches the switch can change (round 5: every regime
# used to run on the same nine cases, 144 tests of which a third repeated the default path under a switch their kernels do
# not read):   0 k3 32->32 | 1 k3 96->96 | 2 k3 (96|32)->96 | 3 k3 L1 64->128 | 4 k3 L1 (256|128)->256 | 5 k5 6->32 |
#              6 down 32->32 | 8 up 96->96 | 9 up L1 128->96
_ALL = [0, 1, 2, 3, 4, 5, 6, 8, 9]
_SPLIT = [1, 3, 4, 6, 9]            # switches of the split-K path: the level-1 maps (always split at this size) + one of each level-0 kind
_WGRAD = [0, 1, 2, 3, 5, 6, 8]      # weight-gradient switches: every block shape (2x2, 3x3, 4x4, two sources, the stem, K = 8)
_ORDER = [1, 2, 5, 8]               # dispatch-order switches: one case per rulebook kind
REGIME_CASES = {
    'unsplit': _ALL, 'unsplit_compiler_tracked_loads': _ALL,
    'split_compiler_tracked_loads': _SPLIT, 'unsplit_32_column_strips': [1, 2, 4, 8, 9], 'atomic_combine': _SPLIT,
    'no_chunk_slices': _SPLIT, 'many_slices': _SPLIT, 'two_slices': _ALL, 'wgrad_plain': _WGRAD, 'wgrad_compiler_tracked_loads': _WGRAD,
    'wgrad_hand_issued_loads_square_blocks_only': _WGRAD, 'wgrad_64_tile_chunks': _WGRAD,
    'wgrad_of_transposed_maps_over_the_up_rulebook': [8, 9], 'wgrad_one_offset_per_workgroup': [0, 5, 6],
    'deterministic': [1, 3, 6, 8, 9], 'deterministic_up_rulebook': [8, 9],
    'no_xcd_order': _ORDER, 'xcd_equal_tile_counts': _ORDER, 'unsplit_64bit': _ALL,
}


def _regime_matrix():
    from test_gpu_ops import CONV_CASES
    assert set(REGIME_CASES) == set(REGIMES)
    return [pytest.param(r, *CONV_CASES[i], id='%s-%s-L%d-%s-%d' % ((r,) + CONV_CASES[i][:2] + ('+'.join(map(str, CONV_CASES[i][2])), CONV_CASES[i][3])))
            for r in sorted(REGIMES) for i in REGIME_CASES[r]]


@pytest.mark.parametrize('regime,kind,level,cins,cout,bias', _regime_matrix())
def test_conv_forced_regime(maps, monkeypatch, regime, kind, level, cins, cout, bias):
    """The C library reads its switches at every call, so each regime is forced on the small maps."""
    from test_gpu_ops import _conv_case
    for k, v in REGIMES[regime].items():
        monkeypatch.setenv(k, v)
    _conv_case(maps, kind, level, cins, cout, bias)


def _one_by_one_cases():
    from test_gpu_ops import CONV_CASES
    return [c for c in CONV_CASES if c[0] == '1x1']


@pytest.mark.parametrize('kind,level,cins,cout,bias', _one_by_one_cases())
def test_conv_1x1_through_the_general_kernel(maps, monkeypatch, kind, level, cins, cout, bias):
    """1x1 layers with whole 16-channel chunks take the streaming-GEMM kernel (conv_1x1.h) by default (test_gpu_ops);
    B2M_CONV_1X1=0 sends them through the identity-rulebook variant of the general kernel, which still carries the
    layers with fewer than 16 input channels per chunk."""
    from test_gpu_ops import _conv_case
    monkeypatch.setenv('B2M_CONV_1X1', '0')
    monkeypatch.setenv('B2M_WGRAD_PIPE_IDENT', '0')       # and their weight gradient through the plain kernel
    _conv_case(maps, kind, level, cins, cout, bias)


# ------------------------------------------------------------------ 2. benchmark-size layers
@pytest.fixture(scope='module')
def scene150k():
    from box2mask_amd import synth
    from box2mask_amd.sparse import CoordinateManager
    b = synth.make_batch(1, seed0=3, target_voxels=150_000)
    m = CoordinateManager(b['vox_coords'], reorder=True)      # Morton rows, as the network runs them
    coords = m.coords[0].cpu().numpy()
    return m, coords


def _full_case(m, coords, ksize, c1, c2, cout, seed):
    from box2mask_amd import functional as F_
    from oracle import sparse_ref as S
    n = len(coords)
    cin = c1 + c2
    K = ksize ** 3
    torch.manual_seed(seed)
    x = torch.randn(n, cin)
    w = torch.randn(K, cin, cout) / (cin * min(K, 10)) ** 0.5 if K > 1 else torch.randn(cin, cout) / cin ** 0.5
    gy = torch.randn(n, cout)
    if K > 1:
        rb = m.rulebook_same(0, ksize)
        nbr = S.kernel_map_same(coords, ksize, 1)
        yo, dxo, dwo = S.conv_nbr_explicit(x, w, nbr, gy)
    else:
        rb = None
        yo = x @ w; dxo = gy @ w.t(); dwo = x.t() @ gy
    xg = x.cuda().requires_grad_(True); wg = w.cuda().requires_grad_(True)
    x1, x2 = (xg[:, :c1].contiguous(), xg[:, c1:].contiguous()) if c2 else (xg, None)
    yg = F_.sparse_conv(x1, x2, wg, None, rb, rb, K > 1, n)
    yg.backward(gy.cuda())
    torch.cuda.synchronize()
    e = (_close(yg, yo, 'forward'), _close(xg.grad, dxo, 'dgrad'), _close(wg.grad, dwo, 'wgrad'))
    print('rows %d K %d %d->%d: fwd %.2e dgrad %.2e wgrad %.2e' % (n, K, cin, cout, *e))


@pytest.mark.parametrize('ksize,c1,c2,cout', [(3, 96, 0, 96), (3, 96, 32, 96), (5, 6, 0, 32), (1, 128, 0, 96),
                                              (3, 32, 0, 32),
                                              # 1x1 streaming-GEMM kernel: 3 / 2 / 1 strips per wave, two sources
                                              (1, 96, 32, 128), (1, 64, 0, 64), (1, 96, 0, 32)])
def test_full_size_layer_150k(scene150k, ksize, c1, c2, cout):
    m, coords = scene150k
    _full_case(m, coords, ksize, c1, c2, cout, zlib.crc32(repr((ksize, c1, c2, cout)).encode()) % 1000)


def test_conv_1x1_bias_and_accumulate_150k(scene150k):
    """conv_1x1_kernel epilogue at benchmark size: Y = Y0 + X W + b (accumulate = 1, bias) with a row pitch wider than
    the channel count, against torch."""
    from box2mask_amd import functional as F_
    m, coords = scene150k
    n = len(coords)
    torch.manual_seed(5)
    for cin, cout in ((128, 96), (96, 64), (64, 20)):
        x = torch.randn(n, cin, device='cuda'); w = torch.randn(cin, cout, device='cuda') / cin ** 0.5
        b = torch.randn(1, cout, device='cuda')
        wide = torch.randn(n, cout + 12, device='cuda')
        y0 = wide[:, 4:4 + cout]                                   # 16-byte aligned rows, pitch cout + 12
        ref = y0.double() + x.double() @ w.double() + b.double()
        out = F_.conv_raw(x, None, F_.weight_pack(w.unsqueeze(0)), 1, b, None, n, cout, out=y0, accumulate=True)
        torch.cuda.synchronize()
        assert out.data_ptr() == y0.data_ptr()
        _close(out, ref, '1x1 %d->%d accumulate + bias' % (cin, cout), 1e-5)


def test_full_size_layer_1p2m_rows():
    """The level-0 map of the benchmark batch (8 scenes, ~1.2 M rows): un-split path over ~37 k items, XCD runs over
    the whole grid, weight-gradient chunks of 32 tiles."""
    from box2mask_amd import synth
    from box2mask_amd.sparse import CoordinateManager
    b = synth.make_batch(8, seed0=0, target_voxels=150_000)
    m = CoordinateManager(b['vox_coords'], reorder=True)
    coords = m.coords[0].cpu().numpy()
    assert len(coords) > 1_000_000
    _full_case(m, coords, 3, 96, 0, 96, 11)


# ------------------------------------------------------------------ 3. dense equivalence, random occupancy, no oracle
def _random_sites(B, X, Y, Z, density, seed):
    rng = np.random.default_rng(seed)
    occ = rng.random((B, X, Y, Z)) < density
    occ[:, 0, 0, 0] = True                       # pin the grid origin so that dense and sparse indices agree
    c = np.argwhere(occ).astype(np.int32)
    return c[rng.permutation(len(c))]            # rows in random order


def _dense_weight(w, k):
    """(K,Cin,Cout), offset index x-fastest -> conv3d weight (Cout,Cin,kx,ky,kz) for a dense [b,c,x,y,z] grid."""
    K, ci, co = w.shape
    return w.reshape(k, k, k, ci, co).permute(4, 3, 2, 1, 0).contiguous()


def _scatter_dense(feat, coords, shape, ts=1):
    B, X, Y, Z = shape
    d = torch.zeros(B, feat.shape[1], X, Y, Z, dtype=feat.dtype)
    c = torch.as_tensor(coords).long()
    d[c[:, 0], :, c[:, 1] // ts, c[:, 2] // ts, c[:, 3] // ts] = feat
    return d


def _read_dense(d, coords, ts=1):
    c = torch.as_tensor(coords).long()
    return d[c[:, 0], :, c[:, 1] // ts, c[:, 2] // ts, c[:, 3] // ts]


@pytest.mark.parametrize('ksize,level,cin,cout', [(3, 0, 32, 48), (3, 0, 96, 96), (5, 0, 6, 32), (3, 1, 64, 64)])
def test_dense_equivalence_stride1_random_occupancy(ksize, level, cin, cout):
    from box2mask_amd import functional as F_
    from box2mask_amd.sparse import CoordinateManager
    shape = (2, 44, 36, 28) if cin * cout < 96 * 96 else (2, 32, 28, 24)      # keeps the dense fp64 reference at seconds
    c0 = _random_sites(*shape, 0.3, 17 + ksize + level)
    m = CoordinateManager(torch.from_numpy(c0))
    m.ensure_level(level)
    coords = m.coords[level].cpu().numpy()
    ts = 1 << level
    n = len(coords)
    gshape = (shape[0],) + tuple((s + ts - 1) // ts for s in shape[1:])
    torch.manual_seed(ksize * 10 + level)
    x = torch.randn(n, cin, dtype=torch.float64)
    w = torch.randn(ksize ** 3, cin, cout, dtype=torch.float64) / (cin * 10) ** 0.5
    gy = torch.randn(n, cout, dtype=torch.float64)
    xd = x.clone().requires_grad_(True); wd = w.clone().requires_grad_(True)
    yd = _read_dense(F.conv3d(_scatter_dense(xd, coords, gshape, ts), _dense_weight(wd, ksize), padding=ksize // 2), coords, ts)
    yd.backward(gy)
    rb = m.rulebook_same(level, ksize)
    xg = x.float().cuda().requires_grad_(True); wg = w.float().cuda().requires_grad_(True)
    yg = F_.sparse_conv(xg, None, wg, None, rb, rb, True, n)
    yg.backward(gy.float().cuda())
    _close(yg, yd, 'forward'); _close(xg.grad, xd.grad, 'dgrad'); _close(wg.grad, wd.grad, 'wgrad')


@pytest.mark.parametrize('cin,cout', [(32, 32), (96, 64)])
def test_dense_equivalence_k2s2_and_transpose_random_occupancy(cin, cout):
    from box2mask_amd import functional as F_
    from box2mask_amd.sparse import CoordinateManager
    shape = (2, 44, 36, 28)                                   # even sides: coarse cells align with the dense stride-2 grid
    c0 = _random_sites(*shape, 0.25, 23)
    m = CoordinateManager(torch.from_numpy(c0))
    m.ensure_level(1)
    fine, coarse = m.coords[0].cpu().numpy(), m.coords[1].cpu().numpy()
    cshape = (shape[0],) + tuple(s // 2 for s in shape[1:])
    nf, nc = len(fine), len(coarse)
    torch.manual_seed(cin)
    # ---- strided convolution fine -> coarse
    x = torch.randn(nf, cin, dtype=torch.float64); w = torch.randn(8, cin, cout, dtype=torch.float64) / cin ** 0.5
    gy = torch.randn(nc, cout, dtype=torch.float64)
    xd = x.clone().requires_grad_(True); wd = w.clone().requires_grad_(True)
    yd = _read_dense(F.conv3d(_scatter_dense(xd, fine, shape), _dense_weight(wd, 2), stride=2), coarse, 2)
    yd.backward(gy)
    xg = x.float().cuda().requires_grad_(True); wg = w.float().cuda().requires_grad_(True)
    yg = F_.sparse_conv(xg, None, wg, None, m.rulebook_down(0), m.rulebook_up(0), False, nc)
    yg.backward(gy.float().cuda())
    _close(yg, yd, 'down forward'); _close(xg.grad, xd.grad, 'down dgrad'); _close(wg.grad, wd.grad, 'down wgrad')
    # ---- transposed convolution coarse -> the EXISTING fine sites
    x = torch.randn(nc, cin, dtype=torch.float64); w = torch.randn(8, cin, cout, dtype=torch.float64) / cin ** 0.5
    gy = torch.randn(nf, cout, dtype=torch.float64)
    xd = x.clone().requires_grad_(True); wd = w.clone().requires_grad_(True)
    wt = wd.reshape(2, 2, 2, cin, cout).permute(3, 4, 2, 1, 0).contiguous()       # conv_transpose3d: (Cin,Cout,kx,ky,kz)
    yd = _read_dense(F.conv_transpose3d(_scatter_dense(xd, coarse, cshape, 2), wt, stride=2), fine)
    yd.backward(gy)
    xg = x.float().cuda().requires_grad_(True); wg = w.float().cuda().requires_grad_(True)
    yg = F_.sparse_conv(xg, None, wg, None, m.rulebook_up(0), m.rulebook_down(0), False, nf)
    yg.backward(gy.float().cuda())
    _close(yg, yd, 'up forward'); _close(xg.grad, xd.grad, 'up dgrad'); _close(wg.grad, wd.grad, 'up wgrad')


# ------------------------------------------------------------------ 4. BatchNorm statistics from the convolution's epilogue
@pytest.mark.parametrize('regime', ['unsplit', 'split4', 'split2', 'many_slices', 'unsplit_two_launch_reduction'])
@pytest.mark.parametrize('cin,cout', [(96, 96), (32, 64), (64, 128)])
def test_conv_tile_stats_feed_batchnorm(maps, monkeypatch, regime, cin, cout):
    """b2m_conv_fwd_stats: the per-tile column sums the convolution kernel leaves behind equal the sums of its output,
    and BatchNorm fed by them equals BatchNorm that reads the output (b2m_bn_stats) -- forward, running statistics and
    the gradients.  `many_slices` (atomic combine across workgroups) cannot provide them and must fall back.  The tile sums
    are reduced + finalized in one launch (bn_tilestats_finalize_one_kernel; maps of <= 8192 tiles) or in two
    (B2M_BN_TS_ONE=0: the form the 19 k-tile level-0 maps take)."""
    from box2mask_amd import functional as F_
    if regime == 'unsplit_two_launch_reduction':
        monkeypatch.setenv('B2M_BN_TS_ONE', '0')
        regime = 'unsplit'
    env = {'unsplit': {'B2M_CONV_TARGET': '0'}, 'split2': {'B2M_CONV_TARGET': '0', 'B2M_CONV_SPLIT2': '100000000'}, 'split4': {'B2M_CONV_TARGET': '800', 'B2M_CONV_CHUNKSPLIT': '0'}, 'many_slices': {'B2M_CONV_TARGET': '100000', 'B2M_CONV_MAXSLICE': '16'}}
    for k, v in env[regime].items():
        monkeypatch.setenv(k, v)
    m, _ = maps
    rb = m.rulebook_same(0, 3)
    n = rb.n_out
    torch.manual_seed(zlib.crc32(repr((regime, cin, cout)).encode()))
    x = torch.randn(n, cin, device='cuda')
    w = (torch.randn(27, cin, cout, device='cuda') * 0.05).requires_grad_(True)
    gamma = (torch.rand(cout, device='cuda') + 0.5).requires_grad_(True)
    beta = torch.randn(cout, device='cuda').requires_grad_(True)

    def run(stats):
        monkeypatch.setenv('B2M_CONV_STATS', '1' if stats else '0')
        rm, rv = torch.zeros(cout, device='cuda'), torch.ones(cout, device='cuda')
        y = F_.sparse_conv(x, None, w, None, rb, rb, True, n, collect_stats=True)
        ts = getattr(y, '_b2m_tile_stats', None)
        out = F_.batch_norm(y, gamma, beta, rm, rv, True, relu=True)
        g = torch.autograd.grad(out.square().sum(), (w, gamma, beta))
        return y.detach(), ts, out.detach(), rm, rv, g
    y1, ts, o1, rm1, rv1, g1 = run(True)
    y0, ts0, o0, rm0, rv0, g0 = run(False)
    assert ts0 is None
    if regime == 'many_slices':
        assert ts is None                          # more than 4 slices: partial sums of different workgroups, no owner
        return
    assert ts is not None and ts[1] == (n + 63) // 64
    pad = torch.zeros(ts[1] * 64 - n, cout, device='cuda')
    yt = torch.cat([y1, pad]).reshape(ts[1], 64, cout).double()
    _close(ts[0][:, 0].double(), yt.sum(1), 'tile sums', 1e-5)
    _close(ts[0][:, 1].double(), (yt * yt).sum(1), 'tile sums of squares', 1e-5)
    _close(o1, o0, 'BN output', 1e-5)
    _close(rm1, rm0, 'running mean', 1e-5)
    _close(rv1, rv0, 'running var', 1e-5)
    for a, b, nm in zip(g1, g0, ('dW', 'dgamma', 'dbeta')):
        _close(a, b, nm, 1e-4)


@pytest.mark.parametrize('stem', ['1', '0'])
def test_stem_kernel_and_its_tile_stats(maps, monkeypatch, stem):
    """The network's first layer (5x5x5, 6 -> 32: conv_stem_kernel, B2M_CONV_STEM=0: conv_fwd_kernel) through the autograd
    operator with its 6-channel input: output and the per-tile column sums against the oracle / the output itself, and the
    two kernels against each other bit for bit (same summation order)."""
    from box2mask_amd import functional as F_
    from oracle import sparse_ref as S
    m, h = maps
    rb = m.rulebook_same(0, 5)
    n = rb.n_out
    torch.manual_seed(5)
    x = torch.randn(n, 6)
    w = torch.randn(125, 6, 32) * 0.05
    monkeypatch.setenv('B2M_CONV_STEM', stem)
    monkeypatch.setenv('B2M_CONV_TARGET', '0')          # un-split, as the 19 k-tile map of a benchmark batch runs (small maps are split)
    y = F_.sparse_conv(x.cuda(), None, w.cuda(), None, rb, rb, True, n, collect_stats=True)
    ts = getattr(y, '_b2m_tile_stats', None)
    # oracle on the manager's (Morton) row order: permute the oracle's rows by coordinates
    kg = S.pack_keys(m.coords[0].cpu().numpy()); ko = S.pack_keys(h.coords[0])
    order = np.argsort(kg); to_gpu = torch.from_numpy(order[np.searchsorted(kg[order], ko)])
    xo = torch.empty_like(x); xo[:] = x[to_gpu]            # oracle row r holds the device row to_gpu[r]
    yo = S.conv_nbr(xo, w, h.k_first())
    _close(y.cpu()[to_gpu], yo, 'stem forward (B2M_CONV_STEM=%s)' % stem)
    if stem == '1':
        assert ts is not None and ts[1] == (n + 63) // 64
        pad = torch.zeros(ts[1] * 64 - n, 32, device='cuda')
        yt = torch.cat([y, pad]).reshape(ts[1], 64, 32).double()
        _close(ts[0][:, 0].double(), yt.sum(1), 'tile sums', 1e-5)
        _close(ts[0][:, 1].double(), (yt * yt).sum(1), 'tile sums of squares', 1e-5)
        monkeypatch.setenv('B2M_CONV_STEM', '0')
        y_old = F_.sparse_conv(x.cuda(), None, w.cuda(), None, rb, rb, True, n)
        assert torch.equal(y, y_old), 'the two stem kernels differ'
    else:
        assert ts is None


# ------------------------------------------------------------------ 4. transposed k2s2 maps in scatter form (b2m_conv_up)
def _updown_cases():
    from test_gpu_ops import CONV_CASES
    return [c for c in CONV_CASES if c[0] in ('up', 'down')]


@pytest.mark.parametrize('kind,level,cins,cout,bias', _updown_cases())
def test_transposed_maps_in_scatter_form_vs_oracle(maps, monkeypatch, kind, level, cins, cout, bias):
    """Forward of the transposed convolution and data gradient of the strided one walk the DOWN rulebook with the roles of its
    row numbers exchanged (conv_fwd_flow_kernel<.., UP>); by default only from 450 (tile, strip) items on -- forced here on
    the small maps.  (/root/reference/models/detection_net.py:52-129.)"""
    from test_gpu_ops import _conv_case
    monkeypatch.setenv('B2M_CONV_UP_MIN_ITEMS', '1')
    _conv_case(maps, kind, level, cins, cout, bias)


def test_scatter_form_equals_the_fine_row_tiling_bit_for_bit(monkeypatch):
    """b2m_conv_up against b2m_conv_fwd over the UP rulebook on the same operands: every output element is ONE offset's sum over
    the input channels, accumulated in the same chunk order by both -- equal bits, with and without a second source, bias,
    accumulate (a read-modify-write of the tensor already there) and the inference epilogue (scale, shift, residual,
    ReLU); the hook proves which kernel ran."""
    from box2mask_amd import _lib, functional as F_, synth
    from box2mask_amd.sparse import CoordinateManager
    b = synth.make_batch(2, seed0=11, target_voxels=20000, pts_per_m2=8000.0)
    m = CoordinateManager(b['vox_coords'].cuda(), reorder=True)
    rb = m.rulebook_up(0)
    nf, nc = rb.n_out, rb.n_in
    assert rb.scatter is m.rulebook_down(0) and nc == rb.scatter.n_out
    torch.manual_seed(3)
    ran = []

    def hook(name, args, meta):
        if name == 'b2m_conv_up':
            return lambda: ran.append(meta['ran'].value)
        return None
    for c1, c2, co in ((96, 0, 96), (64, 32, 128), (32, 0, 32)):
        x1 = torch.randn(nc, c1, device='cuda'); x2 = torch.randn(nc, c2, device='cuda') if c2 else None
        w = torch.randn(8, c1 + c2, co, device='cuda') * 0.1
        bias = torch.randn(1, co, device='cuda'); y0 = torch.randn(nf, co, device='cuda')
        scale = torch.rand(co, device='cuda') + 0.5; shift = torch.randn(co, device='cuda'); res = torch.randn(nf, co, device='cuda')
        wp = F_.weight_pack(w)
        outs = {}
        for form in ('scatter', 'tiled'):
            monkeypatch.setenv('B2M_CONV_UP', '1' if form == 'scatter' else '0')
            monkeypatch.setenv('B2M_CONV_UP_MIN_ITEMS', '1')
            _lib.reload_env()
            del ran[:]
            _lib.set_hook(hook)
            try:
                outs[form] = (F_.conv_raw(x1, x2, wp, 8, None, rb, nf, co),
                              F_.conv_raw(x1, x2, wp, 8, bias, rb, nf, co, out=y0.clone(), accumulate=True),
                              F_.conv_affine(x1, x2, w, rb, nf, scale, shift, res, True),
                              F_.conv_affine(x1, x2, w, rb, nf, scale, shift, None, False))
            finally:
                _lib.set_hook(None)
            torch.cuda.synchronize()
            assert ran == ([1, 1, 1, 1] if form == 'scatter' else [0, 0, 0, 0]), (form, ran)
        for a, b_ in zip(outs['scatter'], outs['tiled']):
            assert torch.equal(a, b_), float((a - b_).abs().max())
