"""Independent checks that pin the CPU restatement of the sparse engine (oracle/sparse_ref.py).
MinkowskiEngine itself is unavailable ("parity unpinned"); these are the substitute:
dense equivalence with torch's conv3d / conv_transpose3d, brute-force kernel maps, BatchNorm1d,
manual segment means and fp64 gradcheck."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import sparse_ref as S


def _dense_block(B, X, Y, Z):
    g = np.stack(np.meshgrid(np.arange(B), np.arange(X), np.arange(Y), np.arange(Z), indexing='ij'), -1)
    return g.reshape(-1, 4).astype(np.int32)      # lexicographic (b,x,y,z)


def _to_dense(feat, coords, B, X, Y, Z, ts=1):
    C = feat.shape[1]
    d = torch.zeros(B, C, X, Y, Z, dtype=feat.dtype)
    c = torch.as_tensor(coords).long()
    d[c[:, 0], :, c[:, 1] // ts, c[:, 2] // ts, c[:, 3] // ts] = feat
    return d


def _w_dense(w, k):
    """(K,Cin,Cout) x-fastest -> conv3d weight (Cout,Cin,kx,ky,kz) for dense[b,c,x,y,z]."""
    K, ci, co = w.shape
    return w.reshape(k, k, k, ci, co).permute(4, 3, 2, 1, 0).contiguous()   # [dz,dy,dx,ci,co] -> [co,ci,dx,dy,dz]


@pytest.mark.parametrize('k', [3, 5])
def test_dense_equivalence_stride1(k):
    torch.manual_seed(0)
    B, X, Y, Z, ci, co = 2, 6, 5, 7, 4, 3
    coords = _dense_block(B, X, Y, Z)
    x = torch.randn(len(coords), ci, dtype=torch.float64)
    w = torch.randn(k ** 3, ci, co, dtype=torch.float64)
    y = S.conv_nbr(x, w, S.kernel_map_same(coords, k, 1))
    yd = F.conv3d(_to_dense(x, coords, B, X, Y, Z), _w_dense(w, k), padding=k // 2)
    c = torch.as_tensor(coords).long()
    assert torch.allclose(y, yd[c[:, 0], :, c[:, 1], c[:, 2], c[:, 3]], atol=1e-10)


def test_dense_equivalence_k2s2_and_transpose():
    torch.manual_seed(1)
    B, X, Y, Z, ci, co = 2, 6, 4, 8, 3, 5
    coords = _dense_block(B, X, Y, Z)
    cc, parent, koff = S.stride_coords(coords, 1)
    x = torch.randn(len(coords), ci, dtype=torch.float64)
    w = torch.randn(8, ci, co, dtype=torch.float64)
    y = S.conv_nbr(x, w, S.child_table(parent, koff, len(cc)))
    yd = F.conv3d(_to_dense(x, coords, B, X, Y, Z), _w_dense(w, 2), stride=2)
    c = torch.as_tensor(cc).long()
    assert torch.allclose(y, yd[c[:, 0], :, c[:, 1] // 2, c[:, 2] // 2, c[:, 3] // 2], atol=1e-10)
    # transposed: coarse -> existing fine map
    xc = torch.randn(len(cc), co, dtype=torch.float64)
    wt = torch.randn(8, co, ci, dtype=torch.float64)
    yt = S.conv_nbr(xc, wt, S.up_table(parent, koff))
    wd = wt.reshape(2, 2, 2, co, ci).permute(3, 4, 2, 1, 0).contiguous()     # conv_transpose3d: (Cin,Cout,kx,ky,kz)
    ytd = F.conv_transpose3d(_to_dense(xc, cc, B, X // 2, Y // 2, Z // 2, ts=2), wd, stride=2)
    f = torch.as_tensor(coords).long()
    assert torch.allclose(yt, ytd[f[:, 0], :, f[:, 1], f[:, 2], f[:, 3]], atol=1e-10)


def test_kernel_map_matches_bruteforce_and_is_symmetric():
    rng = np.random.default_rng(3)
    pts = np.unique(rng.integers(0, 9, (400, 4)), axis=0).astype(np.int32)
    pts[:, 0] %= 2
    pts = np.unique(pts, axis=0)
    for ts in (1, 2):
        c = pts.copy(); c[:, 1:] *= ts
        for k in (3, 5):
            nbr = S.kernel_map_same(c, k, ts)
            assert np.array_equal(nbr, S.kernel_map_bruteforce(c, k, ts))
            K = k ** 3
            for kk in range(K):      # mirror symmetry used by the data gradient
                o = np.nonzero(nbr[kk] >= 0)[0]
                assert np.array_equal(nbr[K - 1 - kk][nbr[kk][o]], o)


def test_stride_coords_invariants():
    rng = np.random.default_rng(5)
    c = np.unique(rng.integers(0, 40, (3000, 4)), axis=0).astype(np.int32)
    c[:, 0] %= 3
    c = np.unique(c, axis=0)
    co, parent, koff = S.stride_coords(c, 1)
    assert len(np.unique(S.pack_keys(co))) == len(co)                 # unique
    assert (co[:, 1:] % 2 == 0).all()
    assert np.array_equal(co[parent][:, 1:], (c[:, 1:] // 2) * 2)     # parent holds the floored coordinate
    first = np.full(len(co), len(c)); np.minimum.at(first, parent, np.arange(len(c)))
    assert (np.diff(first) > 0).all()                                 # rows ordered by first occurrence
    o = c[:, 1:] - co[parent][:, 1:]
    assert np.array_equal(koff, o[:, 0] + 2 * o[:, 1] + 4 * o[:, 2])
    t = S.child_table(parent, koff, len(co))
    assert (t >= 0).sum() == len(c)                                   # every fine voxel is exactly one pair


def test_batch_norm_and_pool():
    torch.manual_seed(2)
    x = torch.randn(500, 8)
    bn = torch.nn.BatchNorm1d(8)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-1, 1)
    rm, rv = bn.running_mean.clone(), bn.running_var.clone()
    y = S.batch_norm(x, bn.weight, bn.bias, rm, rv, True)
    assert torch.allclose(y, bn(x), atol=1e-6)
    assert torch.allclose(rm, bn.running_mean) and torch.allclose(rv, bn.running_var)
    ids = torch.randint(0, 17, (500,)); ids[:17] = torch.arange(17)
    p = S.segment_pool(x, ids, 17, 'avg')
    for s in range(17):
        assert torch.allclose(p[s], x[ids == s].mean(0), atol=1e-6)
    pm = S.segment_pool(x, ids, 17, 'max')
    for s in range(17):
        assert torch.equal(pm[s], x[ids == s].max(0)[0])


def test_conv_gradcheck_fp64():
    rng = np.random.default_rng(9)
    c = np.unique(rng.integers(0, 4, (40, 4)), axis=0).astype(np.int32); c[:, 0] = 0
    c = np.unique(c, axis=0)
    nbr = S.kernel_map_same(c, 3, 1)
    x = torch.randn(len(c), 3, dtype=torch.float64, requires_grad=True)
    w = torch.randn(27, 3, 2, dtype=torch.float64, requires_grad=True)
    b = torch.randn(1, 2, dtype=torch.float64, requires_grad=True)
    assert torch.autograd.gradcheck(lambda x, w, b: S.conv_nbr(x, w, nbr, b), (x, w, b), atol=1e-8)


def test_unet_oracle_runs_and_orders_rows():
    """Final rows are in input order and pooled row r <-> pooling id r (detection_net.py:347-350)."""
    from box2mask_amd import synth
    from box2mask_amd.config import scannet_config
    from oracle import unet_ref
    batch = synth.make_batch(2, seed0=3, target_voxels=1500, pts_per_m2=1500.0)
    cfg = scannet_config()
    torch.manual_seed(0)
    # random weights with the reference's key names/shapes (subset check happens on the GPU side)
    p = _random_params(cfg)
    out = unet_ref.forward(p, batch['vox_coords'].numpy(), batch['vox_features'], batch['pooling_ids'], cfg,
                           training=True, return_trunk=True)
    S_ = batch['input_location'].shape[0]
    assert out['mlp_offsets'].shape == (S_, 3) and out['mlp_semantics'].shape == (S_, 20)
    assert out['_trunk'].shape == (batch['vox_coords'].shape[0], 96)
    assert all(torch.isfinite(v).all() for v in out.values())


def _random_params(cfg):
    P, A = (32, 64, 128, 256, 256, 128, 96, 96), (256,) * 6
    p = {}

    def conv(name, K, ci, co, bias=False):
        p[name + '.kernel'] = torch.randn(K, ci, co) * (2.0 / (K * co)) ** 0.5 if K > 1 else torch.randn(ci, co) * (2.0 / co) ** 0.5
        if bias:
            p[name + '.bias'] = torch.zeros(1, co)

    def bn(name, c):
        p[name + '.bn.weight'] = torch.ones(c); p[name + '.bn.bias'] = torch.zeros(c)
        p[name + '.bn.running_mean'] = torch.zeros(c); p[name + '.bn.running_var'] = torch.ones(c)

    def layer(name, cin, planes):
        for b in range(cfg.layers):
            ci = cin if b == 0 else planes
            conv('%s.%d.conv1' % (name, b), 27, ci, planes); bn('%s.%d.norm1' % (name, b), planes)
            conv('%s.%d.conv2' % (name, b), 27, planes, planes); bn('%s.%d.norm2' % (name, b), planes)
            if ci != planes:
                conv('%s.%d.downsample.0' % (name, b), 1, ci, planes); bn('%s.%d.downsample.1' % (name, b), planes)

    conv('conv0p1s1', 125, cfg.in_channels, 32); bn('bn0', 32)
    inpl = 32
    for (c, b, blk), pl in zip([('conv1p1s2', 'bn1', 'block1'), ('conv2p2s2', 'bn2', 'block2'), ('conv3p4s2', 'bn3', 'block3'),
                                ('conv4p8s2', 'bn4', 'block4'), ('added_conv1p16s2', 'added_bn1', 'added_block1'),
                                ('added_conv2p32s2', 'added_bn2', 'added_block2'),
                                ('added_conv3p64s2', 'added_bn3', 'added_block3')], (P[0], P[1], P[2], P[3], A[0], A[1], A[2])):
        conv(c, 8, inpl, inpl); bn(b, inpl); layer(blk, inpl, pl); inpl = pl
    skips = [A[1], A[0], P[3], P[2], P[1], P[0], 32]
    planes = [A[3], A[4], A[5], P[4], P[5], P[6], P[7]]
    for (c, b, blk), pl, sk in zip([('added_convtr4p128s2', 'added_bntr4', 'added_block4'),
                                    ('added_convtr5p64s2', 'added_bntr5', 'added_block5'),
                                    ('added_convtr6p32s2', 'added_bntr6', 'added_block6'), ('convtr4p16s2', 'bntr4', 'block5'),
                                    ('convtr5p8s2', 'bntr5', 'block6'), ('convtr6p4s2', 'bntr6', 'block7'),
                                    ('convtr7p2s2', 'bntr7', 'block8')], planes, skips):
        conv(c, 8, inpl, pl); bn(b, pl); layer(blk, pl + sk, pl); inpl = pl
    for head, od in (('mlp_offsets', 3), ('mlp_bounds', 3), ('mlp_score', 1), ('mlp_semantics', 20)):
        conv(head + '.0', 1, 96, 96, True); bn(head + '.2', 96)
        conv(head + '.3', 1, 96, 96, True); bn(head + '.5', 96)
        conv(head + '.6', 1, 96, od, True)
    return p


def test_explicit_gradients_equal_autograd():
    """conv_nbr_explicit (the tape-free form used for full-size layers) == autograd of conv_nbr."""
    rng = np.random.default_rng(5)
    c = np.unique(rng.integers(0, 12, (900, 4)), axis=0).astype(np.int32)
    c[:, 0] %= 2
    c = np.unique(c, axis=0)
    nbr = S.kernel_map_same(c, 3, 1)
    torch.manual_seed(5)
    x = torch.randn(len(c), 7, dtype=torch.float64, requires_grad=True)
    w = torch.randn(27, 7, 5, dtype=torch.float64, requires_grad=True)
    gy = torch.randn(len(c), 5, dtype=torch.float64)
    y = S.conv_nbr(x, w, nbr)
    y.backward(gy)
    y2, dx2, dw2 = S.conv_nbr_explicit(x.detach(), w.detach(), nbr, gy)
    assert torch.allclose(y2, y.detach(), atol=1e-12)
    assert torch.allclose(dx2, x.grad, atol=1e-12) and torch.allclose(dw2, w.grad, atol=1e-12)
    # strided map (different in/out row counts)
    cc, parent, koff = S.stride_coords(c, 1)
    tab = S.child_table(parent, koff, len(cc))
    x = torch.randn(len(c), 4, dtype=torch.float64, requires_grad=True)
    w = torch.randn(8, 4, 6, dtype=torch.float64, requires_grad=True)
    gy = torch.randn(len(cc), 6, dtype=torch.float64)
    y = S.conv_nbr(x, w, tab); y.backward(gy)
    y2, dx2, dw2 = S.conv_nbr_explicit(x.detach(), w.detach(), tab, gy)
    assert torch.allclose(y2, y.detach(), atol=1e-12) and torch.allclose(dx2, x.grad, atol=1e-12) and torch.allclose(dw2, w.grad, atol=1e-12)
