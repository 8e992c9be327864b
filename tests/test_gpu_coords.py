"""GPU parity: coordinate maps, kernel maps and tile rulebooks are bit-exact against the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _decode_rulebook(rb):
    """tile rulebook -> neighbour table (K, n_out)."""
    from box2mask_amd.sparse import TILE
    K, n_out, nt = rb.K, rb.n_out, rb.ntiles
    ldr = nt * TILE
    rin = rb.rb_in[:K * ldr].cpu().numpy().reshape(K, nt, TILE)
    rout = rb.rb_out[:K * ldr].cpu().numpy().reshape(K, nt, TILE)
    cnt = rb.rb_cnt[:K * nt].cpu().numpy().reshape(K, nt)
    nbr = np.full((K, nt * TILE), -1, np.int32)
    for k in range(K):
        for t in range(nt):
            c = cnt[k, t]
            assert (rin[k, t, c:] == -1).all() and (rout[k, t, c:] == 0).all()
            o = rout[k, t, :c].astype(np.int64)
            assert (np.diff(o) > 0).all()                      # compacted in output-row order
            nbr[k, t * TILE + o] = rin[k, t, :c]
    return nbr[:, :n_out], int(cnt.sum())


def _check(coords, levels=4, k0=5):
    from box2mask_amd.sparse import CoordinateManager
    from oracle import sparse_ref as S
    m = CoordinateManager(torch.from_numpy(coords), keep_tables=True)
    h = S.Hierarchy(coords, n_levels=levels, k0=k0)
    assert int(m.dup_count.item()) == 0
    for l in range(levels):
        assert np.array_equal(m.coords[l].cpu().numpy(), h.coords[l]), 'level %d coords' % l
        if l + 1 < levels:
            m.ensure_level(l + 1)
            assert np.array_equal(m.parent[l].cpu().numpy(), h.parent[l])
            assert np.array_equal(m.koff[l].cpu().numpy(), h.koff[l])
        rb = m.rulebook_same(l, 3)
        assert np.array_equal(rb.nbr.cpu().numpy()[:, :h.n(l)], h.k3(l)), 'level %d k3 map' % l
        dec, pairs = _decode_rulebook(rb)
        assert np.array_equal(dec, h.k3(l)) and pairs == int((h.k3(l) >= 0).sum()) == rb.pairs
    rb5 = m.rulebook_same(0, k0)
    assert np.array_equal(rb5.nbr.cpu().numpy()[:, :h.n(0)], h.k_first())
    assert np.array_equal(_decode_rulebook(rb5)[0], h.k_first())
    for l in range(levels - 1):
        assert np.array_equal(_decode_rulebook(m.rulebook_down(l))[0], h.down(l)), 'down %d' % l
        assert np.array_equal(_decode_rulebook(m.rulebook_up(l))[0], h.up(l)), 'up %d' % l
    # the production path (no neighbour table: probes go straight into the rulebook) must give the same bits
    m2 = CoordinateManager(torch.from_numpy(coords), keep_tables=False)
    for l, ks in [(l, 3) for l in range(levels)] + [(0, k0)]:
        a, b = m.rulebook_same(l, ks), m2.rulebook_same(l, ks)
        assert b.nbr is None
        for f in ('rb_in', 'rb_out', 'rb_cnt'):
            x, y = getattr(a, f), getattr(b, f)
            if f == 'rb_cnt' and a.ntiles < 64:          # (the tail behind the counts is written from 64 tiles on)
                x, y = x[:a.K * a.ntiles], y[:a.K * a.ntiles]
            assert torch.equal(x, y), (l, ks, f)
    return m, h


def test_random_coords():
    rng = np.random.default_rng(0)
    c = rng.integers(0, 60, (20000, 4)).astype(np.int32)
    c[:, 0] %= 3
    c = np.unique(c, axis=0)          # lexicographic, as np.unique in the reference data loader
    _check(c, levels=5)


@pytest.mark.parametrize('n', [1, 2, 63, 64, 65, 127, 128, 129, 300])
def test_ragged_sizes(n):
    rng = np.random.default_rng(n)
    c = rng.integers(0, 12, (4 * n, 4)).astype(np.int32)
    c[:, 0] = 0
    c = np.unique(c, axis=0)[:n]
    _check(c, levels=3)


def test_synthetic_scene_batch():
    from box2mask_amd import synth
    b = synth.make_batch(2, seed0=1, target_voxels=12000, pts_per_m2=6000.0)
    m, h = _check(b['vox_coords'].numpy(), levels=8)
    assert [m.n(l) for l in range(8)] == [h.n(l) for l in range(8)]


def test_duplicates_are_counted_and_bounds_checked():
    from box2mask_amd.sparse import CoordinateManager
    c = np.array([[0, 1, 1, 1], [0, 1, 1, 1], [0, 2, 2, 2]], np.int32)
    m = CoordinateManager(torch.from_numpy(c))
    assert int(m.dup_count.item()) == 1
    with pytest.raises(ValueError):
        CoordinateManager(torch.tensor([[0, -1, 0, 0]], dtype=torch.int32))
    with pytest.raises(ValueError):
        CoordinateManager(torch.tensor([[0, 70000, 0, 0]], dtype=torch.int32))


def test_full_size_scene_properties():
    """BASELINE size (150k voxels): size-independent properties instead of the slow oracle."""
    from box2mask_amd import synth
    from box2mask_amd.sparse import CoordinateManager
    sc = synth.make_scene(0)
    c = synth.batched_coordinates([sc['vox_coords']])
    m = CoordinateManager(c, keep_tables=True)
    rb = m.rulebook_same(0, 3)
    nbr = rb.nbr.cpu().numpy()
    n = m.n(0)
    assert np.array_equal(nbr[13], np.arange(n))                      # centre offset = identity
    for k in (0, 5, 12):                                              # mirror symmetry
        o = np.nonzero(nbr[k] >= 0)[0]
        assert np.array_equal(nbr[26 - k][nbr[k][o]], o)
    cc = m.coords[0].cpu().numpy()
    offs = np.array([[(k % 3) - 1, (k // 3) % 3 - 1, k // 9 - 1] for k in range(27)])
    for k in (1, 20):
        o = np.nonzero(nbr[k] >= 0)[0][:5000]
        assert np.array_equal(cc[nbr[k][o]][:, 1:], cc[o][:, 1:] + offs[k])
    m.ensure_level(7)
    ns = [m.n(l) for l in range(8)]
    assert all(a > b for a, b in zip(ns, ns[1:]))
    tot = 0
    for l in range(7):
        tot = m.rulebook_down(l).pairs
        assert tot == m.n(l)                                          # every fine voxel is exactly one pair


def _expected_runs(cnt):
    """XCD work boundaries as include/b2m.h states them (rb_cnt tail), restated with numpy."""
    K, nt = cnt.shape
    cost = np.where(cnt > 0, 3 * ((cnt + 15) // 16) + 1, 0).sum(0).astype(np.int64)
    prefix = np.cumsum(cost)
    total = int(prefix[-1]) if nt else 0
    start = [nt * x // 8 for x in range(8)] + [nt]
    if total > 0:
        for x in range(1, 8):
            target = (total * x + 7) // 8
            start[x] = int(np.searchsorted(prefix, target, side='left')) + 1
    cap = (nt * 5 + 31) // 32
    for x in range(1, 8):
        start[x] = min(max(start[x], start[x - 1]), start[x - 1] + cap)
    for x in range(7, 0, -1):
        start[x] = max(start[x], start[x + 1] - cap)
    return cost, np.array(start), cap


@pytest.mark.parametrize('case', ['scene', 'skewed', 'tiny', 'empty_half'])
def test_xcd_runs_carry_equal_work(case):
    """The tail of rb_cnt: per-tile cost and the nine run boundaries, exactly as stated; runs cover all tiles, none is
    longer than the cap the launch grids are sized for; on a real scene the runs' work is level to one tile's cost."""
    from box2mask_amd import synth
    from box2mask_amd.sparse import CoordinateManager
    rng = np.random.default_rng(3)
    if case == 'scene':
        coords = synth.make_batch(2, seed0=4, target_voxels=30000, pts_per_m2=8000.0)['vox_coords']
    elif case == 'skewed':      # a dense block (27 neighbours everywhere) next to isolated voxels (1 active offset each)
        g = np.stack(np.meshgrid(*[np.arange(36)] * 3, indexing='ij'), -1).reshape(-1, 3)
        iso = np.stack(np.meshgrid(*[np.arange(40) * 3 + 200] * 3, indexing='ij'), -1).reshape(-1, 3)
        xyz = np.concatenate([g, iso])
        coords = torch.from_numpy(np.concatenate([np.zeros((len(xyz), 1), np.int64), xyz], 1)).int()
    elif case == 'tiny':
        xyz = rng.integers(0, 12, (300, 3))
        xyz = np.unique(xyz, axis=0)
        coords = torch.from_numpy(np.concatenate([np.zeros((len(xyz), 1), np.int64), xyz], 1)).int()
    else:                       # isolated voxels only in the first half of the rows: costs 1,1,1,... then the block
        iso = np.stack(np.meshgrid(*[np.arange(30) * 3] * 3, indexing='ij'), -1).reshape(-1, 3)
        g = np.stack(np.meshgrid(*[np.arange(20)] * 3, indexing='ij'), -1).reshape(-1, 3) + 400
        xyz = np.concatenate([iso, g])
        coords = torch.from_numpy(np.concatenate([np.zeros((len(xyz), 1), np.int64), xyz], 1)).int()
    m = CoordinateManager(coords, reorder=True)
    for rb in (m.rulebook_same(0, 3), m.rulebook_down(0), m.rulebook_up(0), m.rulebook_same(1, 3)):
        K, nt = rb.K, rb.ntiles
        raw = rb.rb_cnt.cpu().numpy()
        assert raw.shape[0] == K * nt + 16 + 2 * nt
        if nt < 64:                 # small rulebooks: the convolutions take equal tile counts, the tail stays unwritten
            continue
        cnt = raw[:K * nt].reshape(K, nt).astype(np.int64)
        cost, start, cap = _expected_runs(cnt)
        assert (raw[K * nt + 16:K * nt + 16 + nt] == cost).all()
        order = raw[K * nt + 16 + nt:]
        assert (np.sort(order) == np.arange(nt)).all()          # a permutation of the tiles
        got = raw[K * nt:K * nt + 9]
        assert (got == start).all(), (case, got, start)
        assert got[0] == 0 and got[8] == nt and (np.diff(got) >= 0).all() and (np.diff(got) <= cap).all()
        for x in range(8):                                      # every run: row order, then the window heavy -> light
            s0, s1 = int(got[x]), int(got[x + 1])
            w0 = max(s0, s1 - 768)
            assert (order[s0:w0] == np.arange(s0, w0)).all()
            win = order[w0:s1]
            assert (np.sort(win) == np.arange(w0, s1)).all()
            if len(win):
                cmax = int(cost[w0:s1].max())
                cls = 7 - cost[win] * 8 // (cmax + 1)
                assert (np.diff(cls) >= 0).all()                                    # heaviest class first
                assert all((np.diff(win[cls == c]) > 0).all() for c in range(8))      # row order inside a class
        if case == 'scene' and nt >= 64:
            work = np.array([cost[got[x]:got[x + 1]].sum() for x in range(8)])
            assert work.max() - work.min() <= 2 * cost.max(), work
        if case == 'skewed' and rb.K == 27 and rb is m.rulebook_same(0, 3):
            assert (np.diff(got) == cap).any()          # the clamp is reached (and the convolution still covers every tile)


def test_conv_on_skewed_runs_matches_equal_counts(monkeypatch):
    """Runs at the clamp (dense block + isolated voxels): forward and weight gradient agree with the equal-count order."""
    from box2mask_amd import functional as F_
    from box2mask_amd.sparse import CoordinateManager
    g = np.stack(np.meshgrid(*[np.arange(36)] * 3, indexing='ij'), -1).reshape(-1, 3)
    iso = np.stack(np.meshgrid(*[np.arange(40) * 3 + 200] * 3, indexing='ij'), -1).reshape(-1, 3)
    xyz = np.concatenate([g, iso])
    coords = torch.from_numpy(np.concatenate([np.zeros((len(xyz), 1), np.int64), xyz], 1)).int()
    m = CoordinateManager(coords, reorder=True)
    rb = m.rulebook_same(0, 3)
    n = rb.n_out
    torch.manual_seed(1)
    x = torch.randn(n, 96, device='cuda'); w = torch.randn(27, 96, 96, device='cuda') * 0.05; dy = torch.randn(n, 96, device='cuda')

    def run():
        y = F_.conv_raw(x, None, F_.weight_pack(w), 27, None, rb, n, 96)
        dw = torch.zeros_like(w)
        F_.wgrad_raw(x, dy, rb, 27, dw, 0)
        torch.cuda.synchronize()
        return y, dw
    y1, dw1 = run()
    monkeypatch.setenv('B2M_XCD_BALANCE', '0')
    y0, dw0 = run()
    assert torch.equal(y1, y0)                           # the same wave code per tile: the same bits
    assert float((dw1 - dw0).abs().max()) / float(dw0.abs().max()) < 1e-5


@pytest.mark.parametrize('n', [1, 2, 63, 64, 65, 4095, 4096, 4097, 100_003, 1_300_000])
def test_radix_argsort_matches_stable_sort(n):
    """b2m_radix_argsort (the Morton row order of CoordinateManager(reorder=True)): permutation and inverse equal a stable
    sort's, for key sets with few distinct values (long runs of equal digits), full 64-bit keys and masked passes."""
    from box2mask_amd import _lib
    g = torch.Generator().manual_seed(n)
    cases = {
        'few values': (torch.randint(0, 7, (n,), generator=g), 0xFF),
        'morton-like': ((torch.randint(0, 8, (n,), generator=g) << 48) | torch.randint(0, 1 << 30, (n,), generator=g),
                        ((1 << 30) - 1) | (0x3FF << 48)),
        'all 63 bits': (torch.randint(0, (1 << 62), (n,), generator=g), 0xFFFFFFFFFFFFFFFF),
        'all equal': (torch.full((n,), 12345), 0),
    }
    for what, (keys, mask) in cases.items():
        k = keys.to(torch.int64).cuda()
        perm = torch.empty(n, dtype=torch.int64, device='cuda'); inv = torch.empty_like(perm)
        scratch = torch.empty((_lib.load().b2m_radix_argsort_scratch(n) + 7) // 8, dtype=torch.int64, device='cuda')
        _lib.call('b2m_radix_argsort', k.data_ptr(), n, mask, perm.data_ptr(), inv.data_ptr(), scratch.data_ptr())
        ref = torch.sort(keys.to(torch.int64), stable=True).indices
        assert torch.equal(perm.cpu(), ref), what
        assert torch.equal(inv.cpu()[ref], torch.arange(n)), what


def _hilbert_keys_numpy(c, bits):
    """Skilling's transform, vectorised (the restatement b2m_hilbert_keys is checked against)."""
    X = c[:, 1:4].astype(np.uint64).copy()
    M = np.uint64(1) << np.uint64(bits - 1)
    Q = M
    while Q > 1:
        P = Q - np.uint64(1)
        for a in range(3):
            sel = (X[:, a] & Q) != 0
            X[sel, 0] ^= P
            ns = ~sel
            t = (X[ns, 0] ^ X[ns, a]) & P
            X[ns, 0] ^= t
            X[ns, a] ^= t
        Q >>= np.uint64(1)
    X[:, 1] ^= X[:, 0]; X[:, 2] ^= X[:, 1]
    t = np.zeros(len(X), np.uint64)
    Q = M
    while Q > 1:
        t[(X[:, 2] & Q) != 0] ^= (Q - np.uint64(1))
        Q >>= np.uint64(1)
    X ^= t[:, None]
    k = c[:, 0].astype(np.uint64) << np.uint64(48)
    for bit in range(bits):
        for a in range(3):
            k |= ((X[:, a] >> np.uint64(bit)) & np.uint64(1)) << np.uint64(3 * bit + (2 - a))
    return k.astype(np.int64)


def test_hilbert_keys_walk_a_dense_cube_cell_by_cell_and_match_the_restatement():
    """(i) On a dense 16^3 cube the rows sorted by key form ONE path through face-adjacent cells -- the defining property of a
    Hilbert curve, independent of any implementation; (ii) random coordinates of two scenes: keys equal the numpy restatement
    of Skilling's transform, the batch index leads; (iii) the manager's default row order is that order."""
    from box2mask_amd import _lib
    from box2mask_amd.sparse import CoordinateManager
    g = np.stack(np.meshgrid(np.arange(16), np.arange(16), np.arange(16), indexing='ij'), -1).reshape(-1, 3)
    c = np.concatenate([np.zeros((len(g), 1), np.int64), g], 1).astype(np.int32)
    ct = torch.from_numpy(c).cuda()
    keys = torch.empty(len(c), dtype=torch.int64, device='cuda')
    _lib.call('b2m_hilbert_keys', ct.data_ptr(), len(c), 4, keys.data_ptr())
    k = keys.cpu().numpy()
    assert len(np.unique(k)) == len(k) and k.min() == 0 and k.max() == 16 ** 3 - 1
    path = g[np.argsort(k)]
    assert (np.abs(np.diff(path, axis=0)).sum(1) == 1).all()
    rng = np.random.default_rng(3)
    c = np.concatenate([rng.integers(0, 2, (5000, 1)), rng.integers(0, 700, (5000, 3))], 1).astype(np.int32)
    ct = torch.from_numpy(c).cuda()
    keys = torch.empty(len(c), dtype=torch.int64, device='cuda')
    _lib.call('b2m_hilbert_keys', ct.data_ptr(), len(c), 10, keys.data_ptr())
    assert np.array_equal(keys.cpu().numpy(), _hilbert_keys_numpy(c, 10))
    cu = np.unique(c, axis=0)
    m = CoordinateManager(torch.from_numpy(cu), reorder=True)
    order = np.argsort(_hilbert_keys_numpy(cu, int(cu.max()).bit_length()), kind='stable')
    assert np.array_equal(m.perm.cpu().numpy(), order)


def test_row_order_switch_morton_keeps_the_z_order(monkeypatch):
    """B2M_ROW_ORDER=morton: the manager's rows follow b2m_morton_keys (batch index, then interleaved x / y / z bits) -- the order
    of rounds 1-3, kept for A/B -- and a network forward is the same function of its input in either order."""
    from box2mask_amd.sparse import CoordinateManager
    rng = np.random.default_rng(5)
    c = np.unique(np.concatenate([rng.integers(0, 2, (4000, 1)), rng.integers(0, 300, (4000, 3))], 1), axis=0).astype(np.int32)

    def zkey(c):
        k = c[:, 0].astype(np.uint64) << np.uint64(48)
        for bit in range(16):
            for a, sh in ((1, 0), (2, 1), (3, 2)):
                k |= ((c[:, a].astype(np.uint64) >> np.uint64(bit)) & np.uint64(1)) << np.uint64(3 * bit + sh)
        return k
    monkeypatch.setenv('B2M_ROW_ORDER', 'morton')
    m = CoordinateManager(torch.from_numpy(c), reorder=True)
    assert np.array_equal(m.perm.cpu().numpy(), np.argsort(zkey(c), kind='stable'))
    monkeypatch.delenv('B2M_ROW_ORDER')
    h = CoordinateManager(torch.from_numpy(c), reorder=True)
    assert not np.array_equal(h.perm.cpu().numpy(), m.perm.cpu().numpy())
    # the same kernel map in both orders: pairs of level-0 k3, as (input coordinate, output coordinate, offset) sets
    assert m.rulebook_same(0, 3).pairs == h.rulebook_same(0, 3).pairs
