"""B2M_DETERMINISTIC=1: every order-dependent reduction of the path takes its ordered form (two-stage weight-gradient
combine, un-split maps instead of the atomic split-K combine, sorted segment mean), so two runs of the same training
step give the same BITS; and the ordered forms agree with the fast (atomic) ones to rounding."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _step(seed):
    from box2mask_amd import synth
    from box2mask_amd.config import scannet_config
    from box2mask_amd.model import Model
    torch.manual_seed(seed)
    model = Model(scannet_config(), *synth.scannet_tables())
    model.train()
    # scenes large enough that levels 2.. take the split-K path and every weight-gradient launch has several tile chunks
    batch = synth.make_batch(4, seed0=21, target_voxels=20000, pts_per_m2=8000.0)
    losses = model.compute_loss(batch, 150)
    losses['optimization_loss'].backward()
    torch.cuda.synchronize()
    grads = {n: p.grad.detach().cpu().clone() for n, p in model.detection_model.named_parameters() if p.grad is not None}
    return float(losses['optimization_loss']), grads


def test_two_runs_give_the_same_bits(monkeypatch):
    monkeypatch.setenv('B2M_DETERMINISTIC', '1')
    l1, g1 = _step(5)
    l2, g2 = _step(5)
    assert l1 == l2
    assert g1.keys() == g2.keys() and len(g1) > 250
    bad = [n for n in g1 if not torch.equal(g1[n], g2[n])]
    assert not bad, 'gradients differ between two deterministic runs: %s' % bad[:5]


def test_ordered_forms_agree_with_the_fast_ones(monkeypatch):
    """Operator level (no BatchNorm amplification in between): weight gradient, split-K forward, segment mean."""
    from box2mask_amd import functional as F_, synth
    from box2mask_amd.sparse import CoordinateManager
    b = synth.make_batch(2, seed0=2, target_voxels=6000, pts_per_m2=6000.0)
    m = CoordinateManager(b['vox_coords'], reorder=True)
    rb = m.rulebook_same(0, 3)
    n = rb.n_out
    torch.manual_seed(0)
    x = torch.randn(n, 96, device='cuda'); w = torch.randn(27, 96, 96, device='cuda') * 0.05; dy = torch.randn(n, 96, device='cuda')
    ids = torch.randint(0, 300, (n,), device='cuda')

    def run():
        dw = torch.zeros_like(w)
        F_.wgrad_raw(x, dy, rb, 27, dw, 0)
        y = F_.conv_raw(x, None, F_.weight_pack(w), 27, None, rb, n, 96)
        p = F_.segment_pool(x, ids, 300, 'avg')
        torch.cuda.synchronize()
        return dw, y, p
    fast = run()
    monkeypatch.setenv('B2M_DETERMINISTIC', '1')
    det1, det2 = run(), run()
    for a, b_, c, what in zip(fast, det1, det2, ('wgrad', 'forward', 'segment mean')):
        assert torch.equal(b_, c), what + ': two deterministic runs differ'
        err = float((a - b_).abs().max()) / float(b_.abs().max())
        assert err < 1e-5, (what, err)


def test_prefetched_batch_gives_the_same_bits(monkeypatch):
    """Model.prefetch builds the sparse tensor (coordinate maps, kernel maps, foreground row list) on a second stream
    ahead of time: loss and gradients are the SAME BITS as when compute_loss builds them itself; a prefetch for other
    tensors is dropped (the batch is built in place); host tensors are copied by the side stream."""
    from box2mask_amd import synth
    from box2mask_amd.config import scannet_config
    from box2mask_amd.model import Model
    monkeypatch.setenv('B2M_DETERMINISTIC', '1')
    torch.manual_seed(5)
    model = Model(scannet_config(), *synth.scannet_tables())
    model.train()
    batch = synth.make_batch(3, seed0=31, target_voxels=12000, pts_per_m2=8000.0)
    other = synth.make_batch(2, seed0=77, target_voxels=5000, pts_per_m2=8000.0)

    def run(prefetch_of=None, **kw):
        model.detection_model.zero_grad()
        if prefetch_of is not None:
            model.prefetch(prefetch_of, **kw)
        taken = model._prefetched is not None and model._prefetched[0] == model._batch_key(batch)
        losses = model.compute_loss(batch, 150)
        assert model._prefetched is None                    # consumed or dropped, never kept for a later batch
        losses['optimization_loss'].backward()
        torch.cuda.synchronize()
        return taken, float(losses['optimization_loss']), {
            n: p.grad.detach().clone() for n, p in model.detection_model.named_parameters() if p.grad is not None}

    t0, l0, g0 = run()
    assert not t0
    host = dict(batch, vox_coords=batch['vox_coords'].cpu(), vox_features=batch['vox_features'].cpu())
    ev = torch.cuda.Event(); ev.record()
    for what, (taken, l, g) in (('default', run(batch)), ('ready=True', run(batch, ready=True)),
                                ('event', run(batch, ready=ev)), ('other batch', run(other))):
        assert taken == (what != 'other batch'), what
        assert l == l0, what
        bad = [n for n in g0 if not torch.equal(g0[n], g[n])]
        assert not bad, (what, bad[:5])
    # host tensors: the key is the host tensors' identity, so the step must be handed the same dict
    model.detection_model.zero_grad()
    model.prefetch(host, ready=True)
    assert model._prefetched[0] == model._batch_key(host)
    losses = model.compute_loss(host, 150)
    assert float(losses['optimization_loss']) == l0


def test_passthrough_inputs_give_the_same_gradients(monkeypatch):
    """Block inputs and encoder outputs have two consumers.  By default the second consumer takes the alias the first
    one (a convolution) hands back and that convolution's data-gradient kernel adds onto the other gradient in place;
    B2M_CONV_PASSTHROUGH=0 leaves the sum to an add kernel of autograd's.  Same loss bits, gradients equal to rounding
    (the order of one addition per element differs)."""
    from box2mask_amd import synth
    from box2mask_amd.config import scannet_config
    from box2mask_amd.model import Model
    monkeypatch.setenv('B2M_DETERMINISTIC', '1')
    batch = synth.make_batch(3, seed0=41, target_voxels=12000, pts_per_m2=8000.0)

    def run():
        torch.manual_seed(7)
        model = Model(scannet_config(), *synth.scannet_tables())
        model.train()
        losses = model.compute_loss(batch, 150)
        losses['optimization_loss'].backward()
        torch.cuda.synchronize()
        return float(losses['optimization_loss']), {
            n: p.grad.detach().clone() for n, p in model.detection_model.named_parameters() if p.grad is not None}
    l1, g1 = run()
    monkeypatch.setenv('B2M_CONV_PASSTHROUGH', '0')
    l0, g0 = run()
    assert l1 == l0
    assert g1.keys() == g0.keys() and len(g1) > 250
    worst = max(float((g1[n] - g0[n]).abs().max()) / max(float(g0[n].abs().max()), 1e-12) for n in g0)
    assert worst < 2e-4, worst


def test_torch_add_between_two_convolutions(monkeypatch):
    """A torch elementwise add on FEATURE tensors between two convolutions of this package (`box2mask_amd.nn` used as "ME" by
    a model other than SelectionNet).  AddBackward0 hands ONE gradient tensor to both of its inputs; one of them is the
    passed-through input of a convolution whose data-gradient kernel adds onto such gradients in place.  The mark that
    permits the in-place add names the backward node a gradient was made for (functional._own), so the tensor that went
    through the add is summed out of place and the other branch keeps its gradient: every gradient equals the run without
    passed-through inputs."""
    from box2mask_amd import functional as F_, synth
    from box2mask_amd.sparse import CoordinateManager
    monkeypatch.setenv('B2M_DETERMINISTIC', '1')
    b = synth.make_batch(2, seed0=3, target_voxels=6000, pts_per_m2=8000.0)
    m = CoordinateManager(b['vox_coords'].cuda(), reorder=True)
    rb = m.rulebook_same(0, 3)
    n, c = rb.n_out, 32

    def run(passthrough):
        torch.manual_seed(11)
        a = torch.randn(n, c, device='cuda', requires_grad=True)
        ws = [torch.nn.Parameter(torch.randn(27, c, c, device='cuda') * 0.05) for _ in range(4)]
        r1, r2 = torch.randn(n, c, device='cuda'), torch.randn(n, c, device='cuda')
        conv = lambda x, w, **kw: F_.sparse_conv(x, None, w, None, rb, rb, True, n, **kw)
        h = conv(a, ws[0])
        t = conv(a, ws[3])      # (made BEFORE the convolution below: the engine then runs that one's backward first, and an
        #                         in-place add onto the shared gradient would be read by this branch)
        if passthrough:
            y1, hp, _ = conv(h, ws[1], passthrough=True)      # the other consumers of h take the alias hp
        else:
            y1, hp = conv(h, ws[1]), h
        z = torch.add(hp, t)                                   # a torch operator between two package convolutions
        v = conv(z, ws[2])
        ((y1 * r1).sum() + (v * r2).sum()).backward()
        torch.cuda.synchronize()
        return [a.grad.clone()] + [w.grad.clone() for w in ws]
    g1 = run(True)
    g0 = run(False)
    for x1, x0 in zip(g1, g0):
        assert float((x1 - x0).abs().max()) <= 1e-5 * max(float(x0.abs().max()), 1e-12)
