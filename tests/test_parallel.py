"""N>1 host logic on the CPU with gloo, world_size 2: scene sharding, bucketed gradient all-reduce
(with and without backward overlap) and the SyncBN statistics merge."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from box2mask_amd.parallel import GradAllReduce, init_distributed, shard_scenes
    from box2mask_amd.functional import merge_bn_sums
    r, w = init_distributed('gloo')
    assert (r, w) == (rank, world)
    out = {}
    # --- scene sharding: disjoint, complete
    out['shard'] = shard_scenes(16, rank, world)
    # --- gradient all-reduce, overlap via hooks, several buckets
    torch.manual_seed(0)                                  # same weights on all ranks
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4))
    dp = GradAllReduce(list(net.parameters()), bucket_bytes=256, overlap=True)
    dp.broadcast_parameters()
    assert len(dp.buckets) >= 2
    torch.manual_seed(100 + rank)                         # different data per rank
    x = torch.randn(5, 8)
    for it in range(2):                                   # two steps: the hook state must reset
        net.zero_grad(set_to_none=(it == 1))
        net(x).pow(2).sum().backward()                    # the reference's loop: nothing between backward() and step()
    out['grads'] = [p.grad.clone() for p in net.parameters()]
    dp.all_reduce_mean()                                  # explicit call after the fact: must be a no-op
    out['grads_again'] = [p.grad.clone() for p in net.parameters()]
    # --- a parameter that receives no gradient keeps .grad None, the others are still averaged
    net3 = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4))
    net3.load_state_dict(net.state_dict())
    unused = torch.nn.Parameter(torch.ones(3))
    dp3 = GradAllReduce(list(net3.parameters()) + [unused], bucket_bytes=256, overlap=True)
    net3(x).pow(2).sum().backward()
    assert unused.grad is None
    out['grads3'] = [p.grad.clone() for p in net3.parameters()]
    # --- gradient arena: operators write into slots of one flat buffer, the all-reduce runs in place on bucket ranges
    from box2mask_amd.grad_arena import GradArena, grad_slot

    class ArenaLinear(torch.autograd.Function):           # stands in for the HIP operators (they need a GPU)
        @staticmethod
        def forward(ctx, inp, w):
            ctx.save_for_backward(inp, w)
            return inp @ w

        @staticmethod
        def backward(ctx, g):
            inp, w = ctx.saved_tensors
            dw = grad_slot(w)
            if dw is None:
                dw = torch.zeros_like(w)
            dw += inp.t() @ g                             # accumulate into the pre-zeroed slot
            return g @ w.t(), dw

    torch.manual_seed(1)
    ws = [torch.nn.Parameter(torch.randn(8, 8) * 0.3) for _ in range(5)]
    arena = GradArena(ws)
    dpa = GradAllReduce(ws, bucket_bytes=600, overlap=True, arena=arena)
    assert len(dpa.buckets) >= 2

    def run():
        arena.begin_pass()
        h = x
        for w_ in ws:
            h = torch.tanh(ArenaLinear.apply(h, w_))
        h.pow(2).sum().backward()
    run()
    base = arena.buffers[0].data_ptr()
    out['arena_in_place'] = all(w_.grad.data_ptr() == base + 4 * arena.offset[id(w_)] for w_ in ws)
    out['arena_g1'] = [w_.grad.clone() for w_ in ws]
    run()                                                 # gradients kept: second pass goes to the other buffer and adds up
    out['arena_g2'] = [w_.grad.clone() for w_ in ws]
    for w_ in ws:
        w_.grad = None
    run()                                                 # after zero_grad(set_to_none): buffer 0 again
    out['arena_g3'] = [w_.grad.clone() for w_ in ws]
    out['arena_ws'] = [w_.detach().clone() for w_ in ws]
    out['x'] = x
    # --- no-overlap path gives the same result
    net2 = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4))
    net2.load_state_dict(net.state_dict())
    dp2 = GradAllReduce(list(net2.parameters()), bucket_bytes=1 << 20, overlap=False)
    net2(x).pow(2).sum().backward()
    dp2.all_reduce_mean()
    out['grads2'] = [p.grad.clone() for p in net2.parameters()]
    # --- broadcast_parameters also carries the module buffers (BatchNorm running statistics, step counters): a rank-local
    # load_state_dict before the wrapper is built must not leave the ranks with different normalisation in eval mode
    bnm = torch.nn.BatchNorm1d(4)
    with torch.no_grad():
        bnm.running_mean.fill_(float(rank + 1)); bnm.running_var.fill_(float(2 * rank + 3)); bnm.num_batches_tracked.fill_(7 + rank)
        bnm.weight.fill_(float(rank))
    GradAllReduce(list(bnm.parameters()), buffers=list(bnm.buffers())).broadcast_parameters()
    out['bn_buffers'] = [bnm.running_mean.clone(), bnm.running_var.clone(), bnm.num_batches_tracked.clone().double(), bnm.weight.detach().clone()]
    # --- SyncBN statistics merge == statistics of the concatenated rows
    torch.manual_seed(7 + rank)
    feats = torch.randn(30 + 10 * rank, 6, dtype=torch.float64)
    sums = torch.cat([feats.sum(0), (feats * feats).sum(0)])
    gs, cnt = merge_bn_sums(sums, feats.shape[0], dist.group.WORLD)
    out['bn'] = (gs, cnt, feats)
    def plain(v):
        if torch.is_tensor(v):
            return v.detach().numpy().copy()
        if isinstance(v, (list, tuple)):
            return [plain(u) for u in v]
        return v
    q.put((rank, {k: plain(v) for k, v in out.items()}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_world_size_2_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = dict(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    def back(v):
        import numpy as np
        if isinstance(v, np.ndarray):
            return torch.from_numpy(v)
        if isinstance(v, list):
            return [back(u) for u in v]
        return v
    a, b = ({k: back(v) for k, v in res[r].items()} for r in (0, 1))
    assert sorted(a['shard'] + b['shard']) == list(range(16)) and not set(a['shard']) & set(b['shard'])
    # reference: mean of the two ranks' gradients computed in one process
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4))
    gs = []
    for x in (a['x'], b['x']):
        net.zero_grad()
        net(x).pow(2).sum().backward()
        gs.append([p.grad.clone() for p in net.parameters()])
    for i in range(len(gs[0])):
        ref = (gs[0][i] + gs[1][i]) / 2
        for r in (a, b):
            assert torch.allclose(r['grads'][i], ref, atol=1e-6)
            assert torch.allclose(r['grads2'][i], ref, atol=1e-6)
            assert torch.equal(r['grads_again'][i], r['grads'][i])
            assert torch.allclose(r['grads3'][i], ref, atol=1e-6)
    # arena path: in place, equal to the single-process mean; kept gradients add up; buffer reuse after zero_grad
    assert a['arena_in_place'] and b['arena_in_place']
    ws = [w.clone().requires_grad_(True) for w in a['arena_ws']]
    gsum = None
    for x in (a['x'], b['x']):
        h = x
        for w in ws:
            h = torch.tanh(h @ w)
        g = torch.autograd.grad(h.pow(2).sum(), ws)
        gsum = g if gsum is None else [u + v for u, v in zip(gsum, g)]
    for i in range(len(ws)):
        ref = gsum[i] / 2
        for r in (a, b):
            assert torch.allclose(r['arena_g1'][i], ref, atol=1e-6)
            assert torch.allclose(r['arena_g2'][i], 2 * ref, atol=1e-6)
            assert torch.allclose(r['arena_g3'][i], ref, atol=1e-6)
    for r in (a, b):                                     # rank 0's values everywhere
        assert torch.equal(r['bn_buffers'][0], torch.full((4,), 1.0)) and torch.equal(r['bn_buffers'][1], torch.full((4,), 3.0))
        assert float(r['bn_buffers'][2]) == 7.0 and torch.equal(r['bn_buffers'][3], torch.zeros(4))
    feats = torch.cat([a['bn'][2], b['bn'][2]])
    for r in (a, b):
        s, cnt, _ = r['bn']
        assert cnt == feats.shape[0]
        mean = s[:6] / cnt; var = s[6:] / cnt - mean * mean
        assert torch.allclose(mean, feats.mean(0), atol=1e-12) and torch.allclose(var, feats.var(0, unbiased=False), atol=1e-12)


def test_single_process_is_a_noop():
    from box2mask_amd.parallel import GradAllReduce, shard_scenes
    net = torch.nn.Linear(3, 2)
    dp = GradAllReduce(list(net.parameters()))
    net(torch.ones(1, 3)).sum().backward()
    g = net.weight.grad.clone()
    dp.all_reduce_mean()
    assert torch.equal(net.weight.grad, g)
    assert shard_scenes(5, 0, 1) == [0, 1, 2, 3, 4]


def test_arena_slot_is_handed_out_once_per_pass():
    """Two forward passes summed into one backward, and a weight shared by two layers: the backward operator of one
    parameter runs twice inside ONE backward() call; the second call must not receive the same arena slot (autograd would
    add the view to itself: exactly 2x the gradient)."""
    from box2mask_amd.grad_arena import GradArena, grad_slot

    class ArenaLinear(torch.autograd.Function):
        @staticmethod
        def forward(ctx, inp, w):
            ctx.save_for_backward(inp, w)
            return inp @ w

        @staticmethod
        def backward(ctx, g):
            inp, w = ctx.saved_tensors
            dw = grad_slot(w)
            if dw is None:
                dw = torch.zeros_like(w)
            dw += inp.t() @ g
            return g @ w.t(), dw

    torch.manual_seed(3)
    ws = [torch.nn.Parameter(torch.randn(6, 6) * 0.4) for _ in range(3)]
    xa, xb = torch.randn(5, 6), torch.randn(7, 6)
    arena = GradArena(ws)

    def net(x, op):
        h = x
        for w_ in ws:
            h = torch.tanh(op(h, w_))
        return torch.tanh(op(h, ws[0])).pow(2).sum()          # ws[0] is shared by the first and the last layer

    def forward(x):
        arena.begin_pass()
        return net(x, ArenaLinear.apply)
    (forward(xa) + forward(xb)).backward()
    ref = torch.autograd.grad(net(xa, torch.matmul) + net(xb, torch.matmul), ws)
    for w_, r in zip(ws, ref):
        assert torch.allclose(w_.grad, r, atol=1e-6), float((w_.grad / r).mean())
    # the normal loop still lands in the arena
    for w_ in ws:
        w_.grad = None
    arena.begin_pass()
    h = xa
    for w_ in ws:
        h = torch.tanh(ArenaLinear.apply(h, w_))
    h.sum().backward()
    base = arena.buffers[arena.current].data_ptr()
    assert all(w_.grad.data_ptr() == base + 4 * arena.offset[id(w_)] for w_ in ws)
