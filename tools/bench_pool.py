"""Segment pooling (avg) forward / backward on a level-0-sized tensor: ms and TB/s of algorithmic bytes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from box2mask_amd import functional as F_

n, c, nseg = 1201955, 96, 9752
torch.manual_seed(0)
ids = torch.sort(torch.randint(0, nseg, (n,), device='cuda'))[0]
ids = ids[torch.randperm(n, device='cuda')[:n]].contiguous() if os.environ.get('SHUFFLE') else ids
x = torch.randn(n, c, device='cuda', requires_grad=True)
s = torch.cuda.Event(enable_timing=True); m = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
best = [1e9, 1e9]
for _ in range(6):
    x.grad = None
    s.record(); y = F_.segment_pool(x, ids, nseg, 'avg'); m.record(); y.backward(torch.ones_like(y)); e.record()
    torch.cuda.synchronize()
    best = [min(best[0], s.elapsed_time(m)), min(best[1], m.elapsed_time(e))]
b = 4.0 * n * c
print('segment mean %d x %d -> %d: fwd %.3f ms (%.2f TB/s), bwd %.3f ms (%.2f TB/s)' % (n, c, nseg, best[0], b / best[0] / 1e9, best[1], b / best[1] / 1e9))
