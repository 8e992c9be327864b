"""BatchNorm kernels on level-0-sized tensors: achieved HBM bandwidth per kernel (algorithmic bytes / HIP-event time)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from box2mask_amd import _lib, functional as F_

rec = []
def hook(name, a, meta=None):
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True); s.record()
    def done():
        e.record(); rec.append((name, s, e))
    return done

for n, c in ((1201955, 96), (1201955, 32), (290828, 96), (68925, 128)):
    x = torch.randn(n, c, device='cuda', requires_grad=True)
    res = torch.randn(n, c, device='cuda', requires_grad=True)
    g, b = torch.ones(c, device='cuda', requires_grad=True), torch.zeros(c, device='cuda', requires_grad=True)
    rm, rv = torch.zeros(c, device='cuda'), torch.ones(c, device='cuda')
    for with_res in (False, True):
        for it in range(3):
            rec.clear()
            _lib.set_hook(hook if it == 2 else None)
            y = F_.batch_norm(x, g, b, rm, rv, True, residual=res if with_res else None, relu=True)
            y.backward(torch.ones_like(y))
            torch.cuda.synchronize()
        _lib.set_hook(None)
        T = 4.0 * n * c
        alg = {'b2m_bn_stats_finalize': T, 'b2m_bn_apply': T * (3 if with_res else 2),
               'b2m_bn_bwd_reduce': T * (3 if with_res else 2), 'b2m_bn_bwd_apply': T * ((3 if with_res else 2) + (2 if with_res else 1))}
        line = '%8d x %3d %s' % (n, c, 'res ' if with_res else 'nores')
        for name, s, e in rec:
            ms = s.elapsed_time(e)
            line += ' | %s %.3f ms %.2f TB/s' % (name[7:], ms, alg.get(name, 0) / ms / 1e9)
        print(line)
