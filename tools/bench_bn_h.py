"""The half BatchNorm kernels alone on the chip: GB/s of each pass (algorithmic bytes / time) beside the fp32 kernels' on the same shape.
    python tools/bench_bn_h.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from box2mask_amd import functional as F_, half_train as HT


def timeit(f, n=20):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3


for n, c in ((1290000, 96), (1290000, 32), (420000, 128), (420000, 64), (110000, 256), (110000, 128), (26000, 256)):
    line = '%8d x %3d' % (n, c)
    for half in (True, False):
        dt = torch.float16 if half else torch.float32
        esz = 2 if half else 4
        x = torch.randn(n, c, device='cuda').to(dt).requires_grad_(True)
        res = torch.randn(n, c, device='cuda').to(dt)
        g = torch.ones(c, device='cuda', requires_grad=True); b = torch.zeros(c, device='cuda', requires_grad=True)
        rm, rv = torch.zeros(c, device='cuda'), torch.ones(c, device='cuda')
        gy = torch.randn(n, c, device='cuda').to(dt)
        if half:
            fwd = lambda: HT.batch_norm(x, g, b, rm, rv, 0.1, 1e-5, None, True)
        else:
            fwd = lambda: F_.batch_norm(x, g, b, rm, rv, True, 0.1, 1e-5, None, True, False, None)
        t_f = timeit(fwd)
        y = fwd()

        def bwd():
            x.grad = None; g.grad = None; b.grad = None
            y.backward(gy, retain_graph=True)
        t_b = timeit(bwd)
        bytes_f = 3 * n * c * esz          # stats: read x; apply: read x, write y
        bytes_b = 7 * n * c * esz          # reduce: dy, x, y; apply: dy, x, y -> dx
        line += ' | %s fwd %6.1f us %5.2f TB/s  bwd %6.1f us %5.2f TB/s' % ('half' if half else 'fp32', t_f * 1e6, bytes_f / t_f / 1e12, t_b * 1e6, bytes_b / t_b / 1e12)
    print(line, flush=True)
