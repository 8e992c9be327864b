"""A/B of conv_wgrad_flow_kernel with hand-issued loads against the hipcc-tracked variant on the same tensors, three runs per
shape (a race shows as run-to-run differences): PYTHONPATH=. python tools/ab_wgrad_handloads.py  (GPU)."""
import os, sys, zlib
import numpy as np, torch
sys.path.insert(0, 'tests')
from test_gpu_ops import _scene
from box2mask_amd import functional as F_, _lib
from box2mask_amd.sparse import CoordinateManager
b = _scene()
m = CoordinateManager(b['vox_coords'])
level = 0
rb = m.rulebook_same(level, 3); K = 27; n = m.n(level)
print('n', n)
torch.manual_seed(1)
def run(x, dy, cin_total, ci0, hl):
    os.environ['B2M_WGRAD_HANDLOADS'] = str(hl); _lib.reload_env()
    dw = torch.zeros(K, cin_total, dy.shape[1], device='cuda')
    F_.wgrad_raw(x, dy, rb, K, dw, ci0)
    torch.cuda.synchronize()
    return dw
for (c1, c2, cout) in [(96, 0, 96), (96, 32, 96), (96, 32, 96), (64, 0, 64), (128, 0, 128)]:
    x = torch.randn(n, c1 + c2, device='cuda'); dy = torch.randn(n, cout, device='cuda')
    x1 = x[:, :c1].contiguous()
    for rep in range(3):
        a = run(x1, dy, c1 + c2, 0, 1); r = run(x1, dy, c1 + c2, 0, 0)
        d = (a - r).abs()
        sc = float(r.abs().max())
        bad = (d > 1e-3 * sc)
        print((c1, c2, cout), 'rep', rep, 'max rel', float(d.max()) / sc, 'bad', int(bad.sum()), 'nan', int(torch.isnan(a).sum()))
        if bad.any():
            idx = bad.nonzero()
            print('  k', sorted(set(idx[:, 0].tolist()))[:30])
            print('  ci', sorted(set(idx[:, 1].tolist()))[:100])
            print('  co', sorted(set(idx[:, 2].tolist()))[:100])
    if c2:
        x2 = x[:, c1:].contiguous()
        a = run(x2, dy, c1 + c2, c1, 1); r = run(x2, dy, c1 + c2, c1, 0)
        print('  x2 part', float((a - r).abs().max()) / float(r.abs().max()))
