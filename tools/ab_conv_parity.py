"""Two settings of the convolution switches on the same inputs (forward, two sources, accumulate, bias): outputs compared, three
runs of variant B compared with each other (a race shows as run-to-run differences).
    A="B2M_CONV_CHAIN=0" B="B2M_CONV_CHAIN=1" python tools/ab_conv_parity.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from box2mask_amd import synth, functional as F_, _lib
from box2mask_amd.sparse import CoordinateManager
bs = int(os.environ.get('BS', '8'))
b = synth.make_batch(bs, seed0=0)
m = CoordinateManager(b['vox_coords'], reorder=True)
rb0 = m.rulebook_same(0, 3); m.ensure_level(4); rb1 = m.rulebook_same(1, 3); rb2 = m.rulebook_same(2, 3)
cases = [('L0 96->96', rb0, 96, 0, 96), ('L0 128(96|32)->96', rb0, 96, 32, 96), ('L0 32->32', rb0, 32, 0, 32), ('L1 96->96', rb1, 96, 0, 96),
         ('L1 128->128', rb1, 128, 0, 128), ('L1 64->64', rb1, 64, 0, 64), ('L2 128->128', rb2, 128, 0, 128)]
def env_of(s): return dict(kv.split('=') for kv in s.split(',') if kv)
EA, EB = env_of(os.environ.get('A', '')), env_of(os.environ.get('B', ''))
def run(env, fn):
    for k in set(EA) | set(EB): os.environ.pop(k, None)
    os.environ.update(env); _lib.reload_env()
    return fn()
ok = True
for name, rb, c1, c2, co in cases:
    torch.manual_seed(1)
    x1 = torch.randn(rb.n_in, c1, device='cuda'); x2 = torch.randn(rb.n_in, c2, device='cuda') if c2 else None
    w = torch.randn(27, c1 + c2, co, device='cuda') * 0.05
    bias = torch.randn(co, device='cuda'); y_init = torch.randn(rb.n_out, co, device='cuda')
    wp = F_.weight_pack(w)
    for mode in ('plain', 'bias+accumulate'):
        def f():
            if mode == 'plain': return F_.conv_raw(x1, x2, wp, 27, None, rb, rb.n_out, co).clone()
            return F_.conv_raw(x1, x2, wp, 27, bias, rb, rb.n_out, co, out=y_init.clone(), accumulate=True)
        y0 = run(EA, f)
        ys = [run(EB, f) for _ in range(3)]
        torch.cuda.synchronize()
        scale = float(y0.abs().max())
        e = max(float((y - y0).abs().max()) for y in ys) / scale
        rr = max(float((ys[0] - ys[1]).abs().max()), float((ys[0] - ys[2]).abs().max())) / scale
        good = e < 2e-5 and rr < 2e-5
        ok &= good
        print('%-20s %-16s rows %8d  B vs A %.2e   B run-to-run %.2e  %s' % (name, mode, rb.n_out, e, rr, 'ok' if good else 'MISMATCH'))
print('ALL OK' if ok else 'FAILED')
sys.exit(0 if ok else 1)
