"""conv_fwd_coop_kernel against conv_fwd_flow_kernel on the benchmark's maps: same inputs, outputs compared (the two differ in
the summation order over offsets only), three runs of the cooperative kernel compared with each other (a race shows as
run-to-run differences far above rounding)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from box2mask_amd import synth, functional as F_, _lib
from box2mask_amd.sparse import CoordinateManager
bs = int(os.environ.get('BS', '2'))
b = synth.make_batch(bs, seed0=0)
m = CoordinateManager(b['vox_coords'], reorder=True)
rb0 = m.rulebook_same(0, 3); m.ensure_level(4); rb1 = m.rulebook_same(1, 3); rb2 = m.rulebook_same(2, 3)
cases = [('L0 96->96', rb0, 96, 0, 96), ('L0 128(96|32)->96', rb0, 96, 32, 96), ('L0 32->32', rb0, 32, 0, 32), ('L1 96->96', rb1, 96, 0, 96),
         ('L1 128->128', rb1, 128, 0, 128), ('L1 64->64', rb1, 64, 0, 64), ('L2 128->128', rb2, 128, 0, 128)]
def run(env, fn):
    for k in ('B2M_CONV_COOP', 'B2M_CONV_COOP_MIN_TILES'): os.environ.pop(k, None)
    os.environ.update(env); _lib.reload_env()
    return fn()
ok = True
for name, rb, c1, c2, co in cases:
    torch.manual_seed(1)
    x1 = torch.randn(rb.n_in, c1, device='cuda'); x2 = torch.randn(rb.n_in, c2, device='cuda') if c2 else None
    w = torch.randn(27, c1 + c2, co, device='cuda') * 0.05
    wp = F_.weight_pack(w)
    f = lambda: F_.conv_raw(x1, x2, wp, 27, None, rb, rb.n_out, co).clone()
    y0 = run({}, f)
    ys = [run({'B2M_CONV_COOP': '1', 'B2M_CONV_COOP_MIN_TILES': '1'}, f) for _ in range(3)]
    torch.cuda.synchronize()
    scale = float(y0.abs().max())
    e = [float((y - y0).abs().max()) / scale for y in ys]
    rr = max(float((ys[0] - ys[1]).abs().max()), float((ys[0] - ys[2]).abs().max())) / scale
    good = max(e) < 2e-5 and rr < 2e-5
    ok &= good
    print('%-22s rows %8d  coop vs flow %.2e %.2e %.2e   run-to-run %.2e  %s' % (name, rb.n_out, e[0], e[1], e[2], rr, 'ok' if good else 'MISMATCH'))
print('ALL OK' if ok else 'FAILED')
sys.exit(0 if ok else 1)
