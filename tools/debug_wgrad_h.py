import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from box2mask_amd import functional as F_, half_train as HT, _lib
from box2mask_amd.sparse import CoordinateManager
from test_gpu_ops import _scene
b = _scene()
m = CoordinateManager(b['vox_coords']); m.ensure_level(2)
HT.loss_scale[0] = 1.0
for (lvl, cin, cout) in ((0, 32, 32), (0, 64, 64), (0, 32, 128), (0, 128, 32), (0, 32, 96), (0, 96, 96), (1, 128, 128), (1, 16, 48), (1, 48, 16), (2, 256, 128)):
    rb = m.rulebook_same(lvl, 3); n = m.n(lvl)
    torch.manual_seed(1)
    x = torch.randn(n, cin, device='cuda').half(); dy = torch.randn(n, cout, device='cuda').half()
    for env in ({}, {'B2M_WGRAD_TRH': '0'}, {'B2M_WGRAD_TRH': '0', 'B2M_WGRAD_PIPE': '0'}, {'B2M_WGRAD_KPACK': '0'}, {'B2M_XCD_BALANCE': '0'}):
        for k_ in ('B2M_WGRAD_KPACK', 'B2M_WGRAD_PIPE', 'B2M_XCD_BALANCE', 'B2M_WGRAD_HANDLOADS', 'B2M_WGRAD_TRH'): os.environ.pop(k_, None)
        os.environ.update(env); _lib.reload_env()
        dwh = torch.zeros(27, cin, cout, device='cuda'); dwf = torch.zeros(27, cin, cout, device='cuda')
        HT._wgrad_h(x, dy, rb, 27, dwh, 0, 1.0)
        F_.wgrad_raw(x.float(), dy.float(), rb, 27, dwf, 0)
        torch.cuda.synchronize()
        if 'ref' not in dir() or ref_key != (lvl, cin, cout):
            ref, ref_key = dwf.clone(), (lvl, cin, cout)          # (the first variant's fp32 result = hand-issued flow kernel)
        print('      fp32 kernel of this variant vs the default fp32 kernel: %.3e' % float((dwf - ref).abs().max() / ref.abs().max()))
        e = float((dwh - dwf).abs().max() / dwf.abs().max())
        per_k = ((dwh - dwf).abs().amax((1, 2)) / dwf.abs().amax()).cpu().numpy()
        print('L%d %d->%d %-26s rel %.3e   bad offsets: %s' % (lvl, cin, cout, env, e, [i for i, v in enumerate(per_k) if v > 1e-3][:30]))
