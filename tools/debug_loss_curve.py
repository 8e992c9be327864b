"""How far apart are two 30-step Adam trajectories of this network anyway?  fp32 from the same weights | fp32 from weights perturbed
by a relative 5e-4 (one half rounding) and 1e-5 | half (three forms of its weight gradient).  The yardstick for
tests/test_gpu_half_train.py::test_half_training_loss_curve_follows_fp32."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from box2mask_amd import synth, _lib
from box2mask_amd.config import scannet_config
from box2mask_amd.model import Model
os.environ['B2M_DETERMINISTIC'] = '1'
batch = synth.make_batch(8, seed0=60, target_voxels=6000, pts_per_m2=6000.0)
STEPS = int(os.environ.get('STEPS', '30'))


def run(half, perturb=0.0, pseed=0, env=None, scale=1024.0):
    for k_ in ('B2M_WGRAD_TRH', 'B2M_WGRAD_PIPE'): os.environ.pop(k_, None)
    os.environ.update(env or {}); _lib.reload_env()
    torch.manual_seed(7)
    model = Model(scannet_config(half_training=half, half_loss_scale=scale), *synth.scannet_tables())
    if perturb:
        g = torch.Generator(device='cuda'); g.manual_seed(pseed)
        with torch.no_grad():
            for p in model.parameters():
                p.mul_(1.0 + perturb * (2 * torch.rand(p.shape, device=p.device, generator=g) - 1))
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
    out = []
    for _ in range(STEPS):
        opt.zero_grad()
        ld = model.compute_loss(batch, 150)
        ld['optimization_loss'].backward()
        opt.step()
        out.append(float(ld['optimization_loss'].detach()))
    return np.array(out)


ref = run(False)
print('fp32                     ', ' '.join('%.3f' % v for v in ref[::3]), flush=True)
for name, kw in (('fp32 again', {}), ('fp32 weights * (1 +- 1e-5)', dict(perturb=1e-5)), ('fp32 weights * (1 +- 5e-4)', dict(perturb=5e-4)),
                 ('fp32 weights * (1 +- 5e-4) #2', dict(perturb=5e-4, pseed=1)), ('fp32 weights * (1 +- 5e-4) #3', dict(perturb=5e-4, pseed=2)),
                 ('half (f16 MFMA wgrad)', dict(half=True)), ('half (cvt wgrad)', dict(half=True, env={'B2M_WGRAD_TRH': '0'})),
                 ('half (plain wgrad)', dict(half=True, env={'B2M_WGRAD_TRH': '0', 'B2M_WGRAD_PIPE': '0'})),
                 ('half, loss scale 128', dict(half=True, scale=128.0)), ('half, loss scale 8192', dict(half=True, scale=8192.0))):
    half = kw.pop('half', False)
    c = run(half, **kw)
    d = np.abs(c - ref) / ref
    print('%-32s %s | max %.3f mean %.3f tail5 %.3f' % (name, ' '.join('%.3f' % v for v in c[::3]), d.max(), d.mean(),
          abs(c[-5:].mean() - ref[-5:].mean()) / ref[-5:].mean()), flush=True)
