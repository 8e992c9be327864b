"""Debug: compare GPU gradients and fp32-oracle gradients against an fp64 oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from box2mask_amd import synth
from box2mask_amd.config import scannet_config
from box2mask_amd.detection_net import SelectionNet
from box2mask_amd import nn as ME
from oracle import unet_ref

def rel(a, b):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)

cfg = scannet_config()
valid, _, _, is_fg = synth.scannet_tables()
torch.manual_seed(0)
net = SelectionNet(cfg, 'cuda', valid, is_fg, out_channels=[96, 96, 6]).cuda().train()
batch = synth.make_batch(8, seed0=4, target_voxels=2500, pts_per_m2=6000.0)
S_ = batch['input_location'].shape[0]
p_cpu = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
sin = ME.SparseTensor(batch['vox_features'], batch['vox_coords'])
out = net(sin, batch['pooling_ids'].cuda(), S_)
heads = ['mlp_offsets', 'mlp_bounds', 'mlp_bb_scores', 'mlp_semantics']
torch.manual_seed(1)
gws = {h: torch.randn(out[h].F.shape) for h in heads}
(sum((out[h].F * gws[h].cuda()).sum() for h in heads)).backward()
torch.cuda.synchronize()
res = {}
for dt in (torch.float32, torch.float64):
    p = {k: (v.to(dt).clone().requires_grad_(True) if v.is_floating_point() and 'running' not in k else (v.to(dt) if v.is_floating_point() else v))
         for k, v in p_cpu.items()}
    o = unet_ref.forward(p, batch['vox_coords'].numpy(), batch['vox_features'].to(dt), batch['pooling_ids'], cfg, training=True, n_segments=S_)
    (sum((o[h] * gws[h].to(dt)).sum() for h in heads)).backward()
    res[dt] = (p, o)
p32, o32 = res[torch.float32]; p64, o64 = res[torch.float64]
print('forward: gpu-vs-64 / oracle32-vs-64')
for h in heads:
    print('  %-14s %.3e %.3e' % (h, rel(out[h].F, o64[h]), rel(o32[h], o64[h])))
rows = []
for name, prm in net.named_parameters():
    g64 = p64[name].grad
    rows.append((rel(prm.grad, g64), rel(p32[name].grad, g64), float(g64.abs().max()), name))
rows.sort(reverse=True)
print('grads: gpu-vs-64, oracle32-vs-64, max|g|, name')
for r in rows[:25]:
    print('  %.3e %.3e %.3e %s' % r)
print('median gpu err %.3e, median oracle32 err %.3e' % (sorted(r[0] for r in rows)[len(rows)//2], sorted(r[1] for r in rows)[len(rows)//2]))
