"""Which switch of the default mode moves the network-level gradients (tests/test_gpu_default_mode.py)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from box2mask_amd import synth, _lib, nn as ME, functional as F_
from box2mask_amd.config import scannet_config
from box2mask_amd.model import Model
from oracle import unet_ref

HEADS = ['mlp_offsets', 'mlp_bounds', 'mlp_bb_scores', 'mlp_semantics']
W = {'mlp_offsets': 3, 'mlp_bounds': 3, 'mlp_bb_scores': 1, 'mlp_semantics': 20}
EVAL = os.environ.get('EVAL_BN', '1') == '1'


def rel(a, b):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)


cfg = scannet_config()
torch.manual_seed(11)
model = Model(cfg, *synth.scannet_tables())
net = model.detection_model
batch = synth.make_batch(6, seed0=300, target_voxels=5000, pts_per_m2=8000.0)
S_ = batch['input_location'].shape[0]
if EVAL:
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, ME.MinkowskiBatchNorm):
                m.bn.momentum = 1.0
                m.bn.weight.uniform_(0.6, 1.4); m.bn.bias.uniform_(-0.3, 0.3)
        net.train()
        net(ME.SparseTensor(batch['vox_features'], batch['vox_coords']), batch['pooling_ids'].cuda(), S_)
    net.eval()
sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
torch.manual_seed(20)
gws = {h: torch.randn(S_, W[h]) for h in HEADS}
p = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and 'running' not in k else v) for k, v in sd.items()}
out = unet_ref.forward(p, batch['vox_coords'].numpy(), batch['vox_features'], batch['pooling_ids'], cfg, training=not EVAL, n_segments=S_)
sum((out[h] * gws[h]).sum() for h in HEADS).backward()

variants = [('default', {}), ('one stream', {'B2M_WGRAD_STREAM': '0'}), ('no passthrough', {'B2M_CONV_PASSTHROUGH': '0'}),
            ('one stream, no passthrough', {'B2M_WGRAD_STREAM': '0', 'B2M_CONV_PASSTHROUGH': '0'}),
            ('deterministic', {'B2M_DETERMINISTIC': '1'}), ('no split', {'B2M_CONV_TARGET': '0'}),
            ('plain wgrad', {'B2M_WGRAD_PIPE': '0'}), ('default again', {})]
keys = set(k for _, e in variants for k in e)
for name, env in variants:
    for k in keys:
        os.environ.pop(k, None)
    os.environ.update(env)
    _lib.reload_env()
    net.load_state_dict(sd)
    for q in net.parameters():
        q.grad = None
    o = net(ME.SparseTensor(batch['vox_features'], batch['vox_coords']), batch['pooling_ids'].cuda(), S_)
    sum((o[h].F * gws[h].cuda()).sum() for h in HEADS).backward()
    torch.cuda.synchronize()
    rows = sorted(((rel(q.grad, p[n].grad), n) for n, q in net.named_parameters() if q.grad is not None), reverse=True)
    fwd = max(rel(o[h].F, out[h]) for h in HEADS)
    print('%-28s fwd %.2e  grads: n=%d worst %.3e %s | #>1e-3: %d | median %.2e' % (
        name, fwd, len(rows), rows[0][0], rows[0][1], sum(1 for r in rows if r[0] > 1e-3), rows[len(rows) // 2][0]), flush=True)
    if name == 'default':
        for r in rows[:12]:
            print('     %.3e %s' % r)
