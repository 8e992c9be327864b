"""Per-layer distance between a half-precision training pass (half_train.py) and the fp32 pass on the same weights and batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from box2mask_amd import synth, nn as ME
from box2mask_amd.config import scannet_config
from box2mask_amd.detection_net import SelectionNet
os.environ.setdefault('B2M_DETERMINISTIC', '1')
cfg = scannet_config()
valid, _, _, is_fg = synth.scannet_tables()
torch.manual_seed(3)
net = SelectionNet(cfg, 'cuda', valid, is_fg, out_channels=[96, 96, 6]).cuda()
net.train()
batch = synth.make_batch(int(os.environ.get('SCENES', '32')), seed0=40, target_voxels=int(os.environ.get('VOX', '2000')), pts_per_m2=6000.0)
S_ = batch['input_location'].shape[0]
sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
res = {}
for half in (False, True):
    net.load_state_dict(sd)
    net.half_training = half
    net._trace = {}
    out = net(ME.SparseTensor(batch['vox_features'], batch['vox_coords']), batch['pooling_ids'].cuda(), S_)
    res[half] = ({k: v.detach().float().clone() for k, v in net._trace.items()}, {h: out[h].F.detach().clone() for h in out})
for name in res[False][0]:
    a, b = res[True][0][name], res[False][0][name]
    print('%-14s rows %7d  max-rel %.3e  rms-rel %.3e' % (name, a.shape[0], float((a - b).abs().max() / b.abs().max()),
                                                         float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt())))
for h in res[False][1]:
    a, b = res[True][1][h], res[False][1][h]
    print('head %-14s max-rel %.3e rms-rel %.3e' % (h, float((a - b).abs().max() / b.abs().max()), float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt())))

# ---- gradients: per parameter, in the network's order, for a few loss scales
heads = ['mlp_offsets', 'mlp_bounds', 'mlp_bb_scores', 'mlp_semantics']
gws = None
def grads(half, scale=1024.0):
    global gws
    net.load_state_dict(sd)
    net.half_training = half
    net.cfg.half_loss_scale = scale
    net._trace = None
    for p in net.parameters():
        p.grad = None
    out = net(ME.SparseTensor(batch['vox_features'], batch['vox_coords']), batch['pooling_ids'].cuda(), S_)
    if gws is None:
        gws = {h: torch.randn(out[h].F.shape, device='cuda') for h in heads}
    sum(((out[h].F - gws[h]) ** 2).mean() for h in heads).backward()
    torch.cuda.synchronize()
    return {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
g32 = grads(False)
for scale in (1.0, 1024.0, 65536.0):
    g16 = grads(True, scale)
    rel = {n: float((g16[n] - g32[n]).abs().max() / g32[n].abs().max().clamp_min(1e-30)) for n in g32}
    a = torch.cat([g16[n].reshape(-1).double() for n in g32]); b = torch.cat([g32[n].reshape(-1).double() for n in g32])
    print('loss scale %g: cosine %.5f' % (scale, float((a * b).sum() / (a.norm() * b.norm()))))
    names = list(g32)
    for n in names[:12] + names[60:70] + names[140:150] + names[-40:-24]:
        print('   %-42s rel %.3e   |g32| %.3e |g16| %.3e' % (n, rel[n], float(g32[n].abs().max()), float(g16[n].abs().max())))

# ---- the same comparison with the ReLU decisions of the fp32 pass replayed in the half pass (every BatchNorm-ReLU and ReLU)
from box2mask_amd import functional as F_, half_train as HT
os.environ['B2M_BN_PAIR'] = '0'
masks, mode, idx = [], ['record'], [0]
bn32, bn16, relu32 = F_.batch_norm, HT.batch_norm, F_.relu
def take(z):
    if mode[0] == 'record':
        m = z.detach() > 0
        masks.append(m)
        return m
    m = masks[idx[0]]; idx[0] += 1
    assert m.shape == z.shape, (m.shape, z.shape)
    return m
def p_bn32(x, gamma, beta, rm, rv, training, momentum=0.1, eps=1e-5, residual=None, relu=False, sync=False, count_key=None):
    z = bn32(x, gamma, beta, rm, rv, training, momentum, eps, residual, False, sync, count_key)
    return z * take(z).to(z.dtype) if relu else z
def p_bn16(x, gamma, beta, rm, rv, momentum, eps, residual=None, relu=False):
    z = bn16(x, gamma, beta, rm, rv, momentum, eps, residual, False)
    return z * take(z).to(z.dtype) if relu else z
def p_relu(x):
    return x * take(x).to(x.dtype)
F_.batch_norm, HT.batch_norm, F_.relu = p_bn32, p_bn16, p_relu
import box2mask_amd.nn as nnmod
mode[0] = 'record'; masks.clear()
g32 = grads(False)
mode[0] = 'replay'; idx[0] = 0
g16 = grads(True, 1024.0)
print('ReLU decisions recorded %d, replayed %d' % (len(masks), idx[0]))
rel = {n: float((g16[n] - g32[n]).abs().max() / g32[n].abs().max().clamp_min(1e-30)) for n in g32}
a = torch.cat([g16[n].reshape(-1).double() for n in g32]); b = torch.cat([g32[n].reshape(-1).double() for n in g32])
import numpy as np
v = np.array(sorted(rel.values()))
print('REPLAYED masks: cosine %.5f  median %.3e  p95 %.3e  worst %.3e' % (float((a * b).sum() / (a.norm() * b.norm())), np.median(v), np.percentile(v, 95), v[-1]))
for n in list(g32)[:8] + list(g32)[100:108]:
    print('   %-42s rel %.3e' % (n, rel[n]))
