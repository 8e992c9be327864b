"""Default vs deterministic mode on the block chain of tests/test_gpu_default_mode.py, per parameter, plus single-switch variants."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
from box2mask_amd import synth, _lib, nn as ME, functional as F_
from box2mask_amd.resnet import BasicBlock

b = synth.make_batch(3, seed0=70, target_voxels=30000, pts_per_m2=8000.0)


class Chain(nn.Module):
    def __init__(self):
        super().__init__()
        short = nn.Sequential(ME.MinkowskiConvolution(32, 64, kernel_size=1, dimension=3), ME.MinkowskiBatchNorm(64))
        self.b0 = BasicBlock(32, 64, downsample=short, dimension=3)
        self.b1 = BasicBlock(64, 64, dimension=3)
        self.down = ME.MinkowskiConvolution(64, 64, kernel_size=2, stride=2, dimension=3)
        self.bn = ME.MinkowskiBatchNorm(64)
        self.b2 = BasicBlock(64, 64, dimension=3)
        self.up = ME.MinkowskiConvolutionTranspose(64, 32, kernel_size=2, stride=2, dimension=3)
        self.bnu = ME.MinkowskiBatchNorm(32)
        short2 = nn.Sequential(ME.MinkowskiConvolution(96, 48, kernel_size=1, dimension=3), ME.MinkowskiBatchNorm(48))
        self.b3 = BasicBlock(96, 48, downsample=short2, dimension=3)

    def forward(self, x):
        e = self.b1(self.b0(x))
        d, e = self.down(e, passthrough=True)
        d = d.new(self.bn.apply_bn(d.F, relu=True))
        d = self.b2(d)
        u = self.up(d)
        u = u.new(self.bnu.apply_bn(u.F, relu=True))
        y = self.b3(ME.cat(u, e))
        ME.flush_batch_counters()
        return y


def rel(a, b_):
    a = a.detach().double(); b_ = b_.detach().double()
    return float((a - b_).abs().max()) / max(float(b_.abs().max()), 1e-30)


def run(env):
    for k in ('B2M_DETERMINISTIC', 'B2M_WGRAD_STREAM', 'B2M_CONV_PASSTHROUGH', 'B2M_CONV_STATS', 'B2M_WGRAD_PIPE', 'B2M_CONV_TARGET',
              'B2M_WGRAD_KPACK', 'B2M_BN_PAIR', 'B2M_BN_SMALL_ROWS', 'B2M_CONV_PIPE', 'B2M_XCD_BALANCE'):
        os.environ.pop(k, None)
    os.environ.update(env)
    _lib.reload_env()
    torch.manual_seed(9)
    net = Chain().cuda().train()
    torch.manual_seed(10)
    sin = ME.SparseTensor(torch.randn(b['vox_coords'].shape[0], 32), b['vox_coords'])
    sin.F.requires_grad_(True)
    F_.packed_weights.begin_pass()
    y = net(sin).F
    gy = torch.randn(y.shape, device='cuda')
    (y * gy).sum().backward()
    torch.cuda.synchronize()
    return y.detach().clone(), sin.F.grad.clone(), {n: p.grad.clone() for n, p in net.named_parameters()}


base = run({'B2M_DETERMINISTIC': '1'})
for name, env in (('deterministic again', {'B2M_DETERMINISTIC': '1'}), ('default', {}), ('one stream', {'B2M_WGRAD_STREAM': '0'}),
                  ('no passthrough', {'B2M_CONV_PASSTHROUGH': '0'}), ('stats pass', {'B2M_CONV_STATS': '0'}),
                  ('plain wgrad', {'B2M_WGRAD_PIPE': '0'}), ('no kpack', {'B2M_WGRAD_KPACK': '0'}), ('no split', {'B2M_CONV_TARGET': '0'}),
                  ('no pair', {'B2M_BN_PAIR': '0'}), ('plain conv', {'B2M_CONV_PIPE': '0'}),
                  ('one stream + plain wgrad + no split', {'B2M_WGRAD_STREAM': '0', 'B2M_WGRAD_PIPE': '0', 'B2M_CONV_TARGET': '0'})):
    y, dx, g = run(env)
    rows = sorted(((rel(g[n], base[2][n]), n) for n in g), reverse=True)
    print('%-36s y %.2e dx max %.2e L2 %.2e | worst %s' % (name, rel(y, base[0]), rel(dx, base[1]),
          float((dx - base[1]).norm() / base[1].norm()), ' '.join('%s %.1e' % (n, e) for e, n in rows[:6])), flush=True)
    if name == 'default':
        for e, n in rows:
            print('      %.2e %s' % (e, n))

# ---- ground truth: the chain on the CPU oracle in fp64, ReLU decisions taken from the device run
import numpy as np
from oracle import sparse_ref as S


def oracle(masks, mgr, feats, gy, sd):
    hier = S.Hierarchy(b['vox_coords'].numpy(), n_levels=2)
    to_gpu = []
    for l in range(2):
        kg = S.pack_keys(mgr.coords[l].cpu().numpy()); ko = S.pack_keys(hier.coords[l])
        order = np.argsort(kg); pos = np.searchsorted(kg[order], ko)
        to_gpu.append(torch.from_numpy(order[pos]))
    it = iter(masks)

    def relu(x, level):
        m = next(it).cpu()[to_gpu[level]]
        return x * m.to(x.dtype)
    p = {k: v.double().clone().requires_grad_(True) if v.is_floating_point() and 'running' not in k else v for k, v in sd.items()}
    bn = lambda name, x: S.batch_norm(x, p[name + '.bn.weight'], p[name + '.bn.bias'], None, None, True)

    def block(name, x, nbr, level, short):
        out = S.conv_nbr(x, p[name + '.conv1.kernel'], nbr)
        out = relu(bn(name + '.norm1', out), level)
        out = bn(name + '.norm2', S.conv_nbr(out, p[name + '.conv2.kernel'], nbr))
        res = bn(name + '.downsample.1', S.conv_nbr(x, p[name + '.downsample.0.kernel'], None)) if short else x
        return relu(out + res, level)
    x0 = feats.double()[to_gpu[0].argsort()] if False else None
    # device rows are in Morton order; bring the input to oracle (input) order
    inv0 = torch.empty_like(to_gpu[0]); inv0[to_gpu[0]] = torch.arange(len(inv0))
    x = feats.double().cpu()[to_gpu[0]].clone().requires_grad_(True)
    e = block('b1', block('b0', x, hier.k3(0), 0, True), hier.k3(0), 0, False)
    d = relu(bn('bn', S.conv_nbr(e, p['down.kernel'], hier.down(0))), 1)
    d = block('b2', d, hier.k3(1), 1, False)
    u = relu(bn('bnu', S.conv_nbr(d, p['up.kernel'], hier.up(0))), 0)
    y = block('b3', torch.cat([u, e], 1), hier.k3(0), 0, True)
    (y * gy.double().cpu()[to_gpu[0]]).sum().backward()
    return y, x.grad, {k: v.grad for k, v in p.items() if torch.is_tensor(v) and v.requires_grad}, to_gpu[0]


def run_rec(env):
    for k in ('B2M_DETERMINISTIC', 'B2M_CONV_STATS', 'B2M_CONV_TARGET'):
        os.environ.pop(k, None)
    os.environ.update(env)
    _lib.reload_env()
    masks = []
    bn0 = F_.batch_norm

    def bnrec(x, g_, b_, rm, rv, tr, mom=0.1, eps=1e-5, residual=None, relu=False, sync=False, count_key=None):
        y = bn0(x, g_, b_, rm, rv, tr, mom, eps, residual, relu, sync, count_key)
        if relu:
            masks.append(y.detach() > 0)
        return y
    F_.batch_norm = bnrec
    os.environ['B2M_BN_PAIR'] = '0'         # (the recorder hooks F_.batch_norm only)
    try:
        torch.manual_seed(9)
        net = Chain().cuda().train()
        sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
        torch.manual_seed(10)
        feats = torch.randn(b['vox_coords'].shape[0], 32)
        sin = ME.SparseTensor(feats, b['vox_coords'])
        sin.F.requires_grad_(True)
        F_.packed_weights.begin_pass()
        y = net(sin).F
        gy = torch.randn(y.shape, device='cuda')
        (y * gy).sum().backward()
        torch.cuda.synchronize()
    finally:
        F_.batch_norm = bn0
    # gy / feats in device (Morton) order -> the oracle permutes by coordinates
    oy, odx, og, perm = oracle(masks, sin.manager, sin.F.detach().cpu(), gy, sd)
    g = {n: p_.grad.cpu() for n, p_ in net.named_parameters()}
    rows = sorted(((rel(g[n], og[n]), n) for n in g), reverse=True)
    print('%-22s vs fp64 oracle (shared ReLU decisions): y %.2e dx %.2e | worst %s' % (
        str(env), rel(y.detach().cpu()[perm], oy), rel(sin.F.grad.cpu()[perm], odx), ' '.join('%s %.1e' % (n, e) for e, n in rows[:5])), flush=True)


print()
for env in ({'B2M_DETERMINISTIC': '1'}, {}, {'B2M_CONV_STATS': '0'}, {'B2M_CONV_TARGET': '0'}):
    run_rec(env)
