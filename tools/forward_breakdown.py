"""Device time of the FORWARD pass by C-ABI entry (HIP events around every b2m_* launch, train-mode BatchNorm, no autograd
graph) -- the part of a training step nothing overlaps."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from box2mask_amd import synth, _lib, nn as ME
from box2mask_amd.config import scannet_config
from box2mask_amd.model import Model

cfg = scannet_config()
torch.manual_seed(0)
model = Model(cfg, *synth.scannet_tables())
net = model.detection_model.train()
batch = synth.make_batch(8, seed0=0)
rec = []


def hook(name, a, meta=None):
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    shape = ''
    if name.startswith('b2m_conv_fwd'):
        shape = ' K=%d %d->%d n=%d' % (a[8], a[2] + a[5], a[16], a[13])
    elif name in ('b2m_bn_apply', 'b2m_bn_small_fwd', 'b2m_bn_apply2'):
        shape = ' n=%d c=%d' % ((a[2], a[3]) if name != 'b2m_bn_apply2' else (a[4], a[5]))

    def done():
        e.record(); rec.append((name, shape, s, e))
    return done


S_ = batch['input_location'].shape[0]
with torch.no_grad():
    for it in range(3):
        sin = ME.SparseTensor(batch['vox_features'], batch['vox_coords'])
        sin.manager.prefetch(8, same=[(0, 5)] + [(l, 3) for l in range(8)], strided=True)
        torch.cuda.synchronize()
        rec.clear()
        _lib.set_hook(hook if it == 2 else None)
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        net(sin, batch['pooling_ids'].cuda(), S_)
        t1.record(); torch.cuda.synchronize()
        _lib.set_hook(None)
        print('forward wall %.2f ms%s' % (t0.elapsed_time(t1), ' (every launch bracketed)' if it == 2 else ''))
by = collections.defaultdict(lambda: [0.0, 0])
for name, shape, s, e in rec:
    by[name][0] += s.elapsed_time(e); by[name][1] += 1
print('by entry:')
for k, (ms, n) in sorted(by.items(), key=lambda kv: -kv[1][0]):
    print('  %-28s %7.3f ms %4d launches' % (k, ms, n))
bs = collections.defaultdict(lambda: [0.0, 0])
for name, shape, s, e in rec:
    if shape:
        bs[name[4:] + shape][0] += s.elapsed_time(e); bs[name[4:] + shape][1] += 1
print('largest shapes:')
for k, (ms, n) in sorted(bs.items(), key=lambda kv: -kv[1][0])[:28]:
    print('  %-50s %7.3f ms x%d' % (k, ms, n))
