"""A/B of a library switch on the half training step inside ONE process (alternating, so the box's drift cancels):
    AB="B2M_CONV_TW4_H=1;B2M_CONV_TW4_H=0" python tools/half_step_ab.py        (ROUNDS=3, STEPS=15, HALF=1, BS=8, TV=150000)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from box2mask_amd import synth, _lib, functional as F_, half_train as HT
from box2mask_amd.config import scannet_config
from box2mask_amd.model import Model
torch.manual_seed(0)
half = os.environ.get('HALF', '1') == '1'
model = Model(scannet_config(half_training=half), *synth.scannet_tables())
opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
batch = synth.make_batch(int(os.environ.get('BS', '8')), seed0=0, target_voxels=int(os.environ.get('TV', '150000')))
for k in list(batch):
    if torch.is_tensor(batch[k]):
        batch[k] = batch[k].cuda()
model.train()


def step():
    opt.zero_grad()
    model.compute_loss(batch, 150)['optimization_loss'].backward()
    opt.step()


variants = [dict(kv.split('=') for kv in v.split(',') if kv) for v in os.environ.get('AB', 'X=1;X=0').split(';')]
keys = sorted({k for v in variants for k in v})
res = {i: [] for i in range(len(variants))}
for rnd in range(int(os.environ.get('ROUNDS', '3'))):
    for i, v in enumerate(variants):
        for k in keys:
            os.environ.pop(k, None)
        os.environ.update(v); _lib.reload_env()
        HT.images.__init__(); F_.invalidate_half_images(); F_.packed_weights.__init__()
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = int(os.environ.get('STEPS', '15'))
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        res[i].append((time.perf_counter() - t0) / n * 1e3)
for i, v in enumerate(variants):
    print('%-40s %s  min %.2f ms' % (v, ' '.join('%.2f' % t for t in res[i]), min(res[i])), flush=True)
