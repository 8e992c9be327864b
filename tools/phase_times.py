"""Device time (HIP events, no profiler) of the phases of a training step that consist of small launches: heads forward +
loss terms, and loss backward + heads backward -- against the sum of their kernels these show how long the device waits for
the host there."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from box2mask_amd import synth, functional as F_
from box2mask_amd.config import scannet_config
from box2mask_amd.model import Model

cfg = scannet_config()
torch.manual_seed(0)
model = Model(cfg, *synth.scannet_tables())
opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
batch = synth.make_batch(8, seed0=0)
for k in ('vox_coords', 'vox_features', 'pooling_ids', 'input_location', 'gt_bb_offsets', 'gt_bb_bounds', 'gt_semantics', 'fg_instances', 'batch_ids'):
    batch[k] = batch[k].cuda()
ev = {}
pool0 = F_.segment_pool


def mark(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); ev[name] = e


def pool(x, ids, n_seg, mode='avg'):
    y = pool0(x, ids, n_seg, mode)
    mark('pooled')
    if y.requires_grad:
        y.register_hook(lambda g: (mark('heads_bwd_done'), g)[1])
    return y


F_.segment_pool = pool
model.train()
res = []
for it in range(8):
    opt.zero_grad()
    mark('start')
    losses = model.compute_loss(batch, 150)
    mark('loss_done')
    model.prefetch(batch, ready=True)
    losses['optimization_loss'].backward()
    mark('bwd_done')
    opt.step()
    mark('end')
    torch.cuda.synchronize()
    t = lambda a, b: ev[a].elapsed_time(ev[b])
    res.append((t('start', 'pooled'), t('pooled', 'loss_done'), t('loss_done', 'heads_bwd_done'), t('heads_bwd_done', 'bwd_done'), t('bwd_done', 'end'), t('start', 'end')))
for r in res[3:]:
    print('forward to pooling %.2f | heads fwd + loss %.2f | loss bwd + heads bwd %.2f | rest of backward %.2f | optimizer %.2f | step %.2f ms' % r)
