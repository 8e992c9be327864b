"""How long does the host need to ENQUEUE one training step (no sync) vs the GPU to execute it?"""
import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from box2mask_amd import synth
from box2mask_amd.config import scannet_config
from box2mask_amd.model import Model
cfg = scannet_config()
torch.manual_seed(0)
model = Model(cfg, *synth.scannet_tables())
opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
bs = int(os.environ.get('BS', '8')); tv = int(os.environ.get('TV', '150000'))
batch = synth.make_batch(bs, seed0=0, target_voxels=tv)
for k in ('vox_coords', 'vox_features', 'pooling_ids', 'input_location', 'gt_bb_offsets', 'gt_bb_bounds', 'gt_semantics', 'fg_instances', 'batch_ids'):
    batch[k] = batch[k].cuda()
model.train()
def step():
    opt.zero_grad()
    l = model.compute_loss(batch, 150)
    l['optimization_loss'].backward()
    opt.step()
for _ in range(2): step()
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print('enqueue %.1f ms, total %.1f ms' % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
pr = cProfile.Profile(); pr.enable(); step(); pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(18); print(s.getvalue()[:3500])
