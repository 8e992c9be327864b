"""Host-side cost of ENQUEUEING one batch-size-1 forward pass (maps prefetched: no host read inside): cProfile of
Model.get_prediction(..., to_cpu=False), and enqueue time against the device's time for it."""
import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from box2mask_amd import synth
from box2mask_amd.config import scannet_config
from box2mask_amd.model import Model
cfg = scannet_config()
torch.manual_seed(0)
model = Model(cfg, *synth.scannet_tables(), device='cuda:0')
batch = synth.make_batch(1, seed0=100, target_voxels=int(os.environ.get('TV', '150000')))
for k in ('vox_coords', 'vox_features', 'pooling_ids'):
    batch[k] = batch[k].cuda()
model.eval()
def once():
    model.prefetch(batch, ready=True, loss_rows=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pred = model.get_prediction(batch, with_grad=False, to_cpu=False, min_size=True)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) * 1e3, (t2 - t0) * 1e3
for _ in range(5): once()
for _ in range(4):
    print('enqueue %.2f ms, done %.2f ms' % once())
model.prefetch(batch, ready=True, loss_rows=False); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
model.get_prediction(batch, with_grad=False, to_cpu=False, min_size=True)
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28); print(s.getvalue()[:6000])
