"""Batch-size-1 inference forward passes for a kernel trace:  rocprofv3 --kernel-trace -- python3 tools/inference_trace.py
(then tools/inference_gaps.py on the CSV): what part of a forward pass is kernels, what part is the device waiting."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from box2mask_amd import synth
from box2mask_amd.config import scannet_config
from box2mask_amd.model import Model
torch.manual_seed(0)
cfg = scannet_config()
model = Model(cfg, *synth.scannet_tables(), device='cuda:0')
model.eval()
batch = synth.make_batch(1, seed0=100, target_voxels=150_000)
for k in ('vox_coords', 'vox_features', 'pooling_ids'):
    batch[k] = batch[k].cuda()
n = int(os.environ.get('PASSES', '12'))
for it in range(n + 3):
    if it == 3:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    model.get_prediction(batch, with_grad=False, to_cpu=False, min_size=False)
torch.cuda.synchronize()
print('forward wall %.3f ms per pass (%d passes, host clock)' % ((time.perf_counter() - t0) / n * 1e3, n))
