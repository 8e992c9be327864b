"""A few training steps with the trunk in half (half_train.py) for a kernel trace: which kernels the half step consists of.
    rocprofv3 --kernel-trace --stats ... -- python3 tools/half_step_trace.py        (BS / TV / HALF=0 for the fp32 step)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from box2mask_amd import synth
from box2mask_amd.config import scannet_config
from box2mask_amd.model import Model
torch.manual_seed(0)
model = Model(scannet_config(half_training=os.environ.get('HALF', '1') == '1'), *synth.scannet_tables())
opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
batch = synth.make_batch(int(os.environ.get('BS', '8')), seed0=0, target_voxels=int(os.environ.get('TV', '150000')))
for k in ('vox_coords', 'vox_features', 'pooling_ids', 'input_location', 'gt_bb_offsets', 'gt_bb_bounds', 'gt_semantics', 'fg_instances', 'batch_ids'):
    batch[k] = batch[k].cuda()
model.train()
for _ in range(int(os.environ.get('STEPS', '8'))):
    opt.zero_grad()
    l = model.compute_loss(batch, 150)
    l['optimization_loss'].backward()
    opt.step()
torch.cuda.synchronize()
print('done', float(l['optimization_loss']))
