"""Time of one `packed_weights.begin_pass()`: every layer's forward and data-gradient weight image in one launch
(weight_pack_batch_kernel), as at the start of every training step."""
import torch, sys, os
sys.path.insert(0, os.getcwd())
from box2mask_amd import synth, functional as F_
from box2mask_amd.config import scannet_config
from box2mask_amd.model import Model
m = Model(scannet_config(), *synth.scannet_tables()); m.train()
b = synth.make_batch(2, seed0=1, target_voxels=20000)
for _ in range(2):
    l = m.compute_loss(b, 150); l['optimization_loss'].backward()
torch.cuda.synchronize()
s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
for w in m.detection_model.parameters():
    w.data.add_(0.0)            # bump versions so that every image is repacked
ts = []
for _ in range(5):
    for w in m.detection_model.parameters(): w.data.mul_(1.0)
    torch.cuda.synchronize(); s.record(); F_.packed_weights.begin_pass(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
print('pack all layers: %.3f ms' % min(ts))
