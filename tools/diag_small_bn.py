"""Upper bound of what fusing the small-map BatchNorm launches into their convolutions could save (review item 4 of round 5):
the training step with EVERY BatchNorm of a map with <= 4096 rows deleted (identity; wrong results) against the real step.
A fused conv + BatchNorm launch cannot beat no BatchNorm at all."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from box2mask_amd import synth, functional as F_
from box2mask_amd.config import scannet_config
from box2mask_amd.model import Model
torch.manual_seed(0)
cfg = scannet_config()
model = Model(cfg, *synth.scannet_tables(), device='cuda:0')
opt = torch.optim.Adam(model.parameters(), lr=cfg.lr, fused=True)
model.train()
batch = synth.make_batch(8, seed0=0)
batch = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}
bn0, pair0 = F_.batch_norm, F_.batch_norm_pair
skipped = [0]
def bn_skip(x, gamma, beta, rm, rv, training, momentum=0.1, eps=1e-5, residual=None, relu=False, sync=False, count_key=None):
    if x.shape[0] <= 4096:
        skipped[0] += 1
        return x if residual is None else x + 0 * residual
    return bn0(x, gamma, beta, rm, rv, training, momentum, eps, residual, relu, sync, count_key)
def run(n):
    for _ in range(3):
        opt.zero_grad(); model.compute_loss(batch, 150)['optimization_loss'].backward(); opt.step()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        opt.zero_grad(); model.compute_loss(batch, 150)['optimization_loss'].backward(); opt.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
for rnd in range(2):
    F_.batch_norm = bn0
    a = run(10)
    F_.batch_norm = bn_skip
    skipped[0] = 0
    b = run(10)
    print('step %.2f ms | without the BatchNorms of maps <= 4096 rows (%d per step deleted, identity) %.2f ms | difference %.2f ms' % (a, skipped[0] // 13, b, a - b))
F_.batch_norm = bn0
