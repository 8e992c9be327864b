"""Where the HOST spends a training step of the ARKit-shaped workload (bench.py --workload arkit: four scenes, 430 k voxels at
4 cm -- the step the host paces): cProfile over STEPS free-running steps, fp32 and half (HALF=1).
    python tools/host_profile.py            (HALF=0|1, STEPS, WORKLOAD=arkit|scannet|s3dis)"""
import argparse, cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from box2mask_amd.model import Model
args = argparse.Namespace(workload=os.environ.get('WORKLOAD', 'arkit'), batch_size=8, target_voxels=150000)
w, cfg, tables, batch = bench.make_workload(args, 0, 1)
cfg.half_training = os.environ.get('HALF', '0') == '1'
model = Model(cfg, *tables)
opt = torch.optim.Adam(model.parameters(), lr=cfg.lr, fused=True)
model.train()
for k in list(batch):
    if torch.is_tensor(batch[k]):
        batch[k] = batch[k].cuda()


def step():
    opt.zero_grad(set_to_none=True)
    ld = model.compute_loss(batch, 150)
    ld['optimization_loss'].backward()
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
STEPS = int(os.environ.get('STEPS', '20'))
t0 = time.perf_counter()
for _ in range(STEPS):
    step()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print('%s half=%s: %.2f ms per step (host returned after %.2f ms per step)' % (args.workload, cfg.half_training, t_all / STEPS * 1e3, t_host / STEPS * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(STEPS):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)
st.sort_stats('cumulative').print_stats(34)
