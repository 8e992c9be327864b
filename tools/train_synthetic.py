"""End-to-end run of everything the repository builds, on synthetic scenes: raw points + weak box labels -> device
batch (prepare.voxelize_scene / box_supervision / collate) -> training steps as models/training.py:63-70 does them
-> a checkpoint in the reference's format -> predictions -> instance masks -> ScanNet AP against the scenes' own
instances (eval_metric.compute_eval).

    python tools/train_synthetic.py --scenes 4 --voxels 20000 --steps 200 [--half 1]
"""
import argparse
import os
import sys
import tempfile
import time
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from box2mask_amd import eval_metric, prepare, synth          # noqa: E402
from box2mask_amd.config import scannet_config                # noqa: E402
from box2mask_amd.model import Model                          # noqa: E402


def build_batch(seeds, voxels, mode, cfg_sup):
    items, raws = [], []
    for s in seeds:
        raw = synth.make_scene(s, target_voxels=voxels, points_only=True, pts_per_m2=8000.0)
        raw['name'] = 'synth%04d' % s
        it = prepare.voxelize_scene(raw, 0.02)
        it['scene'] = {'name': raw['name']}
        if mode == 'train':
            prepare.box_supervision(it, raw['labels'], cfg_sup)
        items.append(it); raws.append(raw)
    return prepare.collate(items, mode), raws


def ground_truth_ids(raw):
    """label*1000 + instance + 1 per point for furniture, label*1000 for floor / walls."""
    lab = raw['labels']
    inst = lab['seg2inst'][raw['segments']]
    sem = lab['per_instance_semantics'][inst].astype(np.int64)
    return np.where((sem > 2) & (sem != 22), sem * 1000 + inst + 1, sem * 1000)


def evaluate(model, batch, raws):
    model.eval()
    pred = model.get_prediction(batch, with_grad=False, to_cpu=True, min_size=True)
    res = model.pred2mask(batch, pred, 'eval')
    gts = {r['name']: ground_truth_ids(r) for r in raws}
    avg, _ = eval_metric.compute_eval(res, gts)
    model.train()
    return avg


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--scenes', type=int, default=4)
    ap.add_argument('--voxels', type=int, default=20000)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--lr', type=float, default=1e-3)
    ap.add_argument('--eval-every', type=int, default=50)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--half', type=int, default=0, help='1: cfg.half_training (half activations / gradients in the trunk, half_train.py)')
    args = ap.parse_args(argv)
    torch.manual_seed(args.seed)
    cfg = scannet_config(lr=args.lr, mlp_bb_scores_start_epoch=0, half_training=bool(args.half))       # score head trained from the first step
    cfg.checkpoint_path = tempfile.mkdtemp(prefix='b2m_ckpt_') + '/'
    sup = SimpleNamespace(smallest_bb_heuristic=True)
    batch, raws = build_batch(range(args.scenes), args.voxels, 'train', sup)
    model = Model(cfg, *synth.scannet_tables())
    opt = torch.optim.Adam(model.parameters(), lr=cfg.lr, fused=True)
    model.train()
    history = []
    t0 = time.time()
    for step in range(args.steps):
        opt.zero_grad()
        losses = model.compute_loss(batch, epoch=step)
        losses['optimization_loss'].backward()
        opt.step()
        if step % 10 == 0 or step == args.steps - 1:
            history.append((step, float(losses['optimization_loss'].detach())))
        if args.eval_every and (step % args.eval_every == 0 or step == args.steps - 1):
            avg = evaluate(model, batch, raws)
            print('step %4d  loss %.4f  AP50 %.3f  AP25 %.3f  AP %.3f  (%.1f s)'
                  % (step, float(losses['optimization_loss'].detach()), avg['all_ap_50%'], avg['all_ap_25%'], avg['all_ap'],
                     time.time() - t0), flush=True)
    # checkpoint in the reference's format (models/training.py:212-226) and reload through Model.load_checkpoint
    secs = float(int(time.time() - t0))          # the reference names checkpoints by the float training time
    name = 'checkpoint_{}h:{}m:{}s_{}'.format(int(secs // 3600), int((secs // 60) % 60), int(secs % 60), secs)
    torch.save({'epoch': args.steps, 'training_time': secs, 'iteration_num': args.steps,
                'model_state_dict': model.state_dict(), 'optimizer_state_dict': opt.state_dict()},
               cfg.checkpoint_path + name + '.tar')
    fresh = Model(cfg, *synth.scannet_tables())
    epoch, _, loaded, _ = fresh.load_checkpoint()
    a1, a2 = evaluate(model, batch, raws), evaluate(fresh, batch, raws)
    assert a1['all_ap_50%'] == a2['all_ap_50%'], 'a reloaded checkpoint must predict the same masks'
    print('checkpoint %s reloaded: AP50 %.3f' % (loaded, a2['all_ap_50%']))
    return history, a2


if __name__ == '__main__':
    main()
