"""The whole chain on the device: raw points + weak boxes -> prepared batch -> optimisation steps -> checkpoint in
the reference's format -> reload -> masks -> ScanNet AP (tools/train_synthetic.py)."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_synthetic_scenes_are_learned(monkeypatch):
    monkeypatch.setenv('B2M_DETERMINISTIC', '1')           # ordered reductions: the same trajectory on every run
    spec = importlib.util.spec_from_file_location('train_synthetic', os.path.join(ROOT, 'tools', 'train_synthetic.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    history, avg = mod.main(['--scenes', '3', '--voxels', '12000', '--steps', '200', '--eval-every', '0'])
    first, last = history[0][1], history[-1][1]
    print('loss %.3f -> %.3f, AP25 %.3f' % (first, last, avg['all_ap_25%']))
    assert last < 0.1 * first, history                     # 40 -> ~1 in the first hundred steps
    assert avg['all_ap_25%'] > 0.05, avg                   # instances start to come out (AP25 ~0.3 here, AP50 0.7 after 800 steps)


def test_forward_follows_a_fused_optimizer_step():
    """Fused optimizers update parameters without bumping their version counters; the packed weight images must be
    rebuilt anyway (regression: a version-validated cache kept serving the images of the initial weights)."""
    import torch
    from box2mask_amd import nn as ME, synth
    from box2mask_amd.config import scannet_config
    from box2mask_amd.detection_net import SelectionNet
    from oracle import unet_ref
    cfg = scannet_config()
    valid, _, _, is_fg = synth.scannet_tables()
    torch.manual_seed(1)
    net = SelectionNet(cfg, 'cuda', valid, is_fg, out_channels=[96, 96, 6]).cuda()
    net.train()
    batch = synth.make_batch(8, seed0=21, target_voxels=1500, pts_per_m2=6000.0)
    S_ = batch['input_location'].shape[0]
    opt = torch.optim.Adam(net.parameters(), lr=1e-2, fused=True)
    for _ in range(2):
        out = net(ME.SparseTensor(batch['vox_features'], batch['vox_coords']), batch['pooling_ids'].cuda(), S_)
        loss = sum((v.F ** 2).mean() for v in out.values())
        opt.zero_grad()
        loss.backward()
        opt.step()
    out = net(ME.SparseTensor(batch['vox_features'], batch['vox_coords']), batch['pooling_ids'].cuda(), S_)
    p_cpu = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    ref = unet_ref.forward(p_cpu, batch['vox_coords'].numpy(), batch['vox_features'], batch['pooling_ids'], cfg,
                           training=True, n_segments=S_)
    for h, v in out.items():
        err = float((v.F.detach().cpu() - ref[h]).abs().max()) / max(float(ref[h].abs().max()), 1e-9)
        assert err < 1e-3, (h, err)
