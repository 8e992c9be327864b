"""Model.load_checkpoint against checkpoints written exactly as the reference's trainer writes them
(/root/reference/models/training.py:216-224: `checkpoint_{h}h:{m}m:{s}s_{seconds}.tar` under cfg.checkpoint_path with the keys
training_time / epoch / iteration_num / model_state_dict / optimizer_state_dict), selected as /root/reference/models/model.py:264-288
selects them: newest, `closest_to` hours, by name; the 2-tuple (0, 0) for an empty directory.  SURVEY.md §8 a-19."""
import os
import types

import pytest
import torch

from box2mask_amd import synth
from box2mask_amd.config import scannet_config


def _convert_secs(sec):                 # utils/util.py:94-98
    return int(sec / 3600), int((sec / 60) % 60), int(sec % 60)


def _save_like_the_trainer(root, state_dict, training_time, epoch, iteration_num, optimizer_state=None):
    """training.py:216-224, same file name and the same dictionary."""
    path = root + 'checkpoint_{}h:{}m:{}s_{}.tar'.format(*[*_convert_secs(training_time), training_time])
    if not os.path.exists(path):
        torch.save({'training_time': training_time, 'epoch': epoch, 'iteration_num': iteration_num,
                    'model_state_dict': state_dict, 'optimizer_state_dict': optimizer_state or {}}, path)
    return os.path.basename(path)[:-4]


def _checkpoints(root, make_state):
    """Three checkpoints at 0.5 h, 2 h and 5.25 h of training time (integer and float seconds, as time.time() differences are)."""
    names = {}
    for secs, epoch, it in ((1800, 3, 120), (7200.5, 11, 480), (18900.25, 27, 1260)):
        names[epoch] = _save_like_the_trainer(root, make_state(epoch), secs, epoch, it)
    return names


def test_load_checkpoint_selection_rules_cpu(tmp_path):
    """The selection logic alone, through the unbound method on a stand-in (no GPU, no network): which file is read, what is
    handed to load_state_dict, the returned 4-tuple."""
    from box2mask_amd.model import Model
    root = str(tmp_path) + '/'
    cfg = scannet_config(checkpoint_path=root)
    loaded = []
    me = types.SimpleNamespace(cfg=cfg, device='cpu', load_state_dict=lambda sd, strict=True: loaded.append(sd))
    assert Model.load_checkpoint(me) == (0, 0)                                   # empty directory: model.py:267-269
    assert loaded == []
    names = _checkpoints(root, lambda epoch: {'w': torch.full((3,), float(epoch))})
    assert names[3] == 'checkpoint_0h:30m:0s_1800' and names[11] == 'checkpoint_2h:0m:0s_7200.5'
    # newest
    assert Model.load_checkpoint(me) == (27, 18900.25, names[27], 1260)
    assert float(loaded[-1]['w'][0]) == 27.0
    # closest to a training time in hours (model.py:275-276)
    assert Model.load_checkpoint(me, closest_to=2.2) == (11, 7200.5, names[11], 480)
    assert Model.load_checkpoint(me, closest_to=0.1) == (3, 1800, names[3], 120)
    assert Model.load_checkpoint(me, closest_to=100) == (27, 18900.25, names[27], 1260)
    # by name (model.py:281-282)
    assert Model.load_checkpoint(me, checkpoint=names[3]) == (3, 1800, names[3], 120)
    assert float(loaded[-1]['w'][0]) == 3.0
    with pytest.raises(FileNotFoundError):
        Model.load_checkpoint(me, checkpoint='checkpoint_9h:9m:9s_1')
    # a stray file: the reference fails converting its name to a float (np.array(..., dtype=float)); a ValueError here too
    open(root + 'notes.txt', 'w').write('x')
    with pytest.raises(ValueError):
        Model.load_checkpoint(me)


@pytest.mark.gpu
def test_load_checkpoint_restores_a_trained_model(tmp_path, monkeypatch):
    """The real Model: state dicts of two different weight sets saved as the trainer saves them (with an optimizer state),
    the newest / the closest one restored bit for bit into a fresh model, and the restored model predicts what the saved one
    predicted."""
    monkeypatch.setenv('B2M_DETERMINISTIC', '1')        # ordered reductions: equal weights then give equal bits
    from box2mask_amd.model import Model
    root = str(tmp_path) + '/'
    cfg = scannet_config(checkpoint_path=root)
    tables = synth.scannet_tables()
    torch.manual_seed(5)
    src = Model(cfg, *tables)
    assert src.load_checkpoint() == (0, 0)
    opt = torch.optim.Adam(src.parameters(), lr=1e-3)
    batch = synth.make_batch(2, seed0=3, target_voxels=3000, pts_per_m2=6000.0)
    src.train()
    states, preds = {}, {}
    for epoch, secs in ((1, 60.5), (2, 4000)):
        opt.zero_grad()
        src.compute_loss(batch, 150)['optimization_loss'].backward()
        opt.step()
        states[epoch] = {k: v.detach().cpu().clone() for k, v in src.state_dict().items()}
        name = _save_like_the_trainer(root, src.state_dict(), secs, epoch, 10 * epoch, opt.state_dict())
        src.eval()
        preds[epoch] = src.get_prediction(batch, with_grad=False, to_cpu=True, min_size=False)
        src.train()
        states[epoch, 'name'] = name
    torch.manual_seed(99)
    dst = Model(cfg, *tables)                                   # other initial weights
    assert dst.load_checkpoint() == (2, 4000, states[2, 'name'], 20)
    for k, v in dst.state_dict().items():
        assert torch.equal(v.cpu(), states[2][k]), k
    dst.eval()
    out = dst.get_prediction(batch, with_grad=False, to_cpu=True, min_size=False)
    for h in cfg.network_heads:
        assert torch.equal(out[h], preds[2][h]), h
    assert dst.load_checkpoint(closest_to=0.01) == (1, 60.5, states[1, 'name'], 10)
    for k, v in dst.state_dict().items():
        assert torch.equal(v.cpu(), states[1][k]), k
    out = dst.get_prediction(batch, with_grad=False, to_cpu=True, min_size=False)
    for h in cfg.network_heads:
        assert torch.equal(out[h], preds[1][h]), h
