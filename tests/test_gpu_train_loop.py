"""The whole chain on the device: raw points + weak boxes -> prepared batch -> optimisation steps -> checkpoint in
the reference's format -> reload -> masks -> ScanNet AP (tools/train_synthetic.py)."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_synthetic_scenes_are_learned():
    spec = importlib.util.spec_from_file_location('train_synthetic', os.path.join(ROOT, 'tools', 'train_synthetic.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    history, avg = mod.main(['--scenes', '3', '--voxels', '12000', '--steps', '200', '--eval-every', '0'])
    first, last = history[0][1], history[-1][1]
    assert last < 0.15 * first, history                    # 40 -> ~1 in the first hundred steps
    assert avg['all_ap_25%'] > 0.02, avg                   # instances start to come out (AP25 ~0.3 here, AP50 0.7 after 800 steps)
