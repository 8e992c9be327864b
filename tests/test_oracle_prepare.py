"""The CPU restatement of scene preparation (oracle/prepare_ref.py) against outputs of the real reference code
(tests/golden/prepare.npz, made by tools/gen_golden.py from /root/reference/models/dataloader.py)."""
import os

import numpy as np
import pytest

from oracle import prepare_ref as R

KEYS = ('vox_coords', 'vox2point', 'point2vox', 'vox_segments', 'vox_features', 'vox_world_coords', 'seg2vox',
        'seg2point', 'input_location')


@pytest.fixture(scope='module')
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, 'prepare.npz'))


def _scene(gold, i):
    return {k: gold['s%d_in_%s' % (i, k)] for k in ('positions', 'colors', 'normals', 'segments')}, \
        float(gold['s%d_in_voxel_size' % i])


def test_voxelize_bit_exact(gold):
    for i in range(int(gold['n_scenes'])):
        sc, vs = _scene(gold, i)
        r = R.voxelize_scene(sc['positions'], sc['colors'], sc['normals'], sc['segments'], vs)
        for k in KEYS:
            want = gold['s%d_%s' % (i, k)]
            assert r[k].shape == want.shape and r[k].dtype == want.dtype, (i, k, r[k].dtype, want.dtype)
            assert np.array_equal(r[k], want), (i, k)


def test_ball_tree_is_the_exact_nearest_point(gold):
    """The independent fp64 brute force picks the same points as the reference's ball tree (no exact ties in the
    fixtures), which is what the device kernel is specified against."""
    for i in range(int(gold['n_scenes'])):
        sc, vs = _scene(gold, i)
        shift = min(0, np.min(sc['positions']))
        ic = (sc['positions'] - shift) / vs
        nn = R.nearest_bruteforce(ic, gold['s%d_vox_coords' % i])
        assert np.array_equal(nn, gold['s%d_point2vox' % i]), i


def test_collate_bit_exact(gold):
    items = []
    for i in (0, 1):
        sc, vs = _scene(gold, i)
        items.append(R.voxelize_scene(sc['positions'], sc['colors'], sc['normals'], sc['segments'], vs))
    b = R.collate(items)
    for k in ('vox_features', 'batch_ids', 'input_location', 'pooling_ids'):
        want = gold['collate_%s' % k]
        assert b[k].dtype == want.dtype and np.array_equal(b[k], want), k
    n0 = len(items[0]['vox_coords'])
    assert b['vox_coords'].dtype == np.int32 and b['vox_coords'].shape == (n0 + len(items[1]['vox_coords']), 4)
    assert np.array_equal(b['vox_coords'][:n0, 1:], items[0]['vox_coords'].astype(np.int32))
    assert (b['vox_coords'][:n0, 0] == 0).all() and (b['vox_coords'][n0:, 0] == 1).all()
