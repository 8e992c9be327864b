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


def test_box_supervision_bit_exact(gold):
    for i in (0, 1, 2):
        sc, vs = _scene(gold, i)
        labels = {k: gold['s%d_label_%s' % (i, k)] for k in ('unique_instances', 'per_instance_semantics',
                                                            'per_instance_bb_centers', 'per_instance_bb_bounds',
                                                            'seg2inst')}
        item = R.voxelize_scene(sc['positions'], sc['colors'], sc['normals'], sc['segments'], vs)
        ipp, ips = R.approx_association(sc['positions'], sc['segments'], labels, item['unique_vox_segments'], True)
        assert np.array_equal(ipp, gold['s%d_inst_per_point' % i]) and np.array_equal(ips, gold['s%d_inst_per_seg' % i])
        assert len(set(ips.tolist())) > (3 if i < 2 else 2)   # background and several instances
        if i == 2:                                         # overlapping boxes: the heuristic branch decided some
            sem = labels['per_instance_semantics']; fgm = (sem > 2) & (sem != 22)
            lo = (labels['per_instance_bb_centers'] - labels['per_instance_bb_bounds'] - 0.005)[fgm]
            hi = (labels['per_instance_bb_centers'] + labels['per_instance_bb_bounds'] + 0.005)[fgm]
            cnt = ((sc['positions'][None] >= lo[:, None]).all(-1) & (sc['positions'][None] <= hi[:, None]).all(-1)).sum(0)
            multi = [s_ for s_ in item['unique_vox_segments'] if cnt[sc['segments'] == s_].min() > 1]
            assert len(multi) > 3
        gt = R.bbs_supervision(item, labels, ips)
        for k, v in gt.items():
            want = gold['s%d_%s' % (i, k)]
            assert v.dtype == want.dtype and np.array_equal(v, want), (i, k)


# ---------------------------------------------------------------- the other association branches (prepare2.npz)
def _p2():
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from golden_scenes import prepare2_scenes
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'prepare2.npz'), allow_pickle=True), \
        prepare2_scenes()


def test_oracle_other_association_branches_match_reference():
    """oracle/prepare_ref.py's majority-vote / point / oriented-box / S3DIS associations == the real dataset classes
    (tools/gen_golden.py prepare2), bit for bit."""
    g, scenes = _p2()
    for i, sc in enumerate(scenes):
        pos, seg, lab = sc['positions'], sc['segments'], sc['labels']
        for vs, tag in ((sc['voxel_size'], 'scannet'), (0.04, 'arkit')):
            v = R.voxelize_scene(pos, sc['colors'], sc['normals'], seg, vs)
            useg = v['unique_vox_segments']
            if tag == 'scannet':
                for k in ('vox_coords', 'vox2point', 'point2vox', 'vox_segments', 'seg2vox', 'seg2point'):   # S3DIS.__getitem__
                    assert np.array_equal(v[k], g['s%d_s3dis_%s' % (i, k)]), k
                for h in (1, 0):
                    pp, ps = R.approx_association_points(pos, seg, lab, useg, bool(h), True)
                    assert np.array_equal(pp, g['s%d_scannet_majority_h%d_pseudo0' % (i, h)])
                    assert np.array_equal(ps, g['s%d_scannet_majority_h%d_pseudo1' % (i, h)])
                    pp, _ = R.approx_association_points(pos, seg, lab, useg, bool(h), False)
                    assert np.array_equal(pp, g['s%d_scannet_point_h%d_pseudo0' % (i, h)])
                for ign in (1, 0):
                    a = R.s3dis_association(pos, seg, lab, useg, False, bool(ign))
                    for j in range(4):
                        assert np.array_equal(a[j], g['s%d_s3dis_i%d_assoc%d' % (i, ign, j)]), (i, ign, j)
                    a = R.s3dis_association(pos, seg, lab, useg, True, bool(ign))
                    assert np.array_equal(a[0], g['s%d_s3dis_i%d_point0' % (i, ign)])
                    assert np.array_equal(a[1], g['s%d_s3dis_i%d_point1' % (i, ign)])
            else:
                for k in ('vox_coords', 'vox2point', 'point2vox', 'vox_segments', 'seg2vox', 'seg2point'):
                    assert np.array_equal(v[k], g['s%d_arkit_%s' % (i, k)]), k
                pp, ps = R.arkit_association(pos, seg, lab, useg, False)
                assert np.array_equal(pp, g['s%d_arkit_seg_pseudo0' % i]) and np.array_equal(ps, g['s%d_arkit_seg_pseudo1' % i])
                pp, _ = R.arkit_association(pos, seg, lab, useg, True)
                assert np.array_equal(pp, g['s%d_arkit_point_pseudo0' % i])
