"""GPU parity of scene preparation (include/b2m_prepare.h, box2mask_amd/prepare.py) against the golden vectors of
the real reference (tests/golden/prepare.npz) and against the CPU oracle on other inputs.  Integer outputs (voxel
rows, inverse maps, associated points, segment ranks) are bit-exact; float32 features are bit-exact; segment
centroids are compared in fp64 with rtol 1e-12 (integer sums vs numpy's running mean) and as float32 exactly."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, 'prepare.npz'))


def _scene(gold, i):
    return {k: gold['s%d_in_%s' % (i, k)] for k in ('positions', 'colors', 'normals', 'segments')}, \
        float(gold['s%d_in_voxel_size' % i])


def _check_item(item, want, tag):
    from box2mask_amd import prepare
    n = lambda t: t.cpu().numpy()
    assert np.array_equal(n(item['vox_coords'][:, 1:]), want['vox_coords'].astype(np.int32)), tag + ' vox_coords'
    assert (n(item['vox_coords'][:, 0]) == 0).all()
    for k in ('vox2point', 'point2vox', 'vox_segments', 'seg2vox', 'seg2point'):
        got = n(item[k])
        assert got.dtype == np.int64 and np.array_equal(got, want[k]), tag + ' ' + k
    assert np.array_equal(n(item['vox_features']), want['vox_features'].astype(np.float32)), tag + ' features'
    assert np.array_equal(n(prepare.vox_world_coords(item)), want['vox_world_coords']), tag + ' world coords'
    loc = n(item['input_location'])
    assert np.allclose(loc, want['input_location'], rtol=1e-12, atol=1e-12), tag + ' centroids'
    assert np.array_equal(loc.astype(np.float32), want['input_location'].astype(np.float32)), tag + ' centroids f32'


def test_voxelize_matches_reference_golden(gold):
    from box2mask_amd import prepare
    for i in range(int(gold['n_scenes'])):
        sc, vs = _scene(gold, i)
        item = prepare.voxelize_scene(sc, vs)
        want = {k: gold['s%d_%s' % (i, k)] for k in ('vox_coords', 'vox2point', 'point2vox', 'vox_segments',
                                                      'vox_features', 'vox_world_coords', 'seg2vox', 'seg2point',
                                                      'input_location')}
        _check_item(item, want, 'scene %d' % i)


def test_collate_matches_reference_golden(gold):
    from box2mask_amd import prepare
    items = [prepare.voxelize_scene(*_scene(gold, i)) for i in (0, 1)]
    b = prepare.collate(items, 'test')
    for k in ('vox_features', 'batch_ids', 'input_location', 'pooling_ids'):
        got = b[k].cpu().numpy()
        want = gold['collate_%s' % k]
        assert got.dtype == want.dtype and np.array_equal(got, want), k
    n0 = items[0]['vox_coords'].shape[0]
    c = b['vox_coords'].cpu().numpy()
    assert c.dtype == np.int32 and (c[:n0, 0] == 0).all() and (c[n0:, 0] == 1).all()
    assert np.array_equal(c[:n0, 1:], gold['s0_vox_coords'].astype(np.int32))
    assert np.array_equal(c[n0:, 1:], gold['s1_vox_coords'].astype(np.int32))


@pytest.mark.parametrize('seed,target,vs', [(3, 20000, 0.02), (4, 40000, 0.04)])
def test_voxelize_matches_oracle(seed, target, vs):
    from box2mask_amd import prepare, synth
    from oracle import prepare_ref as R
    sc = synth.make_scene(seed, target_voxels=target, points_only=True, pts_per_m2=8000.0)
    sc['positions'] = sc['positions'] - 0.83            # negative coordinates: exercises the shift
    want = R.voxelize_scene(sc['positions'], sc['colors'], sc['normals'], sc['segments'], vs)
    _check_item(prepare.voxelize_scene(sc, vs), want, 'seed %d' % seed)


def test_edge_cases():
    """One point; exact duplicates (lowest index wins); half-way coordinates (np.round is half-to-even); points of a
    neighbouring cell closer to a centre than the cell's own point."""
    from box2mask_amd import prepare
    from oracle import prepare_ref as R
    one = {'positions': np.array([[0.3, 0.1, 0.2]]), 'colors': np.ones((1, 3)), 'normals': np.ones((1, 3)),
           'segments': np.array([4])}
    it = prepare.voxelize_scene(one, 0.02)
    assert it['vox_coords'].shape == (1, 4) and it['point2vox'].tolist() == [0] and it['seg2vox'].tolist() == [0]
    vs = 0.5
    pos = np.array([[0.25, 0.25, 0.25],      # 0.5 -> rounds to 0 (half to even)
                    [0.75, 0.75, 0.75],      # 1.5 -> 2
                    [1.25, 1.25, 1.25],      # 2.5 -> 2
                    [1.25, 1.25, 1.25],      # duplicate of point 2
                    [0.49, 0.0, 0.0],        # cell (1,0,0), at 0.98: own point, far corner side
                    [0.26, 0.0, 0.0]])       # cell (1,0,0) as well (0.52 -> 1) but closer to centre (0,0,0)? no: 0.52
    sc = {'positions': pos, 'colors': np.arange(18.).reshape(6, 3), 'normals': np.zeros((6, 3)),
          'segments': np.array([1, 1, 2, 2, 3, 3])}
    it = prepare.voxelize_scene(sc, vs)
    want = R.voxelize_scene(pos, sc['colors'], sc['normals'], sc['segments'], vs)
    assert np.array_equal(it['vox_coords'][:, 1:].cpu().numpy(), want['vox_coords'].astype(np.int32))
    assert np.array_equal(it['vox2point'].cpu().numpy(), want['vox2point'])
    ic = pos / vs
    d = ((want['vox_coords'][:, None, :] - ic[None]) ** 2).sum(-1)
    got = it['point2vox'].cpu().numpy()
    assert np.array_equal(d[np.arange(len(got)), got], d.min(1))            # a nearest point
    assert np.array_equal(got, np.argmin(d, 1))                             # and the lowest index among ties
    with pytest.raises(ValueError):
        prepare.voxelize_scene({'positions': np.array([[0., 0., 0.], [1e9, 0., 0.]]), 'colors': np.zeros((2, 3)),
                                'normals': np.zeros((2, 3)), 'segments': np.array([0, 0])}, 0.02)


@pytest.mark.parametrize('n', [2, 64, 4096, 8192, 1 << 17])
def test_sort_u64(n):
    from box2mask_amd import _lib
    g = torch.Generator().manual_seed(n)
    k = torch.randint(0, 1 << 62, (n,), generator=g, dtype=torch.int64)
    k[: n // 3] = k[n // 3: 2 * (n // 3)][: n // 3]              # repeated keys
    d = k.cuda()
    _lib.call('b2m_sort_u64', d.data_ptr(), n)
    assert torch.equal(d.cpu(), torch.sort(k)[0])


def test_full_size_scene_properties():
    """~150 k voxels / ~1.2 M points (BASELINE configs[1] scene): properties that do not need the oracle."""
    from box2mask_amd import prepare, synth
    sc = synth.make_scene(11, points_only=True)
    vs = 0.02
    it = prepare.voxelize_scene(sc, vs)
    c = it['vox_coords'][:, 1:].cpu().numpy().astype(np.int64)
    key = (c[:, 0] << 42) | (c[:, 1] << 21) | c[:, 2]
    assert (np.diff(key) > 0).all()                                         # sorted, unique
    shift = min(0, sc['positions'].min())
    vox = np.round((sc['positions'] - shift) / vs).astype(np.int64)
    v2p = it['vox2point'].cpu().numpy()
    assert np.array_equal(c[v2p], vox)                                       # the inverse map of np.unique
    p2v = it['point2vox'].cpu().numpy()
    ic = (sc['positions'] - shift) / vs
    d_assoc = ((ic[p2v] - c) ** 2).sum(1)
    assert d_assoc.max() <= 0.75 + 1e-9
    first = np.full(len(c), -1, np.int64); first[v2p[::-1]] = np.arange(len(v2p))[::-1]
    assert (d_assoc <= ((ic[first] - c) ** 2).sum(1) + 1e-15).all()         # never worse than an own point
    s2v = it['seg2vox'].cpu().numpy()
    assert np.array_equal(np.unique(it['vox_segments'].cpu().numpy())[s2v], it['vox_segments'].cpu().numpy())


def test_prepared_batch_drives_the_model(gold):
    """raw points -> device batch -> SelectionNet forward -> instance masks, nothing through the host."""
    from box2mask_amd import prepare, synth
    from box2mask_amd.config import scannet_config
    from box2mask_amd.model import Model
    items = [prepare.voxelize_scene(*_scene(gold, i)) for i in (0, 1)]
    for it in items:
        it['scene'] = {'name': it['scene'].get('name', 'x') if isinstance(it['scene'], dict) else 'x'}
    items[0]['scene']['name'], items[1]['scene']['name'] = 'a', 'b'
    batch = prepare.collate(items, 'test')
    torch.manual_seed(0)
    model = Model(scannet_config(), *synth.scannet_tables())
    model.eval()
    pred = model.get_prediction(batch, with_grad=False, to_cpu=True, min_size=True)
    assert pred['mlp_offsets'].shape == (batch['input_location'].shape[0], 3)
    res = model.pred2mask(batch, pred, 'eval')
    assert set(res) == {'a', 'b'}
    for r, it in zip((res['a'], res['b']), items):
        assert r['mask'].shape[1] == it['vox2point'].shape[0]


def _labels(gold, i):
    return {k: gold['s%d_label_%s' % (i, k)] for k in ('unique_instances', 'per_instance_semantics',
                                                      'per_instance_bb_centers', 'per_instance_bb_bounds', 'seg2inst')}


def test_box_supervision_matches_reference_golden(gold):
    """approx_association + bbs_supervision (dataloader.py:165-314): instance per point / per segment bit-exact,
    targets exact except the offsets, which carry the centroid's few-ulp difference (rtol 1e-12; equal as float32)."""
    from types import SimpleNamespace
    from box2mask_amd import prepare
    cfg = SimpleNamespace(smallest_bb_heuristic=True, point_association=False, majority_vote=False,
                          dropout_boxes=None, noisy_boxes=None)
    items = []
    for i in (0, 1, 2):
        sc, vs = _scene(gold, i)
        sc['name'] = 's%d' % i
        it = prepare.box_supervision(prepare.voxelize_scene(sc, vs), _labels(gold, i), cfg)
        items.append(it)
        n = lambda t: t.cpu().numpy()
        assert np.array_equal(n(it['pseudo_inst'][0]), gold['s%d_inst_per_point' % i]), i
        assert np.array_equal(n(it['pseudo_inst'][1]), gold['s%d_inst_per_seg' % i]), i
        assert np.array_equal(n(it['fg_instances']), gold['s%d_fg_instances' % i])
        assert np.array_equal(n(it['gt_semantics']), gold['s%d_gt_semantics' % i])
        assert np.array_equal(n(it['gt_bb_bounds']), gold['s%d_gt_bb_bounds' % i])
        assert np.allclose(n(it['gt_bb_offsets']), gold['s%d_gt_bb_offsets' % i], rtol=1e-12, atol=1e-12)
    b = prepare.collate(items[:2], 'train')
    for k in ('gt_bb_bounds', 'gt_semantics', 'fg_instances'):
        got = b[k].cpu().numpy()
        assert got.dtype == gold['collate_%s' % k].dtype and np.array_equal(got, gold['collate_%s' % k]), k
    assert np.allclose(b['gt_bb_offsets'].cpu().numpy(), gold['collate_gt_bb_offsets'], rtol=0, atol=1e-6)
    assert (b['gt_bb_offsets'].cpu().numpy() != gold['collate_gt_bb_offsets']).mean() < 1e-3


def test_box_supervision_options_match_oracle():
    """no heuristic (-2 stays), no boxes at all, and a segment without voxels."""
    from types import SimpleNamespace
    from box2mask_amd import prepare, synth
    from oracle import prepare_ref as R
    sc = synth.make_scene(5, target_voxels=12000, points_only=True, pts_per_m2=7000.0)
    labels = sc['labels']
    # overlapping copies of the first boxes so that some segments sit in two boxes
    labels = dict(labels)
    labels['per_instance_bb_bounds'] = labels['per_instance_bb_bounds'] * np.float32(1.8)
    for heuristic in (False, True):
        cfg = SimpleNamespace(smallest_bb_heuristic=heuristic)
        it = prepare.box_supervision(prepare.voxelize_scene(sc, 0.02), labels, cfg)
        ref = R.voxelize_scene(sc['positions'], sc['colors'], sc['normals'], sc['segments'], 0.02)
        ipp, ips = R.approx_association(sc['positions'], sc['segments'], labels, ref['unique_vox_segments'], heuristic)
        assert np.array_equal(it['pseudo_inst'][1].cpu().numpy(), ips)
        assert np.array_equal(it['pseudo_inst'][0].cpu().numpy(), ipp)
        if not heuristic:
            assert (ips == -2).any()
        gt = R.bbs_supervision(ref, labels, ips)
        assert np.array_equal(it['gt_semantics'].cpu().numpy(), gt['gt_semantics'])
        assert np.array_equal(it['fg_instances'].cpu().numpy(), gt['fg_instances'])
    empty = dict(labels)
    empty['per_instance_semantics'] = np.zeros_like(labels['per_instance_semantics'])     # nothing is foreground
    it = prepare.box_supervision(prepare.voxelize_scene(sc, 0.02), empty, SimpleNamespace(smallest_bb_heuristic=True))
    assert (it['pseudo_inst'][1] == -1).all() and not it['fg_instances'].any()


def test_prepared_training_batch_trains(gold):
    """raw points + weak boxes -> device batch -> compute_loss -> backward."""
    from types import SimpleNamespace
    from box2mask_amd import prepare, synth
    from box2mask_amd.config import scannet_config
    from box2mask_amd.model import Model
    cfg_sup = SimpleNamespace(smallest_bb_heuristic=True)
    items = []
    for i in (0, 1):
        sc, vs = _scene(gold, i)
        sc['name'] = 's%d' % i
        items.append(prepare.box_supervision(prepare.voxelize_scene(sc, vs), _labels(gold, i), cfg_sup))
    batch = prepare.collate(items, 'train')
    torch.manual_seed(0)
    model = Model(scannet_config(), *synth.scannet_tables())
    model.train()
    losses = model.compute_loss(batch, 150)
    losses['optimization_loss'].backward()
    assert torch.isfinite(losses['optimization_loss'])
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())


def test_box_supervision_dropout_and_noise_match_reference_golden(gold):
    """cfg.dropout_boxes / cfg.noisy_boxes (dataloader.py:210-232): the random streams are seeded by the scene name,
    so the kept boxes, the noisy corners and the resulting targets must equal the reference's."""
    from types import SimpleNamespace
    from box2mask_amd import prepare
    sc, vs = _scene(gold, 1)
    sc['name'] = str(gold['s1_name'])
    cfg = SimpleNamespace(smallest_bb_heuristic=True, dropout_boxes=0.15, noisy_boxes=0.004)
    it = prepare.box_supervision(prepare.voxelize_scene(sc, vs), _labels(gold, 1), cfg)
    assert np.array_equal(it['noisy_bbs'][0], gold['s1_noisy_bbs_min'])
    assert np.array_equal(it['noisy_bbs'][1], gold['s1_noisy_bbs_max'])
    assert np.array_equal(it['pseudo_inst'][1].cpu().numpy(), gold['s1_noisy_inst_per_seg'])
    assert np.array_equal(it['pseudo_inst'][0].cpu().numpy(), gold['s1_noisy_inst_per_point'])
    for k in ('fg_instances', 'gt_bb_bounds', 'gt_semantics'):
        assert np.array_equal(it[k].cpu().numpy(), gold['s1_noisy_%s' % k]), k


# ---------------------------------------------------------------- the other dataset branches (tests/golden/prepare2.npz)
def _close(got, want, name):
    got = got.cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    if want.dtype.kind == 'f':
        # offsets / locations inherit the segment centroid's few-ulp difference (integer sums vs numpy's running mean)
        assert np.allclose(got, want, rtol=1e-12, atol=1e-12), name
    else:
        assert np.array_equal(got, want), name


def test_other_dataset_branches_match_reference_golden(golden_dir):
    """ScanNet majority_vote / point_association / mask_supervision, ARKitScenes 4 cm voxelisation + oriented-box
    association, S3DIS two-stage box association + mask supervision: device path == outputs of the REAL dataset classes
    (tools/gen_golden.py prepare2): instance ids, masks and integer targets bit-exact, fp64 targets to 1e-12."""
    import sys
    from types import SimpleNamespace
    from box2mask_amd import prepare
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from golden_scenes import prepare2_scenes
    g = np.load(os.path.join(golden_dir, 'prepare2.npz'))
    TGT = ('fg_instances', 'gt_bb_bounds', 'gt_bb_offsets', 'gt_semantics', 'gt_per_vox_semantics', 'instance_ids',
           'vox_instances')

    def check(tag, it):
        n = 0
        for k in TGT:
            if '%s_%s' % (tag, k) in g.files:
                _close(it[k], g['%s_%s' % (tag, k)], tag + ' ' + k)
                n += 1
        for j in (0, 1):
            if '%s_pseudo%d' % (tag, j) in g.files:
                _close(it['pseudo_inst'][j], g['%s_pseudo%d' % (tag, j)], tag + ' pseudo')
                n += 1
        assert n >= 3, tag

    for i, sc in enumerate(prepare2_scenes()):
        lab, vs = sc['labels'], sc['voxel_size']
        for h in (1, 0):
            cfg = SimpleNamespace(smallest_bb_heuristic=bool(h), point_association=False, majority_vote=True)
            check('s%d_scannet_majority_h%d' % (i, h), prepare.box_supervision(prepare.voxelize_scene(sc, vs), lab, cfg))
            cfg = SimpleNamespace(smallest_bb_heuristic=bool(h), point_association=True, majority_vote=False)
            it = prepare.box_supervision(prepare.voxelize_scene(sc, vs, do_segment_pooling=False), lab, cfg)
            check('s%d_scannet_point_h%d' % (i, h), it)
            _close(it['input_location'], g['s%d_scannet_point_h%d_input_location' % (i, h)], 'input_location')
            with pytest.raises(RuntimeError):                  # dataloader.py:173-174
                prepare.box_supervision(prepare.voxelize_scene(sc, vs), lab, cfg)
        for pool in (1, 0):
            it = prepare.mask_supervision(prepare.voxelize_scene(sc, vs, do_segment_pooling=bool(pool)), lab, None)
            check('s%d_scannet_mask_p%d' % (i, pool), it)
        # ---- ARKitScenes: 4 cm voxels, oriented boxes
        it = prepare.voxelize_scene(sc, 0.04)
        for k in ('vox2point', 'point2vox', 'vox_segments', 'seg2vox', 'seg2point'):
            _close(it[k], g['s%d_arkit_%s' % (i, k)], 'arkit ' + k)
        assert np.array_equal(it['vox_coords'][:, 1:].cpu().numpy(), g['s%d_arkit_vox_coords' % i].astype(np.int32))
        assert np.array_equal(it['vox_features'].cpu().numpy(), g['s%d_arkit_vox_features' % i].astype(np.float32))
        _close(it['input_location'], g['s%d_arkit_input_location' % i], 'arkit input_location')
        cfg = SimpleNamespace(point_association=False)
        check('s%d_arkit_seg' % i, prepare.box_supervision(it, lab, cfg, 'arkitscenes'))
        pp, none = prepare.approx_association(prepare.voxelize_scene(sc, 0.04), lab, SimpleNamespace(point_association=True),
                                              'arkitscenes')
        assert none is None
        _close(pp, g['s%d_arkit_point_pseudo0' % i], 'arkit point')
        check('s%d_arkit_mask' % i, prepare.mask_supervision(prepare.voxelize_scene(sc, 0.04), lab, None, 'arkitscenes'))
        # ---- S3DIS: foreground boxes, then background boxes; majority vote per segment
        for ign in (1, 0):
            cfg = SimpleNamespace(point_association=False, ignore_wall_ceiling_floor=bool(ign))
            it = prepare.voxelize_scene(sc, vs)
            a = prepare.approx_association(it, lab, cfg, 's3dis')
            for j in range(4):
                _close(a[j], g['s%d_s3dis_i%d_assoc%d' % (i, ign, j)], 's3dis assoc %d' % j)
            check('s%d_s3dis_i%d' % (i, ign), prepare.box_supervision(it, lab, cfg, 's3dis'))
            cfg = SimpleNamespace(point_association=True, ignore_wall_ceiling_floor=bool(ign))
            a = prepare.approx_association(prepare.voxelize_scene(sc, vs), lab, cfg, 's3dis')
            _close(a[0], g['s%d_s3dis_i%d_point0' % (i, ign)], 's3dis point inst')
            _close(a[1], g['s%d_s3dis_i%d_point1' % (i, ign)], 's3dis point sem')
        cfg = SimpleNamespace(ignore_wall_ceiling_floor=True)
        it = prepare.voxelize_scene(sc, vs)
        check('s%d_s3dis_mask' % i, prepare.mask_supervision(it, lab, cfg, 's3dis'))
        for k in ('vox2point', 'point2vox', 'vox_segments', 'seg2vox', 'seg2point'):        # S3DIS.__getitem__ (:671-730)
            _close(it[k], g['s%d_s3dis_%s' % (i, k)], 's3dis ' + k)
        assert np.array_equal(it['vox_coords'][:, 1:].cpu().numpy(), g['s%d_s3dis_vox_coords' % i].astype(np.int32))
        assert np.array_equal(it['vox_features'].cpu().numpy(), g['s%d_s3dis_vox_features' % i].astype(np.float32))
        _close(it['input_location'], g['s%d_s3dis_input_location' % i], 's3dis input_location')


def test_segment_mode_and_oriented_boxes_on_a_large_scene():
    """b2m_seg_mode / b2m_obb_membership against the oracle on a 300 k-point scene with many overlapping rotated boxes."""
    from types import SimpleNamespace
    from box2mask_amd import prepare, synth
    from oracle import prepare_ref as R
    sc = synth.make_scene(9, target_voxels=30000, points_only=True, pts_per_m2=9000.0)
    lab = dict(sc['labels'])
    n_inst = len(lab['unique_instances'])
    rng = np.random.default_rng(3)
    ang = rng.uniform(0, np.pi, n_inst)
    rot = np.zeros((n_inst, 3, 3))
    rot[:, 0, 0] = np.cos(ang); rot[:, 0, 1] = -np.sin(ang); rot[:, 1, 0] = np.sin(ang); rot[:, 1, 1] = np.cos(ang); rot[:, 2, 2] = 1
    lab['per_instance_bb_rotations'] = rot.reshape(n_inst, 9)
    lab['per_instance_bb_bounds'] = lab['per_instance_bb_bounds'] * np.float32(1.5)
    it = prepare.voxelize_scene(sc, 0.04)
    ref = R.voxelize_scene(sc['positions'], sc['colors'], sc['normals'], sc['segments'], 0.04)
    useg = ref['unique_vox_segments']
    for pa in (False, True):
        got = prepare.approx_association(it, lab, SimpleNamespace(point_association=pa), 'arkitscenes')
        want = R.arkit_association(sc['positions'], sc['segments'], lab, useg, pa)
        assert np.array_equal(got[0].cpu().numpy(), want[0])
        if not pa:
            assert np.array_equal(got[1].cpu().numpy(), want[1])
    it2 = prepare.voxelize_scene(sc, 0.02)
    ref2 = R.voxelize_scene(sc['positions'], sc['colors'], sc['normals'], sc['segments'], 0.02)
    got = prepare.approx_association(it2, lab, SimpleNamespace(point_association=False, majority_vote=True,
                                                               smallest_bb_heuristic=True), 'scannet')
    want = R.approx_association_points(sc['positions'], sc['segments'], lab, ref2['unique_vox_segments'], True, True)
    assert np.array_equal(got[0].cpu().numpy(), want[0]) and np.array_equal(got[1].cpu().numpy(), want[1])
    got = prepare.approx_association(it2, lab, SimpleNamespace(point_association=False, ignore_wall_ceiling_floor=True), 's3dis')
    want = R.s3dis_association(sc['positions'], sc['segments'], lab, ref2['unique_vox_segments'], False, True)
    for a, b in zip(got, want):
        assert np.array_equal(a.cpu().numpy(), b)


def test_scenes_voxelised_together_equal_one_by_one():
    """prepare.voxelize_scenes queues every scene's kernels of a stage before the stage's counts come back in ONE copy (two host
    reads per batch instead of four per scene): every tensor of every item equals voxelize_scene's, bit for bit; a scene with a
    point out of range or a negative segment id is still refused (dataloader.py:61-123)."""
    from box2mask_amd import prepare, synth
    scenes = [synth.make_scene(300 + i, target_voxels=(3000, 9000, 1500)[i], points_only=True) for i in range(3)]
    together = prepare.voxelize_scenes(scenes, 0.02)
    for sc, it in zip(scenes, together):
        one = prepare.voxelize_scene(sc, 0.02)
        assert set(one) == set(it)
        for k, v in one.items():
            if torch.is_tensor(v):
                assert v.dtype == it[k].dtype and torch.equal(v, it[k]), k
    bad = dict(scenes[1], segments=np.where(np.arange(len(scenes[1]['segments'])) == 5, -3, scenes[1]['segments']))
    with pytest.raises(ValueError):
        prepare.voxelize_scenes([scenes[0], bad], 0.02)
    far = dict(scenes[0], positions=np.concatenate([scenes[0]['positions'], [[1e9, 0.0, 0.0]]]),
               colors=np.concatenate([scenes[0]['colors'], [[0.0, 0.0, 0.0]]]), normals=np.concatenate([scenes[0]['normals'], [[0.0, 0.0, 1.0]]]),
               segments=np.concatenate([scenes[0]['segments'], [0]]))
    with pytest.raises(ValueError):
        prepare.voxelize_scenes([far, scenes[2]], 0.02)
