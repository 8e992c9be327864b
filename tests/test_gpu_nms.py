"""GPU parity of the votes -> instance-mask path: bit-exact against golden vectors from the real
reference (tests/golden/*.npz) and, at sizes the reference takes seconds for, against the oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
CASES = ['votes1', 'votes64', 'votes512', 'votes2000', 'degenerate', 'edge_th']


@pytest.fixture(scope='module')
def g(golden_dir):
    return np.load(os.path.join(golden_dir, 'iou_nms.npz'))


@pytest.mark.parametrize('case', CASES)
@pytest.mark.parametrize('where', ['cpu', 'cuda'])
def test_nms_clustering_golden(g, case, where):
    from box2mask_amd import iou_nms
    boxes = torch.from_numpy(g[case + '_boxes']).to(where)
    reps, clusters, heat = iou_nms.NMS_clustering(boxes, float(g[case + '_th']))
    assert reps.dtype == torch.int64 and reps.device.type == where
    assert np.array_equal(reps.cpu().numpy(), g[case + '_reps'])
    assert np.array_equal(heat.cpu().numpy().view(np.uint32), g[case + '_heat'].view(np.uint32))
    assign = np.full(len(boxes), -1, np.int32)
    for c, idx in enumerate(clusters):
        assign[idx.cpu().numpy()] = c
    assert np.array_equal(assign, g[case + '_assign'])
    # members of a cluster are listed in visiting (descending score) order, like the reference
    sc = g[case + '_boxes'][:, 0]
    for idx in clusters:
        s = sc[idx.cpu().numpy()]
        assert (np.diff(s) <= 0).all()


@pytest.mark.parametrize('case', CASES)
def test_mask_nms_golden(g, case):
    from box2mask_amd import iou_nms
    masks = torch.from_numpy(g[case + '_masks'])
    kept, supp = iou_nms.mask_NMS(masks, 0.6)
    assert np.array_equal(kept.numpy(), g[case + '_mask_kept'])
    assert len(supp) == len(kept)


def test_small_functions_golden(g):
    from box2mask_amd import iou_nms
    from box2mask_amd.util import to_bbs_min_max, to_unique
    iou = iou_nms.set_IOUs(torch.from_numpy(g['set_a']), torch.from_numpy(g['set_b']))
    assert np.array_equal(iou.numpy().view(np.uint32), g['set_iou'].view(np.uint32))
    one = iou_nms.torch_IOUs(torch.from_numpy(g['set_a'][0]), torch.from_numpy(g['set_b']))
    from oracle import nms_ref
    assert np.array_equal(one.numpy().view(np.uint32), nms_ref.box_ious(g['set_a'][0], g['set_b']).view(np.uint32))
    s = iou_nms.semIOU(torch.from_numpy(g['sem_pred']).cuda(), torch.from_numpy(g['sem_gt']).cuda())
    assert np.array_equal(s, g['sem_iou'])
    bbs = to_bbs_min_max(*(torch.from_numpy(g[k]) for k in ('bbs_loc', 'bbs_off', 'bbs_bnd', 'bbs_sc')))
    assert np.array_equal(bbs.numpy().view(np.uint32), g['bbs_out'].view(np.uint32))
    assert np.array_equal(to_unique([g['uniq_in0'].copy(), g['uniq_in1'].copy(), g['uniq_in2'].copy()]).numpy(), g['uniq_out'])
    m = torch.from_numpy(g['votes512_masks'])
    mi = iou_nms.masks_iou(m[0], m[1:])
    assert np.array_equal(mi.numpy().view(np.uint32), nms_ref.masks_iou(g['votes512_masks'][0], g['votes512_masks'][1:]).view(np.uint32))


def test_empty_input_raises_like_reference():
    from box2mask_amd import iou_nms
    with pytest.raises(ValueError):
        iou_nms.NMS_clustering(torch.zeros(0, 7), 0.5)


@pytest.mark.parametrize('case', ['a', 'b'])
@pytest.mark.parametrize('mode', ['eval', 'train'])
def test_detection2mask_golden(golden_dir, case, mode):
    """Model.pred2mask against SelectionNet.detection2mask of the real reference."""
    from box2mask_amd import synth
    from box2mask_amd.config import scannet_config
    from box2mask_amd.detection_net import SelectionNet
    d = np.load(os.path.join(golden_dir, 'detection2mask.npz'))
    cfg = scannet_config()
    valid, _, _, is_fg = synth.scannet_tables()
    net = SelectionNet(cfg, 'cuda', valid, is_fg, out_channels=[96, 96, 6])
    names = [str(s) for s in d['d2m_%s_names' % case]]
    batch = {'input_location': torch.from_numpy(d['d2m_%s_input_location' % case]),
             'batch_ids': torch.from_numpy(d['d2m_%s_batch_ids' % case]),
             'scene': [{'name': n} for n in names],
             'seg2vox': [d['d2m_%s_seg2vox%d' % (case, i)] for i in range(len(names))],
             'vox2point': [d['d2m_%s_vox2point%d' % (case, i)] for i in range(len(names))]}
    pred = {h: torch.from_numpy(d['d2m_%s_pred_%s' % (case, h)]) for h in
            ('mlp_offsets', 'mlp_bounds', 'mlp_bb_scores', 'mlp_semantics')}
    res = net.detection2mask(batch, pred, cfg, mode, True, *d['d2m_%s_ths' % case].tolist())
    total = 0
    for si, n in enumerate(names):
        pre = 'd2m_%s_%s_s%d_' % (case, mode, si)
        r = res[n]
        assert np.array_equal(r['conf'].numpy().view(np.uint32), d[pre + 'conf'].view(np.uint32))
        assert r['label_id'].dtype == np.int32 and np.array_equal(r['label_id'], d[pre + 'label_id'])
        assert r['mask'].dtype == torch.bool and tuple(r['mask'].shape) == tuple(d[pre + 'mask_shape'])
        assert np.array_equal(np.packbits(r['mask'].numpy(), axis=1), d[pre + 'mask'])
        if mode != 'eval':
            assert np.array_equal(r['cluster_representatives'].numpy(), d[pre + 'reps'])
        total += r['mask'].shape[0]
    assert total > 0


@pytest.mark.parametrize('n,nobj', [(6000, 120), (1500, 30)])
def test_s3dis_sized_votes_against_oracle(n, nobj):
    from box2mask_amd import iou_nms, synth
    from oracle import nms_ref
    boxes = synth.make_votes(5, n_obj=nobj, n_seg=n)
    reps, clusters, heat = iou_nms.NMS_clustering(torch.from_numpy(boxes), 0.5)
    oreps, oclusters, oheat = nms_ref.nms_clustering(boxes, 0.5)
    assert np.array_equal(reps.numpy(), oreps)
    assert np.array_equal(heat.numpy().view(np.uint32), oheat.view(np.uint32))
    masks = oheat > np.float32(0.3)
    kept, _ = iou_nms.mask_NMS(torch.from_numpy(masks), 0.6)
    assert np.array_equal(kept.numpy(), nms_ref.mask_nms(masks, 0.6)[0])


def test_score_ties_are_visited_in_row_order():
    from box2mask_amd import iou_nms
    from oracle import nms_ref
    b = np.zeros((6, 7), np.float32)
    for i in range(6):
        b[i] = [0.5, 3 * i, 0, 0, 3 * i + 1, 1, 1]       # disjoint boxes, identical scores
    reps, _, _ = iou_nms.NMS_clustering(torch.from_numpy(b), 0.5)
    assert reps.tolist() == list(range(6)) == nms_ref.nms_clustering(b, 0.5)[0].tolist()


def test_large_n_global_sort_path():
    """n > 4096 takes the global-memory bitonic sort; idempotence: clustering the representatives
    again leaves every one of them its own cluster."""
    from box2mask_amd import iou_nms, synth
    from oracle import nms_ref
    boxes = synth.make_votes(9, n_obj=200, n_seg=9000)
    reps, clusters, heat = iou_nms.NMS_clustering(torch.from_numpy(boxes), 0.5)
    oreps, _, _ = nms_ref.nms_clustering(boxes, 0.5)
    assert np.array_equal(reps.numpy(), oreps)
    assert sum(len(c) for c in clusters) == len(boxes)
    reps2, _, _ = iou_nms.NMS_clustering(torch.from_numpy(boxes[reps.numpy()]), 0.5)
    assert len(reps2) == len(reps)


@pytest.mark.parametrize('case', ['a', 'b'])
@pytest.mark.parametrize('mode', ['eval', 'train'])
def test_detection2mask_without_segment_pooling_golden(golden_dir, case, mode):
    """cfg.do_segment_pooling = False (detection_net.py:436-445): per-voxel votes, no seg2vox projection; against the real
    reference on per-voxel predictions that are foreground everywhere (the inputs its branch can execute on), bit for bit."""
    from box2mask_amd import synth
    from box2mask_amd.config import scannet_config
    from box2mask_amd.detection_net import SelectionNet
    d = np.load(os.path.join(golden_dir, 'detection2mask_nopool.npz'))
    cfg = scannet_config()
    cfg.do_segment_pooling = False
    valid, _, _, is_fg = synth.scannet_tables()
    net = SelectionNet(cfg, 'cuda', valid, is_fg, out_channels=[96, 96, 6])
    names = [str(s) for s in d['np_%s_names' % case]]
    batch = {'input_location': torch.from_numpy(d['np_%s_input_location' % case]),
             'batch_ids': torch.from_numpy(d['np_%s_batch_ids' % case]),
             'scene': [{'name': n} for n in names],
             'vox2point': [d['np_%s_vox2point%d' % (case, i)] for i in range(len(names))]}      # (no 'seg2vox': it is not read)
    pred = {h: torch.from_numpy(d['np_%s_pred_%s' % (case, h)]) for h in
            ('mlp_offsets', 'mlp_bounds', 'mlp_bb_scores', 'mlp_semantics')}
    res = net.detection2mask(batch, pred, cfg, mode, True, *d['np_%s_ths' % case].tolist())
    total = 0
    for si, n in enumerate(names):
        pre = 'np_%s_%s_s%d_' % (case, mode, si)
        r = res[n]
        assert np.array_equal(r['conf'].numpy().view(np.uint32), d[pre + 'conf'].view(np.uint32))
        assert r['label_id'].dtype == np.int32 and np.array_equal(r['label_id'], d[pre + 'label_id'])
        assert r['mask'].dtype == torch.bool and tuple(r['mask'].shape) == tuple(d[pre + 'mask_shape'])
        assert np.array_equal(np.packbits(r['mask'].numpy(), axis=1), d[pre + 'mask'])
        if mode != 'eval':
            assert np.array_equal(r['cluster_representatives'].numpy(), d[pre + 'reps'])
        total += len(r['conf'])
    assert total > 50


def test_detection2mask_without_segment_pooling_pads_background_votes():
    """Background voxels among per-voxel votes (where the reference's branch stops with a shape mismatch): the result equals
    the pooled flow with one segment per voxel (seg2vox = identity), which zero-pads the background votes."""
    from box2mask_amd import synth
    from box2mask_amd.config import scannet_config
    from box2mask_amd.detection_net import SelectionNet
    valid, _, _, is_fg = synth.scannet_tables()
    rng = np.random.default_rng(3)
    n = 3000
    loc = rng.uniform(0, 3, (n, 3)).astype(np.float32)
    centres = rng.uniform(0.5, 2.5, (12, 3)).astype(np.float32)
    obj = rng.integers(0, 12, n)
    pred = {'mlp_offsets': torch.from_numpy(centres[obj] - loc + rng.normal(0, 0.02, (n, 3)).astype(np.float32)),
            'mlp_bounds': torch.from_numpy((0.3 + rng.normal(0, 0.02, (n, 3))).astype(np.float32)),
            'mlp_bb_scores': torch.from_numpy(rng.normal(0.5, 2, (n, 1)).astype(np.float32)),
            'mlp_semantics': torch.from_numpy(rng.normal(0, 3, (n, len(valid))).astype(np.float32))}     # mixed fore- / background
    base = {'input_location': torch.from_numpy(loc), 'batch_ids': torch.zeros(n, dtype=torch.long), 'scene': [{'name': 's'}],
            'vox2point': [rng.integers(0, n, 7000)]}
    out = []
    for pooling in (False, True):
        cfg = scannet_config()
        cfg.do_segment_pooling = pooling
        net = SelectionNet(cfg, 'cuda', valid, is_fg, out_channels=[96, 96, 6])
        batch = dict(base)
        if pooling:
            batch['seg2vox'] = [np.arange(n)]
        out.append(net.detection2mask(batch, pred, cfg, 'eval', True, 0.5, 0.05, 0.3, 0.6)['s'])
    a, b = out
    assert len(a['conf']) > 3
    assert torch.equal(a['conf'], b['conf']) and np.array_equal(a['label_id'], b['label_id']) and torch.equal(a['mask'], b['mask'])


@pytest.mark.parametrize('mode', ['eval', 'train'])
def test_detection2mask_s3dis_flow_golden(golden_dir, mode):
    """Per-voxel semantics head (S3DIS config): segment majority vote, no mask NMS (detection_net.py:398-415,449)."""
    from box2mask_amd.config import scannet_config
    from box2mask_amd.detection_net import SelectionNet
    d = np.load(os.path.join(golden_dir, 'detection2mask.npz'))
    cfg = scannet_config(network_heads=['mlp_offsets', 'mlp_bounds', 'mlp_bb_scores', 'mlp_per_vox_semantics'])
    net = SelectionNet(cfg, 'cuda', torch.Tensor(np.arange(13)), lambda s: s > 2, out_channels=[96, 96, 6])
    assert net.requires_voxel_outputs
    name = str(d['d2m_s3_names'][0])
    batch = {'input_location': torch.from_numpy(d['d2m_s3_input_location']),
             'batch_ids': torch.from_numpy(d['d2m_s3_batch_ids']), 'scene': [{'name': name}],
             'seg2vox': [d['d2m_s3_seg2vox0']], 'vox2point': [d['d2m_s3_vox2point0']],
             'vox_segments': [d['d2m_s3_vox_segments0']]}
    pred = {h: torch.from_numpy(d['d2m_s3_pred_' + h]) for h in
            ('mlp_offsets', 'mlp_bounds', 'mlp_bb_scores', 'mlp_per_vox_semantics')}
    r = net.detection2mask(batch, pred, cfg, mode, True, *d['d2m_s3_ths'].tolist())[name]
    pre = 'd2m_s3_%s_s0_' % mode
    assert np.array_equal(r['conf'].numpy().view(np.uint32), d[pre + 'conf'].view(np.uint32))
    assert np.array_equal(r['label_id'], d[pre + 'label_id'])
    assert tuple(r['mask'].shape) == tuple(d[pre + 'mask_shape']) and r['mask'].shape[0] > 0
    assert np.array_equal(np.packbits(r['mask'].numpy(), axis=1), d[pre + 'mask'])


def test_detection2mask_no_cluster_survives_the_score_filter():
    """All scores below the threshold: every scene must come back with zero instances (the reference returns empty
    tensors there), not fail on empty work."""
    from box2mask_amd import synth
    from box2mask_amd.config import scannet_config
    from box2mask_amd.model import Model
    cfg = scannet_config()
    batch = synth.make_batch(2, seed0=3, target_voxels=4000, pts_per_m2=6000.0)
    valid, id2idx, _, _ = synth.scannet_tables()
    S = batch['input_location'].shape[0]
    sem_idx = id2idx[batch['gt_semantics']].clamp_min(0)
    pred = {cfg.mlp_offsets: batch['gt_bb_offsets'].clone(), cfg.mlp_bounds: batch['gt_bb_bounds'].clamp_min(0.04),
            cfg.mlp_bb_scores: torch.full((S, 1), -20.0),                     # sigmoid ~ 2e-9 < score_th 0.05
            cfg.mlp_semantics: torch.nn.functional.one_hot(sem_idx, len(valid)).float()}
    model = Model(cfg, *synth.scannet_tables())
    for mode in ('eval', 'vox'):
        res = model.pred2mask(batch, pred, mode)
        for b, sc in enumerate(batch['scene']):
            r = res[sc['name']]
            assert len(r['conf']) == 0 and len(r['label_id']) == 0 and r['mask'].shape[0] == 0
            n = len(batch['vox2point'][b]) if mode == 'eval' else len(batch['seg2vox'][b])
            assert r['mask'].shape == (0, n)


def test_detection2mask_batch_stages_with_an_empty_scene_against_oracle():
    """The batched mask stages (one launch per stage over a descriptor table): 6 scenes of different sizes, one of them with
    no cluster above the score threshold (ksel = 0 in the middle of the table) -- every scene bit for bit against the CPU
    oracle's per-scene walk (oracle/nms_ref.py, pinned to the reference)."""
    from box2mask_amd import synth
    from box2mask_amd.config import scannet_config
    from box2mask_amd.model import Model
    from oracle import nms_ref
    cfg = scannet_config()
    items = [synth.make_scene(50 + s, target_voxels=tv, pts_per_m2=7000.0) for s, tv in enumerate((3000, 9000, 1500, 6000, 12000, 2500))]
    batch = synth.collate(items)
    valid, id2idx, _, is_fg = synth.scannet_tables()
    S = batch['input_location'].shape[0]
    g = torch.Generator().manual_seed(11)
    sem_idx = id2idx[batch['gt_semantics']].clamp_min(0)
    scores = 2.0 * torch.randn(S, 1, generator=g)
    scores[batch['batch_ids'] == 2] = -20.0                     # scene 2: nothing survives the score filter
    pred = {cfg.mlp_offsets: batch['gt_bb_offsets'] + 0.025 * torch.randn(S, 3, generator=g),
            cfg.mlp_bounds: (batch['gt_bb_bounds'] + 0.025 * torch.randn(S, 3, generator=g)).clamp_min(cfg.min_bb_size),
            cfg.mlp_bb_scores: scores,
            cfg.mlp_semantics: torch.nn.functional.one_hot(sem_idx, len(valid)).float()}
    model = Model(cfg, *synth.scannet_tables())
    res = model.pred2mask(batch, pred, 'eval')
    total = 0
    for b, sc in enumerate(batch['scene']):
        m = (batch['batch_ids'] == b).numpy()
        bbs = nms_ref.to_bbs_min_max(batch['input_location'][m].numpy(), pred[cfg.mlp_offsets][m].numpy(),
                                     pred[cfg.mlp_bounds][m].numpy(), torch.sigmoid(pred[cfg.mlp_bb_scores])[m].numpy())
        sem = valid[sem_idx[m]].long().numpy()
        ref = nms_ref.detection2mask_scene(bbs, sem, lambda x: (x > 2) & (x != 22), np.asarray(batch['seg2vox'][b]),
                                           np.asarray(batch['vox2point'][b]), list(cfg.eval_ths), 'eval')
        got = res[sc['name']]
        assert np.array_equal(ref['conf'], got['conf'].numpy()), b
        assert np.array_equal(ref['label_id'], got['label_id']), b
        assert np.array_equal(ref['mask'], got['mask'].numpy()), b
        if b == 2:
            assert got['mask'].shape[0] == 0
        total += got['mask'].shape[0]
    assert total > 10


def test_mask_gather_through_the_voxel_major_image_equals_the_row_lookups():
    """b2m_mask_gather_batch_t (the kept rows transposed into one word per voxel and 64 rows, then one look-up per point) against
    b2m_mask_gather_batch (one look-up per point and row) on a hand-made table: 150 kept rows of 200 (three words per voxel, a
    row subset in scrambled order) with a point count that is not a multiple of 16, a scene without kept rows in the middle, a
    scene with the identity index (training mode) and a 16-aligned one that takes the 16-byte stores."""
    from box2mask_amd import _lib
    g = torch.Generator().manual_seed(5)
    dev = 'cuda:0'
    scenes = [dict(n_vox=1000, ksel=200, kk=150, n_pts=4099, ident=False), dict(n_vox=300, ksel=4, kk=0, n_pts=640, ident=False),
              dict(n_vox=777, ksel=9, kk=5, n_pts=777, ident=True), dict(n_vox=2048, ksel=70, kk=65, n_pts=8192, ident=False)]
    keep = []
    desc = np.zeros((len(scenes), 20), np.int64)
    for s_, sc in enumerate(scenes):
        words = (sc['n_vox'] + 63) // 64
        bits = torch.randint(-2**62, 2**62, (sc['ksel'], words), generator=g, dtype=torch.int64).to(dev)
        rows = torch.randperm(sc['ksel'], generator=g)[:sc['kk']].int().to(dev)
        index = None if sc['ident'] else torch.randint(0, sc['n_vox'], (sc['n_pts'],), generator=g, dtype=torch.int64).to(dev)
        outs = [torch.full((max(sc['kk'], 1), sc['n_pts']), 7, dtype=torch.uint8, device=dev) for _ in range(2)]
        tb = torch.empty(max(sc['n_vox'] * ((sc['kk'] + 63) // 64), 1), dtype=torch.int64, device=dev)
        keep.append((bits, rows, index, outs, tb))
        desc[s_, 6:9] = (sc['n_vox'], bits.data_ptr(), words)
        desc[s_, 11:13] = (rows.data_ptr() if sc['kk'] else 0, sc['kk'])
        desc[s_, 15:17] = (0 if index is None else index.data_ptr(), sc['n_pts'])
    total = sum(sc['kk'] for sc in scenes)
    for which, name in ((0, 'b2m_mask_gather_batch'), (1, 'b2m_mask_gather_batch_t')):
        for s_ in range(len(scenes)):
            desc[s_, 17] = keep[s_][3][which].data_ptr()
        table = np.concatenate([desc.reshape(-1), np.array([k[4].data_ptr() for k in keep], np.int64)])
        d_dev = torch.from_numpy(table).to(dev)
        if which:
            _lib.call(name, d_dev.data_ptr(), len(scenes), total, max(sc['n_pts'] for sc in scenes),
                      max((sc['n_vox'] + 63) // 64 for sc in scenes), d_dev.data_ptr() + 8 * 20 * len(scenes))
        else:
            _lib.call(name, d_dev.data_ptr(), len(scenes), total, max(sc['n_pts'] for sc in scenes))
        torch.cuda.synchronize()
    for s_, sc in enumerate(scenes):
        a, b = keep[s_][3]
        if sc['kk'] == 0:
            assert int((a != 7).sum()) == 0 and int((b != 7).sum()) == 0          # nothing written
            continue
        assert torch.equal(a, b), s_
        # ... and both are the definition: bit (index[p]) of kept row r
        bits, rows, index = keep[s_][0].cpu().numpy().view(np.uint64), keep[s_][1].cpu().numpy(), keep[s_][2]
        v = np.arange(sc['n_pts']) if index is None else index.cpu().numpy()
        ref = ((bits[rows][:, v >> 6] >> (v & 63).astype(np.uint64)) & np.uint64(1)).astype(np.uint8)
        assert np.array_equal(ref, a.cpu().numpy()), s_


def test_unique_insert_few_distinct_keys_and_mixed_waves():
    """b2m_unique_insert through prepare._unique_inverse against np.unique: ground-truth-id-like input (1 M keys, 30 values,
    long runs and random order: the wave-level election path), all-distinct keys (the per-lane path) and waves that mix both."""
    from box2mask_amd import prepare
    rng = np.random.default_rng(0)
    vals = np.sort(rng.choice(10 ** 9, 30, replace=False)).astype(np.int64)
    runs = np.repeat(vals[rng.integers(0, 30, 4000)], 250)                       # long runs
    rnd = vals[rng.integers(0, 30, 300_000)]                                      # random order, few values
    distinct = rng.permutation(500_000).astype(np.int64) * 7 + 1                  # all different
    mixed = np.where(rng.random(400_000) < 0.5, vals[rng.integers(0, 30, 400_000)], rng.integers(0, 10 ** 12, 400_000))
    for name, keys in (('runs', runs), ('random few', rnd), ('distinct', distinct), ('mixed', mixed), ('one', vals[:1]),
                       ('same', np.full(1000, 42, np.int64))):
        u, nu, inv, *_ = prepare._unique_inverse(torch.from_numpy(keys).cuda())
        ru, rinv = np.unique(keys, return_inverse=True)
        assert nu == len(ru), name
        assert np.array_equal(u[:nu].cpu().numpy(), ru), name
        assert np.array_equal(inv.cpu().numpy(), rinv.reshape(-1)), name
