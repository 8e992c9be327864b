"""Generate golden vectors from the REAL reference code (run in the build container only).

    python tools/gen_golden.py            # writes tests/golden/*.npz

It imports /root/reference/models/iou_nms.py and utils/util.py unmodified, and drives
SelectionNet.detection2mask / Model.compute_loss_detection through import stand-ins for the
absent MinkowskiEngine / open3d modules (only needed so `import models.detection_net` succeeds;
the two functions themselves use torch / numpy / scipy only — SURVEY.md Appendix B).
Nothing from /root/reference is copied: the fixtures hold inputs and outputs only.
This script is never needed on the GPU box.
"""
from __future__ import annotations

import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, 'tests', 'golden')
REF = '/root/reference'
sys.path.insert(0, ROOT)

from box2mask_amd import synth  # noqa: E402


def _install_stubs():
    class Holder:
        def __init__(self, F=None, C=None, **kw):
            self.F, self.C = F, C

    me = types.ModuleType('MinkowskiEngine')
    me.SparseTensor = Holder
    me.TensorField = Holder
    me.MinkowskiConvolution = me.MinkowskiBatchNorm = me.MinkowskiReLU = object
    mods = types.ModuleType('MinkowskiEngine.modules')
    rb = types.ModuleType('MinkowskiEngine.modules.resnet_block')
    rb.Bottleneck = type('Bottleneck', (), {'expansion': 4})
    rb.BasicBlock = type('BasicBlock', (), {'expansion': 1})
    sys.modules['MinkowskiEngine'] = me
    sys.modules['MinkowskiEngine.modules'] = mods
    sys.modules['MinkowskiEngine.modules.resnet_block'] = rb
    sys.modules['open3d'] = types.ModuleType('open3d')
    if 'tqdm' not in sys.modules:
        try:
            import tqdm  # noqa: F401
        except ImportError:
            t = types.ModuleType('tqdm'); t.tqdm = lambda x, **k: x; sys.modules['tqdm'] = t
    return Holder


def gen_iou_nms():
    sys.path.insert(0, REF)
    import models.iou_nms as R
    import utils.util as U
    rng = np.random.default_rng(7)
    out = {}
    cases = {}
    for n, nobj in ((1, 1), (64, 6), (512, 30), (2000, 60)):
        cases['votes%d' % n] = synth.make_votes(n, n_obj=nobj, n_seg=n)
    # degenerate boxes: zero side length (the "Invalid boxes" warning path), identical boxes
    b = synth.make_votes(91, n_obj=8, n_seg=200)
    b[5, 4:] = b[5, 1:4]                     # zero volume
    b[17, 4] = b[17, 1]                      # one zero side
    b[30, 1:] = b[31, 1:]                    # identical geometry, different scores
    b[40, 1:] = b[31, 1:]
    cases['degenerate'] = b
    # a box whose IoU with the top box equals the threshold exactly (kept in `remaining` by `<=`)
    e = np.zeros((3, 7), np.float32)
    e[0] = [0.9, 0, 0, 0, 1, 1, 1]
    e[1] = [0.8, 0, 0, 0, 1, 1, 2]            # IoU = 1/(2+1e-6) < 0.5 ; use th computed from actual value
    e[2] = [0.7, 5, 5, 5, 6, 6, 6]
    cases['edge_th'] = e
    for name, boxes in cases.items():
        bt = torch.from_numpy(boxes)
        th = 0.5
        if name == 'edge_th':
            th = float(R.torch_IOUs(bt[0, 1:], bt[:, 1:])[1])     # threshold == IoU exactly
        reps, clusters, heat = R.NMS_clustering(bt, th)
        out[name + '_boxes'] = boxes
        out[name + '_th'] = np.float64(th)
        out[name + '_reps'] = reps.numpy()
        out[name + '_heat'] = heat.numpy()
        out[name + '_assign'] = _assign(clusters, len(boxes))
        out[name + '_order'] = np.concatenate([c.numpy() for c in clusters]) if False else np.zeros(0)
        # mask NMS on the thresholded heat-maps in descending score order of the representatives
        masks = heat > 0.3
        kept, supp = R.mask_NMS(masks, 0.6)
        out[name + '_masks'] = masks.numpy()
        out[name + '_mask_kept'] = kept.numpy()
    # set_IOUs / semIOU / to_bbs_min_max
    a = np.sort(rng.uniform(0, 3, (300, 2, 3)).astype(np.float32), 1).reshape(300, 6)
    c = np.sort(rng.uniform(0, 3, (300, 2, 3)).astype(np.float32), 1).reshape(300, 6)
    out['set_a'], out['set_b'] = a, c
    out['set_iou'] = R.set_IOUs(torch.from_numpy(a), torch.from_numpy(c)).numpy()
    pl = rng.integers(0, 20, 500); gl = rng.integers(0, 20, 500); gl[rng.random(500) < 0.2] = -100
    out['sem_pred'], out['sem_gt'] = pl, gl
    out['sem_iou'] = R.semIOU(torch.from_numpy(pl), torch.from_numpy(gl))
    loc = rng.normal(0, 1, (50, 3)).astype(np.float32); off = rng.normal(0, .3, (50, 3)).astype(np.float32)
    bnd = rng.uniform(.05, .5, (50, 3)).astype(np.float32); sc = rng.random((50, 1)).astype(np.float32)
    out['bbs_loc'], out['bbs_off'], out['bbs_bnd'], out['bbs_sc'] = loc, off, bnd, sc
    out['bbs_out'] = U.to_bbs_min_max(*(torch.from_numpy(v) for v in (loc, off, bnd, sc))).numpy()
    segs = [rng.integers(0, 9, 40), rng.integers(3, 14, 55), rng.integers(0, 5, 30)]
    out['uniq_in0'], out['uniq_in1'], out['uniq_in2'] = segs
    out['uniq_out'] = U.to_unique([s.copy() for s in segs]).numpy()
    np.savez_compressed(os.path.join(OUT, 'iou_nms.npz'), **out)
    print('iou_nms.npz: %d arrays' % len(out))


def _assign(clusters, n):
    a = np.full(n, -1, np.int32)
    for c, idx in enumerate(clusters):
        a[idx.numpy()] = c
    return a


def _scene_inputs(seed, n_scenes=2, target_voxels=12000):
    """Synthetic batch + head outputs that vote for the scene's furniture boxes (so clusters are meaningful)."""
    batch = synth.make_batch(n_scenes, seed0=seed, target_voxels=target_voxels, pts_per_m2=6000.0)
    rng = np.random.default_rng(seed)
    S = batch['input_location'].shape[0]
    fg = batch['fg_instances'].numpy()
    off = batch['gt_bb_offsets'].numpy() + rng.normal(0, 0.03, (S, 3)).astype(np.float32)
    bnd = np.maximum(batch['gt_bb_bounds'].numpy() + rng.normal(0, 0.03, (S, 3)).astype(np.float32), 0.04)
    bnd[~fg] = rng.uniform(0.05, 0.3, ((~fg).sum(), 3))
    logits = rng.normal(0.5, 2.0, (S, 1)).astype(np.float32)
    valid = synth.SCANNET_SEMANTIC_VALID_CLASS_IDS
    sem_logits = rng.normal(0, 1, (S, len(valid))).astype(np.float32)
    gt = batch['gt_semantics'].numpy()
    for s in range(S):           # mostly-correct semantics so that foreground selection is realistic
        if rng.random() < 0.9 and gt[s] in valid:
            sem_logits[s, int(np.nonzero(valid == gt[s])[0][0])] += 6.0
    pred = {'mlp_offsets': torch.from_numpy(off.astype(np.float32)), 'mlp_bounds': torch.from_numpy(bnd.astype(np.float32)),
            'mlp_bb_scores': torch.from_numpy(logits), 'mlp_semantics': torch.from_numpy(sem_logits)}
    return batch, pred


def gen_detection2mask():
    _install_stubs()
    sys.path.insert(0, REF)
    import models.detection_net as dn
    valid, id2idx, _, is_fg = synth.scannet_tables()
    cfg = SimpleNamespace(mlp_per_vox_semantics='mlp_per_vox_semantics', mlp_semantics='mlp_semantics',
                          network_heads=['mlp_offsets', 'mlp_bounds', 'mlp_bb_scores', 'mlp_semantics'],
                          do_segment_pooling=True)
    ns = SimpleNamespace(requires_voxel_outputs=False, semantic_valid_class_ids=valid, is_foreground=is_fg)
    out = {}
    for case, seed in (('a', 11), ('b', 23)):
        batch, pred = _scene_inputs(seed)
        ths = [0.5, 0.05, 0.3, 0.6]
        for mode in ('eval', 'train'):
            res = dn.SelectionNet.detection2mask(ns, batch, {k: v.clone() for k, v in pred.items()}, cfg, mode, True, *ths)
            for si, sc in enumerate(batch['scene']):
                r = res[sc['name']]
                pre = 'd2m_%s_%s_s%d_' % (case, mode, si)
                out[pre + 'conf'] = r['conf'].numpy()
                out[pre + 'label_id'] = np.asarray(r['label_id'])
                out[pre + 'mask'] = np.packbits(r['mask'].numpy(), axis=1)
                out[pre + 'mask_shape'] = np.asarray(r['mask'].shape)
                if mode != 'eval':
                    out[pre + 'reps'] = r['cluster_representatives'].numpy()
        for k, v in pred.items():
            out['d2m_%s_pred_%s' % (case, k)] = v.numpy()
        out['d2m_%s_input_location' % case] = batch['input_location'].numpy()
        out['d2m_%s_batch_ids' % case] = batch['batch_ids'].numpy()
        for si in range(len(batch['scene'])):
            out['d2m_%s_seg2vox%d' % (case, si)] = np.asarray(batch['seg2vox'][si])
            out['d2m_%s_vox2point%d' % (case, si)] = np.asarray(batch['vox2point'][si])
        out['d2m_%s_names' % case] = np.asarray([s['name'] for s in batch['scene']])
        out['d2m_%s_ths' % case] = np.asarray(ths)
    # ---- S3DIS flow (detection_net.py:398-415,449-451): per-voxel semantics head, majority vote per segment,
    # no mask NMS; batch of ONE scene (the reference indexes the whole batch's voxels there)
    s3_valid = torch.Tensor(np.arange(13))
    cfg3 = SimpleNamespace(mlp_per_vox_semantics='mlp_per_vox_semantics', mlp_semantics='mlp_semantics',
                           network_heads=['mlp_offsets', 'mlp_bounds', 'mlp_bb_scores', 'mlp_per_vox_semantics'],
                           do_segment_pooling=True)
    ns3 = SimpleNamespace(requires_voxel_outputs=True, semantic_valid_class_ids=s3_valid, is_foreground=lambda s: s > 2)
    batch, pred = _scene_inputs(31, n_scenes=1, target_voxels=15000)
    rng = np.random.default_rng(31)
    n_vox = batch['vox_coords'].shape[0]
    seg2vox = np.asarray(batch['seg2vox'][0])
    # per-voxel logits: the segment's class (scannet id mapped into 0..12) + noise, so that votes are not unanimous
    seg_cls = (batch['gt_semantics'].numpy() % 13)
    vox_logits = rng.normal(0, 1, (n_vox, 13)).astype(np.float32)
    vox_logits[np.arange(n_vox), seg_cls[seg2vox]] += 2.0
    pred3 = {k: v for k, v in pred.items() if k != 'mlp_semantics'}
    pred3['mlp_per_vox_semantics'] = torch.from_numpy(vox_logits)
    ths3 = [0.5, 0.03, 0.3, 0.6]
    for mode in ('eval', 'train'):
        res = dn.SelectionNet.detection2mask(ns3, batch, {k: v.clone() for k, v in pred3.items()}, cfg3, mode, True, *ths3)
        r = res[batch['scene'][0]['name']]
        pre = 'd2m_s3_%s_s0_' % mode
        out[pre + 'conf'] = r['conf'].numpy()
        out[pre + 'label_id'] = np.asarray(r['label_id'])
        out[pre + 'mask'] = np.packbits(r['mask'].numpy(), axis=1)
        out[pre + 'mask_shape'] = np.asarray(r['mask'].shape)
    for k, v in pred3.items():
        out['d2m_s3_pred_%s' % k] = v.numpy()
    out['d2m_s3_input_location'] = batch['input_location'].numpy()
    out['d2m_s3_batch_ids'] = batch['batch_ids'].numpy()
    out['d2m_s3_seg2vox0'] = seg2vox
    out['d2m_s3_vox2point0'] = np.asarray(batch['vox2point'][0])
    out['d2m_s3_vox_segments0'] = np.asarray(batch['vox_segments'][0])
    out['d2m_s3_names'] = np.asarray([batch['scene'][0]['name']])
    out['d2m_s3_ths'] = np.asarray(ths3)
    np.savez_compressed(os.path.join(OUT, 'detection2mask.npz'), **out)
    print('detection2mask.npz: %d arrays' % len(out))


def gen_detection2mask_nopool():
    """SelectionNet.detection2mask with cfg.do_segment_pooling = False (detection_net.py:436-445: the heat-maps are NOT
    projected through seg2vox, the predictions already live on the voxels).  The reference's branch only executes when every
    voxel of a scene is predicted foreground: it indexes the per-voxel semantics and vox2point with masks that have one
    column per FOREGROUND vote (:463, :470) -- any background voxel is a shape mismatch.  The fixture therefore predicts a
    furniture class everywhere; box2mask_amd pads background votes with zeros as the pooled branch does."""
    _install_stubs()
    sys.path.insert(0, REF)
    import models.detection_net as dn
    valid, id2idx, _, is_fg = synth.scannet_tables()
    cfg = SimpleNamespace(mlp_per_vox_semantics='mlp_per_vox_semantics', mlp_semantics='mlp_semantics',
                          network_heads=['mlp_offsets', 'mlp_bounds', 'mlp_bb_scores', 'mlp_semantics'],
                          do_segment_pooling=False)
    ns = SimpleNamespace(requires_voxel_outputs=False, semantic_valid_class_ids=valid, is_foreground=is_fg)
    out = {}
    for case, seed in (('a', 41), ('b', 47)):
        batch, _ = _scene_inputs(seed, n_scenes=2, target_voxels=2500)
        rng = np.random.default_rng(seed)
        coords = batch['vox_coords'].numpy()
        n_vox = coords.shape[0]
        seg_of_vox = batch['pooling_ids'].numpy()                      # segment row of every voxel (batch-wide)
        loc = (coords[:, 1:].astype(np.float32) * np.float32(0.02))   # vox_world_coords (dataloader.py:98-105)
        seg_loc = batch['input_location'].numpy()[seg_of_vox]
        # every voxel votes for its segment's box (furniture segments) or for a small box of its own
        off = batch['gt_bb_offsets'].numpy()[seg_of_vox] + (seg_loc - loc) + rng.normal(0, 0.03, (n_vox, 3)).astype(np.float32)
        bnd = np.maximum(batch['gt_bb_bounds'].numpy()[seg_of_vox] + rng.normal(0, 0.03, (n_vox, 3)).astype(np.float32), 0.04)
        bg_vox = ~batch['fg_instances'].numpy()[seg_of_vox]           # floor / wall segments: one 1 m box around the segment
        bnd[bg_vox] = (0.5 + rng.normal(0, 0.02, (int(bg_vox.sum()), 3))).astype(np.float32)
        logits = rng.normal(0.5, 2.0, (n_vox, 1)).astype(np.float32)
        fg_classes = np.nonzero(is_fg(torch.from_numpy(valid.numpy() if hasattr(valid, 'numpy') else np.asarray(valid)).long()).numpy())[0]
        sem_logits = rng.normal(0, 1, (n_vox, len(valid))).astype(np.float32)
        pick = fg_classes[(seg_of_vox * 7) % len(fg_classes)]        # one furniture class per segment: every voxel foreground
        sem_logits[np.arange(n_vox), pick] += 12.0
        flip = rng.random(n_vox) < 0.1                                 # some voxels of another (furniture) class
        sem_logits[flip, fg_classes[rng.integers(0, len(fg_classes), int(flip.sum()))]] += 20.0
        pred = {'mlp_offsets': torch.from_numpy(off.astype(np.float32)), 'mlp_bounds': torch.from_numpy(bnd.astype(np.float32)),
                'mlp_bb_scores': torch.from_numpy(logits), 'mlp_semantics': torch.from_numpy(sem_logits)}
        vbatch = {'input_location': torch.from_numpy(loc), 'batch_ids': torch.from_numpy(coords[:, 0].astype(np.int64)),
                  'scene': batch['scene'], 'vox2point': batch['vox2point']}
        ths = [0.5, 0.05, 0.3, 0.6]
        for mode in ('eval', 'train'):
            res = dn.SelectionNet.detection2mask(ns, vbatch, {k: v.clone() for k, v in pred.items()}, cfg, mode, True, *ths)
            for si, sc in enumerate(batch['scene']):
                r = res[sc['name']]
                pre = 'np_%s_%s_s%d_' % (case, mode, si)
                out[pre + 'conf'] = r['conf'].numpy()
                out[pre + 'label_id'] = np.asarray(r['label_id'])
                out[pre + 'mask'] = np.packbits(r['mask'].numpy(), axis=1)
                out[pre + 'mask_shape'] = np.asarray(r['mask'].shape)
                if mode != 'eval':
                    out[pre + 'reps'] = r['cluster_representatives'].numpy()
        for k, v in pred.items():
            out['np_%s_pred_%s' % (case, k)] = v.numpy()
        out['np_%s_input_location' % case] = loc
        out['np_%s_batch_ids' % case] = coords[:, 0].astype(np.int64)
        for si in range(len(batch['scene'])):
            out['np_%s_vox2point%d' % (case, si)] = np.asarray(batch['vox2point'][si])
        out['np_%s_names' % case] = np.asarray([s['name'] for s in batch['scene']])
        out['np_%s_ths' % case] = np.asarray(ths)
    np.savez_compressed(os.path.join(OUT, 'detection2mask_nopool.npz'), **out)
    print('detection2mask_nopool.npz: %d arrays, %s instances' % (len(out), [int(out[k][0]) for k in out if k.endswith('eval_s0_mask_shape')]))


def gen_losses():
    Holder = _install_stubs()
    sys.path.insert(0, REF)
    import models.model as M
    valid, id2idx, _, is_fg = synth.scannet_tables()

    class LUT:                                  # swallows the hard-coded .to('cuda') at model.py:199
        def __init__(self, t): self.t = t
        def __getitem__(self, i):
            r = self.t[i]
            return SimpleNamespace(to=lambda *_: r)

    out = {}
    # a: score loss on; b: IoU loss on, score loss weight 0 (early epoch); c: centre-score head; d: per-voxel
    # semantics head, losses over ALL segments (no foreground selection) -- the S3DIS-style configuration
    for case, seed, epoch in (('a', 5, 150), ('b', 9, 3), ('c', 11, 150), ('d', 13, 150)):
        batch, pred0 = _scene_inputs(seed)
        heads = ['mlp_offsets', 'mlp_bounds', 'mlp_bb_scores', 'mlp_semantics']
        rng = np.random.default_rng(100 + seed)
        if case == 'c':
            heads = heads + ['mlp_center_scores']
            pred0['mlp_center_scores'] = torch.from_numpy(rng.uniform(0, 0.4, (pred0['mlp_offsets'].shape[0], 1)).astype(np.float32))
        if case == 'd':
            heads = ['mlp_offsets', 'mlp_bounds', 'mlp_bb_scores', 'mlp_per_vox_semantics']
            del pred0['mlp_semantics']
            nvox = batch['vox_coords'].shape[0]
            pred0['mlp_per_vox_semantics'] = torch.from_numpy(rng.normal(0, 1, (nvox, 20)).astype(np.float32))
            batch['gt_per_vox_semantics'] = batch['gt_semantics'][batch['pooling_ids']]
        pred = {k: v.clone().requires_grad_(True) for k, v in pred0.items()}
        cfg = SimpleNamespace(mlp_offsets='mlp_offsets', mlp_bounds='mlp_bounds', mlp_bb_scores='mlp_bb_scores',
                              mlp_center_scores='mlp_center_scores', mlp_semantics='mlp_semantics',
                              mlp_per_vox_semantics='mlp_per_vox_semantics',
                              network_heads=heads,
                              loss_on_fg_instances=(case != 'd'), bb_supervision=(case in 'ab'),
                              use_bb_iou_loss=(case == 'b'), loss_weight_center_scores=0.7,
                              loss_weight_per_vox_semantics=0.9,
                              loss_weight_bb_offsets=1.0, loss_weight_bb_bounds=0.5, loss_weight_bb_iou=1.0,
                              loss_weight_bb_scores=1.0, loss_weight_semantics=1.0, min_bb_size=0.04,
                              mlp_bb_scores_start_epoch=100, mlp_center_scores_start_epoch=0)
        ns = SimpleNamespace(cfg=cfg, device='cpu',
                             detection_model=lambda sin, ids: {k: Holder(v) for k, v in pred.items()},
                             BCEWithLogitsLoss=torch.nn.BCEWithLogitsLoss(),
                             semantics_loss=torch.nn.CrossEntropyLoss(ignore_index=-100),
                             semantic_id2idx=LUT(id2idx))
        losses, _ = M.Model.compute_loss_detection(ns, batch, epoch)
        losses['optimization_loss'].backward()
        for k, v in losses.items():
            out['loss_%s_%s' % (case, k)] = np.asarray(v.detach().numpy() if torch.is_tensor(v) else v, dtype=np.float64)
        for k, v in pred.items():
            out['loss_%s_pred_%s' % (case, k)] = pred0[k].numpy()
            out['loss_%s_grad_%s' % (case, k)] = v.grad.numpy() if v.grad is not None else np.zeros_like(pred0[k].numpy())
        for k in ('input_location', 'gt_bb_offsets', 'gt_bb_bounds', 'gt_semantics', 'fg_instances', 'pooling_ids',
                  'gt_per_vox_semantics'):
            if k in batch:
                out['loss_%s_batch_%s' % (case, k)] = batch[k].numpy()
        out['loss_%s_epoch' % case] = np.asarray(epoch)
    np.savez_compressed(os.path.join(OUT, 'losses.npz'), **out)
    print('losses.npz: %d arrays' % len(out))


def gen_prepare():
    """Scene preparation: ScanNet.__getitem__ (dataloader.py:53-123, 'test' mode) and collate_fn (:946-984) of the
    real reference on synthetic raw scenes.  dataprocessing.{scannet,arkitscenes,s3dis} only LOAD scenes (open3d,
    pyviz3d, ... absent here), so they are replaced by empty stand-ins whose process_scene returns the synthetic
    scene; numpy.lib.type_check (an unused import of dataloader.py:4, gone in numpy 2) and
    ME.utils.batched_coordinates likewise.  Everything this fixture records is computed by the reference's own
    lines: np.round / np.unique / sklearn ball tree / the segment loop / to_unique."""
    _install_stubs()
    sys.modules['MinkowskiEngine'].utils = SimpleNamespace(
        batched_coordinates=lambda c, dtype=None: synth.batched_coordinates(c))
    tc = types.ModuleType('numpy.lib.type_check'); tc._is_type_dispatcher = None
    sys.modules['numpy.lib.type_check'] = tc
    dp = types.ModuleType('dataprocessing'); dp.__path__ = []
    sys.modules['dataprocessing'] = dp
    for n in ('scannet', 'arkitscenes', 's3dis'):
        m = types.ModuleType('dataprocessing.' + n); sys.modules['dataprocessing.' + n] = m; setattr(dp, n, m)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import models.dataloader as D

    rng = np.random.default_rng(77)
    scenes = []
    # 0/1: room-shaped surfaces (shifted so that coordinates go negative), 2 cm;  2: a dense blob at 5 cm with many
    # points per voxel;  3: three points
    for seed, tv in ((0, 9000), (1, 6000)):
        sc = synth.make_scene(seed, target_voxels=tv, pts_per_m2=9000.0, points_only=True)
        off = np.array([0.37, 1.21, 0.0]) * (seed + 1)
        sc['positions'] = sc['positions'] - off
        sc['labels']['per_instance_bb_centers'] = (sc['labels']['per_instance_bb_centers'] - off).astype(np.float32)
        sc['voxel_size'] = 0.02
        scenes.append(sc)
    P = 4000
    bp = rng.normal(0, 0.25, (P, 3))
    cell = np.floor((bp + 2.0) / 0.2).astype(np.int64)
    blob_labels = {     # overlapping boxes: segments inside an overlap go to the smallest box (smallest_bb_heuristic)
        'unique_instances': np.arange(5),
        'per_instance_semantics': np.array([5, 7, 9, 2, 0], np.int32),
        'per_instance_bb_centers': np.array([[0, 0, 0], [0.15, 0, 0], [0, 0.1, 0], [0, 0, -1], [3, 3, 3]], np.float32),
        'per_instance_bb_bounds': np.array([[.4, .4, .4], [.4, .4, .4], [.25, .25, .25], [1, 1, .1], [.1, .1, .1]], np.float32),
    }
    blob_seg = (cell[:, 0] * 400 + cell[:, 1] * 20 + cell[:, 2]) * 3 + 1
    blob_labels['seg2inst'] = rng.integers(0, 5, int(blob_seg.max()) + 1).astype(np.int32)
    scenes.append({'name': 'blob', 'positions': bp, 'colors': rng.uniform(0, 1, (P, 3)),
                   'normals': rng.normal(size=(P, 3)), 'segments': blob_seg, 'voxel_size': 0.05,
                   'labels': blob_labels})
    scenes.append({'name': 'tiny', 'positions': np.array([[0.1, 0.2, 0.3], [0.101, 0.2, 0.3], [1.0, -0.5, 0.25]]),
                   'colors': rng.uniform(0, 1, (3, 3)), 'normals': rng.normal(size=(3, 3)),
                   'segments': np.array([5, 5, 9]), 'voxel_size': 0.02})
    out = {'n_scenes': np.array(len(scenes))}
    items = []
    for i, sc in enumerate(scenes):
        D.scannet.process_scene = lambda name, mode, cfg, do_augmentations=False, _sc=sc: (_sc, None)
        ds = D.ScanNet.__new__(D.ScanNet)
        ds.cfg = SimpleNamespace(voxel_size=sc['voxel_size'], use_normals_input=True, do_segment_pooling=True)
        ds.mode = 'test'; ds.do_augmentations = False; ds.data_list = [sc['name']]
        ret = ds[0]
        items.append(ret)
        for k in ('positions', 'colors', 'normals', 'segments'):
            out['s%d_in_%s' % (i, k)] = np.asarray(sc[k])
        out['s%d_in_voxel_size' % i] = np.array(sc['voxel_size'])
        for k in ('vox_coords', 'vox2point', 'point2vox', 'vox_segments', 'vox_features', 'vox_world_coords',
                  'seg2vox', 'seg2point', 'input_location'):
            out['s%d_%s' % (i, k)] = np.asarray(ret[k])
    # ---- 'train' mode with weak box supervision (configs/scannet.txt: bb_supervision, smallest_bb_heuristic):
    # approx_association + bbs_supervision (dataloader.py:165-314) on the two room scenes.  np.int (removed from
    # numpy 1.24 on) is what dataloader.py:187,244,... still spells: aliased for this run only.
    if not hasattr(np, 'int'):
        np.int = int
    train_items = []
    for i in (0, 1, 2):
        sc = scenes[i]
        D.scannet.process_scene = lambda name, mode, cfg, do_augmentations=False, _sc=sc: (_sc, _sc['labels'])
        ds = D.ScanNet.__new__(D.ScanNet)
        ds.cfg = SimpleNamespace(voxel_size=sc['voxel_size'], use_normals_input=True, do_segment_pooling=True,
                                 bb_supervision=True, point_association=False, majority_vote=False,
                                 smallest_bb_heuristic=True, dropout_boxes=None, noisy_boxes=None)
        ds.mode = 'train'; ds.do_augmentations = False; ds.data_list = [sc['name']]
        ret = ds[0]
        train_items.append(ret)
        for k in ('unique_instances', 'per_instance_semantics', 'per_instance_bb_centers', 'per_instance_bb_bounds',
                  'seg2inst'):
            out['s%d_label_%s' % (i, k)] = np.asarray(sc['labels'][k])
        out['s%d_inst_per_point' % i] = np.asarray(ret['pseudo_inst'][0])
        out['s%d_inst_per_seg' % i] = np.asarray(ret['pseudo_inst'][1])
        for k in ('fg_instances', 'gt_bb_bounds', 'gt_bb_offsets', 'gt_semantics'):
            out['s%d_%s' % (i, k)] = np.asarray(ret[k])
    # the two randomised supervision options (seeded by the scene name, dataloader.py:210-232) on scene 1
    sc = scenes[1]
    D.scannet.process_scene = lambda name, mode, cfg, do_augmentations=False, _sc=sc: (_sc, _sc['labels'])
    ds = D.ScanNet.__new__(D.ScanNet)
    ds.cfg = SimpleNamespace(voxel_size=sc['voxel_size'], use_normals_input=True, do_segment_pooling=True,
                             bb_supervision=True, point_association=False, majority_vote=False,
                             smallest_bb_heuristic=True, dropout_boxes=0.15, noisy_boxes=0.004)
    ds.mode = 'train'; ds.do_augmentations = False; ds.data_list = [sc['name']]
    ret = ds[0]
    out['s1_name'] = np.array(sc['name'])
    out['s1_noisy_inst_per_seg'] = np.asarray(ret['pseudo_inst'][1])
    out['s1_noisy_inst_per_point'] = np.asarray(ret['pseudo_inst'][0])
    out['s1_noisy_bbs_min'], out['s1_noisy_bbs_max'] = (np.asarray(v) for v in ret['noisy_bbs'])
    for k in ('fg_instances', 'gt_bb_bounds', 'gt_semantics'):
        out['s1_noisy_%s' % k] = np.asarray(ret[k])
    tb = D.collate_fn(SimpleNamespace(do_segment_pooling=True), 'train')(train_items[:2])
    for k in ('gt_bb_bounds', 'gt_bb_offsets', 'gt_semantics', 'fg_instances'):
        out['collate_%s' % k] = tb[k].numpy()
    # collate of the two 2 cm scenes (one voxel size per config in the reference)
    cf = D.collate_fn(SimpleNamespace(do_segment_pooling=True), 'test')
    b = cf([items[0], items[1]])
    for k in ('vox_features', 'batch_ids', 'input_location', 'pooling_ids'):
        out['collate_%s' % k] = b[k].numpy()
    np.savez_compressed(os.path.join(OUT, 'prepare.npz'), **out)
    print('prepare.npz: %d arrays, %s voxels' % (len(out), [len(it['vox_coords']) for it in items]))


def _dataloader_module():
    """models/dataloader.py of the reference behind stand-ins for the modules that only LOAD data (see gen_prepare)."""
    _install_stubs()
    sys.modules['MinkowskiEngine'].utils = SimpleNamespace(
        batched_coordinates=lambda c, dtype=None: synth.batched_coordinates(c))
    tc = types.ModuleType('numpy.lib.type_check'); tc._is_type_dispatcher = None
    sys.modules['numpy.lib.type_check'] = tc
    dp = types.ModuleType('dataprocessing'); dp.__path__ = []
    sys.modules['dataprocessing'] = dp
    for n in ('scannet', 'arkitscenes', 's3dis'):
        m = types.ModuleType('dataprocessing.' + n); sys.modules['dataprocessing.' + n] = m; setattr(dp, n, m)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import models.dataloader as D
    if not hasattr(np, 'int'):
        np.int = int                      # dataloader.py still spells np.int (removed from numpy 1.24 on)
    # dataloader.py:267,592,868,916 index stats.mode(x, None)[0][0]: the array-valued result of SciPy < 1.11.  The SciPy
    # of this image returns scalars; keepdims=True is that old behaviour (same values, same tie rule).
    import scipy.stats as st
    D.stats = SimpleNamespace(mode=lambda a, axis=0: st.mode(a, axis, keepdims=True))
    # the one function of dataprocessing/s3dis.py the dataset class calls (s3dis.py:79-82, two comparisons)
    D.s3dis.semantics_to_forground_mask = lambda semantics, cfg=None: (semantics > 2) if cfg.ignore_wall_ceiling_floor \
        else (semantics >= 0)
    return D


def gen_prepare2():
    """The remaining branches of the dataset classes from the REAL reference: ScanNet
    majority_vote / point_association / mask_supervision (dataloader.py:138-272), ARKitScenes.__getitem__ at 4 cm with
    oriented-box association (:385-621) and S3DIS box / mask supervision (:737-927), on the two labelled scenes of
    prepare2_scenes()."""
    D = _dataloader_module()
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from golden_scenes import prepare2_scenes
    scenes = prepare2_scenes()
    out = {}

    def run(cls, mod, sc, mode, **cfgkw):
        mod.process_scene = lambda name, m, cfg, do_augmentations=False, subsample_rate=None, _sc=sc: (_sc, _sc['labels'])
        ds = cls.__new__(cls)
        ds.cfg = SimpleNamespace(voxel_size=sc['voxel_size'], use_normals_input=True, dropout_boxes=None, noisy_boxes=None,
                                 **cfgkw)
        ds.mode = mode; ds.do_augmentations = False; ds.data_list = [sc['name']]; ds.data_class = mod
        ds.subsample_rate = None
        return ds[0]

    def put(tag, ret, keys):
        for k in keys:
            if k == 'pseudo_inst':
                for j, v in enumerate(ret.get(k, ())):
                    if v is not None:
                        out['%s_pseudo%d' % (tag, j)] = np.asarray(v)
            elif k in ret:
                out['%s_%s' % (tag, k)] = np.asarray(ret[k])
    TGT = ('pseudo_inst', 'fg_instances', 'gt_bb_bounds', 'gt_bb_offsets', 'gt_semantics', 'gt_per_vox_semantics',
           'instance_ids', 'vox_instances')
    for i, sc in enumerate(scenes):
        out['s%d_rotations' % i] = sc['labels']['per_instance_bb_rotations']
        # ---- ScanNet
        for heur in (True, False):
            r = run(D.ScanNet, D.scannet, sc, 'train', do_segment_pooling=True, bb_supervision=True, point_association=False,
                    majority_vote=True, smallest_bb_heuristic=heur)
            put('s%d_scannet_majority_h%d' % (i, heur), r, TGT)
            r = run(D.ScanNet, D.scannet, sc, 'train', do_segment_pooling=False, bb_supervision=True, point_association=True,
                    majority_vote=False, smallest_bb_heuristic=heur)
            put('s%d_scannet_point_h%d' % (i, heur), r, TGT + ('input_location',))
        for pool in (True, False):
            r = run(D.ScanNet, D.scannet, sc, 'train', do_segment_pooling=pool, bb_supervision=False)
            put('s%d_scannet_mask_p%d' % (i, pool), r, TGT)
        # ---- ARKitScenes (4 cm voxels as configs/arkitscenes.txt)
        sc4 = dict(sc, voxel_size=0.04)
        r = run(D.ARKitScenes, D.arkitscenes, sc4, 'test', do_segment_pooling=True)
        put('s%d_arkit' % i, r, ('vox_coords', 'vox2point', 'point2vox', 'vox_segments', 'vox_features', 'seg2vox',
                                 'seg2point', 'input_location'))
        r = run(D.ARKitScenes, D.arkitscenes, sc4, 'train', do_segment_pooling=True, bb_supervision=True,
                point_association=False)
        out['s%d_arkit_seg_pseudo0' % i], out['s%d_arkit_seg_pseudo1' % i] = (np.asarray(v) for v in
                                                                                 D.ARKitScenes.approx_association(
            SimpleNamespace(cfg=SimpleNamespace()), sc['labels'], sc, False, np.unique(r['vox_segments'])))
        put('s%d_arkit_seg' % i, r, TGT)
        # (the item itself cannot be made with point_association: :550 takes len(unique_segs) of None without segment
        # pooling and :511 raises with it -- only the association is pinned)
        out['s%d_arkit_point_pseudo0' % i] = np.asarray(D.ARKitScenes.approx_association(
            SimpleNamespace(cfg=SimpleNamespace()), sc['labels'], sc, True, np.unique(r['vox_segments']))[0])
        r = run(D.ARKitScenes, D.arkitscenes, sc4, 'train', do_segment_pooling=True, bb_supervision=False)
        put('s%d_arkit_mask' % i, r, TGT)
        # ---- S3DIS
        for ign in (True, False):
            r = run(D.S3DIS, D.s3dis, sc, 'train', do_segment_pooling=True, bb_supervision=True, point_association=False,
                    ignore_wall_ceiling_floor=ign)
            a = D.S3DIS.approx_association(SimpleNamespace(cfg=SimpleNamespace(ignore_wall_ceiling_floor=ign)),
                                           sc['labels'], sc, False, np.unique(r['vox_segments']))
            for j, v in enumerate(a):
                out['s%d_s3dis_i%d_assoc%d' % (i, ign, j)] = np.asarray(v)
            put('s%d_s3dis_i%d' % (i, ign), r, TGT)
            a = D.S3DIS.approx_association(SimpleNamespace(cfg=SimpleNamespace(ignore_wall_ceiling_floor=ign)),
                                           sc['labels'], sc, True, np.unique(r['vox_segments']))
            out['s%d_s3dis_i%d_point0' % (i, ign)], out['s%d_s3dis_i%d_point1' % (i, ign)] = (np.asarray(v) for v in a)
        r = run(D.S3DIS, D.s3dis, sc, 'val', do_segment_pooling=True, bb_supervision=False, ignore_wall_ceiling_floor=True)
        put('s%d_s3dis_mask' % i, r, TGT)
        # the voxelisation block of S3DIS.__getitem__ itself (:671-730)
        put('s%d_s3dis' % i, r, ('vox_coords', 'vox2point', 'point2vox', 'vox_segments', 'vox_features', 'seg2vox',
                                 'seg2point', 'input_location'))
    np.savez_compressed(os.path.join(OUT, 'prepare2.npz'), **out)
    print('prepare2.npz: %d arrays, %.1f kB' % (len(out), os.path.getsize(os.path.join(OUT, 'prepare2.npz')) / 1e3))


def gen_eval():
    """AP evaluation: assign_instances_for_scan / evaluate_matches / compute_averages of the real
    /root/reference/utils/eval_metric.py on synthetic predictions (noisy copies of the ground-truth instances,
    duplicates, wrong labels, tiny masks, void regions).  np.float / np.bool (removed numpy aliases the file still
    spells, eval_metric.py:111,139) are aliased for this run only."""
    import tempfile
    for a, t in (('float', float), ('bool', bool), ('int', int)):
        if not hasattr(np, a):
            setattr(np, a, t)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import utils.eval_metric as E
    rng = np.random.default_rng(5)
    valid = E.VALID_CLASS_IDS
    out = {'n_scenes': np.array(3)}
    matches = {}
    for si in range(3):
        n = 30000 + 5000 * si
        # ground truth: contiguous index ranges as instances; labels incl. void classes 1/2, one unannotated block (0)
        cuts = np.sort(rng.choice(np.arange(200, n - 200), 24 + si, replace=False))
        inst_of = np.searchsorted(cuts, np.arange(n), side='right')
        n_inst = inst_of.max() + 1
        inst_label = rng.choice(np.concatenate([valid[:6], [1, 2]]), n_inst)
        inst_label[0] = 0
        gt_ids = np.where(inst_label[inst_of] == 0, 0, inst_label[inst_of] * 1000 + inst_of + 1).astype(np.int64)
        # tiny ground-truth instance (< 100 vertices) of a valid class inside instance 3
        tiny = np.nonzero(inst_of == 3)[0][:60]
        gt_ids[tiny] = valid[2] * 1000 + 900
        masks, labels, confs = [], [], []
        for g in range(n_inst):
            if rng.random() < 0.15:
                continue                                        # missed instance
            for rep in range(1 + (rng.random() < 0.2)):         # sometimes a duplicate detection
                m = inst_of == g
                flip = rng.random(n) < rng.choice([0.01, 0.05, 0.2, 0.45])
                m = np.where(flip & (np.abs(np.arange(n) - np.nonzero(inst_of == g)[0].mean()) < 3000), ~m, m)
                masks.append(m)
                lab = inst_label[g] if rng.random() < 0.85 else rng.choice(valid[:6])
                labels.append(lab if lab > 2 else valid[0])
                confs.append(rng.random())
        for _ in range(3):                                      # small and spurious detections
            m = np.zeros(n, bool); s0 = rng.integers(0, n - 400); m[s0:s0 + rng.choice([40, 99, 100, 350])] = True
            masks.append(m); labels.append(rng.choice(valid[:6])); confs.append(rng.random())
        masks.append(np.zeros(n, bool)); labels.append(1); confs.append(0.9)     # label outside the benchmark
        pred = {'conf': np.array(confs, np.float32), 'label_id': np.array(labels, np.int32), 'mask': np.stack(masks)}
        name = 'scene%d' % si
        with tempfile.NamedTemporaryFile('w', suffix='.txt', delete=False) as f:
            f.write('\n'.join(str(int(v)) for v in gt_ids) + '\n')
        gt2pred, pred2gt = E.assign_instances_for_scan(name, pred, f.name)
        os.unlink(f.name)
        matches[name] = {'gt': gt2pred, 'pred': pred2gt}
        out['s%d_gt_ids' % si] = gt_ids
        out['s%d_conf' % si] = pred['conf']; out['s%d_label_id' % si] = pred['label_id']
        out['s%d_mask' % si] = np.packbits(pred['mask'], axis=1)
        out['s%d_n' % si] = np.array(n)
    ap, _ = E.evaluate_matches(matches)
    avgs = E.compute_averages(ap)
    out['ap'] = ap
    out['all_ap'] = np.array([avgs['all_ap'], avgs['all_ap_50%'], avgs['all_ap_25%']])
    out['class_ap'] = np.array([[avgs['classes'][c]['ap'], avgs['classes'][c]['ap50%'], avgs['classes'][c]['ap25%']]
                                for c in E.CLASS_LABELS])
    # a one-scene table as well (exercises has_gt / has_pred per class differently)
    ap1, _ = E.evaluate_matches({'scene0': matches['scene0']})
    out['ap_scene0'] = ap1
    np.savez_compressed(os.path.join(OUT, 'eval_metric.npz'), **out)
    print('eval_metric.npz: mAP %.4f  AP50 %.4f  AP25 %.4f' % tuple(out['all_ap']))


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    which = sys.argv[1:] or ['iou_nms', 'detection2mask', 'detection2mask_nopool', 'losses', 'prepare', 'prepare2', 'eval']
    if 'iou_nms' in which:
        gen_iou_nms()
    if 'detection2mask' in which:
        gen_detection2mask()
    if 'detection2mask_nopool' in which:
        gen_detection2mask_nopool()
    if 'losses' in which:
        gen_losses()
    if 'prepare' in which:
        gen_prepare()
    if 'prepare2' in which:
        gen_prepare2()
    if 'eval' in which:
        gen_eval()
