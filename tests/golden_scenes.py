"""Inputs of tests/golden/prepare2.npz: the two labelled raw scenes, rebuilt from their seeds (the fixture holds the
reference's OUTPUTS only).  Shared by tools/gen_golden.py (which feeds them to the real dataset classes) and the tests."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from box2mask_amd import synth  # noqa: E402


def prepare2_scenes():
    """The labelled raw scenes of the prepare2 fixture (scene 1 and the blob of gen_prepare, rebuilt from the same
    seeds) with the extra labels the other dataset classes read."""
    rng = np.random.default_rng(77)
    sc = synth.make_scene(1, target_voxels=6000, pts_per_m2=9000.0, points_only=True)
    off = np.array([0.37, 1.21, 0.0]) * 2
    sc['positions'] = sc['positions'] - off
    sc['labels']['per_instance_bb_centers'] = (sc['labels']['per_instance_bb_centers'] - off).astype(np.float32)
    P = 4000
    bp = rng.normal(0, 0.25, (P, 3))
    cell = np.floor((bp + 2.0) / 0.2).astype(np.int64)
    blob_labels = {
        'unique_instances': np.arange(5),
        'per_instance_semantics': np.array([5, 7, 9, 2, 0], np.int32),
        'per_instance_bb_centers': np.array([[0, 0, 0], [0.15, 0, 0], [0, 0.1, 0], [0, 0, -1], [3, 3, 3]], np.float32),
        'per_instance_bb_bounds': np.array([[.4, .4, .4], [.4, .4, .4], [.25, .25, .25], [1, 1, .1], [.1, .1, .1]], np.float32),
    }
    blob_seg = (cell[:, 0] * 400 + cell[:, 1] * 20 + cell[:, 2]) * 3 + 1
    blob_labels['seg2inst'] = rng.integers(0, 5, int(blob_seg.max()) + 1).astype(np.int32)
    blob = {'name': 'blob', 'positions': bp, 'colors': rng.uniform(0, 1, (P, 3)), 'normals': rng.normal(size=(P, 3)),
            'segments': blob_seg, 'labels': blob_labels}
    out = []
    for scene, vs in ((sc, 0.02), (blob, 0.05)):
        lab = scene['labels']
        n_inst = len(lab['unique_instances'])
        r2 = np.random.default_rng(5 + n_inst)
        ang = r2.uniform(0, np.pi, n_inst)
        rot = np.zeros((n_inst, 3, 3))
        rot[:, 0, 0] = np.cos(ang); rot[:, 0, 1] = -np.sin(ang); rot[:, 1, 0] = np.sin(ang); rot[:, 1, 1] = np.cos(ang)
        rot[:, 2, 2] = 1
        lab['per_instance_bb_rotations'] = rot.reshape(n_inst, 9)
        inst = np.asarray(lab['seg2inst'])[np.asarray(scene['segments'])]
        lab['semantics'] = np.asarray(lab['per_instance_semantics'])[inst]           # per-point labels (no pooling)
        lab['bb_bounds'] = np.asarray(lab['per_instance_bb_bounds'])[inst]
        lab['bb_centers'] = np.asarray(lab['per_instance_bb_centers'])[inst]
        scene['voxel_size'] = vs
        out.append(scene)
    return out
