"""The ~25 configuration fields the hot path reads, with the defaults of
/root/reference/config_loader.py and the values of configs/scannet.txt.  The reference's
configargparse front end is out of scope; any object with these attributes works."""
from __future__ import annotations

from types import SimpleNamespace


def scannet_config(**overrides):
    cfg = SimpleNamespace(
        # head-name constants (config_loader.py:284-331)
        mlp_offsets='mlp_offsets', mlp_bounds='mlp_bounds', mlp_bb_scores='mlp_bb_scores',
        mlp_center_scores='mlp_center_scores', mlp_semantics='mlp_semantics',
        mlp_per_vox_semantics='mlp_per_vox_semantics',
        network_heads=['mlp_offsets', 'mlp_bounds', 'mlp_bb_scores', 'mlp_semantics'],   # configs/scannet.txt:12
        in_channels=6, layers=2, load_unused_head=False,
        do_segment_pooling=True, max_pool_segments_detection_net=False, mlp_bounds_relu=False,
        min_bb_size=0.04, multigpu=False,
        bb_supervision=True, loss_on_fg_instances=True, use_bb_iou_loss=False,
        loss_weight_bb_offsets=1.0, loss_weight_bb_bounds=0.5, loss_weight_bb_iou=1.0,
        loss_weight_bb_scores=1.0, loss_weight_center_scores=1.0, loss_weight_semantics=1.0,
        loss_weight_per_vox_semantics=1.0,
        mlp_bb_scores_start_epoch=100, mlp_center_scores_start_epoch=0,
        eval_ths=[0.5, 0.05, 0.3, 0.6],                                                    # configs/scannet.txt:15
        checkpoint_path='experiments/scannet/checkpoints/', voxel_size=0.02, batch_size=8, lr=1e-3,
        half_inference=False,        # build extension (no reference field): inference on half activations, see model.Model
        half_training=False,         # build extension: training passes with the trunk's activations / gradients in half (half_train.py)
        half_loss_scale=1024.0,      # ... and the factor its gradients are scaled by inside the half region (a power of two)
    )
    for k, v in overrides.items():
        setattr(cfg, k, v)
    return cfg
