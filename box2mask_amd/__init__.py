"""box2mask_amd — MI355X (gfx950) native hot path of Box2Mask.

Sparse-voxel U-Net forward/backward (coordinate hashing, tile rulebooks, MFMA gather-GEMM with
LDS-resident output strips, batch norm, segment pooling) and box-vote clustering
(non-maximum clustering, mask projection, mask NMS, label histogram) as hand-written HIP kernels
behind a C ABI (include/b2m.h), driven through the reference's own Python class surface:

    from box2mask_amd.model import Model            # /root/reference/models/model.py
    from box2mask_amd.detection_net import SelectionNet
    import box2mask_amd.nn as ME                    # MinkowskiEngine-named layers
    from box2mask_amd import iou_nms                # NMS_clustering, mask_NMS, set_IOUs, ...

Importing the package does not need a GPU; constructing a Model / calling any operator does, and
fails loudly when the HIP extension or the device is missing (there is no CPU fallback).
"""
__version__ = '0.1.0'
