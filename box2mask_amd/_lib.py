"""ctypes binding of the C ABI declared in include/b2m.h.

There is no CPU fallback: if the shared library is missing the import fails loudly and tells the
user how to build it.  Device pointers are passed as integers (``tensor.data_ptr()``) and the
stream is torch's current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('B2M_LIB_PATH', os.path.join(_HERE, 'libb2m_hip.so'))   # override: A/B of two builds

P = C.c_void_p
I32 = C.c_int32
I64 = C.c_int64
F32 = C.c_float
F64 = C.c_double

# name -> argument types (return type is always int).  Must list EVERY symbol of include/*.h;
# tests/test_abi.py cross-checks this table against the header.
PROTOTYPES = {
    'b2m_coords_build': [P, I64, P, P, I64, P, P],
    'b2m_morton_keys': [P, I64, P, P],
    'b2m_hilbert_keys': [P, I64, I32, P, P],
    'b2m_coords_stride': [P, I64, I32, P, P, P, P, P, I64, P, C.POINTER(I64), P],
    'b2m_kernel_map': [P, I64, I32, I32, P, P, I64, P, I32, I32, I32, P, I64, P],
    'b2m_occupancy': [P, I64, I32, I32, I32, I32, P, I64, P],
    'b2m_kernel_map_rulebook': [P, I64, I32, I32, P, P, I64, P, I32, I32, I32, P, P, P, P],
    'b2m_stride_tables': [P, P, I64, I64, P, I64, P, I64, P],
    'b2m_rulebook': [P, I64, I32, I64, P, P, P, P, P],
    'b2m_rulebook_balance': [P, I32, I64, P],
    'b2m_detection_loss': [P, I64, P, I64, P, I64, P, I64, I32, P, P, P, P, P, I64, F64, P, F32, F32, F32, F32, F32, P, P, P, P, P, P, P, P],
    'b2m_radix_argsort': [P, I64, C.c_uint64, P, P, P, P],
    'b2m_conv_fwd': [P, I64, I32, P, I64, I32, I64, P, I32, P, P, P, P, I64, P, I64, I32, I32, P],
    'b2m_conv_fwd_affine': [P, I64, I32, P, I64, I32, I64, P, I32, P, P, P, I64, P, I64, I32, P, P, P, I64, I32, P, P],
    'b2m_conv_fwd_h': [P, I64, I32, P, I64, I32, I64, P, I32, P, P, P, I64, P, I64, I32, P, P, P, I64, I32, P],
    'b2m_conv_fwd_h_stats': [P, I64, I32, P, I64, I32, I64, P, I32, P, P, P, I64, P, I64, I32, P, P],
    'b2m_weight_pack_h': [P, I64, I32, I32, I32, I32, P, P],
    'b2m_conv_fwd_stats': [P, I64, I32, P, I64, I32, I64, P, I32, P, P, P, P, I64, P, I64, I32, I32, P, P, P],
    'b2m_conv_wgrad_h': [P, I64, I32, I64, P, I64, I32, P, P, P, I64, I32, P, I64, I64, I32, F32, P],
    'b2m_weight_pack_h_t': [P, I32, I32, I32, I32, I32, I32, P, P],
    'b2m_bn_stats_finalize_h': [P, I64, I64, I32, P, P, P, F32, F32, P, P, P, P, P, P, P],
    'b2m_bn_stats_h': [P, I64, I64, I32, P, P, P],
    'b2m_bn_apply_h': [P, I64, I64, I32, P, P, P, I64, I32, P, I64, P],
    'b2m_bn_bwd_reduce_h': [P, I64, P, I64, P, I64, I64, I32, P, P, I32, P, P, P, P, P, P, F32, P],
    'b2m_bn_bwd_apply_h': [P, I64, P, I64, P, I64, I64, I32, P, P, P, P, F64, P, I32, P, P, P, I64, P, I64, P],
    'b2m_clock_probe': [P, I32, P],
    'b2m_xchg_allreduce': [P, I32, P, I32, I32, C.c_uint64, P, P, P],
    'b2m_conv_up': [P, I64, I32, P, I64, I32, I64, P, I32, P, P, P, P, P, I64, I32, I64, I32, P, P, P, I64, I32, P, P],
    'b2m_weight_pack': [P, I64, I32, I32, I32, I32, I32, I32, I32, P, P],
    'b2m_weight_pack_run': [P, I32, I64, P],
    'b2m_weight_pack_h_run': [P, I32, I64, P],
    'b2m_conv_wgrad': [P, I64, I32, I64, P, I64, I32, P, P, P, I64, I32, P, I64, I64, P, P],
    'b2m_conv_wgrad_tr': [P, I64, I32, I64, P, I64, I32, P, P, P, I64, I32, P, I64, I64, P, P],
    'b2m_bn_stats': [P, I64, I64, I32, P, P, P],
    'b2m_bn_tilestats': [P, I64, I32, P, P, P],
    'b2m_bn_tilestats_finalize': [P, I64, I64, I32, P, P, P, P, F32, F32, P, P, P, P, P, P, P],
    'b2m_bn_finalize': [P, F64, P, I32, P, P, F32, F32, P, P, P, P, P, P, P],
    'b2m_bn_apply': [P, I64, I64, I32, P, P, P, I64, I32, P, I64, P],
    'b2m_bn_bwd_reduce': [P, I64, P, I64, P, I64, I64, I32, P, P, I32, P, P, P, P, P, P, P],
    'b2m_bn_stats_finalize': [P, I64, I64, I32, P, P, P, P, F32, F32, P, P, P, P, P, P, P],
    'b2m_bn_bwd_apply': [P, I64, P, I64, P, I64, I64, I32, P, P, P, P, F64, P, I32, P, P, P, I64, P, I64, P],
    'b2m_bn_apply2': [P, I64, P, I64, I64, I32, P, P, P, P, I32, P, I64, P],
    'b2m_bn_bwd_reduce2': [P, I64, P, I64, P, I64, P, I64, I64, I32, P, P, P, P, I32, P, P, P],
    'b2m_bn_bwd_apply2': [P, I64, P, I64, P, I64, P, I64, I64, I32, P, P, P, P, P, P, P, F64, P, I32, P, I64, P, I64, P, P, P, P, P, P],
    'b2m_bn_small_fwd': [P, I64, I64, I32, P, P, F32, F32, P, P, P, P, P, P, P, I64, I32, P, I64, P],
    'b2m_bn_small_fwd_stats': [P, I64, I64, I32, P, P],
    'b2m_bn_small_fwd_apply': [P, P, I64, I64, I32, P, P, F32, F32, P, P, P, P, P, P, P, I64, I32, P, I64, P],
    'b2m_bn_small_bwd_phase': [I32, P, I64, P, I64, P, I64, I64, I32, P, P, P, I32, P, P, P, P, P, I64, P, I64, P, P, P],
    'b2m_bn_small_bwd': [P, I64, P, I64, P, I64, I64, I32, P, P, P, I32, P, P, P, P, P, I64, P, I64, P],
    'b2m_relu_fwd': [P, I64, P, P],
    'b2m_relu_bwd': [P, P, I64, P, P],
    'b2m_add': [P, P, I64, P, P],
    'b2m_segment_pool_fwd': [P, I64, I64, I32, P, I64, I32, P, P, P, P, P],
    'b2m_segment_mean_sorted': [P, I64, I64, I32, P, P, I64, P, P, P],
    'b2m_segment_pool_bwd': [P, I64, I32, P, I64, I32, P, P, P, I64, P],
    'b2m_nmc': [P, I32, F32, I32, P, P, P, P, P, P],
    'b2m_nmc_batch': [P, P, I32, I32, F32, P, P, P, P, P, P],
    'b2m_mask_project': [P, I32, P, I32, P, P, I64, F32, P, I64, P],
    'b2m_mask_nms': [P, I32, I64, F32, P, P, P, P],
    'b2m_label_hist': [P, I64, P, I32, P, I64, I32, P, P],
    'b2m_mask_gather': [P, I64, P, I32, P, I64, P, P],
    'b2m_mask_project_batch': [P, I32, I64, I64, F32, P],
    'b2m_mask_nms_batch': [P, I32, I32, F32, P],
    'b2m_label_hist_batch': [P, I32, I64, I32, P],
    'b2m_mask_gather_batch': [P, I32, I64, I64, P],
    'b2m_mask_gather_batch_t': [P, I32, I64, I64, I64, P, P],
    'b2m_mask_hist': [P, I64, I32, P, I64, I32, P, P],
    'b2m_mask_pack': [P, I32, I64, P, I64, P],
    'b2m_set_ious': [P, P, I64, P, P],
    # include/b2m_prepare.h
    'b2m_vox_shift': [P, I64, P, P, P],
    'b2m_unique_insert_async': [P, I64, P, I64, P, P, P, P],
    'b2m_vox_keys': [P, I64, P, F64, P, P, P],
    'b2m_sort_u64': [P, I64, P],
    'b2m_unique_rank': [P, I64, P, P, I64, P, I64, P, P],
    'b2m_vox_decode': [P, I64, I32, P, P],
    'b2m_vox_nearest': [P, I64, P, F64, P, P, I64, I64, P, P, P],
    'b2m_vox_gather': [P, I64, P, P, P, P, P, P],
    'b2m_seg_centroid': [P, P, I64, I64, F64, P, P, P, P, P],
    'b2m_box_membership': [P, I64, P, P, P, I32, P, P, P, P],
    'b2m_seg_box_vote': [P, I64, P, P, I64, I64, P, P, P, P, I32, P, P, P, P, P],
    'b2m_obb_membership': [P, I64, P, P, P, I32, P, P, P],
    'b2m_seg_rank': [P, I64, P, P, I64, P, P],
    'b2m_seg_mode': [P, P, I64, I64, I32, P, P, P],
}
PLAIN = {'b2m_last_error': (C.c_char_p, []), 'b2m_version': (C.c_int, []), 'b2m_device_ok': (C.c_int, []),
         'b2m_reload_env': (C.c_int, []),
         'b2m_weight_pack_size': (C.c_int64, [I32, I32, I32]),
         'b2m_conv_wgrad_workspace': (C.c_int64, [I32, I32, I32]),
         'b2m_weight_pack_h_size': (C.c_int64, [I32, I32, I32, I32]),
         'b2m_unique_insert': (C.c_int64, [P, I64, P, I64, P, P, P, P]),
         'b2m_weight_pack_plan_size': (C.c_int32, []),
         'b2m_weight_pack_h_plan_size': (C.c_int32, []),
         'b2m_weight_pack_h_plan': (C.c_int64, [I32, P, P, P, P, P, P, P, P, P, P, P]),
         'b2m_rulebook_cnt_size': (C.c_int64, [I32, I64]),
         'b2m_radix_argsort_scratch': (C.c_int64, [I64]),
         'b2m_xchg_size': (C.c_int64, []), 'b2m_xchg_max_doubles': (C.c_int32, []), 'b2m_xchg_max_ranks': (C.c_int32, []),
         'b2m_xchg_alloc': (C.c_int, [C.POINTER(C.c_void_p), P]), 'b2m_xchg_open': (C.c_int, [P, C.POINTER(C.c_void_p)]),
         'b2m_xchg_close': (C.c_int, [P]), 'b2m_xchg_free': (C.c_int, [P]),
         'b2m_xchg_is_finegrained': (C.c_int32, [P]),
         'b2m_weight_pack_plan': (C.c_int64, [I32, P, P, P, P, P, P, P, P, P, P, P])}

_lib = None


class B2MError(RuntimeError):
    pass


def load():
    """Load libb2m_hip.so (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            'box2mask_amd: the HIP extension %s is missing. Build it with '
            '`python -m box2mask_amd.build` (hipcc --offload-arch=gfx950); there is no CPU fallback.' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, args in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    for name, (res, args) in PLAIN.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = res
    _lib = lib
    return lib


def ptr(t):
    """Device (or host) address of a tensor, or None."""
    if t is None:
        return None
    return t.data_ptr()


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_raw_device = getattr(torch._C, '_cuda_getDevice', None)


def stream():
    """Handle of torch's current HIP stream on the current device.  (torch.cuda.current_stream() builds a Stream object per
    call: 9 us of the 18 us a launch through this module cost, on ~1300 launches per training step and ~210 per batch-size-1
    inference pass; the raw getter is what torch's own extensions use.)"""
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


# optional observer used by bench.py to bracket launches with HIP events: hook(name, args, meta) -> finish()
_hook = None


def set_hook(fn):
    global _hook
    _hook = fn


def call(name, *args, meta=None):
    """Invoke a b2m_* entry on the current stream; raise B2MError on a negative return.  `meta`: facts about the call the
    raw arguments do not carry (the logical channel count of a padded input), for the observer only."""
    lib = _lib if _lib is not None else load()
    done = _hook(name, args, meta) if _hook is not None else None
    rc = getattr(lib, name)(*args, stream())
    if done is not None:
        done()
    if rc != 0:
        raise B2MError('%s failed (%d): %s' % (name, rc, lib.b2m_last_error().decode()))


def call_ret(name, *args):
    """An entry of PLAIN that takes the stream and returns a count (>= 0) or a negative error code."""
    lib = load()
    done = _hook(name, args, None) if _hook is not None else None
    rc = getattr(lib, name)(*args, stream())
    if done is not None:
        done()
    if rc < 0:
        raise B2MError('%s failed (%d): %s' % (name, rc, lib.b2m_last_error().decode()))
    return rc


env_epoch = [0]      # advanced by reload_env: whatever was built under the old switches (packed weight images: their strip width is a
                     # switch) is rebuilt on its next use


def reload_env():
    """The library caches the B2M_* switches per process: call this after changing one in os.environ."""
    load().b2m_reload_env()
    env_epoch[0] += 1


def require_gpu():
    if not torch.cuda.is_available():
        raise B2MError('box2mask_amd needs a ROCm GPU (MI355X / gfx950); there is no CPU path in the product.')
    load()
