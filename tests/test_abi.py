"""The C-ABI shared library loads and exports every symbol include/*.h declares, and the ctypes
prototype table matches the header (argument counts and kinds).  No compute calls (no GPU here)."""
import ctypes
import os
import re

import pytest

from box2mask_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_decls():
    inc = os.path.join(ROOT, 'include')
    src = ''.join(open(os.path.join(inc, f)).read() for f in sorted(os.listdir(inc)) if f.endswith('.h'))
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    decls = {}
    for m in re.finditer(r'\b(int64_t|int32_t|int|const char\*)\s+(b2m_\w+)\s*\(([^;]*?)\)\s*;', src, flags=re.S):
        args = [a.strip() for a in m.group(3).replace('\n', ' ').split(',')]
        if args == ['void']:
            args = []
        decls[m.group(2)] = args
    return decls


def _kind(arg):
    if '*' in arg:
        return 'p'
    t = arg.split()[0] if not arg.startswith('const') else arg.split()[1]
    return {'int64_t': 'i64', 'uint64_t': 'u64', 'int32_t': 'i32', 'float': 'f32', 'double': 'f64', 'int': 'i32'}[t]


def _ckind(t):
    if t in (_lib.P,) or (isinstance(t, type) and issubclass(t, ctypes._Pointer)):
        return 'p'
    return {_lib.I64: 'i64', ctypes.c_uint64: 'u64', _lib.I32: 'i32', _lib.F32: 'f32', _lib.F64: 'f64'}[t]


def test_library_is_built_and_loads():
    # (a library OLDER than its sources is rebuilt here, and a source that no longer compiles fails HERE: round 6 ran two hours of
    # GPU leases on a library one edit behind a source that assembled to text but not to an object)
    from box2mask_amd import build
    if not os.path.exists(_lib.LIB_PATH) or (_lib.LIB_PATH == build.LIB and build._stale()):
        build.build(verbose=False)
    lib = _lib.load()
    assert lib.b2m_version() >= 1


def test_every_header_symbol_is_exported_and_bound():
    decls = _header_decls()
    assert len(decls) >= 25
    lib = _lib.load()
    for name, args in decls.items():
        assert hasattr(lib, name), 'missing export: ' + name
        if name in _lib.PLAIN:
            continue
        assert name in _lib.PROTOTYPES, 'no ctypes prototype for ' + name
        proto = _lib.PROTOTYPES[name]
        assert len(proto) == len(args), (name, len(proto), len(args))
        assert [_ckind(t) for t in proto] == [_kind(a) for a in args], name
    for name in _lib.PROTOTYPES:
        assert name in decls, 'prototype without header declaration: ' + name


def test_product_has_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from box2mask_amd import iou_nms
    with pytest.raises(_lib.B2MError):
        iou_nms.NMS_clustering(torch.rand(4, 7).abs(), 0.5)
    from box2mask_amd.model import Model
    from box2mask_amd.config import scannet_config
    from box2mask_amd import synth
    with pytest.raises(_lib.B2MError):
        Model(scannet_config(), *synth.scannet_tables())


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, 'box2mask_amd')
    for f in os.listdir(pkg):
        if f.endswith('.py'):
            txt = open(os.path.join(pkg, f)).read()
            assert 'import oracle' not in txt and 'from oracle' not in txt, f


def test_round6_entries_check_their_arguments_on_the_host():
    """The entries added in round 6 validate their arguments before anything touches a device (negative return + b2m_last_error, the
    convention of include/b2m.h) -- and b2m_weight_pack_h_plan, which is host code altogether, lays a plan out as documented."""
    import ctypes as C
    import numpy as np
    lib = _lib.load()
    err = lambda: lib.b2m_last_error().decode()
    # the plan of two images of a (27, 96, 128) layer: forward image (96 | 0 input channels) and the transposed image of channels 32..95
    n = 2
    i64 = lambda v: np.ascontiguousarray(v, dtype=np.int64)
    i32 = lambda v: np.ascontiguousarray(v, dtype=np.int32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    cols = [i64([4096, 4096]), i64([8192, 16384]), i32([27, 27]), i32([96, 96]), i32([128, 128]), i32([96, 0]), i32([0, 1]), i32([0, 1]),
            i32([0, 32]), i32([0, 64])]
    size = lib.b2m_weight_pack_h_plan_size()
    assert 64 <= size <= 128
    host = np.zeros(n * size, np.uint8)
    blocks = lib.b2m_weight_pack_h_plan(n, *[p(c) for c in cols], p(host))
    total0 = lib.b2m_weight_pack_h_size(27, 96, 0, 128)
    total1 = lib.b2m_weight_pack_h_size(27, 128, 0, 64)
    assert total0 == 27 * 96 * 128 and total1 == 27 * 128 * 64            # (complete strips and chunks: no padding)
    assert blocks == -(-total0 // 2048) + -(-total1 // 2048)
    bad = [c.copy() for c in cols]
    bad[8][1] = 64                                                          # slice 64 .. 127 of 96 input channels
    assert lib.b2m_weight_pack_h_plan(n, *[p(c) for c in bad], p(host)) < 0 and 'slice' in err()
    bad = [c.copy() for c in cols]
    bad[5][0] = 40                                                          # 40 | 56 input channels: not multiples of 16
    assert lib.b2m_weight_pack_h_plan(n, *[p(c) for c in bad], p(host)) < 0 and '16' in err()
    assert lib.b2m_weight_pack_h_run(None, 0, 0, None) == 0                 # nothing to do: no launch, no device needed
    # half weight gradient: 2-byte aligned operands
    a = [4097, 96, 96, 1000, 8192, 96, 96, 64, 64, 64, 1000, 27, 4096, 96, 96 * 96, 0, 1.0, None]
    assert lib.b2m_conv_wgrad_h(*a) < 0 and 'aligned' in err()
    # F16 convolution with tile sums: the sums' buffer is not optional
    a = [4096, 96, 96, None, 0, 0, 1000, 8192, 27, 64, 64, 64, 1000, 16384, 96, 96, None, None]
    assert lib.b2m_conv_fwd_h_stats(*a) < 0 and 'tile_stats' in err()
    # half BatchNorm backward: ReLU needs the stored output or the forward's scale / shift
    a = [4096, 96, None, 0, 8192, 96, 1000, 96, 64, 64, 1, None, None, 64, 64, None, None, 1.0, None]
    assert lib.b2m_bn_bwd_reduce_h(*a) < 0 and 'relu' in err()
    a = [4096, 96, None, 0, 8192, 96, 1000, 96, 64, 64, None, 64, 1000.0, None, 1, None, None, 16384, 96, None, 0, None]
    assert lib.b2m_bn_bwd_apply_h(*a) < 0 and 'relu' in err()
