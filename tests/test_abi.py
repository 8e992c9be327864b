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
    if not os.path.exists(_lib.LIB_PATH):
        from box2mask_amd import build
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
