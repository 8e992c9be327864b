"""Build recipe for the HIP extension (one shared library with a C ABI, include/b2m.h).

`python -m box2mask_amd.build` cross-compiles for gfx950 with hipcc; no GPU is needed to build.
The library is built in-tree (box2mask_amd/libb2m_hip.so) so that it travels with the repository
snapshot to the GPU box and shows up as loaded native code.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libb2m_hip.so')
SOURCES = ['coords.hip', 'conv.hip', 'norm.hip', 'nms.hip', 'voxelize.hip', 'xchg.hip']
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-munsafe-fp-atomics', '-Wall',
         '-Wno-unused-function']


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    inc = os.path.join(HERE, '..', 'include')
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if not f.endswith('.o')] + [os.path.join(inc, f) for f in os.listdir(inc)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not _stale():
        return LIB
    objs = []
    procs = []
    for s in SOURCES:
        o = os.path.join(CSRC, s.replace('.hip', '.o'))
        objs.append(o)
        cmd = [HIPCC] + FLAGS + ['-c', os.path.join(CSRC, s), '-o', o]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = None
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0 and failed is None:
            failed = (s, out.decode())
        elif verbose and out.strip():
            print(out.decode())
    if failed is not None:
        # the library of an OLDER source must not survive a failed build: whatever runs next would run it in silence (round 6:
        # two hours of GPU leases measured a library one edit behind its source)
        if os.path.exists(LIB):
            os.remove(LIB)
        raise RuntimeError('hipcc failed on %s (the stale %s was removed):\n%s' % (failed[0], os.path.basename(LIB), failed[1]))
    cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', LIB]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIB)
