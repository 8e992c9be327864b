"""Diagnostic: where a conv_fwd wave spends its cycles (s_memtime stamps, build with -DB2M_STAMPS).

    python tools/stamps.py build        # -> tools/micro/libb2m_stamps.so (cross-compiles without a GPU)
    B2M_LIB_PATH=tools/micro/libb2m_stamps.so python tools/stamps.py run

Prints, per case, cycles per active offset in the three phases of the offset walk (pair-list fetch, load+MFMA
loop, LDS flush), the strip write-out, and the wave lifetime.  Stamps cost about 10 % themselves."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, 'tools', 'micro', 'libb2m_stamps.so')


def build():
    from box2mask_amd import build as B
    objs = []
    for s in B.SOURCES:
        o = os.path.join(ROOT, 'tools', 'micro', 'stamps_' + s.replace('.hip', '.o'))
        subprocess.check_call([B.HIPCC] + B.FLAGS + ['-DB2M_STAMPS', '-I', os.path.join(ROOT, 'include'), '-c',
                                                    os.path.join(B.CSRC, s), '-o', o])
        objs.append(o)
    subprocess.check_call([B.HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', OUT])
    print(OUT)


def run():
    import torch
    from box2mask_amd import _lib, synth, functional as F_
    from box2mask_amd.sparse import CoordinateManager
    lib = C.CDLL(os.environ['B2M_LIB_PATH'])
    lib.b2m_debug_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    b = synth.make_batch(int(os.environ.get('BS', '4')), seed0=0)
    m = CoordinateManager(b['vox_coords'])
    rb0 = m.rulebook_same(0, 3); m.ensure_level(2); rb1 = m.rulebook_same(1, 3)
    cases = [('L0 k3 96->96', rb0, 27, 96, 0, 96), ('L0 k3 128(96|32)->96', rb0, 27, 96, 32, 96),
             ('L0 k3 32->32', rb0, 27, 32, 0, 32), ('L1 k3 128->128', rb1, 27, 128, 0, 128),
             ('L0 1x1 128->96', None, 1, 128, 0, 96)]
    for name, rb, K, c1, c2, co in cases:
        n_out = rb.n_out if rb is not None else m.n(0)
        n_in = rb.n_in if rb is not None else m.n(0)
        x1 = torch.randn(n_in, c1, device='cuda'); x2 = torch.randn(n_in, c2, device='cuda') if c2 else None
        wp = F_.weight_pack(torch.randn(K, c1 + c2, co, device='cuda') * 0.05)
        for it in range(2):
            torch.cuda.synchronize(); lib.b2m_debug_stamps(None, 1)
            s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
            s.record(); F_.conv_raw(x1, x2, wp, K, None, rb, n_out, co); e.record(); torch.cuda.synchronize()
        v = (C.c_ulonglong * 12)(); lib.b2m_debug_stamps(v, 0)
        idx, loop, epi, life, noff, waves, tail, groups, ini, cnt = [float(x) for x in v][:10]
        print('%-22s %.3f ms  waves %d  offsets/wave %.1f  groups/offset %.2f | cycles per offset: pairs %.0f  '
              'loads+mfma %.0f  flush %.0f | per wave: lifetime %.0f  walk %.0f  write-out %.0f  other %.0f (strip init %.0f, counts %.0f)'
              % (name, s.elapsed_time(e), waves, noff / waves, groups / max(noff, 1), idx / noff, loop / noff,
                 epi / noff, life / waves, (idx + loop + epi) / waves, tail / waves,
                 (life - tail - idx - loop - epi) / waves, ini / waves, cnt / waves))


if __name__ == '__main__':
    build() if sys.argv[1:] == ['build'] else run()
