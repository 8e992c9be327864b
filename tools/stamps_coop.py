"""Diagnostic: what the lock of conv_fwd_coop_kernel costs (s_memtime stamps; build tools/micro/libb2m_stamps.so with
`python tools/stamps.py build`, run with B2M_LIB_PATH=tools/micro/libb2m_stamps.so B2M_CONV_COOP=1)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from box2mask_amd import _lib, synth, functional as F_
from box2mask_amd.sparse import CoordinateManager
lib = C.CDLL(os.environ['B2M_LIB_PATH'])
lib.b2m_debug_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
b = synth.make_batch(int(os.environ.get('BS', '8')), seed0=0)
m = CoordinateManager(b['vox_coords'], reorder=True)
rb0 = m.rulebook_same(0, 3); m.ensure_level(2); rb1 = m.rulebook_same(1, 3)
for name, rb, c1, co in [('L0 96->96', rb0, 96, 96), ('L1 96->96', rb1, 96, 96), ('L0 32->32', rb0, 32, 32), ('L1 128->128', rb1, 128, 128)]:
    x1 = torch.randn(rb.n_in, c1, device='cuda')
    wp = F_.weight_pack(torch.randn(27, c1, co, device='cuda') * 0.05)
    for it in range(2):
        torch.cuda.synchronize(); lib.b2m_debug_stamps(None, 1)
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record(); F_.conv_raw(x1, None, wp, 27, None, rb, rb.n_out, co); e.record(); torch.cuda.synchronize()
    v = (C.c_ulonglong * 12)(); lib.b2m_debug_stamps(v, 0)
    wait, hold, fail, life, units, waves, pro, endw = [float(x) for x in v][:8]
    print('%-12s %.3f ms waves %d units/wave %.1f | per unit: wait for the lock %.0f cycles (%.2f failed attempts), hold %.0f | per wave: lifetime %.0f, prologue to first barrier %.0f, wait at the last barrier %.0f'
          % (name, s.elapsed_time(e), waves, units / waves, wait / units, fail / units, hold / units, life / waves, pro / waves, endw / waves))
