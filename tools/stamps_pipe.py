"""s_memtime stamps of conv_fwd_pipe_kernel (build: python tools/stamps.py build; run with B2M_LIB_PATH=tools/micro/libb2m_stamps.so)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from box2mask_amd import synth, functional as F_
from box2mask_amd.sparse import CoordinateManager
lib = C.CDLL(os.environ['B2M_LIB_PATH'])
lib.b2m_debug_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
b = synth.make_batch(int(os.environ.get('BS', '4')), seed0=0)
m = CoordinateManager(b['vox_coords'], reorder=True)
rb0 = m.rulebook_same(0, 3); m.ensure_level(2); rb1 = m.rulebook_same(1, 3)
for name, rb, c1, c2, co in [('L0 k3 96->96', rb0, 96, 0, 96), ('L0 k3 128(96|32)->96', rb0, 96, 32, 96), ('L0 k3 96->32', rb0, 96, 0, 32), ('L1 k3 96->96', rb1, 96, 0, 96)]:
    x1 = torch.randn(rb.n_in, c1, device='cuda'); x2 = torch.randn(rb.n_in, c2, device='cuda') if c2 else None
    wp = F_.weight_pack(torch.randn(27, c1 + c2, co, device='cuda') * 0.05)
    for it in range(2):
        torch.cuda.synchronize(); lib.b2m_debug_stamps(None, 1)
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record(); F_.conv_raw(x1, x2, wp, 27, None, rb, rb.n_out, co); e.record(); torch.cuda.synchronize()
    v = (C.c_ulonglong * 12)(); lib.b2m_debug_stamps(v, 0)
    mf, iss, fl, life, offs, waves, wout, steps, setup, adv = [float(x) for x in v][:10]
    print('%-22s %.3f ms waves %d offsets/wave %.1f steps/offset %.1f | per step: mfma block %.0f  load issue %.0f | per offset: flush %.0f  advance %.0f | per wave: lifetime %.0f  setup %.0f  write-out %.0f  loop %.0f'
          % (name, s.elapsed_time(e), waves, offs / waves, steps / offs, mf / steps, iss / steps, fl / offs, adv / offs, life / waves, setup / waves, wout / waves, (mf + iss + fl + adv) / waves))
