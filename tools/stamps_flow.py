"""Diagnostic: where a conv_fwd_flow_kernel wave (un-split maps, hand-issued loads) spends its cycles -- s_memtime stamps around
the phases of the walk, twelve waves per CU as in the product.  Build tools/micro/libb2m_stamps.so with
`python tools/stamps.py build`; run with B2M_LIB_PATH=tools/micro/libb2m_stamps.so."""
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
    ts = []
    for it in range(2):
        torch.cuda.synchronize(); lib.b2m_debug_stamps(None, 1)
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record(); F_.conv_raw(x1, None, wp, 27, None, rb, rb.n_out, co, tile_stats=ts); e.record(); torch.cuda.synchronize()
    v = (C.c_ulonglong * 12)(); lib.b2m_debug_stamps(v, 0)
    pro, loop, flush, adv, tail, life, vis, waves, stat, init = [float(x) for x in v][:10]
    print('%-12s %.3f ms waves %d visits/wave %.1f | per wave: lifetime %.0f = prologue %.0f (strip init + counts %.0f) + loads+MFMA %.0f + flush %.0f + advance %.0f + epilogue %.0f (column sums %.0f) | per visit: loop %.0f flush %.0f advance %.0f'
          % (name, s.elapsed_time(e), waves, vis / waves, life / waves, pro / waves, init / waves, loop / waves, flush / waves, adv / waves, tail / waves, stat / waves,
             loop / vis, flush / vis, adv / vis))
