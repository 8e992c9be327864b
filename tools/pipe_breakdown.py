"""Launch time of the pipelined forward kernel with single components removed (diagnostic builds, wrong results):
what do the strip flush, the gathers and the weight loads cost on the level-0 96->96 layer?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from box2mask_amd import synth, functional as F_, _lib
from box2mask_amd.sparse import CoordinateManager
b = synth.make_batch(int(os.environ.get('BS', '4')), seed0=0)
m = CoordinateManager(b['vox_coords'], reorder=True)
rb = m.rulebook_same(0, 3)
x = torch.randn(rb.n_in, 96, device='cuda'); w = torch.randn(27, 96, 96, device='cuda') * 0.05
wp = F_.weight_pack(w)
fl = 2.0 * rb.pairs * 96 * 96
def t(n=5):
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    F_.conv_raw(x, None, wp, 27, None, rb, rb.n_out, 96); s.record()
    for _ in range(n): F_.conv_raw(x, None, wp, 27, None, rb, rb.n_out, 96)
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n
for rnd in range(2):
    for name, env in [('old kernel', {'B2M_CONV_PIPE': '0'}), ('flow', {}),
                      ('flow -flush', {'B2M_PIPE_DBG': '1'}), ('flow -gathers', {'B2M_PIPE_DBG': '2'}),
                      ('flow -weights', {'B2M_PIPE_DBG': '4'}), ('flow -gathers -weights', {'B2M_PIPE_DBG': '6'}),
                      ('flow compiler loads', {'B2M_CONV_HANDLOADS': '0'}),
                      ('flow compiler loads -MFMAs', {'B2M_CONV_HANDLOADS': '0', 'B2M_PIPE_DBG': '8'}),
                      ('flow -MFMAs (loads kept)', {'B2M_PIPE_DBG': '8'})]:
        for k in ('B2M_CONV_PIPE', 'B2M_PIPE_DBG', 'B2M_CONV_HANDLOADS'): os.environ.pop(k, None)
        os.environ.update(env)
        _lib.reload_env()
        ms = t()
        print('%-26s %.3f ms  %.1f TFLOP/s' % (name, ms, fl / ms / 1e9))
# the half inference kernel on the same map (a sixteenth of the MFMA cycles, half the bytes)
for k in ('B2M_CONV_PIPE', 'B2M_PIPE_DBG', 'B2M_CONV_HANDLOADS'): os.environ.pop(k, None)
_lib.reload_env()
xh = x.half()
def th(n=5):
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    F_.conv_affine_h(xh, None, w, rb, rb.n_out); s.record()
    for _ in range(n): F_.conv_affine_h(xh, None, w, rb, rb.n_out)
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n
for rnd in range(2):
    ms = th()
    print('%-26s %.3f ms  %.1f TFLOP/s' % ('half kernel', ms, fl / ms / 1e9))
