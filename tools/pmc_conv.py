"""One L0 conv forward for PMC collection (rocprofv3 --pmc ...)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from box2mask_amd import synth, functional as F_
from box2mask_amd.sparse import CoordinateManager
b = synth.make_batch(4, seed0=0)
m = CoordinateManager(b['vox_coords'], reorder=True)      # Morton rows, as the network runs them
rb = m.rulebook_same(0, 3)
x = torch.randn(rb.n_in, 96, device='cuda'); w = torch.randn(27, 96, 96, device='cuda') * 0.05
for _ in range(3):
    F_.conv_raw(x, None, F_.weight_pack(w), 27, None, rb, rb.n_out, 96)
dy = torch.randn(rb.n_out, 96, device='cuda'); dw = torch.zeros_like(w)
for _ in range(2):
    F_.wgrad_raw(x, dy, rb, 27, dw, 0)
if os.environ.get('HALF'):          # ... and the half-precision inference kernel on the same map (b2m_conv_fwd_h)
    xh = x.half()
    for _ in range(3):
        F_.conv_affine_h(xh, None, w, rb, rb.n_out)
    # ... the half weight gradient on the f16 MFMA (conv_wgrad_trh_kernel) and the 64-column F16 convolution (128 -> 128)
    from box2mask_amd import half_train as HT
    dyh = dy.half(); dwh = torch.zeros_like(w)
    for _ in range(2):
        HT._wgrad_h(xh, dyh, rb, 27, dwh, 0, 1.0)
    x128 = torch.randn(rb.n_in, 128, device='cuda').half(); w128 = torch.randn(27, 128, 128, device='cuda') * 0.05
    for _ in range(2):
        F_.conv_affine_h(x128, None, w128, rb, rb.n_out)
torch.cuda.synchronize()
