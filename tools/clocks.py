"""Diagnostic: the shader clock the forward / data-gradient kernel and the weight-gradient kernel run at -- each alone on
the chip, and side by side on two streams as in the backward pass (build with -DB2M_CLOCKS: two stamps per wave).

    python tools/clocks.py build        # -> tools/micro/libb2m_clocks.so (cross-compiles without a GPU)
    B2M_LIB_PATH=tools/micro/libb2m_clocks.so python tools/clocks.py run

Answers "what saturates when conv_fwd and conv_wgrad share the chip"  as far as a one-GPU box without
co-running PMC collection can: rocprofv3 serialises dispatches under --pmc, so overlapped launches cannot be counted; the
clock and the per-kernel wave time can."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, 'tools', 'micro', 'libb2m_clocks.so')


def build():
    from box2mask_amd import build as B
    objs = []
    for s in B.SOURCES:
        o = os.path.join(ROOT, 'tools', 'micro', 'clocks_' + s.replace('.hip', '.o'))
        subprocess.check_call([B.HIPCC] + B.FLAGS + ['-DB2M_CLOCKS', '-I', os.path.join(ROOT, 'include'), '-c',
                                                    os.path.join(B.CSRC, s), '-o', o])
        objs.append(o)
    subprocess.check_call([B.HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', OUT])
    print(OUT)


def run():
    import torch
    from box2mask_amd import synth, functional as F_
    from box2mask_amd.sparse import CoordinateManager
    lib = C.CDLL(os.environ['B2M_LIB_PATH'])
    lib.b2m_debug_clocks.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    b = synth.make_batch(int(os.environ.get('BS', '8')), seed0=0)
    m = CoordinateManager(b['vox_coords'], reorder=True)
    rb0 = m.rulebook_same(0, 3); m.ensure_level(2); rb1 = m.rulebook_same(1, 3)
    side = torch.cuda.Stream()
    for name, rb, c in (('L0 k3 96->96', rb0, 96), ('L1 k3 96->96', rb1, 96), ('L1 k3 128->128', rb1, 128)):
        x = torch.randn(rb.n_in, c, device='cuda'); dy = torch.randn(rb.n_out, c, device='cuda')
        w = torch.randn(27, c, c, device='cuda') * 0.05
        wp = F_.weight_pack(w); dw = torch.zeros_like(w)
        fl = 2.0 * rb.pairs * c * c
        f_fwd = lambda: F_.conv_raw(x, None, wp, 27, None, rb, rb.n_out, c)
        f_wg = lambda: F_.wgrad_raw(x, dy, rb, 27, dw, 0)

        def both():
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                f_wg()
            f_fwd()
            torch.cuda.current_stream().wait_stream(side)
        for label, fn, nfl in (('forward alone', f_fwd, 1), ('weight gradient alone', f_wg, 1), ('both, two streams', both, 2)):
            for _ in range(3):
                fn()
            torch.cuda.synchronize(); lib.b2m_debug_clocks(None, 1)
            n = 8
            s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(n):
                fn()
            e.record(); torch.cuda.synchronize()
            ms = s.elapsed_time(e) / n
            v = (C.c_ulonglong * 8)(); lib.b2m_debug_clocks(v, 0)
            v = [float(t) for t in v]
            def ghz(c_, r_): return c_ / max(r_, 1) * 0.1
            line = '%-16s %-22s %.3f ms %6.1f TFLOP/s' % (name, label, ms, nfl * fl / ms / 1e9)
            if v[2]:
                line += ' | fwd: clock %.3f GHz, wave life %.0f us x %d waves' % (ghz(v[0], v[1]), v[1] / v[2] / 100, v[2] / n)
            if v[5]:
                line += ' | wgrad: clock %.3f GHz, wave life %.0f us x %d waves' % (ghz(v[3], v[4]), v[4] / v[5] / 100, v[5] / n)
            print(line)


if __name__ == '__main__':
    build() if sys.argv[1:] == ['build'] else run()
