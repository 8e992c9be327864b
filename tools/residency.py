"""Diagnostic: waves resident over time inside ONE conv_fwd_flow launch, per layer shape of the benchmark's batch (VERDICT r05,
item 3: "measure residency, then split only the tail").  Every wave writes its begin / end time (s_memrealtime, 10 ns) into
its own slot (build with -DB2M_RESIDENCY: two stamps per wave); the host turns the intervals into a timeline.

    python tools/residency.py build        # -> tools/micro/libb2m_residency.so (cross-compiles without a GPU)
    B2M_LIB_PATH=tools/micro/libb2m_residency.so python tools/residency.py run

Per shape: launch span (first begin -> last end), waves, median / p90 / max wave lifetime, when the LAST wave began, the mean
number of resident waves in each twentieth of the span (as a share of the peak residency of the launch), and what the span
would be if the work (sum of wave lifetimes) ran at the peak residency throughout -- the tail + ramp is the difference."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, 'tools', 'micro', 'libb2m_residency.so')


def build():
    from box2mask_amd import build as B
    objs = []
    procs = []
    for s in B.SOURCES:
        o = os.path.join(ROOT, 'tools', 'micro', 'residency_' + s.replace('.hip', '.o'))
        procs.append(subprocess.Popen([B.HIPCC] + B.FLAGS + ['-DB2M_RESIDENCY', '-I', os.path.join(ROOT, 'include'), '-c',
                                                             os.path.join(B.CSRC, s), '-o', o]))
        objs.append(o)
    for p in procs:
        assert p.wait() == 0
    subprocess.check_call([B.HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', OUT])
    print(OUT)


def timeline(t0, t1, bins=20):
    import numpy as np
    lo, hi = t0.min(), t1.max()
    span = float(hi - lo)
    edges = lo + span * np.arange(bins + 1) / bins
    res = np.zeros(bins)
    for b in range(bins):           # mean residency in the bin = overlap of every interval with the bin / bin width
        ov = np.clip(np.minimum(t1, edges[b + 1]) - np.maximum(t0, edges[b]), 0, None)
        res[b] = ov.sum() / (span / bins)
    return span, res


def run():
    import torch
    from box2mask_amd import _lib, synth, functional as F_
    from box2mask_amd.sparse import CoordinateManager
    lib = C.CDLL(os.environ['B2M_LIB_PATH'])
    lib.b2m_debug_residency.argtypes = [C.c_void_p, C.c_int]
    cap = lib.b2m_debug_residency(None, 1)
    assert cap > 0
    import numpy as np
    host = np.zeros(3 * cap, dtype=np.uint64)
    b = synth.make_batch(int(os.environ.get('BS', '8')), seed0=0)
    m = CoordinateManager(b['vox_coords'], reorder=True)
    m.ensure_level(4)
    cases = [('L0 k3 96->96', 0, 96, 96), ('L0 k3 32->32', 0, 32, 32), ('L1 k3 96->96', 1, 96, 96), ('L1 k3 32->32', 1, 32, 32),
             ('L1 k3 128->96', 1, 128, 96), ('L2 k3 128->128', 2, 128, 128), ('L2 k3 64->64', 2, 64, 64), ('L3 k3 256->256', 3, 256, 256),
             ('L3 k3 128->128', 3, 128, 128), ('L4 k3 256->256', 4, 256, 256)]
    for name, level, cin, cout in cases:
        rb = m.rulebook_same(level, 3)
        x = torch.randn(rb.n_in, cin, device='cuda')
        wp = F_.weight_pack(torch.randn(27, cin, cout, device='cuda') * 0.05)
        ms = []
        for it in range(3):
            torch.cuda.synchronize()
            assert lib.b2m_debug_residency(None, 1) == cap
            s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
            s.record(); F_.conv_raw(x, None, wp, 27, None, rb, rb.n_out, cout); e.record(); torch.cuda.synchronize()
            ms.append(s.elapsed_time(e))
        assert lib.b2m_debug_residency(host.ctypes.data_as(C.c_void_p), 0) == cap
        v = host.reshape(-1, 3)
        v = v[v[:, 1] > 0]
        t0, t1 = v[:, 0].astype(np.float64) * 10.0, v[:, 1].astype(np.float64) * 10.0        # ns
        life = (t1 - t0) / 1e3
        span, res = timeline(t0, t1)
        peak = res.max()
        ideal = life.sum() * 1e3 / peak                      # ns: the same wave-time at the peak residency throughout
        order = np.argsort(t0)
        last_begin = (t0.max() - t0.min()) / span
        flops = 2.0 * rb.pairs * cin * cout
        print('%-16s rows %7d  %.3f ms (events; %.1f TFLOP/s)  span %.3f ms  waves %6d  lifetime us: median %.0f p90 %.0f max %.0f | '
              'last wave begins at %.2f of the span | peak residency %.0f waves; span at peak residency throughout %.3f ms '
              '(ramp + tail = %.0f %% of the launch)'
              % (name, rb.n_out, min(ms), flops / min(ms) / 1e9, span / 1e6, len(v), np.median(life), np.percentile(life, 90), life.max(),
                 last_begin, peak, ideal / 1e6, 100.0 * (1.0 - ideal / span)))
        print('      resident / peak per twentieth: ' + ' '.join('%3.0f' % (100.0 * r / peak) for r in res))
        # the tail in detail: residency in the last 15 % of the span, and the lifetimes of the waves that END last
        tail = np.argsort(t1)[-8:]
        print('      the last 8 waves to end: began at %s of the span, lived %s us'
              % (' '.join('%.2f' % ((t0[i] - t0.min()) / span) for i in tail), ' '.join('%.0f' % life[i] for i in tail)))


if __name__ == '__main__':
    build() if sys.argv[1:] == ['build'] else run()
