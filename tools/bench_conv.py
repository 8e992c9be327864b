"""Micro-benchmark of the sparse-conv kernels on one synthetic scene batch, with interleaved A/B rounds of
tuning switches inside one process (same device, same clocks).

    python tools/bench_conv.py                       # base vs the variants below
    VARIANTS="tw2:B2M_CONV_TW3=0;old:B2M_CONV_HANDLOADS=0" python tools/bench_conv.py

Switches are re-read by the library per variant (b2m_reload_env): B2M_CONV_TW3 (48-column strips), B2M_CONV_HANDLOADS
(hand-issued loads), B2M_CONV_FAST32 / B2M_WGRAD_FAST32 (24-bit multiply addressing), ...."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from box2mask_amd import synth, functional as F_, _lib
from box2mask_amd.sparse import CoordinateManager

bs = int(os.environ.get('BS', '4'))
b = synth.make_batch(bs, seed0=0)
m = CoordinateManager(b['vox_coords'], reorder=True)      # Morton rows, as the network runs them
rb0 = m.rulebook_same(0, 3); m.ensure_level(7); rb1 = m.rulebook_same(1, 3); rbu = m.rulebook_up(0); rb5 = m.rulebook_same(0, 5)
rb2 = m.rulebook_same(2, 3); rb3 = m.rulebook_same(3, 3); rb4 = m.rulebook_same(4, 3)
rb5l = m.rulebook_same(5, 3); rb6 = m.rulebook_same(6, 3)
rbu1 = m.rulebook_up(1); rbu2 = m.rulebook_up(2); rbu3 = m.rulebook_up(3)
torch.cuda.synchronize()
print('N0', m.n(0), 'pairs k3', rb0.pairs, 'k5', rb5.pairs)

def timeit(fn, n=4):
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

CORUN = os.environ.get('CORUN', '0') == '1'
side = torch.cuda.Stream()
VARIANTS = [('base', {})]
for spec in os.environ.get('VARIANTS', 'tw2:B2M_CONV_TW3=0;slow64:B2M_WGRAD_FAST32=0').split(';'):
    if spec:
        name, kvs = spec.split(':')
        VARIANTS.append((name, dict(kv.split('=') for kv in kvs.split(','))))
SWITCHES = ('B2M_CONV_CHUNK_ITEMS', 'B2M_CONV_SPLIT2', 'B2M_WGRAD_UP', 'B2M_WGRAD_LDS', 'B2M_CONV_CHAIN', 'B2M_WGRAD_HANDLOADS', 'B2M_CONV_HANDLOADS', 'B2M_PIPE_DBG', 'B2M_CONV_UP', 'B2M_CONV_UP_MIN_ITEMS', 'B2M_CONV_STEM', 'B2M_WGRAD_NARROW', 'B2M_WGRAD_PIPE_IDENT', 'B2M_WGRAD_KPACK', 'B2M_CONV_1X1', 'B2M_XCD_ORDER', 'B2M_XCD_BALANCE', 'B2M_CONV_FLOW_SPLIT', 'B2M_CONV_PIPE', 'B2M_CONV_TW3', 'B2M_CONV_FAST32', 'B2M_WGRAD_FAST32', 'B2M_XCD', 'B2M_WGRAD_PIPE', 'B2M_CONV_WGCOMBINE', 'B2M_CONV_CHUNKSPLIT', 'B2M_WGRAD_MIN_TILES', 'B2M_CONV_TARGET', 'B2M_CONV_MAXSLICE')
cases = [('L0 k3 96->96', rb0, 27, 96, 0, 96), ('L0 k3 128(96|32)->96', rb0, 27, 96, 32, 96), ('L0 k3 32->32', rb0, 27, 32, 0, 32),
         ('L1 k3 96->96', rb1, 27, 96, 0, 96), ('L1 k3 32->32', rb1, 27, 32, 0, 32), ('L0 up 96->96', rbu, 8, 96, 0, 96), ('L0 k5 8->32', rb5, 125, 8, 0, 32),
         ('L0 1x1 128->96', None, 1, 128, 0, 96), ('L1 k3 128->128', rb1, 27, 128, 0, 128), ('L1 k3 64->64', rb1, 27, 64, 0, 64),
         ('L2 k3 128->128', rb2, 27, 128, 0, 128), ('L2 k3 64->64', rb2, 27, 64, 0, 64), ('L3 k3 256->256', rb3, 27, 256, 0, 256),
         ('L3 k3 128->128', rb3, 27, 128, 0, 128), ('L4 k3 256->256', rb4, 27, 256, 0, 256),
         ('L5 k3 256->256', rb5l, 27, 256, 0, 256), ('L6 k3 256->256', rb6, 27, 256, 0, 256),
         ('L1 up 128->96', rbu1, 8, 128, 0, 96), ('L2 up 256->128', rbu2, 8, 256, 0, 128), ('L3 up 256->256', rbu3, 8, 256, 0, 256)]
if os.environ.get('CASES'):
    cases = [c for c in cases if any(c[0].startswith(p) for p in os.environ['CASES'].split(','))]
for name, rb, K, c1, c2, co in cases:
    n_out = rb.n_out if rb is not None else m.n(0)
    n_in = rb.n_in if rb is not None else m.n(0)
    x1 = torch.randn(n_in, c1, device='cuda'); x2 = torch.randn(n_in, c2, device='cuda') if c2 else None
    w = torch.randn(K, c1 + c2, co, device='cuda') * 0.05
    P = rb.pairs if rb is not None else n_out
    fl = 2.0 * P * (c1 + c2) * co
    dy = torch.randn(n_out, co, device='cuda'); dw = torch.zeros_like(w)
    xs = x1 if c2 == 0 else torch.cat([x1, x2], 1)
    if os.environ.get('ACC') == '1':       # the data-gradient form: accumulate onto a tensor that is already there
        y_acc = torch.randn(n_out, co, device='cuda')
        f_fwd = lambda: F_.conv_raw(x1, x2, wp, K, None, rb, n_out, co, out=y_acc, accumulate=True)
    else:
        f_fwd = lambda: F_.conv_raw(x1, x2, wp, K, None, rb, n_out, co)
    f_wg = lambda: F_.wgrad_raw(xs, dy, rb, K, dw, 0)

    def f_both():          # data-gradient-shaped launch and weight gradient side by side on two streams, as in the backward pass
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            f_wg()
        f_fwd()
        torch.cuda.current_stream().wait_stream(side)
    res = {v: [[], [], []] for v, _ in VARIANTS}
    for rnd in range(4):
        for v, env in VARIANTS:
            for k_ in SWITCHES: os.environ.pop(k_, None)
            os.environ.update(env)
            _lib.reload_env()
            wp = F_.weight_pack(w)          # the packed layout depends on the strip-width switch
            if rnd == 0: f_fwd(); f_wg(); torch.cuda.synchronize()
            res[v][0].append(timeit(f_fwd)); res[v][1].append(timeit(f_wg))
            if CORUN:
                res[v][2].append(timeit(f_both))
    line = '%-22s' % name
    if os.environ.get('HALF') and (c1 % 16 == 0 and c2 % 16 == 0 and co % 16 == 0):
        # the half-precision inference kernel on the same map (b2m_conv_fwd_h; a 1x1 layer through the identity rulebook)
        for k_ in SWITCHES: os.environ.pop(k_, None)
        _lib.reload_env()
        rbh = rb if rb is not None else m.rulebook_identity(0)
        x1h = x1.half(); x2h = x2.half() if c2 else None
        wh = (w if K > 1 else w[0]).contiguous()
        f_h = lambda: F_.conv_affine_h(x1h, x2h, wh, rbh, n_out)
        f_h(); torch.cuda.synchronize()
        line += ' | half fwd %6.2f TF' % (fl / min(timeit(f_h) for _ in range(4)) / 1e9)
        if K > 1:                       # ... one wave per workgroup (B2M_CONV_GROUP_H=0: rounds 4 - 5) instead of four items per workgroup
            os.environ['B2M_CONV_GROUP_H'] = '0'; _lib.reload_env()
            f_h(); torch.cuda.synchronize()
            line += ' (ungrouped %6.2f)' % (fl / min(timeit(f_h) for _ in range(4)) / 1e9)
            os.environ.pop('B2M_CONV_GROUP_H'); _lib.reload_env()
        if co % 64 == 0 and K > 1:      # ... and in 32-column strips (B2M_CONV_TW4_H=0: the strip width of the fp32 kernels)
            os.environ['B2M_CONV_TW4_H'] = '0'; _lib.reload_env(); F_.invalidate_half_images()
            f_h(); torch.cuda.synchronize()
            line += ' (32-col strips %6.2f)' % (fl / min(timeit(f_h) for _ in range(4)) / 1e9)
            os.environ.pop('B2M_CONV_TW4_H'); _lib.reload_env(); F_.invalidate_half_images()
        if rb is not None and c2 == 0:
            # ... and the half weight gradient (b2m_conv_wgrad_h): f16 MFMA through the transposing LDS read | operands converted on
            # load, fp32 MFMA, flat pipeline | the same, plain kernel
            from box2mask_amd import half_train as HT
            dyh = dy.half(); dwh = torch.zeros_like(w)
            f_wh = lambda: HT._wgrad_h(x1h, dyh, rb, K, dwh, 0, 1.0)
            for tag, env in [('trh', {})] + [('trh ' + kv, dict([kv.split('=')])) for kv in os.environ.get('HALF_WG_VARIANTS', '').split(';') if kv] + [('cvt', {'B2M_WGRAD_TRH': '0'}), ('plain', {'B2M_WGRAD_TRH': '0', 'B2M_WGRAD_PIPE': '0'})]:
                for k_ in SWITCHES + ('B2M_WGRAD_TRH',): os.environ.pop(k_, None)
                os.environ.update(env); _lib.reload_env()
                f_wh(); torch.cuda.synchronize()
                line += ' half wg %s %6.2f TF' % (tag, fl / min(timeit(f_wh) for _ in range(4)) / 1e9)
            for k_ in SWITCHES + ('B2M_WGRAD_TRH',): os.environ.pop(k_, None)
            _lib.reload_env()
    for v, _ in VARIANTS:
        line += ' | %s fwd %6.2f TF wg %6.2f TF' % (v, fl / min(res[v][0]) / 1e9, fl / min(res[v][1]) / 1e9)
        if CORUN:
            line += ' both %6.2f TF' % (2 * fl / min(res[v][2]) / 1e9)
    print(line)
