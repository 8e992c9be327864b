"""Visits, units and full visits of the tile rulebook by tile size (64 ... 512 output rows) and U-Net level, on one synthetic
150 k-voxel scene (CPU only; the oracle's kernel maps: test infrastructure, not the product).  Round 5: what merging the pair
lists of several tiles would buy (the cooperative 256-row tile, profiles/r05_analysis.md section 1) and how many (tile, offset)
visits are FULL -- all 64 pairs, i.e. the identity row mapping that lets consecutive visits chain their accumulators
(conv_fwd_flow_kernel, ConvArgs::chain).  A tile of T rows cuts an offset's n pairs into units of <= 64 pairs = <= 4 MFMA row groups."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fill_stats as FS
from oracle import sparse_ref as S

c = FS.c[FS.order_of(FS.c)]
print('row order:', FS.ORDER)
for lvl in range(4):
    ts = 1 << lvl
    nbr = np.asarray(S.kernel_map_same(c, 3, ts))
    K, N = nbr.shape
    for T in (64, 128, 256, 512):
        nt = (N + T - 1) // T
        pad = nt * T - N
        v = np.concatenate([nbr >= 0, np.zeros((K, pad), bool)], 1).reshape(K, nt, T).sum(2)
        act = v > 0
        g = (v + 15) // 16
        units = (v + 63) // 64                       # visits of <= 4 groups
        lastg = np.where(v % 64 == 0, 4, ((v % 64) + 15) // 16)
        lastg = np.where(v == 0, 0, lastg)
        nfull = np.where(v > 0, units - 1, 0)
        hist = np.zeros(5)
        hist[4] += nfull.sum()
        for G in range(1, 5):
            hist[G] += (lastg[act] == G).sum()
        full64 = (v == T).sum() if T == 64 else 0
        print('level', lvl, 'N', N, 'T', T, 'useful %.3f' % (v.sum() / (16 * g.sum())), 'units/row %.3f' % (units.sum() / N),
              'row groups/unit %.2f' % (g.sum() / units.sum()), 'units by row groups 1..4', np.round(hist[1:] / hist.sum(), 3),
              'MFMA share in one-group units %.3f' % (hist[1] / g.sum()), 'full visits %.3f' % (full64 / act.sum()),
              'units per tile %.1f' % (units.sum() / nt))
    c = S.stride_coords(c, ts)[0].astype(np.int64)
    c = c[FS.order_of(c)]
