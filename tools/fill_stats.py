"""Useful share of the executed MFMAs by tile size, per U-Net level, on one synthetic 150 k-voxel scene (CPU only):
pairs of a (tile, offset) are rounded up to MFMA row groups of 16.  T = 64: one tile per item (conv_fwd_flow_kernel,
conv_flow2_kernel<NT = 1>); T = 128: two tiles per item (conv_flow2_kernel<NT = 2>).  Uses the oracle's kernel maps (test
infrastructure, not the product)."""
import sys, numpy as np
sys.path.insert(0,'/root/repo')
from box2mask_amd import synth
from oracle import sparse_ref as S
b = synth.make_batch(1, seed0=0)
c = b['vox_coords'].numpy().astype(np.int64)
def morton(c):
    x,y,z = c[:,1]-c[:,1].min(), c[:,2]-c[:,2].min(), c[:,3]-c[:,3].min()
    k = np.zeros(len(c), np.uint64)
    for bit in range(12):
        k |= ((x>>bit)&1).astype(np.uint64) << np.uint64(3*bit+2)
        k |= ((y>>bit)&1).astype(np.uint64) << np.uint64(3*bit+1)
        k |= ((z>>bit)&1).astype(np.uint64) << np.uint64(3*bit)
    return np.argsort(k, kind='stable')
def hilbert(c, nbits=12):
    """Row order along a Hilbert curve (Skilling's transform; the product's b2m_hilbert_keys)."""
    X = np.stack([c[:,1]-c[:,1].min(), c[:,2]-c[:,2].min(), c[:,3]-c[:,3].min()], 1).astype(np.uint64).copy()
    M = np.uint64(1) << np.uint64(nbits - 1)
    Q = M
    while Q > 1:
        P = Q - np.uint64(1)
        for a in range(3):
            sel = (X[:, a] & Q) != 0
            X[sel, 0] ^= P
            ns = ~sel
            t = (X[ns, 0] ^ X[ns, a]) & P
            X[ns, 0] ^= t; X[ns, a] ^= t
        Q >>= np.uint64(1)
    X[:, 1] ^= X[:, 0]; X[:, 2] ^= X[:, 1]
    t = np.zeros(len(X), np.uint64)
    Q = M
    while Q > 1:
        t[(X[:, 2] & Q) != 0] ^= (Q - np.uint64(1))
        Q >>= np.uint64(1)
    X ^= t[:, None]
    k = np.zeros(len(X), np.uint64)
    for bit in range(nbits):
        for a in range(3):
            k |= ((X[:, a] >> np.uint64(bit)) & np.uint64(1)) << np.uint64(3*bit + (2-a))
    return np.argsort(k, kind='stable')
import os
ORDER = os.environ.get('ORDER', 'hilbert')             # ORDER=morton: the Z-order of rounds 1-3
def order_of(c):
    return hilbert(c) if ORDER == 'hilbert' else morton(c)


def main():
    global c
    print('row order:', ORDER)
    c = c[order_of(c)]
    for lvl in range(4):
        ts = 1<<lvl
        nbr = np.asarray(S.kernel_map_same(c, 3, ts))
        K, N = nbr.shape
        for T in (64, 128):
            nt = (N + T - 1)//T
            pad = nt*T - N
            v = np.concatenate([nbr >= 0, np.zeros((K,pad),bool)],1).reshape(K, nt, T).sum(2)
            act = v > 0
            g = (v + 15)//16
            print('level', lvl, 'N', N, 'T', T, 'pairs/row %.1f'%(v.sum()/N), 'avg cnt/active %.1f'%v[act].mean(), 'active/tile %.1f'%act.sum(0).mean(),
                  'useful %.3f'%(v.sum()/ (16*g.sum())), 'G hist', np.round(np.bincount(g[act].ravel(), minlength=T//16+1)/act.sum(),2))
        c = S.stride_coords(c, ts)[0].astype(np.int64)
        c = c[hilbert(c) if ORDER == 'hilbert' else morton(c)]


if __name__ == '__main__':
    main()
