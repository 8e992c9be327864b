"""ORACLE (test infrastructure, never shipped, never the thing measured except as `cpu_baseline`).

CPU restatement of the sparse-tensor engine the reference reaches through MinkowskiEngine 0.5.4
(un-vendored third-party dependency, pinned in /root/reference/docs/installation.md:6,42).

PARITY UNPINNED for everything in this file: MinkowskiEngine is neither vendored in
/root/reference nor installed, and the reference ships no tests or golden vectors at this
boundary.  The restatement follows the semantics observable from the reference's
call sites (file:line cited per function) plus ME 0.5.4's published behaviour, and is validated by
independent checks in tests/test_oracle_sparse.py: dense equivalence against
torch.nn.functional.conv3d / conv_transpose3d, brute-force dictionary kernel maps, BatchNorm1d,
scatter_reduce and fp64 gradcheck.

Conventions:
  * coordinates (N,4) int [b,x,y,z] >= 0; a level with tensor stride ts holds multiples of ts
  * kernel offset index enumerates offsets with the first spatial axis (x) fastest
  * odd kernels are centred, even kernels use offsets {0..k-1}*ts, strided output coordinate is
    floor(c / (s*ts)) * (s*ts)
  * transposed convolution re-uses the existing finer coordinate map (no new coordinates)
  * row order of a strided level = order of first occurrence of each coarse voxel in the finer
    level's row order (the build's own canonical, deterministic order; ME's is a hash artefact)
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import numpy as np
import torch


# ----------------------------------------------------------------------------- coordinates
def pack_keys(coords: np.ndarray) -> np.ndarray:
    """(N,4) non-negative ints -> int64 keys, 16 bits per field (b,x,y,z)."""
    c = np.asarray(coords).astype(np.int64)
    assert c.ndim == 2 and c.shape[1] == 4
    assert c.min(initial=0) >= 0 and c.max(initial=0) < 65536
    return (c[:, 0] << 48) | (c[:, 1] << 32) | (c[:, 2] << 16) | c[:, 3]


def kernel_offsets(ksize: int, ts: int) -> np.ndarray:
    """(K,3) integer offsets of a cubic kernel at tensor stride ts, x fastest.
    Odd k: {-k//2..k//2}*ts; even k: {0..k-1}*ts  ([ME-mem])."""
    if ksize % 2 == 1:
        r = np.arange(-(ksize // 2), ksize // 2 + 1)
    else:
        r = np.arange(0, ksize)
    dz, dy, dx = np.meshgrid(r, r, r, indexing='ij')   # x fastest in the flattened order
    return np.stack([dx.ravel(), dy.ravel(), dz.ravel()], 1).astype(np.int64) * ts


def stride_coords(coords: np.ndarray, ts: int):
    """k=2,s=2 coordinate generation (/root/reference/models/detection_net.py:42,48,54,61,68,74,81).
    Returns (coords_out (M,4), parent (N,) row of the coarse voxel, koff (N,) kernel index 0..7).
    Output rows are ordered by first occurrence in the input order."""
    c = np.asarray(coords).astype(np.int64)
    ts2 = 2 * ts
    coarse = c.copy()
    coarse[:, 1:] = (c[:, 1:] // ts2) * ts2
    keys = pack_keys(coarse)
    uniq, first, inv = np.unique(keys, return_index=True, return_inverse=True)
    order = np.argsort(first, kind='stable')            # unique-id -> rank by first occurrence
    rank = np.empty_like(order)
    rank[order] = np.arange(len(order))
    parent = rank[inv.reshape(-1)]
    coords_out = coarse[first[order]]
    o = (c[:, 1:] - coarse[:, 1:]) // ts                 # in {0,1}^3
    koff = o[:, 0] + 2 * o[:, 1] + 4 * o[:, 2]
    return coords_out.astype(np.int32), parent.astype(np.int32), koff.astype(np.int32)


def kernel_map_same(coords: np.ndarray, ksize: int, ts: int) -> np.ndarray:
    """Stride-1 kernel map as a neighbour table nbr[k, o] = input row at coords[o]+offset_k or -1.
    (/root/reference/models/resnet.py:61-65, detection_net.py:37)"""
    c = np.asarray(coords).astype(np.int64)
    keys = pack_keys(c)
    order = np.argsort(keys, kind='stable')
    skeys = keys[order]
    offs = kernel_offsets(ksize, ts)
    K, N = len(offs), len(c)
    nbr = np.full((K, N), -1, np.int32)
    for k in range(K):
        q = c.copy()
        q[:, 1:] += offs[k]
        ok = (q[:, 1:] >= 0).all(1) & (q[:, 1:] < 65536).all(1)
        qk = pack_keys(np.where(ok[:, None], q, 0))
        pos = np.searchsorted(skeys, qk)
        pos = np.minimum(pos, N - 1)
        hit = ok & (skeys[pos] == qk)
        nbr[k, hit] = order[pos[hit]]
    return nbr


def kernel_map_bruteforce(coords: np.ndarray, ksize: int, ts: int) -> np.ndarray:
    """Dictionary O(N*K) kernel map, used only to check kernel_map_same on small inputs."""
    c = [tuple(int(v) for v in r) for r in np.asarray(coords)]
    d = {r: i for i, r in enumerate(c)}
    offs = kernel_offsets(ksize, ts)
    nbr = np.full((len(offs), len(c)), -1, np.int32)
    for o, (b, x, y, z) in enumerate(c):
        for k, (dx, dy, dz) in enumerate(offs):
            nbr[k, o] = d.get((b, x + int(dx), y + int(dy), z + int(dz)), -1)
    return nbr


def child_table(parent: np.ndarray, koff: np.ndarray, n_out: int) -> np.ndarray:
    """k2s2 map as a neighbour table over coarse rows: child[k, o] = fine row or -1."""
    t = np.full((8, n_out), -1, np.int32)
    t[koff, parent] = np.arange(len(parent), dtype=np.int32)
    return t


def up_table(parent: np.ndarray, koff: np.ndarray) -> np.ndarray:
    """Transposed k2s2 map as a neighbour table over fine rows: up[k, i] = parent[i] iff koff[i]==k."""
    n = len(parent)
    t = np.full((8, n), -1, np.int32)
    t[koff, np.arange(n)] = parent
    return t


# ----------------------------------------------------------------------------- dense ops (torch CPU, autograd)
def conv_nbr(x: torch.Tensor, w: torch.Tensor, nbr, bias: torch.Tensor | None = None) -> torch.Tensor:
    """Y[o] = sum_k X[nbr[k,o]] @ W[k]   (+bias).  The per-offset index_select -> mm -> index_add_
    formulation of ME's gather-GEMM-scatter.  w: (K,Cin,Cout) or (Cin,Cout) for 1x1."""
    if nbr is None:                                   # 1x1 conv: plain mm
        y = x @ (w if w.dim() == 2 else w[0])
    else:
        nbr_t = torch.as_tensor(np.asarray(nbr), dtype=torch.long)
        K, n_out = nbr_t.shape
        y = torch.zeros(n_out, w.shape[-1], dtype=x.dtype)
        for k in range(K):
            o = torch.nonzero(nbr_t[k] >= 0).reshape(-1)
            if o.numel() == 0:
                continue
            y = y.index_add(0, o, x.index_select(0, nbr_t[k, o]) @ w[k])
    if bias is not None:
        y = y + bias.reshape(1, -1)
    return y


def conv_nbr_explicit(x: torch.Tensor, w: torch.Tensor, nbr, gy: torch.Tensor | None = None):
    """Forward and, given dL/dY, the data and weight gradients of `conv_nbr`, written out per offset from the
    definition (no autograd tape, in-place accumulation) so that whole 150 k .. 1.2 M-row layers fit the host:
        Y[o]    += X[i] W[k]          dX[i] += dY[o] W[k]^T          dW[k] = X[i]^T dY[o]      over pairs (i,o) of k.
    tests/test_oracle_sparse.py checks it against autograd of `conv_nbr`.  Returns (y, dx, dw); dx/dw None without gy."""
    nbr_t = torch.as_tensor(np.asarray(nbr), dtype=torch.long)
    K, n_out = nbr_t.shape
    w3 = w if w.dim() == 3 else w.unsqueeze(0)
    with torch.no_grad():
        y = torch.zeros(n_out, w3.shape[2], dtype=x.dtype)
        dx = torch.zeros_like(x) if gy is not None else None
        dw = torch.zeros_like(w3) if gy is not None else None
        for k in range(K):
            o = torch.nonzero(nbr_t[k] >= 0).reshape(-1)
            if o.numel() == 0:
                continue
            i = nbr_t[k, o]
            xi = x.index_select(0, i)
            y.index_add_(0, o, xi @ w3[k])
            if gy is not None:
                go = gy.index_select(0, o)
                dx.index_add_(0, i, go @ w3[k].t())
                dw[k] = xi.t() @ go
    return y, dx, (dw.reshape(w.shape) if dw is not None else None)


def batch_norm(x, weight, bias, running_mean, running_var, training, momentum=0.1, eps=1e-5):
    """MinkowskiBatchNorm == BatchNorm1d on the feature matrix."""
    return torch.nn.functional.batch_norm(x, running_mean, running_var, weight, bias, training, momentum, eps)


def segment_pool(x: torch.Tensor, ids: torch.Tensor, n_seg: int, mode: str = 'avg') -> torch.Tensor:
    """Global avg/max pool by batch index after the batch column was overwritten with pooling ids
    (/root/reference/models/detection_net.py:345-352); output row r <-> pooling id r."""
    idx = ids.reshape(-1, 1).expand(-1, x.shape[1])
    out = torch.zeros(n_seg, x.shape[1], dtype=x.dtype)
    if mode == 'avg':
        return out.scatter_reduce(0, idx, x, 'mean', include_self=False)
    return out.scatter_reduce(0, idx, x, 'amax', include_self=False)


# ----------------------------------------------------------------------------- coordinate hierarchy
class Hierarchy:
    """All coordinate maps / kernel maps SelectionNet needs for one batch (8 levels)."""

    def __init__(self, coords, n_levels: int = 8, k0: int = 5):
        c0 = np.asarray(coords).astype(np.int32)
        self.coords = [c0]
        self.parent, self.koff = [], []
        for l in range(n_levels - 1):
            co, p, ko = stride_coords(self.coords[l], 1 << l)
            self.coords.append(co); self.parent.append(p); self.koff.append(ko)
        self._k3, self._k5 = {}, None
        self.k0 = k0

    def n(self, l):
        return len(self.coords[l])

    def k3(self, l):
        if l not in self._k3:
            self._k3[l] = kernel_map_same(self.coords[l], 3, 1 << l)
        return self._k3[l]

    def k_first(self):
        if self._k5 is None:
            self._k5 = kernel_map_same(self.coords[0], self.k0, 1)
        return self._k5

    def down(self, l):      # level l -> l+1, table over coarse rows
        return child_table(self.parent[l], self.koff[l], self.n(l + 1))

    def up(self, l):        # level l+1 -> l, table over fine rows
        return up_table(self.parent[l], self.koff[l])
