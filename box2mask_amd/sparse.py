"""Coordinate manager, tile rulebooks and the SparseTensor holder.

Host-side mirror of what the reference gets from MinkowskiEngine's coordinate manager
([ME-mem]): a SparseTensor owns a feature matrix ``F`` and a key
(level) into a :class:`CoordinateManager` that caches coordinate maps and kernel maps per
(level, kernel) for the lifetime of one batch.  All heavy work happens in the HIP library
(``b2m_coords_*``, ``b2m_kernel_map``, ``b2m_rulebook``); torch only allocates device memory.
"""
from __future__ import annotations

import ctypes
import os
import itertools

import torch

from . import _lib

TILE = 64       # B2M_TILE of include/b2m.h
# Spatial (Morton) row order for tensors built through SparseTensor(features, coordinates): an internal permutation,
# undone at the boundary (per-voxel outputs) -- see CoordinateManager.perm.  tests switch it off to compare
# intermediate levels row by row with the oracle.
REORDER_DEFAULT = True
_serial = itertools.count(1)     # unique ids for coordinate managers (cache keys must not be recycled like id())


def _pow2_at_least(n: int) -> int:
    c = 1
    while c < n:
        c <<= 1
    return c


class Rulebook:
    """Tile rulebook of one kernel map: for every (offset k, tile of TILE output rows) the valid
    (input row, output row) pairs, compacted in output-row order (include/b2m.h: b2m_rulebook)."""

    def __init__(self, nbr, K: int, n_out: int, n_in: int, keep_table: bool = False, device=None):
        """nbr: the (K, n_out) neighbour table to compact, or None when the caller fills the arrays itself
        (b2m_kernel_map_rulebook)."""
        dev = nbr.device if nbr is not None else device
        self.K, self.n_out, self.n_in = K, n_out, n_in
        self.ntiles = (n_out + TILE - 1) // TILE
        ldr = self.ntiles * TILE
        self.rb_in = torch.empty(max(K * ldr, 1), dtype=torch.int32, device=dev)
        self.rb_out = torch.empty(max(K * ldr, 1), dtype=torch.uint8, device=dev)
        # K*ntiles pair counts + the tail b2m_rulebook_balance fills (XCD work boundaries, per-tile cost): b2m.h
        self.rb_cnt = torch.empty(_lib.load().b2m_rulebook_cnt_size(K, n_out), dtype=torch.int32, device=dev)
        if nbr is not None:
            ld = nbr.shape[1] if nbr.dim() == 2 else n_out
            _lib.call('b2m_rulebook', nbr.data_ptr(), ld, K, n_out, self.rb_in.data_ptr(), self.rb_out.data_ptr(),
                      self.rb_cnt.data_ptr(), None)
        self.nbr = nbr if keep_table else None
        self._pairs = None
        self.scatter = None        # an UP rulebook: the DOWN rulebook of the same map (CoordinateManager._stride_tables)

    @property
    def pairs(self) -> int:
        """Total number of (in,out) pairs (syncs once; used for FLOP accounting only)."""
        if self._pairs is None:
            self._pairs = int(self.rb_cnt[:self.K * self.ntiles].sum().item())
        return self._pairs


class CoordinateManager:
    """Coordinate maps of one batch at tensor strides 1,2,4,... and their kernel maps."""

    def __init__(self, coords: torch.Tensor, keep_tables: bool = False, check: bool = True, reorder: bool = False):
        _lib.require_gpu()
        assert coords.dim() == 2 and coords.shape[1] == 4, 'coords must be (N,4) [b,x,y,z]'
        if check and coords.numel():
            mn, mx = int(coords.min()), int(coords.max())
            if mn < 0 or mx >= 65535:
                raise ValueError('coordinates must lie in [0, 65534] (got %d..%d)' % (mn, mx))
        dev = torch.device('cuda', torch.cuda.current_device())
        c0 = coords.to(device=dev, dtype=torch.int32, non_blocking=True).contiguous()
        # optional spatial row order: level-0 rows sorted by Morton key; perm[i] = input row held by internal row i
        self.perm = self.inv_perm = None
        if reorder and c0.shape[0] > 1:
            n0 = c0.shape[0]
            keys = torch.empty(n0, dtype=torch.int64, device=dev)
            bits = max(int(mx).bit_length(), 1) if (check and coords.numel()) else 16
            # row order: Hilbert curve (B2M_ROW_ORDER=morton: Z-order, the order of rounds 1-3)
            if os.environ.get('B2M_ROW_ORDER', 'hilbert') == 'morton':
                _lib.call('b2m_morton_keys', c0.data_ptr(), n0, keys.data_ptr())
            else:
                _lib.call('b2m_hilbert_keys', c0.data_ptr(), n0, bits, keys.data_ptr())
            # radix argsort over the key bits that can differ: 3 x bitlength(largest coordinate) interleaved bits + the
            # batch index at bit 48 (the bounds check above read the maximum; without it all 64 bits)
            mask = ((1 << (3 * bits)) - 1) | (((1 << bits) - 1) << 48)
            self.perm = torch.empty(n0, dtype=torch.int64, device=dev)
            self.inv_perm = torch.empty(n0, dtype=torch.int64, device=dev)
            scratch = torch.empty((_lib.load().b2m_radix_argsort_scratch(n0) + 7) // 8, dtype=torch.int64, device=dev)
            _lib.call('b2m_radix_argsort', keys.data_ptr(), n0, mask & 0xFFFFFFFFFFFFFFFF, self.perm.data_ptr(),
                      self.inv_perm.data_ptr(), scratch.data_ptr())
            c0 = c0[self.perm].contiguous()
        self.device = dev
        self.serial = next(_serial)
        self.keep_tables = keep_tables
        self.coords = [c0]                 # level -> (n,4) int32
        self.tables = []                   # level -> (keys, vals, cap)
        self.parent, self.koff = [], []    # level l -> maps of level l rows into level l+1
        self._rb = {}
        self._occ = False                  # level-0 occupancy bitmap: False = not built yet, None = too large
        self.dup_count = torch.zeros(1, dtype=torch.int32, device=dev)
        self.tables.append(self._build_table(c0, self.dup_count))

    # -- coordinate maps
    def _build_table(self, c, dup=None):
        n = c.shape[0]
        cap = _pow2_at_least(max(2 * n, 16))
        keys = torch.empty(cap, dtype=torch.int64, device=self.device)
        vals = torch.empty(cap, dtype=torch.int32, device=self.device)
        _lib.call('b2m_coords_build', c.data_ptr(), n, keys.data_ptr(), vals.data_ptr(), cap, _lib.ptr(dup))
        return keys, vals, cap

    def n(self, level: int) -> int:
        self.ensure_level(level)
        return self.coords[level].shape[0]

    def ensure_level(self, level: int):
        while len(self.coords) <= level:
            l = len(self.coords) - 1
            c = self.coords[l]
            n = c.shape[0]
            cap = _pow2_at_least(max(2 * n, 16))
            keys = torch.empty(cap, dtype=torch.int64, device=self.device)
            vals = torch.empty(cap, dtype=torch.int32, device=self.device)
            cout = torch.empty((max(n, 1), 4), dtype=torch.int32, device=self.device)
            parent = torch.empty(max(n, 1), dtype=torch.int32, device=self.device)
            koff = torch.empty(max(n, 1), dtype=torch.int32, device=self.device)
            scratch = torch.empty(2 * n + n // 1024 + 2, dtype=torch.int32, device=self.device)
            n_out = ctypes.c_int64(0)
            _lib.call('b2m_coords_stride', c.data_ptr(), n, 1 << l, cout.data_ptr(), parent.data_ptr(),
                      koff.data_ptr(), keys.data_ptr(), vals.data_ptr(), cap, scratch.data_ptr(),
                      ctypes.byref(n_out))
            m = int(n_out.value)
            self.coords.append(cout[:m])
            self.tables.append((keys, vals, cap))
            self.parent.append(parent[:n])
            self.koff.append(koff[:n])

    # -- kernel maps
    OCC_MAX_BYTES = 256 << 20

    def _occupancy(self):
        """(bits, X, Y, Z) of the level-0 voxels, or None when the bounding grid is too large for a bitmap."""
        if self._occ is False:
            c = self.coords[0]
            self._occ = None
            if c.shape[0]:
                b, x, y, z = (int(v) + 1 for v in c.amax(0).tolist())
                words = (b * x * y * z + 63) // 64 + 1
                if words * 8 <= self.OCC_MAX_BYTES:
                    bits = torch.empty(words, dtype=torch.int64, device=self.device)
                    _lib.call('b2m_occupancy', c.data_ptr(), c.shape[0], b, x, y, z, bits.data_ptr(), words)
                    self._occ = (bits, x, y, z)
        return self._occ

    def rulebook_same(self, level: int, ksize: int) -> Rulebook:
        """Stride-1 map of a cubic odd kernel on `level` (in == out coordinate set)."""
        key = ('same', level, ksize)
        if key not in self._rb:
            self.ensure_level(level)
            c = self.coords[level]
            n = c.shape[0]
            keys, vals, cap = self.tables[level]
            K = ksize ** 3
            occ = self._occupancy() if (level == 0 and ksize > 1) else None
            occ_args = (occ[0].data_ptr(), occ[1], occ[2], occ[3]) if occ else (None, 0, 0, 0)
            if self.keep_tables:        # tests compare the neighbour table itself
                nbr = torch.empty((K, max(n, 1)), dtype=torch.int32, device=self.device)
                _lib.call('b2m_kernel_map', c.data_ptr(), n, ksize, 1 << level, keys.data_ptr(), vals.data_ptr(), cap,
                          *occ_args, nbr.data_ptr(), nbr.shape[1])
                self._rb[key] = Rulebook(nbr, K, n, n, self.keep_tables)
            else:                       # straight into the rulebook, no K x n table
                rb = Rulebook(None, K, n, n, device=self.device)
                if n:
                    _lib.call('b2m_kernel_map_rulebook', c.data_ptr(), n, ksize, 1 << level, keys.data_ptr(),
                              vals.data_ptr(), cap, *occ_args, rb.rb_in.data_ptr(), rb.rb_out.data_ptr(),
                              rb.rb_cnt.data_ptr())
                self._rb[key] = rb
        return self._rb[key]

    def rulebook_identity(self, level: int) -> Rulebook:
        """The map of a 1x1 layer on `level` as a tile rulebook (pair j of a tile = (row j, row j)): what the half-precision
        convolution kernel, which only walks rulebooks, takes for the block shortcuts (functional.conv_affine_h)."""
        key = ('ident', level)
        if key not in self._rb:
            self.ensure_level(level)
            n = self.coords[level].shape[0]
            nbr = torch.arange(max(n, 1), dtype=torch.int32, device=self.device).unsqueeze(0)
            self._rb[key] = Rulebook(nbr, 1, n, n)
        return self._rb[key]

    def _stride_tables(self, level: int):
        self.ensure_level(level + 1)
        nf, nc = self.n(level), self.n(level + 1)
        child = torch.empty((8, max(nc, 1)), dtype=torch.int32, device=self.device)
        up = torch.empty((8, max(nf, 1)), dtype=torch.int32, device=self.device)
        _lib.call('b2m_stride_tables', self.parent[level].data_ptr(), self.koff[level].data_ptr(), nf, nc,
                  child.data_ptr(), child.shape[1], up.data_ptr(), up.shape[1])
        self._rb[('down', level)] = Rulebook(child, 8, nc, nf, self.keep_tables)
        self._rb[('up', level)] = Rulebook(up, 8, nf, nc, self.keep_tables)
        # the transposed map in scatter form walks the DOWN rulebook (functional.conv_raw -> b2m_conv_up)
        self._rb[('up', level)].scatter = self._rb[('down', level)]

    def tensors(self):
        """Every device tensor this manager (and its kernel maps) owns -- for `record_stream` when the manager was built
        on one stream and is used on another (Model.prefetch)."""
        for t in self.coords + self.parent + self.koff + [self.perm, self.inv_perm, self.dup_count]:
            if t is not None:
                yield t
        for keys, vals, _ in self.tables:
            yield keys
            yield vals
        if self._occ:
            yield self._occ[0]
        for rb in self._rb.values():
            for t in (rb.rb_in, rb.rb_out, rb.rb_cnt, rb.nbr):
                if t is not None:
                    yield t

    def prefetch(self, n_levels: int, same=(), strided: bool = True):
        """Build the coordinate maps of levels 0 .. n_levels-1 and the kernel maps a network will ask for -- `same`:
        (level, kernel size) pairs, `strided`: every k2s2 map between consecutive levels -- NOW.  Every level's row count
        comes back to the host (it sizes the next allocations); asked for lazily, in the middle of a forward pass, each of
        those reads makes the host wait for all the convolutions queued so far and the device then idles while the host
        catches up.  Done up front, the forward and backward passes are enqueued without a single host read."""
        self.ensure_level(n_levels - 1)
        for level, ksize in same:
            if level < n_levels:
                self.rulebook_same(level, ksize)
        if strided:
            for level in range(n_levels - 1):
                self.rulebook_down(level)

    def rulebook_down(self, level: int) -> Rulebook:
        """k2s2 map level -> level+1, tiled over the coarse (output) rows."""
        if ('down', level) not in self._rb:
            self._stride_tables(level)
        return self._rb[('down', level)]

    def rulebook_up(self, level: int) -> Rulebook:
        """Transposed k2s2 map level+1 -> level, tiled over the fine (output) rows."""
        if ('up', level) not in self._rb:
            self._stride_tables(level)
        return self._rb[('up', level)]


class SparseTensor:
    """Feature matrix + coordinate key; mirrors the part of ME.SparseTensor the reference uses
    (``.F``, ``.C``; /root/reference/models/model.py:43,55-56, detection_net.py:347-348,506-510)."""

    def __init__(self, features, coordinates=None, device=None, coordinate_manager: CoordinateManager | None = None,
                 level: int = 0):
        fresh = coordinate_manager is None
        if fresh:
            assert coordinates is not None
            coordinate_manager = CoordinateManager(coordinates, reorder=REORDER_DEFAULT)
        self.manager = coordinate_manager
        self.level = level
        dev = coordinate_manager.device
        self.F = features if features.device == dev else features.to(dev, non_blocking=True)
        # fp32 features; half only for the tensors the half trunk makes itself (SelectionNet.half_trunk, inference)
        if self.F.dtype != torch.float32 and (fresh or self.F.dtype != torch.float16):
            self.F = self.F.float()
        if fresh and coordinate_manager.perm is not None:
            self.F = self.F[coordinate_manager.perm]        # rows follow the manager's internal (spatial) order
        assert self.F.shape[0] == coordinate_manager.n(level), 'feature rows must match the coordinate map'

    @property
    def C(self):
        """Coordinates row-aligned with F, like ME's (C[i] belongs to F[i]).  DEVIATION from MinkowskiEngine: for a tensor
        built from (features, coordinates) both are in the manager's internal spatial (Morton) row order, not in the order of
        the coordinates handed in -- ME keeps the input order of unique coordinates.  SelectionNet undoes the order at its
        boundary (pooling ids in, per-voxel heads out: detection_net.py:347 of the reference relies on input order there);
        code that uses this module as "ME" directly reads `features_in_input_order()` / `coordinates_in_input_order()`, or
        sets `sparse.REORDER_DEFAULT = False` (rows then stay in input order; the kernels lose their L2 locality).
        tests/test_gpu_net.py::test_sparse_tensor_row_order_contract pins all of this."""
        return self.manager.coords[self.level]

    def coordinates_in_input_order(self):
        """Level-0 coordinates in the row order of the coordinates the tensor was built from (ME's `.C`)."""
        m = self.manager
        if self.level == 0 and m is not None and m.inv_perm is not None:
            return m.coords[0][m.inv_perm]
        return m.coords[self.level]

    def features_in_input_order(self):
        """Level-0 features in the row order of the coordinates the tensor was built from."""
        m = self.manager
        if self.level == 0 and m is not None and m.inv_perm is not None:
            return self.F[m.inv_perm]
        return self.F

    @property
    def tensor_stride(self):
        return 1 << self.level

    def new(self, F, level=None):
        return SparseTensor(F, coordinate_manager=self.manager, level=self.level if level is None else level)

    def __len__(self):
        return self.F.shape[0]
