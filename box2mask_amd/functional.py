"""Autograd operators over the HIP C ABI (include/b2m.h).

torch is used for device memory, streams and the autograd graph only; every forward/backward body
is one or more b2m_* launches on the current stream.  The semantics of each operator are those of
the MinkowskiEngine operator the reference calls (file:line cited per function).
"""
from __future__ import annotations

import ctypes
import itertools
import os
import weakref

import torch
import torch.distributed as dist

from . import _lib
from .grad_arena import grad_slot
from .sparse import Rulebook

_call = _lib.call
_ptr = _lib.ptr


def deterministic() -> bool:
    """B2M_DETERMINISTIC=1: every order-dependent reduction of the path takes its ordered form -- two-stage weight-gradient
    combine, un-split forward/data-gradient maps (no atomic split-K combine), sorted segment mean.  Same bits on every
    run; slower on the small deep-level maps.  The library reads the same variable (csrc/conv.hip)."""
    return os.environ.get('B2M_DETERMINISTIC', '0') == '1'


_wgrad_ws = {}


def _wgrad_workspace(K, cin, cout, device):
    """Partial-dW buffer of the deterministic weight gradient (grown on demand, shared by all layers: launches on one
    stream run in order)."""
    need = _lib.load().b2m_conv_wgrad_workspace(K, cin, cout)
    ws = _wgrad_ws.get(device)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.float32, device=device)
        _wgrad_ws[device] = ws
    return ws


def _f32c(t):
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def weight_pack(w3, transpose: bool = False, mirror: bool = False, slice_begin: int = 0, slice_count: int | None = None,
                out=None):
    """(K,Cin,Cout) weights -> packed MFMA B-fragment image (include/b2m.h: b2m_weight_pack).
    transpose=True packs the data-gradient operand for input channels [slice_begin, slice_begin+slice_count)."""
    K, cin, cout = w3.shape
    if slice_count is None:
        slice_count = cin
    ci, co = (cout, slice_count) if transpose else (cin, cout)
    wp = out
    if wp is None:
        size = _lib.load().b2m_weight_pack_size(K, ci, co)
        wp = torch.empty(size, dtype=torch.float32, device=w3.device)
    _call('b2m_weight_pack', w3.data_ptr(), w3.stride(1), K, cin, cout, 1 if transpose else 0, 1 if mirror else 0,
          slice_begin, slice_count, wp.data_ptr())
    return wp


class _PackedWeights:
    """Packed images of the network's weights, rebuilt for ALL layers with one launch per forward pass.

    Every (weight, variant) a convolution asks for is registered on first use with a persistent image buffer;
    `begin_pass()` (called at the start of SelectionNet.forward) repacks every registered image in a single
    b2m_weight_pack_run launch and opens a new pass; lookups during that forward and its backward are hits, so a
    training step issues 1 pack launch instead of ~200.  begin_pass repacks UNCONDITIONALLY: tensor version counters
    are not a reliable change signal (fused optimizers update parameters without bumping them).  Outside a pass
    opened for the current weights (direct use of the functional API) an image is reused only if the tensor's
    version counter and address are unchanged, otherwise it is repacked on the spot.  Entries die with their tensors."""

    def __init__(self):
        self.entries = {}          # key -> [weakref(weight), args, image, version, data_ptr, pass_id]
        self.plan = None           # (device plan, n, total_blocks, keys)
        self.split = (None, None, None)
        self.group = {}
        self.pending = {}
        self.dirty = True
        self.pass_id = 0
        self.env_epoch = _lib.env_epoch[0]

    @staticmethod
    def _w3(weight):
        w3 = weight if weight.dim() == 3 else weight.unsqueeze(0)
        assert w3.dtype == torch.float32 and w3.is_contiguous()
        return w3

    def get(self, weight, transpose=False, mirror=False, slice_begin=0, slice_count=None):
        if self.env_epoch != _lib.env_epoch[0]:          # the library's switches changed (the images' strip width is one): start over
            self.__init__()
        w3 = self._w3(weight.detach())
        K, cin, cout = w3.shape
        sc = cin if slice_count is None else slice_count
        key = (id(weight), bool(transpose), bool(mirror), int(slice_begin), int(sc))
        e = self.entries.get(key)
        if e is not None and e[0]() is weight and e[4] == weight.data_ptr():
            if self.pending:                                         # packed on the side stream in this pass: first user waits
                ev = self.pending.pop(self.group.get(key), None)
                if ev is not None:
                    torch.cuda.current_stream(weight.device).wait_event(ev)
            if e[5] != self.pass_id or e[3] != weight._version:      # not packed in this pass / changed since
                weight_pack(w3, transpose, mirror, slice_begin, sc, out=e[2])
                e[3], e[5] = weight._version, self.pass_id
            return e[2]
        image = weight_pack(w3, transpose, mirror, slice_begin, sc)
        self.entries[key] = [weakref.ref(weight), (K, cin, cout, bool(transpose), bool(mirror), int(slice_begin), int(sc)),
                             image, weight._version, weight.data_ptr(), self.pass_id]
        self.dirty = True
        return image

    def _plan_of(self, keys):
        """Device plan (descriptor table) of the images `keys`: (tensor, n, blocks, keys) or None."""
        import ctypes as C
        import numpy as np
        n = len(keys)
        if n == 0:
            return None
        lib = _lib.load()
        es = [self.entries[k] for k in keys]
        i64 = lambda v: np.ascontiguousarray(v, dtype=np.int64)
        i32 = lambda v: np.ascontiguousarray(v, dtype=np.int32)
        w = i64([e[4] for e in es]); wp = i64([e[2].data_ptr() for e in es])
        ldw = i64([e[1][2] for e in es])            # contiguous (K,Cin,Cout): row pitch = Cout
        cols = [i32([e[1][j] for e in es]) for j in (0, 1, 2)]
        flags = [i32([int(e[1][j]) for e in es]) for j in (3, 4, 5, 6)]
        host = np.zeros(n * lib.b2m_weight_pack_plan_size(), np.uint8)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        blocks = lib.b2m_weight_pack_plan(n, p(w), p(wp), p(ldw), p(cols[0]), p(cols[1]), p(cols[2]), p(flags[0]),
                                          p(flags[1]), p(flags[2]), p(flags[3]), p(host))
        if blocks < 0:
            raise _lib.B2MError('b2m_weight_pack_plan failed: ' + lib.b2m_last_error().decode())
        dev = es[0][2].device
        return (torch.from_numpy(host).to(dev), n, int(blocks), keys)

    # forward images packed on the main stream in front of the first layer: the first EARLY_BYTES of them in order of first use
    # (the 5x5x5 stem and the encoder's 32- / 64-channel levels; a training pass reaches the first later layer milliseconds in)
    EARLY_BYTES = 6 << 20

    def _build_plan(self):
        dead = [k for k, e in self.entries.items() if e[0]() is None or e[0]().data_ptr() != e[4]]
        for k in dead:
            del self.entries[k]
        keys = list(self.entries)
        self.dirty = False
        self.plan = self._plan_of(keys)
        # the same images in three groups (pack_on_side_stream): forward images needed first | the other forward images | the
        # data-gradient images (transposed: needed by the backward pass only)
        early, late, back, size = [], [], [], 0
        for k in keys:                              # (registration order = order of first use)
            if k[1]:
                back.append(k)
                continue
            size += self.entries[k][2].numel() * 4
            (early if size <= self.EARLY_BYTES else late).append(k)
        self.split = (self._plan_of(early), self._plan_of(late), self._plan_of(back))
        self.group = {k: g for g, ks in enumerate((early, late, back)) for k in ks}

    def begin_pass(self, inference=False):
        """Open a new pass: repack every registered image from the current weights (one launch).
        inference (an eval-mode pass without gradients): the images of the previous pass are kept when no training pass has run
        since they were packed (training_epoch) and no image was registered since -- 0.37 ms per pass that a batch-size-1
        forward pass of 5-7 ms would otherwise spend on weights that cannot have changed; a parameter changed through torch
        (load_state_dict, in-place ops) is still caught by its version counter at lookup (`get`)."""
        if self.env_epoch != _lib.env_epoch[0]:
            self.__init__()
        if inference and not self.dirty and self.plan is not None and getattr(self, '_epoch', None) == training_epoch():
            return
        self._epoch = training_epoch()
        self.pass_id += 1
        if not self.entries:
            return
        if self.dirty or any(e[0]() is None or e[0]().data_ptr() != e[4] for e in self.entries.values()):
            self._build_plan()
        if self.plan is None:
            return
        plan, n, blocks, keys = self.plan
        self.pending = {}                # group -> event the first user of one of its images waits for
        dev = plan.device
        if not inference and pack_on_side_stream() and wgrad_on_side_stream() and torch.is_grad_enabled():
            # A training pass needs a few small images at once and the bulk of them milliseconds later (the 7 MB images of the
            # 256-channel levels, every data-gradient image): those are packed on the side stream, beside the first layers --
            # the stem is bound by latency, not by bandwidth -- and whoever asks for one first makes its stream wait (`get`).
            main, side = torch.cuda.current_stream(dev), _side_stream(dev)
            early, late, back = self.split
            if early is not None:
                _call('b2m_weight_pack_run', early[0].data_ptr(), early[1], early[2])
            side.wait_stream(main)       # the optimizer's update of the weights (and everything before it)
            with torch.cuda.stream(side):
                for g, pl in ((1, late), (2, back)):
                    if pl is not None:
                        _call('b2m_weight_pack_run', pl[0].data_ptr(), pl[1], pl[2])
                        ev = torch.cuda.Event()
                        ev.record(side)
                        self.pending[g] = ev
        else:
            _call('b2m_weight_pack_run', plan.data_ptr(), n, blocks)
        for k in keys:
            e = self.entries[k]
            e[3], e[5] = e[0]()._version, self.pass_id

    def join(self):
        """The current stream waits for every image still being packed on the side stream (before anything that reads images
        outside `get`: nothing in this package does; tests and tools may)."""
        for g in list(self.pending):
            torch.cuda.current_stream().wait_event(self.pending.pop(g))


def pack_on_side_stream() -> bool:
    """B2M_PACK_STREAM=0: every weight image of a pass is packed on the main stream in front of the first layer (rounds 1-5)."""
    return os.environ.get('B2M_PACK_STREAM', '1') == '1'


packed_weights = _PackedWeights()


class _ZeroSlab:
    """Zero-filled output tensors for the convolutions on the tiniest maps.  A map with fewer than 8 tiles (< 512 rows: the two
    deepest levels of a training batch) is split over up to 16 waves per (tile, strip) that meet in the output with fp32 atomics,
    so the output must start at zero -- one hipMemset2DAsync per launch inside b2m_conv_fwd, 52 of the ~100 runtime memsets of a
    training step (profiles/r06_analysis.md).  Here such an output is a piece of ONE zero-filled chunk per pass (a single
    memset of a few MB) and the convolution is told to accumulate onto it: the library then launches nothing but the kernel.
    A chunk is never reused (pieces are saved for backward): when it is used up -- or a new pass begins -- the next one is made;
    the pieces keep their chunk alive."""
    CHUNK = 4 << 20          # floats (16 MB): a training pass of the 8-level U-Net takes ~1 M

    def __init__(self):
        self.chunk = {}      # device -> [tensor, floats used]

    def new_pass(self):
        self.chunk.clear()

    def take(self, n, c, device):
        need = (n * c + 63) // 64 * 64                  # 256-byte aligned pieces
        if need > self.CHUNK // 4:
            return None
        e = self.chunk.get(device)
        if e is None or e[1] + need > e[0].numel():
            e = [torch.zeros(self.CHUNK, dtype=torch.float32, device=device), 0]
            self.chunk[device] = e
        t = e[0][e[1]:e[1] + n * c].view(n, c)
        e[1] += need
        t._b2m_slab = True                              # (a view, but nobody else's: see _accumulation_target)
        return t


zero_slab = _ZeroSlab()
TINY_MAP_ROWS = 8 * 64       # below 8 tiles b2m_conv_fwd splits an item over up to 16 waves that add atomically (csrc/conv.hip)


def conv_zero_slab() -> bool:
    """B2M_CONV_ZERO_SLAB=0: the tiny maps' outputs are plain allocations and b2m_conv_fwd zero-fills each one itself."""
    return os.environ.get('B2M_CONV_ZERO_SLAB', '1') == '1'


def conv_raw(x1, x2, wp, K: int, bias, rb: Rulebook | None, n_out: int, cout: int, out=None, accumulate=False,
             tile_stats: list | None = None, logical_cin: int | None = None):
    """Y = sum_k [x1|x2][in_k] @ B[k] with B given as a packed image for (K, c1+c2, cout).
    tile_stats: a list; if the kernel of this shape can, it also leaves the per-tile column sums of Y (sum and sum of
    squares over each tile of 64 rows: the statistics of the BatchNorm that follows) and the list receives
    (tensor [ntiles, 2, cout], ntiles) -- b2m_conv_fwd_stats."""
    c1 = x1.shape[1]
    c2 = x2.shape[1] if x2 is not None else 0
    if out is None and rb is not None and K > 1 and 0 < n_out < TINY_MAP_ROWS and not deterministic() and conv_zero_slab():
        out = zero_slab.take(n_out, cout, x1.device)     # zero already: nothing for the library to fill (accumulate = 1)
        accumulate = out is not None
    if out is None:
        out = torch.empty((n_out, cout), dtype=torch.float32, device=x1.device)
    meta = None if logical_cin is None else {'cin': logical_cin}
    if rb is None:
        assert K == 1
        rbi = rbo = rbc = None
    else:
        assert rb.K == K and rb.n_out == n_out
        rbi, rbo, rbc = rb.rb_in.data_ptr(), rb.rb_out.data_ptr(), rb.rb_cnt.data_ptr()
    if rb is not None and rb.scatter is not None and n_out > 0 and x1.shape[0] > 0:
        # a transposed k2s2 map: scatter form over the DOWN rulebook where the kernel takes the shape (b2m_conv_up); no per-tile
        # column sums there (tile_stats stays empty: the BatchNorm behind it reads the output)
        sc = rb.scatter
        assert sc.K == K and sc.n_out == x1.shape[0]
        ran = ctypes.c_int32(0)
        m_up = {'ran': ran}
        if logical_cin is not None:
            m_up['cin'] = logical_cin
        _call('b2m_conv_up', x1.data_ptr(), x1.stride(0), c1, _ptr(x2), x2.stride(0) if x2 is not None else 0, c2, x1.shape[0],
              wp.data_ptr(), K, _ptr(bias), sc.rb_in.data_ptr(), sc.rb_out.data_ptr(), sc.rb_cnt.data_ptr(), out.data_ptr(),
              out.stride(0), cout, n_out, 1 if accumulate else 0, None, None, None, 0, 0, ctypes.byref(ran), meta=m_up)
        if ran.value:
            return out
    if tile_stats is not None and rb is not None and n_out > 0:
        ntiles = (n_out + 63) // 64
        ts = torch.empty((ntiles, 2, cout), dtype=torch.float64, device=x1.device)
        wrote = ctypes.c_int32(0)
        _call('b2m_conv_fwd_stats', x1.data_ptr(), x1.stride(0), c1, _ptr(x2), x2.stride(0) if x2 is not None else 0, c2,
              x1.shape[0], wp.data_ptr(), K, _ptr(bias), rbi, rbo, rbc, n_out, out.data_ptr(), out.stride(0), cout,
              1 if accumulate else 0, ts.data_ptr(), ctypes.byref(wrote), meta=meta)
        if wrote.value:
            tile_stats.append((ts, ntiles))
        return out
    _call('b2m_conv_fwd', x1.data_ptr(), x1.stride(0), c1, _ptr(x2), x2.stride(0) if x2 is not None else 0, c2,
          x1.shape[0], wp.data_ptr(), K, _ptr(bias), rbi, rbo, rbc, n_out, out.data_ptr(), out.stride(0), cout,
          1 if accumulate else 0, meta=meta)
    return out


# ---- half activations (inference): b2m_conv_fwd_h
_half_images = {}      # id(weight) -> [weakref(weight), (c1, c2), image, version, data_ptr, epoch]
_half_epoch = [0]


def note_training_pass():
    """Advance the training epoch: everything cached for INFERENCE from parameters and running statistics -- the half weight
    images (weight_pack_h), the eval-mode BatchNorm affine maps (nn.MinkowskiBatchNorm.eval_affine) -- is rebuilt on its next
    use.  Called by every training-mode pass (SelectionNet.forward, every training-mode BatchNorm): an optimizer step follows,
    and neither a fused optimizer's update of the parameters nor this package's update of the running statistics bumps the
    tensors' version counters, so a validation pass between training steps would otherwise run on stale weights / statistics."""
    _half_epoch[0] += 1


def training_epoch() -> int:
    return _half_epoch[0]


invalidate_half_images = note_training_pass


def _after_optimizer_step(optimizer, args, kwargs):
    note_training_pass()


# Parameters also change where no training-mode pass of this package comes first: `eval pass, optimizer.step(), eval pass`
# (the step belongs to a backward pass from before the first eval pass), or a raw-pointer / fused update that bumps no version
# counter.  Every torch optimizer step therefore advances the epoch too (one global post-step hook, an integer increment);
# parallel.GradAllReduce.broadcast_parameters, which writes through `p.data`, calls note_training_pass itself.
try:
    from torch.optim.optimizer import register_optimizer_step_post_hook as _reg_post_hook
    _reg_post_hook(_after_optimizer_step)
except ImportError:          # (torch < 2.0 has no global hook: the training-mode pass is the only signal there)
    pass



def weight_pack_h(weight, c1: int, c2: int):
    """Half image of a layer's forward weights for b2m_conv_fwd_h (include/b2m.h), cached on the tensor's version counter and
    address (load_state_dict, in-place edits) and on the epoch counter that every training pass advances
    (invalidate_half_images)."""
    w3 = weight.detach()
    w3 = w3 if w3.dim() == 3 else w3.unsqueeze(0)
    assert w3.dtype == torch.float32 and w3.is_contiguous()
    K, cin, cout = w3.shape
    assert cin == c1 + c2
    e = _half_images.get(id(weight))
    if (e is not None and e[0]() is weight and e[1] == (c1, c2) and e[3] == weight._version and e[4] == weight.data_ptr()
            and e[5] == (_half_epoch[0], _lib.env_epoch[0])):       # (env_epoch: the image's strip width is a library switch)
        return e[2]
    n = _lib.load().b2m_weight_pack_h_size(K, c1, c2, cout)
    image = torch.empty(n, dtype=torch.float16, device=w3.device)
    _call('b2m_weight_pack_h', w3.data_ptr(), cout, K, c1, c2, cout, image.data_ptr())
    if e is None:                                   # a new parameter: drop the images of parameters that no longer exist
        for k in [k for k, v in _half_images.items() if v[0]() is None]:
            del _half_images[k]
    _half_images[id(weight)] = [weakref.ref(weight), (c1, c2), image, weight._version, weight.data_ptr(), (_half_epoch[0], _lib.env_epoch[0])]
    return image


def conv_affine_h(x1, x2, weight, rb: Rulebook, n_out: int, scale=None, shift=None, residual=None, relu=False):
    """conv_affine on HALF activations: y(half) = [relu](fmaf(conv(x1 | x2), scale, shift) [+ residual(half)]), fp32
    accumulation, through conv_fwd_flow_kernel<.., F16> (b2m_conv_fwd_h).  Inference only; `rb` is a real rulebook (a 1x1
    layer passes CoordinateManager.rulebook_identity)."""
    assert x1.dtype == torch.float16 and (x2 is None or x2.dtype == torch.float16) and rb is not None
    x1 = x1 if x1.stride(1) == 1 else x1.contiguous()
    x2 = None if x2 is None else (x2 if x2.stride(1) == 1 else x2.contiguous())
    w3 = weight if weight.dim() == 3 else weight.unsqueeze(0)
    K, cin, cout = w3.shape
    c1 = x1.shape[1]
    c2 = x2.shape[1] if x2 is not None else 0
    assert rb.K == K and rb.n_out == n_out and c1 + c2 == cin
    wp = weight_pack_h(weight, c1, c2)
    out = torch.empty((n_out, cout), dtype=torch.float16, device=x1.device)
    if n_out == 0:
        return out
    if residual is not None:
        assert residual.dtype == torch.float16 and residual.stride(1) == 1
    _call('b2m_conv_fwd_h', x1.data_ptr(), x1.stride(0), c1, _ptr(x2), x2.stride(0) if x2 is not None else 0, c2, x1.shape[0],
          wp.data_ptr(), K, rb.rb_in.data_ptr(), rb.rb_out.data_ptr(), rb.rb_cnt.data_ptr(), n_out, out.data_ptr(), out.stride(0),
          cout, _ptr(scale), _ptr(shift), _ptr(residual), residual.stride(0) if residual is not None else 0, 1 if relu else 0,
          meta={'half': True})
    return out


def conv_affine(x1, x2, weight, rb: Rulebook | None, n_out: int, scale, shift, residual=None, relu=False):
    """Inference form of conv -> eval-mode BatchNorm (+ residual) (+ ReLU): y = [relu](fmaf(conv(x), scale, shift) [+ res])
    in ONE launch where the layer's kernel can transform its strip on the way out (b2m_conv_fwd_affine), else convolution
    + b2m_bn_apply -- the same bits either way.  No autograd (the caller checks torch.is_grad_enabled()).
    (/root/reference/models/resnet.py:70-83, detection_net.py:234-337 under model.eval().)"""
    if x1.dtype == torch.float16:                       # the half trunk (SelectionNet.half_trunk)
        return conv_affine_h(x1, x2, weight, rb, n_out, scale, shift, residual, relu)
    x1 = _f32c(x1)
    x2 = _f32c(x2) if x2 is not None else None
    c1 = x1.shape[1]
    logical_cin = None
    if x2 is None and c1 % 4 != 0 and c1 < 16:           # the 6-channel network input (see _SparseConv.forward)
        xp = torch.nn.functional.pad(x1, (0, 16 - c1))
        x1 = xp[:, :(c1 + 3) // 4 * 4]
        logical_cin = c1
    w3 = weight if weight.dim() == 3 else weight.unsqueeze(0)
    K, cin, cout = w3.shape
    wp = packed_weights.get(weight)
    c1 = x1.shape[1]
    c2 = x2.shape[1] if x2 is not None else 0
    out = torch.empty((n_out, cout), dtype=torch.float32, device=x1.device)
    if rb is None:
        rbi = rbo = rbc = None
    else:
        assert rb.K == K and rb.n_out == n_out
        rbi, rbo, rbc = rb.rb_in.data_ptr(), rb.rb_out.data_ptr(), rb.rb_cnt.data_ptr()
    if residual is not None:
        residual = _f32c(residual)
    if n_out == 0:
        return out
    if rb is not None and rb.scatter is not None and x1.shape[0] > 0:          # transposed k2s2 map: scatter form (b2m_conv_up)
        sc = rb.scatter
        ran = ctypes.c_int32(0)
        _call('b2m_conv_up', x1.data_ptr(), x1.stride(0), c1, _ptr(x2), x2.stride(0) if x2 is not None else 0, c2, x1.shape[0],
              wp.data_ptr(), K, None, sc.rb_in.data_ptr(), sc.rb_out.data_ptr(), sc.rb_cnt.data_ptr(), out.data_ptr(),
              out.stride(0), cout, n_out, 0, scale.data_ptr(), shift.data_ptr(), _ptr(residual),
              residual.stride(0) if residual is not None else 0, 1 if relu else 0, ctypes.byref(ran), meta={'ran': ran})
        if ran.value:
            return out
    fused = ctypes.c_int32(0)
    _call('b2m_conv_fwd_affine', x1.data_ptr(), x1.stride(0), c1, _ptr(x2), x2.stride(0) if x2 is not None else 0, c2,
          x1.shape[0], wp.data_ptr(), K, rbi, rbo, rbc, n_out, out.data_ptr(), out.stride(0), cout, scale.data_ptr(),
          shift.data_ptr(), _ptr(residual), residual.stride(0) if residual is not None else 0, 1 if relu else 0,
          ctypes.byref(fused), meta=None if logical_cin is None else {'cin': logical_cin})
    if not fused.value:
        _call('b2m_bn_apply', out.data_ptr(), out.stride(0), n_out, cout, scale.data_ptr(), shift.data_ptr(), _ptr(residual),
              residual.stride(0) if residual is not None else 0, 1 if relu else 0, out.data_ptr(), out.stride(0))
    return out


def bn_eval_affine(gamma, beta, running_mean, running_var, eps):
    """(scale, shift) of an eval-mode BatchNorm: y = fmaf(x, scale, shift) (b2m_bn_finalize on the running statistics)."""
    c = running_mean.shape[0]
    scale = torch.empty(c, dtype=torch.float32, device=running_mean.device)
    shift = torch.empty(c, dtype=torch.float32, device=running_mean.device)
    _call('b2m_bn_finalize', None, 1.0, None, c, _ptr(gamma), _ptr(beta), eps, 0.0, running_mean.data_ptr(),
          running_var.data_ptr(), None, None, scale.data_ptr(), shift.data_ptr())
    return scale, shift


def conv_affine_enabled() -> bool:
    """B2M_CONV_AFFINE=0: inference runs convolution and BatchNorm as separate launches (the training-mode layering)."""
    return os.environ.get('B2M_CONV_AFFINE', '1') == '1'


def wgrad_raw(x, dy, rb: Rulebook | None, K: int, dw3, ci0: int, cin: int | None = None):
    """dw3[:, ci0:ci0+cin, :] += sum_pairs x[in, :cin]^T dy[out]   (cin defaults to x.shape[1])."""
    cin_total, cout = dw3.shape[1], dw3.shape[2]
    cin = x.shape[1] if cin is None else cin
    n_out = dy.shape[0]
    if rb is None:
        rbi = rbo = rbc = None
    else:
        rbi, rbo, rbc = rb.rb_in.data_ptr(), rb.rb_out.data_ptr(), rb.rb_cnt.data_ptr()
    ws = _wgrad_workspace(K, cin, cout, x.device) if deterministic() else None
    sc = rb.scatter if rb is not None else None
    if sc is not None and n_out > 0 and x.shape[0] > 0 and wgrad_up_over_down_map():
        # a transposed k2s2 map: the same pairs through the map's DOWN rulebook (tiled over the coarse rows: up to 64 pairs per
        # (tile, offset), where the UP rulebook has ~8 in half-empty 16-pair slots) with the roles of its two row numbers
        # exchanged -- x lives on the tiles' own (coarse) rows, dy on the fine rows the pair lists name
        assert sc.K == K and sc.n_out == x.shape[0]
        _call('b2m_conv_wgrad_tr', x.data_ptr(), x.stride(0), cin, n_out, dy.data_ptr(), dy.stride(0), cout,
              sc.rb_in.data_ptr(), sc.rb_out.data_ptr(), sc.rb_cnt.data_ptr(), x.shape[0], K,
              dw3.data_ptr() + 4 * ci0 * cout, cout, cin_total * cout, _ptr(ws))
        return
    _call('b2m_conv_wgrad', x.data_ptr(), x.stride(0), cin, x.shape[0], dy.data_ptr(), dy.stride(0), cout, rbi, rbo, rbc,
          n_out, K, dw3.data_ptr() + 4 * ci0 * cout, cout, cin_total * cout, _ptr(ws))


def wgrad_up_over_down_map() -> bool:
    """B2M_WGRAD_UP=0: the weight gradient of a transposed k2s2 map walks the UP rulebook (b2m_conv_wgrad) instead of the DOWN
    rulebook with exchanged roles (b2m_conv_wgrad_tr)."""
    return os.environ.get('B2M_WGRAD_UP', '1') == '1'


# ---- weight gradients on a second stream
# The data gradient and the weight gradient of a layer both need only dy: issued on two HIP streams they share the
# chip.  Neither kernel holds every SIMD's wave slots to the end of its launch (the last round of workgroups of a grid
# leaves slots empty, and a conv_fwd_flow wave needs 12 KiB of LDS where a weight-gradient wave needs none), so the two
# fill each other's gaps.  B2M_WGRAD_STREAM=0 keeps everything on one stream; the deterministic mode always does (its
# partial-sum workspace is shared by all layers in stream order).
_side = {'streams': {}, 'armed': False}


def wgrad_on_side_stream() -> bool:
    return os.environ.get('B2M_WGRAD_STREAM', '1') == '1' and not deterministic()


def _side_stream(device):
    st = _side['streams'].get(device)
    if st is None:
        st = torch.cuda.Stream(device=device)
        _side['streams'][device] = st
    return st


def join_side_streams():
    """The current stream waits for the weight gradients issued on the side stream(s).  Runs by itself at the end of
    every backward pass; the data-parallel all-reduce calls it before it launches a bucket."""
    _side['armed'] = False
    for dev, st in _side['streams'].items():
        torch.cuda.current_stream(dev).wait_stream(st)


class _SparseConv(torch.autograd.Function):
    """Sparse convolution Y[o] = sum_k X[in_k(o)] W[k] (+bias).  [ME-mem] MinkowskiConvolution /
    MinkowskiConvolutionTranspose forward+backward (/root/reference/models/resnet.py:61-65,
    detection_net.py:37-135).  `rb_f` maps input rows to output rows; `rb_b` is the reverse map used
    for the data gradient (same rulebook for stride-1 kernels, whose offsets mirror)."""

    @staticmethod
    def forward(ctx, x1, x2, weight, bias, rb_f, rb_b, mirror, n_out, tile_stats=None, passthrough=False):
        in1, in2 = x1, x2
        x1 = _f32c(x1)
        x2 = _f32c(x2) if x2 is not None else None
        c1 = x1.shape[1]
        if x2 is None and c1 % 4 != 0 and c1 < 16:
            # few, odd input channels (the 6-channel network input): zero-pad the rows to a 16-float pitch.  The
            # forward reads an 8-channel view of it (aligned vector gathers; the packed weights are zero there as
            # well), the weight gradient reads whole 16-channel blocks of the padded rows.
            xp = torch.nn.functional.pad(x1, (0, 16 - c1))
            x1 = xp[:, :(c1 + 3) // 4 * 4]
        w3 = weight if weight.dim() == 3 else weight.unsqueeze(0)
        w3 = _f32c(w3)
        K, cin, cout = w3.shape
        wp = packed_weights.get(weight)
        y = conv_raw(x1, x2, wp, K, bias, rb_f, n_out, cout, tile_stats=tile_stats,
                     logical_cin=c1 if x1.shape[1] != c1 else None)
        ctx.save_for_backward(x1, x2, weight, bias)
        ctx.rb_f, ctx.rb_b, ctx.mirror, ctx.c1 = rb_f, rb_b, mirror, c1
        ctx.src1, ctx.src2 = _node_id(in1), _node_id(in2)     # the nodes the data gradients are made for (`_own`)
        ctx._b2m_tag = next(_node_tags)                       # this node's own name (`_accumulation_target`)
        if passthrough:
            # The inputs come back as second / third outputs: whoever else consumes them (the residual branch of a
            # BasicBlock, its 1x1 shortcut) takes THESE, so their gradients arrive here, in one call with dy, and the data
            # gradient is accumulated onto them by the kernel (accumulate = 1) instead of by an add kernel of autograd's
            # (59 adds per step, the level-0 ones 0.25 ms each).  Absent gradients stay None (no zero tensors).
            ctx.set_materialize_grads(False)
            return (y, in1) if in2 is None else (y, in1, in2)
        return y

    @staticmethod
    def backward(ctx, dy, p1=None, p2=None):
        x1, x2, weight, bias = ctx.saved_tensors
        if dy is None:                         # only the passed-through inputs were used downstream
            return p1, p2, None, None, None, None, None, None, None, None
        dy = _f32c(dy)
        w3 = weight if weight.dim() == 3 else weight.unsqueeze(0)
        w3 = _f32c(w3)
        K, cin, cout = w3.shape
        c1 = ctx.c1
        dx1 = dx2 = dw = db = None
        if ctx.needs_input_grad[0]:
            wt = packed_weights.get(weight, True, ctx.mirror, 0, c1)
            acc = _accumulation_target(p1, x1.shape[0], c1, ctx._b2m_tag)
            dx1 = conv_raw(dy, None, wt, K, None, ctx.rb_b, x1.shape[0], c1, out=acc, accumulate=acc is not None)
            if p1 is not None and acc is None:
                dx1 = dx1 + p1
            _own(dx1, ctx.src1)
        if x2 is not None and ctx.needs_input_grad[1]:
            wt = packed_weights.get(weight, True, ctx.mirror, c1, x2.shape[1])
            acc = _accumulation_target(p2, x2.shape[0], x2.shape[1], ctx._b2m_tag)
            dx2 = conv_raw(dy, None, wt, K, None, ctx.rb_b, x2.shape[0], x2.shape[1], out=acc, accumulate=acc is not None)
            if p2 is not None and acc is None:
                dx2 = dx2 + p2
            _own(dx2, ctx.src2)
        if ctx.needs_input_grad[2]:
            # the weight's slot in the model's gradient arena (zeroed once per pass, adopted by autograd as .grad), else a
            # zero-filled tensor of the weight's own shape
            dw = grad_slot(weight)
            if dw is None:
                dw = torch.zeros_like(weight, dtype=torch.float32)
            dw3 = dw if weight.dim() == 3 else dw.unsqueeze(0)
            if wgrad_on_side_stream():
                main, side = torch.cuda.current_stream(dy.device), _side_stream(dy.device)
                side.wait_stream(main)                      # dy, x and the zeroed gradient slot are ready
                with torch.cuda.stream(side):
                    wgrad_raw(x1, dy, ctx.rb_f, K, dw3, 0, c1)
                    if x2 is not None:
                        wgrad_raw(x2, dy, ctx.rb_f, K, dw3, c1, x2.shape[1])
                for t in (dy, x1, x2, dw):                  # the allocator must not hand these out again before the side stream is done
                    if t is not None:
                        t.record_stream(side)
                if not _side['armed']:
                    _side['armed'] = True
                    torch.autograd.Variable._execution_engine.queue_callback(join_side_streams)
            else:
                wgrad_raw(x1, dy, ctx.rb_f, K, dw3, 0, c1)
                if x2 is not None:
                    wgrad_raw(x2, dy, ctx.rb_f, K, dw3, c1, x2.shape[1])
        if bias is not None and ctx.needs_input_grad[3]:
            db = grad_slot(bias)
            if db is not None and db.shape == (1, cout):
                torch.sum(dy, 0, keepdim=True, out=db)
            else:
                db = dy.sum(0, keepdim=True).reshape(bias.shape)
        return dx1, dx2, dw, db, None, None, None, None, None, None


_node_tags = itertools.count(1)


def _node_id(t) -> int:
    """Name of the backward node that will receive the gradient of forward tensor `t`, if that node is one of this package's
    convolutions -- the only nodes that add onto an incoming gradient in place -- else 0 (a leaf, no graph, a torch node).
    The name is a number drawn from a process-wide counter and stored on the node when its forward runs (`ctx._b2m_tag`);
    `id()` of a torch C++ node's Python wrapper is NOT a name: the wrapper is a temporary, and its address can come back as
    the address of a later object."""
    fn = getattr(t, 'grad_fn', None) if t is not None else None
    return getattr(fn, '_b2m_tag', 0) if fn is not None else 0


def _own(t, target: int):
    """Mark a gradient tensor this package produced itself for exactly ONE input, with the backward node it is meant for
    (`target` = _node_id of that input, taken in forward): nobody else holds it, so THAT node's data-gradient kernel may add
    onto it in place."""
    if t is not None:
        t._b2m_own = target
    return t


def _accumulation_target(g, n: int, c: int, node: int):
    """The gradient of a passed-through input, if the data gradient may be added onto it in place: a dense fp32 (n, c)
    tensor that owns its memory AND was produced by one of this package's backward operators for THIS node's output alone
    (`_own` carries the name of the node the gradient was made for; `node` = the convolution's own name, `ctx._b2m_tag`).
    A gradient that went through a torch operator first is not accepted even when it is the very tensor object this
    package produced: AddBackward0 hands ONE tensor to both of its inputs, the mark then names the add node, not this
    one, and adding in place would corrupt the other branch -- such gradients are summed out of place
    (tests/test_gpu_determinism.py::test_torch_add_between_two_convolutions).  The mark is consumed here."""
    if g is None or g.dtype != torch.float32 or tuple(g.shape) != (n, c) or not g.is_contiguous():
        return None
    if g._base is not None and not getattr(g, '_b2m_slab', False):      # a view -- unless it is a piece of the zero slab, made for one tensor
        return None
    mark = g.__dict__.pop('_b2m_own', None)
    if mark is None or mark == 0 or mark != node:
        return None
    return g


def conv_tile_stats() -> bool:
    """B2M_CONV_STATS=0: the BatchNorm statistics come from a pass over the convolution output (b2m_bn_stats) instead of
    the per-tile column sums the convolution kernel leaves behind."""
    return os.environ.get('B2M_CONV_STATS', '1') == '1'


def bn_small_rows() -> int:
    """B2M_BN_SMALL_ROWS: training-mode BatchNorm of maps with at most this many rows runs as ONE launch each way
    (b2m_bn_small_fwd / _bwd; 0 switches it off); under SyncBN as two half-kernels around the statistics exchange."""
    # (round 6: 4096 -> 2048.  The one-launch kernel gives a workgroup four channels of EVERY row -- 16 bytes per row and
    # workgroup, 64 workgroups for 256 channels -- and at 3 k rows takes 25-33 us where statistics-from-tile-sums + apply take
    # 11; launches cost the untraced step nothing (profiles/r06_analysis.md): BatchNorm 8.31 -> 8.01 ms per step.)
    return min(int(os.environ.get('B2M_BN_SMALL_ROWS', '2048')), 16384)       # (B2M_BN_SMALL_MAX_ROWS of the library)


def conv_passthrough() -> bool:
    """B2M_CONV_PASSTHROUGH=0: residual / shortcut branches take the block input itself and autograd adds the two
    gradients of that input with a kernel of its own."""
    return os.environ.get('B2M_CONV_PASSTHROUGH', '1') == '1'


def sparse_conv(x1, x2, weight, bias, rb_f, rb_b, mirror, n_out, collect_stats=False, passthrough=False):
    """collect_stats: the caller will batch-normalise the result in training mode; the per-tile column sums then ride
    along on the returned tensor (attribute `_b2m_tile_stats`, read by batch_norm) when the kernel can provide them.
    passthrough: returns (y, x1, x2) -- x1 / x2 as aliases of the inputs for every OTHER consumer of them, whose gradients
    the data gradient of this convolution is then accumulated onto (see _SparseConv.forward); the inputs themselves when
    no gradient is being recorded."""
    holder = [] if (collect_stats and conv_tile_stats()) else None
    alias = bool(passthrough) and torch.is_grad_enabled() and conv_passthrough() and \
        (x1.requires_grad or (x2 is not None and x2.requires_grad))
    out = _SparseConv.apply(x1, x2, weight, bias, rb_f, rb_b, mirror, n_out, holder, alias)
    y = out[0] if alias else out
    if holder:
        y._b2m_tile_stats = holder[0]
    if not passthrough:
        return y
    if not alias:
        return y, x1, x2
    return (y, out[1], out[2] if x2 is not None else None)


# ----------------------------------------------------------------------------- batch norm
_RED_BLOCKS = 4096

# Collectives of the data-parallel path, counted (and, when bench.py asks, timed with HIP events on the current stream): one
# blocking, latency-bound all-reduce per SyncBN layer and direction, a few large asynchronous ones for the gradient buckets
# (parallel.GradAllReduce).  `collective_stats['timing']` = True makes every SyncBN all-reduce carry an event pair.
collective_stats = {'syncbn': 0, 'grad_buckets': 0, 'bytes': 0, 'timing': False, 'events': [], 'ipc': 0}
ipc_exchange = None        # parallel.IpcExchange of the data-parallel group (B2M_SYNCBN_IPC=1; set by Model.__init__)


def _sync_all_reduce(t, group):
    """A SyncBN statistics exchange: SUM over the ranks, in place, blocking on the current stream -- through the device-side
    mailbox exchange when one is set up for this group (parallel.IpcExchange), else torch.distributed."""
    st = collective_stats
    st['syncbn'] += 1
    st['bytes'] += t.numel() * t.element_size()
    x = ipc_exchange
    if x is not None and x.usable(t) and (group is None or group is x.group or x.group is None):
        st['ipc'] += 1
        reduce = lambda: x.all_reduce_(t)
    else:
        reduce = lambda: dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    if st['timing'] and t.is_cuda:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        reduce()
        e.record()
        st['events'].append((s, e))
    else:
        reduce()



def _sync_group():
    return dist.group.WORLD if (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1) else None


def merge_bn_sums(local_sums: torch.Tensor, local_count: float, group=None):
    """SyncBN statistics exchange: ONE all-reduce of (sum x, sum x^2, row count) over the data-parallel group.
    Device-agnostic (used by the world_size-2 gloo tests).  Replaces the per-layer collectives of
    torch.nn.SyncBatchNorm that ME.MinkowskiSyncBatchNorm.convert_sync_batchnorm installs
    (/root/reference/models/model.py:25).  Returns (global sums, global count) as tensors on the sums' device: the
    count is consumed there (b2m_bn_finalize / b2m_bn_bwd_apply read it through a pointer), no host sync."""
    packed = torch.cat([local_sums, local_sums.new_tensor([float(local_count)])])
    if group is not None:
        _sync_all_reduce(packed, group)
    return packed[:-1], packed[-1:]


class _BatchNorm(torch.autograd.Function):
    """y = BN(x) (+residual) (ReLU).  BatchNorm1d over all rows of the batch
    (/root/reference/models/resnet.py:63,66,73-82); residual add + ReLU fused as the epilogue."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training, momentum, eps, residual, relu, sync,
                count_key=None, tile_stats=None):
        ctx.src_x, ctx.src_res = _node_id(x), _node_id(residual)      # the nodes dx / dres are made for (`_own`)
        x = _f32c(x)
        n, c = x.shape
        dev = x.device
        scale = torch.empty(c, dtype=torch.float32, device=dev)
        shift = torch.empty(c, dtype=torch.float32, device=dev)
        mean = invstd = count_dev = None
        count = float(n)
        small = False
        if training:
            mean = torch.empty(c, dtype=torch.float32, device=dev)
            invstd = torch.empty(c, dtype=torch.float32, device=dev)
            group = _sync_group() if sync else None
            small = n <= bn_small_rows() and c % 4 == 0 and x.stride(0) % 4 == 0
        if small and group is not None:
            # SyncBN on a small map: the one-launch kernel cut in two, the ranks' column sums and row counts meet in between
            # (statistics -> ONE packed all-reduce -> finalize + apply): 2 launches + the collective instead of 5 + it
            if residual is not None:
                residual = _f32c(residual)
            y = torch.empty_like(x)
            stats = torch.empty(2 * c + 1, dtype=torch.float64, device=dev)
            _call('b2m_bn_small_fwd_stats', x.data_ptr(), x.stride(0), n, c, stats.data_ptr())
            _sync_all_reduce(stats, group)
            count_dev = stats[2 * c:]
            _call('b2m_bn_small_fwd_apply', stats.data_ptr(), x.data_ptr(), x.stride(0), n, c, _ptr(gamma), _ptr(beta), eps,
                  momentum, _ptr(running_mean), _ptr(running_var), mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(),
                  shift.data_ptr(), _ptr(residual), residual.stride(0) if residual is not None else 0, 1 if relu else 0,
                  y.data_ptr(), y.stride(0))
        elif small:
            # few rows (deep levels, the heads): statistics + finalize + apply in ONE launch (b2m_bn_small_fwd)
            if residual is not None:
                residual = _f32c(residual)
            y = torch.empty_like(x)
            _call('b2m_bn_small_fwd', x.data_ptr(), x.stride(0), n, c, _ptr(gamma), _ptr(beta), eps, momentum,
                  _ptr(running_mean), _ptr(running_var), mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(),
                  shift.data_ptr(), _ptr(residual), residual.stride(0) if residual is not None else 0, 1 if relu else 0,
                  y.data_ptr(), y.stride(0))
        elif training:
            partial = torch.empty(2 * c * _RED_BLOCKS, dtype=torch.float64, device=dev)
            if tile_stats is not None and (tile_stats[0].shape[2] != c or tile_stats[1] != (n + 63) // 64):
                tile_stats = None                  # not this tensor's sums
            if group is None:
                if tile_stats is not None:         # the producing convolution left the per-tile column sums: no pass over x
                    _call('b2m_bn_tilestats_finalize', tile_stats[0].data_ptr(), tile_stats[1], n, c, partial.data_ptr(),
                          None, _ptr(gamma), _ptr(beta), eps, momentum, _ptr(running_mean), _ptr(running_var),
                          mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(), shift.data_ptr())
                else:
                    _call('b2m_bn_stats_finalize', x.data_ptr(), x.stride(0), n, c, partial.data_ptr(), None, _ptr(gamma),
                          _ptr(beta), eps, momentum, _ptr(running_mean), _ptr(running_var), mean.data_ptr(),
                          invstd.data_ptr(), scale.data_ptr(), shift.data_ptr())
            else:
                # SyncBN: local column sums and the local row count travel in one packed all-reduce; the global count
                # is read by the kernels from device memory (no .item(), no per-level count exchange)
                stats = torch.empty(2 * c + 1, dtype=torch.float64, device=dev)
                if tile_stats is not None:
                    _call('b2m_bn_tilestats', tile_stats[0].data_ptr(), tile_stats[1], c, partial.data_ptr(), stats.data_ptr())
                else:
                    _call('b2m_bn_stats', x.data_ptr(), x.stride(0), n, c, partial.data_ptr(), stats.data_ptr())
                stats[2 * c:].fill_(float(n))
                _sync_all_reduce(stats, group)
                count_dev = stats[2 * c:]
                _call('b2m_bn_finalize', stats.data_ptr(), 0.0, count_dev.data_ptr(), c, _ptr(gamma), _ptr(beta), eps,
                      momentum, _ptr(running_mean), _ptr(running_var), mean.data_ptr(), invstd.data_ptr(),
                      scale.data_ptr(), shift.data_ptr())
        else:
            _call('b2m_bn_finalize', None, 1.0, None, c, _ptr(gamma), _ptr(beta), eps, momentum, running_mean.data_ptr(),
                  running_var.data_ptr(), None, None, scale.data_ptr(), shift.data_ptr())
        if not small:
            if residual is not None:
                residual = _f32c(residual)
            y = torch.empty_like(x)
            _call('b2m_bn_apply', x.data_ptr(), x.stride(0), n, c, scale.data_ptr(), shift.data_ptr(), _ptr(residual),
                  residual.stride(0) if residual is not None else 0, 1 if relu else 0, y.data_ptr(), y.stride(0))
        ctx.training, ctx.relu, ctx.count, ctx.sync, ctx.count_dev = training, relu, count, sync, count_dev
        ctx.small = small
        ctx.has_res = residual is not None
        if training:
            # without a fused residual the ReLU mask is the sign of fmaf(x, scale, shift): the backward recomputes it
            # from x (which it reads anyway) instead of reading y
            ctx.mask_from_x = bool(relu) and residual is None
            ctx.save_for_backward(x, y if (relu and residual is not None) else None, gamma, mean, invstd,
                                  scale if ctx.mask_from_x else None, shift if ctx.mask_from_x else None, beta)
        else:
            ctx.save_for_backward(x, y if relu else None, gamma, scale, None, None, None, beta)
            ctx.eval_mean, ctx.eval_var, ctx.eval_eps = running_mean, running_var, eps
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, gamma, mean, invstd, mscale, mshift, beta = ctx.saved_tensors
        dy = _f32c(dy)
        n, c = x.shape
        dev = x.device
        relu = 1 if ctx.relu else 0
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if (ctx.has_res and ctx.needs_input_grad[8]) else None
        if not ctx.training:
            # eval-mode BN is an affine map: dx = scale * g (mean holds `scale` here)
            g = dy if not relu else dy * (y > 0)
            scale = mean.reshape(1, -1)
            dx = g * scale
            gres = None if dres is None else (g.clone() if g is dy else g)       # (never the incoming tensor itself)
            dgamma = dbeta = None
            if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
                # y = gamma * x_hat + beta with x_hat = (x - running_mean) / sqrt(running_var + eps): the affine parameters
                # have gradients in eval mode too (torch.nn.BatchNorm1d gives them)
                dbeta = g.sum(0)
                invstd = torch.rsqrt(ctx.eval_var + ctx.eval_eps)
                dgamma = ((g * x).sum(0) - dbeta * ctx.eval_mean) * invstd
            return (_own(dx, ctx.src_x), dgamma if ctx.needs_input_grad[1] else None, dbeta if ctx.needs_input_grad[2] else None,
                    None, None, None, None, None, _own(gres, ctx.src_res), None, None, None, None)
        # parameter gradients in buffers of their own: autograd adopts such a tensor as .grad, a view would be cloned
        dbeta, dgamma = grad_slot(beta), grad_slot(gamma)
        if dbeta is None or dgamma is None:
            dbeta = torch.empty(c, dtype=torch.float32, device=dev)
            dgamma = torch.empty(c, dtype=torch.float32, device=dev)
        group = _sync_group() if ctx.sync else None
        if ctx.small and dy.stride(0) % 4 == 0 and group is not None and ctx.count_dev is not None:
            # SyncBN on a small map: reduce (this rank's sums, parameter gradients from them) -> all-reduce -> apply
            xchg = torch.empty(2 * c, dtype=torch.float64, device=dev)
            args = (dy.data_ptr(), dy.stride(0), _ptr(y), y.stride(0) if y is not None else 0, x.data_ptr(), x.stride(0), n, c,
                    mean.data_ptr(), invstd.data_ptr(), _ptr(gamma), relu, _ptr(mscale), _ptr(mshift))
            _call('b2m_bn_small_bwd_phase', 1, *args, dbeta.data_ptr(), dgamma.data_ptr(), None, 0, None, 0, xchg.data_ptr(), None)
            _sync_all_reduce(xchg, group)
            _call('b2m_bn_small_bwd_phase', 2, *args, None, None, dx.data_ptr(), dx.stride(0), _ptr(dres),
                  dres.stride(0) if dres is not None else 0, xchg.data_ptr(), ctx.count_dev.data_ptr())
            return (_own(dx, ctx.src_x), dgamma if ctx.needs_input_grad[1] else None, dbeta if ctx.needs_input_grad[2] else None,
                    None, None, None, None, None, _own(dres, ctx.src_res), None, None, None, None)
        if ctx.small and dy.stride(0) % 4 == 0 and group is None:
            _call('b2m_bn_small_bwd', dy.data_ptr(), dy.stride(0), _ptr(y), y.stride(0) if y is not None else 0, x.data_ptr(),
                  x.stride(0), n, c, mean.data_ptr(), invstd.data_ptr(), _ptr(gamma), relu, _ptr(mscale), _ptr(mshift),
                  dbeta.data_ptr(), dgamma.data_ptr(), dx.data_ptr(), dx.stride(0), _ptr(dres),
                  dres.stride(0) if dres is not None else 0)
            return (_own(dx, ctx.src_x), dgamma if ctx.needs_input_grad[1] else None, dbeta if ctx.needs_input_grad[2] else None,
                    None, None, None, None, None, _own(dres, ctx.src_res), None, None, None, None)
        partial = torch.empty(2 * c * _RED_BLOCKS, dtype=torch.float64, device=dev)
        sums = torch.empty(2 * c, dtype=torch.float64, device=dev)
        _call('b2m_bn_bwd_reduce', dy.data_ptr(), dy.stride(0), _ptr(y), y.stride(0) if y is not None else 0,
              x.data_ptr(), x.stride(0), n, c, mean.data_ptr(), invstd.data_ptr(), relu, _ptr(mscale), _ptr(mshift),
              partial.data_ptr(), sums.data_ptr(), dbeta.data_ptr(), dgamma.data_ptr())
        gsums, count = sums, ctx.count
        group = _sync_group() if ctx.sync else None
        if group is not None:
            gsums = sums.clone()
            _sync_all_reduce(gsums, group)
        _call('b2m_bn_bwd_apply', dy.data_ptr(), dy.stride(0), _ptr(y), y.stride(0) if y is not None else 0,
              x.data_ptr(), x.stride(0), n, c, mean.data_ptr(), invstd.data_ptr(), _ptr(gamma), gsums.data_ptr(),
              count, _ptr(ctx.count_dev), relu, _ptr(mscale), _ptr(mshift), dx.data_ptr(), dx.stride(0), _ptr(dres),
              dres.stride(0) if dres is not None else 0)
        return (_own(dx, ctx.src_x), dgamma if ctx.needs_input_grad[1] else None, dbeta if ctx.needs_input_grad[2] else None,
                None, None, None, None, None, _own(dres, ctx.src_res), None, None, None, None)


def bn_pair() -> bool:
    """B2M_BN_PAIR=0: the two BatchNorms at the end of a BasicBlock with a shortcut convolution run as two launches
    (the shortcut's normalised tensor is stored and read back, two SyncBN exchanges per direction) instead of as one
    paired operator (_BatchNormPair)."""
    return os.environ.get('B2M_BN_PAIR', '1') == '1'


class _BatchNormPair(torch.autograd.Function):
    """y = relu?(BN_a(x_a) + BN_b(x_b)): norm2 + downsample.1 + add + ReLU of a BasicBlock
    (/root/reference/models/resnet.py:73-82) as one operator.  Neither normalisation depends on the other, and both
    backward reductions need only g = dy * (y > 0): one apply, one reduction (sum g, sum g*xhat_a, sum g*xhat_b), one
    backward apply -- and under SyncBN ONE packed all-reduce per direction for the two layers."""

    @staticmethod
    def forward(ctx, xa, ga, ba, rma, rva, xb, gb, bb, rmb, rvb, training, mom_a, eps_a, mom_b, eps_b, relu, sync,
                tile_stats_a=None, tile_stats_b=None):
        ctx.src_a, ctx.src_b = _node_id(xa), _node_id(xb)
        xa, xb = _f32c(xa), _f32c(xb)
        n, c = xa.shape
        assert xb.shape == (n, c)
        dev = xa.device
        f32 = lambda: torch.empty(c, dtype=torch.float32, device=dev)
        sca, sha, scb, shb = f32(), f32(), f32(), f32()
        mean_a = inv_a = mean_b = inv_b = count_dev = None
        if training:
            mean_a, inv_a, mean_b, inv_b = f32(), f32(), f32(), f32()
            partial = torch.empty(2 * c * _RED_BLOCKS, dtype=torch.float64, device=dev)
            group = _sync_group() if sync else None
            sides = ((xa, ga, ba, rma, rva, mom_a, eps_a, mean_a, inv_a, sca, sha, tile_stats_a),
                     (xb, gb, bb, rmb, rvb, mom_b, eps_b, mean_b, inv_b, scb, shb, tile_stats_b))
            if group is None:
                for x, g, b, rm, rv, mom, eps, mean, inv, sc, sh, ts in sides:
                    if ts is not None and (ts[0].shape[2] != c or ts[1] != (n + 63) // 64):
                        ts = None
                    if ts is not None:
                        _call('b2m_bn_tilestats_finalize', ts[0].data_ptr(), ts[1], n, c, partial.data_ptr(), None, _ptr(g),
                              _ptr(b), eps, mom, _ptr(rm), _ptr(rv), mean.data_ptr(), inv.data_ptr(), sc.data_ptr(),
                              sh.data_ptr())
                    else:
                        _call('b2m_bn_stats_finalize', x.data_ptr(), x.stride(0), n, c, partial.data_ptr(), None, _ptr(g),
                              _ptr(b), eps, mom, _ptr(rm), _ptr(rv), mean.data_ptr(), inv.data_ptr(), sc.data_ptr(),
                              sh.data_ptr())
            else:
                # SyncBN: the local column sums of BOTH layers and the local row count in one packed all-reduce
                stats = torch.empty(4 * c + 1, dtype=torch.float64, device=dev)
                for j, (x, g, b, rm, rv, mom, eps, mean, inv, sc, sh, ts) in enumerate(sides):
                    if ts is not None and (ts[0].shape[2] != c or ts[1] != (n + 63) // 64):
                        ts = None
                    dst = stats.data_ptr() + 8 * 2 * c * j
                    if ts is not None:
                        _call('b2m_bn_tilestats', ts[0].data_ptr(), ts[1], c, partial.data_ptr(), dst)
                    else:
                        _call('b2m_bn_stats', x.data_ptr(), x.stride(0), n, c, partial.data_ptr(), dst)
                stats[4 * c:].fill_(float(n))
                _sync_all_reduce(stats, group)
                count_dev = stats[4 * c:]
                for j, (x, g, b, rm, rv, mom, eps, mean, inv, sc, sh, ts) in enumerate(sides):
                    _call('b2m_bn_finalize', stats.data_ptr() + 8 * 2 * c * j, 0.0, count_dev.data_ptr(), c, _ptr(g), _ptr(b),
                          eps, mom, _ptr(rm), _ptr(rv), mean.data_ptr(), inv.data_ptr(), sc.data_ptr(), sh.data_ptr())
        else:
            _call('b2m_bn_finalize', None, 1.0, None, c, _ptr(ga), _ptr(ba), eps_a, mom_a, rma.data_ptr(), rva.data_ptr(),
                  None, None, sca.data_ptr(), sha.data_ptr())
            _call('b2m_bn_finalize', None, 1.0, None, c, _ptr(gb), _ptr(bb), eps_b, mom_b, rmb.data_ptr(), rvb.data_ptr(),
                  None, None, scb.data_ptr(), shb.data_ptr())
        y = torch.empty_like(xa)
        _call('b2m_bn_apply2', xa.data_ptr(), xa.stride(0), xb.data_ptr(), xb.stride(0), n, c, sca.data_ptr(), sha.data_ptr(),
              scb.data_ptr(), shb.data_ptr(), 1 if relu else 0, y.data_ptr(), y.stride(0))
        ctx.training, ctx.relu, ctx.sync, ctx.count, ctx.count_dev = training, relu, sync, float(n), count_dev
        if training:
            ctx.save_for_backward(xa, xb, y, ga, gb, mean_a, inv_a, mean_b, inv_b, ba, bb)
        else:
            ctx.save_for_backward(xa, xb, y, ga, gb, sca, None, scb, None, ba, bb)
            ctx.eval_stats = (rma, rva, eps_a, rmb, rvb, eps_b)
        return y

    @staticmethod
    def backward(ctx, dy):
        xa, xb, y, ga, gb, mean_a, inv_a, mean_b, inv_b, ba, bb = ctx.saved_tensors
        dy = _f32c(dy)
        n, c = xa.shape
        dev = xa.device
        relu = 1 if ctx.relu else 0
        need = ctx.needs_input_grad
        if not ctx.training:
            g = dy if not relu else dy * (y > 0)
            outs = []
            for x, gam, scale, (rm, rv, eps), ig, ib, src in ((xa, ga, mean_a, ctx.eval_stats[0:3], 1, 2, ctx.src_a),
                                                               (xb, gb, mean_b, ctx.eval_stats[3:6], 6, 7, ctx.src_b)):
                dgam = dbet = None
                if need[ig] or need[ib]:
                    dbet = g.sum(0)
                    dgam = ((g * x).sum(0) - dbet * rm) * torch.rsqrt(rv + eps)
                outs.append((_own(g * scale.reshape(1, -1), src), dgam if need[ig] else None, dbet if need[ib] else None))
            (dxa, dga, dba), (dxb, dgb, dbb) = outs
            return (dxa, dga, dba, None, None, dxb, dgb, dbb, None, None) + (None,) * 9
        dxa, dxb = torch.empty_like(xa), torch.empty_like(xb)
        partial = torch.empty(3 * c * _RED_BLOCKS, dtype=torch.float64, device=dev)
        sums = torch.empty(3 * c, dtype=torch.float64, device=dev)

        def slots(beta, gamma):
            db, dg = grad_slot(beta), grad_slot(gamma)
            if db is None or dg is None:
                db = torch.empty(c, dtype=torch.float32, device=dev)
                dg = torch.empty(c, dtype=torch.float32, device=dev)
            return db, dg
        dba, dga = slots(ba, ga)
        dbb, dgb = slots(bb, gb)
        _call('b2m_bn_bwd_reduce2', dy.data_ptr(), dy.stride(0), y.data_ptr(), y.stride(0), xa.data_ptr(), xa.stride(0),
              xb.data_ptr(), xb.stride(0), n, c, mean_a.data_ptr(), inv_a.data_ptr(), mean_b.data_ptr(), inv_b.data_ptr(),
              relu, partial.data_ptr(), sums.data_ptr())
        group = _sync_group() if ctx.sync else None
        gsums = sums
        if group is not None:
            # dx needs the sums over ALL ranks; the parameter gradients are written from this rank's own sums (they are
            # averaged over the ranks with every other gradient afterwards)
            gsums = sums.clone()
            _sync_all_reduce(gsums, group)
        _call('b2m_bn_bwd_apply2', dy.data_ptr(), dy.stride(0), y.data_ptr(), y.stride(0), xa.data_ptr(), xa.stride(0),
              xb.data_ptr(), xb.stride(0), n, c, mean_a.data_ptr(), inv_a.data_ptr(), _ptr(ga), mean_b.data_ptr(),
              inv_b.data_ptr(), _ptr(gb), gsums.data_ptr(), ctx.count, _ptr(ctx.count_dev), relu, dxa.data_ptr(),
              dxa.stride(0), dxb.data_ptr(), dxb.stride(0), dba.data_ptr(), dga.data_ptr(), dbb.data_ptr(), dgb.data_ptr(),
              sums.data_ptr())
        return (_own(dxa, ctx.src_a), dga if need[1] else None, dba if need[2] else None, None, None,
                _own(dxb, ctx.src_b), dgb if need[6] else None, dbb if need[7] else None, None, None) + (None,) * 9


def batch_norm_pair(xa, bn_a, xb, bn_b, training, relu=True, sync=False, count_key=None):
    """bn_a / bn_b: (gamma, beta, running_mean, running_var, momentum, eps) of the two layers.  count_key: the row family
    of the inputs (nn.count_key_of), for observers only."""
    ts_a = getattr(xa, '_b2m_tile_stats', None) if training else None
    ts_b = getattr(xb, '_b2m_tile_stats', None) if training else None
    return _BatchNormPair.apply(xa, bn_a[0], bn_a[1], bn_a[2], bn_a[3], xb, bn_b[0], bn_b[1], bn_b[2], bn_b[3], training,
                                bn_a[4], bn_a[5], bn_b[4], bn_b[5], relu, sync, ts_a, ts_b)


class _BatchNormGroup(torch.autograd.Function):
    """y_j = BN_j(x_j), j = 0 .. m-1: mutually independent training-mode SyncBN layers over the SAME rows -- the layers at
    equal depth of the network's heads (/root/reference/models/detection_net.py:170-194: every head is conv1x1-ReLU-BN x2,
    conv1x1 on the pooled features) -- with ONE packed statistics exchange per direction for all of them: (sum x, sum x^2) of every
    member + the row count forward, (sum g, sum g * xhat) of every member backward.  Four heads: 16 latency-bound all-reduces
    per step become 4.  Same kernels, same arithmetic and the same bits per member as _BatchNorm's SyncBN branch.
    apply(m, group, x_0, gamma_0, beta_0, rm_0, rv_0, momentum_0, eps_0, x_1, ...) -> (y_0, ..., y_{m-1})."""

    @staticmethod
    def forward(ctx, m, group, *flat):
        mem = [flat[7 * j:7 * j + 7] for j in range(m)]
        xs = [_f32c(t[0]) for t in mem]
        n, dev = xs[0].shape[0], xs[0].device
        cs = [x.shape[1] for x in xs]
        assert all(x.shape[0] == n for x in xs)
        off = [0]
        for c in cs:
            off.append(off[-1] + 2 * c)
        stats = torch.empty(off[-1] + 1, dtype=torch.float64, device=dev)
        partial = torch.empty(2 * max(cs) * _RED_BLOCKS, dtype=torch.float64, device=dev)
        for j, x in enumerate(xs):
            _call('b2m_bn_stats', x.data_ptr(), x.stride(0), n, cs[j], partial.data_ptr(), stats.data_ptr() + 8 * off[j])
        stats[off[-1]:].fill_(float(n))
        _sync_all_reduce(stats, group)
        count_dev = stats[off[-1]:]
        ys, saved = [], []
        for j, x in enumerate(xs):
            _, gamma, beta, rm, rv, mom, eps = mem[j]
            c = cs[j]
            f32 = lambda: torch.empty(c, dtype=torch.float32, device=dev)
            mean, invstd, scale, shift = f32(), f32(), f32(), f32()
            _call('b2m_bn_finalize', stats.data_ptr() + 8 * off[j], 0.0, count_dev.data_ptr(), c, _ptr(gamma), _ptr(beta), eps, mom,
                  _ptr(rm), _ptr(rv), mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(), shift.data_ptr())
            y = torch.empty_like(x)
            _call('b2m_bn_apply', x.data_ptr(), x.stride(0), n, c, scale.data_ptr(), shift.data_ptr(), None, 0, 0, y.data_ptr(), y.stride(0))
            ys.append(y)
            saved += [x, gamma, beta, mean, invstd]
        ctx.save_for_backward(*saved)
        ctx.m, ctx.group, ctx.count_dev, ctx.n, ctx.cs, ctx.off = m, group, count_dev, n, cs, off
        ctx.srcs = [_node_id(t[0]) for t in mem]
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        m, n, cs, off = ctx.m, ctx.n, ctx.cs, ctx.off
        sv = ctx.saved_tensors
        dev = sv[0].device
        sums = torch.empty(off[-1], dtype=torch.float64, device=dev)
        partial = torch.empty(2 * max(cs) * _RED_BLOCKS, dtype=torch.float64, device=dev)
        dys = [_f32c(d) for d in dys]
        pg = []
        for j in range(m):
            x, gamma, beta, mean, invstd = sv[5 * j:5 * j + 5]
            c = cs[j]
            dbeta, dgamma = grad_slot(beta), grad_slot(gamma)
            if dbeta is None or dgamma is None:
                dbeta = torch.empty(c, dtype=torch.float32, device=dev)
                dgamma = torch.empty(c, dtype=torch.float32, device=dev)
            # (this rank's sums -> its parameter gradients; the gradient all-reduce averages those over the ranks)
            _call('b2m_bn_bwd_reduce', dys[j].data_ptr(), dys[j].stride(0), None, 0, x.data_ptr(), x.stride(0), n, c, mean.data_ptr(),
                  invstd.data_ptr(), 0, None, None, partial.data_ptr(), sums.data_ptr() + 8 * off[j], dbeta.data_ptr(), dgamma.data_ptr())
            pg.append((dgamma, dbeta))
        gsums = sums.clone()
        _sync_all_reduce(gsums, ctx.group)
        out = [None, None]
        for j in range(m):
            x, gamma, beta, mean, invstd = sv[5 * j:5 * j + 5]
            c = cs[j]
            dx = torch.empty_like(x)
            _call('b2m_bn_bwd_apply', dys[j].data_ptr(), dys[j].stride(0), None, 0, x.data_ptr(), x.stride(0), n, c, mean.data_ptr(),
                  invstd.data_ptr(), _ptr(gamma), gsums.data_ptr() + 8 * off[j], float(n), ctx.count_dev.data_ptr(), 0, None, None,
                  dx.data_ptr(), dx.stride(0), None, 0)
            base = 2 + 7 * j
            out += [_own(dx, ctx.srcs[j]), pg[j][0] if ctx.needs_input_grad[base + 1] else None,
                    pg[j][1] if ctx.needs_input_grad[base + 2] else None, None, None, None, None]
        return tuple(out)


def batch_norm_group(members, group):
    """members: [(x, gamma, beta, running_mean, running_var, momentum, eps), ...] -> [y, ...] (see _BatchNormGroup)."""
    flat = [t for mem in members for t in mem]
    return list(_BatchNormGroup.apply(len(members), group, *flat))


def batch_norm(x, gamma, beta, running_mean, running_var, training, momentum=0.1, eps=1e-5, residual=None,
               relu=False, sync=False, count_key=None):
    # per-tile column sums left by the convolution that produced x (sparse_conv(collect_stats=True))
    tile_stats = getattr(x, '_b2m_tile_stats', None) if training else None
    return _BatchNorm.apply(x, gamma, beta, running_mean, running_var, training, momentum, eps, residual, relu, sync,
                            count_key, tile_stats)


class _ReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _f32c(x)
        y = torch.empty_like(x)
        _call('b2m_relu_fwd', x.data_ptr(), x.numel(), y.data_ptr())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = _f32c(dy)
        dx = torch.empty_like(dy)
        _call('b2m_relu_bwd', dy.data_ptr(), y.data_ptr(), y.numel(), dx.data_ptr())
        return dx


def relu(x):
    return _ReLU.apply(x)


class _SegmentPool(torch.autograd.Function):
    """Segment pooling: out[s] = mean/max of rows with ids == s
    (/root/reference/models/detection_net.py:345-352)."""

    @staticmethod
    def forward(ctx, x, ids, n_seg, mode):
        x = _f32c(x)
        n, c = x.shape
        dev = x.device
        ids = ids.to(device=dev, dtype=torch.int64).contiguous()
        out = torch.empty((n_seg, c), dtype=torch.float32, device=dev)
        counts = torch.empty(max(n_seg, 1), dtype=torch.int32, device=dev)
        argmax = scratch = None
        if mode == 1:
            argmax = torch.empty(max(n_seg * c, 1), dtype=torch.int32, device=dev)
            scratch = torch.empty(max(n_seg * c, 1), dtype=torch.int64, device=dev)
        if mode == 0 and deterministic() and n_seg > 0:
            order = torch.argsort(ids, stable=True)
            seg_start = torch.zeros(n_seg + 1, dtype=torch.int64, device=dev)
            torch.cumsum(torch.bincount(ids, minlength=n_seg), 0, out=seg_start[1:])
            _call('b2m_segment_mean_sorted', x.data_ptr(), x.stride(0), n, c, order.data_ptr(), seg_start.data_ptr(),
                  n_seg, out.data_ptr(), counts.data_ptr())
        else:
            _call('b2m_segment_pool_fwd', x.data_ptr(), x.stride(0), n, c, ids.data_ptr(), n_seg, mode, out.data_ptr(),
                  counts.data_ptr(), _ptr(argmax), _ptr(scratch))
        ctx.save_for_backward(ids, counts, argmax)
        ctx.n, ctx.c, ctx.n_seg, ctx.mode = n, c, n_seg, mode
        return out

    @staticmethod
    def backward(ctx, dout):
        ids, counts, argmax = ctx.saved_tensors
        dout = _f32c(dout)
        dx = torch.empty((ctx.n, ctx.c), dtype=torch.float32, device=dout.device)
        _call('b2m_segment_pool_bwd', dout.data_ptr(), ctx.n, ctx.c, ids.data_ptr(), ctx.n_seg, ctx.mode,
              counts.data_ptr(), _ptr(argmax), dx.data_ptr(), dx.stride(0))
        return dx, None, None, None


class _DetectionLoss(torch.autograd.Function):
    """Box + semantics terms of the training loss, values and gradients in ONE pass over the prediction rows
    (b2m_detection_loss; /root/reference/models/model.py:62-88, 133-176, 194-210).  Returns the 8 result floats
    (include/b2m.h); element 0 is the weighted total and the only one gradients flow through."""

    @staticmethod
    def forward(ctx, off, bnd, sc, sem, gt_off, gt_bnd, loc, fg_u8, gt_sem, n_fg, n_valid, weights, min_bb):
        off, bnd = _f32c(off), _f32c(bnd)
        sc = _f32c(sc) if sc is not None else None
        sem = _f32c(sem) if sem is not None else None
        S, dev = off.shape[0], off.device
        d_off, d_bnd = torch.empty((S, 3), device=dev), torch.empty((S, 3), device=dev)
        d_sc = torch.empty((S, 1), device=dev) if sc is not None else None
        C = sem.shape[1] if sem is not None else 0
        d_sem = torch.empty((S, C), device=dev) if sem is not None else None
        argmax = torch.empty(S, dtype=torch.int64, device=dev) if sem is not None else None
        sums = torch.empty(16, dtype=torch.float64, device=dev)
        result = torch.empty(8, dtype=torch.float32, device=dev)
        _call('b2m_detection_loss', off.data_ptr(), off.stride(0), bnd.data_ptr(), bnd.stride(0), _ptr(sc),
              sc.stride(0) if sc is not None else 0, _ptr(sem), sem.stride(0) if sem is not None else 0, C,
              gt_off.data_ptr(), gt_bnd.data_ptr(), loc.data_ptr(), _ptr(fg_u8), _ptr(gt_sem), S, float(n_fg), _ptr(n_valid),
              weights[0], weights[1], weights[2], weights[3], min_bb, d_off.data_ptr(), d_bnd.data_ptr(), _ptr(d_sc),
              _ptr(d_sem), _ptr(argmax), sums.data_ptr(), result.data_ptr())
        ctx.grads = (d_off, d_bnd, d_sc, d_sem)
        ctx.argmax = argmax
        return result

    @staticmethod
    def backward(ctx, g):
        g0 = g[0]
        live = [d for d in ctx.grads if d is not None]
        scaled = iter(torch._foreach_mul(live, g0))         # one launch for the four heads' gradients
        out = [None if d is None else next(scaled) for d in ctx.grads]
        return (out[0], out[1], out[2], out[3]) + (None,) * 9


def detection_loss(off, bnd, sc, sem, gt_off, gt_bnd, loc, fg_u8, gt_sem, n_fg, n_valid, weights, min_bb):
    """-> (result (8,) float tensor: total, offset_loss, bounds_loss, bb_score_loss, bb_target_scores, bb_scores_correlation,
    semantics_loss, semantics_acc;  predicted class per row or None)."""
    fn = _DetectionLoss
    res = fn.apply(off, bnd, sc, sem, gt_off, gt_bnd, loc, fg_u8, gt_sem, n_fg, n_valid, weights, min_bb)
    node = res.grad_fn
    argmax = None
    if sem is not None:
        # (the argmax rides on the context of the node that made `res`; without autograd the kernel's buffer is re-made)
        argmax = node.argmax if node is not None and hasattr(node, 'argmax') else torch.argmax(sem.detach(), 1)
    return res, argmax


def fused_loss_enabled() -> bool:
    """B2M_FUSED_LOSS=0: the loss terms are evaluated term by term with torch (the reference's formulas as written)."""
    return os.environ.get('B2M_FUSED_LOSS', '1') == '1'


def segment_pool(x, ids, n_seg, mode='avg'):
    return _SegmentPool.apply(x, ids, n_seg, 0 if mode == 'avg' else 1)
