"""Half-precision TRAINING of the trunk (BASELINE configs[4]: "ARKitScenes ... fp16 features on CDNA4";
/root/reference/configs/arkitscenes.txt -- a build extension, the reference trains in fp32).

Between the 5x5x5 stem (fp32) and the segment pooling (fp32) every activation and every activation gradient of the levels at
tensor strides 1 ... 8 -- where the bytes are -- lives in HBM as IEEE binary16; the levels below (a few thousand rows at most)
stay fp32, as do the weights (master copies, the optimizer's); every accumulation is fp32 / fp64:

* forward and data gradient of a layer: `b2m_conv_fwd_h` (conv_fwd_flow_kernel<.., F16>: f16 MFMA, fp32 accumulators, one
  rounding to half on the way out) -- the data gradient with the half image of the transposed (and, for the stride-1 maps,
  mirrored) weights;
* weight gradient: `b2m_conv_wgrad_h` (half operands converted on load, fp32 MFMA, fp32 dW);
* BatchNorm: `b2m_bn_stats_finalize_h` -> `b2m_bn_apply_h`, backward `b2m_bn_bwd_reduce_h` -> `b2m_bn_bwd_apply_h`
  (fp64 statistics, the ReLU mask is the sign of the stored half output);
* loss scaling: the gradient is multiplied by `loss_scale` where it enters the half region (`to_float`), every parameter gradient
  the region produces is multiplied by 1 / loss_scale by the operator that produced it, and the gradient that leaves the region
  towards the stem (`to_half`) likewise -- outside the region nothing is scaled.

Turned on by `SelectionNet.half_training = True` (or `cfg.half_training`); under data parallelism the BatchNorm takes its SyncBN form
(one packed exchange per direction, as the fp32 operator) and the fp32 parameter gradients travel as always.  The layers
(`nn.MinkowskiConvolution` ...) route here by the dtype of their input.  Semantics per operator: those of `functional._SparseConv`
/ `functional._BatchNorm` (/root/reference/models/resnet.py:61-83, detection_net.py:37-135)."""
from __future__ import annotations

import os
import weakref

import torch

from . import _lib
from . import functional as F_
from .grad_arena import grad_slot

_call = _lib.call
_ptr = _lib.ptr

loss_scale = [1024.0]        # set by SelectionNet.forward from cfg.half_loss_scale (a power of two: scaling is exact)


def _hc(t):
    """half, unit column stride (the kernels take a row pitch)."""
    assert t.dtype == torch.float16
    return t if t.stride(1) == 1 else t.contiguous()


# ---- the half images of a training step: forward image and transposed image(s) of every layer, repacked in ONE launch per pass
class _HalfImages:
    """Every (weight, form) a half layer asks for is registered on first use with a persistent image; `begin_pass()` (SelectionNet.
    forward with half_training) repacks all of them from the current weights with one b2m_weight_pack_h_run launch on the side
    stream -- the first layer that asks for an image makes its stream wait -- and opens a new pass; lookups during that forward and
    its backward are hits: 1 launch per step instead of ~90.  The rules are those of functional._PackedWeights: begin_pass repacks
    unconditionally (a fused optimizer bumps no version counter); outside a pass opened for the current weights an image is reused
    only if the tensor's version counter and address are unchanged, else repacked on the spot; entries die with their tensors."""

    def __init__(self):
        self.entries = {}          # key -> [weakref(weight), (K, cin, cout, c1, transposed, mirror, s0, sc), image, version, data_ptr, pass_id]
        self.plan = None           # (device table, n, blocks, keys)
        self.dirty = True
        self.pass_id = 0
        self.pending = None        # event of the side-stream launch of this pass
        self.env_epoch = _lib.env_epoch[0]

    @staticmethod
    def _pack_one(w3, d, image):
        K, cin, cout, c1, transposed, mirror, s0, sc = d
        if transposed:
            _call('b2m_weight_pack_h_t', w3.data_ptr(), K, cin, cout, 1 if mirror else 0, s0, sc, image.data_ptr())
        else:
            _call('b2m_weight_pack_h', w3.data_ptr(), cout, K, c1, cin - c1, cout, image.data_ptr())

    def get(self, weight, c1=0, transposed=False, mirror=False, s0=0, sc=0):
        if self.env_epoch != _lib.env_epoch[0]:          # the library's switches changed (the images' strip width is one): start over
            self.__init__()
        w3 = weight.detach()
        w3 = w3 if w3.dim() == 3 else w3.unsqueeze(0)
        assert w3.dtype == torch.float32 and w3.is_contiguous()
        K, cin, cout = w3.shape
        key = (id(weight), int(c1), bool(transposed), bool(mirror), int(s0), int(sc))
        e = self.entries.get(key)
        if e is not None and e[0]() is weight and e[4] == weight.data_ptr():
            if self.pending is not None:
                torch.cuda.current_stream(weight.device).wait_event(self.pending)
                self.pending = None
            if e[5] != self.pass_id or e[3] != weight._version:      # not packed in this pass / changed since
                self._pack_one(w3, e[1], e[2])
                e[3], e[5] = weight._version, self.pass_id
            return e[2]
        d = (K, cin, cout, int(c1), bool(transposed), bool(mirror), int(s0), int(sc))
        lib = _lib.load()
        n = lib.b2m_weight_pack_h_size(K, cout, 0, sc) if transposed else lib.b2m_weight_pack_h_size(K, c1, cin - c1, cout)
        image = torch.empty(n, dtype=torch.float16, device=w3.device)
        self._pack_one(w3, d, image)
        self.entries[key] = [weakref.ref(weight), d, image, weight._version, weight.data_ptr(), self.pass_id]
        self.dirty = True
        return image

    def _build_plan(self):
        import ctypes as C
        import numpy as np
        for k in [k for k, e in self.entries.items() if e[0]() is None or e[0]().data_ptr() != e[4]]:
            del self.entries[k]
        self.dirty = False
        keys = list(self.entries)
        if not keys:
            self.plan = None
            return
        lib = _lib.load()
        es = [self.entries[k] for k in keys]
        i64 = lambda v: np.ascontiguousarray(v, dtype=np.int64)
        i32 = lambda v: np.ascontiguousarray(v, dtype=np.int32)
        cols = [i64([e[4] for e in es]), i64([e[2].data_ptr() for e in es])] + [i32([int(e[1][j]) for e in es]) for j in range(8)]
        host = np.zeros(len(keys) * lib.b2m_weight_pack_h_plan_size(), np.uint8)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        blocks = lib.b2m_weight_pack_h_plan(len(keys), *[p(c) for c in cols], p(host))
        if blocks < 0:
            raise _lib.B2MError('b2m_weight_pack_h_plan failed: ' + lib.b2m_last_error().decode())
        self.plan = (torch.from_numpy(host).to(es[0][2].device), len(keys), int(blocks), keys)

    def begin_pass(self):
        if self.env_epoch != _lib.env_epoch[0]:
            self.__init__()
        self.pass_id += 1
        self.pending = None
        if not self.entries:
            return
        if self.dirty or any(e[0]() is None or e[0]().data_ptr() != e[4] for e in self.entries.values()):
            self._build_plan()
        if self.plan is None:
            return
        table, n, blocks, keys = self.plan
        dev = table.device
        if F_.pack_on_side_stream() and F_.wgrad_on_side_stream():
            main, side = torch.cuda.current_stream(dev), F_._side_stream(dev)
            side.wait_stream(main)           # the optimizer's update of the weights
            with torch.cuda.stream(side):
                _call('b2m_weight_pack_h_run', table.data_ptr(), n, blocks)
                self.pending = torch.cuda.Event()
                self.pending.record(side)
        else:
            _call('b2m_weight_pack_h_run', table.data_ptr(), n, blocks)
        for k in keys:
            e = self.entries[k]
            e[3], e[5] = e[0]()._version, self.pass_id


images = _HalfImages()


def weight_pack_ht(weight, mirror: bool, s0: int, sc: int):
    """Half image of W'[k] = W[k or K-1-k][s0:s0+sc, :]^T -- the operand with which b2m_conv_fwd_h computes the gradient w.r.t.
    input channels [s0, s0 + sc) of a layer with weights W (K, cin, cout)."""
    return images.get(weight, 0, True, mirror, s0, sc)


def conv_tile_stats() -> bool:
    """B2M_CONV_STATS_H=0: the half BatchNorm reads the convolution's output for its statistics (b2m_bn_stats_h) instead of taking the
    per-tile column sums the convolution kernel leaves behind (b2m_conv_fwd_h_stats)."""
    return os.environ.get('B2M_CONV_STATS_H', '1') == '1' and F_.conv_tile_stats()


def _conv_h(x1, x2, image, K, rb, n_out, cout, tile_stats=None, res=None):
    """tile_stats: a list that receives (tensor [ntiles, 2, cout] fp64, ntiles) -- the per-tile column sums of the output as stored.
    res: a half (n_out, cout) tensor added to the result in the kernel's epilogue (fp32 sum, one rounding)."""
    c1 = x1.shape[1]
    c2 = x2.shape[1] if x2 is not None else 0
    out = torch.empty((n_out, cout), dtype=torch.float16, device=x1.device)
    if n_out == 0:
        return out
    if tile_stats is not None:
        ntiles = (n_out + 63) // 64
        ts = torch.empty((ntiles, 2, cout), dtype=torch.float64, device=x1.device)
        _call('b2m_conv_fwd_h_stats', x1.data_ptr(), x1.stride(0), c1, _ptr(x2), x2.stride(0) if x2 is not None else 0, c2, x1.shape[0],
              image.data_ptr(), K, rb.rb_in.data_ptr(), rb.rb_out.data_ptr(), rb.rb_cnt.data_ptr(), n_out, out.data_ptr(), out.stride(0),
              cout, ts.data_ptr(), meta={'half': True})
        tile_stats.append((ts, ntiles))
        return out
    _call('b2m_conv_fwd_h', x1.data_ptr(), x1.stride(0), c1, _ptr(x2), x2.stride(0) if x2 is not None else 0, c2, x1.shape[0],
          image.data_ptr(), K, rb.rb_in.data_ptr(), rb.rb_out.data_ptr(), rb.rb_cnt.data_ptr(), n_out, out.data_ptr(), out.stride(0),
          cout, None, None, _ptr(res), res.stride(0) if res is not None else 0, 0, meta={'half': True})
    return out


def _wgrad_h(x, dy, rb, K, dw3, ci0, inv):
    """dw3[:, ci0:ci0+cin, :] += inv * sum_pairs x[in]^T dy[out] with half x / dy (b2m_conv_wgrad_h); a transposed map walks its DOWN
    rulebook with the row roles exchanged, as functional.wgrad_raw does."""
    cin_total, cout = dw3.shape[1], dw3.shape[2]
    cin = x.shape[1]
    n_out = dy.shape[0]
    dst = dw3.data_ptr() + 4 * ci0 * cout
    sc = rb.scatter
    if sc is not None and n_out > 0 and x.shape[0] > 0 and F_.wgrad_up_over_down_map():
        _call('b2m_conv_wgrad_h', x.data_ptr(), x.stride(0), cin, n_out, dy.data_ptr(), dy.stride(0), cout, sc.rb_in.data_ptr(),
              sc.rb_out.data_ptr(), sc.rb_cnt.data_ptr(), x.shape[0], K, dst, cout, cin_total * cout, 1, inv)
        return
    _call('b2m_conv_wgrad_h', x.data_ptr(), x.stride(0), cin, x.shape[0], dy.data_ptr(), dy.stride(0), cout, rb.rb_in.data_ptr(),
          rb.rb_out.data_ptr(), rb.rb_cnt.data_ptr(), n_out, K, dst, cout, cin_total * cout, 0, inv)


class _ConvH(torch.autograd.Function):
    """Y = sum_k [x1|x2][in_k] W[k], half in / out (functional._SparseConv without bias, pass-through and tile statistics)."""

    @staticmethod
    def forward(ctx, x1, x2, weight, rb_f, rb_b, mirror, n_out, holder=None, passthrough=False):
        in1, in2 = x1, x2
        x1 = _hc(x1)
        x2 = _hc(x2) if x2 is not None else None
        w3 = weight if weight.dim() == 3 else weight.unsqueeze(0)
        K, cin, cout = w3.shape
        c1 = x1.shape[1]
        c2 = x2.shape[1] if x2 is not None else 0
        assert c1 + c2 == cin and rb_f.K == K and rb_f.n_out == n_out
        y = _conv_h(x1, x2, images.get(weight, c1), K, rb_f, n_out, cout, tile_stats=holder)
        ctx.save_for_backward(x1, x2, weight)
        ctx.rb_f, ctx.rb_b, ctx.mirror = rb_f, rb_b, mirror
        if passthrough:
            # The inputs come back as further outputs, as in functional._SparseConv: whoever else consumes them (the residual
            # branch of a BasicBlock) takes THESE, so their gradient arrives here in one call with dy and the data gradient takes
            # it as the residual of its epilogue (fp32 sum, one rounding, a new tensor: nothing is modified in place) -- instead
            # of an add kernel of autograd's behind the convolution (22 per step).
            ctx.set_materialize_grads(False)
            return (y, in1) if in2 is None else (y, in1, in2)
        return y

    @staticmethod
    def backward(ctx, dy, p1=None, p2=None):
        x1, x2, weight = ctx.saved_tensors
        if dy is None:                         # only the passed-through inputs were used downstream
            return p1, p2, None, None, None, None, None, None, None
        dy = _hc(dy)
        w3 = weight if weight.dim() == 3 else weight.unsqueeze(0)
        K, cin, cout = w3.shape
        c1 = x1.shape[1]
        inv = 1.0 / loss_scale[0]
        dx1 = dx2 = dw = None
        def fusable(p, n, c):
            return (p is not None and p.dtype == torch.float16 and tuple(p.shape) == (n, c) and p.stride(1) == 1 and
                    p.stride(0) % 4 == 0 and p.data_ptr() % 8 == 0 and n > 0)
        if ctx.needs_input_grad[0]:
            f = fusable(p1, x1.shape[0], c1)
            dx1 = _conv_h(dy, None, weight_pack_ht(weight, ctx.mirror, 0, c1), K, ctx.rb_b, x1.shape[0], c1, res=p1 if f else None)
            if p1 is not None and not f:
                dx1 = dx1 + p1
        elif p1 is not None:
            dx1 = p1
        if x2 is not None and ctx.needs_input_grad[1]:
            f = fusable(p2, x2.shape[0], x2.shape[1])
            dx2 = _conv_h(dy, None, weight_pack_ht(weight, ctx.mirror, c1, x2.shape[1]), K, ctx.rb_b, x2.shape[0], x2.shape[1],
                          res=p2 if f else None)
            if p2 is not None and not f:
                dx2 = dx2 + p2
        elif p2 is not None:
            dx2 = p2
        if ctx.needs_input_grad[2]:
            dw = grad_slot(weight)
            if dw is None:
                dw = torch.zeros_like(weight, dtype=torch.float32)
            dw3 = dw if weight.dim() == 3 else dw.unsqueeze(0)

            def run():
                _wgrad_h(x1, dy, ctx.rb_f, K, dw3, 0, inv)      # (inv: the kernel un-scales the block on its way into dW)
                if x2 is not None:
                    _wgrad_h(x2, dy, ctx.rb_f, K, dw3, c1, inv)
            if F_.wgrad_on_side_stream():
                main, side = torch.cuda.current_stream(dy.device), F_._side_stream(dy.device)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    run()
                for t in (dy, x1, x2, dw):
                    if t is not None:
                        t.record_stream(side)
                if not F_._side['armed']:
                    F_._side['armed'] = True
                    torch.autograd.Variable._execution_engine.queue_callback(F_.join_side_streams)
            else:
                run()
        return dx1, dx2, dw, None, None, None, None, None, None


def conv(x1, x2, weight, rb_f, rb_b, mirror, n_out, collect_stats=False, passthrough=False):
    """collect_stats: the caller batch-normalises the result in training mode; the per-tile column sums then ride along on the
    returned tensor (attribute `_b2m_tile_stats`, read by batch_norm), as functional.sparse_conv does for the fp32 layers.
    passthrough: returns (y, x1, x2) with x1 / x2 aliases of the inputs for every OTHER consumer of them (functional.sparse_conv);
    the inputs themselves when no gradient is being recorded."""
    holder = [] if (collect_stats and conv_tile_stats()) else None
    alias = bool(passthrough) and torch.is_grad_enabled() and F_.conv_passthrough() and \
        (x1.requires_grad or (x2 is not None and x2.requires_grad))
    out = _ConvH.apply(x1, x2, weight, rb_f, rb_b, mirror, n_out, holder, alias)
    y = out[0] if alias else out
    if holder:
        y._b2m_tile_stats = holder[0]
    if not passthrough:
        return y
    if not alias:
        return y, x1, x2
    return (y, out[1], out[2] if x2 is not None else None)


_ws = {}


def _workspace(c, device):
    """Partial-sum buffer of the two-stage column reductions (2c x 4096 doubles), kept per device: launches on one stream run in
    order, so consecutive layers can share it."""
    need = 2 * c * F_._RED_BLOCKS
    w = _ws.get(device)
    if w is None or w.numel() < need:
        w = torch.empty(max(need, 2 * 256 * F_._RED_BLOCKS), dtype=torch.float64, device=device)
        _ws[device] = w
    return w


class _BatchNormH(torch.autograd.Function):
    """y = [relu](BN(x) [+ residual]) in training mode, half in / out (functional._BatchNorm's large-map branches).  sync: SyncBN
    (/root/reference/models/model.py:25) -- the ranks' column sums and row counts meet in one packed fp64 all-reduce per direction,
    exactly as in functional._BatchNorm: (sum x, sum x^2, n) forward, (sum g, sum g xhat) backward; the parameter gradients stay this
    rank's own sums (the gradient all-reduce averages them)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps, residual, relu, sync=False, tile_stats=None):
        x = _hc(x)
        n, c = x.shape
        if tile_stats is not None and (tile_stats[0].shape[2] != c or tile_stats[1] != (n + 63) // 64):
            tile_stats = None                  # not this tensor's sums
        dev = x.device
        group = F_._sync_group() if sync else None
        if n <= 1 and group is None:
            raise ValueError('Expected more than 1 value per channel when training, got input size %s' % (tuple(x.shape),))
        residual = _hc(residual) if residual is not None else None
        f32 = lambda: torch.empty(c, dtype=torch.float32, device=dev)
        mean, invstd, scale, shift = f32(), f32(), f32(), f32()
        count_dev = None
        if group is None:
            if tile_stats is not None:         # the producing convolution left the per-tile column sums: no pass over x
                _call('b2m_bn_tilestats_finalize', tile_stats[0].data_ptr(), tile_stats[1], n, c, _workspace(c, dev).data_ptr(), None,
                      _ptr(gamma), _ptr(beta), eps, momentum, _ptr(running_mean), _ptr(running_var), mean.data_ptr(), invstd.data_ptr(),
                      scale.data_ptr(), shift.data_ptr())
            else:
                _call('b2m_bn_stats_finalize_h', x.data_ptr(), x.stride(0), n, c, _workspace(c, dev).data_ptr(), _ptr(gamma), _ptr(beta),
                      eps, momentum, _ptr(running_mean), _ptr(running_var), mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(),
                      shift.data_ptr())
        else:
            stats = torch.empty(2 * c + 1, dtype=torch.float64, device=dev)
            if n > 0 and tile_stats is not None:
                _call('b2m_bn_tilestats', tile_stats[0].data_ptr(), tile_stats[1], c, _workspace(c, dev).data_ptr(), stats.data_ptr())
            elif n > 0:
                _call('b2m_bn_stats_h', x.data_ptr(), x.stride(0), n, c, _workspace(c, dev).data_ptr(), stats.data_ptr())
            else:
                stats.zero_()
            stats[2 * c:].fill_(float(n))
            F_._sync_all_reduce(stats, group)
            count_dev = stats[2 * c:]
            _call('b2m_bn_finalize', stats.data_ptr(), 0.0, count_dev.data_ptr(), c, _ptr(gamma), _ptr(beta), eps, momentum,
                  _ptr(running_mean), _ptr(running_var), mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(), shift.data_ptr())
        y = torch.empty_like(x)
        if n > 0:
            _call('b2m_bn_apply_h', x.data_ptr(), x.stride(0), n, c, scale.data_ptr(), shift.data_ptr(), _ptr(residual),
                  residual.stride(0) if residual is not None else 0, 1 if relu else 0, y.data_ptr(), y.stride(0))
        # without a fused residual the ReLU mask is the sign of fmaf(x, scale, shift): the backward recomputes it from x (which it
        # reads anyway) instead of reading y, and y is not kept (functional._BatchNorm does the same)
        ctx.mask_from_x = bool(relu) and residual is None
        ctx.save_for_backward(x, y if (relu and residual is not None) else None, gamma, beta, mean, invstd,
                              scale if ctx.mask_from_x else None, shift if ctx.mask_from_x else None)
        ctx.relu, ctx.has_res, ctx.sync, ctx.count_dev = bool(relu), residual is not None, bool(sync), count_dev
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, gamma, beta, mean, invstd, mscale, mshift = ctx.saved_tensors
        dy = _hc(dy)
        n, c = x.shape
        dev = x.device
        relu = 1 if ctx.relu else 0
        dbeta, dgamma = grad_slot(beta), grad_slot(gamma)
        if dbeta is None or dgamma is None:
            dbeta = torch.empty(c, dtype=torch.float32, device=dev)
            dgamma = torch.empty(c, dtype=torch.float32, device=dev)
        sums = torch.empty(2 * c, dtype=torch.float64, device=dev)
        if n > 0:
            _call('b2m_bn_bwd_reduce_h', dy.data_ptr(), dy.stride(0), _ptr(y), y.stride(0) if y is not None else 0, x.data_ptr(),
                  x.stride(0), n, c, mean.data_ptr(), invstd.data_ptr(), relu, _ptr(mscale), _ptr(mshift), _workspace(c, dev).data_ptr(),
                  sums.data_ptr(), dbeta.data_ptr(), dgamma.data_ptr(), 1.0 / loss_scale[0])       # (the sums are of the SCALED gradient)
        else:
            sums.zero_(); dbeta.zero_(); dgamma.zero_()
        group = F_._sync_group() if ctx.sync else None
        if group is not None:
            F_._sync_all_reduce(sums, group)          # (dbeta / dgamma were taken from this rank's sums by the kernel above)
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if (ctx.has_res and ctx.needs_input_grad[7]) else None
        if n > 0:
            _call('b2m_bn_bwd_apply_h', dy.data_ptr(), dy.stride(0), _ptr(y), y.stride(0) if y is not None else 0, x.data_ptr(),
                  x.stride(0), n, c, mean.data_ptr(), invstd.data_ptr(), _ptr(gamma), sums.data_ptr(), float(n),
                  _ptr(ctx.count_dev) if group is not None else None, relu, _ptr(mscale), _ptr(mshift),
                  dx.data_ptr(), dx.stride(0), _ptr(dres), dres.stride(0) if dres is not None else 0)
        return (dx, dgamma if ctx.needs_input_grad[1] else None, dbeta if ctx.needs_input_grad[2] else None, None, None, None,
                None, dres, None, None, None)


def batch_norm(x, gamma, beta, running_mean, running_var, momentum, eps, residual=None, relu=False, sync=False):
    tile_stats = getattr(x, '_b2m_tile_stats', None)       # left by conv(collect_stats=True)
    return _BatchNormH.apply(x, gamma, beta, running_mean, running_var, momentum, eps, residual, relu, sync, tile_stats)


class _ToHalf(torch.autograd.Function):
    """fp32 -> half where a half region begins (behind the stem; behind the fp32 deep levels); the gradient leaves the region
    here: un-scaled."""

    @staticmethod
    def forward(ctx, x):
        return x.half()

    @staticmethod
    def backward(ctx, g):
        return g.float() * (1.0 / loss_scale[0])


class _ToFloat(torch.autograd.Function):
    """half -> fp32 where a half region ends (in front of the segment pooling; in front of the fp32 deep levels); the gradient
    enters the region here: scaled."""

    @staticmethod
    def forward(ctx, x):
        return x.float()

    @staticmethod
    def backward(ctx, g):
        return (g * loss_scale[0]).half()


def to_half(x):
    return _ToHalf.apply(x)


def to_float(x):
    return _ToFloat.apply(x)
