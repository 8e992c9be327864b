"""Sparse layers with the MinkowskiEngine names and parameter layout the reference uses.

Import as ``import box2mask_amd.nn as ME`` and the model code reads like
/root/reference/models/detection_net.py / resnet.py.  State-dict keys and shapes follow SURVEY.md
§8(b): ``<conv>.kernel`` (K,Cin,Cout) — 2-D (Cin,Cout) for 1x1 —, ``<conv>.bias`` (1,Cout),
``<bn>.bn.{weight,bias,running_mean,running_var,num_batches_tracked}``  [shapes ME-mem].
"""
from __future__ import annotations

import math

import torch
from torch import nn

from . import functional as F_
from .sparse import CoordinateManager, SparseTensor  # noqa: F401  (re-exported, ME.SparseTensor)


class CatTensor(SparseTensor):
    """Result of :func:`cat`: two feature matrices on the same coordinate key, concatenated lazily.
    Convolutions read both sources directly (no (N, C1+C2) copy is materialised)."""

    def __init__(self, a: SparseTensor, b: SparseTensor):
        assert a.manager is b.manager and a.level == b.level, 'ME.cat needs equal coordinate keys'
        self.manager, self.level = a.manager, a.level
        self.parts = (a.F, b.F)
        self._F = None

    @property
    def F(self):
        if self._F is None:
            self._F = torch.cat(self.parts, 1)
        return self._F


def cat(a: SparseTensor, b: SparseTensor) -> SparseTensor:
    """ME.cat (/root/reference/models/detection_net.py:286-336): channel concat [a, b]."""
    return CatTensor(a, b)


def _sources(x: SparseTensor):
    if isinstance(x, CatTensor):
        return x.parts
    return x.F, None


def _like(x: SparseTensor, f1, f2=None) -> SparseTensor:
    """x with its feature matrix (or its two lazily concatenated parts) replaced."""
    if isinstance(x, CatTensor):
        y = CatTensor.__new__(CatTensor)
        y.manager, y.level, y.parts, y._F = x.manager, x.level, (f1, f2), None
        return y
    return x.new(f1)


class _ConvBase(nn.Module):
    transposed = False

    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False, dimension=3,
                 expand_coordinates=False):
        super().__init__()
        assert dimension == 3 and dilation == 1 and not expand_coordinates
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = kernel_size, stride
        self.kernel_volume = kernel_size ** 3
        if self.kernel_volume == 1:
            self.kernel = nn.Parameter(torch.empty(in_channels, out_channels))
        else:
            self.kernel = nn.Parameter(torch.empty(self.kernel_volume, in_channels, out_channels))
        self.bias = nn.Parameter(torch.empty(1, out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):          # [ME-mem] default init: uniform(+-1/sqrt(fan))
        n = (self.out_channels if self.transposed else self.in_channels) * self.kernel_volume
        stdv = 1.0 / math.sqrt(n)
        with torch.no_grad():
            self.kernel.uniform_(-stdv, stdv)
            if self.bias is not None:
                self.bias.uniform_(-stdv, stdv)

    def extra_repr(self):
        return 'in=%d, out=%d, kernel_size=%d, stride=%d' % (self.in_channels, self.out_channels, self.kernel_size,
                                                             self.stride)


class MinkowskiConvolution(_ConvBase):
    """[ME-mem] stride 1 (odd kernel, centred) or kernel 2 / stride 2."""

    def forward(self, x: SparseTensor, passthrough: bool = False, fuse=None):
        """passthrough: returns (y, x') with x' an alias of x for every other consumer of x (a block's residual /
        shortcut branch): this layer's data gradient is then accumulated onto their gradient by the kernel instead of
        by an add of autograd's (functional._SparseConv).
        fuse = (MinkowskiBatchNorm, residual features or None, relu): inference only (fusable()) -- the layer and the
        eval-mode BatchNorm (+ residual) (+ ReLU) behind it as one launch (functional.conv_affine)."""
        m, l = x.manager, x.level
        x1, x2 = _sources(x)
        if fuse is not None:
            bn, residual, relu = fuse
            if passthrough:
                raise ValueError('fuse= and passthrough= exclude each other (inference has no gradients to pass through)')
            scale, shift = bn.eval_affine()
            if self.bias is not None:       # BN(conv + b) = conv * scale + (shift + scale * b): the bias folds into the shift
                shift = shift + scale * self.bias.detach().reshape(-1).to(scale.dtype)
            if self.kernel_volume == 1:
                # (the half kernel walks rulebooks only: a 1x1 layer brings the identity map of its level)
                rb, n_out, level = (m.rulebook_identity(l) if x1.dtype == torch.float16 else None), x1.shape[0], None
            elif self.stride == 1:
                rb = m.rulebook_same(l, self.kernel_size); n_out, level = rb.n_out, None
            else:
                rb = m.rulebook_down(l); n_out, level = rb.n_out, l + 1
            return x.new(F_.conv_affine(x1, x2, self.kernel, rb, n_out, scale, shift, residual, relu), level=level)

        def result(out, level=None):
            if not passthrough:
                return x.new(out, level=level)
            y, a1, a2 = out
            return x.new(y, level=level), _like(x, a1, a2)
        if x1.dtype == torch.float16:
            # half-precision training (half_train.py): the same layer on binary16 activations; the passed-through gradient is the
            # residual of the data gradient's epilogue
            from . import half_train as HT
            assert self.bias is None, 'the half trunk has no biased convolutions'
            if self.kernel_volume == 1:
                rb_f = rb_b = m.rulebook_identity(l); mirror, level = False, None
            elif self.stride == 1:
                rb_f = rb_b = m.rulebook_same(l, self.kernel_size); mirror, level = True, None
            else:
                rb_f, rb_b, mirror, level = m.rulebook_down(l), m.rulebook_up(l), False, l + 1
            return result(HT.conv(x1, x2, self.kernel, rb_f, rb_b, mirror, rb_f.n_out, collect_stats=self.training,
                                  passthrough=passthrough), level)
        if self.kernel_volume == 1:
            assert self.stride == 1
            return result(F_.sparse_conv(x1, x2, self.kernel, self.bias, None, None, False, x1.shape[0],
                                         passthrough=passthrough))
        # every trunk convolution feeds a BatchNorm (resnet.py:61-66, detection_net.py:37-135): in training mode its
        # kernel also leaves the column sums the normalisation needs
        stats = self.training and self.bias is None
        if self.stride == 1:
            rb = m.rulebook_same(l, self.kernel_size)
            return result(F_.sparse_conv(x1, x2, self.kernel, self.bias, rb, rb, True, rb.n_out, collect_stats=stats,
                                         passthrough=passthrough))
        assert self.stride == 2 and self.kernel_size == 2, 'only k2s2 strided convolutions are on the path'
        rb_f, rb_b = m.rulebook_down(l), m.rulebook_up(l)
        return result(F_.sparse_conv(x1, x2, self.kernel, self.bias, rb_f, rb_b, False, rb_f.n_out, collect_stats=stats,
                                     passthrough=passthrough), level=l + 1)


class MinkowskiConvolutionTranspose(_ConvBase):
    """[ME-mem] kernel 2 / stride 2 transposed convolution onto the EXISTING finer coordinate map
    (required by the ME.cat key equality at detection_net.py:286-336)."""
    transposed = True

    def forward(self, x: SparseTensor, fuse=None) -> SparseTensor:
        assert self.stride == 2 and self.kernel_size == 2 and x.level >= 1
        m, l = x.manager, x.level - 1
        x1, x2 = _sources(x)
        rb_f, rb_b = m.rulebook_up(l), m.rulebook_down(l)
        if fuse is not None:             # inference: see MinkowskiConvolution.forward
            bn, residual, relu = fuse
            scale, shift = bn.eval_affine()
            if self.bias is not None:       # (as above: the bias folds into the shift)
                shift = shift + scale * self.bias.detach().reshape(-1).to(scale.dtype)
            return x.new(F_.conv_affine(x1, x2, self.kernel, rb_f, rb_f.n_out, scale, shift, residual, relu), level=l)
        if x1.dtype == torch.float16:          # half-precision training (half_train.py)
            from . import half_train as HT
            assert self.bias is None
            return x.new(HT.conv(x1, x2, self.kernel, rb_f, rb_b, False, rb_f.n_out, collect_stats=self.training), level=l)
        y = F_.sparse_conv(x1, x2, self.kernel, self.bias, rb_f, rb_b, False, rb_f.n_out,
                           collect_stats=self.training and self.bias is None)
        return x.new(y, level=l)


class MinkowskiBatchNorm(nn.Module):
    """BatchNorm1d on the feature matrix; parameters live under ``.bn`` like ME's wrapper.
    ``sync`` turns on the packed statistics exchange (MinkowskiSyncBatchNorm, model.py:25)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1):
        super().__init__()
        self.bn = nn.BatchNorm1d(num_features, eps=eps, momentum=momentum)
        self.sync = False

    def fusable(self) -> bool:
        """Inference: no gradient is being recorded and the layer normalises with its running statistics -- it is a
        per-channel affine map that the convolution in front of it can apply on its way out (functional.conv_affine)."""
        bn = self.bn
        return (not torch.is_grad_enabled() and not self.training and bn.track_running_stats and bn.running_mean is not None
                and bn.num_features % 4 == 0 and F_.conv_affine_enabled())

    def eval_affine(self):
        """(scale, shift) of the eval-mode layer, recomputed only when a parameter or running statistic may have changed: torch's
        in-place writes (load_state_dict, fill_) bump the tensors' version counters; the running statistics a TRAINING pass of
        this package updates and the parameters a fused optimizer steps do not -- every training-mode BatchNorm call advances
        functional's training epoch instead (F_.note_training_pass), which is part of the key."""
        bn = self.bn
        key = tuple((t.data_ptr(), t._version) if t is not None else None
                    for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var)) + (F_.training_epoch(),)
        cached = getattr(self, '_affine_cache', None)
        if cached is None or cached[0] != key:
            cached = (key, F_.bn_eval_affine(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps))
            self._affine_cache = cached
        return cached[1]

    def apply_bn(self, feats, residual=None, relu=False, count_key=None, defer_counter=False):
        bn = self.bn
        training = self.training or not bn.track_running_stats
        if training and feats.shape[0] == 1:   # torch.nn.functional.batch_norm raises the same (BatchNorm1d)
            raise ValueError('Expected more than 1 value per channel when training, got input size %s'
                             % (tuple(feats.shape),))
        if self.training and bn.track_running_stats and bn.num_batches_tracked is not None:
            if defer_counter:              # one foreach launch per forward pass instead of one add per layer
                _pending_counters.append(bn.num_batches_tracked)
            else:
                bn.num_batches_tracked.add_(1)
        if training:
            F_.note_training_pass()        # running statistics are about to change behind torch's back (eval_affine)
        if feats.dtype == torch.float16:   # half-precision training (half_train.py): binary16 in / out, fp64 statistics
            from . import half_train as HT
            if not training:
                raise RuntimeError('half activations outside inference need training-mode BatchNorm (half_train.py)')
            return HT.batch_norm(feats, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps, residual, relu,
                                 bool(self.sync))
        return F_.batch_norm(feats, bn.weight, bn.bias, bn.running_mean, bn.running_var, training,
                             bn.momentum, bn.eps, residual, relu, self.sync, count_key)

    def forward(self, x: SparseTensor) -> SparseTensor:
        # (inside SelectionNet.forward -- defer_counters(True) -- the heads' layers join the one foreach add of the pass)
        return x.new(self.apply_bn(x.F, count_key=count_key_of(x), defer_counter=_defer_all[0]))

    def _begin(self, feats, defer_counter):
        """The bookkeeping of apply_bn for a layer that runs inside a paired operator: (training?, parameter tuple)."""
        bn = self.bn
        training = self.training or not bn.track_running_stats
        if training and feats.shape[0] == 1:
            raise ValueError('Expected more than 1 value per channel when training, got input size %s'
                             % (tuple(feats.shape),))
        if self.training and bn.track_running_stats and bn.num_batches_tracked is not None:
            if defer_counter:
                _pending_counters.append(bn.num_batches_tracked)
            else:
                bn.num_batches_tracked.add_(1)
        if training:
            F_.note_training_pass()
        return training, (bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps)


def batch_norm_add_relu(norm_a: MinkowskiBatchNorm, feats_a, norm_b: MinkowskiBatchNorm, feats_b, relu=True,
                        defer_counter=False, count_key=None):
    """relu(norm_a(feats_a) + norm_b(feats_b)) -- the end of a BasicBlock with a shortcut convolution (resnet.py:73-82) --
    or None when the paired operator does not apply (different modes of the two layers, small maps outside SyncBN:
    those take the one-launch kernels of functional._BatchNorm)."""
    if not F_.bn_pair() or norm_a.training != norm_b.training or norm_a.sync != norm_b.sync:
        return None
    if feats_a.dtype == torch.float16:       # half-precision training: two layers (half_train.py has no paired form)
        return None
    n, c = feats_a.shape
    # the paired kernels take 16-byte column groups of dense fp32 rows (b2m_bn_apply2 checks the same)
    if c % 4 != 0 or c > 1024 or feats_b.shape != feats_a.shape:
        return None
    sync = norm_a.sync and F_._sync_group() is not None
    if norm_a.training and not sync and n <= F_.bn_small_rows():
        return None
    tr_a, pa = norm_a._begin(feats_a, defer_counter)
    tr_b, pb = norm_b._begin(feats_b, defer_counter)
    assert tr_a == tr_b
    return F_.batch_norm_pair(feats_a, pa, feats_b, pb, tr_a, relu, norm_a.sync, count_key)


def batch_norm_group(norms, tensors):
    """[norm_j(x_j)] for BatchNorm layers that do not depend on one another and see the same rows (the heads' layers at equal
    depth): under SyncBN in training mode ONE packed statistics exchange per direction for all of them
    (functional._BatchNormGroup); otherwise -- a single process, eval mode, one member, B2M_BN_GROUP=0 -- every layer by itself."""
    import os
    sync = F_._sync_group() if all(nm.sync for nm in norms) else None
    same_rows = len({t.F.shape[0] for t in tensors}) == 1
    ok = (sync is not None and len(norms) > 1 and same_rows and all(nm.training for nm in norms) and
          all(t.F.shape[1] % 4 == 0 and t.F.shape[1] <= 1024 for t in tensors) and tensors[0].F.shape[0] > 1 and
          os.environ.get('B2M_BN_GROUP', '1') == '1')
    if not ok:
        return [nm(t) for nm, t in zip(norms, tensors)]
    members = []
    for nm, t in zip(norms, tensors):
        training, (w, b, rm, rv, mom, eps) = nm._begin(t.F, _defer_all[0])
        assert training
        members.append((t.F, w, b, rm, rv, mom, eps))
    ys = F_.batch_norm_group(members, sync)
    return [t.new(y) for t, y in zip(tensors, ys)]


_pending_counters = []
_defer_all = [False]


def defer_counters(on: bool):
    """While on, EVERY training-mode BatchNorm layer (also the ones called as plain modules: the heads) books its
    `num_batches_tracked += 1` for the next flush_batch_counters() instead of launching an add of its own.  The caller
    flushes before it returns (SelectionNet.forward)."""
    _defer_all[0] = bool(on)


def flush_batch_counters():
    """num_batches_tracked += 1 for every BN layer that ran with defer_counter=True since the last flush."""
    if _pending_counters:
        torch._foreach_add_(list(_pending_counters), 1)
        _pending_counters.clear()


def count_key_of(x):
    """Key under which the global (all ranks) row count of x's rows is cached for SyncBN."""
    if getattr(x, 'manager', None) is not None:
        return ('level', x.manager.serial, x.level)
    return ('pooled', x.serial)


class MinkowskiSyncBatchNorm:
    @staticmethod
    def convert_sync_batchnorm(module: nn.Module):
        for m in module.modules():
            if isinstance(m, MinkowskiBatchNorm):
                m.sync = True
        return module


class MinkowskiReLU(nn.Module):
    def __init__(self, inplace=False):
        super().__init__()

    def forward(self, x: SparseTensor) -> SparseTensor:
        return x.new(F_.relu(x.F))


class PooledTensor:
    """Dense per-segment features after segment pooling (row r <-> pooling id r)."""

    def __init__(self, F, serial=None):
        from .sparse import _serial
        self.F = F
        self.manager, self.level = None, None
        self.serial = next(_serial) if serial is None else serial     # row family (same rows -> same SyncBN count)

    def new(self, F, level=None):
        return PooledTensor(F, self.serial)


def segment_pool(x: SparseTensor, pooling_ids, mode='avg') -> PooledTensor:
    """`out.C[:,0] = pooling_ids; ME.SparseTensor(out.F, out.C); global pool`
    (/root/reference/models/detection_net.py:345-352) as one segmented reduction."""
    n_seg = int(pooling_ids.max().item()) + 1 if pooling_ids.numel() else 0
    return PooledTensor(F_.segment_pool(x.F, pooling_ids, n_seg, mode))


def kaiming_normal_(tensor, mode='fan_out', nonlinearity='relu'):
    """[ME-mem] ME.utils.kaiming_normal_ on (K,Cin,Cout) kernels (resnet.py:142)."""
    if tensor.dim() == 2:
        fan_in, fan_out = tensor.size(0), tensor.size(1)
    else:
        fan_in, fan_out = tensor.size(1) * tensor.size(0), tensor.size(2) * tensor.size(0)
    fan = fan_in if mode == 'fan_in' else fan_out
    std = math.sqrt(2.0) / math.sqrt(fan)
    with torch.no_grad():
        return tensor.normal_(0, std)
