"""Flat gradient storage: every parameter gradient of a model is a view into one contiguous fp32 buffer.

Why: the weight-gradient kernels ACCUMULATE (fp32 atomics over tile chunks), so each of the ~110 convolution
weights needed its own zero-filled tensor per step (one fill launch each), and the data-parallel all-reduce had to
concatenate ~300 gradient tensors into flat buckets and copy the result back (2 x 292 MB of traffic, ~600 small
launches).  With the arena a step zeroes the whole buffer with ONE memset, the backward operators write straight into
their slots, autograd adopts the slot views as `.grad`, and the all-reduce runs in place on contiguous bucket
ranges of the same memory (box2mask_amd/parallel.py).

Aliasing rule: a buffer may only be cleared while no live `.grad` points into it.  `optimizer.zero_grad()`
(set_to_none=True, the torch default the reference's train_step relies on, /root/reference/models/training.py:63-70)
drops the views, so in the normal loop buffer 0 is reused every step.  If gradients are kept across backward
passes (accumulation), `.grad` still owns one buffer and the next pass takes the other one; autograd then adds the
new views into the old ones in place.  A caller that keeps a reference to an old `.grad` tensor beyond the second
following backward pass sees it overwritten -- the one behavioural difference to separately allocated gradients.
"""
from __future__ import annotations

import weakref

import torch

_ALIGN = 64          # floats: every slot starts on a 256-byte boundary
_registry = {}       # id(param) -> (weakref(param), weakref(arena), index)


class GradArena:
    def __init__(self, params, order: str = 'reverse'):
        """params: the model's parameters.  Slots are laid out in REVERSE parameter order -- the order backward
        produces gradients -- so that the buckets of parallel.GradAllReduce are contiguous ranges."""
        self.params = [p for p in params if p.requires_grad]
        assert self.params, 'no trainable parameters'
        self.device = self.params[0].device
        seq = list(reversed(self.params)) if order == 'reverse' else list(self.params)
        self.offset, off = {}, 0
        for p in seq:
            assert p.dtype == torch.float32 and p.device == self.device
            self.offset[id(p)] = off
            off += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.size = off
        self.sequence = seq
        self.buffers = [None, None]      # allocated on first use (the second one only if gradients are ever kept)
        self.current = None              # index of the buffer of the pass in flight
        self.handed = set()              # id(p) of the slots handed out since the last begin_pass
        for p in self.params:
            _registry[id(p)] = (weakref.ref(p), weakref.ref(self), id(p))

    # ---- per pass
    def _owned(self, j):
        """True if some live .grad points into buffer j."""
        buf = self.buffers[j]
        if buf is None:
            return False
        base = buf.data_ptr()
        for p in self.params:
            g = p.grad
            if g is not None and g.data_ptr() == base + 4 * self.offset[id(p)]:
                return True
        return False

    def begin_pass(self):
        """Pick a buffer no live gradient points into and clear it (one memset).  Called once per forward pass,
        before any backward operator of that pass can run."""
        self.handed.clear()
        j = 0 if not self._owned(0) else 1
        if j == 1 and self._owned(1):
            # gradients from two earlier passes are both alive (the caller re-assigned .grad by hand): give up the
            # arena for this pass, the operators fall back to separately allocated gradients
            self.current = None
            return
        dev = self.params[0].device                      # follows Model.to(device)
        if self.buffers[j] is None or self.buffers[j].device != dev:
            self.buffers[j] = torch.empty(self.size, dtype=torch.float32, device=dev)
        self.buffers[j].zero_()
        self.current = j

    def slot(self, p):
        """Zero-initialised view for the gradient of `p` in the current pass, or None (no pass open / foreign tensor).
        A slot is handed out ONCE per pass: when the backward operator of one parameter runs a second time inside the
        same backward() -- two forward passes summed into one loss, a weight shared by two layers -- the first result is
        still held by autograd (p.grad is None until AccumulateGrad runs), so the second caller gets None, allocates a
        tensor of its own, and autograd adds the two."""
        if self.current is None:
            return None
        off = self.offset.get(id(p))
        if off is None or id(p) in self.handed:
            return None
        view = self.buffers[self.current][off:off + p.numel()].view(p.shape)
        g = p.grad
        if g is not None and g.data_ptr() == view.data_ptr():
            return None          # a second backward of the same pass: .grad already lives here, it must not be its own addend
        self.handed.add(id(p))
        return view

    # ---- for the all-reduce
    def span(self, plist):
        """(buffer index, start, end) of the contiguous range covering the slots of `plist`, provided every one of
        these parameters has a gradient living in its slot of ONE buffer; else None."""
        which = None
        lo, hi = None, None
        for p in plist:
            g = p.grad
            off = self.offset.get(id(p))
            if g is None or off is None:
                return None
            hit = None
            for j in (0, 1):
                if self.buffers[j] is not None and g.data_ptr() == self.buffers[j].data_ptr() + 4 * off:
                    hit = j
            if hit is None or (which is not None and hit != which):
                return None
            which = hit
            end = off + (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
            lo = off if lo is None else min(lo, off)
            hi = end if hi is None else max(hi, end)
        # contiguity: the padded sizes of the members must fill the range exactly
        total = sum((p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN for p in plist)
        if which is None or total != hi - lo:
            return None
        return which, lo, hi


def arena_of(p):
    """The arena a parameter is registered with (None if it has none or either object is gone)."""
    e = _registry.get(id(p))
    if e is None:
        return None
    pref, aref, _ = e
    if pref() is not p:
        del _registry[id(p)]
        return None
    return aref()


def grad_slot(p):
    """Gradient slot of `p` for the pass in flight, or None -> the caller allocates as usual."""
    a = arena_of(p)
    return a.slot(p) if a is not None else None
