"""Shared helper of the GPU parity tests: the ReLU decisions of a device forward pass, replayed in the CPU oracle."""
import numpy as np
import torch


class MaskRecorder:
    """ReLU decisions of the device forward, replayed in the CPU oracle.

    Why: an element whose pre-activation lies within the forward rounding error of zero (relative 1e-4) may come out on
    different sides of zero on the device and in the oracle.  Each such flip switches one term of every gradient sum on or
    off: with a flipped fraction f the gradients of two CORRECT implementations differ by about sqrt(f) ~ 1e-2 (measured:
    the fp32 CPU oracle against the fp64 CPU oracle, median 1e-2), which would hide a genuine backward bug of a few per
    cent.  Taking the (few) borderline decisions from the device run makes both sides differentiate the SAME piecewise
    linear function; what is left is the backward pass itself, and it must agree to 1e-3 per parameter.

    The device rows are in the manager's internal (Morton) order, the oracle's in input order: masks are permuted per
    level by matching coordinates."""

    def __init__(self, monkeypatch):
        from box2mask_amd import functional as F_
        self.masks = []
        bn0, relu0 = F_.batch_norm, F_.relu

        def bn(x, gamma, beta, rm, rv, training, momentum=0.1, eps=1e-5, residual=None, relu=False, sync=False, count_key=None):
            y = bn0(x, gamma, beta, rm, rv, training, momentum, eps, residual, relu, sync, count_key)
            if relu:
                # count_key = ('level', manager serial, level) for rows of a coordinate map, ('pooled', serial) for segments
                self.masks.append((count_key[2] if count_key[0] == 'level' else None, y.detach() > 0))
            return y

        def relu(x):                                  # (the heads' ReLUs: pooled rows)
            y = relu0(x)
            self.masks.append((None, y.detach() > 0))
            return y
        pair0 = F_.batch_norm_pair

        def pair(xa, bn_a, xb, bn_b, training, relu=True, sync=False, count_key=None):
            y = pair0(xa, bn_a, xb, bn_b, training, relu, sync, count_key)
            if relu:
                self.masks.append((count_key[2] if count_key[0] == 'level' else None, y.detach() > 0))
            return y
        monkeypatch.setattr(F_, 'batch_norm', bn)
        monkeypatch.setattr(F_, 'relu', relu)
        monkeypatch.setattr(F_, 'batch_norm_pair', pair)

    def replay(self, manager, hier, n_seg, monkeypatch):
        """torch.relu of the oracle := multiplication with the recorded masks, in call order."""
        from oracle import sparse_ref
        to_gpu = {}                                   # rows of level l: oracle row r <-> device row to_gpu[l][r]
        for l in range(len(hier.coords)):
            kg = sparse_ref.pack_keys(manager.coords[l].cpu().numpy())
            ko = sparse_ref.pack_keys(hier.coords[l])
            order = np.argsort(kg)
            pos = np.searchsorted(kg[order], ko)
            assert np.array_equal(kg[order][pos], ko)
            to_gpu[l] = torch.from_numpy(order[pos])
        masks = []
        for level, m in self.masks:
            m = m.cpu()
            assert m.shape[0] == (n_seg if level is None else manager.n(level))
            masks.append(m if level is None else m[to_gpu[level]])
        it = iter(masks)

        def masked(x):
            m = next(it)
            assert m.shape == x.shape
            return x * m.to(x.dtype)
        monkeypatch.setattr(torch, 'relu', masked)
        self.to_gpu = to_gpu
        return it
