"""Which torch (aten) operators does one training step issue, and from where?

    python tools/torch_ops.py            (BS=8 TV=150000 PREFETCH=1 HALF=0|1)

A TorchDispatchMode sees every aten operator of the step -- forward, the autograd engine's backward thread, the optimizer --
and books it under the innermost frame of this repository that issued it.  Operators that only make views or allocate are
listed apart: what counts here is what puts a kernel, a memset or a copy on the stream between this package's own launches
(`torch / runtime kernels` of profiles/rNN_summary.md)."""
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.utils._python_dispatch import TorchDispatchMode

from box2mask_amd import synth
from box2mask_amd.config import scannet_config
from box2mask_amd.model import Model

NO_KERNEL = ('aten.empty', 'aten.view', 'aten.as_strided', 'aten.detach', 'aten.alias', 'aten.slice', 'aten.select', 'aten.unsqueeze',
             'aten.squeeze', 'aten.reshape', 'aten._unsafe_view', 'aten.expand', 'aten.t.', 'aten.transpose', 'aten.permute',
             'aten.empty_like', 'aten.empty_strided', 'aten.new_empty', 'aten.unbind', 'aten.split', 'aten.is_pinned',
             'aten._local_scalar_dense', 'aten.record_stream', 'aten.lift_fresh', 'aten.resize_', 'aten.set_', 'aten.narrow',
             'aten.unfold', 'aten.is_same_size', 'aten.sym_', 'aten.stride', 'aten.size', 'aten.numel', 'aten.dim',
             'aten.is_contiguous', 'aten._has_compatible_shallow_copy_type', 'aten.is_nonzero')


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.ops = collections.Counter()
        self.on = False

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        if self.on:
            name = str(func)
            site = '?'
            for fr in reversed(traceback.extract_stack(limit=40)):
                if fr.filename.startswith(ROOT) and not fr.filename.endswith('torch_ops.py'):
                    site = '%s:%d %s' % (os.path.relpath(fr.filename, ROOT), fr.lineno, fr.name)
                    break
            if site == '?':
                for fr in reversed(traceback.extract_stack(limit=40)):
                    if 'torch/optim' in fr.filename or 'autograd' in fr.filename:
                        site = '%s:%d %s' % (fr.filename.split('site-packages/')[-1], fr.lineno, fr.name)
                        break
            self.ops[(name, site)] += 1
        return func(*args, **(kwargs or {}))


def main():
    cfg = scannet_config(half_training=os.environ.get('HALF', '0') == '1')
    torch.manual_seed(0)
    model = Model(cfg, *synth.scannet_tables())
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
    bs = int(os.environ.get('BS', '8')); tv = int(os.environ.get('TV', '150000'))
    batch = synth.make_batch(bs, seed0=0, target_voxels=tv)
    for k in ('vox_coords', 'vox_features', 'pooling_ids', 'input_location', 'gt_bb_offsets', 'gt_bb_bounds', 'gt_semantics',
              'fg_instances', 'batch_ids'):
        batch[k] = batch[k].cuda()
    model.train()
    prefetch = os.environ.get('PREFETCH', '1') == '1'

    def step():
        opt.zero_grad()
        l = model.compute_loss(batch, 150)
        if prefetch:
            model.prefetch(batch, ready=True)
        l['optimization_loss'].backward()
        opt.step()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    log = Log()
    with log:
        log.on = True
        step()
        log.on = False
    torch.cuda.synchronize()
    kern = collections.Counter(); quiet = collections.Counter()
    for (name, site), n in log.ops.items():
        (quiet if name.startswith(NO_KERNEL) else kern)[(name, site)] += n
    print('== operators that put work on the stream: %d per step' % sum(kern.values()))
    by_op = collections.Counter()
    for (name, site), n in kern.items():
        by_op[name] += n
    for name, n in by_op.most_common():
        print('%5d  %s' % (n, name))
    print('== by call site')
    for (name, site), n in sorted(kern.items(), key=lambda kv: -kv[1]):
        print('%5d  %-34s %s' % (n, name, site))
    print('== views / allocations (no device work): %d per step' % sum(quiet.values()))


if __name__ == '__main__':
    main()
