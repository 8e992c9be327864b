"""Data-parallel glue: one process per GPU, scenes sharded across ranks, gradients averaged with
bucketed all-reduces over RCCL/xGMI that overlap the rest of backward.

Replaces torch DistributedDataParallel as the reference intends to use it
(/root/reference/models/model.py:24, training.py:286-297 — the published launcher cannot run,
SURVEY.md Appendix A).  The class is device-agnostic torch code, so the N>1 path is covered by
world_size-2 gloo tests on the CPU; on the GPU the backend "nccl" is RCCL.

Buckets are filled in reverse parameter order (the order backward produces gradients).  When the
last gradient of a bucket has been accumulated an asynchronous all-reduce of the bucket is launched;
a callback queued on the autograd engine finishes the job when the backward pass ends (wait, mean), so
the caller's loop is exactly the reference's `loss.backward(); optimizer.step()` (training.py:63-70).
With a gradient arena (grad_arena.py) a bucket IS a contiguous range of the arena: the collective runs
in place, nothing is packed or copied back.  Large buckets (default 64 MiB) keep the xGMI links busy
with few, large collectives.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_distributed(backend: str | None = None):
    """Initialise torch.distributed from the torchrun environment (RANK / WORLD_SIZE / MASTER_*)."""
    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world <= 1:
        return 0, 1
    rank = int(os.environ['RANK'])
    local = int(os.environ.get('LOCAL_RANK', rank))
    if backend is None:
        # B2M_DIST_BACKEND=gloo: rehearse the N > 1 path where RCCL cannot run (several ranks on one GPU, or no GPU)
        backend = os.environ.get('B2M_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
    if backend == 'nccl':
        torch.cuda.set_device(local)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29500')
    if backend == 'nccl':
        dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device('cuda', local))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world


def shard_scenes(n_scenes: int, rank: int, world: int):
    """Scene indices of this rank (DistributedSampler semantics without shuffling,
    /root/reference/models/dataloader.py:334-341): rank r takes scenes r, r+world, ..."""
    return list(range(rank, n_scenes, world))


class GradAllReduce:
    def __init__(self, params, bucket_bytes: int = 64 << 20, group=None, overlap: bool = True, arena=None, buffers=None):
        self.params = [p for p in params if p.requires_grad]
        self.buffers = list(buffers) if buffers is not None else []      # BatchNorm running statistics / counters
        self.group = group
        self.arena = arena
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.buckets = []          # list of lists of params, reverse parameter order
        cur, size = [], 0
        for p in reversed(self.params):
            cur.append(p)
            size += p.numel() * p.element_size()
            if size >= bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
        if cur:
            self.buckets.append(cur)
        self._bucket_of = {}
        for bi, b in enumerate(self.buckets):
            for p in b:
                self._bucket_of[p] = bi
        self._hooks = []
        self._reset()
        if overlap and self.world > 1:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def _reset(self):
        self._pending = [len(b) for b in self.buckets]
        self._work = [None] * len(self.buckets)
        self._flat = [None] * len(self.buckets)       # (tensor, in_place)
        self._armed = False                           # a finalize callback is queued for the running backward pass

    # -- hooks (overlap with backward)
    def _on_grad(self, p):
        if not self._armed:
            # first gradient of this backward pass: the engine calls _finalize when the pass is complete
            self._armed = True
            torch.autograd.Variable._execution_engine.queue_callback(self._finalize)
        bi = self._bucket_of[p]
        self._pending[bi] -= 1
        if self._pending[bi] < 0:
            raise RuntimeError('GradAllReduce: a gradient arrived twice in one backward pass (parameter used by two '
                               'graphs of one backward call is not supported)')
        if self._pending[bi] == 0:
            self._launch(bi)

    def _launch(self, bi):
        from . import functional as F_
        F_.join_side_streams()                # weight gradients issued on the side stream (functional._SparseConv)
        b = self.buckets[bi]
        span = self.arena.span(b) if self.arena is not None else None
        if span is not None:                  # every gradient of the bucket lives in its arena slot: reduce in place
            j, lo, hi = span
            flat, in_place = self.arena.buffers[j][lo:hi], True
        else:
            flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in b])
            in_place = False
        self._flat[bi] = (flat, in_place)
        from . import functional as F_
        F_.collective_stats['grad_buckets'] += 1
        F_.collective_stats['bytes'] += flat.numel() * flat.element_size()
        self._work[bi] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _finalize(self):
        """Wait for every bucket, turn sums into means and (only for buckets that had to be packed) write the
        result back.  Parameters without a gradient keep `.grad is None`."""
        if self.world <= 1:
            return
        for bi in range(len(self.buckets)):
            if self._work[bi] is None:        # bucket with parameters that received no gradient in this pass
                self._launch(bi)
        inv = 1.0 / self.world
        for bi, b in enumerate(self.buckets):
            self._work[bi].wait()
            flat, in_place = self._flat[bi]
            flat.mul_(inv)                                      # one launch per bucket
            if not in_place:
                views, dst, off = [], [], 0
                for p in b:
                    n = p.numel()
                    if p.grad is not None:
                        views.append(flat[off:off + n].view(p.shape)); dst.append(p.grad)
                    off += n
                if dst:
                    torch._foreach_copy_(dst, views)            # one launch for the whole bucket
        self._reset()

    # -- public
    def broadcast_parameters(self, src: int = 0):
        if self.world <= 1:
            return
        for p in self.params:
            dist.broadcast(p.data, src=src, group=self.group)
        # the BatchNorm running statistics as well: equal at a fresh init, NOT after a rank-local load_state_dict (the
        # reference's DDP broadcasts module buffers too, torch's `broadcast_buffers=True` default; model.py:24)
        for b in self.buffers:
            dist.broadcast(b.data, src=src, group=self.group)
        # written through `.data`: no version counter moved -- what inference caches from the parameters / statistics
        # (packed and half weight images, eval-mode BatchNorm affine maps) is rebuilt on its next use
        from . import functional as F_
        F_.note_training_pass()

    def all_reduce_mean(self):
        """Explicit form for callers without hooks (overlap=False): all-reduce now.  With hooks the work was done
        when backward ended and this is a no-op, so calling it after backward() is always safe."""
        if self.world <= 1:
            return
        if self._hooks and not self._armed and all(w is None for w in self._work):
            return
        self._finalize()
