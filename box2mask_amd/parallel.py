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


class IpcExchange:
    """The SyncBN statistics exchange inside one node without the collective library (include/b2m.h: b2m_xchg_*): every rank
    owns a mailbox in device memory, mapped into its peers through HIP IPC; an exchange is ONE one-workgroup launch that writes
    this rank's doubles into every mailbox, waits (bounded) for the peers' and adds up in rank order.  Opt-in: B2M_SYNCBN_IPC=1
    (functional._sync_all_reduce takes it for float64 device tensors of <= b2m_xchg_max_doubles() elements; everything else and
    every gradient bucket stays on torch.distributed).  Set-up is collective: every rank of `group` constructs it at the same
    point (Model.__init__ under cfg.multigpu does).  Exercised with two processes on ONE GPU (tests/test_gpu_dp.py); ranks on
    different GPUs need peer-visible mailboxes (b2m_xchg_alloc asks for fine-grained memory) -- no lease of the build pool had two."""

    def __init__(self, group=None, device=None):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        if self.world > lib.b2m_xchg_max_ranks():
            raise _lib.B2MError('IpcExchange: at most %d ranks' % lib.b2m_xchg_max_ranks())
        self.max_n = int(lib.b2m_xchg_max_doubles())
        self.device = device if device is not None else torch.device('cuda', torch.cuda.current_device())
        self._lib = lib
        buf, handle = C.c_void_p(), C.create_string_buffer(64)
        if lib.b2m_xchg_alloc(C.byref(buf), handle) != 0:
            raise _lib.B2MError('b2m_xchg_alloc failed: ' + lib.b2m_last_error().decode())
        self._own = buf.value
        handles = [None] * self.world
        dist.all_gather_object(handles, bytes(handle.raw), group=group)
        ptrs, self._opened = [], []
        for r, h in enumerate(handles):
            if r == self.rank:
                ptrs.append(self._own)
                continue
            p = C.c_void_p()
            if lib.b2m_xchg_open(C.create_string_buffer(h, 64), C.byref(p)) != 0:
                raise _lib.B2MError('b2m_xchg_open failed: ' + lib.b2m_last_error().decode())
            ptrs.append(p.value); self._opened.append(p.value)
        # Ranks on DIFFERENT devices need fine-grained (peer-coherent) mailboxes: a waiting kernel must see a peer device's
        # writes without a kernel boundary.  The plain allocation b2m_xchg_alloc falls back to is only coherent between
        # processes that share one device (the two-ranks-on-one-GPU rehearsal) -- refuse anything else.
        fine = bool(lib.b2m_xchg_is_finegrained(C.c_void_p(self._own)))
        where = [None] * self.world
        dist.all_gather_object(where, (_device_identity(self.device), fine), group=group)
        self.fine_grained = all(f for _, f in where)
        if len({d for d, _ in where}) > 1 and not self.fine_grained:
            self.close()
            raise _lib.B2MError('IpcExchange: ranks on different devices need fine-grained mailboxes and this runtime granted '
                                'none (plain device memory is not coherent under a running kernel of a peer device); '
                                'unset B2M_SYNCBN_IPC')
        self.peers = torch.tensor(ptrs, dtype=torch.int64, device=self.device)
        self.err = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._err_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self._err_event = None
        self.epoch = 0
        dist.barrier(group=group)              # every mailbox is mapped everywhere before the first exchange

    def usable(self, t: torch.Tensor) -> bool:
        return t.is_cuda and t.dtype == torch.float64 and t.is_contiguous() and 1 <= t.numel() <= self.max_n

    def all_reduce_(self, t: torch.Tensor):
        """SUM over the ranks, in place, on the current stream (every rank calls it for the same exchange in the same order)."""
        from . import _lib
        self.epoch += 1
        _lib.call('b2m_xchg_allreduce', t.data_ptr(), t.numel(), self.peers.data_ptr(), self.rank, self.world, self.epoch,
                  t.data_ptr(), self.err.data_ptr())

    _MSG = ('IpcExchange: a rank did not arrive within the wait bound (B2M_XCHG_TIMEOUT_S); the statistics of that exchange '
            'were replaced by NaN')

    def check(self):
        """Raises if an exchange gave up waiting for a peer.  A blocking host read: for the end of a run or a test."""
        from . import _lib
        if int(self.err.item()) != 0:
            raise _lib.B2MError(self._MSG)

    def check_async(self):
        """The step-boundary form (GradAllReduce._finalize calls it at the end of every backward pass): looks at the error
        flag as it was copied to pinned host memory at the PREVIOUS boundary -- no wait for the device -- and queues the
        next copy.  A timed-out exchange is therefore reported one step later at the latest; its result was NaN on the
        device from the start, so nothing trained on stale statistics in between."""
        from . import _lib
        if self._err_event is not None and self._err_event.query() and int(self._err_host[0]) != 0:
            raise _lib.B2MError(self._MSG)
        self._err_host.copy_(self.err, non_blocking=True)
        self._err_event = torch.cuda.Event()
        self._err_event.record()

    def close(self):
        for p in self._opened:
            self._lib.b2m_xchg_close(p)
        self._opened = []
        if self._own:
            self._lib.b2m_xchg_free(self._own)
            self._own = None


def _device_identity(device) -> str:
    """What tells two ranks' devices apart across processes: the device's UUID where torch exposes it, else host + visible index."""
    try:
        return str(torch.cuda.get_device_properties(device).uuid)
    except Exception:
        import socket
        return '%s/%s/%s' % (socket.gethostname(), os.environ.get('HIP_VISIBLE_DEVICES', os.environ.get('ROCR_VISIBLE_DEVICES', '')),
                             torch.device(device).index)


def syncbn_ipc_enabled() -> bool:
    return os.environ.get('B2M_SYNCBN_IPC', '0') == '1'


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
        from . import functional as F_
        if F_.ipc_exchange is not None:       # a SyncBN mailbox exchange that timed out in this or the previous pass
            F_.ipc_exchange.check_async()
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
