"""Benchmark of the hot path: ScanNet-shaped scenes/sec for one training step
(coordinate/kernel-map build, forward, losses, backward, gradient all-reduce, Adam step).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[1]: ScanNet 2 cm voxels, batch_size 8 per GPU, ~150 k voxels per scene,
synthetic scenes (box2mask_amd/synth.py), random-init weights, fp32.  One JSON line on rank 0.
`roofline` is measured live with HIP events around every launch of the dominant kernel
(b2m_conv_fwd: conv_fwd_flow_kernel, forward + data-gradient) inside the timed region; `cpu_baseline` times the CPU
oracle (kind "port": MinkowskiEngine itself is unavailable) on a bounded sample on rank 0 at N=1.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
PEAK_HBM_GBS = 8000.0
PEAK_F16_MFMA_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense BF16 / F16 MFMA (~2.5 PFLOP/s)


class LaunchTimer:
    """Brackets selected b2m_* launches with HIP events on torch's current stream (the stream the
    kernels are launched on) and keeps what is needed to compute algorithmic FLOPs afterwards."""

    def __init__(self, names):
        self.names = set(names)
        self.records = []          # (name, event pair, None, meta)
        self.enabled = False
        self._drained = 0
        self.step_id = 0           # advanced by the benchmark's step(): per-step sums of the bracketed launches

    def hook(self, name, args, meta_in=None):
        if not self.enabled or name not in self.names:
            return None
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
        if name == 'b2m_bn_apply':
            # x, ldx, n, c, scale, shift, residual, ldr, relu, y, ldy: streams x (+ residual) in and y out
            meta = dict(bytes=4.0 * args[2] * args[3] * (3 if args[6] else 2))
        elif name in ('b2m_conv_fwd', 'b2m_conv_fwd_stats'):      # same leading arguments
            # x1, ldx1, c1, x2, ldx2, c2, n_in, wp, K, bias, rb_in, rb_out, rb_cnt, n_out, y, ldy, cout, acc
            # (the 6-channel network input is read through a zero-padded 8-channel view: its FLOPs count the logical 6)
            cin = (meta_in or {}).get('cin', args[2] + args[5])
            meta = dict(cin=cin, cout=args[16], K=args[8], n_out=args[13], rb_cnt=args[12], n_in=args[6],
                        acc=int(args[17]))
        elif name == 'b2m_conv_up':
            # x1, ldx1, c1, x2, ldx2, c2, n_coarse, wp, K, bias, rb_in, rb_out, rb_cnt (the DOWN rulebook), y, ldy, cout, n_fine, acc
            # (counted with b2m_conv_fwd; a call the kernel declined -- *ran == 0 -- is followed by b2m_conv_fwd and not recorded)
            meta = dict(cin=args[2] + args[5], cout=args[15], K=args[8], n_out=args[6], rb_cnt=args[12], n_in=args[6],
                        acc=int(args[17]), ran=(meta_in or {}).get('ran'), n_fine=args[16])
        else:   # b2m_conv_wgrad / b2m_conv_wgrad_tr: x, ldx, cin, n_in, dy, lddy, cout, rb_in, rb_out, rb_cnt, n_out, K, ...
            # (_tr: the weight gradient of a transposed map over its DOWN rulebook -- the same pairs, the operands' roles exchanged)
            meta = dict(cin=args[2], cout=args[6], K=args[11], n_out=args[10], rb_cnt=args[9], n_in=args[3])

        meta['step'] = self.step_id

        def done():
            if name == 'b2m_conv_up':
                ran = meta.pop('ran', None)
                if ran is not None and not ran.value:
                    return
            e.record()
            self.records.append((name, _Pair(s, e), None, meta))
        return done

    def drain(self):
        """Turn the event pairs that have completed into plain milliseconds and let the events go.  Called at the start of
        every step: a 20-step run otherwise keeps ~25 000 HIP events alive until the end, and with that many outstanding
        every bracketed launch measured ~36 us longer (conv_fwd 51.7 instead of 44.5 ms per step; 5-step runs did not show it)."""
        for _, pair, _, _ in self.records[self._drained:]:
            if not pair.resolve():
                break
            self._drained += 1


class _Pair:
    """start / end event of one launch; `elapsed_time(None)` keeps the (name, s, e, meta) shape of the records."""

    def __init__(self, s, e):
        self.s, self.e, self.ms = s, e, None

    def resolve(self, wait=False):
        if self.ms is None:
            if not wait and not self.e.query():
                return False
            if wait:
                self.e.synchronize()
            self.ms = self.s.elapsed_time(self.e)
            self.s = self.e = None
        return True

    def elapsed_time(self, _):
        self.resolve(wait=True)
        return self.ms


def clock_probe(dev, iters=1500):
    """Shader clock (MHz) the device holds under fp32-MFMA load right now: b2m_clock_probe runs ~1 ms of the convolution kernels'
    MFMA block on every SIMD and stamps s_memtime / s_memrealtime around it (include/b2m.h)."""
    from box2mask_amd import _lib
    out = torch.zeros(4, dtype=torch.int64, device=dev)
    _lib.call('b2m_clock_probe', out.data_ptr(), iters)
    cyc, ticks, waves, _ = [int(v) for v in out.cpu().tolist()]
    return round(cyc / max(ticks, 1) * 100.0, 1)


def pairs_of(meta, cache, rb_lookup):
    """Total (in,out) pairs of the launch's kernel map (identity map: one pair per row)."""
    if meta['rb_cnt'] is None:
        return meta['n_out']
    key = meta['rb_cnt']
    if key not in cache:
        if key in rb_lookup:
            cnt, K, ntiles = rb_lookup[key]
            cache[key] = int(cnt[:K * ntiles].sum().item())
        else:
            cache[key] = 0
    return cache[key]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch-size', type=int, default=8, help='scenes per GPU (configs/scannet.txt: 8)')
    ap.add_argument('--target-voxels', type=int, default=150_000)
    ap.add_argument('--distinct-batches', type=int, default=4,
                    help='synthetic batches (different scenes, so different voxel counts and kernel maps) the steps cycle through, as '
                         'a data loader would deliver them; 1 = the same batch every step (rounds 1-5)')
    ap.add_argument('--cpu-baseline', type=int, default=1, help='0 disables the CPU oracle timing')
    ap.add_argument('--cpu-voxels', type=int, default=150_000, help='voxels per scene of the CPU baseline sample (default: the metric\'s scene size)')
    ap.add_argument('--cpu-scenes', type=int, default=1, help='scenes in the CPU baseline sample')
    ap.add_argument('--detail', type=int, default=0, help='1 prints a per-layer-shape table of the conv launches to stderr')
    ap.add_argument('--cpu-timeout', type=int, default=240, help='seconds after which the CPU baseline is abandoned')
    ap.add_argument('--votes', type=int, default=1, help='0 skips the votes -> instance masks leg (outside the timed steps)')
    ap.add_argument('--inference', type=int, default=1, help='0 skips the batch-size-1 inference leg (outside the timed steps)')
    ap.add_argument('--prepare', type=int, default=1, help='0 skips the raw points -> device batch leg (outside the timed steps)')
    ap.add_argument('--features', default='f32', choices=['f32', 'f16'],
                    help="f16 adds the `inference_f16` object: the batch-size-1 inference leg with the HALF trunk (activations in "
                         "HBM as IEEE half, f16 MFMA, fp32 accumulation; BASELINE configs[4]).  Never the headline, never the default")
    ap.add_argument('--side-passes', type=int, default=1,
                    help='0 skips the H2D-inclusive and the read-every-loss repeats of the timed steps (value_incl_h2d / value_sync_per_step '
                         'become null): the two-rank rehearsal of the test suite')
    ap.add_argument('--workload', default='scannet', choices=['scannet', 's3dis', 'arkit'],
                    help='scannet = BASELINE configs[1] (the headline); s3dis / arkit = configs[4] / [5], own lines under profiles/, '
                         'never the headline')
    args = ap.parse_args()
    # host-side torch ops (the outputs' clamp, pred2mask's bookkeeping on CPU tensors): the thread count the CPU baseline uses,
    # set whether or not that leg runs (with the default -- every core of the box -- a 2 k-element op costs milliseconds)
    try:
        torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    except AttributeError:
        pass
    if args.workload != 'scannet':       # the side legs and the CPU baseline belong to the headline workload
        args.votes = args.prepare = args.cpu_baseline = args.inference = 0

    # `python bench.py --gpus N` without a launcher: start the N rank processes here, BEFORE this process makes any
    # GPU call (children via subprocess; a process that has initialised the GPU must never exec another program).
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    if args.gpus > 1 and int(os.environ['WORLD_SIZE']) != args.gpus:
        sys.exit('bench.py: --gpus %d but WORLD_SIZE=%s' % (args.gpus, os.environ['WORLD_SIZE']))

    # The CPU baseline runs FIRST, before this process touches the GPU (pure torch-CPU oracle, rank 0, N=1 only).
    cpu_result = cpu_keep = None
    if args.cpu_baseline and int(os.environ.get('WORLD_SIZE', '1')) == 1:
        cpu_result, cpu_keep = cpu_baseline(args.cpu_scenes, args.cpu_voxels, args.cpu_timeout, args.target_voxels)

    from box2mask_amd import _lib, synth
    from box2mask_amd import sparse as sparse_mod
    from box2mask_amd.config import scannet_config
    from box2mask_amd.model import Model
    from box2mask_amd.parallel import init_distributed

    rank, world = init_distributed()
    if world != args.gpus:
        sys.exit('bench.py: started as %d rank(s) but --gpus %d' % (world, args.gpus))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if os.environ.get('B2M_BENCH_ONE_DEVICE') == '1':      # rehearsal of the N > 1 path on a one-GPU box (with B2M_DIST_BACKEND=gloo)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)

    # ---- the CPU oracle's forward of its full-size scene against the device path on the same weights and inputs
    # (whole network at the metric's scene size, train-mode BatchNorm; the parity figure travels with the bench line)
    if cpu_result is not None and cpu_keep is not None:
        cpu_result.update(gpu_vs_oracle(cpu_keep, dev))
        cpu_keep = None
    torch.manual_seed(1234)

    backend = dist.get_backend() if (dist.is_available() and dist.is_initialized()) else None
    workload, cfg, tables, batch = make_workload(args, rank, world)
    # further batches of the same shape, other scenes (scannet only: the other workloads are four fixed scene sizes): the steps
    # cycle through them, so allocation sizes, rulebook shapes and tile counts differ from step to step like a loader's
    batches = [batch]
    if args.workload == 'scannet':
        t_more = time.time()
        # (a loader's batches differ by some per cent in size: 1.0 / 0.93 / 1.07 / 1.0 x the metric's scene size -- mean 1.0)
        jitter = (1.0, 0.93, 1.07, 1.0)
        for j in range(1, max(1, args.distinct_batches)):
            batches.append(synth.make_batch(args.batch_size, seed0=(j * world + rank) * args.batch_size,
                                            target_voxels=int(args.target_voxels * jitter[j % 4])))
        workload['gen_s'] += time.time() - t_more
    model = Model(cfg, *tables, device=dev)
    opt = torch.optim.Adam(model.parameters(), lr=cfg.lr, fused=True)      # same update as training.py:35, one kernel
    model.train()

    # ---- synthetic batch of this rank (weak scaling: every rank gets batch_size scenes), resident in HBM
    t0 = time.time()
    n_vox = int(sum(b_['vox_coords'].shape[0] for b_ in batches) // len(batches))        # mean over the cycle
    vox_of_batches = [int(b_['vox_coords'].shape[0]) for b_ in batches]
    for b_ in batches:
        for k in ('vox_coords', 'vox_features', 'pooling_ids', 'input_location', 'gt_bb_offsets', 'gt_bb_bounds',
                  'gt_semantics', 'fg_instances', 'batch_ids', 'gt_per_vox_semantics'):
            if k in b_:
                b_[k] = b_[k].to(dev)
    torch.cuda.synchronize()
    gen_s = workload['gen_s'] + time.time() - t0

    # keep every rulebook built during the timed steps reachable for the FLOP accounting
    rb_lookup = {}
    orig_init = sparse_mod.Rulebook.__init__

    # (only the pair-count array of a rulebook is kept -- K x ntiles ints; keeping the rulebooks themselves alive pinned
    # ~1 GB per step: the pair lists of the 125-offset map alone are 750 MB, so a 45-step run allocated fresh memory in
    # every step and reported 30.6 GB of `peak_mem_gb` for a step that needs a third of it)
    def rb_init(self, *a, **k):
        orig_init(self, *a, **k)
        rb_lookup[self.rb_cnt.data_ptr()] = (self.rb_cnt, self.K, self.ntiles)
    sparse_mod.Rulebook.__init__ = rb_init

    # HIP events cost device and host time (about 2.5 % of a step when every conv and BN launch is bracketed), and in the
    # timed region the weight gradients run on a second stream BESIDE the data gradients: a bracketed launch's duration is
    # then the time it shared the chip, not the kernel's own.  So the timed region brackets the dominant kernel only
    # (b2m_conv_fwd: `roofline_timed_region`, as it ran), and K more steps AFTER the H2D-inclusive repeat run with the side
    # stream off and all three kernels bracketed (`roofline`, `roofline_wgrad`, `roofline_bn_apply`: every kernel alone on
    # the chip, which is what a roofline fraction is about).
    timer = LaunchTimer(['b2m_conv_fwd', 'b2m_conv_fwd_stats', 'b2m_conv_up'])
    _lib.set_hook(timer.hook)

    # Model.prefetch: the NEXT batch's sparse tensor (Morton order, coordinate hash, 7 strided + 16 kernel maps) is built
    # on a second stream while this step's network runs, as a training loop with a data loader one batch ahead would do.
    # Every step still builds exactly one batch's maps inside the timed region (the one it hands to the next step).
    PREFETCH = os.environ.get('B2M_BENCH_PREFETCH', '1') != '0'
    PREFETCH_AT = os.environ.get('B2M_BENCH_PREFETCH_AT', 'now')
    prefetch_on = [True]
    cursor = [0]                        # steps taken: step i trains on cycle[i % len(cycle)] and prefetches the one after it
    cycle = [batches]                   # (the H2D pass swaps in the pinned-host forms of the same batches)

    def step():
        timer.drain()
        timer.step_id += 1
        cur = cycle[0][cursor[0] % len(cycle[0])]
        nxt = cycle[0][(cursor[0] + 1) % len(cycle[0])]
        cursor[0] += 1
        opt.zero_grad()                 # torch default (set_to_none=True), as the reference's train_step
        cur_b = cur['step'] if 'step' in cur else cur
        ev0 = None
        if PREFETCH_AT == 'forward':
            ev0 = torch.cuda.Event()
            ev0.record()                # the device reaches this point when the step's forward pass begins
        losses = model.compute_loss(cur_b, 150)
        if PREFETCH and prefetch_on[0]:
            # ready=True: the next batch's tensors were complete before this step was enqueued (device tensors from setup, or the
            # pinned host buffers of the H2D pass) -- the side stream starts at once, i.e. wherever the device is while the host
            # runs ahead.  B2M_BENCH_PREFETCH_AT=forward: the side stream waits for the start of THIS step's forward pass, the one
            # phase of a step that runs on a single stream (experiment, profiles/r06_analysis.md)
            model.prefetch(nxt['prefetch'] if 'prefetch' in nxt else nxt, ready=ev0 if ev0 is not None else True)
        losses['optimization_loss'].backward()      # (N > 1: the gradient all-reduce completes inside backward)
        opt.step()
        return losses

    # one-time setup, not part of --warmup: the first two steps register the packed weight images, grow the caching
    # allocator's pools and JIT the fused optimiser kernel (reported as config.setup_steps)
    SETUP_STEPS = 2
    for i_ in range(SETUP_STEPS + args.warmup):
        step()
        # (not after the last one: its prefetch built the maps of the first timed step -- cleared, that step's launches counted
        # no FLOPs in `roofline_timed_region`, 5 % of the region's sum in rounds 2-5)
        if i_ < SETUP_STEPS + args.warmup - 1:
            rb_lookup.clear()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # `roofline_timed_region`: HIP events around the dominant kernel's launches INSIDE the timed region -- on every
    # TIMED_EVERY-th step only (default 5: steps 2, 7, 12, ...): two events per launch on 199 launches cost the step 1.5-2 %
    # (the H2D-inclusive repeat, which carries none, used to come out FASTER than the headline: 117.0 against 114.8 scenes/s),
    # and the headline should not pay for its own instrumentation.  B2M_BENCH_TIMED_EVENTS=1: every step (rounds 1-5); 0: none.
    TIMED_EVERY = int(os.environ.get('B2M_BENCH_TIMED_EVENTS', '5'))
    timed_steps_sampled = 0
    from box2mask_amd import functional as F_
    cstat = F_.collective_stats
    cstat.update(syncbn=0, grad_buckets=0, bytes=0)
    t_start = time.perf_counter()
    for i_ in range(args.steps):
        timer.enabled = TIMED_EVERY > 0 and i_ % TIMED_EVERY == min(TIMED_EVERY // 2, args.steps - 1)
        timed_steps_sampled += int(timer.enabled)
        losses = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_start
    timer.enabled = False
    # collectives of the timed steps (N > 1; zeros for a single process), and one more step with an event pair around every
    # SyncBN all-reduce: what the latency-bound exchanges cost on this fabric (they are blocking on the compute stream)
    collectives = {'syncbn_all_reduces_per_step': cstat['syncbn'] / args.steps, 'gradient_buckets_per_step': cstat['grad_buckets'] / args.steps,
                   'bytes_per_step': cstat['bytes'] // args.steps}
    if world > 1:
        cstat['timing'] = True
        cstat['events'] = []
        step()
        torch.cuda.synchronize()
        cstat['timing'] = False
        ev = cstat['events']
        ms = sorted(s_.elapsed_time(e_) for s_, e_ in ev)
        collectives.update(syncbn_ms_per_step=round(sum(ms), 3), syncbn_median_us=round(ms[len(ms) // 2] * 1e3, 1) if ms else None,
                           syncbn_timed=len(ms))
        cstat['events'] = []
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss_val = float(losses['optimization_loss'].item())
    peak_mem_gb = torch.cuda.max_memory_allocated(dev) / 2 ** 30

    # ---- the same K steps with the batch handed over as pinned HOST buffers (SURVEY 8d defines the metric with the
    # H2D copy; `value` stays the HBM-resident rate, this one is reported beside it)
    host_keys = ('vox_coords', 'vox_features', 'pooling_ids', 'input_location', 'gt_bb_offsets', 'gt_bb_bounds',
                 'gt_semantics', 'fg_instances', 'batch_ids')
    host_keys = tuple(k for k in host_keys + ('gt_per_vox_semantics',) if k in batch)
    pinned = [{k: b_[k].cpu().pin_memory() for k in host_keys} for b_ in batches]
    h2d_bytes = sum(v.numel() * v.element_size() for p_ in pinned for v in p_.values()) // len(pinned)

    class _HostBatch(dict):
        """One batch of the H2D pass: ['step'] = what compute_loss takes -- made when the step asks for it, so that the copies
        are queued inside the step -- and ['prefetch'] = what Model.prefetch takes (the same host tensors for coordinates and
        features: the side stream copies them; compute_loss recognises the batch by those tensors)."""

        def __init__(self, dev_b, pin):
            super().__init__()
            self.dev_b, self.pin = dev_b, pin

        def __contains__(self, k):
            return k in ('step', 'prefetch')

        def __getitem__(self, k):
            keep = ('vox_coords', 'vox_features', 'fg_instances') if k == 'prefetch' else \
                (('vox_coords', 'vox_features') if PREFETCH else ())
            out = dict(self.dev_b)
            for hk in host_keys:
                if k == 'prefetch':
                    if hk in keep:
                        out[hk] = self.pin[hk]
                else:
                    # with prefetch the voxel coordinates / features stay host tensors: the side stream copies them
                    out[hk] = self.pin[hk] if hk in keep else self.pin[hk].to(dev, non_blocking=True)
            return out

    elapsed_h2d = elapsed_sync = None
    if args.side_passes:
        cycle[0] = [_HostBatch(b_, p_) for b_, p_ in zip(batches, pinned)]
        cursor[0] = 0
        step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t_h = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed_h2d = time.perf_counter() - t_h
        if world > 1:
            t = torch.tensor([elapsed_h2d], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed_h2d = float(t.item())
        cycle[0] = batches
        cursor[0] = 0

    # ---- the same K steps with every loss read on the host in every iteration, as the reference's loop does
    # (/root/reference/models/training.py:170-174: `.item()` of each entry of the loss dict): the host cannot run ahead
    # across steps.  `value` stays the rate of the free-running loop; this one is reported beside it.
    def step_sync():
        ld = step()
        return {k_: (v_.item() if hasattr(v_, 'item') else float(v_)) for k_, v_ in ld.items()}
    if args.side_passes:
        step_sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t_y = time.perf_counter()
        for _ in range(args.steps):
            step_sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed_sync = time.perf_counter() - t_y
        if world > 1:
            t = torch.tensor([elapsed_sync], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed_sync = float(t.item())

    # ---- K more steps, one stream, every conv / BN-apply launch bracketed: the kernels' own durations
    timed_records = timer.records
    timer.records = []
    timer._drained = 0
    timer.names = {'b2m_conv_fwd', 'b2m_conv_fwd_stats', 'b2m_conv_up', 'b2m_conv_wgrad', 'b2m_conv_wgrad_tr', 'b2m_bn_apply'}
    prev_wgrad_stream = os.environ.get('B2M_WGRAD_STREAM')        # (a user-set value is restored afterwards)
    os.environ['B2M_WGRAD_STREAM'] = '0'
    _lib.reload_env()
    prefetch_on[0] = False              # (nothing beside the bracketed kernels: the maps are built in front of the forward pass)
    step()
    torch.cuda.synchronize()
    clock_before = clock_probe(dev)     # the clock the chip holds under MFMA load when the bracketed pass begins ...
    import gc
    gc.collect()
    gc.disable()                        # (no collector pause between an event and its launch)
    timer.enabled = True
    t_s = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed_serial = time.perf_counter() - t_s
    timer.enabled = False
    gc.enable()
    clock_after = clock_probe(dev)      # ... and when it ends (a long run heats the lease: both travel with the fractions)
    if prev_wgrad_stream is None:
        os.environ.pop('B2M_WGRAD_STREAM', None)
    else:
        os.environ['B2M_WGRAD_STREAM'] = prev_wgrad_stream
    _lib.reload_env()
    prefetch_on[0] = True

    # ---- `--features f16`: the same K steps with the trunk's activations and their gradients in half (half_train.py; BASELINE
    # configs[4] -- never the headline, never the default).  Same batches, same optimizer, fp32 master weights, static loss scale.
    train_f16 = None
    if args.features == 'f16' and world == 1:
        model.detection_model.half_training = True
        try:
            cursor[0] = 0
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            t_q = time.perf_counter()
            for _ in range(args.steps):
                lq = step()
            torch.cuda.synchronize()
            el_q = time.perf_counter() - t_q
            train_f16 = {'ms_per_step': round(el_q / args.steps * 1e3, 2),
                         'value': round(world * workload['batch_size'] * args.steps / el_q, 3), 'unit': 'scenes/s',
                         'speedup_vs_f32_step': round(elapsed / el_q, 3),
                         'loss_scale': float(getattr(cfg, 'half_loss_scale', 1024.0)),
                         'final_loss': round(float(lq['optimization_loss'].item()), 4),
                         'what': 'the timed step with SelectionNet.half_training: activations and activation gradients of the trunk '
                                 'as IEEE half in HBM (b2m_conv_fwd_h forward + data gradient, b2m_conv_wgrad_h, b2m_bn_*_h), fp32 stem, '
                                 'pooling, heads, master weights and Adam; weight gradient on the f16 MFMA (operands transposed by ds_read_b64_tr_b16; fp32 accumulation)'}
        finally:
            model.detection_model.half_training = False

    if rank != 0:
        if world > 1:
            dist.barrier()
        return

    # ---- roofline of the dominant kernels from the live event timings
    cache = {}
    agg = {}
    hbm = dict(ms=0.0, bytes=0.0, launches=0)
    for name, s, e, meta in timer.records:
        ms = s.elapsed_time(e)
        if name == 'b2m_bn_apply':
            hbm['ms'] += ms; hbm['bytes'] += meta['bytes']; hbm['launches'] += 1
            continue
        if name in ('b2m_conv_fwd_stats', 'b2m_conv_up'):
            name = 'b2m_conv_fwd'
        if name == 'b2m_conv_wgrad_tr':
            name = 'b2m_conv_wgrad'
        P = pairs_of(meta, cache, rb_lookup)
        flops = 2.0 * P * meta['cin'] * meta['cout']
        # bytes any implementation moves: both feature matrices once, the weights once, the pair lists
        # once (5 B per rulebook slot; identity maps have none)
        slots = 0 if meta['rb_cnt'] is None else meta['K'] * ((meta['n_out'] + 63) // 64) * 64
        # (a data gradient that accumulates onto the other consumer's gradient also reads its output once: acc)
        # (b2m_conv_up: n_out is the DOWN rulebook's coarse row count -- for the pair lists; the output it writes has n_fine rows)
        nbytes = 4.0 * (meta['n_in'] * meta['cin'] + (1 + meta.get('acc', 0)) * meta.get('n_fine', meta['n_out']) * meta['cout'] +
                        meta['K'] * meta['cin'] * meta['cout']) + 5.0 * slots
        a = agg.setdefault(name, dict(ms=0.0, flops=0.0, launches=0, bytes=0.0, steps={}))
        a['ms'] += ms; a['flops'] += flops; a['launches'] += 1; a['bytes'] += nbytes
        ps = a['steps'].setdefault(meta['step'], [0.0, 0.0])
        ps[0] += ms; ps[1] += flops

    if args.detail:
        shapes = {}
        for name, s_, e_, meta in timer.records:
            if name == 'b2m_bn_apply':
                continue
            key = ('conv_fwd' if name == 'b2m_conv_fwd_stats' else name[4:], meta['K'], meta['cin'], meta['cout'], meta.get('n_fine', meta['n_out']))
            d = shapes.setdefault(key, [0.0, 0.0, 0])
            d[0] += s_.elapsed_time(e_); d[1] += 2.0 * pairs_of(meta, cache, rb_lookup) * meta['cin'] * meta['cout']; d[2] += 1
        print('%-11s %4s %4s %4s %9s %6s %9s %8s' % ('kernel', 'K', 'cin', 'cout', 'n_out', 'calls', 'ms/step', 'TFLOP/s'), file=sys.stderr)
        for key, d in sorted(shapes.items(), key=lambda kv: -kv[1][0]):
            print('%-11s %4d %4d %4d %9d %6d %9.3f %8.2f' % (key + (d[2] // args.steps, d[0] / args.steps, d[1] / d[0] / 1e9 if d[0] else 0)), file=sys.stderr)

    # HBM-side bytes per launch: NOT measured in this run -- read from the newest committed summary of two separate
    # `rocprofv3 --pmc` passes of this same command (tools/pmc_traffic.py; FETCH_SIZE x2 on gfx950 + WRITE_SIZE, as
    # MI355X_MICROARCH.md prescribes).  `traffic_source` names the file; None when there is none.
    import glob
    pmc, pmc_src = {}, None
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_traffic.json'))):
        try:
            pmc, pmc_src = json.load(open(f)), 'committed PMC passes: profiles/' + os.path.basename(f)
        except (OSError, ValueError):
            pass

    def roof(a, kernel=None):
        # `achieved` / `frac`: the MEDIAN step of the bracketed pass (sum of the kernel's launches of a step).  An event pair
        # also times the gap in which the device waits for the host to submit the launch, and with every launch bracketed
        # the host does not run ahead: one host pause (allocator, interpreter) inside one step inflated round 4's 20-step
        # mean of the weight gradient from 0.51 to 0.47 (profiles/r05_repro.md).  The mean over all steps and the best /
        # worst step are printed beside it; rocprofv3's kernel trace (no host in it) agrees with the median.
        tf_mean = a['flops'] / (a['ms'] * 1e-3) / 1e12 if a['ms'] > 0 else 0.0
        per = sorted((f_ / (m_ * 1e-3) / 1e12, m_) for m_, f_ in a.get('steps', {}).values() if m_ > 0)
        tf = per[len(per) // 2][0] if per else tf_mean
        ms_step = per[len(per) // 2][1] if per else a['ms'] / max(args.steps, 1)
        spread = ({'frac_best_step': round(per[-1][0] / PEAK_FP32_MFMA_TFLOPS, 4), 'frac_worst_step': round(per[0][0] / PEAK_FP32_MFMA_TFLOPS, 4),
                   'frac_mean_all_steps': round(tf_mean / PEAK_FP32_MFMA_TFLOPS, 4), 'steps_bracketed': len(per),
                   'statistic': 'median step of the bracketed pass'} if per else {})
        return {'bound': 'mfma', 'achieved': round(tf, 3), 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                'frac': round(tf / PEAK_FP32_MFMA_TFLOPS, 4), **spread,
                'traffic': pmc.get(kernel, {}).get('traffic_bytes'),
                'traffic_source': pmc_src if pmc.get(kernel, {}).get('traffic_bytes') is not None else None,
                'algorithmic_bytes': round(a.get('bytes', 0.0) / max(a['launches'], 1)),
                'launches_per_step': a['launches'] // max(args.steps, 1),
                'avg_launch_ms': round(ms_step / max(a['launches'] // max(args.steps, 1), 1), 4),
                'gflop_per_step': round(a['flops'] / max(args.steps, 1) / 1e9, 2),
                'ms_per_step': round(ms_step, 3)}

    scenes = world * workload['batch_size'] * args.steps
    fwd = agg.get('b2m_conv_fwd', dict(ms=0.0, flops=0.0, launches=0))
    wg = agg.get('b2m_conv_wgrad', dict(ms=0.0, flops=0.0, launches=0))
    roofline = roof(fwd, 'conv_fwd_kernel')
    roofline['kernel'] = 'b2m_conv_fwd: conv_fwd_flow_kernel (+ conv_1x1_kernel for 1x1 layers, conv_fwd_kernel for the 6-channel stem and the heads), forward + data gradient'
    # the heaviest layer shapes of the dominant kernel, each with its own fraction (same isolated-kernel pass)
    shp = {}
    for name, s_, e_, meta in timer.records:
        if name in ('b2m_conv_fwd', 'b2m_conv_fwd_stats', 'b2m_conv_up'):
            d_ = shp.setdefault((meta['K'], meta['cin'], meta['cout'], meta.get('n_fine', meta['n_out'])), [0.0, 0.0, 0])
            d_[0] += s_.elapsed_time(e_); d_[1] += 2.0 * pairs_of(meta, cache, rb_lookup) * meta['cin'] * meta['cout']; d_[2] += 1
    roofline['top_shapes'] = [
        {'K': k_[0], 'cin': k_[1], 'cout': k_[2], 'rows': k_[3], 'launches_per_step': v_[2] // max(args.steps, 1),
         'ms_per_step': round(v_[0] / max(args.steps, 1), 3), 'achieved': round(v_[1] / v_[0] / 1e9, 2),
         'frac': round(v_[1] / v_[0] / 1e9 / PEAK_FP32_MFMA_TFLOPS, 4)}
        for k_, v_ in sorted(shp.items(), key=lambda kv: -kv[1][0])[:4] if v_[0] > 0]
    # the dominant kernel as it ran inside the timed region (beside the weight-gradient stream)
    tr = dict(ms=0.0, flops=0.0, launches=0, bytes=0.0)
    for name, s_, e_, meta in timed_records:
        tr['ms'] += s_.elapsed_time(e_); tr['launches'] += 1
        tr['flops'] += 2.0 * pairs_of(meta, cache, rb_lookup) * meta['cin'] * meta['cout']
    roofline_timed = roof(tr)
    n_s = max(timed_steps_sampled, 1)
    roofline_timed.update(launches_per_step=tr['launches'] // n_s, gflop_per_step=round(tr['flops'] / n_s / 1e9, 2),
                          ms_per_step=round(tr['ms'] / n_s, 3), avg_launch_ms=round(tr['ms'] / max(tr['launches'], 1), 4),
                          steps_bracketed=timed_steps_sampled)
    roofline_timed['note'] = ('HIP events around every b2m_conv_fwd launch of every %s step of the timed region (%d of %d steps); '
                              'data-gradient launches share the chip with the weight-gradient stream there, so this duration is '
                              "not the kernel's own (see `roofline`)" % ('%d-th' % TIMED_EVERY if TIMED_EVERY > 1 else 'single', timed_steps_sampled, args.steps))
    for k_ in ('traffic', 'traffic_source', 'algorithmic_bytes'):
        roofline_timed.pop(k_, None)
    roofline['measured'] = ('%d steps after the timed region on ONE stream, no prefetch beside them (B2M_WGRAD_STREAM=0: %.2f ms per step incl. the '
                            'events), every launch alone on the chip' % (args.steps, elapsed_serial / args.steps * 1e3))
    roofline['clock_mhz'] = {'before': clock_before, 'after': clock_after,
                             'note': 'shader clock under fp32-MFMA load (b2m_clock_probe: s_memtime / s_memrealtime) at the start and '
                                     'at the end of the bracketed pass; the peak assumes 2400'}
    roofline_wgrad = roof(wg, 'conv_wgrad_kernel')
    roofline_wgrad['clock_mhz'] = roofline['clock_mhz']
    roofline_wgrad['kernel'] = 'b2m_conv_wgrad: conv_wgrad_flow_kernel (+ conv_wgrad_kernel for single-block narrow heads)'

    value = scenes / elapsed
    result = {
        'metric': workload['metric'], 'value': round(value, 3), 'unit': 'scenes/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(elapsed / args.steps * 1e3, 2), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        # the same steps with the batch arriving in pinned host memory (PCIe copy inside the timed region)
        'value_incl_h2d': round(scenes / elapsed_h2d, 3) if elapsed_h2d else None, 'h2d_mb_per_step': round(h2d_bytes / 1e6, 1),
        # every loss `.item()`-ed in every iteration, as training.py:170-174 does (the host cannot run ahead)
        'value_sync_per_step': round(scenes / elapsed_sync, 3) if elapsed_sync else None,
        # /root/reference/README.md:102 quotes "~48GB GPURAM" for this batch size on MinkowskiEngine
        'peak_mem_gb': round(peak_mem_gb, 2),
        'config': {'workload': workload['name'],
                   'global_batch': world * workload['batch_size'], 'voxels_per_scene': n_vox // workload['batch_size'],
                   # what the process group saw (RCCL is backend "nccl" on ROCm); None / 1 for a single process
                   'backend': backend, 'ranks_seen': (dist.get_world_size() if backend else 1),
                   'voxels_per_gpu_batch': n_vox, 'parallelism': 'dp%d' % world, 'final_loss': round(loss_val, 4),
                   # the steps cycle through this many different batches (sizes jitter like a loader's); their voxel counts
                   'distinct_batches': len(batches), 'voxels_of_batches': vox_of_batches,
                   # the SURVEY 8d form of the metric (H2D copy of the batch inside the step) next to `value` -- kept here too
                   # because the driver's record carries `config` in full
                   'value_incl_h2d': round(scenes / elapsed_h2d, 3) if elapsed_h2d else None,
                   'h2d_mb_per_step': round(h2d_bytes / 1e6, 1),
                   'scene_gen_s': round(gen_s, 1), 'setup_steps': SETUP_STEPS,
                   'prefetch': ('next batch: coordinate + kernel maps on a second stream during the step'
                                if PREFETCH else 'off'),
                   'collectives': collectives,
                   'steps_executed': SETUP_STEPS + args.warmup + (4 if args.side_passes else 2) * args.steps + (3 if args.side_passes else 1) + (1 if world > 1 else 0)},
        'train_f16': train_f16,
        'roofline': roofline, 'roofline_timed_region': roofline_timed, 'roofline_wgrad': roofline_wgrad,
        # an HBM-bound kernel of the path, same live HIP-event method: BatchNorm apply (+residual, +ReLU) streams
        # 2-3 tensors per launch; small deep-level layers (launch-latency bound) are part of the average
        'roofline_bn_apply': {'bound': 'hbm', 'achieved': round(hbm['bytes'] / max(hbm['ms'], 1e-9) / 1e6, 1),
                              'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                              'frac': round(hbm['bytes'] / max(hbm['ms'], 1e-9) / 1e6 / PEAK_HBM_GBS, 4),
                              'traffic': pmc.get('bn_apply_kernel', {}).get('traffic_bytes'),
                              'traffic_source': pmc_src if pmc.get('bn_apply_kernel', {}).get('traffic_bytes') is not None else None,
                              'algorithmic_bytes': round(hbm['bytes'] / max(hbm['launches'], 1)),
                              'launches_per_step': hbm['launches'] // max(args.steps, 1),
                              'kernel': 'bn_apply_kernel'},
    }

    # ---- second half of configs[1] ("+ iou_nms on HIP"): votes -> instance masks of the same 8 scenes, reported
    # beside the headline value (never part of it)
    if args.votes and world == 1:          # N=1 only: the other ranks of a scaling run wait at the final barrier
        result['votes_to_masks'] = votes_leg(model, batch, cfg, cpu=bool(args.cpu_baseline), pmc=pmc, pmc_src=pmc_src)

    # ---- inference as the reference's Evaluater runs it: batch_size 1, eval mode, forward + votes -> masks
    if args.inference and world == 1:
        result['inference'] = inference_leg(model, dev, cfg, args.target_voxels, cpu_result, rb_lookup)
        if args.features == 'f16':
            result['inference_f16'] = inference_leg(model, dev, cfg, args.target_voxels, None, rb_lookup, half=True)
            # ... and the forward pass of the step's own batch (bs scenes at once: the launches are long enough for the device,
            # not the host's launch rate, to set the pace), fp32 and half
            result['inference_batched'] = inference_leg(model, dev, cfg, args.target_voxels, None, rb_lookup, own_batch=batch)
            result['inference_batched_f16'] = inference_leg(model, dev, cfg, args.target_voxels, None, rb_lookup, half=True, own_batch=batch)
    elif args.features == 'f16' and world == 1:     # the S3DIS- / ARKit-shaped workloads: forward of their own batch, fp32 and half
        result['inference'] = inference_leg(model, dev, cfg, args.target_voxels, None, rb_lookup, own_batch=batch)
        result['inference_f16'] = inference_leg(model, dev, cfg, args.target_voxels, None, rb_lookup, half=True, own_batch=batch)

    # ---- raw scene points -> voxelised, collated device batch (what feeds the step above)
    if args.prepare and world == 1:
        result['prepare'] = prepare_leg(dev, args.target_voxels, cpu=bool(args.cpu_baseline))

    # ---- CPU baseline: the oracle (port of the reference algorithm) on the host cores, bounded sample
    if cpu_result is not None:
        result['cpu_baseline'] = cpu_result
    print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()


def make_workload(args, rank, world):
    """(description, cfg, class tables, host batch) of this rank.  scannet = BASELINE configs[1], the headline.  The other
    two are BASELINE configs[4] / [5] as bench lines of their own (profiles/r03_workload_*.json), never the headline:
    s3dis -- configs/s3dis_fold5.txt: batch 4, 13 classes, per-voxel semantics head beside the segment heads, rooms of
    0.25-0.6 M voxels (S3DIS rooms after the reference's 0.25 point sampling); arkit -- configs/arkitscenes.txt: batch 4,
    4 cm voxels, 28 classes, mixed scene sizes, features handed over as fp16 (upcast at the boundary)."""
    from box2mask_amd import synth
    from box2mask_amd.config import scannet_config
    t0 = time.time()
    if args.workload == 'scannet':
        cfg = scannet_config(multigpu=(world > 1), batch_size=args.batch_size)
        batch = synth.make_batch(args.batch_size, seed0=rank * args.batch_size, target_voxels=args.target_voxels)
        w = dict(name='ScanNet 2cm voxels, batch_size=%d per GPU, sparse-conv fwd/bwd + losses + Adam '
                      '(BASELINE configs[1])' % args.batch_size,
                 metric='ScanNet scenes/sec (fwd+bwd, ~150k voxels @2cm)', batch_size=args.batch_size)
        tables = synth.scannet_tables()
    elif args.workload == 's3dis':
        cfg = scannet_config(multigpu=(world > 1), batch_size=4, eval_ths=[0.5, 0.03, 0.3, 0.6], loss_weight_bb_scores=3.0,
                             network_heads=['mlp_offsets', 'mlp_bounds', 'mlp_bb_scores', 'mlp_per_vox_semantics'])
        sizes = [250_000, 400_000, 600_000, 350_000]
        items = [synth.make_scene(rank * 4 + i, target_voxels=tv) for i, tv in enumerate(sizes)]
        batch = synth.collate(items)
        n = batch['vox_coords'].shape[0]
        batch['gt_semantics'] = batch['gt_semantics'] % 13
        batch['gt_per_vox_semantics'] = torch.from_numpy(np.random.default_rng(rank).integers(0, 13, n))
        valid = torch.Tensor(np.arange(13)); id2idx = torch.arange(41).long() % 13
        tables = (valid, id2idx, id2idx.clone(), (lambda s_: s_ > 2))
        w = dict(name='S3DIS-shaped rooms (%s voxels @2cm), batch_size=4 per GPU, 13 classes, per-voxel semantics head '
                      '(BASELINE configs[4], configs/s3dis_fold5.txt)' % '/'.join('%dk' % (v // 1000) for v in sizes),
                 metric='S3DIS-shaped scenes/sec (fwd+bwd, 0.25-0.6 M voxels @2cm)', batch_size=4)
    else:
        ids = np.array([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 15, 16, 18, 19, 20, 21, 22, 23, 24, 25, 28, 33, 34, 36, 39])
        cfg = scannet_config(multigpu=(world > 1), batch_size=4, eval_ths=[0.5, 0.05, 0.4, 0.6], loss_weight_bb_scores=3.0,
                             loss_weight_semantics=0.3, voxel_size=0.04)
        sizes = [30_000, 200_000, 80_000, 120_000]
        items = [synth.make_scene(rank * 4 + 10 + i, target_voxels=tv, voxel_size=0.04, pts_per_m2=5000.0)
                 for i, tv in enumerate(sizes)]
        batch = synth.collate(items)
        batch['gt_semantics'] = torch.from_numpy(ids[batch['gt_semantics'].numpy() % len(ids)])
        batch['vox_features'] = batch['vox_features'].half()
        valid = torch.Tensor(ids)
        id2idx = torch.zeros(41).fill_(-100).long(); id2idx[ids] = torch.arange(len(ids)).long()
        tables = (valid, id2idx, id2idx.clone(), (lambda s_: s_ > 2))
        w = dict(name='ARKitScenes-shaped scenes (%s voxels @4cm), batch_size=4 per GPU, 28 classes, fp16 features in '
                      '(BASELINE configs[5], configs/arkitscenes.txt)' % '/'.join('%dk' % (v // 1000) for v in sizes),
                 metric='ARKit-shaped scenes/sec (fwd+bwd, 30-200 k voxels @4cm)', batch_size=4)
    w['gen_s'] = time.time() - t0
    return w, cfg, tables, batch


def gpu_vs_oracle(keep, dev):
    """Forward of the device path on the CPU baseline's own weights and scene; max over the heads and the per-voxel
    trunk features of |gpu - oracle|_max / |oracle|_max (north_star tolerance: 1e-3)."""
    from box2mask_amd import nn as ME, synth
    from box2mask_amd.config import scannet_config
    from box2mask_amd.detection_net import SelectionNet
    cfg = scannet_config()
    valid, _, _, is_fg = synth.scannet_tables()
    net = SelectionNet(cfg, dev, valid, is_fg, out_channels=[96, 96, 6]).to(dev)
    net.load_state_dict(keep['state'])
    net.train()
    b = keep['batch']
    net._trace = {}
    with torch.no_grad():
        sin = ME.SparseTensor(b['vox_features'], b['vox_coords'], device=dev)
        out = net(sin, b['pooling_ids'].to(dev), b['input_location'].shape[0])
        trunk = net._trace['block8']
        if sin.manager.inv_perm is not None:
            trunk = trunk[sin.manager.inv_perm]
    errs, rerrs = {}, {}
    for h, ref in keep['out'].items():
        got = (trunk if h == '_trunk' else out[h].F).detach().cpu().double()
        ref = ref.double()
        tmax = max(float(ref.abs().max()), 1e-30)
        errs[h] = float((got - ref).abs().max()) / tmax
        # per-row figure: a row counts with its own magnitude (at least 5 % of the tensor's maximum) -- rows of small
        # magnitude are invisible to the per-tensor figure
        den = torch.clamp(ref.abs().max(1).values, min=0.05 * tmax)
        rerrs[h] = float(((got - ref).abs().max(1).values / den).max())
    del net, out, trunk, sin
    torch.cuda.empty_cache()
    return {'max_rel_err_vs_gpu': float('%.3e' % max(errs.values())),
            'max_row_rel_err_vs_gpu': float('%.3e' % max(rerrs.values())),
            'row_rel_err_vs_gpu_by_output': {('vox_feats' if k == '_trunk' else k): float('%.3e' % v) for k, v in rerrs.items()},
            'rel_err_vs_gpu_by_output': {('vox_feats' if k == '_trunk' else k): float('%.3e' % v) for k, v in errs.items()},
            'parity_note': 'device forward (default mode, train-mode BatchNorm) on the same weights and the same full-size '
                           'scene as this CPU sample; error = max |gpu - cpu| / max |cpu| per output; row figure = max over rows of '
                           '|gpu - cpu|_inf(row) / max(|cpu|_inf(row), 0.05 max |cpu|)'}


def spawn_ranks(n):
    """One process per GPU on this node, rendezvous on 127.0.0.1; rank 0's JSON line goes to our stdout.
    torch.cuda.device_count() does not initialise the GPU on this image, so it is safe to call here."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if os.environ.get('B2M_BENCH_ONE_DEVICE') == '1':
        have = n
    if have < n:
        print('bench.py: --gpus %d requested but only %d device(s) are visible' % (n, have), file=sys.stderr)
        return 2
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


KERNEL_OF = {'nmc_batch': 'nmc_batch_kernel', 'mask_project': 'mask_project_batch_kernel', 'mask_nms': 'mask_inter_batch_kernel',
             'label_hist': 'label_hist_batch_kernel', 'mask_gather': 'mask_gather_batch_kernel',
             'mask_gather_t': 'mask_gather_t_batch_kernel'}


def synthetic_votes(batch, cfg, seed=7):
    """Head outputs a trained network would produce: every segment votes for the box of its ground-truth object with
    2.5 cm noise on offset and bounds, score logits ~ N(0, 2), semantics = ground truth.  A random-init network's outputs are
    not votes (every segment its own cluster), so the votes -> masks half of every leg runs on these."""
    from box2mask_amd import synth
    g = torch.Generator().manual_seed(seed)
    S = batch['input_location'].shape[0]
    valid, id2idx, _, _ = synth.scannet_tables()
    sem_idx = id2idx[batch['gt_semantics'].cpu()].clamp_min(0)
    pred = {
        cfg.mlp_offsets: batch['gt_bb_offsets'].cpu() + 0.025 * torch.randn(S, 3, generator=g),
        cfg.mlp_bounds: (batch['gt_bb_bounds'].cpu() + 0.025 * torch.randn(S, 3, generator=g)).clamp_min(cfg.min_bb_size),
        cfg.mlp_bb_scores: 2.0 * torch.randn(S, 1, generator=g),
        cfg.mlp_semantics: torch.nn.functional.one_hot(sem_idx, len(valid)).float(),
    }
    return pred, sem_idx


def votes_leg(model, batch, cfg, cpu, pmc=None, pmc_src=None):
    """Evaluater flow (evaluation.py:86-97) on synthetic votes: every segment of the batch votes for the box of
    its ground-truth object with 2.5 cm noise on offset and bounds, score logits ~ N(0,2), semantics = ground
    truth.  Times Model.pred2mask(batch, pred, 'eval') over the whole batch; with `cpu`, scene 0 is
    also run on the CPU oracle (oracle/nms_ref.py, pinned bit for bit to the reference) and compared."""
    from box2mask_amd import synth
    S = batch['input_location'].shape[0]
    valid, id2idx, _, is_fg = synth.scannet_tables()
    pred, sem_idx = synthetic_votes(batch, cfg)
    cpu_batch = dict(batch)
    for k in ('input_location', 'batch_ids'):
        cpu_batch[k] = batch[k].cpu()
    model.eval()
    res = model.pred2mask(cpu_batch, pred, 'eval')          # warm-up
    torch.cuda.synchronize()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        res = model.pred2mask(cpu_batch, pred, 'eval')
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    # one more pass with HIP events around every kernel of the leg (own pass: the events serialise the stream)
    from box2mask_amd import _lib
    rec = []

    def hook(name, a, meta_in=None):
        if name not in ('b2m_nmc_batch', 'b2m_mask_project_batch', 'b2m_mask_nms_batch', 'b2m_label_hist_batch',
                        'b2m_mask_gather_batch', 'b2m_mask_gather_batch_t'):
            return None
        s_ = torch.cuda.Event(enable_timing=True); e_ = torch.cuda.Event(enable_timing=True)
        s_.record()

        def done():
            e_.record()
            rec.append((name, s_, e_, None))
        return done
    _lib.set_hook(hook)
    model.pred2mask(cpu_batch, pred, 'eval')
    torch.cuda.synchronize()
    _lib.set_hook(None)
    kern = {}
    d2m_bytes = getattr(model.detection_model, '_d2m_bytes', {})        # algorithmic bytes per stage (every operand once)
    for name, s_, e_, nb in rec:
        k_ = kern.setdefault(name[4:].replace('_batch', '') if name != 'b2m_nmc_batch' else 'nmc_batch', dict(ms=0.0, bytes=0.0, launches=0))
        k_['ms'] += s_.elapsed_time(e_); k_['launches'] += 1
        k_['bytes'] += d2m_bytes.get(name, 0.0)
    n_scenes = len(batch['scene'])
    fg_votes = int(is_fg(valid[sem_idx].long()).sum())
    # the clustering launch: 28 B per foreground vote read + one fp32 heat-map row per cluster written (the cluster
    # count per scene is read from the kernel's own output by re-running the stage alone)
    from box2mask_amd import iou_nms
    from box2mask_amd.util import to_bbs_min_max
    bbs = to_bbs_min_max(cpu_batch['input_location'], pred[cfg.mlp_offsets], pred[cfg.mlp_bounds],
                         torch.sigmoid(pred[cfg.mlp_bb_scores]))
    fg_all = is_fg(valid[sem_idx].long())
    per_scene = [bbs[(cpu_batch['batch_ids'] == b) & fg_all].float().cuda() for b in range(n_scenes)]
    rs = iou_nms.nmc_device_batch(per_scene, float(cfg.eval_ths[0]))
    if 'nmc_batch' in kern:
        kern['nmc_batch']['bytes'] = float(sum(28 * r.n + 4 * r.k * r.n for r in rs if r is not None))
        kern['nmc_batch']['clusters'] = int(sum(r.k for r in rs if r is not None))
    table = {k: {'launches': v['launches'], 'ms': round(v['ms'], 4), 'algorithmic_bytes': int(v['bytes']),
                 'gb_per_s': round(v['bytes'] / max(v['ms'], 1e-9) / 1e6, 2)} for k, v in kern.items()}
    dom = max(kern, key=lambda k: kern[k]['ms']) if kern else None
    out = {'value': round(n_scenes / dt, 2), 'unit': 'scenes/s', 'ms_per_scene': round(dt / n_scenes * 1e3, 3),
           'scenes': n_scenes, 'segments': int(S), 'foreground_votes': fg_votes,
           'instances': int(sum(len(r['conf']) for r in res.values())),
           'kernels': table,
           # the leg's dominant kernel against the HBM roofline.  nmc_kernel is K sequential IoU sweeps of one workgroup
           # per scene over a few hundred boxes: latency-bound by construction, the fraction says so
           'roofline': None if dom is None else {
               'bound': 'hbm', 'kernel': dom, 'achieved': table[dom]['gb_per_s'], 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
               'frac': round(table[dom]['gb_per_s'] / PEAK_HBM_GBS, 6),
               # HBM-side bytes per launch of that kernel from the committed PMC passes (FETCH_SIZE x2 + WRITE_SIZE)
               'traffic': (pmc or {}).get(KERNEL_OF.get(dom, dom), {}).get('traffic_bytes'),
               'traffic_source': pmc_src if (pmc or {}).get(KERNEL_OF.get(dom, dom), {}).get('traffic_bytes') is not None else None,
               'traffic_all_kernels': {k: (pmc or {}).get(KERNEL_OF.get(k, k), {}).get('traffic_bytes') for k in kern},
               'device_ms_all_kernels': round(sum(v['ms'] for v in kern.values()), 3)},
           'flow': "Model.pred2mask(batch, pred, 'eval') with eval_ths %s; pred on the host as in "
                   "evaluation.py:86, masks returned to the host" % (list(cfg.eval_ths),)}
    # "mAP@0.5 vs ref" half of the metric: ScanNet AP of the device path's masks against the scenes' own instances
    from box2mask_amd import eval_metric
    gts = {sc['name']: synth.gt_instance_ids(cpu_batch, b) for b, sc in enumerate(batch['scene'])}
    t0 = time.perf_counter()
    avg, _ = eval_metric.compute_eval(res, gts)
    out['ap'] = {'ap50': round(float(avg['all_ap_50%']), 6), 'ap': round(float(avg['all_ap']), 6),
                 'ap25': round(float(avg['all_ap_25%']), 6), 'eval_ms': round((time.perf_counter() - t0) * 1e3, 1)}
    if cpu:
        from oracle import nms_ref
        ref, cdt, same = {}, 0.0, True
        for b, sc in enumerate(batch['scene']):
            m = (cpu_batch['batch_ids'] == b).numpy()
            scores = torch.sigmoid(pred[cfg.mlp_bb_scores])[m].numpy()
            bbs = nms_ref.to_bbs_min_max(cpu_batch['input_location'][m].numpy(), pred[cfg.mlp_offsets][m].numpy(),
                                         pred[cfg.mlp_bounds][m].numpy(), scores)
            sem = valid[sem_idx[m]].long().numpy()
            t0 = time.perf_counter()
            r = nms_ref.detection2mask_scene(bbs, sem, lambda x: (x > 2) & (x != 22), np.asarray(batch['seg2vox'][b]),
                                             np.asarray(batch['vox2point'][b]), list(cfg.eval_ths), 'eval')
            cdt += time.perf_counter() - t0
            got = res[sc['name']]
            same = same and np.array_equal(r['conf'], got['conf'].numpy()) and \
                np.array_equal(r['label_id'], got['label_id']) and np.array_equal(r['mask'], got['mask'].numpy())
            ref[sc['name']] = {'conf': r['conf'], 'label_id': r['label_id'], 'mask': r['mask']}
        ravg, _ = eval_metric.compute_eval(ref, gts)
        out['ap']['ap50_cpu_oracle_masks'] = round(float(ravg['all_ap_50%']), 6)
        out['ap']['equal'] = bool(ravg['all_ap_50%'] == avg['all_ap_50%'] and ravg['all_ap'] == avg['all_ap'])
        out['cpu_baseline'] = {'value': round(n_scenes / cdt, 3), 'unit': 'scenes/s', 'cores': 1, 'kind': 'port',
                               'sample': 'the %d scenes of the batch on oracle/nms_ref.py (numpy restatement, bit-exact '
                                         'against the reference on tests/golden): %.2f s' % (n_scenes, cdt),
                               'identical_to_gpu_result': bool(same)}
    model.train()
    return out


def inference_leg(model, dev, cfg, target_voxels, cpu_result, rb_lookup, reps=10, half=False, own_batch=None):
    """_inference_leg with the model's mode switches (half_trunk, train / eval) restored whatever happens inside."""
    prev_half, prev_training = getattr(model.detection_model, 'half_trunk', False), model.detection_model.training
    try:
        return _inference_leg(model, dev, cfg, target_voxels, cpu_result, rb_lookup, reps, half, own_batch)
    finally:
        model.detection_model.half_trunk = prev_half
        model.train() if prev_training else model.eval()


def _inference_leg(model, dev, cfg, target_voxels, cpu_result, rb_lookup, reps=10, half=False, own_batch=None):
    """The reference's evaluation flow (evaluation.py:70-98: batch_size 1, model.eval(), no gradients): one synthetic scene
    of the metric's size, `Model.get_prediction(batch, with_grad=False)` + `Model.pred2mask(batch, pred, 'eval')`, scenes per
    second.  Every trunk convolution applies its eval-mode BatchNorm (+ residual) (+ ReLU) on the way out of its kernel
    (b2m_conv_fwd_affine).  Own roofline: the convolution launches of the forward pass, bracketed by HIP events in an extra
    pass (small-batch regime: a quarter of the rows per launch of the training benchmark's bs = 8 batches).  Never the
    headline."""
    from box2mask_amd import _lib, synth
    # own_batch (the S3DIS- / ARKit-shaped workloads): the forward pass of the workload's own device batch, no masks
    batch = dict(own_batch) if own_batch is not None else synth.make_batch(1, seed0=100, target_voxels=target_voxels)
    n_vox = int(batch['vox_coords'].shape[0])
    for k in ('vox_coords', 'vox_features', 'pooling_ids'):
        batch[k] = batch[k].to(dev)
    was_training = model.detection_model.training
    model.eval()
    pred32 = None
    if half:
        # the half trunk (SelectionNet.half_trunk): the fp32 inference path's outputs on the same scene first -- the parity figure
        pred32 = model.get_prediction(batch, with_grad=False, to_cpu=True, min_size=False)
        model.detection_model.half_trunk = True

    with_masks = own_batch is None
    if with_masks:
        votes, _ = synthetic_votes(batch, cfg, seed=8)
        cpu_batch = dict(batch)
        for k in ('input_location', 'batch_ids'):
            cpu_batch[k] = batch[k].cpu()

    def once(masks=with_masks):
        # the network's forward pass is the real one (its outputs come back to the host as in evaluation.py:86); the masks are
        # made from synthetic votes of the same scene: a random-init network votes every segment into a cluster of its own
        # (own_batch legs: the outputs stay on the device -- the S3DIS-shaped batch returns 0.6 GB of per-voxel features)
        pred = model.get_prediction(batch, with_grad=False, to_cpu=with_masks, min_size=True)
        return model.pred2mask(cpu_batch, votes, 'eval') if masks else pred
    for _ in range(4):                    # (the first passes of a new mix of tensor sizes are the caching allocator's)
        res = once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        res = once()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        once(masks=False)
    torch.cuda.synchronize()
    dt_fwd = (time.perf_counter() - t0) / reps
    # The same loop as an evaluation script with its data loader one scene ahead would run it: the forward pass is enqueued
    # (its outputs stay on the device), the NEXT scene's coordinate hash and kernel maps are built on a second stream while it runs
    # (Model.prefetch: their host reads wait for that stream only), then the outputs come to the host and the masks are made.
    dt_pf = dt_fwd_pf = None
    if not half:
        def once_ahead(masks=with_masks):
            pred = model.get_prediction(batch, with_grad=False, to_cpu=False, min_size=True)     # (takes the maps prefetched for it)
            model.prefetch(batch, ready=True, loss_rows=False)  # the next scene -- here the same one again -- beside this forward pass
            if not with_masks:
                return pred
            pred = {k: v.cpu() for k, v in pred.items()}        # the outputs come to the host as in evaluation.py:86
            return model.pred2mask(cpu_batch, votes, 'eval') if masks else pred
        model.prefetch(batch, ready=True, loss_rows=False)
        for _ in range(3):
            once_ahead()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            once_ahead()
        torch.cuda.synchronize()
        dt_pf = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        for _ in range(reps):
            once_ahead(masks=False)
        torch.cuda.synchronize()
        dt_fwd_pf = (time.perf_counter() - t0) / reps
        model._take_prefetched(batch)           # (drop the last one: the passes below build their own maps)
    # roofline of the forward pass's convolutions (own pass: the events serialise the stream)
    rec, launches = [], {}

    def hook(name, a, meta_in=None):
        if name != 'b2m_conv_up':          # (a b2m_conv_up call the kernel declines launches nothing: counted in done())
            launches[name] = launches.get(name, 0) + 1
        if name not in ('b2m_conv_fwd_affine', 'b2m_conv_fwd', 'b2m_conv_fwd_h', 'b2m_conv_up'):
            return None
        s_ = torch.cuda.Event(enable_timing=True); e_ = torch.cuda.Event(enable_timing=True)
        cin = (meta_in or {}).get('cin', a[2] + a[5])
        # b2m_conv_fwd_affine: x1, ldx1, c1, x2, ldx2, c2, n_in, wp, K, rb_in, rb_out, rb_cnt, n_out, y, ldy, cout, ...
        if name == 'b2m_conv_up':        # ..., n_coarse, wp, K, bias, rb_in, rb_out, rb_cnt (DOWN rulebook), y, ldy, cout, n_fine, ...
            meta = dict(cin=cin, cout=a[15], K=a[8], n_out=a[6], rb_cnt=a[12], half=False)
        elif name != 'b2m_conv_fwd':
            meta = dict(cin=cin, cout=a[15], K=a[8], n_out=a[12], rb_cnt=a[11], half=name == 'b2m_conv_fwd_h')
        else:
            meta = dict(cin=cin, cout=a[16], K=a[8], n_out=a[13], rb_cnt=a[12])
        ran = (meta_in or {}).get('ran')
        s_.record()

        def done():
            if name == 'b2m_conv_up':
                if ran is not None and not ran.value:      # declined: b2m_conv_fwd(_affine) follows
                    return
                launches[name] = launches.get(name, 0) + 1
            e_.record()
            rec.append((s_, e_, meta))
        return done
    _lib.set_hook(hook)
    once(masks=False)
    torch.cuda.synchronize()
    _lib.set_hook(None)
    cache, ms, flops, ms_h, flops_h = {}, 0.0, 0.0, 0.0, 0.0
    for s_, e_, meta in rec:
        t_, f_ = s_.elapsed_time(e_), 2.0 * pairs_of(meta, cache, rb_lookup) * meta['cin'] * meta['cout']
        ms += t_; flops += f_
        if meta.get('half'):
            ms_h += t_; flops_h += f_
    tf = flops / max(ms, 1e-9) / 1e9
    parity = None
    if half:
        pred16 = model.get_prediction(batch, with_grad=False, to_cpu=True, min_size=False)
        parity = {h: float((pred16[h].double() - pred32[h].double()).abs().max() / max(float(pred32[h].abs().max()), 1e-9))
                  for h in cfg.network_heads}
    if was_training:
        model.train()
    n_scenes = (int(batch['batch_ids'].max().item()) + 1) if own_batch is not None else 1
    out = {'value': round(n_scenes / dt, 3), 'unit': 'scenes/s', 'ms_per_scene': round(dt * 1e3 / n_scenes, 3),
           'ms_forward': round(dt_fwd * 1e3, 3), 'voxels': n_vox, 'scenes_per_pass': n_scenes,
           'instances': int(sum(len(r['conf']) for r in res.values())) if with_masks else None,
           'flow': ("batch_size 1: Model.get_prediction(batch, with_grad=False) [real forward, outputs to the host] + "
                    "Model.pred2mask(batch, votes, 'eval') on synthetic votes of the same scene (evaluation.py:70-98); masks "
                    'returned to the host') if with_masks else
                   "Model.get_prediction(batch, with_grad=False, to_cpu=False) on the workload's own batch (forward only, outputs stay on the device)",
           'launches_forward': int(sum(launches.values())),
           'fused_conv_bn_launches': int(launches.get('b2m_conv_fwd_affine', 0) + launches.get('b2m_conv_up', 0)),
           'batchnorm_launches': int(launches.get('b2m_bn_apply', 0) + launches.get('b2m_bn_apply2', 0)),
           'roofline': {'bound': 'mfma', 'achieved': round(tf, 3), 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                        'frac': round(tf / PEAK_FP32_MFMA_TFLOPS, 4), 'traffic': None,
                        'kernel': 'b2m_conv_fwd_affine (conv_fwd_flow_kernel / conv_1x1_kernel / conv_stem_kernel with the '
                                  'BatchNorm epilogue) + b2m_conv_fwd (heads)',
                        'launches': len(rec), 'ms': round(ms, 3), 'gflop': round(flops / 1e9, 2)}}
    if dt_pf is not None:
        out['next_scene_prefetched'] = {
            'value': round(n_scenes / dt_pf, 3), 'unit': 'scenes/s', 'ms_per_scene': round(dt_pf * 1e3 / n_scenes, 3),
            'ms_forward': round(dt_fwd_pf * 1e3, 3),
            'flow': 'the same loop with Model.prefetch(next scene) called right behind the enqueued forward pass: the next scene\'s '
                    'coordinate hash, strided maps and kernel maps are built on a second stream while this scene\'s network runs '
                    '(an evaluation script whose data loader is one scene ahead); same outputs (tests/test_gpu_inference.py)'}
    if half:
        tf_h = flops_h / max(ms_h, 1e-9) / 1e9
        out['features'] = 'f16 (trunk activations and weights IEEE half in HBM, v_mfma_f32_16x16x32_f16 / 16x16x16_f16, fp32 ' \
                          'accumulation and BatchNorm epilogue; 6-channel stem, pooled features and heads fp32)'
        out['half_conv_launches'] = int(launches.get('b2m_conv_fwd_h', 0))
        out['max_rel_err_vs_fp32_path'] = {h: float('%.3e' % e) for h, e in parity.items()}
        out['parity_note'] = ('largest |half - fp32| / max|fp32| per head on this scene; the fp32 path is the one pinned against the '
                              'CPU oracle (gpu_vs_oracle), and tests/test_gpu_half.py compares the half trunk with the oracle '
                              'directly (<= 2e-2)')
        out['roofline'] = {'bound': 'mfma', 'achieved': round(tf_h, 3), 'peak': PEAK_F16_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                           'frac': round(tf_h / PEAK_F16_MFMA_TFLOPS, 4), 'traffic': None,
                           'kernel': 'b2m_conv_fwd_h: conv_fwd_flow_kernel<.., F16> (the half launches only; the kernel is bound by the '
                                     "CU's vector-load path -- 64 bytes per lane-row and KiB weight pieces per MFMA -- not by the f16 "
                                     'MFMA pipe: DESIGN.md §5)',
                           'launches': sum(1 for r in rec if r[2].get('half')), 'ms': round(ms_h, 3), 'gflop': round(flops_h / 1e9, 2)}
    if cpu_result is not None and cpu_result.get('inference_s'):
        out['cpu_baseline'] = {'value': round(1.0 / cpu_result['inference_s'], 4), 'unit': 'scenes/s', 'cores': cpu_result['cores'],
                               'kind': 'port',
                               'sample': 'forward of ONE scene of the metric\'s size on the CPU oracle (oracle/unet_ref.py, '
                                         'coordinate / kernel-map build included): %.1f s; the votes -> masks half on the CPU '
                                         'is votes_to_masks.cpu_baseline' % cpu_result['inference_s']}
    return out


def prepare_leg(dev, target_voxels, cpu, n_scenes=2):
    """Dataset item incl. weak box supervision + collate (dataloader.py:61-314, 946-995) of `n_scenes` synthetic raw scenes (~1.2 M points,
    ~150 k voxels each) with the points already resident in HBM; with `cpu`, scene 0 also runs on the CPU oracle
    (numpy + the reference's sklearn ball tree; bit-exact against the reference on tests/golden) and is compared."""
    from box2mask_amd import prepare, synth
    raw = [synth.make_scene(100 + s, target_voxels=target_voxels, points_only=True) for s in range(n_scenes)]
    scenes = [{k: (torch.as_tensor(v).to(dev) if isinstance(v, np.ndarray) else v) for k, v in sc.items()} for sc in raw]
    torch.cuda.synchronize()

    from types import SimpleNamespace
    sup = SimpleNamespace(smallest_bb_heuristic=True)           # configs/scannet.txt: bb_supervision, smallest_bb_heuristic

    def run():
        # (round 5: the scenes of a batch are voxelised together -- two host reads per BATCH, the voxel and the segment counts)
        items = [prepare.box_supervision(it, sc['labels'], sup) for it, sc in zip(prepare.voxelize_scenes(scenes, 0.02), scenes)]
        return items, prepare.collate(items, 'train')
    run()
    torch.cuda.synchronize()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        items, batch = run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    # device time of the leg's kernels alone (HIP events around every b2m_* launch; own pass): the figure without the
    # two host reads per scene (voxel and segment counts size the next allocations)
    from box2mask_amd import _lib
    rec = []

    def hook(name, a, meta_in=None):
        s_ = torch.cuda.Event(enable_timing=True); e_ = torch.cuda.Event(enable_timing=True)
        s_.record()

        def done():
            e_.record()
            rec.append((name, s_, e_))
        return done
    _lib.set_hook(hook)
    run()
    torch.cuda.synchronize()
    _lib.set_hook(None)
    kernels_ms = sum(s_.elapsed_time(e_) for _, s_, e_ in rec)
    by_kernel = {}
    for name, s_, e_ in rec:
        by_kernel[name[4:]] = round(by_kernel.get(name[4:], 0.0) + s_.elapsed_time(e_) / n_scenes, 4)
    pts = sum(int(sc['positions'].shape[0]) for sc in raw)
    nvox = int(batch['vox_coords'].shape[0])
    # bytes every correct implementation moves per scene: positions (24 B/pt) read by the key pass and by the two
    # association passes, keys + slots + inverse written (8+4+8 B/pt), colours/normals/segment gathered per voxel
    # (56 B) and the voxel rows / features / maps written (16+24+8+8+4 B)
    algo = pts * (3 * 24 + 20) + nvox * (56 + 60)
    out = {'value': round(n_scenes / dt, 2), 'unit': 'scenes/s', 'ms_per_scene': round(dt / n_scenes * 1e3, 3),
           'points': pts, 'voxels': nvox, 'points_per_s': round(pts / dt, 1),
           'roofline': {'bound': 'hbm', 'achieved': round(algo / dt / 1e9, 2), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                        'frac': round(algo / dt / 1e9 / PEAK_HBM_GBS, 4), 'traffic': None,
                        'note': 'whole leg incl. 2 host reads per batch (the voxel and the segment counts of all scenes)',
                        'kernels_only': {'ms_per_scene': round(kernels_ms / n_scenes, 3),
                                         'achieved': round(algo / (kernels_ms * 1e-3) / 1e9, 2),
                                         'frac': round(algo / (kernels_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                                         'ms_per_scene_by_entry': by_kernel}}}
    if cpu:
        from oracle import prepare_ref
        sc = raw[0]
        t0 = time.perf_counter()
        ref = prepare_ref.voxelize_scene(sc['positions'], sc['colors'], sc['normals'], sc['segments'], 0.02)
        ipp, ips = prepare_ref.approx_association(sc['positions'], sc['segments'], sc['labels'],
                                                  ref['unique_vox_segments'], True)
        cdt = time.perf_counter() - t0
        it = items[0]
        same = np.array_equal(it['pseudo_inst'][1].cpu().numpy(), ips) and all(np.array_equal(it[k].cpu().numpy(), ref[k]) for k in ('vox2point', 'point2vox', 'seg2vox')) and \
            np.array_equal(it['vox_coords'][:, 1:].cpu().numpy(), ref['vox_coords'].astype(np.int32)) and \
            np.array_equal(it['vox_features'].cpu().numpy(), ref['vox_features'].astype(np.float32))
        out['cpu_baseline'] = {'value': round(1.0 / cdt, 3), 'unit': 'scenes/s', 'cores': 1, 'kind': 'port',
                               'sample': 'scene 0 (%d points) on oracle/prepare_ref.py (numpy + sklearn ball tree as '
                                         'dataloader.py:61-123, vectorised box association :203-314): %.2f s'
                                         % (len(sc['positions']), cdt),
                               'identical_to_gpu_result': bool(same)}
    return out


def cpu_baseline(n_scenes, voxels, timeout_s, ref_voxels):
    """Coordinate/kernel-map build + forward + backward of ONE full-size synthetic scene (the metric's ~150 k voxels,
    no scaling) on the CPU oracle (torch CPU, per-offset index_select -> mm -> index_add_): scenes per second.
    Threads: min(16, cores available) -- the reference pins MinkowskiEngine's CPU path to
    OMP_NUM_THREADS=16 (/root/reference/config_loader.py:3-4).  Abandoned after `timeout_s` seconds."""
    import signal
    from box2mask_amd import synth
    from box2mask_amd.config import scannet_config
    from box2mask_amd.detection_net import SelectionNet
    from oracle import unet_ref
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(16, avail))
    torch.set_num_threads(cores)
    cfg = scannet_config()
    valid, _, _, is_fg = synth.scannet_tables()
    torch.manual_seed(0)
    net = SelectionNet(cfg, 'cpu', valid, is_fg, out_channels=[96, 96, 6])
    p = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and 'running' not in k else v)
         for k, v in net.state_dict().items()}
    b = synth.make_batch(n_scenes, seed0=0, target_voxels=voxels)
    nvox = int(b['vox_coords'].shape[0])

    class _Timeout(Exception):
        pass

    def _alarm(signum, frame):
        raise _Timeout()

    keep = None
    old = signal.signal(signal.SIGALRM, _alarm)
    signal.alarm(int(timeout_s))
    t0 = time.perf_counter()
    try:
        out = unet_ref.forward(p, b['vox_coords'].numpy(), b['vox_features'], b['pooling_ids'], cfg, training=True,
                               n_segments=b['input_location'].shape[0], return_trunk=True)
        dt_fwd = time.perf_counter() - t0
        keep = {'state': {k: v.detach().clone() for k, v in net.state_dict().items()}, 'batch': b,
                'out': {k: v.detach().clone() for k, v in out.items()}}
        loss = sum(v.abs().mean() for k, v in out.items() if k != '_trunk')
        loss.backward()
        dt = time.perf_counter() - t0
        value = round(n_scenes / dt, 5)
        note = '%.1f s' % dt
    except _Timeout:
        value, note, keep, dt_fwd = None, 'abandoned after %d s' % timeout_s, None, None
    finally:
        signal.alarm(0)
        signal.signal(signal.SIGALRM, old)
    return {'value': value, 'unit': 'scenes/s', 'cores': cores, 'kind': 'port',
            'inference_s': None if (value is None or dt_fwd is None) else round(dt_fwd / n_scenes, 2),
            'sample': '%d synthetic scene(s) of the metric\'s size (%d voxels in all, seed 0..%d; no scaling): coordinate/'
                      'kernel-map build + forward + backward on the CPU oracle (oracle/unet_ref.py, torch %s, %d threads): '
                      '%s; value = scenes / seconds' % (n_scenes, nvox, n_scenes - 1, torch.__version__, cores, note)}, keep


if __name__ == '__main__':
    main()
