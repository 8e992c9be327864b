"""`bench.py --gpus 2` end to end on the one-GPU box: bench.py starts its two ranks itself (before any GPU call), they
rendezvous on 127.0.0.1 with the gloo backend (RCCL needs one device per rank: B2M_DIST_BACKEND=gloo, both ranks on
cuda:0), run SyncBN + the bucketed gradient all-reduce beside the weight-gradient stream, and rank 0 prints ONE JSON line
with n_gpus = 2 and the whole-job scenes/s.  Everything of the N > 1 path except RCCL itself."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_rehearsal():
    env = dict(os.environ, B2M_DIST_BACKEND='gloo', B2M_BENCH_ONE_DEVICE='1')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
                        '--batch-size', '2', '--target-voxels', '20000', '--cpu-baseline', '0', '--side-passes', '0'],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, p.stdout[-2000:]              # rank 0 only
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['config']['global_batch'] == 4 and d['config']['parallelism'] == 'dp2'
    assert d['value'] > 0 and d['scaling'] == 'weak' and 'votes_to_masks' not in d
    # the collectives of a step are counted (and the SyncBN exchanges timed): what an 8-GPU lease will show first
    c = d['config']['collectives']
    assert c['syncbn_all_reduces_per_step'] >= 100 and c['gradient_buckets_per_step'] >= 1 and c['syncbn_ms_per_step'] > 0
    assert c['syncbn_timed'] == c['syncbn_all_reduces_per_step']
    assert abs(d['value'] - 4 * d['steps'] / (d['ms_per_step'] * d['steps'] / 1e3)) < 0.02 * d['value']
