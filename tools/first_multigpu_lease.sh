#!/bin/bash
# The first lease on a node with >= 2 MI355X, as ONE command (run from the repository root):
#     bash tools/first_multigpu_lease.sh [max_gpus]           (default: every visible device, at most 8)
# 1. the RCCL tests that a one-GPU box skips (tests/test_gpu_dp.py -k rccl);
# 2. bench.py --gpus {1,2,4,8} (weak scaling, 8 scenes per rank) with the SyncBN statistics exchange through torch.distributed
#    (B2M_SYNCBN_IPC=0, the default) and through the device-side mailboxes (B2M_SYNCBN_IPC=1);
# 3. one table: scenes/s, ms/step, SyncBN exchanges per step, their median latency and their share of the step.
# Every rank exits non-zero when a mailbox exchange times out (B2M_XCHG_TIMEOUT_S, parallel.IpcExchange.check_async) -- nothing
# is re-executed.  Results land in gpurun_out/multigpu/.
set -o pipefail
out=gpurun_out/multigpu
mkdir -p $out
export HSA_ENABLE_IPC_MODE_LEGACY=0
ngpu=$(python3 -c "import torch; print(torch.cuda.device_count())")
max=${1:-8}
[ "$ngpu" -lt "$max" ] && max=$ngpu
echo "devices visible: $ngpu, running up to $max ranks"
if [ "$ngpu" -lt 2 ]; then echo "needs >= 2 devices"; exit 3; fi
timeout -k 10 900 python3 -m pytest tests/test_gpu_dp.py -m gpu -k rccl -x -q > $out/pytest_rccl.log 2>&1 || { tail -30 $out/pytest_rccl.log; exit 1; }
tail -3 $out/pytest_rccl.log
port=29610
for ipc in 0 1; do
  for n in 1 2 4 8; do
    [ "$n" -gt "$max" ] && continue
    [ "$n" = 1 ] && [ "$ipc" = 1 ] && continue
    port=$((port + 1))
    log=$out/bench_n${n}_ipc${ipc}
    if [ "$n" = 1 ]; then
      B2M_SYNCBN_IPC=$ipc timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-baseline 0 --votes 0 --inference 0 --prepare 0 \
          > $log.json 2> $log.err || { tail -20 $log.err; exit 1; }
    else
      B2M_SYNCBN_IPC=$ipc timeout -k 10 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $port \
          bench.py --gpus $n --steps 20 --warmup 5 > $log.json 2> $log.err || { tail -20 $log.err; exit 1; }
    fi
  done
done
python3 - <<PY
import glob, json, os, re
rows = []
for f in sorted(glob.glob('$out/bench_n*_ipc*.json')):
    n, ipc = map(int, re.search(r'n(\d+)_ipc(\d)', f).groups())
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        rows.append((n, ipc, 'unreadable: %s' % e)); continue
    c = d['config'].get('collectives', {})
    rows.append((n, ipc, d['value'], d['ms_per_step'], d['config'].get('backend'), d['config'].get('ranks_seen'),
                 c.get('syncbn_all_reduces_per_step'), c.get('syncbn_median_us'), c.get('syncbn_ms_per_step'), c.get('gradient_buckets_per_step')))
base = next((r[2] for r in rows if r[0] == 1 and len(r) > 3), None)
with open('$out/table.md', 'w') as fh:
    print('| ranks | SyncBN exchange | scenes/s | ms/step | vs N x 1-GPU | backend | ranks seen | exchanges/step | median us | ms/step in exchanges | gradient buckets |', file=fh)
    print('|---:|---|---:|---:|---:|---|---:|---:|---:|---:|---:|', file=fh)
    for r in sorted(rows):
        if len(r) == 3:
            print('| %d | %s | %s |' % (r[0], 'mailboxes' if r[1] else 'torch.distributed', r[2]), file=fh); continue
        eff = '%.3f' % (r[2] / (base * r[0])) if base else ''
        print('| %d | %s | %.2f | %.2f | %s | %s | %s | %s | %s | %s | %s |' % (r[0], 'mailboxes' if r[1] else 'torch.distributed', r[2], r[3], eff, *r[4:]), file=fh)
print(open('$out/table.md').read())
PY
