#!/bin/bash
# Round-6 diagnostics lease: (1) which torch operators a step issues and from where, (2) untraced step time on one stream and
# in the default mode against the sum / union of the kernels of the same command under a kernel trace -- the idle time of the
# UNTRACED step is (untraced one-stream wall) - (kernel sum), the tracer's own slowdown of the host does not enter.
out=gpurun_out/r06_diag
mkdir -p $out
export TMPDIR=/tmp
root=$(pwd)
B="--steps 20 --warmup 5 --cpu-baseline 0 --votes 0 --inference 0 --prepare 0 --side-passes 0"
timeout -k 10 300 python3 tools/torch_ops.py > $out/torch_ops.log 2>&1 || { tail -20 $out/torch_ops.log; exit 1; }
echo "== torch ops"; head -40 $out/torch_ops.log
timeout -k 10 300 python3 bench.py $B > $out/bench_default.log 2> $out/bench_default.err || { tail -20 $out/bench_default.err; exit 1; }
B2M_WGRAD_STREAM=0 B2M_BENCH_PREFETCH=0 timeout -k 10 300 python3 bench.py $B > $out/bench_one_stream.log 2> $out/bench_one_stream.err || exit 1
python3 - <<PY
import json
for t in ('default', 'one_stream'):
    d = json.loads(open('$out/bench_%s.log' % t).read().strip().splitlines()[-1])
    print(t, 'ms_per_step', d['ms_per_step'], 'value', d['value'])
PY
for tag in default one_stream; do
  if [ $tag = one_stream ]; then export B2M_WGRAD_STREAM=0 B2M_BENCH_PREFETCH=0; fi
  (cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $root/$out/prof_$tag -- python3 $root/bench.py $B > $root/$out/traced_$tag.log 2>&1) || exit 1
  f=$(find $out/prof_$tag -name "*kernel_trace.csv" | head -1)
  python3 tools/idle_analysis.py $f 10 14 > $out/idle_$tag.log
  python3 tools/gap_analysis.py $f > $out/gaps_$tag.log
  python3 tools/step_kernels.py $f 8 26 > $out/step_kernels_$tag.log
  gzip -c $f > $out/trace_$tag.csv.gz
  rm -rf $out/prof_$tag
  echo "== $tag"; tail -3 $out/traced_$tag.log | cut -c1-300; head -12 $out/idle_$tag.log; head -20 $out/step_kernels_$tag.log
done
ls -la $out
