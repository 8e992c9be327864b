#!/bin/bash
# All profile artefacts of a round in one go (run on the GPU box from the repository root):
#   tools/final_profiles.sh <outdir>
# 1. rocprofv3 --kernel-trace --stats of `python3 bench.py --inference 0` (the driver's command without the batch-size-1
#    inference leg, whose launches would be averaged into the per-step table) -> kernel_stats.csv, summary.md
# 2. the same with B2M_WGRAD_STREAM=0 B2M_BENCH_PREFETCH=0 (one stream: a kernel's duration is its own) -> *_one_stream.*
# 3. two PMC passes (FETCH_SIZE / WRITE_SIZE, kernel-trace only) of one bench step, one stream  -> traffic.json
# 4. PMC passes of the conv micro-benchmark (tools/pmc_passes.sh)                               -> pmc/summary.txt
out=${1:-gpurun_out/final}
mkdir -p $out
export TMPDIR=/tmp
root=$(pwd)
stats() {   # $1 = tag, rest = env assignments
  tag=$1; shift
  (cd /tmp; for kv in "$@"; do export "$kv"; done; \
   rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/prof_$tag -- python3 $root/bench.py --inference 0 > $root/$out/bench_$tag.log 2>&1)
  f=$(find $out/prof_$tag -name "*kernel_stats.csv" | head -1)
  cp $f $out/kernel_stats_$tag.csv
  python3 tools/profile_summary.py $out/kernel_stats_$tag.csv $out/bench_$tag.log > $out/summary_$tag.md
  rm -rf $out/prof_$tag
  echo "== $tag"; head -12 $out/summary_$tag.md
}
stats default
stats one_stream B2M_WGRAD_STREAM=0 B2M_BENCH_PREFETCH=0
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && export B2M_WGRAD_STREAM=0 B2M_BENCH_PREFETCH=0 && rocprofv3 --pmc $c --kernel-trace --output-format csv -d $root/$out/pmc_$c -- \
     python3 $root/bench.py --steps 1 --warmup 1 --cpu-baseline 0 --votes 1 --prepare 0 --inference 0 > $root/$out/pmc_$c.log 2>&1)
done
python3 tools/pmc_traffic.py $(find $out/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) \
    $(find $out/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) > $out/traffic.json
rm -rf $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
head -30 $out/traffic.json
bash tools/pmc_passes.sh $out/pmc > $out/pmc_run.log 2>&1
find $out/pmc -name "*.csv" -delete
tail -5 $out/pmc/summary.txt
