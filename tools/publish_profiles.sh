#!/bin/bash
# Copy the artefacts tools/final_profiles.sh left under <outdir> into profiles/ under the round's tag:
#   tools/publish_profiles.sh gpurun_out/r3final r03
src=${1:?outdir}; tag=${2:?tag}
# (nothing is touched unless every input is there: a run on an empty <outdir> once emptied the committed JSON line)
for f in summary_one_stream.md summary_default.md kernel_stats_one_stream.csv kernel_stats_default.csv traffic.json pmc/summary.txt bench_one_stream.log; do
  [ -s "$src/$f" ] || { echo "publish_profiles: $src/$f is missing or empty" >&2; exit 1; }
done
set -e
cp $src/summary_one_stream.md profiles/${tag}_summary.md
cp $src/summary_default.md profiles/${tag}_summary_two_streams.md
cp $src/kernel_stats_one_stream.csv profiles/${tag}_bench_kernel_stats.csv
cp $src/kernel_stats_default.csv profiles/${tag}_bench_kernel_stats_two_streams.csv
cp $src/traffic.json profiles/${tag}_traffic.json
mkdir -p profiles/${tag}_pmc && cp $src/pmc/summary.txt profiles/${tag}_pmc/summary.txt
grep '^{' $src/bench_one_stream.log | tail -1 > profiles/${tag}_bench_under_rocprof.json
ls -la profiles | grep ${tag}
