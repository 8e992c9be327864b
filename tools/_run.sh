set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_inference.py tests/test_gpu_conv_regimes.py -x -q -k "inference or affine or tile_stats or forced_regime" > gpurun_out/t5.log 2>&1 || { tail -40 gpurun_out/t5.log; exit 1; }
tail -3 gpurun_out/t5.log
timeout -k 10 600 python bench.py --steps 5 --warmup 2 > gpurun_out/b5.json 2> gpurun_out/b5.err || { tail -30 gpurun_out/b5.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/b5.json').read().strip().splitlines()[-1])
print('ms/step', d['ms_per_step'], 'value', d['value'], 'roofline', d['roofline']['frac'], 'wgrad', d['roofline_wgrad']['frac'])
print('inference', json.dumps(d.get('inference'))[:1500])
PY
