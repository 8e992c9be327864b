set -e
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_default_mode.py tests/test_gpu_net.py tests/test_gpu_ops.py -x -q -s > gpurun_out/t6.log 2>&1 || { tail -40 gpurun_out/t6.log; exit 1; }
grep -i "replayed\|passed\|failed\|worst relative" gpurun_out/t6.log | tail -20
