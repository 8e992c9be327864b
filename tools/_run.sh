set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_nms.py tests/test_gpu_net.py -x -q -k "detection2mask or row_order" > gpurun_out/t7.log 2>&1 || { tail -40 gpurun_out/t7.log; exit 1; }
tail -3 gpurun_out/t7.log
