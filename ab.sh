mkdir -p gpurun_out/r4
timeout -k 10 400 python -m pytest tests/test_detector_gpu.py -x -q -m gpu -k "growing or overflows" > gpurun_out/r4/newtest.log 2>&1; echo rc=$?; tail -n 3 gpurun_out/r4/newtest.log
