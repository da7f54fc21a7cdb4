mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r4/all_tests.log 2>&1; echo product rc=$?
tail -n 3 gpurun_out/r4/all_tests.log
MOFREAK_HIP_LIBRARY=mofreak_amd/libmofreak_hip_debug.so timeout -k 10 400 python -m pytest tests/test_detector_gpu.py -x -q -m gpu > gpurun_out/r4/tie_tests_debug.log 2>&1; echo debug rc=$?
tail -n 2 gpurun_out/r4/tie_tests_debug.log
timeout -k 10 200 python tests/fuzz_parity_gpu.py 120 111 > gpurun_out/r4/fuzz111.log 2>&1; echo fuzz rc=$?; tail -n 2 gpurun_out/r4/fuzz111.log
MOFREAK_HIP_LIBRARY=mofreak_amd/libmofreak_hip_debug.so timeout -k 10 200 python tests/fuzz_parity_gpu.py 120 112 > gpurun_out/r4/fuzz112.log 2>&1; echo fuzz debug rc=$?; tail -n 2 gpurun_out/r4/fuzz112.log
