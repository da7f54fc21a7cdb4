mkdir -p gpurun_out/r4
MOFREAK_HIP_LIBRARY=mofreak_amd/libmofreak_hip_debug.so timeout -k 10 500 python -m pytest tests -q -m gpu > gpurun_out/r4/all_tests_debug.log 2>&1; echo debug-all rc=$?; tail -n 5 gpurun_out/r4/all_tests_debug.log
timeout -k 10 640 python tests/fuzz_parity_gpu.py 600 131 > gpurun_out/r4/final_fuzz131.log 2>&1; echo fuzz rc=$?; tail -n 1 gpurun_out/r4/final_fuzz131.log
