mkdir -p gpurun_out/r4
MOFREAK_HIP_LIBRARY=mofreak_amd/libmofreak_hip_debug.so timeout -k 10 500 python -m pytest tests -q -m gpu > gpurun_out/r4/all_tests_debug.log 2>&1; echo debug-all rc=$?; tail -n 2 gpurun_out/r4/all_tests_debug.log
MOFREAK_HIP_LIBRARY=mofreak_amd/libmofreak_hip_debug.so timeout -k 10 400 python tests/fuzz_parity_gpu.py 360 204 detector > gpurun_out/r4/detfuzz204.log 2>&1; echo debug rc=$?; tail -n 1 gpurun_out/r4/detfuzz204.log
MOFREAK_HIP_LIBRARY=mofreak_amd/libmofreak_hip_debug.so timeout -k 10 280 python tests/fuzz_parity_gpu.py 240 137 > gpurun_out/r4/fuzz137.log 2>&1; echo debug rc=$?; tail -n 1 gpurun_out/r4/fuzz137.log
