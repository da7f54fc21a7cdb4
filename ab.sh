mkdir -p gpurun_out/r4
MOFREAK_HIP_LIBRARY=mofreak_amd/libmofreak_hip_debug.so timeout -k 10 1150 python tests/fuzz_parity_gpu.py 1110 136 > gpurun_out/r4/fuzz136.log 2>&1; echo debug rc=$?; tail -n 3 gpurun_out/r4/fuzz136.log
