mkdir -p gpurun_out/r4
MOFREAK_HIP_LIBRARY=mofreak_amd/libmofreak_hip_debug.so timeout -k 10 1150 python tests/fuzz_parity_gpu.py 1110 135 > gpurun_out/r4/fuzz135.log 2>&1; echo debug rc=$?; tail -n 1 gpurun_out/r4/fuzz135.log
