mkdir -p gpurun_out/r4
timeout -k 10 640 python tests/fuzz_parity_gpu.py 600 205 detector > gpurun_out/r4/detfuzz205.log 2>&1; echo product rc=$?; tail -n 1 gpurun_out/r4/detfuzz205.log
timeout -k 10 500 python tests/fuzz_parity_gpu.py 460 138 > gpurun_out/r4/fuzz138.log 2>&1; echo product rc=$?; tail -n 1 gpurun_out/r4/fuzz138.log
