mkdir -p gpurun_out/r4
timeout -k 10 250 python bench.py --gpus 2 --backend gloo --share-device --pairs 64 > gpurun_out/r4/final_bench_n2.json 2> gpurun_out/r4/final_bench_n2.err; echo rc=$?
timeout -k 10 250 python bench.py --config C4 --gpus 2 --backend gloo --share-device > gpurun_out/r4/final_bench_c4_n2.json 2> /dev/null; echo rc=$?
timeout -k 10 400 python -m pytest tests/test_dataset_gpu.py tests/test_rccl_gpu.py tests/test_facade.py -x -q -m gpu > gpurun_out/r4/port_tests.log 2>&1; echo rc=$?; tail -n 2 gpurun_out/r4/port_tests.log
