set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3/gpu_tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r3/gpu_tests.log
tail -5 gpurun_out/r3/gpu_tests.log
timeout -k 10 300 python bench.py > gpurun_out/r3/bench_n1.json 2> gpurun_out/r3/bench_n1.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --share-device --pairs 64 > gpurun_out/r3/bench_n2.json 2> gpurun_out/r3/bench_n2.err; echo "bench2 rc=$?"
timeout -k 10 300 python bench.py --config C4 > gpurun_out/r3/bench_c4.json 2> gpurun_out/r3/bench_c4.err; echo "c4 rc=$?"
timeout -k 10 300 python bench.py --config C4 --gpus 2 --backend gloo --share-device > gpurun_out/r3/bench_c4_n2.json 2> gpurun_out/r3/bench_c4_n2.err; echo "c4n2 rc=$?"
timeout -k 10 300 python bench.py --config C5 --stream > gpurun_out/r3/bench_c5.json 2> gpurun_out/r3/bench_c5.err; echo "c5 rc=$?"
