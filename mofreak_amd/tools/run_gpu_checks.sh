#!/bin/bash
# One GPU-box session: the -m gpu tests, then the bench lines of every BASELINE workload (each line into gpurun_out/r3/).
# Steps are joined so that nothing runs after a step that timed out or failed.
set -x
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3/gpu_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r3/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py > gpurun_out/r3/bench_n1.json 2> gpurun_out/r3/bench_n1.err &&
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --share-device --pairs 64 > gpurun_out/r3/bench_n2.json 2> gpurun_out/r3/bench_n2.err &&
timeout -k 10 300 python bench.py --config C2 --no-detector > gpurun_out/r3/bench_c2.json 2> gpurun_out/r3/bench_c2.err &&
timeout -k 10 300 python bench.py --config C4 --steps 3 > gpurun_out/r3/bench_c4.json 2> gpurun_out/r3/bench_c4.err &&
timeout -k 10 300 python bench.py --config C4 --steps 3 --per-clip-calls > gpurun_out/r3/bench_c4_perclip.json 2> gpurun_out/r3/bench_c4_perclip.err &&
timeout -k 10 300 python bench.py --config C4 --gpus 2 --backend gloo --share-device > gpurun_out/r3/bench_c4_n2.json 2> gpurun_out/r3/bench_c4_n2.err &&
timeout -k 10 300 python bench.py --config C5 --stream > gpurun_out/r3/bench_c5.json 2> gpurun_out/r3/bench_c5.err
echo "benches rc=$?"
