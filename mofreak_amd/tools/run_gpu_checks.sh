# The round's final measurements on the GPU box, in two gpurun calls (a call is limited to 20 minutes):
#   bash mofreak_amd/tools/run_gpu_checks.sh 1   tests, the tile kernel's profile + counters, the default bench line, the detector's profile
#   (copy gpurun_out/r05_traffic.json to profiles/traffic.json and profiles/r05_traffic.json)
#   bash mofreak_amd/tools/run_gpu_checks.sh 2   the other bench lines, the C2 counter pass, the fuzz tool on both builds, smoke()
set -x
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r5f
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ "${1:-1}" = "1" ]; then
timeout -k 10 400 python -m pytest tests -x -q -m gpu > gpurun_out/r5f/final_tests.log 2>&1; echo tests rc=$?; tail -n 2 gpurun_out/r5f/final_tests.log
timeout -k 10 500 bash mofreak_amd/tools/profile_tile.sh r05 > gpurun_out/r5f/final_profile_tile.log 2>&1; echo tile prof rc=$?
cp gpurun_out/r05_traffic.json profiles/traffic.json  # (on the box; the same copy is made in the repo afterwards)
timeout -k 10 300 python bench.py > gpurun_out/r5f/final_bench_default.json 2> gpurun_out/r5f/final_bench_default.err; echo rc=$?
timeout -k 10 600 bash mofreak_amd/tools/profile_detector.sh r05 > gpurun_out/r5f/final_profile_detector.log 2>&1; echo det prof rc=$?
exit 0
fi
timeout -k 10 250 python bench.py --no-detector --no-cpu-baseline --steps 50 > gpurun_out/r5f/final_bench_counters.json 2> /dev/null; echo rc=$?
timeout -k 10 250 python bench.py --gpus 1 --backend nccl --force-dist --steps 20 > gpurun_out/r5f/final_bench_nccl_n1.json 2> gpurun_out/r5f/final_bench_nccl_n1.err; echo rc=$?
timeout -k 10 250 python bench.py --gpus 2 --backend gloo --share-device --pairs 64 > gpurun_out/r5f/final_bench_n2.json 2> gpurun_out/r5f/final_bench_n2.err; echo rc=$?
timeout -k 10 250 python bench.py --config C2 --no-detector > gpurun_out/r5f/final_bench_c2.json 2> /dev/null; echo rc=$?
timeout -k 10 250 python bench.py --config C4 --steps 3 > gpurun_out/r5f/final_bench_c4.json 2> /dev/null; echo rc=$?
timeout -k 10 250 python bench.py --config C4 --clips 6766 --steps 1 > gpurun_out/r5f/final_bench_c4_6766.json 2> /dev/null; echo rc=$?
timeout -k 10 250 python bench.py --config C4 --gpus 2 --backend gloo --share-device > gpurun_out/r5f/final_bench_c4_n2.json 2> /dev/null; echo rc=$?
timeout -k 10 250 python bench.py --config C4 --gpus 1 --backend nccl --force-dist --steps 3 > gpurun_out/r5f/final_bench_c4_nccl_n1.json 2> /dev/null; echo rc=$?
timeout -k 10 250 python bench.py --config C4 --steps 3 --write > gpurun_out/r5f/final_bench_c4_write.json 2> /dev/null; echo rc=$?
timeout -k 10 250 python bench.py --config C4 --clips 6766 --steps 1 --write > gpurun_out/r5f/final_bench_c4_6766_write.json 2> /dev/null; echo rc=$?
timeout -k 10 250 python bench.py --config C4 --keypoints brisk --steps 2 > gpurun_out/r5f/final_bench_c4_brisk.json 2> /dev/null; echo rc=$?
timeout -k 10 250 python bench.py --config C4 --keypoints brisk --steps 2 --write > gpurun_out/r5f/final_bench_c4_brisk_write.json 2> /dev/null; echo rc=$?
timeout -k 10 250 python bench.py --config C5 --keypoints brisk --frames 2053 > gpurun_out/r5f/final_bench_c5_brisk.json 2> /dev/null; echo rc=$?
timeout -k 10 200 python mofreak_amd/tools/bench_format.py > gpurun_out/r5f/final_format_bench.jsonl 2> /dev/null; echo rc=$?
timeout -k 10 250 python bench.py --config C5 > gpurun_out/r5f/final_bench_c5.json 2> /dev/null; echo rc=$?
timeout -k 10 300 python bench.py --config C5 --frames 90000 > gpurun_out/r5f/final_bench_c5_90000.json 2> /dev/null; echo rc=$?
for set in FETCH_SIZE WRITE_SIZE; do timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/r5f/c2pmc_$set -- python3 bench.py --config C2 --steps 2 --warmup 1 --no-cpu-baseline --no-detector --no-sustain > /dev/null 2>&1; done
python3 - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(list)
for f in glob.glob('gpurun_out/r5f/c2pmc_*/*/*counter_collection.csv'):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if 'tile_kernel' in r['Kernel_Name']:
            per[(r['Dispatch_Id'], r['Counter_Name'])] += float(r['Counter_Value'])
    for (d, c), v in per.items():
        tot[c].append(v)
avg = {k: sum(v) / len(v) for k, v in tot.items()}
b_alg = (2 * 640 * 480 + 28 * 875) * 1000
traffic = (2 * avg.get('FETCH_SIZE', 0) + avg.get('WRITE_SIZE', 0)) * 1024
open('gpurun_out/r5f/final_c2_traffic.txt', 'w').write(f"C2 (1000 pairs 640x480, 875 keypoints): tile_kernel FETCH_SIZE {avg.get('FETCH_SIZE', 0):.0f} KB, WRITE_SIZE {avg.get('WRITE_SIZE', 0):.0f} KB per launch -> (2 F + W) * 1024 = {traffic / 1e6:.1f} MB = {traffic / b_alg:.3f} x the algorithmic {b_alg / 1e6:.1f} MB\n")
print(open('gpurun_out/r5f/final_c2_traffic.txt').read())
import hashlib, json
json.dump({"kernel": "tile_kernel", "config": "C2", "library_sha256_16": hashlib.sha256(open('mofreak_amd/libmofreak_hip.so', 'rb').read()).hexdigest()[:16], "pairs_per_launch": 1000,
           "FETCH_SIZE_KB": avg.get('FETCH_SIZE', 0), "WRITE_SIZE_KB": avg.get('WRITE_SIZE', 0), "tile_kernel_hbm_bytes_per_launch": traffic, "algorithmic_bytes_per_launch": b_alg,
           "traffic_over_algorithmic": traffic / b_alg, "correction": "(2*FETCH_SIZE + WRITE_SIZE) * 1024, separate --pmc passes"}, open('gpurun_out/r5f/traffic_C2.json', 'w'), indent=1)
PY
rm -rf gpurun_out/r5f/c2pmc_*
timeout -k 10 260 python tests/fuzz_parity_gpu.py 150 521 > gpurun_out/r5f/final_fuzz521.log 2>&1; echo fuzz rc=$?; tail -n 1 gpurun_out/r5f/final_fuzz521.log
MOFREAK_HIP_LIBRARY=mofreak_amd/libmofreak_hip_debug.so timeout -k 10 260 python tests/fuzz_parity_gpu.py 150 522 > gpurun_out/r5f/final_fuzz522.log 2>&1; echo fuzz debug rc=$?; tail -n 1 gpurun_out/r5f/final_fuzz522.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r5f/final_smoke.log 2>&1; echo smoke rc=$?; tail -n 1 gpurun_out/r5f/final_smoke.log
