mkdir -p gpurun_out/r4
timeout -k 10 400 python -m pytest tests/test_detector_gpu.py -x -q -m gpu > gpurun_out/r4/tie_tests.log 2>&1; echo product rc=$?
MOFREAK_HIP_LIBRARY=mofreak_amd/libmofreak_hip_debug.so timeout -k 10 400 python -m pytest tests/test_detector_gpu.py -x -q -m gpu > gpurun_out/r4/tie_tests_debug.log 2>&1; echo debug rc=$?
tail -n 2 gpurun_out/r4/tie_tests.log gpurun_out/r4/tie_tests_debug.log
python mofreak_amd/tools/detector_probe.py 32 12
python mofreak_amd/tools/detector_probe.py 32 12
python mofreak_amd/tools/detector_probe.py 128 6
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4/tieprof -- python3 mofreak_amd/tools/detector_probe.py 32 7 > /dev/null 2>&1
python3 - <<'PY'
import glob, csv
for f in glob.glob('gpurun_out/r4/tieprof/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if 'det_tie' in r['Name']: print(r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
