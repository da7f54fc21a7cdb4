mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r4/all_tests.log 2>&1; echo product rc=$?
tail -n 3 gpurun_out/r4/all_tests.log
python mofreak_amd/tools/detector_probe.py 32 10 describe
python mofreak_amd/tools/detector_probe.py 128 6 loop
python mofreak_amd/tools/detector_probe.py 256 4 loop
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4/loopprof -- python3 mofreak_amd/tools/detector_probe.py 128 7 loop > /dev/null 2>&1
python3 - <<'PY'
import glob, csv
for f in glob.glob('gpurun_out/r4/loopprof/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if 'band_' in r['Name'] or 'describe' in r['Name']: print(r['Name'][:80], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
