mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r4/all_tests.log 2>&1; echo product rc=$?
tail -n 3 gpurun_out/r4/all_tests.log
python mofreak_amd/tools/detector_probe.py 32 10 describe
python mofreak_amd/tools/detector_probe.py 128 6 describe
python mofreak_amd/tools/detector_probe.py 128 6 loop
python mofreak_amd/tools/detector_probe.py 256 4 loop
