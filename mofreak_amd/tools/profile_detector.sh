#!/bin/bash
# Profiles of the detector (+ descriptors) for profiles/ (run on the GPU box, from the repo root):
#   kernel stats of detector_probe.py at 32 pairs per call with and without the descriptors, and SQ / HBM counter passes,
#   each --pmc set in a run of its own, summarised per kernel into gpurun_out/<tag>_detector_*.csv
# usage: bash mofreak_amd/tools/profile_detector.sh r02
set -e
tag=${1:-r02}
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd $root
out=gpurun_out/prof_det_$tag
rm -rf $out && mkdir -p $out
probe=mofreak_amd/tools/detector_probe.py
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $probe 32 6 > $out/stats.log 2>&1
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_detector_kernel_stats.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_dd -- python3 $probe 32 6 describe > $out/stats_dd.log 2>&1
cp $(find $out/stats_dd -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_detector_describe_kernel_stats.csv
# the probe's own figures come from plain runs (under the profiler the host side of every launch is slower)
{ timeout -k 10 120 python3 $probe 32 12; timeout -k 10 120 python3 $probe 32 12 describe; } 2>&1 | grep -h "pairs=" > gpurun_out/${tag}_detector_probe_lines.txt || true
grep -h "pairs=" $out/stats.log $out/stats_dd.log | sed "s/^/under rocprofv3: /" >> gpurun_out/${tag}_detector_probe_lines.txt || true
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVES" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc$i -- python3 $probe 32 2 describe > $out/pmc$i.log 2>&1
done
python3 - $tag $out <<'PY'
import csv, glob, collections, re, sys
tag, out = sys.argv[1], sys.argv[2]
def short(n):
    m = re.search(r'(det_\w+?_kernel|describe_kernel|tile_kernel|band_\w*kernel|bin_\w+?_kernel)', n)
    return m.group(1) if m else None
tot = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(set)
for f in sorted(glob.glob(f'{out}/pmc*/*/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        k = short(r['Kernel_Name'])
        if k is None:
            continue
        tot[k][r['Counter_Name']] += float(r['Counter_Value'])
        launches[(k, r['Counter_Name'])].add(r['Dispatch_Id'])
calls = 3  # probe: 1 warm-up + 2 timed calls of 32 pairs
names = sorted({c for k in tot for c in tot[k]})
with open(f'gpurun_out/{tag}_detector_pmc.csv', 'w') as fh:
    # FETCH_SIZE counts a coalesced read at half its bytes and a scattered 64-byte fetch exactly (profiles/r03_fetch_calibration.txt)
    scattered = {'describe_kernel', 'det_window_kernel', 'det_walk_kernel', 'det_tie_kernel', 'det_tie_first_kernel', 'det_candidates_kernel'}
    fh.write('kernel,launches_per_call,' + ','.join(f'{c}_per_call' for c in names) + ',valu_busy,lds_busy,wait_any_over_wave_cycles,hbm_MB_per_call,hbm_MB_per_call_fetch_x1,hbm_MB_per_call_fetch_x2,fetch_shape\n')
    for k in sorted(tot, key=lambda k: -tot[k].get('SQ_BUSY_CU_CYCLES', 0)):
        c = tot[k]
        per = {n: c.get(n, 0.0) / calls for n in names}
        busy = per.get('SQ_BUSY_CU_CYCLES', 0) or 1.0
        lo = (per.get('FETCH_SIZE', 0) + per.get('WRITE_SIZE', 0)) * 1024 / 1e6
        hi = (2 * per.get('FETCH_SIZE', 0) + per.get('WRITE_SIZE', 0)) * 1024 / 1e6
        hbm = lo if k in scattered else hi
        nl = len(launches[(k, 'SQ_WAVES')]) / calls if (k, 'SQ_WAVES') in launches else 0
        fh.write(f"{k},{nl:.1f}," + ','.join(f'{per[n]:.0f}' for n in names) +
                 f",{per.get('SQ_ACTIVE_INST_VALU', 0) / busy:.3f},{per.get('SQ_LDS_IDX_ACTIVE', 0) / busy:.3f},"
                 f"{per.get('SQ_WAIT_ANY', 0) / max(per.get('SQ_WAVE_CYCLES', 0), 1):.3f},{hbm:.1f},{lo:.1f},{hi:.1f},"
                 f"{'scattered 64-B lines (FETCH_SIZE exact)' if k in scattered else 'coalesced (FETCH_SIZE x2)'}\n")
print(open(f'gpurun_out/{tag}_detector_pmc.csv').read())
PY
