#!/bin/bash
# Profiles of the benchmark command for profiles/ (run on the GPU box, from the repo root):
#   kernel stats (rocprofv3 --kernel-trace --stats), the SQ counter passes and the three HBM traffic passes,
#   each --pmc set in a run of its own, summarised per tile_kernel launch into gpurun_out/<tag>_*.json / .csv.
# usage: bash mofreak_amd/tools/profile_tile.sh r02
set -e
tag=${1:-r02}
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd $root
out=gpurun_out/prof_$tag
rm -rf $out && mkdir -p $out
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-detector --no-sustain > $out/bench_stats.log 2>&1
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_bench_kernel_stats.csv
grep "^{" $out/bench_stats.log > gpurun_out/${tag}_bench_line_under_rocprof.json || true
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc$i -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-detector --no-sustain > $out/pmc$i.log 2>&1
done
python3 - $tag $out <<'PY'
import csv, glob, collections, hashlib, json, sys
tag, out = sys.argv[1], sys.argv[2]
lib_hash = hashlib.sha256(open('mofreak_amd/libmofreak_hip.so', 'rb').read()).hexdigest()[:16]  # the build these counters belong to
tot = collections.defaultdict(list)
rows = []
for f in sorted(glob.glob(f'{out}/pmc*/*/*counter_collection.csv')):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if 'tile_kernel' in r['Kernel_Name']:
            per[r['Dispatch_Id']][r['Counter_Name']] += float(r['Counter_Value'])
    for d, c in per.items():
        for k, v in c.items():
            tot[k].append(v)
pairs, n_kp = 256, 29106
n = pairs * n_kp
avg = {k: sum(v) / len(v) for k, v in tot.items()}
with open(f'gpurun_out/{tag}_pmc_tile_kernel.csv', 'w') as fh:
    fh.write('counter,launches,avg_per_launch,per_descriptor\n')
    for k in sorted(avg):
        fh.write(f'{k},{len(tot[k])},{avg[k]:.1f},{avg[k] / n:.3f}\n')
b_alg = 2 * 1920 * 1080 + 28 * n_kp
traffic = (2 * avg['FETCH_SIZE'] + avg['WRITE_SIZE']) * 1024
pd = {k: round(avg[k] / n, 3) for k in avg if k.startswith('SQ_')}
tj = {"kernel": "tile_kernel", "library_sha256_16": lib_hash, "pairs_per_launch": pairs, "FETCH_SIZE_KB": avg['FETCH_SIZE'], "WRITE_SIZE_KB": avg['WRITE_SIZE'],
      "TCC_HIT_sum": avg['TCC_HIT_sum'], "TCC_MISS_sum": avg['TCC_MISS_sum'],
      "l2_hit_rate": avg['TCC_HIT_sum'] / (avg['TCC_HIT_sum'] + avg['TCC_MISS_sum']),
      "tile_kernel_hbm_bytes_per_launch": traffic,
      "correction": "(2*FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE counts 64 B per 128-B request on gfx950 for wide coalesced reads (MI355X_MICROARCH.md, HBM), WRITE_SIZE is exact; separate --pmc passes",
      "algorithmic_bytes_per_launch": b_alg * pairs, "traffic_over_algorithmic": traffic / (b_alg * pairs),
      "source": f"profiles/{tag}_pmc_tile_kernel.csv",
      "per_descriptor": dict(pd, note="wave-level counters of one tile_kernel launch divided by its 7 451 136 descriptors; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* in quad-cycles summed over a CU's waves, SQ_BUSY_CU_CYCLES / SQ_LDS_* in cycles per CU"),
      "valu_wave_instr_per_descriptor": pd['SQ_INSTS_VALU'], "valu_cycles_per_wave_instr": 2.0,
      "valu_note": "peak = one wave64 instruction per 2 cycles per SIMD (the full-rate class: add/sub/logic/right shifts/f32 fma); most integer ops used here (mul24/mad24, dot, sad, perm, cvt, left shifts, DPP, SDWA) were measured at 1.6x that cost (profiles/r02_valu_microbench.txt)",
      "ratios": {"wait_any_over_wave_cycles": pd['SQ_WAIT_ANY'] / pd['SQ_WAVE_CYCLES'], "lds_conflict_over_lds_active_inst": pd['SQ_LDS_BANK_CONFLICT'] / max(pd['SQ_ACTIVE_INST_LDS'], 1e-9),
                 "lds_conflict_over_lds_idx_active": pd['SQ_LDS_BANK_CONFLICT'] / pd['SQ_LDS_IDX_ACTIVE'], "valu_busy": pd['SQ_ACTIVE_INST_VALU'] / pd['SQ_BUSY_CU_CYCLES'],
                 "lds_busy": pd['SQ_LDS_IDX_ACTIVE'] / pd['SQ_BUSY_CU_CYCLES']}}
json.dump(tj, open(f'gpurun_out/{tag}_traffic.json', 'w'), indent=1)
print(json.dumps(tj['ratios']), tj['traffic_over_algorithmic'], pd['SQ_INSTS_VALU'])
PY
