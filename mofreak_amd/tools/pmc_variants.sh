#!/bin/bash
# SQ counters of tile-kernel builds (the product and mofreak_amd/_exp/libvar_*.so experiment / ablation builds) on the metric's
# workload, one rocprofv3 --pmc pass per library: per-descriptor vector / scalar / LDS instruction counts and busy fractions.
# usage (GPU box, repo root): bash mofreak_amd/tools/pmc_variants.sh OUT.txt LIB [LIB ...]
out=$1; shift
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd $root
: > $out
for lib in "$@"; do
  d=gpurun_out/pmcv_$(basename $lib .so)
  rm -rf $d
  AB_STEPS=2 timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $d -- python3 mofreak_amd/tools/ab_tile.py --one $lib > $d.log 2>&1
  python3 - $lib $d >> $out <<'PY'
import csv, glob, collections, sys
lib, d = sys.argv[1], sys.argv[2]
per = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(f'{d}/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'tile_kernel' in r['Kernel_Name']:
            per[r['Dispatch_Id']][r['Counter_Name']] += float(r['Counter_Value'])
n = 256 * 29106
tot = collections.defaultdict(list)
for c in per.values():
    for k, v in c.items():
        tot[k].append(v)
a = {k: sum(v) / len(v) / n for k, v in tot.items()}
if a:
    print(f"{lib.split('/')[-1]:24s} launches {len(per)}  VALU {a['SQ_INSTS_VALU']:.1f}  SALU {a['SQ_INSTS_SALU']:.1f}  LDS {a['SQ_INSTS_LDS']:.1f}  CU cycles {a['SQ_BUSY_CU_CYCLES']:.1f}  "
          f"valu_busy {a['SQ_ACTIVE_INST_VALU'] / a['SQ_BUSY_CU_CYCLES']:.3f}  lds_busy {a['SQ_LDS_IDX_ACTIVE'] / a['SQ_BUSY_CU_CYCLES']:.3f}  waiting {a['SQ_WAIT_ANY'] / a['SQ_WAVE_CYCLES']:.3f}")
else:
    print(lib, "no counters")
PY
  rm -rf $d
done
cat $out
