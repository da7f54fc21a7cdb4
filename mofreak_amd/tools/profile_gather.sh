set -e
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd $root
out=gpurun_out/prof_gather
rm -rf $out && mkdir -p $out
export AB_STEPS=2
for band in 0; do
  export AB_SORT_BAND=$band
  for set in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD"; do
    tag=$(echo $set | tr ' ' '_' | cut -c1-20)
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/b${band}_$tag -- python3 mofreak_amd/tools/ab_gather.py --one $root/mofreak_amd/libmofreak_hip.so > $out/b${band}_$tag.log 2>&1
  done
done
python3 - $out <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
for band in (0,):
    tot = collections.defaultdict(list)
    for f in sorted(glob.glob(f'{out}/b{band}_*/*/*counter_collection.csv')):
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            if 'describe_kernel' in r['Kernel_Name']:
                per[r['Dispatch_Id']][r['Counter_Name']] += float(r['Counter_Value'])
        for d, c in per.items():
            for k, v in c.items():
                tot[k].append(v)
    print('band', band, {k: round(sum(v) / len(v)) for k, v in sorted(tot.items())})
PY
