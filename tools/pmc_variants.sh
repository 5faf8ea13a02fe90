#!/bin/bash
# SQ counters of k_ransac for every library variant in build/variants (run through gpurun)
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for so in $R/build/variants/*.so; do
  name=$(basename $so .so)
  export OCTREELIB_AMD_LIB=$so
  rocprofv3 --kernel-trace --pmc ${PMC:-SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE} --output-format csv -d $R/gpurun_out/pmcv_$name -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 || echo "FAILED $name"
done
cd $R && python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmcv_*')):
    f = glob.glob(d + '/*/*counter_collection.csv')
    if not f: print(d, 'no counters'); continue
    per = collections.defaultdict(float); disp = set()
    for r in csv.DictReader(open(f[0])):
        if 'k_ransac<' in r['Kernel_Name'] or 'k_ransacI' in r['Kernel_Name']:
            per[r['Counter_Name']] += float(r['Counter_Value']); disp.add(r['Dispatch_Id'])
    n = max(len(disp), 1)
    print(d.split('pmcv_')[1], {k: round(v / n / 1e6, 1) for k, v in per.items()}, 'launches', n)
PY
