#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for rep in 1 2; do
for v in base wgs3; do
  if [ $v = base ]; then unset OCTREELIB_AMD_LIB; else export OCTREELIB_AMD_LIB=$R/build/variants/$v.so; fi
  python bench.py --no-secondary --no-cpu-baseline > gpurun_out/b10_head_${v}_$rep.json 2>/dev/null
done; done
export OCTREELIB_AMD_LIB=$R/build/variants/wgs3.so
python bench.py --workload c4 --steps 4 > gpurun_out/b10_c4_wgs3.json 2>/dev/null
python bench.py --workload c5shard --steps 3 > gpurun_out/b10_c5_wgs3.json 2>/dev/null
unset OCTREELIB_AMD_LIB
python bench.py --workload c4 --steps 4 > gpurun_out/b10_c4_base.json 2>/dev/null
python bench.py --workload c5shard --steps 3 > gpurun_out/b10_c5_base.json 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/b10_head_*.json')):
    d = json.load(open(f)); k = d['kernels']
    print(f.split('b10_')[1], 'step %.3f' % d['ms_per_step'], ' '.join('%s %.3f' % (n, k[n]['ms_per_step']) for n in ('part_hist', 'part_scatter', 'bucket_build', 'bucket_nodes', 'apply_mask', 'ransac')))
for f in sorted(glob.glob('gpurun_out/b10_c4_*.json')):
    d = json.load(open(f)); print(f.split('b10_')[1], d['subdivide_ms'], {k: round(v['ms_per_step'], 3) for k, v in d['roofline_build']['all'].items()})
for f in sorted(glob.glob('gpurun_out/b10_c5_*.json')):
    d = json.load(open(f)); print(f.split('b10_')[1], d['ms'], d['insert_subdivide_only_ms'], {k: round(v['ms_per_step'], 3) for k, v in d['kernels'].items() if k.startswith(('part', 'bucket'))})
PY
