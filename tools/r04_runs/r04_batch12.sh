#!/bin/bash
# how the partition's cost per point depends on the size of what it writes (does the memory-side cache merge the scattered records?)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for n in 1250000 2500000 5000000 7500000 10000000; do
  python bench.py --points $n --no-secondary --no-cpu-baseline --steps 10 > gpurun_out/b12_n$n.json 2>/dev/null
done
python - <<'PY'
import json, glob
for n in (1250000, 2500000, 5000000, 7500000, 10000000):
    d = json.load(open('gpurun_out/b12_n%d.json' % n)); k = d['kernels']
    print(n, 'step %.3f' % d['ms_per_step'], ' '.join('%s %.1f ps/pt' % (m, k[m]['ms_per_step'] * 1e9 / n) for m in ('part_hist', 'part_scatter', 'bucket_build', 'bucket_nodes', 'apply_mask', 'ransac')))
PY
