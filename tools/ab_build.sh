#!/bin/bash
# A/B of library variants under build/variants on the build kernels: headline-only bench, 3 interleaved repetitions
mkdir -p gpurun_out/ab
for rep in 1 2 3; do
for so in build/variants/*.so; do
  name=$(basename $so .so)
  OCTREELIB_AMD_LIB=$PWD/$so python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/ab/${name}_$rep.json 2> gpurun_out/ab/${name}_$rep.err || echo "FAILED $name"
done; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/ab/*.json')):
    try:
        d = json.load(open(f)); k = d['kernels']
        print(f.split('/')[-1][:-5], 'step %.3f' % d['ms_per_step'], ' '.join('%s %.3f' % (n, k[n]['ms_per_step']) for n in ('part_hist', 'part_scatter', 'bucket_build', 'bucket_nodes', 'apply_mask', 'ransac') if n in k), 'kept', d['config']['points_after_ransac'])
    except Exception as e:
        print(f, 'ERR', e)
PY
