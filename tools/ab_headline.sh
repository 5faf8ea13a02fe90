#!/bin/bash
# A/B of library variants under build/variants: headline-only bench, 3 repetitions interleaved
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
        d = json.load(open(f))
        print(f.split('/')[-1][:-5], 'step %.3f ms' % d['ms_per_step'], 'ransac %.3f' % d['roofline']['launch_ms'], 'kept', d['config']['points_after_ransac'])
    except Exception as e:
        print(f, 'ERR', e)
PY
