#!/bin/bash
# run bench.py (short) once per library variant in build/variants; prints ransac ms per variant
for so in build/variants/*.so; do
  name=$(basename $so .so)
  OCTREELIB_AMD_LIB=$PWD/$so python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err || echo "FAILED $name"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/ab_*.json')):
    try:
        d = json.load(open(f))
        print(f.split('ab_')[1][:-5], 'ransac %.3f ms' % d['kernels']['ransac']['ms_per_step'], 'step %.2f ms' % d['ms_per_step'], 'kept', d['config']['points_after_ransac'])
    except Exception as e:
        print(f, 'ERR', e)
PY
