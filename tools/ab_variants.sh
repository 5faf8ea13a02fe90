#!/bin/bash
# run bench.py (short) once per library variant in build/variants; prints key timings per variant
for so in build/variants/*.so; do
  name=$(basename $so .so)
  OCTREELIB_AMD_LIB=$PWD/$so python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err || echo "FAILED $name"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/ab_*.json')):
    try:
        d = json.load(open(f)); k = d['kernels']
        print(f.split('ab_')[1][:-5], 'step %.2f ms' % d['ms_per_step'], 'ransac %.3f' % k['ransac']['ms_per_step'],
              'keygen %.3f' % k['keygen']['ms_per_step'], 'build-only %.3f' % d.get('secondary', {}).get('insert_subdivide_only', {}).get('ms', 0),
              'kept', d['config']['points_after_ransac'])
    except Exception as e:
        print(f, 'ERR', e)
PY
