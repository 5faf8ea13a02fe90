#!/bin/bash
# RANSAC kernel time with parts removed (timing only, results meaningless): 0 = full, 1 = no scoring, 2 = no plane fit
for a in 0 1 2; do
  OCTL_RANSAC_ABLATE=$a python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/abl_$a.json 2> gpurun_out/abl_$a.err || echo FAILED $a
done
python - <<'PY'
import json
for a in (0, 1, 2):
    d = json.load(open(f'gpurun_out/abl_{a}.json'))
    print('ablate', a, 'ransac %.3f ms' % d['kernels']['ransac']['ms_per_step'], 'step %.2f' % d['ms_per_step'])
PY
