#!/usr/bin/env python3
"""More seeds for the tests that build the same forest through the bucket path and through the
level-synchronous path and compare everything bit for bit (run on the GPU box)."""
import sys, traceback
sys.path.insert(0, '.')
from _pytest.monkeypatch import MonkeyPatch
import tests.test_gpu_parity as T

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = [(T.test_voxel_local_build_random_voxels_vs_level_synchronous_build, range(10, 10 + 60 * scale)),
         (T.test_bucket_build_mixed_voxel_populations_vs_level_synchronous_build, range(6, 6 + 24 * scale)),
         (T.test_bucket_build_two_pass_partition_vs_level_synchronous_build, range(3, 3 + 9 * scale))]
bad = 0
for fn, seeds in cases:
    for seed in seeds:
        mp = MonkeyPatch()
        try:
            fn(mp, seed)
        except Exception:
            bad += 1
            print("FAILED", fn.__name__, seed); traceback.print_exc()
        finally:
            mp.undo()
            from octreelib_amd import _native as nat
            nat.get_context().reset_options()   # (the tests switch code paths with Context.set_option)
    print(fn.__name__, "done", flush=True)
print("failures:", bad)
