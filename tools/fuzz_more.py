import sys, traceback
sys.path.insert(0, '.')
from tests.test_gpu_fuzz import test_random_operation_sequences_vs_oracle as f
bad = 0
lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (16, 96)
for seed in range(lo, hi):
    try:
        f(seed)
    except Exception:
        bad += 1
        print("FAILED seed", seed); traceback.print_exc()
    if seed % 10 == 0:
        print("seed", seed, flush=True)
print("done, failures:", bad)
