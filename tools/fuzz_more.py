import sys, traceback
sys.path.insert(0, '.')
from tests.test_gpu_fuzz import test_random_operation_sequences_vs_oracle as f
bad = 0
for seed in range(16, 96):
    try:
        f(seed)
    except Exception:
        bad += 1
        print("FAILED seed", seed); traceback.print_exc()
print("done, failures:", bad)
