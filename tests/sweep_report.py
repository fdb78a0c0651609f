"""Extended randomised differential sweep (the seeds beyond the ones pytest runs): python tests/sweep_report.py [first] [count]
Runs test_fused_train_random_shapes_vs_oracle for many more seeds and prints the failures (none expected)."""
import sys
import traceback

import test_gpu_parity as T

first = int(sys.argv[1]) if len(sys.argv) > 1 else 10
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = []
for seed in range(first, first + count):
    try:
        T.test_fused_train_random_shapes_vs_oracle(seed)
    except Exception:                                  # noqa: BLE001 - report and go on
        bad.append(seed)
        print("seed", seed, "FAILED:", traceback.format_exc().strip().splitlines()[-1][:300])
print(f"{count - len(bad)} / {count} seeds passed; failed: {bad}")
