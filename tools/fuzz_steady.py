"""One-off fuzz of the steady blocks: the randomized on/off comparison of tests/test_gpu_parity.py over many more seeds."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import conftest, test_gpu_parity as T
pkg = conftest.load_package()
bad = 0
first, last = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (5, 65)
for seed in range(first, last):
    try:
        T.test_steady_blocks_random_plans(pkg, seed)
    except AssertionError as e:
        bad += 1
        print("seed", seed, "FAILED:", str(e)[:200])
print("fuzz done, failures:", bad)
