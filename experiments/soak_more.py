"""The random-mix soak of tests/test_gpu_staged.py with more seeds (run on the GPU box)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import basebandboard_amd as gpu
import oracle
import test_gpu_staged as T
for seed in range(10, 10 + int(os.environ.get("SEEDS", "12"))):
    T.test_random_mix_of_calls_on_one_staged_handle(gpu, oracle, seed)
    print("seed", seed, "ok", flush=True)
