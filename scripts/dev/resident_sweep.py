"""dev tool: tests/test_gpu_windowed.py's oracle comparison over random (N, B, blocks) in the resident mode's range"""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bluerov2_amd as ba
from oracle.oracle_ffi import Oracle
import test_gpu_windowed as T
orc = Oracle()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for it in range(n):
    N = int(rng.integers(24, 82)); B = int(rng.choice([1, 2, 3, 5, 17, 40, 64, 100, 200, 256]))
    blocks = None if rng.random() < 0.5 or B < 3 else int(rng.integers(1, B))
    try:
        s, *_ = T._run_against_oracle(ba, orc, N, B, ticks=3, blocks=blocks)
        s.close()
    except AssertionError as e:
        msg = str(e)[:200]
        if "n_qp" in msg or msg == "":
            print(f"N={N} B={B} blocks={blocks}: workload assertion (no active bounds in this draw)")
        else:
            bad += 1; print(f"N={N} B={B} blocks={blocks}: FAILED {msg}")
print("failures:", bad, "of", n)
