"""tools/psnr_probe.py [steps] -- run tests/test_psnr_parity_gpu.py's training-parity comparison with a different step count and print the
trajectories (GPU box; the oracle side takes ~1.2 s per step)."""
import sys, os, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, "cloth-splatting_amd"); sys.path.insert(0, ".")
import test_psnr_parity_gpu as t
t.STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 80
orig_assert = None
# monkeypatch: capture trajectories by wrapping np.abs? simpler: copy of the test with prints -> run and catch assertion
import builtins
try:
    t.test_psnr_parity_hip_vs_oracle_training()
except AssertionError as e:
    print("ASSERT", str(e)[:200])
