"""tools/psnr_spread.py [steps] [runs] -- the HIP training of tests/test_psnr_parity_gpu.py repeated in the DEFAULT mode (K7 sums with float
atomics: the summation order, hence the rounding, differs from run to run) -> the spread of the PSNR trajectory that rounding noise
alone produces on this system.  GPU box."""
import sys
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, "cloth-splatting_amd"); sys.path.insert(0, ".")
import test_psnr_parity_gpu as t
from csplat import native

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 500
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 5
t.STEP_HOOK[0] = lambda *a: None          # (one HIP run per call)
orig = native.lib.csplat_debug_flags
trajs = []
for r in range(runs + 1):
    # run 0: the bit-reproducible mode (what the test uses); runs 1..: atomics
    native.lib.csplat_debug_flags = (lambda f: orig(f)) if r == 0 else (lambda f: orig(0))
    g, _ = t.run_parity(False, steps, hip_only=True)
    trajs.append(g)
    print(f"run {r} ({'DET' if r == 0 else 'atomics'}): final {g[-1]:.4f} dB; at steps 300/400/450: {g[299]:.3f} {g[399]:.3f} {g[449]:.3f}", flush=True)
native.lib.csplat_debug_flags = orig
T = np.stack(trajs)
print("final PSNR: min %.4f max %.4f spread %.4f dB; largest spread along the trajectory %.4f dB at step %d" %
      (T[:, -1].min(), T[:, -1].max(), T[:, -1].max() - T[:, -1].min(), (T.max(0) - T.min(0)).max(), int((T.max(0) - T.min(0)).argmax()) + 1))
