"""tools/psnr_probe.py [steps] [masked] -- the training-parity comparison of tests/test_psnr_parity_gpu.py with THREE trajectories: the HIP
path (fp32), the oracle in fp64 and the oracle in fp32 -- the last one shows how far fp32 arithmetic alone (no kernel of ours involved)
moves the trajectory of this chaotic system away from the fp64 one.  GPU box; the oracle side takes ~0.15 s per step."""
import sys
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, "cloth-splatting_amd"); sys.path.insert(0, ".")
import test_psnr_parity_gpu as t

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 500
masked = len(sys.argv) > 2 and sys.argv[2] == "masked"
g, c64 = t.run_parity(masked, steps)
import torch
t.ORACLE_DTYPE[0] = np.float32
t.TORCH_DTYPE[0] = torch.float32      # the WHOLE CPU side in float32: rasterizer (C oracle), simulator, transform, losses, Adam
_, c32 = t.run_parity(masked, steps)
for name, a, b in (("hip - cpu64", g, c64), ("cpu32 - cpu64", c32, c64), ("hip - cpu32", g, c32)):
    d = np.abs(a - b)
    print(f"{name:22s} final {a[-1] - b[-1]:+.4f} dB  median {np.median(d):.4f}  p95 {np.percentile(d, 95):.4f}  max {d.max():.4f} at step {int(d.argmax()) + 1}")
print("step   hip   cpu fp64  cpu fp32")
for i in list(range(0, steps, max(steps // 25, 1))) + [steps - 1]:
    print(f"{i + 1:4d} {g[i]:8.3f} {c64[i]:8.3f} {c32[i]:8.3f}")
