"""Child process of tests/test_rccl_gpu.py (a FRESH interpreter: the parent has touched the GPU, and RCCL wants its own process):
one rank, backend "nccl" (= RCCL), CSPLAT_FORCE_DIST=1 so that every collective of the view-parallel step is really issued.
Writes a JSON verdict to argv[1]."""
import json
import os
import sys

os.environ.setdefault("CSPLAT_FORCE_DIST", "1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.update(MASTER_ADDR="127.0.0.1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
os.environ.setdefault("MASTER_PORT", str(29500 + (os.getpid() % 2000)))
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import util  # noqa: E402,F401
import test_dist_gpu as tdg  # noqa: E402


def main(out_path):
    from csplat import dist as cd, native, train as tr
    res = {}
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    plain = tdg._run(False, steps=3)                      # the one-rank step, no process group yet
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        assert cd.FORCE_DIST and cd.is_dist() and cd.world_rank() == (1, 0)
        res["backend"] = dist.get_backend()
        # a bare collective first: the library loads, the communicator initialises, sums over one rank are the identity
        t = torch.arange(1 << 20, dtype=torch.float32, device=dev)
        w = dist.all_reduce(t, async_op=True)
        w.wait()
        torch.cuda.synchronize()
        res["bare_allreduce_ok"] = bool(torch.equal(t.cpu(), torch.arange(1 << 20, dtype=torch.float32)))
        # the real step, view_parallel=True: FlatGrads binds the gradients, the early slice leaves from the autograd hook (steps 2, 3),
        # the rest + tail at the end of the step, all through RCCL
        native.lib.csplat_debug_flags(tdg.DET)
        try:
            pc, sim, mopt, cams, bg = tdg._build(dev)
            forced = {}
            for it in range(1, 4):
                ps, loss, stats = tr.train_step(it, cams, pc, sim, mopt, background=bg, view_parallel=True, time_allreduce=(it == 3))
                forced[f"psnr{it}"], forced[f"loss{it}"] = float(ps), float(loss)
                forced[f"vsg{it}"] = stats["viewspace_grad"].cpu().numpy().copy()
                forced[f"radii{it}"] = stats["radii"].cpu().numpy().copy()
            for i, p in enumerate(list(pc.parameters()) + list(sim.parameters())):
                forced[f"p{i}"] = p.detach().cpu().numpy().copy()
            forced["adam_has_state"] = np.array([int(len(pc.optimizer.state.get(p, {})) > 0) for p in pc.parameters()])
            torch.cuda.synchronize()
        finally:
            native.lib.csplat_debug_flags(0)
        fg = pc._flat_grads
        res["early_fired"] = int(fg.early_fired)
        res["allreduce_ms_step3"] = float(stats["allreduce_ms"])
        res["flat_bytes"] = int(fg.flat.numel() * 4)
        diffs = {}
        for k, v in forced.items():
            a, b = np.asarray(v), np.asarray(plain[k])
            if not np.array_equal(a, b):
                diffs[k] = float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max())
        res["differs_from_plain_step"] = diffs
    finally:
        dist.destroy_process_group()
    json.dump(res, open(out_path, "w"))


if __name__ == "__main__":
    main(sys.argv[1])
