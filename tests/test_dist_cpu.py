"""world_size-2 gloo test of the view-parallel helpers (csplat/dist.py) -- the N > 1 path of bench.py / SURVEY 8(e)."""
import os
import sys

import numpy as np
import pytest

import util  # noqa: F401  (sys.path)

torch = pytest.importorskip("torch")


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.join(util.ROOT, "cloth-splatting_amd"))
    from csplat import dist as cd
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        views = list(range(7))
        mine = cd.shard_views(views)
        gen = torch.Generator().manual_seed(100)            # identical "replicated" parameters on both ranks
        params = [torch.nn.Parameter(torch.randn(50, 3, generator=gen)), torch.nn.Parameter(torch.randn(50, 16, 3, generator=gen)),
                  torch.nn.Parameter(torch.randn(50, 1, generator=gen))]
        # each rank's loss only sees its own views; parameter 2 gets no gradient on rank 1
        loss = sum((v + 1) * (params[0] ** 2).sum() + (v + 2) * params[1].sum() for v in mine)
        if rank == 0:
            loss = loss + params[2].sum()
        loss.backward()
        flat = cd.allreduce_gradients(params)
        vg = torch.full((50, 3), float(rank + 1))
        radii = torch.arange(50) * (rank + 1)
        vis = (torch.arange(50) % 2 == rank)
        vg, radii, vis = cd.reduce_densification_stats(vg, radii, vis)
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), mine=np.array(mine), g0=params[0].grad.numpy(), g1=params[1].grad.numpy(),
                 g2=params[2].grad.numpy(), flat=flat.numpy(), vg=vg.numpy(), radii=radii.numpy(), vis=vis.numpy(),
                 p0=params[0].detach().numpy())
    finally:
        dist.destroy_process_group()


def test_view_parallel_allreduce_gloo(tmp_path):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "r0.npz"), np.load(tmp_path / "r1.npz")
    assert list(r0["mine"]) == [0, 2, 4, 6] and list(r1["mine"]) == [1, 3, 5]
    for k in ("g0", "g1", "g2", "flat", "vg", "radii", "vis"):
        np.testing.assert_array_equal(r0[k], r1[k])          # replicas stay identical
    w = sum(v + 1 for v in range(7))
    np.testing.assert_allclose(r0["g0"], 2 * w * r0["p0"], rtol=1e-6)
    np.testing.assert_allclose(r0["g1"], float(sum(v + 2 for v in range(7))))
    np.testing.assert_allclose(r0["g2"], 1.0)
    np.testing.assert_allclose(r0["vg"], 3.0)
    np.testing.assert_array_equal(r0["radii"], np.arange(50) * 2)
    assert r0["vis"].all()


def test_single_process_is_identity():
    from csplat import dist as cd
    assert cd.shard_views([1, 2, 3]) == [1, 2, 3]
    p = torch.nn.Parameter(torch.ones(4)); (p * 2).sum().backward()
    cd.allreduce_gradients([p])
    np.testing.assert_allclose(p.grad.numpy(), 2.0)
