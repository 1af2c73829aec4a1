#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on synthetic scene_1: rendered Mpix/s of the rasterizer hot path, forward +
backward, P = 100k Gaussians, 4 cameras 800x800 per GPU (BASELINE.json configs[1]).

A "step" = one pass of the hot path over one batch: for each of the rank's 4 views, GaussianRasterizer forward
(K1..K6), L1 loss against a fixed target image, backward (K7, K8); with N > 1 ranks (view-parallel, one process per
GPU) the step ends with ONE RCCL all-reduce of the flat per-Gaussian gradient buffer.
All inputs are resident in HBM before the timed region.  Prints ONE JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import contextlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "cloth-splatting_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec); ~6.3 TB/s achievable


def cpu_baseline(scene, cams, P, W, H):
    """Oracle (C restatement, OpenMP) timed on the host cores: ONE step (all views, fwd+bwd) of the same workload."""
    from csplat import synthetic as syn
    from oracle import raster_oracle as ro
    ro.build()
    g = syn.gaussians_at(scene)
    rng = np.random.default_rng(0)
    dpix = rng.normal(size=(3, H, W)).astype(np.float32)
    t0 = time.perf_counter()
    for cam in cams:
        o = ro.forward(g["means3D"], g["opacities"], cam["world_view_transform"], cam["full_proj_transform"],
                       cam["camera_center"], cam["tanfovx"], cam["tanfovy"], W, H, scene["bg"], shs=g["shs"],
                       sh_degree=3, scales=g["scales"], rotations=g["rotations"])
        ro.backward(o, dpix)
    dt = time.perf_counter() - t0
    return {"value": round(len(cams) * W * H / 1e6 / dt, 4), "unit": "Mpix/s", "cores": int(ro.num_threads()),
            "kind": "port",
            "sample": f"one full step: {len(cams)} views {W}x{H}, P={P}, fwd+bwd through oracle/raster_ref.c "
                      f"(OpenMP, fp32), {dt:.2f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--P", type=int, default=100_000)
    ap.add_argument("--res", type=int, default=800)
    ap.add_argument("--views", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train-step", action="store_true",
                    help="skip the auxiliary config-3 train-step measurement (bench_train.py) appended on 1 GPU")
    ap.add_argument("--no-view-streams", dest="view_streams", action="store_false",
                    help="run the views of a step back to back on one stream instead of one HIP stream per view")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # (CSPLAT_BENCH_BACKEND=gloo lets the N>1 code path be exercised on a 1-GPU box: ranks then share the device)
    backend = os.environ.get("CSPLAT_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from csplat import native, synthetic as syn
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

    P, W, H, V = args.P, args.res, args.res, args.views
    scene = syn.scene_1(P=P, W=W, H=H, n_cams=V)
    # view-parallel: rank r renders its own 4 cameras (azimuth offset), Gaussians replicated
    cams = [syn.make_camera(-180.0 + 360.0 * (k + rank / max(world, 1)) / V, W, H) for k in range(V)]
    g = syn.gaussians_at(scene)
    T = lambda a, rg=False: torch.tensor(np.asarray(a, np.float32), device=dev, requires_grad=rg)  # noqa: E731
    params = {k: T(g[k], True) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
    bg = T(scene["bg"])
    settings = [GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=c["tanfovx"], tanfovy=c["tanfovy"], bg=bg, scale_modifier=1.0,
        viewmatrix=T(c["world_view_transform"]), projmatrix=T(c["full_proj_transform"]), sh_degree=3,
        campos=T(c["camera_center"]), prefiltered=False, debug=False) for c in cams]

    def render(i, means2D):
        return GaussianRasterizer(settings[i])(means3D=params["means3D"], means2D=means2D, opacities=params["opacities"],
                                               shs=params["shs"], scales=params["scales"], rotations=params["rotations"])

    # target images = render of a perturbed copy (so the loss and its gradient are non-trivial)
    with torch.no_grad():
        keep = {k: v.detach().clone() for k, v in params.items()}
        gen = torch.Generator(device=dev).manual_seed(1234)
        params["shs"].add_(0.2 * torch.randn(params["shs"].shape, device=dev, generator=gen))
        params["means3D"].add_(0.003 * torch.randn(params["means3D"].shape, device=dev, generator=gen))
        targets = [render(i, torch.zeros(P, 3, device=dev))[0].clone() for i in range(V)]
        for k, v in keep.items():
            params[k].copy_(v)

    targets_stacked = torch.stack(targets).contiguous()
    flat_names = ("means3D", "opacities", "shs", "scales", "rotations")
    R_per_view = [0] * V

    # independent views run on separate HIP streams (diff_gaussian_rasterization.rasterize_views: every view's
    # csplat_forward_begin is issued before the first csplat_forward_finish): the compositing kernels of one view leave
    # most SIMDs idle (their critical path is the deepest 8x8 quadrant), so the views' kernels overlap on the chip
    from diff_gaussian_rasterization import rasterize_views
    from csplat.train import l1_loss

    def step():
        for p in params.values():
            p.grad = None
        m2ds = [torch.zeros(P, 3, device=dev, requires_grad=True) for _ in range(V)]
        if args.view_streams:   # all forwards first (train_step renders every camera, then calls backward once)
            colors, _ = rasterize_views(settings, [dict(means3D=params["means3D"], means2D=m2ds[i], opacities=params["opacities"],
                                                        shs=params["shs"], scales=params["scales"],
                                                        rotations=params["rotations"]) for i in range(V)], stacked=True)
            # one L1 over the [V,3,H,W] batch, as the reference does (train_utils.py:262-285); x V = the sum of the
            # per-view means that the camera-by-camera branch below forms
            return_loss = l1_loss(colors, targets_stacked) * float(V)
        else:
            outs = [render(i, m2ds[i]) for i in range(V)]
            return_loss = torch.stack([l1_loss(outs[i][0], targets[i]) for i in range(V)]).sum()
        loss = return_loss
        loss.backward()         # the batched node fans the views' K7/K8 out over the same per-view streams
        m2d_grads = [m.grad for m in m2ds]
        if world > 1:
            flat = torch.cat([params[k].grad.reshape(P, -1) for k in flat_names] + [sum(m2d_grads)], dim=1)
            dist.all_reduce(flat)
        return loss

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # steady state reached: park everything allocated so far (torch, the scene, the warm-up graphs) in the permanent
    # generation, so that the cyclic collector's periodic full passes do not walk ~10^5 long-lived objects inside the
    # timed region (measured: +1.1 ms/step on the per-view path).  Nothing is skipped: young garbage is still collected.
    import gc
    gc.collect()
    gc.freeze()
    # R (tile instances) per view for the algorithmic-byte count; constant across steps (same inputs)
    import diff_gaussian_rasterization as dgr
    sync()
    if not os.environ.get("CSPLAT_BENCH_NOEVENTS"):
        native.prof_enable(["K7_render_bwd"])
    native.prof_read("K7_render_bwd")
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    k7_ms, k7_n = native.prof_read("K7_render_bwd")
    native.prof_enable([])

    # per-kernel breakdown (separate, untimed pass) and R per view
    native.prof_enable(native.PROF_CLASSES[:8])
    for c in native.PROF_CLASSES[:8]:
        native.prof_read(c)
    step(); torch.cuda.synchronize()
    breakdown = {}
    for c in native.PROF_CLASSES[:8]:
        ms, n = native.prof_read(c)
        breakdown[c] = round(ms / max(n, 1) * 1e3, 2)  # us per launch-bracket
    native.prof_enable([])
    # K7 launched ALONE (views back to back on one stream; untimed, supplementary): the kernel-level reading of the
    # roofline next to the contract's in-step figure, where the views' K7 overlap and each launch lasts longer
    k7_alone_us = None
    if args.view_streams and V > 1:
        args.view_streams = False
        step(); torch.cuda.synchronize()
        native.prof_enable(["K7_render_bwd"]); native.prof_read("K7_render_bwd")
        step(); torch.cuda.synchronize()
        ms_a, n_a = native.prof_read("K7_render_bwd")
        native.prof_enable([])
        args.view_streams = True
        k7_alone_us = ms_a / max(n_a, 1) * 1e3
    with torch.no_grad():
        for i in range(V):
            ctx = type("C", (), {"save_for_backward": lambda s, *a: None, "mark_non_differentiable": lambda s, *a: None})()
            dgr._RasterizeGaussians.forward(ctx, params["means3D"], None, params["shs"], None, params["opacities"],
                                            params["scales"], params["rotations"], None, settings[i])
            R_per_view[i] = ctx.view_state.num_rendered

    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    mpix = world * V * W * H / 1e6
    value = mpix / (dt / args.steps)

    # roofline of the dominant kernel named by BASELINE.json north_star: compositing backward (K7).
    # algorithmic bytes per launch = 84*R + 24*X (instance re-read 44 B + 40 B partials per instance;
    # dL_dpix 16 B (incl. unused depth grad slot) + T 4 B + n_contrib 4 B per pixel), R = that view's tile instances.
    X = W * H
    alg_bytes = float(np.mean([84.0 * r + 24.0 * X for r in R_per_view]))
    k7_avg_s = (k7_ms / max(k7_n, 1)) * 1e-3
    achieved = alg_bytes / k7_avg_s / 1e9 if k7_avg_s > 0 else 0.0
    traffic = None
    import glob
    tpaths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_k7_pmc_traffic.json")))   # newest round's PMC passes
    if tpaths:
        try:
            traffic = json.load(open(tpaths[-1])).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    out = {
        "metric": "rasterizer fwd+bwd rendered Mpix/s (scene_1, 800x800)", "value": round(value, 3), "unit": "Mpix/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"scene_1 synthetic, P={P} Gaussians, {V} cams {W}x{H} per GPU, SH degree 3, "
                               "fwd (K1-K6) + L1 + bwd (K7-K8)" + (", + RCCL all-reduce of flat grads" if world > 1 else ""),
                   "tile_instances_per_view": R_per_view, "parallelism": f"view-parallel x{world}",
                   "streams_per_gpu": V if args.view_streams else 1},
        "roofline": {"bound": "hbm", "kernel": "k_render_bwd (K7 compositing backward)", "achieved": round(achieved, 3),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                     "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_us": round(k7_avg_s * 1e6, 2),
                     "launches_timed": int(k7_n),
                     "alone": None if not k7_alone_us else {
                         "avg_launch_us": round(k7_alone_us, 2), "achieved": round(alg_bytes / (k7_alone_us * 1e-6) / 1e9, 3),
                         "frac": round(alg_bytes / (k7_alone_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 6),
                         "what": "same kernel, views back to back on one stream (untimed extra pass)"},
                     "note": (f"the {V} views' K7 launches of a step run CONCURRENTLY on {V} streams: each launch lasts "
                              "longer than alone (see 'alone') while the step gets shorter; achieved/frac follow the contract "
                              "(bytes of ONE launch / its own duration)") if args.view_streams and V > 1 else None},
        "kernel_us": breakdown,
    }
    # the first half of BASELINE.json's metric ("train-step ms"): BASELINE configs[2], measured by bench_train.py (untimed
    # here, its own timed region; 1 GPU only).  Auxiliary field -- `value` stays the rasterizer fwd+bwd throughput.
    if rank == 0 and world == 1 and not args.no_train_step and (P, W, V) == (100_000, 800, 4):
        try:
            import bench_train
            from types import SimpleNamespace as _NS
            r = bench_train.run(_NS(steps=30, warmup=5, P=100_000, res=800, grid=100), dev)
            out["train_step"] = {"ms": r["value"], "unit": "ms", "rendered_Mpix_per_s": r["rendered_Mpix_per_s"], "steps": r["steps"],
                                 "psnr_first": r["psnr_first"], "psnr_last": r["psnr_last"], "workload": r["config"]["workload"]}
        except Exception as e:      # never let the auxiliary leg take the headline line down
            out["train_step"] = {"error": repr(e)[:200]}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(scene, cams, P, W, H)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
