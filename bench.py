#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on synthetic scene_1: rendered Mpix/s of the rasterizer hot path, forward +
backward, P = 100k Gaussians, 4 cameras 800x800 per GPU (BASELINE.json configs[1]).

A "step" = one pass of the hot path over one batch: for each of the rank's 4 views, GaussianRasterizer forward
(K1..K6), L1 loss against a fixed target image, backward (K7, K8); with N > 1 ranks (view-parallel, one process per
GPU) the step ends with ONE RCCL all-reduce of the flat per-Gaussian gradient buffer the `.grad` tensors are views of
(csplat.dist.FlatGrads).  All inputs are resident in HBM before the timed region.  Prints ONE JSON line on
rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

--mode scenes (BASELINE configs[4], scene-parallel): six seeded scene variants, scene s on rank s mod N, no collective
(replicas only); `value` = all scenes' rendered Mpix over the slowest rank's time, "scaling": "strong".
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
            "what": "C/OpenMP restatement of the upstream algorithm (oracle/raster_ref.c); the reference ships no CPU rasterizer "
                    "and no PyTorch-CPU path exists for this operator (SURVEY F2)",
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
    ap.add_argument("--no-sustained", action="store_true", help="skip the auxiliary 2000-replay sustained-clock pass (profiler runs)")
    ap.add_argument("--no-gnn", action="store_true", help="skip the auxiliary config-4 MeshNet rollout measurement (bench_gnn.py)")
    ap.add_argument("--no-view-streams", dest="view_streams", action="store_false",
                    help="run the views of a step back to back on one stream instead of one HIP stream per view")
    ap.add_argument("--eager", action="store_true", help="launch every step kernel by kernel instead of replaying recorded hipGraphs")
    ap.add_argument("--no-speculation", action="store_true",
                    help="skip the auxiliary 40-step pass with changing P / sizes (counter passes: every launch of a kernel then has the same size)")
    ap.add_argument("--exchange", choices=("auto", "oneshot", "direct", "sliced"), default="auto",
                    help="N > 1: how the step's gradients are summed over the ranks -- oneshot: one RCCL all-reduce of the flat buffer behind "
                         "the step; direct: the same message as all-to-all reduce-scatter + all-gather (full-mesh xGMI); sliced: K8 cut into "
                         "--slices Gaussian ranges, each range's rows leave while the next computes; auto (default): all are timed on this "
                         "node before the timed region and the fastest is used")
    ap.add_argument("--slices", type=int, default=4)
    ap.add_argument("--mode", choices=("views", "scenes"), default="views",
                    help="views: view-parallel weak scaling (default, the BASELINE metric); scenes: 6 scene variants dealt over the ranks")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: start the N ranks as FRESH children (this process has not touched the GPU: importing
        # torch does not initialise it) and pass their exit code on -- one rank per GPU, exactly what the driver's torchrun line does
        import subprocess
        port = 29500 + (os.getpid() % 2000)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N > 1 ranks as\n  python -m torch.distributed.run --nnodes=1 "
                         f"--nproc-per-node {args.gpus} --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus {args.gpus} ...")
    # (CSPLAT_BENCH_BACKEND=gloo lets the N>1 code path be exercised on a 1-GPU box: ranks then share the device)
    backend = os.environ.get("CSPLAT_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    import torch.distributed as dist
    # CSPLAT_FORCE_DIST=1 with one rank: the N > 1 code path (FlatGrads + one all-reduce per step + the `collective` leg) with the
    # collectives really issued -- RCCL exercised on a 1-GPU box (csplat/dist.py, tests/test_rccl_gpu.py)
    dist_on = world > 1 or os.environ.get("CSPLAT_FORCE_DIST", "") == "1"
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29500 + (os.getpid() % 2000)))
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from csplat import native, synthetic as syn
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

    P, W, H, V = args.P, args.res, args.res, args.views
    from csplat import dist as cd
    from csplat.train import l1_loss, step_stats
    from diff_gaussian_rasterization import rasterize_views
    import diff_gaussian_rasterization as dgr_
    T = lambda a, rg=False: torch.tensor(np.asarray(a, np.float32), device=dev, requires_grad=rg)  # noqa: E731

    class Workload:
        """one scene replica: Gaussians, V cameras, target images, and step() = fwd + L1 + bwd (+ the all-reduce)"""

        def __init__(self, seed, cam_phase, reduce_over_ranks):
            self.scene = syn.scene_1(P=P, W=W, H=H, n_cams=V, seed=seed)
            # view-parallel: rank r renders its own V cameras (azimuth offset), Gaussians replicated
            self.cams = [syn.make_camera(-180.0 + 360.0 * (k + cam_phase) / V, W, H) for k in range(V)]
            g = syn.gaussians_at(self.scene)
            self.names = ("means3D", "opacities", "shs", "scales", "rotations")
            self.params = {k: T(g[k], True) for k in self.names}
            bg = T(self.scene["bg"])
            self.settings = [GaussianRasterizationSettings(
                image_height=H, image_width=W, tanfovx=c["tanfovx"], tanfovy=c["tanfovy"], bg=bg, scale_modifier=1.0,
                viewmatrix=T(c["world_view_transform"]), projmatrix=T(c["full_proj_transform"]), sh_degree=3,
                campos=T(c["camera_center"]), prefiltered=False, debug=False) for c in self.cams]
            # target images = render of a perturbed copy (so the loss and its gradient are non-trivial)
            with torch.no_grad():
                keep = {k: v.detach().clone() for k, v in self.params.items()}
                gen = torch.Generator(device=dev).manual_seed(1234)
                self.params["shs"].add_(0.2 * torch.randn(self.params["shs"].shape, device=dev, generator=gen))
                self.params["means3D"].add_(0.003 * torch.randn(self.params["means3D"].shape, device=dev, generator=gen))
                self.targets = [self.render(i, torch.zeros(P, 3, device=dev))[0].clone() for i in range(V)]
                for k, v in keep.items():
                    self.params[k].copy_(v)
            self.targets_stacked = torch.stack(self.targets).contiguous()
            self.zeros = torch.zeros(V, P, 3, device=dev)
            self.one = torch.ones((), device=dev)
            # N > 1: every gradient of the step is a view into ONE flat buffer (+ 3P floats for the summed screen-space
            # gradient that densification consumes) and the step ends with one all-reduce of it -- no cat, no copy back
            self.fg = cd.FlatGrads([self.params[k] for k in self.names], extra=3 * P) if reduce_over_ranks else None

        def render(self, i, means2D):
            pr = self.params
            return GaussianRasterizer(self.settings[i])(means3D=pr["means3D"], means2D=means2D, opacities=pr["opacities"],
                                                        shs=pr["shs"], scales=pr["scales"], rotations=pr["rotations"])

        def grads(self):
            """the step's gradient tensors: the five parameters' and the per-view screen-space ones"""
            return {**{k: self.params[k].grad for k in self.names}, **{f"means2D[{i}]": m.grad for i, m in enumerate(self.m2ds)}}

        def step(self, timed_allreduce=False, skip_allreduce=False, defer=None):
            """defer: a diff_gaussian_rasterization.DeferredK8 -- the backward launches K7 only; finish_sliced() launches the K8 slices, each
            followed by the exchange of its gradient rows, then the statistics"""
            pr = self.params
            if self.fg is not None:
                self.fg.bind()
            else:
                for p_ in pr.values():
                    p_.grad = None
            # the screen-space leaves: V detached views of ONE zero buffer that nothing ever writes into (the rasterizer reads no value
            # of means2D, it returns its gradient) -- resident like every other input, no fill launch per step
            m2ds = self.m2ds = [self.zeros[i].detach().requires_grad_() for i in range(V)]
            # independent views run on separate HIP streams (diff_gaussian_rasterization.rasterize_views): the first phase of
            # all views goes out in four launches, then every view's binning / compositing kernels overlap on the chip
            if args.view_streams:   # all forwards first (train_step renders every camera, then calls backward once)
                colors, outs_ = rasterize_views(self.settings, [dict(means3D=pr["means3D"], means2D=m2ds[i], opacities=pr["opacities"],
                                                                     shs=pr["shs"], scales=pr["scales"], rotations=pr["rotations"])
                                                                for i in range(V)], stacked=True)
                self.radii = [o_[1] for o_ in outs_]
                # one L1 over the [V,3,H,W] batch, as the reference does (train_utils.py:262-285) = the mean of the per-view
                # means that the camera-by-camera branch below forms
                loss = l1_loss(colors, self.targets_stacked)
            else:
                outs = [self.render(i, m2ds[i]) for i in range(V)]
                self.radii = [o_[1] for o_ in outs]
                loss = torch.stack([l1_loss(outs[i][0], self.targets[i]) for i in range(V)]).mean()
            try:
                if defer is not None:
                    with dgr_.deferred_k8(defer):
                        loss.backward(gradient=self.one)
                else:
                    loss.backward(gradient=self.one)   # (a resident 1.0: autograd would launch a fill for the root gradient every step)
            finally:
                if self.fg is not None:
                    self.fg.unbind()
            if defer is not None:
                self.m2d_grads = [m.grad for m in m2ds]         # (filled by the K8 slices)
                return loss
            if self.fg is not None:
                # the screen-space sum densification consumes, written straight into the flat buffer's tail (one launch; K8 wrote every
                # parameter gradient into its slice already: no stock add / stack / sum launches in the step)
                step_stats([m.grad for m in m2ds], self.radii, self.zeros.shape[1], dev, out_vsg=self.fg.tail.view(-1, 3))
                if not skip_allreduce:
                    self.fg.all_reduce(timed=timed_allreduce)
            return loss

        def finish_sliced(self, holder, m2d_grads, radii, G, exchange=True):
            """the tail of a step whose backward was launched with `defer`: K8 slice by slice, the rows of slice g on their way while
            slice g + 1 computes; then the statistics into the buffer's tail and the rest of the exchange"""
            for g_ in range(G):
                holder.launch(g_, G)
                if exchange and self.fg is not None:
                    lo, hi = holder.rows(g_, G)
                    self.fg.start_ranges(self.fg.slice_ranges(P, lo, hi))
            if self.fg is not None:
                step_stats(m2d_grads, radii, self.zeros.shape[1], dev, out_vsg=self.fg.tail.view(-1, 3))
                if exchange:
                    self.fg.finish_sliced(P)

    class GraphedSteps:
        """The workload's step recorded into G hipGraphs and replayed round-robin: csplat.graphs.ReplayedSteps -- the SAME object
        tests/test_raster_gpu.py::test_config2_full_size_faith_replay_vs_oracle holds to the oracle (both phases of the forward launched
        with capacities taken from an eager step's counts, nothing read back; a device word per graph says whether the counts fitted --
        checked after the timed region, together with the replayed gradients themselves)."""

        def __init__(self, w_, G=4, sliced=0):
            """sliced = S > 0 (N > 1): the recordings end behind K7; the host launches the S K8 slices behind every replay, each followed
            by the exchange of its gradient rows (Workload.finish_sliced)"""
            from csplat.graphs import ReplayedSteps
            self.w, self.sliced, self.exchange = w_, int(sliced), True

            def fn():       # (N > 1: the all-reduce follows every replay, launched by the host)
                if self.sliced:
                    holder = dgr_.DeferredK8()
                    loss = w_.step(defer=holder)
                    return loss, w_.grads(), holder, w_.m2d_grads, w_.radii
                loss = w_.step(skip_allreduce=True)
                return loss, w_.grads()
            self.fn = fn
            if self.sliced:     # (the count pass runs fn() eagerly: its deferred K8 must be launched too, or the scratch records stay dirty)
                def counted():
                    out_ = fn()
                    w_.finish_sliced(out_[2], out_[3], out_[4], self.sliced, exchange=False)
                    return out_
                self.rs = ReplayedSteps(counted, dev, G=G)
                self.rs.fn = fn
            else:
                self.rs = ReplayedSteps(fn, dev, G=G)
            self.caps, self.counts = self.rs.caps, self.rs.counts

        def record(self):
            self.rs.record()
            self.graphs = self.rs.graphs

        def step(self):
            out_ = self.rs.step()
            if self.sliced:
                self.w.finish_sliced(out_[2], out_[3], out_[4], self.sliced, exchange=self.exchange)
            elif self.w.fg is not None and self.exchange:       # N > 1: ONE all-reduce of the flat gradient buffer the recorded kernels have just filled
                self.w.fg.all_reduce()

        def all_valid(self):
            return self.rs.all_valid()

    scene_mode = args.mode == "scenes"
    if scene_mode:      # BASELINE configs[4]: six seeded scene variants, scene s on rank s mod N, no data-path collective
        n_scenes = 6
        loads = [Workload(6666 + 17 * s_, 0.0, False) for s_ in range(n_scenes) if s_ % world == rank]
    else:
        n_scenes = world
        loads = [Workload(syn.SEED, rank / max(world, 1), dist_on)]
    wl = loads[0] if loads else None

    def step():
        out_ = None
        for w_ in loads:
            out_ = w_.step()
        return out_

    def sync():
        if dist_on:
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
    import diff_gaussian_rasterization as dgr
    sync()
    # the step as replayed hipGraphs (1 GPU, batched views; --eager / any failure: launch by launch as in rounds 1-3)
    graphed, launch_mode = None, "eager"
    want_graph = not scene_mode and args.view_streams and V >= 2 and not args.eager and wl is not None
    if want_graph:
        try:
            graphed = GraphedSteps(wl)              # (an eager step for the counts the capacities are taken from)
        except Exception as e:
            graphed, launch_mode = None, "eager (no counts: " + repr(e)[:120] + ")"
    if graphed is not None:
        try:
            graphed.record()
            for _ in range(2 * len(graphed.graphs)):
                graphed.step()
            torch.cuda.synchronize()
            if not graphed.all_valid():
                raise RuntimeError("the recorded capacities do not fit the workload's counts")
            launch_mode = f"hipGraph replay ({len(graphed.graphs)} recordings, round-robin)" + (" + one all-reduce per step" if dist_on else "")
        except Exception as e:
            graphed, launch_mode = None, "eager (recording failed: " + repr(e)[:120] + ")"
    if dist_on and want_graph:
        # all ranks replay or none does: a curve that mixes eager and replayed ranks measures the slowest kind
        flag = torch.tensor([1 if graphed is not None else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0 and graphed is not None:
            graphed, launch_mode = None, "eager (recording failed on another rank)"
    # ---- N > 1: HOW the step's gradients are summed over the ranks is measured on this node, not assumed (round 6).  Candidates: one RCCL
    # all-reduce of the flat buffer behind the step ("oneshot"); the same message as an all-to-all reduce-scatter + local sum + all-gather
    # ("direct": xGMI is a full mesh, an all-to-all drives all seven links of every GPU at once); K8 cut into S Gaussian ranges whose rows
    # leave while the next range computes ("sliced").  Each is timed over the same replayed steps (max over ranks), next to the steps
    # WITHOUT any exchange; the fastest is used for the timed region and the line says which, with every candidate's time.
    exchange_info = None
    if dist_on and graphed is not None and wl.fg is not None:
        def time_steps(g_, n=10):
            for _ in range(3):
                g_.step()
            sync(); t_x = time.perf_counter()
            for _ in range(n):
                g_.step()
            sync()
            t_ = torch.tensor([(time.perf_counter() - t_x) / n * 1e3], dtype=torch.float64, device=dev)
            dist.all_reduce(t_, op=dist.ReduceOp.MAX)
            return float(t_.item())
        cand = {}
        graphed.exchange = False
        cand["no_exchange"] = time_steps(graphed)
        graphed.exchange = True
        want = [args.exchange] if args.exchange != "auto" else ["oneshot", "direct", "sliced"]
        sliced_g = None
        for name in want:
            try:
                if name == "sliced":
                    sliced_g = GraphedSteps(wl, sliced=max(1, args.slices))
                    sliced_g.record()
                    wl.fg.algo = "rccl"
                    cand[name] = time_steps(sliced_g)
                    if not sliced_g.all_valid():
                        raise RuntimeError("capacities")
                else:
                    wl.fg.algo = "direct" if name == "direct" else "rccl"
                    cand[name] = time_steps(graphed)
                    if name == "direct" and wl.fg.algo != "direct":
                        cand[name] = None           # (the backend has no all-to-all on device tensors: fell back)
            except Exception as e:
                cand[name] = None
                cand[name + "_error"] = repr(e)[:120]
        ok_ = torch.tensor([0.0 if cand.get(n_) is None else 1.0 for n_ in want], dtype=torch.float64, device=dev)
        dist.all_reduce(ok_, op=dist.ReduceOp.MIN)          # a candidate counts only if it worked on every rank
        live = [n_ for n_, o_ in zip(want, ok_.tolist()) if o_ > 0]
        chosen = min(live, key=lambda n_: cand[n_]) if live else "oneshot"
        wl.fg.algo = "direct" if chosen == "direct" else "rccl"
        if chosen == "sliced":
            graphed = sliced_g
        exchange_info = {"chosen": chosen, "slices": max(1, args.slices) if chosen == "sliced" else 1,
                         "step_ms": {k_: (None if v_ is None else round(v_, 4)) for k_, v_ in cand.items() if not k_.endswith("_error")},
                         "errors": {k_: v_ for k_, v_ in cand.items() if k_.endswith("_error")} or None,
                         "exposed_allreduce_ms": None if cand.get(chosen) is None else round(cand[chosen] - cand["no_exchange"], 4)}
        launch_mode = f"hipGraph replay ({len(graphed.graphs)} recordings, round-robin) + exchange '{chosen}'"
    events_on = not os.environ.get("CSPLAT_BENCH_NOEVENTS")
    # ---- the step under a sustained load, FIRST (round 6: the driver's gpu_busy sampler saw 0 % twice in a 6.9 s run whose timed region
    # lasts 11 ms): > 1 s of back-to-back replays of the step that is timed next -- an extended warm-up by the contract's terms (untimed),
    # and the clocks the timed region then runs at are the sustained ones (DVFS / power management)
    sustained = None
    if graphed is not None and not dist_on and not args.no_sustained:
        n_sus = 2000
        torch.cuda.synchronize(); t_s = time.perf_counter()
        for _ in range(n_sus):
            graphed.step()
        torch.cuda.synchronize()
        sustained = {"steps": n_sus, "ms_per_step": round((time.perf_counter() - t_s) / n_sus * 1e3, 4),
                     "what": "back-to-back hipGraph replays of the timed step, straight BEFORE the timed region (untimed by the contract)"}
        graphed.rs.check()
    if graphed is None and events_on:
        native.prof_enable(["K7_render_bwd"])
    native.prof_read("K7_render_bwd")
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        graphed.step() if graphed is not None else step()
    sync()
    dt = time.perf_counter() - t0
    replay_check = None
    eager_ms_per_step = None
    if graphed is not None:
        if not graphed.all_valid():       # (cannot happen on a static scene; never report a step that did nothing)
            raise SystemExit("bench.py: a replayed step reported counts beyond its capacities")
        # ---- did the replays do the work?    The LAST replay's loss and gradients against one eager step of the same
        # workload: the forward is deterministic (loss bit-equal), the gradients differ by the order of K7's float atomics only (1e-5 of
        # each tensor's scale over 1500 repeats, tools/stress_step.py).  A replay that skipped work, ran on stale buffers or diverged from
        # the eager path ends the run instead of reporting a number.
        if wl.fg is None:
            r_loss, r_grads = graphed.rs.outs[graphed.rs.last()]
            r_loss = float(r_loss.detach())
            r_g = {k: v.detach().clone() for k, v in r_grads.items()}
            e_loss = float(step().detach())
            e_g = wl.grads()
            worst = max(float((r_g[k] - e_g[k]).abs().max()) / max(float(e_g[k].abs().max()), 1e-30) for k in e_g)
            replay_check = {"loss_replayed": r_loss, "loss_eager": e_loss, "grad_max_abs_diff_over_scale": round(worst, 9),
                            "bar": 1e-4, "what": "last timed replay vs one eager step: loss bit-equal, every gradient tensor (5 parameters + "
                                                 f"{V} screen-space) within the float-atomic noise of K7"}
            if (r_loss != e_loss or not (worst <= 1e-4)) and not os.environ.get("CSPLAT_BENCH_ELIMINATION_BUILD"):
                # (CSPLAT_BENCH_ELIMINATION_BUILD=1: tools/ab_libs.sh timing a build with work compiled OUT on purpose -- its line is scratch)
                raise SystemExit(f"bench.py: the replayed step differs from the eager step: {replay_check}")
        # K7's HIP-event bracket: a kernel launched by a graph node cannot be bracketed by timeable events on this ROCm (external
        # event-record nodes: hipEventElapsedTime refuses them), so the SAME K steps run once more launch by launch, straight behind the
        # timed replays, with the event pair around every K7 launch -- same kernel, same launch geometry, same inputs; the rocprofv3
        # trace of this command holds both populations under one kernel name.  Before it, the same K steps launched eagerly WITHOUT brackets are
        # timed: `eager_ms_per_step`.
        for _ in range(3):      # (stream capture emptied torch's allocator cache: the first eager steps behind it re-grow the pools)
            step()
        sync(); t_e = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync()
        eager_ms_per_step = (time.perf_counter() - t_e) / args.steps * 1e3
        if events_on:
            native.prof_enable(["K7_render_bwd"])
        native.prof_read("K7_render_bwd")
        for _ in range(args.steps):
            step()
        sync()
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
    per_camera_ms_per_step = None
    if args.view_streams and V > 1 and wl is not None:
        args.view_streams = False
        for _ in range(4):
            step()
        torch.cuda.synchronize()
        # the reference's own call pattern (scene_reconstruction/train_utils.py:259-292): GaussianRasterizer(...) once per camera, one
        # loss over the images, ONE backward -- timed over the same K steps, no event brackets inside (they cost host time per launch)
        sync(); t_c = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync()
        per_camera_ms_per_step = (time.perf_counter() - t_c) / args.steps * 1e3
        native.prof_enable(["K7_render_bwd"]); native.prof_read("K7_render_bwd")
        step(); step(); torch.cuda.synchronize()
        ms_a, n_a = native.prof_read("K7_render_bwd")
        native.prof_enable([])
        args.view_streams = True
        k7_alone_us = ms_a / max(n_a, 1) * 1e3
    # the collective on its own (untimed extra passes, bracketed by device synchronisation): bytes, ranks, wall time
    collective = None
    if dist_on and not scene_mode:
        ar = []
        for _ in range(5):
            wl.step(timed_allreduce=True)
            ar.append(wl.fg.last_allreduce_ms)
        # the same steps WITHOUT the exchange: what the collective adds to the step as launched (its exposed time).  In this workload
        # every gradient is finished by ONE kernel (K8) at the very end of backward, so nothing is left to hide the exchange behind;
        # the train step (csplat.train.train_step, view_parallel) sends the Gaussian gradients' slice from a backward hook, under the
        # simulator's and the regularisers' backward (csplat/dist.py: early bucket)
        sync(); t_n = time.perf_counter()
        for _ in range(5):
            wl.step(skip_allreduce=True)
        sync(); ms_noar = (time.perf_counter() - t_n) / 5 * 1e3
        sync(); t_n = time.perf_counter()
        for _ in range(5):
            wl.step()
        sync(); ms_ar = (time.perf_counter() - t_n) / 5 * 1e3
        collective = {"backend": "nccl (RCCL)" if backend == "nccl" else backend, "ranks": dist.get_world_size(),
                      "bytes": int(wl.fg.flat.numel() * 4), "allreduce_ms": round(float(np.median(ar)), 4),
                      # the exchange the TIMED region used, chosen by measurement before it (all candidates' step times, max over ranks)
                      "exchange": exchange_info, "slices": (exchange_info or {}).get("slices", 1),
                      "exposed_allreduce_ms": (exchange_info or {}).get("exposed_allreduce_ms", round(ms_ar - ms_noar, 4)),
                      "eager_exposed_allreduce_ms": round(ms_ar - ms_noar, 4), "step_ms_without_allreduce": round(ms_noar, 4),
                      "what": "one all-reduce(sum) per step over the flat gradient buffer (62 floats per Gaussian + 3 for the "
                              "screen-space gradient); timed alone, after the step's kernels have drained"}
    R_per_view = [0] * V
    pairs_per_view = []
    if wl is not None:
        with torch.no_grad():
            for i in range(V):
                ctx = type("C", (), {"save_for_backward": lambda s, *a: None, "mark_non_differentiable": lambda s, *a: None})()
                dgr._RasterizeGaussians.forward(ctx, wl.params["means3D"], None, wl.params["shs"], None, wl.params["opacities"],
                                                wl.params["scales"], wl.params["rotations"], None, wl.settings[i])
                R_per_view[i] = ctx.view_state.num_rendered
                try:        # the (list entry, 4x4 block) pairs K6 marked as blended = the float-atomic requests K7 issues for this view
                    import ctypes as _C
                    o11 = (_C.c_size_t * 11)()
                    native.lib.csplat_binning_fields(int(R_per_view[i]), W, H, o11)
                    raw = ctx.view_state.chunks[1]
                    tiles_ = ((W + 15) // 16) * ((H + 15) // 16)
                    seg_off_ = raw[o11[2]:o11[2] + 4 * (tiles_ + 1)].view(torch.int32).cpu().numpy()
                    nslots = int(seg_off_[tiles_])
                    blk_hi_ = raw[o11[2] + 4 * (tiles_ + 1):o11[2] + 4 * (tiles_ + 1) + tiles_ * 64].view(torch.int32).cpu().numpy().reshape(tiles_, 16)
                    slot_tile_ = raw[o11[3]:o11[3] + 4 * nslots].view(torch.int32).cpu().numpy()
                    bb = raw[o11[9]:o11[9] + nslots * 16 * 32].cpu().numpy().reshape(nslots, 16, 32)
                    # (K7 visits the (segment, block) words whose block still blends at or behind the segment -- blk_hi > segment start;
                    #  the others were never written)
                    seg_lo_ = (np.arange(nslots) - seg_off_[slot_tile_]) * 256
                    live_ = blk_hi_[slot_tile_] > seg_lo_[:, None]
                    pairs_per_view.append(int(np.unpackbits(bb, axis=2).sum(axis=2)[live_].sum()))
                except Exception:
                    pairs_per_view.append(None)

    # forward-only (evaluation) rate, auxiliary: what the reference's render.py prints as FPS (render.py:195,301: render() under no_grad,
    # camera by camera) -- here the V cameras of the batch per call, K1-K6 only
    eval_fwd = None
    if wl is not None and args.view_streams and world == 1:
        with torch.no_grad():
            kws = [dict(means3D=wl.params["means3D"], means2D=None, opacities=wl.params["opacities"], shs=wl.params["shs"],
                        scales=wl.params["scales"], rotations=wl.params["rotations"]) for _ in range(V)]
            for _ in range(3):
                rasterize_views(wl.settings, kws, stacked=True)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(20):
                rasterize_views(wl.settings, kws, stacked=True)
            torch.cuda.synchronize()
            dt_e = (time.perf_counter() - t1) / 20
        eval_fwd = {"ms_per_call": round(dt_e * 1e3, 4), "views_per_call": V, "fps": round(V / dt_e, 1),
                    "Mpix_per_s": round(V * W * H / 1e6 / dt_e, 1), "what": "rasterizer forward only under no_grad (K1-K6), all views of the batch per call"}

    # ---- what the speculative launch costs when training is NOT static.  In the timed region above every step's
    # counts equal the previous step's: the second forward phase is launched on the previous call's capacities and never misses.  Real
    # training changes P at every densification (gaussian_mesh.py:336-431 via train_utils.py:295-304), interleaves evaluation renders of
    # another size (train.py:286-333) and moves R a little every step.  40 steps: P changes every 10th step (+5 % cloned, then -5 %
    # pruned, alternating) -> no history for the new shape: the counts are WAITED for; a two-view 400x400 render under no_grad before
    # every 10th-plus-5 step (its own shape: own history); and at step 23 every Gaussian grows by 30 % for one step -> R up ~1.6x: the
    # capacities do not fit, the phase is repeated with exact sizes (a MISS).  Per step: which of hit / wait / miss the training-sized
    # forward took (diff_gaussian_rasterization.SPEC_STATS) and its wall time, device drained before and after.
    speculation = None
    if wl is not None and args.view_streams and world == 1 and not scene_mode and V >= 2 and not args.no_speculation:
        try:
            base = {k: v.detach().clone() for k, v in wl.params.items()}
            half = [GaussianRasterizationSettings(image_height=H // 2, image_width=W // 2, tanfovx=st_.tanfovx, tanfovy=st_.tanfovy, bg=st_.bg,
                                                  scale_modifier=1.0, viewmatrix=st_.viewmatrix, projmatrix=st_.projmatrix, sh_degree=3,
                                                  campos=st_.campos, prefiltered=False, debug=False) for st_ in wl.settings[:2]]

            def set_P(n):
                idx = torch.arange(n, device=dev) % P
                wl.params = {k: base[k][idx].clone().requires_grad_(True) for k in wl.names}
                wl.zeros = torch.zeros(V, n, 3, device=dev)
            classes = {"hit": [], "wait": [], "miss": []}
            cur_P = P
            for it in range(40):
                if it % 10 == 0 and it > 0:
                    cur_P = P + P // 20 if (it // 10) % 2 == 1 else P - P // 20
                    set_P(cur_P)
                if it % 10 == 5:
                    with torch.no_grad():
                        pr = wl.params
                        rasterize_views(half, [dict(means3D=pr["means3D"], means2D=None, opacities=pr["opacities"], shs=pr["shs"],
                                                    scales=pr["scales"], rotations=pr["rotations"]) for _ in half], stacked=True)
                if it in (23, 24):
                    with torch.no_grad():
                        wl.params["scales"].mul_(1.3 if it == 23 else 1.0 / 1.3)
                before = dict(dgr.SPEC_STATS)
                torch.cuda.synchronize(); t_s = time.perf_counter()
                wl.step()
                torch.cuda.synchronize(); t_e = time.perf_counter()
                kind = next((k for k in ("miss", "wait", "hit") if dgr.SPEC_STATS[k] > before[k]), "hit")
                classes[kind].append((t_e - t_s) * 1e3)
            set_P(P)
            wl.params = {k: base[k].clone().requires_grad_(True) for k in wl.names}
            mean = lambda a: round(float(np.mean(a)), 4) if a else None  # noqa: E731
            speculation = {"steps": 40, "hits": len(classes["hit"]), "waits": len(classes["wait"]), "misses": len(classes["miss"]),
                           "ms_per_hit_step": mean(classes["hit"]), "ms_per_waited_step": mean(classes["wait"]),
                           "ms_per_missed_step": mean(classes["miss"]),
                           "what": "40 steps, device drained around each: P +-5 % every 10th step (new shape: counts waited for), a 2-view "
                                   "400x400 no_grad render before steps 5, 15, 25, 35, every scale x1.3 at step 23 (R up ~1.6x: the second "
                                   "phase is repeated with exact sizes)"}
        except Exception as e:      # never let the auxiliary leg take the headline line down
            speculation = {"error": repr(e)[:200]}

    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if dist_on:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    mpix = n_scenes * V * W * H / 1e6
    value = mpix / (dt / args.steps)

    # roofline of the dominant kernel named by BASELINE.json north_star: compositing backward (K7).
    # algorithmic bytes per launch = 84*R + 24*X (instance re-read 44 B + 40 B partials per instance;
    # dL_dpix 16 B (incl. unused depth grad slot) + T 4 B + n_contrib 4 B per pixel), R = that view's tile instances.
    X = W * H
    # (with the views batched into one K7 launch per step the launch carries all views' bytes)
    k7_launches_per_step = max(1, int(round(k7_n / max(args.steps * max(len(loads), 1), 1))))
    views_per_launch = max(1, V // k7_launches_per_step)
    alg_bytes = float(np.mean([84.0 * r + 24.0 * X for r in R_per_view])) * views_per_launch
    k7_avg_s = (k7_ms / max(k7_n, 1)) * 1e-3
    achieved = alg_bytes / k7_avg_s / 1e9 if k7_avg_s > 0 else 0.0
    # committed counter passes of the same workload on the DEFAULT command (tools/collect_profiles.sh, tools/collect_issue_counters.sh):
    # HBM traffic and VALU instruction count of the K7 launch this run timed.  They are measurements of an earlier run of the same
    # kernel: the profile records the SHA-1 of csplat_raster.hip it was taken with, and the values are withheld (null, with the reason)
    # when the source has changed since.
    import glob
    import hashlib
    K7_NAMES = ("k_composite_bwd_rows_views", "k_composite_bwd_views") if (args.view_streams and V > 1) else ("k_composite_bwd_rows", "k_composite_bwd")
    src_sha = hashlib.sha1(open(os.path.join(ROOT, "cloth-splatting_amd", "csrc", "csplat_raster.hip"), "rb").read()).hexdigest()
    traffic = valu_insts = None
    counters_from = {}

    def newest(pattern):
        paths = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
        return paths[-1] if paths else None
    tp = newest("r*_k7_pmc_traffic.json")
    if tp:
        try:
            doc = json.load(open(tp))
            fresh = doc.get("raster_src_sha1") == src_sha
            counters_from["traffic"] = {"file": os.path.basename(tp), "same_kernel_source": fresh}
            table = doc.get("all_kernels", {}) if (args.view_streams and V > 1) else (doc.get("one_view_per_launch") or {})
            hit = next((v for k, v in table.items() if any(k.startswith(n) and "<true>" not in k for n in K7_NAMES[:1])), None) or \
                next((v for k, v in table.items() if any(k.startswith(n) for n in K7_NAMES)), None)
            if fresh and hit:
                traffic = int(hit["hbm_bytes_per_launch"])
        except Exception as e:
            counters_from["traffic"] = {"error": repr(e)[:120]}
    ip = newest("r*_k67_issue.json")
    if ip:
        try:
            doc = json.load(open(ip))
            fresh = doc.get("raster_src_sha1") == src_sha
            counters_from["issue"] = {"file": os.path.basename(ip), "same_kernel_source": fresh}
            table = doc.get("kernels", {}) if (args.view_streams and V > 1) else doc.get("kernels_one_view_per_launch", {})
            hit = next((table[n] for n in K7_NAMES if n in table), None)
            if fresh and hit:
                valu_insts = hit["SQ_INSTS_VALU"]
        except Exception as e:
            counters_from["issue"] = {"error": repr(e)[:120]}
    # what a wave64 VALU instruction costs a SIMD's vector pipe on this part (tools/valu_rate.hip, profiles/r03_valu_rate.txt, >= 2 waves
    # per SIMD): 2.3 cycles for plain v_fma / v_mul / v_add / v_mov, 4.2 for every DPP form, v_cmp, v_min / v_max, v_cndmask_e64, shifts and
    # conversions, 8.2 for v_exp / v_rcp / v_permlane*_swap.  K7's loop (round 4: 192 VALU per PAIR of groups of four survivors -- 118 / 62
    # / 12 of the three classes, from the ISA; round 3: 156 per group, 82 / 66 / 8 = 3.4) averages 3.3 cycles per instruction.
    SIMDS, CLOCK_GHZ, K7_CYC_PER_VALU = 1024, 2.4, 3.3
    issue = lambda us: None if not (valu_insts and us) else round(valu_insts * K7_CYC_PER_VALU / (SIMDS * CLOCK_GHZ * 1e3 * us), 4)  # noqa: E731
    out = {
        "metric": "rasterizer fwd+bwd rendered Mpix/s (scene_1, 800x800)", "value": round(value, 3), "unit": "Mpix/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "strong" if scene_mode else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": (f"6 seeded scene_1 variants dealt over {world} rank(s), " if scene_mode else "scene_1 synthetic, ") +
                               f"P={P} Gaussians, {V} cams {W}x{H} per " + ("scene" if scene_mode else "GPU") + ", SH degree 3, "
                               "fwd (K1-K6) + L1 + bwd (K7-K8)" +
                               (", + ONE RCCL all-reduce of the flat gradient buffer" if dist_on and not scene_mode else ""),
                   "tile_instances_per_view": R_per_view,
                   "parallelism": (f"scene-parallel x{world} (replicas only)" if scene_mode else f"view-parallel x{world}"),
                   "streams_per_gpu": V if args.view_streams else 1, "launch": launch_mode},
        "roofline": {"bound": "hbm", "kernel": "k_composite_bwd_rows_views (K7 compositing backward, all views of the step in one launch)", "achieved": round(achieved, 3),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                     # PMC bytes (2 x FETCH_SIZE + WRITE_SIZE) of THIS launch (all views of the step), from the committed counter pass of
                     # the default command; null when that pass was taken with a different csplat_raster.hip
                     "traffic": traffic, "counters_from": counters_from,
                     "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_us": round(k7_avg_s * 1e6, 2),
                     "launches_timed": int(k7_n),
                     "views_per_launch": views_per_launch,
                     # what BINDS K7 (round 6, profiles/r06_k7_elimination.txt): its float atomics execute at the memory side at one
                     # chip-wide rate, ~1.3 TB/s priced per 64-byte request (MI355X_MICROARCH.md, Global float atomics); K7 issues ONE
                     # request per (list entry, 4x4 block) pair that blended (counted here from K6's bbits words of this workload)
                     "atomics": None if (not pairs_per_view or any(p_ is None for p_ in pairs_per_view) or k7_avg_s <= 0) else {
                         "requests_per_launch": int(sum(pairs_per_view) * views_per_launch / max(len(pairs_per_view), 1)),
                         "bytes_per_request_priced": 64, "peak_GBps": 1300.0,
                         "achieved_GBps": round(sum(pairs_per_view) * views_per_launch / max(len(pairs_per_view), 1) * 64 / k7_avg_s / 1e9, 1),
                         "frac": round(sum(pairs_per_view) * views_per_launch / max(len(pairs_per_view), 1) * 64 / k7_avg_s / 1e9 / 1300.0, 4),
                         "what": "the resource that binds K7: memory-side float-atomic requests against the measured chip-wide rate; the "
                                 "kernel takes 161 us with the atomics compiled out (elimination build), 211 with them"},
                     "issue_frac": issue(k7_avg_s * 1e6),
                     "issue_note": "VALU wave-instructions of this launch (SQ_INSTS_VALU of the committed counter pass) x 3.3 cycles (the "
                                   "instruction mix of K7's loop priced with tools/valu_rate.hip: 2.3 plain / 4.2 DPP, compare, select, "
                                   "min-max / 8.2 exp, rcp, permlane swap) / (1024 SIMDs x 2.4 GHz x this launch's duration): the share of "
                                   "the vector pipes' cycles the kernel's arithmetic occupies -- the bound that binds it (composite "
                                   "arithmetic, no contraction), not HBM",
                     "alone": None if not k7_alone_us else {
                         "avg_launch_us": round(k7_alone_us, 2),
                         "achieved": round(alg_bytes / views_per_launch / (k7_alone_us * 1e-6) / 1e9, 3),
                         "frac": round(alg_bytes / views_per_launch / (k7_alone_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 6),
                         "issue_frac": None,
                         "what": "same kernel, views back to back on one stream (untimed extra pass)"},
                     "note": (f"the K7 work of the step's {V} views is ONE launch (blockIdx.y = view): bytes and instructions of "
                              f"{views_per_launch} view(s) per launch over that launch's duration; 'alone' = one view per launch, views "
                              "back to back") if args.view_streams and V > 1 else None},
        # the same workload launched three ways: `ms_per_step` = hipGraph replays of the batched step; eager = the batched
        # step (rasterize_views) launch by launch; per-camera = the reference's call pattern, one GaussianRasterizer call per camera
        "eager_ms_per_step": None if eager_ms_per_step is None else round(eager_ms_per_step, 4),
        "per_camera_ms_per_step": None if per_camera_ms_per_step is None else round(per_camera_ms_per_step, 4),
        "replay_check": replay_check,
        "sustained": sustained,
        "kernel_us": breakdown,
        "eval_forward": eval_fwd,
        "speculation": speculation,
        "collective": collective,
    }
    # the first half of BASELINE.json's metric ("train-step ms"): BASELINE configs[2], measured by bench_train.py (untimed
    # here, its own timed region; 1 GPU only).  Auxiliary field -- `value` stays the rasterizer fwd+bwd throughput.
    if rank == 0 and world == 1 and not scene_mode and not args.no_train_step and (P, W, V) == (100_000, 800, 4):
        try:
            import bench_train
            from types import SimpleNamespace as _NS
            r = bench_train.run(_NS(steps=100, warmup=10, P=100_000, res=800, grid=100), dev)
            # ms = mean over the timed steps (the contract's figure); the step is bound by the HOST (Python + ~33 launches), so the
            # mean moves with whatever else runs on the box's cores: median / fastest decile of the same steps beside it
            out["train_step"] = {"ms": r["value"], "median_ms": r["median_ms"], "p10_ms": r["p10_ms"],
                                 "eager_ms": r["eager_ms"], "captured": r["captured"], "captured_stats": r["captured_stats"], "unit": "ms",
                                 "rendered_Mpix_per_s": r["rendered_Mpix_per_s"], "steps": r["steps"],
                                 "psnr_first": r["psnr_first"], "psnr_last": r["psnr_last"], "workload": r["config"]["workload"],
                                 # the tests that hold the CAPTURED path (the figure in `ms`) to the eager step (`eager_ms`): bit for bit in
                                 # the reproducible K7 mode -- every parameter, both Adam moments, the step counts and the statistics
                                 "verified_by": ["tests/test_train_gpu.py::test_captured_train_step_equals_the_eager_step",
                                                 "tests/test_train_gpu.py::test_captured_train_step_survives_a_miss",
                                                 "tests/test_train_gpu.py::test_captured_train_step_across_an_eager_iteration",
                                                 "tests/test_train_gpu.py::test_captured_train_step_alternating_step_shapes",
                                                 "tests/test_train_gpu.py::test_captured_train_step_rerecords_after_a_scratch_eviction",
                                                 "tests/test_train_gpu.py::test_captured_train_step_default_mode_within_calibrated_atomic_noise"]}
        except Exception as e:      # never let the auxiliary leg take the headline line down
            out["train_step"] = {"error": repr(e)[:200]}
    # BASELINE configs[3] (MeshNet rollout, N = 10k, E = 300k, L = 128, M = 15): bench_gnn.py's rollout leg, auxiliary like `train_step`
    if rank == 0 and world == 1 and not scene_mode and not args.no_gnn and (P, W, V) == (100_000, 800, 4):
        try:
            import bench_gnn
            from types import SimpleNamespace as _NS
            torch.cuda.empty_cache()
            r = bench_gnn.run(_NS(steps=20, warmup=3, N=10_000, deg=30), dev, train=False)
            out["gnn"] = {"rollout_ms_per_step": r["rollout_loop_ms_per_step"], "rollout_eager_ms_per_step": r["rollout_loop_eager_ms_per_step"],
                          "rollout_recording": r["rollout_recording"],
                          # the bound that binds the fused layer kernels: fp16 piece-products on the matrix pipe against the 2.5 PFLOP/s
                          # dense peak; HBM from the committed PMC traffic of the layer launches (the rounds 1-5 `frac`
                          # divided the UNFUSED formulation's bytes by the fused kernels' time and was no utilisation)
                          "mfma_frac": r["mfma_frac"], "hbm_GBps": r["hbm_GBps"], "hbm_frac": r["hbm_frac"],
                          "predict_velocity_ms": r["rollout_ms"], "kernel_launches_per_step": r["gnn_kernels"]["launches_per_step"],
                          "unfused_algorithmic_bytes_per_step": r["gnn_kernels"]["algorithmic_bytes_per_step"], "peak_GBps": HBM_PEAK_GBS,
                          "workload": r["config"]["workload"]}
        except Exception as e:
            out["gnn"] = {"error": repr(e)[:200]}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(wl.scene, wl.cams, P, W, H)
    elif rank == 0:
        out["cpu_baseline"] = None
    # (RCCL writes a line of its own -- "Librccl path : ..." -- to stdout when the process group goes down: the JSON line comes AFTER
    #  it, so that it is the last line of the output whatever the library prints)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        try:        # (the library prints through C stdio, which is block-buffered on a pipe: push its buffer out first)
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
