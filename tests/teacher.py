"""Teacher-forced gradient parity of the training step (test infrastructure; may import oracle/).

Reference step being matched: /root/reference/scene_reconstruction/train_utils.py:240-321 (render the step's cameras, stack, L1 +
lambda (1 - SSIM) + cloth regularisers, ONE backward, statistics for densification, two Adam steps).

`capture(pc, sim)` arms the REAL csplat.train.train_step on the HIP path: the next step leaves (a) the parameters it started from,
(b) every parameter gradient it handed the optimizers, (c) the image batch it rendered and (d) dL/dimage as autograd delivered it
to the rasterizer's backward.  `oracle_step(...)` then evaluates the same step from (a) on the CPU: simulator, mesh -> Gaussian
transform, image loss and regularisers as fp64 torch, the rasterizer = oracle/raster_ref.c (fp64 or fp32 build) with its
analytic backward.  The step is held to the oracle NODE BY NODE, every node on the HIP step's own inputs (teacher forcing), so that
no comparison inherits the discontinuities of another node:

  * rasterizer (raster_stage): the tensors the HIP step handed its rasterizer (captured: means3D / rotations per camera, opacity,
    scales, SH) go through the oracle, forward and -- driven with the HIP step's own dL/dimage -- backward: image, radii and the
    gradient of every rasterizer input, with threshold ties counted and required to show in the fp32 build of the oracle too
    (identical inputs: the criterion of tests/test_raster_gpu.py);
  * loss node: dL/dimage of the HIP step against autograd over the fp64 torch formulation of the image loss AT THE HIP IMAGE;
  * everything in front of the rasterizer (pre_stage): simulator, cloth regularisers, mesh -> Gaussian transform, activations as fp64
    torch from the parameters, differentiated with the HIP step's own gradients of the rasterizer inputs: EVERY parameter gradient the
    step handed its optimizers.  These nodes are smooth: no ties, a plain bar;
  * chain (compare_chain) and end to end: the oracle replay from the PARAMETERS with the HIP dL/dimage, and with its own L1 signs --
    reported with counted deviations: there the fp64 mesh transform hands the oracle inputs that differ from the HIP path's by fp32
    rounding, which moves threshold decisions on its own (at 800^2 also the image, by ~1e-4), and the L1 term's sign(render - gt)
    flips wherever two renders that agree to 1e-6 straddle the target.
"""
import contextlib

import numpy as np
import torch

import util  # noqa: F401
from oracle import raster_oracle as ro

GAUSS_NAMES = ["face_bary", "face_offset", "f_dc", "f_rest", "opacity", "scaling", "rotation"]


class Captured:
    def __init__(self):
        self.params = self.grads = self.image = self.dimage = self.names = None
        self.psnr = self.loss = self.stats = None
        self.rin = None            # what the step handed the rasterizer: dict(means3D=[T x [P,3]], rot=[T x [P,4]], op, sc, sh)
        self.rgrad = {}            # ... and the gradients that came back for them (same keys; per-camera ones as ("means3D", i))


@contextlib.contextmanager
def capture(pc, sim):
    """arms the next train_step(s) inside the block; yields the record the LAST of them filled"""
    from csplat import train as tr
    cap = Captured()
    cap.names = GAUSS_NAMES + [n for n, _ in sim.named_parameters()]
    opt = pc.optimizer
    real_step = getattr(opt, "step_now", opt.step)
    real_loss = tr.FusedImageLoss

    def step_wrapper(*a, **k):
        # the Gaussian optimizer steps first (train_utils.py:310-313): backward is complete, nothing has been updated yet
        ps = list(pc.parameters()) + list(sim.parameters())
        cap.params = [p.detach().clone() for p in ps]
        cap.grads = [None if p.grad is None else p.grad.detach().clone() for p in ps]
        return real_step(*a, **k)

    class LossShim:
        @staticmethod
        def apply(image, *rest):
            cap.image = image.detach().clone()
            if image.requires_grad:
                image.register_hook(lambda g: setattr(cap, "dimage", g.detach().clone()))
            return real_loss.apply(image, *rest)

    real_inputs = pc.step_inputs

    def inputs_wrapper(deforms):
        out = real_inputs(deforms)
        if out is None:
            return out
        m3, rq, op, sc, sh = out
        cap.rin = dict(means3D=[t.detach().clone() for t in m3], rot=[t.detach().clone() for t in rq], op=op.detach().clone(),
                       sc=sc.detach().clone(), sh=sh.detach().clone())
        cap.rgrad = {}

        def keep(key):
            return lambda g: cap.rgrad.__setitem__(key, g.detach().clone())
        for i, t in enumerate(m3):
            if t.requires_grad:
                t.register_hook(keep(("means3D", i)))
        for i, t in enumerate(rq):
            if t.requires_grad:
                t.register_hook(keep(("rot", i)))
        for key, t in (("op", op), ("sc", sc), ("sh", sh)):
            if t.requires_grad:
                t.register_hook(keep(key))
        return out

    name = "step_now" if hasattr(opt, "step_now") else "step"
    setattr(opt, name, step_wrapper)
    pc.step_inputs = inputs_wrapper
    tr.FusedImageLoss = LossShim
    try:
        yield cap
    finally:
        delattr(opt, name) if name in opt.__dict__ else None
        pc.__dict__.pop("step_inputs", None)
        tr.FusedImageLoss = real_loss


def _oracle_raster(dtype, side):
    """autograd.Function over the C oracle in the given build; `side` collects radii and the NDC gradient of every call"""

    class OracleRaster(torch.autograd.Function):
        @staticmethod
        def forward(ctx, means3D, opacity, shs, scales, rots, cam, bg_np, sh_degree):
            n = lambda t: t.detach().numpy()  # noqa: E731
            H, W = int(cam.image_height), int(cam.image_width)
            o = ro.forward(n(means3D), n(opacity), n(cam.world_view_transform), n(cam.full_proj_transform), n(cam.camera_center),
                           np.tan(cam.FoVx * 0.5), np.tan(cam.FoVy * 0.5), W, H, bg_np, shs=n(shs), sh_degree=sh_degree,
                           scales=n(scales), rotations=n(rots), dtype=dtype)
            ctx.o = o
            ctx.slot = len(side["radii"])
            side["radii"].append(o.radii.copy())
            side["mean2D"].append(None)
            return torch.from_numpy(np.asarray(o.color, np.float64))

        @staticmethod
        def backward(ctx, g_color):
            g = ro.backward(ctx.o, np.ascontiguousarray(g_color.numpy()))
            side["mean2D"][ctx.slot] = np.asarray(g.mean2D, np.float64)
            t = lambda a: torch.from_numpy(np.asarray(a, np.float64))  # noqa: E731
            return t(g.mean3D), t(g.opacity).reshape(-1, 1), t(g.sh), t(g.scale), t(g.rot), None, None, None
    return OracleRaster


def oracle_step(build_cpu, cams_cpu, params, dimage=None, image_for_loss=None, mask=None, oracle_dtype=np.float64, opt=None):
    """The step on the CPU from the parameter snapshot `params` (list of tensors in the order parameters(pc) + parameters(sim)).

    dimage given  -> CHAIN mode: backward of (image, regularisers) with the image's gradient := dimage (the HIP step's own).
    dimage None   -> END-TO-END: loss = image_losses + regularization differentiated as a whole.
    image_for_loss -> additionally autograd of the image loss alone at that image (the loss node's reference gradient).
    Returns a dict: grads (list), image [B,3,H,W], loss, psnr, radii (max over cameras), vsg (sum of NDC gradients), dimage_ref."""
    from csplat import train as tr
    opt = opt or tr.DEFAULT_OPT
    pc, sim = build_cpu()
    pc.fused = False
    ps = list(pc.parameters()) + list(sim.parameters())
    with torch.no_grad():
        for a, b in zip(ps, params):
            a.copy_(b.detach().cpu().double())
    for p in ps:
        p.grad = None
    side = dict(radii=[], mean2D=[])
    F = _oracle_raster(oracle_dtype, side)
    bg_np = np.ones(3)
    V = pc.mesh.pos.shape[0]
    imgs, verts = [], []
    for c in cams_cpu:
        v = sim(time_vector=torch.tensor(c.time, dtype=pc.mesh.pos.dtype).repeat(V, 1))
        color = F.apply(pc.get_xyz(v), pc.get_opacity, pc.get_features, pc.get_scaling, pc.get_rotation(v), c, bg_np, pc.active_sh_degree)
        imgs.append(color.unsqueeze(0)); verts.append(v[None])
    image = torch.cat(imgs, 0)
    gt = torch.stack([c.original_image for c in cams_cpu]).double()
    reg = tr.regularization(torch.cat(verts, 0), pc, opt)
    img_loss = tr.image_losses(image, gt, opt, mask)
    loss = img_loss + reg
    if dimage is not None:
        torch.autograd.backward([image, reg], [dimage.detach().cpu().double().reshape(image.shape), torch.ones((), dtype=reg.dtype)])
    else:
        loss.backward()
    out = dict(grads=[None if p.grad is None else p.grad.detach().clone() for p in ps], image=image.detach(), loss=float(loss),
               psnr=float(tr.psnr(image.detach(), gt).mean()), radii=np.max(np.stack(side["radii"]), 0),
               vsg=sum(m for m in side["mean2D"] if m is not None))
    if image_for_loss is not None:
        x = image_for_loss.detach().cpu().double().reshape(image.shape).requires_grad_()
        tr.image_losses(x, gt, opt, mask).backward()
        out["dimage_ref"] = x.grad
        out["image_loss_at"] = float(tr.image_losses(x.detach(), gt, opt, mask))
        # the same formulation in fp32 torch on the CPU -- the REFERENCE's own arithmetic (utils/loss_utils.py runs in fp32): how far fp32
        # moves this gradient (the SSIM variances are differences of nearly equal window means once the render matches the target)
        x32 = image_for_loss.detach().cpu().float().reshape(image.shape).requires_grad_()
        tr.image_losses(x32, gt.float(), opt, None if mask is None else mask.float()).backward()
        out["dimage_ref32"] = x32.grad.double()
    return out


def rows_err(got, ref, scale=None):
    """per-row max |got - ref| / max |ref| (rows = leading dimension)"""
    a = np.asarray(got, np.float64)
    b = np.asarray(ref, np.float64)
    a, b = a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1)
    return np.abs(a - b).max(1) / ((np.abs(b).max() if scale is None else scale) + 1e-30)


def loss_node_err(dimage, o64):
    """(error of the HIP dL/dimage, error of the fp32 torch formulation), both against autograd over the fp64 formulation at the same
    image and relative to its largest entry.  The bar for the first is max(1e-4, 3 x the second): the reference's own fp32 arithmetic
    is the yardstick where fp32 cannot deliver 1e-4 (near convergence the SSIM variances cancel to a few digits)."""
    ref = o64["dimage_ref"]
    d = dimage.detach().cpu().double().reshape(ref.shape)
    scale = float(ref.abs().max()) + 1e-30
    return float((d - ref).abs().max()) / scale, float((o64["dimage_ref32"] - ref).abs().max()) / scale


def compare_chain(cap, o64, o32, P, tol=1e-4, tie_frac=1e-3, tie_tol=2e-2, log=print):
    """every parameter gradient of the captured HIP step against the chain-mode oracle results (fp64, and the fp32 build of the C
    rasterizer under the same fp64 torch graph).  Per-Gaussian groups: <= tol of the group's scale, except THRESHOLD TIES -- a pixel
    where fp32 and fp64 arithmetic decide alpha < 1/255 or T (1 - alpha) < 1e-4 differently moves the Gaussians on it by O(alpha):
    counted (<= tie_frac of the rows), bounded (tie_tol), and each must show in the fp32 oracle as well (tests/test_raster_gpu.py:
    _grad_vs_oracles).  Simulator groups are SUMS over every Gaussian and vertex that cancel as training converges (the image term and
    the regularisers balance): the rounding of the fp32 rasterizer arithmetic is then large against the sum itself, whoever does it --
    they are reported next to the distance of the fp32 ORACLE from the fp64 one for the same group (a Gaussian that decides a threshold
    differently on the two sides' slightly different rasterizer inputs shows in the sum).  The rigorous bars
    are raster_stage's and pre_stage's.  Returns {name: (max err, ties or fp32-oracle err)}."""
    res = {}
    for i, name in enumerate(cap.names):
        g, r64, r32 = cap.grads[i], o64["grads"][i], o32["grads"][i] if o32 is not None else None
        if g is None or r64 is None:
            assert g is None and r64 is None, (name, "gradient present on one side only")
            continue
        g = g.cpu().numpy()
        if i < len(GAUSS_NAMES):
            d = rows_err(g, r64.numpy())
            ties = d > tol
            assert ties.sum() <= max(tie_frac * P, 0), (name, int(ties.sum()), float(d.max()))
            assert d.max() <= tie_tol, (name, float(d.max()))
            # (no "the fp32 oracle shares it" requirement HERE: the replay from the parameters hands the oracle rasterizer inputs that
            #  differ from the HIP path's by fp32 rounding, so the two decide different ties; raster_stage holds that criterion on
            #  identical inputs)
            res[name] = (float(d[~ties].max()) if (~ties).any() else 0.0, int(ties.sum()))
        else:
            scale = np.abs(r64.numpy()).max() + 1e-30
            e = float(np.abs(g.astype(np.float64) - r64.numpy()).max() / scale)
            e32 = float(np.abs(r32.numpy() - r64.numpy()).max() / scale) if r32 is not None else 0.0
            # (reported, not asserted: raster_stage + pre_stage hold the bars.  Besides the cancellation, the simulator's two ReLU layers
            #  are discontinuous: a hidden unit whose pre-activation is within fp32 rounding of zero is on in one arithmetic and off in
            #  the other -- seen at step 350 of the parity run: HIP and fp32 torch BOTH 5.8e-2 from fp64 on output.weight, equal to each
            #  other; pre_stage's fp32-torch yardstick is what tells such a step from a defect)
            res[name] = (e, e32)
    log("   chain gradients (HIP vs fp64 oracle, shared dL/dimage; simulator groups: [fp32 oracle vs fp64]): " +
        " ".join(f"{k}:{v[0]:.1e}" + (f"(+{v[1]} ties)" if (isinstance(v[1], int) and v[1]) else (f"[{v[1]:.1e}]" if isinstance(v[1], float) else ""))
                 for k, v in res.items()))
    return res


def _ties_explained(name, got, g32, g64, rows, tol, tie_frac, tie_tol, exp_tie_frac=2e-4):
    """tests/test_raster_gpu.py:_grad_vs_oracles for one gradient on IDENTICAL rasterizer inputs: <= tol against the fp32 oracle up to
    exp() ties (v_exp_f32 vs expf), <= tol against the fp64 oracle up to counted, bounded ties that the fp32 oracle (or an exp tie) shares"""
    got, a32, a64 = (np.asarray(x, np.float64).reshape(rows, -1) for x in (got, g32, g64))
    scale = np.abs(a64).max() + 1e-30
    e32 = np.abs(got - a32).max(1) / scale
    t32 = e32 > tol
    assert t32.sum() <= max(exp_tie_frac * rows, 2), (name, "vs fp32 oracle", int(t32.sum()), float(e32.max()))
    d = np.abs(got - a64).max(1) / scale
    ties = d > tol
    # (ONE pixel whose alpha / transmittance test falls on the other side moves every Gaussian BEHIND it on that pixel -- dozens at the
    #  depths these scenes have -- so the count has a floor; what keeps the check sharp is the next two lines: every deviation bounded,
    #  and every one of them present in the fp32 oracle's own distance from the fp64 one on the SAME inputs)
    assert ties.sum() <= max(tie_frac * rows, 48), (name, int(ties.sum()))
    assert max(d.max(), e32.max()) <= tie_tol, (name, float(d.max()), float(e32.max()))
    d32 = np.abs(a32 - a64).max(1) / scale
    assert np.all((d32[ties] > 0.5 * tol) | t32[ties]), (name, "a deviation from fp64 that neither the fp32 oracle nor an exp tie explains")
    return float(d[~ties].max()) if (~ties).any() else 0.0, int(ties.sum())


def raster_stage(cap, cams_cpu, bg_np, sh_degree=3, tol=1e-4, tie_frac=1e-3, tie_tol=2e-2, radii=None, vsg=None, log=print):
    """The rasterizer node of the captured step against the oracle ON THE SAME INPUTS: image (both builds), radii (bit-exact, max over the
    cameras), and -- backward driven with the step's own dL/dimage -- the gradient of every rasterizer input: means3D and rotations per
    camera, opacity / scales / SH summed over the cameras as the batched backward returns them."""
    from util import image_err
    assert cap.rin is not None and cap.rgrad, "the step did not go through MeshGaussians.step_inputs"
    n = lambda t: t.detach().cpu().numpy()  # noqa: E731
    P = cap.rin["op"].shape[0]
    img, dimg = n(cap.image), n(cap.dimage)
    sums = {k: {"op": 0.0, "sc": 0.0, "sh": 0.0, "m2d": 0.0} for k in (32, 64)}
    rad = []
    out = {}
    for b, cam in enumerate(cams_cpu):
        res = {}
        for dt, key in ((np.float32, 32), (np.float64, 64)):
            o = ro.forward(n(cap.rin["means3D"][b]), n(cap.rin["op"]), n(cam.world_view_transform), n(cam.full_proj_transform),
                           n(cam.camera_center), np.tan(cam.FoVx * 0.5), np.tan(cam.FoVy * 0.5), int(cam.image_width), int(cam.image_height),
                           bg_np, shs=n(cap.rin["sh"]), sh_degree=sh_degree, scales=n(cap.rin["sc"]), rotations=n(cap.rin["rot"][b]), dtype=dt)
            res[key] = (o, ro.backward(o, np.ascontiguousarray(dimg[b].astype(dt))))
        o32, o64 = res[32][0], res[64][0]
        rad.append(o32.radii)
        assert image_err(img[b], o32.color) < tol and image_err(img[b], o64.color, outlier_frac=1e-3) < tol, ("image", b)
        for name, attr in (("means3D", "mean3D"), ("rot", "rot")):
            if (name, b) in cap.rgrad:
                out[f"{name}[{b}]"] = _ties_explained(f"{name}[{b}]", n(cap.rgrad[(name, b)]), getattr(res[32][1], attr), getattr(res[64][1], attr),
                                                      P, tol, tie_frac, tie_tol)
        for key in (32, 64):
            g = res[key][1]
            sums[key]["op"] = sums[key]["op"] + np.asarray(g.opacity, np.float64).reshape(P, -1)
            sums[key]["sc"] = sums[key]["sc"] + np.asarray(g.scale, np.float64).reshape(P, -1)
            sums[key]["sh"] = sums[key]["sh"] + np.asarray(g.sh, np.float64).reshape(P, -1)
            sums[key]["m2d"] = sums[key]["m2d"] + np.asarray(g.mean2D, np.float64).reshape(P, -1)
    for key in ("op", "sc", "sh"):
        if key in cap.rgrad:
            out[key] = _ties_explained(key, n(cap.rgrad[key]), sums[32][key], sums[64][key], P, tol, 4 * tie_frac, tie_tol)
    if vsg is not None:     # the summed screen-space (NDC) gradient densification consumes (train_utils.py:290-292)
        out["viewspace_grad"] = _ties_explained("viewspace_grad", n(vsg), sums[32]["m2d"], sums[64]["m2d"], P, tol, 4 * tie_frac, tie_tol)
    if radii is not None:
        np.testing.assert_array_equal(n(radii), np.max(np.stack(rad), 0))
    log("   rasterizer node on the step's own inputs (HIP vs fp64 oracle, ties shared with the fp32 oracle): " +
        " ".join(f"{k}:{v[0]:.1e}" + (f"(+{v[1]} ties)" if v[1] else "") for k, v in out.items()))
    return out


def pre_stage(build_cpu, cams_cpu, cap, tol=1e-4, opt=None, log=print, build_cpu32=None):
    """Everything in FRONT of the rasterizer, from the parameter snapshot, as fp64 torch (simulator, cloth regularisers, mesh -> Gaussian
    transform, activations), differentiated with the HIP step's own gradients of the rasterizer inputs + the regularisers' unit weight:
    every parameter gradient of the step.  Smooth nodes, no ties -- but not well conditioned: the Kabsch rotation of a face responds to its
    vertices with 1 / edge length (100 on the config-3 mesh), a vertex collects ~30 such terms that largely cancel, and the simulator's
    gradients sum those over all vertices.  The bar is max(tol, 3 x the distance of the SAME nodes evaluated in fp32 torch on the CPU --
    the reference's own arithmetic for this part, gaussian_mesh.py:151-188 / meshnet_network.py:361-373 -- from the fp64 evaluation)."""
    from csplat import train as tr
    opt = opt or tr.DEFAULT_OPT

    def run(builder, dt, flips=(), probe=None):
        """flips: L1 kinks of the regularisers evaluated on the OTHER side -- (kind, index tuple, sign of the residual in fp64): the term
        -2 sign w x is added, so that d|x|/dx reads -sign.  probe: a list that receives the fp64 residuals (rigid, momentum) and weights."""
        pc, sim = builder()
        pc.fused = False
        # the simulator's two ReLU layers (meshnet_network.py:364-366): a hidden unit whose pre-activation is zero to fp32 rounding is on in
        # one arithmetic and off in the other.  The hooks record the pre-activations of every call (probe) and, for a flip ("relu", (layer,
        # call, unit), _), hand the ReLU z - 2 z.detach(): the value moves by 2 |z| ~ 1e-8, the gate stands on the other side
        calls = {"input": 0, "hidden": 0}
        hooks = []
        for lname in ("input", "hidden"):
            lin = getattr(sim, lname, None)
            if lin is None:
                continue

            def hook(_m, _inp, out, lname=lname):
                b_ = calls[lname]
                calls[lname] += 1
                if probe is not None:
                    probe_relu.append((lname, b_, out.detach().reshape(-1).clone()))
                mine = [idx[2] for kind, idx, _ in flips if kind == "relu" and idx[0] == lname and idx[1] == b_]
                if mine:
                    sel = torch.zeros(out.shape[-1], dtype=out.dtype)
                    sel[mine] = 2.0
                    return out - sel * out.detach()
                return None
            hooks.append(lin.register_forward_hook(hook))
        probe_relu = []
        ps = list(pc.parameters()) + list(sim.parameters())
        with torch.no_grad():
            for a, b in zip(ps, cap.params):
                a.copy_(b.detach().cpu().to(dt))
        for p in ps:
            p.grad = None
        V = pc.mesh.pos.shape[0]
        outs, grads, verts = [], [], []
        g_ = lambda t: t.detach().cpu().to(dt)  # noqa: E731
        for b, c in enumerate(cams_cpu):
            v = sim(time_vector=torch.tensor(c.time, dtype=pc.mesh.pos.dtype).repeat(V, 1))
            verts.append(v[None])
            if ("means3D", b) in cap.rgrad:
                outs.append(pc.get_xyz(v)); grads.append(g_(cap.rgrad[("means3D", b)]))
            if ("rot", b) in cap.rgrad:
                outs.append(pc.get_rotation(v)); grads.append(g_(cap.rgrad[("rot", b)]))
        for key, t in (("op", pc.get_opacity), ("sc", pc.get_scaling), ("sh", pc.get_features)):
            if key in cap.rgrad:
                outs.append(t); grads.append(g_(cap.rgrad[key]).reshape(t.shape))
        allv = torch.cat(verts, 0)
        reg = tr.regularization(allv, pc, opt)
        n_c = allv.shape[0]
        ei = pc.mesh.edge_index
        x = m = None
        w_r = w_m = 0.0
        if opt.lambda_rigid > 0:              # train_utils.py:76-237 via csplat.train.regularization: l1_loss(static_norm, deformed_norm)
            x = torch.linalg.norm(allv.index_select(1, ei[1]) - allv.index_select(1, ei[0]), dim=-1) - pc.edge_norm.reshape(1, -1).to(dt)
            w_r = opt.lambda_rigid / x.numel()
        if opt.lambda_momentum > 0 and n_c >= 3:      # ... mean over vertices of the L1 norm of the second difference
            m = allv[2] - 2 * allv[1] + allv[0]
            w_m = opt.lambda_momentum / m.shape[0]
        if probe is not None:
            probe.append((None if x is None else x.detach().clone(), w_r, None if m is None else m.detach().clone(), w_m,
                          pc.edge_norm.reshape(1, -1).detach().clone(), float(allv.detach().abs().max()), probe_relu))
        for kind, idx, sgn in flips:
            if kind != "relu":
                reg = reg - 2.0 * sgn * ((w_r * x[idx]) if kind == "rigid" else (w_m * m[idx]))
        torch.autograd.backward(outs + [reg], grads + [torch.ones((), dtype=reg.dtype)])
        for h_ in hooks:
            h_.remove()
        return [None if p.grad is None else p.grad.detach().double() for p in ps]
    probe = []
    g64 = run(build_cpu, torch.float64, probe=probe)
    g32 = run(build_cpu32, torch.float32) if build_cpu32 is not None else None
    # ---- L1 kinks.  The rigidity term is an L1 distance of edge lengths, the momentum term an L1 norm: where a residual is zero to fp32
    # rounding, the HIP kernel (its own order of operations) and torch may stand on different sides and their gradients differ by twice
    # that term's weight at the edge's two vertices / the vertex's coordinate -- a tie of the same kind as a sign of the image L1
    # (loss_node_err counts those).  Candidates: residuals below 1e-6 of the edge length / of the coordinates' magnitude.  A candidate is
    # FLIPPED on the fp64 / fp32 side when that moves the fp64 gradient TOWARDS the HIP one by more than half the flip's own length; the
    # flips are counted, reported and bounded.
    x64, w_r, m64, w_m, enorm, vmax, relu64 = probe[0]
    cands = []
    if x64 is not None:
        for t_, e_ in torch.nonzero(x64.abs() <= 1e-6 * enorm.to(x64.dtype)).tolist():
            cands.append(("rigid", (t_, e_), 1.0 if float(x64[t_, e_]) >= 0 else -1.0))
    if m64 is not None:
        for v_, c_ in torch.nonzero(m64.abs() <= 1e-6 * vmax).tolist():
            cands.append(("momentum", (v_, c_), 1.0 if float(m64[v_, c_]) >= 0 else -1.0))
    for lname, b_, z in relu64:
        for (u_,) in torch.nonzero(z.abs() <= 1e-6 * float(z.abs().max())).tolist():
            cands.append(("relu", (lname, b_, u_), 0.0))
    flips = []
    if cands and len(cands) <= 64:
        def flat(gs, ref):
            return torch.cat([(torch.zeros_like(r) if g_ is None else g_.detach().cpu().double()).reshape(-1) for g_, r in zip(gs, ref) if r is not None])
        # (a flip's effect on the gradient does not depend on the other flips -- the L1 terms are separable -- but the effects overlap where
        #  edges share vertices: greedy over the candidates with the residual UPDATED after every accepted flip, twice over the list)
        base = flat(g64, g64)
        resid = flat(cap.grads, g64) - base
        deltas = [flat(run(build_cpu, torch.float64, flips=[cnd]), g64) - base for cnd in cands]
        taken = [False] * len(cands)
        for _sweep in range(2):
            for ci, (cnd, delta) in enumerate(zip(cands, deltas)):
                if not taken[ci] and float(resid @ delta) > 0.5 * float(delta @ delta) > 0.0:
                    taken[ci] = True
                    flips.append(cnd)
                    resid = resid - delta
        assert len(flips) <= 8, ("regulariser L1 ties", len(flips), len(cands))
        if flips:
            g64 = run(build_cpu, torch.float64, flips=flips)
            if g32 is not None:
                g32 = run(build_cpu32, torch.float32, flips=flips)
            log(f"   kinks in front of the rasterizer (L1 residuals of the regularisers, ReLU gates of the simulator): {len(flips)} of "
                f"{len(cands)} near-zero candidates stand on the other side in the HIP step: " + " ".join(f"{k}{i}" for k, i, _ in flips))
    res = {}
    for i, (name, g) in enumerate(zip(cap.names, cap.grads)):
        if g is None or g64[i] is None:
            assert g is None and g64[i] is None, (name, "gradient present on one side only")
            continue
        scale = float(g64[i].abs().max()) + 1e-30
        e = float((g.detach().cpu().double() - g64[i]).abs().max()) / scale
        e32 = float((g32[i] - g64[i]).abs().max()) / scale if g32 is not None else 0.0
        assert e <= max(tol, 3.0 * e32), (name, e, e32)
        res[name] = (e, e32)
    log("   nodes in front of the rasterizer (HIP vs fp64 torch [fp32 torch vs fp64], the step's own rasterizer-input gradients): " +
        " ".join(f"{k}:{v[0]:.1e}[{v[1]:.1e}]" for k, v in res.items()))
    return res
