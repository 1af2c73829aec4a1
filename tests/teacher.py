"""Teacher-forced gradient parity of the training step (test infrastructure; may import oracle/).

Reference step being matched: /root/reference/scene_reconstruction/train_utils.py:240-321 (render the step's cameras, stack, L1 +
lambda (1 - SSIM) + cloth regularisers, ONE backward, statistics for densification, two Adam steps).

`capture(pc, sim)` arms the REAL csplat.train.train_step on the HIP path: the next step leaves (a) the parameters it started from,
(b) every parameter gradient it handed the optimizers, (c) the image batch it rendered and (d) dL/dimage as autograd delivered it
to the rasterizer's backward.  `oracle_step(...)` then evaluates the same step from (a) on the CPU: simulator, mesh -> Gaussian
transform, image loss and regularisers as fp64 torch, the rasterizer = oracle/raster_ref.c (fp64 or fp32 build) with its
analytic backward.  Three comparisons come out of it:

  * chain: the oracle's backward is driven with the HIP step's OWN dL/dimage (c -> d is checked separately below).  This holds
    every kernel between the parameters and the image -- simulator, regularisers, mesh transform, activations, K1-K8 -- to the
    oracle without the one chaotic element of the step: the L1 term's sign(render - gt), which flips wherever two renders that
    agree to 1e-6 straddle the target (that, not a kernel, is what separates fp32 and fp64 trajectories).
  * loss node: dL/dimage of the HIP step against autograd over the fp64 torch formulation of the image loss AT THE HIP IMAGE.
  * end to end: the plain fp64 gradient of the whole step (sign flips included) -- reported, with the number of flipped pixels.
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

    name = "step_now" if hasattr(opt, "step_now") else "step"
    setattr(opt, name, step_wrapper)
    tr.FusedImageLoss = LossShim
    try:
        yield cap
    finally:
        delattr(opt, name) if name in opt.__dict__ else None
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
    return out


def rows_err(got, ref, scale=None):
    """per-row max |got - ref| / max |ref| (rows = leading dimension)"""
    a = np.asarray(got, np.float64)
    b = np.asarray(ref, np.float64)
    a, b = a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1)
    return np.abs(a - b).max(1) / ((np.abs(b).max() if scale is None else scale) + 1e-30)


def compare_chain(cap, o64, o32, P, tol=1e-4, tie_frac=1e-3, tie_tol=2e-2, log=print):
    """every parameter gradient of the captured HIP step against the chain-mode oracle results (fp64, and the fp32 build of the C
    rasterizer under the same fp64 torch graph).  Per-Gaussian groups: <= tol of the group's scale, except THRESHOLD TIES -- a pixel
    where fp32 and fp64 arithmetic decide alpha < 1/255 or T (1 - alpha) < 1e-4 differently moves the Gaussians on it by O(alpha):
    counted (<= tie_frac of the rows), bounded (tie_tol), and each must show in the fp32 oracle as well (tests/test_raster_gpu.py:
    _grad_vs_oracles).  Simulator groups (sums over all Gaussians): <= tol outright.  Returns {name: (max err, ties)}."""
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
            if ties.any():
                assert r32 is not None, (name, int(ties.sum()), float(d.max()))
                d32 = rows_err(r32.numpy(), r64.numpy())
                assert np.all(d32[ties] > 0.5 * tol), (name, "a deviation from fp64 that the fp32 oracle does not share", float(d[ties].max()))
            res[name] = (float(d[~ties].max()) if (~ties).any() else 0.0, int(ties.sum()))
        else:
            e = float(np.abs(g.astype(np.float64) - r64.numpy()).max() / (np.abs(r64.numpy()).max() + 1e-30))
            assert e <= tol, (name, e)
            res[name] = (e, 0)
    log("   chain gradients (HIP vs fp64 oracle, shared dL/dimage): " + " ".join(f"{k}:{v[0]:.1e}" + (f"(+{v[1]} ties)" if v[1] else "")
                                                                                  for k, v in res.items()))
    return res
