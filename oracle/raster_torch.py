"""oracle/raster_torch.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Second, independent restatement of the rasterizer forward as a
differentiable torch (CPU, fp64) program.  Its purpose is to check the ANALYTIC backward of
oracle/raster_ref.c (and of the HIP kernels) against torch.autograd, with the upstream gradient
conventions written out explicitly:

  * alpha = min(0.99, opacity*G) is straight-through (gradient passes when capped),
  * means2D receives the gradient w.r.t. an NDC-space offset of the projected centre
    (pixel gradient x 0.5*W / 0.5*H) -- what train_utils.py:290-292 accumulates,
  * the 1.3*tanfov clamp of the view-space centre stops the gradient of that axis,
  * SH colours clamped at 0 get zero gradient; the depth image is not differentiated,
  * tile lists / early termination are taken as constants (computed by oracle/raster_ref.c).

PARITY UNPINNED (see raster_ref.c).  Loops over tiles in Python: small cases only.
Reference call site: gaussian_renderer/__init__.py:156-164.
"""
import numpy as np
import torch

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435]


def _eval_sh(deg, sh, d):
    """sh [P,M,3], d [P,3] unit -> [P,3] (utils/sh_utils.py:57-112 restated for [P,M,3] layout)."""
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    r = C0 * sh[:, 0]
    if deg > 0:
        r = r - C1 * y * sh[:, 1] + C1 * z * sh[:, 2] - C1 * x * sh[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        r = (r + C2[0] * xy * sh[:, 4] + C2[1] * yz * sh[:, 5] + C2[2] * (2 * zz - xx - yy) * sh[:, 6]
             + C2[3] * xz * sh[:, 7] + C2[4] * (xx - yy) * sh[:, 8])
    if deg > 2:
        r = (r + C3[0] * y * (3 * xx - yy) * sh[:, 9] + C3[1] * xy * z * sh[:, 10]
             + C3[2] * y * (4 * zz - xx - yy) * sh[:, 11] + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
             + C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + C3[5] * z * (xx - yy) * sh[:, 14]
             + C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return r


def _rot(q):
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1)
    return R.reshape(-1, 3, 3)


def render(o, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
           cov3D_precomp=None, own_termination=True):
    """Differentiable forward.  `o` is the namespace from raster_oracle.forward() on the SAME inputs
    (supplies radii, tile ranges, sorted ids, n_contrib as constants and the camera).
    All tensor arguments are torch.float64 (leaf tensors with requires_grad as desired).
    own_termination=True recomputes the early-termination / n_contrib logic here instead of reading
    o.n_contrib.  Returns (color[3,H,W], depth[1,H,W], n_contrib[H,W])."""
    i = o._inputs
    W, H = o.W, o.H
    f64 = torch.float64
    V = torch.tensor(np.asarray(i.view, np.float64)).reshape(4, 4)  # row-vector convention: p_row @ V
    Pm = torch.tensor(np.asarray(i.proj, np.float64)).reshape(4, 4)
    campos = torch.tensor(np.asarray(i.campos, np.float64))
    bg = torch.tensor(np.asarray(i.bg, np.float64))
    tanx, tany, mod = float(i.tanfovx), float(i.tanfovy), float(i.scale_mod)
    fx, fy = W / (2 * tanx), H / (2 * tany)
    P = means3D.shape[0]
    ones = torch.ones(P, 1, dtype=f64)
    ph = torch.cat([means3D, ones], 1)
    pv = ph @ V
    hom = ph @ Pm
    pw = 1.0 / (hom[:, 3] + 1e-7)
    ndc = hom[:, :2] * pw[:, None] + means2D[:, :2]
    px = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5
    py = ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5
    if cov3D_precomp is None:
        R = _rot(rotations)
        A = R * (mod * scales)[:, None, :]
        Sig = A @ A.transpose(1, 2)
    else:
        c = cov3D_precomp
        Sig = torch.stack([c[:, 0], c[:, 1], c[:, 2], c[:, 1], c[:, 3], c[:, 4], c[:, 2], c[:, 4], c[:, 5]],
                          1).reshape(-1, 3, 3)
    tz = pv[:, 2]
    # culled Gaussians never reach a tile list; keep the math finite for them
    tz = torch.where(tz > 0.2, tz, torch.ones_like(tz))
    limx, limy = 1.3 * tanx, 1.3 * tany
    txtz, tytz = pv[:, 0] / tz, pv[:, 1] / tz
    inx = (txtz >= -limx) & (txtz <= limx)
    iny = (tytz >= -limy) & (tytz <= limy)
    tx = torch.where(inx, pv[:, 0], (txtz.clamp(-limx, limx) * tz).detach())
    ty = torch.where(iny, pv[:, 1], (tytz.clamp(-limy, limy) * tz).detach())
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -fx * tx / (tz * tz), zero, fy / tz, -fy * ty / (tz * tz)], 1).reshape(-1, 2, 3)
    Rw = V[:3, :3].T  # world->view rotation
    T = J @ Rw
    cov2 = T @ Sig @ T.transpose(1, 2)
    a = cov2[:, 0, 0] + 0.3
    b = cov2[:, 0, 1]
    c = cov2[:, 1, 1] + 0.3
    det = a * c - b * b
    det = torch.where(det == 0, torch.ones_like(det), det)
    conic = torch.stack([c / det, -b / det, a / det], 1)
    if shs is not None:
        d = means3D - campos[None]
        d = d / d.norm(dim=1, keepdim=True)
        rgb = torch.clamp_min(_eval_sh(o.D, shs, d) + 0.5, 0.0)
    else:
        rgb = colors_precomp
    depth = pv[:, 2]
    op = opacities.reshape(-1)

    color = torch.zeros(3, H, W, dtype=f64)
    dimg = torch.zeros(1, H, W, dtype=f64)
    gx = (W + 15) // 16
    ids_all = torch.from_numpy(o.ids.astype(np.int64))
    ncon = torch.from_numpy(o.n_contrib.astype(np.int64))
    ncon_out = ncon.clone() if not own_termination else torch.zeros(H, W, dtype=torch.int64)
    for t in range(o.ranges.shape[0]):
        s, e = int(o.ranges[t, 0]), int(o.ranges[t, 1])
        x0, y0 = (t % gx) * 16, (t // gx) * 16
        x1, y1 = min(x0 + 16, W), min(y0 + 16, H)
        if x1 <= x0 or y1 <= y0:
            continue
        ys, xs = torch.meshgrid(torch.arange(y0, y1), torch.arange(x0, x1), indexing="ij")
        xs = xs.reshape(-1).to(f64); ys = ys.reshape(-1).to(f64)
        npx = xs.shape[0]
        if e <= s:
            color[:, y0:y1, x0:x1] = bg[:, None, None].expand(3, y1 - y0, x1 - x0)
            continue
        g = ids_all[s:e]
        dx = px[g][None, :] - xs[:, None]
        dy = py[g][None, :] - ys[:, None]
        cn = conic[g]
        power = -0.5 * (cn[None, :, 0] * dx * dx + cn[None, :, 2] * dy * dy) - cn[None, :, 1] * dx * dy
        G = torch.exp(torch.clamp_max(power, 0.0))
        araw = op[g][None, :] * G
        alpha = araw + (torch.clamp_max(araw, 0.99) - araw).detach()
        idx = torch.arange(e - s)[None, :].expand(npx, -1)
        live = (power <= 0) & (alpha.detach() >= 1.0 / 255.0)
        if own_termination:
            with torch.no_grad():
                a0 = torch.where(live, alpha, torch.zeros_like(alpha))
                T0 = torch.cumprod(1.0 - a0, dim=1)
                stop = live & (T0 < 1e-4)
                # first stopping entry and everything after it is not blended
                dead = torch.cumsum(stop.to(torch.int64), dim=1) > 0
                live_t = live & ~dead
                last = torch.where(live_t, idx + 1, torch.zeros_like(idx)).max(dim=1).values
            ncon_out[y0:y1, x0:x1] = last.reshape(y1 - y0, x1 - x0)
            live = live_t
        else:
            last = ncon[y0:y1, x0:x1].reshape(-1)  # 1-based index of last blended entry (from raster_ref.c)
            live = live & (idx < last[:, None])
        alpha = torch.where(live, alpha, torch.zeros_like(alpha))
        one_m = 1.0 - alpha
        Tincl = torch.cumprod(one_m, dim=1)
        Tbefore = torch.cat([torch.ones(npx, 1, dtype=f64), Tincl[:, :-1]], 1)
        w = alpha * Tbefore
        Cpix = w @ rgb[g]
        Dpix = w @ depth[g]
        Tfin = Tincl[:, -1]
        Cpix = Cpix + Tfin[:, None] * bg[None, :]
        color[:, y0:y1, x0:x1] = Cpix.T.reshape(3, y1 - y0, x1 - x0)
        dimg[0, y0:y1, x0:x1] = Dpix.reshape(y1 - y0, x1 - x0)
    return color, dimg, ncon_out
