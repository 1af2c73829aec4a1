"""Rotation helpers the mesh->Gaussian transform needs (the reference takes them from the un-pinned pip package
`roma`, requirements.txt:16; call sites scene_reconstruction/gaussian_mesh.py:186-188).  Restated from roma's
published behaviour -- PARITY UNPINNED (roma is not installed here):
  rigid_points_registration(x, y)  Kabsch: R = U diag(1,1,det(U V^T)) V^T of  sum_k yhat_k xhat_k^T
  rotmat_to_unitquat(R)            XYZW, largest-of-(diag, trace) branch (scipy-style)
  quat_product / quat_composition  XYZW Hamilton product
"""
import torch


def rigid_points_registration(x, y):
    """x, y [..., K, 3] -> (R [..., 3, 3], t [..., 3]) minimising sum |R x + t - y|^2."""
    xm, ym = x.mean(dim=-2, keepdim=True), y.mean(dim=-2, keepdim=True)
    xh, yh = x - xm, y - ym
    M = yh.transpose(-1, -2) @ xh
    U, _, Vh = torch.linalg.svd(M)
    d = torch.det(U @ Vh)
    D = torch.ones_like(M[..., 0])
    D = torch.cat([D[..., :2], d.unsqueeze(-1)], dim=-1)
    R = (U * D.unsqueeze(-2)) @ Vh
    t = ym.squeeze(-2) - (R @ xm.transpose(-1, -2)).squeeze(-1)
    return R, t


def _plane_basis(p):
    """orthonormal (u, v, n) spanning the plane of centred triangle points p [..., 3, 3] (rows = points)."""
    u = torch.nn.functional.normalize(p[..., 0, :], dim=-1)
    t = p[..., 1, :] - (p[..., 1, :] * u).sum(-1, keepdim=True) * u
    v = torch.nn.functional.normalize(t, dim=-1)
    return u, v, torch.cross(u, v, dim=-1)


def kabsch_triangles(x, y):
    """Closed form of rigid_points_registration for K = 3 points (the per-Gaussian face registration of
    scene_reconstruction/gaussian_mesh.py:182-186): no SVD.  The 3x3 covariance of two centred triangles has rank 2:
    M = By M2 Bx^T with M2 2x2 in the two triangle planes; R = By Q Bx^T + det(Q) ny nx^T where Q is the orthogonal polar
    factor of M2 -- a rotation (M2 + cof M2, normalised) when det M2 > 0, a reflection (M2 - cof M2, normalised) when
    det M2 < 0, in which case Kabsch's determinant fix flips the (zero) third singular direction.  Equals the SVD
    solution up to rounding; differentiable w.r.t. y; ~25 elementwise kernels instead of two batched 3x3 SVDs."""
    xh = x - x.mean(dim=-2, keepdim=True)
    yh = y - y.mean(dim=-2, keepdim=True)
    ux, vx, nx = _plane_basis(xh)
    uy, vy, ny = _plane_basis(yh)
    xu, xv = (xh * ux.unsqueeze(-2)).sum(-1), (xh * vx.unsqueeze(-2)).sum(-1)      # [..., 3] in-plane coordinates
    yu, yv = (yh * uy.unsqueeze(-2)).sum(-1), (yh * vy.unsqueeze(-2)).sum(-1)
    a, b = (yu * xu).sum(-1), (yu * xv).sum(-1)
    c, d = (yv * xu).sum(-1), (yv * xv).sum(-1)
    pos = (a * d - b * c) > 0
    q00 = torch.where(pos, a + d, a - d)
    q01 = torch.where(pos, b - c, b + c)
    nrm = torch.rsqrt(q00 * q00 + q01 * q01)
    q00, q01 = q00 * nrm, q01 * nrm
    q10 = torch.where(pos, -q01, q01)
    q11 = torch.where(pos, q00, -q00)
    sgn = torch.where(pos, torch.ones_like(a), -torch.ones_like(a))
    o = lambda p, q: p.unsqueeze(-1) * q.unsqueeze(-2)  # noqa: E731  outer product
    e = lambda s_: s_[..., None, None]  # noqa: E731
    return e(q00) * o(uy, ux) + e(q01) * o(uy, vx) + e(q10) * o(vy, ux) + e(q11) * o(vy, vx) + e(sgn) * o(ny, nx)


def rotmat_to_unitquat(R):
    """[..., 3, 3] -> [..., 4] XYZW."""
    shape = R.shape[:-2]
    m = R.reshape(-1, 3, 3)
    diag = torch.stack([m[:, 0, 0], m[:, 1, 1], m[:, 2, 2]], dim=1)
    tr = diag.sum(dim=1)
    dec = torch.cat([diag, tr[:, None]], dim=1)
    choice = dec.argmax(dim=1)
    qs = []
    for i in range(3):
        j, k = (i + 1) % 3, (i + 2) % 3
        q = [None] * 4
        q[i] = 1 - tr + 2 * m[:, i, i]
        q[j] = m[:, j, i] + m[:, i, j]
        q[k] = m[:, k, i] + m[:, i, k]
        q[3] = m[:, k, j] - m[:, j, k]
        qs.append(torch.stack(q, dim=1))
    qs.append(torch.stack([m[:, 2, 1] - m[:, 1, 2], m[:, 0, 2] - m[:, 2, 0], m[:, 1, 0] - m[:, 0, 1], 1 + tr], dim=1))
    q = torch.zeros_like(qs[0])
    for c in range(4):
        q = torch.where((choice == c)[:, None], qs[c], q)
    q = q / q.norm(dim=1, keepdim=True)
    return q.reshape(*shape, 4)


def quat_product(p, q):
    """XYZW Hamilton product p * q."""
    pv, pw, qv, qw = p[..., :3], p[..., 3:], q[..., :3], q[..., 3:]
    v = pw * qv + qw * pv + torch.cross(pv, qv, dim=-1)
    w = pw * qw - (pv * qv).sum(dim=-1, keepdim=True)
    return torch.cat([v, w], dim=-1)


def quat_composition(sequence):
    out = sequence[0]
    for q in sequence[1:]:
        out = quat_product(out, q)
    return out
