"""Synthetic scene_1: the reference ships no data (README.md:46-53), so bench.py,
smoke() and the parity tests all draw their inputs from this seeded generator.  numpy only.

Camera math restates (and tests/golden pins against the reference's own modules):
  pose_spherical            scene_reconstruction/dataset_readers.py:205-224
  c2w -> (R, T)             scene_reconstruction/dataset_readers.py:352-359
  getWorld2View2            utils/graphics_utils.py:38-49
  getProjectionMatrix       utils/graphics_utils.py:51-71
  Camera matrices           scene_reconstruction/cameras.py:57-68
"""
import math

import numpy as np

SEED = 6666  # train.py:360
CAMERA_ANGLE_X = 0.6911
ZNEAR, ZFAR = 0.01, 100.0  # cameras.py:57-58


def pose_spherical(theta_deg, phi_deg, radius):
    def trans_t(t):
        return np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, t], [0, 0, 0, 1]], np.float32)

    def rot_phi(phi):
        return np.array([[1, 0, 0, 0], [0, np.cos(phi), -np.sin(phi), 0], [0, np.sin(phi), np.cos(phi), 0],
                         [0, 0, 0, 1]], np.float32)

    def rot_theta(th):
        return np.array([[np.cos(th), 0, -np.sin(th), 0], [0, 1, 0, 0], [np.sin(th), 0, np.cos(th), 0],
                         [0, 0, 0, 1]], np.float32)

    c2w = trans_t(radius)
    c2w = rot_phi(phi_deg / 180.0 * np.pi) @ c2w
    c2w = rot_theta(theta_deg / 180.0 * np.pi) @ c2w
    c2w = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], np.float32) @ c2w
    return c2w


def c2w_to_RT(c2w):
    """Blender/OpenGL camera-to-world -> (R, T) as the reference stores them (R transposed)."""
    c2w = np.array(c2w, np.float64)
    c2w[:3, 1:3] *= -1
    w2c = np.linalg.inv(c2w)
    return np.transpose(w2c[:3, :3]), w2c[:3, 3]


def world_to_view(R, t, translate=np.array([0.0, 0.0, 0.0]), scale=1.0):
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = R.transpose()
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    C2W = np.linalg.inv(Rt)
    C2W[:3, 3] = (C2W[:3, 3] + translate) * scale
    return np.float32(np.linalg.inv(C2W))


def projection_matrix(znear, zfar, fovX, fovY):
    tanY, tanX = math.tan(fovY / 2), math.tan(fovX / 2)
    top, right = tanY * znear, tanX * znear
    bottom, left = -top, -right
    P = np.zeros((4, 4), np.float32)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def camera_matrices(R, T, fovx, fovy, znear=ZNEAR, zfar=ZFAR):
    """-> (world_view_transform, full_proj_transform, camera_center), all float32, transposed (row-vector) form."""
    wv = world_to_view(R, T).transpose()
    pr = projection_matrix(znear, zfar, fovx, fovy).transpose()
    full = (wv.astype(np.float32) @ pr.astype(np.float32)).astype(np.float32)
    center = np.linalg.inv(wv.astype(np.float64))[3, :3].astype(np.float32)
    return np.ascontiguousarray(wv, np.float32), np.ascontiguousarray(full), center


def make_camera(theta_deg, W, H, time=0.0, phi_deg=-30.0, radius=4.0, fovx=CAMERA_ANGLE_X):
    R, T = c2w_to_RT(pose_spherical(theta_deg, phi_deg, radius))
    focal = W / (2 * math.tan(fovx / 2))
    fovy = 2 * math.atan(H / (2 * focal))
    wv, full, center = camera_matrices(R, T, fovx, fovy)
    return dict(R=R, T=T, FoVx=fovx, FoVy=fovy, image_width=W, image_height=H, time=float(time),
                world_view_transform=wv, full_proj_transform=full, camera_center=center,
                tanfovx=math.tan(fovx * 0.5), tanfovy=math.tan(fovy * 0.5))


def grid_mesh(g):
    """g x g vertices on [-0.5,0.5]^2, z = 0.05 sin(3x) cos(3y); quads split into two triangles."""
    u = np.linspace(-0.5, 0.5, g)
    X, Y = np.meshgrid(u, u, indexing="xy")
    pos = np.stack([X, Y, 0.05 * np.sin(3 * X) * np.cos(3 * Y)], -1).reshape(-1, 3).astype(np.float32)
    idx = np.arange(g * g).reshape(g, g)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel()
    faces = np.concatenate([np.stack([a, b, c], 1), np.stack([b, d, c], 1)], 0).astype(np.int64)
    e = np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]], 0)
    e = np.concatenate([e, e[:, ::-1]], 0)
    e = np.unique(e, axis=0)  # undirected both ways, coalesced (sorted by (row, col))
    return pos, faces, np.ascontiguousarray(e.T)


def mesh_trajectory(pos, n_times=30):
    t = np.arange(n_times, dtype=np.float32)[:, None, None] / n_times
    wave = np.zeros((n_times,) + pos.shape, np.float32)
    wave[..., 2] = (0.02 * t * np.sin(6 * pos[None, :, 0:1] + 4 * t) * np.cos(5 * pos[None, :, 1:2]))[..., 0]
    return pos[None] + wave


def scene_1(P=100_000, W=800, H=800, n_cams=4, grid=100, n_times=30, seed=SEED):
    """Returns a dict of numpy arrays describing synthetic scene_1."""
    rng = np.random.default_rng(seed)
    pos, faces, edge_index = grid_mesh(grid)
    traj = mesh_trajectory(pos, n_times)
    F = faces.shape[0]
    face_ids = rng.integers(0, F, size=P).astype(np.int64)
    bary = rng.dirichlet([1.0, 1.0, 1.0], size=P).astype(np.float32)
    log_scales = rng.normal(math.log(0.01), 0.3, size=(P, 3)).astype(np.float32)
    quats = rng.normal(size=(P, 4)).astype(np.float32)
    quats /= np.linalg.norm(quats, axis=1, keepdims=True)
    opacity_logits = rng.normal(0.0, 1.5, size=(P, 1)).astype(np.float32)
    sh = np.concatenate([rng.normal(0, 1.0, size=(P, 1, 3)), rng.normal(0, 0.1, size=(P, 15, 3))], 1).astype(np.float32)
    thetas = [0.0] if n_cams == 1 else [-180.0 + 360.0 * k / n_cams for k in range(n_cams)]
    cams = [make_camera(th, W, H, time=0.0) for th in thetas]
    return dict(mesh_pos=traj, faces=faces, edge_index=edge_index, face_ids=face_ids, bary=bary,
                log_scales=log_scales, quats=quats, opacity_logits=opacity_logits, sh=sh, cameras=cams,
                bg=np.ones(3, np.float32), sh_degree=3)


def gaussians_at(scene, t=0):
    """Rasterizer-level inputs at mesh timestep t: barycentric centres (gaussian_mesh.py:151-169),
    scales = exp(log_scales), opacities = sigmoid(logits); rotations are the stored unit quaternions."""
    v = scene["mesh_pos"][t][scene["faces"][scene["face_ids"]]]  # [P,3,3]
    b = scene["bary"] / scene["bary"].sum(1, keepdims=True)
    means = np.einsum("pk,pkc->pc", b, v).astype(np.float32)
    return dict(means3D=means, scales=np.exp(scene["log_scales"]).astype(np.float32), rotations=scene["quats"],
                opacities=(1.0 / (1.0 + np.exp(-scene["opacity_logits"]))).astype(np.float32), shs=scene["sh"])
