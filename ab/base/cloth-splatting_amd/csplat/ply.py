"""point_cloud.ply I/O in the layout the reference writes with `plyfile` (first half):
/root/reference/scene_reconstruction/gaussian_model.py:181-212 (attribute list, save) and :219-262 (load),
gaussian_mesh.py:433-481 (+ b1, b2, b3, o, id; every property float32, one `vertex` element, binary little endian --
plyfile's default on a little-endian host).  `plyfile` and `h5py` are not installed in this image: the PLY container is
small enough to write directly (header text + packed records); the mesh side-car `mesh.hdf5` (gaussian_mesh.py:462-465,481)
is written and read by csplat/hdf5min.py -- the flat-root-group, contiguous-dataset subset of HDF5 that file uses -- or by h5py
when it is importable.  (Directories saved by round 1 hold `mesh.npz`; it is still read.)"""
import os

import numpy as np

_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2",
          "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}


def write_ply(path, names, columns):
    """one `vertex` element, all properties float32, binary_little_endian 1.0; columns: [N, len(names)] array"""
    a = np.ascontiguousarray(np.asarray(columns, dtype="<f4"))
    assert a.ndim == 2 and a.shape[1] == len(names)
    header = ["ply", "format binary_little_endian 1.0", f"element vertex {a.shape[0]}"]
    header += [f"property float {n}" for n in names] + ["end_header"]
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        f.write(a.tobytes())


def read_ply(path):
    """-> {property name: 1-D array} of the FIRST element (binary little/big endian or ascii; scalar properties only)"""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, props, count, in_first, seen = None, [], None, False, 0
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: header not terminated")
            tok = line.decode("ascii").split()
            if not tok or tok[0] == "comment" or tok[0] == "obj_info":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                seen += 1
                in_first = seen == 1
                if in_first:
                    count = int(tok[2])
            elif tok[0] == "property" and in_first:
                if tok[1] == "list":
                    raise ValueError(f"{path}: list properties are not supported")
                props.append((tok[2], _TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt == "ascii":
            rows = np.loadtxt(f, max_rows=count, ndmin=2) if count else np.zeros((0, len(props)))
            return {n: rows[:, i].astype(t) for i, (n, t) in enumerate(props)}
        order = "<" if fmt == "binary_little_endian" else ">"
        dt = np.dtype([(n, order + t) for n, t in props])
        rec = np.frombuffer(f.read(count * dt.itemsize), dtype=dt, count=count)
        return {n: np.ascontiguousarray(rec[n]) for n, _ in props}


def attribute_names(n_dc, n_rest, n_scale=3, n_rot=4, mesh=True):
    """gaussian_model.py:181-193 (+ gaussian_mesh.py:433-436)"""
    names = ["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(n_dc)] + [f"f_rest_{i}" for i in range(n_rest)]
    names += ["opacity"] + [f"scale_{i}" for i in range(n_scale)] + [f"rot_{i}" for i in range(n_rot)]
    return names + (["b1", "b2", "b3", "o", "id"] if mesh else [])


def save_gaussians(pc, path):
    """MultiGaussianMesh.save_ply: <path>/point_cloud.ply + the mesh side-car"""
    import torch
    os.makedirs(path, exist_ok=True)
    t = lambda x: x.detach().cpu().numpy()  # noqa: E731
    f_dc = t(pc._features_dc.detach().transpose(1, 2).flatten(start_dim=1).contiguous())
    f_rest = t(pc._features_rest.detach().transpose(1, 2).flatten(start_dim=1).contiguous())
    bary = t(pc.face_bary)
    cols = np.concatenate((t(pc.get_xyz()), np.zeros_like(bary), f_dc, f_rest, t(pc._opacity), t(pc._scaling), t(pc._rotation), bary,
                           t(pc.face_offset), t(pc.face_ids.unsqueeze(1)).astype(np.float32)), axis=1)
    write_ply(os.path.join(path, "point_cloud.ply"), attribute_names(f_dc.shape[1], f_rest.shape[1]), cols)
    mesh = {k: t(v) for k, v in vars(pc.mesh).items() if torch.is_tensor(v)}
    try:
        import h5py
        with h5py.File(os.path.join(path, "mesh.hdf5"), "w") as f:
            for k, v in mesh.items():
                f.create_dataset(k, data=v)
    except ImportError:
        from . import hdf5min
        hdf5min.save(os.path.join(path, "mesh.hdf5"), mesh)
        # hdf5min is written from the file-format specification and has never been opened by libhdf5 (neither it nor h5py exists on
        # this image): the same arrays also go out as `mesh.npz`, which load_gaussians prefers when both are present
        np.savez(os.path.join(path, "mesh.npz"), **mesh)


def load_gaussians(pc, path, device="cuda"):
    """MultiGaussianMesh.load_ply (gaussian_model.py:219-262 + gaussian_mesh.py:467-481) into a MeshGaussians"""
    import torch
    import torch.nn as nn
    d = read_ply(os.path.join(path, "point_cloud.ply"))
    P = d["x"].shape[0]
    col = lambda prefix: np.stack([d[n] for n in sorted((k for k in d if k.startswith(prefix)), key=lambda s: int(s.split("_")[-1]))], 1)  # noqa: E731
    f_dc = np.stack([d["f_dc_0"], d["f_dc_1"], d["f_dc_2"]], 1).reshape(P, 3, 1)
    rest = col("f_rest_")
    assert rest.shape[1] == 3 * (pc.max_sh_degree + 1) ** 2 - 3
    rest = rest.reshape(P, 3, (pc.max_sh_degree + 1) ** 2 - 1)
    par = lambda a: nn.Parameter(torch.tensor(np.ascontiguousarray(a), dtype=torch.float, device=device).requires_grad_(True))  # noqa: E731
    pc._features_dc = nn.Parameter(torch.tensor(f_dc, dtype=torch.float, device=device).transpose(1, 2).contiguous().requires_grad_(True))
    pc._features_rest = nn.Parameter(torch.tensor(rest, dtype=torch.float, device=device).transpose(1, 2).contiguous().requires_grad_(True))
    pc._opacity = par(d["opacity"][:, None])
    pc._scaling, pc._rotation = par(col("scale_")), par(col("rot_"))
    pc.active_sh_degree = pc.max_sh_degree
    pc.face_ids = torch.tensor(d["id"], dtype=torch.long, device=device)
    pc.face_bary = par(np.stack([d["b1"], d["b2"], d["b3"]], 1))
    pc.face_offset = par(d["o"][:, None])
    if os.path.exists(os.path.join(path, "mesh.npz")):      # written by this library (next to mesh.hdf5 when h5py is absent)
        m = np.load(os.path.join(path, "mesh.npz"))
        mesh = {k: torch.tensor(m[k], device=device) for k in m.files}
    else:                                                    # a directory saved by the reference: h5py's file
        from . import hdf5min
        mesh = {k: torch.tensor(v, device=device) for k, v in hdf5min.load(os.path.join(path, "mesh.hdf5")).items()}
    for k, v in mesh.items():
        setattr(pc.mesh, k, v)
    if getattr(pc.mesh, "edge_index", None) is not None and getattr(pc.mesh, "pos", None) is not None:
        ei = pc.mesh.edge_index
        pc.edge_norm = torch.linalg.norm(pc.mesh.pos[ei[1]] - pc.mesh.pos[ei[0]], dim=-1, keepdim=True)
    if hasattr(pc, "invalidate_caches"):
        pc.invalidate_caches()
    return pc
