"""Generates tests/golden/mesh_h5py.hdf5 (+ .npz with the same arrays) with the REAL h5py, using the reference's call pattern for its
`mesh.hdf5` side-car (/root/reference/scene_reconstruction/gaussian_mesh.py:462-465: `h5py.File(path, "w")`, one
`create_dataset(key, data=value)` per mesh attribute).  Run with an interpreter that has h5py -- on this image
    /opt/conda/bin/python3.9 tests/golden/make_h5py_fixture.py
(h5py 3.3.0 on libhdf5 1.10.6; the build's own Python 3.10 has none).  The fixture is data: four small arrays in libhdf5's bytes."""
import os
import sys

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def main(out_h5=os.path.join(HERE, "mesh_h5py.hdf5"), out_npz=os.path.join(HERE, "mesh_h5py.npz")):
    rng = np.random.default_rng(12)
    V, F, E = 25, 32, 80
    mesh = {"pos": rng.normal(size=(V, 3)).astype(np.float32), "norm": rng.normal(size=(V, 3)).astype(np.float32),
            "face": rng.integers(0, V, (3, F)).astype(np.int64), "edge_index": rng.integers(0, V, (2, E)).astype(np.int64)}
    with h5py.File(out_h5, "w") as f:
        for key, value in mesh.items():
            f.create_dataset(key, data=value)
    np.savez(out_npz, **mesh)
    print("h5py", h5py.__version__, "libhdf5", h5py.version.hdf5_version, "->", out_h5)


if __name__ == "__main__":
    main(*sys.argv[1:3])
