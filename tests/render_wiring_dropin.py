"""Run by tests/test_render_wiring_cpu.py in a FRESH interpreter, only where /root/reference exists (the build container).

INTEGRATION.md promises that the build's gaussian_renderer.render() accepts the reference's OWN objects.  Here the reference's
`MultiGaussianMesh` (scene_reconstruction/gaussian_mesh.py) and `Camera` (scene_reconstruction/cameras.py) are imported from
/root/reference (under the stand-ins tests/golden/make_golden.py documents: PyG / h5py / plyfile shims, roma served by scipy),
built from the seeded arrays of render_wiring.npz, and handed to the BUILD's render() with a recording rasterizer in place of the
HIP one (CPU tensors: this is a test of the host-side wiring, not of the kernels).  What the build hands the rasterizer must equal
what the reference's render() handed it (the fixture), field by field.  Test infrastructure, not product code."""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(ROOT, "cloth-splatting_amd"))
import gaussian_renderer as build_gr  # noqa: E402  (the BUILD's module, resolved before the reference root is put on sys.path)
import meshnet  # noqa: E402,F401
assert build_gr.__file__.startswith(ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden as mg  # noqa: E402  (puts /root/reference first on sys.path)

mg.install_mesh_shims()
mg.install_roma_scipy()
G = np.load(os.path.join(ROOT, "tests", "golden", "render_wiring.npz"))


def close(a, b, tol=1e-6):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return a.shape == b.shape and float(np.abs(a - b).max()) <= tol * (float(np.abs(b).max()) + 1e-30)


def same_quats(a, b, tol):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a * np.sign((a * b).sum(1, keepdims=True)) - b).max()) <= tol


with mg.cuda_calls_as_cpu():
    from scene_reconstruction.cameras import Camera
    from scene_reconstruction.gaussian_mesh import MultiGaussianMesh
    assert MultiGaussianMesh.__module__ == "scene_reconstruction.gaussian_mesh" and "/root/reference" in sys.modules[MultiGaussianMesh.__module__].__file__
    d = mg.render_wiring_scene()
    build_gr.GaussianRasterizer = mg.RecordingRasterizer          # the CPU stand-in for libcsplat's rasterizer
    for name, kw in mg.WIRING_CASES.items():
        pc, cam, sim = mg.render_wiring_objects(d, MultiGaussianMesh, Camera)
        kw = dict(kw)
        pipe = types.SimpleNamespace(compute_cov3D_python=bool(kw.pop("pipe_cov", False)), convert_SHs_python=False, debug=False)
        if "override_color" in kw:
            kw["override_color"] = d["override_color"]
        mg.RecordingRasterizer.calls.clear()
        res = build_gr.render(cam, pc, sim, pipe, d["bg"], **kw)
        assert len(mg.RecordingRasterizer.calls) == 1, name
        rs, args = mg.RecordingRasterizer.calls[0]
        g = lambda k: G[f"{name}.{k}"]  # noqa: E731
        for f in rs._fields:
            v = getattr(rs, f)
            v = v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)
            assert close(v, g(f"settings.{f}"), 1e-12), (name, f, v, g(f"settings.{f}"))
        for k, v in args.items():
            assert (v is None) == bool(g(f"arg.{k}.none")), (name, k)
            if v is None:
                continue
            ok = same_quats(v.detach().numpy(), g(f"arg.{k}"), 5e-6) if k == "rotations" else close(v.detach().numpy(), g(f"arg.{k}"), 2e-6)
            assert ok, (name, k)
        assert len(sim.seen) == int(g("sim_calls")) and (not sim.seen or np.array_equal(sim.seen[0].numpy(), g("sim_time_vector"))), name
        for f in res._fields:
            v = getattr(res, f)
            assert (v is None) == bool(g(f"res.{f}.none")), (name, f)
            if v is None or f in ("viewspace_points",):
                continue
            v = v.detach().numpy()
            if f == "rotations":
                assert same_quats(v, g("res.rotations"), 5e-6), name
            elif f in ("projections", "vertice_projections"):
                assert float(np.abs(v - g(f"res.{f}")).max()) < 2e-3, (name, f)
            else:
                assert close(v, g(f"res.{f}"), 2e-6), (name, f)
        assert res.viewspace_points.requires_grad and float(res.viewspace_points.abs().max()) == 0.0
print("render wiring drop-in ok:", ", ".join(mg.WIRING_CASES))
