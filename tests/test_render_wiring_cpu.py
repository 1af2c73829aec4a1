"""The build's render() fed with the REFERENCE's own MultiGaussianMesh / Camera objects (INTEGRATION.md's promise) -- possible only
where /root/reference exists, i.e. in the build container; skipped on the GPU box.  The body runs in a fresh interpreter
(tests/render_wiring_dropin.py) because it installs stand-in modules for packages this image lacks."""
import os
import subprocess
import sys

import numpy as np
import pytest

import util


def test_fixture_holds_the_references_wiring_facts():
    """facts of gaussian_renderer/__init__.py:49-164 the fixture must show (guards the generator, runs everywhere)"""
    G = util.golden("render_wiring.npz")
    fovx, fovy = float(G["in.cam_FoVx"]), float(G["in.cam_FoVy"])
    assert abs(float(G["default.settings.tanfovx"]) - np.tan(fovx * 0.5)) < 1e-12           # tan(FoV * 0.5), :56-57
    assert abs(float(G["default.settings.tanfovy"]) - np.tan(fovy * 0.5)) < 1e-12
    assert int(G["default.settings.sh_degree"]) == 2                                         # pc.active_sh_degree, not max (:70)
    assert float(G["scale_mod.settings.scale_modifier"]) == 1.7 and float(G["default.settings.scale_modifier"]) == 1.0
    assert bool(G["default.arg.colors_precomp.none"]) and not bool(G["default.arg.shs.none"])
    assert bool(G["override_color.arg.shs.none"]) and not bool(G["override_color.arg.colors_precomp.none"])     # :146-147
    np.testing.assert_array_equal(G["override_color.arg.colors_precomp"], G["in.override_color"])
    # python-covariance branch: scales None, cov3D given -- and rotations STILL passed (:83-88,156-164)
    assert bool(G["cov_python.arg.scales.none"]) and not bool(G["cov_python.arg.cov3D_precomp.none"]) and \
        not bool(G["cov_python.arg.rotations.none"])
    assert int(G["static.sim_calls"]) == 0 and int(G["default.sim_calls"]) == 1               # render_static skips the simulator
    np.testing.assert_array_equal(G["static.res.vertice_deform"], G["in.pos"])
    V = G["in.pos"].shape[0]
    assert G["default.sim_time_vector"].shape == (V, 1) and np.all(G["default.sim_time_vector"] == np.float32(G["in.cam_time"]))
    assert bool(G["default.res.vertice_projections.none"]) and not bool(G["project_vertices.res.vertice_projections.none"])
    assert all(bool(G[f"default.res.{f}.none"]) for f in ("shadows", "shadows_mean", "shadows_std"))
    assert float(np.abs(G["default.arg.means2D"]).max()) == 0.0


@pytest.mark.skipif(not os.path.isdir("/root/reference/scene_reconstruction"), reason="needs /root/reference (build container only)")
def test_build_render_accepts_reference_objects():
    r = subprocess.run([sys.executable, os.path.join(util.ROOT, "tests", "render_wiring_dropin.py")], capture_output=True, text=True,
                       timeout=600, env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "render wiring drop-in ok" in r.stdout
