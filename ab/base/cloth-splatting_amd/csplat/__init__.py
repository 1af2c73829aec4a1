"""csplat -- native (HIP, gfx950) core of the cloth-splatting hot path.

This directory's parent (`cloth-splatting_amd/`) is a DROP-IN ROOT: put it first on sys.path and the
reference's own imports resolve to the MI355X implementation:

    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from simple_knn._C import distCUDA2
    from meshnet.graph_network import EncodeProcessDecode
    from gaussian_renderer import render

There is no CPU or PyTorch fallback anywhere on the product path: if libcsplat.so is missing the import
of `csplat.native` raises.
"""
from . import native  # noqa: F401  (raises if the HIP library is absent)

__all__ = ["native"]
