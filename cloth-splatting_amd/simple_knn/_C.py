"""Drop-in for `simple_knn._C` (scene_reconstruction/gaussian_mesh.py:26,250; gaussian_model.py:20,134)."""
import torch

from csplat import native as _n


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    """points [P,3] float32 on the GPU -> [P] mean squared distance to the 3 nearest other points."""
    _n.require_cuda(points)
    pts = points.detach().to(torch.float32).contiguous()
    P = int(pts.shape[0])
    out = torch.empty(P, dtype=torch.float32, device=pts.device)
    with _n.on_device(pts.device):
        if P >= BOXED_FROM:   # Morton order + box pruning (what the upstream extension does); same bits, O(P) candidates
            temp = torch.empty(int(_n.lib.csplat_dist2_temp_bytes(P)), dtype=torch.uint8, device=pts.device)
            _n.check(_n.lib.csplat_dist2_ws(_n.stream_handle(pts.device), P, _n.ptr(pts), _n.ptr(out), _n.ptr(temp)), "csplat_dist2_ws")
        else:
            _n.check(_n.lib.csplat_dist2(_n.stream_handle(pts.device), P, _n.ptr(pts), _n.ptr(out)), "csplat_dist2")
    return out


BOXED_FROM = 4096   # below this the single brute-force kernel is faster than sort + boxes
