"""The hot loop of the reference's rollout (/root/reference/train_meshnet_sim.py:92-265, the part between the data set and the
bookkeeping): per step

    graph features from the CURRENT node positions   (:147-152: `_data_to_graph` + FaceToEdge / Cartesian / Distance; the
                                                      connectivity of a fixed triangulation does not change, the features do)
    v_next = simulator.predict_velocity(cat(velocity history), node_type, edge_index, edge_features)     (:156-160)
    v_next[grasped] = action[step]                                                                        (:176)
    positions += v_next;  history <- (history[1:], v_next)                                                (:256-262)

with the per-step edge features from one HIP kernel (csplat_gnn_edge_features) instead of two PyG transforms.
Returns the predicted velocities [nsteps, N, 3] like the reference's `predictions`."""
import torch

from csplat import native as _n


def edge_features(pos, edge_index):
    """[E, 4] = (pos[row] - pos[col], norm), row / col = edge_index[0] / [1] (PyG Cartesian + Distance, norm=False)"""
    if pos.is_cuda and pos.dtype == torch.float32 and edge_index.dtype == torch.int64:
        pos, edge_index = pos.contiguous(), edge_index.contiguous()
        E = int(edge_index.shape[1])
        out = torch.empty(E, 4, dtype=torch.float32, device=pos.device)
        with _n.on_device(pos.device):
            _n.check(_n.lib.csplat_gnn_edge_features(_n.stream_handle(pos.device), E, _n.ptr(pos), _n.ptr(edge_index), _n.ptr(out)),
                     "csplat_gnn_edge_features")
        return out
    _n.composed_fallback("rollout.edge_features", "dtype", pos)
    d = pos[edge_index[0]] - pos[edge_index[1]]
    return torch.cat([d, d.norm(dim=1, keepdim=True)], 1)


def refine_edge_lengths(pos, v, edge_index, rest_len, grasped_particle=None, iters=10, lr=1e-3):
    """The `real_world` branch of the reference's rollout (/root/reference/train_meshnet_sim.py:211-250), per rollout step: `iters`
    iterations of a FRESH torch.optim.Adam(lr) on the predicted velocities v [N,3] against
        sum_e (|(pos + v)[edge_index[0][e]] - (pos + v)[edge_index[1][e]]| - rest_len[e])^2
    with the one entry zeroed that the reference zeroes (`length_deviation[grasped_particle] *= 0` indexes the EDGE array with the node
    index -- kept as is).  Returns the refined velocities (a new tensor).  On the GPU: csplat_gnn_edge_length_refine, one launch per
    iteration, no autograd graph, no atomics; elsewhere the reference's own torch formulation (reported as a composed fallback)."""
    E = int(edge_index.shape[1])
    if v.is_cuda and v.dtype == torch.float32 and pos.dtype == torch.float32 and edge_index.dtype == torch.int64:
        from .graph_ops import GraphCSR
        dev = v.device
        N = int(v.shape[0])
        csr = GraphCSR.get(edge_index, N)
        out = v.detach().clone().contiguous()
        w = None
        if grasped_particle is not None and E > 0:
            w = torch.ones(E, dtype=torch.float32, device=dev)
            w[grasped_particle] = 0.0
        scratch = torch.empty(9 * max(N, 1), dtype=torch.float32, device=dev)
        rl = rest_len.to(device=dev, dtype=torch.float32).contiguous()
        p_ = pos.contiguous()
        with _n.on_device(dev):
            _n.check(_n.lib.csplat_gnn_edge_length_refine(
                _n.stream_handle(dev), N, E, _n.ptr(p_), _n.ptr(out), _n.ptr(csr.ei), _n.ptr(rl), _n.ptr(w), _n.ptr(csr.rowptr["dst"]),
                _n.ptr(csr.perm["dst"]), _n.ptr(csr.rowptr["src"]), _n.ptr(csr.perm["src"]), int(iters), float(lr), 0.9, 0.999, 1e-8,
                _n.ptr(scratch)), "csplat_gnn_edge_length_refine")
        return out
    _n.composed_fallback("rollout.refine_edge_lengths", "dtype", v)
    with torch.enable_grad():
        vo = v.detach().clone().requires_grad_(True)
        opt = torch.optim.Adam([vo], lr=lr)
        for _ in range(iters):
            opt.zero_grad()
            x = pos.detach() + vo
            d = x[edge_index[0]] - x[edge_index[1]]
            dev_ = torch.norm(d, dim=1) - rest_len
            if grasped_particle is not None:
                keep = torch.ones_like(dev_)
                keep[grasped_particle] = 0
                dev_ = dev_ * keep
            torch.sum(dev_ ** 2).backward()
            opt.step()
    return vo.detach()


@torch.no_grad()
def rollout(simulator, positions, velocity_history, node_type, edge_index, actions, grasped_particle, nsteps, real_world=False,
            original_edge_lengths=None):
    """positions [N,3] (updated in place on a copy), velocity_history [H,N,3], actions [nsteps,3] (velocity of the grasped
    node), grasped_particle: index of the pinned node.  -> (predicted velocities [nsteps,N,3], final positions [N,3]).
    real_world=True (train_meshnet_sim.py:211-250, as meshnet/generate_rw_predictions.py:152 calls it): every predicted velocity is
    refined by ten Adam iterations against the deviation of the edge lengths from `original_edge_lengths` (default: the lengths of the
    initial positions, :114-116) before the grasped node is pinned."""
    pos = positions.clone()
    if real_world and original_edge_lengths is None:
        original_edge_lengths = torch.norm(positions[edge_index[1]] - positions[edge_index[0]], dim=1)
    hist = velocity_history.clone()
    H = hist.shape[0]
    preds = []
    for step in range(nsteps):
        ef = edge_features(pos, edge_index)
        vel = torch.cat([hist[h] for h in range(H)], 1)                       # [N, 3H], oldest first (:145)
        v_next = simulator.predict_velocity(velocities=vel, node_type=node_type, edge_index=edge_index, edge_features=ef)
        if real_world:
            v_next = refine_edge_lengths(pos, v_next, edge_index, original_edge_lengths, grasped_particle)
        v_next[grasped_particle] = actions[step]
        preds.append(v_next)
        pos += v_next
        if H > 1:
            hist[:H - 1] = hist[1:].clone()
        hist[-1] = v_next
    return torch.stack(preds), pos
