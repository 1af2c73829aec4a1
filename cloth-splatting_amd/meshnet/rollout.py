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


@torch.no_grad()
def rollout(simulator, positions, velocity_history, node_type, edge_index, actions, grasped_particle, nsteps):
    """positions [N,3] (updated in place on a copy), velocity_history [H,N,3], actions [nsteps,3] (velocity of the grasped
    node), grasped_particle: index of the pinned node.  -> (predicted velocities [nsteps,N,3], final positions [N,3])"""
    pos = positions.clone()
    hist = velocity_history.clone()
    H = hist.shape[0]
    preds = []
    for step in range(nsteps):
        ef = edge_features(pos, edge_index)
        vel = torch.cat([hist[h] for h in range(H)], 1)                       # [N, 3H], oldest first (:145)
        v_next = simulator.predict_velocity(velocities=vel, node_type=node_type, edge_index=edge_index, edge_features=ef)
        v_next[grasped_particle] = actions[step]
        preds.append(v_next)
        pos += v_next
        if H > 1:
            hist[:H - 1] = hist[1:].clone()
        hist[-1] = v_next
    return torch.stack(preds), pos
