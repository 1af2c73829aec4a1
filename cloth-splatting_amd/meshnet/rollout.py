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


def refine_edge_lengths(pos, v, edge_index, rest_len, grasped_particle=None, iters=10, lr=1e-3, edge_w=None):
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
        w = edge_w          # (a rollout builds the weights once for all its steps)
        if w is None and grasped_particle is not None and E > 0:
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


import os as _os

# the rollout loop recorded: one eager step (weight images packed, caches filled), one step recorded into a hipGraph, the rest replays.
# CSPLAT_ROLLOUT_GRAPH=0 keeps the launch-by-launch loop.
ROLLOUT_GRAPH = _os.environ.get("CSPLAT_ROLLOUT_GRAPH", "1") not in ("", "0")
ROLLOUT_STATS = {"eager_steps": 0, "replayed_steps": 0, "recorded": 0, "record_failed": 0, "repeated_bf16": 0}


def _step_body(simulator, pos, hist, act, node_type, edge_index, grasped_particle, real_world, original_edge_lengths, edge_w, out):
    """one rollout step on the state tensors `pos` [N,3] / `hist` [H,N,3] (both updated IN PLACE), grasped node velocity `act` [3];
    the predicted velocity is written to `out` [N,3]"""
    H = hist.shape[0]
    ef = edge_features(pos, edge_index)
    vel = torch.cat([hist[h] for h in range(H)], 1)                       # [N, 3H], oldest first (:145)
    v_next = simulator.predict_velocity(velocities=vel, node_type=node_type, edge_index=edge_index, edge_features=ef)
    if real_world:
        v_next = refine_edge_lengths(pos, v_next, edge_index, original_edge_lengths, grasped_particle, edge_w=edge_w)
    v_next[grasped_particle] = act
    out.copy_(v_next)
    pos += v_next
    if H > 1:
        hist[:H - 1] = hist[1:].clone()
    hist[-1] = v_next


# recorded steps are kept across rollout() calls (recording costs ~40 ms: a collection pass, the capture, the instantiation -- two steps'
# worth of a 20-step rollout): keyed on everything the recording's raw pointers and baked-in host decisions depend on
_ROLLOUT_CACHE = {}
_ROLLOUT_CACHE_MAX = 2


def _rollout_key(simulator, positions, velocity_history, node_type, edge_index, actions, grasped_particle, real_world):
    from .graph_ops import edge_mlp3_mode
    if not isinstance(grasped_particle, int):
        return None                 # (an index tensor / array: its values are baked into the recording -- not cached)
    parts = [(int(t.data_ptr()), int(t._version)) for t in simulator.state_dict(keep_vars=True).values() if torch.is_tensor(t)]
    for name in ("_output_normalizer", "_node_normalizer"):
        nz = getattr(simulator, name, None)
        if nz is not None:
            for k, v in sorted(vars(nz).items()):
                if torch.is_tensor(v):
                    parts.append((k, int(v.data_ptr()), int(v._version)))
                elif isinstance(v, (int, float, bool, str)) or v is None:
                    parts.append((k, v))
    epd = getattr(simulator, "_encode_process_decode", None)
    return (id(simulator), tuple(parts), tuple(positions.shape), positions.dtype, tuple(velocity_history.shape), tuple(actions.shape[1:]),
            int(node_type.data_ptr()), int(node_type._version), tuple(node_type.shape), int(edge_index.data_ptr()), int(edge_index._version),
            tuple(edge_index.shape), grasped_particle, bool(real_world), edge_mlp3_mode(), bool(getattr(epd, "_bf16_latched", False)),
            bool(simulator.training), _n.SCRATCH_EPOCH[0], positions.device.index)


@torch.no_grad()
def rollout(simulator, positions, velocity_history, node_type, edge_index, actions, grasped_particle, nsteps, real_world=False,
            original_edge_lengths=None, graph=None):
    """positions [N,3] (updated in place on a copy), velocity_history [H,N,3], actions [nsteps,3] (velocity of the grasped
    node), grasped_particle: index of the pinned node.  -> (predicted velocities [nsteps,N,3], final positions [N,3]).
    real_world=True (train_meshnet_sim.py:211-250, as meshnet/generate_rw_predictions.py:152 calls it): every predicted velocity is
    refined by ten Adam iterations against the deviation of the edge lengths from `original_edge_lengths` (default: the lengths of the
    initial positions, :114-116) before the grasped node is pinned.
    graph (default: ROLLOUT_GRAPH on the GPU, nsteps >= 4): the first rollout of a (simulator weights, graph, shapes) combination runs
    step 0 launch by launch and RECORDS step 1 into a hipGraph on state buffers (positions, history, the step's action, the step's output
    row); every later step -- and every step of later rollouts with the same combination -- is one replay: the same launches in the same
    order on the same buffers, so the result equals the eager loop bit for bit; the host issues one graph launch and two small copies
    per step instead of ~35 launches."""
    from .graph_network import deferred_overflow_check
    from .graph_ops import GraphCSR
    if real_world and original_edge_lengths is None:
        original_edge_lengths = torch.norm(positions[edge_index[1]] - positions[edge_index[0]], dim=1)
    dev = positions.device
    use_graph = (ROLLOUT_GRAPH if graph is None else bool(graph)) and positions.is_cuda and nsteps >= 4
    N = positions.shape[0]
    for attempt in range(2):
        key = _rollout_key(simulator, positions, velocity_history, node_type, edge_index, actions, grasped_particle, real_world) if use_graph else None
        ent = _ROLLOUT_CACHE.get(key) if key is not None else None
        preds = torch.empty(nsteps, N, positions.shape[1], dtype=positions.dtype, device=dev)
        if ent is not None:
            # ---- every step a replay of the recording made by an earlier rollout
            ent["pos"].copy_(positions); ent["hist"].copy_(velocity_history); ent["fine"].fill_(True)
            if real_world:
                ent["L0"].copy_(original_edge_lengths)
            for step in range(nsteps):
                ent["act"].copy_(actions[step])
                ent["graph"].replay()
                preds[step].copy_(ent["out"])
            ROLLOUT_STATS["replayed_steps"] += nsteps
            if attempt == 0 and ent["mods"] and not bool(ent["fine"]):
                for m in ent["mods"]:
                    m.latch_bf16()
                ROLLOUT_STATS["repeated_bf16"] += 1
                continue
            return preds, ent["pos"].clone()
        pos = positions.clone()
        hist = velocity_history.clone()
        act = torch.empty(actions.shape[1:], dtype=actions.dtype, device=dev) if nsteps else None
        out = torch.empty_like(pos)
        L0 = original_edge_lengths.to(device=dev, dtype=torch.float32).clone() if real_world else None
        edge_w = None
        if real_world and positions.is_cuda and int(edge_index.shape[1]) > 0:
            edge_w = torch.ones(int(edge_index.shape[1]), dtype=torch.float32, device=dev)
            edge_w[grasped_particle] = 0.0          # (`length_deviation[grasped_particle] *= 0`, train_meshnet_sim.py:234)
        # (the fp16-piece arithmetic of the one-launch kernels reports an overflow as a device word per predict_velocity call: collected
        #  here, read ONCE behind the loop -- no host read per step)
        with deferred_overflow_check() as chk:
            g = None
            for step in range(nsteps):
                act.copy_(actions[step])
                if g is not None:
                    g.replay()
                    ROLLOUT_STATS["replayed_steps"] += 1
                elif use_graph and step == 1:
                    try:
                        from csplat.graphs import capture
                        n_before = len(chk.items)
                        fine_all = torch.ones((), dtype=torch.bool, device=dev)      # (set here, OUTSIDE the recording: replays only AND into it)
                        torch.cuda.synchronize(dev)
                        g = torch.cuda.CUDAGraph()
                        with capture(g):
                            _step_body(simulator, pos, hist, act, node_type, edge_index, grasped_particle, real_world, L0, edge_w, out)
                            # the replays overwrite the step's overflow word: AND it into one that lives across the replays
                            for _m, ok in chk.items[n_before:]:
                                fine_all.logical_and_(ok)
                        mods = list({id(m): m for m, _ok in chk.items[n_before:]}.values())
                        del chk.items[n_before:]
                        chk.items.extend((m, fine_all) for m in mods[:1])
                        ROLLOUT_STATS["recorded"] += 1
                        g.replay()          # (the capture itself executed nothing)
                        ROLLOUT_STATS["replayed_steps"] += 1
                        if key is not None:
                            while len(_ROLLOUT_CACHE) >= _ROLLOUT_CACHE_MAX:
                                _ROLLOUT_CACHE.pop(next(iter(_ROLLOUT_CACHE)))
                            _ROLLOUT_CACHE[key] = {"graph": g, "pos": pos, "hist": hist, "act": act, "out": out, "L0": L0, "edge_w": edge_w,
                                                   "fine": fine_all, "mods": mods,
                                                   # strong references: whatever the recording's raw pointers point into stays alive (and its
                                                   # addresses un-reusable) as long as the recording can be replayed
                                                   "keep": (simulator, edge_index, node_type, GraphCSR.get(edge_index, N))}
                    except Exception:
                        # (a capture that failed half way leaves the state buffers as they were: nothing of a capture executes)
                        g, use_graph = None, False
                        ROLLOUT_STATS["record_failed"] += 1
                        torch.cuda.synchronize(dev)
                        _step_body(simulator, pos, hist, act, node_type, edge_index, grasped_particle, real_world, L0, edge_w, out)
                        ROLLOUT_STATS["eager_steps"] += 1
                else:
                    _step_body(simulator, pos, hist, act, node_type, edge_index, grasped_particle, real_world, L0, edge_w, out)
                    ROLLOUT_STATS["eager_steps"] += 1
                preds[step].copy_(out)
            if attempt == 0 and chk.overflowed():
                ROLLOUT_STATS["repeated_bf16"] += 1
                _ROLLOUT_CACHE.pop(key, None)           # (recorded with the fp16 pieces: not to be replayed for the latched module)
                continue        # a step left fp16's range: the modules concerned now run with bf16 pieces -- the rollout again, from its start
        # (the state buffers belong to the cached recording from here on: hand back a copy)
        return preds, (pos.clone() if (key is not None and key in _ROLLOUT_CACHE) else pos)
    raise AssertionError("unreachable")
