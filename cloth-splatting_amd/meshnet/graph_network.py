"""Drop-in for /root/reference/meshnet/graph_network.py with the torch_geometric dependency replaced by the
csplat HIP kernels.  Class names, constructor / forward signatures and state_dict keys are the reference's
(`_encoder.node_fn.0.NN-0.weight`, `_processor.gnn_stacks.K.edge_fn.0.NN-0.weight`, ...), so `model-N.pt`
checkpoints load unchanged (cloth_network.py:242-243).

Semantics reproduced from PyG `MessagePassing(aggr='add')`:
  x_j = x[edge_index[0]], x_i = x[edge_index[1]];  message = LN(MLP(cat[x_i, x_j, e]))   (graph_network.py:178-199)
  aggregate = sum over edge_index[1], dim_size = N                                    (graph_network.py:136)
  update gets the ORIGINAL edge features -> every layer returns edge_out = 2 * edge_in  (graph_network.py:173-176,222)

MI355X design: the [E, 3L] concat is never built.  W1 of the edge MLP is applied as three column blocks:
x @ W_i^T and x @ W_j^T at node level (N rows), e @ W_e^T at edge level; csplat_gnn_edge_combine_fwd gathers and
adds them (+ReLU) in one HBM pass.  The scatter-add is csplat_gnn_segment_sum over a CSR-by-destination order
(deterministic).  The node MLP's cat[agg, x] is likewise split.  Node-level y and dx GEMMs stay on rocBLAS; every 128 x 128
weight gradient (edge AND node level) is csplat_dw128.
"""
import os
from typing import List

import torch
import torch.nn as nn

from .graph_ops import (EdgeCombine, EdgeFirstLayer, GraphCSR, SegmentSum, absmax, rows_chain, rows_chain_pack, edge_latent_linear, edge_mlp3, edge_mlp3_mode, edge_mlp3_pack, edge_tail_aggregate, gather_rows, mlp3_rows, node_update_pack, node_update_packed, segment_sum_rows, edge_tail_ok, linear_narrow128,
                        layer_norm_rows, report_missed_edge_tail, linear128, linear_rows, node_update)


def build_mlp(input_size: int, hidden_layer_sizes: List[int], output_size: int = None,
              output_activation: nn.Module = nn.Identity, activation: nn.Module = nn.ReLU) -> nn.Module:
    """graph_network.py:7-45 -- Sequential of Linear "NN-i" / activation "Act-i"."""
    layer_sizes = [input_size] + hidden_layer_sizes
    if output_size:
        layer_sizes.append(output_size)
    nlayers = len(layer_sizes) - 1
    act = [activation for _ in range(nlayers)]
    act[-1] = output_activation
    mlp = nn.Sequential()
    for i in range(nlayers):
        mlp.add_module("NN-" + str(i), nn.Linear(layer_sizes[i], layer_sizes[i + 1]))
        mlp.add_module("Act-" + str(i), act[i]())
    return mlp


class Encoder(nn.Module):
    """graph_network.py:48-111"""

    def __init__(self, nnode_in_features: int, nnode_out_features: int, nedge_in_features: int, nedge_out_features: int,
                 nmlp_layers: int, mlp_hidden_dim: int):
        super().__init__()
        self.node_fn = nn.Sequential(*[build_mlp(nnode_in_features, [mlp_hidden_dim for _ in range(nmlp_layers)],
                                                 nnode_out_features), nn.LayerNorm(nnode_out_features)])
        self.edge_fn = nn.Sequential(*[build_mlp(nedge_in_features, [mlp_hidden_dim for _ in range(nmlp_layers)],
                                                 nedge_out_features), nn.LayerNorm(nedge_out_features)])

    def forward(self, x: torch.Tensor, edge_features: torch.Tensor):
        if not torch.is_grad_enabled():
            # rollout: both MLPs on the HIP kernels -- the narrow first Linear in one write-paced launch, the 128-wide layers and the
            # LayerNorm as csplat_linear128 epilogues (stock kernels took 0.86 of the 4.2 ms rollout step here: profiles/r05b)
            xe, ee = _encode_inference(self.node_fn, x), _encode_inference(self.edge_fn, edge_features)
            if xe is not None and ee is not None:
                return xe, ee
        e = _tail(self.edge_fn[0], edge_features, first_has_act=False, skip_first=False)
        return self.node_fn(x), layer_norm_rows(e, self.edge_fn[1])


def _tail(seq_mlp: nn.Sequential, h: torch.Tensor, first_has_act: bool, skip_first: bool = True) -> torch.Tensor:
    """run build_mlp's layers after the first Linear (whose activation has already been applied iff first_has_act);
    skip_first=False runs the whole MLP."""
    mods = list(seq_mlp.children())
    start = (2 if first_has_act else 1) if skip_first else 0  # skip NN-0 (+ Act-0 when it was fused)
    mods = mods[start:]
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.Linear):   # tall inputs: csplat_linear128 both ways (ReLU fused), split-K weight gradient
            fuse = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
            h = linear_rows(h, m.weight, m.bias, relu=fuse)
            i += 2 if fuse else 1
        else:
            h = m(h)
            i += 1
    return h


def _fusable(seq: nn.Sequential, width: int = 128) -> bool:
    """[build_mlp(...), LayerNorm] whose layers after the first are 128 -> 128 Linear + ReLU (Identity last)."""
    mlp, ln = seq[0], seq[1]
    mods = list(mlp.children())
    lins, acts = mods[0::2], mods[1::2]
    if not (isinstance(ln, nn.LayerNorm) and tuple(ln.normalized_shape) == (width,) and ln.elementwise_affine):
        return False
    if len(lins) < 2 or not all(isinstance(m, nn.Linear) and m.out_features == width for m in lins):
        return False
    if not all(m.in_features == width for m in lins[1:]):
        return False
    return all(isinstance(a, nn.ReLU) for a in acts[:-1]) and isinstance(acts[-1], nn.Identity) and \
        lins[0].weight.dtype == torch.float32 and lins[0].weight.is_cuda


def _fused_tail(seq: nn.Sequential, h: torch.Tensor, add_post: torch.Tensor = None) -> torch.Tensor:
    """layers 1.. of a _fusable MLP + its LayerNorm (+ a residual), in place on h (h = activated output of layer 0)."""
    lins = list(seq[0].children())[0::2]
    for i, lin in enumerate(lins[1:], start=1):
        last = i == len(lins) - 1
        h = linear128(h, lin.weight, lin.bias, relu=not last, layer_norm=seq[1] if last else None,
                      add_post=add_post if last else None, out=h)
    return h


def _encode_inference(seq: nn.Sequential, x: torch.Tensor, x_absmax=None):
    """[build_mlp(K -> 128 -> ... -> 128), LayerNorm(128)] of an encoder under no_grad, or None when the shapes are not these"""
    lins = list(seq[0].children())[0::2]
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and 1 <= x.shape[1] <= 32 and len(lins) >= 2 and _fusable(seq)
            and tuple(lins[0].weight.shape) == (128, x.shape[1]) and lins[0].weight.dtype == torch.float32):
        return None
    if ENCODER_FUSED and len(lins) == 3 and x.shape[1] % 4 == 0 and x.shape[0] > 0 and edge_mlp3_mode() == 0:
        # the whole MLP + LayerNorm in ONE launch (csplat_gnn_mlp3_rows: the one-launch edge MLP's kernel on narrow rows, no gathers); the
        # first weight zero-padded to 128 columns, image packed once per weight version
        key = tuple((l.weight._version, l.weight.data_ptr()) for l in lins)
        if getattr(seq, "_mlp3_key", None) != key:
            with torch.no_grad():
                w0 = torch.zeros(128, 128, dtype=torch.float32, device=x.device)
                w0[:, :x.shape[1]] = lins[0].weight
                seq._mlp3_img = edge_mlp3_pack(w0, lins[1].weight, lins[2].weight)
            seq._mlp3_key = key
        return mlp3_rows(x, seq._mlp3_img, lins[0].bias, lins[1].bias, lins[2].bias, seq[1], x_absmax=x_absmax)
    h = linear_narrow128(x, lins[0].weight, lins[0].bias, relu=True)
    return _fused_tail(seq, h)


# rollout: an encoder's MLP + LayerNorm as ONE launch (csplat_gnn_mlp3_rows) instead of three; env CSPLAT_GNN_ENCODER_FUSED=0 goes back
ENCODER_FUSED = os.environ.get("CSPLAT_GNN_ENCODER_FUSED", "1") not in ("", "0")
# rollout: the edge MLP of a layer as ONE launch (csplat_gnn_edge_mlp3: weights resident in registers, the two inner [E,128] activations never in
# HBM) instead of three csplat_linear128 calls: 112-118 us against 235-245 per layer at E = 300k, rollout step 2.24 against 4.4-5.0 ms with
# everything below on (tools/ab_edge_mlp3_rollout.py).  env CSPLAT_GNN_EDGE_FUSED=0 goes back to the three launches.
EDGE_MLP_FUSED = os.environ.get("CSPLAT_GNN_EDGE_FUSED", "1") not in ("", "0")
# ... and that launch sums its messages per destination node itself (edges taken in destination order, per-run "pieces" instead of E message
# rows: include/csplat.h), the segmented sum then runs over ~E / 8 + N piece rows.  env CSPLAT_GNN_EDGE_AGG=0: messages out, segmented sum over E.
EDGE_AGG_FUSED = os.environ.get("CSPLAT_GNN_EDGE_AGG", "1") not in ("", "0")
# the node update on pre-packed 16-bit-piece weights (csplat_gnn_node_update_packed; fp16 or bf16 pieces by edge_mlp3_mode) instead of the
# exact-fp32 MFMA kernel of rounds 2-5; env CSPLAT_GNN_NODE_PACKED=0 goes back
NODE_UPDATE_PACKED = os.environ.get("CSPLAT_GNN_NODE_PACKED", "1") not in ("", "0")


# deferred_overflow_check(): a stack of lists collecting (module, device flag) per EncodeProcessDecode call instead of reading the flag
_OVERFLOW_COLLECT = []


class deferred_overflow_check:
    """`with deferred_overflow_check() as chk:` -- EncodeProcessDecode calls inside the scope do not read their overflow word (no host
    read per call); `chk.overflowed()` reads them all at once (ONE synchronisation) and latches the modules concerned to bf16 pieces."""

    def __enter__(self):
        self.items = []
        _OVERFLOW_COLLECT.append(self.items)
        return self

    def __exit__(self, *exc):
        _OVERFLOW_COLLECT.remove(self.items)
        return False

    def overflowed(self) -> bool:
        if not self.items:
            return False
        fine = torch.stack([b for _m, b in self.items]).cpu().tolist()
        mods = {id(m): m for (m, _b), f in zip(self.items, fine) if not f}
        for m in mods.values():
            m.latch_bf16()
        self.items.clear()
        return bool(mods)


def _is_pow2(v: float) -> bool:
    import math
    return v > 0 and math.frexp(v)[0] == 0.5


class InteractionNetwork(nn.Module):
    """graph_network.py:114-222 (PyG MessagePassing, aggr='add')."""

    def __init__(self, nnode_in: int, nnode_out: int, nedge_in: int, nedge_out: int, nmlp_layers: int, mlp_hidden_dim: int):
        super().__init__()
        self.aggr = 'add'
        self.node_fn = nn.Sequential(*[build_mlp(nnode_in + nedge_out, [mlp_hidden_dim for _ in range(nmlp_layers)],
                                                 nnode_out), nn.LayerNorm(nnode_out)])
        self.edge_fn = nn.Sequential(*[build_mlp(nnode_in + nnode_in + nedge_in, [mlp_hidden_dim for _ in range(nmlp_layers)],
                                                 nedge_out), nn.LayerNorm(nedge_out)])
        self._nnode_in, self._nedge_in = nnode_in, nedge_in

    def forward(self, x, edge_index, edge_features):
        # PyG hands update() the ORIGINAL edge features: edge output = input + input
        return self.message_update(x, edge_index, edge_features, 1.0)[0], edge_features + edge_features

    def message_update(self, x, edge_index, e_base, scale: float = 1.0):
        """the node half of forward() for edge features `scale * e_base`; returns (x_new, e_base to hand to the next layer).
        Because every layer only doubles its edge features (F7), Processor carries the encoder's edge latents e_base and the scalar
        2^l through the stack instead of materialising [E, L] sums per layer; the scale rides in the GEMM's alpha, and e_base is
        CHAINED through the layers (graph_ops.EdgeLatentLinear) so that its gradient is summed inside the layers' input-gradient
        GEMMs instead of by 14 separate [E, L] additions."""
        x_residual = x
        csr = GraphCSR.get(edge_index, x.shape[0])
        n = self._nnode_in
        # ---- message: LN(MLP(cat[x_i, x_j, e])) with the first Linear split into three column blocks; its bias rides on the x_i
        # block (N rows, and its gradient is a column sum over N rows instead of E)
        mlp_e = self.edge_fn[0]
        lin0 = mlp_e[0]
        relu0 = isinstance(mlp_e[1], nn.ReLU)
        W = lin0.weight
        xa = linear_rows(x, W[:, :n], lin0.bias)    # contribution of x_i = x[edge_index[1]]
        xb = linear_rows(x, W[:, n:2 * n], None)    # contribution of x_j = x[edge_index[0]]
        if relu0 and edge_tail_ok(e_base, self.edge_fn):
            # tall fp32 GPU rows: two autograd nodes for the whole message path -- the e-block GEMM with the gathers / ReLU in its
            # epilogue, then the rest of the MLP, the LayerNorm and the sum over destination nodes
            h, e_next = EdgeFirstLayer.apply(e_base, W[:, 2 * n:], scale, xa, xb, csr)
            agg = edge_tail_aggregate(h, csr, self.edge_fn, a0_relu=True)
        else:
            report_missed_edge_tail(e_base)
            ec, e_next = edge_latent_linear(e_base, W[:, 2 * n:], scale)
            h = EdgeCombine.apply(xa, xb, ec, csr, relu0)
            h = _tail(mlp_e, h, relu0)
            msg = layer_norm_rows(h, self.edge_fn[1])
            # ---- aggregate: sum over destination nodes
            agg = SegmentSum.apply(msg, csr)
        # ---- update: LN(MLP(cat[agg, x])), concat folded into two GEMMs
        mlp_n = self.node_fn[0]
        l0 = mlp_n[0]
        a = agg.shape[1]
        hn = linear_rows(agg, l0.weight[:, :a], l0.bias) + linear_rows(x, l0.weight[:, a:], None)
        hn = mlp_n[1](hn)
        hn = _tail(mlp_n, hn, True)
        x_updated = layer_norm_rows(hn, self.node_fn[1])
        return x_updated + x_residual, e_next

    def inference_ok(self, x, edge_features) -> bool:
        n = self._nnode_in
        # (the one-launch kernels gather node rows through 32-bit byte offsets: index * 512 < 2^32 -- graphs beyond 2^23 nodes take the
        #  composed path)
        return (not torch.is_grad_enabled()) and x.is_cuda and x.dtype == torch.float32 and \
            edge_features.dtype == torch.float32 and n == 128 and self._nedge_in == 128 and x.shape[0] < (1 << 23) and \
            _fusable(self.edge_fn) and _fusable(self.node_fn)

    def _split_weights(self):
        """contiguous column blocks of the two first-layer weights, re-cut only when an optimizer step changed them"""
        we, wn = self.edge_fn[0][0].weight, self.node_fn[0][0].weight
        key = (we._version, wn._version, we.data_ptr(), wn.data_ptr())
        if getattr(self, "_wsplit_key", None) != key:
            n, a = self._nnode_in, wn.shape[1] - self._nnode_in
            with torch.no_grad():
                self._wsplit = tuple(t.contiguous() for t in (we[:, :n], we[:, n:2 * n], we[:, 2 * n:], wn[:, :a], wn[:, a:]))
            self._wsplit_key = key
        return self._wsplit

    def _node_image(self, w_agg, w_x, lins, next_layer, nw):
        """the packed register image of the node update's weights (+ the next layer's x_i / x_j blocks), re-packed only when one changed"""
        key = (self._wsplit_key, lins[1].weight._version, lins[2].weight._version, lins[1].weight.data_ptr(), lins[2].weight.data_ptr(),
               None if next_layer is None else next_layer._wsplit_key, edge_mlp3_mode())
        if getattr(self, "_nimg_key", None) != key:
            with torch.no_grad():
                self._nimg = node_update_pack(w_agg, w_x, lins[1].weight, lins[2].weight, nw[0], nw[1])
            self._nimg_key = key
        return self._nimg

    def _edge_image(self, w_e, elins):
        """the packed LDS image of the edge MLP's three weights, re-packed only when a weight changed"""
        key = (self._wsplit_key, elins[1].weight._version, elins[2].weight._version, elins[1].weight.data_ptr(), elins[2].weight.data_ptr(),
               edge_mlp3_mode())
        if getattr(self, "_eimg_key", None) != key:
            with torch.no_grad():
                self._eimg = edge_mlp3_pack(w_e, elins[1].weight, elins[2].weight)
            self._eimg_key = key
        return self._eimg

    def forward_inference(self, x, edge_index, e0, scale: float, xa=None, xb=None, next_layer=None, e0_absmax=None, plan=None):
        """Same arithmetic as forward() for edge features scale * e0 (scale = 2^l after l layers), no autograd.
        Per layer the [E,128] activations make three read+write passes (one per Linear, with gather / bias / ReLU /
        LayerNorm in the epilogues) and one read by the segmented sum; the node level is ONE launch
        (csplat_gnn_node_update) that also forms the x_i / x_j column-block products of `next_layer`'s first edge
        Linear.  Returns (x_new, xa_next, xb_next); xa / xb for this layer are computed here when not handed in."""
        csr = GraphCSR.get(edge_index, x.shape[0])
        w_i, w_j, w_e, w_agg, w_x = self._split_weights()
        if xa is None and NODE_UPDATE_PACKED and x.shape[0] > 0:
            # both node-level products of the first layer in ONE launch on pre-packed 16-bit pieces (csplat_gnn_rows_chain, round 6: 8 us
            # against two exact-fp32 launches of 16 us at N = 10^4); later layers get theirs from the previous layer's node update
            key = (self._wsplit_key, edge_mlp3_mode())
            if getattr(self, "_pair_key", None) != key:
                with torch.no_grad():
                    self._pair_img = rows_chain_pack(0, w_i, w_j)
                self._pair_key = key
            xa, xb = rows_chain(x, self._pair_img, 0)
        elif xa is None:
            xa = linear128(x, w_i)                               # contribution of x_i = x[edge_index[1]]
            xb = linear128(x, w_j)                               # contribution of x_j = x[edge_index[0]]
        elins = list(self.edge_fn[0].children())[0::2]
        if len(elins) == 3 and EDGE_MLP_FUSED and _is_pow2(scale):
            # the whole message MLP + LayerNorm in ONE launch: the two inner [E,128] activations never leave the registers
            # (csplat_gnn_edge_mlp3; the weights' LDS image is packed once per weight version)
            if plan is not None:
                # e0 and the index arrays are in destination order: the launch leaves per-run sums, a node's aggregate = its few pieces
                pieces = torch.empty(max(plan["npieces"], 1), 128, dtype=torch.float32, device=x.device)
                edge_mlp3(e0, scale, xa, plan["dst"], xb, plan["src"], self._edge_image(w_e, elins), elins[0].bias, elins[1].bias,
                          elins[2].bias, self.edge_fn[1], e0_absmax=e0_absmax, agg=(plan["gp0"], pieces))
                msg = None
                lins_n = list(self.node_fn[0].children())[0::2]
                if NODE_UPDATE_PACKED and len(lins_n) == 3:
                    # ... added up by the node update's own row loader
                    nw = next_layer._split_weights() if next_layer is not None else (None, None)
                    return node_update_packed(pieces, x, self._node_image(w_agg, w_x, lins_n, next_layer, nw), lins_n[0].bias, lins_n[1].bias,
                                              lins_n[2].bias, self.node_fn[1], next_layer is not None, piece_ptr=plan["pp"])
                agg = segment_sum_rows(pieces[:plan["npieces"]], plan["pp"], plan["iota"], x.shape[0])
            else:
                msg = edge_mlp3(e0, scale, xa, csr.ei[1], xb, csr.ei[0], self._edge_image(w_e, elins), elins[0].bias, elins[1].bias,
                                elins[2].bias, self.edge_fn[1], e0_absmax=e0_absmax)
        else:
            h = linear128(e0, w_e, self.edge_fn[0][0].bias, alpha=scale, relu=True, gather=(xa, csr.ei[1], xb, csr.ei[0]))
            msg = _fused_tail(self.edge_fn, h)
        if msg is not None:
            agg = SegmentSum.apply(msg, csr)
        lins = list(self.node_fn[0].children())[0::2]
        if len(lins) == 3:
            nw = next_layer._split_weights() if next_layer is not None else (None, None)
            if NODE_UPDATE_PACKED:
                return node_update_packed(agg, x, self._node_image(w_agg, w_x, lins, next_layer, nw), lins[0].bias, lins[1].bias, lins[2].bias,
                                          self.node_fn[1], next_layer is not None)
            return node_update(agg, x, w_agg, w_x, lins[0].bias, lins[1], lins[2], self.node_fn[1], nw[0], nw[1])
        t = linear128(x, w_x, out=xa)
        hn = linear128(agg, w_agg, self.node_fn[0][0].bias, relu=True, add_pre=t, out=t)
        return _fused_tail(self.node_fn, hn, add_post=x), None, None


class Processor(nn.Module):
    """graph_network.py:225-292 (declared aggr='max' upstream but never propagates itself)."""

    def __init__(self, nnode_in: int, nnode_out: int, nedge_in: int, nedge_out: int, nmessage_passing_steps: int,
                 nmlp_layers: int, mlp_hidden_dim: int):
        super().__init__()
        self.aggr = 'max'
        self.gnn_stacks = nn.ModuleList([
            InteractionNetwork(nnode_in=nnode_in, nnode_out=nnode_out, nedge_in=nedge_in, nedge_out=nedge_out,
                               nmlp_layers=nmlp_layers, mlp_hidden_dim=mlp_hidden_dim)
            for _ in range(nmessage_passing_steps)])

    def takes_destination_order(self, x: torch.Tensor, edge_features: torch.Tensor) -> bool:
        """whether forward() will run the one-launch edge MLP with its fused aggregation on these inputs -- the path that wants the edge
        latents in destination order (GraphCSR.agg_plan) and can be handed them that way (dst_order=)"""
        # (piece rows are addressed as index * 512 in 32 bits: E / 8 + N pieces must stay below 2^22 -- larger graphs write message rows)
        return bool(len(self.gnn_stacks)) and all(g.inference_ok(x, edge_features) for g in self.gnn_stacks) and EDGE_MLP_FUSED and \
            EDGE_AGG_FUSED and edge_mlp3_mode() == 0 and edge_features.numel() > 0 and \
            edge_features.shape[0] // 8 + x.shape[0] + 8 < (1 << 22) and \
            all(len(list(g.edge_fn[0].children())[0::2]) == 3 for g in self.gnn_stacks)

    def forward(self, x: torch.Tensor, edge_index: torch.Tensor, edge_features: torch.Tensor, edges_out: bool = True, dst_order=None,
                first_products=None):
        """edges_out=False (EncodeProcessDecode, which drops them): the rollout path returns None for the edge latents instead of spending
        a pass over [E,128] on 2^M * e0.  dst_order=(plan, bound): edge_features ALREADY are in the destination order of `plan`
        (GraphCSR.agg_plan of this edge_index) and `bound` is a device scalar >= max |edge_features| (only when takes_destination_order())"""
        if dst_order is not None:
            assert self.takes_destination_order(x, edge_features) and not edges_out
            plan, amax = dst_order
            e0_run, scale, xa, xb = edge_features.contiguous(), 1.0, None, None
            if first_products is not None:      # (the first layer's x_i / x_j products, already formed: meshnet.rollout's csplat_gnn_rows_chain)
                xa, xb = first_products
            for l, gnn in enumerate(self.gnn_stacks):
                nxt = self.gnn_stacks[l + 1] if l + 1 < len(self.gnn_stacks) else None
                x, xa, xb = gnn.forward_inference(x, edge_index, e0_run, scale, xa, xb, nxt, e0_absmax=amax, plan=plan)
                scale *= 2.0
            return x, None
        if len(self.gnn_stacks) and all(g.inference_ok(x, edge_features) for g in self.gnn_stacks):
            # rollout: every layer doubles the edge features (F7), so carry e0 and the scalar 2^l instead of 15 [E,128] passes
            e0, scale = edge_features.contiguous(), 1.0
            xa = xb = None
            # (the fp16 pieces of the one-launch edge MLP take their scale from max |e0|: one pass for all the layers)
            fp16 = EDGE_MLP_FUSED and edge_mlp3_mode() == 0 and e0.numel() > 0
            plan = e0_run = amax = None
            if self.takes_destination_order(x, edge_features):
                plan = GraphCSR.get(edge_index, x.shape[0]).agg_plan()
                # the edge latents in destination order, once for all the layers (and max |e0| from the same pass)
                e0_run, amax = gather_rows(e0, plan["perm"], with_absmax=True)
            elif fp16:
                amax = absmax(e0)
            for l, gnn in enumerate(self.gnn_stacks):
                nxt = self.gnn_stacks[l + 1] if l + 1 < len(self.gnn_stacks) else None
                x, xa, xb = gnn.forward_inference(x, edge_index, e0 if plan is None else e0_run, scale, xa, xb, nxt, e0_absmax=amax, plan=plan)
                scale *= 2.0
            return x, (e0 * scale if edges_out else None)
        if len(self.gnn_stacks) and not torch.is_grad_enabled():
            # inference on a form the one-pass rollout kernels do not cover (they are built for the 128-wide fp32 network of config 4)
            from csplat import native as _n
            _n.composed_fallback("graph_network.Processor.forward", "dtype" if (x.dtype != torch.float32 or edge_features.dtype != torch.float32)
                                 else "shape", x)
        e_base, scale = edge_features, 1.0
        for gnn in self.gnn_stacks:
            x, e_base = gnn.message_update(x, edge_index, e_base, scale)
            scale *= 2.0
        return x, (e_base * scale if len(self.gnn_stacks) else e_base)


class Decoder(nn.Module):
    """graph_network.py:295-332"""

    def __init__(self, nnode_in: int, nnode_out: int, nmlp_layers: int, mlp_hidden_dim: int):
        super().__init__()
        self.node_fn = build_mlp(nnode_in, [mlp_hidden_dim for _ in range(nmlp_layers)], nnode_out)

    def forward(self, x: torch.Tensor):
        lins = list(self.node_fn.children())[0::2]
        acts = list(self.node_fn.children())[1::2]
        if (not torch.is_grad_enabled()) and NODE_UPDATE_PACKED and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and \
                x.shape[0] > 0 and x.shape[1] == 128 and len(lins) == 3 and tuple(lins[0].weight.shape) == (128, 128) and \
                tuple(lins[1].weight.shape) == (128, 128) and lins[2].in_features == 128 and \
                isinstance(acts[0], nn.ReLU) and isinstance(acts[1], nn.ReLU) and isinstance(acts[2], nn.Identity) and \
                lins[0].weight.dtype == torch.float32:
            # rollout: the two hidden layers in ONE launch on pre-packed 16-bit pieces (csplat_gnn_rows_chain, round 6) instead of two
            # library GEMMs + two ReLU launches; the narrow last Linear stays the library's
            key = tuple((l.weight._version, l.weight.data_ptr()) for l in lins[:2]) + (edge_mlp3_mode(),)
            if getattr(self, "_chain_key", None) != key:
                with torch.no_grad():
                    self._chain_img = rows_chain_pack(1, lins[0].weight, lins[1].weight)
                self._chain_key = key
            h = rows_chain(x, self._chain_img, 1, lins[0].bias, lins[1].bias)
            return torch.nn.functional.linear(h, lins[2].weight, lins[2].bias)
        return self.node_fn(x)


class EncodeProcessDecode(nn.Module):
    """graph_network.py:335-408"""

    def __init__(self, nnode_in_features: int, nnode_out_features: int, nedge_in_features: int, latent_dim: int,
                 nmessage_passing_steps: int, nmlp_layers: int, mlp_hidden_dim: int):
        super().__init__()
        self._encoder = Encoder(nnode_in_features=nnode_in_features, nnode_out_features=latent_dim,
                                nedge_in_features=nedge_in_features, nedge_out_features=latent_dim,
                                nmlp_layers=nmlp_layers, mlp_hidden_dim=mlp_hidden_dim)
        self._processor = Processor(nnode_in=latent_dim, nnode_out=latent_dim, nedge_in=latent_dim, nedge_out=latent_dim,
                                    nmessage_passing_steps=nmessage_passing_steps, nmlp_layers=nmlp_layers,
                                    mlp_hidden_dim=mlp_hidden_dim)
        self._decoder = Decoder(nnode_in=latent_dim, nnode_out=nnode_out_features, nmlp_layers=nmlp_layers,
                                mlp_hidden_dim=mlp_hidden_dim)

    def _edge_latent_bound(self):
        """a device scalar >= |any edge latent the encoder can produce|: its MLP ends in LayerNorm(128), whose normalised values are at most
        sqrt(127) in magnitude, so max|gamma| * sqrt(127) + max|beta| bounds them -- what the one-launch edge MLP takes its fp16 scale from
        without a pass over the [E,128] latents (re-derived when the LayerNorm's parameters change)"""
        ln = self._encoder.edge_fn[1]
        key = (ln.weight._version, ln.bias._version, ln.weight.data_ptr(), ln.weight.device)
        if getattr(self, "_elb_key", None) != key:
            with torch.no_grad():
                self._elb = (ln.weight.abs().max() * (127.0 ** 0.5) + ln.bias.abs().max()).reshape(1).float().contiguous()
            self._elb_key = key
        return self._elb

    def forward(self, x: torch.Tensor, edge_index: torch.Tensor, edge_features: torch.Tensor):
        """The rollout's default arithmetic multiplies with two fp16 pieces per operand (edge_mlp3_mode 0), whose exponent range a trained
        checkpoint with large hidden activations can leave; every ReLU of those kernels lets a NaN through, so an overflow ANYWHERE reaches
        the decoder's output as non-finite rows -- never as finite garbage (csplat_edge_mlp.hip: relu_nan).  Round 6: that is detected
        (one device word per call) and the call is repeated with three bf16 pieces (fp32's exponent range), which this module then keeps
        (`_bf16_latched`, logged once): the reference's fp32 path has no such failure, and neither has the drop-in.  Inside
        `deferred_overflow_check()` (meshnet.rollout.rollout) the word is collected instead of read, so that the loop stays free of
        host reads; the rollout is repeated from its start when a step overflowed."""
        if torch.is_grad_enabled() or not x.is_cuda:
            return self._forward(x, edge_index, edge_features)
        if getattr(self, "_bf16_latched", False) and edge_mlp3_mode() == 0:
            was = edge_mlp3_mode(1)
            try:
                return self._forward(x, edge_index, edge_features)
            finally:
                edge_mlp3_mode(was)
        out = self._forward(x, edge_index, edge_features)
        if edge_mlp3_mode() != 0 or not (EDGE_MLP_FUSED or NODE_UPDATE_PACKED or ENCODER_FUSED):
            return out
        ok = torch.isfinite(out).all()              # (two small launches on [N, out]; read here, or collected and read once per rollout)
        if _OVERFLOW_COLLECT:
            _OVERFLOW_COLLECT[-1].append((self, ok))
            return out
        if not bool(ok):
            was = edge_mlp3_mode(1)
            try:
                again = self._forward(x, edge_index, edge_features)
            finally:
                edge_mlp3_mode(was)
            if bool(torch.isfinite(again).all()):      # (non-finite under fp32's exponent range too: the INPUT was -- nothing to latch)
                self.latch_bf16()
            return again
        return out

    def latch_bf16(self):
        """from now on this module's inference runs with three bf16 pieces (csplat_gnn_edge_mlp3_mode 1)"""
        if not getattr(self, "_bf16_latched", False):
            import warnings
            warnings.warn("meshnet.graph_network.EncodeProcessDecode: an activation left fp16's range under the two-piece fp16 arithmetic "
                          "(non-finite output rows); this module now runs with three bf16 pieces (csplat_gnn_edge_mlp3_mode 1)")
        self._bf16_latched = True

    def _forward(self, x: torch.Tensor, edge_index: torch.Tensor, edge_features: torch.Tensor):
        if not torch.is_grad_enabled() and EDGE_MLP_FUSED and EDGE_AGG_FUSED and edge_mlp3_mode() == 0 and edge_features.is_cuda and \
                edge_features.dim() == 2 and edge_features.shape[0] > 0 and edge_features.dtype == torch.float32 and x.dtype == torch.float32 and \
                isinstance(self._encoder.edge_fn[1], nn.LayerNorm) and self._encoder.edge_fn[1].elementwise_affine and \
                len(self._processor.gnn_stacks) > 0 and all(g._nnode_in == 128 and g._nedge_in == 128 for g in self._processor.gnn_stacks):
            # rollout: the processor's edge launch wants the edges in destination order -- so ENCODE them in that order (a gather of the
            # [E,4] inputs instead of one of the [E,128] latents)
            plan = GraphCSR.get(edge_index, x.shape[0]).agg_plan()
            xe, ee = self._encoder(x, gather_rows(edge_features, plan["perm"]))
            if self._processor.takes_destination_order(xe, ee):
                xp, _edges = self._processor(xe, edge_index, ee, edges_out=False, dst_order=(plan, self._edge_latent_bound()))
                return self._decoder(xp)
            x, edge_features = xe, torch.empty_like(ee).index_copy_(0, plan["perm"], ee)      # (not that path after all: back to edge order)
        else:
            x, edge_features = self._encoder(x, edge_features)
        x, _edges = self._processor(x, edge_index, edge_features, edges_out=False)
        x = self._decoder(x)
        return x
