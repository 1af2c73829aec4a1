"""Autograd wrappers around the csplat_gnn_* C-ABI entry points (include/csplat.h): the gather / concat /
scatter-add that torch_geometric's MessagePassing.propagate performs for InteractionNetwork
(/root/reference/meshnet/graph_network.py:136,173-174,197).  GPU only; no fallback."""
import weakref

import torch

from csplat import native as _n


class GraphCSR:
    """Two CSR orderings (by destination = edge_index[1], by source = edge_index[0]) of one edge list, built once
    per graph on the GPU and reused by every message-passing layer and by the backward pass."""

    _cache = {}

    def __init__(self, edge_index: torch.Tensor, num_nodes: int):
        _n.require_cuda(edge_index)
        assert edge_index.dtype == torch.int64 and edge_index.dim() == 2 and edge_index.shape[0] == 2
        self.ei = edge_index.contiguous()
        self.N, self.E = int(num_nodes), int(edge_index.shape[1])
        dev = edge_index.device
        self.rowptr, self.perm = {}, {}
        tmp = torch.empty(max(int(_n.lib.csplat_gnn_csr_temp_bytes(self.N, self.E)), 256), dtype=torch.uint8, device=dev)
        with _n.on_device(dev):
            for name, row in (("src", 0), ("dst", 1)):
                rp = torch.empty(self.N + 1, dtype=torch.int32, device=dev)
                pm = torch.empty(max(self.E, 1), dtype=torch.int32, device=dev)
                _n.check(_n.lib.csplat_gnn_build_csr(_n.stream_handle(dev), self.N, self.E, self.ei[row].data_ptr(),
                                                     _n.ptr(rp), _n.ptr(pm), _n.ptr(tmp)), "csplat_gnn_build_csr")
                self.rowptr[name], self.perm[name] = rp, pm
        self._deg_dst = None

    @property
    def deg_dst(self):
        """[N] float32: number of edges arriving at each node"""
        if self._deg_dst is None:
            rp = self.rowptr["dst"]
            self._deg_dst = (rp[1:] - rp[:-1]).to(torch.float32)
        return self._deg_dst

    def agg_plan(self):
        """What the one-launch edge MLP needs to sum its messages per destination itself (csplat_gnn_edge_mlp3 with `pieces`, include/csplat.h):
        the edge list in destination order -- perm (int64, position -> edge id), dst / src in that order -- and the numbering of the
        pieces: runs of equal destination, cut additionally every 8 rows.  gp0[g] = first piece of rows 8g .. 8g + 7; node v's pieces are
        pp[v] .. pp[v + 1] - 1.  Built once per graph with stock tensor operations."""
        if getattr(self, "_agg_plan", None) is None:
            dev = self.ei.device
            perm = self.perm["dst"][:self.E].long()
            dst_s, src_s = self.ei[1][perm].contiguous(), self.ei[0][perm].contiguous()
            gp0, pp, npieces = piece_numbering(dst_s, self.rowptr["dst"])
            self._agg_plan = {"perm": perm, "dst": dst_s, "src": src_s, "gp0": gp0, "pp": pp, "npieces": npieces,
                              "iota": torch.arange(max(npieces, 1), dtype=torch.int32, device=dev)}
        return self._agg_plan

    @classmethod
    def get(cls, edge_index, num_nodes):
        """CSR for this edge_index tensor OBJECT (weakly referenced) at its current in-place version."""
        key = id(edge_index)
        hit = cls._cache.get(key)
        if hit is not None:
            ref, version, n, csr = hit
            if ref() is edge_index and version == edge_index._version and n == int(num_nodes):
                return csr
        for k in [k for k, v in cls._cache.items() if v[0]() is None]:
            del cls._cache[k]
        csr = cls(edge_index, num_nodes)
        cls._cache[key] = (weakref.ref(edge_index), edge_index._version, int(num_nodes), csr)
        return csr


def _f32(t):
    return t.contiguous() if t.dtype == torch.float32 else t.float().contiguous()


def piece_numbering(dst_sorted, rowptr, group=8):
    """The numbering of the "pieces" the one-launch edge MLP leaves instead of messages (csplat_gnn_edge_mlp3 with `pieces`): rows are edges in
    destination order; a piece = a maximal run of rows with one destination, cut additionally at every multiple of `group` rows (a wave
    of the kernel sums its own 8 rows); pieces are numbered in row order.  Returns (gp0 [ceil(E / group)] int32: first piece of each
    group of rows, pp [N + 1] int32: node v's pieces are pp[v] .. pp[v + 1] - 1, npieces).  Plain tensor operations: any device."""
    E = int(dst_sorted.numel())
    dev = dst_sorted.device
    if E == 0:
        return torch.zeros(0, dtype=torch.int32, device=dev), torch.zeros(rowptr.numel(), dtype=torch.int32, device=dev), 0
    new = (torch.arange(E, device=dev) % group == 0)
    if E > 1:
        new[1:] |= dst_sorted[1:] != dst_sorted[:-1]
    pidx = torch.cumsum(new.to(torch.int32), 0, dtype=torch.int32) - 1
    npieces = int(pidx[-1].item()) + 1
    gp0 = pidx[0::group].contiguous()
    pidx_ext = torch.cat([pidx, torch.tensor([npieces], dtype=torch.int32, device=dev)])
    return gp0, pidx_ext[rowptr.long()].contiguous(), npieces


class EdgeCombine(torch.autograd.Function):
    """h[e] = relu?(xa[dst[e]] + xb[src[e]] + ec[e]) -- first edge-MLP layer with the [x_i, x_j, e] concat folded away."""

    @staticmethod
    def forward(ctx, xa, xb, ec, csr, relu, grad_premasked=False):
        """grad_premasked: the consumer (EdgeTailAggregate) hands back the gradient of the PRE-activation -- it folds this ReLU's
        backward into the GEMM that produces the gradient -- so backward() must not mask again"""
        xa, xb, ec = _f32(xa), _f32(xb), _f32(ec)
        E, L = ec.shape
        out = torch.empty_like(ec)
        with _n.on_device(ec.device):
            _n.check(_n.lib.csplat_gnn_edge_combine_fwd(_n.stream_handle(ec.device), csr.N, E, L, _n.ptr(csr.ei), _n.ptr(xa),
                                                        _n.ptr(xb), _n.ptr(ec), int(relu), _n.ptr(out)),
                     "csplat_gnn_edge_combine_fwd")
        ctx.csr, ctx.relu = csr, bool(relu) and not grad_premasked
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, g):
        (out,) = ctx.saved_tensors
        csr = ctx.csr
        g = _f32(g)
        E, L = out.shape
        gm = torch.empty_like(g) if ctx.relu else g
        dxa = torch.empty(csr.N, L, dtype=torch.float32, device=g.device)
        dxb = torch.empty_like(dxa)
        with _n.on_device(g.device):
            _n.check(_n.lib.csplat_gnn_edge_combine_bwd(
                _n.stream_handle(g.device), csr.N, E, L, _n.ptr(g), _n.ptr(out), int(ctx.relu), _n.ptr(csr.rowptr["dst"]),
                _n.ptr(csr.perm["dst"]), _n.ptr(csr.rowptr["src"]), _n.ptr(csr.perm["src"]), _n.ptr(gm), _n.ptr(dxa),
                _n.ptr(dxb)), "csplat_gnn_edge_combine_bwd")
        return dxa, dxb, gm, None, None, None


class SegmentSum(torch.autograd.Function):
    """agg[n] = sum_{e: dst(e) = n} msg[e]  (aggr='add', dim_size = N), fixed ascending-edge-id order."""

    @staticmethod
    def forward(ctx, msg, csr):
        msg = _f32(msg)
        E, L = msg.shape
        agg = torch.empty(csr.N, L, dtype=torch.float32, device=msg.device)
        with _n.on_device(msg.device):
            _n.check(_n.lib.csplat_gnn_segment_sum(_n.stream_handle(msg.device), csr.N, E, L, _n.ptr(msg),
                                                   _n.ptr(csr.rowptr["dst"]), _n.ptr(csr.perm["dst"]), _n.ptr(agg)),
                     "csplat_gnn_segment_sum")
        ctx.csr = csr
        return agg

    @staticmethod
    def backward(ctx, g):
        csr = ctx.csr
        g = _f32(g)
        L = g.shape[1]
        out = torch.empty(csr.E, L, dtype=torch.float32, device=g.device)
        with _n.on_device(g.device):
            _n.check(_n.lib.csplat_gnn_gather_rows(_n.stream_handle(g.device), csr.E, L, _n.ptr(g), csr.ei[1].data_ptr(),
                                                   _n.ptr(out)), "csplat_gnn_gather_rows")
        return out, None


class RowsDot(torch.autograd.Function):
    """y[T, R] = h[T, 256] @ W[R, 256]^T + b -- the simulator's 256 -> 3V output layer for the T time values of a step at once
    (csplat_rows_dot_fwd / _bwd: the weight is streamed once per direction; deterministic)."""

    @staticmethod
    def forward(ctx, h, weight, bias, add=None):
        h, weight, bias = _f32(h), _f32(weight), _f32(bias)
        T, K = h.shape
        R = weight.shape[0]
        y = torch.empty(T, R, dtype=torch.float32, device=h.device)
        add = None if add is None else _f32(add.reshape(T, R))
        with _n.on_device(h.device):
            _n.check(_n.lib.csplat_rows_dot_fwd(_n.stream_handle(h.device), T, R, K, _n.ptr(weight), _n.ptr(bias), _n.ptr(h),
                                                _n.ptr(y), None if add is None else _n.ptr(add)), "csplat_rows_dot_fwd")
        ctx.add_shape = None if add is None else add.shape
        ctx.save_for_backward(h, weight)
        return y

    @staticmethod
    def backward(ctx, g):
        h, weight = ctx.saved_tensors
        g = _f32(g)
        T, K = h.shape
        R = weight.shape[0]
        dW, db, dh = torch.empty_like(weight), torch.empty(R, dtype=torch.float32, device=g.device), torch.empty_like(h)
        scratch = torch.empty(_n.lib.csplat_rows_dot_scratch_bytes(T), dtype=torch.uint8, device=g.device)
        with _n.on_device(g.device):
            _n.check(_n.lib.csplat_rows_dot_bwd(_n.stream_handle(g.device), T, R, K, _n.ptr(weight), _n.ptr(h), _n.ptr(g),
                                                _n.ptr(dW), _n.ptr(db), _n.ptr(dh), _n.ptr(scratch)), "csplat_rows_dot_bwd")
        return dh, dW, db, (g if ctx.add_shape is not None and ctx.needs_input_grad[3] else None)


_SIMH_SCRATCH = {}
_n.TICKET_CACHES.append(_SIMH_SCRATCH)


def _sim_hidden_fwd(e, W1, b1, W2, b2):
    e, W1, b1, W2, b2 = (_f32(t) for t in (e, W1, b1, W2, b2))
    T, K0 = int(e.shape[0]), int(e.shape[1])
    h1 = torch.empty(T, 256, dtype=torch.float32, device=e.device)
    h2 = torch.empty(T, 256, dtype=torch.float32, device=e.device)
    with _n.on_device(e.device):
        _n.check(_n.lib.csplat_sim_hidden_fwd(_n.stream_handle(e.device), T, K0, _n.ptr(e), _n.ptr(W1), _n.ptr(b1), _n.ptr(W2), _n.ptr(b2),
                                              _n.ptr(h1), _n.ptr(h2)), "csplat_sim_hidden_fwd")
    return e, W2, h1, h2


def _sim_hidden_bwd(e, W2, h1, h2, g, sinks=(None, None, None, None)):
    """sinks: csplat.native.grad_sink keys of (W1, b1, W2, b2) -- where the gradients are to be written (csplat.dist.FlatGrads)"""
    T, K0 = int(e.shape[0]), int(e.shape[1])
    g = _f32(g)
    dev = g.device
    dW1 = _n.grad_out(sinks[0], (256, K0), dev)
    dW2 = _n.grad_out(sinks[2], (256, 256), dev)
    db = [_n.grad_out(sinks[1], (256,), dev), _n.grad_out(sinks[3], (256,), dev)]
    with _n.on_device(dev):
        key = (str(dev), _n.scratch_stream(dev))
        scratch = _SIMH_SCRATCH.get(key)        # zeroed once per stream: the kernel leaves its ticket word at zero
        if scratch is None:
            scratch = _SIMH_SCRATCH[key] = torch.zeros(int(_n.lib.csplat_sim_hidden_scratch_bytes(8)) // 4, dtype=torch.int32, device=dev)
        _n.check(_n.lib.csplat_sim_hidden_bwd(_n.stream_handle(dev), T, K0, _n.ptr(e), _n.ptr(W2), _n.ptr(h1), _n.ptr(h2), _n.ptr(g),
                                                       _n.ptr(dW1), _n.ptr(db[0]), _n.ptr(dW2), _n.ptr(db[1]), _n.ptr(scratch)),
                 "csplat_sim_hidden_bwd")
    return dW1, db[0], dW2, db[1]


class SimHidden(torch.autograd.Function):
    """relu(Linear(K0, 256)) -> relu(Linear(256, 256)) of the time-conditioned simulator (meshnet_network.py:337-338,364-366) for the
    T <= 8 time rows of a step: csplat_sim_hidden_fwd / _bwd, one launch each way instead of ~20 torch launches."""

    @staticmethod
    def forward(ctx, e, W1, b1, W2, b2):
        saved = _sim_hidden_fwd(e, W1, b1, W2, b2)
        ctx.save_for_backward(*saved)
        return saved[3]

    @staticmethod
    def backward(ctx, g):
        return (None,) + _sim_hidden_bwd(*ctx.saved_tensors, g)


def _rows_dot_fwd(h2, Wo, bo, base=None):
    Wo, bo = _f32(Wo), _f32(bo)
    T, R = int(h2.shape[0]), int(Wo.shape[0])
    y = torch.empty(T, R, dtype=torch.float32, device=h2.device)
    add = None if base is None else _f32(base.reshape(T, R))
    with _n.on_device(h2.device):
        _n.check(_n.lib.csplat_rows_dot_fwd(_n.stream_handle(h2.device), T, R, 256, _n.ptr(Wo), _n.ptr(bo), _n.ptr(h2), _n.ptr(y),
                                            None if add is None else _n.ptr(add)), "csplat_rows_dot_fwd")
    return y, Wo


def _rows_dot_bwd(Wo, h2, g, sinks=(None, None)):
    g = _f32(g)
    T, R = int(h2.shape[0]), int(Wo.shape[0])
    dWo, dbo, dh = _n.grad_out(sinks[0], Wo.shape, g.device), _n.grad_out(sinks[1], (R,), g.device), torch.empty_like(h2)
    scratch = torch.empty(_n.lib.csplat_rows_dot_scratch_bytes(T), dtype=torch.uint8, device=g.device)
    with _n.on_device(g.device):
        _n.check(_n.lib.csplat_rows_dot_bwd(_n.stream_handle(g.device), T, R, 256, _n.ptr(Wo), _n.ptr(h2), _n.ptr(g.reshape(T, R)),
                                            _n.ptr(dWo), _n.ptr(dbo), _n.ptr(dh), _n.ptr(scratch)), "csplat_rows_dot_bwd")
    return dWo, dbo, dh


def sim_residual_applies(e, lin1, lin2, lin_out, base=None):
    return bool(e.is_cuda and e.dim() == 2 and 0 < e.shape[0] <= 8 and e.shape[1] <= 16 and tuple(lin1.weight.shape) == (256, e.shape[1]) and
                tuple(lin2.weight.shape) == (256, 256) and lin1.bias is not None and lin2.bias is not None and not e.requires_grad and
                lin_out.weight.shape[1] == 256 and lin_out.bias is not None and
                (base is None or (base.is_cuda and base.numel() == e.shape[0] * lin_out.weight.shape[0])))


class SimResidual(torch.autograd.Function):
    """the simulator's whole residual MLP for the T time rows of a step as ONE autograd node:
        y = Linear(256, 3V)(relu(Linear(256, 256)(relu(Linear(K0, 256)(e))))) [+ base]          (meshnet_network.py:337-339,364-371)
    = SimHidden followed by RowsDot (the same two launches each way); one node less to record and to walk in a step that is bound
    by the host.  e (the sinusoidal code of the time values) gets no gradient."""

    @staticmethod
    def forward(ctx, e, W1, b1, W2, b2, Wo, bo, base=None):
        e, W2, h1, h2 = _sim_hidden_fwd(e, W1, b1, W2, b2)
        Wo, bo = _f32(Wo), _f32(bo)
        T, R = int(h2.shape[0]), int(Wo.shape[0])
        y = torch.empty(T, R, dtype=torch.float32, device=h2.device)
        add = None if base is None else _f32(base.reshape(T, R))
        with _n.on_device(h2.device):
            _n.check(_n.lib.csplat_rows_dot_fwd(_n.stream_handle(h2.device), T, R, 256, _n.ptr(Wo), _n.ptr(bo), _n.ptr(h2), _n.ptr(y),
                                                None if add is None else _n.ptr(add)), "csplat_rows_dot_fwd")
        ctx.save_for_backward(e, W2, h1, h2, Wo)
        ctx.has_base = base is not None
        return y

    @staticmethod
    def backward(ctx, g):
        e, W2, h1, h2, Wo = ctx.saved_tensors
        g = _f32(g)
        T, R = int(h2.shape[0]), int(Wo.shape[0])
        dWo, dbo, dh = torch.empty_like(Wo), torch.empty(R, dtype=torch.float32, device=g.device), torch.empty_like(h2)
        scratch = torch.empty(_n.lib.csplat_rows_dot_scratch_bytes(T), dtype=torch.uint8, device=g.device)
        with _n.on_device(g.device):
            _n.check(_n.lib.csplat_rows_dot_bwd(_n.stream_handle(g.device), T, R, 256, _n.ptr(Wo), _n.ptr(h2), _n.ptr(g),
                                                _n.ptr(dWo), _n.ptr(dbo), _n.ptr(dh), _n.ptr(scratch)), "csplat_rows_dot_bwd")
        dW1, db1, dW2, db2 = _sim_hidden_bwd(e, W2, h1, h2, dh)
        return None, dW1, db1, dW2, db2, dWo, dbo, (g if ctx.has_base and ctx.needs_input_grad[7] else None)


def sim_residual(e, lin1, lin2, lin_out, base=None):
    """rows_dot(sim_hidden(e, lin1, lin2), lin_out.weight, lin_out.bias, base) as one autograd node when both fused forms apply"""
    if e.is_cuda and e.dim() == 2 and 0 < e.shape[0] <= 8 and e.shape[1] <= 16 and tuple(lin1.weight.shape) == (256, e.shape[1]) and \
            tuple(lin2.weight.shape) == (256, 256) and lin1.bias is not None and lin2.bias is not None and not e.requires_grad and \
            lin_out.weight.shape[1] == 256 and lin_out.bias is not None and \
            (base is None or (base.is_cuda and base.numel() == e.shape[0] * lin_out.weight.shape[0])):
        return SimResidual.apply(e, lin1.weight, lin1.bias, lin2.weight, lin2.bias, lin_out.weight, lin_out.bias, base)
    return rows_dot(sim_hidden(e, lin1, lin2), lin_out.weight, lin_out.bias, base)


def sim_hidden(e, lin1, lin2):
    """relu(lin2(relu(lin1(e)))) for the simulator's two hidden layers; the fused HIP form for T <= 8 fp32 GPU rows of the reference's
    shape (K0 <= 16 -> 256 -> 256), composed torch otherwise"""
    if e.is_cuda and e.dim() == 2 and 0 < e.shape[0] <= 8 and e.shape[1] <= 16 and tuple(lin1.weight.shape) == (256, e.shape[1]) and \
            tuple(lin2.weight.shape) == (256, 256) and lin1.bias is not None and lin2.bias is not None and not e.requires_grad:
        return SimHidden.apply(e, lin1.weight, lin1.bias, lin2.weight, lin2.bias)
    _n.composed_fallback("graph_ops.sim_hidden", "shape", e)
    return torch.relu(lin2(torch.relu(lin1(e))))


def rows_dot(h, weight, bias, add=None):
    """F.linear(h, weight, bias) [+ add, shaped like the result] for few rows of h (<= 8) against a tall 256-column weight, at HBM rate
    on the GPU."""
    if h.is_cuda and h.dim() == 2 and h.shape[1] == 256 and weight.shape[1] == 256 and 0 < h.shape[0] <= 8 and bias is not None and \
            (add is None or (add.is_cuda and add.numel() == h.shape[0] * weight.shape[0])):
        return RowsDot.apply(h, weight, bias, add)
    _n.composed_fallback("graph_ops.rows_dot", "shape", h)
    y = torch.nn.functional.linear(h, weight, bias)
    return y if add is None else add.reshape(y.shape) + y


def _weight_layout(weight):
    """(tensor whose data_ptr is W, ldw, w_transposed) for csplat_linear128_ex: row-major matrices, column slices of wider ones and
    transposes of either are read in place; anything else is copied"""
    w = weight.detach()
    if w.stride(1) == 1 and w.stride(0) >= 128 and w.data_ptr() % 16 == 0:
        return w, int(w.stride(0)), 0
    if w.stride(0) == 1 and w.stride(1) >= 128 and w.data_ptr() % 16 == 0:
        return w, int(w.stride(1)), 1
    return w.contiguous(), 128, 0


_UNIT_LN = {}


def _unit_ln(device, eps):
    """LayerNorm(128) with gamma = 1, beta = 0: the normalisation alone"""
    key = (str(device), float(eps))
    if key not in _UNIT_LN:
        ln = torch.nn.LayerNorm(128, eps=eps).to(device)
        for p_ in ln.parameters():
            p_.requires_grad_(False)
        _UNIT_LN[key] = ln
    return _UNIT_LN[key]


def linear128(A, weight, bias=None, alpha=1.0, relu=False, gather=None, layer_norm=None, add_pre=None, add_post=None, out=None,
              mask=None, ln_stats=None):
    """Fused Linear for the 128-wide MeshNet MLP layers (csplat_linear128_ex, include/csplat.h):
        out = [mask > 0] * ( LN?( relu?( alpha * A @ weight^T + bias + ga[ia] + gb[ib] + add_pre ) ) + add_post )
    A [M,128] fp32, weight [128,128] (torch Linear.weight, a column slice of one, or .t() of either: read in place), gather =
    (ga, ia, gb, ib) or None, layer_norm = nn.LayerNorm(128) or None, add_pre / add_post / mask [M,128] or None, ln_stats [M,2] or
    None (receives each row's (mean, rstd) of the LayerNorm epilogue).  No autograd graph
    is recorded: callers use it under torch.no_grad() or inside an autograd Function."""
    _n.require_cuda(A)
    assert A.dtype == torch.float32 and A.dim() == 2 and A.shape[1] == 128 and tuple(weight.shape) == (128, 128)
    A = A.contiguous()
    weight, ldw, wt = _weight_layout(weight)
    M = A.shape[0]
    out = torch.empty_like(A) if out is None else out
    ga = ia = gb = ib = None
    if gather is not None:
        ga, ia, gb, ib = gather
        ga, gb, ia, ib = _f32(ga), _f32(gb), ia.contiguous(), ib.contiguous()
        assert ia.dtype == torch.int64 and ib.dtype == torch.int64 and ia.numel() == M and ib.numel() == M
    g = b = None
    eps = 0.0
    if layer_norm is not None:
        assert tuple(layer_norm.normalized_shape) == (128,) and layer_norm.elementwise_affine
        g, b, eps = layer_norm.weight.detach().contiguous(), layer_norm.bias.detach().contiguous(), float(layer_norm.eps)
    bias = None if bias is None else bias.detach().contiguous()
    add_pre = None if add_pre is None else _f32(add_pre)
    add_post = None if add_post is None else _f32(add_post)
    mask = None if mask is None else _f32(mask)
    with _n.on_device(A.device):
        _n.check(_n.lib.csplat_linear128_ex(_n.stream_handle(A.device), M, _n.ptr(A), weight.data_ptr(), ldw, wt, _n.ptr(bias), float(alpha),
                                            int(relu), _n.ptr(ga), _n.ptr(ia), _n.ptr(gb), _n.ptr(ib), _n.ptr(g), _n.ptr(b), eps,
                                            _n.ptr(add_pre), _n.ptr(add_post), _n.ptr(mask), _n.ptr(ln_stats), _n.ptr(out)),
                 "csplat_linear128_ex")
    return out


def linear_narrow128(x, weight, bias=None, relu=False):
    """(ReLU)(x @ weight^T + bias) for a narrow input (1 <= K <= 32 columns) and 128 outputs: the first Linear of the encoders' MLPs
    (csplat_linear_narrow128, include/csplat.h).  No autograd graph is recorded."""
    _n.require_cuda(x)
    assert x.dtype == torch.float32 and x.dim() == 2 and 1 <= x.shape[1] <= 32 and tuple(weight.shape) == (128, x.shape[1])
    x = x if x.stride(1) == 1 else x.contiguous()
    w = weight.detach()
    w = w if w.stride(1) == 1 else w.contiguous()
    bias = None if bias is None else bias.detach().contiguous()
    out = torch.empty(x.shape[0], 128, dtype=torch.float32, device=x.device)
    with _n.on_device(x.device):
        _n.check(_n.lib.csplat_linear_narrow128(_n.stream_handle(x.device), x.shape[0], x.shape[1], _n.ptr(x), int(x.stride(0)) if x.shape[0] > 1 else x.shape[1],
                                                w.data_ptr(), int(w.stride(0)), _n.ptr(bias), int(relu), _n.ptr(out)),
                 "csplat_linear_narrow128")
    return out


def gather_rows(rows, keys, with_absmax=False):
    """rows[keys] for [.,L] fp32 rows and int64 keys (csplat_gnn_gather_rows), no autograd; with_absmax: also max |value| of the result as a
    one-element device tensor, from the same pass (csplat_gnn_gather_rows_absmax)"""
    _n.require_cuda(rows)
    rows, keys = _f32(rows), keys.contiguous()
    assert keys.dtype == torch.int64
    out = torch.empty(keys.numel(), rows.shape[1], dtype=torch.float32, device=rows.device)
    with _n.on_device(rows.device):
        if with_absmax:
            am = torch.empty(1, dtype=torch.float32, device=rows.device)
            _n.check(_n.lib.csplat_gnn_gather_rows_absmax(_n.stream_handle(rows.device), keys.numel(), rows.shape[1], _n.ptr(rows), _n.ptr(keys),
                                                          _n.ptr(out), _n.ptr(am)), "csplat_gnn_gather_rows_absmax")
            return out, am
        _n.check(_n.lib.csplat_gnn_gather_rows(_n.stream_handle(rows.device), keys.numel(), rows.shape[1], _n.ptr(rows), _n.ptr(keys), _n.ptr(out)),
                 "csplat_gnn_gather_rows")
    return out


def segment_sum_rows(rows, rowptr, perm, N):
    """out[v] = sum of rows[perm[rowptr[v] .. rowptr[v + 1] - 1]] in that order (csplat_gnn_segment_sum), no autograd"""
    rows = _f32(rows)
    out = torch.empty(N, rows.shape[1], dtype=torch.float32, device=rows.device)
    with _n.on_device(rows.device):
        _n.check(_n.lib.csplat_gnn_segment_sum(_n.stream_handle(rows.device), N, rows.shape[0], rows.shape[1], _n.ptr(rows), _n.ptr(rowptr), _n.ptr(perm),
                                               _n.ptr(out)), "csplat_gnn_segment_sum")
    return out


def edge_mlp3_mode(mode=None):
    """the arithmetic of csplat_gnn_edge_mlp3 (include/csplat.h): 0 = two fp16 pieces per operand (default), 1 = three bf16 pieces (fp32's
    exponent range).  Returns the mode that was set before; None only queries.  An image is packed FOR a mode."""
    return int(_n.lib.csplat_gnn_edge_mlp3_mode(-1 if mode is None else int(mode)))


def absmax(x):
    """max |x| as a one-element device tensor (csplat_absmax: one pass, no host read) -- what edge_mlp3 takes its fp16 scale from"""
    _n.require_cuda(x)
    x = _f32(x)
    assert x.numel() % 4 == 0
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    with _n.on_device(x.device):
        _n.check(_n.lib.csplat_absmax(_n.stream_handle(x.device), x.numel(), _n.ptr(x), _n.ptr(out)), "csplat_absmax")
    return out


def edge_mlp3_pack(w0, w1, w2):
    """the three 128 x 128 weights of an edge MLP as the register image csplat_gnn_edge_mlp3 loads (the 16-bit pieces of every wave's 32
    output rows as MFMA A operands, the contraction index of layers 2 and 3 permuted to the order the previous layer leaves its
    activations in): pack once per weight version and per edge_mlp3_mode"""
    img = torch.empty(int(_n.lib.csplat_gnn_edge_mlp3_image_bytes()), dtype=torch.uint8, device=w0.device)
    ws = []
    for w in (w0, w1, w2):
        w = w.detach()
        assert tuple(w.shape) == (128, 128) and w.dtype == torch.float32 and w.is_cuda
        if not (w.stride(1) == 1 and w.stride(0) >= 128):
            w = w.contiguous()
        ws.append(w)
    with _n.on_device(w0.device):
        _n.check(_n.lib.csplat_gnn_edge_mlp3_pack(_n.stream_handle(w0.device), ws[0].data_ptr(), int(ws[0].stride(0)), ws[1].data_ptr(),
                                                  int(ws[1].stride(0)), ws[2].data_ptr(), int(ws[2].stride(0)), _n.ptr(img)),
                 "csplat_gnn_edge_mlp3_pack")
    return img


def edge_mlp3(e0, alpha, xa, ia, xb, ib, image, b0, b1, b2, layer_norm, out=None, e0_absmax=None, agg=None):
    """Inference-only message of one InteractionNetwork layer in ONE launch (csplat_gnn_edge_mlp3, include/csplat.h):
        out = LN( W2 relu( W1 relu( alpha * W0 e0 + b0 + xa[ia] + xb[ib] ) + b1 ) + b2 )
    the two inner [E,128] activations never leave the chip (graph_network.py:178-199).  image = edge_mlp3_pack(W0, W1, W2) under the
    current edge_mlp3_mode; e0_absmax = absmax(e0) (or of a tensor e0 is a slice of), computed here when not handed in (mode 0 only).
    agg = (gp0, pieces) -- rows in destination order (GraphCSR.agg_plan): the messages are not written, their per-run sums land in
    `pieces` [npieces,128] (returned); mode 0 only."""
    _n.require_cuda(e0)
    e0, xa, xb = _f32(e0), _f32(xa), _f32(xb)
    E = e0.shape[0]
    assert e0.shape[1] == 128 and xa.shape[1] == 128 and xb.shape[1] == 128
    ia, ib = ia.contiguous(), ib.contiguous()
    assert ia.dtype == torch.int64 and ib.dtype == torch.int64 and ia.numel() == E and ib.numel() == E
    gp0 = pieces = None
    if agg is not None:
        gp0, pieces = agg
        assert gp0.dtype == torch.int32 and gp0.numel() == (E + 7) // 8 and pieces.dtype == torch.float32 and pieces.shape[1] == 128
    else:
        out = torch.empty_like(e0) if out is None else out
    if e0_absmax is None and E > 0 and edge_mlp3_mode() == 0:
        e0_absmax = absmax(e0)
    c = lambda t: t.detach().contiguous()  # noqa: E731
    with _n.on_device(e0.device):
        _n.check(_n.lib.csplat_gnn_edge_mlp3(_n.stream_handle(e0.device), E, _n.ptr(e0), float(alpha), _n.ptr(e0_absmax), _n.ptr(xa), _n.ptr(ia),
                                             _n.ptr(xb), _n.ptr(ib), _n.ptr(image), _n.ptr(c(b0)), _n.ptr(c(b1)), _n.ptr(c(b2)),
                                             _n.ptr(c(layer_norm.weight)), _n.ptr(c(layer_norm.bias)), float(layer_norm.eps), _n.ptr(out),
                                             _n.ptr(gp0), _n.ptr(pieces)),
                 "csplat_gnn_edge_mlp3")
    return out if agg is None else pieces


def node_update(agg, x, w_agg, w_x, b0, lin2, lin3, layer_norm, w_i_next=None, w_j_next=None):
    """Inference-only node update of one InteractionNetwork layer in ONE launch (csplat_gnn_node_update, include/csplat.h):
        x_new = LN(W3 relu(W2 relu(Wa agg + Wx x + b0) + b2) + b3) + x,   xa' = x_new Wi'^T,   xb' = x_new Wj'^T
    Returns (x_new, xa', xb') (the last two None without next-layer weights).  All operands fp32, 128 wide."""
    _n.require_cuda(x)
    agg, x = _f32(agg), _f32(x)
    N = x.shape[0]
    assert tuple(agg.shape) == (N, 128) and x.shape[1] == 128
    c = lambda t: t.detach().contiguous()  # noqa: E731
    x_new = torch.empty_like(x)
    have_next = w_i_next is not None
    xa = torch.empty_like(x) if have_next else None
    xb = torch.empty_like(x) if have_next else None
    ops = [c(w_agg), c(w_x), c(b0), c(lin2.weight), c(lin2.bias), c(lin3.weight), c(lin3.bias), c(layer_norm.weight),
           c(layer_norm.bias)]
    nxt = [c(w_i_next), c(w_j_next)] if have_next else [None, None]
    with _n.on_device(x.device):
        _n.check(_n.lib.csplat_gnn_node_update(_n.stream_handle(x.device), N, _n.ptr(agg), _n.ptr(x), *[_n.ptr(t) for t in ops],
                                               float(layer_norm.eps), _n.ptr(nxt[0]), _n.ptr(nxt[1]), _n.ptr(x_new), _n.ptr(xa),
                                               _n.ptr(xb)), "csplat_gnn_node_update")
    return x_new, xa, xb


def mlp3_rows(x, image, b0, b1, b2, layer_norm, x_absmax=None):
    """LayerNorm(W2 relu(W1 relu(W0 x + b0) + b1) + b2) for narrow rows x [M, K] (K a multiple of 4, <= 128) in ONE launch
    (csplat_gnn_mlp3_rows: the one-launch edge MLP's kernel without gathers).  image = edge_mlp3_pack(W0 zero-padded to [128,128], W1, W2)
    under edge_mlp3_mode 0; x_absmax = absmax(x), computed here when not handed in."""
    _n.require_cuda(x)
    x = _f32(x)
    M, K = x.shape
    assert K % 4 == 0 and 4 <= K <= 128 and edge_mlp3_mode() == 0
    out = torch.empty(M, 128, dtype=torch.float32, device=x.device)
    if x_absmax is None and M > 0:
        x_absmax = absmax(x)
    c = lambda t: t.detach().contiguous()  # noqa: E731
    with _n.on_device(x.device):
        _n.check(_n.lib.csplat_gnn_mlp3_rows(_n.stream_handle(x.device), M, _n.ptr(x), K, _n.ptr(x_absmax), _n.ptr(image), _n.ptr(c(b0)), _n.ptr(c(b1)),
                                             _n.ptr(c(b2)), _n.ptr(c(layer_norm.weight)), _n.ptr(c(layer_norm.bias)), float(layer_norm.eps), _n.ptr(out)),
                 "csplat_gnn_mlp3_rows")
    return out


def node_update_pack(w_agg, w_x, w2, w3, w_i_next=None, w_j_next=None):
    """the node update's weights (+ the next layer's x_i / x_j blocks) as the register image csplat_gnn_node_update_packed streams: three bf16
    pieces per matrix as MFMA A operands (csplat_gnn_node_update_pack); pack once per weight version"""
    img = torch.empty(int(_n.lib.csplat_gnn_node_update_image_bytes()), dtype=torch.uint8, device=w_agg.device)
    c = lambda t: None if t is None else t.detach().contiguous()  # noqa: E731
    ws = [c(t) for t in (w_agg, w_x, w2, w3, w_i_next, w_j_next)]
    for t in ws[:4]:
        assert tuple(t.shape) == (128, 128) and t.dtype == torch.float32 and t.is_cuda
    with _n.on_device(w_agg.device):
        _n.check(_n.lib.csplat_gnn_node_update_pack(_n.stream_handle(w_agg.device), *[_n.ptr(t) for t in ws], _n.ptr(img)), "csplat_gnn_node_update_pack")
    return img


def rows_chain_pack(mode, w_first, w_second):
    """the register image of csplat_gnn_rows_chain (mode 0: the pair (Wa, Wb); mode 1: the two layers (W0, W1)); pack once per weight version,
    under the edge_mlp3_mode it will be used with"""
    img = torch.empty(int(_n.lib.csplat_gnn_node_update_image_bytes()), dtype=torch.uint8, device=w_first.device)
    a, b = w_first.detach().contiguous(), w_second.detach().contiguous()
    assert tuple(a.shape) == (128, 128) and tuple(b.shape) == (128, 128) and a.dtype == torch.float32 and a.is_cuda
    with _n.on_device(a.device):
        _n.check(_n.lib.csplat_gnn_rows_chain_pack(_n.stream_handle(a.device), int(mode), _n.ptr(a), _n.ptr(b), _n.ptr(img)), "csplat_gnn_rows_chain_pack")
    return img


def rows_chain(x, image, mode, b0=None, b1=None):
    """csplat_gnn_rows_chain (include/csplat.h), no autograd: mode 0 -> (x Wa^T, x Wb^T); mode 1 -> relu(W1 relu(W0 x + b0) + b1)"""
    _n.require_cuda(x)
    x = _f32(x)
    N = x.shape[0]
    assert x.shape[1] == 128
    out_a = torch.empty_like(x)
    out_b = torch.empty_like(x) if mode == 0 else None
    c = lambda t: None if t is None else t.detach().contiguous()  # noqa: E731
    with _n.on_device(x.device):
        _n.check(_n.lib.csplat_gnn_rows_chain(_n.stream_handle(x.device), N, int(mode), _n.ptr(x), _n.ptr(image), _n.ptr(c(b0)), _n.ptr(c(b1)),
                                              _n.ptr(out_a), _n.ptr(out_b)), "csplat_gnn_rows_chain")
    return (out_a, out_b) if mode == 0 else out_a


def node_update_packed(agg, x, image, b0, b2, b3, layer_norm, has_next, piece_ptr=None):
    """node_update() on the pre-packed weights (csplat_gnn_node_update_packed, include/csplat.h): returns (x_new, xa', xb').  piece_ptr
    (int32 [N + 1]): `agg` is the pieces array of edge_mlp3's fused aggregation, a node's aggregate = the sum of its pieces"""
    _n.require_cuda(x)
    agg, x = _f32(agg), _f32(x)
    N = x.shape[0]
    assert agg.shape[1] == 128 and x.shape[1] == 128 and (piece_ptr is not None or agg.shape[0] == N)
    assert piece_ptr is None or (piece_ptr.dtype == torch.int32 and piece_ptr.numel() == N + 1)
    c = lambda t: t.detach().contiguous()  # noqa: E731
    x_new = torch.empty_like(x)
    xa = torch.empty_like(x) if has_next else None
    xb = torch.empty_like(x) if has_next else None
    with _n.on_device(x.device):
        _n.check(_n.lib.csplat_gnn_node_update_packed(_n.stream_handle(x.device), N, _n.ptr(agg), _n.ptr(x), _n.ptr(image), _n.ptr(c(b0)), _n.ptr(c(b2)),
                                                      _n.ptr(c(b3)), _n.ptr(c(layer_norm.weight)), _n.ptr(c(layer_norm.bias)), float(layer_norm.eps),
                                                      int(bool(has_next)), _n.ptr(x_new), _n.ptr(xa), _n.ptr(xb), _n.ptr(piece_ptr)), "csplat_gnn_node_update_packed")
    return x_new, xa, xb


def ln128_fwd(x, gamma, beta, eps):
    """(LayerNorm(x), per-row (mean, rstd)) for [M, 128] fp32 rows: csplat_ln128_fwd"""
    M = x.shape[0]
    y = torch.empty_like(x)
    stats = torch.empty(M, 2, dtype=torch.float32, device=x.device)
    with _n.on_device(x.device):
        _n.check(_n.lib.csplat_ln128_fwd(_n.stream_handle(x.device), M, _n.ptr(x), _n.ptr(gamma), _n.ptr(beta), float(eps), _n.ptr(y),
                                         _n.ptr(stats)), "csplat_ln128_fwd")
    return y, stats


def ln128_bwd(g, x, stats, gamma, want_dxsum=False, g_rows=None, x_normalized=False):
    """(dx, dgamma, dbeta, column sums of dx or None) of LayerNorm over [M, 128] rows: csplat_ln128_bwd.  g_rows: row r of the incoming
    gradient is g[g_rows[r]] (g then has as many rows as g_rows addresses); x_normalized: x is the normalised row xhat itself."""
    M = x.shape[0]
    dx = torch.empty_like(x)
    dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(gamma)
    dxsum = torch.empty_like(gamma) if want_dxsum else None
    part = torch.empty(3 * int(_n.lib.csplat_ln128_partial_floats(M)), dtype=torch.float32, device=x.device)
    with _n.on_device(x.device):
        _n.check(_n.lib.csplat_ln128_bwd(_n.stream_handle(x.device), M, _n.ptr(g), _n.ptr(x), _n.ptr(stats), _n.ptr(gamma), _n.ptr(dx),
                                         _n.ptr(dgamma), _n.ptr(dbeta), _n.ptr(dxsum), _n.ptr(g_rows), int(x_normalized), _n.ptr(part)),
                 "csplat_ln128_bwd")
    return dx, dgamma, dbeta, dxsum


class LayerNorm128(torch.autograd.Function):
    """nn.LayerNorm(128) on [M, 128] fp32 rows under autograd: one HBM pass forward (csplat_ln128_fwd, keeps mean / rstd per row),
    one backward (csplat_ln128_bwd: input gradient + deterministic gamma / beta gradients)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x, gamma, beta = _f32(x), _f32(gamma), _f32(beta)
        y, stats = ln128_fwd(x, gamma, beta, eps)
        ctx.save_for_backward(x, stats, gamma)
        return y

    @staticmethod
    def backward(ctx, g):
        x, stats, gamma = ctx.saved_tensors
        dx, dgamma, dbeta, _ = ln128_bwd(_f32(g), x, stats, gamma)
        return dx, dgamma, dbeta, None


def layer_norm_rows(x, ln: torch.nn.LayerNorm):
    """ln(x); 128-wide fp32 GPU rows go through LayerNorm128 (same parameters, same state_dict)"""
    if x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[1] == 128 and tuple(ln.normalized_shape) == (128,) and \
            ln.elementwise_affine and ln.bias is not None and torch.is_grad_enabled() and x.shape[0] > 0:
        return LayerNorm128.apply(x, ln.weight, ln.bias, ln.eps)
    if torch.is_grad_enabled() and x.dim() == 2 and x.shape[0] > 0:
        _n.composed_fallback("graph_ops.layer_norm_rows", "dtype" if x.dtype != torch.float32 else "shape", x)
    return ln(x)


def relu_mask_bias128(g, out):
    """(g masked by out > 0, column sums of the masked g): ReLU backward + bias gradient in one pass; out None = no mask"""
    g = _f32(g)
    M = g.shape[0]
    gm = torch.empty_like(g) if out is not None else None
    db = torch.empty(128, dtype=torch.float32, device=g.device)
    part = torch.empty(int(_n.lib.csplat_ln128_partial_floats(M)), dtype=torch.float32, device=g.device)
    with _n.on_device(g.device):
        _n.check(_n.lib.csplat_relu_mask_bias128(_n.stream_handle(g.device), M, _n.ptr(g), _n.ptr(out), _n.ptr(gm), _n.ptr(db), _n.ptr(part)),
                 "csplat_relu_mask_bias128")
    return (gm if gm is not None else g), db


def dw128(g, x, bias=False, x_relu=False):
    """g^T @ x for [M, 128] fp32 rows -> [128, 128] (csplat_dw128: fp32 MFMA on the row-major operands, deterministic split over rows);
    bias=True: (dW, column sums of g); x_relu=True: max(x, 0) in place of x"""
    g, x = _f32(g), _f32(x)
    M = g.shape[0]
    dW = torch.empty(128, 128, dtype=torch.float32, device=g.device)
    db = torch.empty(128, dtype=torch.float32, device=g.device) if bias else None
    ws = torch.empty(max(int(_n.lib.csplat_dw128_workspace_bytes(M)), 256), dtype=torch.uint8, device=g.device)
    with _n.on_device(g.device):
        _n.check(_n.lib.csplat_dw128_bias(_n.stream_handle(g.device), M, _n.ptr(g), _n.ptr(x), int(x_relu), _n.ptr(dW), _n.ptr(db), _n.ptr(ws)),
                 "csplat_dw128_bias")
    return (dW, db) if bias else dW


class SplitKLinear(torch.autograd.Function):
    """y = relu?(x @ weight^T + bias) for 128 -> 128 fp32 layers under autograd (edge level: rows = E ~ 3e5; node level: N ~ 1e4).
    Tall inputs (>= BIG_ROWS) run through csplat_linear128 both ways (forward with bias / ReLU in the epilogue, input gradient as
    the same kernel on the transposed weight): 76 us against 142 us + a ReLU pass for the library call; shorter ones keep the
    library GEMM for y and dx.  The weight gradient g^T @ x is a [128 x M] x [M x 128] product reduced over the ROWS: the library
    call reduces inside 16 workgroups (630 us at M = 3e5, 56 us at M = 1e4) -- csplat_dw128 splits the rows over the chip at every
    M.  Other shapes / dtypes: plain torch, with the weight gradient as a chunked batched GEMM."""

    CHUNK = 3072
    BIG_ROWS = 16384       # (measured again in round 4 with the node level on csplat_linear128 too -- BIG_ROWS = 64: train step 24.3 against 22.4 ms)

    @staticmethod
    def _fast(x, weight):
        return x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 2 and \
            x.shape[1] == 128 and tuple(weight.shape) == (128, 128)

    @staticmethod
    def forward(ctx, x, weight, bias, relu=False):
        kern = SplitKLinear._fast(x, weight)
        fast = kern and x.shape[0] >= SplitKLinear.BIG_ROWS
        if fast:
            out = linear128(x, weight, bias, relu=relu)
        else:
            out = torch.addmm(bias, x, weight.t()) if bias is not None else x @ weight.t()
            if relu:
                out = out.relu_()
        ctx.save_for_backward(x, weight, out if relu else None)
        ctx.has_bias, ctx.relu, ctx.fast, ctx.kern = bias is not None, bool(relu), fast, kern
        return out

    @staticmethod
    def backward(ctx, g):
        x, weight, out = ctx.saved_tensors
        g = g.contiguous()
        db = None
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        kern = ctx.kern and g.dtype == torch.float32 and g.shape[0] > 0
        if kern and (ctx.relu or want_db):
            g, db = relu_mask_bias128(g, out if ctx.relu else None)      # ReLU backward + bias gradient: one pass
        elif ctx.relu:
            g = torch.ops.aten.threshold_backward(g, out, 0)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = linear128(g, weight.t()) if ctx.fast else g @ weight
        dw = None
        if ctx.needs_input_grad[1] and kern:
            dw = dw128(g, x.contiguous())
        elif ctx.needs_input_grad[1]:
            M, C = x.shape[0], x.shape[0] // SplitKLinear.CHUNK
            m0 = C * SplitKLinear.CHUNK
            xc = x.contiguous()
            dw = torch.bmm(g[:m0].view(C, SplitKLinear.CHUNK, -1).transpose(1, 2),
                           xc[:m0].view(C, SplitKLinear.CHUNK, -1)).sum(0) if C else torch.zeros_like(weight)
            if m0 < M:
                dw = dw + g[m0:].t() @ xc[m0:]
        if want_db and db is None:
            db = g.sum(0)
        return dx, dw, (db if want_db else None), None


class EdgeTailAggregate(torch.autograd.Function):
    """S = segment_sum_dst( normalise( Linear_k( relu(... relu(Linear_1(a0)) ...) ) ) ): everything of an InteractionNetwork's message
    path behind the first (split) edge Linear, as ONE autograd node over [E, 128] fp32 rows (graph_network.py:139-150 +
    aggr='add').  `normalise` is the LayerNorm WITHOUT its affine part: sum_e (gamma * xhat_e + beta) = gamma * S + deg * beta is
    applied by the caller on [N, 128] rows, where autograd also finds dgamma and dbeta (edge_tail_aggregate()).
    What one node buys, per layer of the processor:
      * the normalisation is the epilogue of the last Linear's GEMM (csplat_linear128_ex, which also leaves each row's rstd): no
        separate [E, 128] LayerNorm pass, and only xhat is kept for the backward;
      * the ReLU backward of every hidden layer is the mask epilogue of the GEMM that produces its incoming gradient (mask = the
        layer's saved output) -- no separate [E, 128] masking pass;
      * bias gradients are column sums taken where the rows are read anyway: inside csplat_dw128_bias for the hidden layers, inside
        csplat_ln128_bwd (dxsum) for the last one;
      * the backward of the segmented sum (every edge takes its destination node's row) is a gathered read inside csplat_ln128_bwd,
        not an [E, 128] copy;
      * transposed weights are read in place.
    a0_relu: a0 is itself the output of a ReLU whose producer expects the gradient of its PRE-activation (EdgeFirstLayer, or EdgeCombine
    with grad_premasked=True)."""

    @staticmethod
    def forward(ctx, a0, csr, eps, a0_relu, *wb):
        ctx.set_materialize_grads(False)
        k = len(wb) // 2
        acts = [_f32(a0)]
        for i in range(k - 1):
            acts.append(linear128(acts[-1], wb[2 * i], wb[2 * i + 1], relu=True))
        E = acts[0].shape[0]
        stats = torch.empty(E, 2, dtype=torch.float32, device=a0.device)
        unit = _unit_ln(a0.device, eps)
        xhat = linear128(acts[-1], wb[2 * k - 2], wb[2 * k - 1], layer_norm=unit, ln_stats=stats)
        S = torch.empty(csr.N, 128, dtype=torch.float32, device=a0.device)
        with _n.on_device(a0.device):
            _n.check(_n.lib.csplat_gnn_segment_sum(_n.stream_handle(a0.device), csr.N, E, 128, _n.ptr(xhat),
                                                   _n.ptr(csr.rowptr["dst"]), _n.ptr(csr.perm["dst"]), _n.ptr(S)),
                     "csplat_gnn_segment_sum")
        ctx.csr, ctx.k, ctx.a0_relu, ctx.unit = csr, k, bool(a0_relu), unit
        ctx.save_for_backward(stats, xhat, *acts, *wb[0::2])
        return S

    @staticmethod
    def backward(ctx, g_S):
        k, csr = ctx.k, ctx.csr
        saved = ctx.saved_tensors
        stats, xhat, acts, weights = saved[0], saved[1], saved[2:2 + k], saved[2 + k:]
        if g_S is None:
            return (None,) * (4 + 2 * k)
        d, _, _, db_last = ln128_bwd(_f32(g_S), xhat, stats, ctx.unit.weight, want_dxsum=True, g_rows=csr.ei[1], x_normalized=True)
        grads = [None] * (2 * k)
        for i in range(k - 1, -1, -1):                     # Linear i + 1 maps acts[i] to the next activation
            if i == k - 1:
                grads[2 * i], grads[2 * i + 1] = dw128(d, acts[i]), db_last
            else:
                grads[2 * i], grads[2 * i + 1] = dw128(d, acts[i], bias=True)
            if i > 0 or ctx.needs_input_grad[0]:
                d = linear128(d, weights[i].t(), mask=acts[i] if (i > 0 or ctx.a0_relu) else None)
        return (d if ctx.needs_input_grad[0] else None, None, None, None, *grads)


class EdgeFirstLayer(torch.autograd.Function):
    """(a0, e_next) = (relu(scale * e @ weight^T + xa[dst] + xb[src]), e): the first edge Linear of an InteractionNetwork with the
    [x_i, x_j, e] concat folded away (xa / xb: the node-level products of the x_i / x_j column blocks, bias included) -- gathers, sum and
    ReLU are the epilogue of the e-block GEMM, so the [E, 128] product is never written or re-read on its own.  e is chained through
    the layers like in EdgeLatentLinear.  backward() expects the gradient of the PRE-activation (EdgeTailAggregate(a0_relu=True))."""

    @staticmethod
    def forward(ctx, e, weight, scale, xa, xb, csr):
        ctx.set_materialize_grads(False)
        a0 = linear128(e, weight, alpha=float(scale), relu=True, gather=(_f32(xa), csr.ei[1], _f32(xb), csr.ei[0]))
        ctx.save_for_backward(e, weight)
        ctx.scale, ctx.csr = float(scale), csr
        return a0, e.view_as(e)

    @staticmethod
    def backward(ctx, g, g_next):
        e, weight = ctx.saved_tensors
        csr = ctx.csr
        de = dw = dxa = dxb = None
        if g is not None:
            g = _f32(g)
            E = g.shape[0]
            dxa = torch.empty(csr.N, 128, dtype=torch.float32, device=g.device)
            dxb = torch.empty_like(dxa)
            with _n.on_device(g.device):
                _n.check(_n.lib.csplat_gnn_edge_combine_bwd(
                    _n.stream_handle(g.device), csr.N, E, 128, _n.ptr(g), None, 0, _n.ptr(csr.rowptr["dst"]),
                    _n.ptr(csr.perm["dst"]), _n.ptr(csr.rowptr["src"]), _n.ptr(csr.perm["src"]), None, _n.ptr(dxa),
                    _n.ptr(dxb)), "csplat_gnn_edge_combine_bwd")
            if ctx.needs_input_grad[0]:
                de = linear128(g, weight.t(), alpha=ctx.scale, add_post=g_next)
            if ctx.needs_input_grad[1]:
                dw = dw128(g, e)
                if ctx.scale != 1.0:
                    dw = dw * ctx.scale
        elif ctx.needs_input_grad[0]:
            de = g_next
        return de, dw, None, dxa, dxb, None


def edge_tail_ok(a0, seq) -> bool:
    """[build_mlp(...), LayerNorm] whose layers behind the first are 128 -> 128 fp32 Linear (+ ReLU, Identity last) with biases, on
    tall GPU rows under autograd: the shape EdgeTailAggregate covers"""
    mlp, ln = seq[0], seq[1]
    mods = list(mlp.children())
    lins, acts = mods[2::2], mods[3::2]
    return torch.is_grad_enabled() and a0.is_cuda and a0.dtype == torch.float32 and a0.dim() == 2 and a0.shape[1] == 128 and \
        a0.shape[0] >= SplitKLinear.BIG_ROWS and len(lins) >= 1 and isinstance(ln, torch.nn.LayerNorm) and \
        tuple(ln.normalized_shape) == (128,) and ln.elementwise_affine and ln.bias is not None and \
        all(isinstance(m, torch.nn.Linear) and tuple(m.weight.shape) == (128, 128) and m.bias is not None and
            m.weight.dtype == torch.float32 for m in lins) and \
        all(isinstance(a, torch.nn.ReLU) for a in acts[:-1]) and isinstance(acts[-1], torch.nn.Identity) and \
        isinstance(mods[1], torch.nn.ReLU)


def report_missed_edge_tail(a0):
    """InteractionNetwork.message_update took the per-layer path for TALL rows (the fused nodes cover 128-wide fp32 MLPs only)"""
    if torch.is_grad_enabled() and a0.dim() == 2 and a0.shape[0] >= SplitKLinear.BIG_ROWS:
        _n.composed_fallback("graph_network.message_update", "dtype" if a0.dtype != torch.float32 else "shape", a0)


def edge_tail_aggregate(a0, csr, seq, a0_relu=True):
    """sum over destination nodes of seq's output, seq = [build_mlp, LayerNorm] entered behind its first Linear (+ ReLU):
    EdgeTailAggregate for the [E, 128] work, the LayerNorm's affine part on the [N, 128] sums"""
    lins = list(seq[0].children())[2::2]
    wb = [t for m in lins for t in (m.weight, m.bias)]
    S = EdgeTailAggregate.apply(a0, csr, float(seq[1].eps), a0_relu, *wb)
    return S * seq[1].weight + csr.deg_dst[:, None] * seq[1].bias


class EdgeLatentLinear(torch.autograd.Function):
    """(ec, e_next) = (scale * e @ weight^T, e) for the edge latents e [E, 128] that every processor layer multiplies by its own
    weight block (graph_network.py: the layers only ever double their edge features, so all of them read the encoder's output).
    Chaining e through the layers makes the gradient of e a running sum that each layer's backward extends INSIDE its input-gradient
    GEMM (csplat_linear128 with the sum so far as the epilogue addend): 15 GEMMs instead of 15 GEMMs + 14 [E, 128] additions."""

    @staticmethod
    def forward(ctx, e, weight, scale):
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(e, weight)
        ctx.scale = float(scale)
        return linear128(e, weight, alpha=ctx.scale), e.view_as(e)

    @staticmethod
    def backward(ctx, g, g_next):
        e, weight = ctx.saved_tensors
        de = dw = None
        if g is not None:
            g = g.contiguous()
            if ctx.needs_input_grad[0]:
                de = linear128(g, weight.t(), alpha=ctx.scale, add_post=g_next)
            if ctx.needs_input_grad[1]:
                dw = dw128(g, e)
                if ctx.scale != 1.0:
                    dw = dw * ctx.scale
        elif ctx.needs_input_grad[0]:
            de = g_next
        return de, dw, None


def edge_latent_linear(e, weight, scale: float = 1.0):
    """(scale * e @ weight^T, e to hand to the next layer): EdgeLatentLinear for tall fp32 GPU rows under autograd, plain torch otherwise"""
    if torch.is_grad_enabled() and (weight.requires_grad or e.requires_grad) and e.shape[0] >= SplitKLinear.BIG_ROWS and \
            SplitKLinear._fast(e, weight):
        return EdgeLatentLinear.apply(e, weight, scale)
    return linear_rows(e, weight if scale == 1.0 else weight * scale, None), e


def linear_rows(x, lin_weight, lin_bias, min_rows: int = 16384, relu: bool = False):
    """nn.Linear (+ ReLU) forward that switches to SplitKLinear while a graph is being recorded: for tall inputs of any width, and
    for 128 -> 128 fp32 GPU layers from 512 rows up (the library's weight-gradient GEMM reduces the rows inside 16 workgroups)."""
    rows = x.shape[0]
    if torch.is_grad_enabled() and (lin_weight.requires_grad or x.requires_grad) and \
            (rows >= min_rows or (rows >= 512 and lin_weight.requires_grad and SplitKLinear._fast(x, lin_weight))):
        return SplitKLinear.apply(x, lin_weight, lin_bias, relu)
    y = torch.nn.functional.linear(x, lin_weight, lin_bias)
    return y.relu() if relu else y
