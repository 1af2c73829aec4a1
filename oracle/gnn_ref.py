"""oracle/gnn_ref.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy restatement of the MeshNet encode-process-decode network, following
  /root/reference/meshnet/graph_network.py:7-45   build_mlp (Linear "NN-i" + ReLU, Identity on the last)
  /root/reference/meshnet/graph_network.py:48-111 Encoder (node / edge MLP + LayerNorm)
  /root/reference/meshnet/graph_network.py:114-222 InteractionNetwork: message = LN(MLP(cat[x_i, x_j, e])),
        aggr='add' over edge_index[1], update = LN(MLP(cat[agg, x])), residuals
  /root/reference/meshnet/graph_network.py:225-292 Processor, :295-332 Decoder, :335-408 EncodeProcessDecode
  /root/reference/meshnet/model_utils.py:40-96     Normalizer
  /root/reference/meshnet/cloth_network.py:72-193  ClothMeshSimulator feature assembly
PyG semantics (torch_geometric is a pip dependency, un-pinned, README.md:29-30):
  x_j = x[edge_index[0]], x_i = x[edge_index[1]]; sum over edge_index[1] with dim_size N;
  update() receives the ORIGINAL propagate kwargs, so InteractionNetwork returns the INPUT edge features and each
  layer outputs edge_latent_out = 2 * edge_latent_in.
Pinned by tests/golden/gnn.npz (reference modules run under a MessagePassing shim: "shim-derived").
Parameters are passed as a dict of numpy arrays keyed exactly like the reference state_dict.
"""
import numpy as np


def _mlp(p, prefix, x):
    n = 0
    while f"{prefix}.NN-{n}.weight" in p:
        n += 1
    for i in range(n):
        x = x @ p[f"{prefix}.NN-{i}.weight"].T.astype(x.dtype) + p[f"{prefix}.NN-{i}.bias"].astype(x.dtype)
        if i < n - 1:
            x = np.maximum(x, 0)
    return x


def _ln(p, prefix, x, eps=1e-5):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * p[prefix + ".weight"].astype(x.dtype) + p[prefix + ".bias"].astype(x.dtype)


def _mlp_ln(p, prefix, x):
    return _ln(p, prefix + ".1", _mlp(p, prefix + ".0", x))


def interaction(p, prefix, x, ei, e):
    x_j, x_i = x[ei[0]], x[ei[1]]
    m = _mlp_ln(p, prefix + ".edge_fn", np.concatenate([x_i, x_j, e], -1))
    agg = np.zeros((x.shape[0], m.shape[1]), x.dtype)
    np.add.at(agg, ei[1], m)
    xu = _mlp_ln(p, prefix + ".node_fn", np.concatenate([agg, x], -1))
    return xu + x, e + e


def encode_process_decode(p, x, ei, ef, dtype=np.float64):
    x = np.asarray(x, dtype); ef = np.asarray(ef, dtype); ei = np.asarray(ei, np.int64)
    p = {k: np.asarray(v, dtype) for k, v in p.items()}
    h = _mlp_ln(p, "_encoder.node_fn", x)
    e = _mlp_ln(p, "_encoder.edge_fn", ef)
    k = 0
    while f"_processor.gnn_stacks.{k}.node_fn.0.NN-0.weight" in p:
        h, e = interaction(p, f"_processor.gnn_stacks.{k}", h, ei, e)
        k += 1
    return _mlp(p, "_decoder.node_fn", h)


class Normalizer:
    """meshnet/model_utils.py:40-96 (online mean/std accumulated while training)."""

    def __init__(self, size, std_epsilon=1e-8, max_accumulations=10 ** 6):
        self.acc_count = 0.0; self.num = 0.0
        self.acc_sum = np.zeros((1, size)); self.acc_sq = np.zeros((1, size))
        self.eps = std_epsilon; self.max_acc = max_accumulations

    def __call__(self, x, accumulate=True):
        if accumulate and self.num < self.max_acc:
            self.acc_sum += x.sum(0, keepdims=True); self.acc_sq += (x ** 2).sum(0, keepdims=True)
            self.acc_count += x.shape[0]; self.num += 1
        return (x - self.mean()) / self.std()

    def inverse(self, x):
        return x * self.std() + self.mean()

    def mean(self):
        return self.acc_sum / max(self.acc_count, 1.0)

    def std(self):
        c = max(self.acc_count, 1.0)
        return np.maximum(np.sqrt(self.acc_sq / c - self.mean() ** 2), self.eps)
