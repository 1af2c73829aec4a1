"""Drop-in for /root/reference/meshnet/cloth_network.py: ClothMeshSimulator (velocity history + node type ->
acceleration) on top of the HIP-backed EncodeProcessDecode.  Same ctor / method signatures and checkpoint format."""
import torch
import torch.nn as nn

from meshnet._simbase import _SimulatorIO
from meshnet.graph_network import EncodeProcessDecode
from meshnet.model_utils import IdentityNormalizer, Normalizer


class ClothMeshSimulator(_SimulatorIO):
    def __init__(self, simulation_dimensions: int, nnode_in: int, nedge_in: int, latent_dim: int,
                 nmessage_passing_steps: int, nmlp_layers: int, mlp_hidden_dim: int, nnode_types: int,
                 node_type_embedding_size: int, normalize: bool = False, device="cpu"):
        super().__init__()
        self._nnode_types = nnode_types
        self._node_type_embedding_size = node_type_embedding_size
        self._encode_process_decode = EncodeProcessDecode(
            nnode_in_features=nnode_in, nnode_out_features=simulation_dimensions, nedge_in_features=nedge_in,
            latent_dim=latent_dim, nmessage_passing_steps=nmessage_passing_steps, nmlp_layers=nmlp_layers,
            mlp_hidden_dim=mlp_hidden_dim).to(device)
        if normalize:
            self._output_normalizer = Normalizer(size=simulation_dimensions, name='output_normalizer', device=device)
            self._node_normalizer = Normalizer(size=nnode_in, name='node_normalizer', device=device)
        else:  # cloth_network.py:64-65 (both identity normalisers are built with size=simulation_dimensions)
            self._output_normalizer = IdentityNormalizer(size=simulation_dimensions, name='output_normalizer', device=device)
            self._node_normalizer = IdentityNormalizer(size=simulation_dimensions, name='output_normalizer', device=device)
        self._device = device

    def forward(self):
        pass

    def _encoder_preprocessor(self, velocity: torch.Tensor, node_type: torch.Tensor, velocity_noise: torch.Tensor = None):
        """cloth_network.py:72-110: cat(velocity (+noise), one_hot(node_type)) -> node normaliser."""
        v = velocity if velocity_noise is None else velocity + velocity_noise
        onehot = nn.functional.one_hot(torch.squeeze(node_type.long()), self._node_type_embedding_size)
        feats = torch.cat([v, onehot], dim=1)
        return self._node_normalizer(feats, self.training)

    def predict_acceleration(self, velocity, node_type, edge_index, edge_features, target_velocities=None,
                             velocity_noise=None):
        """cloth_network.py:112-161 -> (predicted normalised acceleration, normalised target or None)."""
        feats = self._encoder_preprocessor(velocity, node_type, velocity_noise)
        pred = self._encode_process_decode(feats.to(torch.float32), edge_index, edge_features)
        if target_velocities is None:
            return pred, None
        base = velocity if velocity_noise is None else velocity + velocity_noise
        target_acc = target_velocities - base[:, -3:]
        return pred, self._output_normalizer(target_acc, self.training)

    def predict_velocity(self, velocities, node_type, edge_index, edge_features):
        """cloth_network.py:163-193: rollout step, v_next = v[:, -3:] + de-normalised acceleration."""
        feats = self._encoder_preprocessor(velocities, node_type, velocity_noise=None)
        acc = self._output_normalizer.inverse(self._encode_process_decode(feats, edge_index, edge_features))
        return velocities[:, -3:] + acc
