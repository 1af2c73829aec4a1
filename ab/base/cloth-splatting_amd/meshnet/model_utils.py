"""Drop-in for the hot-path part of /root/reference/meshnet/model_utils.py: NodeType, Normalizer (online
mean / std statistics, :40-96), IdentityNormalizer (:16-37).  The PyG `Data` graph builders of that file are
CPU-side data plumbing and out of scope."""
import enum

import torch
import torch.nn as nn


class NodeType(enum.IntEnum):
    CLOTH = 0


class IdentityNormalizer(nn.Module):
    def __init__(self, size, name='IdentityNormalizer', device='cuda'):
        super().__init__()
        self.name = name
        self._size = size
        self._mean = torch.zeros((1, size), dtype=torch.float32, device=device)
        self._std = torch.ones((1, size), dtype=torch.float32, device=device)

    def forward(self, batched_data, accumulate=True):
        return batched_data

    def inverse(self, normalized_batch_data):
        return normalized_batch_data

    def get_variable(self):
        return {'_size': self._size, '_mean': self._mean, '_std': self._std, 'name': self.name}


class Normalizer(nn.Module):
    """Running sum / sum-of-squares statistics kept as plain tensors (NOT buffers: the reference saves them
    through get_variable(), cloth_network.py:209-213, and they must stay out of state_dict)."""

    def __init__(self, size, max_accumulations=10 ** 6, std_epsilon=1e-8, name='Normalizer', device='cuda'):
        super().__init__()
        self.name = name
        self._max_accumulations = max_accumulations
        f = dict(dtype=torch.float32, device=device)
        self._std_epsilon = torch.tensor(std_epsilon, **f)
        self._acc_count = torch.tensor(0, **f)
        self._num_accumulations = torch.tensor(0, **f)
        self._acc_sum = torch.zeros((1, size), **f)
        self._acc_sum_squared = torch.zeros((1, size), **f)

    def forward(self, batched_data, accumulate=True):
        if accumulate and self._num_accumulations < self._max_accumulations:
            self._accumulate(batched_data.detach())
        return (batched_data - self._mean()) / self._std_with_epsilon()

    def inverse(self, normalized_batch_data):
        return normalized_batch_data * self._std_with_epsilon() + self._mean()

    def _accumulate(self, batched_data):
        self._acc_sum += batched_data.sum(dim=0, keepdim=True)
        self._acc_sum_squared += (batched_data ** 2).sum(dim=0, keepdim=True)
        self._acc_count += batched_data.shape[0]
        self._num_accumulations += 1

    def _safe_count(self):
        return torch.clamp_min(self._acc_count, 1.0)

    def _mean(self):
        return self._acc_sum / self._safe_count()

    def _std_with_epsilon(self):
        std = torch.sqrt(self._acc_sum_squared / self._safe_count() - self._mean() ** 2)
        return torch.maximum(std, self._std_epsilon)

    def get_variable(self):
        return {'_max_accumulations': self._max_accumulations, '_std_epsilon': self._std_epsilon,
                '_acc_count': self._acc_count, '_num_accumulations': self._num_accumulations,
                '_acc_sum': self._acc_sum, '_acc_sum_squared': self._acc_sum_squared, 'name': self.name}


def get_velocity_noise(graph, noise_std, input_sequence_length, device):
    """model_utils.py:98-105"""
    shape = graph.x[:, 0:3 * input_sequence_length].shape
    return torch.normal(std=noise_std, mean=0.0, size=shape).to(device)


def optimizer_to(optim, device):
    """model_utils.py:178-195"""
    for st in optim.state.values():
        items = [st] if isinstance(st, torch.Tensor) else list(st.values()) if isinstance(st, dict) else []
        for t in items:
            if isinstance(t, torch.Tensor):
                t.data = t.data.to(device)
                if t._grad is not None:
                    t._grad.data = t._grad.data.to(device)
