"""save / load shared by MeshSimulator and ClothMeshSimulator (meshnet_network.py:193-252, cloth_network.py:195-254):
checkpoint dict {'model': state_dict, '_output_normalizer': vars, '_node_normalizer': vars}; load(path, 'latest')
picks model-<STEP>.pt with the highest STEP."""
import glob
import os
import re

import torch
import torch.nn as nn


class _SimulatorIO(nn.Module):
    def save(self, path=None):
        torch.save({'model': self.state_dict(), '_output_normalizer': self._output_normalizer.get_variable(),
                    '_node_normalizer': self._node_normalizer.get_variable()}, path)

    def load(self, path: str, file='latest'):
        if file == "latest":
            fnames = glob.glob(os.path.join(path, '*model*pt'))
            if len(fnames) == 0:
                raise ValueError(f"Did not find any pre-trained weights for the meshnet in: {path}")
            expr = re.compile(r'.*model-(\d+).pt')
            steps = [int(m.groups()[0]) for m in map(expr.search, fnames) if m]
            model_file = os.path.join(path, f'model-{max(steps + [0])}.pt')
        else:
            model_file = os.path.join(path, file)
        dicts = torch.load(model_file, weights_only=False)
        self.load_state_dict(dicts["model"])
        for k, v in dicts.items():
            if k == 'model':
                continue
            target = getattr(self, k)
            for para, value in v.items():
                setattr(target, para, value)
        print("Simulator model loaded checkpoint %s" % model_file)
