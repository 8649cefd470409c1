"""Parameter containers and host-PyTorch layers of the hot path.

``ResnetBlockFC`` keeps the reference's parameter names (fc_0, fc_1, shortcut;
reference src/layers.py:8-50) so checkpoints load unchanged.  Inside the decoder
its arithmetic runs in the fused HIP kernel (vtaco_amd/csrc/decode.hip); the
``forward`` here is the plain PyTorch-ROCm path used by the PointNet encoder's
tiny per-point MLP (SURVEY.md K4, 0.16 GFLOP/scene, stays host PyTorch).
"""
from __future__ import annotations

import torch
from torch import nn
from torch.nn import functional as F


class ResnetBlockFC(nn.Module):
    """x -> shortcut(x) + fc_1(relu(fc_0(relu(x)))); fc_1.weight starts at zero."""

    def __init__(self, size_in, size_out=None, size_h=None):
        super().__init__()
        size_out = size_in if size_out is None else size_out
        size_h = min(size_in, size_out) if size_h is None else size_h
        self.size_in, self.size_h, self.size_out = size_in, size_h, size_out
        self.fc_0 = nn.Linear(size_in, size_h)
        self.fc_1 = nn.Linear(size_h, size_out)
        self.shortcut = None if size_in == size_out else nn.Linear(size_in, size_out, bias=False)
        nn.init.zeros_(self.fc_1.weight)

    def forward(self, x):
        dx = self.fc_1(F.relu(self.fc_0(F.relu(x))))
        return (x if self.shortcut is None else self.shortcut(x)) + dx

    def packed(self):
        """(fc_0.weight, fc_0.bias, fc_1.weight, fc_1.bias) for vt_decoder_pack."""
        return self.fc_0.weight, self.fc_0.bias, self.fc_1.weight, self.fc_1.bias
