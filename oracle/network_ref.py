"""oracle/network_ref.py -- TEST INFRASTRUCTURE ONLY.

CPU (torch, fp32) restatement of the reference sigma-MLP, src/network/network.py:6-58:
Linear stack with a skip concat ([encoded_input, x]) in front of the layers listed in `skips`,
LeakyReLU(0.01) between layers, sigmoid / leaky-relu / tanh / identity at the end.
Pinned against golden vectors captured from the imported reference class (tests/golden/make_golden.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def mlp_forward(feat, weights, biases, skips=(2,), last_activation="sigmoid"):
    """Functional form: feat [B, in_dim]; weights[i] is [out_i, in_i] like nn.Linear.weight."""
    x = feat
    n = len(weights)
    for i, (w, b) in enumerate(zip(weights, biases)):
        if i in skips:
            x = torch.cat([feat, x], -1)
        x = F.linear(x, w, b)
        if i < n - 1:
            x = F.leaky_relu(x, 0.01)
        elif last_activation == "sigmoid":
            x = torch.sigmoid(x)
        elif last_activation == "relu":
            x = F.leaky_relu(x, 0.01)
        elif last_activation == "tanh":
            x = torch.tanh(x)
        elif last_activation != "none":
            raise NotImplementedError("Unknown last activation")
    return x


class DensityNetworkRef(torch.nn.Module):
    def __init__(self, encoder, bound=0.2, num_layers=8, hidden_dim=256, skips=(4,), out_dim=1,
                 last_activation="sigmoid"):
        super().__init__()
        self.encoder, self.bound, self.skips = encoder, bound, tuple(skips)
        self.in_dim = encoder.output_dim
        self.last_activation = last_activation
        dims_in = [self.in_dim] + [hidden_dim + (self.in_dim if i in self.skips else 0)
                                   for i in range(1, num_layers - 1)] + [hidden_dim]
        dims_out = [hidden_dim] * (num_layers - 1) + [out_dim]
        self.layers = torch.nn.ModuleList([torch.nn.Linear(i, o) for i, o in zip(dims_in, dims_out)])

    def forward(self, x):
        feat = self.encoder(x, self.bound)
        return mlp_forward(feat, [l.weight for l in self.layers], [l.bias for l in self.layers],
                           self.skips, self.last_activation)
