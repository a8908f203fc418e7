"""Sigma-MLP: host-side mirror of reference src/network/network.py and src/network/__init__.py.

`DensityNetwork(encoder, bound, num_layers, hidden_dim, skips, out_dim, last_activation)` keeps the reference's
constructor, attributes (`.bound`, `.encoder`, `.layers`, `.activations`) and state-dict keys
(`encoder.embeddings`, `layers.{i}.weight|bias`) so reference checkpoints load unchanged.

Two execution paths, both on the GPU:
  * canonical NAF shape (in 32, hidden 32, 4 layers, skip before layer 2, 1 output; config/*.yaml:7-19):
    `naf_field_forward` / the fused renderer -- hash gather + MFMA MLP in libnaf_hip.so;
  * any other shape: HIP hash encoder + `nn.Linear` (rocBLAS GEMMs), exactly the reference's composition.
"""
from __future__ import annotations

import torch
import torch.nn as nn

LAST_ACTIVATIONS = {"sigmoid": 0, "relu": 1, "tanh": 2, "none": 3}


class DensityNetwork(nn.Module):
    def __init__(self, encoder, bound=0.2, num_layers=8, hidden_dim=256, skips=[4], out_dim=1,
                 last_activation="sigmoid"):
        super().__init__()
        self.nunm_layers = num_layers          # (sic) attribute name of the reference, network.py:8
        self.num_layers = num_layers
        self.hidden_dim = hidden_dim
        self.skips = list(skips)
        self.encoder = encoder
        self.in_dim = encoder.output_dim
        self.bound = bound
        self.out_dim = out_dim
        self.last_activation = last_activation

        self.layers = nn.ModuleList(
            [nn.Linear(self.in_dim, hidden_dim)]
            + [nn.Linear(hidden_dim + self.in_dim, hidden_dim) if i in self.skips else nn.Linear(hidden_dim, hidden_dim)
               for i in range(1, num_layers - 1)])
        self.layers.append(nn.Linear(hidden_dim, out_dim))

        self.activations = nn.ModuleList([nn.LeakyReLU() for _ in range(num_layers - 1)])
        if last_activation == "sigmoid":
            self.activations.append(nn.Sigmoid())
        elif last_activation == "relu":
            self.activations.append(nn.LeakyReLU())
        elif last_activation == "tanh":
            self.activations.append(nn.Tanh())
        elif last_activation == "none":
            self.activations.append(nn.Identity())
        else:
            raise NotImplementedError("Unknown last activation")

    # ---- fused-path helpers -----------------------------------------------------------------------------
    def fused_supported(self):
        """True when the network is the canonical NAF MLP the MFMA kernels are specialised for."""
        enc = self.encoder
        return (hasattr(enc, "embeddings") and getattr(enc, "input_dim", 0) == 3 and self.in_dim == 32
                and self.num_layers == 4 and self.hidden_dim == 32 and self.skips == [2] and self.out_dim == 1)

    def packed_mlp(self):
        """The 4225-value parameter block of include/naf_hip.h (differentiable concatenation)."""
        parts = []
        for lyr in self.layers:
            parts += [lyr.weight.reshape(-1), lyr.bias.reshape(-1)]
        return torch.cat(parts).float()

    def forward(self, x):
        if self.fused_supported() and x.is_cuda and not torch.is_grad_enabled():
            from .fused import field_query
            return field_query(self, x)
        x = self.encoder(x, self.bound)
        input_pts = x[..., :self.in_dim]
        for i in range(len(self.layers)):
            if i in self.skips:
                x = torch.cat([input_pts, x], -1)
            x = self.activations[i](self.layers[i](x))
        return x


def get_network(type):
    if type == "mlp":
        return DensityNetwork
    raise NotImplementedError("Unknown network type!")
