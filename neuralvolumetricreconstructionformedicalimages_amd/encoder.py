"""Hash-grid encoder: host-side mirror of reference src/encoder/hashencoder/hashgrid.py and
src/encoder/__init__.py over the HIP kernels in libnaf_hip.so.

Same names, arguments and error behaviour as the reference:
  * `hash_encode(inputs, embeddings, offsets, base_resolution, calc_grad_inputs[, log2_hashmap_size])`  (hashgrid.py:10-74)
  * `HashEncoder(input_dim, num_levels, level_dim, base_resolution, log2_hashmap_size)` with `.output_dim`,
    `.embeddings`, `.offsets`, `forward(inputs, size=1)`                               (hashgrid.py:77-137)
  * `get_encoder(encoding, ...)`                                                       (src/encoder/__init__.py:5-24)
Differences, all MI355X-motivated: the kernel writes the [B, L*C] layout directly (no permute copy,
hashgrid.py:40), table gradients are accumulated in fp32 whatever the table dtype, and the range check runs
on the device (one flag read instead of four `.item()` round trips, hashgrid.py:122-123).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function

from . import _abi


def level_offsets(input_dim, num_levels, base_resolution, log2_hashmap_size):
    """Row offsets per level, int32 [L+1]  (hashgrid.py:92-102)."""
    max_params = 2 ** log2_hashmap_size
    offsets, offset = [], 0
    for i in range(num_levels):
        resolution = base_resolution * 2 ** i
        offsets.append(offset)
        offset += min(max_params, (resolution + 1) ** input_dim)
    offsets.append(offset)
    return np.array(offsets, dtype=np.int32)


class _hash_encode(Function):
    @staticmethod
    def forward(ctx, inputs, embeddings, offsets, base_resolution, calc_grad_inputs=False, log2_hashmap_size=None):
        # inputs [B, D] float in [0,1]; embeddings [sO, C]; offsets [L+1] int32; returns [B, L*C]
        # log2_hashmap_size (an extension of the reference signature, hashgrid.py:13): when given, the backward pass may use the
        # binned scatter with a scratch workspace (naf_hash_encode_backward_ws) instead of one global atomic per corner and channel
        # calc_grad_inputs: False / True (exact input gradient) / _abi.GRAD_INPUTS_REFERENCE (the reference's dy_dx,
        # level scale missing, `nd > gd` dimension pick -- only for comparing against reference runs)
        if torch.is_autocast_enabled():                   # reference: custom_fwd(cast_inputs=torch.half)
            embeddings = embeddings.half()
        inputs = inputs.contiguous().float()
        embeddings = embeddings.contiguous()
        offsets = offsets.contiguous().to(inputs.device)
        B, D = inputs.shape
        L = offsets.shape[0] - 1
        C = embeddings.shape[1]
        H = int(base_resolution)
        dt = embeddings.dtype
        outputs = torch.empty(B, L * C, device=inputs.device, dtype=dt)
        dy_dx = torch.empty(B, L * D * C, device=inputs.device, dtype=dt) if calc_grad_inputs else None
        _abi.check(_abi.lib().naf_hash_encode_forward(
            _abi.ptr(inputs), _abi.ptr(embeddings), _abi.ptr(offsets), _abi.ptr(outputs), B, D, C, L, H,
            int(calc_grad_inputs), _abi.ptr(dy_dx), _abi.dtype_code(dt), _abi.LAYOUT_BLC, _abi.stream_ptr()),
            "hash_encode_forward")
        ctx.save_for_backward(inputs, embeddings, offsets, dy_dx)
        ctx.dims = [B, D, C, L, H]
        ctx.calc_grad_inputs = calc_grad_inputs
        ctx.log2_hashmap_size = log2_hashmap_size
        return outputs

    @staticmethod
    def backward(ctx, grad):
        inputs, embeddings, offsets, dy_dx = ctx.saved_tensors
        B, D, C, L, H = ctx.dims
        calc = ctx.calc_grad_inputs
        grad = grad.contiguous().to(embeddings.dtype)
        grad_embeddings = torch.zeros(embeddings.shape, device=embeddings.device, dtype=torch.float32)
        grad_inputs = torch.zeros_like(inputs) if calc else None
        dtc = _abi.dtype_code(embeddings.dtype)
        need = 0
        if ctx.log2_hashmap_size is not None:
            need = int(_abi.lib().naf_hash_encode_workspace_bytes(B, D, C, L, int(ctx.log2_hashmap_size), dtc))
        if need:
            from . import fused
            ws = fused._grow(inputs.device, need)           # the per-device scratch buffer of the fused calls (stream-ordered reuse)
            _abi.check(_abi.lib().naf_hash_encode_backward_ws(
                _abi.ptr(grad), _abi.ptr(inputs), _abi.ptr(embeddings), _abi.ptr(offsets), _abi.ptr(grad_embeddings),
                B, D, C, L, H, int(calc), _abi.ptr(dy_dx), _abi.ptr(grad_inputs), dtc, _abi.LAYOUT_BLC, int(ctx.log2_hashmap_size),
                _abi.ptr(ws), ws.numel(), _abi.stream_ptr()), "hash_encode_backward_ws")
        else:
            _abi.check(_abi.lib().naf_hash_encode_backward(
                _abi.ptr(grad), _abi.ptr(inputs), _abi.ptr(embeddings), _abi.ptr(offsets), _abi.ptr(grad_embeddings),
                B, D, C, L, H, int(calc), _abi.ptr(dy_dx), _abi.ptr(grad_inputs), dtc, _abi.LAYOUT_BLC, _abi.stream_ptr()),
                "hash_encode_backward")
        grad_embeddings = grad_embeddings.to(embeddings.dtype)
        return (grad_inputs if calc else None), grad_embeddings, None, None, None, None


hash_encode = _hash_encode.apply


class HashEncoder(nn.Module):
    def __init__(self, input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19,
                 strict_range=True, reference_compat=False):
        super().__init__()
        self.input_dim = input_dim
        self.num_levels = num_levels
        self.level_dim = level_dim
        self.log2_hashmap_size = log2_hashmap_size
        self.base_resolution = base_resolution
        self.output_dim = num_levels * level_dim
        self.strict_range = strict_range
        self.reference_compat = reference_compat      # input gradients as the reference computes them (App. A-3)
        if input_dim not in (2, 3) or level_dim not in (1, 2, 4, 8):
            raise RuntimeError("GridEncoding: C must be 1, 2, 4, or 8.")        # hashencoder.cu:310,324
        self.max_params = 2 ** log2_hashmap_size
        offs = level_offsets(input_dim, num_levels, base_resolution, log2_hashmap_size)
        self.offsets = torch.from_numpy(offs)                                   # plain attribute (hashgrid.py:102)
        self.n_params = int(offs[-1]) * level_dim
        self.embeddings = nn.Parameter(torch.zeros(int(offs[-1]), level_dim))
        self._range_flag = None
        self.reset_parameters()

    def reset_parameters(self):
        std = 1e-4
        self.embeddings.data.uniform_(-std, std)

    def __repr__(self):
        return (f"HashEncoder: input_dim={self.input_dim} num_levels={self.num_levels} level_dim={self.level_dim} "
                f"H={self.base_resolution} params={self.embeddings.shape}")

    def _device_offsets(self, device):
        if self.offsets.device != device:
            self.offsets = self.offsets.to(device)
        return self.offsets

    def _normalize(self, inputs, size):
        """[-size,size] -> [0,1] plus the range flag, in one device pass (hashgrid.py:122-125)."""
        flat = inputs.detach().contiguous().float().view(-1)
        if self._range_flag is None or self._range_flag.device != flat.device:
            self._range_flag = torch.empty(3, dtype=torch.int32, device=flat.device)
            self._range_init = torch.tensor([0, 2 ** 31 - 1, -2 ** 31], dtype=torch.int32, device=flat.device)
        self._range_flag.copy_(self._range_init)
        out01 = torch.empty_like(flat)
        _abi.check(_abi.lib().naf_normalize_inputs(_abi.ptr(flat), flat.numel(), float(size), _abi.ptr(out01),
                                                    _abi.ptr(self._range_flag), _abi.stream_ptr()), "normalize_inputs")
        if self.strict_range:
            self.raise_if_out_of_range(size)
        return out01.view(inputs.shape)

    def raise_if_out_of_range(self, size=1):
        """Surface the device-side range flag (the reference raises eagerly, hashgrid.py:122-123)."""
        if self._range_flag is None:
            return
        bad, lo, hi = self._range_flag.tolist()
        if bad:
            def unorder(i):
                i = i if i >= 0 else i ^ 0x7FFFFFFF
                return float(np.array([i], dtype=np.int32).view(np.float32)[0])
            raise ValueError(f"HashGrid encoder: inputs range [{unorder(lo)}, {unorder(hi)}] not in [{-size}, {size}]!")

    def forward(self, inputs, size=1):
        # inputs: [..., input_dim] in [-size, size]  ->  [..., num_levels * level_dim]
        x01 = self._normalize(inputs, size)
        if inputs.requires_grad:                       # dormant in NAF (pts carry no grad): keep autograd's chain rule
            inputs = x01.detach() + (inputs - inputs.detach()) / (2 * size)
        else:
            inputs = x01
        prefix_shape = list(inputs.shape[:-1])
        inputs = inputs.view(-1, self.input_dim)
        calc = inputs.requires_grad
        if calc and self.reference_compat:
            calc = _abi.GRAD_INPUTS_REFERENCE
        outputs = hash_encode(inputs, self.embeddings, self._device_offsets(inputs.device), self.base_resolution, calc, self.log2_hashmap_size)
        return outputs.view(prefix_shape + [self.output_dim])


class FreqEncoder(nn.Module):
    """NeRF-style sin/cos positional encoding: mirror of reference src/encoder/freqencoder.py:5-42.

    No reference config selects it (config/*.yaml:14 all use "hashgrid"), so it is plain tensor ops on the device the
    input lives on rather than a hand-written kernel; it exists so that `get_encoder("frequency")` keeps working."""

    def __init__(self, input_dim, max_freq_log2, N_freqs, log_sampling=True, include_input=True,
                 periodic_fns=(torch.sin, torch.cos)):
        super().__init__()
        self.input_dim = input_dim
        self.include_input = include_input
        self.periodic_fns = periodic_fns
        self.output_dim = (input_dim if include_input else 0) + input_dim * N_freqs * len(periodic_fns)
        if log_sampling:
            bands = 2.0 ** torch.linspace(0.0, max_freq_log2, N_freqs)
        else:
            bands = torch.linspace(2.0 ** 0.0, 2.0 ** max_freq_log2, N_freqs)
        self.freq_bands = bands.numpy().tolist()

    def forward(self, input, bound):
        out = [input] if self.include_input else []
        for freq in self.freq_bands:
            out += [fn(input * freq) for fn in self.periodic_fns]
        return torch.cat(out, dim=-1)


def get_encoder(encoding, input_dim=3, multires=6, degree=4, num_levels=16, level_dim=2, base_resolution=16,
                log2_hashmap_size=19, **kwargs):
    if encoding == "None":
        return lambda x, **kwargs: x, input_dim
    if encoding == "frequency":
        return FreqEncoder(input_dim=input_dim, max_freq_log2=multires - 1, N_freqs=multires, log_sampling=True)
    if encoding == "hashgrid":
        return HashEncoder(input_dim=input_dim, num_levels=num_levels, level_dim=level_dim,
                           base_resolution=base_resolution, log2_hashmap_size=log2_hashmap_size)
    raise NotImplementedError()
