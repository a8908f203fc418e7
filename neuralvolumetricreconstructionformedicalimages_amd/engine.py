"""Training engine for the fused hot path: one `naf_render_train` + two `naf_adam_step` launches per step.

Replaces, for the canonical NAF network, what reference src/trainer.py:134-142 + train.py:48-135 do per step
(6 chunks x ~40 ATen launches, per-chunk 57 MB zero-fill of the table gradient, dense torch.optim.Adam):
  * parameters, Adam moments and gradients are flat fp32 buffers that stay resident in HBM; the module's
    `encoder.embeddings` / `layers.i.weight|bias` are views of them, so state_dict() keeps the reference keys;
  * with a 16-bit table the fp32 master is updated by Adam and the bf16/fp16 shadow the kernels gather from is
    written in the same pass; the gradient buffer is zeroed in that pass too;
  * data parallel (one process per GPU): gradients are summed with one RCCL all-reduce per buffer between the
    backward and the optimiser (see dist.py).
"""
from __future__ import annotations

import ctypes

import torch

from . import _abi
from . import fused


class NAFEngine:
    def __init__(self, net, n_samples, perturb=True, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, table_dtype=torch.float32,
                 mlp_precision=None, seed=0, process_group=None, n_streams=1, chunk_rays=16384):
        if not net.fused_supported():
            raise RuntimeError("NAFEngine needs the canonical NAF network (in 32, hidden 32, 4 layers, skips=[2], out 1)")
        self.net = net
        enc = net.encoder
        dev = enc.embeddings.device
        if dev.type != "cuda":
            raise RuntimeError("NAFEngine: the network must live on the GPU (no CPU path)")
        self.device = dev
        self.n_samples, self.perturb = int(n_samples), bool(perturb)
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.seed = int(seed)
        self.step_count = 0
        self.rays_seen = 0
        self.process_group = process_group

        # ---- flat fp32 master parameters; module parameters become views ---------------------------------
        self.emb = enc.embeddings.data.float().contiguous()
        enc.embeddings.data = self.emb
        self.mlp = net.packed_mlp().detach().clone().contiguous()
        off = 0
        for lyr in net.layers:
            for p in (lyr.weight, lyr.bias):
                n = p.numel()
                p.data = self.mlp[off:off + n].view(p.shape)
                off += n
        assert off == _abi.MLP_PARAMS
        self.table_dtype = table_dtype
        self.emb_lp = None if table_dtype == torch.float32 else self.emb.to(table_dtype)
        self.emb_m, self.emb_v = (torch.zeros_like(self.emb) for _ in range(2))
        self.mlp_m, self.mlp_v = (torch.zeros_like(self.mlp) for _ in range(2))
        # table gradient | MLP gradient | loss in ONE flat buffer (sections 256-byte aligned): a data-parallel step is a
        # single all-reduce
        n_emb, n_mlp = self.emb.numel(), self.mlp.numel()
        o_mlp = (n_emb + 63) // 64 * 64
        o_loss = o_mlp + (n_mlp + 63) // 64 * 64
        self.grad_flat = torch.zeros(o_loss + 64, device=dev)
        self.emb_g = self.grad_flat[:n_emb].view(self.emb.shape)
        self.mlp_g = self.grad_flat[o_mlp:o_mlp + n_mlp]
        self.loss = self.grad_flat[o_loss:o_loss + 1]
        self.acc = None
        self.offsets = enc.offsets.to(dev)
        enc.offsets = self.offsets
        self.mlp_precision = mlp_precision
        if mlp_precision is None:
            self.mlp_precision = _abi.F32 if table_dtype == torch.float32 else _abi.BF16
        # Optional multi-stream execution: the batch is cut into chunks that run their whole forward/backward pipeline on
        # alternating HIP streams, so the gather-bound, VALU-bound and store-bound kernels of different chunks overlap.
        # Each extra stream owns a gradient buffer, a workspace and a loss cell; they are summed before Adam.
        self.n_streams = max(1, int(n_streams))
        self.chunk_rays = int(chunk_rays)
        self._lanes = []
        for _ in range(self.n_streams - 1):
            self._lanes.append({"stream": torch.cuda.Stream(device=dev), "emb_g": torch.zeros_like(self.emb),
                                "mlp_g": torch.zeros_like(self.mlp), "loss": torch.zeros(1, device=dev), "ws": None})

    # -------------------------------------------------------------------------------------------------------
    def _cfg(self, ray_base=0):
        enc = self.net.encoder
        return _abi.RenderCfg(n_samples=self.n_samples, perturb=int(self.perturb), bound=float(self.net.bound),
                              L=enc.num_levels, C=enc.level_dim, H=enc.base_resolution,
                              table_dtype=_abi.dtype_code(self.table_dtype), mlp_precision=int(self.mlp_precision),
                              last_activation=fused.LAST_ACTIVATIONS[self.net.last_activation],
                              seed=(self.seed + 0x9E3779B97F4A7C15 * (self.step_count + 1)) & (2 ** 64 - 1),
                              ray_index_base=int(ray_base), log2_hashmap_size=int(enc.log2_hashmap_size))

    @property
    def table(self):
        return self.emb if self.emb_lp is None else self.emb_lp

    def _launch(self, rays, target, weight, t_rand, ray_base, acc, emb_g, mlp_g, loss, ws):
        n = rays.shape[0]
        cfg = self._cfg(ray_base)
        _abi.check(_abi.lib().naf_render_train(
            _abi.ptr(rays), _abi.ptr(t_rand), _abi.ptr(target), _abi.ptr(weight), _abi.ptr(self.table), _abi.ptr(self.offsets),
            _abi.ptr(self.mlp), _abi.ptr(acc), _abi.ptr(emb_g), _abi.ptr(mlp_g), _abi.ptr(loss), n,
            ctypes.byref(cfg), _abi.ptr(ws), _abi.stream_ptr()), "render_train")

    def backward(self, rays, target, weight, t_rand=None, ray_base=0):
        """Forward + weighted squared error + backward: fills the gradient buffers, returns acc [n]."""
        n = rays.shape[0]
        if self.acc is None or self.acc.numel() < n:
            self.acc = torch.empty(n, device=self.device)
        self.loss.zero_()
        if self.n_streams == 1 or n <= self.chunk_rays:
            cfg = self._cfg(ray_base)
            ws = fused.workspace(cfg, n * self.n_samples, self.device)
            self._launch(rays, target, weight, t_rand, ray_base, self.acc, self.emb_g, self.mlp_g, self.loss, ws)
            fused._bump(self.device)
            return self.acc[:n]
        return self._backward_multistream(rays, target, weight, t_rand, ray_base)

    def _backward_multistream(self, rays, target, weight, t_rand, ray_base):
        n, c = rays.shape[0], self.chunk_rays
        main = torch.cuda.current_stream()
        cfg = self._cfg(ray_base)
        need = int(_abi.lib().naf_render_workspace_bytes(ctypes.byref(cfg), c * self.n_samples))
        lanes = [{"stream": main, "emb_g": self.emb_g, "mlp_g": self.mlp_g, "loss": self.loss,
                  "ws": fused.workspace(cfg, c * self.n_samples, self.device)}] + self._lanes
        start = torch.cuda.Event()
        start.record(main)
        for lane in lanes[1:]:
            if lane["ws"] is None or lane["ws"].numel() < need:
                lane["ws"] = torch.empty(need, dtype=torch.uint8, device=self.device)
            lane["stream"].wait_event(start)                # inputs and parameters are ready
            with torch.cuda.stream(lane["stream"]):
                lane["loss"].zero_()
        for k, b in enumerate(range(0, n, c)):
            e = min(n, b + c)
            lane = lanes[k % len(lanes)]
            with torch.cuda.stream(lane["stream"]):
                tr = None if t_rand is None else t_rand[b:e]
                self._launch(rays[b:e], target[b:e], weight[b:e], tr, ray_base + b, self.acc[b:e], lane["emb_g"], lane["mlp_g"],
                             lane["loss"], lane["ws"])
        for lane in lanes[1:]:                              # fold the side streams' gradients into the main buffers
            done = torch.cuda.Event()
            done.record(lane["stream"])
            main.wait_event(done)
            self.emb_g.add_(lane["emb_g"])
            self.mlp_g.add_(lane["mlp_g"])
            self.loss.add_(lane["loss"])
            lane["emb_g"].zero_()
            lane["mlp_g"].zero_()
        fused._bump(self.device)
        return self.acc[:n]

    def scatter_overflow(self, n_rays):
        """Diagnostic: contributions of the last backward that fell back to atomics (synchronises)."""
        cfg = self._cfg()
        n_points = n_rays * self.n_samples
        ws = fused.workspace(cfg, n_points, self.device)
        out = ctypes.c_uint32(0)
        _abi.check(_abi.lib().naf_scatter_overflow_count(ctypes.byref(cfg), n_points, _abi.ptr(ws), ctypes.byref(out)),
                   "scatter_overflow_count")
        return int(out.value)

    def all_reduce_grads(self):
        if self.process_group is None:
            return
        import torch.distributed as dist
        dist.all_reduce(self.grad_flat, group=self.process_group)       # table + MLP gradients + loss (sum over ranks)

    def optimizer_step(self, grad_scale=1.0):
        self.step_count += 1
        b1, b2 = self.betas
        lp = _abi.ptr(self.emb_lp)
        lp_code = 0 if self.emb_lp is None else _abi.dtype_code(self.table_dtype)
        lib = _abi.lib()
        _abi.check(lib.naf_adam_step(_abi.ptr(self.emb), _abi.ptr(self.emb_m), _abi.ptr(self.emb_v), _abi.ptr(self.emb_g), lp,
                                     lp_code, self.emb.numel(), self.lr, b1, b2, self.eps, self.step_count, grad_scale, 1,
                                     _abi.stream_ptr()), "adam_step(table)")
        _abi.check(lib.naf_adam_step(_abi.ptr(self.mlp), _abi.ptr(self.mlp_m), _abi.ptr(self.mlp_v), _abi.ptr(self.mlp_g), None,
                                     0, self.mlp.numel(), self.lr, b1, b2, self.eps, self.step_count, grad_scale, 1,
                                     _abi.stream_ptr()), "adam_step(mlp)")

    def train_step(self, rays, target, weight, t_rand=None, ray_base=0):
        """One optimisation step on `rays` [n,8]; loss = sum_r weight[r] (acc[r]-target[r])^2.  Returns the loss tensor
        (device, no sync)."""
        self.backward(rays, target, weight, t_rand, ray_base)
        self.all_reduce_grads()
        self.optimizer_step()
        self.rays_seen += rays.shape[0]
        return self.loss

    # ---- optimiser state in torch.optim.Adam's layout (checkpoint compatibility, trainer.py:118-126) ---------
    def optimizer_state_dict(self):
        params = [self.net.encoder.embeddings] + [p for lyr in self.net.layers for p in (lyr.weight, lyr.bias)]
        state = {0: {"step": torch.tensor(float(self.step_count)), "exp_avg": self.emb_m.clone(), "exp_avg_sq": self.emb_v.clone()}}
        off = 0
        for i, p in enumerate(params[1:], start=1):
            n = p.numel()
            state[i] = {"step": torch.tensor(float(self.step_count)), "exp_avg": self.mlp_m[off:off + n].view(p.shape).clone(),
                        "exp_avg_sq": self.mlp_v[off:off + n].view(p.shape).clone()}
            off += n
        group = {"lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": 0, "amsgrad": False,
                 "params": list(range(len(params)))}
        return {"state": state, "param_groups": [group]}

    def load_optimizer_state_dict(self, sd):
        st = sd["state"]
        if len(st) == 0:
            return
        self.step_count = int(float(st[0]["step"]))
        self.emb_m.copy_(st[0]["exp_avg"].to(self.device))
        self.emb_v.copy_(st[0]["exp_avg_sq"].to(self.device))
        off = 0
        for i in range(1, len(st)):
            n = st[i]["exp_avg"].numel()
            self.mlp_m[off:off + n].copy_(st[i]["exp_avg"].reshape(-1).to(self.device))
            self.mlp_v[off:off + n].copy_(st[i]["exp_avg_sq"].reshape(-1).to(self.device))
            off += n
        self.lr = float(sd["param_groups"][0]["lr"])

    def sync_from_module(self):
        """Call after net.load_state_dict(): refresh the low-precision shadow table."""
        if self.emb_lp is not None:
            self.emb_lp.copy_(self.emb)
