"""Training engine for the fused hot path: one `naf_render_train` + two `naf_adam_step` launches per step.

Replaces, for the canonical NAF network, what reference src/trainer.py:134-142 + train.py:48-135 do per step
(6 chunks x ~40 ATen launches, per-chunk 57 MB zero-fill of the table gradient, dense torch.optim.Adam):
  * parameters, Adam moments and gradients are flat fp32 buffers that stay resident in HBM; the module's
    `encoder.embeddings` / `layers.i.weight|bias` are views of them, so state_dict() keeps the reference keys;
  * with a 16-bit table the fp32 master is updated by Adam and the bf16/fp16 shadow the kernels gather from is
    written in the same pass; the gradient buffer is zeroed in that pass too;
  * data parallel (one process per GPU, rays sharded, model replicated): the table gradient is finished bucket by bucket
    (level groups, fine levels first; `naf_render_train_bucketed` records an event per bucket), each bucket's slice of the
    flat gradient buffer is all-reduced (RCCL) on a side stream while the next bucket is still being binned and reduced,
    and Adam runs per bucket as soon as its sum has arrived -- only the last, smallest exchange is exposed (see dist.py).
"""
from __future__ import annotations

import ctypes
import math

import torch

from . import _abi
from . import fused


class NAFEngine:
    def __init__(self, net, n_samples, perturb=True, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, table_dtype=torch.float32,
                 mlp_precision=None, seed=0, process_group=None, n_streams=1, chunk_rays=16384, scatter_mode=None, cfg_flags=None,
                 bucket_levels=None, fuse_table_adam=True, dp_mode="sharded", rays_per_step_hint=None):
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
        if dp_mode not in ("auto", "sharded", "allreduce", "levels"):
            raise ValueError("dp_mode must be 'sharded' (reduce-scatter, per-rank Adam on a table slice, all-gather), 'allreduce', "
                             "'levels' (each rank owns a range of levels; features and their gradients cross in two all-to-alls) or 'auto'")
        self.world, self.rank = 1, 0
        if process_group is not None:
            import torch.distributed as dist
            self.world, self.rank = dist.get_world_size(process_group), dist.get_rank(process_group)
        if dp_mode == "auto":
            from . import dist as naf_dist
            pts = None if rays_per_step_hint is None else int(rays_per_step_hint) * int(n_samples)
            feat_bytes = 4 if (mlp_precision == _abi.F32 or (mlp_precision is None and table_dtype == torch.float32)) else 2
            dp_mode = naf_dist.pick_dp_mode(self.world, enc.num_levels, enc.level_dim, enc.embeddings.numel(), pts, feat_bytes,
                                            4 if table_dtype == torch.float32 else 2)
        self.dp_mode = dp_mode
        self._rays_hint = None if rays_per_step_hint is None else int(rays_per_step_hint)
        self.scatter_mode, self.cfg_flags = scatter_mode, cfg_flags     # None: fused.scatter_mode() default (auto)
        # single-GPU, single-stream steps let the gradient reducer apply the table's Adam update itself (naf_render_train_adam:
        # the gradient table is neither written, re-read nor cleared; bit-identical to backward() + optimizer_step())
        self.fuse_table_adam = bool(fuse_table_adam)

        # ---- flat fp32 master parameters; module parameters become views ---------------------------------
        # The table lives in a flat buffer whose length is rounded up to `pad_to` elements: the exchange ranges of a data-parallel
        # step are multiples of world * 4 elements (equal 16-byte-aligned shards per rank), the last one reaches into the padding.
        n_emb = enc.embeddings.numel()
        self._pad_to = 64 * self.world // math.gcd(64, 4 * self.world) * 4 if self.world > 1 else 64
        n_pad = (n_emb + self._pad_to - 1) // self._pad_to * self._pad_to
        self._emb_flat = torch.zeros(n_pad, device=dev)
        self._emb_flat[:n_emb] = enc.embeddings.data.float().reshape(-1)
        self.emb = self._emb_flat[:n_emb].view(enc.embeddings.shape)
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
        self._lp_flat = None if table_dtype == torch.float32 else self._emb_flat.to(table_dtype)
        self.emb_lp = None if self._lp_flat is None else self._lp_flat[:n_emb].view(self.emb.shape)
        self.emb_m, self.emb_v = (torch.zeros_like(self.emb) for _ in range(2))
        self.mlp_m, self.mlp_v = (torch.zeros_like(self.mlp) for _ in range(2))
        # table gradient | MLP gradient | loss in ONE flat buffer (sections 256-byte aligned): a data-parallel step is a
        # single all-reduce
        n_mlp = self.mlp.numel()
        o_mlp = n_pad
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
        self._dp = None
        self._lv = None
        self._levels_flags = 0
        if process_group is not None:
            if int(n_streams) > 1:
                raise ValueError("NAFEngine: n_streams > 1 cannot be combined with a process group (the bucket events are "
                                 "recorded by the one launch that owns the gradient buffer)")
            if self.dp_mode == "levels":
                self._init_level_parallel()
            else:
                self._init_data_parallel(bucket_levels, None if rays_per_step_hint is None else int(rays_per_step_hint) * self.n_samples)
        self.n_streams = max(1, int(n_streams))
        self.chunk_rays = int(chunk_rays)
        self._lanes = []
        for _ in range(self.n_streams - 1):
            self._lanes.append({"stream": torch.cuda.Stream(device=dev), "emb_g": torch.zeros_like(self.emb),
                                "mlp_g": torch.zeros_like(self.mlp), "loss": torch.zeros(1, device=dev), "ws": None})

    # ---- data parallel -------------------------------------------------------------------------------------
    def _init_data_parallel(self, bucket_levels, points_per_step=None):
        """Buckets = level ranges in the order the scatter finishes them (dist.default_bucket_levels; `points_per_step`, this
        rank's sample points per step when the caller knows them, picks the single-range exchange for small steps)."""
        from . import dist as naf_dist
        L = self.net.encoder.num_levels
        if bucket_levels is None:
            bucket_levels = naf_dist.default_bucket_levels(L, points_per_step)
        bucket_levels = [(int(a), int(b)) for a, b in bucket_levels]
        if len(bucket_levels) > _abi.MAX_GRAD_BUCKETS:
            raise ValueError(f"at most {_abi.MAX_GRAD_BUCKETS} gradient buckets")
        n_emb = self.emb.numel()
        o_mlp = self._emb_flat.numel()
        dp = {"levels": bucket_levels, "comm": torch.cuda.Stream(device=self.device), "time": False, "timings": []}
        # torch creates the underlying hipEvent_t at the first record(): do that now so the handles can be handed to the library
        def event():
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            return ev
        dp["ready"] = [event() for _ in bucket_levels]
        dp["mlp_ready"] = event()
        dp["done"] = [event() for _ in bucket_levels]
        dp["mlp_done"] = event()
        dp["slices"] = naf_dist.grad_bucket_slices(self.offsets.tolist(), self.net.encoder.level_dim, bucket_levels)
        dp["update_slices"] = naf_dist.aligned_update_slices(dp["slices"])              # Adam works on 16-byte groups
        dp["mlp_slice"] = (o_mlp, self.grad_flat.numel())                              # MLP gradient + loss cell
        if self.dp_mode == "sharded":
            # reduce-scatter -> Adam on this rank's slice of every bucket -> all-gather of the updated table:
            # exchange ranges are multiples of world * 4 elements (boundaries moved in favour of the bucket that finishes later,
            # the table's end extended into the buffer's padding), so every rank owns an equal, 16-byte-aligned shard of each
            dp["shard_slices"] = naf_dist.sharded_exchange_slices(dp["slices"], self.world, o_mlp)
            longest = max((b - a) // self.world for a, b in dp["shard_slices"])
            dp["shard_grad"] = [torch.zeros(longest, device=self.device) for _ in bucket_levels]      # reduce-scatter outputs
            dp["rs_done"] = [event() for _ in bucket_levels]
            dp["adam_done"] = [event() for _ in bucket_levels]
            dp["gathered"] = event()
            dp["master_stale"] = False
        st = _abi.GradBuckets()
        st.n_buckets = len(bucket_levels)
        for i, (a, b) in enumerate(bucket_levels):
            st.level_begin[i], st.level_end[i] = a, b
            st.ready[i] = dp["ready"][i].cuda_event
        st.mlp_ready = dp["mlp_ready"].cuda_event
        dp["struct"] = st
        self._dp = dp

    # ---- level parallel -------------------------------------------------------------------------------------
    def _init_level_parallel(self):
        """Rank k owns the levels [k L/N, (k+1) L/N): their rows of the table, of the 16-bit shadow and of the Adam moments are
        current on that rank only (`gather_state` completes them everywhere before an evaluation or a checkpoint)."""
        L, N = self.net.encoder.num_levels, self.world
        if L % N != 0:
            raise ValueError(f"dp_mode 'levels' needs a world size that divides the {L} levels (got {N}); use 'sharded'")
        per = L // N
        # with one or two levels per rank the scatter uses 256 row buckets per level instead of 64 (NAF_CFG_MIN_BUCKETS): 256 / 512 reducer
        # workgroups that each own their rows, so that no launch is split and the reducer applies Adam itself (tools/levels_emulate.py,
        # 8 ranks: reduce + Adam 0.102 -> 0.064 ms per step; with four levels per rank 128 buckets measured no gain: 0.270 against 0.261 ms)
        self._levels_flags = {1: 2, 2: 2}.get(per, 0) << _abi.CFG_MIN_BUCKETS_SHIFT
        offs = [int(v) for v in self.offsets.tolist()]
        C = self.net.encoder.level_dim

        def event():
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            return ev
        self._lv = {"levels": (self.rank * per, (self.rank + 1) * per), "per": per,
                    "rows": [(offs[k * per] * C, offs[(k + 1) * per] * C) for k in range(N)],      # element ranges by owner
                    "comm": torch.cuda.Stream(device=self.device), "mlp_ready": event(), "mlp_done": event(), "buf": {},
                    "grads_ready": event(), "exchange": torch.cuda.Stream(device=self.device),
                    "stale": False, "time": False, "timings": []}

    def _all_to_all(self, out, inp):
        """Equal-split all-to-all of two contiguous device buffers (RCCL; the gloo rehearsal of a one-GPU box stages through the host)."""
        import torch.distributed as dist
        if dist.get_backend(self.process_group) == "gloo" and inp.is_cuda:
            o = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(o.view(torch.uint8).view(-1), inp.cpu().view(torch.uint8).view(-1), group=self.process_group)
            out.copy_(o)
        else:
            dist.all_to_all_single(out.view(-1), inp.view(-1), group=self.process_group)

    def _train_step_levels(self, rays, target, weight, t_rand, ray_base, rays_all=None, global_ray_base=None):
        """One level-parallel step (include/naf_hip.h, naf_levels_*): encode the owned levels for every rank's points -> all-to-all
        -> MLP forward / loss / backward on the own rays -> all-to-all of the feature gradients (+ a 17 KB all-reduce of the MLP
        gradient and the loss behind it, overlapping the scatter) -> scatter + Adam on the owned levels.  Same result as the
        data-parallel step and as one process on the concatenated batch.  Every rank must bring the same number of rays; the
        jitter index of ray j of rank k is global_ray_base + k * n + j; `global_ray_base` defaults to ray_base - rank * n (the convention
        ray_base = (step * world + rank) * n of trainer.py / bench.py) -- a caller with another convention for ray_base passes it
        explicitly (the same value on every rank).  `rays_all` [world * n, 8]: all ranks' rays in rank order when the caller has them (a shared pixel
        draw); otherwise they are all-gathered (32 KB per rank)."""
        import torch.distributed as dist
        N, r, grp, lv = self.world, self.rank, self.process_group, self._lv
        n, S = rays.shape[0], self.n_samples
        enc = self.net.encoder
        L, C = enc.num_levels, enc.level_dim
        lb, le = lv["levels"]
        nl = le - lb
        if n == 0:
            raise ValueError("dp_mode 'levels': every rank needs the same, non-zero number of rays per step")
        main = torch.cuda.current_stream(self.device)
        fixed = lv.get("n")
        if fixed is None:
            # The first step fixes the batch size of the run.  Every rank has its first step at the same time, so the cross-rank check
            # below is issued by ALL ranks or by none (a per-size cache would let one rank skip a collective another rank issues --
            # mismatched collectives, i.e. a hang until the group's timeout); with a rays_per_step_hint (the YAML's n_rays / world:
            # trainer.py, bench.py) the size is validated locally and no collective is needed at all.
            if self._rays_hint is not None:
                if n != self._rays_hint:
                    raise ValueError(f"dp_mode 'levels': this rank brought {n} rays, the engine was built for {self._rays_hint} per rank "
                                     f"and step (rays_per_step_hint); use dp_mode 'sharded' for uneven shards")
            else:
                both = torch.tensor([n, -n], device=self.device, dtype=torch.int64)
                dist.all_reduce(both, op=dist.ReduceOp.MAX, group=grp)
                if int(both[0]) != n or int(both[1]) != -n:
                    raise ValueError(f"dp_mode 'levels': ranks hold different numbers of rays this step (this rank {n}, largest {int(both[0])}, "
                                     f"smallest {-int(both[1])}); use dp_mode 'sharded' for uneven shards")
            lv["n"] = n
        elif n != fixed:
            raise ValueError(f"dp_mode 'levels': {n} rays in this step, {fixed} in the first one -- a level-parallel run keeps one batch "
                             f"size per rank (equal-split collectives); use dp_mode 'sharded' for varying or uneven shards")
        if rays_all is None:
            rays_all = torch.empty(N * n, 8, device=self.device)
            dist.all_gather_into_tensor(rays_all, rays.contiguous(), group=grp)
        elif rays_all.shape[0] != N * n:
            raise ValueError("rays_all must hold world_size * n rays")
        t_all = None
        if t_rand is not None:
            t_all = torch.empty(N * n, t_rand.shape[1], device=self.device)
            dist.all_gather_into_tensor(t_all, t_rand.contiguous(), group=grp)
        fdt = torch.float32 if int(self.mlp_precision) == _abi.F32 else torch.bfloat16
        esz = 4 if fdt == torch.float32 else 2
        run = n * S * C                                            # elements of one (rank, level)
        key = (n, fdt)
        if key not in lv["buf"]:
            lv["buf"].clear()
            mk = lambda *shape: torch.empty(*shape, dtype=fdt, device=self.device)
            lv["buf"][key] = {"send": mk(N, nl, run), "feat": mk(L, run), "dfeat": mk(L, run), "recv": mk(N, nl * run)}
        b = lv["buf"][key]
        if self.acc is None or self.acc.numel() < n:
            self.acc = torch.empty(n, device=self.device)
        g_base = (ray_base - r * n) if global_ray_base is None else int(global_ray_base)
        cfg_all, cfg = self._cfg(g_base & 0xffffffff), self._cfg((g_base + r * n) & 0xffffffff)
        ws = fused.workspace(cfg_all, N * n * S, self.device)
        lib, sp = _abi.lib(), _abi.stream_ptr()
        marks = []

        def mark():
            if lv["time"]:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(main)
                marks.append(ev)
        mark()
        _abi.check(lib.naf_levels_encode(_abi.ptr(rays_all), _abi.ptr(t_all), _abi.ptr(self.table), _abi.ptr(self.offsets), _abi.ptr(b["send"]),
                                         N * n, N, ctypes.byref(cfg_all), lb, le, sp), "levels_encode")      # one block per destination rank
        mark()
        self._all_to_all(b["feat"], b["send"])                      # block k of the result = rank k's levels of MY points: [L][points][C]
        mark()
        _abi.check(lib.naf_levels_field_step(_abi.ptr(rays), _abi.ptr(t_rand), _abi.ptr(target), _abi.ptr(weight), _abi.ptr(b["feat"]),
                                             _abi.ptr(self.mlp), _abi.ptr(self.acc), _abi.ptr(b["dfeat"]), _abi.ptr(self.mlp_g),
                                             _abi.ptr(self.loss), n, ctypes.byref(cfg), _abi.ptr(ws), lv["grads_ready"].cuda_event, sp),
                   "levels_field_step")
        lv["mlp_ready"].record(main)
        mark()
        # the gradients' all-to-all starts behind the MLP backward kernel (the event), not behind the slab reduction that follows it
        ex = lv["exchange"]
        ex.wait_event(lv["grads_ready"])
        with torch.cuda.stream(ex):
            self._all_to_all(b["recv"], b["dfeat"])                 # block k = rank k's gradients of MY levels
        main.wait_stream(ex)
        mark()
        with torch.cuda.stream(lv["comm"]):                         # issued after the all-to-all, so it queues behind it on the links
            lv["comm"].wait_event(lv["mlp_ready"])
            o_mlp = self._emb_flat.numel()
            dist.all_reduce(self.grad_flat[o_mlp:], group=grp)      # MLP gradient + loss
            self.step_count += 1
            self._adam(self.mlp, self.mlp_m, self.mlp_v, self.mlp_g, None, 0, "adam_step(mlp)")      # beside the scatter, not behind it
            lv["mlp_done"].record(lv["comm"])
        b1, b2 = self.betas
        st = _abi.TableAdam()
        st.param, st.exp_avg, st.exp_avg_sq = self.emb.data_ptr(), self.emb_m.data_ptr(), self.emb_v.data_ptr()
        st.param_lp = None if self.emb_lp is None else self.emb_lp.data_ptr()
        st.lp_dtype = 0 if self.emb_lp is None else _abi.dtype_code(self.table_dtype)
        st.n, st.lr, st.beta1, st.beta2, st.eps, st.step, st.grad_scale = self.emb.numel(), self.lr, b1, b2, self.eps, self.step_count, 1.0
        applied = ctypes.c_int(0)
        _abi.check(lib.naf_levels_scatter(_abi.ptr(rays_all), _abi.ptr(t_all), _abi.ptr(b["recv"]), nl * run * esz, N, _abi.ptr(self.offsets),
                                          _abi.ptr(self.emb_g), N * n, ctypes.byref(cfg_all), lb, le, _abi.ptr(ws), ctypes.byref(st),
                                          ctypes.byref(applied), sp), "levels_scatter")
        if not applied.value:
            # the reducer launches were split (few levels per rank) or the batch took the atomic scatter: the gradient of the owned
            # rows sits in emb_g
            self._adam_rows(*lv["rows"][r], what="adam_step(owned levels)")
        main.wait_event(lv["mlp_done"])                              # the next step reads the stepped MLP
        mark()
        if lv["time"]:
            lv["timings"].append(marks)
        lv["stale"] = N > 1
        fused._bump(self.device)

    def broadcast_parameters(self, src=0):
        """Every rank starts from rank `src`'s table and MLP (and refreshes its 16-bit shadow)."""
        if self.process_group is None:
            return
        import torch.distributed as dist
        for t in (self.emb, self.mlp):
            dist.broadcast(t, src=src, group=self.process_group)
        self.sync_from_module()

    # -------------------------------------------------------------------------------------------------------
    def _cfg(self, ray_base=0):
        enc = self.net.encoder
        return _abi.RenderCfg(n_samples=self.n_samples, perturb=int(self.perturb), bound=float(self.net.bound),
                              L=enc.num_levels, C=enc.level_dim, H=enc.base_resolution,
                              table_dtype=_abi.dtype_code(self.table_dtype), mlp_precision=int(self.mlp_precision),
                              last_activation=fused.LAST_ACTIVATIONS[self.net.last_activation],
                              seed=(self.seed + 0x9E3779B97F4A7C15 * (self.step_count + 1)) & (2 ** 64 - 1),
                              ray_index_base=int(ray_base), log2_hashmap_size=int(enc.log2_hashmap_size),
                              scatter_mode=fused._default_scatter_mode if self.scatter_mode is None else int(self.scatter_mode),
                              flags=(fused._default_flags if self.cfg_flags is None else int(self.cfg_flags)) | self._levels_flags)

    @property
    def table(self):
        return self.emb if self.emb_lp is None else self.emb_lp

    def _launch(self, rays, target, weight, t_rand, ray_base, acc, emb_g, mlp_g, loss, ws):
        n = rays.shape[0]
        cfg = self._cfg(ray_base)
        args = (_abi.ptr(rays), _abi.ptr(t_rand), _abi.ptr(target), _abi.ptr(weight), _abi.ptr(self.table), _abi.ptr(self.offsets),
                _abi.ptr(self.mlp), _abi.ptr(acc), _abi.ptr(emb_g), _abi.ptr(mlp_g), _abi.ptr(loss), n, ctypes.byref(cfg), _abi.ptr(ws))
        if self._dp is not None and emb_g is self.emb_g:
            _abi.check(_abi.lib().naf_render_train_bucketed(*args, ctypes.byref(self._dp["struct"]), _abi.stream_ptr()),
                       "render_train_bucketed")
        else:
            _abi.check(_abi.lib().naf_render_train(*args, _abi.stream_ptr()), "render_train")

    def backward(self, rays, target, weight, t_rand=None, ray_base=0):
        """Forward + weighted squared error + backward: fills the gradient buffers, returns acc [n]."""
        if self._lv is not None:
            raise NotImplementedError("dp_mode 'levels' has no separate backward / optimizer_step: a rank holds only its levels' rows "
                                      "(use train_step)")
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

    def scatter_overflow_levels(self, n_rays):
        """Diagnostic: the same per level (list of num_levels counts; synchronises)."""
        cfg = self._cfg()
        n_points = n_rays * self.n_samples
        ws = fused.workspace(cfg, n_points, self.device)
        out = (ctypes.c_uint32 * 32)()
        _abi.check(_abi.lib().naf_scatter_overflow_levels(ctypes.byref(cfg), n_points, _abi.ptr(ws), ctypes.byref(out)),
                   "scatter_overflow_levels")
        return [int(v) for v in out][:self.net.encoder.num_levels]

    def all_reduce_grads(self):
        """Single-buffer form (one all-reduce of table + MLP gradients + loss); the training step uses the bucketed,
        overlapped form below.  Kept for callers that fill the gradient buffers themselves."""
        if self.process_group is None:
            return
        import torch.distributed as dist
        dist.all_reduce(self.grad_flat, group=self.process_group)       # table + MLP gradients + loss (sum over ranks)

    def _adam(self, param, m, v, g, lp, lp_code, what, grad_scale=1.0):
        b1, b2 = self.betas
        _abi.check(_abi.lib().naf_adam_step(_abi.ptr(param), _abi.ptr(m), _abi.ptr(v), _abi.ptr(g), _abi.ptr(lp), lp_code,
                                            param.numel(), self.lr, b1, b2, self.eps, self.step_count, grad_scale, 1,
                                            _abi.stream_ptr()), what)

    def _adam_rows(self, a, e, what="adam_step(table rows)"):
        """Adam on the elements [a, e) of the flat table (master, moments, gradient, 16-bit shadow), gradient cleared: a ragged head
        of up to three elements one by one, the rest in 16-byte groups."""
        lp_code = 0 if self.emb_lp is None else _abi.dtype_code(self.table_dtype)
        emb, m, v, g = (t.view(-1) for t in (self.emb, self.emb_m, self.emb_v, self.emb_g))
        lp = None if self.emb_lp is None else self.emb_lp.view(-1)
        head = min(e, a + (-a) % 4)
        for lo, hi in ((a, head), (head, e)):
            if hi > lo:
                self._adam(emb[lo:hi], m[lo:hi], v[lo:hi], g[lo:hi], None if lp is None else lp[lo:hi], lp_code, what)

    def optimizer_step(self, grad_scale=1.0):
        self.step_count += 1
        lp_code = 0 if self.emb_lp is None else _abi.dtype_code(self.table_dtype)
        self._adam(self.emb, self.emb_m, self.emb_v, self.emb_g, self.emb_lp, lp_code, "adam_step(table)", grad_scale)
        self._adam(self.mlp, self.mlp_m, self.mlp_v, self.mlp_g, None, 0, "adam_step(mlp)", grad_scale)

    def _exchange_and_step(self):
        """Data-parallel tail of a step.  On the side stream, per bucket in the order the scatter finishes them: wait for
        the bucket's event, all-reduce its slice of the flat gradient buffer.  On the main stream: as each sum arrives,
        Adam on exactly that slice of the parameters (so the last exchange overlaps the first bucket's update)."""
        import torch.distributed as dist
        dp = self._dp
        main, comm = torch.cuda.current_stream(self.device), dp["comm"]
        timing = dp["time"]
        marks = []
        order = [("mlp", dp["mlp_ready"], dp["mlp_done"], dp["mlp_slice"])]
        order += [(i, dp["ready"][i], dp["done"][i], dp["slices"][i]) for i in range(len(dp["levels"]))]
        with torch.cuda.stream(comm):
            for tag, ready, done, (a, b) in order:
                comm.wait_event(ready)
                if timing:
                    t0 = torch.cuda.Event(enable_timing=True)
                    t0.record(comm)
                dist.all_reduce(self.grad_flat[a:b], group=self.process_group)
                done.record(comm)
                if timing:
                    t1 = torch.cuda.Event(enable_timing=True)
                    t1.record(comm)
                    marks.append((t0, t1))
        if timing:
            c0 = torch.cuda.Event(enable_timing=True)
            c0.record(main)                                      # end of this rank's own compute
        self.step_count += 1
        lp_code = 0 if self.emb_lp is None else _abi.dtype_code(self.table_dtype)
        emb, m, v, g = (t.view(-1) for t in (self.emb, self.emb_m, self.emb_v, self.emb_g))
        lp = None if self.emb_lp is None else self.emb_lp.view(-1)
        waited = []
        for i, (a, b) in enumerate(dp["update_slices"]):
            main.wait_event(dp["done"][i])
            if timing:
                w = torch.cuda.Event(enable_timing=True)
                w.record(main)
                waited.append(w)
            self._adam(emb[a:b], m[a:b], v[a:b], g[a:b], None if lp is None else lp[a:b], lp_code, "adam_step(table bucket)")
        main.wait_event(dp["mlp_done"])
        self._adam(self.mlp, self.mlp_m, self.mlp_v, self.mlp_g, None, 0, "adam_step(mlp)")
        if timing:
            c1 = torch.cuda.Event(enable_timing=True)
            c1.record(main)
            dp["timings"].append((marks, c0, waited, c1))

    def _exchange_and_step_sharded(self):
        """Data-parallel tail with a sharded optimiser (ZeRO-1 style).  Per bucket, in the order the scatter finishes
        them: reduce-scatter of its gradient range on the side stream (each rank receives the SUM over ranks of its 1/N slice:
        (N-1)/N x 57 MB on the wire instead of twice that for an all-reduce) -> Adam on exactly that slice of parameter and moments
        on the main stream (1/N of the optimiser pass; the other slices' moments are never touched here) -> all-gather of the
        updated slice of the table the kernels read (the 16-bit shadow in 16-bit mode: (N-1)/N x 28.5 MB; the fp32 table itself
        in parity mode).  The fp32 master of the slices other ranks own is refreshed only on demand (`gather_state`, before an
        evaluation or a checkpoint).  The MLP gradient + loss (17 KB) are all-reduced and stepped on every rank."""
        import torch.distributed as dist
        dp = self._dp
        main, comm = torch.cuda.current_stream(self.device), dp["comm"]
        timing = dp["time"]
        marks = []
        world, rank, n_emb = self.world, self.rank, self.emb.numel()
        grp = self.process_group

        def mark(stream):
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(stream)
            return ev

        with torch.cuda.stream(comm):
            comm.wait_event(dp["mlp_ready"])
            t0 = mark(comm) if timing else None
            a, b = dp["mlp_slice"]
            dist.all_reduce(self.grad_flat[a:b], group=grp)
            dp["mlp_done"].record(comm)
            if timing:
                marks.append((t0, mark(comm)))
            for i, (a, b) in enumerate(dp["shard_slices"]):
                comm.wait_event(dp["ready"][i])
                t0 = mark(comm) if timing else None
                sh = (b - a) // world
                dist.reduce_scatter_tensor(dp["shard_grad"][i][:sh], self.grad_flat[a:b], group=grp)
                self.grad_flat[a:b].zero_()                  # the next step's scatter accumulates from zero
                dp["rs_done"][i].record(comm)
                if timing:
                    marks.append((t0, mark(comm)))
        c0 = mark(main) if timing else None                  # end of this rank's own compute
        self.step_count += 1
        lp_code = 0 if self.emb_lp is None else _abi.dtype_code(self.table_dtype)
        emb, m, v = (t.view(-1) for t in (self.emb, self.emb_m, self.emb_v))
        read_flat = self._emb_flat if self._lp_flat is None else self._lp_flat      # what the kernels gather from
        waited = []
        for i, (a, b) in enumerate(dp["shard_slices"]):
            sh = (b - a) // world
            lo = a + rank * sh
            hi = min(lo + sh, n_emb)                         # the last shard of the last range reaches into the padding
            main.wait_event(dp["rs_done"][i])
            if timing:
                waited.append(mark(main))
            if hi > lo:
                lp = None if self._lp_flat is None else self._lp_flat[lo:hi]
                self._adam(emb[lo:hi], m[lo:hi], v[lo:hi], dp["shard_grad"][i][:hi - lo], lp, lp_code, "adam_step(table shard)")
            dp["adam_done"][i].record(main)
        main.wait_event(dp["mlp_done"])
        self._adam(self.mlp, self.mlp_m, self.mlp_v, self.mlp_g, None, 0, "adam_step(mlp)")
        with torch.cuda.stream(comm):
            for i, (a, b) in enumerate(dp["shard_slices"]):
                sh = (b - a) // world
                comm.wait_event(dp["adam_done"][i])
                t0 = mark(comm) if timing else None
                mine = read_flat[a + rank * sh:a + (rank + 1) * sh].clone()      # out of place: no aliasing assumptions on the backend
                dist.all_gather_into_tensor(read_flat[a:b], mine, group=grp)
                if timing:
                    marks.append((t0, mark(comm)))
            dp["gathered"].record(comm)
        main.wait_event(dp["gathered"])                       # the next forward reads the gathered table
        dp["master_stale"] = self._lp_flat is not None and world > 1
        if timing:
            dp["timings"].append((marks, c0, waited, mark(main)))

    def gather_state(self):
        """Sharded data-parallel training keeps the fp32 master (16-bit mode) and the Adam moments current only on the rank that
        owns a slice.  Collective: every rank calls it (before an evaluation or a checkpoint, trainer.py:113-126) and ends up
        with the complete master table and moments.  A no-op for single-process and all-reduce training."""
        if self.dp_mode == "levels" and self.world > 1:
            import torch.distributed as dist
            flats = [self.emb.view(-1), self.emb_m.view(-1), self.emb_v.view(-1)] + ([] if self.emb_lp is None else [self.emb_lp.view(-1)])
            for k, (a, e) in enumerate(self._lv["rows"]):           # every owner hands out its rows
                for t in flats:
                    dist.broadcast(t[a:e], src=dist.get_global_rank(self.process_group, k), group=self.process_group)
            self._lv["stale"] = False
            return
        if self._dp is None or self.dp_mode != "sharded" or self.world == 1:
            return
        import torch.distributed as dist
        dp = self._dp
        torch.cuda.current_stream(self.device).wait_event(dp["gathered"])
        n_emb, world, rank = self.emb.numel(), self.world, self.rank
        full = [self.emb_m.view(-1), self.emb_v.view(-1)] + ([self.emb.view(-1)] if self._lp_flat is not None else [])
        for a, b in dp["shard_slices"]:
            sh = (b - a) // world
            lo = a + rank * sh
            for t in full:
                mine = torch.zeros(sh, device=self.device)
                k = max(0, min(lo + sh, n_emb) - lo)
                mine[:k] = t[lo:lo + k]
                out = torch.empty(b - a, device=self.device)
                dist.all_gather_into_tensor(out, mine, group=self.process_group)
                t[a:min(b, n_emb)] = out[:min(b, n_emb) - a]
        dp["master_stale"] = False

    def comm_timing(self, enable=True):
        """Switch on event timing of the exchange (bench.py); `comm_report()` then returns per-step averages."""
        if self._dp is not None:
            self._dp["time"], self._dp["timings"] = bool(enable), []
        if self._lv is not None:
            self._lv["time"], self._lv["timings"] = bool(enable), []

    def comm_report(self):
        """-> {"allreduce_ms_per_step": time the collectives were in flight on the side stream (sum over buckets),
        "exposed_ms_per_step": time the main stream spent between the end of its own compute and the last Adam launch minus the
        Adam kernels themselves, i.e. what the exchange adds to the step}.  Synchronises."""
        if self._lv is not None and self._lv["timings"]:
            torch.cuda.synchronize(self.device)
            steps = self._lv["timings"]
            names = ("encode_ms", "features_all_to_all_ms", "field_ms", "gradients_all_to_all_ms", "scatter_adam_ms")
            out = {k: sum(m[i].elapsed_time(m[i + 1]) for m in steps) / len(steps) for i, k in enumerate(names)}
            out["allreduce_ms_per_step"] = out["features_all_to_all_ms"] + out["gradients_all_to_all_ms"]
            out["tail_ms_per_step"] = out["allreduce_ms_per_step"]
            return out
        if self._dp is None or not self._dp["timings"]:
            return None
        torch.cuda.synchronize(self.device)
        steps = self._dp["timings"]
        in_flight = sum(sum(a.elapsed_time(b) for a, b in marks) for marks, _, _, _ in steps) / len(steps)
        tail = sum(c0.elapsed_time(c1) for _, c0, _, c1 in steps) / len(steps)
        return {"allreduce_ms_per_step": in_flight, "tail_ms_per_step": tail}

    def sample_depths(self, rays, t_rand=None, ray_base=0):
        """The sample depths z [n, S] the NEXT train_step / backward on these rays will use (explicit jitter, or the counter-based
        generator keyed by this step's seed and the global ray index): what `raw_noise_std` needs to form its term."""
        cfg = self._cfg(ray_base)
        n = rays.shape[0]
        z = torch.empty(n, self.n_samples, device=self.device)
        _abi.check(_abi.lib().naf_sample_rays(_abi.ptr(rays), _abi.ptr(t_rand), _abi.ptr(z), None, n, self.n_samples, int(self.perturb),
                                              float(self.net.bound), cfg.seed, int(ray_base), _abi.stream_ptr()), "sample_rays")
        return z

    def train_step(self, rays, target, weight, t_rand=None, ray_base=0, raw_noise_std=0.0, noise=None, rays_all=None, global_ray_base=None,
                   next_draw=None):
        """One optimisation step on `rays` [n,8]; loss = sum_r weight[r] (acc[r]-target[r])^2.  Returns the loss tensor
        (device, no sync).  `raw_noise_std` > 0 (render.py:196-199): the per-sample noise on sigma adds sum_s noise_s * dist_s to a
        ray's line integral and nothing else (render.noise_line_integral), so the step runs on target - that term; `noise`: explicit
        N(0, 1) draws [n, S] instead of torch.randn.  `rays_all` / `global_ray_base`: level-parallel steps only (_train_step_levels).
        `next_draw` (`RayGenerator.plan_draw`): the pixel draw of the NEXT step, carried along by this one (naf_render_train_adam_draw:
        spare workgroups of the scatter's first launch on the fused single-GPU path, a launch of its own behind the step otherwise)."""
        n = rays.shape[0]
        if float(raw_noise_std) > 0.0 and n > 0:
            from .render import noise_line_integral
            target = target - noise_line_integral(rays, self.sample_depths(rays, t_rand, ray_base), raw_noise_std, noise)
        if self.dp_mode == "levels" and self.process_group is not None:
            self._train_step_levels(rays, target, weight, t_rand, ray_base, rays_all, global_ray_base)
        elif self.fuse_table_adam and self._dp is None and (self.n_streams == 1 or n <= self.chunk_rays) and n > 0:
            self._train_step_fused_adam(rays, target, weight, t_rand, ray_base, next_draw)
            next_draw = None
        else:
            self.backward(rays, target, weight, t_rand, ray_base)
            if self._dp is not None and self.dp_mode == "sharded":
                self._exchange_and_step_sharded()
            elif self._dp is not None:
                self._exchange_and_step()
            else:
                self.optimizer_step()
        if next_draw is not None:
            next_draw.launch()                                 # every other route: the draw as a launch of its own behind the step
        self.rays_seen += n
        return self.loss

    def _train_step_fused_adam(self, rays, target, weight, t_rand, ray_base, next_draw=None):
        """backward() + optimizer_step() in ONE library call: naf_render_train_adam -- the gradient reducer finishes every table
        row with its Adam update, the slab reduction of the MLP gradient does the same for the 4 225 MLP parameters."""
        n = rays.shape[0]
        if self.acc is None or self.acc.numel() < n:
            self.acc = torch.empty(n, device=self.device)
        cfg = self._cfg(ray_base)                              # the jitter seed of step k, as in backward()  (the call overwrites self.loss)
        self.step_count += 1                                   # ... and the Adam step count k + 1, as in optimizer_step()
        ws = fused.workspace(cfg, n * self.n_samples, self.device)
        b1, b2 = self.betas
        st = _abi.TableAdam()
        st.param, st.exp_avg, st.exp_avg_sq = self.emb.data_ptr(), self.emb_m.data_ptr(), self.emb_v.data_ptr()
        st.param_lp = None if self.emb_lp is None else self.emb_lp.data_ptr()
        st.lp_dtype = 0 if self.emb_lp is None else _abi.dtype_code(self.table_dtype)
        st.n, st.lr, st.beta1, st.beta2, st.eps, st.step, st.grad_scale = self.emb.numel(), self.lr, b1, b2, self.eps, self.step_count, 1.0
        st.mlp_param, st.mlp_exp_avg, st.mlp_exp_avg_sq = self.mlp.data_ptr(), self.mlp_m.data_ptr(), self.mlp_v.data_ptr()
        args = (_abi.ptr(rays), _abi.ptr(t_rand), _abi.ptr(target), _abi.ptr(weight), _abi.ptr(self.table), _abi.ptr(self.offsets),
                _abi.ptr(self.mlp), _abi.ptr(self.acc), _abi.ptr(self.emb_g), _abi.ptr(self.mlp_g), _abi.ptr(self.loss), n,
                ctypes.byref(cfg), _abi.ptr(ws), ctypes.byref(st))
        if next_draw is None:
            _abi.check(_abi.lib().naf_render_train_adam(*args, _abi.stream_ptr()), "render_train_adam")
        else:
            _abi.check(_abi.lib().naf_render_train_adam_draw(*args, ctypes.byref(next_draw), _abi.stream_ptr()), "render_train_adam_draw")
        fused._bump(self.device)                               # (the MLP's update rode on the slab reduction of that call)

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
