"""Trainer: host-side mirror of reference src/trainer.py (`Trainer(cfg, device)`, `start`, `train_step`, overridable
`compute_loss(data, global_step, idx_epoch)` / `eval_step(global_step, idx_epoch)`; attributes `net`, `net_fine`,
`optimizer`, `writer`, `eval_dset`, `conf`, `n_rays`, `netchunk`, `evaldir` used by train.py).

Two back ends, selected by the optional `backend.engine` key of the YAML (default "fused"):
  * "fused":  one `naf_render_train` + fused Adam per step (engine.NAFEngine); `compute_loss` is not called, the masked
              chunk-mean loss of train.py:69-127 is expressed as per-ray weights (loss.chunk_mean_weights);
  * "module": `optimizer.zero_grad(); loss = compute_loss(...); loss.backward(); optimizer.step()` exactly like
              trainer.py:134-142, autograd running through the HIP operators.
Checkpoints keep the reference layout {"epoch", "network", "network_fine", "optimizer"} (trainer.py:118-126).
"""
from __future__ import annotations

import json
import os
import os.path as osp
from shutil import copyfile

import torch

from .dataset import TIGREDataset as Dataset
from .encoder import get_encoder
from .engine import NAFEngine
from .loss import chunk_mean_weights
from .network import get_network
from .utils import get_ptycho_mask

try:                                                       # tensorboard is optional in this image
    from torch.utils.tensorboard import SummaryWriter
except Exception:                                          # pragma: no cover
    SummaryWriter = None


class _NullWriter:
    def add_scalar(self, *a, **k):
        pass

    add_text = add_image = add_scalar


class _EngineOptimizer:
    """The slice of the torch.optim.Optimizer surface train.py / the scheduler touch, backed by the fused engine."""

    def __init__(self, engine):
        self.engine = engine
        self.param_groups = [{"lr": engine.lr, "betas": engine.betas, "eps": engine.eps}]

    def sync_lr(self):
        self.engine.lr = float(self.param_groups[0]["lr"])

    def zero_grad(self):
        pass                                               # gradients are cleared inside the fused Adam pass

    def state_dict(self):
        return self.engine.optimizer_state_dict()

    def load_state_dict(self, sd):
        self.engine.load_optimizer_state_dict(sd)
        self.param_groups[0]["lr"] = self.engine.lr


class _StepLR:
    """torch.optim.lr_scheduler.StepLR in its chainable form, for either optimiser: every `step_size`-th call multiplies
    the learning rate THE OPTIMISER CURRENTLY HOLDS by gamma.  The rate is therefore part of the optimiser state: a
    resumed run continues from the decayed rate its checkpoint stored (trainer.py:54-66 -- the reference builds the
    scheduler before it loads the checkpoint and never restores the scheduler itself, so its counter restarts at 0)."""

    def __init__(self, optimizer, step_size, gamma):
        self.optimizer, self.step_size, self.gamma = optimizer, int(step_size), float(gamma)
        self.last_epoch = 0

    def step(self):
        self.last_epoch += 1
        if self.last_epoch % self.step_size == 0:
            for g in self.optimizer.param_groups:
                g["lr"] = g["lr"] * self.gamma
        if hasattr(self.optimizer, "sync_lr"):
            self.optimizer.sync_lr()


_DTYPES = {"float32": torch.float32, "bfloat16": torch.bfloat16, "float16": torch.float16}


class Trainer:
    def __init__(self, cfg, device="cuda"):
        self.global_step = 0
        self.conf = cfg
        self.device = torch.device(device)
        # data parallel when launched one process per GPU (torchrun contract: RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*);
        # rays of a step are sharded over the ranks, the model is replicated
        self.rank, self.world, self.group = 0, 1, None
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            from . import dist as naf_dist
            self.rank, self.world, _, self.group = naf_dist.init_from_env(self.device.type)
        self.n_fine = cfg["render"]["n_fine"]
        self.epochs = cfg["train"]["epoch"]
        self.i_eval = cfg["log"]["i_eval"]
        self.i_save = cfg["log"]["i_save"]
        self.netchunk = cfg["render"]["netchunk"]
        self.n_rays = cfg["train"]["n_rays"]
        backend = cfg.get("backend", {}) or {}
        self.loss_mode = backend.get("loss", "chunk_sum")
        self.i_log = int(cfg["log"].get("i_log", 100))      # steps between train/loss scalars (reading the loss synchronises)
        self._plain_weights = {}
        self._even_shards = cfg["train"]["n_rays"] % max(self.world, 1) == 0     # every rank then holds n_rays / world rays

        self.expdir = osp.join(cfg["exp"]["expdir"], cfg["exp"]["expname"])
        self.ckptdir = osp.join(self.expdir, "ckpt.tar")
        self.ckptdir_backup = osp.join(self.expdir, "ckpt_backup.tar")
        self.evaldir = osp.join(self.expdir, "eval")
        os.makedirs(self.evaldir, exist_ok=True)

        data = cfg["exp"]["datadir"]
        train_dset = Dataset(data, cfg["train"]["n_rays"], "train", device, shard=(self.rank, self.world),
                             seed=backend.get("seed"))
        self.train_dset = train_dset
        self.eval_dset = Dataset(data, cfg["train"]["n_rays"], "val", device) if self.i_eval > 0 else None
        self.train_dloader = torch.utils.data.DataLoader(train_dset, batch_size=cfg["train"]["n_batch"])

        if backend.get("seed") is not None:                # optional: reproducible initialisation (the reference seeds nothing)
            torch.manual_seed(int(backend["seed"]))
        network = get_network(cfg["network"]["net_type"])
        net_cfg = {k: v for k, v in cfg["network"].items() if k != "net_type"}
        encoder = get_encoder(**cfg["encoder"])
        self.net = network(encoder, **net_cfg).to(device)
        grad_vars = list(self.net.parameters())
        self.net_fine = None
        if self.n_fine > 0:
            self.net_fine = network(encoder, **net_cfg).to(device)
            grad_vars += list(self.net_fine.parameters())

        self.engine = None
        want_fused = backend.get("engine", "fused") == "fused"
        # raw_noise_std > 0 (render.py:196-199) is an additive term on the line integral (render.noise_line_integral): the fused engine
        # handles it on the target side, the kernels stay noise-free
        self.raw_noise_std = float(cfg["render"].get("raw_noise_std", 0.0))
        if want_fused and self.n_fine == 0 and self.net.fused_supported() and self.device.type == "cuda":
            self.engine = NAFEngine(self.net, cfg["render"]["n_samples"], perturb=cfg["render"]["perturb"],
                                    lr=cfg["train"]["lrate"], betas=(0.9, 0.999),
                                    table_dtype=_DTYPES[backend.get("table_dtype", "float32")], process_group=self.group,
                                    # 'sharded' (gradient reduce-scatter / per-rank Adam / all-gather) unless the YAML's backend.dp_mode asks
                                    # for 'levels' or 'auto' (dist.pick_dp_mode): the level-parallel step is opt-in until it has run over
                                    # RCCL on more than one rank; level-parallel ranks need equal shards
                                    dp_mode=backend.get("dp_mode", "sharded"),
                                    rays_per_step_hint=-(-int(cfg["train"]["n_rays"]) // max(self.world, 1)))
            self.engine.broadcast_parameters()
            self.optimizer = _EngineOptimizer(self.engine)
        elif self.group is not None:
            raise NotImplementedError("data-parallel training needs the fused engine (backend.engine: fused, n_fine: 0)")
        else:
            if want_fused and self.device.type == "cuda":      # say so instead of silently taking the slower path
                why = "n_fine > 0 (two networks)" if self.n_fine > 0 else \
                    "network / encoder shape outside the fused kernels (32 features, 4 layers of 32, skips [2], 1 output)"
                print(f"[Trainer] {why}: using the module back end (autograd over the HIP operators + torch Linear layers)")
            self.optimizer = torch.optim.Adam(params=grad_vars, lr=cfg["train"]["lrate"], betas=(0.9, 0.999))
        self.lr_scheduler = _StepLR(self.optimizer, cfg["train"]["lrate_step"], cfg["train"]["lrate_gamma"])

        self.epoch_start = 0
        if cfg["train"]["resume"] and osp.exists(self.ckptdir):
            print(f"Load checkpoints from {self.ckptdir}.")
            ckpt = torch.load(self.ckptdir, map_location=device, weights_only=False)
            self.epoch_start = ckpt["epoch"] + 1
            self.net.load_state_dict(ckpt["network"])
            if self.n_fine > 0:
                self.net_fine.load_state_dict(ckpt["network_fine"])
            self.optimizer.load_state_dict(ckpt["optimizer"])
            if self.engine is not None:
                self.engine.sync_from_module()
            self.global_step = self.epoch_start * len(self.train_dloader)
            # the pixel draws are keyed by (seed, item count): a resumed run continues the sequence instead of replaying epoch 0's
            self.train_dset.set_draw_position(self.global_step)

        self.writer = SummaryWriter(self.expdir) if (SummaryWriter is not None and self.rank == 0) else _NullWriter()
        self.writer.add_text("parameters", self.args2string(cfg), global_step=0)

    @property
    def voxels(self):
        """The [n1, n2, n3, 3] voxel-centre list of trainer.py:41 -- built when somebody asks for it: the fused volume query
        generates the grid inside the kernel (`field_query_grid`), and at foot_50's 1024^3 the list alone is 12.9 GB."""
        return self.eval_dset.voxels if self.eval_dset is not None else None

    def args2string(self, hp):
        json_hp = json.dumps(hp, indent=2, default=str)
        return "".join("\t" + line for line in json_hp.splitlines(True))

    # ------------------------------------------------------------------------------------------------------
    def start(self):
        iter_per_epoch = len(self.train_dloader)
        try:
            from tqdm import tqdm
            pbar = tqdm(total=iter_per_epoch * self.epochs, leave=True, disable=self.rank != 0)
            pbar.update(self.epoch_start * iter_per_epoch)
        except Exception:                                  # pragma: no cover
            pbar = None

        for idx_epoch in range(self.epoch_start, self.epochs + 1):
            if self.group is not None and self.engine is not None and self.i_eval > 0 and (idx_epoch % self.i_eval == 0 or idx_epoch == self.epochs):
                # a sharded optimiser / level-parallel run keeps master table and moments current on their owner only: complete them
                # everywhere (collective) before rank 0 reads the parameters
                self.engine.gather_state()
            if self.i_eval > 0 and self.rank == 0 and (idx_epoch % self.i_eval == 0 or idx_epoch == self.epochs):
                self.net.eval()
                with torch.no_grad():
                    loss_test = self.eval_step(global_step=self.global_step, idx_epoch=idx_epoch)
                self.net.train()
                msg = "".join(f", {k}: {float(v):.3g}" for k, v in loss_test.items())
                print(f"[EVAL] epoch: {idx_epoch}/{self.epochs}{msg}")
            if self.group is not None and self.i_eval > 0 and (idx_epoch % self.i_eval == 0 or idx_epoch == self.epochs):
                # rank 0 evaluated alone: the other ranks wait HERE (a host-side barrier on a group with a long timeout of its own,
                # dist.wait_for_rank0), not inside the first gradient exchange of the next step
                from . import dist as naf_dist
                naf_dist.wait_for_rank0(self.group)

            for data in self._batches():
                self.global_step += 1
                self.net.train()
                loss_train = self.train_step(data, global_step=self.global_step, idx_epoch=idx_epoch)
                if pbar is not None:
                    if self.global_step % 50 == 0:         # .item() syncs; the reference does it every step
                        pbar.set_description(f"epoch={idx_epoch}/{self.epochs}, loss={float(loss_train):.3g}, "
                                             f"lr={self.optimizer.param_groups[0]['lr']:.3g}")
                    pbar.update(1)

            if self.i_save > 0 and idx_epoch > 0 and (idx_epoch % self.i_save == 0 or idx_epoch == self.epochs):
                self.save_checkpoint(idx_epoch)

            self.writer.add_scalar("train/lr", self.optimizer.param_groups[0]["lr"], self.global_step)
            self.lr_scheduler.step()
        print(f"Training complete! See logs in {self.expdir}")

    def _batches(self):
        """One epoch of training batches.  The reference iterates a DataLoader with batch_size = n_batch = 1 over a dataset
        whose items already are whole ray batches (trainer.py:32-37); with the fused engine the items are consumed directly --
        the collate step would only copy every tensor once more to add a batch dimension of one."""
        if self.engine is not None and self.conf["train"]["n_batch"] == 1:
            dset = self.train_dset
            return (dset[i] for i in range(len(dset)))
        return iter(self.train_dloader)

    def save_checkpoint(self, idx_epoch):
        if self.group is not None and self.engine is not None:
            self.engine.gather_state()                     # collective: every rank enters save_checkpoint
        if self.rank != 0:                                 # replicas are identical; rank 0 writes
            return
        if osp.exists(self.ckptdir):
            copyfile(self.ckptdir, self.ckptdir_backup)
        print(f"[SAVE] epoch: {idx_epoch}/{self.epochs}, path: {self.ckptdir}")
        torch.save({"epoch": idx_epoch, "network": self.net.state_dict(),
                    "network_fine": self.net_fine.state_dict() if self.n_fine > 0 else None,
                    "optimizer": self.optimizer.state_dict()}, self.ckptdir)

    # ------------------------------------------------------------------------------------------------------
    def ray_weights(self, data, n):
        """Per-ray loss weights: ptycho mask of the full projection sampled at the ray pixels (train.py:59-60,93-95,
        intended semantics of SURVEY.md App. A-6) and the reference's chunked mean (App. A-5)."""
        full_proj = data.get("full_proj")
        if data.get("mask") is not None:                   # our dataset caches the mask per projection
            mask = data["mask"].reshape(-1)
        elif full_proj is not None:
            mask_full = get_ptycho_mask(full_proj.reshape(full_proj.shape[-2:]), threshold=0.007)
            coords = data["coords"].reshape(-1, 2).long()
            mask = mask_full[coords[:, 0], coords[:, 1]]
        else:                                              # no mask: the weights depend on the batch size only
            key = (n, self.loss_mode, self.world)
            if key not in self._plain_weights:
                ones = torch.ones(n, dtype=torch.bool, device=self.device)
                if self.group is not None:
                    w = ones.float() / float(n * self.world) if self._even_shards else None
                else:
                    w = chunk_mean_weights(ones, 200, self.loss_mode)
                self._plain_weights[key] = w
            if self._plain_weights[key] is not None:
                return self._plain_weights[key]
            mask = torch.ones(n, dtype=torch.bool, device=self.device)
        if self.group is not None:                         # global masked mean over all ranks' rays
            from .dist import global_mean_weights
            return global_mean_weights(mask, self.group)
        return chunk_mean_weights(mask, 200, self.loss_mode)

    def train_step(self, data, global_step, idx_epoch):
        if self.engine is not None:
            rays = data["rays"].reshape(-1, 8)
            projs = data["projs"].reshape(-1)
            if projs.dtype != torch.float32:
                projs = projs.float()
            weight = self.ray_weights(data, rays.shape[0])
            n = rays.shape[0]
            loss = self.engine.train_step(rays, projs, weight, ray_base=(global_step * self.world + self.rank) * n,
                                          raw_noise_std=self.raw_noise_std)
            if self.i_log > 0 and global_step % self.i_log == 0:
                self.writer.add_scalar("train/loss", float(loss), global_step)
            return loss
        self.optimizer.zero_grad()
        loss = self.compute_loss(data, global_step, idx_epoch)
        loss.backward()
        self.optimizer.step()
        return loss.detach()

    def compute_loss(self, data, global_step, idx_epoch):
        raise NotImplementedError()

    def eval_step(self, global_step, idx_epoch):
        raise NotImplementedError()
