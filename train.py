#!/usr/bin/env python3
"""python train.py --config config/chest_50.yaml   -- entry point with the reference's CLI (train.py:19-28, 292).

`BasicTrainer.compute_loss` follows train.py:48-135 with the semantics the reference intends:
200-ray chunks, ptycho mask of the full projection sampled at the ray pixels, sum of per-chunk masked means.
`BasicTrainer.eval_step` follows train.py:220-288: render one validation projection, query the whole volume, report
projection MSE/PSNR and volume PSNR (SSIM when scikit-image is present), dump arrays under <expdir>/eval/."""
import argparse
import os
import os.path as osp

import numpy as np
import torch

from neuralvolumetricreconstructionformedicalimages_amd.config import load_config
from neuralvolumetricreconstructionformedicalimages_amd.fused import field_query_grid
from neuralvolumetricreconstructionformedicalimages_amd.loss import calc_mse_loss
from neuralvolumetricreconstructionformedicalimages_amd.render import render, run_network
from neuralvolumetricreconstructionformedicalimages_amd.trainer import Trainer
from neuralvolumetricreconstructionformedicalimages_amd.utils import (get_mse, get_psnr, get_psnr_3d, get_ptycho_mask,
                                                                      get_ssim_3d)


def config_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument("--config", default="./config/abdomen_50.yaml", help="configs file path")
    return parser


class BasicTrainer(Trainer):
    def __init__(self, cfg, device):
        super().__init__(cfg, device)
        print(f"[Start] exp: {cfg['exp']['expname']}, net: Basic network")

    def compute_loss(self, data, global_step, idx_epoch):
        rays = data["rays"].reshape(-1, 8).to(self.device)
        projs = data["projs"].reshape(-1).to(self.device)
        full_proj = data.get("full_proj")
        chunk_size = 200
        loss = {"loss": 0.0}
        mask_full = None
        if full_proj is not None:
            mask_full = get_ptycho_mask(full_proj.reshape(full_proj.shape[-2:]), threshold=0.007).to(self.device)
            coords = data["coords"].reshape(-1, 2).long()
        for i in range(0, rays.shape[0], chunk_size):
            ret = render(rays[i:i + chunk_size], self.net, self.net_fine, n_samples=self.conf["render"]["n_samples"],
                         n_fine=self.conf["render"]["n_fine"], perturb=self.conf["render"]["perturb"],
                         netchunk=self.conf["render"]["netchunk"], raw_noise_std=self.conf["render"]["raw_noise_std"],
                         chunk_size=chunk_size)
            pred = ret["acc"].reshape(-1)
            target = projs[i:i + chunk_size]
            if mask_full is not None:
                c = coords[i:i + chunk_size]
                m = mask_full[c[:, 0], c[:, 1]]
                calc_mse_loss(loss, target[m], pred[m])
            else:
                calc_mse_loss(loss, target, pred)
        for ls in loss.keys():
            self.writer.add_scalar(f"train/{ls}", float(loss[ls].detach()) if torch.is_tensor(loss[ls]) else float(loss[ls]), global_step)
        return loss["loss"]

    def eval_step(self, global_step, idx_epoch):
        select_ind = np.random.choice(len(self.eval_dset))
        projs = self.eval_dset.projs[select_ind].to(self.device)
        rays = self.eval_dset.rays[select_ind].reshape(-1, 8)
        H, W = projs.shape
        projs_pred = []
        # The reference renders the projection n_rays at a time (train.py:235-238), each chunk with its own launches and a
        # host-side NaN check.  The fused path takes 65 536 rays per call instead: the same samples of the same rays
        # (jitter is drawn per ray), 4 calls instead of 256 for a 512 x 512 projection.
        fused = getattr(self.net, "fused_supported", lambda: False)() and self.net_fine is None
        step, inner = (max(self.n_rays, 1 << 16), None) if fused else (self.n_rays, 1024)
        for i in range(0, rays.shape[0], step):
            projs_pred.append(render(rays[i:i + step], self.net, self.net_fine, **self.conf["render"], chunk_size=inner)["acc"])
        projs_pred = torch.cat(projs_pred, 0).reshape(H, W)

        image = self.eval_dset.image
        net_eval = self.net_fine if self.net_fine is not None else self.net
        if getattr(net_eval, "fused_supported", lambda: False)() and hasattr(self.eval_dset, "voxel_axes"):
            # the voxel grid is generated inside the kernel instead of being read as an [n^3, 3] point list (same values)
            image_pred = field_query_grid(net_eval, *self.eval_dset.voxel_axes)
        else:
            image_pred = run_network(self.eval_dset.voxels, net_eval, self.netchunk).squeeze()
        loss = {"proj_mse": get_mse(projs_pred, projs), "proj_psnr": get_psnr(projs_pred, projs),
                "psnr_3d": get_psnr_3d(image_pred, image)}
        try:
            loss["ssim_3d"] = get_ssim_3d(image_pred, image)
        except RuntimeError:
            pass                                           # scikit-image is not installed
        for ls in loss.keys():
            self.writer.add_scalar(f"eval/{ls}", float(loss[ls]), global_step)
        eval_save_dir = osp.join(self.evaldir, f"epoch_{idx_epoch:05d}")
        os.makedirs(eval_save_dir, exist_ok=True)
        np.save(osp.join(eval_save_dir, "image_pred.npy"), image_pred.cpu().detach().numpy())
        np.save(osp.join(eval_save_dir, "image_gt.npy"), image.cpu().detach().numpy())
        np.save(osp.join(eval_save_dir, "proj_pred.npy"), projs_pred.cpu().detach().numpy())
        with open(osp.join(eval_save_dir, "stats.txt"), "w") as f:
            for key, value in loss.items():
                f.write("%s: %f\n" % (key, float(value)))
        return loss


if __name__ == "__main__":
    args = config_parser().parse_args()
    cfg = load_config(args.config)
    if not torch.cuda.is_available():
        raise SystemExit("train.py needs an MI355X: the NAF hot path has no CPU fallback")
    # one process per GPU under `python -m torch.distributed.run --nproc-per-node N train.py --config ...`
    from neuralvolumetricreconstructionformedicalimages_amd.dist import local_device_index
    torch.cuda.set_device(local_device_index())
    BasicTrainer(cfg, torch.device("cuda", local_device_index())).start()
