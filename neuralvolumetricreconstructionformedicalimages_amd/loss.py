"""Training loss: mirror of the part of reference src/loss/loss.py that train.py uses (`calc_mse_loss`, :26-46)
plus the ray weights that express the reference's chunked, masked mean for the fused engine."""
from __future__ import annotations

import torch


def calc_mse_loss(loss, x, y, tv_loss=None):
    """loss["loss"] += mean((x-y)^2); loss["loss_mse"] = that mean (loss.py:36-44)."""
    loss_mse = torch.mean((x - y) ** 2)
    loss["loss"] += loss_mse
    loss["loss_mse"] = loss_mse
    if tv_loss is not None:
        loss["loss"] += tv_loss
        loss["tv_loss"] = tv_loss
    return loss


def chunk_mean_weights(mask, chunk_size=200, mode="chunk_sum"):
    """Per-ray weights w with  sum_r w_r (acc_r - y_r)^2  ==  the reference loss.

    mode "chunk_sum": the reference sums, over consecutive `chunk_size`-ray chunks, the mean over the masked rays of
    each chunk (train.py:69,127 + loss.py:37-38) -> w_r = mask_r / (#masked rays in r's chunk).
    mode "global_mean": one mean over all masked rays (the data-parallel definition)."""
    m = mask.float()
    if mode == "global_mean":
        return m / m.sum().clamp(min=1.0)
    n = m.shape[0]
    pad = (-n) % chunk_size
    mp = torch.cat([m, m.new_zeros(pad)]) if pad else m
    per_chunk = mp.view(-1, chunk_size).sum(1, keepdim=True).clamp(min=1.0)
    return (mp.view(-1, chunk_size) / per_chunk).reshape(-1)[:n]
