"""oracle/loss_metrics_ref.py -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the loss / metric helpers on the training path:
  * calc_mse_loss ............ src/loss/loss.py:26-46
  * get_mse / get_psnr ....... src/utils/util.py:18-51
  * get_psnr_3d .............. src/utils/util.py:55-84     (the PSNR of the +-0.1 dB bar)
  * get_ptycho_mask .......... src/utils/util.py:196-205
  * chunked masked loss ...... train.py:48-135 with the intended semantics of SURVEY App. A-5/A-6
Pinned against golden vectors captured from the imported reference functions (tests/golden/make_golden.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import numpy as np
import torch


def calc_mse_loss(loss, x, y, tv_loss=None):
    mse = torch.mean((x - y) ** 2)
    loss["loss"] = loss["loss"] + mse
    loss["loss_mse"] = mse
    if tv_loss is not None:
        loss["loss"] = loss["loss"] + tv_loss
        loss["tv_loss"] = tv_loss
    return loss


def get_mse(x, y):
    if torch.is_complex(x) and torch.is_complex(y):
        return torch.mean((x.real - y.real) ** 2 + (x.imag - y.imag) ** 2)
    return torch.mean((x - y) ** 2)


def get_psnr(x, y):
    x, y = torch.abs(x), torch.abs(y)
    if torch.max(x) == 0 or torch.max(y) == 0:
        return torch.zeros(1, device=x.device)
    xn = (x - x.min()) / (x.max() - x.min())
    yn = (y - y.min()) / (y.max() - y.min())
    return -10.0 * torch.log10(get_mse(xn, yn))


def get_psnr_3d(a, b, size_average=True, pixel_max=1.0):
    if torch.is_tensor(a):
        a = a.detach().cpu().numpy()
    if torch.is_tensor(b):
        b = b.detach().cpu().numpy()
    a = a[np.newaxis].astype(np.float64)
    b = b[np.newaxis].astype(np.float64)
    mse = ((a - b) ** 2).mean(axis=(1, 2, 3))
    zero = mse == 0
    mse[zero] = 1e-10
    psnr = 20 * np.log10(pixel_max / np.sqrt(mse))
    psnr[zero] = 100
    return psnr.mean() if size_average else psnr


def get_ptycho_mask(hr, threshold=0.007):
    with torch.no_grad():
        m = torch.abs(hr) < threshold
        m[1:] &= m[1:] == m[:-1]
        m[:, 1:] &= m[:, 1:] == m[:, :-1]
        return ~m


def chunked_masked_loss(pred, target, mask=None, chunk_size=200):
    """train.py:69,127 + loss.py:37-38: loss = sum over ray chunks of the per-chunk masked mean."""
    total = 0.0
    for i in range(0, pred.shape[0], chunk_size):
        p, t = pred[i:i + chunk_size], target[i:i + chunk_size]
        if mask is not None:
            m = mask[i:i + chunk_size].bool()
            p, t = p[m], t[m]
        total = total + torch.mean((t - p) ** 2)
    return total
