"""Metrics and masks: mirror of the reference src/utils/util.py functions on the training / evaluation path:
`get_mse` (:18-26), `get_psnr` (:29-51), `get_psnr_3d` (:55-84, the PSNR of the +-0.1 dB bar), `cast_to_image`
(:155-170), `get_ptycho_mask` (:196-205).  `get_ssim_3d` (:87-139) needs scikit-image and is optional."""
from __future__ import annotations

import numpy as np
import torch


def get_mse(x, y):
    if torch.is_complex(x) and torch.is_complex(y):
        return torch.mean((x.real - y.real) ** 2 + (x.imag - y.imag) ** 2)
    return torch.mean((x - y) ** 2)


def get_psnr(x, y):
    x = torch.abs(x)
    y = torch.abs(y)
    if torch.max(x) == 0 or torch.max(y) == 0:
        return torch.zeros(1, device=x.device)
    x_norm = (x - torch.min(x)) / (torch.max(x) - torch.min(x))
    y_norm = (y - torch.min(y)) / (torch.max(y) - torch.min(y))
    return -10.0 * torch.log10(get_mse(x_norm, y_norm))


def get_psnr_3d(arr1, arr2, size_average=True, PIXEL_MAX=1.0):
    if torch.is_tensor(arr1):
        arr1 = arr1.cpu().detach().numpy()
    if torch.is_tensor(arr2):
        arr2 = arr2.cpu().detach().numpy()
    arr1 = arr1[np.newaxis, ...].astype(np.float64)
    arr2 = arr2[np.newaxis, ...].astype(np.float64)
    mse = np.power(arr1 - arr2, 2).mean(axis=1).mean(axis=1).mean(axis=1)
    zero_mse = np.where(mse == 0)
    mse[zero_mse] = 1e-10
    psnr = 20 * np.log10(PIXEL_MAX / np.sqrt(mse))
    psnr[zero_mse] = 100
    return psnr.mean() if size_average else psnr


def get_ssim_3d(arr1, arr2, size_average=True, PIXEL_MAX=1.0):
    """Mean of the slice-wise SSIM along the three axes (util.py:87-139); needs scikit-image."""
    try:
        from skimage.metrics import structural_similarity
    except ImportError as e:                                   # not installed in this image
        raise RuntimeError("get_ssim_3d needs scikit-image (skimage.metrics.structural_similarity)") from e
    if torch.is_tensor(arr1):
        arr1 = arr1.cpu().detach().numpy()
    if torch.is_tensor(arr2):
        arr2 = arr2.cpu().detach().numpy()
    a = arr1[np.newaxis, ...].astype(np.float64)
    b = arr2[np.newaxis, ...].astype(np.float64)
    views = [((0, 2, 3, 1), (0, 2, 3, 1)), ((0, 1, 3, 2), (0, 1, 3, 2)), ((0, 1, 2, 3), (0, 1, 2, 3))]
    total = 0.0
    for pa, pb in views:
        total = total + np.asarray([structural_similarity(x, y) for x, y in zip(np.transpose(a, pa), np.transpose(b, pb))])
    ssim = total / 3
    return ssim.mean() if size_average else ssim


def cast_to_image(tensor, normalize=True):
    """Tensor [H,W] (possibly complex) -> float numpy [H,W,1] in [0,1] (util.py:155-170)."""
    if torch.is_tensor(tensor):
        img = tensor.abs() if torch.is_complex(tensor) else tensor
        img = img.detach().cpu().numpy()
    else:
        img = np.abs(tensor)
    if normalize:
        lo, hi = img.min(), img.max()
        img = (img - lo) / (hi - lo) if hi > lo else np.zeros_like(img)
    return img[..., np.newaxis]


def get_ptycho_mask(hr, threshold=0.007):
    """True where the (complex) projection carries signal: the complement of the 4-connected-smoothed
    |hr| < threshold region (util.py:196-205)."""
    with torch.no_grad():
        mask = torch.abs(hr) < threshold
        mask[1:] &= mask[1:] == mask[:-1]
        mask[:, 1:] &= mask[:, 1:] == mask[:, :-1]
        return ~mask
