"""Metrics and masks: mirror of the reference src/utils/util.py functions on the training / evaluation path:
`get_mse` (:18-26), `get_psnr` (:29-51), `get_psnr_3d` (:55-84, the PSNR of the +-0.1 dB bar), `cast_to_image`
(:155-170), `get_ptycho_mask` (:196-205).  `get_ssim_3d` (:87-139) needs scikit-image and is optional."""
from __future__ import annotations

import numpy as np
import torch


def get_mse(x, y):
    """Mean squared error; complex inputs count real and imaginary parts (util.py:18-26)."""
    diff = x - y
    if torch.is_complex(diff):
        return (diff.real.square() + diff.imag.square()).mean()
    return diff.square().mean()


def _unit_range(t):
    lo, hi = t.min(), t.max()
    return (t - lo) / (hi - lo)


def get_psnr(x, y):
    """PSNR of two projections after each is stretched to [0, 1] on its own (util.py:29-51); 0 if either is all zero."""
    x, y = x.abs(), y.abs()
    if x.max() == 0 or y.max() == 0:
        return torch.zeros(1, device=x.device)
    return -10.0 * torch.log10(get_mse(_unit_range(x), _unit_range(y)))


def _as_volume_batch(a):
    if torch.is_tensor(a):
        a = a.detach().cpu().numpy()
    return np.asarray(a, dtype=np.float64)[np.newaxis]


def get_psnr_3d(arr1, arr2, size_average=True, PIXEL_MAX=1.0):
    """Volume PSNR in dB, 20 log10(PIXEL_MAX / rmse) in float64; identical volumes score 100 (util.py:55-84).
    This is the PSNR of BASELINE.json's +-0.1 dB bar."""
    a, b = _as_volume_batch(arr1), _as_volume_batch(arr2)
    mse = np.square(a - b).reshape(a.shape[0], -1).mean(axis=1)
    psnr = np.full(mse.shape, 100.0)
    hit = mse > 0
    psnr[hit] = 20.0 * np.log10(PIXEL_MAX / np.sqrt(mse[hit]))
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
