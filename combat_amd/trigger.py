"""Host-side constants of the trigger kernels (reference: train_generator.py:47-55, 165;
utils/dct.py:13-111): the orthonormal DCT-II matrix, the low-pass projector
``P = D[:K].T @ D[:K]`` (``low_freq(X) == P X P^T``) and the normalised 3-tap Gaussian whose
sigma torchvision's GaussianBlur draws once per call from torch's global generator."""
from __future__ import annotations

import math

import numpy as np
import torch


def dct_matrix(n: int) -> torch.Tensor:
    k = torch.arange(n, dtype=torch.float64)[:, None]
    i = torch.arange(n, dtype=torch.float64)[None, :]
    d = torch.cos(math.pi * (2 * i + 1) * k / (2 * n)) * math.sqrt(2.0 / n)
    d[0] *= math.sqrt(0.5)
    return d


def lowpass_matrix(n: int, ratio: float) -> torch.Tensor:
    d = dct_matrix(n)
    k = int(n * ratio)
    return (d[:k].T @ d[:k]).float()


def sample_sigma(sigma=(0.1, 1.0)) -> float:
    return float(torch.empty(1).uniform_(float(sigma[0]), float(sigma[1])).item())


def gaussian_kernel1d(sigma: float, kernel_size: int = 3) -> np.ndarray:
    half = (kernel_size - 1) * 0.5
    xs = np.linspace(-half, half, kernel_size, dtype=np.float32)
    pdf = np.exp(-0.5 * (xs / np.float32(sigma)) ** 2).astype(np.float32)
    return pdf / pdf.sum(dtype=np.float32)
