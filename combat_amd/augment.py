"""Host-side parameter sampling for the on-device PostTensorTransform
(reference: utils/dataloader.py:11-21, 45-60; kornia 0.6.6 RandomCrop / RandomRotation /
RandomHorizontalFlip -- absent library, semantics restated; parity unpinned).

The reference gates the crop and the rotation for the *whole batch* with Python's ``random``
(``ProbTransform``, p = 0.8 and 0.5) and lets kornia draw per-sample parameters from torch's
global generator: crop offset uniform over the zero-padded window, rotation applied to each
sample with probability 0.5 at an angle uniform in [-deg, deg], flip with probability 0.5
(CIFAR-10 only).  The device kernel takes one fp32 row per sample:
(crop_dx - pad, crop_dy - pad, angle in radians, flip).
"""
from __future__ import annotations

import math
import random
from typing import Optional

import numpy as np
import torch


class PostTensorTransform:
    """Same constructor contract as the reference class; ``sample(n)`` returns the [n, 4] fp32
    parameter table of one call, or None when the call is the identity."""

    def __init__(self, opt):
        self.option = getattr(opt, "post_transform_option", "use")
        self.crop_pad = int(getattr(opt, "random_crop", 5))
        self.degrees = float(getattr(opt, "random_rotation", 10))
        self.flip = getattr(opt, "dataset", "cifar10") == "cifar10"
        self.p_crop, self.p_rot = 0.8, 0.5

    def sample(self, n: int, generator: Optional[torch.Generator] = None) -> Optional[np.ndarray]:
        if self.option == "no_use":
            return None
        out = np.zeros((n, 4), np.float32)
        if self.option != "use_modified" and random.random() < self.p_crop:
            off = torch.randint(0, 2 * self.crop_pad + 1, (n, 2), generator=generator).numpy()
            out[:, 0:2] = off - self.crop_pad
        if random.random() < self.p_rot:
            apply = torch.rand(n, generator=generator).numpy() < 0.5
            ang = (torch.rand(n, generator=generator).numpy() * 2 - 1) * self.degrees
            out[:, 2] = np.where(apply, ang, 0.0) * (math.pi / 180.0)
        if self.flip:
            out[:, 3] = (torch.rand(n, generator=generator).numpy() < 0.5).astype(np.float32)
        return out


def params_from_oracle_struct(p) -> np.ndarray:
    """tests: convert an oracle ``AugParams`` to the kernel's table."""
    return np.stack([p.crop_dx - p.pad, p.crop_dy - p.pad, np.radians(p.angle_deg), p.flip], 1).astype(np.float32)
