"""Console progress bar (same line format as utils/utils.py:55-94) and a scalar writer with
SummaryWriter's add_scalars/add_image call shape: tensorboard if importable, JSON lines otherwise."""
from __future__ import annotations

import json
import os
import shutil
import sys
import time

TOTAL_BAR_LENGTH = 65.0
_begin = time.time()


def progress_bar(current: int, total: int, msg: str = None) -> None:
    global _begin
    if current == 0:
        _begin = time.time()
    width = shutil.get_terminal_size((160, 24)).columns
    cur = int(TOTAL_BAR_LENGTH * current / total)
    rest = int(TOTAL_BAR_LENGTH - cur) - 1
    out = [" [", "=" * cur, ">", "." * rest, "]"]
    tail = (" | " + msg) if msg else ""
    out.append(tail)
    out.append(" " * max(0, width - int(TOTAL_BAR_LENGTH) - len(tail) - 3))
    out.append("\b" * max(0, width - int(TOTAL_BAR_LENGTH / 2) + 2))
    out.append(" %d/%d " % (current + 1, total))
    out.append("\r" if current < total - 1 else "\n")
    sys.stdout.write("".join(out))
    sys.stdout.flush()


class ScalarWriter:
    def __init__(self, log_dir: str):
        os.makedirs(log_dir, exist_ok=True)
        self.tb = None
        try:
            from torch.utils.tensorboard import SummaryWriter  # noqa: WPS433 (optional dependency)
            self.tb = SummaryWriter(log_dir=log_dir)
        except Exception:
            self.path = os.path.join(log_dir, "scalars.jsonl")

    def add_scalars(self, tag, values, step):
        values = {k: float(v) for k, v in values.items()}
        if self.tb is not None:
            self.tb.add_scalars(tag, values, step)
        else:
            with open(self.path, "a") as f:
                f.write(json.dumps({"tag": tag, "step": int(step), "values": values}) + "\n")

    def add_image(self, tag, img, global_step=0):
        """img: float [3][H][W] in [0, 1] (what make_grid returns).  Without tensorboard the grid is kept as a
        binary PPM under <log_dir>/images/."""
        if self.tb is not None:
            self.tb.add_image(tag, img, global_step=global_step)
            return
        import numpy as np
        a = np.asarray(img.detach().float().cpu().numpy() if hasattr(img, "detach") else img, dtype=np.float32)
        a = (np.clip(a, 0.0, 1.0) * 255.0 + 0.5).astype(np.uint8).transpose(1, 2, 0)
        folder = os.path.join(os.path.dirname(self.path), "images")
        os.makedirs(folder, exist_ok=True)
        with open(os.path.join(folder, "%s_%06d.ppm" % (str(tag).replace("/", "_"), int(global_step))), "wb") as f:
            f.write(b"P6\n%d %d\n255\n" % (a.shape[1], a.shape[0]))
            f.write(np.ascontiguousarray(a).tobytes())


def denormalize(x, opt):
    """networks/models.py:29-44, 66-86: x * std + mean per channel with mean = std = 0.5 for every dataset the
    reference's Denormalizer knows (it raises for others; imagenet10 uses the same 0.5 / 0.5 transform here)."""
    return x * 0.5 + 0.5


def make_grid(batch, nrow: int = 8, padding: int = 2, normalize: bool = True):
    """torchvision.utils.make_grid(batch, normalize=True) as the reference calls it (train_generator.py:314): the
    whole batch min-max scaled to [0, 1] (one range, not per image), images tiled `nrow` per row with `padding`
    pixels of zeros between them.  batch: float [B][C][H][W] -> [C][rows*(H+pad)+pad][cols*(W+pad)+pad]."""
    import math
    import torch
    t = batch.detach().float().cpu().clone()
    if t.dim() == 3:
        t = t[None]
    if t.shape[1] == 1:
        t = t.expand(-1, 3, -1, -1).clone()
    if normalize:
        lo, hi = float(t.min()), float(t.max())
        t = (t.clamp(lo, hi) - lo) / max(hi - lo, 1e-5)
    b, c, h, w = t.shape
    if b == 1:
        return t[0]
    cols = min(nrow, b)
    rows = int(math.ceil(b / cols))
    hh, ww = h + padding, w + padding
    grid = torch.zeros(c, hh * rows + padding, ww * cols + padding)
    for k in range(b):
        y, x = divmod(k, cols)
        grid[:, y * hh + padding:y * hh + padding + h, x * ww + padding:x * ww + padding + w] = t[k]
    return grid


def image_grid(inputs, inputs_bd, opt):
    """The debugging image of the reference loops (train_generator.py:310-314, train_victim_wanet.py:127-133):
    clean images stacked on top of their backdoored copies (dim 2), denormalised, as one grid."""
    import torch
    return make_grid(denormalize(torch.cat([inputs.detach().float().cpu(), inputs_bd.detach().float().cpu()], dim=2), opt),
                     normalize=True)


def SummaryWriter(log_dir: str) -> ScalarWriter:  # reference spelling (train_generator.py:549,565)
    return ScalarWriter(log_dir)
