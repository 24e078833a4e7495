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
        if self.tb is not None:
            self.tb.add_image(tag, img, global_step=global_step)


def SummaryWriter(log_dir: str) -> ScalarWriter:  # reference spelling (train_generator.py:549,565)
    return ScalarWriter(log_dir)
