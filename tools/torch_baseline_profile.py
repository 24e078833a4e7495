"""Where the stock PyTorch-ROCm baseline leg of bench.py spends its step (torch profiler, top device ops)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from oracle import combat_oracle as O  # noqa: E402

dev = torch.device("cuda", 0)
netc, clean, netg, netf = bench._oracle_nets(dev)
bufs_c, bufs_g = [None] * len(O.trainable_names(netc)), [None] * len(O.trainable_names(netg))
g = torch.Generator().manual_seed(1234)
x = (((torch.randint(0, 256, (128, 3, 32, 32), generator=g, dtype=torch.uint8).float() / 255) - 0.5) / 0.5).to(dev)
t = torch.randint(0, 10, (128,), generator=g).to(dev)
rng = np.random.default_rng(0)


def one():
    rnd = O.StepRandomness(6, 0.5, 0.6, [bench._aug_draw(rng, 128) for _ in range(5)])
    O.alternated_step(netc, netg, clean, netf, bufs_c, bufs_g, x, t, rnd, O.StepConfig(), as_written=True,
                      aug_fn=O.post_tensor_transform_batched)


for _ in range(3):
    one()
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        one()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=70))
