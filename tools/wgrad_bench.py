"""3x3 weight-gradient launches at N = 128 (workspace path, as the engines call it): back-to-back, isolated, cold."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from combat_amd import ops  # noqa: E402

bf16 = torch.bfloat16
FLUSH = torch.empty(768 << 20, dtype=torch.uint8, device="cuda")
ws = torch.empty(48 << 20, dtype=torch.uint8, device="cuda")
for name, n, hw, c, k, stride in [("L1 32x32 64->64", 128, 32, 64, 64, 1), ("L2 16x16 128->128", 128, 16, 128, 128, 1),
                                  ("L3 8x8 256->256", 128, 8, 256, 256, 1), ("L4 4x4 512->512", 128, 4, 512, 512, 1),
                                  ("L2s 64->128 s2", 128, 32, 64, 128, 2), ("L3s 128->256 s2", 128, 16, 128, 256, 2),
                                  ("U 16x16 128->64", 128, 16, 128, 64, 1), ("U 32x32 64->64", 128, 32, 64, 64, 1)]:
    x = torch.randn(n, hw, hw, c, device="cuda").to(bf16)
    w = (torch.randn(k, c, 3, 3, device="cuda") / math.sqrt(9 * c)).contiguous(memory_format=torch.channels_last)
    pc = ops.PackedConv(w, stride, 1, c)
    p = hw // stride
    dy = torch.randn(n, p, p, k, device="cuda").to(bf16)
    dw = torch.zeros(k, 9, c, device="cuda")
    run = lambda: ops.conv_wgrad(x, dy, pc, dw, workspace=ws)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    b2b = e0.elapsed_time(e1) / 20 * 1e3
    res = []
    for cold in (False, True):
        v = []
        for _ in range(5):
            if cold:
                FLUSH.fill_(1)
            torch.cuda.synchronize()
            e0.record()
            run()
            e1.record()
            torch.cuda.synchronize()
            v.append(e0.elapsed_time(e1) * 1e3)
        res.append(sorted(v)[2])
    gf = 2.0 * n * p * p * c * k * 9 / 1e9
    print("%-20s b2b %.1f  isolated %.1f  cold %.1f us   (%.0f TF cold)" % (name, b2b, res[0], res[1], gf / res[1] * 1e3), flush=True)
