"""Premise check for a split-K-by-2 pair of 64-channel-tile workgroups on the skinny layers: a launch with half the input
channels and 64-channel tiles has the workgroup lifetime such a half would have."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from combat_amd import ops
bf16 = torch.bfloat16
g = lambda s: torch.Generator().manual_seed(s)

def timed(fn, reps=60):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

N = 128
for (hw, c, k, tile) in [(4, 512, 512, 11), (4, 512, 512, 10), (4, 256, 512, 10), (4, 256, 512, 11), (8, 256, 256, 11), (8, 256, 256, 10), (8, 128, 256, 10), (16, 128, 128, 10), (16, 64, 128, 10)]:
    x = torch.randn(N, hw, hw, c, generator=g(1)).to(bf16).cuda()
    w = torch.randn(k, c, 3, 3, generator=g(2)) / math.sqrt(c * 9)
    pc = ops.PackedConv(w.cuda().contiguous(memory_format=torch.channels_last), 1, 1, c)
    pc.pack()
    y = torch.empty(N, hw, hw, k, dtype=bf16, device="cuda")
    a = ops.conv_args(x, y, pc, 0, tile=tile)
    t = timed(lambda: ops.conv_launch(a))
    tiles_m = N * hw * hw // 128
    print("%2dx%-2d %3d->%-3d tile %d: %5.1f us   workgroups %d" % (hw, hw, c, k, tile, t, tiles_m * (k // (64 if tile == 10 else 32))), flush=True)
