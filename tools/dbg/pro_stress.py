"""Repeat the in-LDS-prologue convolution many times against the chain's result: a race shows as occasional mismatches."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from combat_amd import ops
from combat_amd._lib import lib
bf16 = torch.bfloat16
g = lambda s: torch.Generator().manual_seed(s)
for (n, hw, c, k, tile, stats) in [(16, 32, 64, 64, 11, 5), (16, 32, 64, 64, 11, 0), (128, 32, 64, 64, 10, 5), (16, 16, 128, 128, 11, 5), (128, 16, 128, 128, 10, 5),
                                   (128, 8, 256, 256, 11, 5), (128, 4, 512, 512, 11, 5)]:
    x = (torch.randn(n, c, hw, hw, generator=g(900)) * 1.5 + 0.3).permute(0, 2, 3, 1).contiguous().to(bf16).cuda()
    w = torch.randn(k, c, 3, 3, generator=g(902)) / math.sqrt(c * 9)
    pc = ops.PackedConv(w.cuda().contiguous(memory_format=torch.channels_last), 1, 1, c)
    pc.pack()
    scale = (torch.rand(c, generator=g(1)) + 0.5).cuda()
    shift = (torch.randn(c, generator=g(2)) * 0.3).cuda()
    act = torch.empty_like(x)
    lib.combat_affine_act(x.data_ptr(), n * hw * hw, c, scale.data_ptr(), shift.data_ptr(), 0, 0.0, act.data_ptr(), torch.cuda.current_stream().cuda_stream)
    yref = torch.zeros(n, hw, hw, k, dtype=bf16, device="cuda")
    ar = ops.conv_args(act, yref, pc, 0, tile=tile, stats_kind=stats)
    if stats:
        rows, _ = ops.conv_stats_layout(ar)
        sref = torch.zeros(rows, 2, k, device="cuda")
        ar.stats = sref.data_ptr()
    ops.conv_launch(ar)
    torch.cuda.synchronize()
    ys = [torch.zeros_like(yref) for _ in range(4)]
    sides = [torch.zeros_like(x) for _ in range(4)]
    sts = [torch.zeros_like(sref) for _ in range(4)] if stats else None
    args = []
    for i in range(4):
        a = ops.conv_args(x, ys[i], pc, 0, pro=ops.Affine(scale, shift, 0, True, 0.0), pro_act_dst=sides[i], tile=tile, stats_kind=stats)
        if stats:
            a.stats = sts[i].data_ptr()
        args.append(a)
    bad_y = bad_s = bad_st = 0
    reps = 100
    for r in range(reps):
        for i in range(4):
            ops.conv_launch(args[i])          # back to back, no sync in between
        torch.cuda.synchronize()
        for i in range(4):
            bad_y += int(not torch.equal(ys[i], yref))
            bad_s += int(not torch.equal(sides[i], act))
            if stats:
                bad_st += int(not torch.equal(sts[i], sref))
    print((n, hw, c, k, tile, stats), "launches", 4 * reps, "bad y", bad_y, "bad side", bad_s, "bad stats", bad_st, flush=True)
