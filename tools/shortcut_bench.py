"""A residual block's stride-2 input gradients at 128 images: the 3x3 launch + the 1x1 shortcut launch (through add_pre)
against ONE launch with the shortcut as second reduction source (combat_conv_args.src2); mask + statistics epilogue of the
train backward.  back-to-back / isolated / cold-cache microseconds (tools/conv_bench.py)."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from combat_amd import ops
import conv_bench as cb
bf16 = torch.bfloat16
n = int(os.environ.get("CB_N", "128"))
for hw, c, k in ((32, 64, 128), (16, 128, 256), (8, 256, 512)):
    mk = lambda r, pad: ops.PackedConv((torch.randn(k, c, r, r, device="cuda") / math.sqrt(r * r * c)).contiguous(memory_format=torch.channels_last), 2, pad, c)
    pc3, pc1 = mk(3, 1), mk(1, 0)
    pc3.pack(); pc1.pack()
    p = hw // 2
    dy3, dy1 = (torch.randn(n, p, p, k, device="cuda").to(bf16) for _ in range(2))
    xpre = torch.randn(n, hw, hw, c, device="cuda").to(bf16)
    dx, tsc = (torch.empty(n, hw, hw, c, dtype=bf16, device="cuda") for _ in range(2))
    mask = ops.Affine(torch.rand(c, device="cuda") + 0.5, torch.randn(c, device="cuda") * 0.3, 0, True, 0.0)
    mean, rstd = torch.randn(c, device="cuda") * 0.1, torch.rand(c, device="cuda") + 0.5
    ws = torch.empty(32 << 20, dtype=torch.uint8, device="cuda")
    def args(fused, kind=2 | 4):
        a = ops.conv_args(dy3, dx, pc3, 1, add_pre=None if fused else tsc, mask_x=xpre, mask=mask, stats_kind=kind,
                          xh_mean=mean if kind else None, xh_rstd=rstd if kind else None, workspace=ws,
                          shortcut=(dy1, pc1) if fused else None)
        if kind:
            rows, _ = ops.conv_stats_layout(a)
            a._stats = torch.zeros(rows, 2, c, device="cuda")
            a.stats = a._stats.data_ptr()
        return a
    t3 = cb.timeit(args(False))
    t1 = cb.timeit(ops.conv_args(dy1, tsc, pc1, 1))
    tf = cb.timeit(args(True))
    print("   fused, per-wave rows: %.1f / %.1f / %.1f   fused, mask only: %.1f / %.1f / %.1f" % (cb.timeit(args(True, 2)) + cb.timeit(args(True, 0))), flush=True)
    f = lambda t: "%.1f / %.1f / %.1f" % t
    print("%dx%d %d->%d  3x3: %s   1x1: %s   fused: %s us" % (hw, hw, c, k, f(t3), f(t1), f(tf)), flush=True)
