"""In-LDS prologue (conv3x3_dma_pro_kernel) against the prologue-free ring kernel on the same tile, back to back:
what the transform and the activated side tensor cost per PreActResNet18 layer shape at B = 128.

    python tools/pro_bench.py
"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from combat_amd import ops  # noqa: E402
from combat_amd._lib import lib  # noqa: E402

bf16 = torch.bfloat16
g = lambda s: torch.Generator().manual_seed(s)


def timed(fn, reps=60):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


N = int(os.environ.get("PB_N", 128))
for (hw, c, k, tile) in [(32, 64, 64, 10), (16, 128, 128, 10), (16, 128, 128, 11), (8, 256, 256, 11), (4, 512, 512, 11)]:
    x = (torch.randn(N, c, hw, hw, generator=g(1)) * 1.5).permute(0, 2, 3, 1).contiguous().to(bf16).cuda()
    w = torch.randn(k, c, 3, 3, generator=g(2)) / math.sqrt(c * 9)
    pc = ops.PackedConv(w.cuda().contiguous(memory_format=torch.channels_last), 1, 1, c)
    pc.pack()
    scale, shift = (torch.rand(c, generator=g(3)) + 0.5).cuda(), (torch.randn(c, generator=g(4)) * 0.3).cuda()
    y, side, res = torch.empty(N, hw, hw, k, dtype=bf16, device="cuda"), torch.empty_like(x), torch.randn(N, hw, hw, k, device="cuda").to(bf16)
    aff = ops.Affine(scale, shift, 0, True, 0.0)
    rows = {}
    for name, kw in (("ring, no prologue", dict()), ("ring (auto tile: ws on layer1)", dict(auto=True)), ("prologue, no side tensor", dict(pro=aff)),
                     ("prologue + side tensor", dict(pro=aff, pro_act_dst=side))):
        auto = kw.pop("auto", False)
        a = ops.conv_args(x, y, pc, 0, add_post=res, stats_kind=5, tile=0 if auto else tile, **kw)
        r, _ = ops.conv_stats_layout(a)
        st = torch.zeros(r, 2, k, device="cuda")
        a.stats = st.data_ptr()
        rows[name] = timed(lambda: ops.conv_launch(a))
    act = torch.empty_like(x)
    s0 = torch.cuda.current_stream().cuda_stream
    rows["affine_act alone"] = timed(lambda: lib.combat_affine_act(x.data_ptr(), N * hw * hw, c, scale.data_ptr(), shift.data_ptr(), 0, 0.0, act.data_ptr(), s0))
    print("%3dx%-3d %3d->%-3d tile %d: " % (hw, hw, c, k, tile) + "  ".join("%s %.1f" % (n_, v) for n_, v in rows.items()), flush=True)
