"""Weight-gradient slab reductions at B = 128, back to back: the kernel alone (slabs left behind), + its stand-alone
reduction launch (deterministic / round-3 atomics form: COMBAT_WGRAD_ATOMIC_REDUCE=1 in the environment), and the kernel
carrying the previous launch's reduction (combat_wgrad_args.reduce_first).

    python tools/wgrad_chain_bench.py
"""
import ctypes
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from combat_amd import ops  # noqa: E402
from combat_amd._lib import WgradArgs, lib  # noqa: E402

bf16 = torch.bfloat16
g = lambda s: torch.Generator().manual_seed(s)
st = torch.cuda.current_stream().cuda_stream


def timed(fn, reps=40):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


N = 128
for (hw, c, k) in [(32, 64, 64), (16, 128, 128), (8, 256, 256), (4, 512, 512)]:
    x = torch.randn(N, hw, hw, c, generator=g(1)).to(bf16).cuda()
    dy = torch.randn(N, hw, hw, k, generator=g(2)).to(bf16).cuda()
    w = torch.randn(k, c, 3, 3, generator=g(3)) / math.sqrt(c * 9)
    pc = ops.PackedConv(w.cuda().contiguous(memory_format=torch.channels_last), 1, 1, c)
    dw = torch.zeros(k, 9, c, device="cuda")
    regions = [torch.empty(24 << 20, dtype=torch.uint8, device="cuda") for _ in range(2)]

    def mk(region, defer, first=None):
        a = WgradArgs()
        a.N, a.H, a.W, a.C = x.shape
        _, a.P, a.Q, a.K = dy.shape
        a.R = a.S = 3
        a.stride, a.pad = 1, 1
        a.src, a.dy, a.dw, a.k_real, a.c_real = x.data_ptr(), dy.data_ptr(), dw.data_ptr(), k, c
        a.workspace, a.workspace_bytes, a.defer_reduce = region.data_ptr(), region.numel(), defer
        if first is not None:
            a.reduce_first = ctypes.addressof(first)
        return a

    a_def = mk(regions[0], 1)
    a_own = mk(regions[0], 0)
    a_behind = mk(regions[1], 1, a_def)
    t_kernel = timed(lambda: lib.combat_conv_wgrad(ctypes.byref(a_def), st))
    t_own = timed(lambda: lib.combat_conv_wgrad(ctypes.byref(a_own), st))
    t_red = timed(lambda: lib.combat_conv_wgrad_reduce(ctypes.byref(a_def), st))
    t_behind = timed(lambda: lib.combat_conv_wgrad(ctypes.byref(a_behind), st))
    print("%2dx%-2d %3d->%-3d  kernel alone %.1f  kernel + reduction launch %.1f (reduction alone %.1f)  kernel carrying a reduction %.1f us" % (
        hw, hw, c, k, t_kernel, t_own, t_red, t_behind), flush=True)
