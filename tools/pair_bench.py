"""A residual block's stride-2 first convolution (statistics epilogue) and its 1x1 shortcut over the same activated input at
128 images: ONE combat_conv_gemm_pair launch against two combat_conv_gemm launches, back to back."""
import ctypes, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from combat_amd import ops
from combat_amd._lib import lib
bf16 = torch.bfloat16
n = int(os.environ.get("CB_N", "128"))
def st(): return torch.cuda.current_stream().cuda_stream
def timeit(f, reps=40):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for hw, c, k in ((32, 64, 128), (16, 128, 256), (8, 256, 512)):
    mk = lambda r, pad: ops.PackedConv((torch.randn(k, c, r, r, device="cuda") / math.sqrt(r * r * c)).contiguous(memory_format=torch.channels_last), 2, pad, c)
    pc3, pc1 = mk(3, 1), mk(1, 0)
    pc3.pack(); pc1.pack()
    p = hw // 2
    x = torch.randn(n, hw, hw, c, device="cuda").to(bf16)
    y3, y1 = (torch.empty(n, p, p, k, dtype=bf16, device="cuda") for _ in range(2))
    ws = torch.empty(32 << 20, dtype=torch.uint8, device="cuda")
    a3 = ops.conv_args(x, y3, pc3, 0, stats_kind=1 | 4, workspace=ws)
    rows, _ = ops.conv_stats_layout(a3)
    stats = torch.zeros(rows, 2, k, device="cuda")
    a3.stats = stats.data_ptr()
    a1 = ops.conv_args(x, y1, pc1, 0, workspace=ws)
    t_pair = timeit(lambda: ops.check(lib.combat_conv_gemm_pair(ctypes.byref(a1), ctypes.byref(a3), st()), "pair"))
    t_two = timeit(lambda: (ops.check(lib.combat_conv_gemm(ctypes.byref(a1), st()), "a1"), ops.check(lib.combat_conv_gemm(ctypes.byref(a3), st()), "a3")))
    t3 = timeit(lambda: ops.check(lib.combat_conv_gemm(ctypes.byref(a3), st()), "a3"))
    t1 = timeit(lambda: ops.check(lib.combat_conv_gemm(ctypes.byref(a1), st()), "a1"))
    print("%dx%d %d->%d: pair %.1f us   two launches %.1f us   (3x3 alone %.1f, 1x1 alone %.1f)" % (hw, hw, c, k, t_pair, t_two, t3, t1), flush=True)
