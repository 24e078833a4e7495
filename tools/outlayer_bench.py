"""The generator's output layer (upconv0_0: 64 -> 3 (stored as 8), 32x32, InstanceNorm + LeakyReLU prologue, tanh) and a
stem's input gradient (64-channel dY -> 8): time per tile id (18 = conv_k8, 4 = the generic 16-wide tile), N = 128."""
import math, os, sys, ctypes
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from combat_amd import ops
from combat_amd._lib import lib
import conv_bench as cb
bf16 = torch.bfloat16
n, hw, c = 128, 32, 64
x = torch.randn(n, hw, hw, c, device="cuda").to(bf16)
w = (torch.randn(3, c, 3, 3, device="cuda") / math.sqrt(9 * c)).contiguous(memory_format=torch.channels_last)
pc = ops.PackedConv(w, 1, 1, c)
pc.pack()
y = torch.empty(n, hw, hw, 8, dtype=bf16, device="cuda")
sc, sh = torch.rand(n, c, device="cuda") + 0.5, torch.randn(n, c, device="cuda")
pro = ops.Affine(sc, sh, c, True, 0.2)
bias = torch.zeros(8, device="cuda")
for tile in (0, 18, 4, 2):
    try:
        a = ops.conv_args(x, y, pc, 0, pro=pro, bias=bias, tanh_out=True, tile=tile)
        picked = lib.combat_conv_pick_tile(ctypes.byref(a))
        b2b, iso, cold = cb.timeit(a)
        print("fwd  tile %d -> %d: b2b %.1f / iso %.1f / cold %.1f us" % (tile, picked, b2b, iso, cold), flush=True)
    except Exception as e:
        print("fwd  tile %d: n/a (%s)" % (tile, str(e)[:60]))
ws = (torch.randn(c, 3, 3, 3, device="cuda") / math.sqrt(27)).contiguous(memory_format=torch.channels_last)
pcs = ops.PackedConv(ws, 1, 1, 8)
pcs.pack()
for tile in (0, 18, 4):
    a = ops.conv_args(x, y, pcs, 1, tile=tile)
    picked = lib.combat_conv_pick_tile(ctypes.byref(a))
    b2b, iso, cold = cb.timeit(a)
    print("stem dgrad tile %d -> %d: b2b %.1f / iso %.1f / cold %.1f us" % (tile, picked, b2b, iso, cold), flush=True)
