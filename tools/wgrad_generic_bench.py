import os, sys, math, torch
sys.path.insert(0, os.getcwd())
from combat_amd import ops
bf16 = torch.bfloat16
def run(n, hw, c, k, r, stride, split=0, reps=30):
    w = (torch.randn(k, c, r, r, device='cuda') / 5).contiguous(memory_format=torch.channels_last)
    pc = ops.PackedConv(w, stride, r // 2, c); pc.pack()
    o = hw // stride
    x = torch.randn(n, hw, hw, c, device='cuda').to(bf16)
    dy = torch.randn(n, o, o, pc.Kc, device='cuda').to(bf16)
    dw = torch.zeros(k, r * r, c, device='cuda')
    ws = torch.empty(48 << 20, dtype=torch.uint8, device='cuda')
    for _ in range(3): ops.conv_wgrad(x, dy, pc, dw, split=split, workspace=ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): ops.conv_wgrad(x, dy, pc, dw, split=split, workspace=ws)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for name, a in [('b2.c1 3x3s2 64->128 hw32', (128, 32, 64, 128, 3, 2)), ('b4.c1 128->256 hw16', (128, 16, 128, 256, 3, 2)), ('b6.c1 256->512 hw8', (128, 8, 256, 512, 3, 2)),
                ('b2.sc 1x1s2 64->128', (128, 32, 64, 128, 1, 2)), ('b4.sc 128->256', (128, 16, 128, 256, 1, 2)), ('b6.sc 256->512', (128, 8, 256, 512, 1, 2)),
                ('unet conv1_0 64->128 hw16', (128, 16, 64, 128, 3, 2)), ('conv2_0 128->256 hw8', (128, 8, 128, 256, 3, 2)), ('conv3_0 256->512 hw4', (128, 4, 256, 512, 3, 2)),
                ('upconv0_0 64->3 hw32', (128, 32, 64, 3, 3, 1))]:
    gf = 2.0 * a[0] * (a[1] // a[5]) ** 2 * a[2] * a[3] * a[4] * a[4] / 1e9
    print('%-28s %6.1f us  (%.2f GFLOP -> %.0f TF/s)   split 16: %.1f  64: %.1f  256: %.1f' % (name, run(*a), gf, gf / run(*a) * 1e3, run(*a, 16), run(*a, 64), run(*a, 256)), flush=True)
