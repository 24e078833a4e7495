import os, sys, math, torch
sys.path.insert(0, os.getcwd())
from combat_amd import ops
bf16 = torch.bfloat16
def run(n, hw, k, stride, split, reps=30):
    w = (torch.randn(k, 3, 3, 3, device='cuda') / 5).contiguous(memory_format=torch.channels_last)
    pc = ops.PackedConv(w, stride, 1, 8, dup_hilo=True); pc.pack()
    o = hw // stride
    x = torch.randn(n, hw, hw, 8, device='cuda').to(bf16)
    dy = torch.randn(n, o, o, k, device='cuda').to(bf16)
    dw = torch.zeros(k, 9, 3, device='cuda')
    ws = torch.empty(48 << 20, dtype=torch.uint8, device='cuda')
    for _ in range(3): ops.conv_wgrad(x, dy, pc, dw, split=split, workspace=ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): ops.conv_wgrad(x, dy, pc, dw, split=split, workspace=ws)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for name, a in [('stem N128', (128, 32, 64, 1)), ('conv0_0 s2', (128, 32, 64, 2)), ('stem N256', (256, 32, 64, 1))]:
    print(name, 'generic %.1f us   c8 %.1f us   c8 split 128: %.1f  512: %.1f' % (run(*a, -1), run(*a, 0), run(*a, 128), run(*a, 512)), flush=True)
