import os, sys, math, ctypes, torch
sys.path.insert(0, os.getcwd())
import numpy as np
from combat_amd import ops, _lib
bf16 = torch.bfloat16
lib = ctypes.CDLL(_lib.LIB_PATH)
lib.combat_debug_set_stamps_gather.argtypes = [ctypes.c_void_p]
def run(n, hw, c, k, stride, R, mode, tile=0):
    o = hw // stride
    w = (torch.randn(k, c, R, R, device='cuda') / math.sqrt(R*R*c)).contiguous(memory_format=torch.channels_last)
    pc = ops.PackedConv(w, stride, R // 2, c); pc.pack()
    if mode == 0:
        x = torch.randn(n, hw, hw, c, device='cuda').to(bf16); y = torch.empty(n, o, o, k, dtype=bf16, device='cuda')
    else:
        x = torch.randn(n, o, o, k, device='cuda').to(bf16); y = torch.empty(n, hw, hw, c, dtype=bf16, device='cuda')
    ws = torch.empty(32 << 20, dtype=torch.uint8, device='cuda')
    a = ops.conv_args(x, y, pc, mode, tile=tile, workspace=ws)
    for _ in range(3): ops.conv_launch(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.conv_launch(a)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    stamps = torch.zeros(65536 * 16, dtype=torch.int64, device='cuda')
    lib.combat_debug_set_stamps_gather(ctypes.c_void_p(stamps.data_ptr()))
    ops.conv_launch(a)
    torch.cuda.synchronize()
    lib.combat_debug_set_stamps_gather(ctypes.c_void_p(0))
    s = stamps.cpu().numpy().reshape(-1, 16).astype(np.float64)
    s = s[s[:, 4] > 0]
    st = s[:, 4]
    print('N%d hw%d C%d K%d s%d R%d mode%d: %.1f us; blocks %d steps/wg %.1f | per step: issue %.0f  lds-read %.0f  mfma-issue %.0f  wait+barrier %.0f  = %.0f cyc' % (
        n, hw, c, k, stride, R, mode, us, len(s), st.mean(), (s[:, 0] / st).mean(), (s[:, 1] / st).mean(), (s[:, 2] / st).mean(), (s[:, 3] / st).mean(), (s[:, :4].sum(1) / st).mean()), flush=True)
    wl = s[:, 7] / 100.0
    span = (s[:, 10].max() - s[:, 8].min()) / 100.0 if s[:, 10].any() else float('nan')
    print('     pre-loop %.0f cyc  loop %.0f cyc  epilogue %.0f cyc | start->loop-end %.2f us wall => %.2f GHz | first start -> last end %.1f us' % (s[:, 5].mean(), s[:, 6].mean(), s[:, 9].mean(), wl.mean(), (s[:, 5] + s[:, 6]).mean() / (wl.mean() * 1e3), span), flush=True)
    print('     pre-loop split: pixel rows %.0f | weights+lambdas+first issue %.0f | epilogue rows %.0f | first wait %.0f' % (s[:, 11].mean(), s[:, 12].mean(), s[:, 13].mean(), s[:, 14].mean()), flush=True)
run(128, 8, 256, 512, 2, 3, 0)      # b6.c1
run(128, 16, 128, 256, 2, 3, 0)     # b4.c1
run(128, 32, 64, 128, 2, 3, 0)      # b2.c1
run(128, 8, 256, 512, 2, 3, 1)      # b6.c1 dgrad
run(128, 16, 128, 256, 2, 3, 1)     # b4.c1 dgrad
run(128, 4, 512, 512, 1, 3, 0)      # unet 4x4? (goes to dma kernel probably)
run(128, 2, 512, 512, 1, 3, 0)      # unet 2x2
run(128, 32, 64, 128, 2, 1, 0)      # shortcut 1x1
