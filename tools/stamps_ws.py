"""Per-wave cycle stamps of the weight-stationary conv3x3 kernel (tile 17).  Build the profiling library first:
    python tools/build_stamps_lib.py        (-> combat_amd/libcombat_hip_stamps.so, conv3x3_dma.hip with -DCOMBAT_STAMPS)
    COMBAT_HIP_LIB=combat_amd/libcombat_hip_stamps.so python tools/stamps_ws.py"""
import ctypes, math, os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from combat_amd import ops, _lib
bf16 = torch.bfloat16
lib = ctypes.CDLL(_lib.LIB_PATH)
lib.combat_debug_set_stamps_dma.argtypes = [ctypes.c_void_p]
n, hw, c, k = int(os.environ.get("WS_N", 128)), 32, 64, 64
x = torch.randn(n, hw, hw, c, device="cuda").to(bf16)
w = (torch.randn(k, c, 3, 3, device="cuda") / math.sqrt(9 * c)).contiguous(memory_format=torch.channels_last)
pc = ops.PackedConv(w, 1, 1, c); pc.pack()
y = torch.empty(n, hw, hw, k, dtype=bf16, device="cuda")
act = torch.empty_like(y)
r = torch.randn(n, hw, hw, k, device="cuda").to(bf16)
aff = ops.Affine(torch.rand(k, device="cuda") + .5, torch.randn(k, device="cuda"), 0, True, 0.0)
cases = {"plain": dict(), "eval fwd (act + residual)": dict(add_post=r, act_dst=act, act=aff), "train fwd (stats + residual)": dict(add_post=r, stats_kind=1)}
for name, kw in cases.items():
    a = ops.conv_args(x, y, pc, 0, tile=17, **kw)
    if a.stats_kind:
        rows, _ = ops.conv_stats_layout(a)
        st = torch.zeros(rows, 2, k, device="cuda"); a.stats = st.data_ptr()
    for _ in range(3): ops.conv_launch(a)
    torch.cuda.synchronize()
    stamps = torch.zeros(256 * 8 * 40, dtype=torch.int64, device="cuda")
    lib.combat_debug_set_stamps_dma(ctypes.c_void_p(stamps.data_ptr()))
    ops.conv_launch(a)
    torch.cuda.synchronize()
    lib.combat_debug_set_stamps_dma(ctypes.c_void_p(0))
    s = stamps.cpu().numpy().reshape(256, 8, 40).astype(np.float64)
    s = s - s[:, :1, :1]           # relative to wave 0's entry of each workgroup
    print(name)
    lab = {0: "entry", 1: "prologue DMA issued", 2: "prologue barrier passed", 3: "(B: extra barrier)"}
    for kk in range(2):
        lab.update({4 + 6 * kk: "t%d MFMAs done" % kk, 5 + 6 * kk: "t%d barrier" % kk, 6 + 6 * kk: "t%d DMA issued" % kk,
                    7 + 6 * kk: "t%d epilogue math done" % kk, 8 + 6 * kk: "t%d DMA landed" % kk, 9 + 6 * kk: "t%d stores + fetch issued" % kk})
    for g, gname in ((0, "group A (waves 0-3)"), (1, "group B (waves 4-7)")):
        m = s[:, 4 * g:4 * g + 4, :].mean((0, 1))
        print("  ", gname)
        prev = 0
        for i in range(16):
            if m[i] > 0 or i == 0:
                print("     %-28s %8.0f  (+%6.0f)" % (lab.get(i, str(i)), m[i], m[i] - prev))
                prev = m[i]
