"""In-kernel phase stamps of conv_wgrad3x3_dma_kernel (library built with -DCOMBAT_STAMPS)."""
import ctypes, math, os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from combat_amd import ops, _lib
bf16 = torch.bfloat16
lib = ctypes.CDLL(_lib.LIB_PATH)
lib.combat_debug_set_stamps_wgrad.argtypes = [ctypes.c_void_p]
ws = torch.empty(48 << 20, dtype=torch.uint8, device="cuda")
flush = torch.empty(768 << 20, dtype=torch.uint8, device="cuda")
for name, n, hw, c, k in [("L1", 128, 32, 64, 64), ("L2", 128, 16, 128, 128), ("L3", 128, 8, 256, 256), ("L4", 128, 4, 512, 512)]:
    x = torch.randn(n, hw, hw, c, device="cuda").to(bf16)
    w = (torch.randn(k, c, 3, 3, device="cuda") / math.sqrt(9 * c)).contiguous(memory_format=torch.channels_last)
    pc = ops.PackedConv(w, 1, 1, c)
    dy = torch.randn(n, hw, hw, k, device="cuda").to(bf16)
    dw = torch.zeros(k, 9, c, device="cuda")
    for cold in (False, True):
        for _ in range(2):
            ops.conv_wgrad(x, dy, pc, dw, workspace=ws)
        if cold:
            flush.fill_(1)
        torch.cuda.synchronize()
        st = torch.zeros(8 * 4096 * 8, dtype=torch.int64, device="cuda")
        lib.combat_debug_set_stamps_wgrad(ctypes.c_void_p(st.data_ptr()))
        ops.conv_wgrad(x, dy, pc, dw, workspace=ws)
        torch.cuda.synchronize()
        lib.combat_debug_set_stamps_wgrad(ctypes.c_void_p(0))
        s = st.cpu().numpy().reshape(-1, 8)
        nwg = int((s[:, 3] > 0).sum()) // 8
        per_wave = s[: nwg * 8].reshape(nwg, 8, 8).astype(np.float64)
        print("   per wave id (issue / compute / wait): " + "  ".join(
            "%d:%.0f/%.0f/%.0f" % (w, (per_wave[:, w, 0] / per_wave[:, w, 3]).mean(), (per_wave[:, w, 1] / per_wave[:, w, 3]).mean(),
                                   (per_wave[:, w, 2] / per_wave[:, w, 3]).mean()) for w in range(8)))
        s = per_wave[:, 0, :]
        npatch = s[:, 3].mean()
        print("%s %s: %d workgroups (wave 0), %.1f patches each | per patch: issue %.0f  compute %.0f  wait+barrier %.0f cyc | loop %.0f  epilogue %.0f cyc"
              % (name, "cold" if cold else "warm", len(s), npatch, (s[:, 0] / s[:, 3]).mean(), (s[:, 1] / s[:, 3]).mean(),
                 (s[:, 2] / s[:, 3]).mean(), s[:, 4].mean(), s[:, 5].mean()), flush=True)
