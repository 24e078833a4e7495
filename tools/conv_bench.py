"""Per-shape convolution timings at the benchmarked batch (N = 128): every PreActResNet18 / UNet 3x3 shape, forward
(residual + statistics or activated-output epilogue) and input gradient (mask epilogue), for the automatic tile and
explicit tile ids.  Three numbers per case: back-to-back average (launches overlap their ramps), an isolated launch bracketed by HIP
events after a device sync (what a dependent chain pays), and the same after evicting L2 / Infinity Cache (what
the step's launches find: weights and operands come from HBM).

    python tools/conv_bench.py [shape-substring] [--tiles 0,10,11,12]
"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from combat_amd import ops  # noqa: E402
from combat_amd._lib import lib  # noqa: E402
import ctypes  # noqa: E402

bf16 = torch.bfloat16
SHAPES = [  # name, n, hw, c, k, stride
    ("L1 32x32 64->64", 128, 32, 64, 64, 1), ("L2 16x16 128->128", 128, 16, 128, 128, 1), ("L3 8x8 256->256", 128, 8, 256, 256, 1),
    ("L4 4x4 512->512", 128, 4, 512, 512, 1), ("L2s 32x32 64->128 s2", 128, 32, 64, 128, 2), ("L3s 16x16 128->256 s2", 128, 16, 128, 256, 2),
    ("L4s 8x8 256->512 s2", 128, 8, 256, 512, 2), ("U 16x16 64->64", 128, 16, 64, 64, 1), ("U 8x8 128->128", 128, 8, 128, 128, 1),
    ("U 4x4 256->256", 128, 4, 256, 256, 1), ("U 2x2 512->512", 128, 2, 512, 512, 1), ("U 4x4 512->512", 128, 4, 512, 512, 1),
    ("U 4x4 512->256", 128, 4, 512, 256, 1), ("U 8x8 256->256", 128, 8, 256, 256, 1), ("U 8x8 256->128", 128, 8, 256, 128, 1),
    ("U 16x16 128->128", 128, 16, 128, 128, 1), ("U 16x16 128->64", 128, 16, 128, 64, 1), ("U 32x32 64->64", 128, 32, 64, 64, 1),
]


FLUSH = None


def flush_caches():
    """Evict L2 and the 256 MiB Infinity Cache: the state a convolution finds in the step, where ~400 MB of other
    tensors pass between two uses of a weight tensor."""
    global FLUSH
    if FLUSH is None:
        FLUSH = torch.empty(768 << 20, dtype=torch.uint8, device="cuda")
    FLUSH.fill_(1)


def timeit(a, reps=30):
    for _ in range(3):
        ops.conv_launch(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.conv_launch(a)
    e1.record()
    torch.cuda.synchronize()
    b2b = e0.elapsed_time(e1) / reps * 1e3
    iso = []
    for _ in range(8):
        torch.cuda.synchronize()
        e0.record()
        ops.conv_launch(a)
        e1.record()
        torch.cuda.synchronize()
        iso.append(e0.elapsed_time(e1) * 1e3)
    cold = []
    for _ in range(5):
        flush_caches()
        torch.cuda.synchronize()
        e0.record()
        ops.conv_launch(a)
        e1.record()
        torch.cuda.synchronize()
        cold.append(e0.elapsed_time(e1) * 1e3)
    return b2b, sorted(iso)[len(iso) // 2], sorted(cold)[len(cold) // 2]


def case(n, hw, c, k, stride, tile, mode, ws):
    x = torch.randn(n, hw, hw, c, device="cuda").to(bf16)
    w = (torch.randn(k, c, 3, 3, device="cuda") / math.sqrt(9 * c)).contiguous(memory_format=torch.channels_last)
    pc = ops.PackedConv(w, stride, 1, c)
    pc.pack()
    p = hw // stride
    try:
        if mode == "fwd_train":      # residual + statistics (train-mode forward)
            y = torch.empty(n, p, p, k, dtype=bf16, device="cuda")
            r = torch.randn(n, p, p, k, device="cuda").to(bf16)
            a = ops.conv_args(x, y, pc, 0, add_post=r, stats_kind=1, tile=tile, workspace=ws)
            rows, _ = ops.conv_stats_layout(a)
            st = torch.zeros(rows, 2, k, device="cuda")
            a.stats = st.data_ptr()
            a._keepalive += (st, r, y)
        elif mode == "fwd_eval":     # residual + raw + activated output (eval-mode forward)
            y = torch.empty(n, p, p, k, dtype=bf16, device="cuda")
            act = torch.empty_like(y)
            r = torch.randn(n, p, p, k, device="cuda").to(bf16)
            aff = ops.Affine(torch.rand(k, device="cuda") + 0.5, torch.randn(k, device="cuda"), 0, True, 0.0)
            a = ops.conv_args(x, y, pc, 0, add_post=r, act_dst=act, act=aff, tile=tile, workspace=ws)
        else:                        # input gradient with an activated mask (eval-mode backward)
            dy = torch.randn(n, p, p, k, device="cuda").to(bf16)
            dx = torch.empty(n, hw, hw, c, dtype=bf16, device="cuda")
            xact = torch.relu(torch.randn(n, hw, hw, c, device="cuda")).to(bf16)
            aff = ops.Affine(torch.rand(c, device="cuda") + 0.5, torch.zeros(c, device="cuda"), 0, True, 0.0)
            a = ops.conv_args(dy, dx, pc, 1, mask_x=xact, mask=aff, mask_mul_scale=True, mask_activated=True, tile=tile,
                              workspace=ws)
        picked = lib.combat_conv_pick_tile(ctypes.byref(a))
        if tile and picked != tile:
            return None
        b2b, iso, cold = timeit(a)
        return picked, b2b, iso, cold
    except Exception as e:   # tile not applicable to this shape
        return None


def main():
    sel = [a for a in sys.argv[1:] if not a.startswith("--")]
    tiles = [0, 10, 11, 12, 13, 14, 16]
    for i, a in enumerate(sys.argv):
        if a == "--tiles":
            tiles = [int(v) for v in sys.argv[i + 1].split(",")]
    ws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    n_over = int(os.environ.get("CB_N", 0))      # batch override (what a half-batch chain would pay per launch)
    for name, n, hw, c, k, stride in SHAPES:
        if sel and not any(s in name for s in sel):
            continue
        n = n_over or n
        gf = 2.0 * n * (hw // stride) ** 2 * c * k * 9 / 1e9
        for mode in ("fwd_train", "fwd_eval", "dgrad"):
            cells = []
            for t in tiles:
                r = case(n, hw, c, k, stride, t, mode, ws)
                if r is not None:
                    cells.append("t%d[%d]: %.1f / %.1f / cold %.1f us (%.0f TF)" % (t, r[0], r[1], r[2], r[3], gf / r[2] * 1e3))
            print("%-22s %-9s %s" % (name, mode, "  ".join(cells)), flush=True)


if __name__ == "__main__":
    main()
