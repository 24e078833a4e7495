"""What would running an eval-mode convolution chain as two half-batch chains on two queues buy?  PreActResNet18's 3x3
convolutions (forward with residual + activated output, or input gradient with mask) as ONE chain at N = 128 on one stream
against TWO chains at N = 64 on two streams.  Independent tensors per layer (no real data flow): it times launches only."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from combat_amd import ops  # noqa: E402
import conv_bench as cb  # noqa: E402

LAYERS = [(32, 64, 64, 1)] * 4 + [(32, 64, 128, 2)] + [(16, 128, 128, 1)] * 3 + [(16, 128, 256, 2)] + [(8, 256, 256, 1)] * 3 + \
         [(8, 256, 512, 2)] + [(4, 512, 512, 1)] * 3


def chain(n, mode, ws):
    out = []
    for hw, c, k, s in LAYERS:
        x = torch.randn(n, hw, hw, c, device="cuda").to(cb.bf16)
        w = (torch.randn(k, c, 3, 3, device="cuda") / math.sqrt(9 * c)).contiguous(memory_format=torch.channels_last)
        pc = ops.PackedConv(w, s, 1, c)
        pc.pack()
        p = hw // s
        if mode == "fwd":
            y = torch.empty(n, p, p, k, dtype=cb.bf16, device="cuda")
            act = torch.empty_like(y)
            aff = ops.Affine(torch.rand(k, device="cuda") + 0.5, torch.randn(k, device="cuda"), 0, True, 0.0)
            a = ops.conv_args(x, y, pc, 0, act_dst=act, act=aff, workspace=ws)
        else:
            dy = torch.randn(n, p, p, k, device="cuda").to(cb.bf16)
            dx = torch.empty(n, hw, hw, c, dtype=cb.bf16, device="cuda")
            xact = torch.relu(torch.randn(n, hw, hw, c, device="cuda")).to(cb.bf16)
            aff = ops.Affine(torch.rand(c, device="cuda") + 0.5, torch.zeros(c, device="cuda"), 0, True, 0.0)
            a = ops.conv_args(dy, dx, pc, 1, mask_x=xact, mask=aff, mask_mul_scale=True, mask_activated=True, workspace=ws)
        out.append(a)
    return out


def run(chains, streams, reps=20):
    def once():
        ev = torch.cuda.Event()
        ev.record()
        for ch, st in zip(chains, streams):
            st.wait_event(ev)
            with torch.cuda.stream(st):
                for a in ch:
                    ops.conv_launch(a)
        for st in streams:
            torch.cuda.current_stream().wait_stream(st)
    for _ in range(3):
        once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        once()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    s = [torch.cuda.Stream() for _ in range(4)]
    wss = [torch.empty(64 << 20, dtype=torch.uint8, device="cuda") for _ in range(4)]
    for mode in ("fwd", "dgrad"):
        full = run([chain(128, mode, wss[0])], s[:1])
        two = run([chain(64, mode, wss[i]) for i in range(2)], s[:2])
        four = run([chain(32, mode, wss[i]) for i in range(4)], s[:4])
        half_alone = run([chain(64, mode, wss[0])], s[:1])
        print("%-5s one chain N=128: %.0f us | two chains N=64 on two queues: %.0f us | four chains N=32: %.0f us | one chain N=64 alone: %.0f us"
              % (mode, full, two, four, half_alone), flush=True)


if __name__ == "__main__":
    main()
