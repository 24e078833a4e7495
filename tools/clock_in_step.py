#!/usr/bin/env python3
"""Which engine clock does the chip hold inside the step?  tools/micro/clock_probe.hip's one-wave probe (20 us each)
runs back to back on a stream of its own -- first beside an idle GPU, then beside the alternated step -- and reports
shader-clock cycles per 100-MHz tick for every window, placed on the step's own time axis (a wall-clock stamp kernel
on the main stream marks each step's start).

    hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/micro/clock_probe.hip -o tools/micro/libclockprobe.so
    python tools/clock_in_step.py
"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from combat_amd import step as step_mod  # noqa: E402

probe = ctypes.CDLL(os.path.join(ROOT, "tools", "micro", "libclockprobe.so"))
probe.clock_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
probe.clock_probe_now.argtypes = [ctypes.c_void_p, ctypes.c_void_p]


def run_probes(stream, out, n, spin=2000):
    for i in range(n):
        assert probe.clock_probe_launch(out.data_ptr(), i, spin, stream.cuda_stream) == 0


def mhz(rows):
    return rows[:, 2] / rows[:, 1] * 100.0


def main():
    dev = torch.device("cuda", 0)
    opt = bench.Opt()
    np.random.seed(0)
    torch.manual_seed(100)
    st = step_mod.AlternatedStep(*bench.build_nets(dev), opt)
    batches = bench.synth_batches(8, opt.bs, 0, dev)
    for i in range(10):
        st.run(*batches[i % 8])
    torch.cuda.synchronize()
    ps = torch.cuda.Stream()
    nsteps, per_step = 20, 220
    out = torch.zeros(nsteps * per_step * 3 + 3, dtype=torch.int64, device=dev)
    marks = torch.zeros(nsteps + 1, dtype=torch.int64, device=dev)

    # ---- idle: only the probes
    run_probes(ps, out, 400)
    torch.cuda.synchronize()
    idle = out[:1200].view(-1, 3).cpu().numpy().astype(np.float64)
    print("idle GPU          : %7.1f MHz (min %.1f max %.1f, %d windows of %.1f us)" % (
        mhz(idle).mean(), mhz(idle).min(), mhz(idle).max(), len(idle), idle[:, 1].mean() / 100))

    # ---- beside the step
    out.zero_()
    main_s = torch.cuda.current_stream()
    for k in range(nsteps):
        probe.clock_probe_now(marks[k:].data_ptr(), main_s.cuda_stream)
        st.run(*batches[k % 8])
        run_probes(ps, out[k * per_step * 3:], per_step)
    probe.clock_probe_now(marks[nsteps:].data_ptr(), main_s.cuda_stream)
    torch.cuda.synchronize()
    rows = out[:nsteps * per_step * 3].view(-1, 3).cpu().numpy().astype(np.float64)
    mk = marks.cpu().numpy().astype(np.float64)
    step_us = np.diff(mk) / 100.0
    print("steps             : %.3f ms each (wall-clock stamps on the main queue)" % (step_us[5:].mean() / 1000))
    rows = rows[rows[:, 1] > 0]
    inside = (rows[:, 0] >= mk[5]) & (rows[:, 0] < mk[-1])
    r = rows[inside]
    print("beside the step   : %7.1f MHz (min %.1f max %.1f, %d windows)" % (mhz(r).mean(), mhz(r).min(), mhz(r).max(), len(r)))
    # position inside its step
    k = np.searchsorted(mk, r[:, 0], side="right") - 1
    pos = (r[:, 0] - mk[k]) / (mk[k + 1] - mk[k])
    for lo in np.arange(0.0, 1.0, 0.1):
        sel = (pos >= lo) & (pos < lo + 0.1)
        if sel.any():
            print("  step position %.1f-%.1f: %7.1f MHz (%d windows)" % (lo, lo + 0.1, mhz(r[sel]).mean(), sel.sum()))


if __name__ == "__main__":
    main()
