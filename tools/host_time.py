"""Host-side cost of enqueueing one alternated step (no device wait): if this approaches the step time, the
launch path -- one ctypes call per kernel from Python -- is the bound, not the device."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from combat_amd import step as step_mod  # noqa: E402

dev = torch.device("cuda", 0)
np.random.seed(0)
netc, netg, clean, netf = bench.build_nets(dev)
st = step_mod.AlternatedStep(netc, netg, clean, netf, bench.Opt())
batches = bench.synth_batches(8, 128, 0, dev)
for i in range(8):
    st.run(*batches[i % 8])
torch.cuda.synchronize()
host = []
for rep in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(2):
        st.run(*batches[i % 8])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append(((t1 - t0) / 2 * 1e3, (t2 - t0) / 2 * 1e3))
print("host enqueue ms/step, total ms/step (2 steps from an idle device):", [(round(a, 3), round(b, 3)) for a, b in host])
import cProfile, pstats
pr = cProfile.Profile()
torch.cuda.synchronize()
pr.enable()
for i in range(3):
    st.run(*batches[i % 8])
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
