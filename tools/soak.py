"""Leak check: N alternated steps, host RSS and device memory before / after, step-time drift."""
import os, sys, time, resource
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from combat_amd import step as step_mod
dev = torch.device("cuda", 0)
np.random.seed(0); torch.manual_seed(0)
st = step_mod.AlternatedStep(*bench.build_nets(dev), bench.Opt())
batches = bench.synth_batches(8, 128, 0, dev)
def snap():
    return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024, torch.cuda.memory_allocated() / 2**20, torch.cuda.memory_reserved() / 2**20
for i in range(50): st.run(*batches[i % 8])
torch.cuda.synchronize()
r0 = snap()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
ts = []
for blk in range(6):
    t0 = time.perf_counter()
    for i in range(N // 6): st.run(*batches[i % 8])
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / (N // 6) * 1e3)
    st.read_metrics(reset=True)
r1 = snap()
print("ms/step per block:", [round(t, 3) for t in ts])
print("host maxrss MB %.0f -> %.0f | device allocated MB %.0f -> %.0f | reserved %.0f -> %.0f" % (r0[0], r1[0], r0[1], r1[1], r0[2], r1[2]))
