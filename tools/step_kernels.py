"""Kernel launches of ONE steady-state alternated step by name (torch.profiler), to see what is not a plan call."""
import os, sys, collections
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from combat_amd import step as step_mod
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda", 0)
np.random.seed(0); torch.manual_seed(0)
st = step_mod.AlternatedStep(*bench.build_nets(dev), bench.Opt())
batches = bench.synth_batches(8, 128, 0, dev)
for i in range(10): st.run(*batches[i % 8])
torch.cuda.synchronize()
S = 4
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for i in range(S): st.run(*batches[i % 8])
    torch.cuda.synchronize()
cnt = collections.Counter(); dur = collections.Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA:
        cnt[e.name[:70]] += 1; dur[e.name[:70]] += e.device_time if hasattr(e, "device_time") else e.cuda_time
tot = 0
for k, c in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print("%6.1f/step %8.1f us/step  %s" % (c / S, dur[k] / S, k))
    tot += c / S
print("launches/step", tot)
