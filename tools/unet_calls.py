"""Each call of the generator's forward plan at the benchmarked batch, timed (a) in sequence (events around every call of
one replay, as tools/calls.py) and (b) alone, 20 back-to-back repeats of the same call -- tells a slow kernel from a call
that is slow only where it stands (cold operands, what ran before it)."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
from combat_amd import step as step_mod
device = torch.device("cuda", 0)
opt = bench.Opt()
st = step_mod.AlternatedStep(*bench.build_nets(device), opt)
batches = bench.synth_batches(2, opt.bs, 0, device)
for i in range(3): st.run(*batches[i % 2])
torch.cuda.synchronize()
plan = st.pl[sys.argv[1] if len(sys.argv) > 1 else "G_f"]
s = torch.cuda.current_stream(); stp = s.cuda_stream
seq = []
for cfunc, args, what in plan.calls:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s); cfunc(*args, stp); e1.record(s)
    seq.append((what, e0, e1))
torch.cuda.synchronize()
for (cfunc, args, what), (_, e0, e1) in zip(plan.calls, seq):
    t_seq = e0.elapsed_time(e1) * 1e3
    torch.cuda.synchronize()
    a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a0.record(s)
    for _ in range(20): cfunc(*args, stp)
    a1.record(s); torch.cuda.synchronize()
    print("%-28s in sequence %6.1f us   alone, back to back %6.1f us" % (what, t_seq, a0.elapsed_time(a1) * 1e3 / 20))
