import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
from combat_amd import step as step_mod
device = torch.device("cuda", 0)
opt = bench.Opt()
opt.bs = int(os.environ.get("QS_BS", opt.bs))
np.random.seed(0); torch.manual_seed(100)
st = step_mod.AlternatedStep(*bench.build_nets(device), opt)
batches = bench.synth_batches(8, opt.bs, 0, device)
import contextlib
if os.environ.get("QS_THREADS"):
    torch.set_num_threads(int(os.environ["QS_THREADS"]))
extra = [torch.cuda.Stream(priority=(-1 if os.environ.get("QS_EXTRA_HP") else 0)) for _ in range(int(os.environ.get("QS_EXTRA_STREAMS", 0)))]
for e in extra:      # a stream only takes a hardware queue once it has been used
    with torch.cuda.stream(e):
        torch.zeros(4, device="cuda").add_(1)
torch.cuda.synchronize()
hp = torch.cuda.Stream(priority=-1) if os.environ.get("QS_HIPRI") else None
ctx = torch.cuda.stream(hp) if hp is not None else contextlib.nullcontext()
ctx.__enter__()
for i in range(10): st.run(*batches[i % 8])
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 60
host = 0.0
for i in range(N):
    h0 = time.perf_counter()
    st.run(*batches[i % 8])
    host += time.perf_counter() - h0
t_host_done = time.perf_counter()
torch.cuda.synchronize()
if os.environ.get("QS_PARKED"):
    # pure host cost of enqueueing a step: the device is parked behind a long spin kernel, so no call waits for it
    # (a step is ~700 queue packets; three of them fit the queue)
    for k in range(3):
        torch.cuda.synchronize()
        torch.cuda._sleep(int(2.0e9 * 0.04))
        h0 = time.perf_counter()
        for i in range(3):
            st.run(*batches[i % 8])
        h1 = time.perf_counter()
        torch.cuda.synchronize()
        print("host enqueue with the device parked: %.3f ms/step" % ((h1 - h0) / 3 * 1e3))
from combat_amd.engine import Plan
nd = getattr(Plan, "_ndelay", 0) / (N + 10)
print("delay launches/step %.1f" % nd)
print("%-40s %.3f ms/step" % (" ".join("%s=%s" % (k[11:] if k.startswith("COMBAT_EXP_") else k, v) for k, v in sorted(os.environ.items()) if k.startswith("COMBAT_EXP_") or k.startswith("QS_")) or "-", (time.perf_counter() - t0) / N * 1e3), " host in run() %.3f ms/step, host loop done after %.3f ms/step" % (host / N * 1e3, (t_host_done - t0) / N * 1e3))
