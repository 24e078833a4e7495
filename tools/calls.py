"""Serial per-call timing of every plan call (events around each), aggregated."""
import os, sys, collections, numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
from combat_amd import step as step_mod
from combat_amd.engine import Plan
from combat_amd._lib import lib
device = torch.device("cuda", 0)
opt = bench.Opt()
np.random.seed(0); torch.manual_seed(100)
st = step_mod.AlternatedStep(*bench.build_nets(device), opt)
batches = bench.synth_batches(8, opt.bs, 0, device)
names = {}
for k in dir(lib):
    if k.startswith("combat_"):
        try: names[id(getattr(lib, k))] = k
        except Exception: pass
rec = []
def run(self, prof=None, on_mark=None):
    s = torch.cuda.current_stream(); stp = s.cuda_stream
    for ci, (cfunc, args, what) in enumerate(self.calls):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(s); rc = cfunc(*args, stp); e1.record(s)
        assert rc == 0
        rec.append((self.name, what, getattr(cfunc, "__name__", None) or names.get(id(cfunc), "?"), e0, e1))
Plan.run = run
st.serial = Plan.serial = True
for i in range(3): st.run(*batches[i])
torch.cuda.synchronize()
rec.clear()
S = 5
for i in range(S): st.run(*batches[i])
torch.cuda.synchronize()
byplan = collections.OrderedDict(); byfn = collections.defaultdict(lambda: [0.0, 0])
bycall = collections.OrderedDict()
for pn, what, fn, e0, e1 in rec:
    us = e0.elapsed_time(e1) * 1e3
    byplan.setdefault(pn, [0.0, 0]); byplan[pn][0] += us; byplan[pn][1] += 1
    byfn[fn][0] += us; byfn[fn][1] += 1
    k = (pn, what, fn); bycall.setdefault(k, [0.0, 0]); bycall[k][0] += us; bycall[k][1] += 1
print("== per plan (us/step, calls/step)")
for k, (us, c) in byplan.items(): print("  %-28s %8.1f %5.1f" % (k, us / S, c / S))
print("== per entry point")
for k, (us, c) in sorted(byfn.items(), key=lambda x: -x[1][0]): print("  %-28s %8.1f %5.1f  avg %5.1f" % (k, us / S, c / S, us / c))
print("== per call")
for (pn, what, fn), (us, c) in bycall.items(): print("  %-24s %-34s %-24s %7.1f x%d" % (pn[-24:], what[-34:], fn[-24:], us / c, c // S))
