"""Phase timeline of one step: an event before and after every plan, on the stream it runs on."""
import os, sys, time, collections, numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
from combat_amd import step as step_mod
from combat_amd.engine import Plan
device = torch.device("cuda", 0)
opt = bench.Opt()
np.random.seed(0); torch.manual_seed(100)
nets = bench.build_nets(device)
st = step_mod.AlternatedStep(*nets, opt)
batches = bench.synth_batches(8, opt.bs, 0, device)
orig = Plan.run
rec = None
def run(self, prof=None, on_mark=None):
    if rec is None:
        return orig(self, prof, on_mark)
    s = torch.cuda.current_stream()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(s); orig(self, prof, on_mark); e1.record(s)
    rec.append((self.name, e0, e1))
Plan.run = run
def measure(serial, steps=20):
    global rec
    st.serial = Plan.serial = serial
    for i in range(5): st.run(*batches[i % 8])
    torch.cuda.synchronize()
    agg = collections.OrderedDict(); tot = 0.0
    for i in range(steps):
        rec = []
        b0 = torch.cuda.Event(enable_timing=True); b1 = torch.cuda.Event(enable_timing=True)
        b0.record(); st.run(*batches[i % 8]); b1.record()
        torch.cuda.synchronize()
        tot += b0.elapsed_time(b1)
        for name, e0, e1 in rec:
            a = agg.setdefault(name, [0.0, 0.0, 0.0])
            a[0] += b0.elapsed_time(e0); a[1] += b0.elapsed_time(e1); a[2] += e0.elapsed_time(e1)
        rec = None
    print("serial=%s  step %.3f ms (synchronised per step)" % (serial, tot / steps))
    for name, a in agg.items():
        print("  %-12s start %7.3f  end %7.3f  dur %7.3f" % (name, a[0] / steps, a[1] / steps, a[2] / steps))
    sys.stdout.flush()
measure(True)
measure(False)
