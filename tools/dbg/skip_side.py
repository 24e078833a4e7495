"""How much of the step is the second stream's work?  ms/step with its plans replaced by no-ops (results are garbage:
timing only) -- the upper bound of what making the side work cheaper could buy."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from combat_amd import step as step_mod

class Nop:
    def run(self, *a, **k):
        pass

dev = torch.device("cuda", 0)
for skip in ((), ("K_eval_f", "K_bd_b"), ("C_met_f", "F_f"), ("K_eval_f", "K_bd_b", "C_met_f", "F_f")):
    np.random.seed(0); torch.manual_seed(0)
    opt = bench.Opt()
    st = step_mod.AlternatedStep(*bench.build_nets(dev), opt)
    batches = bench.synth_batches(8, opt.bs, 0, dev)
    st.run(*batches[0])
    for k in skip:
        st.pl[k] = Nop()
    for i in range(10):
        st.run(*batches[i % 8])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(60):
        st.run(*batches[i % 8])
    torch.cuda.synchronize()
    print("skipped %-40s %.3f ms/step" % (",".join(skip) or "(nothing)", (time.perf_counter() - t0) / 60 * 1e3), flush=True)
