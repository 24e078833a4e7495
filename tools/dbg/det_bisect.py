"""Which gradient differs between two runs of one alternated step in deterministic mode?  (keep_grads: the flat gradient
buffers stay as the step computed them.)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
from combat_amd import step as step_mod, engine

engine.set_deterministic(os.environ.get("DET", "1") == "1")
device = torch.device("cuda", 0)


def run():
    opt = bench.Opt()
    import random; random.seed(0); np.random.seed(0); torch.manual_seed(100)
    nets = bench.build_nets(device)
    batches = bench.synth_batches(2, opt.bs, 0, device)
    st = step_mod.AlternatedStep(*nets, opt)
    st.keep_grads = True
    init = [p.detach().clone() for m in nets[:2] for p in m.parameters()]
    st.run(*batches[0])
    torch.cuda.synchronize()
    out = {}
    for tag, eng in (("C", st.eC), ("G", st.eG)):
        for name, (o, n, shape) in eng.fp.offsets.items():
            out[tag + "." + name] = eng.fp.grad[o:o + n].clone()
    extra = {"bd": st.bd.clone(), "d_bd": st.d_bd.clone(), "d_bd2": st.d_bd2.clone(), "inputs": st.inputs.clone()}
    return init, out, extra


ia, ga, ea = run()
ib, gb, eb = run()
print("initial parameters equal:", all(torch.equal(u, v) for u, v in zip(ia, ib)))
for k in ea:
    print("%-8s equal: %s" % (k, torch.equal(ea[k], eb[k])))
bad = [k for k in ga if not torch.equal(ga[k], gb[k])]
print("%d of %d gradients differ" % (len(bad), len(ga)))
for k in bad[:60]:
    d = (ga[k] - gb[k]).abs().max().item()
    print("  %-40s max diff %.3e  (max %.3e, %d elements)" % (k, d, ga[k].abs().max().item(), ga[k].numel()))
