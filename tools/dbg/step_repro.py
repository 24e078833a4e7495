"""Two identical fused steps in one process: which stored tensor of Phase C's forward differs first?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from combat_amd import nets, engine, step as step_mod
from dp_rehearsal import Opt
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
engine.FUSED_PROLOGUE = (len(sys.argv) <= 2 or sys.argv[2] != "chain")
runs = []
for rep in range(3):
    out = []
    for seed, ctor in ((0, nets.PreActResNet18), (1, nets.PreActResNet18), (2, lambda: nets.UnetGenerator(None)), (3, lambda: nets.FrequencyModel(2, 3, 32))):
        torch.manual_seed(seed)
        out.append(ctor().cuda())
    netc, clean, netg, netf = out
    st = step_mod.AlternatedStep(netc, netg, clean.eval(), netf.eval(), Opt())
    st.keep_grads = True
    st.serial = engine.Plan.serial = True
    g = torch.Generator().manual_seed(7)
    x = ((torch.randint(0, 256, (B, 3, 32, 32), generator=g).float() / 255) - 0.5) / 0.5
    t = torch.randint(0, 10, (B,), generator=g)
    t[:3] = 0
    xd = x.cuda()
    st.run(xd, t, step_mod.StepRandomness(2, 0.4, 0.7, [None] * 5))
    torch.cuda.synchronize()
    m = st.read_metrics()
    print("rep", rep, "loss_c %.6f" % m["loss_c_sum"])
    d = {k: v.clone() for k, v in st.sC_train.bufs.items()}
    for k, ns in st.sC_train.norm.items():
        d["norm." + k + ".scale"] = ns.scale.clone(); d["norm." + k + ".shift"] = ns.shift.clone()
    d["input"] = st.eC.input(st.sC_train).clone()
    runs.append(d)
order = ["input", "stem", "layer1.0.bn1.part", "norm.layer1.0.bn1.scale", "norm.layer1.0.bn1.shift", "b0.a0", "b0.y1", "layer1.0.bn2.part",
         "norm.layer1.0.bn2.scale", "b0.a1t", "b0.out", "layer1.1.bn1.part", "norm.layer1.1.bn1.scale", "b1.a0", "b1.y1", "b1.a1t", "b1.out", "b2.a0", "b2.y1", "b2.sc", "b2.out"]
for k in order:
    if k in runs[0]:
        print("%-28s run0 vs run1: %s   run1 vs run2: %s" % (k, "equal" if torch.equal(runs[0][k], runs[1][k]) else "DIFF", "equal" if torch.equal(runs[1][k], runs[2][k]) else "DIFF"))
