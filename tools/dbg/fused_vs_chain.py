"""PreActEngine train forward + backward with the in-LDS prologue against the round-3 chain: every stored tensor and the gradient."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from combat_amd import nets, engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
res = {}
for fused in (False, True):
    engine.FUSED_PROLOGUE = fused
    torch.manual_seed(0)
    net = nets.PreActResNet18().cuda()
    eng = net._net_engine()
    eng.refresh()
    slot = eng.slot("dbg", B, 32)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, 32, 32, 8, generator=g).to(torch.bfloat16).cuda()
    eng.input(slot).copy_(x)
    h = eng.head_bufs(slot)
    h["targets"].copy_(torch.randint(0, 10, (B,), generator=g))
    fwd = eng.forward_plan(slot, True)
    bwd = eng.backward_train_plan(slot)
    fwd.run()
    bwd.run()
    torch.cuda.synchronize()
    res[fused] = ({k: v.clone() for k, v in slot.bufs.items()}, eng.fp.grad.clone(), [c[0] if isinstance(c, tuple) else str(c) for c in getattr(fwd, "calls", [])])
    print("fused", fused, "forward calls", len(fwd), "backward calls", len(bwd))
a, b = res[False], res[True]
for k in sorted(a[0]):
    if k in b[0] and a[0][k].shape == b[0][k].shape and a[0][k].dtype == b[0][k].dtype:
        eq = torch.equal(a[0][k], b[0][k])
        if not eq:
            d = (a[0][k].float() - b[0][k].float()).abs()
            print("%-24s DIFF max %.4g  frac %.4f" % (k, float(d.max()), float((d > 0).float().mean())))
print("grad rel diff", float((a[1] - b[1]).norm() / a[1].norm()))
fp = eng.fp
worst = []
for name, (o, nn, _) in fp.offsets.items():
    ga, gb = a[1][o:o + nn], b[1][o:o + nn]
    r = float((ga - gb).norm() / max(float(ga.norm()), 1e-30))
    worst.append((r, name, float(ga.norm()), float(gb.norm())))
for r, name, na, nb in sorted(worst, reverse=True)[:25]:
    print("%-36s rel %.4f  |chain| %.4g |fused| %.4g" % (name, r, na, nb))
