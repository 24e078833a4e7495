"""AlternatedStep (one step) with the in-LDS prologue against the round-3 chain: Phase C forward tensors, loss, gradient."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from combat_amd import nets, engine, step as step_mod
from dp_rehearsal import Opt
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
serial = len(sys.argv) > 2 and sys.argv[2] == "serial"
res = {}
for fused in (False, True):
    engine.FUSED_PROLOGUE = fused
    out = []
    for seed, ctor in ((0, nets.PreActResNet18), (1, nets.PreActResNet18), (2, lambda: nets.UnetGenerator(None)), (3, lambda: nets.FrequencyModel(2, 3, 32))):
        torch.manual_seed(seed)
        out.append(ctor().cuda())
    netc, clean, netg, netf = out
    st = step_mod.AlternatedStep(netc, netg, clean.eval(), netf.eval(), Opt())
    st.keep_grads = True
    if serial:
        st.serial = engine.Plan.serial = True
    g = torch.Generator().manual_seed(7)
    x = ((torch.randint(0, 256, (B, 3, 32, 32), generator=g).float() / 255) - 0.5) / 0.5
    t = torch.randint(0, 10, (B,), generator=g)
    t[:3] = 0
    st.eC.refresh()
    blk0 = st.eC.blocks[0]
    wf_old = blk0.conv1.wf.clone()
    xd = x.cuda()
    st.run(xd, t, step_mod.StepRandomness(2, 0.4, 0.7, [None] * 5))
    torch.cuda.synchronize()
    m = st.read_metrics()
    from combat_amd import ops as _ops
    wf_new = blk0.conv1.wf.clone()
    blk0.conv1.wf.copy_(wf_old)
    bufs_ = st.sC_train.bufs
    y_old = torch.zeros_like(bufs_["b0.y1"])
    _ops.conv_launch(_ops.conv_args(bufs_["b0.a0"], y_old, blk0.conv1, 0, tile=11))
    torch.cuda.synchronize()
    blk0.conv1.wf.copy_(wf_new)
    print("   stored y1 vs conv(a0, weights before the step):", "equal" if torch.equal(y_old, bufs_["b0.y1"]) else "DIFF frac %.3f" % float((y_old.float() != bufs_["b0.y1"].float()).float().mean()))
    print("fused", fused, "loss_c %.6f loss_ce %.6f" % (m["loss_c_sum"], m["loss_ce_sum"]))
    res[fused] = ({k: v.clone() for k, v in st.sC_train.bufs.items()}, st.eC.fp.grad.clone())
a, b = res[False], res[True]
n = 0
for k in sorted(a[0]):
    if k in b[0] and a[0][k].shape == b[0][k].shape and a[0][k].dtype == b[0][k].dtype and not k.startswith("g."):
        if not torch.equal(a[0][k], b[0][k]):
            d = (a[0][k].float() - b[0][k].float()).abs()
            print("%-24s DIFF max %.4g  frac %.4f" % (k, float(d.max()), float((d > 0).float().mean())))
            n += 1
            if n > 30:
                break
print("grad rel diff", float((a[1] - b[1]).norm() / a[1].norm()))
# ---- which side deviates?  recompute b0.y1 from the stored tensors of the FUSED run
import ctypes
from combat_amd import ops
from combat_amd._lib import lib
eng = st.eC
bufs = st.sC_train.bufs
blk = eng.blocks[0]
stn = st.sC_train.norm[blk.bn1.prefix]
y_plain = torch.zeros_like(bufs["b0.y1"])
ops.conv_launch(ops.conv_args(bufs["b0.a0"], y_plain, blk.conv1, 0))
y_pro = torch.zeros_like(bufs["b0.y1"])
side = torch.zeros_like(bufs["b0.a0"])
apro = ops.conv_args(bufs["stem"], y_pro, blk.conv1, 0, pro=ops.Affine(stn.scale, stn.shift, 0, True, 0.0), pro_act_dst=side)
print("tile of the prologue launch", lib.combat_conv_pick_tile(ctypes.byref(apro)))
ops.conv_launch(apro)
torch.cuda.synchronize()
f = lambda u, v: "equal" if torch.equal(u, v) else "DIFF frac %.3f" % float((u.float() != v.float()).float().mean())
print("stored(fused) y1 vs conv(a0):", f(bufs["b0.y1"], y_plain))
print("stored(fused) y1 vs pro conv now:", f(bufs["b0.y1"], y_pro))
print("pro conv now vs conv(a0):", f(y_pro, y_plain), " side vs a0:", f(side, bufs["b0.a0"]))
print("stored(chain) y1 vs conv(a0):", f(a[0]["b0.y1"], y_plain), " chain a0 vs fused a0:", f(a[0]["b0.a0"], bufs["b0.a0"]))
print("scale ptr %x shift ptr %x" % (stn.scale.data_ptr(), stn.shift.data_ptr()))
ya, yb = a[0]["b0.y1"].float(), b[0]["b0.y1"].float()
bad = (ya != yb)
print("b0.y1 mismatch per channel(8 groups):", [round(float(bad[..., i*8:(i+1)*8].float().mean()), 2) for i in range(8)])
print("per image:", [round(float(bad[i].float().mean()), 2) for i in range(bad.shape[0])])
print("per row:", [round(float(bad[:, i].float().mean()), 2) for i in range(32)])
print("per col:", [round(float(bad[:, :, i].float().mean()), 2) for i in range(32)])
print("mean abs diff", float((ya - yb).abs().mean()), "mean abs", float(ya.abs().mean()))
pa, pb = a[0]["layer1.0.bn2.part"], b[0]["layer1.0.bn2.part"]
print("bn2.part shapes", pa.shape, pb.shape)
