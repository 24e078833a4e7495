"""Phase C's train forward alone (no optimiser step) on real-image-shaped input, fused vs chain, + recomputation of b0.y1."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from combat_amd import nets, engine, ops
from combat_amd._lib import lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
mode = sys.argv[2] if len(sys.argv) > 2 else "image"
res = {}
for fused in (False, True):
    engine.FUSED_PROLOGUE = fused
    torch.manual_seed(0)
    net = nets.PreActResNet18().cuda()
    eng = net._net_engine()
    eng.refresh()
    slot = eng.slot("dbg", B, 32)
    g = torch.Generator().manual_seed(7)
    x = ((torch.randint(0, 256, (B, 3, 32, 32), generator=g).float() / 255) - 0.5) / 0.5
    if mode == "image":
        xd = x.cuda()
        inp = eng.input(slot)
        ops.check(lib.combat_image_to_c8(xd.data_ptr(), B, 32, inp.data_ptr(), torch.cuda.current_stream().cuda_stream), "c8")
        torch.cuda.synchronize()
    else:
        eng.input(slot).copy_(torch.randn(B, 32, 32, 8, generator=g).to(torch.bfloat16).cuda())
    eng.head_bufs(slot)["targets"].copy_(torch.randint(0, 10, (B,), generator=g))
    fwd = eng.forward_plan(slot, True)
    fwd.run()
    torch.cuda.synchronize()
    bufs = slot.bufs
    blk = eng.blocks[0]
    y_plain = torch.zeros_like(bufs["b0.y1"])
    ops.conv_launch(ops.conv_args(bufs["b0.a0"], y_plain, blk.conv1, 0, tile=11 if B <= 16 else 0))
    torch.cuda.synchronize()
    f = lambda u, v: "equal" if torch.equal(u, v) else "DIFF frac %.3f max %.3g" % (float((u.float() != v.float()).float().mean()), float((u.float() - v.float()).abs().max()))
    print("fused", fused, "stored y1 vs conv(a0) now:", f(bufs["b0.y1"], y_plain), " loss", float(eng.head_bufs(slot)["loss"]))
    res[fused] = {k: v.clone() for k, v in bufs.items()}
for k in ("stem", "b0.a0", "b0.y1", "b0.a1t", "b0.out", "b1.y1"):
    print(k, "equal" if torch.equal(res[False][k], res[True][k]) else "DIFF")
