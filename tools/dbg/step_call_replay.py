"""After one fused step: replay single calls of Phase C's forward plan in isolation and check run-to-run determinism."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from combat_amd import nets, engine, step as step_mod
from combat_amd._lib import lib, ConvArgs
from dp_rehearsal import Opt
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
out = []
for seed, ctor in ((0, nets.PreActResNet18), (1, nets.PreActResNet18), (2, lambda: nets.UnetGenerator(None)), (3, lambda: nets.FrequencyModel(2, 3, 32))):
    torch.manual_seed(seed)
    out.append(ctor().cuda())
netc, clean, netg, netf = out
st = step_mod.AlternatedStep(netc, netg, clean.eval(), netf.eval(), Opt())
st.serial = engine.Plan.serial = True
g = torch.Generator().manual_seed(7)
x = ((torch.randint(0, 256, (B, 3, 32, 32), generator=g).float() / 255) - 0.5) / 0.5
t = torch.randint(0, 10, (B,), generator=g)
t[:3] = 0
xd = x.cuda()
st.run(xd, t, step_mod.StepRandomness(2, 0.4, 0.7, [None] * 5))
torch.cuda.synchronize()
P = st.pl["C_train_f"]
s0 = torch.cuda.current_stream().cuda_stream
bufs = st.sC_train.bufs
for i, (cfunc, args, what) in enumerate(P.calls[:12]):
    if cfunc is lib.combat_conv_gemm:
        a = args[0]._obj
        print(i, what, "tile", lib.combat_conv_pick_tile(ctypes.byref(a)), "pro", bool(a.pro_scale), "side", bool(a.pro_act_dst), "stats_kind", a.stats_kind,
              "ws", bool(a.workspace), "add_post", bool(a.add_post), "N H W C K", a.N, a.H, a.W, a.C, a.K)
        if a.pro_scale:
            # which slot buffer is dst?
            name = [k for k, v in bufs.items() if v.data_ptr() == a.dst][0]
            for rep_ in range(2):
                prev = None
                pat = ""
                for r in range(24):
                    bufs[name].zero_()
                    cfunc(*args, s0)
                    torch.cuda.synchronize()
                    cur_ = bufs[name].clone()
                    pat += "." if prev is None or torch.equal(prev, cur_) else "X"
                    prev = cur_
                print("    isolated replay (X = differs from the previous launch):", pat, " zeros left:", int((cur_.float() == 0).sum()))
    else:
        print(i, what)
# replay the plan prefix [0, k) repeatedly: which prefix length makes b0.y1 nondeterministic?
for k in (3, 4, 5):
    res = []
    for r in range(6):
        for (cfunc, args, what) in P.calls[:k]:
            cfunc(*args, s0)
        torch.cuda.synchronize()
        res.append(bufs["b0.y1"].clone())
    print("prefix", k, [w for _, _, w in P.calls[:k]][-1], "b0.y1 deterministic:", all(torch.equal(res[0], r_) for r_ in res))
# ---- bisect call 2: change one field at a time
import copy
cfunc, args, what = P.calls[2]
a0_ = args[0]._obj
def variant(name, **kw):
    a = ConvArgs()
    ctypes.memmove(ctypes.byref(a), ctypes.byref(a0_), ctypes.sizeof(ConvArgs))
    keep = []
    for k, v in kw.items():
        if isinstance(v, torch.Tensor):
            keep.append(v)
            v = v.data_ptr()
        setattr(a, k, v)
    dstbuf = torch.zeros_like(bufs["b0.y1"]) if "dst" not in kw else kw["dst"]
    if "dst" not in kw:
        a.dst = dstbuf.data_ptr()
    res = []
    for r in range(30):
        dstbuf.fill_(7.0)
        lib.combat_conv_gemm(ctypes.byref(a), s0)
        torch.cuda.synchronize()
        res.append(dstbuf.clone())
    nd = sum(int(not torch.equal(res[0], r_)) for r_ in res[1:])
    left = int((res[-1].float() == 7.0).sum())
    print("%-44s differing %2d / 29   elements still 7.0: %d" % (name, nd, left))
variant("as recorded (fresh dst)")
variant("dst = the slot's own y1", dst=bufs["b0.y1"])
variant("no statistics", stats_kind=0, stats=None)
variant("no side tensor", pro_act_dst=None)
variant("src = clone", src=bufs["stem"].clone())
#variant("scale/shift = clones", pro_scale=torch.empty(64, device="cuda").copy_(st.sC_train.norm["layer1.0.bn1"].scale), pro_shift=torch.empty(64, device="cuda").copy_(st.sC_train.norm["layer1.0.bn1"].shift))
variant("weights = clone", wpack=st.eC.blocks[0].conv1.wf.clone())
variant("tile 10", tile=10)
print("scale finite:", bool(torch.isfinite(st.sC_train.norm["layer1.0.bn1"].scale).all()), "stem finite:", bool(torch.isfinite(bufs["stem"].float()).all()),
      "weights finite:", bool(torch.isfinite(st.eC.blocks[0].conv1.wf.float()).all()))
print("ptrs: src %x dst %x side %x stats %x scale %x shift %x w %x" % (a0_.src, a0_.dst, a0_.pro_act_dst, a0_.stats, a0_.pro_scale, a0_.pro_shift, a0_.wpack))
