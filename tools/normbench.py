import os, sys, torch
sys.path.insert(0, os.getcwd())
from combat_amd import ops
from combat_amd._lib import lib
bf16 = torch.bfloat16
def st(): return torch.cuda.current_stream().cuda_stream
def run(groups, pxg, c, gran=32, reps=50):
    rows = groups * pxg
    x = torch.randn(rows, c, device='cuda').to(bf16)
    dz = torch.randn(rows, c, device='cuda').to(bf16)
    rpg = max(pxg // gran, 1)
    part = torch.randn(groups * rpg, 2, c, device='cuda')
    mean, rstd, scale, shift = (torch.empty(groups, c, device='cuda') for _ in range(4))
    act = torch.empty_like(x); dx = torch.empty_like(x)
    scratch = torch.empty(ops.norm_scratch_bytes(groups, c) // 4, device='cuda')
    gamma = torch.ones(c, device='cuda'); beta = torch.zeros(c, device='cuda')
    def f():
        ops.check(lib.combat_norm_act_fused(x.data_ptr(), part.data_ptr(), groups, rpg, pxg, c, 1e-5, 0.0, gamma.data_ptr() if groups == 1 else None,
                  beta.data_ptr() if groups == 1 else None, mean.data_ptr(), rstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), None, None, 0.1, None,
                  scratch.data_ptr(), scratch.numel() * 4, act.data_ptr(), st()), "f")
    def b():
        ops.check(lib.combat_norm_bwd_fused(dz.data_ptr(), x.data_ptr(), None, part.data_ptr(), groups, rpg, pxg, c, gamma.data_ptr() if groups == 1 else None,
                  mean.data_ptr(), rstd.data_ptr(), None, None, scratch.data_ptr(), scratch.numel() * 4, dx.data_ptr(), st()), "b")
    def a():
        ops.check(lib.combat_affine_act(x.data_ptr(), rows, c, scale.data_ptr(), shift.data_ptr(), pxg if groups > 1 else 0, 0.0, act.data_ptr(), st()), "a")
    out = []
    for fn in (f, b, a):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / reps * 1e3)
    print('groups %4d pxg %6d C %3d rows/group %4d: act_fused %.1f us  bwd_fused %.1f us  (affine_act alone %.1f us)  tensor %.1f MB' % (groups, pxg, c, rpg, out[0], out[1], out[2], rows * c * 2 / 1e6), flush=True)
run(1, 131072, 64, 512); run(1, 32768, 128, 128); run(1, 8192, 256, 128); run(1, 2048, 512, 128)   # BatchNorm: one row per workgroup
run(1, 131072, 64); run(1, 32768, 128)     # ... per wave (norm_stage1 first)
run(128, 256, 64); run(128, 64, 128); run(128, 256, 128)
