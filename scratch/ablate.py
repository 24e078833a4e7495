import os, sys, math, torch
sys.path.insert(0, os.getcwd())
from combat_amd import ops
bf16 = torch.bfloat16
def run(n, hw, c, k, tile, pro=True, resid=True, stats=True, reps=30):
    x = torch.randn(n, hw, hw, c, device='cuda').to(bf16)
    w = (torch.randn(k, c, 3, 3, device='cuda') / math.sqrt(9*c)).contiguous(memory_format=torch.channels_last)
    pc = ops.PackedConv(w, 1, 1, c); pc.pack()
    y = torch.empty(n, hw, hw, k, dtype=bf16, device='cuda')
    sc, sh = torch.rand(c, device='cuda') + .5, torch.randn(c, device='cuda')
    r = torch.randn(n, hw, hw, k, device='cuda').to(bf16)
    a = ops.conv_args(x, y, pc, 0, pro=ops.Affine(sc, sh, 0, True, 0.0) if pro else None, add_post=r if resid else None, stats_kind=1 if stats else 0, tile=tile)
    rows, _ = ops.conv_stats_layout(a)
    st = torch.zeros(rows, 2, k, device='cuda'); a.stats = st.data_ptr()
    for _ in range(3): ops.conv_launch(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): ops.conv_launch(a)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    fl = 2.0 * n*hw*hw * k * c * 9
    return us, fl / us / 1e6
if __name__ == "__main__":
  for name, args in [('layer1 128x32x32 64->64 H256x64', (128, 32, 64, 64, 6)), ('layer2 128x16x16 128->128 H128x128', (128, 16, 128, 128, 7)),
                     ('layer3 128x8x8 256->256 H128x64', (128, 8, 256, 256, 8)), ('layer4 128x4x4 512->512 H128x128', (128, 4, 512, 512, 7)), ('layer4 H64x64', (128, 4, 512, 512, 9))]:
      us, tf = run(*args)
      us2, tf2 = run(*args, pro=False, resid=False, stats=False)
      print('%-40s dbg=%s  full-epilogue %.1f us %.0f TF/s | plain %.1f us %.0f TF/s' % (name, os.environ.get('COMBAT_DEBUG_SKIP', '0'), us, tf, us2, tf2))
