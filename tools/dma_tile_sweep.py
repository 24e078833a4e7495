import os, sys, math, torch
sys.path.insert(0, os.getcwd())
from combat_amd import ops
bf16 = torch.bfloat16
def run(n, hw, c, k, tile, reps=40):
    x = torch.randn(n, hw, hw, c, device='cuda').to(bf16)
    w = (torch.randn(k, c, 3, 3, device='cuda') / math.sqrt(9*c)).contiguous(memory_format=torch.channels_last)
    pc = ops.PackedConv(w, 1, 1, c); pc.pack()
    y = torch.empty(n, hw, hw, k, dtype=bf16, device='cuda')
    r = torch.randn(n, hw, hw, k, device='cuda').to(bf16)
    try:
        a = ops.conv_args(x, y, pc, 0, add_post=r, stats_kind=1, tile=tile)
        rows, _ = ops.conv_stats_layout(a)
    except Exception as e:
        return float('nan')
    st = torch.zeros(rows, 2, k, device='cuda'); a.stats = st.data_ptr()
    for _ in range(3): ops.conv_launch(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): ops.conv_launch(a)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for name, shape in [('layer1 32x32 64', (128, 32, 64, 64)), ('layer2 16x16 128', (128, 16, 128, 128)), ('layer2 N256', (256, 16, 128, 128)),
                    ('layer3 8x8 256', (128, 8, 256, 256)), ('layer3 N256', (256, 8, 256, 256)), ('layer4 4x4 512', (128, 4, 512, 512)), ('layer4 N256', (256, 4, 512, 512)),
                    ('unet 16x16 64', (128, 16, 64, 64)), ('unet 16x16 128->64', (128, 16, 128, 64)), ('unet 8x8 128', (128, 8, 128, 128)), ('unet 8x8 256', (128, 8, 256, 256)), ('unet 4x4 256', (128, 4, 256, 256))]:
    print('%-22s %s' % (name, '  '.join('t%d:%.1f' % (t, run(*shape, t)) for t in (0, 10, 11))), flush=True)
