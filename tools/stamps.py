import os, sys, math, ctypes, torch
sys.path.insert(0, os.getcwd())
import numpy as np
from combat_amd import ops, _lib
bf16 = torch.bfloat16
lib = ctypes.CDLL(_lib.LIB_PATH)
lib.combat_debug_set_stamps.argtypes = [ctypes.c_void_p]
lib.combat_debug_set_stamps_dma.argtypes = [ctypes.c_void_p]
def run(n, hw, c, k, tile, mode=0):
    x = torch.randn(n, hw, hw, c, device='cuda').to(bf16)
    w = (torch.randn(k, c, 3, 3, device='cuda') / math.sqrt(9*c)).contiguous(memory_format=torch.channels_last)
    pc = ops.PackedConv(w, 1, 1, c); pc.pack()
    y = torch.empty(n, hw, hw, k, dtype=bf16, device='cuda')
    sc, sh = torch.rand(c, device='cuda') + .5, torch.randn(c, device='cuda')
    r = torch.randn(n, hw, hw, k, device='cuda').to(bf16)
    a = ops.conv_args(x, y, pc, 0, pro=None if tile >= 10 else ops.Affine(sc, sh, 0, True, 0.0), add_post=r, stats_kind=1, tile=tile)
    rows, _ = ops.conv_stats_layout(a)
    st = torch.zeros(rows, 2, k, device='cuda'); a.stats = st.data_ptr()
    for _ in range(3): ops.conv_launch(a)
    torch.cuda.synchronize()
    stamps = torch.zeros(65536 * 16, dtype=torch.int64, device='cuda')
    setter = lib.combat_debug_set_stamps_dma if tile >= 10 else lib.combat_debug_set_stamps
    setter(ctypes.c_void_p(stamps.data_ptr()))
    ops.conv_launch(a)
    torch.cuda.synchronize()
    setter(ctypes.c_void_p(0))
    s = stamps.cpu().numpy().reshape(-1, 16)
    nb = int((s[:, 0] != 0).sum())
    s = s[:nb]
    cyc = s[:, :5].astype(np.float64); wall = s[:, 8:13].astype(np.float64)
    if tile >= 10: wall[:, 1:4] = wall[:, :1]
    d = np.diff(cyc, axis=1)
    t0 = wall[:, 0].min()
    print('  blocks %d  kernel span %.1f us (wall clock 100MHz)' % (nb, (wall[:, 4].max() - t0) / 100.0))
    names = ['stage halo+w', 'chunk0 taps', 'other chunks', 'epilogue']
    for i, nm in enumerate(names):
        print('   %-14s mean %7.0f cyc  p10 %7.0f  p90 %7.0f' % (nm, d[:, i].mean(), np.percentile(d[:, i], 10), np.percentile(d[:, i], 90)))
    c8 = s[:, :8].astype(np.float64)
    if c8[:, 5].any():
        print('   epilogue split: barrier+transpose %.0f  math %.0f  stores %.0f  stats %.0f' % ((c8[:, 5] - c8[:, 3]).mean(), (c8[:, 6] - c8[:, 5]).mean(), (c8[:, 7] - c8[:, 6]).mean(), (c8[:, 4] - c8[:, 7]).mean()))
    if s[:, 14].any(): print('   setup before the first DMA: mean %.0f cyc' % (s[:, 0].astype(np.float64) - s[:, 14].astype(np.float64)).mean())
    tot = cyc[:, 4] - cyc[:, 0]
    print('   %-14s mean %7.0f cyc   (wall mean %.2f us -> %.2f GHz)' % ('total', tot.mean(), ((wall[:, 4] - wall[:, 0]) / 100).mean(), tot.mean() / ((wall[:, 4] - wall[:, 0]).mean() * 10)))
    hw = s[:, 15]
    if hw.any():
        xcc = (hw >> 32) & 0xf; cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
        key = (xcc * 8 + se) * 32 + sh * 16 + cu
        uniq, cnt = np.unique(key, return_counts=True)
        print('   distinct CUs %d  blocks/CU min %d max %d;  per XCC %s' % (len(uniq), cnt.min(), cnt.max(), np.bincount(xcc.astype(int)).tolist()))
        # concurrency per CU: max overlapping blocks
        mx = []
        for u in uniq[:64]:
            sel = key == u
            ev = sorted([(w0, 1) for w0 in wall[sel, 0]] + [(w4, -1) for w4 in wall[sel, 4]])
            c = m = 0
            for _, d in ev:
                c += d; m = max(m, c)
            mx.append(m)
        print('   max concurrent blocks on a CU (first 64 CUs): %s' % np.bincount(mx).tolist())
    st_ = (wall[:, 0] - t0) / 100.0
    print('   block start times us: p0 %.1f p25 %.1f p50 %.1f p75 %.1f p100 %.1f' % tuple(np.percentile(st_, [0, 25, 50, 75, 100])))
for name, shape, tile in [('layer1 B128 t10', (128, 32, 64, 64), 10), ('layer1 B128 t16', (128, 32, 64, 64), 16), ('layer2 B128 t10', (128, 16, 128, 128), 10),
                    ('layer2 B128 t16', (128, 16, 128, 128), 16), ('layer3 B128 t11', (128, 8, 256, 256), 11), ('layer4 B128 t11', (128, 4, 512, 512), 11)]:
    print(name, flush=True)
    run(*shape, tile)
