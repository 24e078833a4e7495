"""Kernel-level parity on the MI355X: every C-ABI entry point against a plain fp32 PyTorch
reference of the same operator, computed on the CPU from the same (bf16-rounded) inputs.

Tolerances: bf16 storage has 8 significant bits, so a stored output is compared with
rel-L2 <= 4e-3 (pure rounding of the result) unless the test says otherwise; fp32 outputs
(statistics, gradients of weights, logits) with rel-L2 <= 2e-3 or tighter.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

bf16 = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    from combat_amd import ops as o
    return o


def dev(t):
    return t.to("cuda")


def rel_l2(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return float((a - b).norm() / max(float(b.norm()), 1e-30))


def nhwc(x):  # fp32 NCHW (cpu) -> bf16 NHWC (cuda)
    return dev(x.permute(0, 2, 3, 1).contiguous().to(bf16))


def nchw(x):  # bf16 NHWC (cuda) -> fp32 NCHW (cpu)
    return x.float().cpu().permute(0, 3, 1, 2).contiguous()


def rb(x):  # round to bf16 and back (what the kernel reads)
    return x.to(bf16).float()


def g(seed):
    return torch.Generator().manual_seed(seed)


def make_conv(ops, k, c, r, stride, pad, seed, c_pad=None, dup=False):
    w = torch.randn(k, c, r, r, generator=g(seed)) * (1.0 / math.sqrt(c * r * r))
    wd = dev(w).contiguous(memory_format=torch.channels_last)
    pc = ops.PackedConv(wd, stride, pad, c_pad or c, dup_hilo=dup)
    pc.pack()
    return w, pc


CONV_CASES = [
    # n, hw, c, k, r, stride, pad, tile
    (4, 32, 64, 64, 3, 1, 1, 0),
    (4, 32, 64, 64, 3, 1, 1, 3),      # 64x64 tile
    (2, 32, 64, 128, 3, 2, 1, 0),
    (2, 32, 64, 128, 1, 2, 0, 0),     # 1x1 stride-2 shortcut
    (8, 16, 128, 128, 3, 1, 1, 1),    # 128x128 tile
    (8, 16, 128, 128, 3, 1, 1, 5),    # 64x128 tile
    (8, 4, 512, 512, 3, 1, 1, 0),     # layer4 shape: skinny M
    (3, 5, 64, 64, 3, 1, 1, 0),       # ragged M (75 rows)
    (6, 2, 512, 512, 3, 1, 1, 0),     # UNet bottleneck 2x2
]


HALO_CASES = [
    # n, hw, c, k, tile   (3x3 / stride 1 / pad 1 with the input patch held in LDS)
    (4, 32, 64, 64, 6), (4, 32, 64, 64, 8), (4, 32, 64, 64, 9), (2, 32, 128, 64, 6),
    (8, 16, 128, 128, 7), (8, 16, 128, 128, 8), (8, 16, 64, 128, 9), (2, 16, 64, 64, 6),
    (3, 8, 256, 256, 7), (3, 8, 256, 256, 6), (5, 8, 128, 64, 8),      # several images per tile, ragged N
    (8, 4, 512, 512, 9), (9, 4, 512, 512, 8), (6, 2, 512, 512, 9), (4, 32, 64, 64, 0),
]


@pytest.mark.parametrize("n,hw,c,k,tile", HALO_CASES)
def test_conv3x3_halo_forward_and_dgrad(ops, n, hw, c, k, tile):
    from combat_amd._lib import lib
    import ctypes
    x = torch.randn(n, c, hw, hw, generator=g(1))
    w, pc = make_conv(ops, k, c, 3, 1, 1, 2)
    y = torch.empty(n, hw, hw, k, dtype=bf16, device="cuda")
    a = ops.conv_args(nhwc(x), y, pc, 0, tile=tile)
    picked = lib.combat_conv_pick_tile(ctypes.byref(a))
    assert picked >= 6 and (tile == 0 or picked == tile), picked
    ops.conv_launch(a)
    assert rel_l2(nchw(y), F.conv2d(rb(x), rb(w), padding=1)) < 4e-3
    dy = torch.randn(n, k, hw, hw, generator=g(3))
    dx = torch.empty(n, hw, hw, c, dtype=bf16, device="cuda")
    a = ops.conv_args(nhwc(dy), dx, pc, 1, tile=0 if c != k else tile)
    assert lib.combat_conv_pick_tile(ctypes.byref(a)) >= 6
    ops.conv_launch(a)
    assert rel_l2(nchw(dx), torch.nn.grad.conv2d_input((n, c, hw, hw), rb(w), rb(dy), padding=1)) < 4e-3


@pytest.mark.parametrize("n,hw,k,stride", [(5, 32, 64, 1), (3, 16, 128, 1), (4, 32, 64, 2), (2, 6, 64, 1), (3, 10, 64, 2)])
def test_conv_c8_direct_kernel(ops, n, hw, k, stride):
    """C = 8 inputs (c8 images / 3-channel gradients): the kernel that feeds MFMA operands straight from
    global memory must agree with the generic gather tile on the same arguments -- forward with bias,
    residual, statistics and an activated second output; stride-1 input gradient with a per-image
    InstanceNorm mask and the norm-backward sums -- and with torch."""
    from combat_amd._lib import lib
    import ctypes
    c = 8
    x = torch.randn(n, c, hw, hw, generator=g(201))
    w, pc = make_conv(ops, k, c, 3, stride, 1, 202)
    p, q = pc.out_hw(hw, hw)
    bias = torch.randn(k, generator=g(203)) * 0.1
    res = torch.randn(n, k, p, q, generator=g(204))
    sc, sh = torch.rand(k, generator=g(205)) + 0.5, torch.randn(k, generator=g(206)) * 0.3
    outs = []
    for tile in (15, 2):
        y = torch.empty(n, p, q, k, dtype=bf16, device="cuda")
        act = torch.empty_like(y)
        a = ops.conv_args(nhwc(x), y, pc, 0, bias=dev(bias), add_post=nhwc(res), stats_kind=1, tile=tile, act_dst=act,
                          act=ops.Affine(dev(sc), dev(sh), 0, True, 0.1))
        assert lib.combat_conv_pick_tile(ctypes.byref(a)) == tile
        rows, rpi = ops.conv_stats_layout(a)
        stats = torch.zeros(rows, 2, k, device="cuda")
        a.stats = stats.data_ptr()
        ops.conv_launch(a)
        outs.append((y, act, stats.sum(0)))
    assert rel_l2(outs[0][0].float(), outs[1][0].float()) < 2e-3 and rel_l2(outs[0][1].float(), outs[1][1].float()) < 2e-3
    assert rel_l2(outs[0][2], outs[1][2]) < 1e-3
    ref = F.conv2d(rb(x), rb(w), bias, stride=stride, padding=1) + rb(res)
    assert rel_l2(nchw(outs[0][0]), ref) < 4e-3
    a = ops.conv_args(nhwc(x), torch.empty(n, p, q, k, dtype=bf16, device="cuda"), pc, 0)
    assert lib.combat_conv_pick_tile(ctypes.byref(a)) == 15          # automatic choice
    if stride != 1:
        return
    # input gradient of a K = 8 convolution (here: C = k inputs, 8 outputs), per-image mask + sums
    w2, pc2 = make_conv(ops, 8, k, 3, 1, 1, 207)
    dy = torch.randn(n, 8, hw, hw, generator=g(208))
    xpre = torch.randn(n, k, hw, hw, generator=g(209))
    msc, msh = torch.rand(n, k, generator=g(210)) + 0.5, torch.randn(n, k, generator=g(211)) * 0.3
    mean, rstd = torch.randn(n, k, generator=g(212)) * 0.1, torch.rand(n, k, generator=g(213)) + 0.5
    outs = []
    for tile in (15, 2):
        dx = torch.empty(n, hw, hw, k, dtype=bf16, device="cuda")
        a = ops.conv_args(nhwc(dy), dx, pc2, 1, mask_x=nhwc(xpre), mask=ops.Affine(dev(msc), dev(msh), k, True, 0.2),
                          stats_kind=2, xh_mean=dev(mean), xh_rstd=dev(rstd), tile=tile)
        assert lib.combat_conv_pick_tile(ctypes.byref(a)) == tile
        rows, rpi = ops.conv_stats_layout(a)
        stats = torch.zeros(rows, 2, k, device="cuda")
        a.stats = stats.data_ptr()
        ops.conv_launch(a)
        outs.append((dx, stats.sum(0), rpi))
    assert rel_l2(outs[0][0].float(), outs[1][0].float()) < 2e-3 and rel_l2(outs[0][1], outs[1][1]) < 2e-3
    assert outs[0][2] == outs[1][2]
    gin = torch.nn.grad.conv2d_input((n, k, hw, hw), rb(w2), rb(dy), padding=1)
    keep = (rb(xpre) * msc[:, :, None, None] + msh[:, :, None, None]) > 0
    assert rel_l2(nchw(outs[0][0]), torch.where(keep, gin, 0.2 * gin)) < 4e-3


DMA_CASES = [
    # n, hw, c, k, tile   (prologue-free 3x3 convolutions with both operands DMA'd into LDS)
    (4, 32, 64, 64, 10), (2, 32, 128, 128, 10), (3, 16, 128, 128, 10), (5, 16, 64, 128, 10), (3, 16, 256, 64, 10),
    (3, 8, 256, 256, 10), (5, 8, 128, 64, 10), (2, 8, 64, 64, 10),      # two images per tile, ragged N
    (9, 4, 512, 512, 10), (8, 4, 128, 256, 10), (3, 4, 64, 64, 10),     # eight images per tile
    (4, 32, 64, 64, 11), (3, 16, 128, 64, 11), (5, 8, 128, 64, 11), (9, 4, 512, 512, 11), (7, 4, 256, 256, 0),  # 32-channel tiles
    (3, 32, 64, 64, 14), (2, 16, 128, 128, 14), (7, 8, 128, 64, 14), (5, 8, 64, 128, 14),     # 256-pixel tiles, ragged N
    (3, 32, 64, 64, 16), (2, 32, 128, 128, 16), (3, 16, 128, 128, 16), (5, 16, 64, 128, 16), (2, 16, 256, 64, 16),  # 64-pixel wave tiles
    (3, 32, 64, 64, 17), (5, 16, 64, 64, 17), (40, 32, 64, 64, 17), (81, 32, 64, 64, 17),   # weight-stationary persistent kernel: 1, 1, 2, 3 tiles per workgroup
]


@pytest.mark.parametrize("n,hw,c,k,tile", DMA_CASES)
def test_conv3x3_dma_forward_and_dgrad(ops, n, hw, c, k, tile):
    from combat_amd._lib import lib
    import ctypes
    x = torch.randn(n, c, hw, hw, generator=g(1))
    w, pc = make_conv(ops, k, c, 3, 1, 1, 2)
    y = torch.empty(n, hw, hw, k, dtype=bf16, device="cuda")
    res = torch.randn(n, k, hw, hw, generator=g(6))
    a = ops.conv_args(nhwc(x), y, pc, 0, add_post=nhwc(res), stats_kind=1, tile=tile)
    assert lib.combat_conv_pick_tile(ctypes.byref(a)) == (tile or 11)   # automatic: a skinny layer takes 32-channel tiles
    rows, rpi = ops.conv_stats_layout(a)
    stats = torch.zeros(rows, 2, k, device="cuda")
    a.stats = stats.data_ptr()
    ops.conv_launch(a)
    ref = F.conv2d(rb(x), rb(w), padding=1) + rb(res)
    assert rel_l2(nchw(y), ref) < 4e-3
    yr = nchw(y)
    assert rel_l2(stats.sum(0).cpu()[0], yr.sum((0, 2, 3))) < 1e-4
    if rpi:
        per_img = stats[: n * rpi].view(n, rpi, 2, k).sum(1).cpu()
        assert rel_l2(per_img[:, 1], (yr * yr).sum((2, 3))) < 1e-4
    dy = torch.randn(n, k, hw, hw, generator=g(3))
    dx = torch.empty(n, hw, hw, c, dtype=bf16, device="cuda")
    a = ops.conv_args(nhwc(dy), dx, pc, 1, tile=tile)
    assert lib.combat_conv_pick_tile(ctypes.byref(a)) == (tile or 11)
    ops.conv_launch(a)
    assert rel_l2(nchw(dx), torch.nn.grad.conv2d_input((n, c, hw, hw), rb(w), rb(dy), padding=1)) < 4e-3


@pytest.mark.parametrize("n,hw", [(3, 32), (40, 32), (81, 32), (128, 32), (67, 16), (200, 16)])
def test_conv3x3_weight_stationary_equals_ring_kernel(ops, n, hw):
    """COMBAT_TILE_S128x64 (filter bank resident in LDS, persistent workgroups, two wave groups half a tile apart)
    against COMBAT_TILE_D128x64 on the same arguments: same tiles, same MFMA order, same fused epilogue, so every
    output -- raw tensor, activated tensor, statistics rows -- must be BIT-identical, for each of the five epilogue
    flavours the step launches and for 1 ... 7 tiles per workgroup (odd counts leave the second group a tile short)."""
    from combat_amd._lib import lib
    import ctypes
    c = k = 64
    w, pc = make_conv(ops, k, c, 3, 1, 1, 40)
    x = nhwc(torch.randn(n, c, hw, hw, generator=g(41)))
    dy = nhwc(torch.randn(n, k, hw, hw, generator=g(42)))
    res = nhwc(torch.randn(n, k, hw, hw, generator=g(43)))
    xpre = nhwc(torch.relu(torch.randn(n, c, hw, hw, generator=g(44))))
    sc = dev((torch.rand(k, generator=g(45)) + 0.5) * torch.where(torch.rand(k, generator=g(46)) < 0.2, -1.0, 1.0))
    sh = dev(torch.randn(k, generator=g(47)) * 0.3)
    mean, rstd = dev(torch.randn(k, generator=g(48)) * 0.1), dev(torch.rand(k, generator=g(49)) + 0.5)
    aff = ops.Affine(sc, sh, 0, True, 0.0)
    cases = {
        "plain+residual": lambda t, o: ops.conv_args(x, o["y"], pc, 0, add_post=res, tile=t),
        "activated output": lambda t, o: ops.conv_args(x, o["y"], pc, 0, add_post=res, act_dst=o["act"], act=aff, tile=t),
        "activated output only": lambda t, o: ops.conv_args(x, None, pc, 0, act_dst=o["act"], act=aff, tile=t),
        "statistics": lambda t, o: ops.conv_args(x, o["y"], pc, 0, add_post=res, stats_kind=1, tile=t),
        "eval backward": lambda t, o: ops.conv_args(dy, o["y"], pc, 1, add_pre=res, mask_x=xpre, mask=aff, mask_mul_scale=True,
                                                     mask_activated=True, add_post=res, tile=t),
        "train backward": lambda t, o: ops.conv_args(dy, o["y"], pc, 1, add_pre=res, mask_x=xpre, mask=aff, stats_kind=2,
                                                      xh_mean=mean, xh_rstd=rstd, tile=t),
    }
    for name, make in cases.items():
        outs = []
        for tile in (10, 17):
            o = dict(y=torch.zeros(n, hw, hw, k, dtype=bf16, device="cuda"), act=torch.zeros(n, hw, hw, k, dtype=bf16, device="cuda"))
            a = make(tile, o)
            assert lib.combat_conv_pick_tile(ctypes.byref(a)) == tile, name
            if a.stats_kind:
                rows, rpi = ops.conv_stats_layout(a)
                o["stats"] = torch.zeros(rows, 2, k, device="cuda")
                a.stats = o["stats"].data_ptr()
                o["layout"] = (rows, rpi)
            ops.conv_launch(a)
            torch.cuda.synchronize()
            outs.append(o)
        for key in outs[0]:
            if key == "layout":
                assert outs[0][key] == outs[1][key], name
            else:
                assert torch.equal(outs[0][key], outs[1][key]), (name, key, n, hw)
    # and the automatic choice: the persistent kernel from two tiles per CU upwards
    a = ops.conv_args(x, torch.empty(n, hw, hw, k, dtype=bf16, device="cuda"), pc, 0)
    tiles = n * (hw // 16) * (hw // 8)
    assert lib.combat_conv_pick_tile(ctypes.byref(a)) == (17 if tiles >= 512 else (11 if tiles <= 256 else 10))


LDS_PRO_CASES = [
    # n, hw, c, k, tile: every 3x3 / stride-1 shape of PreActResNet18 (and the generator's) at B = 128, both channel
    # tiles; then ragged image counts on the maps where a tile holds several images, and other widths
    (128, 32, 64, 64, 10), (128, 16, 128, 128, 10), (128, 16, 128, 128, 11), (128, 8, 256, 256, 10), (128, 8, 256, 256, 11),
    (128, 4, 512, 512, 10), (128, 4, 512, 512, 11), (128, 8, 256, 128, 11), (128, 16, 128, 64, 10), (128, 4, 512, 256, 11),
    (5, 32, 64, 64, 11), (5, 8, 128, 64, 10), (9, 4, 256, 256, 11), (3, 16, 64, 128, 10), (2, 64, 64, 64, 10),
]


@pytest.mark.parametrize("n,hw,c,k,tile", LDS_PRO_CASES)
def test_conv_lds_prologue_equals_norm_act_then_conv(ops, n, hw, c, k, tile):
    """the train-mode BatchNorm + ReLU of a convolution's input applied IN LDS by the DMA-staged
    kernel (combat_conv_args.pro_* + pro_act_dst) against the chain it replaces -- combat_norm_act_fused materialises
    relu(bn(x)), then the prologue-free convolution on the same tile.  Same tiles, same MFMA order, same epilogue:
    the raw output, the statistics rows and the activated side tensor must be BIT-identical, with and without the
    residual / statistics epilogue (the two flavours a train-mode network launches)."""
    from combat_amd._lib import lib
    import ctypes
    st = torch.cuda.current_stream().cuda_stream
    x = nhwc(torch.randn(n, c, hw, hw, generator=g(900)) * 1.5 + 0.3)
    res = nhwc(torch.randn(n, k, hw, hw, generator=g(901)))
    w, pc = make_conv(ops, k, c, 3, 1, 1, 902)
    gamma = dev(torch.rand(c, generator=g(903)) + 0.5) * dev(torch.where(torch.rand(c, generator=g(904)) < 0.2, -1.0, 1.0))
    beta = dev(torch.randn(c, generator=g(905)) * 0.3)
    # ---- the chain: statistics rows of x (as a producing convolution's epilogue leaves them) -> one fused launch
    m = n * hw * hw
    parts = m // 32
    part = torch.zeros(parts, 2, c, device="cuda")
    ops.check(lib.combat_group_stats(x.data_ptr(), parts, 32, c, part.data_ptr(), st), "group_stats")
    mean, rstd, scale, shift = (torch.zeros(c, device="cuda") for _ in range(4))
    scratch = torch.zeros(ops.norm_scratch_bytes(1, c) // 4, device="cuda")
    act = torch.zeros_like(x)
    ops.check(lib.combat_norm_act_fused(x.data_ptr(), part.data_ptr(), 1, parts, m, c, 1e-5, 0.0, gamma.data_ptr(), beta.data_ptr(),
                                        mean.data_ptr(), rstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), None, None, 0.1, None,
                                        scratch.data_ptr(), scratch.numel() * 4, act.data_ptr(), st), "norm_act_fused")
    aff = ops.Affine(scale, shift, 0, True, 0.0)
    for name, kw in (("plain + residual", dict(add_post=res)), ("statistics + residual", dict(add_post=res, stats_kind=1 | 4)),
                     ("statistics", dict(stats_kind=1))):
        outs = []
        for fused in (False, True):
            o = dict(y=torch.zeros(n, hw, hw, k, dtype=bf16, device="cuda"))
            if fused:
                o["side"] = torch.full_like(x, 7.0)
                a = ops.conv_args(x, o["y"], pc, 0, pro=aff, pro_act_dst=o["side"], tile=tile, **kw)
            else:
                o["side"] = act
                a = ops.conv_args(act, o["y"], pc, 0, tile=tile, **kw)
            assert lib.combat_conv_pick_tile(ctypes.byref(a)) == tile, (name, fused)
            if a.stats_kind:
                rows, rpi = ops.conv_stats_layout(a)
                o["stats"] = torch.zeros(rows, 2, k, device="cuda")
                a.stats = o["stats"].data_ptr()
                o["layout"] = (rows, rpi)
            ops.conv_launch(a)
            torch.cuda.synchronize()
            outs.append(o)
        for key in outs[0]:
            if key == "layout":
                assert outs[0][key] == outs[1][key], name
            else:
                assert torch.equal(outs[0][key], outs[1][key]), (name, key, n, hw, c, k, tile)
    # the prologue without the side tensor, leaky slope, automatic tile: against fp32 torch
    y = torch.zeros(n, hw, hw, k, dtype=bf16, device="cuda")
    a = ops.conv_args(x, y, pc, 0, pro=ops.Affine(scale, shift, 0, True, 0.2))
    assert lib.combat_conv_pick_tile(ctypes.byref(a)) in (10, 11)
    ops.conv_launch(a)
    xa = rb(F.leaky_relu(nchw(x) * scale.cpu()[None, :, None, None] + shift.cpu()[None, :, None, None], 0.2))
    assert rel_l2(nchw(y), F.conv2d(xa, rb(w), padding=1)) < 4e-3


@pytest.mark.parametrize("n,hw,c,k,stride,tile,waves", [
    (5, 32, 64, 64, 1, 10, 4), (3, 16, 128, 128, 1, 10, 4), (5, 8, 128, 64, 1, 10, 4), (9, 4, 256, 256, 1, 11, 4),   # ring kernel (ragged N on small maps)
    (3, 32, 64, 64, 1, 14, 8), (7, 8, 128, 64, 1, 14, 8),                   # ... 256-pixel tiles, eight waves
    (3, 32, 64, 64, 1, 17, 0), (40, 32, 64, 64, 1, 17, 0), (81, 32, 64, 64, 1, 17, 0), (128, 32, 64, 64, 1, 17, 0),   # weight-stationary: 1, 1-2, 3, 4 tiles per workgroup
    (5, 32, 64, 128, 2, 12, 4), (3, 16, 128, 256, 2, 13, 4), (3, 10, 64, 64, 2, 12, 4),   # gather kernel (stride 2; 3 x 25 = 75 pixels: ragged last tile)
    (5, 32, 8, 64, 1, 15, 4)])                                               # C = 8 kernel
def test_conv_statistics_one_row_per_workgroup(ops, n, hw, c, k, stride, tile, waves):
    """COMBAT_STATS_PER_WORKGROUP: the same launch with one statistics row per workgroup instead of one per wave --
    outputs unchanged, the layout call reports the smaller row count, and (kernels that meet in LDS: waves > 0) every row
    is the sum of its workgroup's wave rows added in wave order, BIT-exactly; the persistent kernel's rows (one per
    workgroup, over all its tiles) must add up to the same totals.  Forward (sum, second moment) and, at stride 1,
    the train-mode input gradient (sum dz, sum dz * xhat behind the mask)."""
    from combat_amd._lib import lib, STATS_PER_WORKGROUP
    import ctypes
    x = nhwc(torch.randn(n, c, hw, hw, generator=g(301)))
    w, pc = make_conv(ops, k, c, 3, stride, 1, 302)
    p, q = pc.out_hw(hw, hw)
    res = nhwc(torch.randn(n, k, p, q, generator=g(303)))
    cases = [("forward", lambda o, kind: ops.conv_args(x, o, pc, 0, add_post=res, stats_kind=kind, tile=tile), (n, p, q, k))]
    if stride == 1 and c >= 64:
        dy = nhwc(torch.randn(n, k, hw, hw, generator=g(304)))
        xpre = nhwc(torch.randn(n, c, hw, hw, generator=g(305)))
        aff = ops.Affine(dev(torch.rand(c, generator=g(306)) + 0.5), dev(torch.randn(c, generator=g(307)) * 0.3), 0, True, 0.0)
        mean, rstd = dev(torch.randn(c, generator=g(308)) * 0.1), dev(torch.rand(c, generator=g(309)) + 0.5)
        cases.append(("train backward", lambda o, kind: ops.conv_args(dy, o, pc, 1, mask_x=xpre, mask=aff, stats_kind=kind,
                                                                      xh_mean=mean, xh_rstd=rstd, tile=tile), (n, hw, hw, c)))
    for name, make, oshape in cases:
        outs = []
        for extra in (0, STATS_PER_WORKGROUP):
            o = torch.zeros(oshape, dtype=bf16, device="cuda")
            a = make(o, (1 if name == "forward" else 2) | extra)
            assert lib.combat_conv_pick_tile(ctypes.byref(a)) in ((12, 13) if tile in (12, 13) else (tile,)), name   # (the gather kernel sizes its channel tile itself)
            rows, rpi = ops.conv_stats_layout(a)
            stats = torch.full((rows + 1, 2, oshape[3]), 7.0, device="cuda")      # (+ a guard row)
            a.stats = stats.data_ptr()
            ops.conv_launch(a)
            torch.cuda.synchronize()
            assert torch.all(stats[rows] == 7.0), (name, "wrote beyond the reported rows")
            outs.append((o, stats[:rows], rows, rpi))
        (o_w, s_w, rows_w, _), (o_g, s_g, rows_g, rpi_g) = outs
        assert torch.equal(o_w, o_g), name
        assert rows_g < rows_w, (name, rows_g, rows_w)
        if waves:
            pad = rows_g * waves - rows_w             # (a ragged last tile has fewer waves inside the tensor)
            assert 0 <= pad < waves, (name, rows_g, rows_w)
            per = torch.cat([s_w, torch.zeros(pad, 2, s_w.shape[2], device="cuda")]).view(rows_g, waves, 2, -1)
            acc = per[:, 0].clone()
            for wv in range(1, waves):
                acc += per[:, wv]
            assert torch.equal(acc, s_g), name
            if rpi_g:     # rows of one image are contiguous and complete
                ref = (o_g.float().view(n, -1, oshape[3]).sum(1), s_g.view(n, rpi_g, 2, -1)[:, :, 0].sum(1))
                if name == "forward":
                    assert rel_l2(ref[1], ref[0]) < 1e-4
        else:
            tiles = rows_w // 4
            per = -(-tiles // 256)                       # tiles per persistent workgroup
            assert rows_g == -(-tiles // per), (rows_g, rows_w)
            assert rel_l2(s_g.sum(0).double(), s_w.sum(0).double()) < 1e-6, name


@pytest.mark.parametrize("n,hw,c,k,r,stride,tile,keep_raw", [
    (4, 16, 64, 64, 3, 1, 0, True), (4, 16, 64, 128, 3, 1, 0, False),     # DMA-staged kernel
    (4, 16, 64, 64, 3, 1, 17, True), (41, 32, 64, 64, 3, 1, 17, False),   # ... weight-stationary persistent form
    (4, 16, 64, 64, 3, 1, 16, True), (3, 32, 64, 128, 3, 1, 16, False),   # ... with 64-pixel wave tiles
    (4, 16, 64, 64, 3, 1, 8, False), (3, 8, 128, 128, 3, 1, 9, True),     # halo kernels
    (4, 16, 64, 128, 3, 2, 0, False), (4, 16, 8, 64, 3, 1, 0, True), (4, 16, 64, 128, 1, 2, 0, True)])  # gather kernel
def test_conv_activation_output_and_activated_mask(ops, n, hw, c, k, r, stride, tile, keep_raw):
    """Second epilogue output = the next layer's eval BatchNorm + ReLU applied to the stored value
    (bit-identical to what that layer's prologue would compute), with or without the raw tensor; and
    the input-gradient mask taken from such an activated tensor (kept-test x > 0, scale multiplies)."""
    pad = 1 if r == 3 else 0
    x = torch.randn(n, c, hw, hw, generator=g(1))
    w, pc = make_conv(ops, k, c, r, stride, pad, 2)
    p, q = pc.out_hw(hw, hw)
    sc = (torch.rand(k, generator=g(4)) + 0.5) * torch.where(torch.rand(k, generator=g(7)) < 0.2, -1.0, 1.0)
    sh = torch.randn(k, generator=g(5)) * 0.3
    aff = ops.Affine(sc.cuda(), sh.cuda(), 0, True, 0.0)
    y = torch.empty(n, p, q, k, dtype=bf16, device="cuda") if keep_raw else None
    act = torch.empty(n, p, q, k, dtype=bf16, device="cuda")
    ops.conv_launch(ops.conv_args(nhwc(x), y, pc, 0, act_dst=act, act=aff, tile=tile))
    y_ref = rb(F.conv2d(rb(x), rb(w), stride=stride, padding=pad))
    if keep_raw:
        assert rel_l2(nchw(y), y_ref) < 4e-3
        stored = nchw(y)
    else:
        stored = y_ref
    # fused multiply-add (one rounding), as the kernels' prologue / activation epilogue compute it
    fma = (stored.double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)).float()
    act_ref = rb(torch.relu(fma))
    if keep_raw:   # exact: the activation is computed from the stored bf16 value
        assert (nchw(act) != act_ref).sum().item() <= 2
    else:
        assert rel_l2(nchw(act), act_ref) < 6e-3
    if r == 3 and stride == 1 and c >= 64:
        dy = torch.randn(n, k, hw, hw, generator=g(3))
        xact = rb(torch.relu(torch.randn(n, c, hw, hw, generator=g(8))))
        msc = torch.rand(c, generator=g(9)) - 0.3
        maff = ops.Affine(msc.cuda(), torch.zeros(c).cuda(), 0, True, 0.0)
        dx = torch.empty(n, hw, hw, c, dtype=bf16, device="cuda")
        ops.conv_launch(ops.conv_args(nhwc(dy), dx, pc, 1, mask_x=nhwc(xact), mask=maff, mask_mul_scale=True,
                                      mask_activated=True, tile=tile if c == k else 0))
        ref = torch.nn.grad.conv2d_input((n, c, hw, hw), rb(w), rb(dy), padding=1)
        ref = ref * (xact > 0).float() * msc.view(1, -1, 1, 1)
        assert rel_l2(nchw(dx), ref) < 4e-3


@pytest.mark.parametrize("n,hw,c,groups_per_image,tile", [(4, 32, 64, True, 0), (3, 8, 128, True, 0), (8, 4, 128, True, 9),
                                                        (8, 4, 128, False, 0)])
def test_conv3x3_halo_instance_stats_layout(ops, n, hw, c, groups_per_image, tile):
    """Statistics rows of the 3x3 kernels are image-aligned whenever one wave's rows stay inside an
    image (4x4 images: only with the 64-pixel halo tile; the automatic 128-pixel DMA tile puts two
    images into a wave's 32 rows); the per-image sums must then match."""
    x = torch.randn(n, c, hw, hw, generator=g(4))
    w, pc = make_conv(ops, c, c, 3, 1, 1, 5)
    y = torch.empty(n, hw, hw, c, dtype=bf16, device="cuda")
    a = ops.conv_args(nhwc(x), y, pc, 0, stats_kind=1, tile=tile)
    rows, rpi = ops.conv_stats_layout(a)
    assert (rpi > 0) == groups_per_image
    stats = torch.zeros(rows, 2, c, device="cuda")
    a.stats = stats.data_ptr()
    ops.conv_launch(a)
    yr = nchw(y)
    assert rel_l2(stats.sum(0).cpu()[0], yr.sum((0, 2, 3))) < 1e-4
    if rpi:
        per_img = stats[: n * rpi].view(n, rpi, 2, c).sum(1).cpu()
        assert rel_l2(per_img[:, 0], yr.sum((2, 3))) < 1e-4
        assert rel_l2(per_img[:, 1], (yr * yr).sum((2, 3))) < 1e-4


@pytest.mark.parametrize("n,hw,c,k,r,stride,pad,tile", CONV_CASES)
def test_conv_forward_plain(ops, n, hw, c, k, r, stride, pad, tile):
    x = torch.randn(n, c, hw, hw, generator=g(1))
    w, pc = make_conv(ops, k, c, r, stride, pad, 2)
    p, q = pc.out_hw(hw, hw)
    y = torch.empty(n, p, q, k, dtype=bf16, device="cuda")
    ops.conv_launch(ops.conv_args(nhwc(x), y, pc, 0, tile=tile))
    ref = F.conv2d(rb(x), rb(w), stride=stride, padding=pad)
    assert rel_l2(nchw(y), ref) < 4e-3


@pytest.mark.parametrize("n,hw,c,k,r,stride", [
    (3, 12, 64, 128, 3, 2), (2, 20, 128, 64, 3, 2), (5, 6, 64, 64, 3, 2), (3, 12, 64, 128, 1, 2), (2, 10, 64, 64, 3, 1),
    (7, 4, 256, 512, 3, 2), (4, 32, 64, 128, 3, 2), (9, 2, 512, 512, 3, 1)])
def test_gather_dma_odd_shapes_forward_and_dgrad(ops, n, hw, c, k, r, stride):
    """The gather kernel's per-lane row offsets and tap-validity masks are computed once per workgroup (shift
    decode for power-of-two maps, divisions otherwise; parity-class pixel order for stride-2 input gradients;
    split reductions for skinny layers): sizes that are not powers of two, ragged last tiles, 1x1 filters."""
    from combat_amd._lib import lib
    import ctypes
    pad = 1 if r == 3 else 0
    x = torch.randn(n, c, hw, hw, generator=g(301))
    w, pc = make_conv(ops, k, c, r, stride, pad, 302)
    p, q = pc.out_hw(hw, hw)
    ws = torch.empty(32 << 20, dtype=torch.uint8, device="cuda")
    y = torch.empty(n, p, q, k, dtype=bf16, device="cuda")
    a = ops.conv_args(nhwc(x), y, pc, 0, workspace=ws)
    tile = lib.combat_conv_pick_tile(ctypes.byref(a))
    assert tile in (10, 11, 12, 13), tile
    ops.conv_launch(a)
    assert rel_l2(nchw(y), F.conv2d(rb(x), rb(w), stride=stride, padding=pad)) < 4e-3
    dy = torch.randn(n, k, p, q, generator=g(303))
    dx = torch.empty(n, hw, hw, c, dtype=bf16, device="cuda")
    a = ops.conv_args(nhwc(dy), dx, pc, 1, workspace=ws)
    assert lib.combat_conv_pick_tile(ctypes.byref(a)) in (10, 11, 12, 13)
    ops.conv_launch(a)
    ref = torch.nn.grad.conv2d_input((n, c, hw, hw), rb(w), rb(dy), stride=stride, padding=pad)
    assert rel_l2(nchw(dx), ref) < 4e-3


def test_conv_forward_bn_relu_prologue_residual_stats(ops):
    """PreAct conv2: relu(bn(x)) prologue, `out += shortcut` epilogue, next-BN statistics."""
    n, hw, c = 4, 16, 128
    x = torch.randn(n, c, hw, hw, generator=g(3)) * 2 + 0.5
    sc = torch.rand(c, generator=g(4)) + 0.5
    sh = torch.randn(c, generator=g(5)) * 0.3
    res = torch.randn(n, c, hw, hw, generator=g(6))
    w, pc = make_conv(ops, c, c, 3, 1, 1, 7)
    y = torch.empty(n, hw, hw, c, dtype=bf16, device="cuda")
    a = ops.conv_args(nhwc(x), y, pc, 0, pro=ops.Affine(dev(sc), dev(sh), 0, True, 0.0), add_post=nhwc(res),
                      stats_kind=1)
    rows, _ = ops.conv_stats_layout(a)
    stats = torch.zeros(rows, 2, c, device="cuda")
    a.stats = stats.data_ptr()
    ops.conv_launch(a)
    torch.cuda.synchronize()
    act = rb(F.relu(rb(x) * sc[None, :, None, None] + sh[None, :, None, None]))
    ref = F.conv2d(act, rb(w), padding=1) + rb(res)
    assert rel_l2(nchw(y), ref) < 4e-3
    yr = nchw(y)
    s = stats.sum(0).cpu()
    assert rel_l2(s[0], yr.sum((0, 2, 3))) < 1e-4
    assert rel_l2(s[1], (yr * yr).sum((0, 2, 3))) < 1e-4


def test_conv_forward_instnorm_leaky_bias_tanh(ops):
    """UNet: per-(image, channel) affine + LeakyReLU(0.2) prologue, bias, tanh, K=3 padded to 8."""
    n, hw, c = 5, 16, 64
    x = torch.randn(n, c, hw, hw, generator=g(8))
    sc = torch.rand(n, c, generator=g(9)) + 0.5
    sh = torch.randn(n, c, generator=g(10)) * 0.3
    w = torch.randn(3, c, 3, 3, generator=g(11)) * 0.05
    b = torch.randn(3, generator=g(12)) * 0.1
    pc = ops.PackedConv(dev(w).contiguous(memory_format=torch.channels_last), 1, 1, c)
    pc.pack()
    b8 = torch.zeros(8)
    b8[:3] = b
    y = torch.empty(n, hw, hw, 8, dtype=bf16, device="cuda")
    ops.conv_launch(ops.conv_args(nhwc(x), y, pc, 0, pro=ops.Affine(dev(sc), dev(sh), c, True, 0.2), bias=dev(b8),
                                  tanh_out=True))
    act = rb(F.leaky_relu(rb(x) * sc[:, :, None, None] + sh[:, :, None, None], 0.2))
    ref = torch.tanh(F.conv2d(act, rb(w), b, padding=1))
    out = nchw(y)
    assert rel_l2(out[:, :3], ref) < 4e-3
    assert float(out[:, 3:].abs().max()) == 0.0


@pytest.mark.parametrize("n,hw,c", [(5, 16, 64), (3, 32, 64), (2, 32, 128), (2, 64, 64)])
def test_conv_eight_output_channels_kernel(ops, n, hw, c):
    """COMBAT_TILE_K8 (conv_k8.hip): 3x3 stride-1 convolutions into the 8-channel image layout -- the generator's output
    layer (per-(image, channel) InstanceNorm affine + LeakyReLU(0.2) prologue, bias, tanh; also with a per-channel
    prologue and with none) and a stem's input gradient (mode 1 over a 64-channel dY) -- against the generic 16-wide
    tile on the same arguments and against torch; it is the automatic choice for these launches."""
    from combat_amd._lib import lib, TILE_K8
    import ctypes
    x = torch.randn(n, c, hw, hw, generator=g(401))
    w = torch.randn(3, c, 3, 3, generator=g(402)) * 0.05
    b = torch.randn(3, generator=g(403)) * 0.1
    pc = ops.PackedConv(dev(w).contiguous(memory_format=torch.channels_last), 1, 1, c)
    pc.pack()
    b8 = torch.zeros(8)
    b8[:3] = b
    sc_i, sh_i = torch.rand(n, c, generator=g(404)) + 0.5, torch.randn(n, c, generator=g(405)) * 0.3
    sc_c, sh_c = torch.rand(c, generator=g(406)) + 0.5, torch.randn(c, generator=g(407)) * 0.3
    cases = {
        "instance prologue + bias + tanh": (dict(pro=ops.Affine(dev(sc_i), dev(sh_i), c, True, 0.2), bias=dev(b8), tanh_out=True),
                                            lambda: torch.tanh(F.conv2d(rb(F.leaky_relu(rb(x) * sc_i[:, :, None, None] + sh_i[:, :, None, None], 0.2)), rb(w), b, padding=1))),
        "channel prologue, no activation": (dict(pro=ops.Affine(dev(sc_c), dev(sh_c), 0, False, 0.0)),
                                            lambda: F.conv2d(rb(rb(x) * sc_c[None, :, None, None] + sh_c[None, :, None, None]), rb(w), padding=1)),
        "plain + bias": (dict(bias=dev(b8)), lambda: F.conv2d(rb(x), rb(w), b, padding=1)),
    }
    for name, (kw, ref) in cases.items():
        outs = []
        for tile in (TILE_K8, 4, 0):
            y = torch.full((n, hw, hw, 8), 9.0, dtype=bf16, device="cuda")
            a = ops.conv_args(nhwc(x), y, pc, 0, tile=tile, **kw)
            assert lib.combat_conv_pick_tile(ctypes.byref(a)) == (tile or TILE_K8), name
            ops.conv_launch(a)
            outs.append(nchw(y))
        assert rel_l2(outs[0][:, :3], ref()) < 4e-3, name
        assert rel_l2(outs[0], outs[1]) < 2e-3, name
        assert float(outs[0][:, 3:].abs().max()) == 0.0 and torch.equal(outs[0], outs[2]), name
    # a stem's input gradient: dY with c channels -> the 3 (of 8) image channels, taps mirrored
    ws, pcs = make_conv(ops, c, 3, 3, 1, 1, 408, c_pad=8)
    dy = torch.randn(n, c, hw, hw, generator=g(409))
    outs = []
    for tile in (TILE_K8, 4, 0):
        dx = torch.full((n, hw, hw, 8), 9.0, dtype=bf16, device="cuda")
        a = ops.conv_args(nhwc(dy), dx, pcs, 1, tile=tile)
        assert lib.combat_conv_pick_tile(ctypes.byref(a)) == (tile or TILE_K8)
        ops.conv_launch(a)
        outs.append(nchw(dx))
    gin = torch.nn.grad.conv2d_input((n, 3, hw, hw), rb(ws), rb(dy), padding=1)
    assert rel_l2(outs[0][:, :3], gin) < 4e-3 and rel_l2(outs[0], outs[1]) < 2e-3 and torch.equal(outs[0], outs[2])
    assert float(outs[0][:, 3:].abs().max()) == 0.0
    # what it does not cover stays with the general kernels: a residual operand, a ragged tile grid
    y = torch.empty(n, hw, hw, 8, dtype=bf16, device="cuda")
    a = ops.conv_args(nhwc(x), y, pc, 0, add_post=y)
    assert lib.combat_conv_pick_tile(ctypes.byref(a)) != TILE_K8
    a = ops.conv_args(nhwc(x[:, :, :12, :12].contiguous()), torch.empty(n, 12, 12, 8, dtype=bf16, device="cuda"), pc, 0)
    assert lib.combat_conv_pick_tile(ctypes.byref(a)) != TILE_K8


def test_conv_forward_hilo_stem(ops):
    """3-channel image as c8 hi/lo split: the stem sees ~16 mantissa bits of the pixels."""
    n, hw = 4, 32
    x = torch.rand(n, 3, hw, hw, generator=g(13)) * 2 - 1
    w, pc = make_conv(ops, 64, 3, 3, 1, 1, 14, c_pad=8, dup=True)
    xc8 = torch.empty(n, hw, hw, 8, dtype=bf16, device="cuda")
    ops.image_to_c8(dev(x), xc8)
    y = torch.empty(n, hw, hw, 64, dtype=bf16, device="cuda")
    ops.conv_launch(ops.conv_args(xc8, y, pc, 0))
    ref = F.conv2d(x, rb(w), padding=1)  # un-rounded image
    assert rel_l2(nchw(y), ref) < 3e-3
    y2 = torch.empty(n, 16, 16, 64, dtype=bf16, device="cuda")  # stride 2 (UNet conv0_0)
    w2, pc2 = make_conv(ops, 64, 3, 3, 2, 1, 15, c_pad=8, dup=True)
    ops.conv_launch(ops.conv_args(xc8, y2, pc2, 0))
    assert rel_l2(nchw(y2), F.conv2d(x, rb(w2), stride=2, padding=1)) < 3e-3


@pytest.mark.parametrize("n,hw,c,k,r,stride,pad", [
    (4, 16, 64, 64, 3, 1, 1), (2, 32, 64, 128, 3, 2, 1), (2, 32, 64, 128, 1, 2, 0), (8, 4, 512, 512, 3, 1, 1),
    (3, 6, 128, 256, 3, 2, 1), (4, 32, 3, 64, 3, 1, 1), (4, 16, 64, 3, 3, 1, 1),
    (3, 10, 3, 64, 3, 2, 1), (5, 16, 3, 128, 3, 2, 1), (2, 12, 3, 64, 3, 1, 1)])   # 8-channel inputs: all-taps kernel
def test_conv_dgrad_and_wgrad(ops, n, hw, c, k, r, stride, pad):
    c_pad = 8 if c == 3 else c
    x = torch.randn(n, c, hw, hw, generator=g(20))
    w, pc = make_conv(ops, k, c, r, stride, pad, 21, c_pad=c_pad, dup=(c == 3))
    p, q = pc.out_hw(hw, hw)
    dy = torch.randn(n, k, p, q, generator=g(22))
    kc = pc.Kc
    dy_d = torch.zeros(n, p, q, kc, dtype=bf16, device="cuda")
    dy_d[..., :k] = nhwc(dy)
    xr = rb(x).requires_grad_(True)
    wr = rb(w).requires_grad_(True)
    F.conv2d(xr, wr, stride=stride, padding=pad).backward(rb(dy))
    dx = torch.empty(n, hw, hw, c_pad, dtype=bf16, device="cuda")
    ops.conv_launch(ops.conv_args(dy_d, dx, pc, 1))
    assert rel_l2(nchw(dx)[:, :c], xr.grad) < 4e-3
    if c == 3:
        xin = torch.empty(n, hw, hw, 8, dtype=bf16, device="cuda")
        ops.image_to_c8(dev(rb(x)), xin)
    else:
        xin = nhwc(x)
    dw = torch.zeros(k, r * r, c, device="cuda")
    ops.conv_wgrad(xin, dy_d, pc, dw)
    ref = wr.grad.permute(0, 2, 3, 1).reshape(k, r * r, c)
    assert rel_l2(dw, ref) < 2e-3


@pytest.mark.parametrize("n,hw,c,k,r", [(8, 16, 64, 128, 3), (16, 8, 256, 512, 3), (8, 16, 128, 256, 1), (3, 8, 64, 64, 3)])
def test_conv_stride2_dgrad_parity_classes(ops, n, hw, c, k, r):
    """Input gradient of a stride-2 convolution: destination pixels are walked parity-class-major so
    that invalid filter taps are skipped; mask, scale and per-channel statistics in the epilogue must
    not care (the last case's class size is not a multiple of the tile: linear order)."""
    pad = 1 if r == 3 else 0
    w, pc = make_conv(ops, k, c, r, 2, pad, 40)
    p = hw // 2
    dy = torch.randn(n, k, p, p, generator=g(41))
    xpre = torch.randn(n, c, hw, hw, generator=g(42))
    sc = torch.rand(c, generator=g(43)) - 0.3
    sh = torch.randn(c, generator=g(44)) * 0.3
    ref = torch.nn.grad.conv2d_input((n, c, hw, hw), rb(w), rb(dy), stride=2, padding=pad)
    ref = ref * ((rb(xpre) * sc[None, :, None, None] + sh[None, :, None, None]) > 0).float() * sc[None, :, None, None]
    dx = torch.empty(n, hw, hw, c, dtype=bf16, device="cuda")
    a = ops.conv_args(nhwc(dy), dx, pc, 1, mask_x=nhwc(xpre), mask=ops.Affine(dev(sc), dev(sh), 0, True, 0.0),
                      mask_mul_scale=True, stats_kind=1)
    rows, _ = ops.conv_stats_layout(a)
    stats = torch.zeros(rows, 2, c, device="cuda")
    a.stats = stats.data_ptr()
    ops.conv_launch(a)
    assert rel_l2(nchw(dx), ref) < 4e-3
    got = nchw(dx)
    assert rel_l2(stats.sum(0).cpu()[0], got.sum((0, 2, 3))) < 1e-4
    assert rel_l2(stats.sum(0).cpu()[1], (got * got).sum((0, 2, 3))) < 1e-4


@pytest.mark.parametrize("n,hw,c,k", [(8, 32, 64, 128), (4, 16, 128, 256), (16, 8, 256, 512), (3, 32, 64, 128), (5, 10, 64, 64)])
def test_conv_stride2_dgrad_with_shortcut_as_second_source(ops, n, hw, c, k):
    """combat_conv_args.src2: the input gradient of a residual block's 1x1 / stride-2 shortcut rides along in the 3x3 /
    stride-2 input-gradient launch as extra reduction steps on the centre tap's pixels (parity-class-major order: the
    (even, even) class only; linear order -- class size not a multiple of the tile, odd maps -- by the tap's validity
    bits; skinny layers with a split reduction).  Against the two-launch form (shortcut gradient through add_pre: what
    the plans did before) with the same mask + statistics epilogue, and against torch."""
    from combat_amd._lib import lib
    import ctypes
    w3, pc3 = make_conv(ops, k, c, 3, 2, 1, 501)
    w1, pc1 = make_conv(ops, k, c, 1, 2, 0, 502)
    p = (hw + 1) // 2
    dy3, dy1 = torch.randn(n, k, p, p, generator=g(503)), torch.randn(n, k, p, p, generator=g(504))
    xpre = torch.randn(n, c, hw, hw, generator=g(505))
    sc, sh = torch.rand(c, generator=g(506)) - 0.3, torch.randn(c, generator=g(507)) * 0.3
    mean, rstd = dev(torch.randn(c, generator=g(508)) * 0.1), dev(torch.rand(c, generator=g(509)) + 0.5)
    mask = ops.Affine(dev(sc), dev(sh), 0, True, 0.0)
    ws = torch.empty(32 << 20, dtype=torch.uint8, device="cuda")

    def run(fused):
        dx = torch.empty(n, hw, hw, c, dtype=bf16, device="cuda")
        tsc = None
        if not fused:
            tsc = torch.empty(n, hw, hw, c, dtype=bf16, device="cuda")
            ops.conv_launch(ops.conv_args(nhwc(dy1), tsc, pc1, 1))
        a = ops.conv_args(nhwc(dy3), dx, pc3, 1, add_pre=tsc, mask_x=nhwc(xpre), mask=mask, stats_kind=2, xh_mean=mean,
                          xh_rstd=rstd, workspace=ws, shortcut=(nhwc(dy1), pc1) if fused else None)
        assert lib.combat_conv_pick_tile(ctypes.byref(a)) in (12, 13)
        rows, _ = ops.conv_stats_layout(a)
        stats = torch.zeros(rows, 2, c, device="cuda")
        a.stats = stats.data_ptr()
        ops.conv_launch(a)
        torch.cuda.synchronize()
        return nchw(dx), stats.sum(0).cpu()

    if hw % 2:      # (odd maps: the 1x1's output is one pixel larger than the 3x3's would need -- not a shortcut geometry)
        pytest.skip("odd map")
    (dx_f, st_f), (dx_2, st_2) = run(True), run(False)
    gin = (torch.nn.grad.conv2d_input((n, c, hw, hw), rb(w3), rb(dy3), stride=2, padding=1) +
           torch.nn.grad.conv2d_input((n, c, hw, hw), rb(w1), rb(dy1), stride=2, padding=0))
    keep = (rb(xpre) * sc[None, :, None, None] + sh[None, :, None, None]) > 0
    ref = torch.where(keep, gin, torch.zeros_like(gin))
    assert rel_l2(dx_f, ref) < 4e-3
    assert rel_l2(dx_f, dx_2) < 3e-3           # (the two-launch form rounds the shortcut's gradient to bf16 in between)
    assert rel_l2(st_f, st_2) < 3e-3
    assert ops.shortcut_fusable(nhwc(dy3), torch.empty(n, hw, hw, c, dtype=bf16, device="cuda"), pc3, pc1)
    # not a shortcut geometry: stride-1 3x3, or a 3x3 "shortcut"
    w31, pc31 = make_conv(ops, k, c, 3, 1, 1, 510)
    assert not ops.shortcut_fusable(nhwc(torch.randn(n, k, hw, hw)), torch.empty(n, hw, hw, c, dtype=bf16, device="cuda"), pc31, pc1)
    assert not ops.shortcut_fusable(nhwc(dy3), torch.empty(n, hw, hw, c, dtype=bf16, device="cuda"), pc3, pc3)


@pytest.mark.parametrize("n,hw,c,k,r,stride,mode", [(16, 2, 512, 512, 3, 1, 0), (16, 2, 512, 512, 3, 1, 1), (9, 4, 256, 256, 3, 1, 0),
                                                   (32, 8, 256, 512, 3, 2, 0), (10, 4, 512, 128, 1, 1, 0)])
def test_conv_split_reduction(ops, n, hw, c, k, r, stride, mode):
    """Skinny layers (few tiles, long reductions) given a workspace: the reduction steps of a tile are divided
    among several workgroups, slabs combined by a second launch that runs the fused epilogue."""
    from combat_amd._lib import lib
    import ctypes
    pad = 1 if r == 3 else 0
    x = torch.randn(n, c, hw, hw, generator=g(80))
    w, pc = make_conv(ops, k, c, r, stride, pad, 81)
    p = hw // stride
    ws = torch.empty(32 << 20, dtype=torch.uint8, device="cuda")
    if mode == 0:
        res = torch.randn(n, k, p, p, generator=g(82))
        y = torch.empty(n, p, p, k, dtype=bf16, device="cuda")
        a = ops.conv_args(nhwc(x), y, pc, 0, add_post=nhwc(res), stats_kind=1, workspace=ws)
        assert a.workspace and lib.combat_conv_pick_tile(ctypes.byref(a)) in (12, 13)
        rows, _ = ops.conv_stats_layout(a)
        stats = torch.zeros(rows, 2, k, device="cuda")
        a.stats = stats.data_ptr()
        ops.conv_launch(a)
        ref = F.conv2d(rb(x), rb(w), stride=stride, padding=pad) + rb(res)
        assert rel_l2(nchw(y), ref) < 4e-3
        assert rel_l2(stats.sum(0).cpu()[0], nchw(y).sum((0, 2, 3))) < 1e-4
    else:
        dy = torch.randn(n, k, p, p, generator=g(83))
        dx = torch.empty(n, hw, hw, c, dtype=bf16, device="cuda")
        a = ops.conv_args(nhwc(dy), dx, pc, 1, workspace=ws)
        assert a.workspace and lib.combat_conv_pick_tile(ctypes.byref(a)) in (12, 13)
        ops.conv_launch(a)
        assert rel_l2(nchw(dx), torch.nn.grad.conv2d_input((n, c, hw, hw), rb(w), rb(dy), stride=stride, padding=pad)) < 4e-3


def test_conv_gather_dma_ragged_statistics_rows(ops):
    """A last tile that reaches beyond the tensor must not write statistics rows it does not own (the
    array has exactly ceil(M / 32) rows): guard rows after the array stay untouched."""
    n, hw, c, k = 3, 8, 64, 64          # stride-2 forward: M = 3 * 16 = 48 pixels, one 128-row tile
    x = torch.randn(n, c, hw, hw, generator=g(70))
    w, pc = make_conv(ops, k, c, 3, 2, 1, 71)
    y = torch.empty(n, hw // 2, hw // 2, k, dtype=bf16, device="cuda")
    a = ops.conv_args(nhwc(x), y, pc, 0, stats_kind=1)
    from combat_amd._lib import lib
    import ctypes
    assert lib.combat_conv_pick_tile(ctypes.byref(a)) in (12, 13)
    rows, _ = ops.conv_stats_layout(a)
    assert rows == 2
    stats = torch.full((rows + 6, 2, k), 777.0, device="cuda")
    a.stats = stats.data_ptr()
    ops.conv_launch(a)
    assert bool((stats[rows:] == 777.0).all())
    yr = nchw(y)
    assert rel_l2(stats[:rows].sum(0).cpu()[0], yr.sum((0, 2, 3))) < 1e-4
    assert rel_l2(nchw(y), F.conv2d(rb(x), rb(w), stride=2, padding=1)) < 4e-3


def test_conv_dgrad_epilogue_mask_stats(ops):
    """Train-mode BN backward, reduction half fused into dgrad: dz = (dgrad + add_pre) * relu'(bn(x)),
    partial sums of dz and dz*xhat; eval-mode variant multiplies by the BN scale and adds the
    identity-shortcut gradient."""
    n, hw, c = 4, 16, 128
    w, pc = make_conv(ops, c, c, 3, 1, 1, 30)
    dy = torch.randn(n, c, hw, hw, generator=g(31))
    xpre = torch.randn(n, c, hw, hw, generator=g(32))
    extra = torch.randn(n, c, hw, hw, generator=g(33))
    post = torch.randn(n, c, hw, hw, generator=g(34))
    sc = torch.rand(c, generator=g(35)) + 0.5
    sh = torch.randn(c, generator=g(36)) * 0.3
    mean = torch.randn(c, generator=g(37)) * 0.1
    rstd = torch.rand(c, generator=g(38)) + 0.5
    da = torch.nn.grad.conv2d_input((n, c, hw, hw), rb(w), rb(dy), padding=1) + rb(extra)
    msk = ((rb(xpre) * sc[None, :, None, None] + sh[None, :, None, None]) > 0).float()
    dz_ref = da * msk
    dz = torch.empty(n, hw, hw, c, dtype=bf16, device="cuda")
    a = ops.conv_args(nhwc(dy), dz, pc, 1, add_pre=nhwc(extra), mask_x=nhwc(xpre),
                      mask=ops.Affine(dev(sc), dev(sh), 0, True, 0.0), stats_kind=2, xh_mean=dev(mean),
                      xh_rstd=dev(rstd))
    rows, _ = ops.conv_stats_layout(a)
    stats = torch.zeros(rows, 2, c, device="cuda")
    a.stats = stats.data_ptr()
    ops.conv_launch(a)
    assert rel_l2(nchw(dz), dz_ref) < 4e-3
    dzr = nchw(dz)
    xhat = (rb(xpre) - mean[None, :, None, None]) * rstd[None, :, None, None]
    s = stats.sum(0).cpu()
    assert rel_l2(s[0], dzr.sum((0, 2, 3))) < 1e-4
    assert rel_l2(s[1], (dzr * xhat).sum((0, 2, 3))) < 1e-4
    # eval-mode: dx = (dgrad) * mask * scale + post
    dx = torch.empty(n, hw, hw, c, dtype=bf16, device="cuda")
    ops.conv_launch(ops.conv_args(nhwc(dy), dx, pc, 1, mask_x=nhwc(xpre), mask=ops.Affine(dev(sc), dev(sh), 0, True, 0.0),
                                  mask_mul_scale=True, add_post=nhwc(post)))
    ref = torch.nn.grad.conv2d_input((n, c, hw, hw), rb(w), rb(dy), padding=1) * msk * sc[None, :, None, None] + rb(post)
    assert rel_l2(nchw(dx), ref) < 4e-3


def test_wgrad_with_prologue(ops):
    n, hw, c, k = 4, 16, 64, 128
    x = torch.randn(n, c, hw, hw, generator=g(40))
    sc = torch.rand(n, c, generator=g(41)) + 0.5
    sh = torch.randn(n, c, generator=g(42)) * 0.3
    w, pc = make_conv(ops, k, c, 3, 2, 1, 43)
    dy = torch.randn(n, k, 8, 8, generator=g(44))
    act = rb(F.leaky_relu(rb(x) * sc[:, :, None, None] + sh[:, :, None, None], 0.2))
    ref = torch.nn.grad.conv2d_weight(act, (k, c, 3, 3), rb(dy), stride=2, padding=1)
    dw = torch.zeros(k, 9, c, device="cuda")
    ops.conv_wgrad(nhwc(x), nhwc(dy), pc, dw, pro=ops.Affine(dev(sc), dev(sh), c, True, 0.2))
    assert rel_l2(dw, ref.permute(0, 2, 3, 1).reshape(k, 9, c)) < 2e-3


@pytest.mark.parametrize("n,hw,c,k,per_image", [
    (4, 32, 64, 64, False), (3, 16, 128, 64, True), (5, 8, 64, 128, True), (6, 4, 128, 128, False),
    (9, 4, 64, 64, True), (7, 2, 128, 64, True), (33, 2, 64, 64, False)])
def test_wgrad3x3_halo_kernel(ops, n, hw, c, k, per_image):
    """3x3/s1 weight gradient with the input patch in LDS and all nine taps per workgroup: BatchNorm
    (per-channel) and InstanceNorm (per-image) prologues, several images per 64-pixel tile, ragged N;
    also against the generic kernel (split < 0 forces it)."""
    x = torch.randn(n, c, hw, hw, generator=g(45))
    shape = (n, c) if per_image else (c,)
    sc, sh = torch.rand(shape, generator=g(46)) + 0.5, torch.randn(shape, generator=g(47)) * 0.3
    w, pc = make_conv(ops, k, c, 3, 1, 1, 48)
    dy = torch.randn(n, k, hw, hw, generator=g(49))
    bsc = sc[:, :, None, None] if per_image else sc[None, :, None, None]
    bsh = sh[:, :, None, None] if per_image else sh[None, :, None, None]
    act = rb(F.leaky_relu(rb(x) * bsc + bsh, 0.2))
    ref = torch.nn.grad.conv2d_weight(act, (k, c, 3, 3), rb(dy), padding=1).permute(0, 2, 3, 1).reshape(k, 9, c)
    pro = ops.Affine(dev(sc), dev(sh), c if per_image else 0, True, 0.2)
    xd, dyd = nhwc(x), nhwc(dy)
    dw = torch.zeros(k, 9, c, device="cuda")
    ops.conv_wgrad(xd, dyd, pc, dw, pro=pro)
    assert rel_l2(dw, ref) < 2e-3
    dw2 = torch.zeros(k, 9, c, device="cuda")
    ops.conv_wgrad(xd, dyd, pc, dw2, pro=pro, split=-1)
    assert rel_l2(dw2, ref) < 2e-3
    dw3 = torch.zeros(k, 9, c, device="cuda")
    ops.conv_wgrad(xd, dyd, pc, dw3, pro=pro, split=3)      # explicit number of pixel ranges
    assert rel_l2(dw3, ref) < 2e-3


@pytest.mark.parametrize("n,hw,c,k", [(4, 32, 64, 64), (3, 16, 128, 64), (5, 8, 64, 128), (6, 4, 128, 128), (9, 4, 64, 64),
                                      (7, 2, 128, 64), (33, 2, 64, 64), (16, 16, 64, 64)])
def test_wgrad3x3_dma_kernel(ops, n, hw, c, k):
    """3x3/s1 weight gradient of an already-activated input (no prologue): both operands DMA'd into
    LDS.  All tile geometries (16-, 8-, 4- and 2-pixel-wide patches, several images per patch, ragged N),
    automatic and explicit pixel-range split; cross-checked against the generic kernel."""
    x = torch.randn(n, c, hw, hw, generator=g(55))
    w, pc = make_conv(ops, k, c, 3, 1, 1, 56)
    dy = torch.randn(n, k, hw, hw, generator=g(57))
    ref = torch.nn.grad.conv2d_weight(rb(x), (k, c, 3, 3), rb(dy), padding=1).permute(0, 2, 3, 1).reshape(k, 9, c)
    xd, dyd = nhwc(x), nhwc(dy)
    for split, ws in ((0, None), (3, None), (-1, None), (0, True), (5, True)):   # ws: slab stores + reduction launch
        dw = torch.zeros(k, 9, c, device="cuda")
        ops.conv_wgrad(xd, dyd, pc, dw, split=split, workspace=ws)
        assert rel_l2(dw, ref) < 2e-3, (split, ws)


def test_wgrad_reduction_rides_in_the_next_weight_gradient(ops):
    """combat_wgrad_args.reduce_first (round 4): a chain of weight gradients of different layer shapes, each leaving its
    partial-sum slabs (defer_reduce) for the NEXT launch to fold into the earlier dw first; only the last gets a
    reduction launch.  The carried reduction has a fixed summation order and no atomics, so (1) every dw of the chain
    equals the same layer launched alone with its own reduction launch up to summation order, (2) a second chain run
    reproduces the first BIT for bit, (3) a launch that cannot take its predecessor along (generic kernel) reduces it
    first.  (Opt-in in the engines -- COMBAT_REDUCE_BEHIND=1 -- because it measured slower: engine.py.)"""
    import ctypes
    from combat_amd._lib import WgradArgs, lib
    st = torch.cuda.current_stream().cuda_stream
    shapes = [(16, 32, 64, 64), (16, 16, 128, 128), (24, 8, 256, 128), (64, 4, 512, 256), (16, 32, 64, 64)]
    layers = []
    for i, (n, hw, c, k) in enumerate(shapes):
        x = nhwc(torch.randn(n, c, hw, hw, generator=g(300 + i)))
        dy = nhwc(torch.randn(n, k, hw, hw, generator=g(320 + i)))
        w, pc = make_conv(ops, k, c, 3, 1, 1, 340 + i)
        layers.append((x, dy, pc, (k, 9, c), (x.float().cpu(), dy.float().cpu())))

    def args_of(x, dy, pc, dw, ws=None, defer=0, first=None):
        a = WgradArgs()
        a.N, a.H, a.W, a.C = x.shape
        _, a.P, a.Q, a.K = dy.shape
        a.R = a.S = 3
        a.stride, a.pad = 1, 1
        a.src, a.dy, a.dw, a.k_real, a.c_real = x.data_ptr(), dy.data_ptr(), dw.data_ptr(), pc.K, pc.c_real
        if ws is not None:
            a.workspace, a.workspace_bytes, a.defer_reduce = ws.data_ptr(), ws.numel(), defer
        if first is not None:
            a.reduce_first = ctypes.addressof(first)
        return a

    # alone: slabs + the stand-alone (deterministic) reduction launch
    alone = []
    for x, dy, pc, shp, _ in layers:
        dw = torch.zeros(shp, device="cuda")
        probe = args_of(x, dy, pc, dw)
        need = int(lib.combat_conv_wgrad_workspace_bytes(ctypes.byref(probe)))
        assert need > 0
        ws = torch.empty(need, dtype=torch.uint8, device="cuda")
        ops.check(lib.combat_conv_wgrad(ctypes.byref(args_of(x, dy, pc, dw, ws)), st), "wgrad alone")
        alone.append(dw)
    torch.cuda.synchronize()
    for (x, dy, pc, shp, (xf, dyf)), dw in zip(layers, alone):
        n, hw, _, c = x.shape
        ref = torch.nn.grad.conv2d_weight(xf.permute(0, 3, 1, 2), (shp[0], c, 3, 3), dyf.permute(0, 3, 1, 2), padding=1)
        assert rel_l2(dw, ref.permute(0, 2, 3, 1).reshape(shp)) < 2e-3

    def chain():
        regions = [torch.empty(24 << 20, dtype=torch.uint8, device="cuda") for _ in range(2)]
        dws, prev, keep = [], None, []
        for i, (x, dy, pc, shp, _) in enumerate(layers):
            dw = torch.zeros(shp, device="cuda")
            a = args_of(x, dy, pc, dw, regions[i % 2], defer=1, first=prev)
            ops.check(lib.combat_conv_wgrad(ctypes.byref(a), st), "wgrad chain %d" % i)
            dws.append(dw)
            keep.append(a)
            prev = a
        ops.check(lib.combat_conv_wgrad_reduce(ctypes.byref(prev), st), "last reduce")
        torch.cuda.synchronize()
        return dws

    first, second = chain(), chain()
    for i in range(len(layers)):
        assert rel_l2(first[i], alone[i]) < 1e-6, i          # (the stand-alone reduction adds the ranges in another order)
        if i + 1 < len(layers):
            assert torch.equal(first[i], second[i]), i        # carried reduction: fixed order, no atomics: bit-reproducible
        else:     # (the chain's last launch gets the stand-alone reduction: atomics unless COMBAT_WGRAD_DET_REDUCE=1)
            assert rel_l2(first[i], second[i]) < 1e-6
    # a stride-2 launch (generic kernel) behind a slab-leaving one: the predecessor is reduced by a launch of its own
    x, dy, pc, shp, _ = layers[0]
    ws = torch.empty(24 << 20, dtype=torch.uint8, device="cuda")
    dw0 = torch.zeros(shp, device="cuda")
    a0 = args_of(x, dy, pc, dw0, ws, defer=1)
    ops.check(lib.combat_conv_wgrad(ctypes.byref(a0), st), "wgrad")
    w2, pc2 = make_conv(ops, 128, 64, 3, 2, 1, 399)
    dy2 = nhwc(torch.randn(16, 128, 16, 16, generator=g(398)))
    dw2 = torch.zeros(128, 9, 64, device="cuda")
    a2 = WgradArgs()
    a2.N, a2.H, a2.W, a2.C = x.shape
    _, a2.P, a2.Q, a2.K = dy2.shape
    a2.R = a2.S = 3
    a2.stride, a2.pad = 2, 1
    a2.src, a2.dy, a2.dw, a2.k_real, a2.c_real = x.data_ptr(), dy2.data_ptr(), dw2.data_ptr(), 128, 64
    a2.reduce_first = ctypes.addressof(a0)
    ops.check(lib.combat_conv_wgrad(ctypes.byref(a2), st), "wgrad stride 2")
    torch.cuda.synchronize()
    assert rel_l2(dw0, alone[0]) < 1e-6


# ---- the benchmarked shape: every convolution of PreActResNet18 (classifier_models/preact_resnet.py:21,23,27-29,77)
# and of the UnetGenerator (networks/models.py:275-314) at the metric's per-GPU batch N = 128.  At this size the
# dispatcher takes branches the small cases above never reach (DMA-tile thresholds on the workgroup count,
# split reductions below 96 tiles, channel-tile-fastest / pixel-tile-fastest XCD order, 128-workgroup weight-
# gradient sizing with slab workspaces) -- so forward, input gradient and weight gradient are each checked here
# with the AUTOMATIC tile against fp32 torch on the CPU, same bf16-rounded operands.
B128_SHAPES = [
    # hw, c, k, r, stride                         PreActResNet18
    (32, 64, 64, 3, 1), (32, 64, 128, 3, 2), (32, 64, 128, 1, 2), (16, 128, 128, 3, 1), (16, 128, 256, 3, 2),
    (16, 128, 256, 1, 2), (8, 256, 256, 3, 1), (8, 256, 512, 3, 2), (8, 256, 512, 1, 2), (4, 512, 512, 3, 1),
    # UnetGenerator (shapes not already above)
    (16, 64, 64, 3, 1), (16, 64, 128, 3, 2), (8, 128, 128, 3, 1), (8, 128, 256, 3, 2), (4, 256, 256, 3, 1),
    (4, 256, 512, 3, 2), (2, 512, 512, 3, 1), (4, 512, 256, 3, 1), (8, 256, 128, 3, 1), (16, 128, 64, 3, 1),
    # 3-channel ends: classifier stem / UNet conv0_0 (c8 hi/lo images) and the UNet output layer (K = 3 -> 8)
    (32, 3, 64, 3, 1), (32, 3, 64, 3, 2), (32, 64, 3, 3, 1),
]


@pytest.mark.parametrize("hw,c,k,r,stride", B128_SHAPES)
def test_conv_shapes_at_batch_128(ops, hw, c, k, r, stride):
    n, pad = 128, (1 if r == 3 else 0)
    c_pad = 8 if c == 3 else c
    x = torch.randn(n, c, hw, hw, generator=g(700)) if c != 3 else torch.rand(n, 3, hw, hw, generator=g(700)) * 2 - 1
    w, pc = make_conv(ops, k, c, r, stride, pad, 701, c_pad=c_pad, dup=(c == 3))
    p, q = pc.out_hw(hw, hw)
    kc = pc.Kc
    ws = torch.empty(32 << 20, dtype=torch.uint8, device="cuda")
    if c == 3:
        xin = torch.empty(n, hw, hw, 8, dtype=bf16, device="cuda")
        ops.image_to_c8(dev(x), xin)
        x_seen = x                       # hi/lo split: the stem sees the un-rounded pixels (to ~16 bits)
    else:
        xin, x_seen = nhwc(x), rb(x)
    # forward with the statistics epilogue (what a train-mode layer launches)
    y = torch.empty(n, p, q, kc, dtype=bf16, device="cuda")
    a = ops.conv_args(xin, y, pc, 0, stats_kind=1, workspace=ws)
    rows, rpi = ops.conv_stats_layout(a)
    stats = torch.zeros(rows, 2, kc, device="cuda")
    a.stats = stats.data_ptr()
    ops.conv_launch(a)
    ref = F.conv2d(x_seen, rb(w), stride=stride, padding=pad)
    yr = nchw(y)[:, :k]
    assert rel_l2(yr, ref) < 4e-3
    tot = stats.sum(0).cpu()
    assert rel_l2(tot[0, :k], yr.sum((0, 2, 3))) < 1e-4 and rel_l2(tot[1, :k], (yr * yr).sum((0, 2, 3))) < 1e-4
    # input gradient
    dy = torch.randn(n, k, p, q, generator=g(702))
    dy_d = torch.zeros(n, p, q, kc, dtype=bf16, device="cuda")
    dy_d[..., :k] = nhwc(dy)
    dx = torch.empty(n, hw, hw, c_pad, dtype=bf16, device="cuda")
    ops.conv_launch(ops.conv_args(dy_d, dx, pc, 1, workspace=ws))
    dx_ref = torch.nn.grad.conv2d_input((n, c, hw, hw), rb(w), rb(dy), stride=stride, padding=pad)
    assert rel_l2(nchw(dx)[:, :c], dx_ref) < 4e-3
    # weight gradient: atomics path and slab-workspace path (the engines pass a workspace)
    dw_ref = torch.nn.grad.conv2d_weight(x_seen if c != 3 else rb(x), (k, c, r, r), rb(dy), stride=stride, padding=pad)
    dw_ref = dw_ref.permute(0, 2, 3, 1).reshape(k, r * r, c)
    if c == 3:
        ops.image_to_c8(dev(rb(x)), xin)
    for use_ws in (None, True):
        dw = torch.zeros(k, r * r, c, device="cuda")
        ops.conv_wgrad(xin, dy_d, pc, dw, workspace=use_ws)
        assert rel_l2(dw, dw_ref) < 2e-3, use_ws


@pytest.mark.parametrize("per_image,with_affine", [(False, True), (True, True), (False, False)])
def test_affine_act_matches_the_conv_prologue(ops, per_image, with_affine):
    """combat_affine_act materialises what a convolution prologue computes: a prologue-free convolution of
    its output must equal the convolution with the prologue, bit for bit."""
    from combat_amd._lib import lib
    n, hw, c, k = 5, 8, 64, 64
    x = torch.randn(n, c, hw, hw, generator=g(60))
    shape = (n, c) if per_image else (c,)
    sc, sh = torch.rand(shape, generator=g(61)) + 0.5, torch.randn(shape, generator=g(62)) * 0.3
    aff = ops.Affine(dev(sc) if with_affine else None, dev(sh) if with_affine else None, c if per_image else 0, True, 0.2)
    xd = nhwc(x)
    act = torch.empty_like(xd)
    ops.check(lib.combat_affine_act(xd.data_ptr(), n * hw * hw, c, aff.scale.data_ptr() if with_affine else None,
                                    aff.shift.data_ptr() if with_affine else None, hw * hw if per_image else 0, 0.2,
                                    act.data_ptr(), torch.cuda.current_stream().cuda_stream), "combat_affine_act")
    w, pc = make_conv(ops, k, c, 3, 1, 1, 63)
    y_pro = torch.empty(n, hw, hw, k, dtype=bf16, device="cuda")
    y_act = torch.empty(n, hw, hw, k, dtype=bf16, device="cuda")
    ops.conv_launch(ops.conv_args(xd, y_pro, pc, 0, pro=aff, tile=9))
    ops.conv_launch(ops.conv_args(act, y_act, pc, 0, tile=9))
    assert torch.equal(y_pro, y_act)
    bsc = (sc[:, :, None, None] if per_image else sc[None, :, None, None]) if with_affine else 1.0
    bsh = (sh[:, :, None, None] if per_image else sh[None, :, None, None]) if with_affine else 0.0
    assert rel_l2(nchw(act), F.leaky_relu(rb(x) * bsc + bsh, 0.2)) < 4e-3


def test_pack_weights_layouts(ops):
    k, c = 24, 16
    w = torch.randn(k, c, 3, 3, generator=g(50))
    pc = ops.PackedConv(dev(w).contiguous(memory_format=torch.channels_last), 1, 1, c)
    pc.pack()
    wf = pc.wf.float().cpu()
    ref = rb(w).permute(0, 2, 3, 1).reshape(k, 9 * c)
    assert torch.equal(wf[:k, :9 * c], ref)
    assert float(wf[k:].abs().max()) == 0 and float(wf[:, 9 * c:].abs().max()) == 0
    wd = pc.wd.float().cpu()
    refd = rb(w).permute(1, 2, 3, 0).reshape(c, 9 * k)
    assert torch.equal(wd[:c, :9 * k], refd)


# ---------------------------------------------------------------- normalisation


def test_norm_finalize_batchnorm_and_running_stats(ops):
    rows, c, gran = 4096, 64, 32
    x = torch.randn(rows * 1, c, generator=g(60)) * 1.7 + 0.8
    xb = dev(x.to(bf16))
    parts = rows // gran
    partials = torch.empty(parts, 2, c, device="cuda")
    ops.group_stats(xb, parts, gran, partials)
    gamma, beta = torch.rand(c, generator=g(61)) + 0.5, torch.randn(c, generator=g(62))
    rm, rv = dev(torch.zeros(c)), dev(torch.ones(c))
    nbt = torch.zeros((), dtype=torch.int64, device="cuda")
    mean, rstd, scale, shift = (torch.empty(c, device="cuda") for _ in range(4))
    scratch = torch.empty(ops.norm_scratch_bytes(1, c) // 4, device="cuda")
    ops.norm_finalize(partials, 1, parts, c, rows, gamma=dev(gamma), beta=dev(beta), mean=mean, rstd=rstd,
                      scale=scale, shift=shift, running_mean=rm, running_var=rv, nbt=nbt, scratch=scratch)
    xr = rb(x)
    bn = torch.nn.BatchNorm1d(c)
    with torch.no_grad():
        bn.weight.copy_(gamma)
        bn.bias.copy_(beta)
    ref = bn(xr)
    got = xr * scale.cpu() + shift.cpu()
    assert rel_l2(got, ref) < 1e-5
    assert rel_l2(rm, bn.running_mean) < 1e-5 and rel_l2(rv, bn.running_var) < 1e-5
    assert int(nbt) == 1
    assert rel_l2(mean, xr.mean(0)) < 1e-5
    assert rel_l2(rstd, 1 / torch.sqrt(xr.var(0, unbiased=False) + 1e-5)) < 1e-5


def test_instance_norm_small_groups_and_backward(ops):
    """InstanceNorm on 2x2 maps (4 rows per group) forward stats + full backward through
    finalize/apply, against autograd."""
    n, hw2, c = 6, 4, 64
    x = torch.randn(n, hw2, c, generator=g(63)) * 1.3 + 0.2
    dy = torch.randn(n, hw2, c, generator=g(64))
    xb, dyb = dev(x.to(bf16)), dev(dy.to(bf16))
    partials = torch.empty(n, 2, c, device="cuda")
    ops.group_stats(xb, n, hw2, partials)
    mean, rstd, scale, shift = (torch.empty(n, c, device="cuda") for _ in range(4))
    ops.norm_finalize(partials, n, 1, c, hw2, mean=mean, rstd=rstd, scale=scale, shift=shift)
    xr = rb(x).requires_grad_(True)
    y = F.instance_norm(xr.permute(0, 2, 1), eps=1e-5).permute(0, 2, 1)
    assert rel_l2(xr.detach() * scale.cpu()[:, None] + shift.cpu()[:, None], y) < 1e-5
    y.backward(rb(dy))
    ops.group_stats_bwd(dyb, xb, n, hw2, 1, mean, rstd, partials)
    ca, cb, cc = (torch.empty(n, c, device="cuda") for _ in range(3))
    ops.norm_bwd_finalize(partials, n, 1, c, hw2, gamma=None, mean=mean, rstd=rstd, ca=ca, cb=cb, cc=cc)
    dx = torch.empty(n, hw2, c, dtype=bf16, device="cuda")
    ops.norm_bwd_apply(dyb, xb, dx, ca, cb, cc, rows_per_group=hw2)
    assert rel_l2(dx.float(), xr.grad) < 6e-3


def test_batchnorm_backward_coefficients(ops):
    rows, c, gran = 2048, 128, 32
    x = torch.randn(rows, c, generator=g(65)) * 1.5 + 0.3
    dz = torch.randn(rows, c, generator=g(66))
    gamma = torch.rand(c, generator=g(67)) + 0.5
    xb, dzb = dev(x.to(bf16)), dev(dz.to(bf16))
    parts = rows // gran
    partials = torch.empty(parts, 2, c, device="cuda")
    ops.group_stats(xb, parts, gran, partials)
    mean, rstd = torch.empty(c, device="cuda"), torch.empty(c, device="cuda")
    ops.norm_finalize(partials, 1, parts, c, rows, mean=mean, rstd=rstd)
    ops.group_stats_bwd(dzb, xb, parts, gran, 0, mean, rstd, partials)
    ca, cb, cc, dg, db = (torch.empty(c, device="cuda") for _ in range(5))
    ops.norm_bwd_finalize(partials, 1, parts, c, rows, gamma=dev(gamma), mean=mean, rstd=rstd, ca=ca, cb=cb, cc=cc,
                          dgamma=dg, dbeta=db)
    dx = torch.empty(rows, c, dtype=bf16, device="cuda")
    ops.norm_bwd_apply(dzb, xb, dx, ca, cb, cc)
    xr = rb(x).requires_grad_(True)
    gp = gamma.clone().requires_grad_(True)
    bp = torch.zeros(c, requires_grad=True)
    F.batch_norm(xr, None, None, gp, bp, training=True).backward(rb(dz))
    assert rel_l2(dx.float(), xr.grad) < 6e-3
    assert rel_l2(dg, gp.grad) < 1e-4 and rel_l2(db, bp.grad) < 1e-4


def _stream():
    return torch.cuda.current_stream().cuda_stream


@pytest.mark.parametrize("rows,c,gran,groups", [(4096, 64, 32, 1), (2048, 128, 32, 1), (36864, 64, 32, 1),
                                                (6 * 256, 64, 32, 6), (5 * 64, 136, 32, 5)])
def test_norm_act_fused_equals_finalize_plus_activation(ops, rows, c, gran, groups):
    """combat_norm_act_fused = combat_norm_finalize + combat_affine_act in one launch: same published
    statistics (bit for bit: same rows, same fp64 finalisation) and the same activation tensor, for
    BatchNorm (one group, incl. > 256 partial rows -> stage 1, running statistics) and InstanceNorm."""
    from combat_amd._lib import lib
    x = torch.randn(rows, c, generator=g(160)) * 1.7 + 0.8
    xb = dev(x.to(bf16))
    parts = rows // gran
    partials = torch.empty(parts, 2, c, device="cuda")
    ops.group_stats(xb, parts, gran, partials)
    bn = groups == 1
    gamma = dev(torch.rand(c, generator=g(161)) + 0.5) if bn else None
    beta = dev(torch.randn(c, generator=g(162))) if bn else None
    scratch = torch.empty(ops.norm_scratch_bytes(groups, c) // 4, device="cuda")
    res = []
    for fused in (False, True):
        rm, rv = (dev(torch.zeros(c)), dev(torch.ones(c))) if bn else (None, None)
        nbt = torch.zeros((), dtype=torch.int64, device="cuda") if bn else None
        mean, rstd, scale, shift = (torch.empty(groups, c, device="cuda") for _ in range(4))
        act = torch.empty_like(xb)
        if fused:
            ops.check(lib.combat_norm_act_fused(
                xb.data_ptr(), partials.data_ptr(), groups, parts // groups, rows // groups, c, 1e-5, 0.2,
                gamma.data_ptr() if bn else None, beta.data_ptr() if bn else None, mean.data_ptr(), rstd.data_ptr(),
                scale.data_ptr(), shift.data_ptr(), rm.data_ptr() if bn else None, rv.data_ptr() if bn else None, 0.1,
                nbt.data_ptr() if bn else None, scratch.data_ptr(), scratch.numel() * 4, act.data_ptr(), _stream()),
                "combat_norm_act_fused")
        else:
            ops.norm_finalize(partials, groups, parts // groups, c, rows // groups, gamma=gamma, beta=beta, mean=mean,
                              rstd=rstd, scale=scale, shift=shift, running_mean=rm, running_var=rv, nbt=nbt,
                              scratch=scratch)
            ops.check(lib.combat_affine_act(xb.data_ptr(), rows, c, scale.data_ptr(), shift.data_ptr(),
                                            rows // groups if groups > 1 else 0, 0.2, act.data_ptr(), _stream()),
                      "combat_affine_act")
        res.append((mean, rstd, scale, shift, act, rm, rv, nbt))
    for a, b in zip(*res):
        if a is not None:
            assert torch.equal(a, b)
    xr = rb(x).view(groups, rows // groups, c)
    mu, var = xr.mean(1, keepdim=True), xr.var(1, unbiased=False, keepdim=True)
    ref = (xr - mu) / torch.sqrt(var + 1e-5)
    if bn:
        ref = ref * gamma.cpu() + beta.cpu()
    assert rel_l2(res[1][4].float().view(groups, -1, c), F.leaky_relu(ref, 0.2)) < 4e-3


@pytest.mark.parametrize("n,px,c", [(6, 4, 64), (5, 16, 256), (3, 256, 64), (2, 1024, 72)])
def test_instance_norm_fused_from_the_tensors(ops, n, px, c):
    """No partial rows: both fused kernels take their sums from the tensors (small InstanceNorm groups --
    replaces group_stats(+_bwd) + finalize + apply).  Forward against F.instance_norm, backward against
    autograd, with a residual term."""
    from combat_amd._lib import lib
    x = torch.randn(n, px, c, generator=g(163)) * 1.3 + 0.2
    dy, add = torch.randn(n, px, c, generator=g(164)), torch.randn(n, px, c, generator=g(165))
    xb, dyb, addb = dev(x.to(bf16)), dev(dy.to(bf16)), dev(add.to(bf16))
    mean, rstd, scale, shift = (torch.empty(n, c, device="cuda") for _ in range(4))
    act = torch.empty_like(xb)
    ops.check(lib.combat_norm_act_fused(xb.data_ptr(), None, n, 0, px, c, 1e-5, 0.2, None, None, mean.data_ptr(),
                                        rstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), None, None, 0.1, None, None,
                                        0, act.data_ptr(), _stream()), "combat_norm_act_fused")
    xr = rb(x).requires_grad_(True)
    y = F.instance_norm(xr.permute(0, 2, 1), eps=1e-5).permute(0, 2, 1)
    assert rel_l2(act.float(), F.leaky_relu(y, 0.2)) < 4e-3
    assert rel_l2(mean, xr.detach().mean(1)) < 1e-5
    assert rel_l2(xr.detach() * scale.cpu()[:, None] + shift.cpu()[:, None], y) < 1e-5
    y.backward(rb(dy))
    dx = torch.empty_like(xb)
    ops.check(lib.combat_norm_bwd_fused(dyb.data_ptr(), xb.data_ptr(), addb.data_ptr(), None, n, 0, px, c, None,
                                        mean.data_ptr(), rstd.data_ptr(), None, None, None, 0, dx.data_ptr(), _stream()),
              "combat_norm_bwd_fused")
    assert rel_l2(dx.float(), xr.grad + rb(add)) < 6e-3
    # argument checks: more pixels per group than the direct form reduces
    assert lib.combat_norm_act_fused(xb.data_ptr(), None, 1, 0, 2048, c, 1e-5, 0.2, None, None, mean.data_ptr(),
                                     rstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), None, None, 0.1, None, None,
                                     0, act.data_ptr(), _stream()) == -1


@pytest.mark.parametrize("rows,c", [(2048, 128), (16384, 64)])
def test_batchnorm_backward_fused(ops, rows, c):
    """combat_norm_bwd_fused from partial rows = combat_norm_bwd_finalize + combat_norm_bwd_apply, bit for bit,
    and against autograd (dx, dgamma, dbeta)."""
    from combat_amd._lib import lib
    gran = 32
    x = torch.randn(rows, c, generator=g(166)) * 1.5 + 0.3
    dz = torch.randn(rows, c, generator=g(167))
    gamma = torch.rand(c, generator=g(168)) + 0.5
    xb, dzb = dev(x.to(bf16)), dev(dz.to(bf16))
    parts = rows // gran
    partials = torch.empty(parts, 2, c, device="cuda")
    ops.group_stats(xb, parts, gran, partials)
    mean, rstd = torch.empty(c, device="cuda"), torch.empty(c, device="cuda")
    scratch = torch.empty(ops.norm_scratch_bytes(1, c) // 4, device="cuda")
    ops.norm_finalize(partials, 1, parts, c, rows, mean=mean, rstd=rstd, scratch=scratch)
    ops.group_stats_bwd(dzb, xb, parts, gran, 0, mean, rstd, partials)
    ca, cb, cc, dg, db, dg2, db2 = (torch.empty(c, device="cuda") for _ in range(7))
    ops.norm_bwd_finalize(partials, 1, parts, c, rows, gamma=dev(gamma), mean=mean, rstd=rstd, ca=ca, cb=cb, cc=cc,
                          dgamma=dg, dbeta=db, scratch=scratch)
    dx, dx2 = torch.empty(rows, c, dtype=bf16, device="cuda"), torch.empty(rows, c, dtype=bf16, device="cuda")
    ops.norm_bwd_apply(dzb, xb, dx, ca, cb, cc)
    gm = dev(gamma)
    ops.check(lib.combat_norm_bwd_fused(dzb.data_ptr(), xb.data_ptr(), None, partials.data_ptr(), 1, parts, rows, c,
                                        gm.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dg2.data_ptr(), db2.data_ptr(),
                                        scratch.data_ptr(), scratch.numel() * 4, dx2.data_ptr(), _stream()),
              "combat_norm_bwd_fused")
    assert torch.equal(dx, dx2) and torch.equal(dg, dg2) and torch.equal(db, db2)
    xr = rb(x).requires_grad_(True)
    gp = gamma.clone().requires_grad_(True)
    bp = torch.zeros(c, requires_grad=True)
    F.batch_norm(xr, None, None, gp, bp, training=True).backward(rb(dz))
    assert rel_l2(dx2.float(), xr.grad) < 6e-3
    assert rel_l2(dg2, gp.grad) < 1e-4 and rel_l2(db2, bp.grad) < 1e-4


def test_bn_eval_fold(ops):
    c = 96
    gm, bt = torch.rand(c, generator=g(68)) + 0.5, torch.randn(c, generator=g(69))
    rm, rv = torch.randn(c, generator=g(70)), torch.rand(c, generator=g(71)) + 0.2
    sc, sh = torch.empty(c, device="cuda"), torch.empty(c, device="cuda")
    ops.bn_eval_fold(dev(gm), dev(bt), dev(rm), dev(rv), sc, sh)
    x = torch.randn(5, c, generator=g(72))
    assert rel_l2(x * sc.cpu() + sh.cpu(), F.batch_norm(x, rm, rv, gm, bt, training=False)) < 1e-6


def test_copy3_one_launch_for_three_copies(ops):
    """combat_copy3: device and pinned-host sources (the latter read through the device mapping), unaligned pointers and
    lengths, skipped entries, an unpinned host source (falls back to an ordinary copy)."""
    from combat_amd._lib import lib
    st = torch.cuda.current_stream().cuda_stream
    a = torch.randn(300_001, generator=g(601))                       # 1.2 MB, length not a multiple of 16 bytes
    a_dev, a_pin, a_host = a.cuda(), a.clone().pin_memory(), a.clone()
    small = torch.arange(3, dtype=torch.float32).cuda()
    raw = torch.randint(0, 255, (4099,), dtype=torch.uint8, generator=g(602)).pin_memory()
    d0, d1, d2 = torch.zeros_like(a_dev), torch.zeros(8, device="cuda"), torch.zeros(4099, dtype=torch.uint8, device="cuda")
    ops.check(lib.combat_copy3(d0.data_ptr(), a_pin.data_ptr(), a.numel() * 4, d1.data_ptr(), small.data_ptr(), 12,
                               d2.data_ptr(), raw.data_ptr(), 4099, st), "copy3")
    torch.cuda.synchronize()
    assert torch.equal(d0.cpu(), a) and torch.equal(d1.cpu(), torch.tensor([0., 1., 2., 0., 0., 0., 0., 0.])) and torch.equal(d2.cpu(), raw)
    # unaligned device views, a skipped entry, an unpinned host source
    e0, e2 = torch.zeros(300_001, device="cuda"), torch.zeros_like(a_dev)
    ops.check(lib.combat_copy3(e0[1:].data_ptr(), a_dev[3:].data_ptr(), (a.numel() - 3) * 4, None, None, 0,
                               e2.data_ptr(), a_host.data_ptr(), a.numel() * 4, st), "copy3")
    torch.cuda.synchronize()
    assert torch.equal(e0[1:-2].cpu(), a[3:]) and float(e0[0]) == 0.0 and torch.equal(e2.cpu(), a)
    assert lib.combat_copy3(None, a_dev.data_ptr(), 16, None, None, 0, None, None, 0, st) == -1     # bytes without a destination
    assert lib.combat_copy3(None, None, 0, None, None, 0, None, None, 0, st) == 0                   # nothing to do


# ---------------------------------------------------------------- UNet glue


@pytest.mark.parametrize("n,hw,c", [(3, 8, 64), (2, 16, 128), (2, 6, 64), (2, 4, 512)])
def test_unet_up_forward_backward(ops, n, hw, c):
    y = torch.randn(n, c, hw, hw, generator=g(80))
    s = torch.randn(n, c, hw, hw, generator=g(81))
    sy, ty = torch.rand(n, c, generator=g(82)) + 0.5, torch.randn(n, c, generator=g(83)) * 0.2
    ss, ts = torch.rand(n, c, generator=g(84)) + 0.5, torch.randn(n, c, generator=g(85)) * 0.2
    out = torch.empty(n, 2 * hw, 2 * hw, c, dtype=bf16, device="cuda")
    ops.unet_up_fwd(nhwc(y), dev(sy), dev(ty), out, nhwc(s), dev(ss), dev(ts))
    u = (rb(y) * sy[:, :, None, None] + ty[:, :, None, None]
         + F.leaky_relu(rb(s) * ss[:, :, None, None] + ts[:, :, None, None], 0.2)).requires_grad_(True)
    ref = F.leaky_relu(F.interpolate(u, scale_factor=2, mode="bilinear", align_corners=False), 0.2)
    assert rel_l2(nchw(out), ref) < 4e-3
    out2 = torch.empty_like(out)
    ops.unet_up_fwd(nhwc(y), dev(sy), dev(ty), out2)  # no skip
    ref2 = F.leaky_relu(F.interpolate(rb(y) * sy[:, :, None, None] + ty[:, :, None, None], scale_factor=2,
                                      mode="bilinear", align_corners=False), 0.2)
    assert rel_l2(nchw(out2), ref2) < 4e-3
    # backward uses the sign of the *stored* output
    d_out = torch.randn(n, c, 2 * hw, 2 * hw, generator=g(86))
    outr = nchw(out)
    du = torch.empty(n, hw, hw, c, dtype=bf16, device="cuda")
    ops.unet_up_bwd(nhwc(d_out), out, du)
    gmask = torch.where(outr > 0, torch.ones_like(outr), torch.full_like(outr, 0.2)) * rb(d_out)
    uu = torch.zeros(n, c, hw, hw, requires_grad=True)
    F.interpolate(uu, scale_factor=2, mode="bilinear", align_corners=False).backward(gmask)
    assert rel_l2(nchw(du), uu.grad) < 4e-3


@pytest.mark.parametrize("n,hw,c,rows", [(5, 2, 512, 0), (4, 4, 256, 0), (3, 8, 128, 2), (3, 16, 64, 8), (2, 8, 72, 0)])
def test_unet_up_fused_equals_finalize_plus_upsample(ops, n, hw, c, rows):
    """combat_unet_up_fused = (group statistics ->) combat_norm_finalize -> combat_unet_up_fwd in one launch:
    same published InstanceNorm statistics and the same upsampled tensor (bit for bit with partial rows; with
    the sums taken from the tensor, against torch), with and without the skip operand."""
    from combat_amd._lib import lib
    y = torch.randn(n, hw, hw, c, generator=g(180)) * 1.4 + 0.3
    sk = torch.randn(n, hw, hw, c, generator=g(181))
    yb, skb = dev(y.to(bf16)), dev(sk.to(bf16))
    ss, ts = dev(torch.rand(n, c, generator=g(182)) + 0.5), dev(torch.randn(n, c, generator=g(183)) * 0.2)
    pq = hw * hw
    mean, rstd, scale, shift = (torch.empty(n, c, device="cuda") for _ in range(4))
    for skip in (True, False):
        part = None
        if rows:
            part = torch.empty(n * rows, 2, c, device="cuda")
            ops.group_stats(yb.view(n * pq, c), n * rows, pq // rows, part)
        out = torch.empty(n, 2 * hw, 2 * hw, c, dtype=bf16, device="cuda")
        ops.check(lib.combat_unet_up_fused(yb.data_ptr(), part.data_ptr() if rows else None, rows, skb.data_ptr() if skip else None,
                                           ss.data_ptr() if skip else None, ts.data_ptr() if skip else None, n, hw, hw, c, 1e-5,
                                           mean.data_ptr(), rstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), out.data_ptr(),
                                           torch.cuda.current_stream().cuda_stream), "combat_unet_up_fused")
        yr = rb(y).permute(0, 3, 1, 2)
        u = F.instance_norm(yr, eps=1e-5)
        if skip:
            u = u + F.leaky_relu(rb(sk).permute(0, 3, 1, 2) * ss.cpu()[:, :, None, None] + ts.cpu()[:, :, None, None], 0.2)
        ref = F.leaky_relu(F.interpolate(u, scale_factor=2, mode="bilinear", align_corners=False), 0.2)
        assert rel_l2(nchw(out), ref) < 4e-3
        assert rel_l2(mean, yr.mean((2, 3))) < 1e-5
        if rows:   # the two-launch chain on the same partial rows
            m2, r2, sc2, sh2 = (torch.empty(n, c, device="cuda") for _ in range(4))
            ops.norm_finalize(part, n, rows, c, pq, mean=m2, rstd=r2, scale=sc2, shift=sh2)
            out2 = torch.empty_like(out)
            ops.unet_up_fwd(yb, sc2, sh2, out2, *((skb, ss, ts) if skip else ()))
            assert torch.equal(out, out2) and torch.equal(scale, sc2) and torch.equal(shift, sh2) and torch.equal(rstd, r2)


@pytest.mark.parametrize("n,hw,c", [(5, 2, 512), (4, 4, 256), (3, 8, 128), (3, 16, 64), (2, 8, 72)])
def test_unet_up_bwd_fused_equals_two_launches(ops, n, hw, c):
    """combat_unet_up_bwd_fused = combat_unet_up_bwd + combat_norm_bwd_fused (sums from the tensors), bit for bit."""
    from combat_amd._lib import lib
    d_out = dev(torch.randn(n, 2 * hw, 2 * hw, c, generator=g(190)).to(bf16))
    out = dev(torch.randn(n, 2 * hw, 2 * hw, c, generator=g(191)).to(bf16))
    y = dev((torch.randn(n, hw, hw, c, generator=g(192)) * 1.3 + 0.2).to(bf16))
    mean, rstd = dev(torch.randn(n, c, generator=g(193)) * 0.2), dev(torch.rand(n, c, generator=g(194)) + 0.5)
    st = torch.cuda.current_stream().cuda_stream
    du, dx, du2, dx2 = (torch.empty(n, hw, hw, c, dtype=bf16, device="cuda") for _ in range(4))
    ops.check(lib.combat_unet_up_bwd_fused(d_out.data_ptr(), out.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), n,
                                           hw, hw, c, du.data_ptr(), dx.data_ptr(), st), "combat_unet_up_bwd_fused")
    ops.unet_up_bwd(d_out, out, du2)
    ops.check(lib.combat_norm_bwd_fused(du2.data_ptr(), y.data_ptr(), None, None, n, 0, hw * hw, c, None, mean.data_ptr(),
                                        rstd.data_ptr(), None, None, None, 0, dx2.data_ptr(), st), "combat_norm_bwd_fused")
    assert torch.equal(du, du2)
    assert rel_l2(dx.float(), dx2.float()) < 1e-3 and float((dx.float() - dx2.float()).abs().max()) <= 2 * float(dx2.float().abs().max()) * 2 ** -8


# ---------------------------------------------------------------- trigger / augmentation / DCT


@pytest.mark.parametrize("hw,sigma", [(32, 0.35), (32, 0.9), (64, 0.6)])
def test_trigger_forward_backward(ops, hw, sigma):
    from oracle import combat_oracle as O
    n = 5
    x = ((torch.randint(0, 256, (n, 3, hw, hw), generator=g(90)).float() / 255) - 0.5) / 0.5
    noise = torch.tanh(torch.randn(n, 3, hw, hw, generator=g(91)) * 3)
    n8 = torch.zeros(n, hw, hw, 8, dtype=bf16, device="cuda")
    n8[..., :3] = nhwc(noise)
    pm = dev(O.lowpass_matrix(hw, 0.65))
    k1 = dev(O.gaussian_kernel1d(sigma))
    out = torch.empty(n, 3, hw, hw, device="cuda")
    out8 = torch.empty(n, hw, hw, 8, dtype=bf16, device="cuda")
    mse = torch.empty(3 * n, device="cuda")   # per (image, channel)
    ops.trigger_fwd(dev(x), n8, pm, k1, 0.08, out, out8, mse)
    nz = rb(noise).requires_grad_(True)
    ref = O.trigger_mix(x, nz, 0.08, 0.65, sigma)
    assert float((out.cpu() - ref).abs().max()) < 2e-5
    assert rel_l2(mse.view(n, 3), ((ref - x) ** 2).sum((2, 3))) < 1e-4
    hi, lo = out8[..., :3].float().cpu(), out8[..., 3:6].float().cpu()
    assert float(((hi + lo).permute(0, 3, 1, 2) - ref).abs().max()) < 3e-5
    # the poisoned sub-batch: output image i from row src_index[i] of the images and of the generator output
    idx = torch.tensor([3, 0, 4], dtype=torch.int32)
    sub = torch.empty(3, 3, hw, hw, device="cuda")
    ops.trigger_fwd(dev(x), n8, pm, k1, 0.08, sub, src_index=dev(idx))
    assert torch.equal(sub, out[idx.long().cuda()])
    d_out = torch.randn(n, 3, hw, hw, generator=g(92))
    l2s = 0.02 / ref.numel()
    (ref * d_out).sum().add(l2s * ((ref - x) ** 2).sum()).backward()
    dn = torch.empty(n, hw, hw, 8, dtype=bf16, device="cuda")
    ops.trigger_bwd(dev(x), n8, pm, k1, 0.08, dev(d_out * 0.25), out, l2s, dn, d_out2=dev(d_out * 0.75))   # two shares of one gradient
    assert rel_l2(nchw(dn)[:, :3], nz.grad) < 4e-3


@pytest.mark.parametrize("hw", [32, 112])      # 112 > 96: the large-image kernels (global gather / fp32 atomics)
def test_augment_forward_backward(ops, hw):
    from oracle import combat_oracle as O
    n = 8
    x = torch.rand(n, 3, hw, hw, generator=g(95)) * 2 - 1
    p = O.AugParams(np.array([5, 0, 10, 3, 5, 7, 5, 2], np.int32), np.array([5, 10, 0, 8, 5, 5, 1, 9], np.int32),
                    np.array([0, 0, 7.5, -9.0, 3.0, 0, -4.0, 10.0], np.float32),
                    np.array([0, 1, 0, 1, 1, 0, 0, 1], np.int32))
    perm = torch.tensor([3, 1, 0, 2, 7, 6, 5, 4], dtype=torch.int32)
    par = torch.tensor(np.stack([p.crop_dx - 5, p.crop_dy - 5, np.radians(p.angle_deg), p.flip], 1).astype(np.float32))
    out8 = torch.empty(n, hw, hw, 8, dtype=bf16, device="cuda")
    outf = torch.empty(n, 3, hw, hw, device="cuda")
    ops.augment_fwd(dev(x), n, hw, out8, dev(par), dev(perm), outf)
    xr = x[perm.long()].clone().requires_grad_(True)
    ref = O.post_tensor_transform(xr, p)
    assert float((outf.cpu() - ref).abs().max()) < 2e-5
    d = torch.randn(n, 3, hw, hw, generator=g(96))
    ref.backward(rb(d))
    d8 = torch.zeros(n, hw, hw, 8, dtype=bf16, device="cuda")
    d8[..., :3] = nhwc(d)
    dx = torch.empty(n, 3, hw, hw, device="cuda")
    ops.augment_bwd(d8, n, hw, dx, dev(par))
    tol = 2e-5 * hw / 32     # bilinear weights are differences of fp32 coordinates up to hw: their ulp grows with hw
    assert float((dx.cpu() - xr.grad).abs().max()) < tol
    ops.augment_bwd(d8, n, hw, dx, dev(par), accumulate=True)
    assert float((dx.cpu() - 2 * xr.grad).abs().max()) < 2 * tol
    # identity (post_transform_option no_use): exact copy, hi + lo == x to ~2^-16
    ops.augment_fwd(dev(x), n, hw, out8, None, None, outf)
    assert torch.equal(outf.cpu(), x)
    assert float(((out8[..., :3].float() + out8[..., 3:6].float()).cpu().permute(0, 3, 1, 2) - x).abs().max()) < 4e-5


@pytest.mark.parametrize("n,hw", [(4, 32), (2, 224)])       # 224 > 64: the strip kernel
def test_dct_u8(ops, n, hw):
    from oracle import combat_oracle as O
    x = ((torch.randint(0, 256, (n, 3, hw, hw), generator=g(97)).float() / 255) - 0.5) / 0.5 * 0.999
    out8 = torch.empty(n, hw, hw, 8, dtype=bf16, device="cuda")
    ops.dct_u8(dev(x), dev(O.dct_matrix(hw)), out8)
    ref = O.frequency_input(x)
    got = (out8[..., :3].float() + out8[..., 3:6].float()).cpu().permute(0, 3, 1, 2)
    assert rel_l2(got, ref) < 1e-4


# ---------------------------------------------------------------- head / SGD / misc


@pytest.mark.parametrize("hw,classes", [(4, 10), (8, 8)])
def test_head_forward_backward(ops, hw, classes):
    n, c = 6, 512
    feat = torch.randn(n, c, hw, hw, generator=g(100))
    ph = hw // 4
    w = torch.randn(classes, c * ph * ph, generator=g(101)) * 0.05
    b = torch.randn(classes, generator=g(102)) * 0.1
    t = torch.randint(0, classes, (n,), generator=g(103))
    fb = nhwc(feat)
    logits = torch.empty(n, classes, device="cuda")
    pooled = torch.empty(n, c * ph * ph, device="cuda")
    loss = torch.zeros(1, device="cuda")
    correct = torch.zeros(1, dtype=torch.int32, device="cuda")
    ops.head_fwd(fb, dev(w), dev(b), logits, targets=dev(t), loss_weight=0.8, pooled=pooled, loss_sum=loss,
                 correct=correct)
    fr = rb(feat).requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    lg = F.linear(F.avg_pool2d(fr, 4).flatten(1), wr, br)
    ls = 0.8 * F.cross_entropy(lg, t)
    ls.backward()
    assert rel_l2(logits, lg) < 1e-5
    assert abs(float(loss) - float(ls)) < 1e-5
    assert int(correct) == int((lg.argmax(1) == t).sum())
    dl = torch.empty(n, classes, device="cuda")
    dfeat = torch.empty(n, hw, hw, c, dtype=bf16, device="cuda")
    dw, db = torch.zeros_like(dev(w)), torch.zeros(classes, device="cuda")   # accumulated into
    ops.head_bwd(pooled, n, hw, c, dev(w), logits, dev(t), 0.8, dl, dfeat, dw, db)
    assert rel_l2(nchw(dfeat), fr.grad) < 4e-3
    assert rel_l2(dw, wr.grad) < 1e-5 and rel_l2(db, br.grad) < 1e-5
    # one launch for the forward + the feature gradient (combat_head_fwd_bwd), the weight half on its own
    # (combat_head_bwd_weights): the same values as the two calls above, bit for bit where nothing is an atomic sum
    from combat_amd._lib import lib
    st = torch.cuda.current_stream().cuda_stream
    logits2, pooled2, dl2, dfeat2 = torch.empty_like(logits), torch.empty_like(pooled), torch.empty_like(dl), torch.empty_like(dfeat)
    loss2, correct2 = torch.zeros(1, device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda")
    wd, bd, td = dev(w), dev(b), dev(t)
    ops.check(lib.combat_head_fwd_bwd(fb.data_ptr(), n, hw, c, wd.data_ptr(), bd.data_ptr(), classes, td.data_ptr(), 0.8,
                                      pooled2.data_ptr(), logits2.data_ptr(), loss2.data_ptr(), correct2.data_ptr(), None, None,
                                      dl2.data_ptr(), dfeat2.data_ptr(), st), "combat_head_fwd_bwd")
    dw2, db2 = torch.zeros_like(dw), torch.zeros_like(db)
    ops.check(lib.combat_head_bwd_weights(dl2.data_ptr(), pooled2.data_ptr(), n, hw, c, classes, dw2.data_ptr(), db2.data_ptr(), st),
              "combat_head_bwd_weights")
    torch.cuda.synchronize()
    for u, v in ((logits2, logits), (pooled2, pooled), (dl2, dl), (dfeat2, dfeat)):
        assert torch.equal(u, v)
    assert int(correct2) == int(correct) and abs(float(loss2) - float(loss)) < 1e-6
    assert rel_l2(dw2, dw) < 1e-6 and rel_l2(db2, db) < 1e-6


def test_sgd_nesterov_multi_tensor(ops):
    from oracle import combat_oracle as O
    shapes = [(64, 3, 3, 3), (64,), (128, 64, 3, 3), (10, 512)]
    ps = [torch.randn(s, generator=g(110 + i)) for i, s in enumerate(shapes)]
    dps = [dev(p.clone()) for p in ps]
    dbufs = [torch.zeros_like(p) for p in dps]
    dgr = [torch.empty_like(p) for p in dps]
    table = torch.tensor([[p.data_ptr(), gr.data_ptr(), bf.data_ptr()] for p, gr, bf in zip(dps, dgr, dbufs)],
                         dtype=torch.int64, device="cuda")
    sizes = torch.tensor([p.numel() for p in dps], dtype=torch.int64, device="cuda")
    bufs = [None] * len(ps)
    for step in range(3):
        gs = [torch.randn(s, generator=g(120 + 10 * step + i)) for i, s in enumerate(shapes)]
        for d, s in zip(dgr, gs):
            d.copy_(s * 4)
        ops.sgd_nesterov(table, sizes, len(ps), max(p.numel() for p in ps), 1e-2, 0.9, 5e-4, 0.25, step == 0)
        O.sgd_nesterov_step(ps, gs, bufs, 1e-2)
    for a, b in zip(dps, ps):
        assert rel_l2(a, b) < 1e-6


def test_small_kernels(ops):
    n, c, hw = 3, 32, 8
    x = torch.randn(n, c, hw, hw, generator=g(130))
    xb = nhwc(x)
    mp = torch.empty(n, hw // 2, hw // 2, c, dtype=bf16, device="cuda")
    ops.maxpool2(xb, mp)
    assert torch.equal(nchw(mp), F.max_pool2d(rb(x), 2))
    sc, sh = torch.rand(c, generator=g(131)) + 0.5, torch.randn(c, generator=g(132))
    ea = torch.empty_like(xb)
    ops.elu_affine(xb, dev(sc), dev(sh), ea)
    assert rel_l2(nchw(ea), F.elu(rb(x)) * sc[None, :, None, None] + sh[None, :, None, None]) < 4e-3
    cs = torch.empty(c, device="cuda")
    ops.colsum(xb, c, cs)
    assert rel_l2(cs, rb(x).sum((0, 2, 3))) < 1e-5
    w, b = torch.randn(2, c * hw * hw, generator=g(133)) * 0.02, torch.randn(2, generator=g(134))
    lg = torch.empty(n, 2, device="cuda")
    ops.linear_nhwc(xb, dev(w), dev(b), lg)
    assert rel_l2(lg, F.linear(rb(x).flatten(1), w, b)) < 1e-5
    back = torch.empty(n, 5, hw, hw, device="cuda")
    ops.nhwc_to_nchw_f32(xb, 5, back)
    assert torch.equal(back.cpu(), rb(x)[:, :5])
    img = torch.rand(n, 3, hw, hw, generator=g(135))
    o16 = torch.empty(n, hw, hw, 16, dtype=bf16, device="cuda")
    ops.nchw_to_nhwc_bf16(dev(img), o16)
    assert torch.equal(nchw(o16)[:, :3], rb(img)) and float(o16[..., 3:].float().abs().max()) == 0


def test_invalid_arguments_are_rejected(ops):
    """Error behaviour of the boundary: bad shapes raise, nothing is launched."""
    from combat_amd._lib import CombatHipError
    w, pc = make_conv(ops, 64, 64, 3, 1, 1, 140)
    x = torch.zeros(1, 4, 4, 64, dtype=bf16, device="cuda")
    y = torch.zeros(1, 4, 4, 64, dtype=bf16, device="cuda")
    a = ops.conv_args(x, y, pc, 0)
    a.stride = 3
    with pytest.raises(CombatHipError):
        ops.conv_launch(a)
    a = ops.conv_args(x, y, pc, 0)
    a.stats_kind = 1  # no stats buffer
    with pytest.raises(CombatHipError):
        ops.conv_launch(a)
    with pytest.raises(CombatHipError):
        ops.maxpool2(torch.zeros(1, 3, 3, 8, dtype=bf16, device="cuda"), y)


@pytest.mark.parametrize("n,hw", [(5, 32), (3, 64)])
def test_log_terms_equal_the_aten_spelling(n, hw):
    """combat_log_terms against the reference's own expressions (train_generator.py:234-247): MSE from per-plane partial
    sums, loss_grad_l2 through F.pad(., (1, 1, 2, 1)) and the two difference tensors, argmax == 1 count."""
    import torch.nn.functional as F
    from combat_amd import ops as O_
    from combat_amd._lib import lib
    from combat_amd.step import AlternatedStep
    x = torch.rand(n, 3, hw, hw, generator=g(300)).cuda() * 2 - 1
    xb = (x + 0.05 * torch.randn(n, 3, hw, hw, generator=g(301)).cuda()).clamp(-1, 1)
    mse = ((xb - x) ** 2).sum((2, 3)).reshape(-1).contiguous()
    logits = torch.randn(n, 2, generator=g(302)).cuda()
    logits[0, 1] = logits[0, 0]          # a tie is class 0 (argmax returns the first maximum)
    acc = torch.tensor([1.5, 2.5, 0, 0, 0, 0, 0, 0], dtype=torch.float64, device="cuda")
    hits = torch.tensor(3.0, dtype=torch.float64, device="cuda")
    O_.check(lib.combat_log_terms(x.data_ptr(), xb.data_ptr(), mse.data_ptr(), n, hw, logits.data_ptr(), acc.data_ptr(),
                                  hits.data_ptr(), torch.cuda.current_stream().cuda_stream), "log terms")
    torch.cuda.synchronize()
    assert abs(float(acc[0]) - 1.5 - float(F.mse_loss(xb, x))) < 1e-6 * float(F.mse_loss(xb, x)) + 1e-9
    ref = float(AlternatedStep._grad_l2(x, xb))
    assert abs(float(acc[1]) - 2.5 - ref) < 1e-5 * ref + 1e-9, (float(acc[1]) - 2.5, ref)
    assert float(hits) == 3.0 + float((logits.argmax(1) == 1).sum())


@pytest.mark.parametrize("n,hw,c,k", [(128, 32, 64, 128), (128, 8, 256, 512), (6, 16, 128, 256)])
def test_conv_pair_equals_two_launches(n, hw, c, k):
    """combat_conv_gemm_pair (a residual block's stride-2 3x3 convolution + its 1x1 stride-2 shortcut over the same input,
    preact_resnet.py:33-36) against the two combat_conv_gemm launches it replaces: bit-identical outputs and statistics;
    and a pair that cannot be grouped (a stride-1 3x3 on the DMA kernel + a 1x1) runs as two launches."""
    import ctypes, math
    from combat_amd import ops as O_
    from combat_amd._lib import lib
    st = torch.cuda.current_stream().cuda_stream
    x = torch.relu(torch.randn(n, hw, hw, c, generator=g(400))).to(bf16).cuda()
    w3 = (torch.randn(k, c, 3, 3, generator=g(401)) / math.sqrt(9 * c)).cuda().contiguous(memory_format=torch.channels_last)
    w1 = (torch.randn(k, c, 1, 1, generator=g(402)) / math.sqrt(c)).cuda().contiguous(memory_format=torch.channels_last)
    pc3, pc1 = O_.PackedConv(w3, 2, 1, c), O_.PackedConv(w1, 2, 0, c)
    pc3.pack()
    pc1.pack()
    p = hw // 2
    aff = O_.Affine(torch.rand(k, device="cuda") + 0.5, torch.randn(k, device="cuda"), 0, True, 0.0)

    def make():
        y3 = torch.zeros(n, p, p, k, dtype=bf16, device="cuda")
        a3t = torch.zeros_like(y3)
        y1 = torch.zeros_like(y3)
        a = O_.conv_args(x, y3, pc3, 0, stats_kind=1, act_dst=a3t, act=aff)
        rows, _ = O_.conv_stats_layout(a)
        stt = torch.zeros(rows, 2, k, device="cuda")
        a.stats = stt.data_ptr()
        b = O_.conv_args(x, y1, pc1, 0)
        return a, b, (y3, a3t, y1, stt)

    a, b, sep = make()
    O_.check(lib.combat_conv_gemm(ctypes.byref(a), st), "a")
    O_.check(lib.combat_conv_gemm(ctypes.byref(b), st), "b")
    a2, b2, grp = make()
    O_.check(lib.combat_conv_gemm_pair(ctypes.byref(b2), ctypes.byref(a2), st), "pair")     # shortcut first, as the engines record it
    torch.cuda.synchronize()
    for u, v in zip(sep, grp):
        assert torch.equal(u, v)
    assert float(sep[0].float().abs().max()) > 0 and float(sep[2].float().abs().max()) > 0
    # not groupable: stride-1 3x3 (LDS-DMA halo kernel) + 1x1 stride 1
    if hw >= 8:
        w3s = (torch.randn(c, c, 3, 3, generator=g(403)) / math.sqrt(9 * c)).cuda().contiguous(memory_format=torch.channels_last)
        pcs = O_.PackedConv(w3s, 1, 1, c)
        pcs.pack()
        w1s = (torch.randn(c, c, 1, 1, generator=g(404)) / math.sqrt(c)).cuda().contiguous(memory_format=torch.channels_last)
        pc1s = O_.PackedConv(w1s, 1, 0, c)
        pc1s.pack()
        outs = [torch.zeros(n, hw, hw, c, dtype=bf16, device="cuda") for _ in range(4)]
        O_.check(lib.combat_conv_gemm(ctypes.byref(O_.conv_args(x, outs[0], pcs, 0)), st), "s")
        O_.check(lib.combat_conv_gemm(ctypes.byref(O_.conv_args(x, outs[1], pc1s, 0)), st), "s1")
        aa, bb = O_.conv_args(x, outs[2], pcs, 0), O_.conv_args(x, outs[3], pc1s, 0)
        O_.check(lib.combat_conv_gemm_pair(ctypes.byref(aa), ctypes.byref(bb), st), "pair fallback")
        torch.cuda.synchronize()
        assert torch.equal(outs[0], outs[2]) and torch.equal(outs[1], outs[3])
