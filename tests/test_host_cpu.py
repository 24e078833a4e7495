"""CPU-only tests: the C-ABI library loads and exports what include/combat_hip.h declares (no
compute calls), struct layouts agree with the header, and the host logic around the step (flags,
tables, sampling, data, checkpoint plumbing, data-parallel exchange over gloo)."""
import ctypes
import json
import math
import os
import random
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "combat_hip.h")


@pytest.fixture(scope="module")
def L():
    from combat_amd import _lib
    return _lib


# ---------------------------------------------------------------- C ABI


def declared_functions():
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    return sorted(set(re.findall(r"\b(combat_\w+)\s*\(", src)) - {"combat_conv_args", "combat_wgrad_args"})


def test_library_exports_every_declared_symbol(L):
    names = declared_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L.lib, n), "libcombat_hip.so does not export %s" % n
        assert n in L.SIGNATURES, "ctypes binding has no signature for %s" % n
    assert set(L.SIGNATURES) == set(names)
    assert L.lib.combat_version().startswith(b"combat_hip gfx950")
    assert L.lib.combat_abi_version() >= 1


def test_ctypes_struct_layout_matches_header(L, tmp_path):
    """Compile a C program against the header and compare sizeof/offsetof with the ctypes mirrors."""
    fields = {"combat_conv_args": [f for f, _ in L.ConvArgs._fields_],
              "combat_wgrad_args": [f for f, _ in L.WgradArgs._fields_]}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "combat_hip.h"', "int main(void){"]
    for s, fl in fields.items():
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (s, s))
        for f in fl:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (s, f, s, f))
    lines.append("return 0;}")
    c = tmp_path / "layout.c"
    c.write_text("\n".join(lines))
    exe = str(tmp_path / "layout")
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(c), "-o", exe], check=True)
    out = dict(l.split() for l in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.splitlines())
    for s, cls in (("combat_conv_args", L.ConvArgs), ("combat_wgrad_args", L.WgradArgs)):
        assert int(out[s]) == ctypes.sizeof(cls)
        for f, _ in cls._fields_:
            assert int(out["%s.%s" % (s, f)]) == getattr(cls, f).offset, (s, f)


def test_boundary_rejects_bad_arguments_without_a_gpu(L):
    """Validation happens before any HIP call: COMBAT_EINVAL (-1), nothing launched."""
    a = L.ConvArgs()
    assert L.lib.combat_conv_gemm(ctypes.byref(a), None) == -1
    a.src = a.wpack = a.dst = 4096
    a.N = a.H = a.W = a.P = a.Q = 4
    a.C = a.K = 64
    a.R = a.S = 3
    a.stride, a.pad, a.kpad, a.rows_pad = 3, 1, 576, 128           # stride 3 is unsupported
    assert L.lib.combat_conv_gemm(ctypes.byref(a), None) == -1
    a.stride, a.stats_kind = 1, 1                                   # statistics without a buffer
    assert L.lib.combat_conv_gemm(ctypes.byref(a), None) == -1
    assert L.lib.combat_memset_zero(None, 16, None) == -1
    assert L.lib.combat_maxpool2(4096, 1, 3, 3, 8, 4096, None) == -1
    assert L.lib.combat_conv_stats_granule(99) == -1
    with pytest.raises(L.CombatHipError, match="combat_x.*invalid"):
        L.check(-1, "combat_x", "shape")


def test_tile_choice_and_stats_layout_are_host_queries(L):
    a = L.ConvArgs()
    a.N, a.H, a.W, a.C, a.P, a.Q, a.K = 128, 32, 32, 64, 32, 32, 64
    a.R = a.S = 3
    a.stride, a.pad, a.kpad, a.rows_pad = 1, 1, 576, 128
    assert L.lib.combat_conv_pick_tile(ctypes.byref(a)) == L.TILE_S128x64      # no prologue, C = K = 64, >= 2 tiles per CU:
    #                                                                             weight-stationary persistent kernel
    rows, rpi = ctypes.c_int32(), ctypes.c_int32()
    def layout(kind):
        a.stats_kind = kind
        assert L.lib.combat_conv_stats_layout(ctypes.byref(a), ctypes.byref(rows), ctypes.byref(rpi)) == 0
        a.stats_kind = 0
        return rows.value, rpi.value
    assert layout(1) == (4096, 32)                                              # one row per wave: 32 per image
    assert layout(1 | L.STATS_PER_WORKGROUP) == (256, 0)                        # one per persistent workgroup (4 tiles each)
    a.N = 48                                                                    # 384 tiles: the ring kernel
    assert L.lib.combat_conv_pick_tile(ctypes.byref(a)) == L.TILE_D128x64
    assert layout(1) == (1536, 32) and layout(2 | L.STATS_PER_WORKGROUP) == (384, 8)   # ... one per 128-pixel tile
    a.N = 128
    a.stats_kind = 8                                                            # unknown statistics bits
    assert L.lib.combat_conv_gemm(ctypes.byref(a), None) == -1
    a.stats_kind = L.STATS_PER_WORKGROUP                                        # the row-form bit without a kind
    assert L.lib.combat_conv_gemm(ctypes.byref(a), None) == -1
    a.stats_kind = 0
    a.pro_act = 1
    assert L.lib.combat_conv_pick_tile(ctypes.byref(a)) == L.TILE_H128x64      # prologue, big layer: halo 128x64
    assert L.lib.combat_conv_stats_layout(ctypes.byref(a), ctypes.byref(rows), ctypes.byref(rpi)) == 0
    assert rows.value == 128 * 1024 // 32 and rpi.value == 32
    a.stats_kind = 1 | L.STATS_PER_WORKGROUP                                    # (the register-staged kernels ignore the bit)
    assert L.lib.combat_conv_stats_layout(ctypes.byref(a), ctypes.byref(rows), ctypes.byref(rpi)) == 0
    assert rows.value == 128 * 1024 // 32 and rpi.value == 32
    a.stats_kind = 0
    a.H = a.W = a.P = a.Q = 4
    a.C = a.K = 512
    a.kpad, a.rows_pad = 4608, 512
    assert L.lib.combat_conv_pick_tile(ctypes.byref(a)) == L.TILE_H64x64       # skinny layer: halo 64x64
    a.stride, a.P, a.Q = 2, 2, 2
    a.pro_act = 0
    assert L.lib.combat_conv_pick_tile(ctypes.byref(a)) in (L.TILE_G128x64, L.TILE_G128x32)   # strided, no prologue: DMA gather
    a.pro_act = 1
    assert L.lib.combat_conv_pick_tile(ctypes.byref(a)) in (L.TILE_64x64, L.TILE_64x128)      # with a prologue: register-staged gather


def test_eight_channel_and_shortcut_launch_forms_are_host_queries(L):
    """COMBAT_TILE_K8 and the second reduction source (combat_conv_args.src2) are decided on the host: which launches take
    them, and that a malformed one is rejected before any HIP call."""
    a = L.ConvArgs()
    a.src = a.wpack = a.dst = 4096
    a.N, a.H, a.W, a.C, a.P, a.Q, a.K = 128, 32, 32, 64, 32, 32, 8              # generator output layer / stem input gradient
    a.R = a.S = 3
    a.stride, a.pad, a.kpad, a.rows_pad = 1, 1, 576, 16
    assert L.lib.combat_conv_pick_tile(ctypes.byref(a)) == L.TILE_K8
    a.mode = 1
    assert L.lib.combat_conv_pick_tile(ctypes.byref(a)) == L.TILE_K8
    a.add_post = 4096                                                            # a residual operand: the general kernels
    assert L.lib.combat_conv_pick_tile(ctypes.byref(a)) != L.TILE_K8
    a.add_post, a.H, a.W, a.P, a.Q = None, 12, 12, 12, 12                        # not whole 16 x 8 tiles
    assert L.lib.combat_conv_pick_tile(ctypes.byref(a)) != L.TILE_K8
    # a block's stride-2 input gradient with its shortcut as second source: src = dY [N,16,16,128] -> dst = dX [N,32,32,64]
    b = L.ConvArgs()
    b.src = b.wpack = b.dst = 4096
    b.N, b.H, b.W, b.C, b.P, b.Q, b.K = 128, 16, 16, 128, 32, 32, 64
    b.R = b.S = 3
    b.stride, b.pad, b.kpad, b.rows_pad, b.mode = 2, 1, 1152, 64, 1
    plain = L.lib.combat_conv_pick_tile(ctypes.byref(b))
    assert plain in (L.TILE_G128x64, L.TILE_G128x32)
    b.src2, b.wpack2, b.kpad2, b.rows_pad2 = 4096, 4096, 128, 64
    assert L.lib.combat_conv_pick_tile(ctypes.byref(b)) == plain
    b.mode = 0                                                                   # forward launches have no such form
    assert L.lib.combat_conv_pick_tile(ctypes.byref(b)) == 0
    b.mode, b.kpad2 = 1, 64                                                      # operand shorter than the reduction
    assert L.lib.combat_conv_pick_tile(ctypes.byref(b)) == 0
    b.kpad2, b.wpack2 = 128, None                                                # a source without its operand
    assert L.lib.combat_conv_gemm(ctypes.byref(b), None) == -1


def test_modules_have_no_cpu_fallback():
    from combat_amd import nets
    from combat_amd._lib import CombatHipError
    with pytest.raises(CombatHipError):
        nets.PreActResNet18()(torch.zeros(2, 3, 32, 32))
    with pytest.raises(CombatHipError):
        nets.UnetGenerator(None)(torch.zeros(2, 3, 32, 32))


# ---------------------------------------------------------------- flags


def test_flags_match_the_reference_parser():
    import importlib
    sys.path.insert(0, ROOT)
    cfg = importlib.import_module("config")
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "config_flags.json")))
    ours = {a.dest: a for a in cfg.get_arguments()._actions if a.dest != "help"}
    for name, spec in ref.items():
        assert name in ours, name
        a = ours[name]
        d = list(a.default) if isinstance(a.default, (list, tuple)) else a.default
        assert d == spec["default"], (name, d, spec["default"])
        assert getattr(a.type, "__name__", None) == spec["type"], name
        assert (a.nargs == 0) == spec["store_true"], name
        assert (list(a.choices) if a.choices else None) == spec["choices"], name
    opt = cfg.get_arguments().parse_args(["--dataset", "cifar10", "--pc", "0.5", "--noise_rate", "0.08",
                                          "--saving_prefix", "train_generator"])
    assert (opt.bs, opt.lr_C, opt.ratio, opt.L2_weight, opt.clean_model_weight) == (128, 1e-2, 0.65, 0.02, 0.8)


# ---------------------------------------------------------------- step host logic


class Opt:
    noise_rate, ratio, kernel_size, sigma = 0.08, 0.65, 3, (0.1, 1.0)
    pc, target_label, attack_mode, num_classes = 0.5, 0, "all2one", 10
    input_height = input_width = 32
    dataset, post_transform_option, random_crop, random_rotation = "cifar10", "use", 5, 10


def test_poison_tables_match_oracle_ordering():
    from combat_amd.step import create_targets_bd, poison_tables
    from oracle import combat_oracle as O
    g = torch.Generator().manual_seed(0)
    for n, nb in ((16, 0), (16, 3), (128, 7)):
        t = torch.randint(0, 10, (n,), generator=g)
        t[:8] = 0
        bd = create_targets_bd(t, Opt())
        perm, tot, idx_small, idx_total = poison_tables(t, bd, nb)
        o_perm, o_tot = O.poison_order(t, bd, nb)
        assert perm.tolist() == o_perm.tolist() and tot.tolist() == o_tot.tolist()
        assert idx_small[:nb].tolist() == o_perm[:nb].tolist() and idx_small[nb:].abs().sum() == 0
        assert idx_total[:nb].tolist() == list(range(n, n + nb)) and idx_total[nb:].tolist() == o_perm[nb:].tolist()
    o = Opt()
    o.attack_mode = "all2all"
    assert create_targets_bd(torch.tensor([9, 0, 4]), o).tolist() == [0, 1, 5]
    o.attack_mode = "nope"
    with pytest.raises(Exception, match="not implemented"):
        create_targets_bd(torch.tensor([1]), o)


def test_draw_randomness_consumes_the_three_rng_streams_in_reference_order():
    from combat_amd.augment import PostTensorTransform
    from combat_amd.step import draw_randomness
    t = torch.tensor([0, 0, 0, 0, 0, 0, 3, 4, 5, 0, 1, 2])
    bd = torch.zeros_like(t)

    def draw():
        np.random.seed(5)
        random.seed(5)
        torch.manual_seed(5)
        return draw_randomness(t, bd, Opt(), PostTensorTransform(Opt()))

    a, b = draw(), draw()
    assert a.num_bd == b.num_bd and a.sigma_c == b.sigma_c and a.sigma_g == b.sigma_g
    np.random.seed(5)
    assert a.num_bd == int(np.sum(np.random.rand(7) < 0.5))          # numpy global RNG, :183
    assert 0.1 <= a.sigma_c <= 1.0 and 0.1 <= a.sigma_g <= 1.0 and a.sigma_c != a.sigma_g
    assert len(a.aug) == 5
    for x, y in zip(a.aug, b.aug):
        assert (x is None and y is None) or np.array_equal(x, y)


def test_augmentation_sampler_distributions():
    from combat_amd.augment import PostTensorTransform
    random.seed(0)
    torch.manual_seed(0)
    tf = PostTensorTransform(Opt())
    n, calls = 256, 400
    crops = rots = 0
    flips = []
    for _ in range(calls):
        p = tf.sample(n)
        assert p.shape == (n, 4) and p.dtype == np.float32
        if np.any(p[:, :2] != 0):
            crops += 1
            assert p[:, :2].min() >= -5 and p[:, :2].max() <= 5 and np.all(p[:, :2] == np.round(p[:, :2]))
        if np.any(p[:, 2] != 0):
            rots += 1
            assert np.abs(p[:, 2]).max() <= math.radians(10) + 1e-6
            assert 0.35 < np.mean(p[:, 2] != 0) < 0.65          # kornia applies the rotation to ~half the samples
        flips.append(p[:, 3].mean())
        assert set(np.unique(p[:, 3])) <= {0.0, 1.0}
    assert 0.72 < crops / calls < 0.88 and 0.42 < rots / calls < 0.58   # batch-level gates p=0.8 / p=0.5
    assert 0.47 < np.mean(flips) < 0.53
    o = Opt()
    o.post_transform_option = "no_use"
    assert PostTensorTransform(o).sample(8) is None
    o.post_transform_option = "use_modified"
    assert all(np.all(PostTensorTransform(o).sample(8)[:, :2] == 0) for _ in range(20))


def test_trigger_constants_agree_with_oracle():
    from combat_amd import trigger
    from oracle import combat_oracle as O
    for n in (32, 64):
        assert torch.allclose(trigger.lowpass_matrix(n, 0.65), O.lowpass_matrix(n, 0.65), atol=1e-7)
        assert torch.allclose(trigger.dct_matrix(n).float(), O.dct_matrix(n), atol=1e-7)
    for s in (0.1, 0.37, 1.0):
        assert np.allclose(trigger.gaussian_kernel1d(s), O.gaussian_kernel1d(s).numpy(), atol=1e-7)
    torch.manual_seed(3)
    a = trigger.sample_sigma((0.1, 1.0))
    torch.manual_seed(3)
    assert a == O.sample_sigma((0.1, 1.0))


def test_bucket_rule_for_poisoned_sub_batch():
    from combat_amd.step import bucket
    assert [bucket(n) for n in (1, 8, 9, 17, 33, 128)] == [8, 8, 16, 32, 64, 128]
    with pytest.raises(ValueError):
        bucket(5000)


# ---------------------------------------------------------------- parameters / checkpoints


def test_flat_params_keep_state_dict_semantics():
    from combat_amd import nets
    from combat_amd.engine import FlatParams
    torch.manual_seed(0)
    m = nets.UnetGenerator(None)
    before = {k: v.clone() for k, v in m.state_dict().items()}
    fp = FlatParams(m)
    after = m.state_dict()
    assert list(after) == list(before)
    for k in before:
        assert torch.equal(after[k], before[k]) and after[k].shape == before[k].shape
    w = m.conv1_0.weight
    assert w.permute(0, 2, 3, 1).is_contiguous()                      # physical [K][R][S][C]
    o, n, _ = fp.offsets["conv1_0.weight"]
    assert torch.equal(fp.flat[o:o + n].view(128, 3, 3, 64), before["conv1_0.weight"].permute(0, 2, 3, 1))
    fp.grad[o:o + n] = torch.arange(n, dtype=torch.float32)
    g = fp.logical(fp.grad, "conv1_0.weight")
    assert g.shape == (128, 64, 3, 3) and float(g[1, 2, 0, 1]) == float(((1 * 3 + 0) * 3 + 1) * 64 + 2)
    # load_state_dict writes through to the flat buffer; torch.save/load round trip keeps values
    sd = {k: torch.randn_like(v) for k, v in before.items()}
    m.load_state_dict(sd)
    assert torch.equal(fp.flat[o:o + n].view(128, 3, 3, 64), sd["conv1_0.weight"].permute(0, 2, 3, 1))
    assert all(o % 64 == 0 for o, _, _ in fp.offsets.values())


def test_data_loader_normalisation_sharding_and_poison_flags():
    from combat_amd.data import ArrayLoader, get_dataloader, poison_flags, synthetic_cifar10

    class O2(Opt):
        bs, synthetic, synthetic_size, debug, data_root = 32, True, 200, False, "/nonexistent"
    dl = get_dataloader(O2(), True)
    assert len(dl) == 7
    xs = [b for b in dl]
    assert xs[0][0].shape == (32, 3, 32, 32) and xs[0][0].dtype == torch.float32 and xs[-1][0].shape[0] == 8
    x, y = synthetic_cifar10(200, 1234)
    assert xs[0][0].min() >= -1 and xs[0][0].max() <= 1
    assert torch.allclose(torch.sort(torch.cat([b[0] for b in xs]).flatten())[0][::50000],
                          torch.sort(((torch.from_numpy(x).float() / 255) - 0.5).flatten() / 0.5)[0][::50000])
    torch.manual_seed(1)
    a = torch.cat([b[1] for b in ArrayLoader(x, np.arange(200), 16, True, rank=0, world=2)])
    torch.manual_seed(1)
    b = torch.cat([b[1] for b in ArrayLoader(x, np.arange(200), 16, True, rank=1, world=2)])
    assert len(a) == len(b) == 100 and set(a.tolist()).isdisjoint(b.tolist())
    random.seed(0)
    flags = poison_flags(y, O2(), 10)
    assert flags.sum() == int(0.5 * (y == 0).sum()) and np.all(y[flags] == 0)
    dlp = get_dataloader(O2(), True, poisoned=True)
    xb, yb, pb = next(iter(dlp))
    assert pb.dtype == torch.bool and pb.shape == (32,)
    O2.synthetic = False
    with pytest.raises(FileNotFoundError, match="--synthetic"):
        get_dataloader(O2(), True)


def test_array_loader_shards_are_disjoint_equal_and_rng_independent():
    """data-parallel shards of an epoch are disjoint parts of ONE permutation that depends
    only on (base_seed, epoch) -- not on what a rank drew from its global generators meanwhile -- and every
    rank walks the same number of batches (a rank one step ahead would hang in the gradient all-reduce)."""
    from combat_amd.data import ArrayLoader
    x = np.zeros((203, 3, 4, 4), np.uint8)      # 203 = 3 * 67 + 2: ragged over 3 ranks
    loaders = [ArrayLoader(x, np.arange(203), 16, True, rank=r, world=3, base_seed=11) for r in range(3)]
    assert len({len(l) for l in loaders}) == 1 and len(loaders[0]) == 5          # ceil(68 / 16) on every rank
    for epoch in range(3):
        ids = []
        for r, l in enumerate(loaders):
            torch.manual_seed(1000 * r + epoch)           # ranks diverge in their global generator ...
            torch.rand(r + 1)
            got = torch.cat([b[1] for b in l])
            assert len(got) == 68
            ids.append(got)
        allids = torch.cat(ids).tolist()
        assert set(allids) == set(range(203)) and len(allids) == 204              # one wrapped sample pads the epoch
        assert torch.equal(ids[0], loaders[0].epoch_order(epoch))                  # ... and the shards do not care
    a, b2 = loaders[0].epoch_order(0), loaders[0].epoch_order(1)
    assert not torch.equal(a, b2)
    # evaluation loaders: exact (possibly uneven) strided shards, nothing counted twice
    ev = [ArrayLoader(x, np.arange(203), 16, False, rank=r, world=3) for r in range(3)]
    got = torch.cat([torch.cat([b[1] for b in l]) for l in ev])
    assert sorted(got.tolist()) == list(range(203))
    # unseeded construction draws the base seed from the global generator (DataLoader's RandomSampler does)
    torch.manual_seed(5)
    s1 = ArrayLoader(x, np.arange(203), 16, True).base_seed
    torch.manual_seed(5)
    assert ArrayLoader(x, np.arange(203), 16, True).base_seed == s1


def test_progress_bar_and_scalar_writer(tmp_path, capsys):
    from combat_amd.log import SummaryWriter, progress_bar
    progress_bar(0, 3, "Clean Acc: 10.0000")
    progress_bar(2, 3, "Clean Acc: 11.0000")
    out = capsys.readouterr().out
    assert " [" in out and "Clean Acc: 11.0000" in out and " 3/3 " in out and out.endswith("\n")
    w = SummaryWriter(str(tmp_path / "log"))
    w.add_scalars("Clean Accuracy", {"Clean": torch.tensor(12.5), "Bd": 3}, 0)
    if w.tb is None:
        rec = json.loads(open(tmp_path / "log" / "scalars.jsonl").read())
        assert rec["values"] == {"Clean": 12.5, "Bd": 3.0} and rec["tag"] == "Clean Accuracy"


# ---------------------------------------------------------------- data parallel (gloo, world size 2)


def test_bucket_ranges_are_contiguous_and_reversed():
    from combat_amd.dist import bucket_ranges
    r = bucket_ranges(10_000_000, [100, 2_000_000, 2_500_000, 6_000_000, 9_900_000])
    assert r == [(6_000_000, 10_000_000), (2_000_000, 6_000_000), (0, 2_000_000)]
    assert bucket_ranges(1000, [10, 500]) == [(0, 1000)]


def _dp_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from combat_amd import dist as cdist
    from combat_amd.data import ArrayLoader
    from oracle import combat_oracle as O
    r, _, w = cdist.init("gloo")
    assert (r, w) == (rank, world)
    # identical replicas after broadcast
    torch.manual_seed(100 + rank)
    lin = torch.nn.Linear(64, 32)
    cdist.broadcast_module(lin)
    # every rank computes the gradient of its own shard; bucketed all-reduce; SGD with grad_scale 1/world
    torch.manual_seed(7)
    xs, ys = torch.randn(64, 64), torch.randn(64, 32)
    shard = slice(rank * 32, (rank + 1) * 32)
    loss = ((lin(xs[shard]) - ys[shard]) ** 2).mean()
    gw, gb = torch.autograd.grad(loss, [lin.weight, lin.bias])
    total = 4096
    flat = torch.zeros(total)
    flat[: gw.numel()] = gw.flatten()
    flat[3000: 3000 + 32] = gb
    ranges = cdist.bucket_ranges(total, [1024, 3000], min_elems=512)
    red = cdist.GradReducer(flat, ranges)
    for i in range(len(ranges)):          # buckets are launched in backward order, as the plans do
        red.launch(i)
    red.wait()
    gw_avg = flat[: gw.numel()].view_as(gw) * red.grad_scale
    gb_avg = flat[3000: 3032] * red.grad_scale
    # reference: the full-batch gradient (mean over 64 = average of the two shard means)
    lin2 = torch.nn.Linear(64, 32)
    lin2.load_state_dict(lin.state_dict())
    full = ((lin2(xs) - ys) ** 2).mean()
    fw, fb = torch.autograd.grad(full, [lin2.weight, lin2.bias])
    ok = torch.allclose(gw_avg, fw, atol=1e-6) and torch.allclose(gb_avg, fb, atol=1e-6)
    params, bufs = [lin.weight.detach().clone(), lin.bias.detach().clone()], [None, None]
    O.sgd_nesterov_step(params, [gw_avg, gb_avg], bufs, 1e-2)
    gathered = [torch.zeros_like(params[0]) for _ in range(world)]
    dist.all_gather(gathered, params[0])
    ok = ok and torch.equal(gathered[0], gathered[1])                 # replicas stay identical
    counts = cdist.all_reduce_counters([rank + 1, 10])
    ok = ok and counts == [3.0, 20.0]
    torch.manual_seed(1 + 17 * rank)     # unseeded ranks: the loader's base seed is rank 0's draw, broadcast
    dl = ArrayLoader(np.zeros((51, 3, 4, 4), np.uint8), np.arange(51), 8, True, rank=rank, world=world)
    for epoch in range(2):
        ids = torch.cat([b[1] for b in dl])
        allids = [torch.zeros(26, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(allids, ids)
        ok = ok and len(ids) == 26 and len(dl) == 4
        ok = ok and set(torch.cat(allids).tolist()) == set(range(51))   # disjoint (one wrapped sample), complete epoch shards
    # BatchNorm running statistics are rank-local in training and averaged before evaluation / checkpoint
    bn = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 1), torch.nn.BatchNorm2d(4))
    with torch.no_grad():
        bn[1].running_mean.fill_(1.0 + rank)
        bn[1].running_var.fill_(2.0 + 2 * rank)
        bn[1].num_batches_tracked.fill_(7)
    moved = cdist.average_bn_buffers(bn)
    ok = ok and moved == 2 and torch.equal(bn[1].running_mean, torch.full((4,), 1.5)) and \
        torch.equal(bn[1].running_var, torch.full((4,), 3.0)) and int(bn[1].num_batches_tracked) == 7
    open(os.path.join(tmp, "ok%d" % rank), "w").write(str(bool(ok)))
    dist.destroy_process_group()


def test_data_parallel_gradient_exchange_gloo_world2(tmp_path):
    import torch.multiprocessing as mp
    port = 29600 + os.getpid() % 300
    mp.spawn(_dp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert open(tmp_path / "ok0").read() == "True" and open(tmp_path / "ok1").read() == "True"


def test_loader_batches_equal_totensor_normalize_bit_for_bit():
    """ArrayLoader converts in numpy (one thread, into a reused buffer on CUDA hosts): the batch must equal the
    reference's ToTensor + Normalize(0.5, 0.5) chain (utils/dataloader.py:35-39) exactly, ragged last batch included,
    and successive batches must not alias each other within the ring's depth."""
    from combat_amd.data import ArrayLoader, synthetic_cifar10
    x, y = synthetic_cifar10(300, 3, 32, 10)
    ld = ArrayLoader(x, y, 128, True, base_seed=5)
    order = ld.epoch_order(0)
    got = [(xb.clone(), yb.clone(), xb) for xb, yb in ld]
    assert [g[0].shape[0] for g in got] == [128, 128, 44]
    for i, (xb, yb, _) in enumerate(got):
        idx = order[i * 128:(i + 1) * 128]
        ref = ((torch.from_numpy(x)[idx].float() / 255.0) - 0.5) / 0.5
        assert torch.equal(xb, ref) and torch.equal(yb, torch.from_numpy(y)[idx])
    assert got[0][2].data_ptr() != got[1][2].data_ptr()


def test_make_grid_matches_torchvision_semantics(tmp_path):
    """combat_amd.log.make_grid / image_grid / ScalarWriter.add_image: the reference's image logging
    (train_generator.py:310-315: torch.cat([inputs, inputs_bd], dim=2) -> Denormalizer -> torchvision.utils.make_grid(
    normalize=True) -> add_image).  torchvision is absent; the expectations below are make_grid's documented
    behaviour: ONE min-max range over the whole batch, nrow = 8 images per row, 2 pixels of zero padding around and
    between images, single-channel images repeated to three channels."""
    from combat_amd import log
    b = torch.arange(3 * 1 * 2 * 2, dtype=torch.float32).view(3, 1, 2, 2) - 4.0      # values -4 .. 7
    g = log.make_grid(b, normalize=True)
    assert tuple(g.shape) == (3, 2 + 2 * 2, 3 * (2 + 2) + 2)
    assert float(g.min()) == 0.0 and abs(float(g.max()) - 1.0) < 1e-6
    assert torch.equal(g[0], g[1]) and torch.equal(g[1], g[2])
    want = (b[1, 0] + 4.0) / 11.0
    assert torch.allclose(g[0, 2:4, 6:8], want, atol=1e-6)                           # second image at column 2 + (2 + 2)
    assert float(g[:, :2].abs().sum()) == 0.0 and float(g[:, :, 4:6].abs().sum()) == 0.0   # padding stays zero
    ten = log.make_grid(torch.rand(10, 3, 4, 4), normalize=True)
    assert tuple(ten.shape) == (3, 2 * (4 + 2) + 2, 8 * (4 + 2) + 2)                 # 8 per row, 2 rows

    class O:
        dataset = "cifar10"

    x, xb = torch.rand(4, 3, 8, 8) * 2 - 1, torch.rand(4, 3, 8, 8) * 2 - 1
    grid = log.image_grid(x, xb, O())
    assert tuple(grid.shape) == (3, 16 + 4, 4 * (8 + 2) + 2)                          # clean images on top of their copies
    assert torch.allclose(log.denormalize(x, O()), x * 0.5 + 0.5)
    w = log.ScalarWriter(str(tmp_path))
    w.add_image("Images", grid, global_step=20)
    if w.tb is None:
        path = tmp_path / "images" / "Images_000020.ppm"
        raw = path.read_bytes()
        assert raw.startswith(b"P6\n42 20\n255\n") and len(raw) == len(b"P6\n42 20\n255\n") + 42 * 20 * 3


def test_conv_pair_rejects_what_the_single_call_rejects(L):
    """combat_conv_gemm_pair promises the results of two combat_conv_gemm calls, so it must return EINVAL
    -- before any kernel dereferences a shape -- wherever the single call does, for either of its two problems."""
    def good():
        a = L.ConvArgs()
        a.src = a.wpack = a.dst = 4096
        a.N, a.H, a.W, a.C, a.P, a.Q, a.K = 4, 16, 16, 64, 8, 8, 128
        a.R = a.S = 3
        a.stride, a.pad, a.kpad, a.rows_pad = 2, 1, 576, 128
        return a

    def breakers():
        def f(a): a.act_dst = 4096                       # activated output without its tables
        yield f
        def f(a): a.pro_scale = 4096                     # prologue scale without shift
        yield f
        def f(a): a.mask_scale = 4096                    # mask scale without shift
        yield f
        def f(a): a.mask_mul_scale = 1                   # multiply by a scale that is not there
        yield f
        def f(a): a.stats_kind = 2; a.stats = 4096       # norm-backward statistics without mask / xhat tables
        yield f
        def f(a): a.stats_kind = 1                       # statistics without a buffer
        yield f
        def f(a): a.kpad = 100                           # kpad not a multiple of 64
        yield f
        def f(a): a.C = 48; a.kpad = 448                 # channel count not a power of two
        yield f
        def f(a): a.stride = 3
        yield f
        def f(a): a.R = 2; a.S = 2
        yield f
        def f(a): a.N = 1 << 30                          # M overflow
        yield f
        def f(a): a.src = None
        yield f

    for brk in breakers():
        bad = good()
        brk(bad)
        assert L.lib.combat_conv_gemm(ctypes.byref(bad), None) == -1
        assert L.lib.combat_conv_gemm_pair(ctypes.byref(bad), ctypes.byref(good()), None) == -1
        assert L.lib.combat_conv_gemm_pair(ctypes.byref(good()), ctypes.byref(bad), None) == -1
    # the plan API rejects nonsense without touching a device
    assert L.lib.combat_plan_record(None, -1) == -1 and L.lib.combat_plan_size(None) == -1
    cp = L.lib.combat_plan_create()
    assert L.lib.combat_plan_set_after(cp, 0) == -1                         # nothing recorded yet
    assert L.lib.combat_plan_record(cp, -1) == 0 and L.lib.combat_memset_zero(4096, 16, None) == 0
    assert L.lib.combat_plan_record(cp, 0) == 0 and L.lib.combat_memset_zero(4096, 16, None) == 0
    assert L.lib.combat_plan_size(cp) == 2 and L.lib.combat_plan_set_after(cp, 0) == 0 and L.lib.combat_plan_set_after(cp, 1) == -1
    assert L.lib.combat_plan_run(cp, 0, 3, None, None, 0) == -1             # range beyond the plan
    assert L.lib.combat_plan_record_cancel() == 0                           # (nothing left armed)
    L.lib.combat_plan_destroy(cp)
