"""Network-level and step-level parity on the MI355X against the CPU oracle and the golden
vectors recorded from the reference modules.

Two references, two tolerances (both stated at each check):

  * tests/bf16_emu.py -- the oracle's dataflow with bf16 rounding at exactly the points where the
    HIP path stores bf16 (straight-through gradients).  Forward: the engines agree with it layer
    by layer (early layers <= 3e-3; deeper ones drift as single-ulp differences pass through
    1/sigma of small normalisation groups).  Backward: the emulation is *teacher-forced* with the
    tensors the engine actually stored, so activation masks and statistics are identical and the
    gradients must agree to rel-L2 <= 4e-2 (the engine additionally rounds ~20 gradient tensors
    to bf16 along the way) -- this is the check that pins the backward wiring.
  * the fp32 goldens recorded from the reference modules.  The distance between ANY bf16-activation
    implementation and fp32 at these randomly initialised test points is dominated by
    ReLU/LeakyReLU mask flips of pre-activations within one bf16 ulp of zero (~0.3 % of units per
    layer => ~5 % gradient rel-L2 per layer): measured with the emulation on the CPU,
    forward 1-4e-2, gradients 0.2-0.45 rel-L2, independent of batch size (B=4..128).  The golden
    checks therefore bound forward error at 5e-2 and gradient error at 0.5, and assert that the
    engine is no further from fp32 than the emulation is (x1.6).
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bf16_emu as E  # noqa: E402

pytestmark = pytest.mark.gpu
bf16 = torch.bfloat16


def T(a):
    return torch.from_numpy(np.asarray(a))


def rel_l2(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return float((a - b).norm() / max(float(b.norm()), 1e-30))


def seeded(ctor, seed):
    torch.manual_seed(seed)
    return ctor()


def grads_vs_summary(g, prefix, fp, names, tol_l2):
    """Whole-network relative error from the recorded per-tensor (l2, sampled values)."""
    num = den = 0.0
    worst = (0.0, None)
    for k in names:
        ours = fp.logical(fp.grad, k).double().cpu().flatten()
        ref_l2 = float(g["%s/%s/l2" % (prefix, k)])
        idx, ref = g["%s/%s/idx" % (prefix, k)], g["%s/%s/val" % (prefix, k)]
        d = ours[idx].numpy() - ref
        num += float((d ** 2).sum())
        den += float((ref ** 2).sum())
        e = abs(float(ours.norm()) - ref_l2) / max(ref_l2, 1e-12)
        if ref_l2 > 1e-6 and e > worst[0]:
            worst = (e, k)
    return (num / max(den, 1e-30)) ** 0.5, worst


@pytest.fixture(scope="module")
def mods():
    from combat_amd import engine, nets, ops, step
    return dict(engine=engine, nets=nets, ops=ops, step=step)


def stored(slot, names, c=None):
    """Engine slot buffers (bf16 NHWC) -> {name: fp32 NCHW on the CPU} for teacher forcing."""
    out = {}
    for k in names:
        if k in slot.bufs:
            v = slot.bufs[k].float().cpu().permute(0, 3, 1, 2).contiguous()
            out[k] = v[:, :c] if (c and k == "noise") else v
    return out


def flat_grads(fp, names):
    return torch.cat([fp.logical(fp.grad, k).reshape(-1).float().cpu() for k in names])


def sampled_err(g, prefix, named):
    num = den = 0.0
    for k, v in named:
        idx, ref = g["%s/%s/idx" % (prefix, k)], g["%s/%s/val" % (prefix, k)]
        d = v.double().flatten()[idx].numpy() - ref
        num += float((d ** 2).sum())
        den += float((ref ** 2).sum())
    return (num / max(den, 1e-30)) ** 0.5


def test_unet_forward_backward(mods, golden):
    g = golden("unet")
    nets, ops = mods["nets"], mods["ops"]
    for tag in ("b4", "b1"):
        m = seeded(lambda: nets.UnetGenerator(None), int(g["seed"]))
        names = [k for k, _ in m.named_parameters()]
        # bf16 emulation on the CPU
        p = {k: v.clone().requires_grad_(True) for k, v in m.state_dict().items()}
        x, cot = T(g[tag + "/x"]), T(g[tag + "/g"])
        taps = {}
        y_emu = E.unet_forward_emu(p, x, taps)
        ge = torch.autograd.grad(y_emu, [p[k] for k in names], cot, allow_unused=True)
        ge = [torch.zeros_like(p[k]) if a is None else a for k, a in zip(names, ge)]
        # engine
        m = m.cuda()
        eng = m._net_engine()
        eng.refresh()
        n, _, hw, _ = x.shape
        slot = eng.slot("t", n, hw)
        ops.image_to_c8(x.cuda(), eng.input(slot))
        eng.forward_plan(slot).run()
        y = eng.output(slot)[..., :3].float().permute(0, 3, 1, 2)
        # layer by layer against the emulation: a logic error shows as a jump at one layer, while
        # rounding differences (a stored value landing on the other side of a bf16 tie, then
        # amplified by 1/sigma of 4-pixel InstanceNorm groups) grow smoothly with depth
        errs = {k: rel_l2(slot.bufs["t." + k].float().permute(0, 3, 1, 2), v.detach()) for k, v in taps.items()}
        assert errs["conv0_0"] < 1e-3 and errs["conv0_1"] < 3e-3 and errs["conv1_0"] < 5e-3, errs
        assert max(errs.values()) < 4e-2, errs
        assert rel_l2(y, y_emu.detach()) < 3e-2, (tag, errs)         # vs emulation
        e_fp32, emu_fp32 = rel_l2(y, T(g[tag + "/y"])), rel_l2(y_emu.detach(), T(g[tag + "/y"]))
        assert e_fp32 < 5e-2 and e_fp32 < 1.25 * emu_fp32 + 1e-3, (tag, e_fp32, emu_fp32)
        gz = cot.cuda() * (1 - y * y)
        z = slot.buf("g.z", (n, hw, hw, 8))
        z.zero_()
        z[..., :3] = gz.permute(0, 2, 3, 1).to(bf16)
        eng.backward_plan(slot).run()
        torch.cuda.synchronize()
        ours = flat_grads(eng.fp, names)
        force = stored(slot, ["t." + k for k in taps] + ["up0", "up1", "up2", "up3", "noise"], 3)
        pf = {k: v.clone().requires_grad_(True) for k, v in m.state_dict().items()}
        pf = {k: v.cpu() for k, v in pf.items()}
        pf = {k: v.detach().clone().requires_grad_(True) for k, v in pf.items()}
        y_f = E.unet_forward_emu(pf, x, force=force)
        gf = torch.autograd.grad(y_f, [pf[k] for k in names], cot, allow_unused=True)
        gf = torch.cat([(torch.zeros_like(pf[k]) if a is None else a).reshape(-1) for k, a in zip(names, gf)])
        e_tf = rel_l2(ours, gf)
        assert e_tf < 4e-2, (tag, e_tf)                               # vs teacher-forced emulation
        e_fp32 = sampled_err(g, tag + "/gp", [(k, eng.fp.logical(eng.fp.grad, k).cpu()) for k in names])
        emu_fp32 = sampled_err(g, tag + "/gp", zip(names, ge))
        assert e_fp32 < 0.5 and e_fp32 < 1.6 * emu_fp32, (tag, e_fp32, emu_fp32)


def test_unet_module_call_and_empty_batch(mods):
    nets = mods["nets"]
    from oracle import combat_oracle as O
    m = seeded(lambda: nets.UnetGenerator(None), 5).cuda()
    x = (torch.rand(3, 3, 32, 32, generator=torch.Generator().manual_seed(1)) * 2 - 1)
    y = m(x.cuda())
    ref = O.unet_forward({k: v.detach().cpu().contiguous() for k, v in m.state_dict().items()}, x)
    assert rel_l2(y, ref) < 5e-2   # vs fp32 (see module docstring)
    assert tuple(m(torch.zeros(0, 3, 32, 32, device="cuda")).shape) == (0, 3, 32, 32)
    with pytest.raises(Exception):
        m(x)  # CPU tensor: no fallback path


def test_preact_train_eval(mods, golden):
    """eval (fresh stats) -> train (batch stats, running-stat update, all gradients) -> eval
    (updated stats), the sequence recorded in the golden file."""
    g = golden("preact")
    nets, ops = mods["nets"], mods["ops"]
    m = seeded(nets.PreActResNet18, int(g["seed"]))
    names = [k for k, _ in m.named_parameters()]
    pe = {k: v.clone() for k, v in m.state_dict().items()}   # emulation state (CPU)
    for k in names:
        pe[k].requires_grad_(True)
    m = m.cuda()
    eng = m._net_engine()

    def run(tag, train):
        eng.refresh()
        x, t = T(g[tag + "/x"]), T(g[tag + "/t"])
        xe = x.clone().requires_grad_(True)
        pe_before = {k: v.clone() for k, v in pe.items() if "running" in k or "num_batches" in k}
        lg_e = E.preact_forward_emu(pe, xe, train)
        loss_e = F.cross_entropy(lg_e, t)
        gr_e = torch.autograd.grad(loss_e, [xe] + [pe[k] for k in names])
        n, _, hw, _ = x.shape
        slot = eng.slot(tag, n, hw)
        ops.image_to_c8(x.cuda(), eng.input(slot))
        h = eng.head_bufs(slot)
        h["targets"].copy_(t.cuda())
        h["loss"].zero_()
        eng.forward_plan(slot, train).run()
        assert rel_l2(h["logits"], lg_e.detach()) < 2e-2, tag                     # vs emulation
        assert abs(float(h["loss"]) - float(loss_e.detach())) < 5e-3, tag
        assert rel_l2(h["logits"], T(g[tag + "/logits"])) < 3e-2, tag             # vs fp32
        assert abs(float(h["loss"]) - float(g[tag + "/loss"])) < 1e-2, tag
        # teacher-forced emulation: same stored tensors => same masks and statistics
        keys = ["stem"] + ["b%d.%s" % (b, s) for b in range(8) for s in ("y1", "out", "sc")]
        xf = x.clone().requires_grad_(True)
        pf = dict(pe)
        pf.update({k: v.clone() for k, v in pe_before.items()})   # the forced pass starts from the same buffers
        lg_f = E.preact_forward_emu(pf, xf, train, force=stored(slot, keys))
        assert rel_l2(h["logits"], lg_f.detach()) < 2e-3, tag
        gr_f = torch.autograd.grad(F.cross_entropy(lg_f, t), [xf] + [pe[k] for k in names])
        return slot, gr_e, gr_f

    def check_gx(slot, gr_e, gr_f, tag):
        eng.backward_eval_plan(slot, 1.0).run()
        gx = slot.bufs["g.img"][..., :3].float().permute(0, 3, 1, 2)
        assert rel_l2(gx, gr_f[0]) < 6e-2, tag                                    # vs teacher-forced
        e, emu = rel_l2(gx, T(g[tag + "/gx"])), rel_l2(gr_e[0], T(g[tag + "/gx"]))
        assert e < 0.5 and e < 1.6 * emu + 1e-3, (tag, e, emu)                    # vs fp32

    slot, gr_e, gr_f = run("eval0", False)
    check_gx(slot, gr_e, gr_f, "eval0")
    slot, gr_e, gr_f = run("train", True)
    eng.backward_train_plan(slot).run()
    torch.cuda.synchronize()
    ours = flat_grads(eng.fp, names)
    e_tf = rel_l2(ours, torch.cat([a.reshape(-1) for a in gr_f[1:]]))
    assert e_tf < 4e-2, e_tf                                                      # vs teacher-forced
    e = sampled_err(g, "train/gp", [(k, eng.fp.logical(eng.fp.grad, k).cpu()) for k in names])
    emu = sampled_err(g, "train/gp", zip(names, gr_e[1:]))
    assert e < 0.5 and e < 1.6 * emu, (e, emu)                                    # vs fp32
    for k, v in m.state_dict().items():  # running statistics after one train-mode forward
        if "running" in k:
            assert rel_l2(v, T(g["train/buf/" + k])) < 5e-3, k
        if "num_batches" in k:
            assert int(v) == int(g["train/buf/" + k])
    eng.fold_bn()
    slot, gr_e, gr_f = run("eval1", False)
    check_gx(slot, gr_e, gr_f, "eval1")


def test_resnet18_train_eval(mods, golden):
    """The CelebA classifier (post-activation ResNet18, 64 x 64, 8 classes): train-mode forward with batch
    statistics + all parameter gradients, then eval-mode forward (folded BatchNorms) + input gradient, as
    recorded in the golden file; tight against the bf16 emulation (teacher-forced for gradients)."""
    g = golden("resnet")
    nets, ops = mods["nets"], mods["ops"]
    mk = lambda: nets.ResNet18(num_classes=8, input_size=64)
    m = seeded(mk, int(g["seed"]))
    names = [k for k, _ in m.named_parameters()]
    pe = {k: v.clone() for k, v in m.state_dict().items()}
    for k in names:
        pe[k].requires_grad_(True)
    m = m.cuda()
    eng = m._net_engine()
    blocks = range(8)
    keys_train = ["stem", "stem.a"] + ["b%d.%s" % (b, s) for b in blocks for s in ("y1", "a1", "y2", "ys", "out")]
    keys_eval = ["stem.a"] + ["b%d.%s" % (b, s) for b in blocks for s in ("a1", "scv", "out")]

    def run(tag, train):
        eng.mark_weights_dirty()
        eng.refresh()
        x, t = T(g[tag + "/x"]), T(g[tag + "/t"])
        xe = x.clone().requires_grad_(True)
        pe_before = {k: v.clone() for k, v in pe.items() if "running" in k or "num_batches" in k}
        lg_e = E.resnet_forward_emu(pe, xe, train)
        loss_e = F.cross_entropy(lg_e, t)
        gr_e = torch.autograd.grad(loss_e, [xe] + [pe[k] for k in names], allow_unused=True)
        n, _, hw, _ = x.shape
        slot = eng.slot(tag, n, hw)
        ops.image_to_c8(x.cuda(), eng.input(slot))
        h = eng.head_bufs(slot)
        h["targets"].copy_(t.cuda())
        h["loss"].zero_()
        eng.forward_plan(slot, train).run()
        assert rel_l2(h["logits"], lg_e.detach()) < 2e-2, tag                     # vs emulation
        assert abs(float(h["loss"]) - float(loss_e.detach())) < 5e-3, tag
        assert rel_l2(h["logits"], T(g[tag + "/logits"])) < 3e-2, tag             # vs fp32
        assert abs(float(h["loss"]) - float(g[tag + "/loss"])) < 1e-2, tag
        keys = [k for k in (keys_train if train else keys_eval) if k in slot.bufs]
        xf = x.clone().requires_grad_(True)
        pf = dict(pe)
        pf.update({k: v.clone() for k, v in pe_before.items()})
        lg_f = E.resnet_forward_emu(pf, xf, train, force=stored(slot, keys))
        assert rel_l2(h["logits"], lg_f.detach()) < 2e-3, tag
        gr_f = torch.autograd.grad(F.cross_entropy(lg_f, t), [xf] + [pe[k] for k in names], allow_unused=True)
        return slot, gr_e, gr_f

    slot, gr_e, gr_f = run("train", True)
    eng.backward_train_plan(slot).run()
    torch.cuda.synchronize()
    ours = flat_grads(eng.fp, names)
    e_tf = rel_l2(ours, torch.cat([a.reshape(-1) for a in gr_f[1:]]))
    assert e_tf < 4e-2, e_tf                                                      # vs teacher-forced
    e = sampled_err(g, "train/gp", [(k, eng.fp.logical(eng.fp.grad, k).cpu()) for k in names])
    emu = sampled_err(g, "train/gp", zip(names, gr_e[1:]))
    assert e < 0.5 and e < 1.6 * emu, (e, emu)                                    # vs fp32
    for k, v in m.state_dict().items():
        if "running" in k:
            assert rel_l2(v, T(g["train/buf/" + k])) < 5e-3, k
        if "num_batches" in k:
            assert int(v) == int(g["train/buf/" + k])
    slot, gr_e, gr_f = run("eval1", False)
    eng.backward_eval_plan(slot, 1.0).run()
    gx = slot.bufs["g.img"][..., :3].float().permute(0, 3, 1, 2)
    assert rel_l2(gx, gr_f[0]) < 6e-2                                             # vs teacher-forced
    e, emu = rel_l2(gx, T(g["eval1/gx"])), rel_l2(gr_e[0], T(g["eval1/gx"]))
    assert e < 0.5 and e < 1.6 * emu + 1e-3, (e, emu)                             # vs fp32


def test_frequency_model_vs_golden(mods, golden):
    g = golden("freq")
    nets, ops = mods["nets"], mods["ops"]
    from combat_amd import trigger
    m = seeded(lambda: nets.FrequencyModel(2, 3, 32), int(g["seed"]))
    sd = m.state_dict()
    for k in sd:
        if "buf/" + k in g:
            sd[k] = T(g["buf/" + k])
    m.load_state_dict(sd)
    m = m.cuda().eval()
    eng = m._net_engine()
    eng.refresh()
    img = T(g["img"]).cuda()
    slot = eng.slot("t", img.shape[0], 32)
    ops.dct_u8(img, trigger.dct_matrix(32).float().cuda(), eng.input(slot))
    eng.forward_plan(slot).run()
    ref = T(g["logits"])
    assert rel_l2(slot.bufs["logits"], ref) < 2e-2
    assert torch.equal(slot.bufs["logits"].argmax(1).cpu(), ref.argmax(1))
    # module call signature on an already-transformed input
    assert rel_l2(m(T(g["dct_in"]).cuda()), ref) < 2e-2


class Opt:
    noise_rate, ratio, kernel_size, sigma = 0.08, 0.65, 3, (0.1, 1.0)
    pc, target_label, attack_mode, num_classes = 0.5, 0, "all2one", 10
    L2_weight, clean_model_weight, lr_C, lr_G = 0.02, 0.8, 1e-2, 1e-2
    input_height = input_width = 32
    dataset, post_transform_option, random_crop, random_rotation = "cifar10", "no_use", 5, 10


def _build(mods, seeds):
    nets = mods["nets"]
    s0, s1, s2, s3 = seeds
    netc = seeded(nets.PreActResNet18, s0)
    clean = seeded(nets.PreActResNet18, s1)
    netg = seeded(lambda: nets.UnetGenerator(None), s2)
    netf = seeded(lambda: nets.FrequencyModel(2, 3, 32), s3).eval()
    return netc, clean, netg, netf


def _oracle_state(m):
    return {k: v.detach().clone() for k, v in m.state_dict().items()}


def bench_batch(i, bs=128, rank=0, n_batches=8):
    """Batch i of bench.py's synthetic pool (same generator walk as bench.py::synth_batches and
    tests/golden/make_golden.py::bench_batch)."""
    gen = torch.Generator().manual_seed(1234 + rank)
    out = None
    for _ in range(i % n_batches + 1):
        u8 = torch.randint(0, 256, (bs, 3, 32, 32), generator=gen, dtype=torch.uint8)
        t = torch.randint(0, 10, (bs,), generator=gen)
        out = (((u8.float() / 255) - 0.5) / 0.5, t)
    return out


@pytest.mark.parametrize("with_aug,b", [(False, 16), (True, 16), (False, 128), (True, 128)])
def test_alternated_step_vs_oracle(mods, golden, with_aug, b):
    """One alternated step on the GPU against the CPU oracle with bf16-emulating networks.

    Phase C is compared from the identical start state.  Phase G is compared from the engine's
    own post-Phase-C state (netC's update already differs between two bf16 realisations by the
    mask-flip noise described in the module docstring, and eval-mode BN with day-one running
    statistics amplifies that), with the generator emulation teacher-forced.

    b = 128 is the benchmarked per-GPU batch (bench.py's own first batch and network seeds): DMA-tile
    thresholds, split reductions, XCD orders, the statistics stage-1 launch and the 128-workgroup weight
    gradients all take other branches there than at b = 16."""
    from oracle import combat_oracle as O
    from combat_amd.augment import params_from_oracle_struct
    g = golden("step" if b == 16 else "step_b128")
    step_mod, nets = mods["step"], mods["nets"]
    seeds = [int(s) for s in g["seeds"]]
    netc, clean, netg, netf = _build(mods, seeds)
    oc, ok, og, of = (_oracle_state(m) for m in (netc, clean, netg, netf))
    old_c, old_g = _oracle_state(netc), _oracle_state(netg)
    if b == 16:
        x, t = T(g["step0/inputs"]), T(g["step0/targets"])
    else:
        x, t = bench_batch(0)
        assert abs(float(x.double().sum()) - float(g["step0/x_sum"])) < 1e-6 and torch.equal(t, T(g["step0/targets"]))
    rng = np.random.default_rng(3)
    augs_o, augs_k = [None] * 5, [None] * 5
    if with_aug:
        for i in range(5):
            p = O.AugParams(rng.integers(0, 11, b).astype(np.int32), rng.integers(0, 11, b).astype(np.int32),
                            np.where(rng.random(b) < 0.5, rng.uniform(-10, 10, b), 0).astype(np.float32),
                            (rng.random(b) < 0.5).astype(np.int32))
            augs_o[i], augs_k[i] = p, params_from_oracle_struct(p)
    nb, sc, sg = int(g["num_bd"][0]), float(g["sigma_c"][0]), float(g["sigma_g"][0])
    bufs_c, bufs_g = [None] * len(O.trainable_names(oc)), [None] * len(O.trainable_names(og))
    ref = O.alternated_step(oc, og, ok, of, bufs_c, bufs_g, x, t, O.StepRandomness(nb, sc, sg, augs_o), O.StepConfig(),
                            clf_fn=E.preact_forward_emu, gen_fn=E.unet_forward_emu)
    if not with_aug:   # bf16 step vs the fp32 trace of the reference modules
        assert abs(ref["loss_c"] - float(g["trace/loss_c"][0])) < 2e-2

    opt = Opt()
    opt.post_transform_option = "use" if with_aug else "no_use"
    netc, clean, netg, netf = netc.cuda(), clean.cuda().eval(), netg.cuda(), netf.cuda().eval()
    st = step_mod.AlternatedStep(netc, netg, clean, netf, opt)
    st.keep_grads = True      # the gradient buffers are inspected below
    st.run(x.cuda(), t, step_mod.StepRandomness(nb, sc, sg, augs_k))
    torch.cuda.synchronize()
    m = st.read_metrics()
    tol = lambda r: 1e-2 * max(1.0, abs(r))

    # ---------------- Phase C (identical start state)
    assert abs(m["loss_c_sum"] - ref["loss_c"]) < tol(ref["loss_c"])
    assert abs(m["clean_model_correct"] - ref["clean_model_correct"]) <= 1
    gn_c = float(st.eC.fp.grad.double().norm())
    assert abs(gn_c - ref["gnorm_c"]) < 3e-2 * ref["gnorm_c"], (gn_c, ref["gnorm_c"])
    num = den = 0.0
    for k in O.trainable_names(oc):
        d_ref = (oc[k] - old_c[k]).double()
        d_our = (netc.state_dict()[k].detach().cpu() - old_c[k]).double()
        num += float(((d_our - d_ref) ** 2).sum())
        den += float((d_ref ** 2).sum())
    assert (num / den) ** 0.5 < 0.35, (num / den) ** 0.5      # bound only (mask flips); wiring is pinned above
    for k, v in netc.state_dict().items():
        if "running_mean" in k or "running_var" in k:
            assert rel_l2(v, oc[k]) < 1e-2, k
    # the fused optimiser applied exactly the engine's own gradient: p1 = p0 - lr*(1+mu)*(g + wd*p0)
    fp = st.eC.fp
    for k in ("conv1.weight", "layer2.0.bn1.bias", "layer4.1.conv2.weight", "linear.bias"):
        gk = fp.logical(fp.grad, k).cpu()
        exp = old_c[k] - 1e-2 * 1.9 * (gk + 5e-4 * old_c[k])
        assert rel_l2(netc.state_dict()[k].detach().cpu(), exp) < 1e-6, k

    # ---------------- Phase G (from the engine's post-Phase-C state)
    oc2 = {k: v.detach().cpu().clone() for k, v in netc.state_dict().items()}
    names_g = O.trainable_names(old_g)
    pg = {k: v.clone().requires_grad_(k in names_g) for k, v in old_g.items()}
    keys = ["t." + n for n, *_ in nets.UNET_LAYERS] + ["up0", "up1", "up2", "up3", "noise"]
    noise = E.unet_forward_emu(pg, x, force=stored(st.sG, keys, 3))
    ibd = O.trigger_mix(x, noise, 0.08, 0.65, sg)
    assert float((st.bd.cpu() - ibd.detach()).abs().max()) < 3e-5
    bd_t = torch.zeros_like(t)
    leaf = ibd.detach().clone().requires_grad_(True)
    pred_bd = E.preact_forward_emu(oc2, O.post_tensor_transform(leaf, augs_o[3]), False)
    cm_pred = E.preact_forward_emu(ok, O.post_tensor_transform(leaf, augs_o[4]), False)
    loss_ce, cm_loss = F.cross_entropy(pred_bd, bd_t), F.cross_entropy(cm_pred, t)
    assert abs(m["loss_ce_sum"] - float(loss_ce.detach())) < tol(float(loss_ce.detach()))
    assert abs(m["clean_model_loss_sum"] - float(cm_loss.detach())) < tol(float(cm_loss.detach()))
    l2 = float(F.mse_loss(ibd.detach(), x))
    assert abs(m["loss_l2_sum"] - l2) < 1e-3 * l2 + 1e-7
    assert abs(m["loss_grad_l2_sum"] - ref["loss_grad_l2"]) < 5e-2 * ref["loss_grad_l2"] + 1e-6
    assert abs(m["bd_correct"] - int((pred_bd.argmax(1) == bd_t).sum())) <= 1
    assert abs(m["clean_model_bd_ba"] - int((cm_pred.argmax(1) == t).sum())) <= 1
    assert abs(m["clean_model_bd_asr"] - int((cm_pred.argmax(1) == bd_t).sum())) <= 1
    (d_bd,) = torch.autograd.grad(loss_ce + 0.8 * cm_loss, leaf)
    assert rel_l2((st.d_bd + st.d_bd2).cpu(), d_bd) < 0.25            # un-forced classifiers: mask-flip bound
    total = (ibd * (st.d_bd + st.d_bd2).cpu()).sum() + 0.02 * F.mse_loss(ibd, x)   # engine's own image gradient (both classifiers' shares) as cotangent
    gr = torch.autograd.grad(total, [pg[k] for k in names_g], allow_unused=True)
    gr = torch.cat([(torch.zeros_like(pg[k]) if a is None else a).reshape(-1) for k, a in zip(names_g, gr)])
    assert rel_l2(flat_grads(st.eG.fp, names_g), gr) < 5e-2    # teacher-forced: pins trigger bwd + UNet bwd
    fp = st.eG.fp
    for k in ("conv0_0.weight", "conv3_1.weight", "upconv0_0.bias", "upconv1_0.bias"):
        gk = fp.logical(fp.grad, k).cpu()
        exp = old_g[k] - 1e-2 * 1.9 * (gk + 5e-4 * old_g[k])
        assert rel_l2(netg.state_dict()[k].detach().cpu(), exp) < 1e-6, k

    if not with_aug:
        # ---------------- the HIP step against the fp32 trace recorded from the REFERENCE modules (first step;
        # Phase G there starts from the reference's own fp32 Phase-C update, so this also bounds how far the
        # bf16 realisation of one netC update moves the Phase-G losses): 2e-2 * max(1, |ref|), loss_l2 2 % rel.
        for ours, key in (("loss_c_sum", "loss_c"), ("loss_ce_sum", "loss_ce"), ("clean_model_loss_sum", "clean_model_loss")):
            r = float(g["trace/" + key][0])
            assert abs(m[ours] - r) < 2e-2 * max(1.0, abs(r)), (key, m[ours], r)
        r = float(g["trace/loss_l2"][0])
        assert abs(m["loss_l2_sum"] - r) < 2e-2 * r, (m["loss_l2_sum"], r)
        for ours, key in (("clean_model_correct", "clean_model_correct"), ("clean_model_bd_ba", "clean_model_bd_ba"),
                          ("clean_model_bd_asr", "clean_model_bd_asr"), ("f_correct", "f_correct")):
            assert abs(m[ours] - float(g["trace/" + key][0])) <= max(2, 0.03 * b), (key, m[ours], float(g["trace/" + key][0]))


def test_backward_plans_do_not_share_weight_gradient_scratch(mods):
    """the last weight gradient of a backward plan runs on the plan's own stream while the
    previous ones may still be writing / reducing their partial-sum slabs on the auxiliary stream; they used
    to share one scratch base.  At N = 128 (where every 3x3 weight gradient takes the slab path) the
    multi-stream backward must give, per tensor, what the in-line (one stream) replay gives."""
    nets, ops, engine = mods["nets"], mods["ops"], mods["engine"]
    gen = torch.Generator().manual_seed(4)
    x = torch.rand(128, 3, 32, 32, generator=gen) * 2 - 1
    t = torch.randint(0, 10, (128,), generator=gen)
    # ---- classifier: stem.wgrad is the aux=False launch
    m = seeded(nets.PreActResNet18, 0).cuda()
    eng = m._net_engine()
    eng.refresh()
    slot = eng.slot("race", 128, 32)
    ops.image_to_c8(x.cuda(), eng.input(slot))
    eng.head_bufs(slot)["targets"].copy_(t.cuda())
    eng.forward_plan(slot, True).run()
    bwd = eng.backward_train_plan(slot)
    names = [k for k, _ in m.named_parameters()]

    def grads(fp, names, run):
        run()
        torch.cuda.synchronize()
        return {k: fp.logical(fp.grad, k).detach().float().cpu().clone() for k in names}

    engine.Plan.serial = True
    try:
        ref = grads(eng.fp, names, bwd.run)
    finally:
        engine.Plan.serial = False
    for rep in range(4):
        got = grads(eng.fp, names, bwd.run)
        for k in names:
            assert rel_l2(got[k], ref[k]) < 1e-5, ("preact", rep, k, rel_l2(got[k], ref[k]))
    # ---- generator: conv0_0.wgrad is the aux=False launch
    gm = seeded(lambda: nets.UnetGenerator(None), 2).cuda()
    eg = gm._net_engine()
    eg.refresh()
    sg = eg.slot("race", 128, 32)
    ops.image_to_c8(x.cuda(), eg.input(sg))
    eg.forward_plan(sg).run()
    z = sg.buf("g.z", (128, 32, 32, 8))
    z.zero_()
    z[..., :3] = (torch.randn(128, 32, 32, 3, generator=gen) * 1e-2).to(bf16).cuda()
    gb = eg.backward_plan(sg)
    gnames = [k for k, _ in gm.named_parameters()]
    engine.Plan.serial = True
    try:
        ref = grads(eg.fp, gnames, gb.run)
    finally:
        engine.Plan.serial = False
    for rep in range(4):
        got = grads(eg.fp, gnames, gb.run)
        for k in gnames:
            assert rel_l2(got[k], ref[k]) < 1e-5 or float(ref[k].abs().max()) == 0.0, ("unet", rep, k, rel_l2(got[k], ref[k]))
    # ---- the same plan dealt to TWO auxiliary queues (each queue has its own scratch): weight gradients of
    # neighbouring layers then run concurrently
    engine.Plan.default_aux_queues = 2
    try:
        sg2 = eg.slot("race2", 128, 32)
        ops.image_to_c8(x.cuda(), eg.input(sg2))
        eg.forward_plan(sg2).run()
        sg2.buf("g.z", (128, 32, 32, 8)).copy_(z)
        gb2 = eg.backward_plan(sg2)
        assert gb2.aux_queues == 2 and set(gb2.aux.values()) == {0, 1}
        for rep in range(3):
            got = grads(eg.fp, gnames, gb2.run)
            for k in gnames:
                assert rel_l2(got[k], ref[k]) < 1e-5 or float(ref[k].abs().max()) == 0.0, ("unet 2 queues", rep, k, rel_l2(got[k], ref[k]))
    finally:
        engine.Plan.default_aux_queues = 1


def test_deferred_weight_gradient_reductions(mods):
    """combat_wgrad_args.defer_reduce + combat_conv_wgrad_reduce + combat_plan_set_after (opt-in: COMBAT_DEFER_REDUCE=1;
    measured slower on the step, see engine.Plan.defer_reduces): a training backward plan whose slab reductions run
    on the plan's own stream, each behind its weight-gradient launch on the auxiliary queue, gives the gradients of
    the default plan (reduction right behind each kernel) -- in C replay, Python replay and in line."""
    nets, ops, engine = mods["nets"], mods["ops"], mods["engine"]
    from combat_amd._lib import lib
    gen = torch.Generator().manual_seed(4)
    x = torch.rand(128, 3, 32, 32, generator=gen) * 2 - 1
    t = torch.randint(0, 10, (128,), generator=gen)
    m = seeded(nets.PreActResNet18, 0).cuda()
    eng = m._net_engine()
    eng.refresh()
    names = [k for k, _ in m.named_parameters()]

    def build(tag, defer):
        engine.Plan.defer_reduces = defer
        try:
            slot = eng.slot(tag, 128, 32)
            ops.image_to_c8(x.cuda(), eng.input(slot))
            eng.head_bufs(slot)["targets"].copy_(t.cuda())
            eng.forward_plan(slot, True).run()
            return eng.backward_train_plan(slot)
        finally:
            engine.Plan.defer_reduces = False

    def grads(run):
        run()
        torch.cuda.synchronize()
        return {k: eng.fp.logical(eng.fp.grad, k).detach().float().cpu().clone() for k in names}

    ref_plan, plan = build("defer.ref", False), build("defer.on", True)
    engine.REDUCE_BEHIND = True          # opt-in (COMBAT_REDUCE_BEHIND=1): the reductions ride in the next weight gradient
    try:
        behind = build("defer.behind", False)
    finally:
        engine.REDUCE_BEHIND = False
    n_red = sum(1 for f, _, _ in plan.calls if f is lib.combat_conv_wgrad_reduce)
    assert n_red >= 12 and len(plan.after) == n_red and not any(f is lib.combat_conv_wgrad_reduce for f, _, _ in ref_plan.calls)
    # reduce-behind (round 4, opt-in): the slab-leaving launches form ONE chain per plan -- each carries its
    # predecessor (reduce_first), only the last one gets a reduction call
    chained = [a for _, a in behind.wgrads if a.defer_reduce]
    assert len(chained) == n_red and sum(1 for a in chained if a.reduce_first) == n_red - 1
    assert sum(1 for f, _, _ in behind.calls if f is lib.combat_conv_wgrad_reduce) == 1
    ref = grads(ref_plan.run)
    first = grads(behind.run)
    for k in names:
        assert rel_l2(first[k], ref[k]) < 1e-5, ("reduce-behind", k, rel_l2(first[k], ref[k]))
    for mode in ("c", "py", "serial"):
        engine.Plan.compiled, engine.Plan.serial = mode != "py", mode == "serial"
        try:
            got = grads(plan.run)
        finally:
            engine.Plan.compiled, engine.Plan.serial = True, False
        for k in names:
            assert rel_l2(got[k], ref[k]) < 1e-5, (mode, k, rel_l2(got[k], ref[k]))


def test_plan_replay_in_c_equals_python_replay(mods):
    """combat_plan_run (csrc/plan.cpp: the launch list walked in C, hand-off events reused) against the Python replay
    of the same plans.  Deterministic plans (the eval-mode forward: no atomics anywhere) must give BIT-identical
    tensors; one whole alternated step from identical states -- whose weight gradients of the stride-2 / 1x1 / 3-channel
    layers use fp32 atomics, so two replays of EITHER kind differ in the last bits -- must agree as closely as two
    Python replays agree with each other (and to 1e-5), with the auxiliary weight-gradient queue in play and in line."""
    engine, step_mod, nets, ops = mods["engine"], mods["step"], mods["nets"], mods["ops"]
    opt = Opt()
    # ---- deterministic plan: bit-identical
    m = seeded(nets.PreActResNet18, 0).cuda().eval()
    eng = m._net_engine()
    eng.refresh()
    x, t = bench_batch(0, 32)
    slot = eng.slot("creplay", 32, 32)
    ops.image_to_c8(x.cuda(), eng.input(slot))
    eng.head_bufs(slot)["targets"].copy_(t.cuda())
    plan = eng.forward_plan(slot, False)
    outs = []
    for compiled in (False, True, True):
        engine.Plan.compiled = compiled
        try:
            plan.run()
            torch.cuda.synchronize()
        finally:
            engine.Plan.compiled = True
        outs.append({k: v.clone() for k, v in slot.bufs.items() if v.dtype in (bf16, torch.float32) and k != "loss"})
    assert plan._cplan and len(plan._prog_nomark) == 1 and plan._prog_nomark[0][0] == "c"   # one foreign call
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]) and torch.equal(outs[1][k], outs[2][k]), k

    # ---- one whole step
    def run(compiled, serial):
        engine.Plan.compiled = compiled
        engine.Plan.serial = serial
        try:
            netc, clean, netg, netf = (mm.cuda() for mm in _build(mods, (0, 1, 2, 3)))
            clean.eval()
            st = step_mod.AlternatedStep(netc, netg, clean, netf, opt)
            st.serial = serial
            xb, tb = bench_batch(1, 32)
            st.run(xb.cuda(), tb, step_mod.StepRandomness(3, 0.4, 0.7, [None] * 5))
            torch.cuda.synchronize()
            state = {k: v.detach().float().clone() for mm in (netc, netg) for k, v in mm.state_dict().items()}
            return state, st.read_metrics()
        finally:
            engine.Plan.compiled, engine.Plan.serial = True, False

    def dist(a, b):
        return max(rel_l2(a[0][k].cpu(), b[0][k].cpu()) for k in a[0] if a[0][k].numel() > 1)

    for serial in (False, True):
        py1, py2, c1 = run(False, serial), run(False, serial), run(True, serial)
        noise = dist(py1, py2)
        # (two samples underestimate the noise: a single tensor whose fp32-atomic reduction happened to run in another
        # order moves by ~5e-7 -- seen between two replays of EITHER kind; real ordering bugs move results by >= 1e-3)
        assert dist(py1, c1) <= max(3 * noise, 2e-6) and dist(py1, c1) < 1e-5, (serial, noise, dist(py1, c1))
        for k in ("clean_correct", "bd_correct", "train_correct", "clean_model_correct"):
            assert py1[1][k] == c1[1][k]
        for k in ("loss_c_sum", "loss_ce_sum", "clean_model_loss_sum", "loss_l2_sum"):
            assert abs(py1[1][k] - c1[1][k]) <= 1e-5 * max(1.0, abs(py1[1][k])), (k, py1[1][k], c1[1][k])


def test_plan_replay_reports_the_failing_call(mods):
    """A recorded call whose arguments the entry point rejects surfaces as CombatHipError naming the call, as in the
    Python replay."""
    engine, lib_mod = mods["engine"], __import__("combat_amd._lib", fromlist=["lib"])
    P = engine.Plan("probe")
    buf = torch.zeros(64, device="cuda")
    P.hold(buf)
    P.add("ok", lib_mod.lib.combat_memset_zero, buf.data_ptr(), 256)
    P.add("bad", lib_mod.lib.combat_memset_zero, None, -4)
    with pytest.raises(lib_mod.CombatHipError, match="probe/bad"):
        P.run()


def _ema(v, a=0.1):
    out, m = [], float(v[0])
    for x in v:
        m = (1 - a) * m + a * float(x)
        out.append(m)
    return np.array(out)


@pytest.mark.parametrize("name,steps", [("trajectory_lr2e3", 100), ("trajectory", 40), ("trajectory_celeba", 40)])
def test_trajectory_vs_reference_trace(mods, golden, name, steps):
    trajectory_check(mods, golden, name, steps)


def trajectory_check(mods, golden, name, steps, pinned=None):
    """(pinned: tests/test_deterministic_gpu.py -- in deterministic mode a run is reproducible, so the bounds can sit just
    above what THE run measures instead of covering the run-to-run spread: dict(ema={loss: relative bound}, counter=max
    per-step deviation in images, total=relative bound on the counters' sums).)

    SURVEY 8(d) / north_star "loss curves must match": alternated steps at B = 32 over a pool of 25 fixed
    synthetic batches (augmentation off, recorded num_bd / sigma per step) against the trace the REFERENCE's
    own modules + torch.optim.SGD produced in fp32 (tests/golden/make_golden.py::golden_trajectory*).

    `trajectory_lr2e3` (--lr_C 2e-3 --lr_G 2e-3), 100 steps: exponential moving averages (alpha 0.1) of loss_c,
    clean_model_loss and loss_l2 within 2 % of the reference's at EVERY step; loss_ce -- whose raw curve
    alternates between ~0 and ~2 with the recorded blur sigma of the step -- within 2 % + 0.02 up to step 60 and
    10 % + 0.02 after; per-step (to step 60; 10 images after)
    counters within 4 images of 32 and their 100-step totals within 1.5 %.  (The fp32 oracle driven with the
    CPU bf16 emulation, the idealised form of this design, measures 0.23 % / 0 % / 0.14 % and 7 % on loss_ce's
    EMA at step 99 = 0.05 absolute.)
    `trajectory` (the default lr 1e-2): the reference's own run turns chaotic after ~45 steps on these
    random-label batches (loss_ce jumps between 0 and 50-290), so only its smooth first 40 steps are compared:
    loss_c / clean_model_loss / loss_l2 EMAs within 5 %, loss_ce within 5 % + 0.03.
    `trajectory_celeba` (BASELINE config 4's shape: 3 x 64 x 64, 8 classes, ResNet18 surrogate and clean model, B = 16,
    lr 2e-3, 40 steps -- the surrogate memorises the 160 images, loss_c 2.08 -> 0.03, and the trigger wins within 20 steps,
    loss_ce -> 1e-5): loss_c / clean_model_loss / loss_l2 EMAs within 5 %, loss_ce within 5 % + 0.03."""
    step_mod, nets = mods["step"], mods["nets"]
    g = golden(name)
    seeds = [int(s) for s in g["seeds"]]
    celeba = name == "trajectory_celeba"
    opt = Opt()
    if celeba:
        mk = lambda: nets.ResNet18(num_classes=8, input_size=64)
        netc, clean = seeded(mk, seeds[0]), seeded(mk, seeds[1])
        netg = seeded(lambda: nets.UnetGenerator(None), seeds[2])
        netf = seeded(lambda: nets.FrequencyModel(2, 3, 64), seeds[3]).eval()
        opt.num_classes, opt.input_height, opt.input_width, opt.dataset = 8, 64, 64, "celeba"
    else:
        netc, clean, netg, netf = _build(mods, seeds)
    st = step_mod.AlternatedStep(netc.cuda(), netg.cuda(), clean.cuda().eval(), netf.cuda().eval(), opt)
    b, pool_n, lr = (16 if celeba else 32), int(g["pool"]), float(g["lr"])
    hw, ncls = opt.input_height, opt.num_classes
    s_img, s_lab = (int(v) for v in g["pool_seeds"])

    def synth(i):
        u8 = torch.randint(0, 256, (b, 3, hw, hw), generator=torch.Generator().manual_seed(s_img + i), dtype=torch.uint8)
        return (u8.float() / 255 - 0.5) / 0.5, torch.randint(0, ncls, (b,), generator=torch.Generator().manual_seed(s_lab + i))

    pool = [synth(i) for i in range(pool_n)]
    keys = (("loss_c_sum", "loss_c"), ("loss_ce_sum", "loss_ce"), ("clean_model_loss_sum", "clean_model_loss"),
            ("loss_l2_sum", "loss_l2"), ("clean_correct", "clean_correct"), ("bd_correct", "bd_correct"),
            ("clean_model_correct", "clean_model_correct"), ("clean_model_bd_ba", "clean_model_bd_ba"),
            ("clean_model_bd_asr", "clean_model_bd_asr"), ("train_correct", "train_correct"))
    ours = {k: [] for _, k in keys}
    for s in range(steps):
        x, t = pool[s % pool_n]
        st.reset_metrics()
        st.run(x.cuda(), t, step_mod.StepRandomness(int(g["num_bd"][s]), float(g["sigma_c"][s]), float(g["sigma_g"][s]),
                                                    [None] * 5), lr_c=lr, lr_g=lr)
        m = st.read_metrics()
        for mk, k in keys:
            ours[k].append(m[mk])
    tight = name == "trajectory_lr2e3"
    rel = 0.02 if tight else 0.05
    report = {}
    for k, floor in (("loss_c", 0.0), ("clean_model_loss", 0.0), ("loss_l2", 0.0), ("loss_ce", 0.02 if tight else 0.03)):
        e_o, e_r = _ema(ours[k]), _ema(g["trace/" + k][:steps])
        dev = np.abs(e_o - e_r) - floor
        worst = int(np.argmax(dev / np.maximum(np.abs(e_r), 1e-9)))
        report[k] = (float((dev / np.maximum(np.abs(e_r), 1e-9)).max()), worst)
        tol = np.full(steps, rel)
        if pinned is not None:
            tol[:] = pinned["ema"][k]
        elif k == "loss_ce" and tight:
            tol[60:] = 0.10      # past step 60 loss_ce alternates 0 <-> 2 with the step's blur sigma and small
            #                      differences are amplified: the CPU bf16 emulation itself is 7 % off fp32 at step 99,
            #                      HIP runs measured 3-6 % (they differ run to run: fp32 atomics reorder sums)
        assert np.all(dev <= tol * np.abs(e_r)), (name, k, report[k], e_o[worst], e_r[worst])
    print(name, "max relative EMA deviations (value, step):", report)
    for k in ("clean_correct", "bd_correct", "clean_model_correct", "clean_model_bd_ba", "clean_model_bd_asr", "train_correct"):
        o, r = np.array(ours[k], dtype=np.float64), g["trace/" + k][:steps]
        # per-step counters of 32 images: within 4 of the reference's on >= 95 % of the first 60 steps and never more
        # than 10 off.  bd_correct sits at 0 or 32 and crosses over within two or three steps (steps 12-13, 40-45 of the
        # lr 2e-3 trace); where exactly a run crosses depends on its summation order (fp32 atomics in the weight
        # gradients), so a single step of a crossing can be 5-8 images off in one run and exact in the next.
        # (past step 60 bd_correct follows the spiky loss_ce, see above)
        d = np.abs(o - r)
        near, far = max(2, b // 8), max(4, 10 * b // 32)       # 4 and 10 images at B = 32
        if pinned is not None:
            far = pinned["counter"]
        print("%s %-20s max |ours - reference| %2d (step %d), first 60 steps: %.1f %% within %d, totals %d vs %d" % (
            name, k, int(d.max()), int(d.argmax()), 100.0 * (d[: min(60, steps)] <= near).mean(), near, int(o.sum()), int(r.sum())))
        assert np.all(d <= far), (name, k, int(d.argmax()), d.max())
        head = d[: min(60, steps)]
        assert (head <= near).mean() >= 0.95, (name, k, np.nonzero(head > near)[0].tolist(), head.max())
        assert abs(o.sum() - r.sum()) <= (pinned["total"] * steps * b if pinned is not None else max(0.015 * steps * b, 8)), (name, k, o.sum(), r.sum())


@pytest.mark.parametrize("b", [8, 128])
def test_alternated_step_celeba_shape_resnet18(mods, b):
    """BASELINE config 4's shape (CelebA: 3 x 64 x 64, 8 classes, ResNet18 surrogate and clean model, UNet at
    64 x 64), at a toy batch and at the configuration's own per-GPU batch of 128 (where DMA-tile thresholds, split
    reductions and the statistics stage-1 launches take other branches): one alternated step against the CPU oracle
    driven with the bf16-emulating networks, with the CIFAR test's method and tolerances -- Phase C from the identical
    start state (loss 1e-2, gradient norm 3e-2, running statistics 1e-2, optimiser identity 1e-6), Phase G from the
    engine's own post-Phase-C classifier with the generator emulation teacher-forced (losses 1e-2 * max(1, |ref|),
    loss_l2 1e-3, counters +-1, generator gradient 5e-2)."""
    from oracle import combat_oracle as O
    step_mod, nets = mods["step"], mods["nets"]
    mk = lambda: nets.ResNet18(num_classes=8, input_size=64)
    netc, clean = seeded(mk, 11), seeded(mk, 12)
    netg = seeded(lambda: nets.UnetGenerator(None), 13)
    netf = seeded(lambda: nets.FrequencyModel(2, 3, 64), 14).eval()
    oc, ok, og, of = (_oracle_state(m) for m in (netc, clean, netg, netf))
    old_c, old_g = _oracle_state(netc), _oracle_state(netg)
    gen = torch.Generator().manual_seed(5)
    x = ((torch.randint(0, 256, (b, 3, 64, 64), generator=gen, dtype=torch.uint8).float() / 255) - 0.5) / 0.5
    t = torch.randint(0, 8, (b,), generator=gen)
    t[::2][:4] = 0
    nb, sc, sg = 2, 0.6, 0.9
    cfg = O.StepConfig(num_classes=8, classifier="resnet18")
    bufs_c, bufs_g = [None] * len(O.trainable_names(oc)), [None] * len(O.trainable_names(og))
    ref = O.alternated_step(oc, og, ok, of, bufs_c, bufs_g, x, t, O.StepRandomness(nb, sc, sg, [None] * 5), cfg,
                            clf_fn=E.resnet_forward_emu, gen_fn=E.unet_forward_emu)

    opt = Opt()
    opt.num_classes, opt.input_height, opt.input_width, opt.dataset = 8, 64, 64, "celeba"
    netc, clean, netg, netf = netc.cuda(), clean.cuda().eval(), netg.cuda(), netf.cuda().eval()
    st = step_mod.AlternatedStep(netc, netg, clean, netf, opt)
    st.keep_grads = True      # the gradient buffers are inspected below
    st.run(x.cuda(), t, step_mod.StepRandomness(nb, sc, sg, [None] * 5))
    torch.cuda.synchronize()
    m = st.read_metrics()
    assert all(np.isfinite(v) for v in m.values())
    tol = lambda r: 1e-2 * max(1.0, abs(r))
    # ---------------- Phase C (identical start state)
    assert abs(m["loss_c_sum"] - ref["loss_c"]) < tol(ref["loss_c"])
    assert abs(m["clean_model_correct"] - ref["clean_model_correct"]) <= 1
    gn_c = float(st.eC.fp.grad.double().norm())
    assert abs(gn_c - ref["gnorm_c"]) < 3e-2 * ref["gnorm_c"], (gn_c, ref["gnorm_c"])
    num = den = 0.0
    for k in O.trainable_names(oc):
        d_ref = (oc[k] - old_c[k]).double()
        d_our = (netc.state_dict()[k].detach().cpu() - old_c[k]).double()
        num += float(((d_our - d_ref) ** 2).sum())
        den += float((d_ref ** 2).sum())
    assert (num / den) ** 0.5 < 0.35, (num / den) ** 0.5      # bound only (mask flips); wiring is pinned by the module tests
    for k, v in netc.state_dict().items():
        if "running_mean" in k or "running_var" in k:
            assert rel_l2(v, oc[k]) < 1e-2, k
    fp = st.eC.fp
    for k in ("conv1.weight", "layer2.0.bn1.bias", "layer4.1.conv2.weight", "linear.bias"):
        gk = fp.logical(fp.grad, k).cpu()
        exp = old_c[k] - 1e-2 * 1.9 * (gk + 5e-4 * old_c[k])
        assert rel_l2(netc.state_dict()[k].detach().cpu(), exp) < 1e-6, k
    # ---------------- Phase G (from the engine's post-Phase-C state)
    oc2 = {k: v.detach().cpu().clone() for k, v in netc.state_dict().items()}
    names_g = O.trainable_names(old_g)
    pg = {k: v.clone().requires_grad_(k in names_g) for k, v in old_g.items()}
    keys = ["t." + n for n, *_ in nets.UNET_LAYERS] + ["up0", "up1", "up2", "up3", "noise"]
    noise = E.unet_forward_emu(pg, x, force=stored(st.sG, keys, 3))
    ibd = O.trigger_mix(x, noise, 0.08, 0.65, sg)
    assert float((st.bd.cpu() - ibd.detach()).abs().max()) < 3e-5
    bd_t = torch.zeros_like(t)
    leaf = ibd.detach().clone().requires_grad_(True)
    pred_bd = E.resnet_forward_emu(oc2, leaf, False)
    cm_pred = E.resnet_forward_emu(ok, leaf, False)
    loss_ce, cm_loss = F.cross_entropy(pred_bd, bd_t), F.cross_entropy(cm_pred, t)
    assert abs(m["loss_ce_sum"] - float(loss_ce.detach())) < tol(float(loss_ce.detach())), (m["loss_ce_sum"], float(loss_ce.detach()))
    assert abs(m["clean_model_loss_sum"] - float(cm_loss.detach())) < tol(float(cm_loss.detach()))
    l2 = float(F.mse_loss(ibd.detach(), x))
    assert abs(m["loss_l2_sum"] - l2) < 1e-3 * l2 + 1e-7
    assert abs(m["loss_grad_l2_sum"] - ref["loss_grad_l2"]) < 5e-2 * ref["loss_grad_l2"] + 1e-6
    assert abs(m["bd_correct"] - int((pred_bd.argmax(1) == bd_t).sum())) <= 1
    assert abs(m["clean_model_bd_ba"] - int((cm_pred.argmax(1) == t).sum())) <= 1
    assert abs(m["clean_model_bd_asr"] - int((cm_pred.argmax(1) == bd_t).sum())) <= 1
    (d_bd,) = torch.autograd.grad(loss_ce + 0.8 * cm_loss, leaf)
    assert rel_l2((st.d_bd + st.d_bd2).cpu(), d_bd) < 0.25            # un-forced classifiers: mask-flip bound
    total = (ibd * (st.d_bd + st.d_bd2).cpu()).sum() + 0.02 * F.mse_loss(ibd, x)
    gr = torch.autograd.grad(total, [pg[k] for k in names_g], allow_unused=True)
    gr = torch.cat([(torch.zeros_like(pg[k]) if a is None else a).reshape(-1) for k, a in zip(names_g, gr)])
    assert rel_l2(flat_grads(st.eG.fp, names_g), gr) < 5e-2    # teacher-forced: pins trigger bwd + UNet bwd at 64 x 64


def test_alternated_step_runs_with_sampled_randomness_and_empty_poison(mods):
    """Default path: randomness drawn on the host as the reference does; also num_bd == 0 and a
    ragged last batch (B=80 -> here 12) must work (train_generator.py:190-194)."""
    step_mod = mods["step"]
    netc, clean, netg, netf = _build(mods, [0, 1, 2, 3])
    opt = Opt()
    opt.post_transform_option = "use"
    st = step_mod.AlternatedStep(netc.cuda(), netg.cuda(), clean.cuda().eval(), netf.cuda().eval(), opt)
    gen = torch.Generator().manual_seed(0)
    for b in (16, 12):
        x = (torch.randint(0, 256, (b, 3, 32, 32), generator=gen).float() / 255 - 0.5) / 0.5
        t = torch.randint(1, 10, (b,), generator=gen)   # no target-class image: num_bd is 0
        st.run(x.cuda(), t)
        t[:5] = 0
        st.run(x.cuda(), t)
    torch.cuda.synchronize()
    m = st.read_metrics(reset=True)
    assert m["samples"] == 2 * (16 + 12)
    assert all(np.isfinite(v) for v in m.values())
    for p in list(netc.parameters()) + list(netg.parameters()):
        assert torch.isfinite(p).all()
    # streams are process-wide, not per step object (a second object must not add hardware queues: see step.shared_stream)
    st2 = step_mod.AlternatedStep(netc, netg, clean, netf, opt)
    st2.run(x.cuda(), t)
    torch.cuda.synchronize()
    assert st2._side is st._side and st._side is not None


def test_preact_train_forward_with_lds_prologue_equals_the_chain(mods):
    """COMBAT_FUSED_PROLOGUE=1 (engine.FUSED_PROLOGUE): PreActResNet18's train forward with relu(bn(.)) applied in LDS
    by the consuming convolutions (conv3x3_dma_pro_kernel + combat_norm_finalize) against the default chain
    (combat_norm_act_fused + prologue-free convolution).  At B = 128 both run the same tiles on every layer (the
    weight-stationary kernel of the chain's 64-channel layers is bit-identical to the ring tile), so every stored
    tensor of the forward -- raw outputs, activated tensors, statistics, running statistics, loss -- is BIT-identical
    (both modes recorded with the ring tiles: the persistent kernel groups its statistics rows per workgroup);
    the backward reads the same tensors (gradient: fp32-atomic noise of a few weight-gradient launches only).  The
    opt-in path is slower (engine.py) and stays off; this keeps it correct."""
    engine, nets = mods["engine"], mods["nets"]
    b = 128
    res = {}
    prev = engine.FUSED_PROLOGUE
    try:
        for fused in (False, True):
            engine.FUSED_PROLOGUE = fused
            net = seeded(nets.PreActResNet18, 0).cuda()
            eng = net._net_engine()
            eng.refresh()
            slot = eng.slot("pro.%d" % fused, b, 32)
            x, t = bench_batch(0)
            ops_ = mods["ops"]
            xd = x.cuda()
            ops_.image_to_c8(xd, eng.input(slot))
            eng.head_bufs(slot)["targets"].copy_(t)
            with engine.short_workgroups():       # (ring tiles in both modes: the persistent kernel groups its statistics rows differently)
                fwd, bwd = eng.forward_plan(slot, True), eng.backward_train_plan(slot)
            names = [w for _, _, w in fwd.calls]
            assert any(w.endswith(".finalize") for w in names) == fused
            assert sum(w.endswith(".finact") for w in names) == (3 if fused else 16)      # the stride-2 blocks' inputs stay materialised
            fwd.run()
            bwd.run()
            torch.cuda.synchronize()
            res[fused] = ({k: v.clone() for k, v in slot.bufs.items() if not k.startswith("g.")}, eng.fp.grad.clone(),
                          {k: v.detach().clone() for k, v in net.state_dict().items() if "running" in k or "num_batches" in k})
    finally:
        engine.FUSED_PROLOGUE = prev
    (ba, ga, ra), (bb, gb, rb_) = res[False], res[True]
    for k in sorted(ba):
        if k == "loss":      # (summed over samples with an fp32 atomic: last-bit noise between any two runs)
            assert abs(float(ba[k]) - float(bb[k])) < 1e-5
        elif k in bb and ba[k].shape == bb[k].shape and not k.endswith(".part"):     # (row counts differ: one row per persistent workgroup / per tile)
            assert torch.equal(ba[k], bb[k]), k
    for k in ra:
        assert torch.equal(ra[k], rb_[k]), k
    assert rel_l2(gb, ga) < 1e-5


def test_fused_head_plans_equal_the_two_launch_plans(mods):
    """PreActEngine.forward_plan(head_bwd=True) + backward_*_plan(head_done=True) (COMBAT_FUSED_HEAD=1 in the steps): the head's
    forward and feature gradient as one launch, the linear layer's weight gradient on the auxiliary queue -- the feature
    gradient bit-identical, parameter gradients equal up to the atomic sums of a few weight-gradient launches, the
    image gradient of the eval pass bit-identical."""
    nets, ops_ = mods["nets"], mods["ops"]
    net = seeded(nets.PreActResNet18, 0).cuda()
    eng = net._net_engine()
    eng.refresh()
    x, t = bench_batch(0, bs=32)
    out = {}
    for fused in (False, True):
        slot = eng.slot("head.%d" % fused, 32, 32)
        ops_.image_to_c8(x.cuda(), eng.input(slot))
        eng.head_bufs(slot)["targets"].copy_(t)
        kw_f, kw_b = (dict(head_bwd=True), dict(head_done=True)) if fused else ({}, {})
        fwd, bwd = eng.forward_plan(slot, True, **kw_f), eng.backward_train_plan(slot, **kw_b)
        assert sum("head" in w for _, _, w in fwd.calls) == 1 and any(w == "head_bwd.w" for _, _, w in bwd.calls) == fused
        fwd.run()
        bwd.run()
        torch.cuda.synchronize()
        res = [slot.bufs["g.feat"].clone(), eng.fp.grad.clone(), float(eng.head_bufs(slot)["loss"])]
        eslot = eng.slot("head.e%d" % fused, 32, 32)
        ops_.image_to_c8(x.cuda(), eng.input(eslot))
        eng.head_bufs(eslot)["targets"].copy_(t)
        eng.forward_plan(eslot, False, 1.0, False, **kw_f).run()
        eng.backward_eval_plan(eslot, 1.0, **kw_b).run()
        torch.cuda.synchronize()
        out[fused] = res + [eslot.bufs["g.img"].clone()]
    assert torch.equal(out[False][0], out[True][0]) and torch.equal(out[False][3], out[True][3])
    assert rel_l2(out[True][1], out[False][1]) < 1e-5 and abs(out[True][2] - out[False][2]) < 1e-5


@pytest.mark.parametrize("switch", ["merge_c_eval", "fused_head", "reduce_behind", "fused_prologue"])
def test_step_ab_switches_keep_the_results(mods, switch):
    """Round 4's opt-in forms of the step (all measured no faster; kept selectable): netC's two Phase-G
    eval forwards as one 2n pass (COMBAT_MERGE_C_EVAL), one head launch per differentiated pass (COMBAT_FUSED_HEAD),
    weight-gradient reductions carried by the next launch (COMBAT_REDUCE_BEHIND), BatchNorm + ReLU in the consuming
    convolution's LDS (COMBAT_FUSED_PROLOGUE).  Each must leave a step's losses, counters and updated weights where the
    default leaves them (up to the summation order of the kernels it swaps)."""
    step_mod, engine = mods["step"], mods["engine"]
    gen = torch.Generator().manual_seed(11)
    x = (torch.randint(0, 256, (32, 3, 32, 32), generator=gen).float() / 255 - 0.5) / 0.5
    t = torch.randint(0, 10, (32,), generator=gen)
    t[:5] = 0
    flags = {"merge_c_eval": (step_mod, "MERGE_C_EVAL"), "fused_head": (step_mod, "FUSED_HEAD"),
             "reduce_behind": (engine, "REDUCE_BEHIND"), "fused_prologue": (engine, "FUSED_PROLOGUE")}
    mod, name = flags[switch]
    out = []
    for on in (False, True):
        prev = getattr(mod, name)
        setattr(mod, name, on)
        try:
            netc, clean, netg, netf = _build(mods, [0, 1, 2, 3])
            st = step_mod.AlternatedStep(netc.cuda(), netg.cuda(), clean.cuda().eval(), netf.cuda().eval(), Opt())
            for i in range(2):
                st.run(x.cuda(), t, step_mod.StepRandomness(3, 0.4, 0.7, [None] * 5))
            torch.cuda.synchronize()
            out.append((st.read_metrics(), torch.cat([p.detach().flatten() for p in netc.parameters()]).clone(),
                        torch.cat([p.detach().flatten() for p in netg.parameters()]).clone()))
        finally:
            setattr(mod, name, prev)
    (m0, c0, g0), (m1, c1, g1) = out
    for k in ("loss_c_sum", "loss_ce_sum", "clean_model_loss_sum", "loss_l2_sum"):
        assert abs(m0[k] - m1[k]) <= 2e-3 * max(1.0, abs(m0[k])), (switch, k, m0[k], m1[k])
    for k in ("clean_correct", "bd_correct", "clean_model_correct", "clean_model_bd_ba", "clean_model_bd_asr", "train_correct"):
        assert abs(m0[k] - m1[k]) <= 1, (switch, k, m0[k], m1[k])
    assert rel_l2(c1, c0) < 2e-3 and rel_l2(g1, g0) < 2e-3, (switch, rel_l2(c1, c0), rel_l2(g1, g0))


def test_step_keeps_a_dropped_pinned_batch_alive(mods):
    """combat_copy3 reads a pinned host batch through its device mapping, unseen by torch's caching
    host allocator.  A caller that drops its per-batch pin_memory() tensor right after run() (a DataLoader with
    pin_memory=True) must not have the block recycled under the device: the staging set holds the tensor until the
    copy's event has completed.  Here the block would be handed out again and overwritten at once."""
    step_mod = mods["step"]
    gen = torch.Generator().manual_seed(3)
    x = (torch.randint(0, 256, (16, 3, 32, 32), generator=gen).float() / 255 - 0.5) / 0.5
    t = torch.randint(0, 10, (16,), generator=gen)
    t[:4] = 0
    rnd = lambda: step_mod.StepRandomness(2, 0.4, 0.7, [None] * 5)
    outs = []
    for pinned in (False, True):
        netc, clean, netg, netf = _build(mods, [0, 1, 2, 3])
        st = step_mod.AlternatedStep(netc.cuda(), netg.cuda(), clean.cuda().eval(), netf.cuda().eval(), Opt())
        torch.cuda.synchronize()
        if pinned:
            xb = x.clone().pin_memory()
            ptr = xb.data_ptr()
            blocker = torch.cuda._sleep(200_000_000) if hasattr(torch.cuda, "_sleep") else None   # device busy: the copy runs late
            st.run(xb, t, rnd())
            held = st._stage[(st._stage_i - 1) % st.kStage]["inputs_ref"]
            assert held is not None and held.data_ptr() == ptr
            del xb, held
            junk = [torch.full((16, 3, 32, 32), 9.0).pin_memory() for _ in range(4)]    # would reuse a freed block
            assert all(j.data_ptr() != ptr for j in junk)
        else:
            st.run(x.cuda(), t, rnd())
        torch.cuda.synchronize()
        outs.append((st.inputs.clone(), st.read_metrics()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[1][0].cpu(), x)
    for k in ("loss_ce_sum", "loss_l2_sum", "clean_model_loss_sum"):
        assert abs(outs[0][1][k] - outs[1][1][k]) <= 1e-3 * max(1.0, abs(outs[0][1][k])), k   # (atomics-ordered weight gradients)


def test_classifier_step_metrics_cover_ragged_batches(mods):
    """CIFAR-10's last batch is ragged (50000 % 128 = 80); ClassifierStep keeps a slot,
    plans and loss / accuracy cells per batch size, and read_metrics sums (and resets) all of them."""
    step_mod, nets = mods["step"], mods["nets"]
    opt = Opt()
    gen = torch.Generator().manual_seed(8)
    batches = [((torch.rand(n, 3, 32, 32, generator=gen) * 2 - 1), torch.randint(0, 10, (n,), generator=gen)) for n in (48, 48, 20, 48)]
    totals = []
    for per_batch in (False, True):
        netc = seeded(nets.PreActResNet18, 0).cuda()
        st = step_mod.ClassifierStep(netc, opt)
        acc = {"loss_sum": 0.0, "correct": 0}
        for x, t in batches:
            st.run(x.cuda(), t)
            if per_batch:
                m = st.read_metrics(reset=True)
                acc = {k: acc[k] + m[k] for k in acc}
        if not per_batch:
            acc = st.read_metrics(reset=True)
            again = st.read_metrics()
            assert again["loss_sum"] == 0.0 and again["correct"] == 0      # every slot was reset
        totals.append(acc)
    n_all = sum(x.shape[0] for x, _ in batches)
    assert abs(totals[0]["loss_sum"] - totals[1]["loss_sum"]) < 1e-3 * totals[1]["loss_sum"], totals
    assert totals[0]["correct"] == totals[1]["correct"] and 0 <= totals[0]["correct"] <= n_all
    # the reference accumulates the batch-MEAN loss (train_victim.py:139): ~ln(10) per batch, over ALL four batches
    assert 1.5 * len(batches) < totals[0]["loss_sum"] < 4.0 * len(batches)


# ---------------------------------------------------------------- evaluation loops and the victim / clean-classifier step


def _eval_victim_nets(g, mods):
    """The networks of tests/golden/eval_victim.npz: seeds + seeded BatchNorm running statistics."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import test_oracle_golden as TO
    return TO.eval_victim_nets(g), TO


def test_eval_loops_vs_reference_counters(mods, golden, tmp_path, monkeypatch):
    """`eval.py::eval` (reference eval.py:108-152) and the in-training
    `train_generator.py::eval` (reference :321-465) on a 101-image synthetic test set (batches of 64 and 37, as
    recorded) with the recorded blur sigmas, against (1) the counters the REFERENCE modules produced
    (tests/golden/eval_victim.npz) and (2) the oracle's eval_batch driven with the bf16 emulation.  Counters are
    integer argmax counts over 101 / ~90 images of freshly initialised networks (small logit margins): +-3 images
    against fp32, +-2 against the emulation."""
    import importlib
    from oracle import combat_oracle as O
    from combat_amd import trigger
    from combat_amd.data import ArrayLoader
    from combat_amd.dist import NullWriter
    g = golden("eval_victim")
    (netc, clean, netg, netf), TO = _eval_victim_nets(g, mods)
    sd = lambda m: {k: v.detach().clone() for k, v in m.state_dict().items()}
    oc, ok, og, of = sd(netc), sd(clean), sd(netg), sd(netf)
    s_img, s_lab = (int(v) for v in g["eval/seeds"])
    u8, lab = [], []
    for s, b in enumerate(int(v) for v in g["eval/batch"]):
        u8.append(torch.randint(0, 256, (b, 3, 32, 32), generator=torch.Generator().manual_seed(s_img + s), dtype=torch.uint8))
        lab.append(torch.randint(0, 10, (b,), generator=torch.Generator().manual_seed(s_lab + s)))
    test_dl = ArrayLoader(torch.cat(u8).numpy(), torch.cat(lab).numpy(), 64, False)
    sig = [float(v) for v in g["sigma"]]
    # oracle with the bf16 emulation on the same batches
    emu = {}
    for s, (xb, tb) in enumerate(test_dl):
        r = O.eval_batch(oc, og, xb, tb, sig[s], O.StepConfig(), clean=ok, netf=of, clf_fn=E.preact_forward_emu,
                         gen_fn=E.unet_forward_emu)
        emu = {k: emu.get(k, 0) + v for k, v in r.items()}
    ref = {k[5:]: int(g[k].sum()) for k in g if k.startswith("eval/") and k not in ("eval/batch", "eval/seeds")}
    assert ref["clean_n"] == 101 and emu["clean_n"] == 101

    class EOpt(Opt):
        device = "cuda"
        ckpt_path = str(tmp_path / "ck.pth.tar")

    opt = EOpt()
    netc, clean, netg, netf = netc.cuda().eval(), clean.cuda().eval(), netg.cuda().eval(), netf.cuda().eval()
    draws = []
    monkeypatch.setattr(trigger, "sample_sigma", lambda rng=(0.1, 1.0): draws.pop(0))

    def near(ours, key, n_key):
        n = ref[n_key]
        assert abs(ours * n / 100.0 - ref[key]) <= 3.01, (key, ours * n / 100.0, ref[key])
        assert abs(ours * n / 100.0 - emu[key]) <= 2.01, (key, ours * n / 100.0, emu[key])

    # ---- eval.py (reference eval.py:108-152): clean accuracy, Bd BA, Bd ASR
    ev = importlib.import_module("eval")
    draws[:] = list(sig)
    acc_clean, acc_ba, acc_asr = ev.eval(netc, netg, test_dl, NullWriter(), opt)
    near(acc_clean, "clean_correct", "clean_n")
    near(acc_ba, "bd_ba", "bd_n")
    near(acc_asr, "bd_correct", "bd_n")
    # ---- train_generator.py::eval (reference :321-465): six accuracies + the checkpoint
    tg = importlib.import_module("train_generator")
    oC = torch.optim.SGD(netc.parameters(), 1e-2, momentum=0.9, weight_decay=5e-4, nesterov=True)
    oG = torch.optim.SGD(netg.parameters(), 1e-2, momentum=0.9, weight_decay=5e-4, nesterov=True)
    sC = torch.optim.lr_scheduler.MultiStepLR(oC, [100, 150], 0.1)
    sG = torch.optim.lr_scheduler.MultiStepLR(oG, [100, 150], 0.1)
    draws[:] = list(sig)
    best = tg.eval(netc, oC, sC, netg, oG, sG, netf, clean, test_dl, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, NullWriter(), 0, opt)
    for ours, key, n_key in zip(best, ("clean_correct", "bd_correct", "f_correct", "clean_model_correct", "clean_model_bd_ba",
                                       "clean_model_bd_asr"), ("clean_n", "bd_n", "bd_n", "clean_n", "bd_n", "bd_n")):
        near(ours, key, n_key)
    ck = torch.load(opt.ckpt_path, map_location="cpu", weights_only=True)
    assert abs(ck["best_clean_acc"] - best[0]) < 1e-9 and len(ck["netC"]) == 102


def test_victim_and_clean_classifier_step_vs_oracle(mods, golden, monkeypatch):
    """ClassifierStep = the loop bodies of train_victim.py:102-141 (frozen generator poisons the
    flagged images; D3 intent) and train_clean_classifier.py:87-110, against the trace of the REFERENCE modules
    (tests/golden/eval_victim.npz: loss within 1e-2, accuracy count +-2, gradient norm within 3 %, parameter update
    within the mask-flip bound 0.35) and, teacher-forced, against the oracle's victim_step driven with the bf16
    emulation (gradients rel-L2 <= 4e-2: pins batch order, labels, trigger and backward wiring)."""
    from oracle import combat_oracle as O
    from combat_amd import trigger
    step_mod = mods["step"]
    g = golden("eval_victim")
    vi, vl = (int(v) for v in g["victim/seeds"])
    u8 = torch.randint(0, 256, (48, 3, 32, 32), generator=torch.Generator().manual_seed(vi), dtype=torch.uint8)
    x = (u8.float() / 255 - 0.5) / 0.5
    t = torch.randint(0, 10, (48,), generator=torch.Generator().manual_seed(vl))
    t[:6] = 0
    sigma = float(g["victim/sigma"])
    monkeypatch.setattr(trigger, "sample_sigma", lambda rng=(0.1, 1.0): sigma)
    for tag, pz in (("victim", torch.from_numpy(g["victim/poisoned"])), ("cleanclf", None)):
        (netc, clean, netg, netf), TO = _eval_victim_nets(g, mods)
        sd = lambda m: {k: v.detach().clone() for k, v in m.state_dict().items()}
        oc, og = sd(netc), sd(netg)
        p0 = {k: v.detach().clone() for k, v in netc.named_parameters()}
        names = [k for k, _ in netc.named_parameters()]
        netc, netg = netc.cuda(), netg.cuda().eval()
        opt = Opt()
        st = step_mod.ClassifierStep(netc, opt, netg if pz is not None else None)
        st.run(x.cuda(), t, pz)
        torch.cuda.synchronize()
        m = st.read_metrics()
        assert abs(m["loss_sum"] - float(g[tag + "/loss"])) < 1e-2, (tag, m["loss_sum"], float(g[tag + "/loss"]))
        assert abs(m["correct"] - int(g[tag + "/correct"])) <= 2
        fp = st.eC.fp
        gn = float(fp.grad.double().norm())
        assert abs(gn - float(g[tag + "/gnorm"])) < 3e-2 * float(g[tag + "/gnorm"]), (tag, gn, float(g[tag + "/gnorm"]))
        # teacher-forced emulation through the oracle's victim_step
        keys = ["stem"] + ["b%d.%s" % (b, s) for b in range(8) for s in ("y1", "out", "sc")]
        force = stored(st.slot, keys)
        clf = lambda p, xx, train: E.preact_forward_emu(p, xx, train, force=force)
        r = O.victim_step(oc, [None] * len(names), x, t, O.StepConfig(), netg=og if pz is not None else None, poisoned=pz,
                          sigma=sigma, clf_fn=clf, gen_fn=E.unet_forward_emu)
        ours = flat_grads(fp, names)
        e_tf = rel_l2(ours, torch.cat([a.reshape(-1) for a in r["grads"]]))
        assert e_tf < 4e-2, (tag, e_tf)
        assert abs(m["loss_sum"] - r["loss"]) < 5e-3 and abs(m["correct"] - r["correct"]) <= 1
        # the optimiser applied the engine's own gradient; the update stays within the mask-flip bound of fp32's
        num = den = 0.0
        for k in names:
            ref_after_idx, ref_after = g["%s/after/%s/idx" % (tag, k)], g["%s/after/%s/val" % (tag, k)]
            d_ref = ref_after - p0[k].double().flatten()[ref_after_idx].numpy()
            d_our = (dict(netc.named_parameters())[k].detach().cpu().double().flatten()[ref_after_idx] -
                     p0[k].double().flatten()[ref_after_idx]).numpy()
            num, den = num + float(((d_our - d_ref) ** 2).sum()), den + float((d_ref ** 2).sum())
        assert (num / den) ** 0.5 < 0.35, (tag, (num / den) ** 0.5)
        for k in ("conv1.weight", "layer3.0.shortcut.0.weight", "linear.bias"):
            gk = fp.logical(fp.grad, k).cpu()
            exp = p0[k] - 1e-2 * 1.9 * (gk + 5e-4 * p0[k])
            assert rel_l2(dict(netc.named_parameters())[k].detach().cpu(), exp) < 1e-6, k
