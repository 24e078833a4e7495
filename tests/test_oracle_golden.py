"""Pin the CPU oracle (oracle/combat_oracle.py) against golden vectors recorded from the
reference's own modules (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from combat_amd import nets
from oracle import combat_oracle as O


def T(a):
    return torch.from_numpy(np.asarray(a))


def state(module):
    return {k: v.clone() for k, v in module.state_dict().items()}


def check_summary(g, prefix, named, rtol=2e-4, atol=1e-6):
    """Compare (l2, sampled entries) of each tensor with the recorded summaries."""
    for k, v in named:
        v = v.detach().double().flatten()
        l2 = float(g["%s/%s/l2" % (prefix, k)])
        assert abs(float(v.norm()) - l2) <= rtol * max(l2, 1e-12) + atol, (prefix, k)
        idx = g["%s/%s/idx" % (prefix, k)]
        ref = g["%s/%s/val" % (prefix, k)]
        np.testing.assert_allclose(v[idx].numpy(), ref, rtol=rtol, atol=atol + rtol * l2 / max(1, v.numel()) ** 0.5,
                                   err_msg="%s/%s" % (prefix, k))


def seeded(ctor, seed):
    torch.manual_seed(seed)
    return ctor()


# ---------------------------------------------------------------- parameter mirrors


@pytest.mark.parametrize("name,ctor", [
    ("unet", lambda: nets.UnetGenerator(None)),
    ("preact", lambda: nets.PreActResNet18()),
    ("resnet", lambda: nets.ResNet18(num_classes=8, input_size=64)),
    ("freq", lambda: nets.FrequencyModel(num_classes=2, n_input=3, input_size=32)),
])
def test_mirror_modules_reproduce_reference_init(golden, name, ctor):
    """Same seed => same parameters, same state_dict keys and order as the reference class."""
    g = golden(name)
    m = seeded(ctor, int(g["seed"]))
    sd = m.state_dict()
    ref_keys = [k[len("param/"):-len("/l2")] for k in g if k.startswith("param/") and k.endswith("/l2")]
    assert list(sd.keys()) == ref_keys
    fresh = {k: v for k, v in sd.items() if not (name == "freq" and "running" in k)}
    check_summary(g, "param", fresh.items(), rtol=1e-6, atol=0)


def test_state_dict_entry_counts():
    assert len(nets.PreActResNet18().state_dict()) == 102
    assert len(nets.ResNet18().state_dict()) == 122
    assert len(nets.UnetGenerator(None).state_dict()) == 32
    assert sum(p.numel() for p in nets.PreActResNet18().parameters()) == 11171146
    assert sum(p.numel() for p in nets.UnetGenerator(None).parameters()) == 9370243


# ---------------------------------------------------------------- DCT / low-pass


@pytest.mark.parametrize("n", [32, 64])
def test_dct_matches_reference_fft_path(golden, n):
    g = golden("dct")
    x = T(g["x%d" % n])
    np.testing.assert_allclose(O.dct_2d(x).numpy(), g["dct%d" % n], rtol=0, atol=2e-3)  # scale ~4e3
    np.testing.assert_allclose(O.idct_2d(x).numpy(), g["idct%d" % n], rtol=0, atol=2e-3)
    np.testing.assert_allclose(O.dct_2d(T(g["u8_%d" % n])).numpy(), g["dct_u8_%d" % n], rtol=0, atol=4e-3)


@pytest.mark.parametrize("n", [32, 64])
def test_low_freq_and_closed_form(golden, n):
    g = golden("dct")
    x = T(g["lf_x%d" % n]).requires_grad_(True)
    y = O.low_freq(x, 0.65)
    np.testing.assert_allclose(y.detach().numpy(), g["lf_y%d" % n], atol=2e-5)
    (gx,) = torch.autograd.grad(y, x, T(g["lf_g%d" % n]))
    np.testing.assert_allclose(gx.numpy(), g["lf_gx%d" % n], atol=2e-5)
    p = O.lowpass_matrix(n, 0.65)
    np.testing.assert_allclose((p @ x.detach() @ p.T).numpy(), g["lf_y%d" % n], atol=2e-5)
    np.testing.assert_allclose((p @ T(g["lf_g%d" % n]) @ p).numpy(), g["lf_gx%d" % n], atol=2e-5)
    np.testing.assert_allclose((p @ p).numpy(), p.numpy(), atol=1e-6)  # idempotent
    np.testing.assert_allclose(p.numpy(), p.T.numpy(), atol=1e-7)


# ---------------------------------------------------------------- UNet


@pytest.mark.parametrize("tag", ["b4", "b1", "c64"])
def test_unet_forward_backward(golden, tag):
    g = golden("unet")
    m = seeded(lambda: nets.UnetGenerator(None), int(g["seed"]))
    p = {k: v.requires_grad_(True) for k, v in state(m).items()}
    x = T(g[tag + "/x"]).requires_grad_(True)
    y = O.unet_forward(p, x)
    np.testing.assert_allclose(y.detach().numpy(), g[tag + "/y"], atol=3e-5)
    names = [k for k, _ in m.named_parameters()]
    grads = torch.autograd.grad(y, [x] + [p[k] for k in names], T(g[tag + "/g"]))
    np.testing.assert_allclose(grads[0].numpy(), g[tag + "/gx"], atol=5e-5, rtol=1e-3)
    check_summary(g, tag + "/gp", zip(names, grads[1:]), rtol=2e-3, atol=1e-5)


def test_unet_empty_batch(golden):
    g = golden("unet")
    m = nets.UnetGenerator(None)
    y = O.unet_forward(state(m), torch.zeros(0, 3, 32, 32))
    assert tuple(y.shape) == tuple(g["b0/shape"])
    assert bool(g["train_equals_eval"])


# ---------------------------------------------------------------- classifiers


def _clf_case(g, tag, fwd, p, train, names):
    x = T(g[tag + "/x"]).requires_grad_(True)
    t = T(g[tag + "/t"])
    logits = fwd(p, x, train)
    np.testing.assert_allclose(logits.detach().numpy(), g[tag + "/logits"], atol=5e-5, rtol=1e-4)
    loss = F.cross_entropy(logits, t)
    assert abs(float(loss.detach()) - float(g[tag + "/loss"])) < 1e-5
    grads = torch.autograd.grad(loss, [x] + [p[k] for k in names])
    np.testing.assert_allclose(grads[0].numpy(), g[tag + "/gx"], atol=1e-6, rtol=2e-3)
    check_summary(g, tag + "/gp", zip(names, grads[1:]), rtol=3e-3, atol=1e-6)
    if train:
        for k, v in p.items():
            if "running" in k or "num_batches" in k:
                np.testing.assert_allclose(v.detach().numpy(), g["%s/buf/%s" % (tag, k)], atol=1e-6, rtol=1e-5,
                                           err_msg=k)


def test_preact_resnet18(golden):
    g = golden("preact")
    m = seeded(nets.PreActResNet18, int(g["seed"]))
    names = [k for k, _ in m.named_parameters()]
    p = state(m)
    for k in names:
        p[k].requires_grad_(True)
    _clf_case(g, "eval0", O.preact_resnet18_forward, p, False, names)
    _clf_case(g, "train", O.preact_resnet18_forward, p, True, names)
    _clf_case(g, "eval1", O.preact_resnet18_forward, p, False, names)


def test_resnet18_celeba_shape(golden):
    g = golden("resnet")
    m = seeded(lambda: nets.ResNet18(num_classes=8, input_size=64), int(g["seed"]))
    names = [k for k, _ in m.named_parameters()]
    p = state(m)
    for k in names:
        p[k].requires_grad_(True)
    _clf_case(g, "train", O.resnet18_forward, p, True, names)
    _clf_case(g, "eval1", O.resnet18_forward, p, False, names)


def test_frequency_model(golden):
    g = golden("freq")
    m = seeded(lambda: nets.FrequencyModel(2, 3, 32), int(g["seed"]))
    p = state(m)
    for k in p:
        if "buf/" + k in g:
            p[k] = T(g["buf/" + k])
    img = T(g["img"])
    inp = O.frequency_input(img)
    np.testing.assert_allclose(inp.numpy(), g["dct_in"], atol=4e-3)
    logits = O.frequency_model_forward(p, T(g["dct_in"]))
    np.testing.assert_allclose(logits.numpy(), g["logits"], rtol=2e-4, atol=2e-3)


# ---------------------------------------------------------------- optimiser / selection


def test_sgd_matches_torch_optim():
    torch.manual_seed(0)
    ps = [torch.randn(5, 3), torch.randn(7)]
    ref = [p.clone().requires_grad_(True) for p in ps]
    opt = torch.optim.SGD(ref, 1e-2, momentum=0.9, weight_decay=5e-4, nesterov=True)
    bufs = [None, None]
    for _ in range(3):
        gs = [torch.randn_like(p) for p in ps]
        for r, gr in zip(ref, gs):
            r.grad = gr.clone()
        opt.step()
        O.sgd_nesterov_step(ps, gs, bufs, 1e-2)
    for a, b in zip(ps, ref):
        np.testing.assert_allclose(a.numpy(), b.detach().numpy(), rtol=1e-6, atol=1e-7)


def test_poison_order():
    t = torch.tensor([3, 0, 5, 0, 0, 7])
    bd = O.create_targets_bd(t)
    perm, tt = O.poison_order(t, bd, 2)
    assert perm.tolist() == [1, 3, 4, 0, 2, 5]
    assert tt.tolist() == [0, 0, 0, 3, 5, 7]
    assert O.create_targets_bd(t, "all2all").tolist() == [4, 1, 6, 1, 1, 8]
    with pytest.raises(Exception):
        O.create_targets_bd(t, "nope")


# ---------------------------------------------------------------- whole step


def test_alternated_step_trace(golden):
    """3 steps, B=16: losses, metric counts and final parameters match the trace produced by
    driving the reference modules with torch.optim.SGD in train_generator.py order."""
    g = golden("step")
    s0, s1, s2, s3 = [int(s) for s in g["seeds"]]
    netc = state(seeded(nets.PreActResNet18, s0))
    clean = state(seeded(nets.PreActResNet18, s1))
    netg_m = seeded(lambda: nets.UnetGenerator(None), s2)
    netg = state(netg_m)
    netf = state(seeded(lambda: nets.FrequencyModel(2, 3, 32), s3))
    bufs_c = [None] * len(O.trainable_names(netc))
    bufs_g = [None] * len(O.trainable_names(netg))
    cfg = O.StepConfig()
    for s in range(3):
        rnd = O.StepRandomness(int(g["num_bd"][s]), float(g["sigma_c"][s]), float(g["sigma_g"][s]))
        out = O.alternated_step(netc, netg, clean, netf, bufs_c, bufs_g, T(g["step%d/inputs" % s]),
                                T(g["step%d/targets" % s]), rnd, cfg, as_written=(s == 1))
        for k in ("loss_c", "loss_ce", "loss_l2", "loss_grad_l2", "clean_model_loss"):
            ref = float(g["trace/" + k][s])
            assert abs(out[k] - ref) <= 2e-4 * max(1.0, abs(ref)), (s, k, out[k], ref)
        # step 0 starts from identical state: tight.  Later steps inherit fp32 rounding noise that
        # this ill-conditioned start (B=16, fresh init, train-mode BN) amplifies ~1000x per step
        # (a 1e-7 relative nudge of conv1.weight moves it by 8e-5 after three steps).
        for k, tol0 in (("gnorm_c", 1e-5), ("gnorm_g", 2e-4)):  # Phase G already sees the updated netC
            ref = float(g["trace/" + k][s])
            assert abs(out[k] - ref) <= (tol0 if s == 0 else 5e-3) * ref, (s, k, out[k], ref)
        for k in ("clean_correct", "bd_correct", "f_correct", "clean_model_correct", "clean_model_bd_ba",
                  "clean_model_bd_asr"):
            assert out[k] == int(g["trace/" + k][s]), (s, k)
    check_summary(g, "final/netc", netc.items(), rtol=5e-3, atol=4e-4)
    check_summary(g, "final/netg", netg.items(), rtol=5e-3, atol=4e-4)


# ---------------------------------------------------------------- the bf16 emulation is pinned to the oracle


def _rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / max(float(b.norm()), 1e-30))


def test_bf16_emulation_without_rounding_is_the_oracle():
    """tests/bf16_emu.py is the tight reference of the GPU engine tests.  With its rounding switched off it
    must reproduce the (golden-pinned) oracle forwards AND their autograd gradients: same dataflow, same
    statistics, same running-stat updates.  Run in float64 so that what is compared is the dataflow and not
    fp32 re-association (the emulation applies BatchNorm as x*scale+shift, drops the conv biases that an
    InstanceNorm cancels and, for the eval-mode ResNet18, folds bn2 / the shortcut norm into the weights, as
    the engine does): rel-L2 <= 1e-9 everywhere (VERDICT r1 asked for <= 1e-6)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import bf16_emu as E
    from combat_amd import nets
    from oracle import combat_oracle as O
    gen = torch.Generator().manual_seed(77)
    f64 = torch.float64
    dbl = lambda sd: {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    E.ROUND = False
    try:
        # ---- UNet
        torch.manual_seed(3)
        sd = dbl(nets.UnetGenerator(None).state_dict())
        names = O.trainable_names(sd)
        x = torch.rand(3, 3, 32, 32, generator=gen, dtype=f64) * 2 - 1
        cot = torch.randn(3, 3, 32, 32, generator=gen, dtype=f64)
        outs = []
        for fn in (O.unet_forward, E.unet_forward_emu):
            p = {k: v.clone().requires_grad_(k in names) for k, v in sd.items()}
            xi = x.clone().requires_grad_(True)
            y = fn(p, xi)
            g = torch.autograd.grad(y, [xi] + [p[k] for k in names], cot, allow_unused=True)
            outs.append((y.detach(), [torch.zeros_like(t) if a is None else a for t, a in zip([xi] + [p[k] for k in names], g)]))
        assert _rel(outs[1][0], outs[0][0]) < 1e-9
        for k, a, b in zip(["x"] + names, outs[1][1], outs[0][1]):
            if k.endswith(".bias") and k not in ("upconv0_0.bias", "conv0_0.bias"):
                # a bias in front of an InstanceNorm: the emulation does not add it (exact zero gradient); the
                # oracle's gradient for it is zero up to rounding
                assert float(a.abs().max()) == 0.0 and float(b.abs().max()) < 1e-9, k
                continue
            assert _rel(a, b) < 1e-9, (k, _rel(a, b))
        # ---- classifiers, train then eval (running statistics carried over)
        cases = ((nets.PreActResNet18, O.preact_resnet18_forward, E.preact_forward_emu, 32, 10),
                 (lambda: nets.ResNet18(num_classes=8, input_size=64), O.resnet18_forward, E.resnet_forward_emu, 64, 8))
        for ctor, f_o, f_e, hw, classes in cases:
            torch.manual_seed(5)
            sd = dbl(ctor().state_dict())
            names = O.trainable_names(sd)
            x = torch.rand(4, 3, hw, hw, generator=gen, dtype=f64) * 2 - 1
            t = torch.randint(0, classes, (4,), generator=gen)
            states = []
            for fn in (f_o, f_e):
                p = {k: v.clone() for k, v in sd.items()}
                for k in names:
                    p[k].requires_grad_(True)
                res = []
                for train in (True, False):
                    xi = x.clone().requires_grad_(True)
                    lg = fn(p, xi, train)
                    g = torch.autograd.grad(torch.nn.functional.cross_entropy(lg, t), [xi] + [p[k] for k in names])
                    res.append((lg.detach(), g))
                states.append((res, {k: v.detach().clone() for k, v in p.items() if "running" in k or "num_batches" in k}))
            (ro, bo), (re_, be) = states
            for mode in (0, 1):
                assert _rel(re_[mode][0], ro[mode][0]) < 1e-9, (hw, mode)
                for k, a, b in zip(["x"] + names, re_[mode][1], ro[mode][1]):
                    assert _rel(a, b) < 1e-9 or float(b.abs().max()) < 1e-12, (hw, mode, k, _rel(a, b))
            for k in bo:
                assert _rel(be[k].double(), bo[k].double()) < 1e-9, k
    finally:
        E.ROUND = True


def test_batched_augmentation_equals_the_per_sample_loop():
    """oracle.post_tensor_transform_batched (used when the oracle step is timed on a device as the stock
    PyTorch baseline) is the same differentiable map as the per-sample restatement."""
    import numpy as np
    from oracle import combat_oracle as O
    rng = np.random.default_rng(5)
    b = 12
    p = O.AugParams(rng.integers(0, 11, b).astype(np.int32), rng.integers(0, 11, b).astype(np.int32),
                    np.where(rng.random(b) < 0.5, rng.uniform(-10, 10, b), 0).astype(np.float32),
                    (rng.random(b) < 0.5).astype(np.int32))
    x = torch.rand(b, 3, 32, 32, generator=torch.Generator().manual_seed(1)) * 2 - 1
    cot = torch.randn(b, 3, 32, 32, generator=torch.Generator().manual_seed(2))
    outs = []
    for fn in (O.post_tensor_transform, O.post_tensor_transform_batched):
        xi = x.clone().requires_grad_(True)
        y = fn(xi, p)
        (gx,) = torch.autograd.grad(y, xi, cot)
        outs.append((y.detach(), gx))
    # (the loop evaluates cos/sin of the angle in double, the batched form in fp32: ~3e-6 on values in [-1, 1])
    assert float((outs[0][0] - outs[1][0]).abs().max()) < 1e-5
    assert float((outs[0][1] - outs[1][1]).abs().max()) < 1e-4
    assert O.post_tensor_transform_batched(x, None) is x


# ---------------------------------------------------------------- evaluation loop body, victim / clean-classifier step


def randomize_bn_buffers(net, seed):
    """tests/golden/make_golden.py::randomize_bn_buffers on combat_amd's mirror modules (same module order)."""
    i = 0
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.05, generator=torch.Generator().manual_seed(seed + i))
                m.running_var.uniform_(0.6, 1.4, generator=torch.Generator().manual_seed(seed + 1000 + i))
                i += 1
    return net


def eval_victim_nets(g):
    from combat_amd import nets
    seeds, bns = [int(v) for v in g["seeds"]], [int(v) for v in g["bn_seeds"]]
    out = []
    for ctor, sd, bn in zip((nets.PreActResNet18, nets.PreActResNet18, lambda: nets.UnetGenerator(None),
                             lambda: nets.FrequencyModel(2, 3, 32)), seeds, bns):
        torch.manual_seed(sd)
        m = ctor()
        out.append(randomize_bn_buffers(m, bn) if bn else m)
    return out      # netc, clean, netg, netf


def synth_images(b, hw, seed):
    u8 = torch.randint(0, 256, (b, 3, hw, hw), generator=torch.Generator().manual_seed(seed), dtype=torch.uint8)
    return (u8.float() / 255 - 0.5) / 0.5


def test_eval_batch_and_victim_step_match_the_reference_modules(golden):
    """oracle.eval_batch / oracle.victim_step against counters, losses and gradients recorded from the reference's
    own modules in the order of train_generator.py:353-391 / eval.py:119-143 and train_victim.py:102-141 (D3
    intent) / train_clean_classifier.py:87-110 (tests/golden/make_golden.py::golden_eval_victim)."""
    from oracle import combat_oracle as O
    g = golden("eval_victim")
    netc, clean, netg, netf = eval_victim_nets(g)
    sd = lambda m: {k: v.detach().clone() for k, v in m.state_dict().items()}
    oc, ok, og, of = sd(netc), sd(clean), sd(netg), sd(netf)
    cfg = O.StepConfig()
    s_img, s_lab = (int(v) for v in g["eval/seeds"])
    for s, b in enumerate(int(v) for v in g["eval/batch"]):
        x = synth_images(b, 32, s_img + s)
        t = torch.randint(0, 10, (b,), generator=torch.Generator().manual_seed(s_lab + s))
        r = O.eval_batch(oc, og, x, t, float(g["sigma"][s]), cfg, clean=ok, netf=of)
        for k, v in r.items():
            assert v == int(g["eval/" + k][s]), (s, k, v, int(g["eval/" + k][s]))
        r2 = O.eval_batch(oc, og, x, t, float(g["sigma"][s]), cfg)          # eval.py's form: no detector / clean model
        assert (r2["clean_correct"], r2["bd_correct"], r2["bd_ba"]) == (r["clean_correct"], r["bd_correct"], r["bd_ba"])
    t0 = torch.zeros(5, dtype=torch.int64)                                   # only target-class images: nothing to trigger
    assert O.eval_batch(oc, og, x[:5], t0, 0.5, cfg, clean=ok, netf=of)["bd_n"] == 0
    # ---- victim step, then clean-classifier step on the same (updated) classifier
    vi, vl = (int(v) for v in g["victim/seeds"])
    x = synth_images(48, 32, vi)
    t = torch.randint(0, 10, (48,), generator=torch.Generator().manual_seed(vl))
    t[:6] = 0
    names = O.trainable_names(oc)
    oc_start = {k: v.clone() for k, v in oc.items()}
    for tag, pz, ng in (("victim", torch.from_numpy(g["victim/poisoned"]), og), ("cleanclf", None, None)):
        oc = {k: v.clone() for k, v in oc_start.items()}            # both steps from the same start state
        bufs = [None] * len(names)
        r = O.victim_step(oc, bufs, x, t, cfg, netg=ng, poisoned=pz, sigma=float(g["victim/sigma"]))
        assert abs(r["loss"] - float(g[tag + "/loss"])) < 1e-5 and r["correct"] == int(g[tag + "/correct"])
        assert abs(r["gnorm"] - float(g[tag + "/gnorm"])) < 1e-4 * float(g[tag + "/gnorm"])
        # Gradient tolerance 5e-3: the triggered images differ by ~1e-6 between the reference's FFT low-pass and the
        # oracle's closed form P X P^T, and this freshly initialised train-mode network amplifies a 1e-6
        # perturbation of 3 of the 48 images into a 1.2e-3 relative change of the stem gradient (measured with the
        # reference module itself; without triggered images the oracle's gradients equal the module's bit for bit).
        num = den = 0.0       # (relative L2 over the sampled entries of all tensors: single entries move more)
        for k, v in zip(names, r["grads"]):
            d = v.double().flatten()[g["%s/gp/%s/idx" % (tag, k)]].numpy() - g["%s/gp/%s/val" % (tag, k)]
            num, den = num + float((d ** 2).sum()), den + float((g["%s/gp/%s/val" % (tag, k)] ** 2).sum())
            assert abs(float(v.double().norm()) - float(g["%s/gp/%s/l2" % (tag, k)])) < 5e-3 * float(g["%s/gp/%s/l2" % (tag, k)]) + 1e-9, k
        assert (num / den) ** 0.5 < (5e-3 if tag == "victim" else 1e-5), (tag, (num / den) ** 0.5)
        check_summary(g, tag + "/after", oc.items(), 1e-4)
    assert r["num_bd"] == 0


# ---------------------------------------------------------------- WaNet: GridGenerator, warp, alternated step


def test_grid_generator_is_a_constant_field_and_matches_the_reference(golden):
    """networks/models.py:344-385.  The reference pools an affine-free InstanceNorm output, whose spatial mean is 0
    by construction: the recorded outputs are the same for every input (spread 3e-8), the encoder's gradients are
    rounding noise (< 1e-6) next to the head's (O(1)) -- so the trigger is tanh(fc2(lrelu(fc1.bias))), which is what
    the HIP path computes (combat_amd.engine.GridEngine)."""
    from combat_amd import nets
    from oracle import combat_oracle as O
    g = golden("wanet")

    class WOpt:
        s = 2

    torch.manual_seed(int(g["seed"]))
    m = nets.GridGenerator(WOpt())
    check_summary(g, "param", state(m).items(), 1e-6)
    p = {k: v.clone().requires_grad_(True) for k, v in state(m).items()}
    x = T(g["gg/x"]).requires_grad_(True)
    y = O.grid_generator_forward(p, x)
    np.testing.assert_allclose(y.detach().numpy(), g["gg/y"], atol=1e-6)
    assert float(g["gg/spread"]) < 1e-7 and float(g["gg/gx_max"]) < 1e-6
    names = list(p)
    grads = torch.autograd.grad(y, [x] + [p[k] for k in names], T(g["gg/cot"]))
    for k, gr in zip(names, grads[1:]):
        if k.startswith("fc") and k != "fc1.weight":
            np.testing.assert_allclose(gr.numpy(), g["gg/grad/" + k], rtol=1e-4, atol=1e-6)
        else:
            assert float(g["gg/gmax/" + k]) < 1e-6 and float(gr.abs().max()) < 1e-5, k
    # the closed form
    b1, w2, b2 = p["fc1.bias"].detach(), p["fc2.weight"].detach(), p["fc2.bias"].detach()
    const = torch.tanh(F.linear(F.leaky_relu(b1, 0.2), w2, b2)).reshape(2, 2, 2)
    assert float((y.detach() - const[None]).abs().max()) < 1e-6


@pytest.mark.parametrize("tag,rescale", [("warp32", 0.15), ("warp64", 0.15), ("warpbig", 0.9)])
def test_wanet_warp_matches_the_reference_calls(golden, tag, rescale):
    """train_generator_wanet.py:151-157 (bicubic upsample align_corners=True, identity blend, clamp, bilinear
    grid_sample) with gradients w.r.t. the images and the warp field."""
    from oracle import combat_oracle as O
    g = golden("wanet")
    x, nz = T(g[tag + "/x"]).requires_grad_(True), T(g[tag + "/noise"]).requires_grad_(True)
    ibd, ng = O.wanet_warp(x, nz, rescale)
    np.testing.assert_allclose(ibd.detach().numpy(), g[tag + "/out"], atol=2e-6)
    loss = (ibd * T(g[tag + "/cot"])).sum() + (0.02 * F.mse_loss(ng, ng * 0) if tag != "warpbig" else 0.0)
    gx, gn = torch.autograd.grad(loss, [x, nz])
    np.testing.assert_allclose(gx.numpy(), g[tag + "/gx"], atol=2e-6)
    np.testing.assert_allclose(gn.numpy(), g[tag + "/gnoise"], rtol=1e-4, atol=1e-4)
    assert tuple(O.wanet_identity_grid(8).shape) == (1, 8, 8, 2)
    assert float(O.wanet_identity_grid(8)[0, 2, 5, 0]) == pytest.approx(-1 + 2 * 5 / 7)     # [..., 0] is x (the column)


def test_wanet_alternated_step_trace(golden):
    """Two WaNet steps (train_generator_wanet.py:132-237) against the reference modules + torch.optim.SGD."""
    from combat_amd import nets
    from oracle import combat_oracle as O
    g = golden("wanet")

    class WOpt:
        s = 2

    seeds = [int(v) for v in g["seeds"]]
    mods = [seeded(nets.PreActResNet18, seeds[0]), seeded(nets.PreActResNet18, seeds[1]),
            seeded(lambda: nets.GridGenerator(WOpt()), seeds[2]), seeded(lambda: nets.FrequencyModel(2, 3, 32), seeds[3])]
    oc, ok, og, of = (state(m) for m in mods)
    bufs_c, bufs_g = [None] * len(O.trainable_names(oc)), [None] * len(O.trainable_names(og))
    cfg = O.StepConfig(trigger="wanet")
    s_img, s_lab = (int(v) for v in g["step_seeds"])
    for s in range(2):
        x = synth_images(16, 32, s_img + s)
        t = torch.randint(0, 10, (16,), generator=torch.Generator().manual_seed(s_lab + s))
        t[:4] = 0
        assert torch.equal(t, T(g["step%d/targets" % s]))
        r = O.alternated_step(oc, og, ok, of, bufs_c, bufs_g, x, t, O.StepRandomness(int(g["num_bd"][s]), 0.0, 0.0), cfg,
                              as_written=(s == 1))
        for k in ("loss_c", "loss_ce", "loss_l2", "loss_grad_l2", "clean_model_loss", "gnorm_c", "gnorm_g"):
            ref = float(g["trace/" + k][s])
            assert abs(r[k] - ref) < 2e-4 * max(1.0, abs(ref)) + 1e-7, (s, k, r[k], ref)
        for k in ("clean_correct", "bd_correct", "f_correct", "clean_model_correct", "clean_model_bd_ba", "clean_model_bd_asr"):
            assert r[k] == int(g["trace/" + k][s]), (s, k)
    check_summary(g, "final/netg", og.items(), 1e-4)


def test_wanet_trajectory_first_steps(golden):
    """The first six steps of the 60-step WaNet trace (reference modules + torch.optim.SGD, lr 2e-3, B = 32) through the
    oracle: losses to 2e-4, counters exact, the generator's field to 1e-5."""
    from combat_amd import nets
    from oracle import combat_oracle as O
    g = golden("wanet_trajectory")

    class WOpt:
        s = 2

    seeds = [int(v) for v in g["seeds"]]
    mods = [seeded(nets.PreActResNet18, seeds[0]), seeded(nets.PreActResNet18, seeds[1]),
            seeded(lambda: nets.GridGenerator(WOpt()), seeds[2]), seeded(lambda: nets.FrequencyModel(2, 3, 32), seeds[3])]
    oc, ok, og, of = (state(m) for m in mods)
    bufs_c, bufs_g = [None] * len(O.trainable_names(oc)), [None] * len(O.trainable_names(og))
    lr = float(g["lr"])
    cfg = O.StepConfig(trigger="wanet", lr_c=lr, lr_g=lr)
    s_img, s_lab = (int(v) for v in g["pool_seeds"])
    for s in range(6):
        i = s % int(g["pool"])
        x = synth_images(32, 32, s_img + i)
        t = torch.randint(0, 10, (32,), generator=torch.Generator().manual_seed(s_lab + i))
        field = O.grid_generator_forward(og, x[:1]).flatten()
        np.testing.assert_allclose(field.numpy(), g["trace/field"][s], atol=1e-5)
        r = O.alternated_step(oc, og, ok, of, bufs_c, bufs_g, x, t, O.StepRandomness(int(g["num_bd"][s]), 0.0, 0.0), cfg)
        for k in ("loss_c", "loss_ce", "loss_l2", "loss_grad_l2", "clean_model_loss"):
            ref = float(g["trace/" + k][s])
            assert abs(r[k] - ref) < 2e-4 * max(1.0, abs(ref)) + 1e-7, (s, k, r[k], ref)
        for k in ("clean_correct", "bd_correct", "f_correct", "clean_model_correct", "clean_model_bd_ba", "clean_model_bd_asr"):
            assert r[k] == int(g["trace/" + k][s]), (s, k)


def wanet_victim_nets(g):
    """(netc, netg) of tests/golden/victim_wanet.npz: the mirror modules under the recorded seeds, BatchNorm buffers
    and generator head randomised by the same calls as make_golden.py::golden_victim_wanet."""
    from combat_amd import nets

    class WOpt:
        s = 2

    s_c, s_g = (int(v) for v in g["seeds"])
    torch.manual_seed(s_c)
    netc = randomize_bn_buffers(nets.PreActResNet18(), int(g["bn_seed"]))
    torch.manual_seed(s_g)
    netg = nets.GridGenerator(WOpt()).eval()
    h1, h2, h3 = (int(v) for v in g["head_seeds"])
    with torch.no_grad():
        netg.fc1.bias.normal_(0, 1.0, generator=torch.Generator().manual_seed(h1))
        netg.fc2.weight.normal_(0, 0.5, generator=torch.Generator().manual_seed(h2))
        netg.fc2.bias.normal_(0, 0.5, generator=torch.Generator().manual_seed(h3))
    return netc, netg


def test_wanet_victim_step_and_eval_match_the_reference_modules(golden):
    """oracle.victim_step / oracle.eval_batch with cfg.trigger == "wanet" against one training batch and two evaluation
    batches of train_victim_wanet.py (:72-112 with the D3 intent, :150-181) recorded from the reference's PreActResNet18
    and GridGenerator through its own F.upsample / F.grid_sample calls (make_golden.py::golden_victim_wanet)."""
    from oracle import combat_oracle as O
    g = golden("victim_wanet")
    netc, netg = wanet_victim_nets(g)
    check_summary(g, "netg", [(k, v) for k, v in netg.state_dict().items() if k.startswith("fc")], 1e-6)
    sd = lambda m: {k: v.detach().clone() for k, v in m.state_dict().items()}
    oc, og = sd(netc), sd(netg)
    cfg = O.StepConfig(trigger="wanet")
    s_img, s_lab = (int(v) for v in g["eval/seeds"])
    for s, b in enumerate(int(v) for v in g["eval/batch"]):
        x = synth_images(b, 32, s_img + s)
        t = torch.randint(0, 10, (b,), generator=torch.Generator().manual_seed(s_lab + s))
        r = O.eval_batch(oc, og, x, t, 0.5, cfg)
        for k in ("clean_n", "clean_correct", "bd_n", "bd_correct", "bd_ba"):
            assert r[k] == int(g["eval/" + k][s]), (s, k, r[k], int(g["eval/" + k][s]))
    vi, vl = (int(v) for v in g["victim/seeds"])
    x = synth_images(48, 32, vi)
    t = torch.randint(0, 10, (48,), generator=torch.Generator().manual_seed(vl))
    t[:6] = 0
    pz = torch.from_numpy(g["victim/poisoned"])
    ibd, _ = O.wanet_warp(x[pz], O.grid_generator_forward(og, x[pz]), cfg.grid_rescale)
    assert float((ibd - torch.from_numpy(g["victim/inputs_bd"])).abs().max()) < 1e-5
    names = O.trainable_names(oc)
    r = O.victim_step(oc, [None] * len(names), x, t, cfg, netg=og, poisoned=pz)
    assert abs(r["loss"] - float(g["victim/loss"])) < 1e-5 and r["correct"] == int(g["victim/correct"]) and r["num_bd"] == 3
    assert abs(r["gnorm"] - float(g["victim/gnorm"])) < 1e-4 * float(g["victim/gnorm"])
    num = den = 0.0
    for k, v in zip(names, r["grads"]):
        d = v.double().flatten()[g["victim/gp/%s/idx" % k]].numpy() - g["victim/gp/%s/val" % k]
        num, den = num + float((d ** 2).sum()), den + float((g["victim/gp/%s/val" % k] ** 2).sum())
    assert (num / den) ** 0.5 < 5e-3, (num / den) ** 0.5      # (same amplification of 1e-6 image differences as the UNet victim step)
    check_summary(g, "victim/after", oc.items(), 1e-4)
