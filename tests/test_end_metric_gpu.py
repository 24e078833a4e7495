"""End-metric parity (the acceptance criterion north_star ends on): clean accuracy / backdoor benign accuracy /
attack success rate of the whole pipeline -- train_clean_classifier -> train_generator -> train_victim -> eval.py --
against the same pipeline driven through the REFERENCE's nn.Modules on the CPU (tests/golden/end_metric.npz, made by
tests/golden/make_golden.py::golden_end_metric from /root/reference in the build container).

Both sides see the same bytes (combat_amd.data.synthetic_structured: a learnable class-prototype set; real CIFAR-10 is
not in the image) and the same random draws: the golden records every epoch permutation, num_bd, blur sigma and the
poisoned index set, and this test replays the repo's own train() / eval() functions of the four entry scripts on them
(the loaders' order and the two RNG draw sites are injected; everything else -- the steps, the evaluation loops, the
keep-best checkpoint logic, checkpoint files handed from script to script -- runs as shipped, --post_transform_option
no_use because the augmentation library is absent on the reference side).

What "parity" can mean here.  The alternated training is chaotic at the reference's lr = 1e-2 (DESIGN.md section 4:
two fp32 runs of the reference modules that differ only in summation order diverge within ~45 steps), so eval.py's
numbers are a DISTRIBUTION, on both sides: the golden pipeline re-run with nothing changed but the CPU thread count
gives Bd BA 84.8 / 85.2 / ... and Bd ASR 9.9 / 8.9 / ... (tests/golden/end_metric_perturbed.npz), and this path --
whose few atomics-based weight gradients reorder sums from run to run -- gives Bd ASR between 8.3 and 12.9 over six
runs of this very test.  A single-run |delta| <= 0.5 pp on BA / ASR would test luck.  So the test runs the pipeline
REPS times and compares MEANS:  |mean_ours - mean_reference| <= max(0.5 pp, 3 x standard error of the difference),
for each of eval.py's three numbers (reference eval.py:108-152).  (Measured over 11 runs of this path: Bd ASR 10.8 +- 1.9
against the four reference runs' 9.7 +- 0.9, Bd BA 84.2 +- 1.6 against 84.7 +- 0.8 -- equal within 1.5 standard errors;
a one-point shift of the attack's success under bf16 can be neither shown nor excluded with samples this small.)  Clean accuracy is converged (99.4-99.8 % on both
sides, spread 0.1 pp) and is additionally held to north_star's 0.5 pp in EVERY run.  The per-epoch counters of every
stage are printed (mid-training epochs move by tens of points per epoch and are not asserted)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class NullWriter:
    def add_scalars(self, *a, **k):
        pass

    def add_image(self, *a, **k):
        pass


REPS = 8


def _pipeline(g, tmp_path, monkeypatch, verbose):
    import config
    import eval as eval_script
    import train_clean_classifier as tcc
    import train_generator as tg
    import train_victim as tv
    from combat_amd import nets, step as step_mod, trigger
    from combat_amd.data import ArrayLoader, synthetic_structured

    n_train, n_test, bs = int(g["cfg/n_train"]), int(g["cfg/n_test"]), int(g["cfg/bs"])
    ea, eb, ec = int(g["cfg/epochs_a"]), int(g["cfg/epochs_b"]), int(g["cfg/epochs_c"])
    s_clean, s_netc, s_netg, s_victim = (int(v) for v in g["cfg/seeds"])
    signal, noise_rate = float(g["cfg/signal"]), float(g["cfg/noise_rate"])
    xtr, ytr = synthetic_structured(n_train, int(g["cfg/seed_train"]), signal=signal)
    xte, yte = synthetic_structured(n_test, int(g["cfg/seed_test"]), signal=signal)
    assert int(xtr.astype(np.int64).sum()) == int(g["data/train_sum"]) and int(xte.astype(np.int64).sum()) == int(g["data/test_sum"])

    opt = config.get_arguments().parse_args([
        "--dataset", "cifar10", "--bs", str(bs), "--post_transform_option", "no_use", "--noise_rate", str(noise_rate),
        "--checkpoints", str(tmp_path), "--log_interval", "1000", "--lr_C", str(float(g["cfg/lr"])),
        "--lr_G", str(float(g["cfg/lr"])), "--lr_clean", str(float(g["cfg/lr"]))])
    nets.configure_dataset(opt)
    opt.device = "cuda"

    class ReplayLoader(ArrayLoader):
        """ArrayLoader whose epoch permutations are the recorded ones."""

        def __init__(self, perms, *a, **k):
            super().__init__(*a, **k)
            self.perms = perms

        def epoch_order(self, epoch):
            return torch.from_numpy(np.asarray(self.perms[epoch]).astype(np.int64))

    sig = {"q": iter(())}
    monkeypatch.setattr(trigger, "sample_sigma", lambda *a, **k: float(next(sig["q"])))
    tmp_path = str(tmp_path)
    test_dl = ArrayLoader(xte, yte, bs, False)
    report = {}

    def ckpt(stage):
        opt.ckpt_folder = os.path.join(str(tmp_path), stage)
        os.makedirs(opt.ckpt_folder, exist_ok=True)
        opt.ckpt_path = os.path.join(opt.ckpt_folder, stage + ".pth.tar")
        return opt.ckpt_path

    # ---- A: train_clean_classifier.py
    path_a = ckpt("clean")
    torch.manual_seed(s_clean)
    netA, optA, schA = tcc.get_model(opt)
    dl = ReplayLoader(g["A/perm"], xtr, ytr, bs, True)
    best, correct_a = 0.0, []
    for ep in range(ea):
        tcc.train(netA, optA, schA, dl, NullWriter(), ep, opt)
        best = tcc.eval(netA, optA, schA, test_dl, best, NullWriter(), ep, opt)
        netA.eval()
        with torch.no_grad():
            correct_a.append(sum(int((netA(x.cuda()).argmax(1).cpu() == t).sum()) for x, t in test_dl))
    report["A clean correct / epoch"] = (correct_a, g["A/correct"].tolist())

    # ---- B: train_generator.py
    path_b = ckpt("generator")
    torch.manual_seed(s_netc)
    netC = nets.default_classifier(opt).to(opt.device)
    torch.manual_seed(s_netg)
    netG = nets.UnetGenerator(opt).to(opt.device)
    torch.manual_seed(3)
    netF = nets.FrequencyModel(num_classes=2, n_input=3, input_size=32).to(opt.device).eval()   # (metric only; no draw)
    clean_model = nets.default_classifier(opt).to(opt.device)
    clean_model.load_state_dict(torch.load(path_a, map_location=opt.device, weights_only=True)["netC"])
    clean_model.eval()
    sgd = lambda m: torch.optim.SGD(m.parameters(), float(g["cfg/lr"]), momentum=0.9, weight_decay=5e-4, nesterov=True)
    sch = lambda o: torch.optim.lr_scheduler.MultiStepLR(o, [100, 150], 0.1)
    optC, optG = sgd(netC), sgd(netG)
    schC, schG = sch(optC), sch(optG)
    draws = iter(zip(g["B/num_bd"].tolist(), g["B/sigma_c"].tolist(), g["B/sigma_g"].tolist()))
    monkeypatch.setattr(step_mod.AlternatedStep, "_draw",
                        lambda self, t, bt: step_mod.StepRandomness(*[f(v) for f, v in zip((int, float, float), next(draws))], [None] * 5))
    dl = ReplayLoader(g["B/perm"], xtr, ytr, bs, True)
    bests = (0.0,) * 6
    ev_b = {"clean": [], "bd": []}
    for ep in range(eb):
        tg.train(netC, optC, schC, netG, optG, schG, netF, clean_model, dl, NullWriter(), ep, opt)
        sig["q"] = iter(g["B/eval_sigma"][ep].tolist())
        before = bests
        bests = tg.eval(netC, optC, schC, netG, optG, schG, netF, clean_model, test_dl, *bests, NullWriter(), ep, opt)
        ev_b["saved" if bests != before else "kept"] = ep
    report["B best (clean acc, bd acc)"] = (bests[:2], None)
    sdB = torch.load(path_b, map_location=opt.device, weights_only=True)
    report["B saved epoch"] = (int(sdB["epoch_current"]), int(g["B/best_epoch"]))

    # ---- C: train_victim.py
    path_c = ckpt("victim")
    torch.manual_seed(s_victim)
    netV = nets.default_classifier(opt).to(opt.device)
    optV = sgd(netV)
    schV = sch(optV)
    netGv = nets.UnetGenerator(opt).to(opt.device)
    netGv.load_state_dict(sdB["netG"])
    netGv.eval()
    netGv.requires_grad_(False)
    dl = ReplayLoader(g["C/perm"], xtr, ytr, bs, True, poisoned=g["C/poisoned"])
    train_sig = iter(g["C/sigma"].tolist())
    best_c = best_b = 0.0
    for ep in range(ec):
        sig["q"] = train_sig
        tv.train(netV, optV, schV, netGv, dl, NullWriter(), ep, opt)
        sig["q"] = iter(g["C/eval_sigma"][ep].tolist())
        best_c, best_b = tv.eval(netV, optV, schV, netGv, test_dl, best_c, best_b, NullWriter(), ep, opt)
    sdC = torch.load(path_c, map_location=opt.device, weights_only=True)
    report["C saved epoch"] = (int(sdC["epoch_current"]), int(g["C/best_epoch"]))
    report["C best clean acc"] = (best_c, float(g["C/eval_clean"].max()) * 100.0 / n_test)

    # ---- D: eval.py on the victim's checkpoint
    netD = nets.default_classifier(opt).to(opt.device)
    netD.load_state_dict(sdC["netC"])
    netD.eval()
    sig["q"] = iter(g["D/eval_sigma"].tolist())
    acc_clean, acc_ba, acc_asr = eval_script.eval(netD, netGv, test_dl, NullWriter(), opt)
    bd_n = int(g["D/bd_n"])
    ref = (int(g["D/clean"]) * 100.0 / n_test, int(g["D/bd_ba"]) * 100.0 / bd_n, int(g["D/bd_asr"]) * 100.0 / bd_n)
    report["D clean acc / Bd BA / Bd ASR"] = ((acc_clean, acc_ba, acc_asr), ref)
    if verbose:
        for k, v in report.items():
            print("end metric | %-32s ours %s   reference %s" % (k, v[0], v[1]))
    return (acc_clean, acc_ba, acc_asr), ref, correct_a[-1]


@pytest.mark.parametrize("fixture", ["end_metric", "end_metric_attack"])
def test_end_metrics_match_the_reference_pipeline(golden, tmp_path, monkeypatch, fixture):
    """fixture "end_metric": the reference's default --noise_rate 0.08, 6 + 6 + 8 epochs (fast; the trigger is a tenth of
    this set's class signal and the victim never learns it: Bd ASR = chance on both sides, so the comparison pins the
    clean path and the plumbing).  fixture "end_metric_attack" (round 4): --noise_rate 0.3, 6 + 12 + 12
    epochs -- the regime in which the attack TAKES: the reference modules' own runs give clean accuracy 99.6-99.8 %
    and Bd ASR 54-79 % (tests/golden/end_metric_attack*.npz), so the same statistical comparison now pins the trigger
    path's training dynamics (generator, low-pass, clamp-mix, blur, poisoned-victim training) at the end metric."""
    g, gp = golden(fixture), golden(fixture + "_perturbed")
    n_test, bd_n = int(g["cfg/n_test"]), int(g["D/bd_n"])
    # the reference sample: the recorded run (8 threads) + the perturbed re-runs
    refs = np.array([(int(g["D/clean"]) * 100.0 / n_test, int(g["D/bd_ba"]) * 100.0 / bd_n, int(g["D/bd_asr"]) * 100.0 / bd_n)] +
                    [(c * 100.0 / n_test, ba * 100.0 / n, asr * 100.0 / n) for c, ba, asr, n in
                     zip(gp["runs/clean"], gp["runs/bd_ba"], gp["runs/bd_asr"], gp["runs/bd_n"])])
    ours = []
    for rep in range(REPS):
        d = tmp_path / ("rep%d" % rep)
        d.mkdir()
        (m, ref, last_a) = _pipeline(g, d, monkeypatch, verbose=rep == 0)
        ours.append(m)
        print("end metric | run %d: clean acc %.3f  Bd BA %.3f  Bd ASR %.3f" % ((rep,) + m))
        # converged, every run: north_star's 0.5-pp bound against the reference's runs (they differ among themselves by
        # 0.2 pp with nothing but the thread count changed; a run of this path is one more draw, e.g. 99.22 once in 33
        # runs against 99.6-99.8 otherwise: within 0.5 pp of the nearest reference run)
        assert np.abs(refs[:, 0] - m[0]).min() <= 0.5, ("clean acc of a single run", m[0], refs[:, 0].tolist())
        assert abs(last_a - int(g["A/correct"][-1])) * 100.0 / n_test <= 1.0
    assert tuple(ref) == tuple(refs[0])
    ours = np.array(ours)
    for j, name in enumerate(("clean acc", "Bd BA", "Bd ASR")):
        mo, mr = ours[:, j].mean(), refs[:, j].mean()
        se = np.sqrt(ours[:, j].var(ddof=1) / len(ours) + refs[:, j].var(ddof=1) / len(refs))
        tol = max(0.5, 3.0 * se)
        print("end metric | %-9s ours %.3f +- %.3f (n=%d)   reference %.3f +- %.3f (n=%d)   |delta| %.3f   tolerance %.3f" % (
            name, mo, ours[:, j].std(ddof=1), len(ours), mr, refs[:, j].std(ddof=1), len(refs), abs(mo - mr), tol))
        assert abs(mo - mr) <= tol, (name, mo, mr, tol)
    if fixture == "end_metric_attack":      # the regime this fixture exists for: the backdoor works, on both sides
        assert refs[:, 2].mean() >= 50.0 and ours[:, 2].mean() >= 50.0, (refs[:, 2].tolist(), ours[:, 2].tolist())
