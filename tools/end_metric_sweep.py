#!/usr/bin/env python3
"""Regime search for the end-metric fixture: which (signal, noise_rate, epochs) make the attack
TAKE on the synthetic structured set -- Bd ASR well above chance -- so that tests/golden/end_metric*.npz pin the
trigger path's training dynamics and not two chance-level noise distributions.

Runs the repo's own pipeline (tests/test_end_metric_gpu.py::_pipeline: train_clean_classifier -> train_generator ->
train_victim -> eval.py, as shipped) on the GPU with draws generated here in the order
tests/golden/make_golden.py::golden_end_metric consumes them -- the reference side is NOT involved (a config costs
seconds here and ~1 h of CPU there); the chosen configuration is then handed to make_golden.py.

    python tools/end_metric_sweep.py "signal=0.05,epochs_b=20,epochs_c=12" "signal=0.05,noise_rate=0.2" ...
"""
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

BASE = dict(n_train=2048, n_test=1024, bs=128, epochs_a=6, epochs_b=6, epochs_c=8, lr=1e-2, noise_rate=0.08, signal=0.15,
            seed_train=1234, seed_test=4321, seeds=dict(clean=11, netc=12, netg=13, victim=14), draw_seed=777)


def draws_for(cfg):
    """Every random draw of golden_end_metric, in its order (they depend on the labels only, not on any network)."""
    from combat_amd.data import synthetic_structured
    xtr, ytr = synthetic_structured(cfg["n_train"], cfg["seed_train"], signal=cfg["signal"])
    xte, yte = synthetic_structured(cfg["n_test"], cfg["seed_test"], signal=cfg["signal"])
    n, bs = cfg["n_train"], cfg["bs"]
    g = np.random.default_rng(cfg["draw_seed"])
    out = {"cfg/" + k: np.int64(v) for k, v in cfg.items() if isinstance(v, int)}
    out["cfg/lr"], out["cfg/noise_rate"], out["cfg/signal"] = np.float64(cfg["lr"]), np.float64(cfg["noise_rate"]), np.float64(cfg["signal"])
    out["cfg/seeds"] = np.array([cfg["seeds"][k] for k in ("clean", "netc", "netg", "victim")])
    out["data/train_sum"], out["data/test_sum"] = np.int64(xtr.astype(np.int64).sum()), np.int64(xte.astype(np.int64).sum())
    test_nt = [int((yte[i:i + bs] != 0).sum()) for i in range(0, cfg["n_test"], bs)]
    out["A/perm"] = np.stack([g.permutation(n) for _ in range(cfg["epochs_a"])])
    out["A/correct"] = np.zeros(cfg["epochs_a"], np.int64)
    perms, nbs, sc, sg, evs = [], [], [], [], []
    for ep in range(cfg["epochs_b"]):
        perm = g.permutation(n)
        perms.append(perm)
        for i in range(0, n, bs):
            t = ytr[perm[i:i + bs]]
            nbs.append(int(np.sum(g.random(int((t == 0).sum())) < 0.5)))
            sc.append(float(g.uniform(0.1, 1.0)))
            sg.append(float(g.uniform(0.1, 1.0)))
        evs.append([float(g.uniform(0.1, 1.0)) for _ in test_nt])
    out["B/perm"], out["B/num_bd"], out["B/sigma_c"], out["B/sigma_g"] = np.stack(perms), np.array(nbs), np.array(sc), np.array(sg)
    out["B/eval_sigma"], out["B/best_epoch"] = np.array(evs), np.int64(-1)
    ids = np.array([i for i, l in enumerate(ytr.tolist()) if l == 0])
    flags = np.zeros(n, np.bool_)
    flags[g.choice(ids, size=int(0.5 * len(ids)), replace=False)] = True
    out["C/poisoned"] = flags
    perms, sig, evs = [], [], []
    for ep in range(cfg["epochs_c"]):
        perm = g.permutation(n)
        perms.append(perm)
        for i in range(0, n, bs):
            if flags[perm[i:i + bs]].any():
                sig.append(float(g.uniform(0.1, 1.0)))
        evs.append([float(g.uniform(0.1, 1.0)) for _ in test_nt])
    out["C/perm"], out["C/sigma"], out["C/eval_sigma"] = np.stack(perms), np.array(sig), np.array(evs)
    out["C/best_epoch"], out["C/eval_clean"] = np.int64(-1), np.zeros(cfg["epochs_c"], np.int64)
    out["D/eval_sigma"] = np.array([float(g.uniform(0.1, 1.0)) for _ in test_nt])
    out["D/bd_n"] = np.int64(sum(test_nt))
    out["D/clean"] = out["D/bd_ba"] = out["D/bd_asr"] = np.int64(0)
    return out


def parse(spec):
    cfg = dict(BASE)
    for kv in filter(None, spec.split(",")):
        k, v = kv.split("=")
        cfg[k] = type(BASE[k])(v) if not isinstance(BASE[k], int) else int(v)
    return cfg


def main():
    from _pytest.monkeypatch import MonkeyPatch
    from test_end_metric_gpu import _pipeline
    reps = int(os.environ.get("SWEEP_REPS", 1))
    for spec in sys.argv[1:] or [""]:
        cfg = parse(spec)
        g = draws_for(cfg)
        for rep in range(reps):
            t0 = time.time()
            with tempfile.TemporaryDirectory() as d:
                mp = MonkeyPatch()
                try:
                    (acc, ba, asr), _, _ = _pipeline(g, d, mp, verbose=False)
                finally:
                    mp.undo()
            print("sweep | %-60s run %d: clean acc %.2f  Bd BA %.2f  Bd ASR %.2f   (%.0f s)" % (spec or "(base)", rep, acc, ba, asr,
                                                                                           time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
