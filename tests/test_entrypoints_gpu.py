"""The reference's command-line workflow end to end on the MI355X with tiny synthetic data:
train_clean_classifier -> train_generator (fresh, then --continue_training) -> train_victim -> eval,
checking the checkpoint layout (paths and keys of reference train_generator.py:441-457, 497-499)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(script, *args, cwd):
    cmd = [sys.executable, os.path.join(ROOT, script), "--synthetic", "--synthetic_size", "256", "--bs", "64",
           "--checkpoints", os.path.join(cwd, "ckpt"), "--allow_missing_F", "--log_interval", "1"] + list(args)
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run(cmd, cwd=cwd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + "\n" + r.stderr[-3000:]
    return r.stdout


def test_reference_workflow_on_synthetic_data(tmp_path):
    cwd = str(tmp_path)
    run("train_clean_classifier.py", "--saving_prefix", "classifier_clean", "--n_iters", "1", cwd=cwd)
    clean = os.path.join(cwd, "ckpt", "classifier_clean", "cifar10", "cifar10_classifier_clean.pth.tar")
    sd = torch.load(clean, map_location="cpu", weights_only=False)
    assert set(sd) == {"netC", "schedulerC", "optimizerC", "best_clean_acc", "epoch_current"}
    assert len(sd["netC"]) == 102 and sd["netC"]["conv1.weight"].shape == (64, 3, 3, 3)

    out = run("train_generator.py", "--saving_prefix", "train_generator", "--load_checkpoint_clean", "classifier_clean",
              "--n_iters", "1", cwd=cwd)
    assert "Clean Acc:" in out and "Saving..." in out
    gen = os.path.join(cwd, "ckpt", "train_generator_clean", "cifar10", "cifar10_train_generator_clean.pth.tar")
    sd = torch.load(gen, map_location="cpu", weights_only=False)
    assert set(sd) == {"netC", "schedulerC", "optimizerC", "netG", "schedulerG", "optimizerG", "clean_model",
                       "best_clean_acc", "best_bd_acc", "best_F_acc", "best_clean_model_acc", "best_clean_model_bd_ba",
                       "best_clean_model_bd_asr", "epoch_current"}
    assert len(sd["netG"]) == 32 and len(sd["netC"]) == 102
    mom = sd["optimizerG"]["state"]
    assert len(mom) == 32 and all("momentum_buffer" in v for v in mom.values())
    assert all(torch.isfinite(v).all() for v in sd["netG"].values())
    # a torch module of the reference layout can consume the checkpoint as is
    ref_like = torch.nn.Conv2d(3, 64, 3, 2, 1)
    ref_like.load_state_dict({"weight": sd["netG"]["conv0_0.weight"], "bias": sd["netG"]["conv0_0.bias"]})

    out = run("train_generator.py", "--saving_prefix", "train_generator", "--load_checkpoint_clean", "classifier_clean",
              "--n_iters", "2", "--continue_training", cwd=cwd)
    assert "Continue training!!" in out

    run("train_victim.py", "--saving_prefix", "train_victim", "--load_checkpoint", "train_generator_clean",
        "--n_iters", "1", cwd=cwd)
    vic = os.path.join(cwd, "ckpt", "train_victim", "cifar10", "cifar10_train_victim.pth.tar")
    assert set(torch.load(vic, map_location="cpu", weights_only=False)) == {
        "netC", "schedulerC", "optimizerC", "netG", "best_clean_acc", "best_bd_acc", "epoch_current"}

    out = run("eval.py", "--saving_prefix", "train_generator", "--load_checkpoint_clean", "train_victim",
              "--load_checkpoint", "train_generator_clean", cwd=cwd)
    assert "Bd ASR:" in out


def test_celeba_workflow_on_synthetic_data(tmp_path):
    """The same three scripts with --dataset celeba (64 x 64, 8 classes, ResNet18 surrogate / clean model,
    train_generator.py:93-96): checkpoints carry the reference's ResNet18 state-dict layout (122 entries)."""
    cwd = str(tmp_path)
    run("train_clean_classifier.py", "--dataset", "celeba", "--saving_prefix", "classifier_clean", "--n_iters", "1", cwd=cwd)
    clean = os.path.join(cwd, "ckpt", "classifier_clean", "celeba", "celeba_classifier_clean.pth.tar")
    sd = torch.load(clean, map_location="cpu", weights_only=False)
    assert len(sd["netC"]) == 122 and sd["netC"]["linear.weight"].shape == (8, 2048)
    out = run("train_generator.py", "--dataset", "celeba", "--saving_prefix", "train_generator",
              "--load_checkpoint_clean", "classifier_clean", "--n_iters", "1", cwd=cwd)
    assert "Clean Acc:" in out and "Saving..." in out
    gen = os.path.join(cwd, "ckpt", "train_generator_clean", "celeba", "celeba_train_generator_clean.pth.tar")
    sd = torch.load(gen, map_location="cpu", weights_only=False)
    assert len(sd["netC"]) == 122 and len(sd["netG"]) == 32
    assert all(torch.isfinite(v).all() for v in sd["netG"].values())
    assert all(torch.isfinite(v.float()).all() for v in sd["netC"].values())
    out = run("eval.py", "--dataset", "celeba", "--saving_prefix", "train_generator", "--load_checkpoint_clean",
              "classifier_clean", "--load_checkpoint", "train_generator_clean", cwd=cwd)
    assert "Bd ASR:" in out
