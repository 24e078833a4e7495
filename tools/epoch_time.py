"""End-to-end rate of the entry script (loader, progress bar, eval, checkpoint included): trains a clean classifier for a
few batches, then runs train_generator.py for 2 epochs of 200 synthetic batches and reports wall time per training step."""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = tempfile.mkdtemp()
common = [sys.executable, None, "--synthetic", "--bs", "128", "--checkpoints", os.path.join(d, "ckpt"), "--allow_missing_F"]


def run(script, *args):
    cmd = list(common)
    cmd[1] = os.path.join(ROOT, script)
    t0 = time.perf_counter()
    r = subprocess.run(cmd + list(args), cwd=d, env=dict(os.environ, PYTHONPATH=ROOT), capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return time.perf_counter() - t0, r.stdout


def epochs(out):
    return "\n".join(l for l in out.replace("\r", "\n").splitlines() if "train " in l and " s, eval" in l)


_, out = run("train_clean_classifier.py", "--synthetic_size", "25600", "--saving_prefix", "classifier_clean", "--n_iters", "3")
print("train_clean_classifier.py (200 train + 200 eval batches per epoch):\n" + epochs(out))
t1, _ = run("train_generator.py", "--synthetic_size", "25600", "--saving_prefix", "g1", "--load_checkpoint_clean", "classifier_clean", "--n_iters", "1")
t3, out = run("train_generator.py", "--synthetic_size", "25600", "--saving_prefix", "g3", "--load_checkpoint_clean", "classifier_clean", "--n_iters", "3")
per_epoch = (t3 - t1) / 2      # start-up (imports, first-launch set-up) cancels
print("train_generator.py: %.2f s per epoch of 200 train + 200 eval batches of 128 (1 epoch %.1f s, 3 epochs %.1f s)" % (per_epoch, t1, t3))
print(epochs(out))
_, out = run("train_victim.py", "--synthetic_size", "25600", "--saving_prefix", "victim", "--load_checkpoint", "g3_clean", "--n_iters", "3")
print("train_victim.py:\n" + epochs(out))
_, out = run("train_generator_wanet.py", "--synthetic_size", "25600", "--saving_prefix", "w3", "--load_checkpoint_clean", "classifier_clean", "--n_iters", "2")
print("train_generator_wanet.py:\n" + epochs(out))
