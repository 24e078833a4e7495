#!/usr/bin/env python3
"""Throughput of COMBAT's alternated generator+surrogate step on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (train_generator.py:170-290: Phase C + Phase G, both SGD
updates, and for N > 1 both gradient all-reduces) over one synthetic CIFAR-10-shaped batch of 128
images per GPU that is already resident in HBM.  `--gpus N` without a torchrun environment starts the
N ranks itself (a child `python -m torch.distributed.run ...`, spawned before anything touches the GPU).
Prints ONE JSON line on rank 0 (contract in the task statement): whole-job images/sec, plus

  roofline      the dominant kernel (the convolution kernel family with the most device time):
                algorithmic FLOPs of its launches / their HIP-event durations, measured in a second,
                instrumented replay of the same K steps (events on the launch stream), against the
                2.5 PFLOP/s dense bf16 MFMA peak; `per_shape` = every PreActResNet18 convolution shape
                (classifier_models/preact_resnet.py:21,23,27-29,77) x {fwd, dgrad, wgrad}: us, TFLOP/s, frac;
  cpu_baseline  the CPU oracle (oracle/combat_oracle.py, the fp32 restatement of the reference step,
                as written incl. its discarded work) timed on this host's cores on a bounded sample;
  pytorch_rocm_baseline
                the same restatement run by stock PyTorch-ROCm on this GPU (ATen + MIOpen), fp32 and bf16
                autocast: BASELINE.json configs[1]'s "vs PyTorch-ROCm baseline" row.

Before anything is timed the first two steps at B = 128 (augmentation off, recorded num_bd / sigma) are
checked against tests/golden/step_b128.npz -- losses recorded from the reference's own modules on the same
batches and seeds -- and the run aborts on a mismatch: the number is for a shape whose results are checked.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "images/sec per alternated generator+surrogate step, CIFAR-10 bs=128, 1/2/4/8 GPU"
PEAK_BF16_TFLOPS = 2500.0
STEP_GFLOP = 1493.95      # algorithmic minimum of one alternated step at B = 128
def _pmc_traffic_files():
    """profiles/rNN_<tag>_pmc_hbm_traffic.csv, newest (highest round, then tag) first."""
    import glob
    return [os.path.basename(p) for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic.csv")), reverse=True)]
TILE_NAMES = {1: "conv_gemm_kernel<128,128>", 2: "conv_gemm_kernel<128,64>", 3: "conv_gemm_kernel<64,64>",
              4: "conv_gemm_kernel<128,16>", 5: "conv_gemm_kernel<64,128>",
              6: "conv3x3_halo1_kernel<256,64>", 7: "conv3x3_halo1_kernel<128,128>", 8: "conv3x3_halo1_kernel<128,64>",
              9: "conv3x3_halo_kernel<64,64>", 10: "conv3x3_dma_kernel<64>", 11: "conv3x3_dma_kernel<32>",
              12: "conv_gather_dma_kernel<64>", 13: "conv_gather_dma_kernel<32>", 14: "conv3x3_dma_kernel<64,256px>",
              15: "conv_c8_kernel", 16: "conv3x3_dma_kernel<64,256px,w64>", 17: "conv3x3_ws_kernel<64>"}


def log(msg):
    if int(os.environ.get("RANK", 0)) == 0:
        print("[bench] " + msg, file=sys.stderr, flush=True)


class Opt:
    """config.py defaults of the reference for `--dataset cifar10 --pc 0.5 --noise_rate 0.08`."""
    dataset = "cifar10"
    input_height = input_width = 32
    input_channel, num_classes, bs = 3, 10, 128
    noise_rate, ratio, kernel_size, sigma = 0.08, 0.65, 3, (0.1, 1.0)
    pc, target_label, attack_mode = 0.5, 0, "all2one"
    L2_weight, clean_model_weight, lr_C, lr_G = 0.02, 0.8, 1e-2, 1e-2
    post_transform_option, random_crop, random_rotation = "use", 5, 10


def synth_batches(n_batches, bs, rank, device, hw=32, classes=10):
    """uint8 pixels -> ToTensor + Normalize(0.5, 0.5); labels uniform."""
    g = torch.Generator().manual_seed(1234 + rank)
    out = []
    for _ in range(n_batches):
        u8 = torch.randint(0, 256, (bs, 3, hw, hw), generator=g, dtype=torch.uint8)
        x = ((u8.float() / 255) - 0.5) / 0.5
        t = torch.randint(0, classes, (bs,), generator=g)
        out.append((x.to(device), t))
    return out


def build_nets(device, dataset="cifar10"):
    from combat_amd import nets
    clf = nets.PreActResNet18 if dataset == "cifar10" else (lambda: nets.ResNet18(num_classes=8, input_size=64))
    torch.manual_seed(0)
    netc = clf()
    torch.manual_seed(1)
    clean = clf().eval()
    torch.manual_seed(2)
    netg = nets.UnetGenerator(None)
    torch.manual_seed(3)
    netf = nets.FrequencyModel(2, 3, 32 if dataset == "cifar10" else 64).eval()   # shipped detector weights do not travel: default init
    return netc.to(device), netg.to(device), clean.to(device), netf.to(device)


# --------------------------------------------------------------------------------------------- parity gate


def golden_gate(device):
    """Two alternated steps at the benchmarked shape against the trace recorded from the reference modules
    (tests/golden/step_b128.npz: bench batches 0 and 1, network seeds 0..3, augmentation off).  Tolerance as
    in tests/test_engine_gpu.py: 2e-2 * max(1, |ref|) per loss (bf16 activations vs fp32), loss_l2 2 % rel."""
    from combat_amd import step as step_mod
    path = os.path.join(ROOT, "tests", "golden", "step_b128.npz")
    if not os.path.exists(path):
        return {"checked": False, "why": "tests/golden/step_b128.npz absent"}
    g = dict(np.load(path))
    opt = Opt()
    opt.post_transform_option = "no_use"
    netc, netg, clean, netf = build_nets(device)
    st = step_mod.AlternatedStep(netc, netg, clean, netf, opt)
    batches = synth_batches(2, opt.bs, 0, device)
    report = {"checked": True, "reference": "tests/golden/step_b128.npz (reference modules, fp32)", "steps": []}
    ok = True
    for s in range(2):
        x, t = batches[s]
        assert abs(float(x.double().sum()) - float(g["step%d/x_sum" % s])) < 1e-6, "synthetic batch differs from the golden's"
        st.reset_metrics()
        st.run(x, t, step_mod.StepRandomness(int(g["num_bd"][s]), float(g["sigma_c"][s]), float(g["sigma_g"][s]), [None] * 5))
        torch.cuda.synchronize()
        m = st.read_metrics()
        row = {}
        for ours, key, rel in (("loss_c_sum", "loss_c", False), ("loss_ce_sum", "loss_ce", False),
                               ("clean_model_loss_sum", "clean_model_loss", False), ("loss_l2_sum", "loss_l2", True)):
            r = float(g["trace/" + key][s])
            tol = 2e-2 * (abs(r) if rel else max(1.0, abs(r))) * (1.0 if s == 0 else 2.0)   # step 2 starts from two bf16 updates
            row[key] = [round(m[ours], 6), round(r, 6)]
            ok = ok and abs(m[ours] - r) < tol
        report["steps"].append(row)
    report["ok"] = bool(ok)
    return report


# --------------------------------------------------------------------------------------------- roofline


def conv_flops(a):
    """Algorithmic FLOPs of one conv launch: 2 * output pixels of the convolution * Cout * Cin * taps
    with the REAL channel counts (padding channels and the masked taps of a strided dgrad are not work)."""
    pc = a._keepalive[2]
    pix = a.N * a.P * a.Q if a.mode == 0 else a.N * a.H * a.W
    fl = 2.0 * pix * pc.K * pc.c_real * pc.taps
    if a.src2:      # the block's 1x1 shortcut rides along (combat_conv_args.src2): its input gradient's work too
        pc2 = a._keepalive[15][1]
        fl += 2.0 * pix * pc2.K * pc2.c_real * pc2.taps
    return fl


def wgrad_flops(a):
    return 2.0 * a.N * a.P * a.Q * a.k_real * a.c_real * a.R * a.S


def pmc_traffic(kernel_prefix):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE x 2
    per the gfx950 correction + WRITE_SIZE, separate --pmc runs): PMC counters cannot be collected from
    inside this process.  (bytes, file) or (None, None) if no summary is present."""
    for name in _pmc_traffic_files():
        path = os.path.join(ROOT, "profiles", name)
        try:
            tot = cnt = 0.0
            for line in open(path):
                if line.startswith("#") or line.startswith("kernel,"):
                    continue
                f = line.rstrip("\n").rsplit(",", 6)
                if kernel_prefix in f[0]:
                    tot += float(f[1]) * float(f[6]) * 1024.0
                    cnt += float(f[1])
            if cnt:
                return round(tot / cnt), name
        except OSError:
            continue
    return None, None


def shape_key(src_hw, c, k, r, stride):
    return "%dx%d s%d %d->%d @%dx%d" % (r, r, stride, c, k, src_hw, src_hw)


def roofline_from(prof):
    from combat_amd._lib import ConvArgs, lib
    import ctypes
    groups, per_shape, per_shape_unet = {}, {}, {}
    for what, a, e0, e1 in prof:
        sec = e0.elapsed_time(e1) * 1e-3
        if isinstance(a, ConvArgs):
            fl = conv_flops(a)
            tile = lib.combat_conv_pick_tile(ctypes.byref(a))
            g = groups.setdefault(tile, [0.0, 0.0, 0])
            g[0] += fl
            g[1] += sec
            g[2] += 1
            pc = a._keepalive[2]
            # the convolution's input map: src of a forward launch, dst (= dx) of an input-gradient launch
            key = shape_key(a.H if a.mode == 0 else a.P, pc.c_real, pc.K, pc.R, pc.stride)
            kind = "fwd" if a.mode == 0 else "dgrad"
        else:
            key, kind = shape_key(a.H, a.c_real, a.k_real, a.R, a.stride), "wgrad"
            if what.endswith(".reduce"):     # the deferred reduction of a weight gradient: time, no work, no launch of its own
                tab = per_shape if what.startswith("preact.") else (per_shape_unet if what.startswith("unet.") else None)
                if tab is not None:
                    tab.setdefault(key, {}).setdefault(kind, [0.0, 0.0, 0])[1] += sec
                continue
            fl = wgrad_flops(a)
        tab = per_shape if what.startswith("preact.") else (per_shape_unet if what.startswith("unet.") else None)
        if tab is not None:
            d = tab.setdefault(key, {}).setdefault(kind, [0.0, 0.0, 0])
            d[0] += fl
            d[1] += sec
            d[2] += 1
    # the dominant kernel = the __global__ template with the most device time, over all its tile
    # instantiations (rocprofv3 lists each instantiation; "all_conv_tiles" below does too)
    fams = {}
    for t, (f, s_, c) in groups.items():
        fam = TILE_NAMES.get(t, str(t)).split("<")[0]
        g = fams.setdefault(fam, [0.0, 0.0, 0])
        g[0] += f
        g[1] += s_
        g[2] += c
    fam, (fl, sec, cnt) = max(fams.items(), key=lambda kv: kv[1][1])
    achieved = fl / sec / 1e12
    traffic, traffic_file = pmc_traffic(fam + "<")
    def table(tab):
        return {key: {kind: {"launches": c, "us": round(s_ / c * 1e6, 2), "tflops": round(f / s_ / 1e12, 1),
                             "frac": round(f / s_ / 1e12 / PEAK_BF16_TFLOPS, 4)}
                      for kind, (f, s_, c) in sorted(kinds.items())} for key, kinds in sorted(tab.items())}
    shapes, shapes_unet = table(per_shape), table(per_shape_unet)
    traffic_pass_ms = None
    if traffic_file:        # the bench line of the profiling pass the counters were collected in (same file prefix)
        try:
            with open(os.path.join(ROOT, "profiles", traffic_file.replace("_pmc_hbm_traffic.csv", "_bench.json"))) as f:
                traffic_pass_ms = json.loads([l for l in f.read().splitlines() if l.startswith("{")][-1]).get("ms_per_step")
        except (OSError, ValueError, IndexError):
            pass
    return {
        "bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
        "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
        "traffic": traffic,
        "traffic_unit": "HBM bytes per launch, launch-weighted over the instantiations (rocprofv3 PMC passes, "
                        "profiles/%s)" % traffic_file,
        "traffic_source": {"file": "profiles/%s" % traffic_file if traffic_file else None,
                           "pass_ms_per_step": traffic_pass_ms,
                           "note": "counters come from a committed profiling pass of this same command, not from this "
                                   "run: PMC collection needs rocprofv3 around the process"},
        "kernel": fam + " (all tile instantiations)", "launches": cnt, "avg_launch_us": round(sec / cnt * 1e6, 2),
        "gflop_per_launch": round(fl / cnt / 1e9, 3),
        "all_conv_tiles": {TILE_NAMES.get(t, str(t)): {"launches": c, "avg_us": round(s_ / c * 1e6, 2),
                                                      "tflops": round(f / s_ / 1e12, 1)}
                           for t, (f, s_, c) in sorted(groups.items())},
        "per_shape": shapes,
        "per_shape_unet": shapes_unet,
        "per_shape_unet_note": "the UNet generator's convolutions (networks/models.py:275-314), same keys; its 3-channel "
                               "output layer and first layer count their REAL channels (3), so their fractions are "
                               "small by construction",
        "per_shape_note": "PreActResNet18 convolutions (surrogate + clean model plans), keyed 'RxS stride Cin->Cout "
                          "@input HxW'; wgrad = weight-gradient launch incl. its partial-sum reduction launch; the 1x1 "
                          "stride-2 shortcuts' input gradients ride along in the 3x3 stride-2 dgrad launches (second "
                          "reduction source), their FLOPs are counted there",
    }


# --------------------------------------------------------------------------------------------- baselines


def _oracle_state(seed_ctor):
    return {k: v.clone() for k, v in seed_ctor().state_dict().items()}


def _oracle_nets(device=None):
    from combat_amd import nets
    out = []
    for seed, ctor in ((0, nets.PreActResNet18), (1, nets.PreActResNet18), (2, lambda: nets.UnetGenerator(None)),
                       (3, lambda: nets.FrequencyModel(2, 3, 32))):
        torch.manual_seed(seed)
        sd = _oracle_state(ctor)
        out.append({k: (v.to(device) if device is not None else v) for k, v in sd.items()})
    return out


def _aug_draw(rng, bs):
    from oracle import combat_oracle as O
    return O.AugParams(rng.integers(0, 11, bs).astype(np.int32), rng.integers(0, 11, bs).astype(np.int32),
                       np.where(rng.random(bs) < 0.5, rng.uniform(-10, 10, bs), 0).astype(np.float32),
                       (rng.random(bs) < 0.5).astype(np.int32))


def cpu_baseline(seconds_budget=25.0):
    """The oracle's step, as the reference writes it (loss.backward() into every leaf, all five
    forwards building graphs), fp32, anomaly detection off, on this host's cores."""
    from oracle import combat_oracle as O
    # the GPU box gives one GPU a share of 16 host cores (os.cpu_count() reports the whole host):
    # use the affinity mask, capped at that share, so the CPU leg is not oversubscribed
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    threads = int(os.environ.get("COMBAT_CPU_THREADS", min(cores, 16)))
    torch.set_num_threads(threads)
    netc, clean, netg, netf = _oracle_nets()
    bufs_c, bufs_g = [None] * len(O.trainable_names(netc)), [None] * len(O.trainable_names(netg))
    g = torch.Generator().manual_seed(1234)
    bs = 128
    x = ((torch.randint(0, 256, (bs, 3, 32, 32), generator=g, dtype=torch.uint8).float() / 255) - 0.5) / 0.5
    t = torch.randint(0, 10, (bs,), generator=g)
    rng = np.random.default_rng(0)

    def one():
        n_trg = int((t == 0).sum())
        rnd = O.StepRandomness(int(np.sum(rng.random(n_trg) < 0.5)), 0.5, 0.6, [_aug_draw(rng, bs) for _ in range(5)])
        O.alternated_step(netc, netg, clean, netf, bufs_c, bufs_g, x, t, rnd, O.StepConfig(), as_written=True)

    log("cpu_baseline: %d threads, warm-up step" % threads)
    one()  # warm-up (allocator, MKL-DNN primitive caches)
    t0 = time.perf_counter()
    steps = 0
    while True:
        one()
        steps += 1
        el = time.perf_counter() - t0
        log("cpu_baseline: step %d done, %.1f s" % (steps, el))
        if el > seconds_budget or steps >= 8:
            break
    return {"value": round(bs * steps / el, 2), "unit": "images/sec", "cores": threads, "kind": "port",
            "sample": "%d alternated steps of 128 images after 1 warm-up (%.1f s), fp32 torch-CPU oracle of the "
                      "reference step as written, set_detect_anomaly off" % (steps, el)}


def pytorch_rocm_baseline(device, steps=10, budget_s=90.0):
    """BASELINE.json configs[1] "vs PyTorch-ROCm baseline": the reference step as written (the oracle's
    functional restatement: same ATen operator sequence as train_generator.py:170-255, all dead work included,
    anomaly detection off) executed by stock PyTorch-ROCm on this GPU -- convolutions by MIOpen -- in fp32
    and under bf16 autocast, batched augmentation, same batch / seeds as the HIP path.  A baseline leg: the
    oracle is the thing timed here, never the product."""
    from oracle import combat_oracle as O
    out = {"unit": "images/sec", "kind": "stock PyTorch-ROCm (ATen + MIOpen), reference step as written, 1 GPU",
           "torch": torch.__version__}
    bs = 128
    g = torch.Generator().manual_seed(1234)
    x = (((torch.randint(0, 256, (bs, 3, 32, 32), generator=g, dtype=torch.uint8).float() / 255) - 0.5) / 0.5).to(device)
    t = torch.randint(0, 10, (bs,), generator=g).to(device)
    n_trg = int((t == 0).sum())
    for tag, autocast in (("fp32", False), ("bf16_autocast", True)):
        netc, clean, netg, netf = _oracle_nets(device)
        bufs_c, bufs_g = [None] * len(O.trainable_names(netc)), [None] * len(O.trainable_names(netg))
        rng = np.random.default_rng(0)

        def one():
            # num_bd fixed at its expectation (6 of ~13 target-class images): every new poisoned-sub-batch size is
            # a new set of convolution shapes for MIOpen to select / compile kernels for, which a 10-step sample
            # would mostly measure (302 ms/step with num_bd drawn per step, vs the figure reported here)
            rnd = O.StepRandomness(min(6, n_trg), 0.5, 0.6, [_aug_draw(rng, bs) for _ in range(5)])
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
                O.alternated_step(netc, netg, clean, netf, bufs_c, bufs_g, x, t, rnd, O.StepConfig(), as_written=True,
                                  aug_fn=O.post_tensor_transform_batched)

        t_start = time.perf_counter()
        try:
            for _ in range(3):      # warm-up: MIOpen solver selection / kernel compilation happens here
                one()
            torch.cuda.synchronize()
            warm = time.perf_counter() - t_start
            log("pytorch_rocm_baseline[%s]: warm-up %.1f s" % (tag, warm))
            if warm > budget_s:
                out[tag] = {"value": None, "why": "warm-up alone took %.0f s (MIOpen kernel compilation)" % warm}
                continue
            t0 = time.perf_counter()
            for _ in range(steps):
                one()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            out[tag] = {"value": round(bs * steps / el, 1), "ms_per_step": round(el / steps * 1e3, 2), "steps": steps}
            log("pytorch_rocm_baseline[%s]: %.2f ms/step" % (tag, el / steps * 1e3))
        except Exception as e:   # a baseline leg must never take the headline measurement down with it
            out[tag] = {"value": None, "why": "%s: %s" % (type(e).__name__, str(e)[:200])}
        del netc, clean, netg, netf, bufs_c, bufs_g
        torch.cuda.empty_cache()
    return out


# --------------------------------------------------------------------------------------------- main


def spawn_ranks(args):
    """`python bench.py --gpus N` outside torchrun: start the N ranks as a CHILD process (nothing in this
    process has touched the GPU yet -- a process that has must never exec) and exit with its status."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("no torchrun environment: starting %d ranks: %s" % (args.gpus, " ".join(cmd[1:9])))
    return subprocess.run(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-torch-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-golden-gate", action="store_true")
    ap.add_argument("--dataset", default="cifar10", choices=("cifar10", "celeba"),
                    help="cifar10 = BASELINE configs[1] (the metric's configuration); celeba = configs[3]'s shape "
                         "(64 x 64, 8 classes, ResNet18), informational")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    from combat_amd import dist as cdist_
    cdist_.limit_host_threads()      # (the cpu_baseline leg sets its own count)
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    # COMBAT_DIST_BACKEND=gloo: rehearsal of the multi-rank path on a box with fewer GPUs than ranks (RCCL
    # refuses two ranks on one device); the contract's runs use nccl = RCCL, one rank per GPU
    backend = os.environ.get("COMBAT_DIST_BACKEND", "nccl")
    local_rank = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    pg = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=device)
        else:
            torch.distributed.init_process_group(backend)
        pg = torch.distributed.group.WORLD

    from combat_amd import step as step_mod
    gate = None
    if not args.no_golden_gate and args.dataset == "cifar10" and rank == 0:
        gate = golden_gate(device)
        log("golden gate (B=128 vs reference-module trace): %s" % json.dumps(gate))
        if gate.get("checked") and not gate["ok"]:
            raise SystemExit("bench.py: results at the benchmarked shape differ from tests/golden/step_b128.npz: %s" % gate)
    opt = Opt()
    if args.dataset == "celeba":
        opt.dataset, opt.input_height, opt.input_width, opt.num_classes = "celeba", 64, 64, 8
    np.random.seed(rank)
    import random
    random.seed(rank)
    torch.manual_seed(100 + rank)
    netc, netg, clean, netf = build_nets(device, args.dataset)
    if world > 1:   # identical replicas: rank 0's parameters everywhere
        for m in (netc, netg, clean, netf):
            for p in list(m.parameters()) + list(m.buffers()):
                torch.distributed.broadcast(p.data, 0)
    st = step_mod.AlternatedStep(netc, netg, clean, netf, opt, process_group=pg)
    batches = synth_batches(8, opt.bs, rank, device, opt.input_height, opt.num_classes)

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    log("setup done; %d warm-up steps" % args.warmup)
    for i in range(args.warmup):
        x, t = batches[i % len(batches)]
        st.run(x, t)
        if i == 0:
            torch.cuda.synchronize()
            log("first step done")
    sync()
    log("warm-up done; timing %d steps" % args.steps)
    t0 = time.perf_counter()
    for i in range(args.steps):
        x, t = batches[i % len(batches)]
        st.run(x, t)
    host_s = time.perf_counter() - t0          # launches enqueued (the device is still working them off)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tt)
    log("timed region: %.3f s (%.3f ms/step; host thread inside run() %.3f ms/step -- it runs ahead of the device until the "
        "hardware queues fill and then moves at the device's pace)" % (elapsed, elapsed / args.steps * 1e3, host_s / args.steps * 1e3))
    # the host's own cost of enqueueing a step: the device is parked behind a spin kernel, so no call waits for queue space
    # (three steps = ~2 100 queue packets fit); outside the timed region
    host_ms = None
    if world == 1 and not args.no_roofline:      # (not in profiling runs: the spin kernel would sit in their traces)
        torch.cuda._sleep(int(2.0e9 * 0.04))
        h0 = time.perf_counter()
        for i in range(3):
            x, t = batches[i % len(batches)]
            st.run(x, t)
        host_ms = (time.perf_counter() - h0) / 3 * 1e3
        torch.cuda.synchronize()
        log("host enqueue %.3f ms/step (device parked: the cost of the %s replay + the step's own host work)" % (
            host_ms, "C (combat_plan_run)" if os.environ.get("COMBAT_PLAN_PY", "0") != "1" else "Python"))
    metrics = st.read_metrics()
    finite = all(np.isfinite(v) for v in metrics.values())

    roof = None
    if not args.no_roofline:
        # (every rank replays -- the step contains collectives -- rank 0 reports)
        # kernel-level numbers: replay with every launch in line on one stream (the timed region above
        # overlaps three streams, which stretches every kernel it brackets by what runs beside it)
        from combat_amd.engine import Plan
        prof = []
        st.serial = Plan.serial = True
        for i in range(min(args.steps, 20)):
            x, t = batches[i % len(batches)]
            st.run(x, t, prof=prof)
        torch.cuda.synchronize()
        st.serial = Plan.serial = False
        if rank == 0:
            roof = roofline_from(prof)
            if args.dataset == "cifar10":   # the WHOLE step against the MFMA peak: SURVEY 8(d)'s 1 493.95 GFLOP per 128-image step
                ms = elapsed / args.steps * 1e3
                roof["step_frac"] = round(STEP_GFLOP * world / (ms * 1e-3) / 1e3 / (PEAK_BF16_TFLOPS * world), 4)
                roof["step_tflops"] = round(STEP_GFLOP / (ms * 1e-3) / 1e3, 1)
                roof["step_frac_note"] = ("%.2f GFLOP algorithmic per 128-image step (SURVEY 8(d)) / ms_per_step of the timed "
                                          "region / 2.5 PFLOP/s, per GPU" % STEP_GFLOP)
            roof["replay"] = "serial (one stream), HIP events around each convolution / weight-gradient launch, %d steps" % min(args.steps, 20)
            log("instrumented replay done: %s %.1f TFLOP/s" % (roof["kernel"], roof["achieved"]))
    if world > 1:
        torch.distributed.barrier()

    if rank == 0:
        out = {
            "metric": METRIC, "value": round(opt.bs * world * args.steps / elapsed, 2), "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "train_generator.py --dataset cifar10 --pc 0.5 --noise_rate 0.08: alternated "
                                   "PreActResNet18 surrogate + UNet generator step (Phase C + Phase G, both "
                                   "Nesterov-SGD updates), bs=128 per GPU, on-device augmentation on",
                       "per_gpu_batch": opt.bs, "global_batch": opt.bs * world,
                       "parallelism": "dp%d (RCCL all-reduce of netC grads in Phase C, netG grads in Phase G)" % world,
                       "gflop_per_image_algorithmic": 11.67, "losses_finite": finite,
                       "golden_gate": gate, "host_enqueue_ms_per_step": None if host_ms is None else round(host_ms, 3)},
            "roofline": roof,
        }
        if args.dataset != "cifar10":
            out["config"]["workload"] = ("train_generator.py --dataset celeba: alternated ResNet18 surrogate + UNet generator "
                                         "step at 64 x 64, 8 classes, bs=128 per GPU (informational; the metric's "
                                         "configuration is cifar10)")
            out["config"].pop("gflop_per_image_algorithmic", None)
        if world == 1 and args.dataset == "cifar10":
            del st, netc, netg, clean, netf
            torch.cuda.empty_cache()
            if not args.no_torch_baseline:
                out["pytorch_rocm_baseline"] = pytorch_rocm_baseline(device)
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
