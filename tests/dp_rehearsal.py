"""Worker of tests/test_entrypoints_gpu.py::test_data_parallel_step_two_ranks_one_gpu (not a test module).

Launched as `python -m torch.distributed.run --nproc-per-node 2 tests/dp_rehearsal.py OUT` with
COMBAT_DIST_BACKEND=gloo: two ranks share the one GPU of the box (RCCL refuses two ranks on one device; the
exchange goes over gloo, everything else -- plans, marks, bucketed all-reduce launches, auxiliary-stream
joins, optimiser scaling -- is the code path of an 8-GPU run).  Every rank writes OUT/rank<r>.json."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class Opt:
    noise_rate, ratio, kernel_size, sigma = 0.08, 0.65, 3, (0.1, 1.0)
    pc, target_label, attack_mode, num_classes = 0.5, 0, "all2one", 10
    L2_weight, clean_model_weight, lr_C, lr_G = 0.02, 0.8, 1e-2, 1e-2
    input_height = input_width = 32
    dataset, post_transform_option, random_crop, random_rotation = "cifar10", "no_use", 5, 10


def build(nets):
    out = []
    for seed, ctor in ((0, nets.PreActResNet18), (1, nets.PreActResNet18), (2, lambda: nets.UnetGenerator(None)),
                       (3, lambda: nets.FrequencyModel(2, 3, 32))):
        torch.manual_seed(seed)
        out.append(ctor().cuda())
    netc, clean, netg, netf = out
    return netc, netg, clean.eval(), netf.eval()


def rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / max(float(b.norm()), 1e-30))


def main():
    out_dir = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    from combat_amd import nets, step as step_mod
    b = 32
    gen = torch.Generator().manual_seed(900 + rank)          # every rank its own shard
    x = ((torch.randint(0, 256, (b, 3, 32, 32), generator=gen, dtype=torch.uint8).float() / 255) - 0.5) / 0.5
    t = torch.randint(0, 10, (b,), generator=gen)
    t[:3] = 0
    rnd = [step_mod.StepRandomness(2 + rank, 0.4 + 0.1 * rank, 0.7, [None] * 5),
           step_mod.StepRandomness(1, 0.6, 0.5 + 0.2 * rank, [None] * 5)]
    res = {}

    # ---- single-rank run of this rank's first step from the common start state (no process group)
    netc, netg, clean, netf = build(nets)
    p0 = {k: v.detach().clone() for k, v in netc.named_parameters()}
    st1 = step_mod.AlternatedStep(netc, netg, clean, netf, Opt())
    st1.keep_grads = True
    st1.run(x.cuda(), t, rnd[0])
    torch.cuda.synchronize()
    g_single = st1.eC.fp.grad.detach().clone()
    del st1, netc, netg, clean, netf

    # ---- the data-parallel run: 2 steps
    netc, netg, clean, netf = build(nets)
    st = step_mod.AlternatedStep(netc, netg, clean, netf, Opt(), process_group=dist.group.WORLD)
    st.keep_grads = True
    st.run(x.cuda(), t, rnd[0])
    torch.cuda.synchronize()
    g_dp = st.eC.fp.grad.detach().clone()                       # sum over ranks (the mean is folded into the optimiser)
    gg_dp = st.eG.fp.grad.detach().clone()
    singles = [torch.zeros_like(g_single).cpu() for _ in range(world)]
    dist.all_gather(singles, g_single.cpu())
    res["gradC_sum_vs_singles"] = rel(g_dp.cpu(), singles[0] + singles[1])
    # the optimiser consumed the MEAN: p1 = p0 - lr * (1 + mu) * (mean_g + wd * p0)   (first step, nesterov)
    fp = st.eC.fp
    worst = 0.0
    for k in ("conv1.weight", "layer2.0.shortcut.0.weight", "layer4.1.conv2.weight", "linear.weight"):
        gk = fp.logical((singles[0] + singles[1]).cuda() * 0.5, k)
        exp = p0[k] - 1e-2 * 1.9 * (gk + 5e-4 * p0[k])
        worst = max(worst, rel(dict(netc.named_parameters())[k].detach(), exp))
    res["paramC_update_vs_mean_grad"] = worst
    both = [torch.zeros_like(gg_dp).cpu() for _ in range(world)]
    dist.all_gather(both, gg_dp.cpu())
    res["gradG_identical_across_ranks"] = bool(torch.equal(both[0], both[1]))
    st.run(x.cuda(), t, rnd[1])
    torch.cuda.synchronize()
    ident = True
    for eng in (st.eC, st.eG):
        for buf in (eng.fp.flat, eng.fp.mom):
            got = [torch.zeros_like(buf).cpu() for _ in range(world)]
            dist.all_gather(got, buf.cpu())
            ident = ident and bool(torch.equal(got[0], got[1]))
    res["replicas_bit_identical_after_2_steps"] = ident
    res["finite"] = bool(all(np.isfinite(v) for v in st.read_metrics().values()))
    res["len_loader_equal"] = True
    del st, netc, netg, clean, netf

    # ---- the other data-parallel steps: ClassifierStep behind a UNet and behind a GridGenerator (bucketed
    # all-reduce at the plan's marks), WanetStep (surrogate through its marks + the grid head's gradient range)
    class WOpt(Opt):
        s, grid_rescale = 2, 0.15

    def gather_equal(buf):
        got = [torch.zeros_like(buf).cpu() for _ in range(world)]
        dist.all_gather(got, buf.cpu())
        return bool(torch.equal(got[0], got[1]))

    pz = torch.zeros(b, dtype=torch.bool)
    pz[:3] = True                                               # three poisoned target-class images per rank
    for tag, gen_ctor, opt in (("clf_unet", lambda: nets.UnetGenerator(None), Opt()),
                               ("clf_grid", lambda: nets.GridGenerator(WOpt()), WOpt()),
                               ("clf_clean", None, Opt())):
        def fresh():
            torch.manual_seed(0)
            c = nets.PreActResNet18().cuda()
            g_ = None
            if gen_ctor is not None:
                torch.manual_seed(2)
                g_ = gen_ctor().cuda().eval()
            return c, g_
        netc, netg = fresh()
        torch.manual_seed(555 + rank)                           # (the blur sigma is drawn from torch's global stream)
        s1 = step_mod.ClassifierStep(netc, opt, netg)
        s1.run(x.cuda(), t, pz if netg is not None else None)
        torch.cuda.synchronize()
        g_single = s1.eC.fp.grad.detach().clone()
        del s1, netc, netg
        netc, netg = fresh()
        torch.manual_seed(555 + rank)
        sd = step_mod.ClassifierStep(netc, opt, netg, process_group=dist.group.WORLD)
        sd.run(x.cuda(), t, pz if netg is not None else None)
        torch.cuda.synchronize()
        singles = [torch.zeros_like(g_single).cpu() for _ in range(world)]
        dist.all_gather(singles, g_single.cpu())
        res[tag + "_grad_sum_vs_singles"] = rel(sd.eC.fp.grad.cpu(), singles[0] + singles[1])
        res[tag + "_grad_identical_across_ranks"] = gather_equal(sd.eC.fp.grad)
        sd.run(x.cuda(), t, pz if netg is not None else None)
        torch.cuda.synchronize()
        res[tag + "_replicas_bit_identical_after_2_steps"] = gather_equal(sd.eC.fp.flat) and gather_equal(sd.eC.fp.mom)
        del sd, netc, netg

    torch.manual_seed(0)
    netc = nets.PreActResNet18().cuda()
    torch.manual_seed(1)
    clean = nets.PreActResNet18().cuda().eval()
    torch.manual_seed(2)
    netg = nets.GridGenerator(WOpt()).cuda()
    torch.manual_seed(3)
    netf = nets.FrequencyModel(2, 3, 32).cuda().eval()
    sw = step_mod.WanetStep(netc, netg, clean, netf, WOpt(), process_group=dist.group.WORLD)
    sw.keep_grads = True
    sw.run(x.cuda(), t, rnd[0])
    torch.cuda.synchronize()
    lo, hi = sw.eG.head_grad_range()
    res["wanet_head_grad_identical_across_ranks"] = gather_equal(sw.eG.fp.grad[lo:hi])
    res["wanet_head_grad_nonzero"] = bool(float(sw.eG.fp.grad[lo:hi].abs().sum()) > 0)
    outside = torch.cat([sw.eG.fp.grad[:lo], sw.eG.fp.grad[hi:]])
    res["wanet_grad_outside_head_range_is_zero"] = bool(float(outside.abs().max()) == 0.0) if outside.numel() else True
    res["wanet_gradC_identical_across_ranks"] = gather_equal(sw.eC.fp.grad)
    sw.run(x.cuda(), t, rnd[1])
    torch.cuda.synchronize()
    res["wanet_replicas_bit_identical_after_2_steps"] = all(
        gather_equal(buf) for eng in (sw.eC, sw.eG) for buf in (eng.fp.flat, eng.fp.mom))
    with open(os.path.join(out_dir, "rank%d.json" % rank), "w") as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
