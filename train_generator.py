"""Alternated training of the trigger generator and the surrogate classifier on MI355X.

Drop-in for the reference script of the same name: same flags (config.py), same
``get_model`` / ``train`` / ``eval`` / ``main`` call signatures, same checkpoint path and keys
(reference train_generator.py:80-128, 131-318, 321-465, 468-609) -- the per-batch loop body
(:170-290) runs as ``combat_amd.step.AlternatedStep`` on the HIP kernels.  CIFAR-10 + the default
default classifier of the dataset (PreActResNet18 for cifar10, ResNet18 for celeba) / UNet / "original" detector;
other --model / --dataset values raise.

Data parallel: ``python -m torch.distributed.run --nproc-per-node N train_generator.py ...`` gives
every rank a disjoint shard of each epoch and averages the gradients over RCCL (combat_amd.dist).
"""
import os
import random
import time

import numpy as np
import torch

import config
from combat_amd import api, dist as cdist
from combat_amd.data import get_dataloader
from combat_amd.log import SummaryWriter, image_grid, progress_bar
from combat_amd.nets import FrequencyModel, UnetGenerator, configure_dataset, default_classifier
from combat_amd.step import AlternatedStep, WanetStep, create_targets_bd  # noqa: F401  (re-exported like the reference)


def create_dir(path_dir):
    os.makedirs(path_dir, exist_ok=True)


def get_model(opt):
    if opt.model != "default" or opt.model_clean != "default" or opt.F_model not in ("original", "original_holdout"):
        raise Exception("only the default classifier / UNet / 'original' detector run on the HIP path")
    netC = default_classifier(opt).to(opt.device)
    clean_model = default_classifier(opt).to(opt.device)
    netG = UnetGenerator(opt).to(opt.device)
    netF = FrequencyModel(num_classes=2, n_input=opt.input_channel, input_size=opt.input_height).to(opt.device)
    optimizerC = torch.optim.SGD(netC.parameters(), opt.lr_C, momentum=0.9, weight_decay=5e-4, nesterov=True)
    schedulerC = torch.optim.lr_scheduler.MultiStepLR(optimizerC, opt.schedulerC_milestones, opt.schedulerC_lambda)
    optimizerG = torch.optim.SGD(netG.parameters(), opt.lr_G, momentum=0.9, weight_decay=5e-4, nesterov=True)
    schedulerG = torch.optim.lr_scheduler.MultiStepLR(optimizerG, opt.schedulerG_milestones, opt.schedulerG_lambda)
    return netC, optimizerC, schedulerC, netG, optimizerG, schedulerG, netF, clean_model


def _step_of(netC, netG, clean_model, netF, opt) -> AlternatedStep:
    st = netC.__dict__.get("_alt_step")
    if st is None:
        pg = torch.distributed.group.WORLD if torch.distributed.is_initialized() else None
        cls = WanetStep if getattr(netG, "arch", "") == "gridgen" else AlternatedStep
        st = cls(netC, netG, clean_model, netF, opt, process_group=pg)
        netC.__dict__["_alt_step"] = st
    return st


def train(netC, optimizerC, schedulerC, netG, optimizerG, schedulerG, netF, clean_model, train_dl, tf_writer, epoch, opt):
    print(" Train:")
    netC.train()
    netG.train()
    clean_model.eval()
    st = _step_of(netC, netG, clean_model, netF, opt)
    st.reset_metrics()
    n_batches = len(train_dl)
    every = max(1, int(getattr(opt, "log_interval", 20)))
    m = None
    for batch_idx, (inputs, targets) in enumerate(train_dl):
        st.run(inputs.to(opt.device, non_blocking=True), targets,
               lr_c=optimizerC.param_groups[0]["lr"], lr_g=optimizerG.param_groups[0]["lr"])
        last = batch_idx == n_batches - 1 or (opt.max_steps and batch_idx + 1 >= opt.max_steps)
        if batch_idx % every == 0 or last:   # the reference formats device tensors every step (one sync each)
            m = st.read_metrics()
            ts = m["samples"]
            progress_bar(
                batch_idx, n_batches,
                "Clean Acc: {:.4f} | Bd Acc: {:.4f} | F Acc: {:.4f} | Clean Model Acc: {:.4f} | Clean Model Bd BA: {:.4f} "
                "| Clean Model Bd ASR: {:.4f}".format(
                    m["clean_correct"] * 100.0 / ts, m["bd_correct"] * 100.0 / ts, m["f_correct"] * 100.0 / ts,
                    m["clean_model_correct"] * 100.0 / ts, m["clean_model_bd_ba"] * 100.0 / ts,
                    m["clean_model_bd_asr"] * 100.0 / ts))
        if last:
            break
    ts = m["samples"]
    if not epoch % 1:
        tf_writer.add_scalars("Clean Accuracy", {
            "Clean": m["clean_correct"] * 100.0 / ts, "Bd": m["bd_correct"] * 100.0 / ts, "F": m["f_correct"] * 100.0 / ts,
            "CleanModel Acc": m["clean_model_correct"] * 100.0 / ts,
            "CleanModel Bd BA": m["clean_model_bd_ba"] * 100.0 / ts,
            "CleanModel Bd ASR": m["clean_model_bd_asr"] * 100.0 / ts,
            "L2 Loss": m["loss_l2_sum"] / ts, "Grad L2 Loss": m["loss_grad_l2_sum"] / ts,
            "CleanModel Loss": m["clean_model_loss_sum"] / ts}, epoch)
    if not epoch % 20 and not isinstance(tf_writer, cdist.NullWriter):    # :310-315: the last batch and its backdoored copy (Phase G's inputs_bd) as an image grid
        tf_writer.add_image("Images", image_grid(st.inputs, st.bd, opt), global_step=epoch)
    schedulerC.step()
    schedulerG.step()


def eval(netC, optimizerC, schedulerC, netG, optimizerG, schedulerG, netF, clean_model, test_dl, best_clean_acc, best_bd_acc,
         best_F_acc, best_clean_model_acc, best_clean_model_bd_ba, best_clean_model_bd_asr, tf_writer, epoch, opt):
    print(" Eval:")
    netC.eval()
    cdist.average_bn_buffers(netC)    # data parallel: one model on every rank and in the checkpoint (combat_amd/dist.py)
    netG.eval()
    clean_model.eval()
    c = dict(clean_n=0, bd_n=0, clean=0, bd=0, F=0, cm=0, cm_ba=0, cm_asr=0)
    for batch_idx, (inputs, targets) in enumerate(test_dl):
        with torch.no_grad():
            inputs, targets = inputs.to(opt.device), targets.to(opt.device)
            preds_clean = netC(inputs)
            c["clean_n"] += len(inputs)
            ntrg = (targets != opt.target_label).nonzero()[:, 0]
            inputs_toChange, targets_toChange = inputs[ntrg], targets[ntrg]
            inputs_bd = api.create_backdoor(netG, inputs_toChange, opt)     # random sigma at eval too (:353,:373)
            targets_bd = create_targets_bd(targets_toChange, opt).to(opt.device)
            c["bd_n"] += len(ntrg)
            # the six counters of this batch in ONE device->host read (the reference reads each one separately)
            cnt = [(preds_clean.argmax(1) == targets).sum(), (clean_model(inputs).argmax(1) == targets).sum()]
            if len(ntrg):
                preds_bd = netC(inputs_bd)
                cm_bd = clean_model(inputs_bd).argmax(1)
                cnt += [(preds_bd.argmax(1) == targets_bd).sum(),
                        (api.frequency_logits(netF, inputs_bd, opt).argmax(1) == 1).sum(),
                        (cm_bd == targets_toChange).sum(), (cm_bd == targets_bd).sum()]
            cnt = torch.stack(cnt).cpu().tolist()
            c["clean"] += int(cnt[0])
            c["cm"] += int(cnt[1])
            if len(ntrg):
                c["bd"] += int(cnt[2])
                c["F"] += int(cnt[3])
                c["cm_ba"] += int(cnt[4])
                c["cm_asr"] += int(cnt[5])
        acc_clean = c["clean"] * 100.0 / c["clean_n"]
        bd_n = max(c["bd_n"], 1)
        acc_bd, acc_F = c["bd"] * 100.0 / bd_n, c["F"] * 100.0 / bd_n
        acc_clean_model = c["cm"] * 100.0 / c["clean_n"]
        bd_ba_clean_model, bd_asr_clean_model = c["cm_ba"] * 100.0 / bd_n, c["cm_asr"] * 100.0 / bd_n
        progress_bar(batch_idx, len(test_dl),
                     "Clean Acc: {:.4f} - Best: {:.4f} | Bd Acc: {:.4f} - Best: {:.4f} | F Acc: {:.4f} - Best: {:.4f} | "
                     "Clean Model Acc: {:.4f} - Best: {:.4f} | Clean Model Bd BA: {:.4f} - Best: {:.4f} | "
                     "Clean Model Bd ASR: {:.4f} - Best: {:.4f}".format(
                         acc_clean, best_clean_acc, acc_bd, best_bd_acc, acc_F, best_F_acc, acc_clean_model,
                         best_clean_model_acc, bd_ba_clean_model, best_clean_model_bd_ba, bd_asr_clean_model,
                         best_clean_model_bd_asr))
    if torch.distributed.is_initialized():   # every rank evaluated its shard of the test set
        vals = cdist.all_reduce_counters([c[k] for k in sorted(c)], device=opt.device)
        c = dict(zip(sorted(c), vals))
        bd_n = max(c["bd_n"], 1)
        acc_clean, acc_bd, acc_F = c["clean"] * 100.0 / c["clean_n"], c["bd"] * 100.0 / bd_n, c["F"] * 100.0 / bd_n
        acc_clean_model = c["cm"] * 100.0 / c["clean_n"]
        bd_ba_clean_model, bd_asr_clean_model = c["cm_ba"] * 100.0 / bd_n, c["cm_asr"] * 100.0 / bd_n
    if not epoch % 1:
        tf_writer.add_scalars("Test Accuracy", {
            "Clean": acc_clean, "Bd": acc_bd, "F": acc_F, "Clean Model Acc": acc_clean_model,
            "Clean Model Bd BA": bd_ba_clean_model, "Clean Model Bd ASR": bd_asr_clean_model}, epoch)
    if acc_clean > best_clean_acc or (acc_clean == best_clean_acc and acc_bd > best_bd_acc):
        print(" Saving...")
        best_clean_acc, best_bd_acc, best_F_acc = acc_clean, acc_bd, acc_F
        best_clean_model_acc, best_clean_model_bd_ba = acc_clean_model, bd_ba_clean_model
        best_clean_model_bd_asr = bd_asr_clean_model
        if int(os.environ.get("RANK", 0)) == 0:
            api.sync_momentum_to_optimizer(optimizerC, netC)
            api.sync_momentum_to_optimizer(optimizerG, netG)
            torch.save({
                "netC": netC.state_dict(), "schedulerC": schedulerC.state_dict(), "optimizerC": optimizerC.state_dict(),
                "netG": netG.state_dict(), "schedulerG": schedulerG.state_dict(), "optimizerG": optimizerG.state_dict(),
                "clean_model": clean_model.state_dict(), "best_clean_acc": acc_clean, "best_bd_acc": acc_bd,
                "best_F_acc": acc_F, "best_clean_model_acc": best_clean_model_acc,
                "best_clean_model_bd_ba": best_clean_model_bd_ba, "best_clean_model_bd_asr": best_clean_model_bd_asr,
                "epoch_current": epoch}, opt.ckpt_path)
    return (best_clean_acc, best_bd_acc, best_F_acc, best_clean_model_acc, best_clean_model_bd_ba,
            best_clean_model_bd_asr)


def detector_checkpoint_path(opt):
    """The reference looks under <F_checkpoints>/<dataset>/<F_model>/ (:504-507) while its shipped
    files sit one level up as <dataset>_original_detector.pth.tar: probe both."""
    folder = os.path.join(opt.F_checkpoints, opt.dataset)
    name = "{}_{}_detector.pth.tar".format(opt.dataset, opt.F_model)
    for p in (os.path.join(folder, opt.F_model, name), os.path.join(folder, name)):
        if os.path.exists(p):
            return p
    return os.path.join(folder, opt.F_model, name)


def main(get_model=None, train=None, eval=None):
    """``get_model`` / ``train`` / ``eval`` default to this module's; train_generator_wanet.py passes its own."""
    get_model = get_model or globals()["get_model"]
    train = train or globals()["train"]
    eval = eval or globals()["eval"]
    opt = config.get_arguments().parse_args()
    configure_dataset(opt)
    rank, local_rank, world = cdist.init()
    if opt.device == "cuda":
        opt.device = "cuda:%d" % local_rank
    if opt.seed is not None:
        torch.manual_seed(opt.seed)
        np.random.seed(opt.seed + rank)
        random.seed(opt.seed + rank)

    train_dl = get_dataloader(opt, True, rank=rank, world=world)
    test_dl = get_dataloader(opt, False, shuffle=False, rank=rank, world=world)
    netC, optimizerC, schedulerC, netG, optimizerG, schedulerG, netF, clean_model = get_model(opt)

    mode = opt.saving_prefix
    opt.ckpt_folder = os.path.join(opt.checkpoints, "{}_clean".format(mode), opt.dataset)
    opt.ckpt_path = os.path.join(opt.ckpt_folder, "{}_{}_clean.pth.tar".format(opt.dataset, mode))
    opt.log_dir = os.path.join(opt.ckpt_folder, "log_dir")

    opt.F_ckpt_path = detector_checkpoint_path(opt)
    print(f"Loading {opt.F_model} at {opt.F_ckpt_path}")
    if os.path.exists(opt.F_ckpt_path):
        netF.load_state_dict(torch.load(opt.F_ckpt_path, map_location=opt.device, weights_only=True)["netC"])
    elif not opt.allow_missing_F:
        print("Error: {} not found (pass --allow_missing_F to run with an untrained detector)".format(opt.F_ckpt_path))
        exit()
    netF.eval()
    print("Done")

    load_path = os.path.join(opt.checkpoints, opt.load_checkpoint_clean or "", opt.dataset,
                             "{}_{}.pth.tar".format(opt.dataset, opt.load_checkpoint_clean))
    if not os.path.exists(load_path):
        print("Error: {} not found".format(load_path))
        exit()
    clean_model.load_state_dict(torch.load(load_path, map_location=opt.device, weights_only=True)["netC"])
    clean_model.eval()

    if opt.continue_training:
        if not os.path.exists(opt.ckpt_path):
            print("Pretrained model doesnt exist")
            exit()
        print("Continue training!!")
        sd = torch.load(opt.ckpt_path, map_location=opt.device, weights_only=True)
        netC.load_state_dict(sd["netC"])
        optimizerC.load_state_dict(sd["optimizerC"])
        schedulerC.load_state_dict(sd["schedulerC"])
        netG.load_state_dict(sd["netG"])
        optimizerG.load_state_dict(sd["optimizerG"])
        schedulerG.load_state_dict(sd["schedulerG"])
        clean_model.load_state_dict(sd["clean_model"])
        api.load_momentum_from_optimizer(optimizerC, netC)
        api.load_momentum_from_optimizer(optimizerG, netG)
        best = [sd[k] for k in ("best_clean_acc", "best_bd_acc", "best_F_acc", "best_clean_model_acc",
                                "best_clean_model_bd_ba", "best_clean_model_bd_asr")]
        epoch_current = sd["epoch_current"]
    else:
        print("Train from scratch!!!")
        best = [0.0] * 6
        epoch_current = 0
        cdist.fresh_start(opt.ckpt_folder, rank)   # the reference wipes the folder too (:562); rank 0 only, then a barrier
    if world > 1:
        for m in (netC, netG, clean_model, netF):
            cdist.broadcast_module(m)
    if rank == 0:    # one writer, one checkpoint file: rank 0's (every rank holds the same replicas)
        create_dir(opt.log_dir)
        tf_writer = SummaryWriter(log_dir=opt.log_dir)
    else:
        tf_writer = cdist.NullWriter()

    train_dl.epoch = epoch_current      # a seeded, resumed run continues the sequence of epoch permutations
    for epoch in range(epoch_current, opt.n_iters):
        print("Epoch {}:".format(epoch + 1))
        t0 = time.perf_counter()
        train(netC, optimizerC, schedulerC, netG, optimizerG, schedulerG, netF, clean_model, train_dl, tf_writer, epoch, opt)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        best = list(eval(netC, optimizerC, schedulerC, netG, optimizerG, schedulerG, netF, clean_model, test_dl, *best,
                         tf_writer, epoch, opt))
        print(" train {:.2f} s, eval + checkpoint {:.2f} s".format(t1 - t0, time.perf_counter() - t1))


if __name__ == "__main__":
    main()
