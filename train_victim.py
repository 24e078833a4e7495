"""Victim training on data poisoned by the frozen generator (reference train_victim.py:93-165 loop,
:168-231 eval, :221-229 checkpoint keys; dataset flags from utils/dataloader_cleanbd.py:131-158)."""
import os
import time

import torch

import config
from combat_amd import api, dist as cdist
from combat_amd.data import get_dataloader
from combat_amd.log import SummaryWriter, image_grid, progress_bar
from combat_amd.nets import UnetGenerator, configure_dataset, default_classifier
from combat_amd.step import ClassifierStep, create_targets_bd


def get_model(opt):
    netC = default_classifier(opt).to(opt.device)
    netG = UnetGenerator(opt).to(opt.device)
    optimizerC = torch.optim.SGD(netC.parameters(), opt.lr_C, momentum=0.9, weight_decay=5e-4, nesterov=True)
    schedulerC = torch.optim.lr_scheduler.MultiStepLR(optimizerC, opt.schedulerC_milestones, opt.schedulerC_lambda)
    return netC, optimizerC, schedulerC, netG


def train(netC, optimizerC, schedulerC, netG, train_dl, tf_writer, epoch, opt):
    print(" Train:")
    netC.train()
    step = netC.__dict__.get("_clf_step")
    if step is None:
        pg = torch.distributed.group.WORLD if torch.distributed.is_initialized() else None
        step = netC.__dict__["_clf_step"] = ClassifierStep(netC, opt, netG, process_group=pg)
    if step.N:
        step.read_metrics(reset=True)
    total = 0
    pair = None
    for batch_idx, (inputs, targets, poisoned) in enumerate(train_dl):
        step.run(inputs.to(opt.device, non_blocking=True), targets, poisoned, lr=optimizerC.param_groups[0]["lr"])
        total += inputs.shape[0]
        if step.wanet and (not batch_idx % 5 or pair is None):   # train_victim_wanet.py:127-133: every fifth batch with a poisoned
            pair = step.poisoned_pair() or pair                   # image (until one is found: any batch, so that an epoch logs a grid)
        last = batch_idx == len(train_dl) - 1 or (opt.max_steps and batch_idx + 1 >= opt.max_steps)
        if batch_idx % max(1, opt.log_interval) == 0 or last:
            m = step.read_metrics()
            progress_bar(batch_idx, len(train_dl), "CE Loss: {:.4f} | Clean Acc: {:.4f}".format(
                m["loss_sum"] / total, m["correct"] * 100.0 / total))
        if last:
            break
    tf_writer.add_scalars("Clean Accuracy", {"Clean": m["correct"] * 100.0 / total}, epoch)
    if step.wanet and pair is not None and not isinstance(tf_writer, cdist.NullWriter):   # train_victim_wanet.py:136
        tf_writer.add_image("Images", image_grid(pair[0], pair[1], opt), global_step=epoch)
    schedulerC.step()


def eval(netC, optimizerC, schedulerC, netG, test_dl, best_clean_acc, best_bd_acc, tf_writer, epoch, opt):
    print(" Eval:")
    netC.eval()
    cdist.average_bn_buffers(netC)    # data parallel: one model on every rank and in the checkpoint (combat_amd/dist.py)
    n = nb = correct = bd = 0
    for batch_idx, batch in enumerate(test_dl):
        inputs, targets = batch[0].to(opt.device), batch[1].to(opt.device)
        with torch.no_grad():
            correct += int((netC(inputs).argmax(1) == targets).sum())
            n += len(inputs)
            ntrg = (targets != opt.target_label).nonzero()[:, 0]
            if len(ntrg):
                inputs_bd = api.create_backdoor(netG, inputs[ntrg], opt)
                targets_bd = create_targets_bd(targets[ntrg], opt).to(opt.device)
                bd += int((netC(inputs_bd).argmax(1) == targets_bd).sum())
                nb += len(ntrg)
        acc_clean, acc_bd = correct * 100.0 / n, bd * 100.0 / max(nb, 1)
        progress_bar(batch_idx, len(test_dl), "Clean Acc: {:.4f} - Best: {:.4f} | Bd Acc: {:.4f} - Best: {:.4f}".format(
            acc_clean, best_clean_acc, acc_bd, best_bd_acc))
    if torch.distributed.is_initialized():   # every rank evaluated its shard of the test set
        n, nb, correct, bd = cdist.all_reduce_counters([n, nb, correct, bd], device=opt.device)
        acc_clean, acc_bd = correct * 100.0 / n, bd * 100.0 / max(nb, 1)
    tf_writer.add_scalars("Test Accuracy", {"Clean": acc_clean, "Bd": acc_bd}, epoch)
    if acc_clean > best_clean_acc:
        print(" Saving...")
        best_clean_acc, best_bd_acc = acc_clean, acc_bd
        if int(os.environ.get("RANK", 0)) == 0:
            api.sync_momentum_to_optimizer(optimizerC, netC)
            state = {"netC": netC.state_dict(), "schedulerC": schedulerC.state_dict(), "optimizerC": optimizerC.state_dict(),
                     "netG": netG.state_dict(), "best_clean_acc": acc_clean, "best_bd_acc": acc_bd, "epoch_current": epoch}
            if getattr(netG, "arch", "") == "gridgen":          # train_victim_wanet.py:199
                state["grid_rescale"] = opt.grid_rescale
            torch.save(state, opt.ckpt_path)
    return best_clean_acc, best_bd_acc


def load_generator(netG, load_path, opt):
    """The frozen generator of the attack (:262-280).  A checkpoint of the other trigger family is refused by name
    instead of by a state-dict key mismatch deep inside torch."""
    sd = torch.load(load_path, map_location=opt.device, weights_only=True)["netG"]
    is_grid = any(k.startswith("fc1.") for k in sd)
    want_grid = getattr(netG, "arch", "") == "gridgen"
    if is_grid != want_grid:
        raise SystemExit("Error: {} holds a {} generator; use {}".format(
            load_path, "GridGenerator (WaNet)" if is_grid else "UnetGenerator",
            "train_victim_wanet.py" if is_grid else "train_victim.py"))
    netG.load_state_dict(sd)
    netG.eval()
    netG.requires_grad_(False)      # :279-280


def main(get_model=None, wanet=False):
    get_model = get_model or globals()["get_model"]
    opt = config.get_arguments().parse_args()
    configure_dataset(opt)
    rank, local_rank, world = cdist.init()
    if opt.device == "cuda":
        opt.device = "cuda:%d" % local_rank
    if opt.seed is not None:
        import random
        import numpy as np
        torch.manual_seed(opt.seed)
        np.random.seed(opt.seed + rank)
        random.seed(opt.seed)     # the poisoned index set must be the same on every rank (it is also broadcast)
    train_dl = get_dataloader(opt, True, poisoned=True, rank=rank, world=world)
    test_dl = get_dataloader(opt, False, shuffle=False, poisoned=True, rank=rank, world=world)
    netC, optimizerC, schedulerC, netG = get_model(opt)
    # train_victim.py:255-257 saves under <prefix>/, train_victim_wanet.py:241-243 under <prefix>_clean/
    mode = "{}_clean".format(opt.saving_prefix) if wanet else opt.saving_prefix
    opt.ckpt_folder = os.path.join(opt.checkpoints, mode, opt.dataset)
    opt.ckpt_path = os.path.join(opt.ckpt_folder, "{}_{}.pth.tar".format(opt.dataset, mode))
    opt.log_dir = os.path.join(opt.ckpt_folder, "log_dir")
    load_path = os.path.join(opt.checkpoints, opt.load_checkpoint, opt.dataset,
                             "{}_{}.pth.tar".format(opt.dataset, opt.load_checkpoint))
    if not os.path.exists(load_path):
        print("Error: {} not found".format(load_path))
        exit()
    load_generator(netG, load_path, opt)
    best_clean_acc = best_bd_acc = 0.0
    epoch_current = 0
    if opt.continue_training and os.path.exists(opt.ckpt_path):
        sd = torch.load(opt.ckpt_path, map_location=opt.device, weights_only=True)
        netC.load_state_dict(sd["netC"])
        optimizerC.load_state_dict(sd["optimizerC"])
        schedulerC.load_state_dict(sd["schedulerC"])
        api.load_momentum_from_optimizer(optimizerC, netC)
        best_clean_acc, best_bd_acc, epoch_current = sd["best_clean_acc"], sd["best_bd_acc"], sd["epoch_current"]
    else:
        cdist.fresh_start(opt.ckpt_folder, rank)
    if world > 1:
        cdist.broadcast_module(netC)
    if rank == 0:
        os.makedirs(opt.log_dir, exist_ok=True)
        tf_writer = SummaryWriter(log_dir=opt.log_dir)
    else:
        tf_writer = cdist.NullWriter()
    train_dl.epoch = epoch_current      # a seeded, resumed run continues the sequence of epoch permutations
    for epoch in range(epoch_current, opt.n_iters):
        print("Epoch {}:".format(epoch + 1))
        t0 = time.perf_counter()
        train(netC, optimizerC, schedulerC, netG, train_dl, tf_writer, epoch, opt)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        best_clean_acc, best_bd_acc = eval(netC, optimizerC, schedulerC, netG, test_dl, best_clean_acc, best_bd_acc,
                                           tf_writer, epoch, opt)
        print(" train {:.2f} s, eval + checkpoint {:.2f} s".format(t1 - t0, time.perf_counter() - t1))


if __name__ == "__main__":
    main()
