"""Plain supervised training of the classifier that `train_generator.py` later loads as `clean_model`
(reference train_clean_classifier.py:75-121 loop, :153-160 checkpoint keys, :191-193 path
<checkpoints>/<saving_prefix>/<dataset>/<dataset>_<saving_prefix>.pth.tar)."""
import os
import time

import torch

import config
from combat_amd import api, dist as cdist
from combat_amd.data import get_dataloader
from combat_amd.log import SummaryWriter, progress_bar
from combat_amd.nets import configure_dataset, default_classifier
from combat_amd.step import ClassifierStep


def get_model(opt):
    netC = default_classifier(opt).to(opt.device)
    optimizerC = torch.optim.SGD(netC.parameters(), opt.lr_clean, momentum=0.9, weight_decay=5e-4, nesterov=True)
    schedulerC = torch.optim.lr_scheduler.MultiStepLR(optimizerC, opt.scheduler_clean_milestones, opt.scheduler_clean_lambda)
    return netC, optimizerC, schedulerC


def train(netC, optimizerC, schedulerC, train_dl, tf_writer, epoch, opt, step=None):
    print(" Train:")
    netC.train()
    step = step or netC.__dict__.get("_clf_step")
    if step is None:
        pg = torch.distributed.group.WORLD if torch.distributed.is_initialized() else None
        step = netC.__dict__["_clf_step"] = ClassifierStep(netC, opt, process_group=pg)
    step.read_metrics(reset=True) if step.N else None
    total = 0
    for batch_idx, (inputs, targets) in enumerate(train_dl):
        step.run(inputs.to(opt.device, non_blocking=True), targets, lr=optimizerC.param_groups[0]["lr"])
        total += inputs.shape[0]
        last = batch_idx == len(train_dl) - 1 or (opt.max_steps and batch_idx + 1 >= opt.max_steps)
        if batch_idx % max(1, opt.log_interval) == 0 or last:
            m = step.read_metrics()
            progress_bar(batch_idx, len(train_dl), "CE Loss: {:.4f} | Clean Acc: {:.4f}".format(
                m["loss_sum"] / total, m["correct"] * 100.0 / total))
        if last:
            break
    tf_writer.add_scalars("Clean Accuracy", {"Clean": m["correct"] * 100.0 / total}, epoch)
    schedulerC.step()


def eval(netC, optimizerC, schedulerC, test_dl, best_clean_acc, tf_writer, epoch, opt):
    print(" Eval:")
    netC.eval()
    cdist.average_bn_buffers(netC)    # data parallel: one model on every rank and in the checkpoint (combat_amd/dist.py)
    n = correct = 0
    for batch_idx, (inputs, targets) in enumerate(test_dl):
        with torch.no_grad():
            preds = netC(inputs.to(opt.device))
        n += len(inputs)
        correct += int((preds.argmax(1).cpu() == targets).sum())
        progress_bar(batch_idx, len(test_dl), "Clean Acc: {:.4f} - Best: {:.4f}".format(correct * 100.0 / n, best_clean_acc))
    if torch.distributed.is_initialized():   # every rank evaluated its shard of the test set
        n, correct = cdist.all_reduce_counters([n, correct], device=opt.device)
    acc = correct * 100.0 / n
    tf_writer.add_scalars("Test Accuracy", {"Clean": acc}, epoch)
    if acc > best_clean_acc:
        print(" Saving...")
        best_clean_acc = acc
        if int(os.environ.get("RANK", 0)) == 0:
            api.sync_momentum_to_optimizer(optimizerC, netC)
            torch.save({"netC": netC.state_dict(), "schedulerC": schedulerC.state_dict(),
                        "optimizerC": optimizerC.state_dict(), "best_clean_acc": acc, "epoch_current": epoch}, opt.ckpt_path)
    return best_clean_acc


def main():
    opt = config.get_arguments().parse_args()
    configure_dataset(opt)
    rank, local_rank, world = cdist.init()
    if opt.device == "cuda":
        opt.device = "cuda:%d" % local_rank
    if opt.seed is not None:
        torch.manual_seed(opt.seed)
    train_dl = get_dataloader(opt, True, rank=rank, world=world)
    test_dl = get_dataloader(opt, False, shuffle=False, rank=rank, world=world)
    netC, optimizerC, schedulerC = get_model(opt)
    mode = opt.saving_prefix
    opt.ckpt_folder = os.path.join(opt.checkpoints, mode, opt.dataset)
    opt.ckpt_path = os.path.join(opt.ckpt_folder, "{}_{}.pth.tar".format(opt.dataset, mode))
    opt.log_dir = os.path.join(opt.ckpt_folder, "log_dir")
    best, epoch_current = 0.0, 0
    if opt.continue_training and os.path.exists(opt.ckpt_path):
        sd = torch.load(opt.ckpt_path, map_location=opt.device, weights_only=True)
        netC.load_state_dict(sd["netC"])
        optimizerC.load_state_dict(sd["optimizerC"])
        schedulerC.load_state_dict(sd["schedulerC"])
        api.load_momentum_from_optimizer(optimizerC, netC)
        best, epoch_current = sd["best_clean_acc"], sd["epoch_current"]
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
        train(netC, optimizerC, schedulerC, train_dl, tf_writer, epoch, opt)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        best = eval(netC, optimizerC, schedulerC, test_dl, best, tf_writer, epoch, opt)
        print(" train {:.2f} s, eval + checkpoint {:.2f} s".format(t1 - t0, time.perf_counter() - t1))


if __name__ == "__main__":
    main()
