"""Final evaluation: clean accuracy, backdoor benign accuracy (Bd BA) and attack success rate (Bd ASR)
of a classifier against a trained generator (reference eval.py:108-152 loop, :155-219 main).
As in the reference, the classifier is loaded from --load_checkpoint_clean (pass the victim's folder
there to evaluate a victim) and the generator from --load_checkpoint; nothing is saved."""
import os

import torch

import config
from combat_amd import api
from combat_amd import dist as cdist
from combat_amd.data import get_dataloader
from combat_amd.log import SummaryWriter, progress_bar
from combat_amd.nets import GridGenerator, UnetGenerator, configure_dataset, default_classifier
from combat_amd.step import create_targets_bd


def get_model(opt):
    return default_classifier(opt).to(opt.device), UnetGenerator(opt).to(opt.device)


def eval(netC, netG, test_dl, tf_writer, opt):
    print(" Eval:")
    n = nb = clean = ba = asr = 0
    for batch_idx, (inputs, targets) in enumerate(test_dl):
        with torch.no_grad():
            inputs, targets = inputs.to(opt.device), targets.to(opt.device)
            clean += int((netC(inputs).argmax(1) == targets).sum())
            n += len(inputs)
            ntrg = (targets != opt.target_label).nonzero()[:, 0]
            if len(ntrg):
                inputs_bd = api.create_backdoor(netG, inputs[ntrg], opt)
                targets_bd = create_targets_bd(targets[ntrg], opt).to(opt.device)
                pred = netC(inputs_bd).argmax(1)
                ba += int((pred == targets[ntrg]).sum())
                asr += int((pred == targets_bd).sum())
                nb += len(ntrg)
        acc_clean, acc_ba, acc_asr = clean * 100.0 / n, ba * 100.0 / max(nb, 1), asr * 100.0 / max(nb, 1)
        progress_bar(batch_idx, len(test_dl), "Clean Acc: {:.4f} | Bd BA: {:.4f} | Bd ASR: {:.4f}".format(
            acc_clean, acc_ba, acc_asr))
    tf_writer.add_scalars("Test Accuracy", {"Clean": acc_clean, "Bd BA": acc_ba, "Bd ASR": acc_asr}, 0)
    return acc_clean, acc_ba, acc_asr


def main():
    opt = config.get_arguments().parse_args()
    configure_dataset(opt)
    cdist.limit_host_threads()
    test_dl = get_dataloader(opt, False, shuffle=False)
    netC, netG = get_model(opt)
    mode = opt.saving_prefix
    opt.ckpt_folder = os.path.join(opt.checkpoints, "{}_clean".format(mode), opt.dataset)
    opt.log_dir = os.path.join(opt.ckpt_folder, "log_dir")
    os.makedirs(opt.log_dir, exist_ok=True)
    for key, ck in (("netC", opt.load_checkpoint_clean), ("netG", opt.load_checkpoint)):
        path = os.path.join(opt.checkpoints, ck or "", opt.dataset, "{}_{}.pth.tar".format(opt.dataset, ck))
        if not os.path.exists(path):
            print("Error: {} not found".format(path))
            exit()
        sd = torch.load(path, map_location=opt.device, weights_only=True)[key]
        if key == "netG" and any(k.startswith("fc1.") for k in sd):
            # a WaNet generator (train_generator_wanet.py / train_victim_wanet.py checkpoints; the reference has no
            # stand-alone evaluation script for them: its eval.py builds a UnetGenerator and fails on the keys)
            netG = GridGenerator(opt).to(opt.device)
        net = netC if key == "netC" else netG
        net.load_state_dict(sd)
        net.eval()
    eval(netC, netG, test_dl, SummaryWriter(log_dir=opt.log_dir), opt)


if __name__ == "__main__":
    main()
