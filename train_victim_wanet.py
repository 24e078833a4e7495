"""Victim training on data poisoned by the frozen WaNet-style warping generator on MI355X.

Drop-in for the reference script of the same name (reference train_victim_wanet.py:36-55 get_model, :58-135
train, :138-207 eval, :210-285 main): the frozen GridGenerator's field, upsampled (bicubic, align_corners) and
blended with the identity grid by --grid_rescale, resamples the images the dataset flags as poisoned; the
classifier is trained on [warped poisoned images, all others] with cross-entropy; evaluation warps the
non-target-class test images and counts how many land on the attack target.  The per-batch body (:72-112) runs
as ``combat_amd.step.ClassifierStep`` (which dispatches on the generator's type: ``combat_warp_fwd`` here, the
UNet trigger in train_victim.py), eval through ``combat_amd.api.create_backdoor``.  Differences of the two
reference scripts that are kept: the checkpoint lives under ``<saving_prefix>_clean/`` (:241-243), carries a
``grid_rescale`` key (:199) and the image grid is logged every epoch (:136).
"""
import torch

import train_victim as base
from combat_amd.nets import GridGenerator, default_classifier
from combat_amd.step import create_targets_bd  # noqa: F401  (re-exported like the reference)

create_dir = base.os.makedirs


def get_model(opt):
    netC = default_classifier(opt).to(opt.device)
    netG = GridGenerator(opt).to(opt.device)
    optimizerC = torch.optim.SGD(netC.parameters(), opt.lr_C, momentum=0.9, weight_decay=5e-4, nesterov=True)
    schedulerC = torch.optim.lr_scheduler.MultiStepLR(optimizerC, opt.schedulerC_milestones, opt.schedulerC_lambda)
    return netC, optimizerC, schedulerC, netG


def identity_grid_of(opt):
    """The reference builds this in main() (:263-265) and threads it through train/eval; the HIP path generates
    the same grid on the device (combat_wanet_grid), so the argument is accepted and unused."""
    a = torch.linspace(-1, 1, steps=opt.input_height)
    x, y = torch.meshgrid(a, a, indexing="ij")
    return torch.stack((y, x), 2)[None, ...].to(opt.device)


def train(netC, optimizerC, schedulerC, netG, train_dl, identity_grid, tf_writer, epoch, opt):
    return base.train(netC, optimizerC, schedulerC, netG, train_dl, tf_writer, epoch, opt)


def eval(netC, optimizerC, schedulerC, netG, test_dl, identity_grid, best_clean_acc, best_bd_acc, tf_writer, epoch, opt):
    return base.eval(netC, optimizerC, schedulerC, netG, test_dl, best_clean_acc, best_bd_acc, tf_writer, epoch, opt)


def main():
    base.main(get_model=get_model, wanet=True)


if __name__ == "__main__":
    main()
