"""Alternated training with the WaNet-style warping trigger on MI355X.

Drop-in for the reference script of the same name (reference train_generator_wanet.py:53-93
get_model, :96-318 train, :321-452 eval, :455-603 main): the generator is the GridGenerator whose
s x s field is upsampled (bicubic, align_corners) to the image size, mixed with the identity grid by
--grid_rescale and used to resample the batch bilinearly; no low-pass, no blur.  The per-batch body
(:140-268) runs as ``combat_amd.step.WanetStep`` on the HIP kernels (combat_amd/csrc/warp.hip), the
evaluation loop and checkpoint logic are train_generator.py's (the reference's two scripts differ
only in how the backdoor batch is made, which ``combat_amd.api.create_backdoor`` dispatches on the
generator's type).  cifar10 / celeba / imagenet10 with the dataset's default classifier.
"""
import torch

import train_generator as base
from combat_amd.nets import FrequencyModel, GridGenerator, default_classifier
from combat_amd.step import WanetStep, create_targets_bd  # noqa: F401  (re-exported like the reference)

create_dir = base.create_dir


def get_model(opt):
    if opt.F_model not in ("original", "original_holdout"):
        raise Exception("only the 'original' detector runs on the HIP path")
    netC = default_classifier(opt).to(opt.device)
    clean_model = default_classifier(opt).to(opt.device)
    netG = GridGenerator(opt).to(opt.device)
    netF = FrequencyModel(num_classes=2, n_input=opt.input_channel, input_size=opt.input_height).to(opt.device)
    optimizerC = torch.optim.SGD(netC.parameters(), opt.lr_C, momentum=0.9, weight_decay=5e-4, nesterov=True)
    schedulerC = torch.optim.lr_scheduler.MultiStepLR(optimizerC, opt.schedulerC_milestones, opt.schedulerC_lambda)
    optimizerG = torch.optim.SGD(netG.parameters(), opt.lr_G, momentum=0.9, weight_decay=5e-4, nesterov=True)
    schedulerG = torch.optim.lr_scheduler.MultiStepLR(optimizerG, opt.schedulerG_milestones, opt.schedulerG_lambda)
    return netC, optimizerC, schedulerC, netG, optimizerG, schedulerG, netF, clean_model


def identity_grid_of(opt):
    """The reference builds this in main() (:579-581) and threads it through train/eval; the HIP path
    generates the same grid on the device (combat_wanet_grid), so the argument is accepted and unused."""
    a = torch.linspace(-1, 1, steps=opt.input_height)
    x, y = torch.meshgrid(a, a, indexing="ij")
    return torch.stack((y, x), 2)[None, ...].to(opt.device)


def train(netC, optimizerC, schedulerC, netG, optimizerG, schedulerG, netF, clean_model, train_dl, identity_grid,
          tf_writer, epoch, opt):
    return base.train(netC, optimizerC, schedulerC, netG, optimizerG, schedulerG, netF, clean_model, train_dl,
                      tf_writer, epoch, opt)


def eval(netC, optimizerC, schedulerC, netG, optimizerG, schedulerG, netF, clean_model, test_dl, identity_grid,
         best_clean_acc, best_bd_acc, best_F_acc, best_clean_model_acc, best_clean_model_bd_ba,
         best_clean_model_bd_asr, tf_writer, epoch, opt):
    return base.eval(netC, optimizerC, schedulerC, netG, optimizerG, schedulerG, netF, clean_model, test_dl,
                     best_clean_acc, best_bd_acc, best_F_acc, best_clean_model_acc, best_clean_model_bd_ba,
                     best_clean_model_bd_asr, tf_writer, epoch, opt)


def main():
    base.main(get_model=get_model)


if __name__ == "__main__":
    main()
