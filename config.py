"""Command-line flags of the COMBAT entry scripts -- same names, types and defaults as the
reference's single flat parser (reference config.py:4-86), so existing command lines keep working;
extra flags for this implementation are grouped at the end."""
import argparse

# (flag, kwargs) -- order follows the reference for diff-ability of `--help`
_FLAGS = [
    ("--data_root", dict(type=str, default="./data")),
    ("--checkpoints", dict(type=str, default="./checkpoints")),
    ("--temps", dict(type=str, default="./temps")),
    ("--device", dict(type=str, default="cuda")),
    ("--continue_training", dict(action="store_true")),
    ("--saving_prefix", dict(type=str, help="Folder in /checkpoints for saving ckpt")),
    ("--attack_mode", dict(default="all2one")),
    ("--load_checkpoint", dict(default="")),
    ("--load_checkpoint_clean", dict(type=str)),
    ("--dataset", dict(type=str, default="cifar10")),
    ("--input_height", dict(type=int, default=32)),
    ("--input_width", dict(type=int, default=32)),
    ("--input_channel", dict(type=int, default=3)),
    ("--num_classes", dict(type=int, default=10)),
    ("--bs", dict(type=int, default=128)),
    ("--lr_C", dict(type=float, default=1e-2)),
    ("--lr_G", dict(type=float, default=1e-2)),
    ("--lr_clean", dict(type=float, default=1e-2)),
    # the reference declares these three with type=list (and --sigma with type=tuple), which splits a
    # command-line value into characters: only the defaults are usable there; kept as is
    ("--schedulerC_milestones", dict(type=list, default=[100, 150])),
    ("--schedulerG_milestones", dict(type=list, default=[100, 150])),
    ("--scheduler_clean_milestones", dict(type=list, default=[100, 150])),
    ("--schedulerC_lambda", dict(type=float, default=0.1)),
    ("--schedulerG_lambda", dict(type=float, default=0.1)),
    ("--scheduler_clean_lambda", dict(type=float, default=0.1)),
    ("--n_iters", dict(type=int, default=200)),
    ("--num_workers", dict(type=int, default=6)),
    ("--lambda_cov", dict(type=float, default=1)),
    ("--noise_rate", dict(type=float, default=0.08)),
    ("--target_label", dict(type=int, default=0)),
    ("--pc", dict(type=float, default=0.5)),
    ("--cross_rate", dict(type=float, default=1)),
    ("--s", dict(type=int, default=2)),
    ("--grid_rescale", dict(type=float, default=0.15)),
    ("--ratio", dict(type=float, default=0.65, help="scale ratio for DCT of noise")),
    ("--kernel_size", dict(type=int, default=3, help="kernel size for Gaussian blur")),
    ("--sigma", dict(type=tuple, default=(0.1, 1.0), help="sigma for Gaussian blur")),
    ("--random_rotation", dict(type=int, default=10)),
    ("--random_crop", dict(type=int, default=5)),
    ("--scale", dict(type=float, default=1)),
    ("--S2", dict(type=int, default=8)),
    ("--clamp", dict(action="store_true")),
    ("--nearest", dict(type=float, default=0)),
    ("--lnoise", dict(type=int, default=8)),
    ("--model", dict(type=str, default="default")),
    ("--tv_weight", dict(type=float, default=0.01)),
    ("--L2_weight", dict(type=float, default=0.02)),
    ("--F_checkpoints", dict(type=str, default="./defenses/frequency_based/checkpoints")),
    ("--F_model", dict(type=str, default="original")),
    ("--F_dropout", dict(type=float, default=0.5)),
    ("--F_num_ensemble", dict(type=int, default=3)),
    ("--model_clean", dict(type=str, default="default")),
    ("--clean_model_weight", dict(type=float, default=0.8)),
    ("--noise_only", dict(action="store_true", default=False)),
    ("--post_transform_option", dict(type=str, default="use", choices=["use", "no_use", "use_modified"])),
    ("--scale_noise_rate", dict(type=float, default=1.0)),
    ("--cross_weight", dict(type=float, default=0.2)),
    ("--debug", dict(action="store_true", default=False)),
    ("--r", dict(type=float, default=1 / 4)),
    ("--scale_factor", dict(type=float, default=0.5)),
    ("--scale_mode", dict(type=str, default="bicubic")),
]

_EXTRA = [
    ("--synthetic", dict(action="store_true", help="CIFAR-10-shaped random data instead of --data_root")),
    ("--synthetic_size", dict(type=int, default=0, help="images per synthetic split (0 = dataset size)")),
    ("--synthetic_kind", dict(type=str, default="noise", help="noise: uniform bytes, random labels; structured: a learnable class-prototype set")),
    ("--max_steps", dict(type=int, default=0, help="stop each epoch after this many batches (0 = all)")),
    ("--log_interval", dict(type=int, default=20, help="batches between progress-bar refreshes (each one syncs)")),
    ("--seed", dict(type=int, default=None, help="seed torch / numpy / random (the reference never seeds)")),
    ("--allow_missing_F", dict(action="store_true", help="random-init frequency detector if its checkpoint is absent")),
]


def get_arguments():
    parser = argparse.ArgumentParser()
    for flag, kw in _FLAGS + _EXTRA:
        parser.add_argument(flag, **kw)
    return parser
