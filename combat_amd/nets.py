"""Host-side mirrors of the reference networks on the hot path.

Each class keeps the reference's constructor signature, ``state_dict()`` keys / shapes / order
(fp32, OIHW) and default initialisation order -- so ``torch.manual_seed(s); Net()`` yields the
same parameters as the reference class under the same seed and checkpoints move both ways --
but holds no compute: ``forward`` hands the tensors to the HIP engine (``combat_amd.engine``),
which fails loudly when the gfx950 library is missing.

Mirrored interfaces:
  UnetGenerator     networks/models.py:268-341
  GridGenerator     networks/models.py:344-385
  PreActResNet18    classifier_models/preact_resnet.py:13-40, 72-110
  ResNet18          classifier_models/resnet.py:15-37, 68-106
  FrequencyModel    defenses/frequency_based/model.py:8-52
"""
from __future__ import annotations

import torch
from torch import nn

# (name, cin multiplier, cout multiplier, stride, followed by InstanceNorm?)
UNET_LAYERS = (
    ("conv0_0", 0, 1, 2, False), ("conv0_1", 1, 1, 1, True),
    ("conv1_0", 1, 2, 2, True), ("conv1_1", 2, 2, 1, True),
    ("conv2_0", 2, 4, 2, True), ("conv2_1", 4, 4, 1, True),
    ("conv3_0", 4, 8, 2, True), ("conv3_1", 8, 8, 1, True),
    ("upconv3_1", 8, 8, 1, True), ("upconv3_0", 8, 4, 1, True),
    ("upconv2_1", 4, 4, 1, True), ("upconv2_0", 4, 2, 1, True),
    ("upconv1_1", 2, 2, 1, True), ("upconv1_0", 2, 1, 1, True),
    ("upconv0_1", 1, 1, 1, True), ("upconv0_0", 1, 0, 1, False),
)
INPUT_SIZE_TO_SCALER = {32: 1, 64: 4, 224: 49}  # 224 added: the reference raises KeyError


def _engine():
    from . import engine  # deferred: importing nets must work without the HIP library
    return engine


class _HipModule(nn.Module):
    """Common plumbing: a lazily built engine keyed on parameter identity."""

    arch = ""

    def _net_engine(self):
        eng = self.__dict__.get("_eng")
        if eng is None:
            eng = _engine().build_engine(self)
            self.__dict__["_eng"] = eng
        return eng

    def _apply(self, fn, *a, **k):  # .to()/.cuda() move parameters: drop packed device state
        self.__dict__.pop("_eng", None)
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        eng = self.__dict__.get("_eng")
        if eng is not None:
            eng.mark_weights_dirty()
        return r

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """float32 [N,C,H,W] -> float32; differentiable w.r.t. x and the parameters."""
        return _engine().module_forward(self, x)


class UnetGenerator(_HipModule):
    arch = "unet"

    def __init__(self, opt=None, in_channels=3, nf=64, use_bias=True, out_channel=None):
        super().__init__()
        if out_channel is None:
            out_channel = in_channels
        self.in_channels, self.nf, self.out_channel = in_channels, nf, out_channel
        for name, ci, co, stride, _ in UNET_LAYERS:
            cin = in_channels if ci == 0 else nf * ci
            cout = out_channel if co == 0 else nf * co
            setattr(self, name, nn.Conv2d(cin, cout, kernel_size=3, stride=stride, padding=1, bias=use_bias))


class GridGenerator(_HipModule):
    arch = "gridgen"

    def __init__(self, opt, in_channels=3, nf=64, use_bias=True):
        super().__init__()
        self.S = opt.s
        self.in_channels, self.nf = in_channels, nf
        for name, ci, co, stride, _ in UNET_LAYERS[:8]:
            cin = in_channels if ci == 0 else nf * ci
            setattr(self, name, nn.Conv2d(cin, nf * co, kernel_size=3, stride=stride, padding=1, bias=use_bias))
        self.fc1 = nn.Linear(nf * 8, nf)
        self.fc2 = nn.Linear(nf, self.S * self.S * 2)


class _PreActBlock(nn.Module):
    def __init__(self, in_planes, planes, stride):
        super().__init__()
        self.stride = stride
        self.bn1 = nn.BatchNorm2d(in_planes)
        self.conv1 = nn.Conv2d(in_planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        if stride != 1 or in_planes != planes:
            self.shortcut = nn.Sequential(nn.Conv2d(in_planes, planes, 1, stride, bias=False))


class _BasicBlock(nn.Module):
    def __init__(self, in_planes, planes, stride):
        super().__init__()
        self.stride = stride
        self.conv1 = nn.Conv2d(in_planes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.shortcut = nn.Sequential()
        if stride != 1 or in_planes != planes:
            self.shortcut = nn.Sequential(nn.Conv2d(in_planes, planes, 1, stride, bias=False),
                                          nn.BatchNorm2d(planes))


class _ResNetBase(_HipModule):
    block = None
    stem_bn = False

    def __init__(self, num_classes=10, n_input=3, scaler=1):
        super().__init__()
        self.num_classes, self.n_input, self.scaler = num_classes, n_input, scaler
        self.conv1 = nn.Conv2d(n_input, 64, 3, 1, 1, bias=False)
        if self.stem_bn:
            self.bn1 = nn.BatchNorm2d(64)
        in_planes = 64
        for i, (planes, stride) in enumerate(((64, 1), (128, 2), (256, 2), (512, 2)), start=1):
            blocks = []
            for s in (stride, 1):
                blocks.append(self.block(in_planes, planes, s))
                in_planes = planes
            setattr(self, "layer%d" % i, nn.Sequential(*blocks))
        self.linear = nn.Linear(512 * scaler, num_classes)


class PreActResNet(_ResNetBase):
    arch = "preact_resnet18"
    block = _PreActBlock
    stem_bn = False


class ResNet(_ResNetBase):
    arch = "resnet18"
    block = _BasicBlock
    stem_bn = True


def PreActResNet18(num_classes=10, n_input=3, input_size=32):
    return PreActResNet(num_classes=num_classes, n_input=n_input, scaler=INPUT_SIZE_TO_SCALER[input_size])


def ResNet18(num_classes=10, n_input=3, input_size=64):
    return ResNet(num_classes=num_classes, n_input=n_input, scaler=INPUT_SIZE_TO_SCALER[input_size])


class FrequencyModel(_HipModule):
    arch = "freq"
    WIDTHS = (32, 32, 64, 64, 128, 128)

    def __init__(self, num_classes=2, n_input=3, input_size=32):
        super().__init__()
        self.num_classes, self.n_input, self.input_size = num_classes, n_input, input_size
        cin = n_input
        for i, w in enumerate(self.WIDTHS, start=1):
            setattr(self, "conv%d" % i, nn.Conv2d(cin, w, (3, 3), padding="same"))
            setattr(self, "bn%d" % i, nn.BatchNorm2d(w))
            cin = w
        self.linear6 = nn.Linear(2048 * INPUT_SIZE_TO_SCALER[input_size], num_classes)


def configure_dataset(opt) -> None:
    """Input dimensions / class count per dataset, as every entry script's main() sets them
    (train_generator.py:470-486; imagenet10: train_generator_wanet.py:472-477 -- 224 x 224, 10 classes, bs 32)."""
    if opt.dataset == "cifar10":
        opt.input_height, opt.input_width, opt.input_channel = 32, 32, 3
    elif opt.dataset == "celeba":
        opt.input_height, opt.input_width, opt.input_channel = 64, 64, 3
        opt.num_classes = 8
    elif opt.dataset == "imagenet10":
        opt.input_height, opt.input_width, opt.input_channel = 224, 224, 3
        opt.num_classes = 10
        opt.bs = 32
    else:
        raise Exception("Invalid Dataset")


def default_classifier(opt):
    """The `--model default` classifier of a dataset (train_generator.py:90-96): PreActResNet18 for CIFAR-10,
    ResNet18(num_classes) at 64 x 64 for CelebA."""
    if getattr(opt, "model", "default") != "default":
        raise Exception("only the default classifier of each dataset runs on the HIP path")
    if opt.dataset == "cifar10":
        return PreActResNet18()
    if opt.dataset == "celeba":
        return ResNet18(num_classes=opt.num_classes)
    if opt.dataset == "imagenet10":   # train_generator_wanet.py:72-74 (the reference's scaler table lacks 224: SURVEY D4)
        return ResNet18(num_classes=opt.num_classes, input_size=opt.input_height)
    raise Exception("Invalid Dataset")
