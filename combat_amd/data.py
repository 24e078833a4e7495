"""Input side of the step (reference: utils/dataloader.py:24-42, 98-123; utils/dataloader_cleanbd.py
:131-193) without torchvision: CIFAR-10 is read from the standard python or binary batches, images
get the reference's ToTensor + Normalize(0.5, 0.5) ((u8/255 - 0.5)/0.5), batches come out as pinned
float32 [B,3,32,32] + int64 labels on the HOST, as a torch DataLoader would yield them.  The whole
set (150 MB of uint8) stays in RAM, so there are no worker processes; shuffling draws one
``torch.randperm`` per epoch from a private generator whose base seed comes from torch's global generator
(DataLoader's RandomSampler does the same).  Nothing is
downloaded: a missing dataset is an error unless ``--synthetic`` asks for CIFAR-10-shaped noise."""
from __future__ import annotations

import os
import pickle
import random
from typing import Iterator, Optional, Tuple

import numpy as np
import torch

CIFAR_TRAIN, CIFAR_TEST = 50000, 10000


def _load_cifar10(root: str, train: bool) -> Tuple[np.ndarray, np.ndarray]:
    py = os.path.join(root, "cifar-10-batches-py")
    bn = os.path.join(root, "cifar-10-batches-bin")
    if os.path.isdir(py):
        names = ["data_batch_%d" % i for i in range(1, 6)] if train else ["test_batch"]
        xs, ys = [], []
        for n in names:
            with open(os.path.join(py, n), "rb") as f:
                d = pickle.load(f, encoding="latin1")   # the public dataset's own format (user data)
            xs.append(np.asarray(d["data"], np.uint8).reshape(-1, 3, 32, 32))
            ys.append(np.asarray(d.get("labels", d.get("fine_labels")), np.int64))
        return np.concatenate(xs), np.concatenate(ys)
    if os.path.isdir(bn):
        names = ["data_batch_%d.bin" % i for i in range(1, 6)] if train else ["test_batch.bin"]
        raw = np.concatenate([np.fromfile(os.path.join(bn, n), np.uint8).reshape(-1, 3073) for n in names])
        return raw[:, 1:].reshape(-1, 3, 32, 32).copy(), raw[:, 0].astype(np.int64)
    raise FileNotFoundError(
        "CIFAR-10 not found under %r (expected cifar-10-batches-py/ or cifar-10-batches-bin/); nothing is "
        "downloaded here -- pass --synthetic for CIFAR-10-shaped random data" % root)


def _load_celeba(root: str, train: bool, hw: int) -> Tuple[np.ndarray, np.ndarray]:
    """CelebA as the reference uses it (utils/dataloader.py:63-80: images resized to 64 x 64, label =
    (attr[18] << 2) + (attr[31] << 1) + attr[21] over torchvision's 40 attribute columns, 8 classes), from a
    pre-decoded cache `celeba_{train,test}_64.npz` with uint8 `images` [N,64,64,3] (or [N,3,64,64]) and `attr`
    [N,40] (0/1).  JPEG decoding is outside this package (no image library is assumed); the cache is what
    a one-off conversion of torchvision.datasets.CelebA(split=train / test, target_type="attr") writes."""
    path = os.path.join(root, "celeba_%s_%d.npz" % ("train" if train else "test", hw))
    if not os.path.exists(path):
        raise FileNotFoundError("CelebA cache %r not found (uint8 images [N,%d,%d,3] + attr [N,40]); nothing is "
                                "downloaded.  --synthetic runs the same shapes on generated data." % (path, hw, hw))
    z = np.load(path)
    attr = z["attr"].astype(np.int64)
    labels = (attr[:, 18] << 2) + (attr[:, 31] << 1) + attr[:, 21]
    img = z["images"]
    if img.shape[-1] == 3:   # NHWC -> the loader's NCHW
        img = img.transpose(0, 3, 1, 2)
    return np.ascontiguousarray(img), labels.astype(np.int64)


def _load_imagenet10(root: str, train: bool, hw: int) -> Tuple[np.ndarray, np.ndarray]:
    """ImageNet-10 as the reference uses it (utils/dataloader.py:83-95: torchvision.datasets.ImageNet under
    <data_root>/imagenet10, split train / val, resized to 224 x 224 by get_transform :27), from a pre-decoded cache
    `imagenet10_{train,val}_224.npz` with uint8 `images` [N,224,224,3] (or [N,3,224,224]) and int `labels` [N] (0..9).
    JPEG decoding is outside this package, as for CelebA.  (np.load ignores mmap_mode for .npz archives: the array is
    read once, and only copied again if it is stored NHWC.)"""
    path = os.path.join(root, "imagenet10_%s_%d.npz" % ("train" if train else "val", hw))
    if not os.path.exists(path):
        raise FileNotFoundError("ImageNet-10 cache %r not found (uint8 images [N,%d,%d,3] + labels [N]); nothing is "
                                "downloaded.  --synthetic runs the same shapes on generated data." % (path, hw, hw))
    z = np.load(path)
    img = z["images"]
    if img.shape[-1] == 3:
        img = np.ascontiguousarray(img.transpose(0, 3, 1, 2))
    return img, np.asarray(z["labels"]).astype(np.int64)


def synthetic_cifar10(n: int, seed: int, hw: int = 32, classes: int = 10) -> Tuple[np.ndarray, np.ndarray]:
    g = np.random.default_rng(seed)
    return g.integers(0, 256, (n, 3, hw, hw), dtype=np.uint8), g.integers(0, classes, n).astype(np.int64)


def synthetic_structured(n: int, seed: int, hw: int = 32, classes: int = 10, proto_seed: int = 2024,
                         signal: float = 0.07, clutter: float = 0.2, grain: float = 0.08) -> Tuple[np.ndarray, np.ndarray]:
    """A LEARNABLE CIFAR-shaped set for end-metric checks (clean accuracy / attack success rate need something to
    learn; uniform noise with random labels has nothing): class c has a fixed smooth prototype field (a 3 x 8 x 8
    Gaussian draw upsampled bilinearly to hw x hw, the same for every split: `proto_seed`), an image is
    0.5 + signal * prototype[label] + clutter-weighted smooth field of its own + grain-weighted white noise, clipped
    and quantised to uint8 like a stored dataset.  Everything comes from numpy's PCG64 streams (`default_rng`),
    which are specified bit for bit, so the build container (reference modules, tests/golden/make_golden.py) and
    the GPU box generate identical bytes from (n, seed)."""
    def smooth(g, count):
        z = g.standard_normal((count, 3, 8, 8)).astype(np.float32)
        # separable bilinear upsampling 8 -> hw (align_corners=False), written out so that no library kernel decides a bit
        pos = (np.arange(hw, dtype=np.float32) + 0.5) * (8.0 / hw) - 0.5
        i0 = np.clip(np.floor(pos), 0, 7).astype(np.int64)
        i1 = np.clip(i0 + 1, 0, 7)
        w1 = np.clip(pos - i0, 0.0, 1.0).astype(np.float32)
        zr = z[:, :, i0, :] * (1 - w1)[None, None, :, None] + z[:, :, i1, :] * w1[None, None, :, None]
        return zr[:, :, :, i0] * (1 - w1)[None, None, None, :] + zr[:, :, :, i1] * w1[None, None, None, :]

    protos = smooth(np.random.default_rng(proto_seed), classes)
    g = np.random.default_rng(seed)
    labels = g.integers(0, classes, n).astype(np.int64)
    img = np.float32(0.5) + np.float32(signal) * protos[labels] + np.float32(clutter) * smooth(g, n) \
        + np.float32(grain) * g.standard_normal((n, 3, hw, hw)).astype(np.float32)
    return np.clip(np.rint(img * 255.0), 0, 255).astype(np.uint8), labels


class ArrayLoader:
    """Minimal DataLoader stand-in over in-memory uint8 images: ``len()`` = batches per epoch, iteration
    yields (inputs, targets[, poisoned]).

    Data parallel (``rank``/``world``): every epoch's permutation comes from a PRIVATE ``torch.Generator``
    seeded ``base_seed + epoch`` -- the same on every rank whatever the ranks have drawn from their global
    generators in between (augmentation and blur draws are rank-local and would desynchronise a shared
    stream) -- and rank r takes ``order[r::world]``: disjoint shards of one permutation.  ``base_seed``
    defaults to a draw from the global generator at construction (the unseeded behaviour of DataLoader's
    RandomSampler); with a process group it is rank 0's draw.  A shuffling (training) loader gives EVERY
    rank the same number of samples, hence of batches -- a rank that ran one step more would wait forever
    in its gradient all-reduce: the permutation is padded by wrapping around, like
    ``DistributedSampler(drop_last=False)``.  A non-shuffling (evaluation) loader keeps the exact, possibly
    uneven, strided shards: its loop contains no collective and its counters must not count a sample twice."""

    def __init__(self, images: np.ndarray, labels: np.ndarray, bs: int, shuffle: bool, poisoned: Optional[np.ndarray] = None,
                 rank: int = 0, world: int = 1, drop_last: bool = False, base_seed: Optional[int] = None):
        self.x = torch.from_numpy(np.ascontiguousarray(images))
        self.y = torch.from_numpy(np.ascontiguousarray(labels))
        self.poisoned = None if poisoned is None else torch.from_numpy(poisoned.astype(np.bool_))
        self.bs, self.shuffle, self.rank, self.world, self.drop_last = bs, shuffle, rank, world, drop_last
        self.dataset = self  # len(loader.dataset)
        if base_seed is None:
            base_seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
            if world > 1 and torch.distributed.is_available() and torch.distributed.is_initialized():
                box = [base_seed]
                torch.distributed.broadcast_object_list(box, src=0)
                base_seed = int(box[0])
        self.base_seed, self.epoch = int(base_seed), 0

    def _count(self) -> int:
        """Samples this rank walks per epoch."""
        n = self.x.shape[0]
        if self.shuffle:
            return (n + self.world - 1) // self.world          # same on every rank (wrap-around padding)
        return (n - self.rank + self.world - 1) // self.world   # exact strided shard

    def __len__(self) -> int:
        n = self._count()
        return n // self.bs if self.drop_last else (n + self.bs - 1) // self.bs

    def epoch_order(self, epoch: int) -> torch.Tensor:
        """This rank's sample indices of `epoch` (deterministic in base_seed, epoch, rank, world)."""
        n = self.x.shape[0]
        if not self.shuffle:
            return torch.arange(n)[self.rank::self.world]
        g = torch.Generator().manual_seed(self.base_seed + epoch)
        order = torch.randperm(n, generator=g)
        total = self._count() * self.world
        if total > n:
            order = torch.cat([order, order[: total - n]])
        return order[self.rank::self.world]

    RING = 8    # pinned staging buffers (batches the host may be ahead of the device's copies)

    def _staging(self):
        """A ring of pinned float batches, allocated once.  (`tensor.pin_memory()` per batch cost 18 ms each on the
        MI355X box -- a fresh page-locked allocation whenever the device is behind -- and turned a 4.2-ms step into a
        28-ms one.)  A slot is rewritten only after the event recorded behind its consumer's work has completed."""
        if getattr(self, "_ring", None) is None:
            shape = (self.RING, self.bs) + tuple(self.x.shape[1:])
            self._ring = torch.empty(shape, dtype=torch.float32).pin_memory()
            self._ring_ev = [None] * self.RING
            self._ring_i = 0
        return self._ring

    def __iter__(self) -> Iterator:
        order = self.epoch_order(self.epoch)
        self.epoch += 1
        cuda = torch.cuda.is_available()
        x_np, half, c255 = self.x.numpy(), np.float32(0.5), np.float32(255.0)
        for i in range(len(self)):
            idx = order[i * self.bs:(i + 1) * self.bs]
            if cuda:
                ring = self._staging()
                k = self._ring_i
                self._ring_i = (k + 1) % self.RING
                if self._ring_ev[k] is not None:
                    self._ring_ev[k].synchronize()
                xb = ring[k, : idx.numel()]
            else:
                xb = torch.empty((idx.numel(),) + tuple(self.x.shape[1:]), dtype=torch.float32)
            # ToTensor + Normalize(0.5, 0.5) (utils/dataloader.py:35-39) in numpy: ONE thread.  torch's CPU kernels
            # open an OpenMP region over every hardware thread the box shows (128 on the MI355X boxes, 16 of them
            # usable): 1.8 ms for this 1.5-MB batch idle, ~18 ms beside the thread that drives the GPU.
            out = xb.numpy()
            np.divide(x_np[idx.numpy()], c255, out=out, dtype=np.float32)
            np.subtract(out, half, out=out)
            np.divide(out, half, out=out)
            if self.poisoned is None:
                yield xb, self.y[idx]
            else:
                yield xb, self.y[idx], self.poisoned[idx]
            if cuda:    # the consumer has enqueued its copy of this batch by the time it asks for the next one
                ev = torch.cuda.Event()
                ev.record()
                self._ring_ev[k] = ev


def poison_flags(labels: np.ndarray, opt, n_classes: int) -> np.ndarray:
    """utils/dataloader_cleanbd.py:142-150: a fixed random subset (``random.sample``, Python's global
    RNG) of int(pc * #target-class images) images is marked poisoned, once per dataset."""
    target = {opt.target_label} if opt.attack_mode == "all2one" else set(range(n_classes))
    ids = [i for i, l in enumerate(labels.tolist()) if int(l) in target]
    num = max(0, int(opt.pc * len(ids)))
    print(f"Poison {num} images ({opt.pc * len(ids)})")
    flags = np.zeros(len(labels), np.bool_)
    flags[random.sample(ids, num)] = True
    return flags


def get_dataloader(opt, train: bool = True, pretensor_transform: bool = False, bs: Optional[int] = None,
                   shuffle: bool = True, poisoned: bool = False, rank: int = 0, world: int = 1) -> ArrayLoader:
    """Same call shape as the reference's ``get_dataloader`` (utils/dataloader.py:98,
    utils/dataloader_cleanbd.py:161 with ``poisoned=True``)."""
    bs = opt.bs if bs is None else bs
    if opt.dataset not in ("cifar10", "celeba", "imagenet10"):
        raise Exception("Invalid Dataset")
    if getattr(opt, "synthetic", False):
        n = getattr(opt, "synthetic_size", 0) or (CIFAR_TRAIN if train else CIFAR_TEST)
        if not getattr(opt, "synthetic_size", 0):      # keep the default split under ~2 GB of uint8 whatever the image size
            n = max(bs, min(n, (2 << 30) // (3 * opt.input_height * opt.input_height)))
        gen = synthetic_structured if getattr(opt, "synthetic_kind", "noise") == "structured" else synthetic_cifar10
        x, y = gen(n, 1234 if train else 4321, opt.input_height, opt.num_classes)
    elif opt.dataset == "celeba":
        x, y = _load_celeba(opt.data_root, train, opt.input_height)
    elif opt.dataset == "imagenet10":
        x, y = _load_imagenet10(opt.data_root, train, opt.input_height)
    else:
        x, y = _load_cifar10(opt.data_root, train)
    if getattr(opt, "debug", False):                         # utils/dataloader.py:118-119
        x, y = x[:1000], y[:1000]
    flags = poison_flags(y, opt, opt.num_classes) if poisoned else None
    if flags is not None and world > 1 and torch.distributed.is_initialized():
        box = [flags]     # ONE poisoned index set (utils/dataloader_cleanbd.py:142-150 draws it once per dataset): rank 0's
        torch.distributed.broadcast_object_list(box, src=0)
        flags = box[0]
    seed = getattr(opt, "seed", None)
    return ArrayLoader(x, y, bs, shuffle, flags, rank, world, base_seed=None if seed is None else int(seed) + 7919)
