"""Data parallelism for the alternated step: one process per GPU, ``torch.distributed`` with the
``nccl`` backend (= RCCL over xGMI on ROCm); ``gloo`` on CPU for tests.

The reference has no distributed code.  Added semantics: every rank
holds full replicas and applies the reference's step to its own 128-image shard -- own poison
selection, own augmentation draws, rank-local BatchNorm statistics (DDP's default) -- and the
parameter gradients are averaged: netC's after Phase C, netG's after Phase G.  Each network's
gradients live in ONE flat fp32 buffer (engine.FlatParams), so the exchange is a few large
all-reduces over contiguous slices (bucketed so the first buckets travel while the rest of the
backward still runs); xGMI is point-to-point, so few large messages beat many small ones."""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def limit_host_threads() -> None:
    """The host side of this package is one thread feeding the GPU plus batch-sized CPU tensor ops.  torch sizes its
    intra-op pool to every hardware thread it sees (128 on the MI355X boxes, whatever the process may actually use),
    and a 128-thread OpenMP region around a 1.5-MB operation costs milliseconds: cap it (COMBAT_HOST_THREADS, default
    8 or the current setting if smaller)."""
    want = int(os.environ.get("COMBAT_HOST_THREADS", 0)) or min(torch.get_num_threads(), 8)
    if want != torch.get_num_threads():
        torch.set_num_threads(want)


def init(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(rank, local_rank, world) from the torchrun environment; no-op for a single process."""
    limit_host_threads()
    world = int(os.environ.get("WORLD_SIZE", 1))
    rank, local = int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return rank, local, world


def broadcast_module(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Identical replicas: parameters and buffers of rank `src` everywhere."""
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src, group=group)


def average_bn_buffers(module: torch.nn.Module, group=None) -> int:
    """BatchNorm running statistics are rank-local during training (each rank normalises with the statistics of its
    own 128-image shard: DDP's default), so after an epoch the replicas' `running_mean` / `running_var`
    differ slightly while every parameter is bit-identical.  Before anything reads them -- the per-epoch evaluation,
    whose counters are summed over ranks, and the checkpoint rank 0 writes -- they are AVERAGED over the ranks (the
    exponential average is linear in the batch statistics, so the mean of the ranks' running means is the running mean
    of the mean batch statistic; `num_batches_tracked` is equal everywhere).  Every rank then evaluates, and rank 0
    saves, the same model.  Returns the number of buffers exchanged (0 for a single process)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 0
    bufs = [b for name, b in module.named_buffers() if name.endswith(("running_mean", "running_var"))]
    if not bufs:
        return 0
    flat = torch.cat([b.detach().reshape(-1).float() for b in bufs])
    dist.all_reduce(flat, group=group)
    flat /= dist.get_world_size(group)
    o = 0
    for b in bufs:
        b.data.copy_(flat[o:o + b.numel()].view_as(b))
        o += b.numel()
    eng = module.__dict__.get("_eng")
    if eng is not None:
        eng.mark_weights_dirty()          # the folded eval-mode scale / shift tables are stale
    return len(bufs)


def bucket_ranges(total: int, marks: Sequence[int], min_elems: int = 1 << 20) -> List[Tuple[int, int]]:
    """Split [0, total) at the given offsets (layer boundaries of the flat gradient buffer, ascending)
    into contiguous buckets of at least `min_elems` elements, returned in REVERSE order -- the order in
    which a backward pass finishes them."""
    cuts = [0]
    for m in sorted(set(int(x) for x in marks)):
        if 0 < m < total and m - cuts[-1] >= min_elems and total - m >= min_elems:
            cuts.append(m)
    cuts.append(total)
    return [(cuts[i], cuts[i + 1]) for i in range(len(cuts) - 1)][::-1]


class GradReducer:
    """Sum-all-reduce of a flat gradient buffer in buckets; `average()` scaling is folded into the
    optimiser (grad_scale = 1/world) so no extra pass touches the gradients."""

    def __init__(self, flat_grad: torch.Tensor, ranges: Sequence[Tuple[int, int]], group=None):
        self.flat, self.ranges, self.group = flat_grad, list(ranges), group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.force = os.environ.get("COMBAT_FORCE_ALLREDUCE", "0") == "1"   # (see step.FORCE_ALLREDUCE)
        self._pending = []

    def launch(self, i: int) -> None:
        """Start bucket i (call as soon as the backward has produced it)."""
        if self.world == 1 and not self.force:
            return
        lo, hi = self.ranges[i]
        self._pending.append(dist.all_reduce(self.flat[lo:hi], group=self.group, async_op=True))

    def launch_all(self) -> None:
        for i in range(len(self.ranges)):
            self.launch(i)

    def wait(self) -> None:
        for w in self._pending:
            w.wait()
        self._pending.clear()

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world


def all_reduce_counters(values: Sequence[float], group=None, device=None) -> List[float]:
    """Epoch-level metric counters (train_generator.py:344-351 style integer sums)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return list(values)
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    dist.all_reduce(t, group=group)
    return t.tolist()


def barrier(group=None) -> None:
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.barrier(group=group)


def fresh_start(folder: str, rank: int, group=None) -> None:
    """The reference wipes the checkpoint folder on a fresh start (train_generator.py:562).  With several
    ranks only rank 0 may do that, and nobody may create files under it (log_dir, writers) before the
    wipe has finished: wipe on rank 0, barrier, then continue."""
    import shutil
    if rank == 0:
        shutil.rmtree(folder, ignore_errors=True)
    barrier(group)


class NullWriter:
    """SummaryWriter call shape that records nothing (ranks other than 0)."""

    def add_scalars(self, *a, **k):
        pass

    def add_image(self, *a, **k):
        pass
