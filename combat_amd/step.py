"""The alternated generator/surrogate training step on the HIP engines
(reference: the loop body train_generator.py:170-290).

Phase C updates the surrogate classifier ``netC`` on a batch whose first ``num_bd`` target-class
images carry the generator's trigger; Phase G updates the generator ``netG`` against the just
updated (eval-mode) ``netC`` and the frozen ``clean_model``.  Work whose results the reference
discards is not launched: the generator backward of Phase C (:220 zeroes it),
the classifier / clean-model weight gradients of Phase G (:179 zeroes netC's; clean_model has no
optimiser), and autograd graphs for the three metric-only forwards (:214, :227, :245-247).

Every random draw the reference makes is taken on the host *before* anything is launched
(``StepRandomness``) and shipped to the device as small tables, so a step is a fixed sequence of
kernel launches with no host synchronisation; metric counters stay on the device until
``read_metrics``.
"""
from __future__ import annotations

import contextlib
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import os

import numpy as np
import torch

from . import ops, trigger
from ._lib import lib
from .augment import PostTensorTransform
from .dist import GradReducer, bucket_ranges
from .engine import FreqEngine, PreActEngine, UnetEngine, f32, short_workgroups

BUCKETS = (8, 16, 32, 64, 128, 256, 512, 1024)


def bucket(n: int) -> int:
    for b in BUCKETS:
        if n <= b:
            return b
    raise ValueError("poisoned sub-batch of %d images exceeds the largest bucket" % n)


@dataclass
class StepRandomness:
    """The draws of one step, in the order the reference makes them (train_generator.py:183,
    194, 196, 214, 226-228, 250): num_bd (numpy global RNG), the two blur sigmas (torch global
    RNG) and five augmentation tables (Python ``random`` + torch RNG)."""

    num_bd: int
    sigma_c: float
    sigma_g: float
    aug: List[Optional[np.ndarray]] = field(default_factory=lambda: [None] * 5)


def create_targets_bd(targets: torch.Tensor, opt) -> torch.Tensor:
    """train_generator.py:70-77."""
    if opt.attack_mode == "all2one":
        return torch.ones_like(targets) * opt.target_label
    if opt.attack_mode == "all2all":
        return (targets + 1) % opt.num_classes
    raise Exception("{} attack mode is not implemented".format(opt.attack_mode))


def draw_randomness(targets_cpu: torch.Tensor, bd_targets_cpu: torch.Tensor, opt, transforms: PostTensorTransform,
                    sigma_range=(0.1, 1.0)) -> StepRandomness:
    """Consume the three RNG streams exactly where the reference does."""
    n = targets_cpu.shape[0]
    n_trg = int((targets_cpu == bd_targets_cpu).sum())
    num_bd = int(np.sum(np.random.rand(n_trg) < opt.pc))                 # :183
    sigma_c = trigger.sample_sigma(sigma_range) if num_bd else 0.5      # :194 (skipped when empty)
    aug0 = transforms.sample(n)                                         # :196
    aug1 = transforms.sample(n)                                         # :214
    sigma_g = trigger.sample_sigma(sigma_range)                         # :226
    aug2, aug3 = transforms.sample(n), transforms.sample(n)             # :227, :228
    aug4 = transforms.sample(n)                                         # :250
    return StepRandomness(num_bd, sigma_c, sigma_g, [aug0, aug1, aug2, aug3, aug4])


def poison_tables(targets: torch.Tensor, bd_targets: torch.Tensor, num_bd: int):
    """Host-side index tables of Phase C (train_generator.py:181-204).  The batch is re-ordered
    [first num_bd target-class images (poisoned), remaining target-class images, all others]:
      perm         int32 [n]  source image of every row of the re-ordered batch
      total_targets int64 [n] labels of the re-ordered batch (poisoned rows carry bd_targets)
      idx_small    int32 [n]  first num_bd entries = images handed to the generator
      idx_total    int32 [n]  gather indices into the [inputs ; triggered images] staging buffer
                               (rows < num_bd read the triggered copies at n + i)"""
    n = targets.shape[0]
    trg = (targets == bd_targets).nonzero()[:, 0]
    ntrg = (targets != bd_targets).nonzero()[:, 0]
    perm = torch.cat([trg, ntrg]).to(torch.int32)
    total_targets = targets[perm.long()].clone()
    total_targets[:num_bd] = bd_targets[perm[:num_bd].long()]
    idx_small = torch.zeros(n, dtype=torch.int32)
    idx_small[:num_bd] = perm[:num_bd]
    idx_total = perm.clone()
    if num_bd:
        idx_total[:num_bd] = torch.arange(n, n + num_bd, dtype=torch.int32)
    return perm, total_targets, idx_small, idx_total


_STREAMS: Dict = {}


def shared_stream(device, kind: str, priority: int = 0) -> torch.cuda.Stream:
    """The process-wide stream of a kind ("main", "side") on a device.  Streams are NOT per step object: the HIP
    runtime multiplexes a process's streams onto a handful of hardware queues, and with the nine streams that two
    step objects with private streams add up to (bench.py's golden gate + the timed step) launches started to block on
    the host -- 10.6 instead of 4.3 ms/step, 9 ms of it inside the enqueue calls."""
    dev = torch.device(device)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), kind)
    s = _STREAMS.get(key)
    if s is None:
        s = low_priority_stream(dev) if (kind != "main" and LOW_PRIO_SIDE) else torch.cuda.Stream(device=dev, priority=priority)
        _STREAMS[key] = s
    return s


LOW_PRIO_SIDE = os.environ.get("COMBAT_LOW_PRIO_SIDE", "0") == "1"   # experiment: second / auxiliary queues below the critical one
_HIP = None


def low_priority_stream(dev) -> torch.cuda.Stream:
    """A stream of the LOWEST priority the device offers (torch only exposes normal / high)."""
    import ctypes
    global _HIP
    if _HIP is None:
        _HIP = ctypes.CDLL("libamdhip64.so")
    least, greatest = ctypes.c_int(), ctypes.c_int()
    assert _HIP.hipDeviceGetStreamPriorityRange(ctypes.byref(least), ctypes.byref(greatest)) == 0
    h = ctypes.c_void_p()
    with torch.cuda.device(dev):
        assert _HIP.hipStreamCreateWithPriority(ctypes.byref(h), 1, least.value) == 0      # 1 = hipStreamNonBlocking
    print("low-priority stream: range least %d greatest %d" % (least.value, greatest.value), flush=True)
    return torch.cuda.ExternalStream(h.value, device=dev)


SIDE_SHORT_WORKGROUPS = os.environ.get("COMBAT_SIDE_WS", "0") != "1"   # (COMBAT_SIDE_WS=1: A/B, the persistent kernel on the second stream too)
# COMBAT_MERGE_C_EVAL=1 (A/B): netC's two eval-mode forwards of Phase G (train_generator.py:227-228) as ONE 2n-image
# pass on the critical queue -- [aug(inputs) ; aug(inputs_bd)], backward on the second half -- as the clean model's
# already are, instead of the metric-only half on the second stream.  Fewer, fatter launches against a longer critical chain.
MERGE_C_EVAL = os.environ.get("COMBAT_MERGE_C_EVAL", "0") == "1"
# COMBAT_FUSED_HEAD=1: one head launch per differentiated pass (combat_head_fwd_bwd) and the linear layer's weight gradient
# beside the input-gradient chain.  OFF by default: two dependent launches and a 6.5-us bubble fewer on the critical queue
# per pass, and the step did not move (3.765 / 3.762 against 3.735 / 3.742 ms on one box, 3.868 against 3.848 on another):
# the step is bound by the CU-time of its heavy launches, not by the length of the critical queue's chain.
FUSED_HEAD = os.environ.get("COMBAT_FUSED_HEAD", "0") == "1"
FORCE_ALLREDUCE = os.environ.get("COMBAT_FORCE_ALLREDUCE", "0") == "1"   # issue the bucketed all-reduces even at world size 1
#                                                                          (tests: RCCL's streams beside the step's on ONE GPU)


def backward_allreduce(plan, eng, pg, world: int, reducers: dict, prof=None) -> None:
    """Run a backward plan; with several ranks, all-reduce the flat gradient buffer in buckets, each launched as
    soon as the plan has enqueued the kernels that complete it (`Plan.mark`), so that the first buckets travel
    while the rest of the backward still runs.  `reducers` caches the GradReducer per plan."""
    if world == 1 and not (FORCE_ALLREDUCE and pg is not None):
        plan.run(prof)
        return
    red = reducers.get(id(plan))
    if red is None:
        ranges = bucket_ranges(eng.fp.total, plan.marks.values())
        red = (GradReducer(eng.fp.grad, ranges, pg), ranges)
        reducers[id(plan)] = red
    reducer, ranges = red
    launched = set()

    def on_mark(offset):
        for i, (lo, hi) in enumerate(ranges):
            if lo >= offset and i not in launched:
                launched.add(i)
                reducer.launch(i)

    plan.run(prof, on_mark=on_mark)
    on_mark(0)
    reducer.wait()


class AlternatedStep:
    """Owns the engines, slots and small device tables of one rank's step."""

    serial = False   # True: no second stream (bench.py's instrumented replay times one kernel at a time)
    keep_grads = False   # True: leave the flat gradient buffers as the step computed them (tests inspect them);
    #                      default: they are zeroed behind each optimiser step, off the critical queue
    kStage = 4   # pinned staging sets (host steps in flight before it has to wait for the device)

    def __init__(self, netC, netG, clean_model, netF, opt, process_group=None):
        self.opt = opt
        self.dev = next(netC.parameters()).device
        self.eC: PreActEngine = netC._net_engine()
        self.eG: UnetEngine = netG._net_engine()
        self.eK: PreActEngine = clean_model._net_engine()
        self.eF: Optional[FreqEngine] = netF._net_engine() if netF is not None else None
        self.netC, self.netG, self.clean_model, self.netF = netC, netG, clean_model, netF
        self.pg = process_group
        self.world = torch.distributed.get_world_size(process_group) if process_group is not None else 1
        if self.world > 1 or (FORCE_ALLREDUCE and process_group is not None):
            self.eC.flush_at_marks = self.eG.flush_at_marks = True    # gradients above an all-reduce mark must be final there
        self.hw = opt.input_height
        self.N = 0
        dev = self.dev
        self.P = trigger.lowpass_matrix(self.hw, opt.ratio).to(dev)
        self.D = trigger.dct_matrix(self.hw).float().to(dev)
        self.transforms = PostTensorTransform(opt)
        self.acc = torch.zeros(8, dtype=torch.float64, device=dev)  # running sums for logging
        self.acc_side = torch.zeros((), dtype=torch.float64, device=dev)   # detector hits (counted on the second stream)
        self._side = None
        self._main = None
        self._host = None
        self.steps_done = 0
        self.samples = 0                 # images since the last metric reset
        self._reducers = {}
        self._sets: Dict[int, dict] = {}

    # ------------------------------------------------------------------ buffers per batch size
    _PER_N = ("inputs", "cat_src", "bd", "d_bd", "d_bd2", "mse", "tab", "tab_f", "tab_i", "k1", "_stage", "_stage_i",
              "d_targets", "sC_train", "sC_eval", "sC_met", "sC_bd", "sK_eval", "sK_bd", "sG", "sF", "pl",
              )

    def _setup(self, n: int):
        """Buffers, slots and plans are per batch size and cached: the ragged last batch of an epoch
        (80 images for CIFAR-10) gets its own set once and the metric counters of every set survive."""
        if n == self.N:
            return
        if self.N:
            self._sets[self.N] = {k: getattr(self, k) for k in self._PER_N}
        if n in self._sets:
            for k, v in self._sets[n].items():
                setattr(self, k, v)
            self.N = n
            return
        self.N, dev, hw = n, self.dev, self.hw
        self.cat_src = torch.zeros(2 * n, 3, hw, hw, dtype=f32, device=dev)   # [inputs ; poisoned images]
        self.inputs = self.cat_src[:n]
        self.bd = torch.empty(n, 3, hw, hw, dtype=f32, device=dev)
        self.d_bd = torch.empty(n, 3, hw, hw, dtype=f32, device=dev)
        self.d_bd2 = torch.empty(n, 3, hw, hw, dtype=f32, device=dev)   # clean-model share (second stream)
        self.mse = torch.empty(3 * n, dtype=f32, device=dev)   # per (image, channel)
        # ONE host->device table per step (one copy instead of a dozen 10-us ones in front of the generator
        # forward): [5 aug tables | index_small, index_total | blur kernels of sigma_c, sigma_g | label rows].
        # Label rows: 0 targets, 1 bd_targets, 2 total_targets (re-ordered batch), 3-4 targets twice (the clean
        # model's 2n batch), 5-6 its second label set (row 6 = bd_targets); the heads' label buffers are
        # views of these rows.
        self.tab = torch.zeros(self._table_bytes(n), dtype=torch.uint8, device=dev)
        self.tab_f, self.tab_i, self.k1, self.d_targets = self._table_views(self.tab, n)
        # pinned staging, a ring of kStage sets: the host runs several steps ahead of the device, and an
        # asynchronous copy reads its pinned source when the DEVICE gets to it -- a set is rewritten only
        # after the event behind its last copy has completed
        self._stage = []
        for _ in range(self.kStage):
            raw = torch.zeros(self._table_bytes(n), dtype=torch.uint8).pin_memory()
            tf, ti, k1, tg = self._table_views(raw, n)
            self._stage.append(dict(raw=raw, tab_f=tf, tab_i=ti, k1=k1, targets=tg, done=None))
        self._stage_i = 0
        eC, eK, eG = self.eC, self.eK, self.eG
        # The eval-mode forwards of one network are independent per sample.  clean_model (off the critical
        # path, second stream): the metric-only forward and the differentiated one run as ONE 2n-image batch
        # [aug(inputs) ; aug(inputs_bd)] (train_generator.py:214+250 -- clean_model is frozen and :214
        # reads only `inputs`, so running it in Phase G changes nothing), backward on the triggered half
        # through a view slot.  netC (:227+228): the differentiated forward is on the critical path and runs
        # alone; the accuracy-only forward on the clean images goes to the second stream.
        self.sC_train = eC.slot("C.train", n, hw)
        self.merge_c = MERGE_C_EVAL and type(self) is AlternatedStep
        if self.merge_c:
            self.sC_eval = eC.slot("C.eval2", 2 * n, hw)   # [metric half ; loss half]
            self.sC_met = self.sC_eval
        else:
            self.sC_eval = eC.slot("C.evalbd", n, hw)      # netC on the triggered images: on the critical path, so on its own
            self.sC_met = eC.slot("C.metric", n, hw)       # netC on the clean images (accuracy only): second stream
        self.sK_eval = eK.slot("K.eval2", 2 * n, hw)
        self.sG = self._gen_slot(n)
        self.sF = self.eF.slot("F", n, hw) if self.eF is not None else None
        # the heads read their labels straight out of the step table (bound before the plans marshal pointers)
        lab = self.d_targets
        if self.merge_c:
            self.sC_train.bufs["targets"], self.sC_eval.bufs["targets"] = lab[2], lab[0:2].view(-1)
        else:
            self.sC_train.bufs["targets"], self.sC_eval.bufs["targets"], self.sC_met.bufs["targets"] = lab[2], lab[1], lab[0]
        self.sK_eval.bufs["targets"], self.sK_eval.bufs["targets2"] = lab[3:5].view(-1), lab[5:7].view(-1)
        w_cm = float(self.opt.clean_model_weight)
        # one head launch per differentiated pass (forward + feature gradient: combat_head_fwd_bwd), where the engine has it
        hb = FUSED_HEAD and type(eC) is PreActEngine
        self.pl = dict(
            C_train_f=eC.forward_plan(self.sC_train, True, **(dict(head_bwd=True) if hb else {})),
            C_train_b=eC.backward_train_plan(self.sC_train, **(dict(head_done=True) if hb else {})),
            C_eval_f=eC.forward_plan(self.sC_eval, False, 1.0, False, **(dict(split_head=True) if self.merge_c else dict(head_bwd=True) if hb else {})),
        )
        self.pl.update(self._gen_plans())
        self.sC_bd = self.sC_eval.half_view(n, n, eC.FWD_SHARED) if self.merge_c else self.sC_eval
        self.pl["C_bd_b"] = eC.backward_eval_plan(self.sC_bd, 1.0, **(dict(head_done=True) if hb and not self.merge_c else {}))
        # the second stream's passes run beside the critical queue: one tile per workgroup (engine.short_workgroups)
        with (short_workgroups() if SIDE_SHORT_WORKGROUPS else contextlib.nullcontext()):
            if not self.merge_c:
                self.pl["C_met_f"] = eC.forward_plan(self.sC_met, False, 1.0, False)
            self.pl["K_eval_f"] = eK.forward_plan(self.sK_eval, False, w_cm, True, split_head=True)
            self.sK_bd = self.sK_eval.half_view(n, n, eK.FWD_SHARED)
            self.pl["K_bd_b"] = eK.backward_eval_plan(self.sK_bd, w_cm)
            if self.sF is not None:
                self.pl["F_f"] = self.eF.forward_plan(self.sF)

    def _gen_slot(self, n):
        return self.eG.slot("G", n, self.hw)

    def _gen_plans(self) -> dict:
        return dict(G_f=self.eG.forward_plan(self.sG), G_b=self.eG.backward_plan(self.sG))

    @staticmethod
    def _table_bytes(n: int) -> int:
        return 5 * n * 16 + 2 * n * 4 + 24 + 7 * n * 8

    @staticmethod
    def _table_views(raw: torch.Tensor, n: int):
        o1 = 5 * n * 16
        o2 = o1 + 2 * n * 4
        o3 = o2 + 24
        return (raw[:o1].view(f32).view(5, n, 4), raw[o1:o2].view(torch.int32).view(2, n),
                raw[o2:o3].view(f32).view(2, 3), raw[o3:].view(torch.int64).view(7, n))

    def _side_stream(self) -> torch.cuda.Stream:
        if self.serial:            # kernel-level measurements: everything in line on the caller's stream
            return torch.cuda.current_stream()
        if self._side is None:
            self._side = shared_stream(self.dev, "side")
        return self._side

    # ------------------------------------------------------------------ one step
    def run(self, inputs: torch.Tensor, targets_cpu: torch.Tensor, rnd: Optional[StepRandomness] = None,
            lr_c: Optional[float] = None, lr_g: Optional[float] = None, prof: Optional[list] = None) -> None:
        """inputs: float32 [B,3,H,W] (device or pinned host); targets_cpu: int64 [B] on the host,
        as the DataLoader yields them (train_generator.py:170-171).

        The step runs on the caller's stream plus two process-wide ones (second stream, auxiliary weight-gradient
        queue).  COMBAT_OWN_STREAM=1 moves the critical chain to a high-priority stream of its own (joined with the
        caller's on both sides): 4.18 -> 4.15 ms/step on a quiet process, but the HIP runtime multiplexes a process's
        streams onto few hardware queues and a high-priority stream makes that mapping fragile -- ONE more stream in
        use anywhere in the process (a collective library's, a second step object's) and launches block on the host:
        7-11 ms/step measured.  Without it the step tolerates four foreign streams (4.28-4.33 ms/step; 5.4 at six).
        Off by default for that reason."""
        if self.serial or os.environ.get("COMBAT_OWN_STREAM", "0") != "1":
            return self._run(inputs, targets_cpu, rnd, lr_c, lr_g, prof)
        caller = torch.cuda.current_stream()
        if self._main is None:
            self._main = shared_stream(self.dev, "main", priority=-1)
        self._main.wait_stream(caller)
        with torch.cuda.stream(self._main):
            self._run(inputs, targets_cpu, rnd, lr_c, lr_g, prof)
        caller.wait_stream(self._main)

    def _run(self, inputs, targets_cpu, rnd, lr_c, lr_g, prof) -> None:
        opt = self.opt
        n = inputs.shape[0]
        self._setup(n)
        hw, st = self.hw, torch.cuda.current_stream().cuda_stream
        targets_cpu = targets_cpu.cpu()
        bd_targets_cpu = create_targets_bd(targets_cpu, opt).cpu()
        if rnd is None:
            rnd = self._draw(targets_cpu, bd_targets_cpu)
        nb = rnd.num_bd
        # ---- host tables (train_generator.py:181-204: batch order [poisoned, rest of target class, others])
        perm, tot, idx_small, idx_total = poison_tables(targets_cpu, bd_targets_cpu, nb)
        hs = self._stage[self._stage_i]
        self._stage_i = (self._stage_i + 1) % self.kStage
        if hs["done"] is not None:
            hs["done"].synchronize()       # (only ever waits if the device is kStage steps behind)
        h_targets, h_tab_i, h_tab_f, h_k1 = hs["targets"], hs["tab_i"], hs["tab_f"], hs["k1"]
        h_targets[0].copy_(targets_cpu)
        h_targets[1].copy_(bd_targets_cpu)
        h_targets[2].copy_(tot)
        h_targets[3].copy_(targets_cpu)
        h_targets[4].copy_(targets_cpu)
        h_targets[6].copy_(bd_targets_cpu)
        h_tab_i[0].copy_(idx_small)
        h_tab_i[1].copy_(idx_total)
        aug_ptr = []
        for i, a in enumerate(rnd.aug):
            if a is not None:
                h_tab_f[i].copy_(torch.from_numpy(a))
            aug_ptr.append(self.tab_f[i].data_ptr() if a is not None else None)
        h_k1[0].copy_(torch.from_numpy(trigger.gaussian_kernel1d(rnd.sigma_c, opt.kernel_size)))
        h_k1[1].copy_(torch.from_numpy(trigger.gaussian_kernel1d(rnd.sigma_g, opt.kernel_size)))
        # ---- the step's small copies as ONE launch (combat_copy3): table (pinned host memory, read through its device
        # mapping), batch, and the generator's re-packed output bias
        extra = self.eG.small_refresh_copy() if hasattr(self.eG, "small_refresh_copy") else None
        direct = inputs.dtype == f32 and inputs.is_contiguous() and (inputs.is_cuda or inputs.is_pinned()) and \
            tuple(inputs.shape) == tuple(self.inputs.shape)
        if os.environ.get("COMBAT_NO_COPY3") == "1":          # A/B: three asynchronous copies
            self.tab.copy_(hs["raw"], non_blocking=True)
            direct = False
            if extra:
                self.eG.bias8_taken = False
        else:
            ops.check(lib.combat_copy3(self.tab.data_ptr(), hs["raw"].data_ptr(), hs["raw"].numel(),
                                       self.inputs.data_ptr() if direct else None, inputs.data_ptr() if direct else None,
                                       inputs.numel() * 4 if direct else 0,
                                       extra[0] if extra else None, extra[1] if extra else None, extra[2] if extra else 0, st),
                      "combat_copy3")
        hs["done"] = torch.cuda.Event()
        hs["done"].record()
        # combat_copy3 reads a pinned host batch through its device mapping, which torch's caching host allocator cannot
        # see (no event is recorded against the block, unlike copy_(non_blocking=True)): keep the batch alive in this
        # staging set until `done` has completed, so a caller may drop its pinned tensor right after run()
        hs["inputs_ref"] = inputs if direct and not inputs.is_cuda else None
        if not direct:
            self.inputs.copy_(inputs, non_blocking=True)
        eC, eG, eK, eF, pl = self.eC, self.eG, self.eK, self.eF, self.pl
        for e in (eC, eG, eK) + ((eF,) if eF is not None else ()):
            e.refresh()
        k1c, k1g = self.k1[0].data_ptr(), self.k1[1].data_ptr()
        x_ptr = self.inputs.data_ptr()

        # ---- generator forward of the whole batch, ONCE for both phases.  The reference runs netG twice
        # (train_generator.py:187 on the poisoned images in eval mode, :216 on the batch in train mode);
        # the UNet has InstanceNorm without running statistics and no dropout, and its weights only
        # change at the end of Phase G, so the first call's outputs are rows of the second's.
        self._gen_forward(x_ptr, n, st, prof)
        ev_fork = torch.cuda.Event()           # what the second stream's chain depends on (see the fork below)
        ev_fork.record()

        # ================= Phase C (train_generator.py:175-212) =================
        if nb:   # the poisoned images: rows index_small[:nb] of the batch and of the generator output (:186-194)
            self._trigger_c(x_ptr, nb, k1c, st)
        ops.check(lib.combat_augment_fwd(self.cat_src.data_ptr(), self.tab_i[1].data_ptr(), aug_ptr[0], n, hw,
                                         eC.input(self.sC_train).data_ptr(), None, st), "augment 0")
        pl["C_train_f"].run(prof)
        # ---- fork (enqueued after the surrogate forward so that the main queue is never short of work while the
        # host feeds the second stream).  Everything Phase G does with the clean model and the detector depends only on the
        # generator forward above -- not on Phase C -- so that chain (trigger, two augmentations,
        # clean-model forward + input-gradient backward: ~1.3 ms of launches that individually
        # leave most of the chip idle) runs on a second stream underneath Phase C.  It owns the clean
        # model's and the detector's engines and the buffers bd / mse / d_bd2; main waits for `ev_bd`
        # before it reads the poisoned images and for `ev_side` before the generator backward.
        side = self._side_stream()
        side.wait_event(ev_fork)
        xK, xC = eK.input(self.sK_eval), eC.input(self.sC_eval)     # [2n, ...]: metric half, loss half / [n, ...]: loss images
        bd_ptr = self.bd.data_ptr()
        with torch.cuda.stream(side):
            s2 = side.cuda_stream
            self._trigger_g(x_ptr, n, k1g, s2)                                                    # :224-226
            ev_bd = torch.cuda.Event()
            ev_bd.record()
            ops.check(lib.combat_augment_fwd(x_ptr, None, aug_ptr[1], n, hw, xK.data_ptr(), None, s2), "augment 1")  # :214
            ops.check(lib.combat_augment_fwd(bd_ptr, None, aug_ptr[4], n, hw, xK[n:].data_ptr(), None, s2), "augment 4")
            pl["K_eval_f"].run(prof)               # :214 (metric half) + :250-251 (loss half)
            pl["K_bd_b"].run(prof)
            ops.check(lib.combat_augment_bwd(self.sK_bd.bufs["g.img"].data_ptr(), 8, aug_ptr[4], n, hw,
                                             self.d_bd2.data_ptr(), 0, s2), "augment 4 bwd")
            ev_side = torch.cuda.Event()
            ev_side.record()

        self._backward_allreduce(pl["C_train_b"], eC, prof)
        eC.fp.sgd_step(float(lr_c if lr_c is not None else opt.lr_C), grad_scale=1.0 / self.world)
        if not self.keep_grads:
            eC.fp.zero_grad_behind()
        eC.mark_weights_dirty()
        eC.refresh()                       # re-pack bf16 operands, fold the new running stats

        # ================= Phase G (train_generator.py:216-255; generator forward and clean-model chain: above) =====
        torch.cuda.current_stream().wait_event(ev_bd)
        if self.merge_c:
            ops.check(lib.combat_augment_fwd(x_ptr, None, aug_ptr[2], n, hw, xC.data_ptr(), None, st), "augment 2")
            ops.check(lib.combat_augment_fwd(bd_ptr, None, aug_ptr[3], n, hw, xC[n:].data_ptr(), None, st), "augment 3")
        else:
            ops.check(lib.combat_augment_fwd(bd_ptr, None, aug_ptr[3], n, hw, xC.data_ptr(), None, st), "augment 3")
        pl["C_eval_f"].run(prof)               # :228, :231 (merged: + :227)
        # ---- everything that is only logged (:227 accuracy of the updated netC on the clean images, :245-247
        # detector, :234-243 L2 / gradient-L2 terms) is forked to the second stream HERE: underneath the surrogate's
        # input-gradient pass and the generator backward.  Measured on one box (ms/step): forked after the
        # input-gradient pass 4.26, here 4.18, before the surrogate's eval forward 4.23.
        ev_late = torch.cuda.Event()
        ev_late.record()
        pl["C_bd_b"].run(prof)
        ops.check(lib.combat_augment_bwd(self.sC_bd.bufs["g.img"].data_ptr(), 8, aug_ptr[3], n, hw,
                                         self.d_bd.data_ptr(), 0, st), "augment 3 bwd")
        with torch.cuda.stream(side):
            side.wait_event(ev_late)
            s2 = side.cuda_stream
            if not self.merge_c:
                ops.check(lib.combat_augment_fwd(x_ptr, None, aug_ptr[2], n, hw, eC.input(self.sC_met).data_ptr(), None, s2),
                          "augment 2")
                pl["C_met_f"].run(prof)
            if eF is not None:
                ops.check(lib.combat_dct_u8(bd_ptr, self.D.data_ptr(), n, hw, eF.input(self.sF).data_ptr(), s2), "dct")
                pl["F_f"].run(prof)
            self._log_terms(n, s2, self.sF.bufs["logits"] if eF is not None else None)
            ev_met = torch.cuda.Event()
            ev_met.record()
        torch.cuda.current_stream().wait_event(ev_side)     # ---- join
        self._gen_backward(x_ptr, n, k1g, st, prof)
        eG.fp.sgd_step(float(lr_g if lr_g is not None else opt.lr_G), grad_scale=1.0 / self.world)
        if not self.keep_grads:
            eG.fp.zero_grad_behind()
        eG.mark_weights_dirty()
        # the next step rewrites the images, the step table and netC's operands under the second stream's readers
        torch.cuda.current_stream().wait_event(ev_met)
        self.samples += n
        self.steps_done += 1

    # ------------------------------------------------------------------ trigger-specific pieces (UNet + low-pass / clamp-mix / blur)
    def _draw(self, targets_cpu, bd_targets_cpu) -> StepRandomness:
        return draw_randomness(targets_cpu, bd_targets_cpu, self.opt, self.transforms, getattr(self.opt, "sigma", (0.1, 1.0)))

    def _gen_forward(self, x_ptr, n, st, prof) -> None:
        ops.check(lib.combat_image_to_c8(x_ptr, n, self.hw, self.eG.input(self.sG).data_ptr(), st), "c8 G")
        self.pl["G_f"].run(prof)

    def _trigger_c(self, x_ptr, nb, k1c, st) -> None:
        """Poisoned images of Phase C -> cat_src[n:] (train_generator.py:186-194)."""
        ops.check(lib.combat_trigger_fwd(x_ptr, self.eG.output(self.sG).data_ptr(), self.P.data_ptr(), k1c, float(self.opt.noise_rate),
                                         nb, self.hw, self.tab_i[0].data_ptr(), self.cat_src[self.N:].data_ptr(), None, None, st),
                  "trigger C")

    def _trigger_g(self, x_ptr, n, k1g, s2) -> None:
        """Triggered copy of the whole batch -> self.bd (+ per-plane squared error) (train_generator.py:224-226)."""
        ops.check(lib.combat_trigger_fwd(x_ptr, self.eG.output(self.sG).data_ptr(), self.P.data_ptr(), k1g, float(self.opt.noise_rate),
                                         n, self.hw, None, self.bd.data_ptr(), None, self.mse.data_ptr(), s2), "trigger G")

    def _log_terms(self, n, s2, f_logits) -> None:
        """loss_l2 / loss_grad_l2 running sums (train_generator.py:234-243; the second is logged only) and the
        detector's hit count (:245-247), in one launch (the ATen spelling of the same is _grad_l2 below: 14 launches)."""
        ops.check(lib.combat_log_terms(self.inputs.data_ptr(), self.bd.data_ptr(), self.mse.data_ptr(), n, self.hw,
                                       f_logits.data_ptr() if f_logits is not None else None, self.acc.data_ptr(),
                                       self.acc_side.data_ptr(), s2), "log terms")

    def _gen_backward(self, x_ptr, n, k1g, st, prof) -> None:
        """Image gradient (d_bd + d_bd2) + the L2 term -> generator parameter gradients (train_generator.py:253-254)."""
        hw = self.hw
        l2_scale = float(self.opt.L2_weight) / float(n * 3 * hw * hw)          # :234, :253
        ops.check(lib.combat_trigger_bwd(x_ptr, self.eG.output(self.sG).data_ptr(), self.P.data_ptr(), k1g, float(self.opt.noise_rate),
                                         n, hw, self.d_bd.data_ptr(), self.d_bd2.data_ptr(), self.bd.data_ptr(), l2_scale, 1,
                                         self.sG.buf("g.z", (n, hw, hw, 8)).data_ptr(), st), "trigger bwd")   # d_bd + d_bd2
        self._backward_allreduce(self.pl["G_b"], self.eG, prof)

    @staticmethod
    def _grad_l2(x: torch.Tensor, xb: torch.Tensor) -> torch.Tensor:
        """train_generator.py:235-243 (logged, not part of the loss)."""
        F = torch.nn.functional
        e, eb = F.pad(x, (1, 1, 2, 1)), F.pad(xb, (1, 1, 2, 1))
        return F.mse_loss(e[:, :, 1:] - e[:, :, :-1], eb[:, :, 1:] - eb[:, :, :-1]) + \
            F.mse_loss(e[:, :, :, 1:] - e[:, :, :, :-1], eb[:, :, :, 1:] - eb[:, :, :, :-1])

    def _backward_allreduce(self, plan, eng, prof) -> None:
        backward_allreduce(plan, eng, self.pg, self.world, self._reducers, prof)

    # ------------------------------------------------------------------ metrics
    def _slot_sets(self):
        """(sC_train, sC_eval, sK_eval) of every batch size seen so far."""
        names = ("sC_train", "sC_eval", "sK_eval", "sC_met")
        out = [tuple(getattr(self, k) for k in names)] if self.N else []
        out += [tuple(d[k] for k in names) for n, d in self._sets.items() if n != self.N]
        return out

    def read_metrics(self, reset: bool = False) -> Dict[str, float]:
        """One host sync: the running sums the reference prints each step (:257-290), over every
        batch size run since the last reset."""
        acc = self.acc.cpu()
        total = max(float(self.samples), 1.0)
        w_cm = float(self.opt.clean_model_weight) or 1.0
        out = {"samples": total, "loss_c_sum": 0.0, "loss_ce_sum": 0.0, "clean_model_loss_sum": 0.0,
               "loss_l2_sum": float(acc[0]), "loss_grad_l2_sum": float(acc[1]), "clean_correct": 0, "bd_correct": 0,
               "f_correct": int(self.acc_side.cpu()), "clean_model_correct": 0, "clean_model_bd_ba": 0, "clean_model_bd_asr": 0,
               "train_correct": 0}
        for sCt, sCe, sKe, sCm in self._slot_sets():
            cC, cE, kE = self.eC.head_bufs(sCt), self.eC.head_bufs(sCe), self.eK.head_bufs(sKe)
            out["loss_c_sum"] += float(cC["loss"])
            out["loss_ce_sum"] += float(cE["loss"])
            out["clean_model_loss_sum"] += float(kE["loss"]) / w_cm
            out["clean_correct"] += int(sCm.bufs["correct0"][0]) if "correct0" in sCm.bufs else int(self.eC.head_bufs(sCm)["correct"][0])
            out["bd_correct"] += int(cE["correct"][0])
            out["clean_model_correct"] += int(sKe.bufs["correct0"][0])
            out["clean_model_bd_ba"] += int(kE["correct"][0])
            out["clean_model_bd_asr"] += int(kE["correct"][1])
            out["train_correct"] += int(cC["correct"][0])
        if reset:
            self.reset_metrics()
        return out

    def reset_metrics(self) -> None:
        self.samples = 0
        self.acc.zero_()
        self.acc_side.zero_()
        for sCt, sCe, sKe, sCm in self._slot_sets():
            for eng, slot in ((self.eC, sCt), (self.eC, sCe), (self.eK, sKe), (self.eC, sCm)):
                h = eng.head_bufs(slot)
                h["loss"].zero_()
                h["correct"].zero_()
                for k in ("loss0", "correct0"):
                    if k in slot.bufs:
                        slot.bufs[k].zero_()


class WanetStep(AlternatedStep):
    """The alternated step with the warping trigger (reference train_generator_wanet.py:132-237): the generator is a
    GridGenerator whose [2][S][S] field is bicubically upsampled, blended with the identity grid at ``grid_rescale``
    and used to ``grid_sample`` the images (:151-157, :196-202); loss_l2 = MSE(noise_grid, 0) (:212).  Phase
    structure, streams, poison selection, augmentation, classifiers and optimisers are AlternatedStep's.  No Gaussian
    blur is drawn (the reference's WaNet loop has none), so the torch RNG stream is consumed by the augmentation only."""

    WARP_GROUPS = 8     # image ranges of the warp backward (partial sums: deterministic, no atomics)

    def _draw(self, targets_cpu, bd_targets_cpu) -> StepRandomness:
        """train_generator_wanet.py:148, :159, :182, :203-204, :226: num_bd and five augmentation calls -- no blur."""
        n = targets_cpu.shape[0]
        n_trg = int((targets_cpu == bd_targets_cpu).sum())
        num_bd = int(np.sum(np.random.rand(n_trg) < self.opt.pc))
        return StepRandomness(num_bd, 0.5, 0.5, [self.transforms.sample(n) for _ in range(5)])

    def _gen_slot(self, n):
        return None

    def _gen_plans(self) -> dict:
        return {}

    def _setup(self, n: int):
        super()._setup(n)
        if getattr(self, "_wpartial_n", 0) != n:
            self._wpartial = torch.zeros(self.WARP_GROUPS, self.hw, self.hw, 2, dtype=f32, device=self.dev)
            self._wpartial_n = n

    def _gen_forward(self, x_ptr, n, st, prof) -> None:
        self.g = self.eG.forward_grid(self.hw, float(self.opt.grid_rescale), st)

    def _trigger_c(self, x_ptr, nb, k1c, st) -> None:
        ops.check(lib.combat_warp_fwd(x_ptr, self.tab_i[0].data_ptr(), self.g["grid"].data_ptr(), 0, nb, self.hw,
                                      self.cat_src[self.N:].data_ptr(), st), "warp C")

    def _trigger_g(self, x_ptr, n, k1g, s2) -> None:
        ops.check(lib.combat_warp_fwd(x_ptr, None, self.g["grid"].data_ptr(), 0, n, self.hw, self.bd.data_ptr(), s2), "warp G")

    def _log_terms(self, n, s2, f_logits) -> None:
        if f_logits is not None:
            self.acc_side += (f_logits.argmax(1) == 1).sum()
        ng = self.g["noise_grid"]                      # [H][H][2]; the reference's noise_grid is B equal copies
        self.acc[0] += ng.pow(2).mean()
        F = torch.nn.functional
        e = F.pad(ng[None], (1, 1, 2, 1))              # F.pad of the [B, H, H, 2] tensor pads (H, 2): reproduce that (:214)
        self.acc[1] += (e[:, :, 1:] - e[:, :, :-1]).pow(2).mean() + (e[:, :, :, 1:] - e[:, :, :, :-1]).pow(2).mean()

    def _gen_backward(self, x_ptr, n, k1g, st, prof) -> None:
        eG = self.eG
        ops.check(_zero_grad(eG.fp, st), "zero_grad G")
        ops.check(lib.combat_warp_bwd(x_ptr, self.d_bd.data_ptr(), self.d_bd2.data_ptr(), self.g["grid"].data_ptr(), 0, n, self.hw,
                                      self.WARP_GROUPS, self._wpartial.data_ptr(), st), "warp bwd")
        eG.backward_field(self._wpartial, self.WARP_GROUPS, self.hw, float(self.opt.grid_rescale), float(self.opt.L2_weight), st)
        if self.world > 1:
            # only fc1.bias, fc2.weight and fc2.bias ever receive a gradient (GridEngine): one contiguous range of
            # ~640 floats travels, not the 19-MB flat buffer whose remainder is exactly zero on every rank
            lo, hi = eG.head_grad_range()
            torch.distributed.all_reduce(eG.fp.grad[lo:hi], group=self.pg)


def _zero_grad(fp, st):
    from .engine import _zero_grad_call
    return _zero_grad_call(fp, st)


class ClassifierStep:
    """Phase-C-only training step: the loop bodies of train_clean_classifier.py:75-121 (no generator) and
    train_victim.py:102-141 (frozen generator poisons the images the dataset flags).  Batch order as in
    the victim script: [poisoned images with the trigger, all other images] (:125-127)."""

    def __init__(self, netC, opt, netG=None, process_group=None):
        self.opt, self.netC, self.netG = opt, netC, netG
        self.dev = next(netC.parameters()).device
        self.eC: PreActEngine = netC._net_engine()
        self.eG = netG._net_engine() if netG is not None else None      # UnetEngine, or GridEngine (WaNet)
        # train_victim_wanet.py:85-96: the frozen GridGenerator's field warps the poisoned images (no low-pass, no blur)
        self.wanet = netG is not None and getattr(netG, "arch", "") == "gridgen"
        if netG is not None and getattr(netG, "arch", "") not in ("unet", "gridgen"):
            raise ValueError("ClassifierStep: unsupported generator %r (UnetGenerator or GridGenerator expected)" % type(netG).__name__)
        self.pg = process_group
        self.world = torch.distributed.get_world_size(process_group) if process_group is not None else 1
        if self.world > 1 or (FORCE_ALLREDUCE and process_group is not None):
            self.eC.flush_at_marks = True
        self.hw = opt.input_height
        self.P = trigger.lowpass_matrix(self.hw, opt.ratio).to(self.dev)
        self.k1 = torch.zeros(3, dtype=f32, device=self.dev)
        self.transforms = PostTensorTransform(opt)
        self.N = 0
        self._sets: Dict[int, dict] = {}
        self._small: Dict[int, tuple] = {}
        self._reducers: Dict = {}

    _PER_N = ("cat_src", "tab_i", "tab_f", "slot", "fwd", "bwd")

    def _setup(self, n):
        """Buffers, slot and plans are per batch size and cached (as AlternatedStep._setup): the ragged last
        batch of an epoch gets its own set once, and the loss / accuracy cells of every set survive until
        `read_metrics` sums (and resets) them all."""
        if n == self.N:
            return
        if self.N:
            self._sets[self.N] = {k: getattr(self, k) for k in self._PER_N}
        if n in self._sets:
            for k, v in self._sets[n].items():
                setattr(self, k, v)
            self.N = n
            return
        self.N, hw, dev = n, self.hw, self.dev
        self.cat_src = torch.zeros(2 * n, 3, hw, hw, dtype=f32, device=dev)
        self.tab_i = torch.zeros(2, n, dtype=torch.int32, device=dev)
        self.tab_f = torch.zeros(n, 4, dtype=f32, device=dev)
        self.slot = self.eC.slot("V.train", n, hw)
        hb = FUSED_HEAD and type(self.eC) is PreActEngine
        self.fwd = self.eC.forward_plan(self.slot, True, **(dict(head_bwd=True) if hb else {}))
        self.bwd = self.eC.backward_train_plan(self.slot, **(dict(head_done=True) if hb else {}))

    def run(self, inputs: torch.Tensor, targets_cpu: torch.Tensor, poisoned_cpu: Optional[torch.Tensor] = None,
            lr: Optional[float] = None) -> None:
        opt = self.opt
        n = inputs.shape[0]
        self._setup(n)
        hw, st = self.hw, torch.cuda.current_stream().cuda_stream
        targets_cpu = targets_cpu.cpu()
        if poisoned_cpu is None or self.eG is None:
            poisoned_cpu = torch.zeros(n, dtype=torch.bool)
        trg = poisoned_cpu.cpu().nonzero()[:, 0]
        ntrg = (~poisoned_cpu.cpu()).nonzero()[:, 0]      # intent of train_victim.py:121
        nb = int(trg.numel())
        perm = torch.cat([trg, ntrg]).to(torch.int32)
        tot = targets_cpu[perm.long()].clone()
        if nb:
            tot[:nb] = create_targets_bd(targets_cpu[trg], opt)
        idx = torch.zeros(2, n, dtype=torch.int32)
        idx[0, :nb] = perm[:nb]
        idx[1] = perm
        if nb:
            idx[1, :nb] = torch.arange(n, n + nb, dtype=torch.int32)
        self.tab_i.copy_(idx)
        self.last_poisoned = (trg, nb)     # (image logging: train_victim_wanet.py:127-133)
        aug = self.transforms.sample(n)
        if aug is not None:
            self.tab_f.copy_(torch.from_numpy(aug))
        self.cat_src[:n].copy_(inputs)
        h = self.eC.head_bufs(self.slot)
        h["targets"].copy_(tot)
        self.eC.refresh()
        if nb and self.wanet:
            g = self.eG.forward_grid(hw, float(opt.grid_rescale), st)
            ops.check(lib.combat_warp_fwd(self.cat_src.data_ptr(), self.tab_i[0].data_ptr(), g["grid"].data_ptr(), 0, nb, hw,
                                          self.cat_src[n:].data_ptr(), st), "warp")
        elif nb:
            self.eG.refresh()
            nbk = min(bucket(nb), n)
            if nbk not in self._small:
                s = self.eG.slot("V.small", nbk, hw)
                self._small[nbk] = (s, self.eG.forward_plan(s), torch.zeros(nbk, 3, hw, hw, dtype=f32, device=self.dev))
            sS, plan_small, tochange = self._small[nbk]
            self.k1.copy_(torch.from_numpy(trigger.gaussian_kernel1d(
                trigger.sample_sigma(getattr(opt, "sigma", (0.1, 1.0))), opt.kernel_size)))
            ops.check(lib.combat_augment_fwd(self.cat_src.data_ptr(), self.tab_i[0].data_ptr(), None, nbk, hw,
                                             self.eG.input(sS).data_ptr(), tochange.data_ptr(), st), "gather poisoned")
            plan_small.run()
            ops.check(lib.combat_trigger_fwd(tochange.data_ptr(), self.eG.output(sS).data_ptr(), self.P.data_ptr(),
                                             self.k1.data_ptr(), float(opt.noise_rate), nb, hw, None,
                                             self.cat_src[n:].data_ptr(), None, None, st), "trigger")
        ops.check(lib.combat_augment_fwd(self.cat_src.data_ptr(), self.tab_i[1].data_ptr(),
                                         self.tab_f.data_ptr() if aug is not None else None, n, hw,
                                         self.eC.input(self.slot).data_ptr(), None, st), "augment")
        self.fwd.run()
        backward_allreduce(self.bwd, self.eC, self.pg, self.world, self._reducers)   # buckets overlap the backward
        self.eC.fp.sgd_step(float(lr if lr is not None else opt.lr_C), grad_scale=1.0 / self.world)
        self.eC.mark_weights_dirty()

    def poisoned_pair(self):
        """(inputs_toChange, inputs_bd) of the last batch -- the two tensors the reference's debugging image stacks
        (train_victim_wanet.py:129) -- or None if that batch had no poisoned image."""
        trg, nb = getattr(self, "last_poisoned", (None, 0))
        if not nb:
            return None
        n = self.N
        return self.cat_src[:n][trg.to(self.dev)].clone(), self.cat_src[n:n + nb].clone()

    def _slots(self):
        out = [self.slot] if self.N else []
        return out + [d["slot"] for n, d in self._sets.items() if n != self.N]

    def read_metrics(self, reset=False):
        """Running loss sum / correct count over every batch size run since the last reset (one host sync)."""
        out = {"loss_sum": 0.0, "correct": 0}
        for slot in self._slots():
            h = self.eC.head_bufs(slot)
            out["loss_sum"] += float(h["loss"])
            out["correct"] += int(h["correct"][0])
            if reset:
                h["loss"].zero_()
                h["correct"].zero_()
        return out
