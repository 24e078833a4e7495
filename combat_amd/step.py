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

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np
import torch

from . import ops, trigger
from ._lib import lib
from .augment import PostTensorTransform
from .engine import FreqEngine, PreActEngine, UnetEngine, f32

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


class AlternatedStep:
    """Owns the engines, slots and small device tables of one rank's step."""

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
        self.hw = opt.input_height
        self.N = 0
        dev = self.dev
        self.P = trigger.lowpass_matrix(self.hw, opt.ratio).to(dev)
        self.D = trigger.dct_matrix(self.hw).float().to(dev)
        self.k1 = torch.zeros(2, 3, dtype=f32, device=dev)      # sigma_c, sigma_g kernels
        self.transforms = PostTensorTransform(opt)
        self.acc = torch.zeros(8, dtype=torch.float64, device=dev)  # running sums for logging
        self._host = None
        self.steps_done = 0

    # ------------------------------------------------------------------ buffers per batch size
    def _setup(self, n: int):
        if n == self.N:
            return
        self.N, dev, hw = n, self.dev, self.hw
        self.inputs = torch.empty(n, 3, hw, hw, dtype=f32, device=dev)
        self.cat_src = torch.zeros(2 * n, 3, hw, hw, dtype=f32, device=dev)   # [inputs ; poisoned images]
        self.bd = torch.empty(n, 3, hw, hw, dtype=f32, device=dev)
        self.d_bd = torch.empty(n, 3, hw, hw, dtype=f32, device=dev)
        self.mse = torch.empty(n, dtype=f32, device=dev)
        # one host->device table per step: [5 aug tables | index_small | index_total | k1 x2]
        self.tab_f = torch.zeros(5, n, 4, dtype=f32, device=dev)
        self.tab_i = torch.zeros(2, n, dtype=torch.int32, device=dev)
        self.h_tab_f = torch.zeros(5, n, 4, dtype=f32).pin_memory()
        self.h_tab_i = torch.zeros(2, n, dtype=torch.int32).pin_memory()
        self.h_k1 = torch.zeros(2, 3, dtype=f32).pin_memory()
        self.h_targets = torch.zeros(3, n, dtype=torch.int64).pin_memory()    # targets, bd_targets, total_targets
        self.d_targets = torch.zeros(3, n, dtype=torch.int64, device=dev)
        eC, eK, eG = self.eC, self.eK, self.eG
        self.sC_train = eC.slot("C.train", n, hw)
        self.sC_clean = eC.slot("C.clean", n, hw)
        self.sC_bd = eC.slot("C.bd", n, hw)
        self.sK_clean = eK.slot("K.clean", n, hw)
        self.sK_bd = eK.slot("K.bd", n, hw)
        self.sG = eG.slot("G", n, hw)
        self.sF = self.eF.slot("F", n, hw) if self.eF is not None else None
        w_cm = float(self.opt.clean_model_weight)
        self.pl = dict(
            C_train_f=eC.forward_plan(self.sC_train, True), C_train_b=eC.backward_train_plan(self.sC_train),
            C_clean_f=eC.forward_plan(self.sC_clean, False),
            C_bd_f=eC.forward_plan(self.sC_bd, False), C_bd_b=eC.backward_eval_plan(self.sC_bd, 1.0),
            K_clean_f=eK.forward_plan(self.sK_clean, False),
            K_bd_f=eK.forward_plan(self.sK_bd, False, w_cm, True), K_bd_b=eK.backward_eval_plan(self.sK_bd, w_cm),
            G_f=eG.forward_plan(self.sG), G_b=eG.backward_plan(self.sG),
        )
        if self.sF is not None:
            self.pl["F_f"] = self.eF.forward_plan(self.sF)
        # targets live in the head buffers of each slot: point them at the shared table once
        self._targets_of = {
            "C.train": (self.sC_train, 2, None), "C.clean": (self.sC_clean, 0, None), "C.bd": (self.sC_bd, 1, None),
            "K.clean": (self.sK_clean, 0, None), "K.bd": (self.sK_bd, 0, 1)}
        self._gen_small: Dict[int, tuple] = {}

    def _small(self, nbk: int):
        """Generator slot + plan for a poisoned sub-batch bucket."""
        if nbk not in self._gen_small:
            s = self.eG.slot("G.small", nbk, self.hw)
            self._gen_small[nbk] = (s, self.eG.forward_plan(s),
                                    torch.zeros(nbk, 3, self.hw, self.hw, dtype=f32, device=self.dev))
        return self._gen_small[nbk]

    # ------------------------------------------------------------------ one step
    def run(self, inputs: torch.Tensor, targets_cpu: torch.Tensor, rnd: Optional[StepRandomness] = None,
            lr_c: Optional[float] = None, lr_g: Optional[float] = None, prof: Optional[list] = None) -> None:
        """inputs: float32 [B,3,H,W] (device or pinned host); targets_cpu: int64 [B] on the host,
        as the DataLoader yields them (train_generator.py:170-171)."""
        opt = self.opt
        n = inputs.shape[0]
        self._setup(n)
        hw, st = self.hw, torch.cuda.current_stream().cuda_stream
        targets_cpu = targets_cpu.cpu()
        bd_targets_cpu = create_targets_bd(targets_cpu, opt).cpu()
        if rnd is None:
            rnd = draw_randomness(targets_cpu, bd_targets_cpu, opt, self.transforms, getattr(opt, "sigma", (0.1, 1.0)))
        nb = rnd.num_bd
        # ---- host tables (train_generator.py:181-204: batch order [poisoned, rest of target class, others])
        trg = (targets_cpu == bd_targets_cpu).nonzero()[:, 0]
        ntrg = (targets_cpu != bd_targets_cpu).nonzero()[:, 0]
        perm = torch.cat([trg, ntrg]).to(torch.int32)
        self.h_targets[0].copy_(targets_cpu)
        self.h_targets[1].copy_(bd_targets_cpu)
        tot = targets_cpu[perm.long()].clone()
        tot[:nb] = bd_targets_cpu[perm[:nb].long()]
        self.h_targets[2].copy_(tot)
        idx_small, idx_total = self.h_tab_i[0], self.h_tab_i[1]
        idx_small.zero_()
        idx_small[:nb] = perm[:nb]
        idx_total.copy_(perm)
        if nb:
            idx_total[:nb] = torch.arange(n, n + nb, dtype=torch.int32)
        aug_ptr = []
        for i, a in enumerate(rnd.aug):
            if a is not None:
                self.h_tab_f[i].copy_(torch.from_numpy(a))
            aug_ptr.append(self.tab_f[i].data_ptr() if a is not None else None)
        self.h_k1[0].copy_(torch.from_numpy(trigger.gaussian_kernel1d(rnd.sigma_c, opt.kernel_size)))
        self.h_k1[1].copy_(torch.from_numpy(trigger.gaussian_kernel1d(rnd.sigma_g, opt.kernel_size)))
        self.tab_f.copy_(self.h_tab_f, non_blocking=True)
        self.tab_i.copy_(self.h_tab_i, non_blocking=True)
        self.k1.copy_(self.h_k1, non_blocking=True)
        self.d_targets.copy_(self.h_targets, non_blocking=True)
        self.inputs.copy_(inputs, non_blocking=True)
        self.cat_src[:n].copy_(self.inputs)
        for name, (slot, ti, t2) in self._targets_of.items():
            eng = self.eK if name.startswith("K") else self.eC
            h = eng.head_bufs(slot)
            h["targets"].copy_(self.d_targets[ti])
            if t2 is not None:
                h["targets2"].copy_(self.d_targets[t2])
        eC, eG, eK, eF, pl = self.eC, self.eG, self.eK, self.eF, self.pl
        for e in (eC, eG, eK) + ((eF,) if eF is not None else ()):
            e.refresh()
        P_, k1c, k1g = self.P.data_ptr(), self.k1[0].data_ptr(), self.k1[1].data_ptr()
        rate = float(opt.noise_rate)
        x_ptr = self.inputs.data_ptr()

        # ================= Phase C (train_generator.py:175-212) =================
        if nb:
            nbk = min(bucket(nb), n)
            sS, plan_small, tochange = self._small(nbk)
            ops.check(lib.combat_augment_fwd(x_ptr, self.tab_i[0].data_ptr(), None, nbk, hw, eG.input(sS).data_ptr(),
                                             tochange.data_ptr(), st), "gather poisoned")
            plan_small.run(prof)
            ops.check(lib.combat_trigger_fwd(tochange.data_ptr(), eG.output(sS).data_ptr(), P_, k1c, rate, nb, hw,
                                             self.cat_src[n:].data_ptr(), None, None, st), "trigger C")
        ops.check(lib.combat_augment_fwd(self.cat_src.data_ptr(), self.tab_i[1].data_ptr(), aug_ptr[0], n, hw,
                                         eC.input(self.sC_train).data_ptr(), None, st), "augment 0")
        pl["C_train_f"].run(prof)
        pl["C_train_b"].run(prof)
        self._allreduce(eC)
        eC.fp.sgd_step(float(lr_c if lr_c is not None else opt.lr_C), grad_scale=1.0 / self.world)
        eC.mark_weights_dirty()
        eC.refresh()                       # re-pack bf16 operands, fold the new running stats
        ops.check(lib.combat_augment_fwd(x_ptr, None, aug_ptr[1], n, hw, eK.input(self.sK_clean).data_ptr(), None, st),
                  "augment 1")
        pl["K_clean_f"].run(prof)              # :214 metric only

        # ================= Phase G (train_generator.py:216-255) =================
        ops.check(lib.combat_image_to_c8(x_ptr, n, hw, eG.input(self.sG).data_ptr(), st), "c8 G")
        pl["G_f"].run(prof)
        noise = eG.output(self.sG)
        ops.check(lib.combat_trigger_fwd(x_ptr, noise.data_ptr(), P_, k1g, rate, n, hw, self.bd.data_ptr(), None,
                                         self.mse.data_ptr(), st), "trigger G")
        bd_ptr = self.bd.data_ptr()
        ops.check(lib.combat_augment_fwd(x_ptr, None, aug_ptr[2], n, hw, eC.input(self.sC_clean).data_ptr(), None, st),
                  "augment 2")
        pl["C_clean_f"].run(prof)              # :227 metric only
        ops.check(lib.combat_augment_fwd(bd_ptr, None, aug_ptr[3], n, hw, eC.input(self.sC_bd).data_ptr(), None, st),
                  "augment 3")
        pl["C_bd_f"].run(prof)                 # :228, :231
        pl["C_bd_b"].run(prof)
        ops.check(lib.combat_augment_bwd(self.sC_bd.bufs["g.img"].data_ptr(), 8, aug_ptr[3], n, hw,
                                         self.d_bd.data_ptr(), 0, st), "augment 3 bwd")
        if eF is not None:                 # :245-247 metric only
            ops.check(lib.combat_dct_u8(bd_ptr, self.D.data_ptr(), n, hw, eF.input(self.sF).data_ptr(), st), "dct")
            pl["F_f"].run(prof)
            self.acc[4] += (self.sF.bufs["logits"].argmax(1) == 1).sum()
        ops.check(lib.combat_augment_fwd(bd_ptr, None, aug_ptr[4], n, hw, eK.input(self.sK_bd).data_ptr(), None, st),
                  "augment 4")
        pl["K_bd_f"].run(prof)                 # :250-251
        pl["K_bd_b"].run(prof)
        ops.check(lib.combat_augment_bwd(self.sK_bd.bufs["g.img"].data_ptr(), 8, aug_ptr[4], n, hw,
                                         self.d_bd.data_ptr(), 1, st), "augment 4 bwd")
        l2_scale = float(opt.L2_weight) / float(n * 3 * hw * hw)          # :234, :253
        ops.check(lib.combat_trigger_bwd(x_ptr, noise.data_ptr(), P_, k1g, rate, n, hw, self.d_bd.data_ptr(), bd_ptr,
                                         l2_scale, 1, self.sG.buf("g.z", (n, hw, hw, 8)).data_ptr(), st), "trigger bwd")
        pl["G_b"].run(prof)
        self._allreduce(eG)
        eG.fp.sgd_step(float(lr_g if lr_g is not None else opt.lr_G), grad_scale=1.0 / self.world)
        eG.mark_weights_dirty()
        # ---- logged-only terms (:234-243)
        self.acc[0] += self.mse.sum() / float(n * 3 * hw * hw)
        self.acc[1] += self._grad_l2(self.inputs, self.bd)
        self.acc[7] += n
        self.steps_done += 1

    @staticmethod
    def _grad_l2(x: torch.Tensor, xb: torch.Tensor) -> torch.Tensor:
        """train_generator.py:235-243 (logged, not part of the loss)."""
        F = torch.nn.functional
        e, eb = F.pad(x, (1, 1, 2, 1)), F.pad(xb, (1, 1, 2, 1))
        return F.mse_loss(e[:, :, 1:] - e[:, :, :-1], eb[:, :, 1:] - eb[:, :, :-1]) + \
            F.mse_loss(e[:, :, :, 1:] - e[:, :, :, :-1], eb[:, :, :, 1:] - eb[:, :, :, :-1])

    def _allreduce(self, eng) -> None:
        if self.world > 1:
            torch.distributed.all_reduce(eng.fp.grad, group=self.pg)

    # ------------------------------------------------------------------ metrics
    def read_metrics(self, reset: bool = False) -> Dict[str, float]:
        """One host sync: the running sums the reference prints each step (:257-290)."""
        hb = lambda eng, slot: eng.head_bufs(slot)
        cC, cCl, cBd = hb(self.eC, self.sC_train), hb(self.eC, self.sC_clean), hb(self.eC, self.sC_bd)
        kCl, kBd = hb(self.eK, self.sK_clean), hb(self.eK, self.sK_bd)
        acc = self.acc.cpu()
        total = max(float(acc[7]), 1.0)
        w_cm = float(self.opt.clean_model_weight) or 1.0
        out = {
            "samples": total,
            "loss_c_sum": float(cC["loss"]), "loss_ce_sum": float(cBd["loss"]),
            "clean_model_loss_sum": float(kBd["loss"]) / w_cm,
            "loss_l2_sum": float(acc[0]), "loss_grad_l2_sum": float(acc[1]),
            "clean_correct": int(cCl["correct"][0]), "bd_correct": int(cBd["correct"][0]),
            "f_correct": int(acc[4]), "clean_model_correct": int(kCl["correct"][0]),
            "clean_model_bd_ba": int(kBd["correct"][0]), "clean_model_bd_asr": int(kBd["correct"][1]),
            "train_correct": int(cC["correct"][0]),
        }
        if reset:
            self.reset_metrics()
        return out

    def reset_metrics(self) -> None:
        self.acc.zero_()
        for eng, slot in ((self.eC, self.sC_train), (self.eC, self.sC_clean), (self.eC, self.sC_bd),
                          (self.eK, self.sK_clean), (self.eK, self.sK_bd)):
            h = eng.head_bufs(slot)
            h["loss"].zero_()
            h["correct"].zero_()
