"""Per-network forward/backward schedules over the combat_hip C ABI.

A network is executed as a *plan*: the ordered list of C-ABI calls (arguments marshalled once,
buffers pre-allocated in a ``Slot``) that the reference's ``module(x)`` / ``loss.backward()``
expand to, with the normalisation / activation / residual element-wise work folded into the
convolution prologues and epilogues.  Plans are replayed with no Python work per kernel beyond
one ctypes call, and are what gets captured into a HIP graph.

Reference call sites: PreActResNet forward classifier_models/preact_resnet.py:32-40, 93-102;
UnetGenerator forward networks/models.py:321-341; FrequencyModel forward
defenses/frequency_based/model.py:49-52; their backward passes are what autograd derives.
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, List, Optional, Tuple

import torch

from . import ops
from ._lib import STATS_PER_WORKGROUP, TILE_D128x32, TILE_D128x64, TILE_S128x64, CombatHipError, ConvArgs, lib
from .nets import UNET_LAYERS
from .ops import Affine, PackedConv, bf16

# Statistics over the whole batch (BatchNorm) ask the convolution for one partial row per workgroup instead of one per
# wave: 256 rows instead of 1024-4096 on the 32 x 32 and 16 x 16 layers, so the fused normalisation launches reduce them
# themselves and no norm_stage1 launch stands between a convolution and its normalisation (18 launches per alternated
# step).  COMBAT_STATS_PER_WAVE=1 restores the per-wave rows (A/B measurements).
BATCH_STATS_ROWS = 0 if os.environ.get("COMBAT_STATS_PER_WAVE") == "1" else STATS_PER_WORKGROUP

f32 = torch.float32


def _p(t):
    return None if t is None else t.data_ptr()


class Plan:
    """An ordered list of C-ABI calls with pre-marshalled arguments (stream appended at run)."""

    serial = False   # True: auxiliary-stream calls run in line (kernel-level measurements: one kernel at a time)
    serial_shortcuts = False   # True: shortcut input gradients as launches of their own (per-launch profiling, A/B)
    default_aux_queues = 1   # auxiliary queues a new plan deals its aux calls to (see aux_queues below)
    flush_at_marks = False   # data parallel: deferred weight-gradient reductions are issued at every all-reduce mark
    #                          (the gradients above a mark must be final there); otherwise all of them at the end of
    #                          the plan, on its own stream, each behind its weight gradient only
    # Deferred weight-gradient reductions (combat_wgrad_args.defer_reduce + Plan.flush_reduces) are OFF by default:
    # measured on the alternated step (tools/quick_step.py, same box, same process order) 4.31 ms/step with every
    # reduction right behind its kernel on the auxiliary queue, 4.45-4.51 with the reductions on the plan's own stream
    # behind per-call events (private 19-MB slab regions or a ring of two: no difference; private regions WITHOUT
    # deferral: 4.32, so it is not the slabs leaving the Infinity Cache) -- twenty event records between the auxiliary
    # queue's kernels and twenty cross-queue waits cost more than the 8-us reductions they take off that queue.
    defer_reduces = os.environ.get("COMBAT_DEFER_REDUCE", "0") == "1"

    def __init__(self, name: str):
        self.name = name
        self.calls: List[Tuple] = []
        self._keep: List = []
        self.marks: Dict[int, int] = {}   # call index -> flat-gradient offset complete after that call
        self.aux: Dict[int, int] = {}     # call index -> auxiliary queue of the calls that may run beside the plan's own stream
        self.aux_queues = Plan.default_aux_queues   # auxiliary queues the aux calls are dealt to, round-robin (each with its own
        #                                   weight-gradient scratch).  Measured: 2 for the generator's backward 4.19 -> 4.23
        #                                   ms/step, 2 for the surrogate's 4.14 -> 4.17: one queue it stays
        self.wgrads: List[Tuple] = []     # (call index, WgradArgs) of the weight-gradient launches
        self._ws = None
        self._cplan = None                # combat_plan* of the compiled form (built on first run)
        self._prog = None
        self.after: Dict[int, int] = {}   # call index -> index of an earlier call it waits for (on whatever queue that ran)
        self.pending_reduces: List[Tuple] = []   # (what, WgradArgs, index of its weight-gradient call)
        self.chain: Dict[int, Tuple] = {}        # auxiliary queue -> (what, WgradArgs) of the launch whose slabs await the next one
        self.chain_count: Dict[int, int] = {}    # auxiliary queue -> slab-leaving launches so far (alternates the two regions)

    def add(self, what: str, cfunc, *args, aux: bool = False, after: Optional[int] = None) -> None:
        """aux: the call's result is only consumed after the plan (weight gradients: by the optimiser /
        all-reduce), so it may run on the auxiliary stream beside the calls that follow it.
        after: index of an earlier call (typically an aux one) this call must wait for."""
        self.calls.append((cfunc, args, what))
        self._prog = None
        if aux:
            self.aux[len(self.calls) - 1] = self.next_aux_queue()
        if after is not None:
            self.after[len(self.calls) - 1] = after

    def flush_reduces(self) -> None:
        """Issue the reductions of the weight gradients recorded with defer_reduce so far: on the plan's own stream,
        each waiting for its weight-gradient launch only.  The auxiliary queue then holds a chain of weight-gradient
        kernels instead of (kernel, reduction) pairs -- the reductions were 10 of every 40 us of what is the critical
        path of both training backward passes -- and the reductions run while it works on the next layers."""
        for what, a, ci in self.pending_reduces:
            self.add(what, lib.combat_conv_wgrad_reduce, ctypes.byref(a), after=ci)
        self.pending_reduces = []
        self.chain = {}      # (reduce-behind: the chains' last launches were in the list above; the next launch starts a new chain)

    def next_aux_queue(self) -> int:
        """Queue the next aux call will be given (round-robin over `aux_queues`)."""
        return len(self.aux) % self.aux_queues

    def hold(self, *objs) -> None:
        self._keep.extend(objs)

    def merge_convs(self, i: int) -> None:
        """Calls i and i + 1 are two combat_conv_gemm launches with no data dependence between them (a residual
        block's shortcut and its first convolution over the same input): make them ONE combat_conv_gemm_pair call.
        Only while nothing recorded after them is indexed by position (forward plans: no marks, no aux calls)."""
        (f0, a0, w0), (f1, a1, w1) = self.calls[i], self.calls[i + 1]
        assert f0 is lib.combat_conv_gemm and f1 is lib.combat_conv_gemm and not self.marks and not self.aux
        self.calls[i:i + 2] = [(lib.combat_conv_gemm_pair, (a0[0], a1[0]), w0 + "+" + w1)]
        self._prog = None

    def workspace(self, device) -> torch.Tensor:
        """Scratch of this plan's convolutions (split reductions of skinny layers).  Per plan: the calls of
        one plan run one after the other on one stream, two plans may run concurrently."""
        if self._ws is None:
            self._ws = torch.empty(32 << 20, dtype=torch.uint8, device=device)
        return self._ws

    def mark(self, grad_offset: int) -> None:
        """Every gradient at flat offset >= grad_offset is final once the calls recorded so far ran."""
        if self.flush_at_marks or grad_offset == 0:
            self.flush_reduces()
        self.marks[len(self.calls) - 1] = grad_offset

    compiled = os.environ.get("COMBAT_PLAN_PY", "0") != "1"   # False: replay every plan from Python (debugging / A-B timing)

    def _program(self):
        """The C-side form of this plan (csrc/plan.cpp), built on first use: every C-ABI call is captured into a
        combat_plan with its queue; Python callables (the gradient-zeroing hand-off) and the all-reduce marks split
        it into ranges.  Returns [("c", begin, end) | ("py", func, args, what) | ("mark", offset)]."""
        if self._prog is not None:
            return self._prog
        if self._cplan:
            lib.combat_plan_destroy(self._cplan)
        cp = self._cplan = lib.combat_plan_create()
        if not cp:
            raise CombatHipError("combat_plan_create failed")
        prog, names, begin = [], [], None
        cindex = {}      # plan call index -> index inside the combat_plan
        for ci, (cfunc, args, what) in enumerate(self.calls):
            captured = False
            if getattr(cfunc, "argtypes", None) is not None:       # a ctypes entry point
                ops.check(lib.combat_plan_record(cp, self.aux.get(ci, -1)), "combat_plan_record")
                try:
                    rc = cfunc(*args, None)
                finally:
                    captured = not lib.combat_plan_record_cancel()
                if captured and rc:
                    raise CombatHipError("%s/%s: recording failed (%d)" % (self.name, what, rc))
            if captured:
                names.append(what)
                cindex[ci] = len(names) - 1
                if ci in self.after:
                    ops.check(lib.combat_plan_set_after(cp, cindex[self.after[ci]]), "combat_plan_set_after")
                if begin is None:
                    begin = len(names) - 1
            else:
                if begin is not None:
                    prog.append(("c", begin, len(names)))
                    begin = None
                assert ci not in self.aux and ci not in self.after, "only C-ABI calls can run on an auxiliary queue / wait for a call"
                prog.append(("py", cfunc, args, what))
            if ci in self.marks:
                if begin is not None:
                    prog.append(("c", begin, len(names)))
                    begin = None
                prog.append(("mark", self.marks[ci]))
        if begin is not None:
            prog.append(("c", begin, len(names)))
        assert lib.combat_plan_size(cp) == len(names)
        # without an all-reduce callback the marks are no-ops: neighbouring C ranges become one foreign call
        merged = []
        for it in prog:
            if it[0] == "mark":
                continue
            if it[0] == "c" and merged and merged[-1][0] == "c" and merged[-1][2] == it[1]:
                merged[-1] = ("c", merged[-1][1], it[2])
            else:
                merged.append(it)
        self._prog, self._prog_nomark, self._cnames = prog, merged, names
        return prog

    def run(self, prof: Optional[list] = None, on_mark=None) -> None:
        """Replay.  The launch list is walked in C (combat_plan_run: one foreign call per plan, or per range between
        two all-reduce marks / Python hand-offs), with the auxiliary-queue hand-off events created once per plan.
        `prof` (a list) selects the Python replay with per-launch HIP-event bracketing instead (`_run_py`).
        `on_mark(offset)` is called right after the call that completes the gradients above `offset`
        has been enqueued (data-parallel bucketed all-reduce overlapping the rest of the backward)."""
        if prof is not None or not Plan.compiled:
            return self._run_py(prof, on_mark)
        prog = self._program()
        if on_mark is None:
            prog = self._prog_nomark
        stream = torch.cuda.current_stream()
        st = stream.cuda_stream
        n_aux = self.aux_queues if (self.aux and not Plan.serial) else 0
        auxp = _aux_pointers(stream, n_aux) if n_aux else None
        cp = self._cplan
        for it in prog:
            kind = it[0]
            if kind == "c":
                rc = lib.combat_plan_run(cp, it[1], it[2], st, auxp, n_aux)
                if rc:
                    what = self._cnames[lib.combat_plan_failed_call(cp)]
                    kindname = {-1: "invalid shape/argument", -2: "HIP launch failed"}.get(rc, "status %d" % rc)
                    raise CombatHipError("%s/%s: %s" % (self.name, what, kindname))
            elif kind == "py":
                rc = it[1](*it[2], st)
                if rc:
                    raise CombatHipError("%s/%s: status %d" % (self.name, it[3], rc))
            else:
                if n_aux:   # the gradients above the mark include auxiliary-stream results
                    ops.check(lib.combat_plan_join(cp, st, auxp, n_aux), "combat_plan_join")
                on_mark(it[1])
        if n_aux:           # join: whoever runs after the plan sees every result
            ops.check(lib.combat_plan_join(cp, st, auxp, n_aux), "combat_plan_join")

    def __del__(self):
        try:
            if self._cplan:
                lib.combat_plan_destroy(self._cplan)
        except Exception:
            pass

    def _run_py(self, prof: Optional[list] = None, on_mark=None) -> None:
        """Replay from Python, one foreign call per launch (the instrumented form).  `prof` (a list) switches on per-launch HIP-event bracketing of the convolution and
        weight-gradient launches on the launch stream: it receives (plan/what, ConvArgs | WgradArgs,
        start_event, end_event).
        `on_mark(offset)` is called right after the call that completes the gradients above `offset`
        has been enqueued (data-parallel bucketed all-reduce overlapping the rest of the backward)."""
        stream = torch.cuda.current_stream()
        st = stream.cuda_stream
        auxs = []
        if self.aux and not Plan.serial:
            auxs = [_aux_stream(stream, k) for k in range(self.aux_queues)]
        used = set()
        awaited = set(self.after.values()) if auxs else set()
        done: Dict[int, torch.cuda.Event] = {}
        for ci, (cfunc, args, what) in enumerate(self.calls):
            if auxs and ci in self.after and self.after[ci] in done:
                (auxs[self.aux[ci]] if ci in self.aux else stream).wait_event(done[self.after[ci]])
            if prof is not None and cfunc is lib.combat_conv_gemm_pair:     # measured one kernel at a time
                rc = 0
                for sub, arg in zip(what.split("+"), args):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                    rc = rc or lib.combat_conv_gemm(arg, st)
                    e1.record(stream)
                    prof.append((self.name + "/" + sub, arg._obj, e0, e1))
            elif prof is not None and (cfunc is lib.combat_conv_gemm or cfunc is lib.combat_conv_wgrad or
                                       cfunc is lib.combat_conv_wgrad_reduce):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                rc = cfunc(*args, st)
                e1.record(stream)
                prof.append((self.name + "/" + what, args[0]._obj, e0, e1))
            elif auxs and ci in self.aux:
                # everything enqueued so far (the producers of this call's operands) happens-before it
                q = self.aux[ci]
                ev = torch.cuda.Event()
                ev.record(stream)
                auxs[q].wait_event(ev)
                rc = cfunc(*args, auxs[q].cuda_stream)
                used.add(q)
            else:
                rc = cfunc(*args, st)
            if rc:
                kind = {-1: "invalid shape/argument", -2: "HIP launch failed"}.get(rc, "status %d" % rc)
                raise CombatHipError("%s/%s: %s" % (self.name, what, kind))
            if ci in awaited:
                done[ci] = torch.cuda.Event()
                done[ci].record(auxs[self.aux[ci]] if ci in self.aux else stream)
            if on_mark is not None and ci in self.marks:
                for q in used:   # the gradients above the mark include auxiliary-stream results
                    ev = torch.cuda.Event()
                    ev.record(auxs[q])
                    stream.wait_event(ev)
                on_mark(self.marks[ci])
        for q in used:           # join: whoever runs after the plan sees every result
            ev = torch.cuda.Event()
            ev.record(auxs[q])
            stream.wait_event(ev)

    def __len__(self):
        return len(self.calls)


_AUX_STREAMS: Dict = {}
_AUX_POINTERS: Dict = {}


def _aux_pointers(parent: torch.cuda.Stream, n: int):
    """void*[n] of the parent stream's auxiliary streams, for combat_plan_run (built once per parent stream)."""
    key = (parent.device, parent.cuda_stream, n)
    arr = _AUX_POINTERS.get(key)
    if arr is None:
        arr = (ctypes.c_void_p * n)(*[_aux_stream(parent, k).cuda_stream for k in range(n)])
        _AUX_POINTERS[key] = arr
    return arr


def _aux_stream(parent: torch.cuda.Stream, k: int = 0) -> torch.cuda.Stream:
    """The auxiliary stream of the stream a plan runs on, for calls marked aux (weight gradients beside
    the input-gradient chain).  Per parent stream: two chains that run concurrently must not order each
    other through a shared auxiliary queue.  (Tried and dropped: a second auxiliary stream for the
    shortcut convolutions with a join in front of their consumer -- each cross-stream wait costs the
    critical path more than the 15 us convolution it hides: 5.63 -> 5.93 ms/step.)"""
    key = (parent.device, parent.cuda_stream, k)
    s = _AUX_STREAMS.get(key)
    if s is None:
        if os.environ.get("COMBAT_LOW_PRIO_SIDE", "0") == "1":
            from .step import low_priority_stream
            s = low_priority_stream(parent.device)
        else:
            s = torch.cuda.Stream(device=parent.device)
        _AUX_STREAMS[key] = s
    return s


class Slot:
    """Named device buffers of one forward(+backward) instance at a fixed batch size."""

    def __init__(self, device, n: int, hw: int):
        self.device, self.N, self.hw = device, n, hw
        self.bufs: Dict[str, torch.Tensor] = {}
        self.norm: Dict[str, "NormState"] = {}
        self.plans: Dict[str, Plan] = {}

    def half_view(self, lo: int, n: int, share: Tuple[str, ...]) -> "Slot":
        """A slot over images [lo, lo+n) of this one: the named buffers are views (batch-major
        layouts make any image range contiguous), everything else is allocated on demand.  Used to run
        the backward of only the triggered half of a merged eval forward."""
        s = Slot(self.device, n, self.hw)
        for k in share:
            if k in self.bufs:
                s.bufs[k] = self.bufs[k][lo:lo + n]
        s.feat_hw = getattr(self, "feat_hw", None)
        return s

    def buf(self, name: str, shape, dtype=bf16, zero: bool = False) -> torch.Tensor:
        t = self.bufs.get(name)
        if t is None:
            t = (torch.zeros if zero else torch.empty)(tuple(shape), dtype=dtype, device=self.device)
            self.bufs[name] = t
        assert tuple(t.shape) == tuple(shape) and t.dtype == dtype, (name, t.shape, shape)
        return t


class NormState:
    """Statistics of one normalised tensor: per-channel (groups=1) or per-(image, channel)."""

    def __init__(self, slot: Slot, key: str, groups: int, c: int):
        mk = lambda s: slot.buf("%s.%s" % (key, s), (groups, c), f32)
        self.groups, self.C = groups, c
        self.mean, self.rstd, self.scale, self.shift = mk("mean"), mk("rstd"), mk("scale"), mk("shift")
        self.pending = self.bpending = None    # statistics waiting for their fused finalize+apply launch
        self._slot, self._key = slot, key


class FlatParams:
    """All parameters of a module re-homed into one fp32 buffer, with same-layout gradient and
    momentum buffers.  Conv weights are stored [K][R][S][C] (the module keeps seeing them as OIHW
    tensors with channels_last strides), so the wgrad kernel writes gradients in place, SGD is one
    launch and the data-parallel all-reduce is one flat tensor (bucketed by offset ranges)."""

    ALIGN = 64  # floats

    def __init__(self, module: torch.nn.Module):
        params = list(module.named_parameters())
        dev = params[0][1].device
        self.offsets: Dict[str, Tuple[int, int, tuple]] = {}
        off = 0
        for name, p in params:
            self.offsets[name] = (off, p.numel(), tuple(p.shape))
            off += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.total = off
        self.flat = torch.zeros(off, dtype=f32, device=dev)
        self.grad = torch.zeros(off, dtype=f32, device=dev)
        self.mom = torch.zeros(off, dtype=f32, device=dev)
        with torch.no_grad():
            for name, p in params:
                o, n, shape = self.offsets[name]
                dst = self.flat[o:o + n]
                if p.dim() == 4:
                    k, c, r, s = shape
                    dst.copy_(p.detach().permute(0, 2, 3, 1).reshape(-1))
                    p.data = dst.view(k, r, s, c).permute(0, 3, 1, 2)
                else:
                    dst.copy_(p.detach().reshape(-1))
                    p.data = dst.view(shape)
        self.sgd_ptrs = torch.tensor([[self.flat.data_ptr(), self.grad.data_ptr(), self.mom.data_ptr()]],
                                     dtype=torch.int64, device=dev)
        self.sgd_sizes = torch.tensor([self.total], dtype=torch.int64, device=dev)

    def _slice(self, base: torch.Tensor, name: str) -> torch.Tensor:
        o, n, _ = self.offsets[name]
        return base[o:o + n]

    def grad_phys(self, name: str) -> torch.Tensor:
        """Gradient in the kernel's physical layout ([K][taps][C] for conv weights)."""
        o, n, shape = self.offsets[name]
        g = self.grad[o:o + n]
        if len(shape) == 4:
            k, c, r, s = shape
            return g.view(k, r * s, c)
        return g.view(shape)

    def logical(self, base: torch.Tensor, name: str) -> torch.Tensor:
        """`base` (grad / momentum) slice viewed with the parameter's logical shape."""
        o, n, shape = self.offsets[name]
        g = base[o:o + n]
        if len(shape) == 4:
            k, c, r, s = shape
            return g.view(k, r, s, c).permute(0, 3, 1, 2)
        return g.view(shape)

    zero_ev = None   # set while an early zeroing of `grad` is in flight on the auxiliary stream

    def zero_grad_behind(self) -> None:
        """Call right after the optimiser consumed the gradients: the buffer is zeroed on the auxiliary stream
        now, instead of by the first call of the next backward plan on the critical queue (9 us + a launch
        gap per network and step).  The plan's `zero_grad` entry then only waits for the event."""
        if Plan.serial:
            return
        main = torch.cuda.current_stream()
        aux = _aux_stream(main)
        ev = torch.cuda.Event()
        ev.record(main)
        aux.wait_event(ev)
        ops.check(lib.combat_memset_zero(self.grad.data_ptr(), self.total * 4, aux.cuda_stream), "zero_grad")
        self.zero_ev = torch.cuda.Event()
        self.zero_ev.record(aux)

    def sgd_step(self, lr: float, momentum: float = 0.9, weight_decay: float = 5e-4, grad_scale: float = 1.0):
        """torch.optim.SGD(nesterov=True) on the whole buffer: the momentum buffer starts at zero,
        so `buf = mu*buf + g` equals torch's first-step `buf = g`."""
        ops.sgd_nesterov(self.sgd_ptrs, self.sgd_sizes, 1, self.total, lr, momentum, weight_decay, grad_scale, False)


# --------------------------------------------------------------------------------------------
# recording helpers
# --------------------------------------------------------------------------------------------


def _zero_grad_call(fp, st):
    """First call of a backward plan: zero the flat gradient buffer -- or, if FlatParams.zero_grad_behind()
    already did that behind the previous optimiser step, just order this stream after it."""
    ev = fp.zero_ev
    if ev is not None:
        fp.zero_ev = None
        torch.cuda.current_stream().wait_event(ev)
        return 0
    return lib.combat_memset_zero(fp.grad.data_ptr(), fp.total * 4, st)


_SHORT_WORKGROUPS = False


class short_workgroups:
    """Plans recorded inside this context keep to one tile per workgroup: the weight-stationary persistent kernel
    (COMBAT_TILE_S128x64) is replaced by the ring kernel on the same tiles.  For passes that run BESIDE the critical
    queue: a persistent workgroup holds its CU (and 141 KB of its LDS) for 4-8 tiles, 17-34 us at 128-256 images, and
    the critical queue's launches wait for CUs that long."""

    def __enter__(self):
        global _SHORT_WORKGROUPS
        self.prev, _SHORT_WORKGROUPS = _SHORT_WORKGROUPS, True

    def __exit__(self, *exc):
        global _SHORT_WORKGROUPS
        _SHORT_WORKGROUPS = self.prev


def _tile_preference(a) -> None:
    if _SHORT_WORKGROUPS and a.tile == 0 and lib.combat_conv_pick_tile(ctypes.byref(a)) == TILE_S128x64:
        a.tile = TILE_D128x64


FUSE_SHORTCUT = os.environ.get("COMBAT_NO_FUSED_SHORTCUT", "0") != "1"
# COMBAT_REDUCE_BEHIND=1: a weight gradient's slab reduction rides in the NEXT weight-gradient launch of its queue
# (rec_wgrad, combat_wgrad_args.reduce_first): 23 of the 24 reduction launches of a step disappear and the 3x3 weight
# gradients become bit-reproducible (fixed summation order, no atomics).  OFF by default: measured 3.84 against 3.74
# ms/step -- the 128 workgroups of a weight gradient fold 147 KB of slabs each at a CU's ~30 GB/s before their first
# patch (+8 us per launch, +18 on the 512-channel layers), the round-3 reduction launch spreads the same bytes over the
# whole chip in 4-8 us (tools/wgrad_chain_bench.py).
REDUCE_BEHIND = os.environ.get("COMBAT_REDUCE_BEHIND", "0") == "1"
# COMBAT_FUSED_PROLOGUE=1: train-mode BatchNorm + ReLU applied by the CONSUMING convolution in LDS
# (PreActEngine._forward_train_body, conv3x3_dma_pro_kernel) instead of one combat_norm_act_fused launch per BatchNorm.
# OFF by default: built, bit-identical (tests/test_kernels_gpu.py::test_conv_lds_prologue_equals_norm_act_then_conv) and
# measured SLOWER -- 3.885 against 3.757 ms/step on one box (profiles/r04_a_*): every channel tile of the consumer
# repeats the transform (16x on a 512-channel layer with 32-channel tiles), the patch of a 4 x 4 map is three times
# its pixels, and the 64-channel layers lose the weight-stationary kernel; tools/pro_bench.py has the per-shape table
# (back to back: +3.8 us on 64 -> 64 @ 32 x 32, +9 on 256 -> 256 @ 8 x 8, +17 on 512 -> 512 @ 4 x 4, against
# 4-7 us for the stand-alone activation pass).  DESIGN.md section 5, round 4.
FUSED_PROLOGUE = os.environ.get("COMBAT_FUSED_PROLOGUE", "0") == "1"


def fuse_shortcut(dy, dx, pc: PackedConv, dy_sc, pc_sc: PackedConv):
    """The `shortcut=` argument of the block's first convolution's input gradient (dy -> dx through `pc`, 3x3 / stride
    2) if the 1x1 / stride-2 shortcut's input gradient (dy_sc through `pc_sc`) can ride along as a second reduction
    source (combat_conv_args.src2: one launch, no intermediate tensor), else None (two launches)."""
    if FUSE_SHORTCUT and not Plan.serial_shortcuts and ops.shortcut_fusable(dy, dx, pc, pc_sc) and tuple(dy_sc.shape) == tuple(dy.shape):
        return (dy_sc, pc_sc)
    return None


def rec_conv(plan: Plan, what: str, src, dst, pc: PackedConv, mode: int, **kw):
    a = ops.conv_args(src, dst, pc, mode, workspace=plan.workspace(src.device), **kw)
    _tile_preference(a)
    plan.hold(a)
    plan.add(what, lib.combat_conv_gemm, ctypes.byref(a))
    return a


def rec_wgrad(plan: Plan, what: str, src, dy, pc: PackedConv, dw, pro: Optional[Affine] = None, aux: bool = True):
    """aux=False: the LAST weight gradient of a backward pass stays on the plan's own stream, beside the
    second-to-last one on the auxiliary stream (nothing is left to hide it behind)."""
    a = ops.WgradArgs()
    a.N, a.H, a.W, a.C = src.shape
    _, a.P, a.Q, a.K = dy.shape
    a.R = a.S = pc.R
    a.stride, a.pad = pc.stride, pc.pad
    a.src, a.dy, a.dw, a.k_real, a.c_real = src.data_ptr(), dy.data_ptr(), dw.data_ptr(), pc.K, pc.c_real
    if pro is not None:
        a.pro_scale, a.pro_shift = _p(pro.scale), _p(pro.shift)
        a.pro_group_stride, a.pro_act, a.pro_slope = pro.group_stride, int(pro.act), pro.slope
    a.split = 0
    queue = plan.next_aux_queue() if aux else -1
    ws = _wgrad_workspace(src.device, queue)
    a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
    need = int(lib.combat_conv_wgrad_workspace_bytes(ctypes.byref(a)))
    prev = plan.chain.get(queue) if aux else None
    if aux and REDUCE_BEHIND and not Plan.defer_reduces and need > 0:      # (need > 0: the DMA-staged 3x3 kernel, several pixel ranges)
        # Reduce-behind (round 4): a launch that leaves partial-sum slabs does not reduce them; the NEXT weight gradient
        # of this queue folds them into this launch's dw before it starts on its own patches
        # (combat_wgrad_args.reduce_first), and only the last one of a chain gets a reduction launch (Plan.flush_reduces).
        # Slabs live in one of two private regions per queue, used alternately: the reader of region k % 2 (launch
        # k + 1) has completed before launch k + 2 writes it.
        if prev is not None:
            a.reduce_first = ctypes.addressof(prev[1])
            plan.pending_reduces = [q for q in plan.pending_reduces if q[1] is not prev[1]]
            plan.hold(prev[1])
            plan.chain.pop(queue)
        k = plan.chain_count.get(queue, 0)
        plan.chain_count[queue] = k + 1
        ws = _wgrad_slab_region(src.device, queue, k % 2, need)
        a.workspace, a.workspace_bytes, a.defer_reduce = ws.data_ptr(), ws.numel(), 1
        plan.chain[queue] = (what, a)
    elif aux and need > 0 and Plan.defer_reduces:
        # a launch that leaves partial-sum slabs: its reduction is deferred (Plan.flush_reduces), so the slabs need a
        # region of their own until then -- HBM is 288 GB, a backward plan's ~20 regions of 19 MB are not
        ws = torch.empty(need, dtype=torch.uint8, device=src.device)
        a.workspace, a.workspace_bytes, a.defer_reduce = ws.data_ptr(), need, 1
    plan.hold(a, src, dy, dw, pro, ws)
    plan.add(what, lib.combat_conv_wgrad, ctypes.byref(a), aux=aux)
    plan.wgrads.append((len(plan.calls) - 1, a))
    if a.defer_reduce:
        plan.pending_reduces.append((what + ".reduce", a, len(plan.calls) - 1))


def set_deterministic(on: bool) -> None:
    """Process-wide deterministic mode of the kernel library (include/combat_hip.h, combat_set_deterministic; default:
    COMBAT_DETERMINISTIC=1 in the environment): every parameter-gradient reduction and the augmentation adjoint in
    a fixed summation order, so two runs on the same inputs and draws give bit-identical parameters.  The launches
    read the switch when they run, so recorded plans follow it; weight-gradient calls always carry the per-queue
    workspace the ordered reductions need."""
    lib.combat_set_deterministic(1 if on else 0)


def deterministic() -> bool:
    return bool(lib.combat_get_deterministic())


_WGRAD_SLABS: Dict = {}


def _wgrad_slab_region(device, queue: int, which: int, need: int) -> torch.Tensor:
    """One of the two slab regions of a queue's reduce-behind chain (rec_wgrad)."""
    key = (device, int(queue), int(which))
    t = _WGRAD_SLABS.get(key)
    if t is None or t.numel() < need:
        t = torch.empty(max(need, 24 << 20), dtype=torch.uint8, device=device)
        _WGRAD_SLABS[key] = t
    return t


def balance_wgrads(plan: Plan, device) -> None:
    """Weight gradients run on the auxiliary stream beside the input-gradient chain -- but they are the longer of
    the two chains (20 launches of ~40 us against ~480 us of input gradients + norm backward in PreActResNet18's
    backward), so the plan ended with the main stream idle behind ~400 us of queued weight gradients.  The LAST
    `tail` weight gradients of the plan (the early layers) are therefore issued on the plan's own stream: at the
    end of the plan both queues run weight gradients side by side (each is sized for half the chip)."""
    tail = int(os.environ.get("COMBAT_WGRAD_MAIN_TAIL", WGRAD_MAIN_TAIL))
    if tail <= 0 or Plan.serial:
        return
    aux_calls = [(ci, a) for ci, a in plan.wgrads if ci in plan.aux]
    ws = _wgrad_workspace(device, -1)
    plan._prog = None
    for ci, a in aux_calls[-tail:]:
        plan.aux.pop(ci, None)
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
        plan.hold(ws)


WGRAD_MAIN_TAIL = 0


_WGRAD_WS: Dict = {}


def _wgrad_workspace(device, queue: int = 0) -> torch.Tensor:
    """Scratch for the weight-gradient partial sums (pixel ranges x tiles <= ~256 + tiles slabs of 147 KB; a
    launch that would need more falls back to fp32 atomics by itself).  One buffer per device AND per queue
    the launch is recorded for: launches of one queue run one after the other, but the last weight gradient of
    a backward plan (queue -1) runs on the plan's own stream while its predecessors may still be writing /
    reducing their slabs on an auxiliary stream (queue 0, 1, ...) -- sharing one base pointer between two queues let
    those launches overwrite each other's partial sums."""
    key = (device, int(queue))
    ws = _WGRAD_WS.get(key)
    if ws is None:
        ws = torch.empty(48 << 20, dtype=torch.uint8, device=device)
        _WGRAD_WS[key] = ws
    return ws


def rec_act(plan: Plan, what: str, src, dst, aff: Affine):
    """dst = lrelu(src * scale + shift) as a tensor (what a convolution prologue with `aff` computes)."""
    c = src.shape[-1]
    rows = src.numel() // c
    group_rows = (src.shape[1] * src.shape[2]) if aff.group_stride else 0
    assert aff.act and aff.group_stride in (0, c)
    plan.hold(src, dst, aff)
    plan.add(what, lib.combat_affine_act, src.data_ptr(), rows, c, _p(aff.scale), _p(aff.shift), group_rows,
             float(aff.slope), dst.data_ptr())


def _pow2_part(pq: int, cap: int = 32) -> int:
    g = 1
    while g * 2 <= cap and pq % (g * 2) == 0:
        g *= 2
    return g


class NetEngine:
    """Shared machinery: parameters, packed operands, scratch, slots."""

    _flush_at_marks = False

    @property
    def flush_at_marks(self) -> bool:
        """Set by a data-parallel step (see Plan.flush_at_marks).  Backward plans copy it when they are built and are
        cached per slot, and an engine is shared by every step object of its module: changing the value drops the
        cached backward plans, so that a plan built for a single-rank step is never replayed under all-reduce marks."""
        return self._flush_at_marks

    @flush_at_marks.setter
    def flush_at_marks(self, value: bool) -> None:
        value = bool(value)
        if value != self._flush_at_marks:
            for slot in getattr(self, "slots", {}).values():
                for k in [k for k in slot.plans if k.startswith("bwd")]:
                    del slot.plans[k]
        self._flush_at_marks = value

    def __init__(self, module: torch.nn.Module):
        self.module = module
        self.device = next(module.parameters()).device
        if self.device.type != "cuda":
            raise CombatHipError("combat_amd engines run on the GPU only (module is on %s); there is no CPU path"
                                 % self.device)
        self.fp = FlatParams(module)
        self.convs: List[PackedConv] = []
        self.slots: Dict[Tuple, Slot] = {}
        self._scratch = torch.empty(ops.norm_scratch_bytes(1, 512) // 4, dtype=f32, device=self.device)
        self.weights_dirty = True

    def mark_weights_dirty(self) -> None:
        self.weights_dirty = True

    def slot(self, name: str, n: int, hw: int) -> Slot:
        key = (name, n, hw)
        s = self.slots.get(key)
        if s is None:
            s = Slot(self.device, n, hw)
            self.slots[key] = s
        return s

    def _pc(self, weight, stride, pad, cin_pad, dup=False, need_dgrad=True, row_scale=None) -> PackedConv:
        pc = PackedConv(weight, stride, pad, cin_pad, dup_hilo=dup, need_dgrad=need_dgrad, row_scale=row_scale)
        self.convs.append(pc)
        return pc

    def refresh(self) -> None:
        """Re-derive bf16 operands (and folded eval-BN) after a parameter update: one launch packs every
        convolution of the network (device-side descriptor table, rebuilt if a weight tensor moved)."""
        if not self.weights_dirty:
            return
        self._pack_all()
        self._refresh_extra()
        self.weights_dirty = False

    def _pack_all(self) -> None:
        ptrs = tuple(pc.weight.data_ptr() for pc in self.convs)
        if getattr(self, "_pack_ptrs", None) != ptrs:
            masters = [pc.master() for pc in self.convs]
            if any(m.data_ptr() != q for m, q in zip(masters, ptrs)):   # a weight is not channels_last: no stable view
                self._pack_ptrs = None
                for pc in self.convs:
                    pc.pack()
                return
            from ._lib import PackDesc
            tab = (PackDesc * len(self.convs))()
            for d, pc, m in zip(tab, self.convs, masters):
                pc.fill_desc(d, m)
            self._pack_tab = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(self.device)
            self._pack_ptrs = ptrs
        ops.check(lib.combat_pack_weights_batch(self._pack_tab.data_ptr(), len(self.convs),
                                                torch.cuda.current_stream().cuda_stream), "combat_pack_weights_batch")

    def _refresh_extra(self) -> None:
        pass

    # ---- normalisation plumbing -----------------------------------------------------------
    FUSED_DIRECT_PX = 1024   # pixels per group the fused norm kernels reduce themselves (kFusedMaxDirect, norm.hip)

    def _conv_norm(self, plan: Plan, slot: Slot, key: str, src, dst, pc: PackedConv, *, groups: int,
                   gamma=None, beta=None, running=None, act_dst=None, slope: float = 0.0, defer: bool = False,
                   **conv_kw) -> NormState:
        """conv + statistics of its raw output + finalize.  groups == 1: batch statistics
        (optionally updating running stats); groups == N: instance statistics.
        act_dst: also materialise lrelu(norm(dst), slope) -- finalize and activation are then ONE launch
        (combat_norm_act_fused).  defer: leave the finalize to the consumer's `_norm_act` (the activation
        buffer belongs to the consuming layer)."""
        n, p, q, c = dst.shape
        pq, m = p * q, n * p * q
        a = ops.conv_args(src, dst, pc, 0, workspace=plan.workspace(src.device), **conv_kw)
        a.stats_kind = 1 | (BATCH_STATS_ROWS if groups == 1 else 0)    # (the row layout depends on it)
        _tile_preference(a)      # (after every field the tile choice reads -- the epilogue flavour -- is final)
        rows, rpi = ops.conv_stats_layout(a)
        fused = (groups == 1) or (rpi > 0)
        st = slot.norm.get(key)
        if st is None:
            st = NormState(slot, key, groups, c)
            slot.norm[key] = st
        part = None
        if fused:
            part = slot.buf(key + ".part", (rows, 2, c), f32)
            a.stats = part.data_ptr()
            rpg = rows if groups == 1 else rpi
        else:
            a.stats_kind = 0
        plan.hold(a, part)
        plan.add(key + ".conv", lib.combat_conv_gemm, ctypes.byref(a))
        one_launch = act_dst is not None or defer
        if not fused and not (one_launch and pq <= self.FUSED_DIRECT_PX):
            g = _pow2_part(pq)
            parts = m // g
            part = slot.buf(key + ".part", (parts, 2, c), f32)
            plan.add(key + ".stats", lib.combat_group_stats, dst.data_ptr(), parts, g, c, part.data_ptr())
            rpg = pq // g
        rm = rv = nbt = None
        if running is not None:
            rm, rv, nbt = running
        plan.hold(gamma, beta, rm, rv, nbt)
        if one_launch:
            st.pending = dict(x=dst, part=part, rpg=rpg if part is not None else 0, pxg=m // groups, key=key,
                              gamma=gamma, beta=beta, rm=rm, rv=rv, nbt=nbt)
            if act_dst is not None:
                self._norm_act(plan, st, act_dst, slope)
            return st
        plan.add(key + ".finalize", lib.combat_norm_finalize, part.data_ptr(), groups, rpg, c, float(m // groups),
                 1e-5, _p(gamma), _p(beta), st.mean.data_ptr(), st.rstd.data_ptr(), st.scale.data_ptr(),
                 st.shift.data_ptr(), _p(rm), _p(rv), 0.1, _p(nbt), self._scratch.data_ptr(),
                 self._scratch.numel() * 4)
        return st

    def _lds_prologue_ok(self, src, pc: PackedConv) -> bool:
        """Would a forward 3x3 convolution of `src` through `pc` with a per-channel BatchNorm + ReLU prologue and the
        activated side output take the DMA-staged kernel that applies the prologue in LDS (combat_conv_args.pro_act_dst)?"""
        if not FUSED_PROLOGUE or pc.R != 3 or pc.stride != 1:
            return False
        n, h, w, _ = src.shape
        probe = getattr(self, "_pro_probe", None)
        if probe is None:
            t = torch.zeros(512, dtype=f32, device=self.device)
            probe = self._pro_probe = Affine(t, t, 0, True, 0.0)
        a = ConvArgs()
        a.N, a.H, a.W, a.C = src.shape
        a.P, a.Q, a.K = h, w, pc.Kc
        a.R = a.S = 3
        a.stride, a.pad, a.mode = 1, pc.pad, 0
        a.src, a.dst, a.pro_act_dst = src.data_ptr(), src.data_ptr(), src.data_ptr()
        a.wpack, a.kpad, a.rows_pad = pc.wf.data_ptr(), pc.kpad_f, pc.rows_f
        a.pro_scale, a.pro_shift, a.pro_act = probe.scale.data_ptr(), probe.shift.data_ptr(), 1
        return lib.combat_conv_pick_tile(ctypes.byref(a)) in (TILE_D128x64, TILE_D128x32)

    def _norm_act(self, plan: Plan, st: NormState, act_dst, slope: float):
        """The deferred finalize of `st` + its activation tensor, one launch."""
        q = st.pending
        st.pending = None
        x = q["x"]
        plan.hold(x, act_dst, q["part"])
        plan.add(q["key"] + ".finact", lib.combat_norm_act_fused, x.data_ptr(), _p(q["part"]), st.groups, q["rpg"],
                 q["pxg"], st.C, 1e-5, float(slope), _p(q["gamma"]), _p(q["beta"]), st.mean.data_ptr(),
                 st.rstd.data_ptr(), st.scale.data_ptr(), st.shift.data_ptr(), _p(q["rm"]), _p(q["rv"]), 0.1,
                 _p(q["nbt"]), self._scratch.data_ptr(), self._scratch.numel() * 4, act_dst.data_ptr())

    def _dgrad_norm(self, plan: Plan, slot: Slot, key: str, dy, dz, pc: PackedConv, x_pre, st: NormState, *,
                    group_stride: int, slope: float, gamma=None, dgamma=None, dbeta=None, add_pre=None, shortcut=None):
        """dgrad whose epilogue applies the activation mask of the conv input (recomputed from the
        saved pre-norm tensor x_pre) and emits the two norm-backward reductions.  The caller's
        `_bwd_apply` turns them into coefficients and applies those in one launch."""
        n, p, q, c = dz.shape
        pq, m = p * q, n * p * q
        groups = st.groups
        mask = Affine(st.scale, st.shift, group_stride, True, slope)
        a = ops.conv_args(dy, dz, pc, 1, add_pre=add_pre, mask_x=x_pre, mask=mask, workspace=plan.workspace(dy.device),
                          shortcut=shortcut)
        a.stats_kind = 2 | (BATCH_STATS_ROWS if groups == 1 else 0)
        a.xh_mean, a.xh_rstd = st.mean.data_ptr(), st.rstd.data_ptr()
        _tile_preference(a)      # (after stats_kind / xh_*: the tile query and the launch must see the same flavour)
        rows, rpi = ops.conv_stats_layout(a)
        fused = (groups == 1) or (rpi > 0)
        part, rpg = None, 0
        if fused:
            part = slot.buf(key + ".bpart", (rows, 2, c), f32)
            a.stats = part.data_ptr()
            rpg = rows if groups == 1 else rpi
        else:
            a.stats_kind, a.xh_mean, a.xh_rstd = 0, None, None
        plan.hold(a, part)
        plan.add(key + ".dgrad", lib.combat_conv_gemm, ctypes.byref(a))
        if not fused and pq > self.FUSED_DIRECT_PX:
            part, rpg = self._stats_bwd(plan, slot, key, dz, x_pre, st)
        st.bpending = dict(part=part, rpg=rpg, gamma=gamma, dgamma=dgamma, dbeta=dbeta)

    def _stats_bwd(self, plan: Plan, slot: Slot, key: str, dz, x_pre, st: NormState):
        n, p, q, c = dz.shape
        pq, m = p * q, n * p * q
        g = _pow2_part(pq if st.groups > 1 else m)
        parts = m // g
        part = slot.buf(key + ".bpart", (parts, 2, c), f32)
        ppi = (pq // g) if st.groups > 1 else 0
        plan.add(key + ".bstats", lib.combat_group_stats_bwd, dz.data_ptr(), x_pre.data_ptr(), parts, g, c, ppi,
                 st.mean.data_ptr(), st.rstd.data_ptr(), part.data_ptr())
        return part, (pq // g if st.groups > 1 else parts)

    def _bwd_apply(self, plan: Plan, slot: Slot, key: str, dz, x_pre, dx, st: NormState, add=None):
        """Norm backward: (sum dz, sum dz*xhat) -> coefficients -> dx = ca*dz + cb*x + cc (+ add), one launch.
        The sums come from the rows `_dgrad_norm` left behind, or (no such rows) from dz / x_pre directly."""
        n, p, q, c = dz.shape
        pq, m = p * q, n * p * q
        pend = getattr(st, "bpending", None) or dict(part=None, rpg=0, gamma=None, dgamma=None, dbeta=None)
        st.bpending = None
        part, rpg = pend["part"], pend["rpg"]
        pxg = m // st.groups
        if part is None and (st.groups == 1 or pq > self.FUSED_DIRECT_PX):
            part, rpg = self._stats_bwd(plan, slot, key, dz, x_pre, st)
        plan.hold(dz, x_pre, dx, add, part, pend["gamma"], pend["dgamma"], pend["dbeta"])
        plan.add(key + ".bfused", lib.combat_norm_bwd_fused, dz.data_ptr(), x_pre.data_ptr(), _p(add), _p(part),
                 st.groups, rpg, pxg, c, _p(pend["gamma"]), st.mean.data_ptr(), st.rstd.data_ptr(),
                 _p(pend["dgamma"]), _p(pend["dbeta"]), self._scratch.data_ptr(), self._scratch.numel() * 4,
                 dx.data_ptr())


# --------------------------------------------------------------------------------------------
# classifiers
# --------------------------------------------------------------------------------------------


class _BN:
    def __init__(self, eng: "NetEngine", mod: torch.nn.BatchNorm2d, prefix: str):
        self.prefix, self.C = prefix, mod.num_features
        self.gamma, self.beta = mod.weight.data, mod.bias.data
        self.rm, self.rv, self.nbt = mod.running_mean, mod.running_var, mod.num_batches_tracked
        self.escale = torch.empty(self.C, dtype=f32, device=eng.device)
        self.eshift = torch.empty(self.C, dtype=f32, device=eng.device)

    def fold(self):
        ops.bn_eval_fold(self.gamma, self.beta, self.rm, self.rv, self.escale, self.eshift)

    def eval_affine(self) -> Affine:
        return Affine(self.escale, self.eshift, 0, True, 0.0)


class _Blk:
    pass


class PreActEngine(NetEngine):
    """PreActResNet18 (classifier_models/preact_resnet.py): stem conv, 8 pre-activation blocks,
    avg-pool + linear + cross-entropy head."""

    def __init__(self, module):
        super().__init__(module)
        m = module
        self.classes = m.linear.out_features
        self.stem = self._pc(m.conv1.weight.data, 1, 1, 8, dup=True)
        self.blocks: List[_Blk] = []
        self.bns: List[_BN] = []
        for li in range(1, 5):
            for bi, blk in enumerate(getattr(m, "layer%d" % li)):
                e = _Blk()
                e.prefix = "layer%d.%d." % (li, bi)
                e.stride, e.cin, e.planes = blk.stride, blk.conv1.in_channels, blk.conv1.out_channels
                e.bn1, e.bn2 = _BN(self, blk.bn1, e.prefix + "bn1"), _BN(self, blk.bn2, e.prefix + "bn2")
                e.conv1 = self._pc(blk.conv1.weight.data, blk.stride, 1, e.cin)
                e.conv2 = self._pc(blk.conv2.weight.data, 1, 1, e.planes)
                e.sc = self._pc(blk.shortcut[0].weight.data, blk.stride, 0, e.cin) if hasattr(blk, "shortcut") else None
                self.blocks.append(e)
                self.bns += [e.bn1, e.bn2]
        self.lin_w, self.lin_b = m.linear.weight.data, m.linear.bias.data

    def _refresh_extra(self):
        self.fold_bn()

    def fold_bn(self):
        """Parameters or running stats changed: refresh every folded eval scale/shift (one launch)."""
        ptrs = tuple(t.data_ptr() for bn in self.bns for t in (bn.gamma, bn.beta, bn.rm, bn.rv))
        if getattr(self, "_bn_ptrs", None) != ptrs:
            from ._lib import BnDesc
            tab = (BnDesc * len(self.bns))()
            for d, bn in zip(tab, self.bns):
                d.gamma, d.beta, d.running_mean, d.running_var = (t.data_ptr() for t in (bn.gamma, bn.beta, bn.rm, bn.rv))
                d.scale, d.shift, d.C = bn.escale.data_ptr(), bn.eshift.data_ptr(), bn.C
            self._bn_tab = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(self.device)
            self._bn_ptrs = ptrs
        ops.check(lib.combat_bn_eval_fold_batch(self._bn_tab.data_ptr(), len(self.bns), 1e-5,
                                                torch.cuda.current_stream().cuda_stream), "combat_bn_eval_fold_batch")

    # ---- buffers of the head
    def head_bufs(self, slot: Slot):
        n = slot.N
        return dict(
            logits=slot.buf("logits", (n, self.classes), f32), dlogits=slot.buf("dlogits", (n, self.classes), f32),
            pooled=slot.buf("pooled", (n, self.lin_w.shape[1]), f32), loss=slot.buf("loss", (1,), f32, zero=True),
            correct=slot.buf("correct", (2,), torch.int32, zero=True),
            targets=slot.buf("targets", (n,), torch.int64, zero=True),
            targets2=slot.buf("targets2", (n,), torch.int64, zero=True))

    def input(self, slot: Slot) -> torch.Tensor:
        return slot.buf("x", (slot.N, slot.hw, slot.hw, 8))

    FWD_SHARED = ("stem", "logits", "dlogits", "pooled", "targets", "targets2") + tuple(
        "b%d.%s" % (b, s) for b in range(8) for s in ("y1", "out", "sc", "a1", "act")) + ("stem.act",)

    def forward_plan(self, slot: Slot, train: bool, loss_weight: float = 1.0, with_targets2: bool = False,
                     split_head: bool = False, head_bwd: bool = False) -> Plan:
        """split_head (eval only): the batch is two independent halves [metric-only images ; images whose
        loss is differentiated]; the head runs once per half with separate loss / counter cells
        (first half -> correct[0] of slot 'loss0/correct0' cells, second half -> the usual ones).
        head_bwd: the head launch also produces dlogits and the gradient w.r.t. the feature map ('g.feat':
        combat_head_fwd_bwd), for a pass whose backward plan (built with head_done=True) follows at once."""
        key = "fwd.%s.%g.%d.%d%s" % ("train" if train else "eval", loss_weight, with_targets2, split_head, ".hb" if head_bwd else "")
        if key in slot.plans:
            return slot.plans[key]
        P = Plan("preact." + key)
        n, hw = slot.N, slot.hw
        x = self.input(slot)
        if not train:
            cur, chw = self._forward_eval_body(P, slot, x)
        else:
            cur, chw = self._forward_train_body(P, slot, x)
        h = self.head_bufs(slot)
        P.hold(h)
        if head_bwd:
            assert not split_head and type(self) is PreActEngine
            d_feat = slot.buf("g.feat", cur.shape)
            P.hold(d_feat)
            P.add("head+bwd", lib.combat_head_fwd_bwd, cur.data_ptr(), n, chw, cur.shape[-1], self.lin_w.data_ptr(),
                  self.lin_b.data_ptr(), self.classes, h["targets"].data_ptr(), loss_weight, h["pooled"].data_ptr(),
                  h["logits"].data_ptr(), h["loss"].data_ptr(), h["correct"].data_ptr(),
                  h["targets2"].data_ptr() if with_targets2 else None,
                  h["correct"][1:].data_ptr() if with_targets2 else None, h["dlogits"].data_ptr(), d_feat.data_ptr())
        elif not split_head:
            P.add("head", lib.combat_head_fwd, cur.data_ptr(), n, chw, cur.shape[-1], self.lin_w.data_ptr(),
                  self.lin_b.data_ptr(), self.classes, h["targets"].data_ptr(), loss_weight, h["pooled"].data_ptr(),
                  h["logits"].data_ptr(), h["loss"].data_ptr(), h["correct"].data_ptr(),
                  h["targets2"].data_ptr() if with_targets2 else None,
                  h["correct"][1:].data_ptr() if with_targets2 else None)
        else:
            assert not train and n % 2 == 0
            m = n // 2
            loss0 = slot.buf("loss0", (1,), f32, zero=True)
            correct0 = slot.buf("correct0", (2,), torch.int32, zero=True)
            views = [cur[:m], cur[m:], h["targets"][:m], h["targets"][m:], h["pooled"][:m], h["pooled"][m:],
                     h["logits"][:m], h["logits"][m:], h["targets2"][m:]]
            P.hold(views, loss0, correct0)
            P.add("head.metric", lib.combat_head_fwd, views[0].data_ptr(), m, chw, cur.shape[-1], self.lin_w.data_ptr(),
                  self.lin_b.data_ptr(), self.classes, views[2].data_ptr(), 1.0, views[4].data_ptr(),
                  views[6].data_ptr(), loss0.data_ptr(), correct0.data_ptr(), None, None)
            P.add("head.loss", lib.combat_head_fwd, views[1].data_ptr(), m, chw, cur.shape[-1], self.lin_w.data_ptr(),
                  self.lin_b.data_ptr(), self.classes, views[3].data_ptr(), loss_weight, views[5].data_ptr(),
                  views[7].data_ptr(), h["loss"].data_ptr(), h["correct"].data_ptr(),
                  views[8].data_ptr() if with_targets2 else None,
                  h["correct"][1:].data_ptr() if with_targets2 else None)
        slot.plans[key] = P
        slot.feat_hw = chw
        return P

    def _forward_train_body(self, P: Plan, slot: Slot, x):
        """Train mode: BatchNorm uses batch statistics, produced by the epilogue of the convolution that
        writes the normalised tensor.  relu(bn(.)) is then materialised once per BatchNorm ('b%d.a0' =
        block input through bn1, 'b%d.a1t' = conv1 output through bn2): the forward convolutions and
        the weight-gradient passes that read it need no prologue and take their operands by LDS-DMA.

        Round 4 (FUSED_PROLOGUE): where the consumer is a 3x3 / stride-1 convolution, relu(bn(.)) is not a launch of
        its own any more: the producer's statistics are finalised (per-channel scale / shift) and the CONSUMING
        convolution applies them in LDS after its DMA'd patch has landed (combat_conv_args.pro_* on the DMA-staged
        kernel), writing the activated tensor out on the side for its weight gradient (pro_act_dst).  The blocks with
        a stride-2 first convolution + 1x1 shortcut still get their input materialised by combat_norm_act_fused."""
        n, hw = slot.N, slot.hw
        cur = slot.buf("stem", (n, hw, hw, 64))

        def fuse(blk, src):    # does blk.conv1 read its raw input `src` with the in-LDS prologue?
            return blk.sc is None and blk.stride == 1 and self._lds_prologue_ok(src, blk.conv1)

        first = self.blocks[0].bn1
        a0 = slot.buf("b0.a0", cur.shape)
        st = self._conv_norm(P, slot, first.prefix, x, cur, self.stem, groups=1, gamma=first.gamma, beta=first.beta,
                             running=(first.rm, first.rv, first.nbt), act_dst=None if fuse(self.blocks[0], cur) else a0)
        chw = hw
        for b, blk in enumerate(self.blocks):
            ohw = chw // blk.stride
            i_sc = len(P.calls)
            if blk.sc is not None:
                resid = slot.buf("b%d.sc" % b, (n, ohw, ohw, blk.planes))
                rec_conv(P, "b%d.sc" % b, a0, resid, blk.sc, 0)
            else:
                resid = cur
            y1 = slot.buf("b%d.y1" % b, (n, ohw, ohw, blk.planes))
            out = slot.buf("b%d.out" % b, (n, ohw, ohw, blk.planes))
            nxt = self.blocks[b + 1].bn1 if b + 1 < len(self.blocks) else None
            a1 = slot.buf("b%d.a1t" % b, y1.shape)
            fuse2 = self._lds_prologue_ok(y1, blk.conv2)
            bn2 = dict(groups=1, gamma=blk.bn2.gamma, beta=blk.bn2.beta, running=(blk.bn2.rm, blk.bn2.rv, blk.bn2.nbt),
                       act_dst=None if fuse2 else a1)
            if fuse(blk, cur):     # conv1 normalises + activates its raw input in LDS and leaves a0 for its weight gradient
                st2 = self._conv_norm(P, slot, blk.bn2.prefix, cur, y1, blk.conv1, **bn2,
                                      pro=Affine(st.scale, st.shift, 0, True, 0.0), pro_act_dst=a0)
            else:
                st2 = self._conv_norm(P, slot, blk.bn2.prefix, a0, y1, blk.conv1, **bn2)
            if blk.sc is not None:      # shortcut + first convolution: one launch (both read a0)
                P.merge_convs(i_sc)
            # conv2 is 3x3 / stride 1 in every block: relu(bn2(y1)) in LDS, a1 on the side
            c2 = dict(add_post=resid)
            src2 = a1
            if fuse2:
                c2.update(pro=Affine(st2.scale, st2.shift, 0, True, 0.0), pro_act_dst=a1)
                src2 = y1
            if nxt is not None:
                a0 = slot.buf("b%d.a0" % (b + 1), out.shape)
                st = self._conv_norm(P, slot, nxt.prefix, src2, out, blk.conv2, groups=1, gamma=nxt.gamma, beta=nxt.beta,
                                     running=(nxt.rm, nxt.rv, nxt.nbt),
                                     act_dst=None if fuse(self.blocks[b + 1], out) else a0, **c2)
            else:
                rec_conv(P, "b%d.c2" % b, src2, out, blk.conv2, 0, **c2)
            cur, chw = out, ohw
        return cur, chw

    def _forward_eval_body(self, P: Plan, slot: Slot, x):
        """Eval mode: every BatchNorm is a fixed per-channel affine known before its producer runs, so the
        producing convolution writes relu(bn(.)) itself (second epilogue output) and no convolution
        needs a prologue -- the 3x3 / stride-1 layers then run on the DMA-staged kernel.  Tensors:
        'stem.act' / 'b%d.act' = relu(bn1_next(block output)), 'b%d.a1' = relu(bn2(conv1 output));
        the raw block output ('stem' / 'b%d.out') is kept only where an identity shortcut or the head
        reads it."""
        n, hw = slot.N, slot.hw
        blocks = self.blocks
        raw = slot.buf("stem", (n, hw, hw, 64)) if blocks[0].sc is None else None
        act = slot.buf("stem.act", (n, hw, hw, 64))
        rec_conv(P, "stem", x, raw, self.stem, 0, act_dst=act, act=blocks[0].bn1.eval_affine())
        chw = hw
        for b, blk in enumerate(blocks):
            ohw = chw // blk.stride
            shape = (n, ohw, ohw, blk.planes)
            if blk.sc is not None:
                resid = slot.buf("b%d.sc" % b, shape)
                rec_conv(P, "b%d.sc" % b, act, resid, blk.sc, 0)
            else:
                resid = raw
            a1 = slot.buf("b%d.a1" % b, shape)
            rec_conv(P, "b%d.c1" % b, act, None, blk.conv1, 0, act_dst=a1, act=blk.bn2.eval_affine())
            if blk.sc is not None:
                P.merge_convs(len(P.calls) - 2)
            nxt = blocks[b + 1] if b + 1 < len(blocks) else None
            keep_raw = nxt is None or nxt.sc is None
            out = slot.buf("b%d.out" % b, shape) if keep_raw else None
            if nxt is not None:
                nact = slot.buf("b%d.act" % b, shape)
                rec_conv(P, "b%d.c2" % b, a1, out, blk.conv2, 0, add_post=resid, act_dst=nact, act=nxt.bn1.eval_affine())
            else:
                nact = None
                rec_conv(P, "b%d.c2" % b, a1, out, blk.conv2, 0, add_post=resid)
            raw, act, chw = out, nact, ohw
        return raw, chw

    def _block_io(self, slot: Slot, b: int):
        xin = slot.bufs["stem"] if b == 0 else slot.bufs["b%d.out" % (b - 1)]
        return xin, slot.bufs["b%d.y1" % b], slot.bufs["b%d.out" % b]

    def backward_train_plan(self, slot: Slot, loss_weight: float = 1.0, head_done: bool = False) -> Plan:
        """Backward of a train-mode forward: all parameter gradients (into fp.grad, which the plan
        zeroes first); no input gradient (Phase C never needs it, train_generator.py:220).
        head_done: the forward plan was built with head_bwd=True -- dlogits and 'g.feat' exist; the linear layer's
        weight gradient is then an auxiliary-queue launch behind the first input-gradient launch."""
        key = "bwd.train.%g%s" % (loss_weight, ".hd" if head_done else "")
        if key in slot.plans:
            return slot.plans[key]
        P = Plan("preact." + key)
        P.flush_at_marks = self.flush_at_marks
        fp, n = self.fp, slot.N
        h = self.head_bufs(slot)
        P.add("zero_grad", _zero_grad_call, fp)
        feat = slot.bufs["b%d.out" % (len(self.blocks) - 1)]
        d_out = slot.buf("g.feat", feat.shape)
        if not head_done:
            P.add("head_bwd", lib.combat_head_bwd, h["pooled"].data_ptr(), n, slot.feat_hw, feat.shape[-1],
                  self.lin_w.data_ptr(), self.classes, h["logits"].data_ptr(), h["targets"].data_ptr(), loss_weight,
                  h["dlogits"].data_ptr(), d_out.data_ptr(), fp.grad_phys("linear.weight").data_ptr(),
                  fp.grad_phys("linear.bias").data_ptr())
        head_w_pending = head_done
        for b in reversed(range(len(self.blocks))):
            blk = self.blocks[b]
            xin, y1, _ = self._block_io(slot, b)
            st1, st2 = slot.norm[blk.bn1.prefix], slot.norm[blk.bn2.prefix]
            a0, a1 = slot.bufs["b%d.a0" % b], slot.bufs["b%d.a1t" % b]   # relu(bn1(xin)), relu(bn2(y1)) of the forward
            pre = blk.prefix
            if not head_w_pending:
                rec_wgrad(P, "b%d.c2.wgrad" % b, a1, d_out, blk.conv2, fp.grad_phys(pre + "conv2.weight"))
            dz2 = slot.buf("g.b%d.dz2" % b, y1.shape)
            self._dgrad_norm(P, slot, "g." + blk.bn2.prefix, d_out, dz2, blk.conv2, y1, st2, group_stride=0,
                             slope=0.0, gamma=blk.bn2.gamma, dgamma=fp.grad_phys(pre + "bn2.weight"),
                             dbeta=fp.grad_phys(pre + "bn2.bias"))
            if head_w_pending:    # (auxiliary calls behind the plan's first kernel launch: it carries their hand-off event)
                head_w_pending = False
                P.hold(h)
                P.add("head_bwd.w", lib.combat_head_bwd_weights, h["dlogits"].data_ptr(), h["pooled"].data_ptr(), n, slot.feat_hw,
                      feat.shape[-1], self.classes, fp.grad_phys("linear.weight").data_ptr(), fp.grad_phys("linear.bias").data_ptr(),
                      aux=True)
                rec_wgrad(P, "b%d.c2.wgrad" % b, a1, d_out, blk.conv2, fp.grad_phys(pre + "conv2.weight"))
            dy1 = slot.buf("g.b%d.dy1" % b, y1.shape)
            self._bwd_apply(P, slot, "g." + blk.bn2.prefix, dz2, y1, dy1, st2)
            tsc, fused = None, None
            dz1 = slot.buf("g.b%d.dz1" % b, xin.shape)
            if blk.sc is not None:
                rec_wgrad(P, "b%d.sc.wgrad" % b, a0, d_out, blk.sc, fp.grad_phys(pre + "shortcut.0.weight"))
                fused = fuse_shortcut(dy1, dz1, blk.conv1, d_out, blk.sc)    # the shortcut's input gradient rides along below
                if fused is None:
                    tsc = slot.buf("g.b%d.tsc" % b, xin.shape)
                    rec_conv(P, "b%d.sc.dgrad" % b, d_out, tsc, blk.sc, 1)
            rec_wgrad(P, "b%d.c1.wgrad" % b, a0, dy1, blk.conv1, fp.grad_phys(pre + "conv1.weight"))
            self._dgrad_norm(P, slot, "g." + blk.bn1.prefix, dy1, dz1, blk.conv1, xin, st1, group_stride=0,
                             slope=0.0, gamma=blk.bn1.gamma, dgamma=fp.grad_phys(pre + "bn1.weight"),
                             dbeta=fp.grad_phys(pre + "bn1.bias"), add_pre=tsc, shortcut=fused)
            dxin = slot.buf("g.b%d.dx" % b, xin.shape)
            self._bwd_apply(P, slot, "g." + blk.bn1.prefix, dz1, xin, dxin, st1,
                            add=None if blk.sc is not None else d_out)
            d_out = dxin
            if b in (6, 4, 2):   # layer4 / layer3 / layer2 complete: their gradient range can travel
                P.mark(fp.offsets[pre + "bn1.weight"][0])
        rec_wgrad(P, "stem.wgrad", self.input(slot), d_out, self.stem, fp.grad_phys("conv1.weight"), aux=False)
        P.mark(0)
        balance_wgrads(P, self.device)
        slot.plans[key] = P
        return P

    def backward_eval_plan(self, slot: Slot, loss_weight: float, head_done: bool = False) -> Plan:
        """Backward of an eval-mode forward w.r.t. the input image only (Phase G: the classifier
        and clean-model weight gradients are never consumed, train_generator.py:179,254).
        Result: slot buffer 'g.img' (bf16 NHWC c8, channels 0..2).  head_done: as backward_train_plan."""
        key = "bwd.eval.%g%s" % (loss_weight, ".hd" if head_done else "")
        if key in slot.plans:
            return slot.plans[key]
        P = Plan("preact." + key)
        n = slot.N
        h = self.head_bufs(slot)
        feat = slot.bufs["b%d.out" % (len(self.blocks) - 1)]
        d_out = slot.buf("g.feat", feat.shape)
        if not head_done:
            P.add("head_bwd", lib.combat_head_bwd, None, n, slot.feat_hw, feat.shape[-1], self.lin_w.data_ptr(),
                  self.classes, h["logits"].data_ptr(), h["targets"].data_ptr(), loss_weight, h["dlogits"].data_ptr(),
                  d_out.data_ptr(), None, None)
        for b in reversed(range(len(self.blocks))):
            blk = self.blocks[b]
            # activated tensors of the eval forward: the ReLU mask is (act > 0), the BatchNorm scale multiplies
            xact = slot.bufs["stem.act"] if b == 0 else slot.bufs["b%d.act" % (b - 1)]
            a1 = slot.bufs["b%d.a1" % b]
            dy1 = slot.buf("g.b%d.dy1" % b, a1.shape)
            rec_conv(P, "b%d.c2.dgrad" % b, d_out, dy1, blk.conv2, 1, mask_x=a1, mask=blk.bn2.eval_affine(),
                     mask_mul_scale=True, mask_activated=True)
            tsc, fused = None, None
            dxin = slot.buf("g.b%d.dx" % b, xact.shape)
            if blk.sc is not None:
                fused = fuse_shortcut(dy1, dxin, blk.conv1, d_out, blk.sc)
                if fused is None:
                    tsc = slot.buf("g.b%d.tsc" % b, xact.shape)
                    rec_conv(P, "b%d.sc.dgrad" % b, d_out, tsc, blk.sc, 1)
            rec_conv(P, "b%d.c1.dgrad" % b, dy1, dxin, blk.conv1, 1, shortcut=fused, add_pre=tsc, mask_x=xact,
                     mask=blk.bn1.eval_affine(), mask_mul_scale=True, mask_activated=True,
                     add_post=None if blk.sc is not None else d_out)
            d_out = dxin
        gimg = slot.buf("g.img", (n, slot.hw, slot.hw, 8))
        rec_conv(P, "stem.dgrad", d_out, gimg, self.stem, 1)
        slot.plans[key] = P
        return P


class ResNetEngine(PreActEngine):
    """ResNet18 (classifier_models/resnet.py:15-37, 68-106; the CelebA classifier): post-activation
    BasicBlocks  out = relu(bn2(conv2(relu(bn1(conv1(x))))) + shortcut(x)),  shortcut = conv1x1 + BatchNorm
    where the shape changes, stem conv + BatchNorm + ReLU, avg_pool2d(4) + Linear.

    A BatchNorm here FOLLOWS its convolution, so
      * eval mode: bn1 / the stem's norm are the producing convolution's activated-output epilogue, as in
        the pre-activation net; bn2 and the shortcut's norm sit in front of the residual sum and are folded
        into the packed weights (row_scale = gamma * rstd, the shift travels as the bias): every block is
        three launches and no tensor is stored raw;
      * train mode: batch statistics from the conv epilogues; bn2 + residual (+ the shortcut's own,
        separately finalised norm) + ReLU are one fused launch (combat_norm_add_act_fused).
    Tensors: 'stem.a' = relu(bn(conv(x))), 'b%d.a1' = relu(bn1(conv1(.))), 'b%d.out' = the block output
    (activated); train only: raw 'stem', 'b%d.y1', 'b%d.y2', 'b%d.ys' for the norm backward passes."""

    def __init__(self, module):
        NetEngine.__init__(self, module)
        m = module
        self.classes = m.linear.out_features
        self.stem = self._pc(m.conv1.weight.data, 1, 1, 8, dup=True)
        self.bn0 = _BN(self, m.bn1, "bn1")
        self.blocks: List[_Blk] = []
        self.bns: List[_BN] = [self.bn0]
        for li in range(1, 5):
            for bi, blk in enumerate(getattr(m, "layer%d" % li)):
                e = _Blk()
                e.prefix = "layer%d.%d." % (li, bi)
                e.stride, e.cin, e.planes = blk.stride, blk.conv1.in_channels, blk.conv1.out_channels
                e.bn1, e.bn2 = _BN(self, blk.bn1, e.prefix + "bn1"), _BN(self, blk.bn2, e.prefix + "bn2")
                e.conv1 = self._pc(blk.conv1.weight.data, blk.stride, 1, e.cin)
                e.conv2 = self._pc(blk.conv2.weight.data, 1, 1, e.planes)
                e.conv2e = self._pc(blk.conv2.weight.data, 1, 1, e.planes, row_scale=e.bn2.escale)
                e.sc = e.sce = e.bns = None
                if len(blk.shortcut):      # (an empty nn.Sequential for identity shortcuts, resnet.py:24)
                    e.bns = _BN(self, blk.shortcut[1], e.prefix + "shortcut.1")
                    e.sc = self._pc(blk.shortcut[0].weight.data, blk.stride, 0, e.cin)
                    e.sce = self._pc(blk.shortcut[0].weight.data, blk.stride, 0, e.cin, row_scale=e.bns.escale)
                    self.bns.append(e.bns)
                self.blocks.append(e)
                self.bns += [e.bn1, e.bn2]
        self.lin_w, self.lin_b = m.linear.weight.data, m.linear.bias.data
        self.ones = torch.ones(512, dtype=f32, device=self.device)
        self.zeros = torch.zeros(512, dtype=f32, device=self.device)

    def refresh(self) -> None:
        """The folded eval operands read the eval scale of their BatchNorm: fold first, then pack."""
        if not self.weights_dirty:
            return
        self.fold_bn()
        self._pack_all()
        self.weights_dirty = False

    FWD_SHARED = ("stem.a", "logits", "dlogits", "pooled", "targets", "targets2") + tuple(
        "b%d.%s" % (b, s) for b in range(8) for s in ("a1", "out", "scv"))

    def _block_in(self, slot: Slot, b: int):
        return slot.bufs["stem.a"] if b == 0 else slot.bufs["b%d.out" % (b - 1)]

    # ---- forward
    def _forward_eval_body(self, P: Plan, slot: Slot, x):
        n, hw = slot.N, slot.hw
        cur = slot.buf("stem.a", (n, hw, hw, 64))
        rec_conv(P, "stem", x, None, self.stem, 0, act_dst=cur, act=self.bn0.eval_affine())
        relu = Affine(self.ones, self.zeros, 0, True, 0.0)
        chw = hw
        for b, blk in enumerate(self.blocks):
            ohw = chw // blk.stride
            shape = (n, ohw, ohw, blk.planes)
            a1 = slot.buf("b%d.a1" % b, shape)
            rec_conv(P, "b%d.c1" % b, cur, None, blk.conv1, 0, act_dst=a1, act=blk.bn1.eval_affine())
            if blk.sc is not None:
                resid = slot.buf("b%d.scv" % b, shape)
                rec_conv(P, "b%d.sc" % b, cur, resid, blk.sce, 0, bias=blk.bns.eshift)
                P.merge_convs(len(P.calls) - 2)
            else:
                resid = cur
            out = slot.buf("b%d.out" % b, shape)
            rec_conv(P, "b%d.c2" % b, a1, None, blk.conv2e, 0, bias=blk.bn2.eshift, add_post=resid, act_dst=out, act=relu)
            cur, chw = out, ohw
        return cur, chw

    def _norm_add_act(self, plan: Plan, st: NormState, act_dst, add, add_scale, add_shift):
        q = st.pending
        st.pending = None
        x = q["x"]
        plan.hold(x, act_dst, q["part"], add, add_scale, add_shift)
        plan.add(q["key"] + ".finaddact", lib.combat_norm_add_act_fused, x.data_ptr(), _p(q["part"]), st.groups, q["rpg"],
                 q["pxg"], st.C, 1e-5, 0.0, _p(q["gamma"]), _p(q["beta"]), st.mean.data_ptr(), st.rstd.data_ptr(),
                 st.scale.data_ptr(), st.shift.data_ptr(), _p(q["rm"]), _p(q["rv"]), 0.1, _p(q["nbt"]),
                 self._scratch.data_ptr(), self._scratch.numel() * 4, add.data_ptr(), _p(add_scale), _p(add_shift),
                 act_dst.data_ptr())

    def _forward_train_body(self, P: Plan, slot: Slot, x):
        n, hw = slot.N, slot.hw
        bnk = lambda bn: dict(gamma=bn.gamma, beta=bn.beta, running=(bn.rm, bn.rv, bn.nbt))
        y0 = slot.buf("stem", (n, hw, hw, 64))
        cur = slot.buf("stem.a", y0.shape)
        self._conv_norm(P, slot, self.bn0.prefix, x, y0, self.stem, groups=1, act_dst=cur, **bnk(self.bn0))
        chw = hw
        for b, blk in enumerate(self.blocks):
            ohw = chw // blk.stride
            shape = (n, ohw, ohw, blk.planes)
            y1, a1 = slot.buf("b%d.y1" % b, shape), slot.buf("b%d.a1" % b, shape)
            self._conv_norm(P, slot, blk.bn1.prefix, cur, y1, blk.conv1, groups=1, act_dst=a1, **bnk(blk.bn1))
            y2 = slot.buf("b%d.y2" % b, shape)
            st2 = self._conv_norm(P, slot, blk.bn2.prefix, a1, y2, blk.conv2, groups=1, defer=True, **bnk(blk.bn2))
            out = slot.buf("b%d.out" % b, shape)
            if blk.sc is not None:
                ys = slot.buf("b%d.ys" % b, shape)
                sts = self._conv_norm(P, slot, blk.bns.prefix, cur, ys, blk.sc, groups=1, **bnk(blk.bns))
                self._norm_add_act(P, st2, out, ys, sts.scale, sts.shift)
            else:
                self._norm_add_act(P, st2, out, cur, None, None)
            cur, chw = out, ohw
        return cur, chw

    # ---- backward
    def _head_grad(self, P: Plan, slot: Slot, loss_weight: float, train: bool):
        """Gradient w.r.t. the last block's output, through its ReLU."""
        fp, n = self.fp, slot.N
        h = self.head_bufs(slot)
        feat = slot.bufs["b%d.out" % (len(self.blocks) - 1)]
        d_feat = slot.buf("g.feat", feat.shape)
        P.add("head_bwd", lib.combat_head_bwd, h["pooled"].data_ptr() if train else None, n, slot.feat_hw, feat.shape[-1],
              self.lin_w.data_ptr(), self.classes, h["logits"].data_ptr(), h["targets"].data_ptr(), loss_weight,
              h["dlogits"].data_ptr(), d_feat.data_ptr(), fp.grad_phys("linear.weight").data_ptr() if train else None,
              fp.grad_phys("linear.bias").data_ptr() if train else None)
        d_out = slot.buf("g.feat.m", feat.shape)
        P.add("head_bwd.relu", lib.combat_relu_mask, d_feat.data_ptr(), feat.data_ptr(), feat.numel(), d_out.data_ptr())
        return d_out

    def backward_train_plan(self, slot: Slot, loss_weight: float = 1.0) -> Plan:
        key = "bwd.train.%g" % loss_weight
        if key in slot.plans:
            return slot.plans[key]
        P = Plan("resnet." + key)
        P.flush_at_marks = self.flush_at_marks
        fp = self.fp
        G = lambda name, like: slot.buf("g." + name, like.shape)
        P.add("zero_grad", _zero_grad_call, fp)
        d_out = self._head_grad(P, slot, loss_weight, True)

        def bn_bwd(bn, key_, dz, x_pre, name):
            """BatchNorm backward of dz w.r.t. the raw tensor x_pre (sums taken from the tensors)."""
            st = slot.norm[bn.prefix]
            st.bpending = dict(part=None, rpg=0, gamma=bn.gamma, dgamma=fp.grad_phys(bn.prefix + ".weight"),
                               dbeta=fp.grad_phys(bn.prefix + ".bias"))
            dx = G(name, x_pre)
            self._bwd_apply(P, slot, key_, dz, x_pre, dx, st)
            return dx

        for b in reversed(range(len(self.blocks))):
            blk, pre = self.blocks[b], self.blocks[b].prefix
            cur_in = self._block_in(slot, b)
            y1, a1, y2 = slot.bufs["b%d.y1" % b], slot.bufs["b%d.a1" % b], slot.bufs["b%d.y2" % b]
            dy2 = bn_bwd(blk.bn2, "g." + blk.bn2.prefix, d_out, y2, "b%d.dy2" % b)
            tsc, dys = None, None
            if blk.sc is not None:
                dys = bn_bwd(blk.bns, "g." + blk.bns.prefix, d_out, slot.bufs["b%d.ys" % b], "b%d.dys" % b)
                rec_wgrad(P, "b%d.sc.wgrad" % b, cur_in, dys, blk.sc, fp.grad_phys(pre + "shortcut.0.weight"))
            rec_wgrad(P, "b%d.c2.wgrad" % b, a1, dy2, blk.conv2, fp.grad_phys(pre + "conv2.weight"))
            st1 = slot.norm[blk.bn1.prefix]
            dz1 = G("b%d.dz1" % b, y1)
            self._dgrad_norm(P, slot, "g." + blk.bn1.prefix, dy2, dz1, blk.conv2, y1, st1, group_stride=0, slope=0.0,
                             gamma=blk.bn1.gamma, dgamma=fp.grad_phys(pre + "bn1.weight"),
                             dbeta=fp.grad_phys(pre + "bn1.bias"))
            dy1 = G("b%d.dy1" % b, y1)
            self._bwd_apply(P, slot, "g." + blk.bn1.prefix, dz1, y1, dy1, st1)
            rec_wgrad(P, "b%d.c1.wgrad" % b, cur_in, dy1, blk.conv1, fp.grad_phys(pre + "conv1.weight"))
            dxin = G("b%d.dx" % b, cur_in) if b > 0 else None
            fused = None
            if blk.sc is not None:       # the shortcut's share of the block-input gradient: along with conv1's, or a launch
                fused = fuse_shortcut(dy1, dxin, blk.conv1, dys, blk.sc) if b > 0 else None
                if fused is None:
                    tsc = G("b%d.tsc" % b, cur_in)
                    rec_conv(P, "b%d.sc.dgrad" % b, dys, tsc, blk.sc, 1)
            other = (None if fused is not None else tsc) if blk.sc is not None else d_out
            if b > 0:   # through the previous block's final ReLU
                rec_conv(P, "b%d.c1.dgrad" % b, dy1, dxin, blk.conv1, 1, shortcut=fused, add_pre=other, mask_x=cur_in,
                         mask_activated=True)
                d_out = dxin
                if b in (6, 4, 2):
                    P.mark(fp.offsets[pre + "conv1.weight"][0])
            else:       # into the stem: ReLU + BatchNorm of the raw stem output
                y0, st0 = slot.bufs["stem"], slot.norm[self.bn0.prefix]
                dz0 = G("stem.dz", y0)
                self._dgrad_norm(P, slot, "g." + self.bn0.prefix, dy1, dz0, blk.conv1, y0, st0, group_stride=0, slope=0.0,
                                 gamma=self.bn0.gamma, dgamma=fp.grad_phys("bn1.weight"), dbeta=fp.grad_phys("bn1.bias"),
                                 add_pre=other)
                dy0 = G("stem.dy", y0)
                self._bwd_apply(P, slot, "g." + self.bn0.prefix, dz0, y0, dy0, st0)
                rec_wgrad(P, "stem.wgrad", self.input(slot), dy0, self.stem, fp.grad_phys("conv1.weight"), aux=False)
        P.mark(0)
        slot.plans[key] = P
        return P

    def backward_eval_plan(self, slot: Slot, loss_weight: float) -> Plan:
        """Input gradient of an eval-mode forward (result: 'g.img'); masks from the activated tensors, the
        BatchNorm scales from the folded operands (bn2, shortcut) or the mask tables (bn1, stem)."""
        key = "bwd.eval.%g" % loss_weight
        if key in slot.plans:
            return slot.plans[key]
        P = Plan("resnet." + key)
        n = slot.N
        d_out = self._head_grad(P, slot, loss_weight, False)
        for b in reversed(range(len(self.blocks))):
            blk = self.blocks[b]
            cur_in, a1 = self._block_in(slot, b), slot.bufs["b%d.a1" % b]
            dy1 = slot.buf("g.b%d.dy1" % b, a1.shape)
            rec_conv(P, "b%d.c2.dgrad" % b, d_out, dy1, blk.conv2e, 1, mask_x=a1, mask=blk.bn1.eval_affine(),
                     mask_mul_scale=True, mask_activated=True)
            other, fused = d_out, None
            dxin = slot.buf("g.b%d.dx" % b, cur_in.shape)
            if blk.sc is not None:
                fused = fuse_shortcut(dy1, dxin, blk.conv1, d_out, blk.sce) if b > 0 else None
                other = None
                if fused is None:
                    other = slot.buf("g.b%d.tsc" % b, cur_in.shape)
                    rec_conv(P, "b%d.sc.dgrad" % b, d_out, other, blk.sce, 1)
            if b > 0:
                rec_conv(P, "b%d.c1.dgrad" % b, dy1, dxin, blk.conv1, 1, shortcut=fused, add_pre=other, mask_x=cur_in,
                         mask_activated=True)
            else:
                rec_conv(P, "b%d.c1.dgrad" % b, dy1, dxin, blk.conv1, 1, add_pre=other, mask_x=cur_in,
                         mask=self.bn0.eval_affine(), mask_mul_scale=True, mask_activated=True)
            d_out = dxin
        gimg = slot.buf("g.img", (n, slot.hw, slot.hw, 8))
        rec_conv(P, "stem.dgrad", d_out, gimg, self.stem, 1)
        slot.plans[key] = P
        return P


# --------------------------------------------------------------------------------------------
# UNet generator
# --------------------------------------------------------------------------------------------


class UnetEngine(NetEngine):
    """UnetGenerator (networks/models.py:268-341).  Tensors: t[name] = raw conv outputs,
    up[level] = LeakyReLU(bilinear_up(...)) decoder inputs; InstanceNorm + LeakyReLU live in the
    prologue of the consuming convolution."""

    LR = 0.2

    def __init__(self, module):
        super().__init__(module)
        m = module
        self.nf = m.nf
        self.pc: Dict[str, PackedConv] = {}
        self.bias: Dict[str, torch.Tensor] = {}
        for name, ci, co, stride, _ in UNET_LAYERS:
            conv = getattr(m, name)
            cin_pad = 8 if ci == 0 else conv.in_channels
            self.pc[name] = self._pc(conv.weight.data, stride, 1, cin_pad, dup=(ci == 0), need_dgrad=(ci != 0))
            self.bias[name] = conv.bias.data if conv.bias is not None else None
        self.out_bias8 = torch.zeros(8, dtype=f32, device=self.device)
        self.ones = torch.ones(64 * 8, dtype=f32, device=self.device)
        self.zeros = torch.zeros(64 * 8, dtype=f32, device=self.device)

    def _refresh_extra(self):
        if self.bias8_taken:          # the caller copied it with its own small copies (small_refresh_copy)
            self.bias8_taken = False
            return
        b = self.bias["upconv0_0"]
        if b is not None:
            self.out_bias8[: b.numel()].copy_(b)

    bias8_taken = False

    def small_refresh_copy(self):
        """(dst pointer, src pointer, bytes) of the one small copy the next refresh() would issue by itself (the output
        layer's bias into its 8-channel form), for a caller that batches it with copies of its own (combat_copy3) --
        or None if the operands are current.  The next refresh() then skips it."""
        b = self.bias["upconv0_0"]
        if not self.weights_dirty or b is None:
            return None
        self.bias8_taken = True
        return self.out_bias8.data_ptr(), b.data_ptr(), 4 * b.numel()

    def input(self, slot: Slot) -> torch.Tensor:
        return slot.buf("x", (slot.N, slot.hw, slot.hw, 8))

    def output(self, slot: Slot) -> torch.Tensor:
        return slot.buf("noise", (slot.N, slot.hw, slot.hw, 8))

    def _in_aff(self, st: NormState) -> Affine:
        return Affine(st.scale, st.shift, st.C, True, self.LR)

    def forward_plan(self, slot: Slot) -> Plan:
        if "fwd" in slot.plans:
            return slot.plans["fwd"]
        P = Plan("unet.fwd")
        n, hw, nf = slot.N, slot.hw, self.nf
        T = lambda name, s, c: slot.buf("t." + name, (n, s, s, c))
        x = self.input(slot)
        h1, h2, h3, h4 = hw // 2, hw // 4, hw // 8, hw // 16
        pc, bs = self.pc, self.bias
        t00 = T("conv0_0", h1, nf)
        # conv0_1 reads LeakyReLU(conv0_0 output), no norm in between: the activation is conv0_0's second output
        a01 = slot.buf("a.conv0_1", t00.shape)
        rec_conv(P, "conv0_0", x, t00, pc["conv0_0"], 0, bias=bs["conv0_0"], act_dst=a01,
                 act=Affine(self.ones, self.zeros, 0, True, self.LR))

        def cn(name, src, s, c, pro, defer=False):
            # pro: None (src is used as is), an Affine (activation only), or the NormState of src whose
            # finalize was deferred to here.  No bias: InstanceNorm removes any per-(image, channel) constant, so
            # IN(conv(x) + b) == IN(conv(x)) exactly, and leaving b out keeps the stored bf16
            # tensor better centred (less rounding error amplified by 1/sigma)
            dst = T(name, s, c)
            if pro is not None:   # materialise the activated input once ('a.<conv>'): forward and wgrad then need no prologue
                act = slot.buf("a." + name, src.shape)
                if isinstance(pro, NormState):
                    self._norm_act(P, pro, act, self.LR)
                else:
                    rec_act(P, name + ".act", src, act, pro)
                src = act
            st = self._conv_norm(P, slot, name, src, dst, pc[name], groups=n, defer=defer)
            return dst, st

        # defer=True: the layer's InstanceNorm is finalised by the launch that writes the next conv's input
        t01, s01 = cn("conv0_1", a01, h1, nf, None, True)
        t10, s10 = cn("conv1_0", t01, h2, nf * 2, s01, True)
        t11, s11 = cn("conv1_1", t10, h2, nf * 2, s10, True)
        t20, s20 = cn("conv2_0", t11, h3, nf * 4, s11, True)
        t21, s21 = cn("conv2_1", t20, h3, nf * 4, s20, True)
        t30, s30 = cn("conv3_0", t21, h4, nf * 8, s21, True)
        t31, s31 = cn("conv3_1", t30, h4, nf * 8, s30, True)

        def up(level, y, sy, skip, ss, s_in, c):
            o = slot.buf("up%d" % level, (n, 2 * s_in, 2 * s_in, c))
            P.hold(y, skip)
            if sy.pending is not None:   # y's InstanceNorm is finalised by the launch that upsamples it
                q = sy.pending
                sy.pending = None
                P.hold(q["part"])
                P.add("up%d.fin" % level, lib.combat_unet_up_fused, y.data_ptr(), _p(q["part"]), q["rpg"], _p(skip),
                      ss.scale.data_ptr() if ss else None, ss.shift.data_ptr() if ss else None, n, s_in, s_in, c, 1e-5,
                      sy.mean.data_ptr(), sy.rstd.data_ptr(), sy.scale.data_ptr(), sy.shift.data_ptr(), o.data_ptr())
                return o
            P.add("up%d" % level, lib.combat_unet_up_fwd, y.data_ptr(), sy.scale.data_ptr(), sy.shift.data_ptr(),
                  _p(skip), ss.scale.data_ptr() if ss else None, ss.shift.data_ptr() if ss else None, n, s_in, s_in, c,
                  o.data_ptr())
            return o

        u3 = up(3, t31, s31, None, None, h4, nf * 8)
        tu31, su31 = cn("upconv3_1", u3, h3, nf * 8, None, True)
        tu30, su30 = cn("upconv3_0", tu31, h3, nf * 4, su31, True)
        u2 = up(2, tu30, su30, t21, s21, h3, nf * 4)
        tu21, su21 = cn("upconv2_1", u2, h2, nf * 4, None, True)
        tu20, su20 = cn("upconv2_0", tu21, h2, nf * 2, su21, True)
        u1 = up(1, tu20, su20, t11, s11, h2, nf * 2)
        tu11, su11 = cn("upconv1_1", u1, h1, nf * 2, None, True)
        tu10, su10 = cn("upconv1_0", tu11, h1, nf, su11, True)
        u0 = up(0, tu10, su10, t01, s01, h1, nf)
        tu01, su01 = cn("upconv0_1", u0, hw, nf, None)
        rec_conv(P, "upconv0_0", tu01, self.output(slot), pc["upconv0_0"], 0, pro=self._in_aff(su01),
                 bias=self.out_bias8, tanh_out=True)
        slot.plans["fwd"] = P
        return P

    def backward_plan(self, slot: Slot) -> Plan:
        """Given slot buffer 'g.z' (gradient w.r.t. the pre-tanh output, bf16 NHWC c8) produce all
        parameter gradients in fp.grad (zeroed first).  Conv biases that feed an InstanceNorm get
        an exactly-zero gradient (the norm removes any per-channel constant)."""
        if "bwd" in slot.plans:
            return slot.plans["bwd"]
        P = Plan("unet.bwd")
        P.flush_at_marks = self.flush_at_marks
        fp, n, hw, nf = self.fp, slot.N, slot.hw, self.nf
        pc = self.pc
        t = lambda name: slot.bufs["t." + name]
        stn = lambda name: slot.norm[name]
        G = lambda name, like: slot.buf("g." + name, like.shape)
        P.add("zero_grad", _zero_grad_call, fp)
        gz = slot.buf("g.z", (n, hw, hw, 8))

        def through_norm(name, dy, pcv, src_name, add_pre=None, first=False):
            """dy = gradient w.r.t. raw output of conv `pcv` whose input is LR(IN(t[src_name])).
            wgrad of pcv, then gradient w.r.t. t[src_name] (returned).  first: the plan's first layer -- its auxiliary
            calls are recorded BEHIND its input-gradient launch, so that they are handed the plan stream's state by that
            launch's completion event: an auxiliary call that opens a plan needs an event record in the plan's queue,
            which held the first kernel of the generator's backward back by ~20 us (tools/timeline.py)."""
            src, st = t(src_name), stn(src_name)
            act = slot.bufs.get("a." + name)   # LR(IN(src)) of the forward (all but the K = 8 output layer)

            def wgrad():
                if act is not None:
                    rec_wgrad(P, name + ".wgrad", act, dy, pcv, fp.grad_phys(name + ".weight"))
                else:
                    rec_wgrad(P, name + ".wgrad", src, dy, pcv, fp.grad_phys(name + ".weight"), self._in_aff(st))
            if not first:
                wgrad()
            dz = G(src_name + ".dz", src)
            self._dgrad_norm(P, slot, "g." + src_name, dy, dz, pcv, src, st, group_stride=st.C, slope=self.LR,
                             add_pre=add_pre)
            if first:
                P.add("db.upconv0_0", lib.combat_colsum, gz.data_ptr(), n * hw * hw, 8, 3,
                      fp.grad_phys("upconv0_0.bias").data_ptr(), aux=True)
                wgrad()
            dx = G(src_name + ".dx", src)
            self._bwd_apply(P, slot, "g." + src_name, dz, src, dx, st)
            return dx

        def through_up(name, dy, pcv, level, y_name):
            """dy = gradient w.r.t. raw output of conv `pcv` whose input is up[level].  Returns
            (du, d_y): du = gradient w.r.t. the pre-upsample sum, d_y = gradient w.r.t. t[y_name]."""
            u = slot.bufs["up%d" % level]
            rec_wgrad(P, name + ".wgrad", u, dy, pcv, fp.grad_phys(name + ".weight"), None)
            du_full = G("up%d.d" % level, u)
            rec_conv(P, name + ".dgrad", dy, du_full, pcv, 1)
            y, st = t(y_name), stn(y_name)
            du = G("u%d" % level, y)
            dy_out = G(y_name + ".dx", y)
            if y.shape[1] * y.shape[2] <= 64:   # upsample adjoint + InstanceNorm backward, one launch (one workgroup per
                # (image, 64 channels): on 16 x 16 maps that is half the chip doing all the work -- two launches there)
                P.hold(du_full, u, y, du, dy_out)
                P.add("up%d.bwd.norm" % level, lib.combat_unet_up_bwd_fused, du_full.data_ptr(), u.data_ptr(), y.data_ptr(),
                      st.mean.data_ptr(), st.rstd.data_ptr(), n, y.shape[1], y.shape[2], y.shape[3], du.data_ptr(),
                      dy_out.data_ptr())
                return du, dy_out
            P.add("up%d.bwd" % level, lib.combat_unet_up_bwd, du_full.data_ptr(), u.data_ptr(), n, y.shape[1],
                  y.shape[2], y.shape[3], du.data_ptr())
            self._bwd_apply(P, slot, "g." + y_name, du, y, dy_out, st)    # sums taken from du / y directly
            return du, dy_out

        d = through_norm("upconv0_0", gz, pc["upconv0_0"], "upconv0_1", first=True)
        du0, d = through_up("upconv0_1", d, pc["upconv0_1"], 0, "upconv1_0")
        d = through_norm("upconv1_0", d, pc["upconv1_0"], "upconv1_1")
        du1, d = through_up("upconv1_1", d, pc["upconv1_1"], 1, "upconv2_0")
        d = through_norm("upconv2_0", d, pc["upconv2_0"], "upconv2_1")
        du2, d = through_up("upconv2_1", d, pc["upconv2_1"], 2, "upconv3_0")
        d = through_norm("upconv3_0", d, pc["upconv3_0"], "upconv3_1")
        _, d = through_up("upconv3_1", d, pc["upconv3_1"], 3, "conv3_1")
        P.mark(fp.offsets["upconv3_1.weight"][0])          # decoder gradients complete
        d = through_norm("conv3_1", d, pc["conv3_1"], "conv3_0")
        d = through_norm("conv3_0", d, pc["conv3_0"], "conv2_1", add_pre=du2)
        P.mark(fp.offsets["conv3_0.weight"][0])            # + the two 512-channel encoder layers (14 MB of the rest)
        d = through_norm("conv2_1", d, pc["conv2_1"], "conv2_0")
        d = through_norm("conv2_0", d, pc["conv2_0"], "conv1_1", add_pre=du1)
        d = through_norm("conv1_1", d, pc["conv1_1"], "conv1_0")
        d = through_norm("conv1_0", d, pc["conv1_0"], "conv0_1", add_pre=du0)
        # conv0_1 reads LeakyReLU(conv0_0 output): no norm in between
        t00 = t("conv0_0")
        lr_only = Affine(None, None, 0, True, self.LR)
        rec_wgrad(P, "conv0_1.wgrad", slot.bufs["a.conv0_1"], d, pc["conv0_1"], fp.grad_phys("conv0_1.weight"))
        d00 = G("conv0_0.dx", t00)
        rec_conv(P, "conv0_1.dgrad", d, d00, pc["conv0_1"], 1, mask_x=t00, mask=Affine(None, None, 0, True, self.LR))
        P.add("db.conv0_0", lib.combat_colsum, d00.data_ptr(), d00.numel() // d00.shape[-1], d00.shape[-1],
              d00.shape[-1], fp.grad_phys("conv0_0.bias").data_ptr(), aux=True)
        rec_wgrad(P, "conv0_0.wgrad", self.input(slot), d00, pc["conv0_0"], fp.grad_phys("conv0_0.weight"), aux=False)
        P.mark(0)
        balance_wgrads(P, self.device)
        slot.plans["bwd"] = P
        return P


# --------------------------------------------------------------------------------------------
# WaNet grid generator
# --------------------------------------------------------------------------------------------


class GridEngine(NetEngine):
    """GridGenerator (networks/models.py:344-385) + the warp of train_generator_wanet.py:151-157.

    The reference pools an affine-free InstanceNorm output -- spatial mean exactly 0 -- so the network's output
    is tanh(fc2(lrelu(fc1.bias))) for every input and its encoder receives no gradient (pinned against the
    reference module: tests/test_oracle_golden.py::test_grid_generator_is_a_constant_field...).  The engine
    therefore computes the [2][S][S] field from the three head tensors, the [H][H][2] sampling grid shared by the
    batch, and in the backward the gradients of fc1.bias, fc2.weight and fc2.bias; every other parameter keeps a
    zero gradient (weight decay still applies in the optimiser, as in the reference)."""

    def __init__(self, module):
        super().__init__(module)
        m = module
        self.S, self.nf = int(m.S), int(m.fc1.out_features)
        self.nout = 2 * self.S * self.S
        self.field = torch.zeros(self.nout, dtype=f32, device=self.device)
        self._grids: Dict[int, dict] = {}

    def refresh(self) -> None:   # no packed operands
        self.weights_dirty = False

    def grid_bufs(self, hw: int) -> dict:
        g = self._grids.get(hw)
        if g is None:
            # U[H][S]: the linear map of F.upsample(size=H, mode="bicubic", align_corners=True) along one axis
            # (train_generator_wanet.py:152), taken from the operator itself on the basis vectors
            eye = torch.eye(self.S, dtype=f32).view(self.S, 1, self.S, 1)
            u = torch.nn.functional.interpolate(eye, size=(hw, 1), mode="bicubic", align_corners=True)   # [S,1,H,1]
            U = u[:, 0, :, 0].t().contiguous()
            g = dict(U=U.to(self.device), noise_grid=torch.zeros(hw, hw, 2, dtype=f32, device=self.device),
                     grid=torch.zeros(hw, hw, 2, dtype=f32, device=self.device))
            self._grids[hw] = g
        return g

    def _head(self):
        fp = self.fp
        return (fp._slice(fp.flat, "fc1.bias"), fp._slice(fp.flat, "fc2.weight"), fp._slice(fp.flat, "fc2.bias"))

    def forward_grid(self, hw: int, rescale: float, st=None) -> dict:
        """field, noise_grid and the sampling grid for H = hw (two launches)."""
        st = torch.cuda.current_stream().cuda_stream if st is None else st
        g = self.grid_bufs(hw)
        b1, w2, b2 = self._head()
        ops.check(lib.combat_grid_head_fwd(b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), self.nf, self.nout,
                                           self.field.data_ptr(), st), "combat_grid_head_fwd")
        ops.check(lib.combat_wanet_grid(self.field.data_ptr(), g["U"].data_ptr(), self.S, hw, float(rescale),
                                        g["noise_grid"].data_ptr(), g["grid"].data_ptr(), st), "combat_wanet_grid")
        return g

    def head_grad_range(self) -> Tuple[int, int]:
        """[lo, hi) of the flat gradient buffer that holds every gradient this engine ever writes (fc1.bias,
        fc2.weight, fc2.bias are consecutive parameters); the rest of fp.grad stays exactly zero."""
        offs = [self.fp.offsets[k] for k in ("fc1.bias", "fc2.weight", "fc2.bias")]
        lo = min(o for o, _, _ in offs)
        hi = max(o + n for o, n, _ in offs)
        inside = sum(n for o, n, _ in self.fp.offsets.values() if lo <= o < hi)
        assert inside == sum(n for _, n, _ in offs), "head parameters are not contiguous in the flat buffer"
        return lo, hi

    def backward_field(self, partial: torch.Tensor, groups: int, hw: int, rescale: float, l2_scale: float, st=None) -> None:
        """d(loss)/d(grid) partial sums -> gradients of the head in fp.grad (the rest of fp.grad must be zero)."""
        st = torch.cuda.current_stream().cuda_stream if st is None else st
        g, fp = self.grid_bufs(hw), self.fp
        b1, w2, _ = self._head()
        ops.check(lib.combat_wanet_field_bwd(partial.data_ptr(), groups, g["noise_grid"].data_ptr(), g["U"].data_ptr(), self.S, hw,
                                             float(rescale), float(l2_scale), self.field.data_ptr(), b1.data_ptr(), w2.data_ptr(),
                                             self.nf, fp._slice(fp.grad, "fc1.bias").data_ptr(),
                                             fp._slice(fp.grad, "fc2.weight").data_ptr(),
                                             fp._slice(fp.grad, "fc2.bias").data_ptr(), None, st), "combat_wanet_field_bwd")


# --------------------------------------------------------------------------------------------
# Frequency detector (metric only)
# --------------------------------------------------------------------------------------------


class FreqEngine(NetEngine):
    """FrequencyModel (defenses/frequency_based/model.py:8-52) in eval mode: 6 x
    (conv+bias -> ELU -> BatchNorm(running stats)), 2x2 max-pool after every second, Linear."""

    def __init__(self, module):
        super().__init__(module)
        m = module
        self.layers = []
        cin = 8
        for i, w in enumerate(m.WIDTHS, start=1):
            conv, bn = getattr(m, "conv%d" % i), getattr(m, "bn%d" % i)
            self.layers.append((self._pc(conv.weight.data, 1, 1, cin, dup=(i == 1), need_dgrad=False), conv.bias.data,
                                _BN(self, bn, "bn%d" % i)))
            cin = w
        self.lin_w, self.lin_b = m.linear6.weight.data, m.linear6.bias.data
        self.classes = m.linear6.out_features

    def _refresh_extra(self):
        for _, _, bn in self.layers:
            bn.fold()

    def input(self, slot: Slot) -> torch.Tensor:
        return slot.buf("x", (slot.N, slot.hw, slot.hw, 8))

    def forward_plan(self, slot: Slot) -> Plan:
        if "fwd" in slot.plans:
            return slot.plans["fwd"]
        P = Plan("freq.fwd")
        n, s = slot.N, slot.hw
        cur = self.input(slot)
        for i, (pc, bias, bn) in enumerate(self.layers, start=1):
            raw = slot.buf("c%d" % i, (n, s, s, pc.K))
            rec_conv(P, "conv%d" % i, cur, raw, pc, 0, bias=bias)
            act = slot.buf("a%d" % i, (n, s, s, pc.K))
            P.add("elu_bn%d" % i, lib.combat_elu_affine, raw.data_ptr(), n * s * s, pc.K, bn.escale.data_ptr(),
                  bn.eshift.data_ptr(), act.data_ptr())
            cur = act
            if i % 2 == 0:
                pooled = slot.buf("p%d" % i, (n, s // 2, s // 2, pc.K))
                P.add("pool%d" % i, lib.combat_maxpool2, cur.data_ptr(), n, s, s, pc.K, pooled.data_ptr())
                cur, s = pooled, s // 2
        logits = slot.buf("logits", (n, self.classes), f32)
        P.add("linear", lib.combat_linear_nhwc, cur.data_ptr(), n, s, s, cur.shape[-1], self.lin_w.data_ptr(),
              self.lin_b.data_ptr(), self.classes, logits.data_ptr())
        slot.plans["fwd"] = P
        return P


# --------------------------------------------------------------------------------------------
# drop-in module(x) with autograd (reference call signature; not the fast path)
# --------------------------------------------------------------------------------------------

ENGINES = {"preact_resnet18": PreActEngine, "resnet18": ResNetEngine, "unet": UnetEngine, "freq": FreqEngine,
           "gridgen": GridEngine}


def build_engine(module) -> NetEngine:
    cls = ENGINES.get(module.arch)
    if cls is None:
        raise NotImplementedError("no HIP engine for architecture %r yet" % module.arch)
    return cls(module)


def _images_to_c8(x: torch.Tensor, dst: torch.Tensor) -> None:
    ops.image_to_c8(x.contiguous().float(), dst)


def pad_batch(n: int) -> int:
    """Inference batches are rounded up to a multiple of 16 so that ragged sizes (the non-target subset
    of a test batch) share slots and plans; every sample is independent in eval mode (BatchNorm uses
    running statistics, InstanceNorm is per sample), so the padding rows change nothing."""
    return max(16, (n + 15) // 16 * 16)


class _ClassifierFn(torch.autograd.Function):
    """logits = netC(x) through the HIP plans.  Backward supports the two uses the reference
    makes of a classifier: train mode -> parameter gradients; eval mode -> input gradient."""

    @staticmethod
    def forward(ctx, module, x, *params):
        eng: PreActEngine = module._net_engine()
        eng.refresh()
        n, _, hw, _ = x.shape
        train = module.training
        slot = eng.slot("module.train" if train else "module.eval", n if train else pad_batch(n), hw)
        _images_to_c8(x, eng.input(slot))
        eng.forward_plan(slot, train).run()
        if train:
            eng.fold_bn()
        ctx.module, ctx.slot, ctx.train = module, slot, train
        return eng.head_bufs(slot)["logits"][:n].clone()

    @staticmethod
    def backward(ctx, dlogits):
        raise NotImplementedError(
            "differentiate through combat_amd.step.AlternatedStep (fused loss heads); module(x) is inference-only")


class _GeneratorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, x, *params):
        eng: UnetEngine = module._net_engine()
        eng.refresh()
        n, _, hw, _ = x.shape
        if n == 0:
            return x.new_zeros(x.shape)
        slot = eng.slot("module", pad_batch(n), hw)
        _images_to_c8(x, eng.input(slot))
        eng.forward_plan(slot).run()
        out = torch.empty(slot.N, 3, hw, hw, dtype=f32, device=x.device)
        ops.nhwc_to_nchw_f32(eng.output(slot), 3, out)
        return out[:n]

    @staticmethod
    def backward(ctx, g):
        raise NotImplementedError(
            "differentiate through combat_amd.step.AlternatedStep; module(x) is inference-only")


class _GridFn(torch.autograd.Function):
    """[B, 2, S, S] field of a GridGenerator: the same for every sample (see GridEngine).  Like the other modules'
    `module(x)` it is inference-only: a backward through it raises instead of handing the caller's own training
    loop a silent zero gradient (the differentiated path is combat_amd.step.WanetStep)."""

    @staticmethod
    def forward(ctx, module, x, *params):
        eng = module._net_engine()
        st = torch.cuda.current_stream().cuda_stream
        b1, w2, b2 = eng._head()
        ops.check(lib.combat_grid_head_fwd(b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), eng.nf, eng.nout,
                                           eng.field.data_ptr(), st), "combat_grid_head_fwd")
        return eng.field.view(1, 2, eng.S, eng.S).expand(x.shape[0], -1, -1, -1).clone()

    @staticmethod
    def backward(ctx, g):
        raise NotImplementedError(
            "differentiate through combat_amd.step.WanetStep; module(x) is inference-only")


def module_forward(module, x: torch.Tensor) -> torch.Tensor:
    """The reference's `module(x)` call signature (float32 NCHW in, float32 out) on the HIP path."""
    if x.device.type != "cuda":
        raise CombatHipError("combat_amd modules run on the GPU only; got a %s tensor" % x.device)
    params = tuple(module.parameters())
    if module.arch == "unet":
        return _GeneratorFn.apply(module, x, *params)
    if module.arch in ("preact_resnet18", "resnet18"):
        return _ClassifierFn.apply(module, x, *params)
    if module.arch == "gridgen":
        return _GridFn.apply(module, x, *params)
    if module.arch == "freq":
        eng = module._net_engine()
        eng.refresh()
        n, _, hw, _ = x.shape
        slot = eng.slot("module", pad_batch(n), hw)
        ops.image_to_c8(x.contiguous().float(), eng.input(slot))  # hi/lo split keeps the DCT's dynamic range
        eng.forward_plan(slot).run()
        return slot.bufs["logits"][:n].clone()
    raise NotImplementedError(module.arch)
