"""Device-side helpers shared by the entry scripts: the pieces of the reference scripts that are
tensor arithmetic (train_generator.py:47-55, 190-194, 245-247, 353-391) routed to the HIP kernels."""
from __future__ import annotations

import torch

from . import ops, trigger

_consts = {}


def _c(hw: int, ratio: float, device):
    key = (hw, ratio, str(device))
    if key not in _consts:
        _consts[key] = (trigger.lowpass_matrix(hw, ratio).to(device), trigger.dct_matrix(hw).float().to(device))
    return _consts[key]


def create_backdoor(netG, inputs: torch.Tensor, opt, sigma: float = None) -> torch.Tensor:
    """netG -> low_freq -> clamp(x + noise*rate) -> GaussianBlur (one sigma per call, drawn here from
    torch's global generator like T.GaussianBlur) for a float32 NCHW device batch; no gradient."""
    n, _, hw, _ = inputs.shape
    if n == 0:
        return inputs
    eng = netG._net_engine()
    eng.refresh()
    inputs = inputs.contiguous().float()
    if netG.arch == "gridgen":   # WaNet: warp by the generator's field (train_generator_wanet.py:151-157, :346-352)
        from ._lib import lib
        g = eng.forward_grid(hw, float(opt.grid_rescale))
        out = torch.empty_like(inputs)
        ops.check(lib.combat_warp_fwd(inputs.data_ptr(), None, g["grid"].data_ptr(), 0, n, hw, out.data_ptr(),
                                      torch.cuda.current_stream().cuda_stream), "combat_warp_fwd")
        return out
    from .engine import pad_batch
    slot = eng.slot("api", pad_batch(n), hw)
    ops.image_to_c8(inputs, eng.input(slot))
    eng.forward_plan(slot).run()
    if sigma is None:
        sigma = trigger.sample_sigma(getattr(opt, "sigma", (0.1, 1.0)))
    pm, _ = _c(hw, opt.ratio, inputs.device)
    k1 = torch.from_numpy(trigger.gaussian_kernel1d(sigma, opt.kernel_size)).to(inputs.device)
    out = torch.empty_like(inputs)
    ops.trigger_fwd(inputs, eng.output(slot), pm, k1, float(opt.noise_rate), out)
    return out


def frequency_logits(netF, inputs_bd: torch.Tensor, opt) -> torch.Tensor:
    """netF(dct_2d(((x + 1) / 2 * 255).byte())) (train_generator.py:245-247, 381-383)."""
    n, _, hw, _ = inputs_bd.shape
    eng = netF._net_engine()
    eng.refresh()
    from .engine import pad_batch
    slot = eng.slot("api", pad_batch(n), hw)
    _, dm = _c(hw, opt.ratio, inputs_bd.device)
    ops.dct_u8(inputs_bd.contiguous().float(), dm, eng.input(slot))
    eng.forward_plan(slot).run()
    return slot.bufs["logits"][:n].clone()


def sync_momentum_to_optimizer(optimizer, module) -> None:
    """The fused SGD keeps momentum in the engine's flat buffer; mirror it into the torch optimiser's
    state so `optimizer.state_dict()` (checkpoint key optimizerC/optimizerG) has the reference layout."""
    eng = module._net_engine()
    for name, p in module.named_parameters():
        optimizer.state[p]["momentum_buffer"] = eng.fp.logical(eng.fp.mom, name).detach().clone()


def load_momentum_from_optimizer(optimizer, module) -> None:
    eng = module._net_engine()
    for name, p in module.named_parameters():
        buf = optimizer.state.get(p, {}).get("momentum_buffer")
        if buf is not None:
            eng.fp.logical(eng.fp.mom, name).copy_(buf)
