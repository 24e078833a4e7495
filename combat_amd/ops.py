"""Tensor-level wrappers over the C ABI: torch tensors in, raw pointers out.

torch is used only for device memory and streams.  Every wrapper launches on
``torch.cuda.current_stream()`` and raises :class:`combat_amd._lib.CombatHipError` (naming the
kernel and shape) on a non-zero status.  Activations are ``[N, H, W, C]`` bf16 contiguous
tensors (C a multiple of 8); see include/combat_hip.h for every convention.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib
from ._lib import ConvArgs, WgradArgs, check, lib

bf16 = torch.bfloat16


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _round_up(v: int, m: int) -> int:
    return (v + m - 1) // m * m


def rows_pad(n: int) -> int:
    return 16 if n <= 16 else _round_up(n, 128)


@dataclass
class Affine:
    """Per-channel (group_stride 0) or per-(image, channel) (group_stride C) scale/shift with an
    optional leaky activation: the fused BatchNorm/InstanceNorm + (Leaky)ReLU that feeds a conv."""

    scale: Optional[torch.Tensor] = None
    shift: Optional[torch.Tensor] = None
    group_stride: int = 0
    act: bool = False
    slope: float = 0.0


class PackedConv:
    """One convolution's geometry plus its bf16 packed operands (forward and dgrad layouts).

    ``weight`` is the fp32 master parameter (logical OIHW, channels_last memory so its physical
    order is [K][R][S][c_real]); ``pack()`` re-derives the bf16 operands after an update."""

    def __init__(self, weight: torch.Tensor, stride: int, pad: int, c_in_padded: int, dup_hilo: bool = False,
                 need_dgrad: bool = True, row_scale: Optional[torch.Tensor] = None):
        """row_scale ([K] fp32, device): pack row_scale[n] * w[n] -- an eval-mode BatchNorm that follows the
        convolution, folded into the operands (re-read at every pack)."""
        k, c_real, r, s = weight.shape
        self.row_scale = row_scale
        assert r == s
        self.weight = weight
        self.K, self.c_real, self.R, self.stride, self.pad = k, c_real, r, stride, pad
        self.C = c_in_padded
        self.Kc = _round_up(k, 8)
        self.dup_hilo = dup_hilo
        self.taps = r * s
        dev = weight.device
        self.rows_f, self.kpad_f = rows_pad(self.Kc), _round_up(self.taps * self.C, 64)
        self.wf = torch.empty(self.rows_f, self.kpad_f, dtype=bf16, device=dev)
        self.wd = None
        self.rows_d = self.kpad_d = 0
        if need_dgrad:
            self.rows_d, self.kpad_d = rows_pad(self.C), _round_up(self.taps * self.Kc, 64)
            self.wd = torch.empty(self.rows_d, self.kpad_d, dtype=bf16, device=dev)

    def master(self) -> torch.Tensor:
        """The fp32 weights in [K][taps][c_real] physical order (a view when already channels_last)."""
        w = self.weight.detach()
        return w.permute(0, 2, 3, 1).contiguous() if not w.permute(0, 2, 3, 1).is_contiguous() else w.permute(0, 2, 3, 1)

    def fill_desc(self, d, master: torch.Tensor) -> None:
        """combat_pack_desc of this convolution (`master` = self.master(), kept alive by the caller)."""
        d.w, d.wf, d.wd = master.data_ptr(), self.wf.data_ptr(), (self.wd.data_ptr() if self.wd is not None else None)
        d.K, d.taps, d.c_real, d.C, d.dup_hilo = self.K, self.taps, self.c_real, self.C, int(self.dup_hilo)
        d.rows_pad_f, d.kpad_f = self.rows_f, self.kpad_f
        d.rows_pad_d, d.kpad_d = (self.rows_d, self.kpad_d) if self.wd is not None else (0, 0)
        d.row_scale = _p(self.row_scale)

    def pack(self) -> None:
        w = self.master()
        if self.row_scale is not None:   # the scaled form exists as a batch descriptor only
            from ._lib import PackDesc
            tab = (PackDesc * 1)()
            self.fill_desc(tab[0], w)
            dtab = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(self.weight.device)
            check(lib.combat_pack_weights_batch(dtab.data_ptr(), 1, _stream()), "combat_pack_weights_batch")
            torch.cuda.current_stream().synchronize()   # (dtab must outlive the launch)
            return
        check(lib.combat_pack_weights(w.data_ptr(), self.K, self.taps, self.c_real, self.C, int(self.dup_hilo),
                                      self.wf.data_ptr(), self.rows_f, self.kpad_f, _p(self.wd), self.rows_d,
                                      self.kpad_d, _stream()),
              "combat_pack_weights", "K=%d C=%d taps=%d" % (self.K, self.C, self.taps))

    def out_hw(self, h: int, w: int):
        return ((h + 2 * self.pad - self.R) // self.stride + 1, (w + 2 * self.pad - self.R) // self.stride + 1)


def conv_args(src, dst, pc: PackedConv, mode: int, *, pro: Optional[Affine] = None, bias=None, add_pre=None,
              mask_x=None, mask: Optional[Affine] = None, mask_mul_scale=False, add_post=None, tanh_out=False,
              stats_kind=0, stats=None, xh_mean=None, xh_rstd=None, tile=0, act_dst=None,
              act: Optional[Affine] = None, mask_activated=False, workspace=None, shortcut=None,
              pro_act_dst=None) -> ConvArgs:
    """act_dst / act: second output bf16(lrelu(y * act.scale + act.shift)) of the stored value y (the
    next layer's eval-mode BatchNorm + ReLU); dst may then be None.  mask_activated: mask_x is such an
    activation (kept-test mask_x > 0).  pro_act_dst: src-shaped tensor that receives the prologue's result
    pro(src) (combat_conv_args.pro_act_dst: DMA-staged 3x3 kernel with a per-channel prologue)."""
    a = ConvArgs()
    a.N, a.H, a.W, a.C = src.shape
    _, a.P, a.Q, a.K = (dst if dst is not None else act_dst).shape
    a.R = a.S = pc.R
    a.stride, a.pad, a.mode = pc.stride, pc.pad, mode
    a.src, a.dst = src.data_ptr(), _p(dst)
    if act_dst is not None:
        assert act is not None and act.group_stride == 0
        a.act_dst, a.act_scale, a.act_shift, a.act_slope = act_dst.data_ptr(), _p(act.scale), _p(act.shift), act.slope
    a.mask_activated = int(mask_activated)
    if mode == 0:
        a.wpack, a.kpad, a.rows_pad = pc.wf.data_ptr(), pc.kpad_f, pc.rows_f
    else:
        a.wpack, a.kpad, a.rows_pad = pc.wd.data_ptr(), pc.kpad_d, pc.rows_d
    if pro is not None:
        a.pro_scale, a.pro_shift = _p(pro.scale), _p(pro.shift)
        a.pro_group_stride, a.pro_act, a.pro_slope = pro.group_stride, int(pro.act), pro.slope
    if pro_act_dst is not None:
        assert pro is not None and tuple(pro_act_dst.shape) == tuple(src.shape)
        a.pro_act_dst = pro_act_dst.data_ptr()
    a.bias, a.add_pre, a.mask_x, a.add_post = _p(bias), _p(add_pre), _p(mask_x), _p(add_post)
    if mask is not None:
        a.mask_scale, a.mask_shift = _p(mask.scale), _p(mask.shift)
        a.mask_group_stride, a.mask_slope = mask.group_stride, mask.slope
    a.mask_mul_scale = int(mask_mul_scale)
    a.tanh_out = int(tanh_out)
    a.stats_kind, a.stats = stats_kind, _p(stats)
    a.xh_mean, a.xh_rstd = _p(xh_mean), _p(xh_rstd)
    a.tile = tile
    if shortcut is not None:    # (dY of the block's 1x1 / stride-2 shortcut, its PackedConv): second reduction source
        src2, pc2 = shortcut
        assert mode == 1 and tuple(src2.shape) == tuple(src.shape)
        a.src2, a.wpack2, a.kpad2, a.rows_pad2 = src2.data_ptr(), pc2.wd.data_ptr(), pc2.kpad_d, pc2.rows_d
    if workspace is not None:   # scratch for a split reduction (skinny layers); set before any layout query
        need = int(lib.combat_conv_workspace_bytes(ctypes.byref(a)))
        if 0 < need <= workspace.numel() * workspace.element_size():
            a.workspace, a.workspace_bytes = workspace.data_ptr(), workspace.numel() * workspace.element_size()
    # the struct holds raw pointers: keep every tensor alive as long as the struct is
    a._keepalive = (src, dst, pc, pro, bias, add_pre, mask_x, mask, add_post, stats, xh_mean, xh_rstd, act_dst, act,
                    workspace, shortcut, pro_act_dst)
    return a


def shortcut_fusable(dy, dx, pc: PackedConv, pc_sc: PackedConv) -> bool:
    """Can the input gradient of `pc` (3x3 / stride 2) take the block's 1x1 / stride-2 shortcut `pc_sc` along as a second
    reduction source (combat_conv_args.src2)?  True when a kernel takes such a launch."""
    if pc_sc.R != 1 or pc_sc.stride != 2 or pc_sc.wd is None:
        return False
    a = conv_args(dy, dx, pc, 1, shortcut=(dy, pc_sc))
    return lib.combat_conv_pick_tile(ctypes.byref(a)) > 0


def conv_tile_granule(a: ConvArgs):
    tile = lib.combat_conv_pick_tile(ctypes.byref(a))
    return tile, lib.combat_conv_stats_granule(tile)


def conv_stats_layout(a: ConvArgs):
    """(rows, rows_per_image) of the statistics array this launch writes (rows_per_image == 0:
    rows are not aligned to images)."""
    rows, rpi = ctypes.c_int32(0), ctypes.c_int32(0)
    check(lib.combat_conv_stats_layout(ctypes.byref(a), ctypes.byref(rows), ctypes.byref(rpi)),
          "combat_conv_stats_layout")
    return rows.value, rpi.value


def conv_launch(a: ConvArgs) -> None:
    check(lib.combat_conv_gemm(ctypes.byref(a), _stream()), "combat_conv_gemm",
          "mode=%d src=%dx%dx%dx%d dst=%dx%dx%d R=%d stride=%d" % (a.mode, a.N, a.H, a.W, a.C, a.P, a.Q, a.K, a.R,
                                                                   a.stride))


def conv_wgrad(src, dy, pc: PackedConv, dw, *, pro: Optional[Affine] = None, split=0, workspace=None) -> None:
    """dw (fp32 [K][taps][c_real], pre-zeroed) += wgrad.  workspace=True allocates the scratch the launch
    can use (partial sums by plain stores + one reduction instead of fp32 atomics); a tensor is used as is."""
    a = WgradArgs()
    a.N, a.H, a.W, a.C = src.shape
    _, a.P, a.Q, a.K = dy.shape
    a.R = a.S = pc.R
    a.stride, a.pad = pc.stride, pc.pad
    a.src, a.dy, a.dw, a.k_real, a.c_real = src.data_ptr(), dy.data_ptr(), dw.data_ptr(), pc.K, pc.c_real
    if pro is not None:
        a.pro_scale, a.pro_shift = _p(pro.scale), _p(pro.shift)
        a.pro_group_stride, a.pro_act, a.pro_slope = pro.group_stride, int(pro.act), pro.slope
    a.split = split
    if workspace is True:
        nbytes = int(lib.combat_conv_wgrad_workspace_bytes(ctypes.byref(a)))
        workspace = torch.empty(max(nbytes, 4), dtype=torch.uint8, device=src.device) if nbytes else None
    if workspace is not None:
        a.workspace, a.workspace_bytes = workspace.data_ptr(), workspace.numel() * workspace.element_size()
    check(lib.combat_conv_wgrad(ctypes.byref(a), _stream()), "combat_conv_wgrad",
          "src=%s dy=%s" % (tuple(src.shape), tuple(dy.shape)))


def norm_scratch_bytes(groups: int, c: int) -> int:
    return int(lib.combat_norm_scratch_bytes(groups, c))


def norm_finalize(partials, groups, rows_per_group, c, count, *, eps=1e-5, gamma=None, beta=None, mean=None,
                  rstd=None, scale=None, shift=None, running_mean=None, running_var=None, momentum=0.1, nbt=None,
                  scratch=None) -> None:
    check(lib.combat_norm_finalize(partials.data_ptr(), groups, rows_per_group, c, float(count), eps, _p(gamma),
                                   _p(beta), _p(mean), _p(rstd), _p(scale), _p(shift), _p(running_mean),
                                   _p(running_var), momentum, _p(nbt), _p(scratch),
                                   0 if scratch is None else scratch.numel() * scratch.element_size(), _stream()),
          "combat_norm_finalize", "groups=%d rows=%d C=%d" % (groups, rows_per_group, c))


def norm_bwd_finalize(partials, groups, rows_per_group, c, count, *, gamma, mean, rstd, ca, cb, cc, dgamma=None,
                      dbeta=None, scratch=None) -> None:
    check(lib.combat_norm_bwd_finalize(partials.data_ptr(), groups, rows_per_group, c, float(count), _p(gamma),
                                       mean.data_ptr(), rstd.data_ptr(), ca.data_ptr(), cb.data_ptr(), cc.data_ptr(),
                                       _p(dgamma), _p(dbeta), _p(scratch),
                                       0 if scratch is None else scratch.numel() * scratch.element_size(), _stream()),
          "combat_norm_bwd_finalize", "groups=%d rows=%d C=%d" % (groups, rows_per_group, c))


def bn_eval_fold(gamma, beta, rm, rv, scale, shift, eps=1e-5) -> None:
    check(lib.combat_bn_eval_fold(gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(), rv.data_ptr(), eps,
                                  gamma.numel(), scale.data_ptr(), shift.data_ptr(), _stream()), "combat_bn_eval_fold")


def group_stats(x, parts, rows_per_part, partials) -> None:
    check(lib.combat_group_stats(x.data_ptr(), parts, rows_per_part, x.shape[-1], partials.data_ptr(), _stream()),
          "combat_group_stats", "parts=%d rows=%d C=%d" % (parts, rows_per_part, x.shape[-1]))


def group_stats_bwd(dz, x, parts, rows_per_part, parts_per_image, xh_mean, xh_rstd, partials) -> None:
    check(lib.combat_group_stats_bwd(dz.data_ptr(), x.data_ptr(), parts, rows_per_part, x.shape[-1], parts_per_image,
                                     xh_mean.data_ptr(), xh_rstd.data_ptr(), partials.data_ptr(), _stream()),
          "combat_group_stats_bwd", "parts=%d rows=%d C=%d" % (parts, rows_per_part, x.shape[-1]))


def norm_bwd_apply(dz, x, dx, ca, cb, cc, *, rows_per_group=0, add=None) -> None:
    c = x.shape[-1]
    rows = x.numel() // c
    check(lib.combat_norm_bwd_apply(dz.data_ptr(), x.data_ptr(), _p(add), dx.data_ptr(), rows, c, rows_per_group,
                                    int(rows_per_group > 0), ca.data_ptr(), cb.data_ptr(), cc.data_ptr(), _stream()),
          "combat_norm_bwd_apply", "rows=%d C=%d" % (rows, c))


def unet_up_fwd(y, sy, ty, out, skip=None, ss=None, ts=None) -> None:
    n, h, w, c = y.shape
    check(lib.combat_unet_up_fwd(y.data_ptr(), sy.data_ptr(), ty.data_ptr(), _p(skip), _p(ss), _p(ts), n, h, w, c,
                                 out.data_ptr(), _stream()), "combat_unet_up_fwd", str(tuple(y.shape)))


def unet_up_bwd(d_out, out, du) -> None:
    n, h, w, c = du.shape
    check(lib.combat_unet_up_bwd(d_out.data_ptr(), out.data_ptr(), n, h, w, c, du.data_ptr(), _stream()),
          "combat_unet_up_bwd", str(tuple(du.shape)))


def trigger_fwd(x, noise, p_mat, k1, noise_rate, out, out_c8=None, mse_partial=None, src_index=None) -> None:
    """src_index (int32 [n_out], device): output image i is built from row src_index[i] of x and noise."""
    hw = x.shape[2]
    n = x.shape[0] if src_index is None else src_index.numel()
    check(lib.combat_trigger_fwd(x.data_ptr(), noise.data_ptr(), p_mat.data_ptr(), k1.data_ptr(), noise_rate, n, hw,
                                 _p(src_index), out.data_ptr(), _p(out_c8), _p(mse_partial), _stream()),
          "combat_trigger_fwd", str(tuple(x.shape)))


def trigger_bwd(x, noise, p_mat, k1, noise_rate, d_out, out, l2_scale, d_noise, pre_tanh=False, d_out2=None) -> None:
    n, _, hw, _ = x.shape
    check(lib.combat_trigger_bwd(x.data_ptr(), noise.data_ptr(), p_mat.data_ptr(), k1.data_ptr(), noise_rate, n, hw,
                                 _p(d_out), _p(d_out2), _p(out), l2_scale, int(pre_tanh), d_noise.data_ptr(), _stream()),
          "combat_trigger_bwd", str(tuple(x.shape)))


def augment_fwd(x, n, hw, out_c8, params=None, index=None, out_f32=None) -> None:
    check(lib.combat_augment_fwd(x.data_ptr(), _p(index), _p(params), n, hw, out_c8.data_ptr(), _p(out_f32),
                                 _stream()), "combat_augment_fwd", "n=%d hw=%d" % (n, hw))


def augment_bwd(d_c8, n, hw, d_x, params=None, accumulate=False) -> None:
    check(lib.combat_augment_bwd(d_c8.data_ptr(), d_c8.shape[-1], _p(params), n, hw, d_x.data_ptr(),
                                 int(accumulate), _stream()),
          "combat_augment_bwd", "n=%d hw=%d" % (n, hw))


def head_fwd(feat, weight, bias, logits, *, targets=None, loss_weight=1.0, pooled=None, loss_sum=None, correct=None,
             targets2=None, correct2=None) -> None:
    n, hw, _, c = feat.shape
    check(lib.combat_head_fwd(feat.data_ptr(), n, hw, c, weight.data_ptr(), bias.data_ptr(), weight.shape[0],
                              _p(targets), loss_weight, _p(pooled), logits.data_ptr(), _p(loss_sum), _p(correct),
                              _p(targets2), _p(correct2), _stream()), "combat_head_fwd", str(tuple(feat.shape)))


def head_bwd(pooled, n, hw, c, weight, logits, targets, loss_weight, dlogits, d_feat=None, dw=None, db=None) -> None:
    check(lib.combat_head_bwd(_p(pooled), n, hw, c, weight.data_ptr(), weight.shape[0], logits.data_ptr(),
                              targets.data_ptr(), loss_weight, dlogits.data_ptr(), _p(d_feat), _p(dw), _p(db),
                              _stream()), "combat_head_bwd", "n=%d hw=%d C=%d" % (n, hw, c))


def sgd_nesterov(ptrs, sizes, count, max_size, lr, momentum, weight_decay, grad_scale, first_step) -> None:
    check(lib.combat_sgd_nesterov(ptrs.data_ptr(), sizes.data_ptr(), count, max_size, lr, momentum, weight_decay,
                                  grad_scale, int(first_step), _stream()), "combat_sgd_nesterov", "count=%d" % count)


def image_to_c8(x, out_c8) -> None:
    n, _, hw, _ = x.shape
    check(lib.combat_image_to_c8(x.data_ptr(), n, hw, out_c8.data_ptr(), _stream()), "combat_image_to_c8")


def nhwc_to_nchw_f32(x, c, out) -> None:
    n, h, w, cc = x.shape
    check(lib.combat_nhwc_to_nchw_f32(x.data_ptr(), n, h, w, cc, c, out.data_ptr(), _stream()),
          "combat_nhwc_to_nchw_f32")


def nchw_to_nhwc_bf16(x, out) -> None:
    n, c, h, w = x.shape
    check(lib.combat_nchw_to_nhwc_bf16(x.data_ptr(), n, c, h, w, out.shape[-1], out.data_ptr(), _stream()),
          "combat_nchw_to_nhwc_bf16")


def colsum(x, c_out, out) -> None:
    c = x.shape[-1]
    check(lib.combat_colsum(x.data_ptr(), x.numel() // c, c, c_out, out.data_ptr(), _stream()), "combat_colsum")


def maxpool2(x, out) -> None:
    n, h, w, c = x.shape
    check(lib.combat_maxpool2(x.data_ptr(), n, h, w, c, out.data_ptr(), _stream()), "combat_maxpool2")


def elu_affine(x, scale, shift, out) -> None:
    c = x.shape[-1]
    check(lib.combat_elu_affine(x.data_ptr(), x.numel() // c, c, scale.data_ptr(), shift.data_ptr(), out.data_ptr(),
                                _stream()), "combat_elu_affine")


def dct_u8(x, d_mat, out_c8) -> None:
    n, _, hw, _ = x.shape
    check(lib.combat_dct_u8(x.data_ptr(), d_mat.data_ptr(), n, hw, out_c8.data_ptr(), _stream()), "combat_dct_u8")


def linear_nhwc(x, weight, bias, logits) -> None:
    n, h, w, c = x.shape
    check(lib.combat_linear_nhwc(x.data_ptr(), n, h, w, c, weight.data_ptr(), bias.data_ptr(), weight.shape[0],
                                 logits.data_ptr(), _stream()), "combat_linear_nhwc")
