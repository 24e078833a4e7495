"""bf16-emulating references for the network engines (test infrastructure).

Same dataflow as oracle/combat_oracle.py, but every tensor the HIP path stores in bf16 is rounded
at the same point (straight-through for autograd), conv weights are the bf16 operands, and
normalisation statistics are taken from the rounded tensors -- exactly what the kernels do.  The
engines must match these tightly (only fp32 summation order and the bf16 rounding of *gradient*
tensors differ); the distance between these emulations and the fp32 goldens is the inherent bf16
error of the design, reported by the tests that compare against the goldens.
"""
import torch
import torch.nn.functional as F

bf16 = torch.bfloat16

ROUND = True   # False: no bf16 rounding anywhere -- the emulation must then BE the fp32 oracle (dataflow pin:
#                tests/test_oracle_golden.py::test_bf16_emulation_without_rounding_is_the_oracle)


def q(t, force=None, key=None):
    """Round to bf16, identity gradient.  With ``force[key]`` given (the tensor the HIP engine
    actually stored, NCHW fp32) the *value* is replaced by it while the gradient still flows to
    ``t``: "teacher forcing", so that activation masks and normalisation statistics downstream are
    exactly the engine's and a gradient comparison is not dominated by mask flips."""
    if force is not None and key in force:
        return t + (force[key] - t).detach()
    if not ROUND:
        return t
    return t + (t.to(bf16).float() - t).detach()


def qw(p, name):
    return q(p[name])


def _bn_affine(p, name, x, train, eps=1e-5):
    c = x.shape[1]
    if train:
        mean = x.mean((0, 2, 3))
        var = x.var((0, 2, 3), unbiased=False)
        with torch.no_grad():   # nn.BatchNorm2d's running-statistics update (momentum 0.1, unbiased var)
            cnt = x.numel() // c
            p[name + ".running_mean"].mul_(0.9).add_(0.1 * mean)
            p[name + ".running_var"].mul_(0.9).add_(0.1 * var * cnt / max(cnt - 1, 1))
            if name + ".num_batches_tracked" in p:
                p[name + ".num_batches_tracked"] += 1
    else:
        mean, var = p[name + ".running_mean"], p[name + ".running_var"]
    rstd = 1 / torch.sqrt(var + eps)
    scale = p[name + ".weight"] * rstd
    shift = p[name + ".bias"] - mean * scale
    return x * scale.view(1, c, 1, 1) + shift.view(1, c, 1, 1)


def preact_forward_emu(p, x, train, force=None):
    """force: optional {engine slot buffer name: NCHW fp32 tensor} (see :func:`q`)."""
    t = q(F.conv2d(x, qw(p, "conv1.weight"), padding=1), force, "stem")
    b = 0
    for layer, stride0 in ((1, 1), (2, 2), (3, 2), (4, 2)):
        for blk in (0, 1):
            pre = "layer%d.%d." % (layer, blk)
            stride = stride0 if blk == 0 else 1
            a1 = q(F.relu(_bn_affine(p, pre + "bn1", t, train)))
            sck = pre + "shortcut.0.weight"
            sc = q(F.conv2d(a1, qw(p, sck), stride=stride), force, "b%d.sc" % b) if sck in p else t
            y1 = q(F.conv2d(a1, qw(p, pre + "conv1.weight"), stride=stride, padding=1), force, "b%d.y1" % b)
            a2 = q(F.relu(_bn_affine(p, pre + "bn2", y1, train)))
            t = q(F.conv2d(a2, qw(p, pre + "conv2.weight"), padding=1) + sc, force, "b%d.out" % b)
            b += 1
    feat = F.avg_pool2d(t, 4).flatten(1)
    return F.linear(feat, p["linear.weight"], p["linear.bias"])


def _bn_eval_tables(p, name, eps=1e-5):
    scale = p[name + ".weight"] / torch.sqrt(p[name + ".running_var"] + eps)
    return scale, p[name + ".bias"] - p[name + ".running_mean"] * scale


def resnet_forward_emu(p, x, train, force=None):
    """ResNet18 (post-activation BasicBlocks) as combat_amd.engine.ResNetEngine computes it.  Train mode:
    raw conv outputs stored in bf16, batch statistics from them, relu(bn(.)) (+ residual) stored in bf16.
    Eval mode: bn1 / stem norm applied to the bf16-rounded conv output by the producer's epilogue; bn2 and
    the shortcut norm folded into bf16 operands (rounding of scale * w), their shifts added as biases."""
    f = force
    if train:
        y0 = q(F.conv2d(x, qw(p, "conv1.weight"), padding=1), f, "stem")
        cur = q(F.relu(_bn_affine(p, "bn1", y0, True)), f, "stem.a")
    else:
        s0, t0 = _bn_eval_tables(p, "bn1")
        y0 = q(F.conv2d(x, qw(p, "conv1.weight"), padding=1))
        cur = q(F.relu(y0 * s0.view(1, -1, 1, 1) + t0.view(1, -1, 1, 1)), f, "stem.a")
    b = 0
    for layer, stride0 in ((1, 1), (2, 2), (3, 2), (4, 2)):
        for blk in (0, 1):
            pre = "layer%d.%d." % (layer, blk)
            stride = stride0 if blk == 0 else 1
            sck = pre + "shortcut.0.weight"
            if train:
                y1 = q(F.conv2d(cur, qw(p, pre + "conv1.weight"), stride=stride, padding=1), f, "b%d.y1" % b)
                a1 = q(F.relu(_bn_affine(p, pre + "bn1", y1, True)), f, "b%d.a1" % b)
                y2 = q(F.conv2d(a1, qw(p, pre + "conv2.weight"), padding=1), f, "b%d.y2" % b)
                if sck in p:
                    ys = q(F.conv2d(cur, qw(p, sck), stride=stride), f, "b%d.ys" % b)
                    scv = _bn_affine(p, pre + "shortcut.1", ys, True)
                else:
                    scv = cur
                cur = q(F.relu(_bn_affine(p, pre + "bn2", y2, True) + scv), f, "b%d.out" % b)
            else:
                s1, t1 = _bn_eval_tables(p, pre + "bn1")
                y1 = q(F.conv2d(cur, qw(p, pre + "conv1.weight"), stride=stride, padding=1))
                a1 = q(F.relu(y1 * s1.view(1, -1, 1, 1) + t1.view(1, -1, 1, 1)), f, "b%d.a1" % b)
                if sck in p:
                    ss, ts = _bn_eval_tables(p, pre + "shortcut.1")
                    scv = q(F.conv2d(cur, q(p[sck] * ss.view(-1, 1, 1, 1)), stride=stride) + ts.view(1, -1, 1, 1), f,
                            "b%d.scv" % b)
                else:
                    scv = cur
                s2, t2 = _bn_eval_tables(p, pre + "bn2")
                v = q(F.conv2d(a1, q(p[pre + "conv2.weight"] * s2.view(-1, 1, 1, 1)), padding=1) + t2.view(1, -1, 1, 1) + scv)
                cur = q(F.relu(v), f, "b%d.out" % b)
            b += 1
    feat = F.avg_pool2d(cur, 4).flatten(1)
    return F.linear(feat, p["linear.weight"], p["linear.bias"])


def unet_forward_emu(p, x, taps=None, force=None):
    """`taps` (dict) receives every stored raw conv output, keyed like the engine's slot buffers;
    `force`: optional {engine slot buffer name: NCHW fp32 tensor} (see :func:`q`)."""
    f = force
    lr = lambda t: F.leaky_relu(t, 0.2)
    inorm = lambda t: F.instance_norm(t, eps=1e-5)

    def conv(name, t, stride=1, bias=False):
        return F.conv2d(t, qw(p, name + ".weight"), p[name + ".bias"] if bias else None, stride=stride, padding=1)

    up = lambda t: F.interpolate(t, scale_factor=2, mode="bilinear", align_corners=False)
    t00 = q(conv("conv0_0", x, 2, True), f, "t.conv0_0")
    t01 = q(conv("conv0_1", q(lr(t00))), f, "t.conv0_1")
    t10 = q(conv("conv1_0", q(lr(inorm(t01))), 2), f, "t.conv1_0")
    t11 = q(conv("conv1_1", q(lr(inorm(t10)))), f, "t.conv1_1")
    t20 = q(conv("conv2_0", q(lr(inorm(t11))), 2), f, "t.conv2_0")
    t21 = q(conv("conv2_1", q(lr(inorm(t20)))), f, "t.conv2_1")
    t30 = q(conv("conv3_0", q(lr(inorm(t21))), 2), f, "t.conv3_0")
    t31 = q(conv("conv3_1", q(lr(inorm(t30)))), f, "t.conv3_1")
    u3 = q(lr(up(inorm(t31))), f, "up3")
    tu31 = q(conv("upconv3_1", u3), f, "t.upconv3_1")
    tu30 = q(conv("upconv3_0", q(lr(inorm(tu31)))), f, "t.upconv3_0")
    u2 = q(lr(up(inorm(tu30) + lr(inorm(t21)))), f, "up2")
    tu21 = q(conv("upconv2_1", u2), f, "t.upconv2_1")
    tu20 = q(conv("upconv2_0", q(lr(inorm(tu21)))), f, "t.upconv2_0")
    u1 = q(lr(up(inorm(tu20) + lr(inorm(t11)))), f, "up1")
    tu11 = q(conv("upconv1_1", u1), f, "t.upconv1_1")
    tu10 = q(conv("upconv1_0", q(lr(inorm(tu11)))), f, "t.upconv1_0")
    u0 = q(lr(up(inorm(tu10) + lr(inorm(t01)))), f, "up0")
    tu01 = q(conv("upconv0_1", u0), f, "t.upconv0_1")
    if taps is not None:
        taps.update({"conv0_0": t00, "conv0_1": t01, "conv1_0": t10, "conv1_1": t11, "conv2_0": t20, "conv2_1": t21,
                     "conv3_0": t30, "conv3_1": t31, "upconv3_1": tu31, "upconv3_0": tu30, "upconv2_1": tu21,
                     "upconv2_0": tu20, "upconv1_1": tu11, "upconv1_0": tu10, "upconv0_1": tu01})
    return q(torch.tanh(conv("upconv0_0", q(lr(inorm(tu01))), 1, True)), f, "noise")
