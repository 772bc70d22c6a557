"""GPU parity of the backward kernels (through the C ABI) against torch autograd in fp32 on 16-bit-rounded operands.

Tolerances: data gradients are 16-bit outputs (1e-3 / 8e-3 of the gradient's scale, as for the forward ops); weight
and bias gradients are fp32 sums of products of 16-bit operands accumulated in fp32 (order differs: 2e-4 of scale)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from climategan_amd import fill

pytestmark = pytest.mark.gpu

DTYPES = [torch.float16, torch.bfloat16]
TOL = {torch.float16: 1e-3, torch.bfloat16: 8e-3}


def q(a, dt):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dt).float()


def to_nhwc(x_cpu, dt):
    from climategan_amd import ops
    return ops.nchw_to_nhwc(x_cpu.cuda(), dt)


def back(y):
    from climategan_amd import ops
    return ops.nhwc_to_nchw(y).cpu()


def rel_err(got, ref):
    return (got - ref).abs().max().item() / max(ref.abs().max().item(), 1e-12)


BWD_CASES = [
    # cin, cout, k, stride, pad, dil, B, H, W
    (4, 64, 4, 2, 1, 1, 2, 32, 40),        # PatchGAN first conv
    (64, 128, 4, 2, 1, 1, 2, 24, 20),      # PatchGAN stride-2 4x4
    (128, 128, 4, 1, 1, 1, 2, 9, 11),      # PatchGAN stride-1 4x4 (H-1)
    (128, 1, 4, 1, 1, 1, 2, 9, 11),        # PatchGAN output conv
    (20, 20, 3, 1, 1, 1, 2, 17, 19),       # Painter main conv, ragged
    (40, 20, 1, 1, 0, 1, 2, 16, 16),       # Painter 1x1 shortcut
    (256, 256, 3, 1, 2, 2, 1, 24, 24),     # ResNet dilated 3x3 (wide-layer kernels on the dgrad)
    (64, 256, 1, 1, 0, 1, 2, 20, 20),      # bottleneck expand
    (128, 128, 3, 2, 1, 1, 1, 33, 35),     # ResNet strided 3x3, odd size (unreached rows -> zero grad)
    (3, 64, 7, 2, 3, 1, 1, 40, 48),        # ResNet stem
    (11, 64, 4, 2, 1, 1, 2, 32, 32),       # ADVENT FC discriminator first conv
    (64, 128, 1, 2, 0, 1, 1, 17, 20),      # ResNet 1x1 stride-2 shortcut (three of the four parity classes have no tap)
    (256, 512, 4, 2, 1, 1, 1, 12, 10),     # PatchGAN deep stride-2 4x4 (dgrad by parity classes, 4 k-steps per tap)
    (3, 128, 3, 1, 1, 1, 2, 20, 24),       # SPADE mlp_shared on the 3-channel cond (wgrad: 8 taps folded per N tile)
    (32, 80, 3, 1, 1, 1, 1, 18, 22),       # cin_s = 32: 2 taps per N tile
    (16, 8, 3, 1, 1, 1, 2, 16, 16),        # mask decoder tail: 4 taps per N tile
]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("case", BWD_CASES)
def test_conv2d_backward(dt, case):
    from climategan_amd import ops
    cin, cout, k, stride, pad, dil, B, H, W = case
    x = q(fill.uniform((B, cin, H, W), 2100 + cin), dt).requires_grad_(True)
    bound = 1.0 / np.sqrt(cin * k * k)
    w = q(fill.uniform((cout, cin, k, k), 2200 + cout, -bound, bound), dt).requires_grad_(True)
    b = torch.from_numpy(fill.uniform((cout,), 2300 + cout)).requires_grad_(True)
    y = F.conv2d(x, w, b, stride=stride, padding=pad, dilation=dil)
    dy = q(fill.uniform(tuple(y.shape), 2400 + cout), dt)
    y.backward(dy)
    dyg = to_nhwc(dy, dt)
    # data gradient
    dx = ops.conv2d_bwd_data(dyg, w.detach().cuda(), (B, H, W), stride=stride, pad=pad, dilation=dil)
    assert dx.t.shape == (B, H, W, ops.cs8(cin))
    e = rel_err(back(dx), x.grad)
    assert e <= TOL[dt], "dgrad %s: rel err %.3g" % (case, e)
    if ops.cs8(cin) != cin:
        assert dx.t[..., cin:].abs().max().item() == 0
    # weight / bias gradient
    dw, db = ops.conv2d_bwd_weight(to_nhwc(x.detach(), dt), dyg, tuple(w.shape), stride=stride, pad=pad, dilation=dil)
    e = rel_err(dw.cpu(), w.grad)
    assert e <= 2e-4, "wgrad %s: rel err %.3g" % (case, e)
    e = rel_err(db.cpu(), b.grad)
    assert e <= 2e-4, "bias grad %s: rel err %.3g" % (case, e)
    # accumulation semantics: a second call adds (this one through the workspace-free atomic path)
    dw2, _ = ops.conv2d_bwd_weight(to_nhwc(x.detach(), dt), dyg, tuple(w.shape), stride=stride, pad=pad, dilation=dil,
                                   want_bias=False, dw=dw.clone(), use_workspace=False)
    assert rel_err(dw2.cpu(), 2 * w.grad) <= 2e-4


def test_dgrad_with_sigma():
    from climategan_amd import ops
    dt = torch.float16
    x = q(fill.uniform((1, 16, 12, 12), 1), dt).requires_grad_(True)
    w = q(fill.uniform((32, 16, 3, 3), 2, -0.1, 0.1), dt)
    sigma = torch.tensor([1.7])
    y = F.conv2d(x, w / sigma, None, padding=1)
    dy = q(fill.uniform(tuple(y.shape), 3), dt)
    y.backward(dy)
    dx = ops.conv2d_bwd_data(to_nhwc(dy, dt), w.cuda(), (1, 12, 12), pad=1, sigma=sigma.cuda())
    assert rel_err(back(dx), x.grad) <= 2e-3     # w / sigma is rounded to fp16 once more than in the reference


def test_dgrad_entry_refuses_reflect_descriptor():
    import ctypes as C
    from climategan_amd import _lib, ops
    lib = _lib.load()
    d = ops._conv_desc(_lib.CGAN_F16, 1, 8, 8, 8, 8, 3, 3, 1, 1, 1, _lib.PAD_REFLECT)
    assert lib.cgan_conv2d_dgrad_packed_weight_bytes(C.byref(d)) == 0
    assert b"zero padding" in lib.cgan_last_error()


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("c,h,w,act", [(64, 20, 24, "lrelu"), (20, 33, 17, "none"), (512, 9, 9, "lrelu")])
def test_instnorm_act_backward(dt, c, h, w, act):
    from climategan_amd import ops
    B = 2
    x = q(fill.uniform((B, c, h, w), 3100 + c, -2, 2), dt).requires_grad_(True)
    y = F.instance_norm(x, eps=1e-5)
    out = F.leaky_relu(y, 0.2) if act == "lrelu" else y
    dy = q(fill.uniform((B, c, h, w), 3200 + c), dt)
    out.backward(dy)
    a = ops.ACT_LRELU if act == "lrelu" else ops.ACT_NONE
    xg = to_nhwc(x.detach(), dt)
    mean, rstd = ops.instnorm_stats(xg)
    outg = ops.norm_act_apply(xg, mean, rstd, act=a)
    dx = ops.instnorm_act_bwd(outg, to_nhwc(dy, dt), rstd, act=a)
    # the kernel recovers y from the 16-bit output: allow twice the forward-op tolerance
    assert rel_err(back(dx), x.grad) <= 2 * TOL[dt]
    if ops.cs8(c) != c:
        assert dx.t[..., c:].abs().max().item() == 0


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("act", ["relu", "lrelu", "tanh", "sigmoid"])
def test_act_backward(dt, act):
    from climategan_amd import ops
    x = q(fill.uniform((2, 24, 9, 11), 3300, -2, 2), dt).requires_grad_(True)
    f = {"relu": F.relu, "lrelu": lambda v: F.leaky_relu(v, 0.2), "tanh": torch.tanh, "sigmoid": torch.sigmoid}[act]
    a = {"relu": ops.ACT_RELU, "lrelu": ops.ACT_LRELU, "tanh": ops.ACT_TANH, "sigmoid": ops.ACT_SIGMOID}[act]
    out = f(x)
    dy = q(fill.uniform((2, 24, 9, 11), 3301), dt)
    out.backward(dy)
    outq = q(out.detach().numpy(), dt)
    dx = ops.act_bwd(to_nhwc(outq, dt), to_nhwc(dy, dt), a)
    assert rel_err(back(dx), x.grad) <= 4 * TOL[dt]     # derivative evaluated at the 16-bit-rounded output


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("target", [1.0, 0.0, 0.87])
def test_bce_logits_loss_and_grad(dt, target):
    from climategan_amd import ops
    x = q(fill.uniform((2, 1, 19, 23), 3400, -4, 4), dt).requires_grad_(True)
    loss = F.binary_cross_entropy_with_logits(x, torch.full_like(x, target))
    (loss * 0.5).backward()
    acc = torch.zeros(1, device="cuda")
    n = x.numel()
    dx = ops.bce_logits(to_nhwc(x.detach(), dt), target, 0.5 / n, acc)
    assert abs(acc.item() - 0.5 * loss.item()) <= 1e-5 * max(1.0, abs(loss.item()))
    got = back(dx)
    assert rel_err(got, x.grad) <= TOL[dt]
    assert dx.t[..., 1:].abs().max().item() == 0
    ops.bce_logits(to_nhwc(x.detach(), dt), target, 0.5 / n, acc, want_grad=False)      # accumulates
    assert abs(acc.item() - loss.item()) <= 2e-5 * max(1.0, abs(loss.item()))


@pytest.mark.parametrize("dt", DTYPES)
def test_l1_loss_and_grad(dt):
    from climategan_amd import ops
    a = q(fill.uniform((2, 16, 9, 7), 3500), dt).requires_grad_(True)
    b = q(fill.uniform((2, 16, 9, 7), 3501), dt)
    b[0, 0, 0, 0] = a.detach()[0, 0, 0, 0]                     # sign(0) = 0
    loss = F.l1_loss(a, b) * 10.0 / 3
    loss.backward()
    acc = torch.zeros(1, device="cuda")
    da = ops.l1_loss(to_nhwc(a.detach(), dt), to_nhwc(b, dt), 10.0 / 3 / a.numel(), acc)
    assert abs(acc.item() - loss.item()) <= 1e-5 * max(1.0, abs(loss.item()))
    assert rel_err(back(da), a.grad) <= TOL[dt]


def test_spectral_norm_backward():
    """Gradient w.r.t. w_bar of L(w_bar / sigma), sigma = u^T w_bar v with u, v constants (norms.py:107-112)."""
    from climategan_amd import ops
    rows, cols = 24, 16 * 9
    w_bar = torch.from_numpy(fill.uniform((rows, cols), 3600)).requires_grad_(True)
    u = torch.from_numpy(fill.uniform((rows,), 3601))
    v = torch.from_numpy(fill.uniform((cols,), 3602))
    g = torch.from_numpy(fill.uniform((rows, cols), 3603))
    sigma = u.dot(w_bar.mv(v))
    ((w_bar / sigma) * g).sum().backward()
    got = ops.spectral_norm_bwd(g.clone().cuda(), w_bar.detach().cuda(), u.cuda(), v.cuda(), sigma.detach().reshape(1).cuda())
    assert rel_err(got.cpu(), w_bar.grad) <= 1e-5


@pytest.mark.parametrize("dt", DTYPES)
def test_conv_fn_upsample_residual_autograd(dt):
    """autograd.ConvFn with the folded x2 upsample on the input and on the residual, LeakyReLU epilogue, spectral-norm
    sigma: all four gradients (x, residual, w_bar, bias) against torch autograd of the unfused fp32 expression."""
    from climategan_amd import ops
    from climategan_amd.autograd import ConvFn
    from oracle import cpu_ref
    B, cin, cout, H, W = 2, 24, 16, 6, 10
    x = q(fill.uniform((B, cin, H, W), 4100), dt).requires_grad_(True)
    res = q(fill.uniform((B, cout, H, W), 4101), dt).requires_grad_(True)
    w_bar = torch.from_numpy(fill.uniform((cout, cin, 3, 3), 4102, -0.2, 0.2)).requires_grad_(True)
    b = torch.from_numpy(fill.uniform((cout,), 4103)).requires_grad_(True)
    u = torch.from_numpy(fill.uniform((cout,), 4104)); u = u / u.norm()
    v = torch.from_numpy(fill.uniform((cin * 9,), 4105)); v = v / v.norm()
    sigma = u.dot(w_bar.reshape(cout, -1).mv(v))
    w_eff = q((w_bar / sigma).detach().numpy(), dt)                      # what the packed 16-bit weights hold
    xu = cpu_ref.nearest_resize(x, (2 * H, 2 * W))
    ru = cpu_ref.nearest_resize(res, (2 * H, 2 * W))
    # straight-through the 16-bit rounding of w: gradients are taken w.r.t. w_bar through w_bar / sigma
    w_used = w_bar / sigma + (w_eff - w_bar / sigma).detach()
    y = F.leaky_relu(F.conv2d(xu, w_used, b, padding=1) + ru, 0.2)
    dy = q(fill.uniform(tuple(y.shape), 4106), dt)
    y.backward(dy)

    xg = to_nhwc(x.detach(), dt).t.requires_grad_(True)
    rg = to_nhwc(res.detach(), dt).t.requires_grad_(True)
    wg = w_bar.detach().cuda().requires_grad_(True)
    bg = b.detach().cuda().requires_grad_(True)
    sig = sigma.detach().reshape(1).cuda()
    pw = ops.pack_conv_weight(wg.detach(), bg.detach(), dt, sig)
    cfg = dict(c_in=cin, stride=1, pad=1, dilation=1, act=ops.ACT_LRELU, slope=0.2, in_upsample=True,
               residual_upsample=True)
    out = ConvFn.apply(xg, wg, bg, rg, pw, cfg, (sig, u.cuda(), v.cuda()))
    assert rel_err(back(ops.NHWC(out.detach(), cout)), y.detach()) <= TOL[dt]
    out.backward(to_nhwc(dy, dt).t)
    assert rel_err(back(ops.NHWC(xg.grad, cin)), x.grad) <= 2 * TOL[dt]
    assert rel_err(back(ops.NHWC(rg.grad, cout)), res.grad) <= 2 * TOL[dt]
    assert rel_err(wg.grad.cpu(), w_bar.grad) <= 5e-3
    assert rel_err(bg.grad.cpu(), b.grad) <= 5e-3


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("C,H,W,ups,act", [(20, 12, 16, False, "lrelu"), (40, 8, 12, True, "none"), (16, 10, 10, True, "lrelu"),
                                           (10, 8, 8, False, "lrelu"), (12, 8, 12, False, "none"),   # pad channels past 2C; a straddling group
                                           # maps of 80 x 80 and up: the fused hidden-map backward (cgan_spade_hidden_bwd)
                                           (20, 80, 96, False, "lrelu"), (12, 88, 84, True, "none")])
def test_spade_fn_backward(dt, C, H, W, ups, act):
    """autograd.SpadeFn (fused forward, re-materialising backward) against torch autograd of the reference expression
    (norms.py:174-186): gradients of x and of the six mlp parameters."""
    from climategan_amd import ops
    from climategan_amd.autograd import SpadeFn
    from oracle import cpu_ref
    B = 2
    hs, ws = (H // 2, W // 2) if ups else (H, W)
    x = q(fill.uniform((B, C, hs, ws), 5100 + C, -2, 2), dt).requires_grad_(True)
    cond = q(fill.uniform((B, 3, 2 * H, 2 * W), 5101), dt)
    shapes = {"mlp_shared.0.weight": (128, 3, 3, 3), "mlp_shared.0.bias": (128,), "mlp_gamma.weight": (C, 128, 3, 3),
              "mlp_gamma.bias": (C,), "mlp_beta.weight": (C, 128, 3, 3), "mlp_beta.bias": (C,)}
    sd = {k: q(v, dt).requires_grad_(True) for k, v in fill.fill_state_dict(shapes, 5102).items()}
    xu = cpu_ref.nearest_resize(x, (H, W)) if ups else x
    seg = cpu_ref.nearest_resize(cond, (H, W))
    actv = F.relu(F.conv2d(seg, sd["mlp_shared.0.weight"], sd["mlp_shared.0.bias"], padding=1))
    gamma = F.conv2d(actv, sd["mlp_gamma.weight"], sd["mlp_gamma.bias"], padding=1)
    beta = F.conv2d(actv, sd["mlp_beta.weight"], sd["mlp_beta.bias"], padding=1)
    y = F.instance_norm(xu, eps=1e-5) * (1 + gamma) + beta
    if act == "lrelu":
        y = F.leaky_relu(y, 0.2)
    dy = q(fill.uniform(tuple(y.shape), 5103), dt)
    y.backward(dy)

    a = ops.ACT_LRELU if act == "lrelu" else ops.ACT_NONE
    xg = to_nhwc(x.detach(), dt)
    xt = xg.t.requires_grad_(True)
    condg = ops.nchw_to_nhwc(cond.cuda(), dt, cs=4)
    ps = {k: v.detach().cuda().requires_grad_(True) for k, v in sd.items()}
    names = ["mlp_shared.0.weight", "mlp_shared.0.bias", "mlp_gamma.weight", "mlp_gamma.bias", "mlp_beta.weight",
             "mlp_beta.bias"]
    pk = ops.pack_spade_weights(*[ps[k].detach() for k in names], dt)
    mean, rstd = ops.instnorm_stats(xg)
    cfg = dict(c=C, cond_c=3, act=a, slope=0.2, x_upsample=ups)
    out = SpadeFn.apply(xt, condg.t, mean, rstd, *[ps[k] for k in names], pk, cfg)
    assert rel_err(back(ops.NHWC(out.detach(), C)), y.detach()) <= TOL[dt]
    out.backward(to_nhwc(dy, dt).t)
    # 16-bit re-materialised hidden map / gamma / gradients: a few 16-bit roundings deep
    # (bf16: the instance-norm backward subtracts two means from 8-bit-mantissa values: up to 7 % on the 5x5 case)
    # (on the 80 x 80-and-up cases the fp16 bound is the instance-norm backward's, fused or not: CGAN_SPADE_FUSED_BWD=0 gives
    # the same 2.9e-2 -- dx does not depend on the hidden map's gradient path)
    if H * W < 6400:
        assert rel_err(back(ops.NHWC(xt.grad, C)), x.grad) <= (4e-3 if dt == torch.float16 else 1e-1)
    else:
        # 3e5 elements: a handful sit within one 16-bit rounding step of the LeakyReLU kink, where the 16-bit y has the other
        # sign than the fp32 one and the slope flips 1 <-> 0.2 (up to 0.6 of the largest gradient on ONE element, fused or not):
        # bound the 99.9th percentile instead of the maximum
        err = (back(ops.NHWC(xt.grad, C)) - x.grad).abs().flatten()
        assert torch.quantile(err, 0.999).item() <= (4e-2 if dt == torch.float16 else 1e-1) * x.grad.abs().max().item()
    for k in names:
        e = rel_err(ps[k].grad.cpu(), sd[k].grad)
        assert e <= (5e-3 if dt == torch.float16 else 8e-2), "%s: rel err %.3g" % (k, e)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("C,H,W,B", [(20, 96, 80, 2), (40, 83, 101, 1), (160, 80, 80, 1), (5, 112, 96, 3)])
def test_spade_hidden_backward_fused_equals_the_unfused_kernels(dt, C, H, W, B):
    """cgan_spade_hidden_bwd (round 5: data gradient of the gamma||beta conv + ReLU mask from a re-computed hidden tile +
    mlp_shared's weight / bias gradient in one kernel, reference norms.py:163-172 under autograd) against the three kernels it
    replaces on the same operands -- hidden map re-materialised, masked data gradient written as a 16-bit map, weight
    gradient over it: the same 16-bit roundings in the same places, so only the fp32 summation order differs -- and against
    torch's fp32 autograd of the same expression.  Ragged extents (tiles past the image), 1 .. 10 channel chunks."""
    from climategan_amd import ops
    dgb_f = q(fill.uniform((B, 2 * C, H, W), 7100 + C, -1, 1), dt)
    seg_f = q(fill.uniform((B, 3, H, W), 7101 + H), dt)
    w_sh = q(fill.uniform((128, 3, 3, 3), 7102, -0.4, 0.4), dt).requires_grad_(True)
    b_sh = torch.from_numpy(fill.uniform((128,), 7103, -0.2, 0.2)).requires_grad_(True)
    w_gb = q(fill.uniform((2 * C, 128, 3, 3), 7104 + C, -0.05, 0.05), dt)
    actv = F.relu(F.conv2d(seg_f, w_sh, b_sh, padding=1))
    gb = F.conv2d(actv, w_gb, None, padding=1)
    gb.backward(dgb_f)
    dgb = to_nhwc(dgb_f, dt)
    seg = to_nhwc(seg_f, dt)
    pw_sh = ops.pack_conv_weight(w_sh.detach().cuda(), b_sh.detach().cuda(), dt)
    dw, db = ops.spade_hidden_bwd(dgb, w_gb.cuda(), seg, pw_sh, C)
    a = ops.conv2d(seg, pw_sh, pad=1, act=ops.ACT_RELU)
    d_pre = ops.conv2d_bwd_data(dgb, w_gb.cuda(), (B, H, W), pad=1, relu_out=a)
    dw_u, db_u = ops.conv2d_bwd_weight(seg, d_pre, (128, 3, 3, 3), pad=1)
    errs = dict(fused_vs_unfused=(rel_err(dw.cpu(), dw_u.cpu()), rel_err(db.cpu(), db_u.cpu())),
                fused_vs_torch=(rel_err(dw.cpu(), w_sh.grad), rel_err(db.cpu(), b_sh.grad)),
                unfused_vs_torch=(rel_err(dw_u.cpu(), w_sh.grad), rel_err(db_u.cpu(), b_sh.grad)))
    print("spade_hidden_bwd %s C=%d %dx%d: %s" % (dt, C, H, W, errs))
    tol = 5e-3 if dt == torch.float16 else 4e-2
    assert max(errs["fused_vs_torch"]) <= tol, errs
    # same products, same 16-bit roundings, the same mask bit for bit (h is rounded to the 16-bit type before the sign test, as
    # the stored map would hold it): only the fp32 summation order differs
    assert max(errs["fused_vs_unfused"]) <= 2e-4, errs
    dw2, db2 = ops.spade_hidden_bwd(dgb, w_gb.cuda(), seg, pw_sh, C)
    assert torch.equal(dw, dw2) and torch.equal(db, db2), "deterministic"


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("C,H,W,ups,act", [(64, 12, 16, False, "lrelu"), (32, 16, 12, True, "lrelu"), (16, 10, 14, True, "none")])
def test_spade_batch_norm_training(dt, C, H, W, ups, act):
    """SPADE with a batch param-free norm in TRAINING mode (the mask decoder's SPADE, norms.py:152-153 with
    masker.py:120-150): batch statistics over (n, h, w), running statistics updated, and the gradients of x and the six
    mlp parameters against torch autograd of the reference expression."""
    from climategan_amd import ops
    from climategan_amd.norms import SPADE
    from oracle import cpu_ref
    B, CN = 3, 15
    hs, ws = (H // 2, W // 2) if ups else (H, W)
    mod = SPADE("batch", 3, C, CN)
    shapes = {k: tuple(v.shape) for k, v in mod.state_dict().items() if k.startswith("mlp_")}
    sd = {k: q(v, dt).requires_grad_(True) for k, v in fill.fill_state_dict(shapes, 5202).items()}
    x = q(fill.uniform((B, C, hs, ws), 5200 + C, -2, 2), dt).requires_grad_(True)
    cond = q(fill.uniform((B, CN, 2 * H, 2 * W), 5201), dt).requires_grad_(True)   # a prediction: wants a gradient
    bn = torch.nn.BatchNorm2d(C, affine=False).train()
    xu = cpu_ref.nearest_resize(x, (H, W)) if ups else x
    seg = cpu_ref.nearest_resize(cond, (H, W))
    actv = F.relu(F.conv2d(seg, sd["mlp_shared.0.weight"], sd["mlp_shared.0.bias"], padding=1))
    gamma = F.conv2d(actv, sd["mlp_gamma.weight"], sd["mlp_gamma.bias"], padding=1)
    beta = F.conv2d(actv, sd["mlp_beta.weight"], sd["mlp_beta.bias"], padding=1)
    pre = bn(xu) * (1 + gamma) + beta
    y = F.leaky_relu(pre, 0.2) if act == "lrelu" else pre
    dy = q(fill.uniform(tuple(y.shape), 5203), dt)
    y.backward(dy)
    # the 16-bit hidden map / gamma / beta move a pre-activation by ~1 % of its scale: within that distance of the
    # LeakyReLU kink the two sides disagree about the slope (a factor 5 on that element's gradient); such elements
    # (about 1 %) are left out of the max-norm comparison of dx, their effect on the sums stays inside the tolerance
    thr = 0.0 if act == "none" else (4e-3 if dt == torch.float16 else 3e-2)
    kink_ok = pre.detach().abs() >= thr
    if ups:
        kink_ok = kink_ok.reshape(B, C, hs, 2, ws, 2).permute(0, 1, 2, 4, 3, 5).reshape(B, C, hs, ws, 4).all(-1)
    assert kink_ok.float().mean().item() > 0.9

    mod.load_state_dict({k: v.detach() for k, v in sd.items()}, strict=False)
    mod = mod.cuda().train()
    xt = to_nhwc(x.detach(), dt).t.requires_grad_(True)
    condg = ops.nchw_to_nhwc(cond.detach().cuda(), dt, cs=ops.cs4(CN))
    condg.t.requires_grad_(True)
    a = ops.ACT_LRELU if act == "lrelu" else ops.ACT_NONE
    out = mod.forward_nhwc(ops.NHWC(xt, C), condg, act=a, x_upsample=ups)
    assert rel_err(back(ops.NHWC(out.t.detach(), C)), y.detach()) <= TOL[dt]
    assert mod.param_free_norm.num_batches_tracked.item() == 1
    assert rel_err(mod.param_free_norm.running_mean.cpu(), bn.running_mean) <= 1e-4
    assert rel_err(mod.param_free_norm.running_var.cpu(), bn.running_var) <= 1e-4
    out.t.backward(to_nhwc(dy, dt).t)
    assert rel_err(back(ops.NHWC(xt.grad, C)) * kink_ok, x.grad * kink_ok) <= (4e-3 if dt == torch.float16 else 1e-1)
    for k in shapes:
        g = dict(mod.named_parameters())[k].grad
        e = rel_err(g.cpu(), sd[k].grad)
        assert e <= (5e-3 if dt == torch.float16 else 8e-2), "%s: rel err %.3g" % (k, e)
    # the conditioning map's gradient: mlp_shared's data gradient summed over the pixels that read each cond pixel
    # (only every other row / column of the 2H x 2W map is read: the rest must be exactly zero)
    gc, rc = back(ops.NHWC(condg.t.grad, CN)), cond.grad
    assert torch.equal(gc == 0, rc == 0) or ((gc == 0) | (rc != 0)).all()
    assert ((gc - rc).abs().mean() / rc.abs().mean()).item() <= (1e-2 if dt == torch.float16 else 8e-2)
    assert rel_err(gc, rc) <= (2e-2 if dt == torch.float16 else 2e-1)
    # eval mode: running statistics, no graph wanted
    mod.eval()
    with torch.no_grad():
        ev = mod.forward_nhwc(ops.NHWC(xt.detach(), C), ops.NHWC(condg.t.detach(), CN), act=a, x_upsample=ups)
    bn.eval()
    with torch.no_grad():
        ye = bn(xu) * (1 + gamma) + beta
        ye = F.leaky_relu(ye, 0.2) if act == "lrelu" else ye
    assert rel_err(back(ev), ye) <= TOL[dt]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("c,cs_in,src,dst", [(15, 16, (8, 10), (16, 20)), (15, 16, (8, 10), (4, 5)), (3, 4, (7, 9), (7, 9)),
                                             (6, 8, (5, 7), (13, 11)), (20, 24, (12, 9), (5, 4)), (1, 8, (3, 2), (10, 9))])
def test_resize_nearest_backward(dt, c, cs_in, src, dst):
    """cgan_resize_nearest_bwd_nhwc: the adjoint of F.interpolate(mode="nearest") (norms.py:179) vs torch autograd,
    up- and down-sampling, non-integer ratios."""
    from climategan_amd import ops
    B = 2
    x = q(fill.uniform((B, c) + src, 7400 + c), dt).requires_grad_(True)
    y = F.interpolate(x, size=dst, mode="nearest")
    dy = q(fill.uniform(tuple(y.shape), 7401), dt)
    y.backward(dy)
    dyg = to_nhwc(dy, dt)
    got = ops.resize_nearest_bwd(dyg, src, cs_in)
    assert got.t.shape == (B,) + src + (cs_in,)
    assert (got.t[..., c:] == 0).all()
    assert rel_err(back(got), x.grad) <= TOL[dt]
    # forward / adjoint consistency: <resize(x), dy> == <x, resize_bwd(dy)>
    xg = ops.nchw_to_nhwc(x.detach().cuda(), dt, cs=cs_in)
    fwd = ops.resize_nearest(xg, dst, cs_out=dyg.t.shape[-1])
    lhs = (fwd.t.float() * dyg.t.float()).sum().item()
    rhs = (xg.t.float() * got.t.float()).sum().item()
    assert abs(lhs - rhs) <= 2e-2 * max(abs(lhs), 1.0)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("with_x", [True, False])
def test_make_m_cond_backward(dt, with_x):
    """cgan_make_m_cond_bwd_nhwc vs torch autograd of cat[normalize(d), softmax(s), bilinear(x)] (generator.py:196-230
    with tutils.normalize, tutils.py:567-576): gradients of d (including the arg-min / arg-max terms) and of s."""
    from climategan_amd import ops
    from climategan_amd.autograd import MakeMCondFn
    from oracle import cpu_ref
    B, H, W, SC = 3, 20, 24, 11
    d = q(fill.uniform((B, 1, H, W), 7500, 0.3, 6.9), dt)
    with torch.no_grad():                                      # unique extrema per sample (ties are a 16-bit artefact)
        for i in range(B):
            d[i, 0, 3 + i, 5] = 0.125
            d[i, 0, 7, 2 + i] = 7.5
    d.requires_grad_(True)
    s = q(fill.uniform((B, SC, H, W), 7501, -3, 3), dt).requires_grad_(True)
    x = q(fill.uniform((B, 3, 4 * H, 4 * W), 7502), dt) if with_x else None
    cond = cpu_ref.make_m_cond(d, s, x)
    g = q(fill.uniform(tuple(cond.shape), 7503), dt)
    cond.backward(g)

    dg, sg = to_nhwc(d.detach(), dt), to_nhwc(s.detach(), dt)
    dg.t.requires_grad_(True)
    sg.t.requires_grad_(True)
    out = MakeMCondFn.apply(dg.t, sg.t, x.cuda() if with_x else None, SC)
    cc = 1 + SC + (3 if with_x else 0)
    assert rel_err(back(ops.NHWC(out.detach(), cc)), cond.detach()) <= TOL[dt]
    out.backward(ops.nchw_to_nhwc(g.cuda(), dt, cs=ops.cs4(cc)).t)
    assert (dg.t.grad[..., 1:] == 0).all() and (sg.t.grad[..., SC:] == 0).all()
    # the arg-min / arg-max pixels carry sums over the whole sample (large values): max-norm comparison covers them
    assert rel_err(back(ops.NHWC(dg.t.grad, 1)), d.grad) <= 2 * TOL[dt]
    assert rel_err(back(ops.NHWC(sg.t.grad, SC)), s.grad) <= 2 * TOL[dt]


@pytest.mark.parametrize("dt", DTYPES)
def test_conv2dblock_spectral_batch_training(dt):
    """Conv2dBlock(norm="spectral_batch", reflect padding, LeakyReLU) in training mode (the projection convs of the
    SPADE mask decoder, masker.py:85-118): spectral-norm conv -> batch-statistics BatchNorm -> activation, forward and
    the gradients of x, weight_bar, gamma, beta against torch."""
    from climategan_amd import ops
    from climategan_amd.blocks import Conv2dBlock
    from climategan_amd.norms import spectral_norm_step_all
    B, cin, cout, H, W = 2, 32, 24, 12, 10
    blk = Conv2dBlock(cin, cout, 3, padding=1, activation="lrelu", pad_type="reflect", norm="spectral_batch")
    shapes = {k: tuple(v.shape) for k, v in blk.state_dict().items() if "num_batches" not in k}
    blk.load_state_dict({k: torch.from_numpy(v) for k, v in fill.fill_state_dict(shapes, 5300).items()}, strict=False)
    with torch.no_grad():
        blk.norm.running_var.abs_().add_(0.5)
    sd = {k: v.clone() for k, v in blk.state_dict().items()}
    x = q(fill.uniform((B, cin, H, W), 5301, -1, 1), dt).requires_grad_(True)
    # torch restatement: one power iteration (norms.py:66-87), conv with w_bar / sigma, BatchNorm (training), LeakyReLU
    wb = sd["conv.module.weight_bar"].clone().requires_grad_(True)
    u, v = sd["conv.module.weight_u"], sd["conv.module.weight_v"]
    wm = wb.detach().reshape(cout, -1)
    v2 = F.normalize(wm.t() @ u, dim=0, eps=1e-12)
    u2 = F.normalize(wm @ v2, dim=0, eps=1e-12)
    sigma = u2 @ (wb.reshape(cout, -1) @ v2)
    bn = torch.nn.BatchNorm2d(cout).train()
    bn.load_state_dict({k[5:]: t for k, t in sd.items() if k.startswith("norm.")})
    bias = sd.get("conv.module.bias")
    pre = F.conv2d(F.pad(x, (1,) * 4, mode="reflect"), wb / sigma, bias)
    z = bn(pre)
    y = F.leaky_relu(z, 0.2)
    dy = q(fill.uniform(tuple(y.shape), 5302), dt)
    y.backward(dy)
    frac_near_kink = (z.detach().abs() < (4e-3 if dt == torch.float16 else 3e-2)).float().mean().item()
    assert frac_near_kink < 0.05          # 16-bit conv outputs that close to the kink may take the other slope

    blk = blk.cuda().train()
    xt = to_nhwc(x.detach(), dt).t.requires_grad_(True)
    spectral_norm_step_all(blk, dt)
    out = blk.forward_nhwc(ops.NHWC(xt, cin))
    assert rel_err(back(ops.NHWC(out.t.detach(), cout)), y.detach()) <= TOL[dt]
    out.t.backward(to_nhwc(dy, dt).t)
    tol = 6e-3 if dt == torch.float16 else 6e-2
    # dx sums 9 taps x 24 channels of dz: a flipped slope on a few of them moves single elements of dx, so dx is
    # compared in the mean (relative L1) and loosely in the max norm
    got, ref = back(ops.NHWC(xt.grad, cin)), x.grad
    assert ((got - ref).abs().mean() / ref.abs().mean()).item() <= tol
    assert rel_err(got, ref) <= 4 * tol
    # (bf16: the BatchNorm backward subtracts two means from 8-bit-mantissa values before the 240-pixel weight sums)
    assert rel_err(blk.conv.module.weight_bar.grad.cpu(), wb.grad) <= (tol if dt == torch.float16 else 1e-1)
    # (a few of each channel's 240 pre-activations sit close enough to the kink to take the other slope in bf16)
    assert rel_err(blk.norm.weight.grad.cpu(), bn.weight.grad) <= (tol if dt == torch.float16 else 1e-1)
    assert rel_err(blk.norm.bias.grad.cpu(), bn.bias.grad) <= (tol if dt == torch.float16 else 1e-1)
    assert rel_err(blk.norm.running_mean.cpu(), bn.running_mean) <= 2e-3      # statistics of a 16-bit conv output
    assert rel_err(blk.norm.running_var.cpu(), bn.running_var) <= 2e-3


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("c,h,w,act,affine", [(64, 20, 24, "relu", True), (256, 9, 11, "none", True), (20, 13, 7, "lrelu", False)])
def test_batchnorm_training_forward_backward(dt, c, h, w, act, affine):
    """autograd.BatchNormActFn (training-mode BatchNorm2d + activation) vs torch: output, running statistics after the
    update, and the gradients of x, gamma, beta."""
    from climategan_amd import ops
    from climategan_amd.autograd import BatchNormActFn
    B = 3
    bn = torch.nn.BatchNorm2d(c, affine=affine)
    if affine:
        with torch.no_grad():
            bn.weight.copy_(torch.from_numpy(1.0 + 0.3 * fill.uniform((c,), 7100)))
            bn.bias.copy_(torch.from_numpy(0.2 * fill.uniform((c,), 7101)))
    bn.running_mean.copy_(torch.from_numpy(0.1 * fill.uniform((c,), 7102)))
    bn.running_var.copy_(torch.from_numpy(1.0 + 0.2 * fill.uniform01((c,), 7103)))
    rm0, rv0 = bn.running_mean.clone(), bn.running_var.clone()
    bn.train()
    x = q(fill.uniform((B, c, h, w), 7104 + c, -2, 2), dt).requires_grad_(True)
    f = {"relu": F.relu, "lrelu": lambda v: F.leaky_relu(v, 0.2), "none": lambda v: v}[act]
    pre = bn(x)
    y = f(pre)
    dy = q(fill.uniform((B, c, h, w), 7105), dt)
    y.backward(dy)
    # The HIP backward recovers the activation mask from the stored 16-bit output: a pre-activation within the type's
    # flush-to-zero range of the kink (x exactly at the batch mean: found by tests/test_gpu_norm_fuzz.py) lands on the
    # other side of it.  Such elements are excluded; their effect on the channel's sums stays inside the tolerance.
    kink_ok = torch.ones_like(pre, dtype=torch.bool) if act == "none" else pre.detach().abs() > 1e-5

    a = {"relu": ops.ACT_RELU, "lrelu": ops.ACT_LRELU, "none": ops.ACT_NONE}[act]
    xt = to_nhwc(x.detach(), dt).t.requires_grad_(True)
    g = bn.weight.detach().clone().cuda().requires_grad_(True) if affine else None
    b = bn.bias.detach().clone().cuda().requires_grad_(True) if affine else None
    rm, rv = rm0.cuda(), rv0.cuda()
    nbt = torch.tensor(41, dtype=torch.int64, device="cuda")          # num_batches_tracked: += 1 per training forward
    out = BatchNormActFn.apply(xt, g, b, rm, rv, c, bn.eps, bn.momentum, a, 0.2, nbt)
    assert nbt.item() == 42 and bn.num_batches_tracked.item() == 1
    assert rel_err(back(ops.NHWC(out.detach(), c)), y.detach()) <= TOL[dt]
    assert rel_err(rm.cpu(), bn.running_mean) <= 1e-5 and rel_err(rv.cpu(), bn.running_var) <= 1e-5
    out.backward(to_nhwc(dy, dt).t)
    assert kink_ok.float().mean().item() > 0.999
    assert rel_err(back(ops.NHWC(xt.grad, c)) * kink_ok, x.grad * kink_ok) <= (4e-3 if dt == torch.float16 else 4e-2)
    if affine:
        assert rel_err(g.grad.cpu(), bn.weight.grad) <= (3e-3 if dt == torch.float16 else 2e-2)
        assert rel_err(b.grad.cpu(), bn.bias.grad) <= (3e-3 if dt == torch.float16 else 2e-2)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("c,h,w,act", [(64, 9, 11, "relu"), (20, 7, 5, "relu"), (256, 6, 6, "none"), (24, 5, 9, "lrelu")])
def test_batchnorm_training_fused_residual(dt, c, h, w, act):
    """relu(bn3(.) + residual), the bottleneck tail (resnet101_v3.py:62-70), with the residual in the BatchNorm apply
    kernel: output and the gradients of x, the residual, gamma, beta vs torch."""
    from climategan_amd import ops
    from climategan_amd.autograd import BatchNormActFn
    B = 3
    bn = torch.nn.BatchNorm2d(c)
    with torch.no_grad():
        bn.weight.copy_(torch.from_numpy(1.0 + 0.3 * fill.uniform((c,), 7200)))
        bn.bias.copy_(torch.from_numpy(0.2 * fill.uniform((c,), 7201)))
    bn.train()
    x = q(fill.uniform((B, c, h, w), 7204 + c, -2, 2), dt).requires_grad_(True)
    r = q(fill.uniform((B, c, h, w), 7304 + c, -1, 1), dt).requires_grad_(True)
    f = {"relu": F.relu, "lrelu": lambda v: F.leaky_relu(v, 0.2), "none": lambda v: v}[act]
    pre = bn(x) + r
    y = f(pre)
    dy = q(fill.uniform((B, c, h, w), 7205), dt)
    y.backward(dy)
    kink_ok = torch.ones_like(pre, dtype=torch.bool) if act == "none" else pre.detach().abs() > 1e-5

    a = {"relu": ops.ACT_RELU, "lrelu": ops.ACT_LRELU, "none": ops.ACT_NONE}[act]
    xt = to_nhwc(x.detach(), dt).t.requires_grad_(True)
    rt = to_nhwc(r.detach(), dt).t.requires_grad_(True)
    g = bn.weight.detach().clone().cuda().requires_grad_(True)
    b = bn.bias.detach().clone().cuda().requires_grad_(True)
    rm, rv = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
    out = BatchNormActFn.apply(xt, g, b, rm, rv, c, bn.eps, bn.momentum, a, 0.2, None, rt)
    assert rel_err(back(ops.NHWC(out.detach(), c)), y.detach()) <= TOL[dt]
    assert (out.detach()[..., c:] == 0).all()                    # storage padding stays zero
    out.backward(to_nhwc(dy, dt).t)
    assert kink_ok.float().mean().item() > 0.999
    assert rel_err(back(ops.NHWC(xt.grad, c)) * kink_ok, x.grad * kink_ok) <= (4e-3 if dt == torch.float16 else 4e-2)
    assert rel_err(back(ops.NHWC(rt.grad, c)) * kink_ok, r.grad * kink_ok) <= TOL[dt]
    assert rel_err(g.grad.cpu(), bn.weight.grad) <= (3e-3 if dt == torch.float16 else 2e-2)
    assert rel_err(b.grad.cpu(), bn.bias.grad) <= (3e-3 if dt == torch.float16 else 2e-2)
    # a residual that wants no gradient gets none
    xt2 = xt.detach().clone().requires_grad_(True)
    out2 = BatchNormActFn.apply(xt2, g, b, rm, rv, c, bn.eps, bn.momentum, a, 0.2, None, rt.detach())
    assert torch.equal(out2.detach(), out.detach())
    out2.backward(to_nhwc(dy, dt).t)
    assert torch.equal(xt2.grad, xt.grad)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("cin,cout,k,pad,H,W", [(64, 64, 3, 1, 20, 24), (16, 32, 3, 1, 9, 13), (8, 8, 7, 3, 12, 10)])
def test_conv_reflect_pad_backward(dt, cin, cout, k, pad, H, W):
    """Reflect-padded conv (Conv2dBlock of the mask / depth decoders): data gradient = pad-0 dgrad over the padded extent
    folded by the reflection's adjoint; weight gradient reads x through the reflection."""
    from climategan_amd import ops
    B = 2
    x = q(fill.uniform((B, cin, H, W), 6100 + cin), dt).requires_grad_(True)
    w = q(fill.uniform((cout, cin, k, k), 6200 + cout, -0.1, 0.1), dt).requires_grad_(True)
    b = torch.from_numpy(fill.uniform((cout,), 6300)).requires_grad_(True)
    y = F.conv2d(F.pad(x, (pad,) * 4, mode="reflect"), w, b)
    dy = q(fill.uniform(tuple(y.shape), 6400), dt)
    y.backward(dy)
    dyg = to_nhwc(dy, dt)
    dx = ops.conv2d_bwd_data(dyg, w.detach().cuda(), (B, H, W), pad=pad, pad_mode=ops.PAD_REFLECT)
    assert dx.t.shape[1:3] == (H, W)
    assert rel_err(back(dx), x.grad) <= 2 * TOL[dt]
    dw, db = ops.conv2d_bwd_weight(to_nhwc(x.detach(), dt), dyg, tuple(w.shape), pad=pad, pad_mode=ops.PAD_REFLECT)
    assert rel_err(dw.cpu(), w.grad) <= 2e-4 and rel_err(db.cpu(), b.grad) <= 2e-4


@pytest.mark.parametrize("dt", DTYPES)
def test_structural_functions_backward(dt):
    """Bilinear resize (both conventions), 3x3/s2 max pool, nearest x2, add+ReLU, multiply and channel concatenation:
    the autograd Functions of the Masker's graph against torch autograd."""
    from climategan_amd import ops
    from climategan_amd import autograd as ag
    B = 2
    tolg = 2 * TOL[dt]

    def run(fn_hip, fn_ref, shapes, seed):
        xs = [q(fill.uniform(s, seed + i, -2, 2), dt).requires_grad_(True) for i, s in enumerate(shapes)]
        y = fn_ref(*xs)
        dy = q(fill.uniform(tuple(y.shape), seed + 50), dt)
        y.backward(dy)
        ts = [to_nhwc(x.detach(), dt).t.requires_grad_(True) for x in xs]
        out = fn_hip(*ts)
        assert rel_err(back(ops.NHWC(out.detach(), y.shape[1])), y.detach()) <= TOL[dt]
        out.backward(to_nhwc(dy, dt).t)
        for x, tt in zip(xs, ts):
            assert rel_err(back(ops.NHWC(tt.grad, x.shape[1])), x.grad) <= tolg

    for align in (True, False):
        run(lambda a: ag.ResizeBilinearFn.apply(a, 24, (23, 31), align),
            lambda a: F.interpolate(a, size=(23, 31), mode="bilinear", align_corners=align), [(B, 24, 10, 14)], 8200)
        run(lambda a: ag.ResizeBilinearFn.apply(a, 11, (9, 7), align),
            lambda a: F.interpolate(a, size=(9, 7), mode="bilinear", align_corners=align), [(B, 11, 20, 16)], 8210)
    run(lambda a: ag.MaxPool3x3s2Fn.apply(a, 16), lambda a: F.max_pool2d(a, 3, 2, 1), [(B, 16, 21, 18)], 8220)
    run(lambda a: ag.ResizeNearest2xFn.apply(a, 8), lambda a: F.interpolate(a, scale_factor=2), [(B, 8, 6, 5)], 8230)
    run(lambda a, b: ag.AddActFn.apply(a, b, 16, ops.ACT_RELU, 0.0), lambda a, b: F.relu(a + b),
        [(B, 16, 7, 9), (B, 16, 7, 9)], 8240)
    run(lambda a, b: ag.MulFn.apply(a, b, 16), lambda a, b: a * b, [(B, 16, 7, 9), (B, 16, 7, 9)], 8250)
    run(lambda a, b, c: ag.ConcatFn.apply([16, 24, 5], a, b, c), lambda a, b, c: torch.cat([a, b, c], 1),
        [(B, 16, 6, 7), (B, 24, 6, 7), (B, 5, 6, 7)], 8260)


def test_conv_backward_padded_1x1_quirk():
    """ASPPv3Plus.conv_out: a 1x1 conv with padding=1 (output grows by 2): data gradient = the interior
    of the transposed conv (pad' = -1), weight gradient over the padded extent."""
    from climategan_amd import ops
    dt = torch.float16
    x = q(fill.uniform((2, 64, 10, 12), 9300), dt).requires_grad_(True)
    w = q(fill.uniform((32, 64, 1, 1), 9301, -0.1, 0.1), dt).requires_grad_(True)
    b = torch.from_numpy(fill.uniform((32,), 9302)).requires_grad_(True)
    y = F.conv2d(x, w, b, padding=1)
    assert y.shape[-2:] == (12, 14)
    dy = q(fill.uniform(tuple(y.shape), 9303), dt)
    y.backward(dy)
    dyg = to_nhwc(dy, dt)
    dx = ops.conv2d_bwd_data(dyg, w.detach().cuda(), (2, 10, 12), pad=1)
    assert rel_err(back(dx), x.grad) <= TOL[dt]
    dw, db = ops.conv2d_bwd_weight(to_nhwc(x.detach(), dt), dyg, tuple(w.shape), pad=1)
    assert rel_err(dw.cpu(), w.grad) <= 2e-4 and rel_err(db.cpu(), b.grad) <= 2e-4


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("case", [(256, 64, 1, 0, 1, 2, 24, 40), (128, 96, 3, 2, 2, 1, 33, 21), (40, 24, 3, 1, 1, 2, 16, 16)])
def test_dgrad_with_added_gradient(dt, case):
    """cgan_conv2d_nhwc_bwd_data_add (ops.conv2d_bwd_data(add=...)): dx = data gradient + the other gradient contribution of
    the same tensor, summed in the conv kernel's epilogue -- every kernel family (wide-layer GEMM, 3x3 LDS tile, general).
    The sum happens in fp32 before the one 16-bit rounding, so it must be at least as close to the fp32 result as the
    two-pass form (gradient rounded, then added in 16 bit)."""
    from climategan_amd import ops

    cin, cout, k, pad, dil, B, H, W = case
    rng = np.random.RandomState(5)
    w = torch.from_numpy(rng.randn(cout, cin, k, k).astype(np.float32) * 0.05).cuda()
    dy = to_nhwc(torch.from_numpy(rng.randn(B, cout, H, W).astype(np.float32)), dt)
    add = to_nhwc(torch.from_numpy(rng.randn(B, cin, H, W).astype(np.float32)), dt)
    dx = ops.conv2d_bwd_data(dy, w, (B, H, W), stride=1, pad=pad, dilation=dil, add=add)
    ref = F.conv_transpose2d(back(dy).float(), q(w.cpu().numpy(), dt), stride=1, padding=pad, dilation=dil) + back(add).float()
    two_pass = ops.conv2d_bwd_data(dy, w, (B, H, W), stride=1, pad=pad, dilation=dil)
    two_pass = (two_pass.t + add.t)
    e1 = rel_err(back(dx).float(), ref)
    e2 = rel_err(back(ops.NHWC(two_pass, cin)).float(), ref)
    assert e1 <= TOL[dt] and e1 <= 1.5 * e2 + 1e-5, (e1, e2)
    assert (dx.t[..., cin:] == 0).all()


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("case", [(256, 64, 1, 0, 1, 1, 2, 24, 40), (128, 96, 3, 2, 2, 1, 1, 33, 21),
                                  (40, 24, 3, 1, 1, 1, 2, 16, 16), (128, 256, 3, 1, 1, 1, 2, 40, 40),
                                  (64, 64, 3, 1, 1, 1, 1, 48, 32), (128, 64, 3, 1, 1, 2, 1, 20, 20)])
def test_dgrad_with_relu_derivative(dt, case):
    """cgan_conv2d_nhwc_bwd_data_relu (ops.conv2d_bwd_data(relu_out=...)): dx = data gradient * [relu_out > 0], the mask
    applied in the conv kernel's epilogue -- every kernel family.  The mask either keeps or zeroes the fp32 value before the
    one 16-bit rounding, so the result is BIT-IDENTICAL to the two-pass form (gradient rounded, then masked).  The last case
    (stride 2) is not a 'same' convolution: the wrapper falls back to the two passes."""
    from climategan_amd import ops

    cin, cout, k, pad, dil, stride, B, H, W = case
    rng = np.random.RandomState(6)
    w = torch.from_numpy(rng.randn(cout, cin, k, k).astype(np.float32) * 0.05).cuda()
    ho = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1
    wo = (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
    dy = to_nhwc(torch.from_numpy(rng.randn(B, cout, ho, wo).astype(np.float32)), dt)
    x = to_nhwc(torch.relu(torch.from_numpy(rng.randn(B, cin, H, W).astype(np.float32))), dt)
    dx = ops.conv2d_bwd_data(dy, w, (B, H, W), stride=stride, pad=pad, dilation=dil, relu_out=x)
    two_pass = ops.act_bwd(x, ops.conv2d_bwd_data(dy, w, (B, H, W), stride=stride, pad=pad, dilation=dil), ops.ACT_RELU)
    assert torch.equal(dx.t, two_pass.t)
    assert 0.2 < (dx.t[..., :cin] == 0).float().mean().item() < 0.8       # the mask did something, and not everything
    ref = F.conv_transpose2d(back(dy).float(), q(w.cpu().numpy(), dt), stride=stride, padding=pad, dilation=dil,
                             output_padding=(H + 2 * pad - dil * (k - 1) - 1) % stride) * (back(x).float() > 0)
    assert rel_err(back(dx).float(), ref) <= TOL[dt]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("case", [(256, 64, 1, 0, 1, 1, 2, 48, 64),       # 2-stage 128 x 128 GEMM tile: fused (has_res = 3)
                                  (1024, 256, 1, 0, 1, 1, 1, 70, 61),     # the same with pixel tails
                                  (2048, 512, 1, 0, 1, 1, 1, 128, 130),   # 256 x 256 tile: fused
                                  (128, 96, 3, 2, 2, 1, 1, 33, 21),       # 3x3: add in the epilogue + the mask as a pass
                                  (40, 24, 3, 1, 1, 1, 2, 16, 16),        # general kernel: likewise
                                  (128, 64, 3, 1, 1, 2, 1, 20, 20)])      # stride 2: the wrapper's own fallback
def test_dgrad_with_added_gradient_and_relu_derivative(dt, case):
    """cgan_conv2d_nhwc_bwd_data_add_relu (ops.conv2d_bwd_data(add=..., relu_out=...)): dx = [relu_out > 0] * (data gradient
    + add) -- what a bottleneck's first conv hands back to the previous block's relu(bn3(.) + skip).  BIT-IDENTICAL to the
    add-in-the-epilogue call followed by the activation-backward pass, whichever kernel the descriptor selects."""
    from climategan_amd import ops

    cin, cout, k, pad, dil, stride, B, H, W = case
    rng = np.random.RandomState(7)
    w = torch.from_numpy(rng.randn(cout, cin, k, k).astype(np.float32) * 0.05).cuda()
    ho = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1
    wo = (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
    dy = to_nhwc(torch.from_numpy(rng.randn(B, cout, ho, wo).astype(np.float32)), dt)
    add = to_nhwc(torch.from_numpy(rng.randn(B, cin, H, W).astype(np.float32)), dt)
    x = to_nhwc(torch.relu(torch.from_numpy(rng.randn(B, cin, H, W).astype(np.float32))), dt)
    dx = ops.conv2d_bwd_data(dy, w, (B, H, W), stride=stride, pad=pad, dilation=dil, add=add, relu_out=x)
    two_pass = ops.act_bwd(x, ops.conv2d_bwd_data(dy, w, (B, H, W), stride=stride, pad=pad, dilation=dil, add=add), ops.ACT_RELU)
    assert torch.equal(dx.t, two_pass.t)
    assert 0.2 < (dx.t[..., :cin] == 0).float().mean().item() < 0.8
    ref = (F.conv_transpose2d(back(dy).float(), q(w.cpu().numpy(), dt), stride=stride, padding=pad, dilation=dil,
                              output_padding=(H + 2 * pad - dil * (k - 1) - 1) % stride) + back(add).float()) * (back(x).float() > 0)
    assert rel_err(back(dx).float(), ref) <= TOL[dt]
    assert (dx.t[..., cin:] == 0).all()


@pytest.mark.parametrize("dt", DTYPES)
def test_relu_mask_in_the_next_blocks_data_gradient_is_bitwise_the_batchnorm_mask(dt):
    """norms.FUSE_RELU_MASK: three bottlenecks in a row, the derivative of relu(bn3(.) + skip) taken by the NEXT block's first
    data-gradient conv (autograd.claim_relu_mask) or by bn3's own backward -- every gradient bit for bit the same; the last
    block's output has another reader (the loss), so its mask stays where it was."""
    from climategan_amd import autograd, norms, ops
    from climategan_amd.deeplab import resnet101_v3 as R

    torch.manual_seed(4)
    # (256 -> 64 -> 256 at 4 x 48 x 48: the first convs' data gradients run on the wide-layer GEMM, i.e. the fused epilogue)
    blocks = [R.Bottleneck(256, 64, 1, 1, torch.nn.Sequential(torch.nn.Conv2d(256, 256, 1, bias=False), torch.nn.BatchNorm2d(256)),
                           torch.nn.BatchNorm2d),
              R.Bottleneck(256, 64, 1, 2, None, torch.nn.BatchNorm2d), R.Bottleneck(256, 64, 1, 1, None, torch.nn.BatchNorm2d)]
    blocks = [b.cuda().train() for b in blocks]
    x0 = torch.randn(4, 256, 48, 48, device="cuda")
    grads, claimed = {}, {}
    for fuse in (True, False):
        norms.FUSE_RELU_MASK = fuse
        try:
            for b in blocks:
                for p in b.parameters():
                    p.grad = None
            x = ops.nchw_to_nhwc(x0, dt)
            x.t.requires_grad_(True)
            y, nodes = x, []
            for i, b in enumerate(blocks):
                y = b.forward_nhwc(y, sole_consumer=i > 0)
                nodes.append(y.t.grad_fn)
            claimed[fuse] = [bool(n.premasked) for n in nodes]
            (y.t.float() * torch.linspace(-1, 1, y.t.numel(), device="cuda").view_as(y.t)).sum().backward()
            grads[fuse] = [x.t.grad.clone()] + [p.grad.clone() for b in blocks for p in b.parameters()]
        finally:
            norms.FUSE_RELU_MASK = True
    assert claimed[True] == [True, True, False] and claimed[False] == [False, False, False]
    for a, b in zip(grads[True], grads[False]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("dt", DTYPES)
def test_conv_pass_fn_matches_autograd_on_a_residual_block(dt):
    """autograd.ConvPassFn: y = act(bn(conv1(x))) ... + x with x handed through conv1's node.  Gradients of x, the conv
    weight and the tail must equal those of the plain graph (two consumers of x, autograd's own accumulation) up to the
    one rounding the fused sum saves."""
    from climategan_amd import ops
    from climategan_amd.deeplab import resnet101_v3 as R

    torch.manual_seed(3)
    blk = R.Bottleneck(64, 16, 1, 1, None, torch.nn.BatchNorm2d).cuda().train()
    x0 = torch.randn(2, 64, 24, 20, device="cuda")
    grads = {}
    for fuse in (True, False):
        R.FUSE_RESIDUAL_GRADIENT = fuse
        try:
            for p in blk.parameters():
                p.grad = None
            x = ops.nchw_to_nhwc(x0, dt)
            x.t.requires_grad_(True)
            y = blk.forward_nhwc(x)
            (y.t.float() * torch.linspace(-1, 1, y.t.numel(), device="cuda").view_as(y.t)).sum().backward()
            grads[fuse] = [x.t.grad.float().clone()] + [p.grad.float().clone() for p in blk.parameters()]
        finally:
            R.FUSE_RESIDUAL_GRADIENT = True
    for a, b in zip(grads[True], grads[False]):
        scale = b.abs().max().item() + 1e-12
        assert (a - b).abs().max().item() <= (2 * TOL[dt]) * scale


def test_batched_weight_pack_is_bitwise_the_single_pack():
    """ops.pack_conv_weights_batched (what norms._PackCache.get_plain uses to refresh every stale plain conv weight after an
    optimizer step in one launch) vs ops.pack_conv_weight, byte for byte, incl. buffer reuse."""
    from climategan_amd import ops

    torch.manual_seed(0)
    shapes = [(64, 3, 7, 7), (256, 64, 1, 1), (64, 64, 3, 3), (20, 40, 3, 3), (1, 8, 3, 3), (512, 2048, 1, 1)]
    params = [(torch.randn(s, device="cuda") * 0.1, torch.randn(s[0], device="cuda") if i % 2 else None)
              for i, s in enumerate(shapes)]
    for dt in DTYPES:
        got = ops.pack_conv_weights_batched(params, dt)
        again = ops.pack_conv_weights_batched([(w * 2, b) for w, b in params], dt, got)       # reuse the buffers
        for (w, b), g, a in zip(params, got, again):
            assert a.w.data_ptr() == g.w.data_ptr()
            ref2 = ops.pack_conv_weight(w * 2, b, dt)
            assert torch.equal(a.w, ref2.w) and torch.equal(a.bias, ref2.bias)


def test_batched_dgrad_pack_equals_the_per_layer_pack():
    """``ops.dgrad_prepack_run`` (all stride-1 data-gradient operators of a backward pass in ONE launch,
    cgan_conv2d_pack_weight_batched with ``transposed``) writes exactly the bytes ``cgan_conv2d_pack_weight_dgrad`` writes
    per layer -- with and without a spectral-norm sigma, 1x1 / 3x3 / 4x4 / 7x7, channel counts that are not multiples of
    the tile sizes -- and ``conv2d_bwd_data`` then gives identical gradients from either."""
    import ctypes as C
    from climategan_amd import _lib, ops

    lib = _lib.load()
    ops.dgrad_prepack_run()                                # whatever earlier forwards of this process left registered
    g = torch.Generator(device="cuda").manual_seed(11)
    shapes = [(64, 256, 1), (256, 64, 1), (20, 40, 3), (3, 20, 3), (128, 128, 3), (48, 256, 1), (512, 4, 4), (1, 8, 3),
              (64, 3, 7), (304, 256, 3)]
    for dt in (torch.bfloat16, torch.float16):
        hs, refs = [], []
        for i, (co, ci, k) in enumerate(shapes):
            w = torch.randn(co, ci, k, k, device="cuda", generator=g) * 0.1
            sigma = torch.rand(1, device="cuda", generator=g) + 0.5 if i % 2 else None
            h = ops.dgrad_register(w, sigma, dt, 1)
            assert h is not None
            hs.append(h)
            d = ops._conv_desc(ops._DT[dt], 2, 24, 24, ci, co, k, k, 1, k // 2, 1, ops.PAD_ZERO, has_bias=False)
            nbytes = lib.cgan_conv2d_dgrad_packed_weight_bytes(C.byref(d))
            ref = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
            _lib.check(lib.cgan_conv2d_pack_weight_dgrad(ops._ptr(w), ops._ptr(sigma), ops._ptr(ref), C.byref(d), ops._stream()),
                       "cgan_conv2d_pack_weight_dgrad")
            refs.append(ref)
        assert ops.dgrad_register(torch.randn(8, 8, 4, 4, device="cuda"), None, dt, 2) is None      # strided: per-call pack
        assert ops.dgrad_prepack_run() == len(shapes)
        for h, ref, shp in zip(hs, refs, shapes):
            assert h.packed is not None and h.packed.numel() == ref.numel(), shp
            assert torch.equal(h.packed, ref), shp
        co, ci, k = shapes[4]
        dy = ops.NHWC(torch.randn(2, 24, 24, co, device="cuda", generator=g).to(dt), co)
        a = ops.conv2d_bwd_data(dy, hs[4].w, (2, 24, 24), pad=1, sigma=hs[4].sigma, prepacked=hs[4])
        b = ops.conv2d_bwd_data(dy, hs[4].w, (2, 24, 24), pad=1, sigma=hs[4].sigma)
        assert torch.equal(a.t, b.t)


@pytest.mark.parametrize("case", [
    # (cin, cout, k, pad, dil, n, h, w, groups): wide layers the plain GEMM kernel takes, one per block tile
    (256, 64, 3, 1, 1, 4, 64, 64, 1),        # 64 couts x 256 pixels
    (256, 256, 3, 1, 1, 8, 80, 80, 2),       # 256 x 128
    (256, 512, 3, 1, 1, 4, 80, 80, 1),       # 128 x 256 (128-pixel chunks)
    (1024, 256, 1, 0, 1, 8, 40, 40, 2), (256, 256, 3, 2, 2, 8, 40, 40, 2), (256, 128, 3, 1, 1, 4, 64, 64, 1),   # 128 x 128
    (256, 1024, 1, 0, 1, 4, 40, 40, 1), (128, 512, 1, 0, 1, 2, 96, 100, 2),   # 128 x 128, 2-stage ring (short-K 1x1)
    (256, 1024, 1, 0, 1, 8, 80, 80, 2), (128, 512, 1, 0, 1, 4, 64, 64, 1), (64, 256, 1, 0, 1, 2, 128, 128, 2),   # x-resident 1x1 kernel
    (1024, 256, 1, 0, 1, 8, 80, 80, 2), (512, 2048, 1, 0, 1, 4, 80, 80, 1), (2048, 512, 1, 0, 1, 4, 64, 64, 2),   # 256 x 256 kernel (long-K 1x1)
    # 512 and more partial rows per group: the in-place pre-merge (round 6) in front of the per-channel walk -- 3 200 rows per
    # group (a last block of 128 of the 256), 1 600 / 800 rows in one group, 1 000 rows (a last block of 232)
    (64, 256, 1, 0, 1, 16, 160, 160, 2), (256, 256, 3, 1, 1, 16, 80, 80, 1), (256, 1024, 1, 0, 1, 20, 80, 80, 2),
])
def test_batchnorm_statistics_from_the_conv_epilogue(case):
    _bn_stats_from_epilogue(case, shifted=False)


@pytest.mark.parametrize("case", [(256, 256, 3, 1, 1, 8, 80, 80, 2), (256, 1024, 1, 0, 1, 8, 80, 80, 2), (1024, 256, 1, 0, 1, 8, 40, 40, 2),
                                  (1024, 256, 1, 0, 1, 8, 80, 80, 2), (64, 256, 1, 0, 1, 16, 160, 160, 2)])
def test_batchnorm_statistics_from_the_conv_epilogue_with_large_channel_means(case):
    """Post-ReLU inputs and weights with a common sign give conv outputs whose channel mean is ten or more standard deviations
    away from zero: the epilogue's M2 must not be formed as sum v^2 - (sum v)^2 / n (round 3: that form cost the encoder 2 %
    of its gradient norm against the reference's step)."""
    _bn_stats_from_epilogue(case, shifted=True)


@pytest.mark.parametrize("case", [
    # npix a multiple of a wave's statistics chunk but NOT of the workgroup's pixel count: the trailing waves of the last
    # block own no chunk (round-3 builds wrote their rows past the end of ``partial``)
    (256, 1024, 1, 0, 1, 4, 28, 28, 1),      # 3136 px = 49 x 64: plain 128-pixel blocks
    (256, 256, 3, 1, 1, 3, 40, 40, 1),       # 4800 px
    (128, 512, 1, 0, 1, 6, 56, 56, 1),       # 18816 px = 147 x 128: x-resident kernel, 256-pixel blocks
    (512, 128, 1, 0, 1, 5, 24, 24, 1),       # 2880 px = 45 x 64
    (1024, 256, 1, 0, 1, 7, 56, 56, 1),      # 21952 px = 343 x 64: the 256 x 256 kernel's last block has three live waves of four
    # whole workgroup tiles (one statistics row per workgroup), every block tile, with and without a bias
    (2048, 512, 1, 0, 1, 8, 80, 80, 1), (256, 256, 3, 1, 1, 8, 80, 80, 1), (512, 128, 1, 0, 1, 8, 80, 80, 1),
    (256, 64, 3, 1, 1, 4, 64, 64, 1), (256, 512, 3, 1, 1, 4, 80, 80, 1), (128, 512, 1, 0, 1, 4, 64, 64, 1),
])
@pytest.mark.parametrize("with_bias", [False, True])
def test_conv_epilogue_statistics_stay_inside_their_buffer(case, with_bias):
    """The partial rows a conv epilogue writes are exactly npix / chunk: a canary behind them stays untouched, the rows
    themselves equal the statistics of the stored y; with a bias (``spectral_batch`` blocks: conv + bias -> BatchNorm) the
    query and the launch agree on the kernel and the statistics are those of round(acc + bias)."""
    import ctypes as C
    from climategan_amd import _lib, ops

    cin, cout, k, pad, dil, n, h, w, _ = case
    dt = torch.bfloat16
    gen = torch.Generator(device="cuda").manual_seed(11)
    xf = torch.randn(n, cin, h, w, device="cuda", generator=gen).to(dt).float()
    wt = (torch.randn(cout, cin, k, k, device="cuda", generator=gen) * (2.0 / (cin * k * k)) ** 0.5).to(dt).float()
    bias = torch.randn(cout, device="cuda", generator=gen) * 3 if with_bias else None
    x = ops.nchw_to_nhwc(xf, dt)
    pw = ops.pack_conv_weight(wt, bias, dt)
    d = ops._conv_desc(x.dtype_id, n, h, w, cin, cout, k, k, 1, pad, dil, ops.PAD_ZERO, has_bias=with_bias)
    lib = _lib.load()
    ppb = lib.cgan_conv2d_stats_chunk_pixels(C.byref(d))
    npix = n * d.h_out * d.w_out
    if ppb <= 0:
        pytest.skip("this descriptor's kernel has no statistics epilogue")
    assert npix % ppb == 0
    rows, cs = npix // ppb, ops.cs8(cout)
    CANARY = 12345.0
    buf = torch.full((rows + 8, cs, 2), CANARY, dtype=torch.float32, device="cuda")
    y = torch.empty((n, d.h_out, d.w_out, cs), dtype=dt, device="cuda")
    _lib.check(lib.cgan_conv2d_nhwc_fwd_stats(ops._ptr(x.t), ops._ptr(pw.w), ops._ptr(pw.bias), ops._ptr(y), ops._ptr(buf),
                                              rows * cs * 2 * 4, C.byref(d), ops._stream()), "cgan_conv2d_nhwc_fwd_stats")
    torch.cuda.synchronize()
    assert bool((buf[rows:] == CANARY).all()), "the epilogue wrote past its npix / chunk partial rows"
    assert bool((buf[:rows, :cout] != CANARY).all()), "a partial row was left unwritten"
    y0 = ops.conv2d(x, pw, pad=pad, dilation=dil)
    assert torch.equal(y, y0.t)
    yv = y.float().view(rows, ppb, cs)[:, :, :cout].double()
    mean_ref = yv.mean(1)
    m2_ref = ((yv - mean_ref[:, None]) ** 2).sum(1)
    scale = max(yv.abs().max().item(), 1.0)
    assert (buf[:rows, :cout, 0].double() - mean_ref).abs().max().item() <= 2e-5 * scale
    assert (buf[:rows, :cout, 1].double() - m2_ref).abs().max().item() <= 2e-4 * max(m2_ref.max().item(), 1.0)


def _bn_stats_from_epilogue(case, shifted):
    """``ops.conv2d_with_stats`` + ``batchnorm_train_stats_from_partials`` (the conv kernel's epilogue reduces its fp32
    accumulators, rounded as they are stored, per chunk of pixels; one finalize launch) against the separate statistics pass
    over the stored y and against float64 over y: same y bit for bit, batch mean / rstd of the STORED values to fp32 accuracy
    (round 3: statistics of the unrounded accumulators amplified the rounding noise of near-constant channels), identical
    running-statistics semantics per group."""
    import torch.nn.functional as F
    from climategan_amd import ops

    cin, cout, k, pad, dil, n, h, w, G = case
    dt = torch.bfloat16
    gen = torch.Generator(device="cuda").manual_seed(3)
    xf = torch.randn(n, cin, h, w, device="cuda", generator=gen).to(dt).float()
    wt = (torch.randn(cout, cin, k, k, device="cuda", generator=gen) * (2.0 / (cin * k * k)) ** 0.5).to(dt).float()
    if shifted:
        xf = (xf.abs() + 1.0).to(dt).float()
        wt = (wt * 0.05 + 1.0 / (cin * k * k)).to(dt).float()
    x = ops.nchw_to_nhwc(xf, dt)
    pw = ops.pack_conv_weight(wt, None, dt)
    y, st = ops.conv2d_with_stats(x, pw, pad=pad, dilation=dil, groups=G)
    y0 = ops.conv2d(x, pw, pad=pad, dilation=dil)
    assert torch.equal(y.t, y0.t)
    assert st is not None, "this layer shape is expected to run the GEMM kernel with the statistics epilogue"
    gamma = torch.rand(cout, device="cuda", generator=gen) + 0.5
    beta = torch.randn(cout, device="cuda", generator=gen)
    npix = n * h * w // G

    def fresh():
        return torch.zeros(cout, device="cuda"), torch.ones(cout, device="cuda"), torch.zeros((), dtype=torch.int64, device="cuda")

    rm_a, rv_a, nbt_a = fresh()
    a = ops.batchnorm_train_stats_from_partials(st, G, npix, cout, gamma, beta, rm_a, rv_a, nbt_a, 1e-5, 0.1)
    rm_b, rv_b, nbt_b = fresh()
    flat = ops.NHWC(y.t.view(G, npix, 1, y.t.shape[-1]), cout)
    b = ops.batchnorm_train_stats(flat, gamma, beta, rm_b, rv_b, nbt_b, 1e-5, 0.1)
    # the statistics are those of the STORED (16-bit) values: the float64 reference is taken over y itself
    ref = ops.nhwc_to_nchw(y).double().view(G, n // G, cout, -1).permute(0, 2, 1, 3).reshape(G, cout, -1)
    mean_ref, var_ref = ref.mean(-1).float(), ref.var(-1, unbiased=False).float()
    if shifted:
        exact = F.conv2d(xf.double(), wt.double(), padding=pad, dilation=dil)
        ratio = (exact.mean((0, 2, 3)).abs() / exact.var((0, 2, 3), unbiased=False).sqrt()).median().item()
        assert ratio > 8, ratio                                                 # the regime this variant is about
    spread = var_ref.sqrt().max().item()
    assert (a[0][:, :cout] - mean_ref).abs().max().item() <= 2e-5 * max(spread, mean_ref.abs().max().item(), 1.0) + 1e-6
    assert ((1.0 / a[1][:, :cout] ** 2 - 1e-5) / var_ref - 1).abs().max().item() <= 2e-4
    for u, v in zip(a, b):                                           # the separate pass reads the same stored values
        assert (u[:, :cout] - v[:, :cout]).abs().max().item() <= 1e-4 * max(v[:, :cout].abs().max().item(), 1.0)
    assert int(nbt_a) == int(nbt_b) == G
    assert (rm_a - rm_b).abs().max().item() <= 1e-4 * max(spread, 1.0) + 1e-6 and (rv_a / rv_b - 1).abs().max().item() <= 1e-3


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("sizes", [((40, 40), (96, 96), 40), ((40, 52), (384, 384), 30), ((24, 24), (17, 31), 24)])
def test_depth_decoder_resize_backward(dt, sizes):
    """The DADA depth decoder's resize when the map's width differs from the target (depth.py:143-153): bicubic to the MiDaS
    size then nearest to the target, under autograd -- Fn.resize_bicubic / Fn.resize_nearest against torch's own
    F.interpolate pair (value and gradient)."""
    import torch.nn.functional as F
    from climategan_amd import functional as Fn, ops

    (h, w), mid, ts = sizes
    g = torch.Generator(device="cuda").manual_seed(17)
    x = torch.randn(2, 1, h, w, device="cuda", generator=g).to(dt).float().requires_grad_(True)
    up = torch.randn(2, 1, ts, ts, device="cuda", generator=g).to(dt).float()
    ref = F.interpolate(F.interpolate(x, size=mid, mode="bicubic", align_corners=False), size=(ts, ts), mode="nearest")
    ref.backward(up)
    xh = ops.NHWC(ops.nchw_to_nhwc(x.detach(), dt).t.requires_grad_(True), 1)
    y = Fn.resize_nearest(Fn.resize_bicubic(xh, mid), (ts, ts))
    got = ops.nhwc_to_nchw(ops.NHWC(y.t.detach(), 1))
    tol = 8e-3 if dt == torch.bfloat16 else 1e-3
    assert (got - ref.detach()).abs().max().item() <= 2 * tol * ref.detach().abs().max().item()      # two 16-bit stores
    y.t.backward(ops.nchw_to_nhwc(up, dt).t)
    dx = ops.nhwc_to_nchw(ops.NHWC(xh.t.grad, 1))
    assert (dx - x.grad).abs().max().item() <= 2 * tol * x.grad.abs().max().item()
    assert bool((xh.t.grad[..., 1:] == 0).all())
