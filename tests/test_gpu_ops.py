"""GPU parity tests of the individual HIP ops (through the C ABI) against the CPU oracle.

Inputs and weights are first rounded to the 16-bit compute type so that the comparison isolates the
kernel (accumulation order + output rounding): tolerance 1e-3 * max|ref| for fp16 (2^-11 output rounding),
8e-3 for bf16 (2^-8 output rounding)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from climategan_amd import fill
from oracle import cpu_ref

pytestmark = pytest.mark.gpu

DTYPES = [torch.float16, torch.bfloat16]
TOL = {torch.float16: 1e-3, torch.bfloat16: 8e-3}


def q(a, dt):
    """round an fp32 numpy array to the 16-bit type, back to fp32 torch (CPU)."""
    return torch.from_numpy(np.ascontiguousarray(a)).to(dt).float()


def to_nhwc(x_cpu, dt, cs=None):
    from climategan_amd import ops
    return ops.nchw_to_nhwc(x_cpu.cuda(), dt, cs=cs)


def back(y):
    from climategan_amd import ops
    return ops.nhwc_to_nchw(y).cpu()


def assert_close(got, ref, dt, what=""):
    scale = max(ref.abs().max().item(), 1e-6)
    err = (got - ref).abs().max().item()
    assert err <= TOL[dt] * scale + 1e-6, "%s: max err %.3g vs scale %.3g (rel %.3g)" % (what, err, scale, err / scale)


def test_library_loads_on_gpu():
    from climategan_amd import _lib
    assert _lib.load().cgan_version() == _lib.ABI_VERSION


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("c,h,w", [(3, 5, 7), (20, 9, 16), (40, 4, 4)])
def test_layout_roundtrip(dt, c, h, w):
    x = q(fill.uniform((2, c, h, w), 3), dt)
    y = back(to_nhwc(x, dt))
    assert torch.equal(y, x)


CONV_CASES = [
    # cin, cout, k, stride, pad, dil, pad_mode, H, W
    (8, 16, 3, 1, 1, 1, "zero", 12, 16),
    (20, 20, 3, 1, 1, 1, "zero", 17, 19),      # channels not multiple of 8, ragged spatial
    (40, 20, 1, 1, 0, 1, "zero", 16, 16),      # conv_s
    (3, 32, 3, 1, 1, 1, "zero", 5, 5),         # fc
    (20, 3, 3, 1, 1, 1, "zero", 16, 24),       # conv_img
    (4, 16, 4, 2, 1, 1, "zero", 32, 40),       # PatchGAN first conv
    (16, 32, 4, 1, 1, 1, "zero", 9, 11),       # PatchGAN stride-1 4x4
    (32, 32, 3, 1, 2, 2, "zero", 14, 14),      # dilated (ResNet layer3)
    (16, 16, 3, 1, 1, 1, "reflect", 10, 13),   # reflect pad (mask decoder)
    (64, 128, 3, 1, 1, 1, "zero", 8, 8),       # multi cout tiles, K = 576
]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d(dt, case):
    from climategan_amd import ops
    cin, cout, k, stride, pad, dil, pmode, H, W = case
    B = 2
    x = q(fill.uniform((B, cin, H, W), 100 + cin), dt)
    w = q(fill.uniform((cout, cin, k, k), 200 + cout, -0.2, 0.2), dt)
    b = torch.from_numpy(fill.uniform((cout,), 300 + cout))
    xin = F.pad(x, (pad,) * 4, mode="reflect") if pmode == "reflect" else x
    ref = F.conv2d(xin, w, b, stride=stride, padding=0 if pmode == "reflect" else pad, dilation=dil)
    pw = ops.pack_conv_weight(w.cuda(), b.cuda(), dt)
    y = ops.conv2d(to_nhwc(x, dt), pw, stride=stride, pad=pad, dilation=dil,
                   pad_mode=ops.PAD_REFLECT if pmode == "reflect" else ops.PAD_ZERO)
    assert y.t.shape == (B, ref.shape[2], ref.shape[3], ops.cs8(cout))
    assert_close(back(y), ref, dt, "conv %s" % (case,))
    # pad channels stay zero
    if ops.cs8(cout) != cout:
        assert y.t[..., cout:].abs().max().item() == 0


@pytest.mark.parametrize("dt", DTYPES)
def test_conv2d_epilogue_residual_upsample_act(dt):
    from climategan_amd import ops
    B, cin, cout, H, W = 2, 24, 24, 6, 10
    x = q(fill.uniform((B, cin, H, W), 1), dt)          # stored pre-upsample
    res = q(fill.uniform((B, cout, H, W), 2), dt)       # stored pre-upsample
    w = q(fill.uniform((cout, cin, 3, 3), 3, -0.2, 0.2), dt)
    b = torch.from_numpy(fill.uniform((cout,), 4))
    xu = cpu_ref.nearest_resize(x, (2 * H, 2 * W))
    ru = cpu_ref.nearest_resize(res, (2 * H, 2 * W))
    ref = F.leaky_relu(F.conv2d(xu, w, b, padding=1) + ru, 0.2)
    pw = ops.pack_conv_weight(w.cuda(), b.cuda(), dt)
    y = ops.conv2d(to_nhwc(x, dt), pw, pad=1, act=ops.ACT_LRELU, residual=to_nhwc(res, dt), in_upsample=True,
                   residual_upsample=True)
    assert_close(back(y), ref, dt, "conv+res+ups+lrelu")
    ref_t = torch.tanh(F.conv2d(x, w, b, padding=1))
    y = ops.conv2d(to_nhwc(x, dt), pw, pad=1, act=ops.ACT_TANH)
    assert_close(back(y), ref_t, dt, "conv+tanh")


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("c,h,w", [(20, 33, 47), (640, 5, 5), (40, 96, 128)])
def test_instnorm_stats(dt, c, h, w):
    from climategan_amd import ops
    x = q(fill.uniform((2, c, h, w), 7, -1, 3), dt)
    mean, rstd = ops.instnorm_stats(to_nhwc(x, dt))
    xd = x.double()
    rm = xd.mean(dim=(2, 3))
    rr = 1.0 / torch.sqrt(xd.var(dim=(2, 3), unbiased=False) + 1e-5)
    assert (mean.cpu()[:, :c].double() - rm).abs().max() < 1e-5 * 3
    assert ((rstd.cpu()[:, :c].double() - rr).abs() / rr).max() < 1e-4


@pytest.mark.parametrize("dt", DTYPES)
def test_instnorm_stats_large_offset(dt):
    """mean >> std: the Chan-merged (mean, M2) reduction must not lose the variance."""
    from climategan_amd import ops
    x = q(10.0 + 0.05 * fill.uniform((1, 8, 128, 160), 9), dt)
    mean, rstd = ops.instnorm_stats(to_nhwc(x, dt))
    xd = x.double()
    rr = 1.0 / torch.sqrt(xd.var(dim=(2, 3), unbiased=False) + 1e-5)
    assert ((rstd.cpu().double() - rr).abs() / rr).max() < 2e-3


SPADE_CASES = [
    # C, H, W, cond_hw, x_upsample, act
    (20, 12, 16, (48, 64), False, "none"),
    (40, 16, 16, (64, 64), False, "lrelu"),
    (24, 20, 20, (40, 40), False, "lrelu"),     # 20x20: ragged 16-px tiles
    (16, 10, 12, (40, 48), True, "lrelu"),      # x stored at 5x6, read through the folded upsample
    (640, 5, 5, (640, 640), False, "lrelu"),    # head_0 shape: channel split across workgroups
    (20, 40, 48, (40, 48), False, "none"),      # cond already at x resolution
]


def _set_variant(v):
    import ctypes
    from climategan_amd import _lib
    _lib.load().cgan_debug_set_spade_variant(ctypes.c_int(v))


def _set_waves(n):
    import ctypes
    from climategan_amd import _lib
    _lib.load().cgan_debug_set_spade_waves(ctypes.c_int(n))


@pytest.mark.usefixtures("dev_lib")
@pytest.mark.parametrize("waves", [8, 4])
@pytest.mark.parametrize("variant", [1, 2, 3, 4, 5])
@pytest.mark.parametrize("case", [(40, 36, 52, (72, 104), False, "lrelu"), (88, 18, 16, (36, 32), True, "none")])
def test_spade_fused_tile_variants(variant, case, waves):
    """Every channel-tiles-per-workgroup instantiation (NCT = 1..5) of both kernels -- the wave-specialised one (8 waves:
    consumers + producers, the default) and the 4-wave one -- on multi-tile, ragged images, including a channel count
    (C=88 -> 11 channel tiles) that leaves a partial last chunk; bf16 and fp16."""
    _set_variant(variant)
    _set_waves(waves)
    try:
        test_spade_fused(torch.float16, case)
        test_spade_fused(torch.bfloat16, case)
    finally:
        _set_variant(0)
        _set_waves(8)


@pytest.mark.usefixtures("dev_lib")
def test_spade_kernels_agree_bitwise():
    """The specialised and the 4-wave kernel run the same arithmetic in the same order: identical bits."""
    from climategan_amd import fill, ops
    from helpers import spade_shapes, t
    C, H, W = 40, 48, 64
    dt = torch.bfloat16
    sd_np = fill.fill_state_dict(spade_shapes("s", C, 3), seed=77)
    pk = ops.pack_spade_weights(*[t(sd_np["s." + k]).cuda() for k in (
        "mlp_shared.0.weight", "mlp_shared.0.bias", "mlp_gamma.weight", "mlp_gamma.bias", "mlp_beta.weight",
        "mlp_beta.bias")], dt)
    x = ops.nchw_to_nhwc(t(fill.uniform((2, C, H, W), 5, -2, 2)).cuda(), dt)
    cond = ops.nchw_to_nhwc(t(fill.uniform((2, 3, 96, 128), 6)).cuda(), dt, cs=4)
    mean, rstd = ops.instnorm_stats(x)
    outs = []
    for waves in (8, 4):
        _set_waves(waves)
        outs.append(ops.spade_fused(x, mean, rstd, cond, pk, act=ops.ACT_LRELU).t.clone())
    _set_waves(8)
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("case", SPADE_CASES)
def test_spade_fused(dt, case):
    from climategan_amd import ops
    from helpers import spade_shapes
    C, H, W, chw, ups, act = case
    B = 2
    sd_np = fill.fill_state_dict(spade_shapes("s", C, 3), seed=C + H)
    sd = {k: q(v, dt) if "weight" in k else torch.from_numpy(v) for k, v in sd_np.items()}
    xs = q(fill.uniform((B, C, H // 2, W // 2) if ups else (B, C, H, W), 55, -2, 2), dt)
    seg = q(fill.uniform((B, 3) + chw, 56), dt)
    x_full = cpu_ref.nearest_resize(xs, (H, W)) if ups else xs
    # oracle with the hidden map rounded to the compute type, as the kernel stores it in LDS
    normalized = cpu_ref.instance_norm(x_full)
    s = cpu_ref.nearest_resize(seg, (H, W))
    actv = F.relu(F.conv2d(s, sd["s.mlp_shared.0.weight"], sd["s.mlp_shared.0.bias"], padding=1)).to(dt).float()
    gamma = F.conv2d(actv, sd["s.mlp_gamma.weight"], sd["s.mlp_gamma.bias"], padding=1)
    beta = F.conv2d(actv, sd["s.mlp_beta.weight"], sd["s.mlp_beta.bias"], padding=1)
    ref = normalized * (1 + gamma) + beta
    if act == "lrelu":
        ref = F.leaky_relu(ref, 0.2)
    g = {k: v.cuda() for k, v in sd.items()}
    pk = ops.pack_spade_weights(g["s.mlp_shared.0.weight"], g["s.mlp_shared.0.bias"], g["s.mlp_gamma.weight"],
                                g["s.mlp_gamma.bias"], g["s.mlp_beta.weight"], g["s.mlp_beta.bias"], dt)
    xn = to_nhwc(xs, dt)
    mean, rstd = ops.instnorm_stats(xn)
    y = ops.spade_fused(xn, mean, rstd, to_nhwc(seg, dt, cs=4), pk, act=ops.ACT_LRELU if act == "lrelu" else ops.ACT_NONE,
                        x_upsample=ups)
    assert y.t.shape == (B, H, W, ops.cs8(C))
    assert_close(back(y), ref, dt, "spade %s" % (case,))
    if ops.cs8(C) != C:
        assert y.t[..., C:].abs().max().item() == 0
    # the training form of the launch (cgan_spade_fused_fwd_train): the same y bit for bit, plus the modulation map gamma the
    # backward consumes
    y2, gam = ops.spade_fused(xn, mean, rstd, to_nhwc(seg, dt, cs=4), pk, act=ops.ACT_LRELU if act == "lrelu" else ops.ACT_NONE,
                              x_upsample=ups, want_gamma=True)
    assert torch.equal(y2.t, y.t)
    assert_close(back(gam), gamma, dt, "spade gamma %s" % (case,))
    if ops.cs8(C) != C:
        assert gam.t[..., C:].abs().max().item() == 0


@pytest.mark.parametrize("rows,cols", [(20, 360), (640, 5760), (1, 8192), (64, 64)])
def test_spectral_norm(rows, cols):
    from climategan_amd import ops
    w = torch.from_numpy(fill.uniform((rows, cols), 1, -0.1, 0.1))
    u = cpu_ref.l2normalize(torch.from_numpy(fill.uniform((rows,), 2)))
    v = cpu_ref.l2normalize(torch.from_numpy(fill.uniform((cols,), 3)))
    _, u_ref, v_ref, s_ref = cpu_ref.spectral_norm_step(w.double(), u.double(), v.double())
    ug, vg = u.cuda().clone(), v.cuda().clone()
    sigma = ops.spectral_norm_power_iter(w.cuda(), ug, vg)
    assert (ug.cpu().double() - u_ref).abs().max() < 1e-5
    assert (vg.cpu().double() - v_ref).abs().max() < 1e-5
    assert abs(sigma.item() - s_ref.item()) < 1e-5 * max(1.0, abs(s_ref.item()))


@pytest.mark.parametrize("dt", DTYPES)
def test_resize_and_avgpool(dt):
    from climategan_amd import ops
    x = q(fill.uniform((2, 4, 17, 23), 5), dt)
    xn = to_nhwc(x, dt)
    y = back(ops.avgpool3x3s2(xn))
    ref = F.avg_pool2d(x, 3, stride=2, padding=1, count_include_pad=False)
    assert_close(y, ref, dt, "avgpool")
    x3 = q(fill.uniform((2, 3, 64, 128), 6), dt)
    y = back(ops.resize_nearest(to_nhwc(x3, dt, cs=4), (2, 4), cs_out=8))
    assert torch.equal(y, cpu_ref.nearest_resize(x3, (2, 4)))


def test_errors_are_loud():
    from climategan_amd import ops
    x = to_nhwc(q(fill.uniform((1, 8, 4, 4), 1), torch.float16), torch.float16)
    w = torch.zeros(8, 16, 3, 3, device="cuda")
    pw = ops.pack_conv_weight(w, None, torch.float16)
    with pytest.raises(RuntimeError):
        ops.conv2d(x, pw, pad=1)  # channel mismatch
    with pytest.raises(RuntimeError):
        ops.nchw_to_nhwc(torch.zeros(1, 3, 4, 4), torch.float16)  # CPU tensor: no CPU path


def test_spectral_norm_group_equals_per_layer():
    """Batched power iteration + batched pack (5 launches for all layers) is bit-identical to the per-layer path."""
    from climategan_amd import ops
    shapes = [(20, 40, 3, 3), (640, 64, 3, 3), (24, 40, 1, 1), (1, 32, 4, 4)]
    params_a, params_b = [], []
    for i, shp in enumerate(shapes):
        w = torch.from_numpy(fill.uniform(shp, 10 + i, -0.1, 0.1)).cuda()
        u = cpu_ref.l2normalize(torch.from_numpy(fill.uniform((shp[0],), 20 + i))).cuda()
        v = cpu_ref.l2normalize(torch.from_numpy(fill.uniform((shp[1] * shp[2] * shp[3],), 30 + i))).cuda()
        b = torch.from_numpy(fill.uniform((shp[0],), 40 + i)).cuda() if i != 2 else None
        params_a.append((w, u.clone(), v.clone(), b))
        params_b.append((w, u.clone(), v.clone(), b))
    grp = ops.SpectralNormGroup(params_a, torch.float16)
    for _ in range(2):  # two forwards: state carries over
        packed = grp.step()
        for (w, u, v, b), pk in zip(params_b, packed):
            sigma = ops.spectral_norm_power_iter(w, u, v)
            ref = ops.pack_conv_weight(w, b, torch.float16, sigma)
            assert torch.equal(ref.w, pk.w) and torch.equal(ref.bias, pk.bias)
        for (wa, ua, va, _), (wb, ub, vb, _) in zip(params_a, params_b):
            assert torch.equal(ua, ub) and torch.equal(va, vb)


LDS_CONV_CASES = [
    # cin, cout, H, W, in_upsample, residual ("none" | "same" | "ups"), act
    (20, 20, 40, 48, False, "same", "lrelu"),     # final_spade conv_1 shape class: 24-channel storage both sides
    (40, 20, 33, 47, False, "none", "none"),      # ragged tiles, cin 40 -> 2 chunks (second one 8/32 full)
    (64, 88, 32, 32, False, "none", "none"),      # 6 channel tiles -> 2 workgroup chunks of 3
    (96, 40, 36, 36, True, "ups", "none"),        # input and residual read through the folded x2 upsample
    (8, 3, 64, 40, False, "none", "tanh"),        # conv_img shape class
]


@pytest.mark.usefixtures("dev_lib")
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("case", LDS_CONV_CASES)
def test_conv3x3_lds_tiled(dt, case):
    """The LDS-tiled 3x3 kernel (conv3x3_lds.hip, used from 32x32 up) against the oracle, and bit-for-bit against
    nothing -- but within tolerance of the general gather kernel forced through the debug knob."""
    import ctypes
    from climategan_amd import _lib, ops
    cin, cout, H, W, ups, resmode, act = case
    B = 2
    hs, ws = (H // 2, W // 2) if ups else (H, W)
    x = q(fill.uniform((B, cin, hs, ws), 500 + cin), dt)
    w = q(fill.uniform((cout, cin, 3, 3), 600 + cout, -0.2, 0.2), dt)
    b = torch.from_numpy(fill.uniform((cout,), 700 + cout))
    xf = cpu_ref.nearest_resize(x, (H, W)) if ups else x
    ref = F.conv2d(xf, w, b, padding=1)
    res = None
    if resmode != "none":
        rs = (H // 2, W // 2) if resmode == "ups" else (H, W)
        res = q(fill.uniform((B, cout) + rs, 800 + cout), dt)
        ref = ref + (cpu_ref.nearest_resize(res, (H, W)) if resmode == "ups" else res)
    if act == "lrelu":
        ref = F.leaky_relu(ref, 0.2)
    elif act == "tanh":
        ref = torch.tanh(ref)
    pw = ops.pack_conv_weight(w.cuda(), b.cuda(), dt)
    kw = dict(pad=1, act={"none": ops.ACT_NONE, "lrelu": ops.ACT_LRELU, "tanh": ops.ACT_TANH}[act],
              residual=to_nhwc(res, dt) if res is not None else None, in_upsample=ups,
              residual_upsample=(resmode == "ups"))
    y = ops.conv2d(to_nhwc(x, dt), pw, **kw)
    assert_close(back(y), ref, dt, "conv3x3 LDS %s" % (case,))
    if ops.cs8(cout) != cout:
        assert y.t[..., cout:].abs().max().item() == 0
    lib = _lib.load()
    lib.cgan_debug_set_conv_kernel(ctypes.c_int(1))
    try:
        y2 = ops.conv2d(to_nhwc(x, dt), pw, **kw)
    finally:
        lib.cgan_debug_set_conv_kernel(ctypes.c_int(0))
    assert_close(back(y2), ref, dt, "conv3x3 gather %s" % (case,))


LDS_PAD_CASES = [
    # cin, cout, H, W, pad, reflect, act: the mask / depth decoders' reflect-padded 3x3 convs (reference blocks.py:21-78,
    # masker.py:45-107) and the 'full' convs that are the data gradients of their pad-0 form
    (16, 8, 48, 40, 1, True, "lrelu"),
    (8, 1, 64, 33, 1, True, "none"),
    (128, 32, 37, 45, 1, True, "lrelu"),
    (64, 32, 32, 32, 1, True, "none"),
    (8, 16, 34, 50, 2, False, "none"),            # 'full': 36 x 52 out
    (32, 64, 40, 40, 2, False, "none"),
    (24, 24, 40, 44, 0, False, "none"),           # 'valid': 38 x 42 out
]


@pytest.mark.usefixtures("dev_lib")
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("case", LDS_PAD_CASES)
def test_conv3x3_lds_tiled_padding_modes(dt, case):
    """Round 5: the tiled 3x3 kernel with reflect padding and with zero padding 0 / 2 (halo origin shifted, input and output
    extents differ) against torch, against the general kernel, and that the dispatcher really takes it."""
    import ctypes
    from climategan_amd import _lib, ops
    cin, cout, H, W, pad, reflect, act = case
    B = 2
    x = q(fill.uniform((B, cin, H, W), 510 + cin + H), dt).requires_grad_(True)
    w = q(fill.uniform((cout, cin, 3, 3), 610 + cout, -0.2, 0.2), dt)
    b = torch.from_numpy(fill.uniform((cout,), 710 + cout))
    xin = F.pad(x, (1,) * 4, mode="reflect") if reflect else x
    ref0 = F.conv2d(xin, w, b, padding=0 if reflect else pad)
    ref = F.leaky_relu(ref0, 0.2) if act == "lrelu" else ref0
    lib = _lib.load()
    pmode = ops.PAD_REFLECT if reflect else ops.PAD_ZERO
    d = ops._conv_desc(ops._DT[dt], B, H, W, cin, cout, 3, 3, 1, pad, 1, pmode)
    assert lib.cgan_conv2d_kernel_kind(ctypes.byref(d), ctypes.c_int32(0)) == 1, "must run on the tiled 3x3 kernel"
    pw = ops.pack_conv_weight(w.cuda(), b.cuda(), dt)
    kw = dict(pad=pad, pad_mode=pmode, act=ops.ACT_LRELU if act == "lrelu" else ops.ACT_NONE)
    xg = to_nhwc(x.detach(), dt)
    y = ops.conv2d(xg, pw, **kw)
    assert y.t.shape == (B, ref.shape[2], ref.shape[3], ops.cs8(cout))
    assert_close(back(y), ref.detach(), dt, "conv3x3 LDS padding %s" % (case,))
    if ops.cs8(cout) != cout:
        assert y.t[..., cout:].abs().max().item() == 0
    lib.cgan_debug_set_conv_kernel(ctypes.c_int(1))
    try:
        y2 = ops.conv2d(xg, pw, **kw)
    finally:
        lib.cgan_debug_set_conv_kernel(ctypes.c_int(0))
    assert_close(back(y2), ref.detach(), dt, "conv3x3 gather padding %s" % (case,))
    # the data gradient (reflect: pad-0 gradient over the padded extent = a 'full' conv on the tiled kernel, folded back)
    dy = q(fill.uniform(tuple(ref0.shape), 810 + cout), dt)
    ref0.backward(dy)
    dx = ops.conv2d_bwd_data(to_nhwc(dy, dt), w.cuda(), (B, H, W), pad=pad, pad_mode=pmode)
    assert_close(back(dx), x.grad, dt, "conv3x3 LDS padding, data gradient %s" % (case,))


GEMM_CONV_CASES = [
    # cin, cout, k, stride, pad, dil, pad_mode, B, H, W, residual, act      (all hit conv_gemm.hip: cin % 32 == 0, cout >= 64)
    (256, 256, 1, 1, 0, 1, "zero", 2, 80, 80, False, "relu"),       # ResNet 1x1 (128x256 tile config)
    (256, 1024, 1, 1, 0, 1, "zero", 2, 40, 48, True, "relu"),       # bottleneck expand + residual + ReLU
    (256, 256, 3, 1, 2, 2, "zero", 2, 40, 40, False, "relu"),       # layer3 dilated 3x3
    (128, 128, 3, 2, 1, 1, "zero", 2, 81, 95, False, "none"),       # strided 3x3, ragged
    (256, 512, 1, 2, 0, 1, "zero", 2, 80, 80, False, "none"),       # strided 1x1 downsample
    (512, 64, 1, 1, 0, 1, "zero", 2, 64, 64, False, "lrelu"),       # 64-cout config (1 x 4 waves)
    (64, 64, 3, 1, 1, 1, "reflect", 2, 48, 80, False, "lrelu"),     # mask decoder: reflect pad
    (96, 72, 3, 1, 4, 4, "zero", 3, 37, 53, True, "none"),          # ragged everything: cout 72 (pad tile), npix % 256 != 0
    (2048, 256, 3, 1, 6, 6, "zero", 1, 48, 48, False, "none"),      # ASPP branch: K = 18432
]


@pytest.mark.usefixtures("dev_lib")
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("case", GEMM_CONV_CASES)
def test_conv_gemm_wide_layers(dt, case):
    """The LDS-tiled implicit-GEMM kernel for wide layers (conv_gemm.hip) against torch fp32 on 16-bit-rounded
    operands, and the general gather kernel (forced through the debug knob) on the same case."""
    import ctypes
    from climategan_amd import _lib, ops
    cin, cout, k, stride, pad, dil, pmode, B, H, W, with_res, act = case
    x = q(fill.uniform((B, cin, H, W), 900 + cin), dt)
    bound = 1.0 / np.sqrt(cin * k * k)
    w = q(fill.uniform((cout, cin, k, k), 1000 + cout, -bound, bound), dt)
    b = torch.from_numpy(fill.uniform((cout,), 1100 + cout))
    xin = F.pad(x, (pad,) * 4, mode="reflect") if pmode == "reflect" else x
    ref = F.conv2d(xin, w, b, stride=stride, padding=0 if pmode == "reflect" else pad, dilation=dil)
    res = None
    if with_res:
        res = q(fill.uniform(tuple(ref.shape), 1200 + cout), dt)
        ref = ref + res
    ref = {"none": lambda v: v, "relu": F.relu, "lrelu": lambda v: F.leaky_relu(v, 0.2)}[act](ref)
    pw = ops.pack_conv_weight(w.cuda(), b.cuda(), dt)
    kw = dict(stride=stride, pad=pad, dilation=dil, pad_mode=ops.PAD_REFLECT if pmode == "reflect" else ops.PAD_ZERO,
              act={"none": ops.ACT_NONE, "relu": ops.ACT_RELU, "lrelu": ops.ACT_LRELU}[act],
              residual=to_nhwc(res, dt) if res is not None else None)
    xg = to_nhwc(x, dt)
    y = ops.conv2d(xg, pw, **kw)
    assert y.t.shape == (B, ref.shape[2], ref.shape[3], ops.cs8(cout))
    assert_close(back(y), ref, dt, "conv GEMM %s" % (case,))
    if ops.cs8(cout) != cout:
        assert y.t[..., cout:].abs().max().item() == 0
    lib = _lib.load()
    lib.cgan_debug_set_conv_kernel(ctypes.c_int(1))
    try:
        y2 = ops.conv2d(xg, pw, **kw)
    finally:
        lib.cgan_debug_set_conv_kernel(ctypes.c_int(0))
    assert_close(back(y2), ref, dt, "conv gather %s" % (case,))
    if k == 1:   # same MFMA k-order in both kernels for 1x1 -> they agree to the last bit (k > 1: taps innermost here)
        lib.cgan_debug_set_conv_kernel(ctypes.c_int(4))      # (without round 5's split-K launches: K slices sum in another order)
        try:
            y3 = ops.conv2d(xg, pw, **kw)
        finally:
            lib.cgan_debug_set_conv_kernel(ctypes.c_int(0))
        assert torch.equal(y3.t, y2.t)


def test_conv_workspace_bindings_are_bounded_and_survive_many_streams():
    """ops._conv_ws (advisor, round 5): a process that walks through many streams must neither fail ("more than 16 streams")
    nor pin 64 MiB per stream for ever, and releasing a binding must happen in the LIBRARY that holds it -- the product and the
    development build keep separate tables: an unbind sent to the wrong one silently moves a layer from the split-K GEMM to
    the general kernel (another fp32 summation order; round 6 saw exactly that as a one-level difference in a smog image,
    inside the full suite only).  A split-K layer (640 -> 640 3x3 at 10 x 10, painter.py:149-160) on 12 streams, on the
    development library in between, and on the first stream again: the same bits every time."""
    from climategan_amd import _lib, ops

    dt = torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(5)
    x = ops.NHWC(torch.randn(2, 10, 10, 640, device="cuda", generator=g).to(dt), 640)
    w = torch.randn(640, 640, 3, 3, device="cuda", generator=g) * 0.02
    pw = ops.pack_conv_weight(w, None, dt)
    y0 = ops.conv2d(x, pw, pad=1).t.clone()
    torch.cuda.synchronize()
    for i in range(12):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            y = ops.conv2d(x, pw, pad=1).t
        st.synchronize()
        assert torch.equal(y, y0), "stream %d" % i
        if i == 5:
            try:
                _lib.load_dev()                                       # its own workspace table: its own bindings
                pw_dev = ops.pack_conv_weight(w, None, dt)
                yd = ops.conv2d(x, pw_dev, pad=1).t
                torch.cuda.synchronize()
                assert torch.equal(yd, y0)
            finally:
                _lib.use_product()
    assert torch.equal(ops.conv2d(x, pw, pad=1).t, y0)                # the calling stream's binding is still the product's
    assert len(ops._CONV_WS) <= ops.CONV_WS_KEEP
    held = sum(b.numel() for _l, b in ops._CONV_WS.values() if b is not None)
    assert held <= ops.CONV_WS_KEEP * ops.CONV_WS_BYTES
