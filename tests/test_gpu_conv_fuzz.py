"""Seeded random-shape sweep of the convolution entry points (forward, data gradient, weight / bias gradient) through
the C ABI against torch fp32 on 16-bit-rounded operands.  The kernel families pick different code paths by shape --
general gather / 3x3 LDS tile / wide-layer GEMM, zero-insertion-free parity classes for strided data gradients, tap
folding (cin_s <= 32), wave-uniform vs per-lane addressing (w_out % 8), the cooperative 128 x 128 weight-gradient tile,
workspace vs atomic reduction, out-of-range buffer lanes for padding / tails -- so the sweep draws ragged sizes, odd
channel counts, strides, dilations and paddings, and repeats the weight gradient with each development knob that forces
or disables a path."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from climategan_amd import fill

pytestmark = pytest.mark.gpu


def q(a, dt):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dt).float()


def rel_err(got, ref):
    return (got - ref).abs().max().item() / max(ref.abs().max().item(), 1e-12)


def draw_cases(n, seed):
    rng = np.random.RandomState(seed)
    cases = []
    while len(cases) < n:
        k = int(rng.choice([1, 3, 3, 3, 4, 5, 7]))
        stride = int(rng.choice([1, 1, 1, 2])) if k > 1 else int(rng.choice([1, 1, 2]))
        dil = int(rng.choice([1, 1, 2, 3])) if (k == 3 and stride == 1) else 1
        pad = int(rng.choice([0, dil * (k // 2), 1])) if k > 1 else int(rng.choice([0, 0, 1]))
        cin = int(rng.choice([1, 3, 4, 8, 11, 16, 20, 24, 32, 40, 64, 96, 128, 160, 256]))
        cout = int(rng.choice([1, 3, 8, 20, 40, 64, 80, 128, 130, 256]))
        B = int(rng.choice([1, 2, 3]))
        H = int(rng.choice([5, 8, 9, 16, 17, 24, 31, 32, 40]))
        W = int(rng.choice([5, 8, 11, 16, 24, 25, 32, 40, 48]))
        eff = dil * (k - 1) + 1
        if H + 2 * pad < eff or W + 2 * pad < eff:
            continue
        if cin * cout * k * k * B * H * W > 3e9:
            continue
        cases.append((cin, cout, k, stride, pad, dil, B, H, W))
    return cases


CASES = draw_cases(36, 20240928) + [
    # shapes that satisfy the cooperative weight-gradient tile's conditions (even block counts, w_out % 8 == 0)
    (128, 128, 3, 1, 1, 1, 2, 16, 16), (256, 128, 1, 1, 0, 1, 2, 8, 24), (8, 128, 3, 1, 1, 1, 2, 16, 16),
    (128, 256, 3, 1, 2, 2, 1, 24, 16), (128, 128, 4, 2, 1, 1, 2, 32, 32),
]


@pytest.mark.usefixtures("dev_lib")
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", CASES)
def test_conv_forward_dgrad_wgrad(dt, case):
    from climategan_amd import _lib, ops
    cin, cout, k, stride, pad, dil, B, H, W = case
    tol16 = 1e-3 if dt == torch.float16 else 8e-3
    x = q(fill.uniform((B, cin, H, W), 100 + cin + H), dt).requires_grad_(True)
    bound = 1.0 / np.sqrt(cin * k * k)
    w = q(fill.uniform((cout, cin, k, k), 200 + cout + k, -bound, bound), dt).requires_grad_(True)
    b = torch.from_numpy(fill.uniform((cout,), 300 + cout, -bound, bound)).requires_grad_(True)
    y = F.conv2d(x, w, b, stride=stride, padding=pad, dilation=dil)
    dy = q(fill.uniform(tuple(y.shape), 400 + cout + W), dt)
    y.backward(dy)

    xg = ops.nchw_to_nhwc(x.detach().cuda(), dt)
    dyg = ops.nchw_to_nhwc(dy.cuda(), dt)
    # forward (bias in the epilogue)
    pw = ops.pack_conv_weight(w.detach().cuda(), b.detach().cuda(), dt)
    yg = ops.conv2d(xg, pw, stride=stride, pad=pad, dilation=dil)
    assert rel_err(ops.nhwc_to_nchw(yg).cpu(), y.detach()) <= tol16, "forward"
    if ops.cs8(cout) != cout:
        assert yg.t[..., cout:].abs().max().item() == 0
    # data gradient
    dx = ops.conv2d_bwd_data(dyg, w.detach().cuda(), (B, H, W), stride=stride, pad=pad, dilation=dil)
    assert rel_err(ops.nhwc_to_nchw(dx).cpu(), x.grad) <= tol16, "dgrad"
    # weight / bias gradient under every path selector: default; no tap folding; per-lane addressing; cooperative tile
    # forced on small shapes; workspace-free atomic reduction
    lib = _lib.load()
    # (name, debug bits, cooperative-tile pixel threshold, workspace, cooperative stage pixels, forced pixel splits)
    variants = [("default", 0, -1, True, 0, 0), ("no fold", 4, -1, True, 0, 0), ("per-lane addressing", 16, -1, True, 0, 0),
                ("cooperative tile", 0, 0, True, 64, 0), ("cooperative tile, 32-pixel stages", 0, 0, True, 32, 0),
                ("cooperative tile, 32-pixel stages, 3 splits", 0, 0, True, 32, 3), ("one split", 0, -1, True, 0, 1),
                ("5 splits", 0, -1, True, 0, 5), ("single-wave tile", 8, -1, True, 0, 0),
                ("atomic reduction", 0, -1, False, 0, 0), ("atomic reduction, cooperative tile", 0, 0, False, 32, 2)]
    try:
        for name, dbg, coop_min, use_ws, chunk, splits in variants:
            lib.cgan_debug_set_wgrad(ctypes.c_int(-splits), ctypes.c_int(dbg))
            lib.cgan_debug_set_wgrad_coop_chunk(ctypes.c_int(chunk))
            lib.cgan_debug_set_wgrad_coop_min_pixels(ctypes.c_int(coop_min if coop_min >= 0 else 32768))
            dw, db = ops.conv2d_bwd_weight(xg, dyg, tuple(w.shape), stride=stride, pad=pad, dilation=dil,
                                           use_workspace=use_ws)
            assert rel_err(dw.cpu(), w.grad) <= 3e-4, "wgrad (%s)" % name
            assert rel_err(db.cpu(), b.grad) <= 3e-4, "bias grad (%s)" % name
        # the bias gradient as a separate channel-sum pass (what the kernels above do with a constant-one GEMM column)
        lib.cgan_debug_set_wgrad(ctypes.c_int(0), ctypes.c_int(0))
        lib.cgan_debug_set_wgrad_bias_fused(ctypes.c_int(0))
        dw, db = ops.conv2d_bwd_weight(xg, dyg, tuple(w.shape), stride=stride, pad=pad, dilation=dil)
        assert rel_err(db.cpu(), b.grad) <= 3e-4, "bias grad (separate pass)"
    finally:
        lib.cgan_debug_set_wgrad_bias_fused(ctypes.c_int(1))
        lib.cgan_debug_set_wgrad(ctypes.c_int(0), ctypes.c_int(0))
        lib.cgan_debug_set_wgrad_coop_chunk(ctypes.c_int(0))
        lib.cgan_debug_set_wgrad_coop_min_pixels(ctypes.c_int(32768))


@pytest.mark.usefixtures("dev_lib")
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", [(64, 40, 1, 256, 256), (128, 24, 2, 128, 256), (64, 130, 1, 256, 256), (192, 8, 1, 260, 288)])
def test_spatially_tiled_3x3_wgrad(dt, case):
    """The spatially tiled weight-gradient kernel of 3 x 3 / pad 1 layers on large maps (conv_wgrad_tile3x3_kernel: halo tile
    staged once for all nine taps): against torch's fp32 gradient of the 16-bit-rounded operands, and against the
    per-tap kernels it replaces (knob off), with the default and with a forced number of spatial splits."""
    from climategan_amd import _lib, ops
    cin, cout, B, H, W = case
    lib = _lib.load()
    x = q(fill.uniform((B, cin, H, W), 900 + cin), dt)
    w = q(fill.uniform((cout, cin, 3, 3), 901 + cout, -0.05, 0.05), dt).requires_grad_(True)
    y = F.conv2d(x, w, None, padding=1)
    dy = q(fill.uniform(tuple(y.shape), 902 + W), dt)
    y.backward(dy)
    xg, dyg = ops.nchw_to_nhwc(x.cuda(), dt), ops.nchw_to_nhwc(dy.cuda(), dt)
    try:
        lib.cgan_debug_set_wgrad_tile3x3(ctypes.c_int(0))
        dw_old, db_old = ops.conv2d_bwd_weight(xg, dyg, (cout, cin, 3, 3), pad=1)
        lib.cgan_debug_set_wgrad_tile3x3(ctypes.c_int(2))           # also where the cooperative kernel would be taken
        for splits in (0, 1, 5):
            lib.cgan_debug_set_wgrad(ctypes.c_int(-splits), ctypes.c_int(0))
            dw, db = ops.conv2d_bwd_weight(xg, dyg, (cout, cin, 3, 3), pad=1)
            assert rel_err(dw.cpu(), w.grad) <= 3e-4, ("tiled", splits)
            assert rel_err(dw.cpu(), dw_old.cpu()) <= 1e-5, ("tiled vs per-tap", splits)
            assert rel_err(db.cpu(), db_old.cpu()) <= 1e-5
            assert rel_err(db.cpu(), dy.float().sum((0, 2, 3))) <= 3e-4, ("tiled: bias gradient", splits)
    finally:
        lib.cgan_debug_set_wgrad(ctypes.c_int(0), ctypes.c_int(0))
        lib.cgan_debug_set_wgrad_tile3x3(ctypes.c_int(1))


def draw_mode_cases(n, seed):
    rng = np.random.RandomState(seed)
    out = []
    for _ in range(n):
        out.append((int(rng.choice([3, 8, 20, 40, 64, 128])), int(rng.choice([3, 20, 64, 128, 130])), int(rng.choice([1, 2])),
                    int(rng.choice([4, 8, 9, 12, 16])), int(rng.choice([4, 8, 10, 16, 20]))))
    return out


@pytest.mark.usefixtures("dev_lib")
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", draw_mode_cases(12, 99))
def test_wgrad_upsample_and_reflect_modes(dt, case):
    """The weight gradient's two other addressing modes: x stored at half resolution and read through the folded nearest
    x2 upsample (SPADE ResNet blocks), and reflect padding (mask / depth decoder convs), each under the per-lane and the
    wave-uniform addressing and with the cooperative tile forced."""
    from climategan_amd import _lib, ops
    cin, cout, B, h, w = case            # stored (half-resolution) extent for the upsample mode
    lib = _lib.load()
    xs = q(fill.uniform((B, cin, h, w), 500 + cin + h), dt)
    wt = q(fill.uniform((cout, cin, 3, 3), 600 + cout, -0.1, 0.1), dt).requires_grad_(True)
    # --- folded upsample
    xu = F.interpolate(xs, scale_factor=2).requires_grad_(False)
    y = F.conv2d(xu, wt, None, padding=1)
    dy = q(fill.uniform(tuple(y.shape), 700 + w), dt)
    y.backward(dy)
    ref_up = wt.grad.clone()
    wt.grad = None
    # --- reflect padding (on the stored extent)
    if h > 1 and w > 1:
        y2 = F.conv2d(F.pad(xs, (1, 1, 1, 1), mode="reflect"), wt, None)
        dy2 = q(fill.uniform(tuple(y2.shape), 800 + w), dt)
        y2.backward(dy2)
        ref_rf = wt.grad.clone()
    xg = ops.nchw_to_nhwc(xs.cuda(), dt)
    try:
        for dbg, coop_min, chunk in ((0, -1, 0), (16, -1, 0), (0, 0, 64), (0, 0, 32)):
            lib.cgan_debug_set_wgrad(ctypes.c_int(0), ctypes.c_int(dbg))
            lib.cgan_debug_set_wgrad_coop_chunk(ctypes.c_int(chunk))
            lib.cgan_debug_set_wgrad_coop_min_pixels(ctypes.c_int(coop_min if coop_min >= 0 else 32768))
            dw, _ = ops.conv2d_bwd_weight(xg, ops.nchw_to_nhwc(dy.cuda(), dt), (cout, cin, 3, 3), pad=1, want_bias=False,
                                          in_upsample=True)
            assert rel_err(dw.cpu(), ref_up) <= 3e-4, ("upsample", dbg, coop_min)
            if h > 1 and w > 1:
                dw, _ = ops.conv2d_bwd_weight(xg, ops.nchw_to_nhwc(dy2.cuda(), dt), (cout, cin, 3, 3), pad=1,
                                              want_bias=False, pad_mode=ops.PAD_REFLECT)
                assert rel_err(dw.cpu(), ref_rf) <= 3e-4, ("reflect", dbg, coop_min)
    finally:
        lib.cgan_debug_set_wgrad(ctypes.c_int(0), ctypes.c_int(0))
        lib.cgan_debug_set_wgrad_coop_chunk(ctypes.c_int(0))
        lib.cgan_debug_set_wgrad_coop_min_pixels(ctypes.c_int(32768))


WS_CASES = [
    # (cin, cout, k, stride, pad, dil, B, H, W, reflect, residual): wide layers the implicit-GEMM kernel takes
    (256, 256, 3, 1, 2, 2, 2, 40, 40, False, False), (512, 192, 3, 1, 1, 1, 1, 33, 47, False, True),
    (1024, 256, 1, 1, 0, 1, 2, 40, 40, False, True), (64, 320, 1, 1, 0, 1, 1, 64, 64, False, False),
    (128, 128, 3, 1, 1, 1, 1, 40, 56, True, False), (256, 512, 4, 2, 1, 1, 2, 48, 48, False, False),
    (512, 320, 3, 1, 3, 3, 1, 130, 131, False, True),        # the automatic K = 64 choice: ragged pixels, partial cout block
    (128, 512, 1, 1, 0, 1, 2, 40, 40, False, False), (256, 1024, 1, 1, 0, 1, 1, 80, 80, False, True),   # 2 / 4 K = 64 stages
    (320, 256, 3, 1, 1, 1, 2, 64, 72, False, False),           # five 64-channel chunks x nine taps, 18 pixel blocks
    (256, 64, 1, 1, 0, 1, 2, 96, 96, False, False), (512, 128, 1, 1, 0, 1, 2, 96, 101, False, True),   # direct 1x1 kernel: 4 / 8 cout tiles, ragged
    (1024, 256, 1, 1, 0, 1, 8, 80, 80, False, True), (2048, 200, 1, 1, 0, 1, 3, 80, 80, False, False),  # ... 16 tiles; 13 of 16
]


@pytest.mark.usefixtures("dev_lib")
@pytest.mark.parametrize("case", WS_CASES)
def test_k64_specialised_gemm_matches_the_plain_kernel(case):
    """The persistent producer / consumer variant of the wide-layer GEMM with K = 64 stages (automatic for long-K 3x3 layers
    in bf16, forced here through cgan_debug_set_gemm_ws; R5 DESIGN 4.2) sums K in another order than the plain kernel: same
    products, results within a bf16 rounding step -- ragged pixel counts, partial cout blocks, dilation, stride, residual +
    activation, several tiles per workgroup.  Layers it does not take (reflect padding, channel counts that are not whole
    64-chunks) fall through to the plain kernel."""
    from climategan_amd import _lib, ops

    cin, cout, k, stride, pad, dil, B, H, W, reflect, residual = case
    lib = _lib.load()
    dt = torch.bfloat16
    torch.manual_seed(1)
    x = ops.nchw_to_nhwc(torch.randn(B, cin, H, W, device="cuda"), dt)
    pw = ops.pack_conv_weight(torch.randn(cout, cin, k, k, device="cuda") * 0.05, torch.randn(cout, device="cuda"), dt)
    kw = dict(stride=stride, pad=pad, dilation=dil, act=ops.ACT_LRELU, slope=0.2,
              pad_mode=ops.PAD_REFLECT if reflect else ops.PAD_ZERO)
    try:
        lib.cgan_debug_set_gemm_ws(ctypes.c_int(1))
        y0 = ops.conv2d(x, pw, **kw)
        res = ops.NHWC(torch.randn_like(y0.t), cout) if residual else None
        if residual:
            y0 = ops.conv2d(x, pw, residual=res, **kw)
        for ws in (5, 6, 8, 10, 11, 12, 9, 0):    # 256 x 128, 128 x 256, the 256 x 256 kernel, the direct 1x1 kernel, the x-resident
            # 1x1 kernel, automatic without it, without any of the three, with all
            lib.cgan_debug_set_gemm_ws(ctypes.c_int(ws))
            y = ops.conv2d(x, pw, residual=res, **kw)
            y2 = ops.conv2d(x, pw, residual=res, **kw)
            assert torch.equal(y.t, y2.t)                                   # deterministic
            d = (y.t.float() - y0.t.float()).abs()
            assert (d <= 2.0 ** -7 * y0.t.float().abs() + 1e-3).all(), (ws, d.max().item())
    finally:
        lib.cgan_debug_set_gemm_ws(ctypes.c_int(0))


@pytest.mark.parametrize("case", [
    # (cin, cout, B, H, W): 4 / 2 / 1 / 3 64-channel chunks; ragged pixel blocks; partial cout blocks and tiles; one cout block
    (256, 1024, 2, 80, 80), (128, 512, 3, 37, 41), (64, 328, 2, 64, 64), (192, 256, 2, 50, 50), (256, 64, 2, 96, 96),
    (256, 1024, 8, 80, 80), (64, 256, 4, 160, 160),
])
@pytest.mark.usefixtures("dev_lib")
def test_x_resident_1x1_kernel_matches_the_plain_kernel(case):
    """conv1x1_xres.hip (short-K 1x1 layers without bias / activation / residual: the bottleneck expands in training mode and
    the reduce layers' data gradients; automatic in bf16 from 256 couts and 16384 pixels, forced here) against the plain
    kernel: same K order, so the same bits; also as a data gradient (transposed pack) and in fp16."""
    from climategan_amd import _lib, ops

    cin, cout, B, H, W = case
    lib = _lib.load()
    for dt in (torch.bfloat16, torch.float16):
        torch.manual_seed(2)
        x = ops.nchw_to_nhwc(torch.randn(B, cin, H, W, device="cuda"), dt)
        wt = torch.randn(cout, cin, 1, 1, device="cuda") * 0.05
        pw = ops.pack_conv_weight(wt, None, dt)
        dy = ops.nchw_to_nhwc(torch.randn(B, cout, H, W, device="cuda"), dt)
        try:
            lib.cgan_debug_set_gemm_ws(ctypes.c_int(1))
            y0 = ops.conv2d(x, pw)
            dx0 = ops.conv2d_bwd_data(dy, wt, (B, H, W)) if cout <= 256 else None
            lib.cgan_debug_set_gemm_ws(ctypes.c_int(11))
            y = [ops.conv2d(x, pw).t.clone() for _ in range(3)]
            assert torch.equal(y[0], y[1]) and torch.equal(y[0], y[2])
            assert torch.equal(y[0], y0.t), (dt, (y[0].float() - y0.t.float()).abs().max().item())
            if dx0 is not None:                   # cout -> cin as a 1x1 conv with K = cout <= 256
                dx = ops.conv2d_bwd_data(dy, wt, (B, H, W))
                assert torch.equal(dx.t, dx0.t)
        finally:
            lib.cgan_debug_set_gemm_ws(ctypes.c_int(0))


@pytest.mark.parametrize("case", [
    # (cin, cout, B, H, W, upsample, residual, stored channels): the SPADE shared conv (3 -> 128), VGG's first conv (3 -> 64),
    # ragged tiles, <= 2 / <= 4 / 8 cout tiles per workgroup, 13 tiles in two workgroups, the folded x2 upsample
    (3, 128, 2, 64, 64, False, False, 8), (3, 64, 1, 50, 37, False, False, 8), (4, 32, 2, 40, 48, False, True, 8),
    (1, 20, 1, 33, 65, False, False, 8), (3, 200, 1, 48, 48, False, False, 8), (3, 128, 1, 64, 64, True, False, 8),
])
@pytest.mark.usefixtures("dev_lib")
def test_folded_tap_3x3_kernel_for_few_input_channels(case):
    """conv3x3_c4_kernel (<= 4 input channels: the 9 taps folded into two MFMA k-steps instead of one k-step per tap) against
    the per-tap kernel it replaces (cgan_debug_set_conv3x3_c4(0)) and against torch: same products, another fp32 summation
    order, so within a rounding step of the 16-bit output."""
    import torch.nn.functional as F
    from climategan_amd import _lib, ops

    cin, cout, B, H, W, ups, residual, cs = case
    lib = _lib.load()
    for dt in (torch.bfloat16, torch.float16):
        torch.manual_seed(3)
        hx, wx = (H // 2, W // 2) if ups else (H, W)
        xf = torch.randn(B, cin, hx, wx, device="cuda").to(dt).float()
        wt = (torch.randn(cout, cin, 3, 3, device="cuda") * 0.2).to(dt).float()
        bias = torch.randn(cout, device="cuda")
        x = ops.nchw_to_nhwc(xf, dt, cs=cs)
        pw = ops.pack_conv_weight(wt, bias, dt)
        res = ops.NHWC(torch.randn(B, H, W, ops.cs8(cout), device="cuda").to(dt), cout) if residual else None
        kw = dict(pad=1, act=ops.ACT_LRELU, slope=0.2, in_upsample=ups, residual=res)
        try:
            lib.cgan_debug_set_conv3x3_c4(ctypes.c_int(0))
            y0 = ops.conv2d(x, pw, **kw)
            lib.cgan_debug_set_conv3x3_c4(ctypes.c_int(1))
            y = ops.conv2d(x, pw, **kw)
            y2 = ops.conv2d(x, pw, **kw)
        finally:
            lib.cgan_debug_set_conv3x3_c4(ctypes.c_int(1))
        assert torch.equal(y.t, y2.t)
        xin = F.interpolate(xf, scale_factor=2, mode="nearest") if ups else xf
        ref = F.conv2d(xin, wt, bias, padding=1)
        if residual:
            ref = ref + ops.nhwc_to_nchw(res)
        ref = F.leaky_relu(ref, 0.2)
        got = ops.nhwc_to_nchw(y)
        eps = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
        assert (got - ref).abs().max().item() <= 1.5 * eps * ref.abs().max().item() + 1e-3, dt
        d = (y.t.float() - y0.t.float()).abs()
        assert (d <= 2 * eps * y0.t.float().abs() + 1e-3).all(), (dt, d.max().item())
        assert y.t[..., cout:].abs().max().item() == 0 if y.t.shape[-1] > cout else True


# ---------------------------------------------------------------------------------------------------------------------
# round 5: parity-class data gradients and split-K launches on the LDS-tiled GEMM (conv_gemm_ext.hip)
# ---------------------------------------------------------------------------------------------------------------------
CLS_CASES = [
    # (cin, cout, k, stride, pad, B, H, W): strided convs whose data gradient has >= 64 output and % 32 input channels
    (64, 128, 4, 2, 1, 2, 32, 32),      # PatchGAN 64 -> 128 (discriminator.py:100-163)
    (128, 256, 4, 2, 1, 1, 24, 40),
    (256, 512, 4, 2, 1, 2, 10, 10),     # few pixels per class
    (64, 64, 4, 2, 1, 1, 17, 23),       # odd extents: classes of different sizes, rows no window reaches
    (128, 128, 3, 2, 1, 2, 33, 31),     # ResNet layer2 3x3 s2: 2 / 1 taps per axis by class
    (256, 512, 1, 2, 0, 2, 16, 16),     # 1x1 s2 shortcut: three classes without taps write zeros
    (72, 96, 4, 2, 1, 1, 12, 20),       # 72 dx channels (a padded cout tile)
]


def _kind(lib, dt, case, bwd):
    from climategan_amd import ops
    cin, cout, k, stride, pad, B, H, W = case
    d = ops._conv_desc(ops._DT[dt], B, H, W, cin, cout, k, k, stride, pad, 1, ops.PAD_ZERO, has_bias=not bwd)
    return lib.cgan_conv2d_kernel_kind(ctypes.byref(d), ctypes.c_int32(1 if bwd else 0))


@pytest.mark.usefixtures("dev_lib")
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", CLS_CASES)
def test_strided_dgrad_by_parity_classes_on_the_tiled_gemm(dt, case):
    from climategan_amd import _lib, ops
    cin, cout, k, stride, pad, B, H, W = case
    tol16 = 1e-3 if dt == torch.float16 else 8e-3
    lib = _lib.load()
    x = q(fill.uniform((B, cin, H, W), 2100 + cin + H), dt).requires_grad_(True)
    bound = 1.0 / np.sqrt(cin * k * k)
    w = q(fill.uniform((cout, cin, k, k), 2200 + cout + k, -bound, bound), dt)
    y = F.conv2d(x, w, None, stride=stride, padding=pad)
    dy = q(fill.uniform(tuple(y.shape), 2300 + cout + W), dt)
    y.backward(dy)
    dyg = ops.nchw_to_nhwc(dy.cuda(), dt)
    assert _kind(lib, dt, case, True) == 2, "this case must run on the tiled GEMM"
    dx = ops.conv2d_bwd_data(dyg, w.cuda(), (B, H, W), stride=stride, pad=pad)
    assert rel_err(ops.nhwc_to_nchw(dx).cpu(), x.grad) <= tol16
    if ops.cs8(cin) != cin:
        assert dx.t[..., cin:].abs().max().item() == 0
    lib.cgan_debug_set_conv_kernel(ctypes.c_int(4))
    try:
        assert _kind(lib, dt, case, True) == 0
        dx2 = ops.conv2d_bwd_data(dyg, w.cuda(), (B, H, W), stride=stride, pad=pad)
    finally:
        lib.cgan_debug_set_conv_kernel(ctypes.c_int(0))
    # same products, another fp32 summation order: within one rounding step of the general kernel's result
    assert rel_err(dx.t.float(), dx2.t.float()) <= (2 ** -10 if dt == torch.float16 else 2 ** -7)


SPLITK_CASES = [
    # (cin, cout, k, stride, pad, B, H, W, residual, act)
    (640, 640, 3, 1, 1, 2, 10, 10, True, "none"),     # Painter G_middle (painter.py:149-160)
    (640, 640, 3, 1, 1, 1, 5, 5, False, "lrelu"),
    (512, 512, 4, 1, 1, 2, 20, 20, False, "lrelu"),   # PatchGAN 512 -> 512 4x4 s1 (19 x 19 out)
    (256, 512, 4, 2, 1, 2, 20, 20, False, "none"),    # strided, 10 x 10 out
    (1280, 128, 3, 1, 1, 1, 10, 10, False, "relu"),   # long K, one cout block
    (512, 72, 3, 1, 1, 2, 9, 11, True, "none"),       # pad tile, ragged
    (512, 1, 4, 1, 1, 2, 19, 19, False, "none"),      # PatchGAN head (discriminator.py:163): one output channel
    (512, 1, 4, 2, 1, 2, 20, 20, False, "none"),
    (1280, 128, 3, 1, 1, 1, 40, 40, False, "none"),   # wide input on a map the tiled 3x3 kernel would otherwise take
]


@pytest.mark.usefixtures("dev_lib")
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", SPLITK_CASES)
def test_small_grids_run_as_k_slices_of_the_tiled_gemm(dt, case):
    from climategan_amd import _lib, ops
    cin, cout, k, stride, pad, B, H, W, with_res, act = case
    tol16 = 1e-3 if dt == torch.float16 else 8e-3
    lib = _lib.load()
    x = q(fill.uniform((B, cin, H, W), 3100 + cin + H), dt).requires_grad_(True)
    bound = 1.0 / np.sqrt(cin * k * k)
    w = q(fill.uniform((cout, cin, k, k), 3200 + cout + k, -bound, bound), dt)
    b = torch.from_numpy(fill.uniform((cout,), 3300 + cout, -bound, bound))
    y0 = F.conv2d(x, w, b, stride=stride, padding=pad)
    dy = q(fill.uniform(tuple(y0.shape), 3400 + cout + W), dt)
    y0.backward(dy)
    ref = y0.detach()
    res = None
    if with_res:
        res = q(fill.uniform(tuple(ref.shape), 3500 + cout), dt)
        ref = ref + res
    ref = {"none": lambda v: v, "relu": F.relu, "lrelu": lambda v: F.leaky_relu(v, 0.2)}[act](ref)
    xg = ops.nchw_to_nhwc(x.detach().cuda(), dt)
    dyg = ops.nchw_to_nhwc(dy.cuda(), dt)
    pw = ops.pack_conv_weight(w.cuda(), b.cuda(), dt)
    kw = dict(stride=stride, pad=pad, act={"none": ops.ACT_NONE, "relu": ops.ACT_RELU, "lrelu": ops.ACT_LRELU}[act],
              residual=ops.nchw_to_nhwc(res.cuda(), dt) if res is not None else None)
    yg = ops.conv2d(xg, pw, **kw)                    # binds this stream's workspace on first use
    assert _kind(lib, dt, case[:8], False) == 2, "this case must run as K slices of the tiled GEMM"
    assert rel_err(ops.nhwc_to_nchw(yg).cpu(), ref) <= tol16, "forward"
    if ops.cs8(cout) != cout:
        assert yg.t[..., cout:].abs().max().item() == 0
    yg_again = ops.conv2d(xg, pw, **kw)
    assert torch.equal(yg.t, yg_again.t), "the ordered reduce is deterministic"
    if stride == 1:
        dx = ops.conv2d_bwd_data(dyg, w.cuda(), (B, H, W), stride=stride, pad=pad)
        assert rel_err(ops.nhwc_to_nchw(dx).cpu(), x.grad) <= tol16, "dgrad"
    lib.cgan_debug_set_conv_kernel(ctypes.c_int(4))
    try:
        assert _kind(lib, dt, case[:8], False) != 2
        y2 = ops.conv2d(xg, pw, **kw)
    finally:
        lib.cgan_debug_set_conv_kernel(ctypes.c_int(0))
    assert rel_err(yg.t.float(), y2.t.float()) <= (2 ** -10 if dt == torch.float16 else 2 ** -7)


@pytest.mark.usefixtures("dev_lib")
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", [
    # (cin, cout, k, stride, pad, dil, B, H, W, reflect): block counts divisible by four on both sides, w_out % 8 == 0
    (256, 256, 3, 1, 2, 2, 2, 24, 16, False),     # ResNet layer3 (resnet101_v3.py:30-50)
    (256, 512, 1, 1, 0, 1, 2, 16, 24, False),
    (512, 256, 1, 1, 0, 1, 1, 40, 40, False),
    (256, 256, 3, 1, 1, 1, 1, 16, 16, True),      # reflect padding
    (8, 256, 4, 2, 1, 1, 2, 32, 32, False),       # folded taps: 4 tap slots
])
def test_weight_gradient_on_the_16_wave_tile(dt, case):
    """Round 5: conv_wgrad_coop_kernel on a 4 x 4 wave grid (256 x 256 tile, 16 waves, one workgroup per CU), forced through
    the development knob on small shapes, with the default and with given pixel splits: against torch's fp32 gradient of the
    16-bit-rounded operands and against the 2 x 2 grid it generalises (same products, same per-quadrant summation order for
    equal splits: bit-identical)."""
    from climategan_amd import _lib, ops
    cin, cout, k, stride, pad, dil, B, H, W, reflect = case
    lib = _lib.load()
    x = q(fill.uniform((B, cin, H, W), 5100 + cin + H), dt)
    bound = 1.0 / np.sqrt(cin * k * k)
    w = q(fill.uniform((cout, cin, k, k), 5200 + cout + k, -bound, bound), dt).requires_grad_(True)
    b = torch.from_numpy(fill.uniform((cout,), 5300 + cout, -bound, bound)).requires_grad_(True)
    xin = F.pad(x, (pad,) * 4, mode="reflect") if reflect else x
    y = F.conv2d(xin, w, b, stride=stride, padding=0 if reflect else pad, dilation=dil)
    dy = q(fill.uniform(tuple(y.shape), 5400 + cout + W), dt)
    y.backward(dy)
    xg, dyg = ops.nchw_to_nhwc(x.cuda(), dt), ops.nchw_to_nhwc(dy.cuda(), dt)
    kw = dict(stride=stride, pad=pad, dilation=dil, pad_mode=ops.PAD_REFLECT if reflect else ops.PAD_ZERO)
    try:
        lib.cgan_debug_set_wgrad_coop_min_pixels(ctypes.c_int(0))
        for splits in (0, 1, 3):
            lib.cgan_debug_set_wgrad(ctypes.c_int(-splits), ctypes.c_int(0))
            lib.cgan_debug_set_wgrad_coop_g(ctypes.c_int(4))
            dw4, db4 = ops.conv2d_bwd_weight(xg, dyg, tuple(w.shape), **kw)
            assert rel_err(dw4.cpu(), w.grad) <= 3e-4, ("16-wave tile", splits)
            assert rel_err(db4.cpu(), b.grad) <= 3e-4, ("16-wave tile, bias", splits)
            if splits:
                lib.cgan_debug_set_wgrad_coop_g(ctypes.c_int(2))
                dw2, _ = ops.conv2d_bwd_weight(xg, dyg, tuple(w.shape), **kw)
                assert torch.equal(dw4, dw2), ("4 x 4 vs 2 x 2 wave grid", splits)
    finally:
        lib.cgan_debug_set_wgrad_coop_g(ctypes.c_int(0))
        lib.cgan_debug_set_wgrad(ctypes.c_int(0), ctypes.c_int(0))
        lib.cgan_debug_set_wgrad_coop_min_pixels(ctypes.c_int(32768))


@pytest.mark.usefixtures("dev_lib")
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", [(4, 64, 2, 64, 64), (3, 64, 1, 96, 80), (1, 32, 2, 64, 96), (4, 128, 1, 66, 70)])
def test_first_layer_dgrad_as_a_subpixel_3x3_conv(dt, case):
    """Round 5: the data gradient of a 4x4 / stride-2 / pad-1 convolution with <= 4 input channels (the discriminators' first
    layer, reference discriminator.py:100-120) as ONE 3x3 convolution of dy with 16 (class, channel) outputs on the tiled 3x3
    kernel + a depth-to-space epilogue: against torch and against the parity-class form on the general kernel."""
    from climategan_amd import _lib, ops
    cin, cout, B, H, W = case
    lib = _lib.load()
    x = q(fill.uniform((B, cin, H, W), 6100 + cin + H), dt).requires_grad_(True)
    bound = 1.0 / np.sqrt(cin * 16)
    w = q(fill.uniform((cout, cin, 4, 4), 6200 + cout, -bound, bound), dt)
    y = F.conv2d(x, w, None, stride=2, padding=1)
    dy = q(fill.uniform(tuple(y.shape), 6300 + cout + W), dt)
    y.backward(dy)
    dyg = ops.nchw_to_nhwc(dy.cuda(), dt)
    assert _kind(lib, dt, (cin, cout, 4, 2, 1, B, H, W), True) == 1, "this case must run on the tiled 3x3 kernel"
    dx = ops.conv2d_bwd_data(dyg, w.cuda(), (B, H, W), stride=2, pad=1)
    assert rel_err(ops.nhwc_to_nchw(dx).cpu(), x.grad) <= (1e-3 if dt == torch.float16 else 8e-3)
    assert dx.t[..., cin:].abs().max().item() == 0
    lib.cgan_debug_set_conv_kernel(ctypes.c_int(4))
    try:
        assert _kind(lib, dt, (cin, cout, 4, 2, 1, B, H, W), True) == 0
        dx2 = ops.conv2d_bwd_data(dyg, w.cuda(), (B, H, W), stride=2, pad=1)
    finally:
        lib.cgan_debug_set_conv_kernel(ctypes.c_int(0))
    assert rel_err(dx.t.float(), dx2.t.float()) <= (2 ** -10 if dt == torch.float16 else 2 ** -7)


@pytest.mark.usefixtures("dev_lib")
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", [
    # (cin, cout, k, stride, pad, B, H, W, act)
    (3, 64, 7, 2, 3, 2, 96, 80, "none"),       # ResNet stem (resnet101_v3.py:176)
    (4, 64, 4, 2, 1, 2, 64, 64, "lrelu"),      # PatchGAN input conv (discriminator.py:100-120)
    (4, 64, 4, 2, 1, 1, 65, 71, "lrelu"),      # odd extents: a row / column no window reaches, ragged tiles
    (8, 40, 5, 1, 2, 2, 40, 48, "none"),       # stride 1, 40 outputs (a pad tile), all 8 channels live
    (1, 16, 2, 1, 0, 1, 33, 40, "relu"),       # one k-step, a single channel tile
    (6, 24, 6, 2, 2, 1, 70, 66, "none"),
])
def test_first_layer_convs_on_the_halo_tiled_kernel(dt, case):
    """Round 5: convolutions with <= 8 input channels and k != 3 (conv_smallcin_kernel: the input halo of a 16 x 16 output
    tile staged once in LDS, four taps per MFMA k-step) against torch, against the general gather kernel, and that the
    dispatcher takes the tiled family for them."""
    from climategan_amd import _lib, ops
    cin, cout, k, stride, pad, B, H, W, act = case
    lib = _lib.load()
    x = q(fill.uniform((B, cin, H, W), 8100 + cin + H), dt)
    bound = 1.0 / np.sqrt(cin * k * k)
    w = q(fill.uniform((cout, cin, k, k), 8200 + cout + k, -bound, bound), dt)
    b = torch.from_numpy(fill.uniform((cout,), 8300 + cout, -bound, bound))
    ref = F.conv2d(x, w, b, stride=stride, padding=pad)
    ref = {"none": lambda v: v, "relu": F.relu, "lrelu": lambda v: F.leaky_relu(v, 0.2)}[act](ref)
    assert _kind(lib, dt, (cin, cout, k, stride, pad, B, H, W), False) == 1, "this case must run on the tiled family"
    xg = ops.nchw_to_nhwc(x.cuda(), dt)
    pw = ops.pack_conv_weight(w.cuda(), b.cuda(), dt)
    kw = dict(stride=stride, pad=pad, act={"none": ops.ACT_NONE, "relu": ops.ACT_RELU, "lrelu": ops.ACT_LRELU}[act])
    y = ops.conv2d(xg, pw, **kw)
    assert y.t.shape == (B, ref.shape[2], ref.shape[3], ops.cs8(cout))
    assert rel_err(ops.nhwc_to_nchw(y).cpu(), ref) <= (1e-3 if dt == torch.float16 else 8e-3)
    if ops.cs8(cout) != cout:
        assert y.t[..., cout:].abs().max().item() == 0
    lib.cgan_debug_set_conv_kernel(ctypes.c_int(1))
    try:
        y2 = ops.conv2d(xg, pw, **kw)
    finally:
        lib.cgan_debug_set_conv_kernel(ctypes.c_int(0))
    assert rel_err(y.t.float(), y2.t.float()) <= (2 ** -10 if dt == torch.float16 else 2 ** -7)


@pytest.mark.parametrize("case", [
    # (n, h, w, cin, cout, k, stride, pad): the data-gradient launches that do not run the plain kernels
    (4, 160, 160, 4, 64, 4, 2, 1),        # sub-pixel form (first PatchGAN / ADVENT layer)
    (4, 80, 80, 64, 128, 4, 2, 1),        # parity classes on the tiled GEMM
    (4, 40, 40, 256, 512, 3, 2, 1),       # parity classes, 3x3 s2
    (2, 10, 10, 640, 640, 3, 1, 1),       # split-K (small grid, long K)
    (4, 20, 20, 512, 512, 4, 1, 1),       # split-K, 4x4 s1 (PatchGAN tail)
])
def test_data_gradient_launch_kinds_are_bitwise_reproducible(case):
    """Advisor (round 5): the sub-pixel, parity-class and split-K data-gradient launches are ordered reductions (no atomics):
    the same call twice -- and once more on another stream -- gives the same bits."""
    from climategan_amd import ops

    n, h, w, cin, cout, k, stride, pad = case
    dt = torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(41)
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    dy = ops.NHWC(torch.randn(n, ho, wo, ops.cs8(cout), device="cuda", generator=g).to(dt), cout)
    if ops.cs8(cout) != cout:
        dy.t[..., cout:] = 0
    wt = torch.randn(cout, cin, k, k, device="cuda", generator=g) * 0.05
    a = ops.conv2d_bwd_data(dy, wt, (n, h, w), stride=stride, pad=pad).t.clone()
    b = ops.conv2d_bwd_data(dy, wt, (n, h, w), stride=stride, pad=pad).t.clone()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        c = ops.conv2d_bwd_data(dy, wt, (n, h, w), stride=stride, pad=pad).t.clone()
    st.synchronize()
    assert torch.equal(a, b) and torch.equal(a, c)
    assert float(a.float().abs().max()) > 0
