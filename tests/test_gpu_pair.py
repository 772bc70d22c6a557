"""Split-precision maps (ops.PairMap, csrc/pair.hip, cgan_conv2d_nhwc_fwd_pair): every op of the inference-time Masker on
fp16 pairs ("pair16") and bf16 triples ("split24") against torch's fp32 / float64 ops on the same fp32 inputs."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DTS = [torch.bfloat16, torch.float16]


def _rand(shape, seed, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.randn(shape, device="cuda", generator=g) * scale


@pytest.mark.parametrize("dt", DTS)
def test_layout_round_trip_and_precision(dt):
    from climategan_amd import ops

    x = _rand((2, 19, 24, 20), 1, 3.0)
    p = ops.pair_from_nchw(x, dt)
    assert tuple(p.t.shape) == (2, 24, 20, ops.store_blocks(dt) * 24) and p.c == 19
    back = ops.nhwc_to_nchw(p)
    rel = ((back - x).abs() / x.abs().clamp_min(0.25)).max().item()        # (fp16 pairs: an absolute floor below |v| ~ 0.1)
    assert rel <= (2.0 ** -21 if dt == torch.float16 else 2.0 ** -23), rel
    assert torch.equal(ops.nhwc_to_nchw(ops.sigmoid(p)), torch.sigmoid(back)) or \
        (ops.nhwc_to_nchw(ops.sigmoid(p)) - torch.sigmoid(back.double()).float()).abs().max().item() <= 2e-7
    r = ops.pair_to_nhwc(p)
    assert isinstance(r, ops.NHWC) and torch.equal(r.t[..., :19], back.permute(0, 2, 3, 1).to(dt))
    assert bool((r.t[..., 19:] == 0).all())


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("case", [
    # (cin, cout, k, stride, pad, dil, reflect, in_upsample, residual, act)
    (24, 40, 3, 1, 1, 1, False, False, True, "relu"),
    (64, 32, 1, 1, 0, 1, False, False, False, "none"),
    (16, 24, 3, 1, 2, 2, False, False, False, "lrelu"),
    (32, 16, 3, 1, 1, 1, True, False, True, "lrelu"),
    (16, 8, 3, 1, 1, 1, True, True, False, "none"),
    (3, 64, 7, 2, 3, 1, False, False, False, "relu"),
    (130, 20, 3, 2, 1, 1, False, False, False, "none"),
])
def test_split_conv_matches_float64(dt, case):
    """hi W_hi + lo W_hi + hi W_lo (+ the bf16 triples' second-order terms) accumulated in fp32 by the MFMA kernel, against
    the float64 convolution of the same fp32 operands: a few 1e-7 of the output's scale -- fp32-grade."""
    from climategan_amd import ops

    cin, cout, k, stride, pad, dil, reflect, ups, with_res, act = case
    n, h, w = 2, 20, 28
    x = _rand((n, cin, h, w), 3)
    wt = _rand((cout, cin, k, k), 4, (2.0 / (cin * k * k)) ** 0.5)
    b = _rand((cout,), 5)
    p = ops.pair_from_nchw(x, dt)
    pw = ops.pack_conv_weight(wt, b, dt, pair=True)
    xr = F.interpolate(x.double(), scale_factor=2, mode="nearest") if ups else x.double()
    xp = F.pad(xr, (pad,) * 4, mode="reflect") if reflect else xr
    ref = F.conv2d(xp, wt.double(), b.double(), stride=stride, padding=0 if reflect else pad, dilation=dil)
    res = None
    if with_res:
        rt = _rand(tuple(ref.shape), 6)
        res = ops.pair_from_nchw(rt, dt)
        ref = ref + ops.nhwc_to_nchw(res).double()                  # (the residual as the split map carries it)
    ref = {"relu": torch.relu, "lrelu": lambda v: F.leaky_relu(v, 0.2), "none": lambda v: v}[act](ref)
    y = ops.conv2d(p, pw, stride=stride, pad=pad, dilation=dil, pad_mode=ops.PAD_REFLECT if reflect else ops.PAD_ZERO,
                   act={"relu": ops.ACT_RELU, "lrelu": ops.ACT_LRELU, "none": ops.ACT_NONE}[act], residual=res, in_upsample=ups)
    assert isinstance(y, ops.PairMap) and y.c == cout
    got = ops.nhwc_to_nchw(y).double()
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    assert err <= (1.5e-6 if dt == torch.float16 else 1e-6), err
    # the ordinary 16-bit conv on the same operands, for scale: three to four orders of magnitude further away
    y16 = ops.nhwc_to_nchw(ops.conv2d(ops.nchw_to_nhwc(x, dt), ops.pack_conv_weight(wt, b, dt), stride=stride, pad=pad,
                                      dilation=dil, pad_mode=ops.PAD_REFLECT if reflect else ops.PAD_ZERO, in_upsample=ups))
    if not with_res and act == "none":
        assert (y16.double() - ref).abs().max().item() / ref.abs().max().item() > 100 * err


@pytest.mark.parametrize("dt", DTS)
def test_glue_ops_match_torch_fp32(dt):
    from climategan_amd import ops

    x = _rand((2, 24, 21, 30), 7, 2.0)
    p = ops.pair_from_nchw(x, dt)
    xq = ops.nhwc_to_nchw(p)                                       # what the map carries (= x to fp32-grade precision)
    tol = 3e-6 if dt == torch.float16 else 6e-7

    def close(got, ref):
        return (got - ref).abs().max().item() <= tol * max(ref.abs().max().item(), 1.0)

    assert torch.equal(ops.nhwc_to_nchw(ops.maxpool3x3s2(p)), F.max_pool2d(xq, 3, 2, 1))
    for size, ac in (((40, 64), True), ((40, 64), False), ((11, 15), False), ((82, 82), True)):
        assert close(ops.nhwc_to_nchw(ops.resize_bilinear(p, size, align_corners=ac)),
                     F.interpolate(xq, size=size, mode="bilinear", align_corners=ac)), (size, ac)
    assert torch.equal(ops.nhwc_to_nchw(ops.resize_nearest(p, (42, 60))), F.interpolate(xq, scale_factor=2, mode="nearest"))
    assert torch.equal(ops.nhwc_to_nchw(ops.resize_nearest(p, (13, 17))), F.interpolate(xq, size=(13, 17), mode="nearest"))
    q = ops.pair_from_nchw(_rand((2, 24, 21, 30), 8), dt)
    assert close(ops.nhwc_to_nchw(ops.eltwise_mul(p, q)), xq * ops.nhwc_to_nchw(q))
    r = ops.pair_from_nchw(_rand((2, 13, 21, 30), 9), dt)
    cat = ops.concat_channels([p, q, r])
    assert cat.c == 61 and torch.equal(ops.nhwc_to_nchw(cat), torch.cat([xq, ops.nhwc_to_nchw(q), ops.nhwc_to_nchw(r)], 1))
    with pytest.raises(RuntimeError):
        ops.conv2d(p, ops.pack_conv_weight(_rand((8, 24, 1, 1), 10), None, dt))       # an ordinary operator on a split map


@pytest.mark.parametrize("dt", DTS)
def test_mask_decoder_glue_on_split_maps(dt):
    """Round 5, the SPADE mask decoder's glue in the split-precision mode against torch's fp32 ops on the same values: the
    bicubic resize of the depth map (depth.py:144-149), make_m_cond (generator.py:196-230: normalize | softmax | bilinear x with
    align_corners) and the plain normalise + LeakyReLU of an eval-mode BatchNorm (cgan_pair_spade_apply without gamma / beta)."""
    from climategan_amd import ops

    d = _rand((2, 1, 20, 24), 21, 2.0)
    pd = ops.pair_from_nchw(d, dt)
    d = ops.nhwc_to_nchw(pd)                               # the values the split map carries
    up = ops.resize_bicubic(pd, (48, 40))
    assert isinstance(up, ops.PairMap) and (up.h, up.w) == (48, 40)
    ref = F.interpolate(d, size=(48, 40), mode="bicubic", align_corners=False)
    assert (ops.nhwc_to_nchw(up) - ref).abs().max().item() <= 3e-6

    s = _rand((2, 11, 20, 24), 22, 3.0)
    ps = ops.pair_from_nchw(s, dt)
    s = ops.nhwc_to_nchw(ps)
    x = _rand((2, 3, 64, 80), 23).clamp(-1, 1)
    for xx in (x, None):
        cond = ops.make_m_cond(pd, ps, xx)
        assert isinstance(cond, ops.PairMap) and cond.c == (15 if xx is not None else 12)
        mn = d.reshape(2, -1).min(1)[0].reshape(2, 1, 1, 1)
        t0 = d - mn
        cats = [t0 / t0.reshape(2, -1).max(1)[0].reshape(2, 1, 1, 1), torch.softmax(s, dim=1)]
        if xx is not None:
            # (torch's CPU kernel: its HIP bilinear kernel rounds the source index differently, 6e-6 on these values; the
            # reference's golden runs are CPU runs)
            cats.append(F.interpolate(xx.cpu(), (20, 24), mode="bilinear", align_corners=True).cuda())
        ref = torch.cat(cats, 1)
        got = ops.nhwc_to_nchw(cond)
        assert (got - ref).abs().max().item() <= 1e-6, (got - ref).abs().max().item()
        assert bool((cond.t.view(2, 20, 24, cond.nb, -1)[..., cond.c:] == 0).all())     # pad channels stay zero
    with pytest.raises(RuntimeError, match="both be split maps"):
        ops.make_m_cond(ops.pair_to_nhwc(pd), ps, None)

    y = _rand((2, 19, 12, 16), 24, 2.0)
    py = ops.pair_from_nchw(y, dt)
    y = ops.nhwc_to_nchw(py)
    mean = _rand((2, 24), 25)
    rstd = _rand((2, 24), 26).abs() + 0.5
    out = ops.norm_act_apply(py, mean, rstd, act=ops.ACT_LRELU, slope=0.2)
    ref = F.leaky_relu((y - mean[:, :19, None, None]) * rstd[:, :19, None, None], 0.2)
    assert isinstance(out, ops.PairMap) and (ops.nhwc_to_nchw(out) - ref).abs().max().item() <= 2e-6


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("case", [
    # (n, h, w, cin, cout, k, stride, pad, dil, residual ("same" | "up" | None), act)
    (2, 24, 28, 64, 64, 3, 1, 1, 1, "same", "relu"),        # a ResNet 3x3
    (2, 24, 28, 128, 256, 1, 1, 0, 1, "same", "none"),      # bottleneck expansion with the shortcut in the epilogue
    (2, 24, 28, 64, 32, 3, 1, 2, 2, None, "lrelu"),         # dilated, <= 64 output channels (the 64-cout block tile)
    (3, 48, 40, 64, 96, 3, 2, 1, 1, None, "none"),          # stride 2
    (2, 32, 32, 128, 40, 3, 1, 1, 1, "up", "lrelu"),        # pad channels + the residual read through the folded x2 upsample
    (1, 40, 40, 320, 160, 3, 1, 1, 1, None, "none"),        # a Painter main conv
])
def test_split_conv_on_the_gemm_tiling(dt, case):
    """Round 5: split-precision convs whose storage channel count is a whole number of 32-channel k-steps (and >= 1024 output
    pixels) run on the LDS-tiled GEMM with the split epilogue (conv_gemm_pair_launch) instead of the gather kernel: same
    float64 yardstick as above, and agreement with the gather kernel (dev knob) to the same few 1e-7."""
    from climategan_amd import _lib, ops

    n, h, w, cin, cout, k, stride, pad, dil, res_kind, act = case
    x = _rand((n, cin, h, w), 13)
    wt = _rand((cout, cin, k, k), 14, (2.0 / (cin * k * k)) ** 0.5)
    b = _rand((cout,), 15)
    p = ops.pair_from_nchw(x, dt)
    pw = ops.pack_conv_weight(wt, b, dt, pair=True)
    ref = F.conv2d(x.double(), wt.double(), b.double(), stride=stride, padding=pad, dilation=dil)
    res = None
    if res_kind:
        shp = tuple(ref.shape) if res_kind == "same" else (n, cout, ref.shape[2] // 2, ref.shape[3] // 2)
        res = ops.pair_from_nchw(_rand(shp, 16), dt)
        r64 = ops.nhwc_to_nchw(res).double()
        ref = ref + (r64 if res_kind == "same" else F.interpolate(r64, scale_factor=2, mode="nearest"))
    ref = {"relu": torch.relu, "lrelu": lambda v: F.leaky_relu(v, 0.2), "none": lambda v: v}[act](ref)
    kw = dict(stride=stride, pad=pad, dilation=dil, act={"relu": ops.ACT_RELU, "lrelu": ops.ACT_LRELU, "none": ops.ACT_NONE}[act],
              residual=res, residual_upsample=res_kind == "up")
    y = ops.conv2d(p, pw, **kw)
    assert isinstance(y, ops.PairMap) and y.c == cout
    got = ops.nhwc_to_nchw(y).double()
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    # (K up to 2880 here: the fp32 accumulation's own rounding, ~ sqrt(K) 6e-8 of the terms' scale, is what is left --
    # torch's fp32 conv2d of the same operands is as far from float64)
    f32 = F.conv2d(x, wt, b, stride=stride, padding=pad, dilation=dil).double()
    f64 = F.conv2d(x.double(), wt.double(), b.double(), stride=stride, padding=pad, dilation=dil)
    floor = (f32 - f64).abs().max().item() / f64.abs().max().item()
    assert err <= max(4e-6, 3 * floor), (err, floor)
    assert bool((y.t.view(n, y.h, y.w, y.nb, -1)[..., cout:] == 0).all())          # pad channels of every block stay zero
    dev = _lib.load_dev()
    try:
        dev.cgan_debug_set_conv_kernel(1)                                            # the gather kernel
        yg = ops.nhwc_to_nchw(ops.conv2d(p, pw, **kw)).double()
    finally:
        dev.cgan_debug_set_conv_kernel(0)
        _lib.use_product()
    assert (got - yg).abs().max().item() / ref.abs().max().item() <= max(4e-6, 3 * floor)     # (another summation order)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("case", [
    # (n, h, w, cin, cout, k, pad, dil, residual ("same" | None), act): >= 192 couts, whole 64-channel K stages, >= 16384 pixels
    (4, 80, 80, 256, 256, 3, 2, 2, None, "relu"),           # ResNet layer3's dilated 3x3
    (4, 80, 80, 256, 1024, 1, 0, 1, "same", "relu"),        # bottleneck expansion with the shortcut in the epilogue
    (3, 80, 80, 1024, 256, 1, 0, 1, None, "none"),          # reduce (K = 6 x 1024 in bf16)
    (3, 80, 96, 512, 200, 3, 4, 4, None, "lrelu"),          # pad channels (200 of 208 couts live), a last pixel block in part
])
def test_split_conv_on_the_256_tile(dt, case):
    """Round 6: split-precision convs with >= 192 output channels and whole 64-channel K stages run on the 256 x 256 / eight-wave
    tile (conv_gemm_big_kernel<.., PAIR>: x read from the stored components through the K-block -> storage-block map, split
    epilogue): float64 yardstick as for the other tilings, and agreement with the 128-pixel tiling (dev knob pair_big = 0)."""
    from climategan_amd import _lib, ops

    n, h, w, cin, cout, k, pad, dil, res_kind, act = case
    x = _rand((n, cin, h, w), 33)
    wt = _rand((cout, cin, k, k), 34, (2.0 / (cin * k * k)) ** 0.5)
    b = _rand((cout,), 35)
    p = ops.pair_from_nchw(x, dt)
    pw = ops.pack_conv_weight(wt, b, dt, pair=True)
    f64 = F.conv2d(x.double(), wt.double(), b.double(), padding=pad, dilation=dil)
    ref = f64
    res = None
    if res_kind:
        res = ops.pair_from_nchw(_rand(tuple(ref.shape), 36), dt)
        ref = ref + ops.nhwc_to_nchw(res).double()
    ref = {"relu": torch.relu, "lrelu": lambda v: F.leaky_relu(v, 0.2), "none": lambda v: v}[act](ref)
    kw = dict(pad=pad, dilation=dil, act={"relu": ops.ACT_RELU, "lrelu": ops.ACT_LRELU, "none": ops.ACT_NONE}[act], residual=res)
    y = ops.conv2d(p, pw, **kw)
    assert isinstance(y, ops.PairMap) and y.c == cout and y.t.shape[3] == ops.store_blocks(dt) * ops.cs8(cout)
    got = ops.nhwc_to_nchw(y).double()
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    f32 = F.conv2d(x, wt, b, padding=pad, dilation=dil).double()
    floor = (f32 - f64).abs().max().item() / f64.abs().max().item()
    assert err <= max(4e-6, 3 * floor), (err, floor)
    assert bool((y.t.view(n, y.h, y.w, y.nb, -1)[..., cout:] == 0).all())          # pad channels of every block stay zero
    dev = _lib.load_dev()
    try:
        dev.cgan_debug_set_pair_big(0)                                               # the 128-pixel tiling of round 5
        pw_dev = ops.pack_conv_weight(wt, b, dt, pair=True)
        ye = ops.nhwc_to_nchw(ops.conv2d(p, pw_dev, **kw)).double()
        dev.cgan_debug_set_pair_big(1)
        yb = ops.nhwc_to_nchw(ops.conv2d(p, pw_dev, **kw)).double()
    finally:
        dev.cgan_debug_set_pair_big(1)
        _lib.use_product()
    assert torch.equal(yb, got)                                                      # the development build runs the same kernel
    assert (got - ye).abs().max().item() / ref.abs().max().item() <= max(4e-6, 3 * floor)     # (another summation order)
