"""Maps of 2 GiB and more: one C-ABI call covers less than 2 GiB per tensor (32-bit offsets in the kernels); the host layer
runs the sample-independent ops on batch slices (``ops._batch_chunked``).  Checked two ways: with the limit lowered, so that
ordinary tensors take the sliced path and must equal the single-call result bit for bit (every sliced op), and with a real
map above 2 GiB (configs[3]'s global batch of 32 per domain on one GPU has several)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture
def small_limit(monkeypatch):
    from climategan_amd import ops

    def set_limit(nbytes):
        monkeypatch.setattr(ops, "MAX_MAP_BYTES", nbytes)
    return set_limit


def _rand(shape, dt, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.randn(shape, device="cuda", generator=g).to(dt)


def _same(a, b):
    from climategan_amd import ops
    if a is None:
        assert b is None
    elif isinstance(a, ops.NHWC):
        assert a.c == b.c and torch.equal(a.t, b.t)
    elif isinstance(a, tuple):
        assert len(a) == len(b)
        for u, v in zip(a, b):
            _same(u, v)
    else:
        assert torch.equal(a, b)


def test_sliced_ops_equal_the_single_call(small_limit):
    from climategan_amd import ops

    dt = torch.bfloat16
    n, h, w = 6, 48, 40
    x = ops.NHWC(_rand((n, h, w, 64), dt, 1), 64)
    x2 = ops.NHWC(_rand((n, h, w, 64), dt, 2), 64)
    wt = torch.randn(128, 64, 3, 3, device="cuda") * 0.05
    b = torch.randn(128, device="cuda")
    pw = ops.pack_conv_weight(wt, b, dt)
    res = ops.NHWC(_rand((n, h, w, 128), dt, 3), 128)
    xf = torch.randn(n, 3, h, w, device="cuda")
    mf = (torch.rand(n, 1, h, w, device="cuda") > 0.5).float()
    wsp = [torch.randn(128, 3, 3, 3, device="cuda") * 0.1, torch.randn(128, device="cuda") * 0.1,
           torch.randn(64, 128, 3, 3, device="cuda") * 0.03, torch.randn(64, device="cuda") * 0.1,
           torch.randn(64, 128, 3, 3, device="cuda") * 0.03, torch.randn(64, device="cuda") * 0.1]
    pk = ops.pack_spade_weights(*wsp, dt)
    cond = ops.nchw_to_nhwc(xf, dt, cs=4)

    def run():
        out = {}
        out["conv"] = ops.conv2d(x, pw, pad=1, act=ops.ACT_LRELU, residual=res)
        out["conv_ups"] = ops.conv2d(ops.NHWC(x.t[:, :h // 2, :w // 2].contiguous(), 64), pw, pad=1, in_upsample=True)
        dy = out["conv"]
        out["dgrad"] = ops.conv2d_bwd_data(dy, wt, (n, h, w), pad=1)
        out["dgrad_add"] = ops.conv2d_bwd_data(dy, wt, (n, h, w), pad=1, add=x2)
        out["dgrad_reflect"] = ops.conv2d_bwd_data(ops.NHWC(dy.t, 128), wt, (n, h, w), pad=1, pad_mode=ops.PAD_REFLECT)
        out["wgrad"] = ops.conv2d_bwd_weight(x, dy, (128, 64, 3, 3), pad=1, use_workspace=True)
        mean, rstd = ops.instnorm_stats(x)
        out["stats"] = (mean, rstd)
        out["apply"] = ops.norm_act_apply(x, mean, rstd, act=ops.ACT_LRELU, residual=x2)
        out["act_bwd"] = ops.act_bwd(out["apply"], x2, ops.ACT_LRELU)
        out["in_bwd"] = ops.instnorm_act_bwd(out["apply"], x2, rstd, act=ops.ACT_LRELU)
        out["spade"] = ops.spade_fused(x, mean, rstd, cond, pk, act=ops.ACT_LRELU)
        out["sumpool"] = ops.sumpool2x2(x)
        out["near"] = ops.resize_nearest(x, (h * 2, w * 2))
        out["near_bwd"] = ops.resize_nearest_bwd(out["near"], (h, w), x.cs)
        out["avg"] = ops.avgpool3x3s2(x)
        out["avg_bwd"] = ops.avgpool3x3s2_bwd(out["avg"], (h, w))
        out["max2"] = ops.maxpool2x2(x)
        out["max2_bwd"] = ops.maxpool2x2_bwd(x, out["max2"])
        out["mul"] = ops.eltwise_mul(x, x2)
        out["to_nhwc"] = ops.nchw_to_nhwc(xf, dt, mask=mf)
        out["to_nchw"] = ops.nhwc_to_nchw(ops.NHWC(out["to_nhwc"].t, 3), paste_x=xf, paste_m=mf)
        fake = ops.NHWC(out["to_nhwc"].t, 3)
        out["heads"] = ops.painter_heads(fake, xf, mf, dt, want_d=True, want_vgg=True)
        out["heads_bwd"] = ops.painter_heads_bwd(out["heads"][0], out["heads"][1], mf)
        acc = torch.zeros(1, device="cuda")
        out["l1"] = ops.l1_loss(x, x2, 0.5, acc)
        out["bce"] = ops.bce_logits(x, 1.0, 0.25, acc)
        out["acc"] = acc
        return out

    ref = run()
    small_limit(1_000_000)      # four samples per slice for the 64-channel maps, two for the 128-channel ones, one upsampled
    got = run()
    for k in ref:
        if k == "acc":                             # sums over slices: another order of fp32 atomics
            assert abs(ref[k].item() - got[k].item()) <= 1e-4 * abs(ref[k].item())
        elif k == "wgrad":                         # the slices accumulate into dw: another summation order
            for a, b_ in zip(ref[k], got[k]):
                assert (a - b_).abs().max().item() <= 2e-4 * a.abs().max().item()
        else:
            try:
                _same(ref[k], got[k])
            except AssertionError:
                raise AssertionError("sliced result of '%s' differs from the single call" % k) from None


def test_an_unsplit_op_refuses_a_map_above_the_limit(small_limit):
    from climategan_amd import ops

    x = ops.NHWC(_rand((4, 32, 32, 64), torch.bfloat16, 4), 64)
    small_limit(x.t.nbytes // 2)
    import climategan_amd.ops as ops_mod
    ops_mod.ABI_MAX_BYTES, keep = x.t.nbytes // 2, ops_mod.ABI_MAX_BYTES
    try:
        with pytest.raises(RuntimeError, match="does not split the batch"):
            ops.sigmoid(x)
    finally:
        ops_mod.ABI_MAX_BYTES = keep
    return
    with pytest.raises(RuntimeError, match="does not split the batch"):
        ops.sigmoid(x)


def test_a_real_map_above_2_gib_equals_its_slices():
    """The SPADE shared conv's re-materialised hidden map at configs[3]'s global batch: 32 x 640 x 640 x 128 bf16 = 3.36 GB
    out of a 105 MB conditioning image; and the weight gradient that reads it."""
    from climategan_amd import ops

    dt = torch.bfloat16
    n = 32
    cond = ops.NHWC(_rand((n, 640, 640, 8), dt, 5), 3)
    cond.t[..., 3:] = 0
    wt = torch.randn(128, 3, 3, 3, device="cuda") * 0.2
    pw = ops.pack_conv_weight(wt, torch.randn(128, device="cuda") * 0.1, dt)
    y = ops.conv2d(cond, pw, pad=1, act=ops.ACT_RELU)
    assert y.t.nbytes > 2 ** 31 and tuple(y.t.shape) == (n, 640, 640, 128)
    for lo in (0, 15, 20, 31):
        one = ops.conv2d(ops.NHWC(cond.t[lo:lo + 1], 3), pw, pad=1, act=ops.ACT_RELU)
        assert torch.equal(one.t[0], y.t[lo])
    dw, db = ops.conv2d_bwd_weight(cond, y, (128, 3, 3, 3), pad=1)
    dw_ref = torch.zeros_like(dw)
    db_ref = torch.zeros_like(db)
    for lo in range(0, n, 8):
        ops.conv2d_bwd_weight(ops.NHWC(cond.t[lo:lo + 8], 3), ops.NHWC(y.t[lo:lo + 8], 128), (128, 3, 3, 3), pad=1,
                              dw=dw_ref, dbias=db_ref)
    assert (dw - dw_ref).abs().max().item() <= 2e-4 * dw_ref.abs().max().item()
    assert (db - db_ref).abs().max().item() <= 2e-4 * db_ref.abs().max().item()
    dx = ops.act_bwd(y, y, ops.ACT_RELU)
    assert dx.t.nbytes > 2 ** 31 and torch.equal(dx.t[17], ops.act_bwd(ops.NHWC(y.t[17:18], 128), ops.NHWC(y.t[17:18], 128),
                                                                          ops.ACT_RELU).t[0])
