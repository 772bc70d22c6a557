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
    # accumulation semantics: a second call adds
    dw2, _ = ops.conv2d_bwd_weight(to_nhwc(x.detach(), dt), dyg, tuple(w.shape), stride=stride, pad=pad, dilation=dil,
                                   want_bias=False, dw=dw.clone())
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


def test_backward_refuses_reflect_and_upsample():
    import ctypes as C
    from climategan_amd import _lib, ops
    lib = _lib.load()
    d = ops._conv_desc(_lib.CGAN_F16, 1, 8, 8, 8, 8, 3, 3, 1, 1, 1, _lib.PAD_REFLECT)
    assert lib.cgan_conv2d_dgrad_packed_weight_bytes(C.byref(d)) == 0
    assert b"zero padding" in lib.cgan_last_error()
