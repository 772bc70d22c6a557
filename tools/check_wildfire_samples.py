"""GPU box: the wildfire event on a batch of repeats next to kernels of a second stream -- per stage of the kernel chain, how
many values of a repeat differ from its original (must be 0), for the 8-outputs-per-thread blur (0) and the reference blur
(1); and sentinels that show whether the side-stream kernels write outside their own tensors.  This is the harness that
found the round-3 packed-fp32 problem (R5 DESIGN 4.6)."""
import ctypes as C
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from climategan_amd import _lib, ops  # noqa: E402

lib = _lib.load_dev()
dt = torch.float16
g = torch.Generator(device="cuda")
g.manual_seed(1)
x2 = torch.rand(2, 3, 640, 640, device="cuda", generator=g) * 2 - 1
s2 = torch.randn(2, 11, 160, 160, device="cuda", generator=g)
s2[:, 9, :60] += 3.0
x, seg = x2.repeat(8, 1, 1, 1), ops.nchw_to_nhwc(s2.repeat(8, 1, 1, 1), dt)
side = torch.cuda.Stream()
xg = ops.NHWC(torch.randn((16, 80, 80, 256), device="cuda", generator=g).to(dt), 256)
pwg = ops.pack_conv_weight(torch.randn(256, 256, 3, 3, device="cuda", generator=g) * 0.05, None, dt)
xc = ops.NHWC(torch.randn((16, 320, 320, 80), device="cuda", generator=g).to(dt), 80)
pwc = ops.pack_conv_weight(torch.randn(80, 80, 3, 3, device="cuda", generator=g) * 0.05, torch.randn(80, device="cuda", generator=g), dt)
xm = ops.NHWC(torch.randn((16, 320, 320, 24), device="cuda", generator=g).to(dt), 20)
pwm = ops.pack_conv_weight(torch.randn(20, 20, 1, 1, device="cuda", generator=g) * 0.05, None, dt)
a = torch.randn(4096, 4096, device="cuda", dtype=dt)


def side_work(kind):
    with torch.no_grad():
        for _ in range(8):
            if kind == "conv_gemm":
                ops.conv2d(xg, pwg, pad=1)
            elif kind == "conv3x3_lds":
                ops.conv2d(xc, pwc, pad=1, act=ops.ACT_LRELU)
            elif kind == "conv_mfma":
                ops.conv2d(xm, pwm)
            elif kind == "torch matmul":
                a @ a


def same(u):
    return sum(int((u[i] != u[i % 2]).sum()) for i in range(2, 16))


def wildfire_stages(kind):
    n, _, h, w = x.shape
    ks = 301
    nbytes = lib.cgan_wildfire_workspace_bytes(n, h, w, seg.h, seg.w, ks)
    ws = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    out = torch.empty_like(x)
    if kind != "none":
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            side_work(kind)
    _lib.check(lib.cgan_wildfire_nchw(ops._ptr(x), ops._ptr(seg.t), seg.dtype_id, ops._ptr(out), n, h, w, seg.h, seg.w, seg.c,
                                      9, ks, 150.5, 200.0, 1, 120.0, ops._ptr(ws), nbytes, ops._stream()), "wildfire")
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    up = lambda b: (b + 255) // 256 * 256       # noqa: E731  (the workspace layout of cgan_wildfire_nchw)
    off = up(n * 3 * 4) + up(ks * 4)
    img = ws[off:off + n * 3 * h * w].view(n, 3, h, w)
    off += up(n * 3 * h * w) + up(n * seg.h * seg.w)
    dil = ws[off:off + n * h * w].view(n, h, w)
    off += up(n * h * w)
    m0 = ws[off:off + n * h * w * 4].view(torch.float32).view(n, h, w)
    off += up(n * h * w * 4)
    m1 = ws[off:off + n * h * w * 4].view(torch.float32).view(n, h, w)
    return {"out": out, "warm + contrast": img, "dilate x": dil, "blur x": m1, "blur y": m0}


sent = [torch.full((96 << 20,), 0x5A, dtype=torch.uint8, device="cuda") for _ in range(4)]
torch.cuda.synchronize()
for kind in ("conv_gemm", "conv3x3_lds"):
    for rep in range(3):
        with torch.cuda.stream(side):
            side_work(kind)
        torch.cuda.synchronize()
    print("sentinels after %-12s: %s bytes changed" % (kind, [int((t_ != 0x5A).sum()) for t_ in sent]), flush=True)
del sent
for blur in (0, 1):
    lib.cgan_debug_set_wf_blur(C.c_int(blur))
    for kind in ("none", "conv_gemm", "conv3x3_lds", "conv_mfma", "torch matmul"):
        print("blur kernel %d next to %-13s" % (blur, kind), {k: same(v) for k, v in wildfire_stages(kind).items()}, flush=True)
lib.cgan_debug_set_wf_blur(C.c_int(0))
