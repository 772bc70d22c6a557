"""Follow-up to gemm_yardstick.py, cold operands only (a rotation of buffers larger than the Infinity Cache):
(a) every wide-layer GEMM kernel variant of the development build on the layer3 / layer4 1 x 1 shapes (cgan_debug_set_gemm_ws /
    _cfg), against hipBLASLt's time for the same GEMM;
(b) the 1 x 1 weight gradients of the same layers (and their data gradients' twins) against torch.matmul(dy^T, x).
usage (GPU box): python tools/micro/gemm_yardstick2.py"""
import ctypes
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from climategan_amd import _lib, ops  # noqa: E402

N, H = 8, 80
SHAPES = [(2048, 512), (512, 2048), (1024, 256), (256, 1024), (1024, 2048), (1024, 512)]
KNOBS = [("auto", 0, 0), ("plain 256x128", 0, 2), ("plain 128x256", 0, 3), ("plain 128x128", 0, 4), ("k64 256x128", 5, 0),
         ("k64 128x256", 6, 0), ("big 256x256", 8, 0), ("xres", 11, 0)]


def timed(fn, reps):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    dt = torch.bfloat16
    lib = _lib.load_dev()
    npix = N * H * H
    for cin, cout in SHAPES:
        nbuf = max(2, int(600e6 // (npix * max(cin, cout) * 2)) + 1)
        xs = [torch.randn(N, H, H, cin, device="cuda").to(dt) for _ in range(nbuf)]
        dys = [torch.randn(N, H, H, cout, device="cuda").to(dt) for _ in range(nbuf)]
        w = torch.randn(cout, cin, 1, 1, device="cuda") * 0.02
        pw = ops.pack_conv_weight(w, None, dt)
        wm = w.view(cout, cin).t().contiguous().to(dt)
        out = torch.empty(npix, cout, device="cuda", dtype=dt)
        reps = 4 * nbuf
        t_mm = timed(lambda i: torch.matmul(xs[i % nbuf].view(npix, cin), wm, out=out), reps)
        line = ["1x1 %4d -> %4d fwd: hipBLASLt %6.1f us |" % (cin, cout, t_mm)]
        for name, ws, cfg in KNOBS:
            lib.cgan_debug_set_gemm_ws(ctypes.c_int(ws))
            lib.cgan_debug_set_gemm_cfg(ctypes.c_int(cfg))
            try:
                t = timed(lambda i: ops.conv2d(ops.NHWC(xs[i % nbuf], cin), pw, stride=1, pad=0, dilation=1, act=ops.ACT_NONE), reps)
                line.append("%s %6.1f" % (name, t))
            except Exception as e:                                    # a forced kernel that does not take the shape
                line.append("%s n/a (%s)" % (name, str(e)[:30]))
        lib.cgan_debug_set_gemm_ws(ctypes.c_int(0))
        lib.cgan_debug_set_gemm_cfg(ctypes.c_int(0))
        print(" ".join(line), flush=True)
        # weight gradient: dW [cout][cin] = dy^T [cout][npix] x [npix][cin]
        dwo = torch.empty(cout, cin, device="cuda", dtype=torch.float32)
        t_mm = timed(lambda i: torch.matmul(dys[i % nbuf].view(npix, cout).t(), xs[i % nbuf].view(npix, cin)), reps)
        t_w = timed(lambda i: ops.conv2d_bwd_weight(ops.NHWC(xs[i % nbuf], cin), ops.NHWC(dys[i % nbuf], cout), (cout, cin, 1, 1),
                                                    want_bias=False, dw=dwo.view(cout, cin, 1, 1)), reps)
        print("1x1 %4d -> %4d wgrad: hipBLASLt (bf16 out) %6.1f us | this package (fp32 out, accumulate) %6.1f us"
              % (cin, cout, t_mm, t_w), flush=True)


if __name__ == "__main__":
    main()
