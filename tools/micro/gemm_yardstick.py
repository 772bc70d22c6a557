"""Yardstick for the layer3 / layer4 1 x 1 layers (n8 80x80 bf16): this package's conv2d against torch.matmul (hipBLASLt)
on the same GEMM, each with WARM operands (one buffer, resident in the Infinity Cache) and COLD operands (a rotation of
buffers larger than the 256 MB cache, the state the train step presents).  Durations from HIP events around back-to-back
launches (>= 30 us each: the queue stays full).
usage (GPU box): python tools/micro/gemm_yardstick.py [--soak]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from climategan_amd import ops  # noqa: E402

SHAPES = [(1024, 256), (256, 1024), (2048, 512), (512, 2048), (512, 128), (128, 512)]
N, H = 8, 80


SOAK = "--soak" in sys.argv      # 1.5 s of dense 8192^3 GEMMs before every timing: the clocks a busy train step sees
_soak = None


def timed(fn, reps):
    global _soak
    if SOAK:
        if _soak is None:
            _soak = (torch.randn(8192, 8192, device="cuda").bfloat16(), torch.randn(8192, 8192, device="cuda").bfloat16())
        torch.cuda.synchronize()
        import time
        t0 = time.time()
        while time.time() - t0 < 1.5:
            for _ in range(20):
                torch.matmul(_soak[0], _soak[1])
            torch.cuda.synchronize()
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
    for cin, cout in SHAPES:
        npix = N * H * H
        nbuf = max(2, int(600e6 // (npix * max(cin, cout) * 2)) + 1)
        xs = [torch.randn(N, H, H, cin, device="cuda").to(dt) for _ in range(nbuf)]
        w = torch.randn(cout, cin, 1, 1, device="cuda") * 0.02
        pw = ops.pack_conv_weight(w, None, dt)
        wm = w.view(cout, cin).t().contiguous().to(dt)
        outs = [torch.empty(npix, cout, device="cuda", dtype=dt) for _ in range(nbuf)]

        def conv(i, cold):
            ops.conv2d(ops.NHWC(xs[i % nbuf if cold else 0], cin), pw, stride=1, pad=0, dilation=1, act=ops.ACT_NONE)

        def mm(i, cold):
            j = i % nbuf if cold else 0
            torch.matmul(xs[j].view(npix, cin), wm, out=outs[j])

        reps = 4 * nbuf
        r = {}
        for name, f in (("conv", conv), ("matmul", mm)):
            for cold in (False, True):
                r[name, cold] = timed(lambda i: f(i, cold), reps)
        mb = npix * (cin + cout) * 2 / 1e6
        print("1x1 %4d -> %4d  %6.1f MB (%.1f us at 8 TB/s)  conv warm %6.1f cold %6.1f us | hipBLASLt warm %6.1f cold %6.1f us"
              % (cin, cout, mb, mb / 8e6 * 1e6, r["conv", False], r["conv", True], r["matmul", False], r["matmul", True]), flush=True)


if __name__ == "__main__":
    main()
