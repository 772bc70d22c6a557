"""The im2col-free upper bound of the k x k wide layers (review of round 4, item 4): the SAME M x N x K as a plain bf16 GEMM
through hipBLASLt (torch.matmul on an [npix, k*k*cin] x [k*k*cin, cout] problem: what the layer would cost if its
im2col matrix existed for free in HBM), cold operands (a rotation of buffers larger than the Infinity Cache), next to this
package's implicit-GEMM kernels on the real layer.
usage (GPU box): python tools/micro/gemm_yardstick_kxk.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from climategan_amd import ops  # noqa: E402

# name, n, h, cin, cout, k, dilation
SHAPES = [
    ("layer3 3x3 d2 256->256 @8x80^2", 8, 80, 256, 256, 3, 2),
    ("layer4 3x3 d4 512->512 @8x80^2", 8, 80, 512, 512, 3, 4),
    ("aspp 3x3 d6 2048->256 @8x80^2", 8, 80, 2048, 256, 3, 6),
    ("decoder 3x3 512->512 @4x80^2", 4, 80, 512, 512, 3, 1),
    ("vgg 3x3 256->256 @4x160^2", 4, 160, 256, 256, 3, 1),
    ("D 4x4 s1 512->512 @8x40^2", 8, 40, 512, 512, 4, 1),
]


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
    print("%-36s %10s %12s %12s %10s %10s" % ("layer", "GFLOP", "hipBLASLt us", "this pkg us", "TF/s blas", "TF/s pkg"))
    for name, n, h, cin, cout, k, dil in SHAPES:
        pad = dil * (k // 2) if k == 3 else 1
        ho = h + 2 * pad - dil * (k - 1)
        npix, K = n * ho * ho, k * k * cin
        nbuf = max(2, int(600e6 // (npix * K * 2)) + 1)
        nbuf = min(nbuf, 4)
        cols = [torch.randn(npix, K, device="cuda").to(dt) for _ in range(nbuf)]           # the im2col matrix, for free
        wm = (torch.randn(K, cout, device="cuda") * 0.02).to(dt)
        out = torch.empty(npix, cout, device="cuda", dtype=dt)
        t_mm = timed(lambda i: torch.matmul(cols[i % nbuf], wm, out=out), 3 * nbuf)
        del cols
        nb2 = max(2, int(600e6 // (n * h * h * cin * 2)) + 1)
        xs = [torch.randn(n, h, h, cin, device="cuda").to(dt) for _ in range(nb2)]
        w = torch.randn(cout, cin, k, k, device="cuda") * 0.02
        pw = ops.pack_conv_weight(w, None, dt)
        t_pk = timed(lambda i: ops.conv2d(ops.NHWC(xs[i % nb2], cin), pw, stride=1, pad=pad, dilation=dil), 3 * nb2)
        gf = 2.0 * npix * K * cout / 1e9
        print("%-36s %10.1f %12.1f %12.1f %10.0f %10.0f" % (name, gf, t_mm, t_pk, gf / (t_mm * 1e-6) / 1e3, gf / (t_pk * 1e-6) / 1e3), flush=True)
        del xs


if __name__ == "__main__":
    main()
