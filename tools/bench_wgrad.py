"""NOTE: shapes that take < ~0.1 ms are host-bound in this loop (Python + ctypes per call): read kernel durations with
rocprofv3 (tools/prof_wgrad.sh) for those.

Time the weight-gradient kernel on the Masker / Painter layer shapes through the C ABI.
usage (GPU box): python tools/bench_wgrad.py [--target WGS] [--dbg BITS] [--bs N]"""
import argparse
import ctypes
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from climategan_amd import _lib, ops  # noqa: E402

SHAPES = [
    # name, cin, cout, k, stride, pad, dil, H (input)
    ("l3 1x1 1024->256 @40", 1024, 256, 1, 1, 0, 1, 40),
    ("l3 3x3 d2 256->256 @40", 256, 256, 3, 1, 2, 2, 40),
    ("l3 1x1 256->1024 @40", 256, 1024, 1, 1, 0, 1, 40),
    ("l4 3x3 d4 512->512 @40", 512, 512, 3, 1, 4, 4, 40),
    ("l4 1x1 512->2048 @40", 512, 2048, 1, 1, 0, 1, 40),
    ("aspp 3x3 d6 2048->256 @40", 2048, 256, 3, 1, 6, 6, 40),
    ("l1 1x1 64->256 @160", 64, 256, 1, 1, 0, 1, 160),
    ("l1 3x3 64->64 @160", 64, 64, 3, 1, 1, 1, 160),
    ("l2 3x3 128->128 @80", 128, 128, 3, 1, 1, 1, 80),
    ("painter 3x3 160->160 @80", 160, 160, 3, 1, 1, 1, 80),
    ("painter 3x3 40->40 @640", 40, 40, 3, 1, 1, 1, 640),
    ("vgg 3x3 64->64 @640", 64, 64, 3, 1, 1, 1, 640),
    ("D 4x4s2 64->128 @320", 64, 128, 4, 2, 2, 1, 320),
    ("spade sh 3->128 @640", 3, 128, 3, 1, 1, 1, 640),
    ("spade gb 128->40 @640", 128, 40, 3, 1, 1, 1, 640),
    ("spade gb 128->80 @640", 128, 80, 3, 1, 1, 1, 640),
    ("spade gb 128->160 @320", 128, 160, 3, 1, 1, 1, 320),
    ("vgg 3x3 128->128 @320", 128, 128, 3, 1, 1, 1, 320),
    ("vgg 3x3 256->256 @160", 256, 256, 3, 1, 1, 1, 160),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--target", type=int, default=0)
    ap.add_argument("--dbg", type=int, default=0)
    ap.add_argument("--bs", type=int, default=12)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--atomic", action="store_true")
    ap.add_argument("--coop-min", type=int, default=-1, help="cgan_debug_set_wgrad_coop_min_pixels")
    ap.add_argument("--only", default="", help="substring filter on the shape name")
    ap.add_argument("--tile", type=int, default=-1, help="cgan_debug_set_wgrad_tile3x3 (0 never, 1 default, 2 wherever it applies)")
    ap.add_argument("--bias", action="store_true", help="with the bias gradient (rides in the weight-gradient kernels)")
    args = ap.parse_args()
    dt = torch.float16 if args.dtype == "fp16" else torch.bfloat16
    lib = _lib.load_dev()
    lib.cgan_debug_set_wgrad(ctypes.c_int(args.target), ctypes.c_int(args.dbg))
    if args.tile >= 0:
        lib.cgan_debug_set_wgrad_tile3x3(ctypes.c_int(args.tile))
    if args.coop_min >= 0:
        lib.cgan_debug_set_wgrad_coop_min_pixels(ctypes.c_int(args.coop_min))
    print("target %d dbg %d bs %d" % (args.target, args.dbg, args.bs))
    for name, cin, cout, k, stride, pad, dil, H in SHAPES:
        if args.only not in name:
            continue
        bs = args.bs if H < 320 else max(args.bs // 3, 1)
        x = ops.NHWC(torch.randn(bs, H, H, ops.cs8(cin), device="cuda").to(dt), cin)       # storage channels: round_up(cin, 8)
        Ho = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1
        dy = ops.NHWC(torch.randn(bs, Ho, Ho, ops.cs8(cout), device="cuda").to(dt), cout)
        dw = torch.zeros(cout, cin, k, k, device="cuda")
        db = torch.zeros(cout, device="cuda")
        for _ in range(2):
            ops.conv2d_bwd_weight(x, dy, (cout, cin, k, k), stride, pad, dil, want_bias=args.bias, dw=dw, dbias=db if args.bias else None, use_workspace=not args.atomic)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 10
        for _ in range(n):
            ops.conv2d_bwd_weight(x, dy, (cout, cin, k, k), stride, pad, dil, want_bias=args.bias, dw=dw, dbias=db if args.bias else None, use_workspace=not args.atomic)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        flops = 2.0 * bs * Ho * Ho * cout * cin * k * k
        print("%-28s %8.3f ms  %7.1f TFLOP/s" % (name, ms, flops / ms / 1e9))


if __name__ == "__main__":
    main()
