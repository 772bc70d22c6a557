"""Time single convolution shape classes of the Masker through the C ABI.
usage (GPU box): python tools/bench_conv.py [--force N]   (N: cgan_debug_set_conv_kernel value)"""
import argparse
import ctypes
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from climategan_amd import _lib, ops  # noqa: E402
import os  # noqa: E402

if os.environ.get("CGAN_LIB"):      # A/B against another build of the library (same box, same call)
    _lib.LIB_PATH = Path(os.environ["CGAN_LIB"]).resolve()

SHAPES = [
    # name, cin, cout, k, stride, pad, dil, H (input), count-in-resnet
    ("l3 1x1 1024->256", 1024, 256, 1, 1, 0, 1, 80),
    ("l3 3x3 d2 256->256", 256, 256, 3, 1, 2, 2, 80),
    ("l3 1x1 256->1024", 256, 1024, 1, 1, 0, 1, 80),
    ("l4 3x3 d4 512->512", 512, 512, 3, 1, 4, 4, 80),
    ("l4 1x1 512->2048", 512, 2048, 1, 1, 0, 1, 80),
    ("l4 1x1 2048->512", 2048, 512, 1, 1, 0, 1, 80),
    ("aspp 3x3 d6 2048->256", 2048, 256, 3, 1, 6, 6, 80),
    ("l1 1x1 64->256", 64, 256, 1, 1, 0, 1, 160),
    ("l1 3x3 64->64", 64, 64, 3, 1, 1, 1, 160),
    ("l2 3x3 128->128", 128, 128, 3, 1, 1, 1, 80),
    ("painter 3x3 160->160 @80", 160, 160, 3, 1, 1, 1, 80),
    ("painter 3x3 320->320 @40", 320, 320, 3, 1, 1, 1, 40),
    ("seg 3x3 256->256 @82", 256, 256, 3, 1, 1, 1, 82),
    ("vgg 3x3 64->64 @640", 64, 64, 3, 1, 1, 1, 640),
    ("vgg 3x3 64->128 @320", 64, 128, 3, 1, 1, 1, 320),
    ("vgg 3x3 128->128 @320", 128, 128, 3, 1, 1, 1, 320),
    ("vgg 3x3 128->256 @160", 128, 256, 3, 1, 1, 1, 160),
    ("spade shared 3x3 3->128 @640", 3, 128, 3, 1, 1, 1, 640),
    ("spade shared 3x3 3->128 @320", 3, 128, 3, 1, 1, 1, 320),
    ("spade dgrad 3x3 40->128 @640", 40, 128, 3, 1, 1, 1, 640),
    ("spade dgrad 3x3 80->128 @640", 80, 128, 3, 1, 1, 1, 640),
    ("spade gb 3x3 128->80 @640", 128, 80, 3, 1, 1, 1, 640),
    ("spade gb 3x3 128->160 @320", 128, 160, 3, 1, 1, 1, 320),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", type=int, default=0)
    ap.add_argument("--bs", type=int, default=16)
    ap.add_argument("--abl", type=int, default=0)
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--only", default="", help="substring filter on the shape name")
    ap.add_argument("--cfg", type=int, default=0, help="cgan_debug_set_gemm_cfg: 1 64x256, 2 256x128, 3 128x256, 4 128x128")
    ap.add_argument("--ws", type=int, default=0, help="cgan_debug_set_gemm_ws: 1 never the K = 64 kernel, 5 force it (256 x 128), 6 (128 x 256)")
    ap.add_argument("--check", action="store_true", help="compare with the plain kernel")
    ap.add_argument("--hw", type=int, default=0, help="override the input extent of every shape")
    ap.add_argument("--res", action="store_true", help="add a residual input (bottleneck expand)")
    ap.add_argument("--nct", type=int, default=0, help="cgan_debug_set_conv3x3_nct: channel tiles per workgroup of the 3x3 LDS kernel")
    ap.add_argument("--bigw", type=int, default=8, help="cgan_debug_set_big_waves: 16 = the sixteen-wave form of the 256 x 256 kernel")
    ap.add_argument("--c4", type=int, default=1, help="cgan_debug_set_conv3x3_c4: 0 no folded-tap kernel, 1 default, 2 all 128 couts per workgroup")
    args = ap.parse_args()
    dt = torch.float16 if args.dtype == "fp16" else torch.bfloat16
    lib = _lib.load_dev()
    lib.cgan_debug_set_conv_kernel(ctypes.c_int(args.force))
    lib.cgan_debug_set_gemm_cfg(ctypes.c_int(args.cfg))
    lib.cgan_debug_set_gemm_ws(ctypes.c_int(args.ws))
    lib.cgan_debug_set_conv3x3_c4(ctypes.c_int(args.c4))
    lib.cgan_debug_set_conv3x3_nct(ctypes.c_int(args.nct))
    if hasattr(lib, "cgan_debug_set_big_waves"):
        lib.cgan_debug_set_big_waves(ctypes.c_int(args.bigw))
    for name, cin, cout, k, stride, pad, dil, H in SHAPES:
        if args.only not in name:
            continue
        H = args.hw or H
        x = ops.NHWC(torch.randn(args.bs, H, H, ops.cs8(cin), device="cuda").to(dt), cin)   # storage channels: round_up(cin, 8)
        w = torch.randn(cout, cin, k, k, device="cuda") * 0.02
        pw = ops.pack_conv_weight(w, None, dt)
        res = None
        if args.res:
            Ho = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1
            res = ops.NHWC(torch.randn(args.bs, Ho, Ho, ops.cs8(cout), device="cuda").to(dt), cout)
        for _ in range(3):
            y = ops.conv2d(x, pw, stride=stride, pad=pad, dilation=dil, act=ops.ACT_NONE if args.res else ops.ACT_RELU, residual=res)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 10
        for _ in range(n):
            y = ops.conv2d(x, pw, stride=stride, pad=pad, dilation=dil, act=ops.ACT_NONE if args.res else ops.ACT_RELU, residual=res)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        flops = 2.0 * y.n * y.h * y.w * cout * cin * k * k
        chk = ""
        if args.check:
            lib.cgan_debug_set_gemm_ws(ctypes.c_int(1))
            y0 = ops.conv2d(x, pw, stride=stride, pad=pad, dilation=dil, act=ops.ACT_RELU)
            lib.cgan_debug_set_gemm_ws(ctypes.c_int(args.ws))
            torch.cuda.synchronize()
            chk = "  identical to the plain kernel: %s (max |diff| %.3g)" % (torch.equal(y.t, y0.t), (y.t.float() - y0.t.float()).abs().max().item())
        print("%-24s %8.3f ms  %7.1f TFLOP/s%s" % (name, ms, flops / ms / 1e9, chk))


if __name__ == "__main__":
    main()
