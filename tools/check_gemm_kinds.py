"""GPU-box check of the wide-layer GEMM kernels against the plain kernel per development knob (cgan_debug_set_gemm_ws):
max deviation, NaNs, run-to-run determinism.  usage: python tools/check_gemm_kinds.py"""
import ctypes
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from climategan_amd import _lib, ops  # noqa: E402

CASES = [  # cin, cout, B, H, W, bias, act
    (256, 256, 2, 64, 64, False, ops.ACT_RELU), (256, 256, 2, 64, 64, True, ops.ACT_NONE), (256, 256, 2, 64, 64, False, ops.ACT_LRELU),
    (256, 256, 2, 64, 64, True, ops.ACT_RELU), (256, 250, 2, 64, 64, False, ops.ACT_NONE), (256, 64, 2, 96, 96, False, ops.ACT_NONE),
    (256, 1024, 2, 80, 80, False, ops.ACT_NONE), (128, 512, 3, 37, 41, True, ops.ACT_LRELU), (192, 256, 2, 50, 50, True, ops.ACT_RELU),
    (64, 256, 8, 160, 160, False, ops.ACT_NONE),
]
lib = _lib.load_dev()
dt = torch.bfloat16
for cin, cout, B, H, W, bias, act in CASES:
    torch.manual_seed(1)
    x = ops.nchw_to_nhwc(torch.randn(B, cin, H, W, device="cuda"), dt)
    pw = ops.pack_conv_weight(torch.randn(cout, cin, 1, 1, device="cuda") * 0.05, torch.randn(cout, device="cuda") if bias else None, dt)
    lib.cgan_debug_set_gemm_ws(ctypes.c_int(1))
    y0 = ops.conv2d(x, pw, act=act, slope=0.2)
    for ws in (11, 0):
        lib.cgan_debug_set_gemm_ws(ctypes.c_int(ws))
        ys = [ops.conv2d(x, pw, act=act, slope=0.2).t.clone() for _ in range(4)]
        torch.cuda.synchronize()
        same = all(torch.equal(ys[0], y) for y in ys[1:])
        d = (ys[0].float() - y0.t.float()).abs()
        bad = (d > 2.0 ** -7 * y0.t.float().abs() + 1e-3)
        print("cin %4d cout %4d n%d %dx%d bias %d act %d ws %2d: deterministic %s nan %d max|d| %.4g bad %d first bad %s" % (
            cin, cout, B, H, W, bias, act, ws, same, int(torch.isnan(ys[0].float()).sum()), float(d.nan_to_num(1e9).max()), int(bad.sum()),
            bad.nonzero()[:3].tolist()), flush=True)
lib.cgan_debug_set_gemm_ws(ctypes.c_int(0))
