"""Duration of single split-precision convolutions (development probe).  usage: python tools/micro/pair_conv_probe.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from climategan_amd import ops  # noqa: E402

dt = torch.bfloat16
for (cin, cout, k, dil, hw, act) in ((512, 512, 3, 1, 80, ops.ACT_RELU), (512, 512, 3, 1, 80, ops.ACT_NONE), (512, 512, 3, 2, 80, ops.ACT_RELU),
                                     (512, 512, 3, 4, 80, ops.ACT_RELU), (256, 256, 3, 2, 80, ops.ACT_RELU), (256, 1024, 1, 1, 80, ops.ACT_NONE),
                                     (1024, 256, 1, 1, 80, ops.ACT_RELU), (2048, 256, 3, 6, 80, ops.ACT_RELU)):
    x = ops.pair_from_nchw(torch.randn(16, cin, hw, hw, device="cuda"), dt)
    w = torch.randn(cout, cin, k, k, device="cuda") * 0.02
    pw = ops.pack_conv_weight(w, None, dt, pair=True)
    pad = dil * (k - 1) // 2
    for _ in range(3):
        y = ops.conv2d(x, pw, pad=pad, dilation=dil, act=act)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        y = ops.conv2d(x, pw, pad=pad, dilation=dil, act=act)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    fl = 2.0 * 16 * hw * hw * cout * cin * k * k * 6
    print("%4d -> %4d k%d d%d act %d: %7.3f ms  %6.1f TFLOP/s (6 products)" % (cin, cout, k, dil, act, ms, fl / ms / 1e9))
