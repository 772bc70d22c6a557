"""Dev tool (GPU box): weight-gradient time (kernel + split reduction, events over 20 calls) against the number of pixel
splits, per layer shape -- the data behind wgrad_plan's choice (conv_wgrad.hip)."""
import ctypes
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from climategan_amd import _lib, ops  # noqa: E402

lib = _lib.load_dev()
dt = torch.bfloat16
SHAPES = [
    # name, cin, cout, k, stride, pad, dil, bs, H, tiles (cooperative: pairs)
    ("l3 3x3 d2 256->256 n8 80", 256, 256, 3, 1, 2, 2, 8, 80, 36),
    ("l3 1x1 256->1024 n8 80", 256, 1024, 1, 1, 0, 1, 8, 80, 16),
    ("l3 1x1 1024->256 n8 80", 1024, 256, 1, 1, 0, 1, 8, 80, 16),
    ("l4 3x3 d4 512->512 n8 80", 512, 512, 3, 1, 4, 4, 8, 80, 144),
    ("l4 1x1 512->2048 n8 80", 512, 2048, 1, 1, 0, 1, 8, 80, 64),
    ("aspp 3x3 d6 2048->256 n8 80", 2048, 256, 3, 1, 6, 6, 8, 80, 288),
    ("l2 3x3 128->128 n8 80", 128, 128, 3, 1, 1, 1, 8, 80, 9),
    ("l2 1x1 128->512 n8 80", 128, 512, 1, 1, 0, 1, 8, 80, 4),
    ("l1 3x3 64->64 n8 160", 64, 64, 3, 1, 1, 1, 8, 160, 9),
    ("l1 1x1 64->256 n8 160", 64, 256, 1, 1, 0, 1, 8, 160, 4),
    ("spade gb 128->80 n4 640", 128, 80, 3, 1, 1, 1, 4, 640, 9),
    ("spade gb 128->40 n4 640", 128, 40, 3, 1, 1, 1, 4, 640, 9),
    ("spade shared 3->128 n4 640", 3, 128, 3, 1, 1, 1, 4, 640, 1),
    ("spade gb 128->160 n4 320", 128, 160, 3, 1, 1, 1, 4, 320, 9),
    ("painter 20->20 n4 640", 20, 20, 3, 1, 1, 1, 4, 640, 1),
    ("D 4x4 s2 256->512 n4 80", 256, 512, 4, 2, 1, 1, 4, 80, 64),
]
only = sys.argv[1] if len(sys.argv) > 1 else ""
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 0           # cooperative kernel: pixels per stage (64 / 32; 0 = the planner decides)
lib.cgan_debug_set_wgrad_coop_chunk(ctypes.c_int(chunk))
slots = 1024 if chunk == 32 else 512
if len(sys.argv) > 3:                                         # cgan_debug_set_wgrad_tile3x3: 0 off, 1 default, 2 wherever it applies
    lib.cgan_debug_set_wgrad_tile3x3(ctypes.c_int(int(sys.argv[3])))
for name, cin, cout, k, stride, pad, dil, bs, H, tiles in SHAPES:
    if only not in name:
        continue
    torch.manual_seed(0)
    x = ops.nchw_to_nhwc(torch.randn(bs, cin, H, H, device="cuda"), dt)
    Ho = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1
    dy = ops.nchw_to_nhwc(torch.randn(bs, cout, Ho, Ho, device="cuda"), dt)
    dw = torch.zeros(cout, cin, k, k, device="cuda")
    flops = 2.0 * bs * Ho * Ho * cout * cin * k * k

    def timed():
        for _ in range(3):
            ops.conv2d_bwd_weight(x, dy, (cout, cin, k, k), stride, pad, dil, want_bias=False, dw=dw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.conv2d_bwd_weight(x, dy, (cout, cin, k, k), stride, pad, dil, want_bias=False, dw=dw)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 50

    lib.cgan_debug_set_wgrad(ctypes.c_int(0), ctypes.c_int(0))
    base = timed()
    out = []
    cands = sorted({max(1, round(f * slots / tiles)) for f in (0.5, 1, 1.5, 2, 3, 4)} |
                   {max(1, (m * slots) // tiles) for m in (1, 2, 3, 4, 6)})
    for sp in cands:
        lib.cgan_debug_set_wgrad(ctypes.c_int(-sp), ctypes.c_int(0))
        out.append((sp, timed()))
    lib.cgan_debug_set_wgrad(ctypes.c_int(0), ctypes.c_int(0))
    best = min(out, key=lambda t: t[1])
    print("%-30s plan %.1f us (%.0f TF) | best %d splits (%d wgs) %.1f us | " % (name, base, flops / base / 1e6, best[0], best[0] * tiles, best[1]) +
          "  ".join("%d:%.1f" % t for t in out), flush=True)
