"""Dev tool (GPU box): time the fused SPADE kernel alone on the Painter's layer shapes, per tile variant."""
import ctypes
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from climategan_amd import _lib, fill, ops  # noqa: E402
import os
if os.environ.get("CGAN_LIB"):      # A/B against another build of the library (same box, same call)
    _lib.LIB_PATH = Path(os.environ["CGAN_LIB"]).resolve()

dt = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else torch.float16
B = int(os.environ.get("SPADE_B", "8"))
shapes = [(40, 640), (20, 640), (80, 320), (160, 160), (640, 20), (640, 5)]
if os.environ.get("SPADE_SHAPES"):   # "C:R,C:R,..."
    shapes = [tuple(int(v) for v in cr.split(":")) for cr in os.environ["SPADE_SHAPES"].split(",")]
variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]   # 0 = default tiles; 1-5 force NCT; +10 = 8-wave kernel
lib = _lib.load_dev()
cond = ops.nchw_to_nhwc(torch.from_numpy(fill.uniform((B, 3, 640, 640), 1)).cuda(), dt, cs=4)
for C, R in shapes:
    g = torch.Generator(device="cuda"); g.manual_seed(C + R)
    w = [torch.randn(s, device="cuda", generator=g) * 0.05 for s in
         [(128, 3, 3, 3), (128,), (C, 128, 3, 3), (C,), (C, 128, 3, 3), (C,)]]
    pk = ops.pack_spade_weights(*w, dt)
    x = ops.NHWC(torch.randn((B, R, R, ops.cs8(C)), device="cuda", generator=g).to(dt), C)
    mean, rstd = ops.instnorm_stats(x)
    flops = B * R * R * 2 * (27 * 128 + 2 * 1152 * C)
    line = "C=%3d R=%3d  %6.1f GFLOP |" % (C, R, flops / 1e9)
    for v in variants:
        lib.cgan_debug_set_spade_variant(ctypes.c_int(v % 10))
        lib.cgan_debug_set_spade_waves(ctypes.c_int(8 if v >= 10 else 4))   # 0-5: 4-wave kernel, 10-15: specialised
        ref = ops.spade_fused(x, mean, rstd, cond, pk, act=ops.ACT_LRELU).t.float()
        if v == variants[0]:
            ref0 = ref
        else:
            line += " (maxdiff %.2g)" % (ref - ref0).abs().max().item()
        for _ in range(2):
            ops.spade_fused(x, mean, rstd, cond, pk, act=ops.ACT_LRELU)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            ops.spade_fused(x, mean, rstd, cond, pk, act=ops.ACT_LRELU)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        line += "  v%d %8.1f us %6.0f TF |" % (v, us, flops / us / 1e6)
    lib.cgan_debug_set_spade_variant(ctypes.c_int(0))
    lib.cgan_debug_set_spade_waves(ctypes.c_int(8))
    print(line)
