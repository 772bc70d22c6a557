"""Dev tool (GPU box): ablation timing of the fused SPADE kernel (which phase costs what)."""
import ctypes, sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from climategan_amd import _lib, fill, ops
dt = torch.bfloat16
B = 8
lib = _lib.load_dev()
cond = ops.nchw_to_nhwc(torch.from_numpy(fill.uniform((B, 3, 640, 640), 1)).cuda(), dt, cs=4)
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0
lib.cgan_debug_set_spade_variant(ctypes.c_int(variant))
for C, R in [(40, 640), (20, 640)]:
    g = torch.Generator(device="cuda"); g.manual_seed(C + R)
    w = [torch.randn(s, device="cuda", generator=g) * 0.05 for s in [(128, 3, 3, 3), (128,), (C, 128, 3, 3), (C,), (C, 128, 3, 3), (C,)]]
    pk = ops.pack_spade_weights(*w, dt)
    x = ops.NHWC(torch.randn((B, R, R, ops.cs8(C)), device="cuda", generator=g).to(dt), C)
    mean, rstd = ops.instnorm_stats(x)
    for bits, name in [(0, "full"), (1, "-hidden"), (2, "-mfma"), (8, "-stores"), (1 | 2, "-hidden-mfma"), (1 | 2 | 8, "-hidden-mfma-stores")]:
        lib.cgan_debug_set_spade_ablation(ctypes.c_int(bits))
        for _ in range(2):
            ops.spade_fused(x, mean, rstd, cond, pk, act=ops.ACT_LRELU)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.spade_fused(x, mean, rstd, cond, pk, act=ops.ACT_LRELU)
        e1.record(); torch.cuda.synchronize()
        print("v%d C=%d %-28s %8.1f us" % (variant, C, name, e0.elapsed_time(e1) / 5 * 1e3))
    lib.cgan_debug_set_spade_ablation(ctypes.c_int(0))
