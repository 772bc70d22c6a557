"""Dev tool: run the fused SPADE kernel a few times on one shape (for rocprofv3 --pmc)."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from climategan_amd import fill, ops
dt = torch.bfloat16
B, C, R = 8, int(sys.argv[1]) if len(sys.argv) > 1 else 40, 640
if len(sys.argv) > 2:     # development ablation bits of the fused kernel (1 skip hidden map, 2 skip main MFMAs, 8 skip stores)
    import ctypes
    from climategan_amd import _lib
    _lib.load_dev().cgan_debug_set_spade_ablation(ctypes.c_int(int(sys.argv[2])))
cond = ops.nchw_to_nhwc(torch.from_numpy(fill.uniform((B, 3, 640, 640), 1)).cuda(), dt, cs=4)
g = torch.Generator(device="cuda"); g.manual_seed(1)
w = [torch.randn(s, device="cuda", generator=g) * 0.05 for s in [(128, 3, 3, 3), (128,), (C, 128, 3, 3), (C,), (C, 128, 3, 3), (C,)]]
pk = ops.pack_spade_weights(*w, dt)
x = ops.NHWC(torch.randn((B, R, R, ops.cs8(C)), device="cuda", generator=g).to(dt), C)
mean, rstd = ops.instnorm_stats(x)
for _ in range(4):
    ops.spade_fused(x, mean, rstd, cond, pk, act=ops.ACT_LRELU)
torch.cuda.synchronize()
