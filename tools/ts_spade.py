"""Dev tool (GPU box): per-phase cycle breakdown of the fused SPADE kernel from in-kernel timestamps."""
import ctypes, sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from climategan_amd import _lib, fill, ops
dt = torch.bfloat16
B = 8
lib = _lib.load()
cond = ops.nchw_to_nhwc(torch.from_numpy(fill.uniform((B, 3, 640, 640), 1)).cuda(), dt, cs=4)
for C, R in [(40, 640), (20, 640)]:
    g = torch.Generator(device="cuda"); g.manual_seed(C + R)
    w = [torch.randn(s, device="cuda", generator=g) * 0.05 for s in [(128, 3, 3, 3), (128,), (C, 128, 3, 3), (C,), (C, 128, 3, 3), (C,)]]
    pk = ops.pack_spade_weights(*w, dt)
    x = ops.NHWC(torch.randn((B, R, R, ops.cs8(C)), device="cuda", generator=g).to(dt), C)
    mean, rstd = ops.instnorm_stats(x)
    nwg = B * (R // 16) ** 2
    for _ in range(3):
        ops.spade_fused(x, mean, rstd, cond, pk, act=ops.ACT_LRELU)
    ts = torch.zeros((nwg, 8), dtype=torch.int64, device="cuda")
    lib.cgan_debug_set_spade_tsbuf(ctypes.c_void_p(ts.data_ptr()))
    ops.spade_fused(x, mean, rstd, cond, pk, act=ops.ACT_LRELU)
    torch.cuda.synchronize()
    lib.cgan_debug_set_spade_tsbuf(ctypes.c_void_p(0))
    t = ts.cpu().double()
    names = ["phase0(cond+lut+prm)", "setup", "hidden h0", "main h0", "barrier", "hidden h1", "main h1", "epilogue"]
    # timestamps: 0 start,1 after phase0,2 after hidden h0,3 after main h0,4 after hidden h1 (incl. barrier),5 after main h1,6 end
    d = {
        "phase0": t[:, 1] - t[:, 0], "hidden h0": t[:, 2] - t[:, 1], "main h0": t[:, 3] - t[:, 2],
        "barrier+hidden h1": t[:, 4] - t[:, 3], "main h1": t[:, 5] - t[:, 4], "epilogue": t[:, 6] - t[:, 5],
        "total": t[:, 6] - t[:, 0],
    }
    span = (t[:, 6].max() - t[:, 0].min()).item()
    print("C=%d: kernel span %.0f ticks; per-WG mean ticks:" % (C, span), {k: int(v.mean().item()) for k, v in d.items()})
