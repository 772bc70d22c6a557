"""Dev tool (GPU box): per-phase cycle breakdown of the fused SPADE kernel from in-kernel timestamps."""
import ctypes, sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from climategan_amd import _lib, fill, ops
dt = torch.bfloat16
B = 8
lib = _lib.load_dev()
if len(sys.argv) > 1:
    lib.cgan_debug_set_spade_ablation(ctypes.c_int(int(sys.argv[1])))
    print('ablation bits', sys.argv[1])
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
    # timestamps: 0 start, 1 after phase 0, 2 after hidden quarter 0, 5 after the K loop, 6 end
    d = {"phase0": t[:, 1] - t[:, 0], "hidden q0": t[:, 2] - t[:, 1], "K loop": t[:, 5] - t[:, 2],
         "epilogue": t[:, 6] - t[:, 5], "total": t[:, 6] - t[:, 0]}
    # concurrency: how many workgroups are resident at once (expect 512 = 2 per CU)
    import numpy as np
    st = t[:, 0].numpy(); en = t[:, 6].numpy()
    ids = ts[:, 7].cpu().numpy()
    cu_key = ((ids >> 32) << 16) | ((ids & 0xffffffff) >> 8 & 0xff)   # (xcc, se/sh/cu)
    keys = np.unique(cu_key)
    concs = []
    for kx in keys:
        m = cu_key == kx
        ev = np.concatenate([np.stack([st[m], np.ones(m.sum())], 1), np.stack([en[m], -np.ones(m.sum())], 1)])
        ev = ev[np.argsort(ev[:, 0], kind="stable")]
        concs.append(np.cumsum(ev[:, 1]).max())
    print("C=%d: %d distinct CUs seen; per-CU max concurrent WGs: min %d max %d mean %.2f; WGs per CU mean %.1f" % (
        C, len(keys), min(concs), max(concs), np.mean(concs), len(st) / len(keys)))
    span = (t[:, 6].max() - t[:, 0].min()).item()
    print("C=%d: kernel span %.0f ticks; per-WG mean ticks:" % (C, span), {k: int(v.mean().item()) for k, v in d.items()})
