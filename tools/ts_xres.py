"""Dev tool (GPU box): per-phase time breakdown of the x-resident 1x1 kernel (conv1x1_xres.hip) from in-kernel stamps.
Ticks are s_memtime ticks; they are scaled to microseconds with the kernel's span against its event-timed duration."""
import ctypes
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from climategan_amd import _lib, ops  # noqa: E402

lib = _lib.load_dev()
dt = torch.bfloat16
for cin, cout, B, H, W, stats in [(256, 1024, 8, 80, 80, False), (256, 1024, 8, 80, 80, True), (64, 256, 8, 160, 160, False), (128, 512, 8, 80, 80, False)]:
    torch.manual_seed(1)
    x = ops.nchw_to_nhwc(torch.randn(B, cin, H, W, device="cuda"), dt)
    pw = ops.pack_conv_weight(torch.randn(cout, cin, 1, 1, device="cuda") * 0.05, None, dt)
    run = (lambda: ops.conv2d_with_stats(x, pw, groups=1)) if stats else (lambda: ops.conv2d(x, pw))
    for ws in (11, 12):
        lib.cgan_debug_set_gemm_ws(ctypes.c_int(ws))
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        print("%d -> %d n%d %dx%d stats %d ws %d: %.1f us per call (events, incl. the statistics finalize)" % (
            cin, cout, B, H, W, stats, ws, e0.elapsed_time(e1) * 50), flush=True)
    lib.cgan_debug_set_gemm_ws(ctypes.c_int(11))
    nwg = ((B * H * W + 255) // 256 + 7) // 8 * 8
    ts = torch.zeros((nwg, 8, 16), dtype=torch.int64, device="cuda")
    lib.cgan_debug_set_xres_tsbuf(ctypes.c_void_p(ts.data_ptr()))
    run()
    torch.cuda.synchronize()
    lib.cgan_debug_set_xres_tsbuf(ctypes.c_void_p(0))
    lib.cgan_debug_set_gemm_ws(ctypes.c_int(0))
    t = ts.cpu().double()
    live = t[:, 0, 0] > 0
    t = t[live]
    ncb = (cout + 255) // 256
    last = 3 + 2 * (ncb - 1)
    span = (t[:, :, last].max() - t[:, :, 0].min()).item()
    print("   %d workgroups; kernel span %.0f ticks; start skew (last WG start - first) %.0f; per-wave means in ticks:" % (
        t.shape[0], span, (t[:, :, 0].max() - t[:, :, 0].min()).item()))
    print("   x tile load %.0f" % (t[:, :, 1] - t[:, :, 0]).mean().item())
    prev = t[:, :, 1]
    for cb in range(ncb):
        k, e = t[:, :, 2 + 2 * cb], t[:, :, 3 + 2 * cb]
        print("   cout block %d: k loop %.0f  epilogue %.0f" % (cb, (k - prev).mean().item(), (e - k).mean().item()))
        prev = e
    print("   workgroup life (wave 0) mean %.0f min %.0f max %.0f" % tuple(
        f((t[:, 0, last] - t[:, 0, 0])).item() for f in (torch.mean, torch.min, torch.max)))
