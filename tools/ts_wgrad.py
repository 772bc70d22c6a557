"""Dev tool (GPU box): where a wave of the cooperative weight-gradient kernel (conv_wgrad.hip) spends its life, from
in-kernel tick sums (cgan_debug_set_wgrad_tsbuf): waiting for its own LDS-DMA pieces, waiting at the chunk barrier,
issuing the next chunk's pieces, fragment reads + MFMAs.  Staging roles: waves 0 / 1 stage dy, waves 2 / 3 stage x."""
import ctypes
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from climategan_amd import _lib, ops  # noqa: E402

lib = _lib.load_dev()
dt = torch.bfloat16
SHAPES = [
    # name, cin, cout, k, pad, dil, bs, H
    ("l3 3x3 d2 256->256 n8 80", 256, 256, 3, 2, 2, 8, 80),
    ("l3 3x3 d2 256->256 n64 80", 256, 256, 3, 2, 2, 64, 80),
    ("l3 1x1 256->1024 n64 80", 256, 1024, 1, 0, 1, 64, 80),
    ("l4 3x3 d4 512->512 n64 80", 512, 512, 3, 4, 4, 64, 80),
    ("l3 1x1 256->1024 n8 80", 256, 1024, 1, 0, 1, 8, 80),
    ("l3 1x1 1024->256 n8 80", 1024, 256, 1, 0, 1, 8, 80),
    ("aspp 3x3 d6 2048->256 n8 80", 2048, 256, 3, 6, 6, 8, 80),
    ("spade gb 128->80 n4 640", 128, 80, 3, 1, 1, 4, 640),
]
for name, cin, cout, k, pad, dil, bs, H in SHAPES:
    torch.manual_seed(0)
    x = ops.nchw_to_nhwc(torch.randn(bs, cin, H, H, device="cuda"), dt)
    dy = ops.nchw_to_nhwc(torch.randn(bs, cout, H, H, device="cuda"), dt)
    dw = torch.zeros(cout, cin, k, k, device="cuda")

    def run():
        ops.conv2d_bwd_weight(x, dy, (cout, cin, k, k), 1, pad, dil, want_bias=False, dw=dw, use_workspace=True)

    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    flops = 2.0 * bs * H * H * cout * cin * k * k
    NWV = int(__import__('os').environ.get('CGAN_TS_WAVES', '4'))      # 16 for the 16-wave tile (CGAN_DEBUG_WGRAD_COOP_G=4)
    ts = torch.zeros((1 << 16, NWV, 8), dtype=torch.int64, device="cuda")
    lib.cgan_debug_set_wgrad_tsbuf(ctypes.c_void_p(ts.data_ptr()))
    run()
    torch.cuda.synchronize()
    lib.cgan_debug_set_wgrad_tsbuf(ctypes.c_void_p(0))
    t = ts.cpu().double()
    t = t[t[:, 0, 0] > 0]
    if t.shape[0] == 0:
        print("%-30s %.1f us (kernel + reduce, events), %.0f TFLOP/s: not the cooperative kernel" % (name, us, flops / us / 1e6))
        continue
    span = (t[:, :, 1].max() - t[:, :, 0].min()).item()
    life = (t[:, :, 1] - t[:, :, 0])
    print("%-30s %.1f us (kernel + reduce, events), %.0f TFLOP/s; %d workgroups, span %.0f ticks, chunks per workgroup %.1f"
          % (name, us, flops / us / 1e6, t.shape[0], span, t[:, 0, 6].mean().item()))
    for w in range(0, NWV, max(NWV // 4, 1)):
        s = t[:, w, 2:6].mean(0)
        print("    wave %d (%s): life %.0f | own DMA wait %.0f  barrier %.0f  issue %.0f  reads + MFMA %.0f   (per chunk: %.0f %.0f %.0f %.0f)" % (
            (w, "dy" if w < 2 else "x ", life[:, w].mean().item()) + tuple(s.tolist()) + tuple((s / t[:, w, 6].mean()).tolist())))
