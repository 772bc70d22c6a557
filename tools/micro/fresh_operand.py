"""Does a kernel that reads a map the PREVIOUS kernel has just written run slower than on a map written long ago?  The 3 -> 128
weight gradient of SPADE's shared conv reads d_pre (420 MB at 4 x 640^2) right after the data-gradient kernel wrote it: 226 us
inside the step, 100 us isolated.  usage (GPU box): python tools/micro/fresh_operand.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from climategan_amd import ops  # noqa: E402

dt = torch.bfloat16
n, H, cin, cout = 4, 640, 3, 128
x = ops.NHWC(torch.randn(n, H, H, 8, device="cuda").to(dt), cin)
src = torch.randn(n, H, H, cout, device="cuda").to(dt)
dy = ops.NHWC(torch.empty_like(src), cout)
dy.t.copy_(src)
dw = torch.zeros(cout, cin, 3, 3, device="cuda")
db = torch.zeros(cout, device="cuda")
other = torch.randn(64 * 1024 * 1024, device="cuda")          # 256 MB of unrelated traffic


def run(prep, reps=10):
    tot = 0.0
    for _ in range(reps):
        prep()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.conv2d_bwd_weight(x, dy, (cout, cin, 3, 3), 1, 1, 1, want_bias=True, dw=dw, dbias=db)
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / reps * 1e3


for _ in range(3):
    ops.conv2d_bwd_weight(x, dy, (cout, cin, 3, 3), 1, 1, 1, want_bias=True, dw=dw, dbias=db)
torch.cuda.synchronize()
print("operand written long ago, device idle before:        %.1f us" % run(lambda: torch.cuda.synchronize()))
print("operand just written by the previous kernel (copy):  %.1f us" % run(lambda: dy.t.copy_(src)))
print("unrelated 256 MB just written by the previous kernel: %.1f us" % run(lambda: other.mul_(1.0001)))
