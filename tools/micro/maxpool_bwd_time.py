"""Dev tool (GPU box): time of the ResNet stem's max-pool backward at the training step's size, and a check against torch."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from climategan_amd import ops  # noqa: E402
from climategan_amd.autograd import MaxPool3x3s2Fn  # noqa: E402

x = ops.nchw_to_nhwc(torch.randn(8, 64, 320, 320, device="cuda"), torch.bfloat16)
xt = x.t.clone().requires_grad_(True)
y = MaxPool3x3s2Fn.apply(xt, 64)
dy = torch.randn_like(y)
for _ in range(3):
    (dx,) = torch.autograd.grad(y, xt, dy, retain_graph=True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    (dx,) = torch.autograd.grad(y, xt, dy, retain_graph=True)
e1.record()
torch.cuda.synchronize()
print("maxpool3x3s2 backward 8x320x320x64 (incl. autograd call): %.1f us" % (e0.elapsed_time(e1) * 50))
ref = ops.nhwc_to_nchw(x).float().requires_grad_(True)
yt = torch.nn.functional.max_pool2d(ref, 3, 2, 1)
yt.backward(ops.nhwc_to_nchw(ops.NHWC(dy, 64)).float())
print("max |dx - torch|:", (ops.nhwc_to_nchw(ops.NHWC(dx, 64)).float() - ref.grad).abs().max().item())
