"""Dev tool (GPU box): host-side cost of the pieces of one op call (stream handle, descriptor, ctypes call)."""
import ctypes as C
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from climategan_amd import _lib, ops  # noqa: E402

torch.cuda.init()
x = torch.zeros(8, device="cuda")


def t(f, n=20000):
    f()
    a = time.perf_counter()
    for _ in range(n):
        f()
    return (time.perf_counter() - a) / n * 1e6


print("torch.cuda.current_stream().cuda_stream  %.2f us" % t(lambda: torch.cuda.current_stream().cuda_stream))
print("ops._stream()                            %.2f us" % t(ops._stream))
print("ops._ptr(x)                              %.2f us" % t(lambda: ops._ptr(x)))
print("torch.empty_like(x)                      %.2f us" % t(lambda: torch.empty_like(x)))
print("ops.NHWC(...)                            %.2f us" % t(lambda: ops.NHWC(x.view(1, 1, 1, 8), 8)))
xx = ops.NHWC(torch.zeros(1, 8, 8, 64, device="cuda", dtype=torch.bfloat16), 64)
mean = torch.zeros(1, 64, device="cuda")
print("ops.norm_act_apply (whole call, tiny map) %.2f us" % t(lambda: ops.norm_act_apply(xx, mean, mean), 5000))
torch.cuda.synchronize()
