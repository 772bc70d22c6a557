"""is tests/test_gpu_vgg._run reproducible run to run, and where does FUSE_RELU_MASK change the gradient?"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent / "tests"))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import test_gpu_vgg as tv
from climategan_amd import norms
for dt in (torch.float16, torch.bfloat16):
    a = tv._run(dt); b = tv._run(dt)
    norms.FUSE_RELU_MASK = False
    c = tv._run(dt); d = tv._run(dt)
    norms.FUSE_RELU_MASK = True
    print(dt, "fused twice equal:", torch.equal(a[1], b[1]), "unfused twice equal:", torch.equal(c[1], d[1]),
          "fused vs unfused:", torch.equal(a[1], c[1]), "max diff", (a[1] - c[1]).abs().max().item(), "scale", c[1].abs().max().item(),
          "n diff", (a[1] != c[1]).sum().item(), "loss", a[0], c[0])
