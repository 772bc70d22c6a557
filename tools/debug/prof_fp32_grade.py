"""apply_events in the fp32-grade mode (G.float()), a few batches: run under rocprofv3 --kernel-trace --stats."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from climategan_amd import fill
from climategan_amd.config import default_opts
from climategan_amd.trainer import Trainer
opts = default_opts(); opts.tasks = ["d", "s", "m", "p"]
T = Trainer(opts, device="cuda").setup(inference=True)
shapes = {k: tuple(v.shape) for k, v in T.G.state_dict().items()}
T.G.load_state_dict({k: torch.from_numpy(v) for k, v in fill.fill_state_dict(shapes, seed=0, gain=1.0).items()}) if False else None
T.G.eval().float()
x = torch.rand(16, 3, 640, 640, device="cuda") * 2 - 1
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    T.infer_all(x, numpy=True, bin_value=0.5, half=False)
torch.cuda.synchronize()
