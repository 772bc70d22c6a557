"""Per-step wall times of the headline joint train step in ONE process (is the box-to-box / run-to-run spread of bench.py's
ms_per_step a property of the process or of the step?).  usage (GPU box): python tools/step_times.py [steps]"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402

dev = torch.device("cuda:0")
T = bench.build_trainer(dev, torch.bfloat16)
T.G.painter.set_latent_shape((4, 3, bench.H, bench.W), True)
batch = bench.joint_batch(4, 0, dev)
for _ in range(5):
    T.train_step(batch)
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
if len(sys.argv) > 2 and sys.argv[2] == "nogc":       # is the periodic long step Python's generational garbage collector?
    import gc
    gc.collect()
    gc.disable()
ts = []
for _ in range(n):
    t0 = time.perf_counter()
    T.train_step(batch)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
s = sorted(ts)
print("steps %d: median %.2f min %.2f p90 %.2f max %.2f ms" % (n, s[n // 2], s[0], s[int(n * 0.9)], s[-1]))
print(" ".join("%.1f" % t for t in ts))
