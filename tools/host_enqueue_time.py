"""Host time to ENQUEUE one joint train step (the call returns before the device is done) against the device time per step:
how far ahead of the device is the host?  usage (GPU box): python tools/host_enqueue_time.py [steps]"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402

dev = torch.device("cuda:0")
T = bench.build_trainer(dev, torch.bfloat16, freeze=True)
T.G.painter.set_latent_shape((4, 3, bench.H, bench.W), True)
batch = bench.joint_batch(4, 0, dev)
for _ in range(5):
    T.train_step(batch)
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
# (a) every step drained before the next: the host's own time to walk one step while the device is never the brake for long
host = []
for _ in range(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    T.train_step(batch)
    host.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize()
# (b) back to back, one drain at the end: the steady-state step time
t0 = time.perf_counter()
for _ in range(n):
    T.train_step(batch)
torch.cuda.synchronize()
dev_ms = (time.perf_counter() - t0) * 1e3 / n
s = sorted(host)
print("host enqueue per step (device idle at entry): median %.1f min %.1f max %.1f ms; steady-state step %.1f ms" % (s[n // 2], s[0], s[-1], dev_ms))
T.close()
