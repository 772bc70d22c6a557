"""cProfile of the HOST side of one joint train step at 4 per domain (the per-rank share at N = 8): where the Python time of the
~3800 launches goes.  The autograd engine runs the backward on its own thread: profiled single-threaded
(torch.autograd.set_multithreading_enabled(False)) so that its Python frames are seen.  usage (GPU box):
python tools/host_profile_step.py [steps]"""
import cProfile
import io
import pstats
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
with torch.autograd.set_multithreading_enabled(False):
    for _ in range(2):
        T.train_step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        torch.cuda.synchronize()
        T.train_step(batch)
    torch.cuda.synchronize()
    print("single-threaded engine, drained per step: %.1f ms per step (host walk + tail)" % ((time.perf_counter() - t0) * 1e3 / n))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        torch.cuda.synchronize()
        T.train_step(batch)
    pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumtime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    print("\n".join(l[:170] for l in s.getvalue().splitlines()))
T.close()
