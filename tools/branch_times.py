"""GPU box: how long the two branches of update_G / update_D take on their streams (CGAN_BRANCH_TIMES=1), next to the step.
usage: python tools/branch_times.py"""
import os
import sys
import time
from pathlib import Path

os.environ["CGAN_BRANCH_TIMES"] = "1"
import torch  # noqa: E402

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402

dev = torch.device("cuda:0")
T = bench.build_trainer(dev, torch.bfloat16)
T.G.painter.set_latent_shape((4, 3, bench.H, bench.W), True)
batch = bench.joint_batch(4, 0, dev)
for _ in range(4):
    T.train_step(batch)
torch.cuda.synchronize()
T.branch_events.clear()
t0 = time.perf_counter()
N = 6
for _ in range(N):
    T.train_step(batch)
torch.cuda.synchronize()
print("step %.1f ms" % ((time.perf_counter() - t0) / N * 1e3))
ev = T.branch_events
for k, name in ((0, "update_G"), (1, "update_D")):
    rows = [e for i, e in enumerate(ev) if i % 2 == k and e[1] is not None]
    if rows:
        m = sum(e[0].elapsed_time(e[1]) for e in rows) / len(rows)
        s = sum(e[0].elapsed_time(e[2]) for e in rows) / len(rows)
        print("%s: main-stream branch (Masker side) ends %.1f ms after the fork, side-stream branch (Painter side) %.1f ms" % (name, m, s))
