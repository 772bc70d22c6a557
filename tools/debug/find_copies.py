"""Which host-side ops of one joint train step launch device-to-device copies (torch profiler, grouped by the op above)?"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
dev = torch.device("cuda:0")
T = bench.build_trainer(dev, torch.bfloat16)
batch = bench.joint_batch(4, 0, dev)
T.G.painter.set_latent_shape((4, 3, bench.H, bench.W), True)
for _ in range(3):
    T.train_step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    T.train_step(batch)
    torch.cuda.synchronize()
ev = prof.events()
cnt = collections.Counter()
for e in ev:
    if e.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::fill_", "aten::zero_", "aten::zeros", "aten::cat", "aten::add", "aten::add_"):
        # climb to the nearest python frame of this package
        st = [s for s in (e.stack or []) if "climategan_amd" in s or "bench.py" in s]
        key = (e.name, str(e.input_shapes)[:60], st[0].split("/")[-1][:80] if st else "(no package frame: autograd engine?)")
        cnt[key] += 1
for k, v in cnt.most_common(45):
    print(v, k)
