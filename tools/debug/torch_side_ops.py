"""aten ops that launch kernels in one joint train step (slice size): counts by op and by calling line (torch.profiler, with_stack)"""
import collections
import sys
from pathlib import Path
import torch
from torch.profiler import ProfilerActivity, profile
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import bench  # noqa: E402

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
T = bench.build_trainer(dev, torch.bfloat16, freeze=True)
T.G.painter.set_latent_shape((bs, 3, bench.H, bench.W), True)
batch = bench.joint_batch(bs, 0, dev)
for _ in range(3):
    T.train_step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    T.train_step(batch)
    torch.cuda.synchronize()
rows = [r for r in prof.key_averages(group_by_stack_n=6) if r.key.startswith("aten::")]
agg = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
for r in rows:
    dt = getattr(r, "device_time_total", getattr(r, "cuda_time_total", 0.0))
    a = agg[r.key]
    a[0] += r.count
    a[1] += dt
    st = [x for x in (r.stack or []) if "climategan_amd" in x or "bench.py" in x]
    a[2][st[0].strip()[-100:] if st else "?"] += r.count
for name, (c, dt, sites) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:16]:
    print("%-30s %5d calls  device %.2f ms" % (name, c, dt / 1e3))
    for sname, k in sites.most_common(6):
        print("      %4d  %s" % (k, sname))
T.close()
