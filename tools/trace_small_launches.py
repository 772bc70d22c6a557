"""Who issues the SMALL torch-side launches (copies, fills, adds, cats) of one joint train step at a rank's share of the
global batch: torch.profiler with Python stacks; every aten op that launched a device kernel is keyed by (op, enclosing
autograd node or op, first climategan_amd frame).  usage (GPU box): python tools/trace_small_launches.py [per_domain]"""
import collections
import re
import sys
from pathlib import Path

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
T = bench.build_trainer(dev, torch.bfloat16, freeze=True)
T.G.painter.set_latent_shape((bs, 3, bench.H, bench.W), True)
batch = bench.joint_batch(bs, 0, dev)
for _ in range(4):
    T.train_step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    T.train_step(batch)
    torch.cuda.synchronize()

def short(n):
    n = re.sub(r"^void ", "", n).replace("(anonymous namespace)::", "").replace("at::native::", "")
    return re.sub(r"\(.*", "", n)[:70]


copies = collections.Counter()
ev = prof.events()
by_kernel = collections.Counter()
sites = collections.Counter()
dur = collections.Counter()
for e in ev:
    if str(e.device_type).endswith("CPU") and e.kernels:            # a host op that launched device work itself
        if e.cpu_children and any(c.kernels for c in e.cpu_children):
            continue                                                 # count the innermost launching op only
        if "cgan" in e.name:
            continue
        frame = next((s for s in (e.stack or []) if "climategan_amd" in s or "bench.py" in s), "")
        frame = frame.split("climategan_amd/")[-1][:60]
        par = e.cpu_parent
        chain = []
        while par is not None and len(chain) < 3:
            chain.append(par.name[:48])
            par = par.cpu_parent
        key = (e.name[:28], " < ".join(chain), frame)
        sites[key] += len(e.kernels)
        dur[key] += sum(k.duration for k in e.kernels)
        for k in e.kernels:
            by_kernel[short(k.name)] += 1
            if "copyBuffer" in k.name or "Memcpy" in k.name or "FillFunctor<float>" in k.name:
                copies[(short(k.name)[:40],) + key] += 1
print("device launches of torch ops in one step at %d per domain: %d" % (bs, sum(sites.values())))
for k, n in by_kernel.most_common(12):
    print("  %5d  %s" % (n, k))
print("by site (launches, device us, op, enclosing, frame):")
for key, n in sites.most_common(45):
    print("  %4d %8.1f  %-28s | %-70s | %s" % (n, dur[key], key[0], key[1], key[2]))
print("copies and fp32 fills by site:")
for key, n in copies.most_common(25):
    print("  %4d  %-40s %-24s | %-60s | %s" % ((n,) + key))
T.close()
