"""Which call sites launch the separate activation-backward / element-wise passes of one joint train step, and on what shapes
(ops.act_bwd, ops.eltwise_*): usage (GPU box): python tools/trace_small_ops.py [per_domain]"""
import collections
import sys
import traceback
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from climategan_amd import ops  # noqa: E402

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
T = bench.build_trainer(dev, torch.bfloat16, freeze=True)
T.G.painter.set_latent_shape((bs, 3, bench.H, bench.W), True)
batch = bench.joint_batch(bs, 0, dev)
for _ in range(2):
    T.train_step(batch)
torch.cuda.synchronize()
log = collections.Counter()


def wrap(name):
    f = getattr(ops, name)

    def g(*a, **k):
        t = a[0].t if hasattr(a[0], "t") else a[0]
        st = [s for s in traceback.extract_stack()[:-1] if "climategan_amd" in s.filename and "ops.py" not in s.filename]
        site = " < ".join("%s:%d" % (Path(s.filename).name, s.lineno) for s in st[-3:][::-1])
        log[(name, tuple(t.shape), site)] += 1
        return f(*a, **k)
    setattr(ops, name, g)


for n in ("act_bwd", "eltwise_mul", "eltwise_relu", "eltwise_scale", "add", "add_act", "sumpool2x2", "resize_nearest"):
    if hasattr(ops, n):
        wrap(n)
T.train_step(batch)
torch.cuda.synchronize()
tot = 0
for (name, shape, site), c in sorted(log.items(), key=lambda kv: -kv[1] * torch.Size(kv[0][1]).numel()):
    mb = c * torch.Size(shape).numel() * 2 / 1e6
    tot += mb
    print("%-14s x%-3d %-24s %8.1f MB/map-pass  %s" % (name, c, shape, mb, site))
T.close()
