import sys, collections
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch, bench
from test_gpu_determinism import _run_steps
dev = torch.device("cuda:0")
T = bench.build_trainer(dev, torch.bfloat16)
T.G.painter.set_latent_shape((2, 3, bench.H, bench.W), True)
batch = bench.joint_batch(2, 0, dev)
sd_g = {k: v.clone() for k, v in T.G.state_dict().items()}
sd_d = {k: v.clone() for k, v in T.D.state_dict().items()}
a = _run_steps(T, batch, sd_g, sd_d, False, steps=1)
b = _run_steps(T, batch, sd_g, sd_d, False, steps=1)
c = _run_steps(T, batch, sd_g, sd_d, True, steps=1)
for name, r in (("again", b), ("two streams", c)):
    bad = [k for k in a if not torch.equal(a[k], r[k])]
    grp = collections.Counter(".".join(k.split(".")[:3]) + (" [bias]" if k.endswith("bias") else "") for k in bad)
    print(name, len(bad), "of", len(a))
    for g, n in sorted(grp.items()): print("   ", g, n)
