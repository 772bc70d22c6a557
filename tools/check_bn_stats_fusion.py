"""GPU box: the Masker encoder's BatchNorm running statistics after ONE training-mode forward, statistics from the conv
epilogue (norms.FUSE_BN_STATS) against the separate statistics pass: per layer, the largest relative difference of
running_var and the largest difference of running_mean in units of the layer's spread."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from climategan_amd import autograd as ag, norms, ops  # noqa: E402

dev = torch.device("cuda:0")
T = bench.build_trainer(dev, torch.bfloat16, tasks=("d", "s", "m"))
batch = bench.joint_batch(4, 0, dev, domains=("r", "s"))
x = torch.cat([batch["r"]["data"]["x"], batch["s"]["data"]["x"]], 0)
sd0 = {k: v.clone() for k, v in T.G.state_dict().items()}
res = {}
for fuse in (True, False):
    T.G.load_state_dict(sd0)
    ops.touch(*T.G.parameters(), *T.G.buffers())
    norms.FUSE_BN_STATS = fuse
    T.G.train()
    with torch.enable_grad(), ag.bn_groups(2):
        z = T.G.encode(x)
    torch.cuda.synchronize()
    res[fuse] = {k: v.clone().float() for k, v in T.G.state_dict().items() if "running_" in k}
    res[fuse]["z"] = z[0].t.detach().float().clone()
wv, wm = [], []
for k in res[True]:
    if k.endswith("running_var"):
        va, vb = (res[True][k] - 0.9) * 10, (res[False][k] - 0.9) * 10          # the two groups' batch variances, averaged-ish
        ma, mb = res[True][k.replace("running_var", "running_mean")], res[False][k.replace("running_var", "running_mean")]
        ok = vb > 1e-6
        wv.append((((va / vb - 1).abs() * ok).max().item(), k))
        wm.append(((((ma - mb).abs() * 10) / vb.clamp_min(1e-6).sqrt() * ok).max().item(), k.replace("running_var", "running_mean")))
for name, w in (("batch variance, largest relative difference", wv), ("batch mean, largest difference in standard deviations", wm)):
    w.sort(reverse=True)
    print(name)
    for v, k in w[:12]:
        print("   %-60s %.3e" % (k, v))
    print("   median over layers %.3e" % sorted(v for v, _ in w)[len(w) // 2])
dz = (res[True]["z"] - res[False]["z"]).abs()
print("z: max |diff| %.4g of max |z| %.4g, mean |diff| %.4g of mean |z| %.4g" % (dz.max(), res[False]["z"].abs().max(), dz.mean(), res[False]["z"].abs().mean()))
