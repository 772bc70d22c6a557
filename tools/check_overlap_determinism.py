"""GPU box: does running the two branches of an update on two streams change any result that is deterministic on one
stream?  One train step from identical states with the overlap on / off: the BatchNorm running statistics (forward convs
+ statistics: no atomics anywhere) must be bit-identical; parameters after the step may differ by the fp32 atomics of the
bias / weight-gradient reductions only (reported)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from climategan_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
T = bench.build_trainer(dev, torch.bfloat16)
T.G.painter.set_latent_shape((4, 3, bench.H, bench.W), True)
batch = bench.joint_batch(4, 0, dev)
sd_g = {k: v.clone() for k, v in T.G.state_dict().items()}
sd_d = {k: v.clone() for k, v in T.D.state_dict().items()}
res = {}
for name, ov in (("one stream", False), ("two streams", True), ("two streams again", True), ("one stream again", False)):
    T.G.load_state_dict(sd_g)
    T.D.load_state_dict(sd_d)
    ops.touch(*T.G.parameters(), *T.G.buffers(), *T.D.parameters(), *T.D.buffers())
    T.g_opt.state.clear()
    T.d_opt.state.clear()
    T.global_step = 0
    T.overlap_branches = ov
    torch.manual_seed(0)
    g, d = T.train_step(batch)
    torch.cuda.synchronize()
    res[name] = ({k: v.clone() for k, v in T.G.state_dict().items()}, float(g), float(d))
base = res["one stream"][0]
for name in ("one stream again", "two streams", "two streams again"):
    cur = res[name][0]
    bn_bad = [k for k in base if "running_" in k and not torch.equal(base[k], cur[k])]
    worst = max(((base[k].float() - cur[k].float()).abs().max().item() / (base[k].float().abs().max().item() + 1e-12), k)
                for k in base if base[k].dtype.is_floating_point and "running_" not in k)
    print("%-18s vs one stream: %d of %d BatchNorm running statistics differ bitwise %s; largest relative parameter "
          "difference %.3g (%s); losses %.6f %.6f vs %.6f %.6f"
          % (name, len(bn_bad), sum("running_" in k for k in base), bn_bad[:3], worst[0], worst[1], res[name][1], res[name][2],
             res["one stream"][1], res["one stream"][2]), flush=True)
