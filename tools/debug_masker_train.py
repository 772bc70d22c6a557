"""Dev tool: one masker forward + backward (domain-r style terms) on the HIP training path; prints loss and grad norms."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
from climategan_amd import fill, ops, losses as L
from climategan_amd.config import default_opts
from climategan_amd.generator import create_generator
opts = default_opts(); opts.tasks = ["d", "s", "m"]
G = create_generator(opts, device="cuda")
shapes = {k: tuple(v.shape) for k, v in G.state_dict().items()}
G.load_state_dict({k: torch.from_numpy(v) for k, v in fill.fill_state_dict(shapes, 3, gain=1.6).items()})
G.train(); G.set_compute_dtype(torch.bfloat16 if "bf16" in sys.argv else torch.float16)
H, W = 128, 160
G.decoders["d"]._target_size = W // 4
G.decoders["s"].set_target_size((H // 4, W // 4))
x = torch.from_numpy(fill.uniform((2, 3, H, W), 5)).cuda()
z = G.encode(x)
d, z_depth = G.decoders["d"].forward_nhwc(z)
s = G.decoders["s"].forward_nhwc(z, z_depth)
m = G.mask_nhwc(z, z_depth=z_depth)
print("shapes", d.t.shape, s.t.shape, m.t.shape, d.t.requires_grad, s.t.requires_grad, m.t.requires_grad)
loss = L.MinentLoss()(L.softmax(s)) * 0.001 + L.TVLoss()(L.sigmoid(m)) + L.MinentLoss(2, 0.1)(L.sigmoid_pair(m)) * 0.5
loss.backward()
torch.cuda.synchronize()
print("loss", loss.item())
tot = 0; none = 0; bad = 0
for k, p in G.named_parameters():
    if not p.requires_grad: continue
    if p.grad is None: none += 1; continue
    tot += 1
    if not torch.isfinite(p.grad).all(): bad += 1
print("params with grad", tot, "without", none, "non-finite", bad)
for k in ("encoder.conv1.weight", "encoder.layer3.10.conv2.weight", "decoders.s.aspp.conv_out.conv.weight", "decoders.m.proj_conv.conv.module.weight_bar"):
    p = dict(G.named_parameters())[k]; print(k, None if p.grad is None else float(p.grad.norm()))
print("max mem GB", torch.cuda.max_memory_allocated() / 2**30)
