import torch, torch.nn.functional as F
from climategan_amd import ops
dt = torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(21)
d = torch.randn((2, 1, 20, 24), device="cuda", generator=g) * 2
pd = ops.pair_from_nchw(d, dt); d = ops.nhwc_to_nchw(pd)
g = torch.Generator(device="cuda").manual_seed(22)
s = torch.randn((2, 11, 20, 24), device="cuda", generator=g) * 3
ps = ops.pair_from_nchw(s, dt); s = ops.nhwc_to_nchw(ps)
g = torch.Generator(device="cuda").manual_seed(23)
x = torch.randn((2, 3, 64, 80), device="cuda", generator=g).clamp(-1, 1)
cond = ops.nhwc_to_nchw(ops.make_m_cond(pd, ps, x))
mn = d.reshape(2, -1).min(1)[0].reshape(2, 1, 1, 1)
t0 = d - mn
ref = torch.cat([t0 / t0.reshape(2, -1).max(1)[0].reshape(2, 1, 1, 1), torch.softmax(s, dim=1),
                 F.interpolate(x.cpu(), (20, 24), mode="bilinear", align_corners=True).cuda()], 1)
refc = torch.cat([(t0 / t0.reshape(2, -1).max(1)[0].reshape(2, 1, 1, 1)).cpu(), torch.softmax(s.cpu(), dim=1),
                 F.interpolate(x.cpu(), (20, 24), mode="bilinear", align_corners=True)], 1).cuda()
print("vs gpu torch", (cond - ref).abs().amax(dim=(0, 2, 3)))
print("vs cpu torch", (cond - refc).abs().amax(dim=(0, 2, 3)))
xh, xw, h, w = 64, 80, 20, 24
sy = torch.tensor((xh - 1), dtype=torch.float32) / torch.tensor((h - 1), dtype=torch.float32)
sx = torch.tensor((xw - 1), dtype=torch.float32) / torch.tensor((w - 1), dtype=torch.float32)
oy = torch.arange(h, dtype=torch.float32, device="cuda"); ox = torch.arange(w, dtype=torch.float32, device="cuda")
fy = oy * sy.cuda(); fx = ox * sx.cuda()
y0 = fy.floor().clamp(max=xh - 1).long(); x0 = fx.floor().clamp(max=xw - 1).long()
y1 = (y0 + 1).clamp(max=xh - 1); x1 = (x0 + 1).clamp(max=xw - 1)
ly = (fy - y0.float())[None, None, :, None]; lx = (fx - x0.float())[None, None, None, :]
g = lambda yy, xx: x[:, :, yy][:, :, :, xx]
mine = (1 - ly) * ((1 - lx) * g(y0, x0) + lx * g(y0, x1)) + ly * ((1 - lx) * g(y1, x0) + lx * g(y1, x1))
tc = F.interpolate(x.cpu(), (20, 24), mode="bilinear", align_corners=True).cuda()
tg = F.interpolate(x, (20, 24), mode="bilinear", align_corners=True)
print("formula vs torch cpu", (mine - tc).abs().max().item(), "vs torch gpu", (mine - tg).abs().max().item(), "kernel vs formula",
      (cond[:, 12:] - mine).abs().max().item(), "cpu vs gpu torch", (tc - tg).abs().max().item())
idx = (cond[:, 12:] - tc).abs().flatten().argmax().item()
print("worst at", idx, "oy/ox", (idx // 24) % 20, idx % 24)
