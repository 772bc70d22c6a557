import torch, torch.nn.functional as F
from climategan_amd import ops
for dt in (torch.bfloat16, torch.float16):
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn((2, 16, 40, 40), device="cuda", generator=g)
    p = ops.pair_from_nchw(x, dt); x = ops.nhwc_to_nchw(p)
    for ac in (True, False):
        for size in ((160, 160), (97, 131), (640, 640)):
            y = ops.nhwc_to_nchw(ops.resize_bilinear(p, size, align_corners=ac))
            rc = F.interpolate(x.cpu(), size, mode="bilinear", align_corners=ac).cuda()
            rg = F.interpolate(x, size, mode="bilinear", align_corners=ac)
            print(dt, ac, size, "vs cpu %.3g vs gpu %.3g" % ((y - rc).abs().max().item(), (y - rg).abs().max().item()))
