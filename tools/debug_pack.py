import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from climategan_amd import fill, ops
shapes = [(20, 40, 3, 3), (640, 64, 3, 3), (24, 40, 1, 1), (1, 32, 4, 4)]
params = []
for i, shp in enumerate(shapes):
    w = torch.from_numpy(fill.uniform(shp, 10 + i, -0.1, 0.1)).cuda()
    u = torch.nn.functional.normalize(torch.from_numpy(fill.uniform((shp[0],), 20 + i)), dim=0).cuda()
    v = torch.nn.functional.normalize(torch.from_numpy(fill.uniform((shp[1] * shp[2] * shp[3],), 30 + i)), dim=0).cuda()
    params.append((w, u, v, None))
grp = ops.SpectralNormGroup(params, torch.float16)
packed = grp.step()
for i, ((w, u, v, b), pk) in enumerate(zip(params, packed)):
    ref = ops.pack_conv_weight(w, None, torch.float16, grp.sigma[i:i+1])
    a = ref.w.view(torch.int16).cpu(); c = pk.w.view(torch.int16).cpu()
    bad = (a != c).nonzero().flatten()
    print(shapes[i], a.numel(), "mismatch", bad.numel(), bad[:10].tolist(), [ (int(a[j]), int(c[j])) for j in bad[:5]])
