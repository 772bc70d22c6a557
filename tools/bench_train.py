"""Painter training step (G update + D update, ExtraAdam) throughput at the default config: 640x640, bs 8 per GPU,
default Painter (latent 640, 7 up-samplings) and 3-scale PatchGAN, GAN + feature-matching + VGG losses.
BASELINE metric M1 restricted to the Painter tasks (the Masker has no training path yet).

usage (GPU box): python tools/bench_train.py [--bs 8] [--steps 6] [--dtype bf16] [--no-vgg]
"""
import argparse
import json
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from climategan_amd import fill  # noqa: E402
from climategan_amd.config import default_opts  # noqa: E402
from climategan_amd.trainer import Trainer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bs", type=int, default=8)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--no-vgg", action="store_true")
    ap.add_argument("--tasks", default="p", help="'p' (Painter step) or 'dsmp' (joint Masker + Painter step: domains r, s, rf)")
    args = ap.parse_args()
    dt = torch.float16 if args.dtype == "fp16" else torch.bfloat16
    opts = default_opts()
    opts.tasks = list(args.tasks)
    if args.no_vgg:
        opts.train.lambdas.G.p.vgg = 0
    T = Trainer(opts, device="cuda").setup(inference=False)
    for mod, seed in ((T.G, 0), (T.D, 1)):
        shapes = {k: tuple(v.shape) for k, v in mod.state_dict().items()}
        mod.load_state_dict({k: torch.from_numpy(v) for k, v in fill.fill_state_dict(shapes, seed=seed).items()})
    T.G.set_compute_dtype(dt)
    T.D.set_compute_dtype(dt)
    x = torch.from_numpy(fill.uniform((args.bs, 3, args.size, args.size), 5)).cuda()
    m = torch.from_numpy(fill.rect_mask(args.bs, args.size, args.size, 6)).cuda()
    T.G.painter.set_latent_shape(x.shape, True)
    batch = {"rf": {"data": {"x": x, "m": m}}}
    if "m" in opts.tasks:
        import numpy as np
        hs = args.size // 4
        for i, dom in enumerate(("r", "s")):
            batch[dom] = {"data": {
                "x": torch.from_numpy(fill.uniform((args.bs, 3, args.size, args.size), 20 + i)).cuda(),
                "d": torch.from_numpy(fill.uniform((args.bs, 1, hs, hs), 30 + i, 0.35, 6.95)).cuda(),
                "s": torch.from_numpy((fill.uniform01((args.bs, 1, hs, hs), 40 + i) * 11).astype(np.int64).clip(0, 10)).cuda(),
                "m": torch.from_numpy(fill.rect_mask(args.bs, args.size, args.size, 50 + i)).cuda()}}
    for _ in range(args.warmup):
        T.train_step(batch)
    torch.cuda.synchronize()
    tg = td = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        a = time.perf_counter()
        T.update_G(batch)
        torch.cuda.synchronize()
        b = time.perf_counter()
        T.update_D(batch)
        torch.cuda.synchronize()
        c = time.perf_counter()
        T.global_step += 1
        tg += b - a
        td += c - b
    dtot = time.perf_counter() - t0
    print(json.dumps({
        "workload": "%s train step (update_G + update_D, ExtraAdam), %dx%d bs %d per domain %s, vgg=%s" % (
            "Painter" if args.tasks == "p" else "joint Masker+Painter (domains r, s, rf)", args.size, args.size, args.bs, args.dtype, not args.no_vgg),
        "images_per_s": round(args.bs * args.steps / dtot, 2), "raw_images_per_s": round(args.bs * len(batch) * args.steps / dtot, 2), "ms_per_step": round(dtot / args.steps * 1e3, 1),
        "update_G_ms": round(tg / args.steps * 1e3, 1), "update_D_ms": round(td / args.steps * 1e3, 1),
        "losses": {k: round(float(v), 4) for k, v in T.loss_log.items()},
        "max_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}))


if __name__ == "__main__":
    main()
