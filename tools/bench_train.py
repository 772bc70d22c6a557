"""Training-step throughput (G update + D update, ExtraAdam) at the default config, 640x640:
--tasks p     Painter step, bs 8 per GPU: default Painter (latent 640, 7 up-samplings), 3-scale PatchGAN, GAN +
              feature-matching + VGG losses;
--tasks dsmp  the joint Masker + Painter step of BASELINE metric M1 (domains r, s, rf; --bs samples per domain).
Also: --wgrad-table / --conv-table (per-shape tables of the conv calls of one step), --cprofile (host side).

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


def wgrad_table(T, batch):
    """Record the (shape, flags) of every conv2d_bwd_weight call of one train step, then time each unique one alone."""
    from collections import Counter
    from climategan_amd import ops
    calls = Counter()
    orig = ops.conv2d_bwd_weight

    def rec(x, dy, w_shape, stride=1, pad=0, dilation=1, want_bias=True, dw=None, dbias=None, in_upsample=False,
            pad_mode=ops.PAD_ZERO, **kw):
        calls[(x.n, x.h, x.w, tuple(w_shape), stride, pad, dilation, bool(want_bias), bool(in_upsample), pad_mode,
               str(x.t.dtype))] += 1
        return orig(x, dy, w_shape, stride, pad, dilation, want_bias, dw, dbias, in_upsample, pad_mode, **kw)

    ops.conv2d_bwd_weight = rec
    try:
        T.train_step(batch)
    finally:
        ops.conv2d_bwd_weight = orig
    torch.cuda.synchronize()
    rows = []
    for key, cnt in calls.items():
        n, h, w, ws, stride, pad, dil, wb, ups, pm, dts = key
        dt = torch.bfloat16 if "bfloat16" in dts else torch.float16
        co, ci, kh, kw_ = ws
        x = ops.NHWC(torch.randn(n, h, w, ops.cs8(ci), device="cuda").to(dt), ci)
        hi, wi = (h * 2, w * 2) if ups else (h, w)
        ho = (hi + 2 * pad - dil * (kh - 1) - 1) // stride + 1
        wo = (wi + 2 * pad - dil * (kw_ - 1) - 1) // stride + 1
        dy = ops.NHWC(torch.randn(n, ho, wo, ops.cs8(co), device="cuda").to(dt), co)
        dw = torch.zeros(ws, device="cuda")
        db = torch.zeros(co, device="cuda")
        for _ in range(2):
            orig(x, dy, ws, stride, pad, dil, wb, dw, db if wb else None, ups, pm)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            orig(x, dy, ws, stride, pad, dil, wb, dw, db if wb else None, ups, pm)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        fl = 2.0 * n * ho * wo * co * ci * kh * kw_
        rows.append((ms * cnt, cnt, ms, fl / ms / 1e9, key))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    print("weight-gradient calls of one step: %d calls, %d shapes, %.2f ms in isolation" % (sum(calls.values()), len(rows), tot))
    for t, cnt, ms, tf, key in rows[:40]:
        n, h, w, ws, stride, pad, dil, wb, ups, pm, dts = key
        print("%7.2f ms  x%-3d %7.3f ms %6.1f TF  n%d %dx%d w%s s%d p%d d%d bias%d ups%d pm%d" % (
            t, cnt, ms, tf, n, h, w, ws, stride, pad, dil, wb, ups, pm))


def conv_table(T, batch):
    """Record every ops.conv2d (forward) and ops.conv2d_bwd_data call of one train step, time each unique shape alone."""
    from collections import Counter
    from climategan_amd import ops
    calls = Counter()
    o_fwd, o_dg = ops.conv2d, ops.conv2d_bwd_data

    def rec_fwd(x, pw, stride=1, pad=0, dilation=1, pad_mode=ops.PAD_ZERO, act=ops.ACT_NONE, slope=0.2, residual=None,
                in_upsample=False, residual_upsample=False):
        calls[("fwd", x.n, x.h, x.w, pw.c_in, pw.c_out, pw.kh, pw.kw, stride, pad, dilation, pad_mode, bool(in_upsample),
               residual is not None, str(x.t.dtype))] += 1
        return o_fwd(x, pw, stride, pad, dilation, pad_mode, act, slope, residual, in_upsample, residual_upsample)

    def rec_dg(dy, w, x_shape, stride=1, pad=0, dilation=1, sigma=None, pad_mode=ops.PAD_ZERO):
        if not (pad_mode == ops.PAD_REFLECT and pad > 0):
            co, ci, kh, kw = w.shape
            calls[("dgrad", x_shape[0], x_shape[1], x_shape[2], ci, co, kh, kw, stride, pad, dilation, 0, False, False,
                   str(dy.t.dtype))] += 1
        return o_dg(dy, w, x_shape, stride, pad, dilation, sigma, pad_mode)

    ops.conv2d, ops.conv2d_bwd_data = rec_fwd, rec_dg
    import climategan_amd.norms as norms_mod
    import climategan_amd.autograd as ag_mod
    try:
        T.train_step(batch)
    finally:
        ops.conv2d, ops.conv2d_bwd_data = o_fwd, o_dg
    torch.cuda.synchronize()
    rows = []
    for key, cnt in calls.items():
        kind, n, h, w, ci, co, kh, kw, stride, pad, dil, pm, ups, res, dts = key
        dt = torch.bfloat16 if "bfloat16" in dts else torch.float16
        hi, wi = (h * 2, w * 2) if ups else (h, w)
        ho = (hi + 2 * pad - dil * (kh - 1) - 1) // stride + 1
        wo = (wi + 2 * pad - dil * (kw - 1) - 1) // stride + 1
        wt = torch.randn(co, ci, kh, kw, device="cuda") * 0.02
        if kind == "fwd":
            x = ops.NHWC(torch.randn(n, h, w, ops.cs8(ci), device="cuda").to(dt), ci)
            pw = ops.pack_conv_weight(wt, None, dt)
            r = ops.NHWC(torch.randn(n, ho, wo, ops.cs8(co), device="cuda").to(dt), co) if res else None
            fn = lambda: o_fwd(x, pw, stride, pad, dil, pm, ops.ACT_NONE, 0.2, r, ups, False)
        else:
            dy = ops.NHWC(torch.randn(n, ho, wo, ops.cs8(co), device="cuda").to(dt), co)
            fn = lambda: o_dg(dy, wt, (n, h, w), stride, pad, dil)
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        fl = 2.0 * n * ho * wo * co * ci * kh * kw
        rows.append((ms * cnt, cnt, ms, fl / ms / 1e9, key))
    rows.sort(reverse=True)
    print("conv calls of one step: %d calls, %d shapes, %.2f ms in isolation (dgrad rows include their weight pack)" % (
        sum(calls.values()), len(rows), sum(r[0] for r in rows)))
    for t, cnt, ms, tf, key in rows[:45]:
        kind, n, h, w, ci, co, kh, kw, stride, pad, dil, pm, ups, res, dts = key
        print("%7.2f ms  x%-3d %7.3f ms %6.1f TF  %-5s n%d %dx%d %d->%d k%dx%d s%d p%d d%d pm%d ups%d res%d" % (
            t, cnt, ms, tf, kind, n, h, w, ci, co, kh, kw, stride, pad, dil, pm, ups, res))


def copy_trace(T, batch):
    """Which Python lines issue the device copies / fills of one train step (each is a launch of its own)."""
    import collections
    import traceback
    counts = collections.Counter()
    nbytes = collections.Counter()
    from torch.utils._python_dispatch import TorchDispatchMode

    class Mode(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            name = func.__name__
            if name.split(".")[0] in ("copy_", "clone", "_to_copy", "fill_", "zero_", "zeros", "zeros_like", "contiguous",
                                      "add", "add_", "mul", "mul_", "cat", "_foreach_copy_"):
                st = [f for f in traceback.extract_stack()[:-1] if "climategan_amd" in f.filename or "tools/" in f.filename]
                site = "%s:%d" % (st[-1].filename.split("/")[-1], st[-1].lineno) if st else "autograd-engine"
                counts[(name, site)] += 1
                t = args[0] if args and isinstance(args[0], torch.Tensor) else None
                if t is not None:
                    nbytes[(name, site)] += t.numel() * t.element_size()
            return func(*args, **(kwargs or {}))

    with Mode():
        T.train_step(batch)
    torch.cuda.synchronize()
    for (name, site), n in counts.most_common(40):
        print("%6d  %-28s %-34s %8.1f MB" % (n, name, site, nbytes[(name, site)] / 1e6))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bs", type=int, default=8)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--no-vgg", action="store_true")
    ap.add_argument("--tasks", default="p", help="'p' (Painter step) or 'dsmp' (joint Masker + Painter step: domains r, s, rf)")
    ap.add_argument("--conv-table", action="store_true", help="per-shape table of the forward / data-gradient conv calls of one step")
    ap.add_argument("--only", default="", choices=["", "G", "D"], help="time / profile only update_G or only update_D")
    ap.add_argument("--ddp-single", action="store_true",
                    help="run under a ONE-rank RCCL group with the gradient reducers active (their single-GPU overhead)")
    ap.add_argument("--cprofile", action="store_true", help="host-side cProfile of one train step")
    ap.add_argument("--copy-trace", action="store_true", help="count the aten::copy_ / fill_ calls of one step by Python call site")
    ap.add_argument("--wgrad-table", action="store_true", help="per-shape table of the weight-gradient calls of one step")
    args = ap.parse_args()
    dt = torch.float16 if args.dtype == "fp16" else torch.bfloat16
    if args.ddp_single:
        import os
        import torch.distributed as dist
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CGAN_DDP_SINGLE_RANK_TEST="1")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
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
    if args.wgrad_table:
        return wgrad_table(T, batch)
    if args.conv_table:
        return conv_table(T, batch)
    if args.copy_trace:
        return copy_trace(T, batch)
    if args.cprofile:
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        T.train_step(batch)
        torch.cuda.synchronize()
        pr.disable()
        st = pstats.Stats(pr)
        st.sort_stats("tottime").print_stats(45)
        st.sort_stats("cumtime").print_stats(60)
        return
    # host-side enqueue time of one step (no synchronisation inside) vs the time the GPU needs to drain it
    torch.cuda.synchronize()
    a = time.perf_counter()
    T.update_G(batch)
    T.update_D(batch)
    b = time.perf_counter()
    torch.cuda.synchronize()
    c = time.perf_counter()
    T.global_step += 1
    enqueue_ms, drain_ms = (b - a) * 1e3, (c - b) * 1e3
    tg = td = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        a = time.perf_counter()
        if args.only != "D":
            T.update_G(batch)
        torch.cuda.synchronize()
        b = time.perf_counter()
        if args.only != "G":
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
        "host_enqueue_ms": round(enqueue_ms, 1), "gpu_drain_after_enqueue_ms": round(drain_ms, 1),
        "update_G_ms": round(tg / args.steps * 1e3, 1), "update_D_ms": round(td / args.steps * 1e3, 1),
        "losses": {k: round(float(v), 4) for k, v in T.loss_log.items()},
        "max_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}))


if __name__ == "__main__":
    main()
