"""bench.py -- throughput of the ClimateGAN hot path on MI355X.

Workload (BASELINE.json configs[1], the single-GPU configuration the metric is quoted on):
  Painter-only SPADE generator forward, 640x640, batch 8 per GPU, bf16 activations / fp32 accumulate,
  synthetic mask + context: one step = ``OmniGenerator.paint(m, x)`` (mask the image, run the 9-block SPADE
  Painter incl. the per-forward spectral-norm power iterations, paste) with inputs resident in HBM.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

N > 1: the path shards by image with no data-path collective (inference): every rank runs the same
per-GPU batch (weak scaling); timing is barrier + synchronize on both sides, max over ranks.

The JSON line also carries
  roofline     : the dominant kernel (fused SPADE, MFMA-bound) -- algorithmic FLOPs of the SPADE layers /
                 their summed duration, timed with events on the launch stream inside the timed region
  cpu_baseline : the oracle's CPU restatement (oracle.cpu_ref, torch fp32 on the host cores) on a bounded
                 sample of the same workload (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

BATCH_PER_GPU = 8
H = W = 640
LATENT = 640
N_UP = 7
MFMA_PEAK_TFLOPS = 2500.0  # dense bf16/fp16 MFMA peak, MI355X_MICROARCH.md chip table


def painter_shapes(latent_dim, n_up):
    """State-dict layout of the reference PainterSpadeDecoder (painter.py:36-113), keys -> shapes."""
    def spade(prefix, c):
        return {prefix + ".mlp_shared.0.weight": (128, 3, 3, 3), prefix + ".mlp_shared.0.bias": (128,),
                prefix + ".mlp_gamma.weight": (c, 128, 3, 3), prefix + ".mlp_gamma.bias": (c,),
                prefix + ".mlp_beta.weight": (c, 128, 3, 3), prefix + ".mlp_beta.bias": (c,)}

    def sn(prefix, cin, cout, k, bias=True):
        d = {prefix + ".module.weight_u": (cout,), prefix + ".module.weight_v": (cin * k * k,),
             prefix + ".module.weight_bar": (cout, cin, k, k)}
        if bias:
            d[prefix + ".module.bias"] = (cout,)
        return d

    def blk(prefix, fin, fout):
        fmid = min(fin, fout)
        d = {}
        d.update(sn(prefix + ".conv_0", fin, fmid, 3))
        d.update(sn(prefix + ".conv_1", fmid, fout, 3))
        d.update(spade(prefix + ".norm_0", fin))
        d.update(spade(prefix + ".norm_1", fmid))
        if fin != fout:
            d.update(sn(prefix + ".conv_s", fin, fout, 1, bias=False))
            d.update(spade(prefix + ".norm_s", fin))
        return d

    d = {"fc.weight": (latent_dim, 3, 3, 3), "fc.bias": (latent_dim,)}
    for b in ("head_0", "G_middle_0", "G_middle_1"):
        d.update(blk(b, latent_dim, latent_dim))
    for i in range(n_up - 2):
        d.update(blk("up_spades.%d" % i, latent_dim // 2 ** i, latent_dim // 2 ** (i + 1)))
    fnc = latent_dim // 2 ** (n_up - 2)
    d.update(blk("final_spade", fnc, fnc))
    d["conv_img.weight"] = (3, fnc, 3, 3)
    d["conv_img.bias"] = (3,)
    return d


def spade_layer_table(latent_dim, n_up, h, w):
    """(C, H, W) of every SPADE layer of the Painter -> algorithmic FLOPs (2*MAC of its three 3x3 convs:
    3->128 shared, 128->C gamma, 128->C beta; reference norms.py:163-172), per image."""
    z_h, z_w = h // 2 ** n_up, w // 2 ** n_up
    layers = []
    res = [(z_h, z_w), (2 * z_h, 2 * z_w), (4 * z_h, 4 * z_w)]
    for r in res:
        layers += [(latent_dim, r), (latent_dim, r)]
    hh, ww = res[-1]
    for i in range(n_up - 2):
        hh, ww = 2 * hh, 2 * ww
        fin, fout = latent_dim // 2 ** i, latent_dim // 2 ** (i + 1)
        layers += [(fin, (hh, ww)), (fin, (hh, ww)), (fout, (hh, ww))]
    fnc = latent_dim // 2 ** (n_up - 2)
    layers += [(fnc, (hh, ww)), (fnc, (hh, ww))]
    flops = 0
    for c, (a, b) in layers:
        flops += a * b * 2 * (3 * 9 * 128 + 2 * 128 * 9 * c)
    return layers, flops


def recorded_spade_traffic():
    """HBM bytes per fused-SPADE launch from the committed PMC summary (tools/summarize_pmc.py); None if absent.
    The PMC passes cannot run inside the timed bench (rocprofv3 wraps the process), so the number is recorded."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_spade_hbm_pmc.csv")))
    if not files:
        return None
    m = re.search(r"= ([0-9.]+) MB per launch", open(files[-1]).read())
    return int(float(m.group(1)) * 1e6) if m else None


def build(device, dtype):
    from climategan_amd import fill
    from climategan_amd.config import default_opts
    from climategan_amd.generator import create_generator

    opts = default_opts()
    opts.tasks = ["p"]
    opts.gen.p.latent_dim = LATENT
    opts.gen.p.spade_n_up = N_UP
    G = create_generator(opts, device=device)
    sd = {k: torch.from_numpy(v) for k, v in fill.fill_state_dict(painter_shapes(LATENT, N_UP), seed=0).items()}
    G.painter.load_state_dict(sd)
    G.set_compute_dtype(dtype)
    G.painter.set_latent_shape((BATCH_PER_GPU, 3, H, W), True)
    return G, sd


def synthetic_batch(rank, device):
    from climategan_amd import fill

    x = torch.from_numpy(fill.uniform((BATCH_PER_GPU, 3, H, W), seed=1000 + rank)).to(device)
    m = torch.from_numpy(fill.rect_mask(BATCH_PER_GPU, H, W, seed=2000 + rank)).to(device)
    return x, m


class SpadeTimer:
    """Event pairs around every fused-SPADE launch (recorded on the stream the kernel is launched on)."""

    def __init__(self):
        self.pairs = []
        self.enabled = False

    def install(self):
        from climategan_amd import ops

        orig = ops.spade_fused
        timer = self

        def timed(*a, **k):
            if not timer.enabled:
                return orig(*a, **k)
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            out = orig(*a, **k)
            e1.record()
            timer.pairs.append((e0, e1))
            return out

        ops.spade_fused = timed
        import climategan_amd.norms as norms_mod

        norms_mod.ops.spade_fused = timed

    def total_ms(self):
        return sum(a.elapsed_time(b) for a, b in self.pairs)


def cpu_baseline(sd):
    """Oracle (CPU restatement of the reference path, torch fp32) on a bounded sample: bs=1, 640x640."""
    from climategan_amd import fill
    from oracle import cpu_ref

    x = torch.from_numpy(fill.uniform((1, 3, H, W), seed=1))
    m = torch.from_numpy(fill.rect_mask(1, H, W, seed=2))
    sdc = {k: v.clone() for k, v in sd.items()}
    z = H // 2 ** N_UP
    ncpu = os.cpu_count() or 1
    best = None
    # torch's intra-op threading stops scaling on this workload well before a 2-socket host's thread count
    # (256 threads: 77 s/image; 8 threads: 1.6 s/image on the same box), so report the best of a short sweep
    for threads in sorted({min(8, ncpu), min(16, ncpu), min(32, ncpu)}):
        torch.set_num_threads(threads)
        with torch.no_grad():
            cpu_ref.paint(sdc, m, x, z, z)  # warm-up
            runs = 2
            t0 = time.perf_counter()
            for _ in range(runs):
                cpu_ref.paint(sdc, m, x, z, z)
            dt = (time.perf_counter() - t0) / runs
        if best is None or dt < best[0]:
            best = (dt, threads)
    dt, threads = best
    return {"value": round(1.0 / dt, 4), "unit": "images/s", "cores": threads, "kind": "port",
            "sample": "oracle.cpu_ref.paint (torch fp32 CPU restatement of generator.py:279-297), bs=1 640x640, "
                      "best of {8,16,32} torch threads (2 runs after 1 warm-up each) on a %d-thread host" % ncpu}


def timed_steps(step, steps, warmup, barrier):
    """W untimed warm-up steps, then exactly K timed steps bracketed by barrier() on both sides."""
    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    return time.perf_counter() - t0


def max_over_ranks(elapsed, dist, device):
    """Whole-job time = the slowest rank's (the only collective on this path)."""
    if dist is None:
        return elapsed
    el = torch.tensor([elapsed], device=device, dtype=torch.float64)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    return el.item()


def result_line(world, steps, warmup, elapsed, dtype_name):
    return {
        "metric": "640x640 images/sec, Painter (SPADE generator) forward, batch 8 per GPU",
        "value": round(world * BATCH_PER_GPU * steps / elapsed, 3),
        "unit": "images/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": round(elapsed / steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": dtype_name,
        "data": "synthetic (counter-hash fill: U(-1,1) images, 3-rectangle masks ~35%, untrained weights "
                "with torch-default conv init ranges)",
        "config": {"workload": "BASELINE configs[1]: Painter-only SPADE generator fwd 640x640 bs=8 "
                               "(OmniGenerator.paint incl. mask, spectral-norm power iterations, paste)",
                   "batch_per_gpu": BATCH_PER_GPU, "global_batch": BATCH_PER_GPU * world,
                   "latent_dim": LATENT, "spade_n_up": N_UP, "parallelism": "independent replicas, image-sharded"},
    }


TRAIN_BS = 4   # per domain per GPU: BASELINE configs[3] is global batch 32 over 8 GPUs


TRAIN_BLOCK_TIMEOUT_S = 420     # watchdog of the supplementary block (see main)


def train_block(train_steps, rank, world, device, dtype, dist, barrier):
    """Supplementary measurement (not `value`): BASELINE's "G+D step" -- the full joint Masker + Painter training
    iteration of the default config (Trainer.train_step = update_G over the real, sim and flooded domains: ResNet-101
    encoder with batch-statistics BatchNorm, depth / seg / mask decoders and their 10 loss terms, ADVENT
    discriminators, Painter with GAN + feature-matching + VGG losses; then update_D; ExtraAdam extrapolate / step),
    640x640, 4 samples per domain per GPU, bf16, data-parallel over the ranks with the bucketed RCCL all-reduce of
    climategan_amd/parallel.py.  Same barrier / synchronize / max-over-ranks timing as the main line.  `images_per_s`
    counts one per-domain sample slot per image; `raw_images_per_s` counts all three domains."""
    import numpy as np

    from climategan_amd import fill
    from climategan_amd.config import default_opts
    from climategan_amd.trainer import Trainer

    opts = default_opts()
    opts.tasks = ["d", "s", "m", "p"]
    opts.gen.p.latent_dim = LATENT
    opts.gen.p.spade_n_up = N_UP
    T = Trainer(opts, device=device).setup(inference=False)
    for mod, seed in ((T.G, 0), (T.D, 1)):
        shapes = {k: tuple(v.shape) for k, v in mod.state_dict().items()}
        mod.load_state_dict({k: torch.from_numpy(v) for k, v in fill.fill_state_dict(shapes, seed=seed, gain=1.6).items()})
    T.G.set_compute_dtype(dtype)
    T.D.set_compute_dtype(dtype)
    bs, hs = TRAIN_BS, H // 4

    def dev(a):
        return torch.from_numpy(a).to(device)

    batch = {"rf": {"data": {"x": dev(fill.uniform((bs, 3, H, W), 100 + rank)),
                             "m": dev(fill.rect_mask(bs, H, W, 200 + rank))}}}
    for i, dom in enumerate(("r", "s")):
        sd = 300 + 10 * i + 1000 * rank
        batch[dom] = {"data": {"x": dev(fill.uniform((bs, 3, H, W), sd)),
                               "d": dev(fill.uniform((bs, 1, hs, hs), sd + 1, 0.35, 6.95)),
                               "s": dev((fill.uniform01((bs, 1, hs, hs), sd + 2) * 11).astype(np.int64).clip(0, 10)),
                               "m": dev(fill.rect_mask(bs, H, W, sd + 3))}}
    T.G.painter.set_latent_shape((bs, 3, H, W), True)
    elapsed = timed_steps(lambda: T.train_step(batch), train_steps, 1, barrier)
    elapsed = max_over_ranks(elapsed, dist, device)
    losses = {k: round(float(v), 4) for k, v in T.loss_log.items()}
    return {"workload": "joint Masker+Painter G+D train step (Trainer.train_step; domains r, s, rf; all default loss "
                        "terms, VGG with random-init weights), 640x640, %d samples per domain per GPU, data-parallel "
                        "bucketed gradient all-reduce" % bs,
            "images_per_s": round(world * bs * train_steps / elapsed, 2),
            "raw_images_per_s": round(world * bs * 3 * train_steps / elapsed, 2),
            "ms_per_step": round(elapsed / train_steps * 1e3, 1), "steps": train_steps, "warmup": 1,
            "global_batch_per_domain": world * bs, "losses_last_step": losses}


def infer_block(steps, rank, world, device, dist, barrier):
    """Supplementary: BASELINE metric M2 -- the apply_events inference loop (Trainer.infer_all: Masker + flood painter
    + wildfire + smog, uint8 outputs copied to the host), 640x640, 16 images per GPU, fp16; images/s over all ranks."""
    from climategan_amd import fill
    from climategan_amd.config import default_opts
    from climategan_amd.trainer import Trainer

    bs = 16
    opts = default_opts()
    opts.tasks = ["d", "s", "m", "p"]
    T = Trainer(opts, device=device).setup(inference=True)
    shapes = {k: tuple(v.shape) for k, v in T.G.state_dict().items()}
    T.G.load_state_dict({k: torch.from_numpy(v) for k, v in fill.fill_state_dict(shapes, seed=0, gain=1.6).items()})
    T.G.set_compute_dtype(torch.float16)
    x = torch.from_numpy(fill.uniform((bs, 3, H, W), 3000 + rank)).to(device)
    out = {}

    def step():
        out.update(T.infer_all(x, numpy=True, bin_value=0.5, half=True))

    for _ in range(2):
        step()
    elapsed = max_over_ranks(timed_steps(step, steps, 0, barrier), dist, device)
    assert set(out) >= {"flood", "wildfire", "smog"} and out["flood"].shape == (bs, H, W, 3)
    return {"workload": "apply_events inference (Trainer.infer_all: flood + wildfire + smog, uint8 results on the host), "
                        "640x640, 16 images per GPU, fp16",
            "images_per_s": round(world * bs * steps / elapsed, 2), "ms_per_batch": round(elapsed / steps * 1e3, 2),
            "steps": steps, "warmup": 2}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--infer-steps", type=int, default=3,
                    help="timed batches of the supplementary apply_events inference measurement (0 = skip)")
    ap.add_argument("--train-steps", type=int, default=3,
                    help="timed steps of the supplementary joint G+D training-step measurement (0 = skip)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                 % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)

    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    G, sd = build(device, dtype)
    x, m = synthetic_batch(rank, device)
    timer = SpadeTimer()
    timer.install()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    out = {}

    def step():
        out["y"] = G.paint(m, x)

    def timed_step():
        timer.enabled = True
        step()

    with torch.no_grad():
        for _ in range(args.warmup):
            step()
        elapsed = timed_steps(timed_step, args.steps, 0, barrier)
        timer.enabled = False
    y = out["y"]
    assert y.shape == (BATCH_PER_GPU, 3, H, W) and torch.isfinite(y).all()
    elapsed = max_over_ranks(elapsed, dist, device)

    # ---- the result line is complete before the supplementary block starts, so that nothing in that block (first
    # RCCL use of the training path on a multi-GPU node, a hung collective, an exception on one rank) can take the
    # headline measurement down with it: a watchdog prints the line and ends the process if the block overruns.
    res = None
    if rank == 0:
        layers, flops_img = spade_layer_table(LATENT, N_UP, H, W)
        spade_ms = timer.total_ms()
        n_launch = len(timer.pairs)
        flops_step = flops_img * BATCH_PER_GPU
        achieved = flops_step * args.steps / (spade_ms * 1e-3) / 1e12 if spade_ms > 0 else 0.0
        res = result_line(world, args.steps, args.warmup, elapsed, args.dtype)
        traffic = recorded_spade_traffic()
        # minimum HBM bytes of the SPADE launches: x read + 4-channel cond read + output write (16-bit, cs8 padding)
        alg_bytes = sum(BATCH_PER_GPU * a * b * 2 * (2 * ((c + 7) // 8 * 8) + 4) for c, (a, b) in layers)
        res.update({
            "roofline": {
                "bound": "mfma",
                "kernel": "spade_fused_kernel (23 launches/step: all SPADE layers of the Painter)",
                "achieved": round(achieved, 2),
                "peak": MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": round(achieved / MFMA_PEAK_TFLOPS, 4),
                "traffic": traffic,
                "traffic_unit": "HBM bytes per launch (mean over the 23 launches), from profiles/*_spade_hbm_pmc.csv: "
                                "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, FETCH_SIZE "
                                "doubled per the gfx950 correction",
                "algorithmic_bytes_per_launch": alg_bytes // max(len(layers), 1),
                "algorithmic_flops_per_step": flops_step,
                "launches_per_step": n_launch // max(args.steps, 1),
                "avg_launch_ms": round(spade_ms / max(n_launch, 1), 4),
                "share_of_step": round(spade_ms / (elapsed * 1e3), 3),
            },
        })
        res["cpu_baseline"] = cpu_baseline(sd) if (world == 1 and not args.no_cpu_baseline) else None

    import threading
    emitted = threading.Lock()
    finished = threading.Event()

    def emit(train):
        if rank == 0 and emitted.acquire(blocking=False):
            res["train_step"] = train
            print(json.dumps(res), flush=True)

    def watchdog():
        if not finished.wait(TRAIN_BLOCK_TIMEOUT_S):
            emit({"error": "supplementary train block did not finish within %d s" % TRAIN_BLOCK_TIMEOUT_S})
            os._exit(0)

    train = infer = None
    if args.train_steps > 0 or args.infer_steps > 0:
        threading.Thread(target=watchdog, daemon=True).start()
    if args.infer_steps > 0:
        try:
            infer = infer_block(args.infer_steps, rank, world, device, dist, barrier)
        except Exception as e:
            infer = {"error": "%s: %s" % (type(e).__name__, e)}
        torch.cuda.empty_cache()
    if rank == 0:
        res["apply_events"] = infer
    if args.train_steps > 0:
        try:
            train = train_block(args.train_steps, rank, world, device, dtype, dist, barrier)
        except Exception as e:  # the main line must survive a failure of the supplementary block
            train = {"error": "%s: %s" % (type(e).__name__, e)}
    emit(train)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    finished.set()


if __name__ == "__main__":
    main()
