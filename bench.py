"""bench.py -- throughput of the ClimateGAN hot path on MI355X.

Headline (BASELINE.json ``metric``: "640x640 images/sec (G+D step) at 1/2/4/8 MI355X"; BASELINE configs[3] at its GLOBAL
batch of 32 per domain -- SURVEY 8d M1: "at 1/2/4/8 GPUs, global bs 32 (4/GPU at 8 GPUs)", reference trainer.py:633,935-939):
  one step = ``Trainer.train_step`` = ``update_G`` + ``update_D`` of the reference's default task set [d, s, m, p]
  (reference trainer.py:989-1032) on one multi-domain batch -- domains r and s through the Masker (ResNet-101 encoder
  with batch-statistics BatchNorm, depth / segmentation / mask decoders, 10 loss terms, ADVENT discriminators), domain rf
  through the Painter (GAN + feature-matching + VGG losses, 3-scale PatchGAN) -- 640x640, **32 samples per domain per
  step over the whole job: 32 / N per rank** (N = 1: all 32 on the one GPU, 137 GB of the 288; N = 8: 4 per GPU), bf16
  activations / fp32 accumulation and parameters, ExtraAdam extrapolation / step.  ``value`` = per-domain sample slots per
  second over all ranks (a step consumes ``bs`` samples from each of the three domains; the raw-image
  figure is 3x and reported next to it); ``scaling`` is "strong": the job's work per step does not change with N.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

N > 1: data parallel, one process per GPU, identical replicas, rank r takes samples [r * 32/N, (r+1) * 32/N) of the global
batch (``climategan_amd.parallel.shard_range``: strong scaling), the G and the D gradients
averaged by the bucketed RCCL all-reduce of ``climategan_amd/parallel.py`` launched from gradient hooks during the
backward; timing is barrier + synchronize on both sides of the K timed steps, max over ranks.

The JSON line also carries
  roofline      the dominant kernel family of the step -- the wide-layer implicit-GEMM convolution kernel
                (``conv_gemm_kernel``: forward and data-gradient convs of the ResNet / ASPP / decoders / VGG / PatchGAN layers
                with >= 64 output channels; MFMA-bound): every launch inside the timed region is bracketed by events on
                the launch stream, achieved = algorithmic FLOPs (2 * output pixels * c_out * c_in * taps, from the call's
                own descriptor) / summed duration;
  cpu_baseline  the oracle's CPU restatement of the same step (``oracle.cpu_ref.joint_train_step``, torch fp32 autograd)
                on a bounded sample -- one step at 1 sample per domain, 640x640 -- on the host cores (rank 0, N = 1);
  sub_blocks    the other BASELINE configurations on the same box, each with >= 20 timed steps: configs[1] Painter
                forward bs 8 bf16 (with the fused-SPADE kernel's MFMA roofline, the kernel north_star names),
                configs[2] Masker train step bs 8, configs[4] apply_events inference bs 16 fp16; and ``per_gpu_slice``,
                the headline step at 4 per domain = one rank's share of an 8-GPU job (the rounds 1-5 headline).
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

# multi-process GPU work on these hosts needs dmabuf IPC (the image exports it; kept for launches from a bare environment)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

BATCH_PER_GPU = 8      # configs[1]: Painter forward
GLOBAL_BS = 32         # configs[3]: samples per domain per step over the WHOLE job (32 / N per rank)
SLICE_BS = 4           # one rank's share at N = 8 (sub_blocks.per_gpu_slice; the kernel-tuning proxy for the 8-GPU point)
MASKER_BS = 8          # configs[2]
INFER_BS = 16          # configs[4]
H = W = 640
LATENT = 640
N_UP = 7
MFMA_PEAK_TFLOPS = 2500.0  # dense bf16/fp16 MFMA peak, MI355X_MICROARCH.md chip table
HBM_PEAK_BYTES = 8.0e12    # HBM3E peak, same table
EVENT_EVERY = 10           # timed steps between two steps whose launches are bracketed by events (see main)
WELL_CONDITIONED = dict(gain=1.0, res_gamma=0.05)   # the fill of the 640x640 parity fixtures (tests/test_gpu_configs_640.py)


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


def recorded_traffic(pattern):
    """HBM bytes per launch from the newest committed PMC summary matching ``profiles/<pattern>`` (tools/summarize_pmc.py);
    None if absent.  The PMC passes cannot run inside the timed bench (rocprofv3 wraps the process): recorded."""
    import glob
    import re
    files = sorted(glob.glob(str(ROOT / "profiles" / pattern)))
    if not files:
        return None
    m = re.search(r"= ([0-9.]+) MB per launch", open(files[-1]).read())
    return int(float(m.group(1)) * 1e6) if m else None


def live_traffic(launches_per_step, global_batch=32, timeout_s=300):
    """HBM bytes per launch of the wide-layer GEMM family MEASURED NOW (round 5): two child runs of this script's headline step
    under ``rocprofv3 --kernel-trace --pmc FETCH_SIZE`` / ``WRITE_SIZE`` (separate passes: the two counters do not fit one),
    the family's dispatches of the last step summed as tools/summarize_pmc_kernel.py does (FETCH_SIZE doubled per the gfx950
    correction of MI355X_MICROARCH.md).  None when rocprofv3 is not on the box, a pass fails or takes too long, or
    CGAN_BENCH_NO_LIVE_PMC=1 -- the caller then reports the newest committed summary and says so."""
    import csv
    import shutil
    import subprocess
    import tempfile
    if os.environ.get("CGAN_BENCH_NO_LIVE_PMC") == "1" or shutil.which("rocprofv3") is None:
        return None
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCPROFILER")) for k in os.environ):      # this run is itself being profiled
        return None
    fam = ("conv_gemm", "conv1x1_xres", "conv1x1_allc")
    tot = {}
    tmp = tempfile.mkdtemp(prefix="cgan_pmc_")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, ctr)
            cmd = ["rocprofv3", "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", out, "-o", "live", "--",
                   sys.executable, os.path.abspath(__file__), "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--sub-steps", "0",
                   "--no-launch-events", "--mfma-table-steps", "0", "--global-batch", str(global_batch)]
            env = dict(os.environ, CGAN_BENCH_NO_LIVE_PMC="1", TMPDIR=os.environ.get("TMPDIR", "/tmp"))
            r = subprocess.run(cmd, cwd=tmp, env=env, capture_output=True, timeout=timeout_s)
            f = None
            for root, _d, files in os.walk(out):
                for name in files:
                    if name.endswith("counter_collection.csv"):
                        f = os.path.join(root, name)
            if r.returncode != 0 or f is None:
                return None
            rows = [x for x in csv.DictReader(open(f)) if x["Counter_Name"] == ctr and any(k in x["Kernel_Name"] for k in fam)]
            rows.sort(key=lambda x: int(x["Dispatch_Id"]))
            rows = rows[-launches_per_step:]
            if len(rows) != launches_per_step:
                return None
            tot[ctr] = sum(float(x["Counter_Value"]) for x in rows) * 1024.0
        return int((2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) / launches_per_step)
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def recorded_mfma_util():
    """Matrix-pipe busy fraction and wave-life split per kernel family from the newest committed SQ-counter summary
    (profiles/*_mfma_util.csv: tools/gpu_pmc_step_sq.sh + tools/mfma_util.py, rocprofv3 --pmc passes of this command's
    headline step); None if absent.  Counter passes wrap the process, so they are recorded, not taken inside the bench."""
    import glob
    files = sorted(glob.glob(str(ROOT / "profiles" / "*_mfma_util.csv")))
    if not files:
        return None
    out, cols = {"source": "profiles/" + os.path.basename(files[-1])}, None
    for line in open(files[-1]):
        if line.startswith("#"):
            continue
        f = line.rstrip("\n").split(",")
        if f[0] == "family":
            cols = f
            continue
        if cols and f[0] in ("gemm 1x1", "gemm kxk", "gemm", "general", "lds3x3", "wgrad", "spade"):
            row = dict(zip(cols, f))
            out[f[0]] = {k: (float(row[k]) if row.get(k) else None)
                         for k in ("mfma_busy", "waves_per_simd", "active", "parked", "issue_stall", "lds_issue", "lds_conflict")}
    return out


# ------------------------------------------------------------------------------------------------ timing helpers
# Sub-blocks with millisecond steps (Painter forward: 6 ms, apply_events: 34 ms) start after seconds of host-only work (model
# construction, weight fill) during which the GPU has dropped to its idle clocks; five warm-up steps = 30 ms do not bring them
# back on every box (one evidence box of round 6 read the Painter block 20 % low and the SAME kernels at their usual time under
# rocprofv3 minutes later).  Their warm-up therefore also lasts at least SUB_WARM_SECONDS[0] of device time; the headline's
# W = 5 steps are 2.8 s by themselves.  (--only runs: 0 unless --sub-warm-seconds is given, so profiles keep their launch counts.)
SUB_WARM_SECONDS = [0.0]


def timed_steps(step, steps, warmup, barrier, min_warm_s=0.0):
    """W untimed warm-up steps (and as many more as ``min_warm_s`` seconds take), then exactly K timed steps bracketed by
    barrier() on both sides."""
    t_w = time.perf_counter()
    for _ in range(warmup):
        step()
    if min_warm_s > 0:
        torch.cuda.synchronize()
        while time.perf_counter() - t_w < min_warm_s:
            for _ in range(4):
                step()
            torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    return time.perf_counter() - t0


def max_over_ranks(elapsed, dist, device):
    """Whole-job time = the slowest rank's."""
    if dist is None:
        return elapsed
    el = torch.tensor([elapsed], device=device, dtype=torch.float64)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    return el.item()


def result_line(world, steps, warmup, elapsed, dtype_name, per_rank, gbs=None):
    gbs = GLOBAL_BS if gbs is None else gbs
    return {
        "metric": "640x640 images/sec (G+D step): per-domain sample slots per second of the joint Masker+Painter "
                  "training step (update_G + update_D), global batch %d per domain" % gbs,
        "value": round(gbs * steps / elapsed, 3),
        "unit": "images/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": round(elapsed / steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": dtype_name,
        "data": "synthetic (counter-hash fill: U(-1,1) images, 3-rectangle masks ~35%, uniform depth / class targets; "
                "untrained weights from the portable fill, VGG-19 random He-scale weights)",
        "raw_images_per_s": round(3 * gbs * steps / elapsed, 3),
        "config": {"workload": "BASELINE configs[3]: full Masker+Painter joint G/D train step (Trainer.train_step; tasks "
                               "d,s,m,p; domains r,s,rf; all default loss terms incl. VGG; ExtraAdam), 640x640, GLOBAL batch "
                               "%d samples per domain per step (%d raw images), %d per domain per GPU on %d GPU%s"
                               % (gbs, 3 * gbs, per_rank, world, "s" if world > 1 else ""),
                   "global_batch": gbs, "batch_per_domain_per_gpu": per_rank,
                   "global_raw_images_per_step": 3 * gbs, "latent_dim": LATENT, "spade_n_up": N_UP,
                   "parallelism": "dp%d: replicas, each rank takes %d of the %d samples per domain, bucketed RCCL all-reduce "
                                  "of G and D gradients from backward hooks" % (world, per_rank, gbs)
                                  if world > 1 else "single GPU"},
    }


class LaunchTimer:
    """Event pairs (recorded on torch's current stream = the stream the library launches on) around selected launches."""

    def __init__(self):
        self.pairs = []          # (event0, event1, flops, algorithmic bytes)
        self.enabled = False
        self.armed = False

    def bracket(self, fn, flops, nbytes=0, tag=None):
        if not self.enabled:
            return fn()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        self.pairs.append((e0, e1, flops, nbytes, tag))
        return out

    def table(self, steps):
        """Per distinct launch shape: launches per step, mean duration, achieved TFLOP/s and algorithmic GB/s, and the
        time the better of the two rooflines would allow (MFMA_PEAK, HBM_PEAK): where the family loses its time."""
        agg = {}
        for e0, e1, fl, nb, tag in self.pairs:
            a = agg.setdefault(tag, [0, 0.0, fl, nb])
            a[0] += 1
            a[1] += e0.elapsed_time(e1)
        rows = []
        for tag, (n, ms, fl, nb) in agg.items():
            us = ms / n * 1e3
            bound_us = max(fl / (MFMA_PEAK_TFLOPS * 1e12), nb / HBM_PEAK_BYTES) * 1e6
            rows.append((ms / steps, n / steps, us, fl / (us * 1e-6) / 1e12, nb / (us * 1e-6) / 1e9, bound_us, tag))
        rows.sort(reverse=True)
        lines = ["%8s %6s %9s %8s %8s %9s  %s" % ("ms/step", "n/step", "us", "TFLOP/s", "GB/s", "bound us", "shape")]
        for r in rows:
            lines.append("%8.3f %6.1f %9.1f %8.1f %8.1f %9.1f  %s" % r)
        return "\n".join(lines)

    def classes(self):
        """The bracketed launches split by what bounds them: 1x1 layers (bottleneck reduce / expand: short K, near the HBM
        roofline) and k x k layers (long K: MFMA-bound), each against BOTH rooflines."""
        out = {}
        for name, sel in (("1x1", lambda t: " k1 " in t), ("kxk", lambda t: " k1 " not in t)):
            ps = [p for p in self.pairs if sel(p[4])]
            ms = sum(p[0].elapsed_time(p[1]) for p in ps)
            if not ps or ms <= 0:
                continue
            fl, nb = sum(p[2] for p in ps), sum(p[3] for p in ps)
            out[name] = {"launches": len(ps), "ms": round(ms, 3), "tflops": round(fl / ms / 1e9, 1),
                         "frac_mfma": round(fl / ms / 1e9 / MFMA_PEAK_TFLOPS, 4),
                         "algorithmic_GBps": round(nb / ms / 1e6, 1), "frac_hbm": round(nb / ms / 1e6 / (HBM_PEAK_BYTES / 1e9), 4),
                         "bound": "hbm" if nb / HBM_PEAK_BYTES > fl / (MFMA_PEAK_TFLOPS * 1e12) else "mfma"}
        return out

    def total_ms(self):
        return sum(p[0].elapsed_time(p[1]) for p in self.pairs)

    def total_flops(self):
        return sum(p[2] for p in self.pairs)

    def total_bytes(self):
        return sum(p[3] for p in self.pairs)


def install_conv_gemm_timer(timer):
    """Bracket every launch of the wide-layer implicit-GEMM kernel (forward and data-gradient entry points): the C ABI says
    which kernel a descriptor selects (``cgan_conv2d_kernel_kind``), and the algorithmic FLOPs come from the same
    descriptor: 2 * n * h_out * w_out * c_out * c_in * kh * kw (forward) / 2 * n * h_in * w_in * c_in * c_out * kh * kw
    (data gradient), logical channels."""
    from climategan_amd import _lib

    lib = _lib.load()
    fwd, bwd, kind = lib.cgan_conv2d_nhwc_fwd, lib.cgan_conv2d_nhwc_bwd_data, lib.cgan_conv2d_kernel_kind_on
    GEMM = 2

    def cs8(c):
        return (c + 7) // 8 * 8

    def alg_bytes(d):
        # minimum HBM traffic of the conv: input + output (+ residual) once, 16-bit, stored channel counts; + the weights
        act = d.n * (d.h_in * d.w_in * cs8(d.c_in) + d.h_out * d.w_out * cs8(d.c_out) * (2 if d.has_residual else 1))
        return 2 * (act + d.c_out * d.c_in * d.kh * d.kw)

    def tag(d, what):
        return "%-8s n%d %dx%d c%d -> %dx%d c%d k%d s%d d%d%s" % (what, d.n, d.h_in, d.w_in, d.c_in, d.h_out, d.w_out, d.c_out,
                                                            d.kh, d.stride, d.dilation, " +res" if d.has_residual else "")

    def timed_fwd(x, w, b, r, y, dref, stream):
        if timer.enabled and kind(dref, 0, stream) == GEMM:
            d = dref._obj
            return timer.bracket(lambda: fwd(x, w, b, r, y, dref, stream),
                                 2.0 * d.n * d.h_out * d.w_out * d.c_out * d.c_in * d.kh * d.kw, alg_bytes(d), tag(d, "fwd"))
        return fwd(x, w, b, r, y, dref, stream)

    def timed_bwd(dy, w, dx, dref, stream):
        if timer.enabled and kind(dref, 1, stream) == GEMM:
            d = dref._obj
            return timer.bracket(lambda: bwd(dy, w, dx, dref, stream),
                                 2.0 * d.n * d.h_in * d.w_in * d.c_in * d.c_out * d.kh * d.kw, alg_bytes(d), tag(d, "bwd_data"))
        return bwd(dy, w, dx, dref, stream)

    bwd_add = lib.cgan_conv2d_nhwc_bwd_data_add

    def timed_bwd_add(dy, w, dx_add, dx, dref, stream):        # data gradient + the other contribution of the same tensor
        if timer.enabled and kind(dref, 1, stream) == GEMM:
            d = dref._obj
            extra = 2 * d.n * d.h_in * d.w_in * cs8(d.c_in)     # the added tensor is read once more
            return timer.bracket(lambda: bwd_add(dy, w, dx_add, dx, dref, stream),
                                 2.0 * d.n * d.h_in * d.w_in * d.c_in * d.c_out * d.kh * d.kw, alg_bytes(d) + extra,
                                 tag(d, "bwd_data+"))
        return bwd_add(dy, w, dx_add, dx, dref, stream)

    bwd_relu = lib.cgan_conv2d_nhwc_bwd_data_relu

    def timed_bwd_relu(dy, w, relu_out, dx, dref, stream):      # data gradient masked by the ReLU output it flows back into
        if timer.enabled and kind(dref, 1, stream) == GEMM:
            d = dref._obj
            extra = 2 * d.n * d.h_in * d.w_in * cs8(d.c_in)     # the activation's output is read once more
            return timer.bracket(lambda: bwd_relu(dy, w, relu_out, dx, dref, stream),
                                 2.0 * d.n * d.h_in * d.w_in * d.c_in * d.c_out * d.kh * d.kw, alg_bytes(d) + extra,
                                 tag(d, "bwd_data*"))
        return bwd_relu(dy, w, relu_out, dx, dref, stream)

    bwd_add_relu = lib.cgan_conv2d_nhwc_bwd_data_add_relu

    def timed_bwd_add_relu(dy, w, dx_add, relu_out, dx, dref, stream):   # both: the added tensor AND the activation's output
        if timer.enabled and kind(dref, 1, stream) == GEMM:
            d = dref._obj
            extra = 4 * d.n * d.h_in * d.w_in * cs8(d.c_in)
            return timer.bracket(lambda: bwd_add_relu(dy, w, dx_add, relu_out, dx, dref, stream),
                                 2.0 * d.n * d.h_in * d.w_in * d.c_in * d.c_out * d.kh * d.kw, alg_bytes(d) + extra,
                                 tag(d, "bwd_data+*"))
        return bwd_add_relu(dy, w, dx_add, relu_out, dx, dref, stream)

    fwd_stats = lib.cgan_conv2d_nhwc_fwd_stats

    def timed_fwd_stats(x, w, b, y, partial, nbytes, dref, stream):     # forward + BatchNorm statistics epilogue: GEMM kernel only
        if timer.enabled:
            d = dref._obj
            return timer.bracket(lambda: fwd_stats(x, w, b, y, partial, nbytes, dref, stream),
                                 2.0 * d.n * d.h_out * d.w_out * d.c_out * d.c_in * d.kh * d.kw, alg_bytes(d), tag(d, "fwd+st"))
        return fwd_stats(x, w, b, y, partial, nbytes, dref, stream)

    lib.cgan_conv2d_nhwc_fwd, lib.cgan_conv2d_nhwc_bwd_data, lib.cgan_conv2d_nhwc_bwd_data_add = timed_fwd, timed_bwd, timed_bwd_add
    lib.cgan_conv2d_nhwc_fwd_stats = timed_fwd_stats
    lib.cgan_conv2d_nhwc_bwd_data_relu = timed_bwd_relu
    lib.cgan_conv2d_nhwc_bwd_data_add_relu = timed_bwd_add_relu

    def uninstall():
        lib.cgan_conv2d_nhwc_fwd, lib.cgan_conv2d_nhwc_bwd_data, lib.cgan_conv2d_nhwc_bwd_data_add = fwd, bwd, bwd_add
        lib.cgan_conv2d_nhwc_fwd_stats = fwd_stats
        lib.cgan_conv2d_nhwc_bwd_data_relu = bwd_relu
        lib.cgan_conv2d_nhwc_bwd_data_add_relu = bwd_add_relu
    return uninstall


def install_all_mfma_timer(timer):
    """Bracket EVERY launch of an MFMA kernel of the train step, whatever kernel its descriptor selects: forward and
    data-gradient convolutions (wide-layer GEMM family / 3x3 LDS-tiled / general gather kernel: ``cgan_conv2d_kernel_kind``),
    weight gradients (all ``conv_wgrad*`` kernels + their split reduction and bias sums: one C call) and the fused SPADE
    forward -- so that 100 % of the MFMA-kernel time of a step has a roofline row (``roofline_all_mfma``, ``--conv-table``).
    Used on extra single-stream steps AFTER the timed region: the headline's timing is untouched."""
    from climategan_amd import _lib

    lib = _lib.load()
    names = ("cgan_conv2d_nhwc_fwd", "cgan_conv2d_nhwc_bwd_data", "cgan_conv2d_nhwc_bwd_data_add", "cgan_conv2d_nhwc_fwd_stats",
             "cgan_conv2d_nhwc_bwd_weight", "cgan_spade_fused_fwd", "cgan_spade_fused_fwd_train",
             "cgan_conv2d_nhwc_bwd_data_relu", "cgan_spade_hidden_bwd", "cgan_conv2d_nhwc_bwd_data_add_relu")
    orig = {n: getattr(lib, n) for n in names}
    kind = lib.cgan_conv2d_kernel_kind_on
    KIND = {0: "general", 1: "lds3x3", 2: "gemm"}

    def cs8(c):
        return (c + 7) // 8 * 8

    def conv_bytes(d, extra=0):
        act = d.n * (d.h_in * d.w_in * cs8(d.c_in) + d.h_out * d.w_out * cs8(d.c_out) * (2 if d.has_residual else 1))
        return 2 * (act + d.c_out * d.c_in * d.kh * d.kw) + extra

    def conv_flops(d):
        return 2.0 * d.n * d.h_out * d.w_out * d.c_out * d.c_in * d.kh * d.kw

    def tag(d, fam, what):
        return "%-7s %-9s n%d %dx%d c%d -> %dx%d c%d k%d s%d d%d%s%s" % (
            fam, what, d.n, d.h_in, d.w_in, d.c_in, d.h_out, d.w_out, d.c_out, d.kh, d.stride, d.dilation,
            " ups" if d.in_upsample else "", " +res" if d.has_residual else "")

    def fwd(x, w, b, r, y, dref, stream):
        d = dref._obj
        return timer.bracket(lambda: orig[names[0]](x, w, b, r, y, dref, stream), conv_flops(d), conv_bytes(d),
                             tag(d, KIND[kind(dref, 0, stream)], "fwd"))

    def bwd(dy, w, dx, dref, stream):
        d = dref._obj      # the FORWARD descriptor: dx has its input shape
        return timer.bracket(lambda: orig[names[1]](dy, w, dx, dref, stream), conv_flops(d), conv_bytes(d),
                             tag(d, KIND[kind(dref, 1, stream)], "bwd_data"))

    def bwd_add(dy, w, dx_add, dx, dref, stream):
        d = dref._obj
        return timer.bracket(lambda: orig[names[2]](dy, w, dx_add, dx, dref, stream), conv_flops(d),
                             conv_bytes(d, 2 * d.n * d.h_in * d.w_in * cs8(d.c_in)), tag(d, KIND[kind(dref, 1, stream)], "bwd_data+"))

    def fwd_stats(x, w, b, y, partial, nbytes, dref, stream):
        d = dref._obj
        return timer.bracket(lambda: orig[names[3]](x, w, b, y, partial, nbytes, dref, stream), conv_flops(d), conv_bytes(d),
                             tag(d, "gemm", "fwd+st"))

    def wgrad(x, dy, dw, db, dref, ws, ws_bytes, stream):
        d = dref._obj
        nb = 2 * d.n * (d.h_in * d.w_in * cs8(d.c_in) // (4 if d.in_upsample else 1) + d.h_out * d.w_out * cs8(d.c_out)) \
            + 4 * d.c_out * d.c_in * d.kh * d.kw
        return timer.bracket(lambda: orig[names[4]](x, dy, dw, db, dref, ws, ws_bytes, stream), conv_flops(d), nb,
                             tag(d, "wgrad", "bwd_w"))

    def spade(x, cond, mean, rstd, packed, y, dref, stream):
        d = dref._obj
        fl = d.n * d.h * d.w * 2.0 * (d.cond_c * 9 * d.hidden + 2 * d.hidden * 9 * d.c)
        nb = d.n * d.h * d.w * 2 * (cs8(d.c) * (1.25 if d.x_upsample else 2) + 4)
        return timer.bracket(lambda: orig[names[5]](x, cond, mean, rstd, packed, y, dref, stream), fl, int(nb),
                             "%-7s %-9s n%d %dx%d c%d%s" % ("spade", "fwd", d.n, d.h, d.w, d.c, " ups" if d.x_upsample else ""))

    def spade_train(x, cond, mean, rstd, packed, y, gamma, dref, stream):      # the training launch also writes gamma
        d = dref._obj
        fl = d.n * d.h * d.w * 2.0 * (d.cond_c * 9 * d.hidden + 2 * d.hidden * 9 * d.c)
        nb = d.n * d.h * d.w * 2 * (cs8(d.c) * (2.25 if d.x_upsample else 3) + 4)
        return timer.bracket(lambda: orig[names[6]](x, cond, mean, rstd, packed, y, gamma, dref, stream), fl, int(nb),
                             "%-7s %-9s n%d %dx%d c%d%s" % ("spade", "fwd+gamma", d.n, d.h, d.w, d.c, " ups" if d.x_upsample else ""))

    def bwd_relu(dy, w, relu_out, dx, dref, stream):
        d = dref._obj
        return timer.bracket(lambda: orig[names[7]](dy, w, relu_out, dx, dref, stream), conv_flops(d),
                             conv_bytes(d, 2 * d.n * d.h_in * d.w_in * cs8(d.c_in)), tag(d, KIND[kind(dref, 1, stream)], "bwd_data*"))

    def spade_hid_bwd(dgb, wdg, cond, wsh, bsh, dw, db, ws, ws_bytes, dref, stream):
        # fused hidden-map backward: the data gradient of the gamma||beta conv (2c -> hidden, 3x3) + the hidden tile's
        # re-computation and mlp_shared's weight gradient (cond_c x 9 x hidden, twice); reads dgb and the conditioning image
        d = dref._obj
        npix = d.n * d.h * d.w
        fl = npix * 2.0 * (2 * d.c * 9 * d.hidden + 2 * d.cond_c * 9 * d.hidden)
        nb = npix * 2 * (cs8(2 * d.c) + cs8(d.cond_c))
        return timer.bracket(lambda: orig[names[8]](dgb, wdg, cond, wsh, bsh, dw, db, ws, ws_bytes, dref, stream), fl, int(nb),
                             "%-7s %-9s n%d %dx%d c%d" % ("spadebw", "hidden", d.n, d.h, d.w, d.c))

    def bwd_add_relu(dy, w, dx_add, relu_out, dx, dref, stream):
        d = dref._obj
        return timer.bracket(lambda: orig[names[9]](dy, w, dx_add, relu_out, dx, dref, stream), conv_flops(d),
                             conv_bytes(d, 4 * d.n * d.h_in * d.w_in * cs8(d.c_in)), tag(d, KIND[kind(dref, 1, stream)], "bwd_data+*"))

    for n, f in zip(names, (fwd, bwd, bwd_add, fwd_stats, wgrad, spade, spade_train, bwd_relu, spade_hid_bwd, bwd_add_relu)):
        setattr(lib, n, f)

    def uninstall():
        for n in names:
            setattr(lib, n, orig[n])
    return uninstall


def all_mfma_summary(timer, steps):
    """Per kernel family of ``install_all_mfma_timer``: launches and ms per step, achieved TFLOP/s and algorithmic GB/s
    against both rooflines."""
    fams = {}
    for e0, e1, fl, nb, tg in timer.pairs:
        a = fams.setdefault(tg.split()[0], [0, 0.0, 0.0, 0.0])
        a[0] += 1
        a[1] += e0.elapsed_time(e1)
        a[2] += fl
        a[3] += nb
    out = {}
    for fam, (n, ms, fl, nb) in sorted(fams.items(), key=lambda kv: -kv[1][1]):
        out[fam] = {"launches_per_step": n // steps, "ms_per_step": round(ms / steps, 3), "tflops": round(fl / ms / 1e9, 1),
                    "frac_mfma": round(fl / ms / 1e9 / MFMA_PEAK_TFLOPS, 4), "algorithmic_GBps": round(nb / ms / 1e6, 1),
                    "frac_hbm": round(nb / ms / 1e6 / (HBM_PEAK_BYTES / 1e9), 4)}
    tot_ms, tot_fl = sum(a[1] for a in fams.values()), sum(a[2] for a in fams.values())
    out["all"] = {"launches_per_step": sum(a[0] for a in fams.values()) // steps, "ms_per_step": round(tot_ms / steps, 3),
                  "tflops": round(tot_fl / tot_ms / 1e9, 1), "frac_mfma": round(tot_fl / tot_ms / 1e9 / MFMA_PEAK_TFLOPS, 4),
                  "algorithmic_flops_per_step": tot_fl / steps}
    return out


def mfma_roofline(run, steps=2, table_path=""):
    """``roofline`` sub-object of a sub-block: ``steps`` extra runs of ``run`` (single-stream) with EVERY MFMA-kernel launch
    bracketed by events (install_all_mfma_timer), per kernel family against both rooflines; the per-shape table goes to
    ``table_path`` when given."""
    t = LaunchTimer()
    un = install_all_mfma_timer(t)
    t.enabled = True
    try:
        for _ in range(steps):
            run()
        torch.cuda.synchronize()
    finally:
        t.enabled = False
        un()
    if not t.pairs:
        return None
    fam = all_mfma_summary(t, steps)
    if table_path:
        with open(table_path, "w") as f:
            f.write(t.table(steps) + "\n")
    return {"bound": "mfma", "kernel": "every MFMA kernel of the workload (wide-layer GEMM family, 3x3 LDS-tiled, general, weight "
                                       "gradients, fused SPADE): launch events on %d extra single-stream runs" % steps,
            "achieved": fam["all"]["tflops"], "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": fam["all"]["frac_mfma"],
            "mfma_kernel_ms": fam["all"]["ms_per_step"], "algorithmic_flops": fam["all"]["algorithmic_flops_per_step"],
            "by_family": {k: v for k, v in fam.items() if k != "all"}}


# ------------------------------------------------------------------------------------------------ workloads
def _dev(a, device):
    return torch.from_numpy(a).to(device)


def joint_batch(bs, rank, device, domains=("r", "s", "rf"), first=None):
    """SURVEY 8d synthetic inputs: x ~ U(-1, 1); 3-rectangle masks; depth ~ U(0.35, 6.95); classes uniform in 0..10.
    ``first`` (the headline): the batch is samples [first, first + bs) of the job's GLOBAL batch -- sample j is drawn from
    its own seed, so the union over the ranks of an N-GPU job is the same 32 samples per domain whatever N is."""
    import numpy as np

    from climategan_amd import fill

    hs = H // 4

    def draw(fn, shape_tail, seed, *a):
        if first is None:
            return fn((bs,) + shape_tail, seed + (1000 * rank if seed >= 300 else rank), *a)
        return np.concatenate([fn((1,) + shape_tail, seed + 7919 * (first + j), *a) for j in range(bs)])

    def masks(seed):
        if first is None:
            return fill.rect_mask(bs, H, W, seed + (1000 * rank if seed >= 300 else rank))
        return np.concatenate([fill.rect_mask(1, H, W, seed + 7919 * (first + j)) for j in range(bs)])

    batch = {}
    if "rf" in domains:
        batch["rf"] = {"data": {"x": _dev(draw(fill.uniform, (3, H, W), 100), device), "m": _dev(masks(200), device)}}
    for i, dom in enumerate(("r", "s")):
        if dom not in domains:
            continue
        sd = 300 + 10 * i
        batch[dom] = {"data": {"x": _dev(draw(fill.uniform, (3, H, W), sd), device),
                               "d": _dev(draw(fill.uniform, (1, hs, hs), sd + 1, 0.35, 6.95), device),
                               "s": _dev((draw(fill.uniform01, (1, hs, hs), sd + 2) * 11).astype(np.int64).clip(0, 10), device),
                               "m": _dev(masks(sd + 3), device)}}
    return {d: batch[d] for d in domains}


def build_trainer(device, dtype, tasks=("d", "s", "m", "p"), freeze=False):
    """``Trainer.setup(inference=False)`` of the default config, every parameter overwritten by the portable fill."""
    from climategan_amd import fill
    from climategan_amd.config import default_opts
    from climategan_amd.trainer import Trainer

    opts = default_opts()
    opts.tasks = list(tasks)
    opts.gen.p.latent_dim = LATENT
    opts.gen.p.spade_n_up = N_UP
    T = Trainer(opts, device=device).setup(inference=False)
    for mod, seed, kw in ((T.G, 0, WELL_CONDITIONED), (T.D, 1, {})):
        shapes = {k: tuple(v.shape) for k, v in mod.state_dict().items()}
        mod.load_state_dict({k: torch.from_numpy(v) for k, v in fill.fill_state_dict(shapes, seed=seed, **kw).items()})
    if "p" in tasks:
        vgg = T.losses["G"]["p"]["vgg"].vgg
        shapes = {k: tuple(v.shape) for k, v in vgg.state_dict().items()}
        vgg.load_state_dict({k: torch.from_numpy(v) for k, v in fill.fill_state_dict(shapes, seed=2, gain=6 ** 0.5).items()})
    # (under torchrun the replicas were broadcast inside setup(); every rank then loads the same portable fill)
    T.G.set_compute_dtype(dtype)
    T.D.set_compute_dtype(dtype)
    # ``freeze`` (bench.py's own runs): this process trains this one trainer at a time, its long-lived Python objects leave
    # the garbage collector's generations (a full collection cost one 160-220 ms step in ~34, R5 DESIGN 4.11); opt-in since
    # round 5 because it is process-global; ``T.close()`` undoes it before the trainer is dropped
    if freeze:
        T.freeze_host_objects()
    return T


def build(device, dtype):
    """configs[1]: the default Painter alone (also used by tests/test_gpu_fullsize.py)."""
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


# ------------------------------------------------------------------------------------------------ CPU baselines
def host_cores():
    """(physical cores, hardware threads) of the host from /proc/cpuinfo."""
    threads = os.cpu_count() or 1
    try:
        phys = set()
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                pid = line.split(":")[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":")[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    phys.add((pid, cid))
                pid = cid = None
        return (len(phys) or threads), threads
    except OSError:
        return threads, threads


CPU_BUDGET_S = 600.0     # hard cap of the CPU leg (a pathological host only: the protocol below takes ~150-180 s)
CPU_THREADS = 32         # torch intra-op threads of the CPU legs, see cpu_baseline_train


def cpu_baseline_train():
    """Oracle (``oracle.cpu_ref.joint_train_step``: the CPU restatement of update_G + ExtraAdam extrapolation + update_D,
    torch fp32 autograd, pinned by the reference's own step -- tests/test_oracle_joint_step.py) beside the headline, SURVEY
    8d's protocol as written: same synthetic inputs, fp32, **3 warm-up steps + 5 timed steps at 1 sample per domain** (3
    images, 640x640, default networks) and **one timed step at 4 per domain** (a step there takes ~70 s: no further
    repeats), ``time.perf_counter``.  Threads: 8d says ``os.cpu_count()``; measured on this host class in round 3 (256
    hardware threads), torch's intra-op threading is 2x SLOWER there than at 16-32 threads (23.4 s against 9.7-12.2 s per
    step), so the leg runs at min(32, physical cores) and says so in ``cores``."""
    import numpy as np

    from climategan_amd import fill
    from oracle import cpu_ref

    t_start = time.perf_counter()
    phys, threads = host_cores()
    use = max(1, min(CPU_THREADS, phys))
    torch.set_num_threads(use)
    shapes_g = json.loads((ROOT / "tests" / "golden" / "generator_masker_shapes.json").read_text())
    shapes_g = {k: tuple(v) for k, v in shapes_g.items()}
    shapes_g.update({"painter." + k: v for k, v in painter_shapes(LATENT, N_UP).items()})
    sd_g = {k: torch.from_numpy(v) for k, v in fill.fill_state_dict(shapes_g, 0, **WELL_CONDITIONED).items()}

    def sn(prefix, cin, cout, k):
        return {prefix + ".module.weight_u": (cout,), prefix + ".module.weight_v": (cin * k * k,),
                prefix + ".module.weight_bar": (cout, cin, k, k), prefix + ".module.bias": (cout,)}

    shapes_d = {}
    for i in range(3):                                     # define_D(4, ndf 64, n_layers 4, num_D 3): discriminator.py:83-169
        chans = [4, 64, 128, 256, 512, 512, 1]
        for j in range(6):
            shapes_d.update(sn("p.discriminator_%d.model%d.0" % (i, j), chans[j], chans[j + 1], 4))
    for task, nc in (("m", 2), ("s", 11)):                 # get_fc_discriminator: discriminator.py:327-361
        chans = [nc, 64, 128, 256, 512, 1]
        for j, idx in enumerate((0, 2, 4, 6, 8)):
            shapes_d.update(sn("%s.Advent.%d" % (task, idx), chans[j], chans[j + 1], 4))
    sd_d = {k: torch.from_numpy(v) for k, v in fill.fill_state_dict(shapes_d, 1).items()}
    sd_v = {k: torch.from_numpy(v) for k, v in fill.fill_state_dict(cpu_ref.vgg19_shapes(), 2, gain=6 ** 0.5).items()}
    hs = H // 4

    def make_batch(bs):
        batch = {}
        for i, dom in enumerate(("r", "s")):
            sd = 300 + 10 * i
            batch[dom] = {"x": torch.from_numpy(fill.uniform((bs, 3, H, W), sd)),
                          "d": torch.from_numpy(fill.uniform((bs, 1, hs, hs), sd + 1, 0.35, 6.95)),
                          "s": torch.from_numpy((fill.uniform01((bs, 1, hs, hs), sd + 2) * 11).astype(np.int64).clip(0, 10)),
                          "m": torch.from_numpy(fill.rect_mask(bs, H, W, sd + 3))}
        batch["rf"] = {"x": torch.from_numpy(fill.uniform((bs, 3, H, W), 100)), "m": torch.from_numpy(fill.rect_mask(bs, H, W, 200))}
        return batch

    def one_step(batch):
        sg, sdd = {k: v.clone() for k, v in sd_g.items()}, {k: v.clone() for k, v in sd_d.items()}
        t0 = time.perf_counter()
        out = cpu_ref.joint_train_step(sg, sdd, sd_v, batch, N_UP, 3, 4)
        dt = time.perf_counter() - t0
        assert all(torch.isfinite(v).all() for v in out["terms"].values())
        return dt

    b1 = make_batch(1)
    warm, runs1 = [], []
    for _ in range(3):
        warm.append(one_step(b1))
        if time.perf_counter() - t_start > CPU_BUDGET_S / 3:         # (hard cap only)
            break
    for _ in range(5):
        runs1.append(one_step(b1))
        if time.perf_counter() - t_start > CPU_BUDGET_S * 2 / 3:
            break
    mean1, best1 = sum(runs1) / len(runs1), min(runs1)
    bs4 = None
    if time.perf_counter() - t_start + 8.0 * best1 < CPU_BUDGET_S:   # measured: 68-74 s at 4 per domain against 10-12 s at 1
        dt4 = one_step(make_batch(4))
        bs4 = {"images_per_s": round(4.0 / dt4, 5), "s_per_step": round(dt4, 2), "timed_steps": 1, "warmup_steps": 0}
    return {"value": round(1.0 / mean1, 5), "unit": "images/s", "cores": use, "kind": "port",
            "physical_cores": phys, "hardware_threads": threads,
            "bs1": {"warmup_s": [round(v, 2) for v in warm], "timed_s": [round(v, 2) for v in runs1],
                    "mean_s": round(mean1, 3), "best_s": round(best1, 3), "images_per_s_best": round(1.0 / best1, 5)},
            "bs4": bs4 if bs4 is not None else "skipped: hard cap of %.0f s reached" % CPU_BUDGET_S,
            "seconds_total": round(time.perf_counter() - t_start, 1),
            "sample": "oracle.cpu_ref.joint_train_step (torch fp32 CPU restatement of trainer.py:989-1032 incl. the ExtraAdam "
                      "extrapolation between the G and the D update), 640x640, SURVEY 8d protocol: %d warm-up + %d timed steps at 1 "
                      "sample per domain (3 images per step; value = 1 / mean step time), %s timed step at 4 per domain; %d torch "
                      "threads on a host with %d physical cores / %d hardware threads"
                      % (len(warm), len(runs1), "1" if bs4 else "no", use, phys, threads)}


def cpu_baseline_paint(sd):
    """configs[1] beside its GPU number: oracle ``cpu_ref.paint`` bs 1, 640x640 (1 warm-up + 2 runs, best of 2 thread counts)."""
    from climategan_amd import fill
    from oracle import cpu_ref

    x = torch.from_numpy(fill.uniform((1, 3, H, W), seed=1))
    m = torch.from_numpy(fill.rect_mask(1, H, W, seed=2))
    sdc = {k: v.clone() for k, v in sd.items()}
    z = H // 2 ** N_UP
    phys, threads = host_cores()
    best = None
    for use in sorted({min(8, phys), min(32, phys)}):
        torch.set_num_threads(use)
        with torch.no_grad():
            cpu_ref.paint(sdc, m, x, z, z)
            t0 = time.perf_counter()
            for _ in range(2):
                cpu_ref.paint(sdc, m, x, z, z)
            dt = (time.perf_counter() - t0) / 2
        if best is None or dt < best[0]:
            best = (dt, use)
    return {"value": round(1.0 / best[0], 4), "unit": "images/s", "cores": best[1], "kind": "port",
            "sample": "oracle.cpu_ref.paint bs 1 640x640, 1 warm-up + 2 runs, %d torch threads (host: %d physical cores)"
                      % (best[1], phys)}


# ------------------------------------------------------------------------------------------------ sub-blocks
def painter_block(steps, warmup, rank, world, device, dtype, dist, barrier, with_cpu):
    """configs[1]: ``OmniGenerator.paint`` bs 8 bf16 with the fused-SPADE MFMA roofline (the kernel north_star names)."""
    from climategan_amd import ops
    import climategan_amd.norms as norms_mod

    G, sd = build(device, dtype)
    x, m = synthetic_batch(rank, device)
    timer = LaunchTimer()
    orig = ops.spade_fused
    layers, flops_img = spade_layer_table(LATENT, N_UP, H, W)
    per_launch = {}

    def timed(xn, *a, **k):
        c, hw = xn.c, (xn.h * (2 if k.get("x_upsample") else 1), xn.w * (2 if k.get("x_upsample") else 1))
        return timer.bracket(lambda: orig(xn, *a, **k), xn.n * hw[0] * hw[1] * 2.0 * (3 * 9 * 128 + 2 * 128 * 9 * c),
                             tag="%dx%d c%d" % (hw[0], hw[1], c))


    ops.spade_fused = norms_mod.ops.spade_fused = timed
    out = {}

    count = {"i": 0}
    sampled = len(range(0, steps, EVENT_EVERY))

    def step():
        timer.enabled = timer.armed and count["i"] % EVENT_EVERY == 0
        count["i"] += 1
        out["y"] = G.paint(m, x)

    try:
        with torch.no_grad():
            timed_steps(step, 0, warmup, barrier, SUB_WARM_SECONDS[0])          # warm-up only
            count["i"] = 0
            timer.armed = True
            elapsed = max_over_ranks(timed_steps(step, steps, 0, barrier), dist, device)
            timer.armed = timer.enabled = False
    finally:
        ops.spade_fused = norms_mod.ops.spade_fused = orig
    assert out["y"].shape == (BATCH_PER_GPU, 3, H, W) and torch.isfinite(out["y"]).all()
    ms, n = timer.total_ms(), len(timer.pairs)
    achieved = timer.total_flops() / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    assert abs(timer.total_flops() - flops_img * BATCH_PER_GPU * sampled) <= 1e-6 * timer.total_flops()
    alg_bytes = sum(BATCH_PER_GPU * a * b * 2 * (2 * ((c + 7) // 8 * 8) + 4) for c, (a, b) in layers)
    # the five full-resolution launches (up_spades' last block + final_spade: 60 % of the 23 launches' FLOPs), on their own
    full = [p for p in timer.pairs if p[4].startswith("%dx%d " % (H, W))]
    full_ms = sum(p[0].elapsed_time(p[1]) for p in full)
    full_tf = sum(p[2] for p in full) / (full_ms * 1e-3) / 1e12 if full_ms > 0 else 0.0
    agg = {}
    for e0, e1, fl, _, tag in timer.pairs:
        a = agg.setdefault(tag, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += e0.elapsed_time(e1)
        a[2] += fl
    by_shape = {tag: {"launches_per_step": a[0] // max(sampled, 1), "avg_us": round(a[1] / a[0] * 1e3, 1),
                      "tflops": round(a[2] / a[1] / 1e9, 1), "frac": round(a[2] / a[1] / 1e9 / MFMA_PEAK_TFLOPS, 3)}
                for tag, a in agg.items() if a[1] > 0}
    res = {"workload": "BASELINE configs[1]: Painter-only SPADE generator fwd 640x640 bs=8 (OmniGenerator.paint incl. mask, "
                       "spectral-norm power iterations, paste), %s" % str(dtype).split(".")[1],
           "images_per_s": round(world * BATCH_PER_GPU * steps / elapsed, 2), "ms_per_step": round(elapsed / steps * 1e3, 3),
           "steps": steps, "warmup": warmup, "warmup_seconds_min": SUB_WARM_SECONDS[0],
           "roofline": {"bound": "mfma", "kernel": "spade_fused_kernel (23 launches/step: all SPADE layers of the Painter)",
                        "achieved": round(achieved, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(achieved / MFMA_PEAK_TFLOPS, 4),
                        "traffic": recorded_traffic("*_spade_hbm_pmc.csv"),
                        "algorithmic_bytes_per_launch": alg_bytes // max(len(layers), 1),
                        "algorithmic_flops_per_step": flops_img * BATCH_PER_GPU,
                        "launches_per_step": n // max(sampled, 1), "avg_launch_ms": round(ms / max(n, 1), 4),
                        "bracketed_steps": "%d of the %d timed steps (every %d-th)" % (sampled, steps, EVENT_EVERY),
                        "share_of_step": round((ms / max(sampled, 1)) / (elapsed / steps * 1e3), 3),
                        "by_shape": by_shape,
                        "roofline_target_set": {
                            "launches": "the %d launches per step at %dx%d (C = 40, 40, 20, 20, 20: up_spades' last block and "
                                        "final_spade)" % (len(full) // max(sampled, 1), H, W),
                            "algorithmic_flops_per_step": sum(p[2] for p in full) / max(sampled, 1),
                            "achieved": round(full_tf, 2), "frac": round(full_tf / MFMA_PEAK_TFLOPS, 4),
                            "avg_launch_ms": round(full_ms / max(len(full), 1), 4)}}}
    if with_cpu:
        res["cpu_baseline"] = cpu_baseline_paint(sd)
    return res


def masker_block(steps, warmup, rank, world, device, dtype, dist, barrier, table_path=""):
    """configs[2]: Masker train step (encoder + depth / seg / mask decoders + ADVENT discriminators), bs 8."""
    T = build_trainer(device, dtype, tasks=("d", "s", "m"), freeze=True)
    batch = joint_batch(MASKER_BS, rank, device, domains=("r", "s"))
    elapsed = max_over_ranks(timed_steps(lambda: T.train_step(batch), steps, warmup, barrier), dist, device)
    assert all(torch.isfinite(v) for v in T.loss_log.values())
    roof = None
    if world == 1:
        T.overlap_branches = False
        roof = mfma_roofline(lambda: T.train_step(batch), 2, table_path)
    T.close()
    return {"workload": "BASELINE configs[2]: Masker train step (Trainer.train_step, tasks d,s,m; domains r,s), 640x640, "
                        "bs 8 per domain per GPU, bf16",
            "images_per_s": round(world * MASKER_BS * steps / elapsed, 2), "ms_per_step": round(elapsed / steps * 1e3, 2),
            "steps": steps, "warmup": warmup, "roofline": roof}


def slice_block(steps, warmup, rank, device, dtype, barrier):
    """The headline step at ONE RANK'S SHARE of an 8-GPU job: 4 samples per domain (the rounds 1-5 headline).  What a rank
    of the N = 8 point computes between two gradient exchanges; ``GLOBAL_BS / ms`` of this block x 8 against the
    headline's value is the strong-scaling ceiling before any communication."""
    T = build_trainer(device, dtype, freeze=True)
    batch = joint_batch(SLICE_BS, rank, device)
    T.G.painter.set_latent_shape((SLICE_BS, 3, H, W), True)
    torch.cuda.reset_peak_memory_stats()
    elapsed = timed_steps(lambda: T.train_step(batch), steps, warmup, barrier)
    assert all(torch.isfinite(v) for v in T.loss_log.values())
    roof = None
    T.overlap_branches = False
    roof = mfma_roofline(lambda: T.train_step(batch), 2, "")
    T.close()
    return {"workload": "BASELINE configs[3], one rank's share at N = 8: joint G+D train step, 640x640, %d samples per domain "
                        "(%d images per step), bf16" % (SLICE_BS, 3 * SLICE_BS),
            "batch_per_domain": SLICE_BS, "images_per_s": round(SLICE_BS * steps / elapsed, 3),
            "ms_per_step": round(elapsed / steps * 1e3, 2), "steps": steps, "warmup": warmup,
            "max_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1), "roofline": roof}


def infer_block(steps, warmup, rank, world, device, dist, barrier, table_path="", only_fp32=False):
    """configs[4]: the apply_events inference loop (Trainer.infer_all: Masker + flood painter + wildfire + smog, uint8
    results copied to the host), 640x640, 16 images per GPU, fp16."""
    from climategan_amd import fill
    from climategan_amd.config import default_opts
    from climategan_amd.trainer import Trainer

    opts = default_opts()
    opts.tasks = ["d", "s", "m", "p"]
    T = Trainer(opts, device=device).setup(inference=True)
    shapes = {k: tuple(v.shape) for k, v in T.G.state_dict().items()}
    T.G.load_state_dict({k: torch.from_numpy(v) for k, v in fill.fill_state_dict(shapes, seed=0, **WELL_CONDITIONED).items()})
    T.G.set_compute_dtype(torch.float16)
    x = torch.from_numpy(fill.uniform((INFER_BS, 3, H, W), 3000 + rank)).to(device)
    out = {}

    def step():
        out.update(T.infer_all(x, numpy=True, bin_value=0.5, half=True))

    if only_fp32:                       # development aid (--only infer32): the fp32-grade modes alone, one after the other
        res = {"workload": "apply_events, fp32-grade modes only"}
        mode = os.environ.get("CGAN_FP32_MODE", "")
        for name in ([mode] if mode else ["split24", "pair16", "split24+fp16painter", "pair16+fp16painter"]):
            T.G.eval()
            T.G.set_compute_dtype(name.split("+")[0])
            if "+" in name:
                T.G.set_painter_compute_dtype(torch.float16)
            dt = timed_steps(lambda: out.update(T.infer_all(x, numpy=True, bin_value=0.5, half=False)), steps, warmup, barrier)
            res[name] = {"images_per_s": round(INFER_BS * steps / dt, 2), "ms_per_batch": round(dt / steps * 1e3, 2)}
        return res
    elapsed = max_over_ranks(timed_steps(step, steps, warmup, barrier, SUB_WARM_SECONDS[0]), dist, device)
    assert set(out) >= {"flood", "wildfire", "smog"} and out["flood"].shape == (INFER_BS, H, W, 3)
    # the opt-in inference mode of SURVEY 8f N2: spectral-norm weights frozen (no power iteration / re-pack per call)
    T.G.freeze_spectral_norm(True)
    frozen = max_over_ranks(timed_steps(step, steps, 2, barrier), dist, device)
    T.G.freeze_spectral_norm(False)
    roof = None
    if world == 1:
        ov = T.overlap_branches
        T.overlap_branches = False
        roof = mfma_roofline(step, 2, table_path)
        T.overlap_branches = ov
    # the reference's DEFAULT apply_events run is fp32 (--half is opt-in, apply_events.py:465-468): G.float() = the
    # split-precision Masker and Painter (bf16 triples through the same MFMA kernels, 6x the multiply work, fp32-grade flood
    # mask and flood image); the event kernels read the maps rounded once to 16 bit
    T.G.eval().float()
    assert T.G.pair_precision

    def step32():
        out.update(T.infer_all(x, numpy=True, bin_value=0.5, half=False))

    n32 = max(3, steps // 2)
    fp32 = max_over_ranks(timed_steps(step32, n32, 2, barrier), dist, device)
    # round 6: the other fp32-grade configurations.  HYBRID = the split-precision Masker (the flood MASK is the fp32-grade one:
    # the bit-exact half of north_star's parity statement) with the Painter back on 16 bit (the painted image then carries the
    # 16-bit Painter's tolerance); "pair16" = fp16 pairs instead of bf16 triples (half the multiplies, fp16's range)
    modes = {}
    for name in ("split24+fp16painter", "pair16", "pair16+fp16painter"):
        T.G.eval()
        T.G.set_compute_dtype(name.split("+")[0])
        if "+" in name:
            T.G.set_painter_compute_dtype(torch.float16)
        dtm = max_over_ranks(timed_steps(step32, n32, 2, barrier), dist, device)
        modes[name] = {"images_per_s": round(world * INFER_BS * n32 / dtm, 2), "ms_per_batch": round(dtm / n32 * 1e3, 2)}
    T.G.set_compute_dtype(torch.float16)
    return {"workload": "BASELINE configs[4]: apply_events inference (Trainer.infer_all: flood + wildfire + smog, uint8 "
                        "results on the host), 640x640, 16 images per GPU, fp16",
            "images_per_s": round(world * INFER_BS * steps / elapsed, 2), "ms_per_batch": round(elapsed / steps * 1e3, 2),
            "images_per_s_frozen_spectral_norm": round(world * INFER_BS * steps / frozen, 2),
            "ms_per_batch_frozen_spectral_norm": round(frozen / steps * 1e3, 2),
            "images_per_s_fp32_grade": round(world * INFER_BS * n32 / fp32, 2),
            "ms_per_batch_fp32_grade": round(fp32 / n32 * 1e3, 2),
            "fp32_grade_mode": "G.float() = set_compute_dtype('split24'): split-precision Masker AND Painter (DESIGN 4.7), %d timed batches" % n32,
            "images_per_s_fp32_grade_mask_16bit_painter": modes["split24+fp16painter"]["images_per_s"],
            "fp32_grade_other_modes": dict(modes, note="hybrid = split-precision Masker (fp32-grade flood mask, tests/test_gpu_configs_640.py::"
                                           "test_hybrid_inference_keeps_the_fp32_grade_mask) + 16-bit Painter (G.set_painter_compute_dtype); pair16 = "
                                           "fp16 pairs, half the multiplies of split24, fp16's range; %d timed batches each" % n32),
            "steps": steps, "warmup": warmup, "warmup_seconds_min": SUB_WARM_SECONDS[0], "roofline": roof}


SUB_BLOCK_TIMEOUT_S = 1200    # watchdog of the sub-blocks (the headline line is complete before they start)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-launch-events", action="store_true", help="do not bracket the conv_gemm launches (no roofline)")
    ap.add_argument("--conv-table", default="", help="write the per-shape table of the bracketed conv_gemm launches here")
    ap.add_argument("--call-log", default="", help="write (C-ABI entry point, algorithmic bytes) of every call of one step here")
    ap.add_argument("--mfma-table-steps", type=int, default=2,
                    help="extra single-stream steps after the timed region with EVERY MFMA-kernel launch bracketed (0 = skip)")
    ap.add_argument("--sub-steps", type=int, default=20, help="timed steps of each sub-block (0 = skip the sub-blocks)")
    ap.add_argument("--sub-warm-seconds", type=float, default=-1.0,
                    help="minimum warm-up time of the millisecond-step blocks (Painter forward, apply_events): default 1 s in "
                         "the default run, 0 with --only")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not re-measure roofline.traffic with two rocprofv3 --pmc child passes (report the committed summary)")
    ap.add_argument("--ddp-bucket-mb", type=float, default=0.0, help="N > 1: gradient bucket size of the reducer (default 25)")
    ap.add_argument("--ddp-bf16-wire", action="store_true", help="N > 1: bf16 gradient buckets on the wire (default fp32)")
    ap.add_argument("--nccl-max-nchannels", type=int, default=0,
                    help="N > 1: NCCL_MAX_NCHANNELS for RCCL (fewer channels = fewer CUs taken from the backward pass)")
    ap.add_argument("--global-batch", type=int, default=GLOBAL_BS,
                    help="samples per domain per step over the whole job (BASELINE configs[3]: 32); anything else is a "
                         "development run and the line says so")
    ap.add_argument("--only", default="", help="run ONE workload as the only measurement (profiling aid): "
                                               "painter | masker | slice | infer | infer32")
    args = ap.parse_args()

    # multi-GPU knobs reach the reducer / RCCL through the environment (read at communicator / reducer construction)
    if args.ddp_bucket_mb > 0:
        os.environ["CGAN_DDP_BUCKET_MB"] = str(args.ddp_bucket_mb)
    if args.ddp_bf16_wire:
        os.environ["CGAN_DDP_BF16_GRADS"] = "1"
    if args.nccl_max_nchannels > 0:
        os.environ["NCCL_MAX_NCHANNELS"] = str(args.nccl_max_nchannels)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                 % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # test hook (tests/test_gpu_bench_two_ranks.py): every rank on device 0 over gloo -- RCCL refuses two ranks on one
    # device, and a one-GPU box is all the tests have; the driver's runs never set these
    backend = os.environ.get("CGAN_BENCH_BACKEND", "nccl")
    if os.environ.get("CGAN_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float16

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    SUB_WARM_SECONDS[0] = args.sub_warm_seconds if args.sub_warm_seconds >= 0 else (0.0 if args.only else 1.0)
    if args.only:
        if args.only == "painter":
            r = painter_block(args.steps, args.warmup, rank, world, device, dtype, dist, barrier, False)
        elif args.only == "masker":
            r = masker_block(args.steps, args.warmup, rank, world, device, dtype, dist, barrier)
        elif args.only == "slice":
            r = slice_block(args.steps, args.warmup, rank, device, dtype, barrier)
        elif args.only == "infer32":
            r = infer_block(args.steps, args.warmup, rank, world, device, dist, barrier, only_fp32=True)
        else:
            r = infer_block(args.steps, args.warmup, rank, world, device, dist, barrier)
        if rank == 0:
            print(json.dumps(r), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return

    # ---------------------------------------------------------------- headline: the joint G+D training step
    # strong scaling: the job's 32 samples per domain are split over the ranks; N = 1 takes them all
    from climategan_amd.parallel import shard_range
    first, per_rank = shard_range(args.global_batch, world, rank)
    T = build_trainer(device, dtype, freeze=True)
    batch = joint_batch(per_rank, rank, device, first=first)
    T.G.painter.set_latent_shape((per_rank, 3, H, W), True)
    timer = LaunchTimer()
    uninstall = (lambda: None) if args.no_launch_events else install_conv_gemm_timer(timer)

    # The launch brackets (two events per launch, 890 per step) go on every EVENT_EVERY-th timed step (the first, eleventh,
    # ...), inside the timed region.  Those steps run the Masker and the Painter branch on ONE stream: with the two-stream
    # overlap of Trainer.update_G / update_D a bracket would time a kernel that shares the chip with the other branch's
    # kernels, which says nothing about the kernel (measured: the same launches read 0.17 of the MFMA peak overlapped,
    # 0.25 alone).  The bracketed steps are therefore ~20 ms slower than the others; ``ms_per_step`` is the mean over all.
    count = {"i": 0}
    sampled = len(range(0, args.steps, EVENT_EVERY)) if not args.no_launch_events else 0
    overlap_default = T.overlap_branches

    # The FIRST warm-up step is single-stream as well (without events): torch's caching allocator keeps a pool per stream, and a
    # one-stream step asks the calling stream's pool for what the side stream's pool holds in a two-stream step -- left to the
    # first bracketed step, that was 84 device mallocs inside the timed region (``allocator`` in the line reports the count).
    serial_warm = {"on": not args.no_launch_events}

    def step():
        bracket = count["i"] % EVENT_EVERY == 0 and (timer.armed or serial_warm["on"])
        timer.enabled = timer.armed and bracket
        T.overlap_branches = overlap_default and not bracket
        count["i"] += 1
        T.train_step(batch)

    timer.armed = False
    for _ in range(args.warmup):
        step()
    serial_warm["on"] = False
    count["i"] = 0
    timer.armed = not args.no_launch_events
    ms0 = torch.cuda.memory_stats(device)
    elapsed = timed_steps(step, args.steps, 0, barrier)
    ms1 = torch.cuda.memory_stats(device)
    # what torch's caching allocator did INSIDE the timed region: a device malloc / free there is a synchronizing driver call
    alloc_stats = {"device_mallocs_in_timed_steps": int(ms1.get("num_device_alloc", 0) - ms0.get("num_device_alloc", 0)),
                   "device_frees_in_timed_steps": int(ms1.get("num_device_free", 0) - ms0.get("num_device_free", 0)),
                   "alloc_retries_in_timed_steps": int(ms1.get("num_alloc_retries", 0) - ms0.get("num_alloc_retries", 0)),
                   "reserved_GB": round(ms1.get("reserved_bytes.all.peak", 0) / 2 ** 30, 1),
                   "allocator_conf": os.environ.get("PYTORCH_HIP_ALLOC_CONF", os.environ.get("PYTORCH_CUDA_ALLOC_CONF", ""))}
    timer.armed = timer.enabled = False
    uninstall()
    elapsed = max_over_ranks(elapsed, dist, device)
    # every MFMA kernel of the step against both rooflines: EXTRA single-stream steps after the timed region (rank 0 of a
    # single-GPU run only: under torchrun an un-reduced extra step would leave the ranks' buckets out of step)
    all_timer = None
    if not args.no_launch_events and world == 1 and args.mfma_table_steps > 0:
        all_timer = LaunchTimer()
        uninstall_all = install_all_mfma_timer(all_timer)
        T.overlap_branches = False
        all_timer.enabled = True
        for i in range(args.mfma_table_steps):
            if args.call_log and i == args.mfma_table_steps - 1:
                from climategan_amd import _lib as _cl
                _cl.CALL_LOG = []
                n0 = len(all_timer.pairs)
            T.train_step(batch)
        torch.cuda.synchronize()
        all_timer.enabled = False
        uninstall_all()
        T.overlap_branches = overlap_default
        if args.call_log:
            # algorithmic bytes of ONE step per C-ABI call (tools/step_hbm_budget.py joins them with the PMC passes by family):
            # the MFMA families from their descriptors (the brackets above), everything else = the operands handed over
            log, _cl.CALL_LOG = _cl.CALL_LOG, None
            mfma_entries = ("cgan_conv2d_nhwc_fwd", "cgan_conv2d_nhwc_bwd_data", "cgan_conv2d_nhwc_bwd_data_add",
                            "cgan_conv2d_nhwc_fwd_stats", "cgan_conv2d_nhwc_bwd_weight", "cgan_spade_fused_fwd",
                            "cgan_spade_fused_fwd_train", "cgan_conv2d_nhwc_bwd_data_relu", "cgan_conv2d_nhwc_bwd_data_add_relu")
            with open(args.call_log, "w") as f:
                for _e0, _e1, _fl, nb, tg in all_timer.pairs[n0:]:
                    f.write("mfma:%s\t%d\t%s\t%.1f\n" % (tg.split()[0], nb, " ".join(tg.split()), _e0.elapsed_time(_e1) * 1e3))
                for entry, nb in log:
                    if entry not in mfma_entries:
                        f.write("%s\t%d\n" % (entry, nb))
    rccl_ranks = None
    if dist is not None:                     # the number of ranks a REAL collective on the job's backend sums over
        one = torch.ones(1, device=device)
        dist.all_reduce(one)
        rccl_ranks = int(one.item())
        if rccl_ranks != world:               # a job whose collective does not span its ranks measures N independent replicas
            sys.exit("bench.py: an all-reduce over the job's process group summed %d ranks, launched with %d" % (rccl_ranks, world))
    losses = {k: float(v) for k, v in T.loss_log.items()}
    assert all(v == v and abs(v) != float("inf") for v in losses.values()), losses
    mem_gb = torch.cuda.max_memory_allocated() / 2 ** 30

    res = None
    if rank == 0:
        res = result_line(world, args.steps, args.warmup, elapsed, args.dtype, per_rank, args.global_batch)
        if args.global_batch != GLOBAL_BS:
            res["config"]["development_run"] = "global batch %d is not BASELINE configs[3]'s %d" % (args.global_batch, GLOBAL_BS)
        if dist is not None:
            res["config"]["rccl_ranks"] = rccl_ranks
            res["config"]["backend"] = dist.get_backend()
            res["config"]["grad_wire_dtype"] = str(T.g_reducer.grad_dtype).split(".")[1] if T.g_reducer is not None else None
            res["config"]["grad_bucket_mb"] = float(os.environ.get("CGAN_DDP_BUCKET_MB", "25"))
            res["config"]["nccl_max_nchannels"] = os.environ.get("NCCL_MAX_NCHANNELS", "default")
        ms, n = timer.total_ms(), len(timer.pairs)
        if n:
            achieved = timer.total_flops() / (ms * 1e-3) / 1e12
            res["roofline"] = {
                "bound": "mfma",
                "kernel": "conv_gemm_kernel (wide-layer implicit GEMM: forward + data-gradient convolutions with >= 64 "
                          "output channels of ResNet-101 / ASPP / decoders / VGG-19 / PatchGAN)",
                "achieved": round(achieved, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / MFMA_PEAK_TFLOPS, 4),
                "traffic": None, "traffic_unit": None,
                "algorithmic_flops_per_step": timer.total_flops() / sampled,
                "algorithmic_bytes_per_launch": int(timer.total_bytes() / n),
                "launches_per_step": n // max(sampled, 1), "avg_launch_ms": round(ms / n, 5),
                "bracketed_steps": "%d of the %d timed steps (every %d-th), run single-stream so that a bracket times the "
                                   "kernel alone; the other steps overlap the Masker and the Painter branch on two streams"
                                   % (sampled, args.steps, EVENT_EVERY),
                "share_of_step": round((ms / sampled) / (elapsed / args.steps * 1e3), 3),
                "by_class": timer.classes()}
            lps = n // max(sampled, 1)
            if sampled and n == lps * sampled:      # the family's time in each bracketed step: how far one step's reading is from the next's
                res["roofline"]["family_ms_per_bracketed_step"] = [
                    round(sum(p[0].elapsed_time(p[1]) for p in timer.pairs[i * lps:(i + 1) * lps]), 2) for i in range(sampled)]
            # (re-measured live further down, once this process has released the trainer's memory: the two child passes need the
            # same 137 GB; until then the newest committed summary stands in)
            res["roofline"]["traffic"] = recorded_traffic("*_conv_gemm_hbm_pmc.csv")
            res["roofline"]["traffic_unit"] = (
                "HBM bytes per launch (mean over the family's %d launches of one step), from the newest committed "
                "profiles/*_conv_gemm_hbm_pmc.csv (rocprofv3 not usable in this run): separate rocprofv3 --pmc FETCH_SIZE / "
                "WRITE_SIZE passes, FETCH_SIZE doubled per the gfx950 correction" % lps)
            mu = recorded_mfma_util()
            if mu:
                gk = [mu[k] for k in ("gemm 1x1", "gemm kxk") if k in mu]
                res["roofline"]["mfma_busy_frac"] = mu.get("gemm", {}).get("mfma_busy") if "gemm" in mu else (
                    {k: mu[k]["mfma_busy"] for k in ("gemm 1x1", "gemm kxk") if k in mu} if gk else None)
                res["mfma_util"] = mu
        else:
            res["roofline"] = None
        if all_timer is not None and all_timer.pairs:
            res["roofline_all_mfma"] = {
                "what": "every launch of an MFMA kernel in %d extra single-stream steps after the timed region (forward / "
                        "data-gradient convs by the kernel their descriptor selects, weight gradients incl. split reduction "
                        "and bias sums, fused SPADE): events on the launch stream, algorithmic FLOPs and bytes from the "
                        "descriptors; frac_mfma against %.0f TFLOP/s, frac_hbm against %.0f GB/s"
                        % (args.mfma_table_steps, MFMA_PEAK_TFLOPS, HBM_PEAK_BYTES / 1e9),
                "by_family": all_mfma_summary(all_timer, args.mfma_table_steps)}
        if n and args.conv_table:
            with open(args.conv_table, "w") as f:
                f.write(timer.table(sampled) + "\n")
                if all_timer is not None and all_timer.pairs:
                    f.write("\n# every MFMA-kernel launch of the step (extra single-stream steps), all families\n")
                    f.write(all_timer.table(args.mfma_table_steps) + "\n")
            # the launches of the LAST bracketed step in launch order (tag, algorithmic bytes, microseconds): joined with the
            # per-dispatch PMC rows of the same command by tools/pmc_by_class.py
            per = n // sampled
            with open(args.conv_table + ".launches", "w") as f:
                for e0, e1, fl, nb, tag in timer.pairs[-per:]:
                    f.write("%s\t%d\t%.1f\n" % (tag, nb, e0.elapsed_time(e1) * 1e3))
        res["config"]["two_stream_overlap"] = bool(overlap_default)
        res["losses_last_step"] = {k: round(v, 4) for k, v in losses.items()}
        res["max_mem_GB"] = round(mem_gb, 1)
        res["allocator"] = alloc_stats
        res["cpu_baseline"] = None

    # ---- the line is complete (minus cpu_baseline / sub-blocks) before anything else runs: a watchdog prints it and
    # ends the process if a sub-block hangs (first RCCL use of a path on a multi-GPU node, an exception on one rank ...)
    import threading
    emitted = threading.Lock()
    finished = threading.Event()

    def emit():
        if rank == 0 and emitted.acquire(blocking=False):
            print(json.dumps(res), flush=True)

    def watchdog():
        if not finished.wait(SUB_BLOCK_TIMEOUT_S):
            if rank == 0:
                res.setdefault("sub_blocks", {})["error"] = "sub-blocks did not finish within %d s" % SUB_BLOCK_TIMEOUT_S
            emit()
            os._exit(0)

    threading.Thread(target=watchdog, daemon=True).start()
    T.close()
    del T, batch
    torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not args.no_live_traffic and res.get("roofline"):
        lps = res["roofline"]["launches_per_step"]
        live = live_traffic(lps, args.global_batch)
        if live is not None:
            res["roofline"]["traffic"] = live
            res["roofline"]["traffic_unit"] = (
                "HBM bytes per launch (mean over the family's %d launches of one step), MEASURED IN THIS RUN (two child passes "
                "of this command's headline step): separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled "
                "per the gfx950 correction" % lps)
    sub = {}
    # the sub-blocks are single-GPU measurements (BASELINE configs[1], [2], [4]); under torchrun only the headline runs: one
    # collective-bearing path less that could leave ranks waiting on each other after the line is out
    if args.sub_steps > 0 and world == 1:
        blocks = (("painter_forward", lambda: painter_block(args.sub_steps, 5, rank, world, device, dtype, dist, barrier,
                                                           world == 1 and not args.no_cpu_baseline)),
                  ("masker_train", lambda: masker_block(args.sub_steps, 3, rank, world, device, dtype, dist, barrier,
                                                        args.conv_table + ".masker" if args.conv_table else "")),
                  ("per_gpu_slice", lambda: slice_block(args.sub_steps, 5, rank, device, dtype, barrier)),
                  ("apply_events", lambda: infer_block(args.sub_steps, 2, rank, world, device, dist, barrier,
                                                       args.conv_table + ".infer" if args.conv_table else "")))
        for name, fn in blocks:
            try:
                sub[name] = fn()
            except Exception as e:          # the headline must survive a failing sub-block
                sub[name] = {"error": "%s: %s" % (type(e).__name__, e)}
            torch.cuda.empty_cache()
    if rank == 0:
        res["sub_blocks"] = sub
        # north_star's named target (>= 40 % of the MFMA peak on the 3x3 SPADE-ResBlk convs at 640x640, bs 8) at the top level
        pf = sub.get("painter_forward") or {}
        if isinstance(pf.get("roofline"), dict):
            ts = pf["roofline"]["roofline_target_set"]
            res["roofline_target_set"] = {"kernel": "spade_fused_kernel, the five 640x640 launches of the Painter forward at bs 8 "
                                                    "(sub_blocks.painter_forward)", "bound": "mfma", "achieved": ts["achieved"],
                                          "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ts["frac"],
                                          "frac_all_23_launches": pf["roofline"]["frac"]}
        if world == 1 and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline_train()
            except Exception as e:
                res["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
    emit()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    finished.set()


if __name__ == "__main__":
    main()
