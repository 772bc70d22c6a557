"""What the joint train step loses while a communication kernel occupies part of the chip (the risk named for the
8-GPU run, measured on one GPU).  A stand-in for RCCL's ring kernels (tools/micro/comm_standin.hip: W persistent workgroups
streaming a[i] += b[i] over a 25 MB bucket on their own stream) runs beside the headline step, ONE launch spanning all timed
steps; the steps are timed by events on the training stream only, and every configuration is bracketed by its own
stand-in-free baseline (the box drifts by milliseconds over a minute).  Two regimes per W: unthrottled (the workgroups move
whatever HBM gives them: an upper bound of the harm) and throttled to a link-like rate (idle slots between 16-byte pieces).
usage (GPU box): python tools/micro/comm_contention.py [--steps 10]"""
import argparse
import ctypes
import subprocess
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--bucket-mb", type=float, default=25.0)
    args = ap.parse_args()
    so = "/tmp/libcomm_standin.so"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC",
                    str(ROOT / "tools/micro/comm_standin.hip"), "-o", so], check=True)
    lib = ctypes.CDLL(so)
    lib.standin_reduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_void_p]
    dev = torch.device("cuda:0")
    T = bench.build_trainer(dev, torch.bfloat16, freeze=True)
    batch = bench.joint_batch(bench.SLICE_BS, 0, dev)
    T.G.painter.set_latent_shape((bench.SLICE_BS, 3, bench.H, bench.W), True)
    n = int(args.bucket_mb * 2 ** 20) // 4 // 4 * 4
    a = torch.zeros(n, device=dev)
    b = torch.ones(n, device=dev)
    comm = torch.cuda.Stream(device=dev)
    for _ in range(8):
        T.train_step(batch)
    torch.cuda.synchronize()

    def standin_ms(wgs, passes, idle):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(comm):
            e0.record()
            lib.standin_reduce(a.data_ptr(), b.data_ptr(), n, wgs, passes, idle, comm.cuda_stream)
            e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)

    def steps_ms(wgs, passes, idle):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if wgs:
            with torch.cuda.stream(comm):
                lib.standin_reduce(a.data_ptr(), b.data_ptr(), n, wgs, passes, idle, comm.cuda_stream)
        e0.record()
        for _ in range(args.steps):
            T.train_step(batch)
        e1.record()
        e1.synchronize()
        still = (not comm.query()) if wgs else True
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / args.steps, still

    print("workgroups idle  base_ms  with_ms  delta_ms  standin_GBps_alone  (bucket %.0f MB, %d steps per arm)" % (args.bucket_mb, args.steps))
    for wgs, idle in ((8, 0), (8, 2), (16, 0), (16, 2), (32, 0), (32, 2), (64, 2), (16, 8)):
        standin_ms(wgs, 1, idle)
        per_pass = standin_ms(wgs, 2, idle) / 2
        alone = 3 * n * 4 / (per_pass * 1e-3) / 1e9
        passes = max(1, int(1.6 * args.steps * 95.0 / per_pass))       # outlasts the timed steps even when they slow down
        base, _ = steps_ms(0, 0, 0)
        ms, still = steps_ms(wgs, passes, idle)
        print("%10d %4d  %7.2f  %7.2f  %+8.2f  %18.1f  %s" % (wgs, idle, base, ms, ms - base, alone,
                                                              "" if still else "(stand-in ended before the steps did)"), flush=True)
    T.close()


if __name__ == "__main__":
    main()
