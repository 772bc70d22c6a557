"""N > 1 path of bench.py on CPU: world_size 2, gloo.  The hot path shards by image with no data-path collective,
so what has to be right is the launch contract: every rank runs K timed steps between barriers, rank-specific
synthetic shards differ, the whole-job time is the MAX over ranks and the JSON line aggregates all ranks."""
import json
import os
import socket
import sys
import time
from pathlib import Path

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from climategan_amd import fill

    calls = {"n": 0}

    def step():
        calls["n"] += 1
        time.sleep(0.01 * (rank + 1))  # rank 1 is the slow one

    elapsed = bench.timed_steps(step, steps=5, warmup=2, barrier=dist.barrier)
    total = bench.max_over_ranks(elapsed, dist, torch.device("cpu"))
    shard = fill.uniform((2, 3, 8, 8), seed=1000 + rank)  # same seeding rule as bench.synthetic_batch
    line = bench.result_line(world, 5, 2, total, "bf16") if rank == 0 else None
    q.put((rank, calls["n"], elapsed, total, float(shard.sum()), json.dumps(line) if line else None))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_timing_and_aggregation():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, n0, e0, t0, s0, line), (r1, n1, e1, t1, s1, _) = res
    assert (n0, n1) == (7, 7)                      # 2 warm-up + 5 timed on every rank
    assert abs(t0 - t1) < 1e-9 and t0 >= max(e0, e1) - 1e-9   # whole-job time = slowest rank, same on all ranks
    assert e1 >= 5 * 0.02 * 0.9                    # the slow rank really ran its 5 timed steps
    assert s0 != s1                                # ranks get different synthetic shards
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["global_batch"] == 16
    assert abs(d["value"] - 2 * 8 * 5 / t0) < 1e-2
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config"):
        assert key in d
