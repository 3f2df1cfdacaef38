#!/usr/bin/env python3
"""BASELINE.json config 4, multi-GPU: (min,+) SSSP / (or,and) BFS to convergence on R-MAT, rows sharded
over the ranks (nnz-balanced), ONE in-place RCCL all-gather of the new vector per iteration.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      tools/iterate_rmat_dist.py [scale=23] [--check] [--chunks=C]

Rank 0 prints one JSON line: iterations, wall time per iteration (max over ranks), and with --check the
bit-exact comparison against the CPU oracle (slow: single thread)."""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparseharness_amd import hostlib as H  # noqa: E402
from sparseharness_amd.distributed import HipLocalStep, ShardedIteration, ShardPlan  # noqa: E402
from sparseharness_amd.engine import MIN_PLUS_F32, OR_AND_I32  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
scale = int(args[0]) if args else 23
check = "--check" in sys.argv
chunks = next((int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--chunks=")), 1)
rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("WORLD_SIZE", 1), ("LOCAL_RANK", 0)))
torch.cuda.set_device(local)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

rp, ci, va = H.rmat(scale)
n = 1 << scale
out = {"workload": f"rmat-{scale}", "rows": n, "nnz": int(rp[-1]), "n_gpus": world, "chunks": chunks}
for name, sr, a, b in [("sssp", MIN_PLUS_F32, 0.0, 0.0), ("bfs", OR_AND_I32, 1, 0)]:
    dt = np.int32 if sr == OR_AND_I32 else np.float32
    vals = va.astype(dt)
    plan = ShardPlan(rp, ci, vals, rank, world, chunks, semiring=sr)
    step = HipLocalStep(plan, sr, local)
    x0 = np.zeros(n, dt)
    if sr == MIN_PLUS_F32:
        x0[:] = np.float32(3.4028235e38)
        x0[0] = 0
    else:
        x0[0] = 1
    drv = ShardedIteration(plan, sr, step)
    drv.run(x0, x0, a, b, 1e-4, 2)            # warm-up (module load, NCCL channels)
    dist.barrier(); torch.cuda.synchronize()
    t = time.perf_counter()
    final, iters, conv = drv.run(x0, x0, a, b, 1e-4, 500)
    torch.cuda.synchronize(); dist.barrier()
    wall = torch.tensor([time.perf_counter() - t, drv.last_loop_seconds], dtype=torch.float64, device="cuda")
    dist.all_reduce(wall, op=dist.ReduceOp.MAX)
    res = {"iterations": iters, "converged": conv, "wall_ms_total_incl_setup_and_readback": round(float(wall[0]) * 1e3, 3),
           "loop_ms_total": round(float(wall[1]) * 1e3, 3),
           "loop_ms_per_iteration": round(float(wall[1]) * 1e3 / iters, 4), "plan": step.A.plan()[0],
           "rank0_rows": plan.rows, "slot_elems": plan.layout.slot}
    if check and rank == 0:
        from oracle import oracle as O
        want, w_it, w_conv = O.iterate(sr, rp, ci, vals, x0, x0, a, b, 1e-4, 500)
        res["bit_exact_vs_oracle"] = bool((iters, conv) == (w_it, w_conv) and np.array_equal(final.view(np.uint32), want.view(np.uint32)))
    out[name] = res
if rank == 0:
    print(json.dumps(out), flush=True)
dist.barrier()
dist.destroy_process_group()
