"""Multi-rank path on CPU: scenario sharding and the statistics all-reduce with the gloo backend, world size 2."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mpc_amd.sharding import allreduce_stats, shard_bounds
    lo, hi = shard_bounds(total, rank, world)
    owned = np.arange(lo, hi)
    # per-rank statistics of "its" scenarios: sums (count, sum of ids) and maxima (largest id, rank-dependent value)
    sums = torch.tensor([float(len(owned)), float(owned.sum())], dtype=torch.float64)
    maxima = torch.tensor([float(owned.max()), 10.0 * (rank + 1)], dtype=torch.float64)
    allreduce_stats(sums, maxima)
    # the dictionary of a controller's stats() (cost and constraint-violation statistics included), fabricated per rank
    from mpc_amd import _lib
    from mpc_amd.sharding import allreduce_solver_stats
    mine = {k: (1e-3 * (rank + 1) if k.startswith("max_") else float(len(owned)) * (i + 1)) for i, k in _lib.STATS_SLOTS.items()}
    red = allreduce_solver_stats(mine)
    q.put((rank, lo, hi, sums.tolist(), maxima.tolist(), red))
    dist.destroy_process_group()


def test_shards_cover_and_statistics_reduce_world2():
    world, total = 2, 4097
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, lo0, hi0, s0, m0, d0), (r1, lo1, hi1, s1, m1, d1) = out
    from mpc_amd import _lib
    assert d0 == d1 and set(d0) == set(_lib.STATS_SLOTS.values())
    for i, k in _lib.STATS_SLOTS.items():                           # maxima take the larger rank's value, the rest add
        assert d0[k] == pytest.approx(2e-3 if k.startswith("max_") else float(total) * (i + 1)), k
    assert {_lib.STATS_SLOTS[i] for i in _lib.STATS_MAX_SLOTS} == {k for k in d0 if k.startswith("max_")}
    for k in ("sum_stage_cost", "sum_lateral_error_sq", "max_bound_violation", "max_heading_pair_violation"):
        assert k in d0                                              # the statistics the runs are judged by travel too
    assert (lo0, hi0, lo1, hi1) == (0, 2049, 2049, 4097)            # contiguous, disjoint, complete
    assert s0 == s1 == [float(total), float(total * (total - 1) // 2)]
    assert m0 == m1 == [float(total - 1), 20.0]


def test_shard_bounds_properties():
    from mpc_amd.sharding import shard_bounds
    for total in (0, 1, 7, 4096, 65536, 1048576 + 3):
        for world in (1, 2, 4, 8):
            edges = [shard_bounds(total, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == total
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)
    # outside a process group the reduction is the identity
    from mpc_amd.sharding import allreduce_stats
    a, b = allreduce_stats(torch.ones(3), torch.zeros(2))
    assert a.tolist() == [1, 1, 1] and b.tolist() == [0, 0]


def test_bench_starts_its_own_ranks_and_reduces_through_the_sharding_helper():
    """`python bench.py --gpus 2` without a launcher starts two ranks itself (fresh child processes under
    torch.distributed.run) and rank 0 prints one line for both; --dry-run keeps the ranks on the CPU (gloo) with
    fabricated per-rank statistics, which travel through bench.reduce_statistics -> sharding.allreduce_stats exactly as
    the measured ones do."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--dry-run", "--steps", "20", "--warmup", "5"],
                         capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                          # one line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["warmup"] == 5
    s, m = d["statistics"]["sum"], d["statistics"]["max"]
    assert s["scenario_steps"] == 2 * 4096 * 20                     # SUM over the ranks
    assert m["elapsed_s"] == 2.0 and m["max_abs_u0"] == 0.2         # MAX over the ranks (rank 1 reports 2.0 s)
    assert d["value"] == s["scenario_steps"] / m["elapsed_s"]       # whole-job units over the slowest rank's time
    assert s["stage_cost"] == pytest.approx(3.0 * 2 * 4096 * 20) and s["closed_loop_steps"] == 2 * 4096 * 20
    assert m["max_heading_pair_violation"] == 2e-14 and m["max_bound_violation"] == 0.0


def test_reduce_statistics_is_identity_for_one_rank():
    import bench
    sums = {k: float(i + 1) for i, k in enumerate(bench.SUM_KEYS)}
    maxima = {k: float(10 * (i + 1)) for i, k in enumerate(bench.MAX_KEYS)}
    s, m = bench.reduce_statistics(sums, maxima)
    assert s == sums and m == maxima
