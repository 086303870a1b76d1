"""The N>1 path on CPU: two gloo ranks shard streams / channels with no data-path collective and reduce the job totals."""
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from dumphfdl_amd import shard
    import bench
    dist.init_process_group("gloo", rank=rank, world_size=world)
    r, lr, w = shard.env_rank()
    freqs = bench.channel_plan(bench.WORKLOADS["cfg2"])
    mine = shard.shard_channels(freqs, r, w)
    seed = shard.stream_seed(3, r, w)
    # every rank "processes" 10 blocks of its own stream; rank 1 is slower
    elapsed, samples, pdus = shard.reduce_job(0.5 + 0.25 * r, 10 * 917504, 7 + r, dist)
    sums = shard.reduce_sums([3 + r, 100 * (r + 1)], dist)
    seeds = shard.gather_ints(seed, dist)
    rows = shard.gather_floats([seed, 0.5 + 0.25 * r, len(mine)], dist)
    dist.barrier()
    dist.destroy_process_group()
    q.put((r, w, mine, seed, elapsed, samples, pdus, sums, seeds, rows))


def test_two_rank_sharding_and_reduction():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    sys.path.insert(0, ROOT)
    import bench
    freqs = bench.channel_plan(bench.WORKLOADS["cfg2"])
    a, b = res
    assert sorted(a[2] + b[2]) == sorted(freqs) and not set(a[2]) & set(b[2])         # partition, no overlap
    assert (a[3], b[3]) == (5, 6)                                                     # independent stream seeds
    for r in res:
        assert r[4] == pytest.approx(0.75) and r[5] == 2 * 10 * 917504 and r[6] == 15  # max time, summed work
        assert r[7] == [7, 300] and r[8] == [5, 6]                                     # summed checks, seeds in rank order on every rank
        assert r[9] == [[5.0, 0.5, 16.0], [6.0, 0.75, 16.0]]                           # per-rank rows, rank order, on every rank


def _worker8(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from dumphfdl_amd import shard
    import bench
    dist.init_process_group("gloo", rank=rank, world_size=world)
    freqs = bench.channel_plan(bench.WORKLOADS["cfg3"])
    mine = shard.shard_channels(freqs, rank, world)
    seed = shard.stream_seed(bench.WORKLOADS["cfg3"]["seed"], rank, world)
    seeds = shard.gather_ints(seed, dist)
    counts = shard.gather_ints(len(mine), dist)
    rows = shard.gather_floats([rank, 2.6 + 0.01 * rank, len(mine)], dist)
    elapsed, samples, pdus = shard.reduce_job(0.67 + 0.001 * rank, 256 * 7340032, 3987, dist)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, mine, seeds, counts, rows, elapsed, samples, pdus))


def test_eight_rank_sharding_of_the_256_channel_plan():
    """BASELINE.json configs[4] as the driver will launch it: 8 ranks.  Stream mode: seeds 5..12 in rank order.  Channel mode: 32 channels
    per rank, a disjoint cover of the 256-channel plan.  Per-rank rows arrive in rank order on every rank; time = max, work = sum."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    sys.path.insert(0, ROOT)
    import bench
    freqs = bench.channel_plan(bench.WORKLOADS["cfg3"])
    cover = []
    for rank, mine, seeds, counts, rows, elapsed, samples, pdus in res:
        assert len(mine) == 32 and mine == freqs[rank::8]
        cover += mine
        assert seeds == list(range(5, 13)) and counts == [32] * 8
        assert [row[0] for row in rows] == [float(r) for r in range(8)] and rows[rank][1] == pytest.approx(2.6 + 0.01 * rank)
        assert elapsed == pytest.approx(0.677) and samples == 8 * 256 * 7340032 and pdus == 8 * 3987
    assert sorted(cover) == sorted(freqs) and len(set(cover)) == 256


def test_single_rank_is_passthrough():
    sys.path.insert(0, ROOT)
    from dumphfdl_amd import shard
    assert shard.reduce_job(1.5, 100, 3) == (1.5, 100, 3)
    assert shard.stream_seed(3, 0, 1) == 3
    assert shard.shard_channels([1, 2, 3], 0, 1) == [1, 2, 3]
    assert shard.reduce_sums([4, 5]) == [4, 5] and shard.gather_ints(9) == [9] and shard.gather_floats([1, 2.5]) == [[1.0, 2.5]]
