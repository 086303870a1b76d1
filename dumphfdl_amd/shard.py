"""Rank -> work rules for multi-GPU runs (one process per GPU, launched by torch.distributed.run).

The HFDL path shards trivially: channels are independent and the only shared datum, the block
spectrum, is recomputed per GPU.  There is therefore NO data-path collective -- torch.distributed (RCCL on GPUs, gloo
in the CPU tests) carries only the timing barrier and the final max-time / sum-count reduction.
"""
import os


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def stream_seed(base_seed, rank, world):
    """BASELINE.json config 5: N independent wideband streams, one per GPU, seeds 5..12; config 3 at N=1."""
    return base_seed if world == 1 else 5 + rank


def shard_channels(freqs, rank, world):
    """One wideband stream over `world` GPUs: round-robin channel partition; every GPU ingests the same raw block."""
    return [f for i, f in enumerate(freqs) if i % world == rank]


def reduce_job(elapsed_s, samples, pdus, dist=None, device="cpu"):
    """Whole-job aggregate: time = max over ranks, samples and PDUs = sum over ranks."""
    if dist is None or not dist.is_available() or not dist.is_initialized():
        return float(elapsed_s), int(samples), int(pdus)
    import torch
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    c = torch.tensor([float(samples), float(pdus)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t.item()), int(c[0].item()), int(c[1].item())


def reduce_sums(values, dist=None, device="cpu"):
    """Sum each of `values` (ints) over the ranks."""
    if dist is None or not dist.is_available() or not dist.is_initialized():
        return [int(v) for v in values]
    import torch
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(v) for v in t.tolist()]


def gather_ints(value, dist=None, device="cpu"):
    """One int per rank, in rank order, on every rank (a one-hot sum: no object collectives needed on RCCL)."""
    if dist is None or not dist.is_available() or not dist.is_initialized():
        return [int(value)]
    import torch
    t = torch.zeros(dist.get_world_size(), dtype=torch.float64, device=device)
    t[dist.get_rank()] = float(value)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(v) for v in t.tolist()]


def gather_floats(values, dist=None, device="cpu"):
    """A row of floats per rank, in rank order, on every rank: result[r] = rank r's `values` (one-hot rows summed: plain all_reduce,
    the same on RCCL and gloo)."""
    if dist is None or not dist.is_available() or not dist.is_initialized():
        return [[float(v) for v in values]]
    import torch
    t = torch.zeros(dist.get_world_size(), len(values), dtype=torch.float64, device=device)
    t[dist.get_rank()] = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [[float(v) for v in row] for row in t.tolist()]
