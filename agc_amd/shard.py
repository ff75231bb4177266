"""Sample sharding across GPUs (SURVEY.md 8e, round-1 form): samples are dealt round-robin,
rank r takes samples r, r+N, r+2N, ...; every rank keeps its own archive shard, so the data
path needs no collective.  Only timings and counters are reduced (max / sum)."""


def samples_of_rank(n_samples, rank, world):
    return list(range(rank, n_samples, world))


def sample_seed(base, step, rank, world):
    """seed of the sample a rank processes at a step: distinct across ranks and steps"""
    return base + step * world + rank


def reduce_job_time(dist, elapsed, device=None):
    """job time = slowest rank"""
    import torch
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def reduce_counters(dist, values, device=None):
    import torch
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.tolist()
