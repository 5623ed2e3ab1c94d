"""Batch sharding of independent GFDM blocks over the GPUs of one node.

GFDM blocks carry no state between them, so the batch is split contiguously: rank r of G processes blocks
[r*B/G, (r+1)*B/G).  There is NO payload collective on the data path; torch.distributed (RCCL over xGMI on the
GPU box, gloo in the CPU tests) is used only to agree on run statistics: blocks processed, an output checksum,
and the slowest rank's time.
"""
import torch
import torch.distributed as dist


def shard_range(total_blocks, rank, world_size):
    """Contiguous, balanced partition: the first (total % world) ranks get one extra block."""
    base, extra = divmod(total_blocks, world_size)
    start = rank * base + min(rank, extra)
    return start, base + (1 if rank < extra else 0)


def output_checksum(out):
    """(sum re, sum im, sum |.|^2) in float64: a size-independent fingerprint of a result tensor."""
    o = out.reshape(-1)
    re = o.real.to(torch.float64)
    im = o.imag.to(torch.float64)
    return torch.stack([re.sum(), im.sum(), (re * re + im * im).sum()])


def reduce_stats(nblocks, checksum, elapsed_s, device):
    """All-reduce run statistics.  Returns (total blocks, summed checksum[3], max elapsed seconds)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return int(nblocks), checksum.detach().to("cpu"), float(elapsed_s)
    sums = torch.cat([torch.tensor([float(nblocks)], dtype=torch.float64, device=device), checksum.to(device)])
    dist.all_reduce(sums, op=dist.ReduceOp.SUM)
    tmax = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    return int(round(sums[0].item())), sums[1:].to("cpu"), float(tmax.item())
