"""Tile-sharded frame across ranks + one film reduce.

One process per GPU.  Rank r renders the 8x8 pixel tiles t with t % world == r into its own zero-initialised
full-frame buffer of *linear* RGB sums; a single `reduce(sum)` onto rank 0 (RCCL over xGMI on GPUs, gloo in the CPU
test) assembles the film; rank 0 resolves (mean, clip, Reinhard, sRGB OETF — non-linear, hence after the reduce).
The renderer is passed in, so the CPU test can drive the identical logic with the oracle."""
import torch
import torch.distributed as dist


def shard_of(rank, world):
    return {"shard_index": rank, "shard_count": max(world, 1)}


def render_frame_sharded(render_accum, accum, rank, world, reduce=True):
    """render_accum(accum_tensor, shard_index, shard_count) adds this rank's tiles into `accum` (a torch tensor on the
    rank's device).  Returns the reduced tensor on rank 0 (other ranks: their partial)."""
    render_accum(accum, **shard_of(rank, world))
    if reduce:
        reduce_film(accum, world)
    return accum


def reduce_film(accum, world):
    """The job's single film exchange: sum the rank-local linear films onto rank 0 (tiles are disjoint, so the sum is
    exact: every pixel has one non-zero contributor).  RCCL over xGMI for device tensors, gloo on the CPU."""
    if world > 1:
        if accum.is_cuda and dist.get_backend() == "gloo":      # rehearsal on a box with fewer GPUs than ranks: stage through the host
            host = accum.cpu()
            dist.reduce(host, dst=0, op=dist.ReduceOp.SUM)
            accum.copy_(host)
        else:
            dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM)
    return accum


def max_over_ranks(seconds, world, device):
    if world <= 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
