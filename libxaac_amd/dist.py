"""Multi-GPU plumbing for the stream-sharded path: one process per
GPU, contiguous stream ranges per rank, no data-path collective.  torch.distributed
(backend "nccl" = RCCL on ROCm, "gloo" in the CPU tests) is used only for the
barrier, the max-over-ranks timing and the optional final PCM gather."""
import os


def shard_range(n_items, rank, world):
    """Contiguous [lo, hi) of n_items owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(int(n_items), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend):
    """Join the job described by the torchrun environment; returns the dist module or None at world 1."""
    rank, _, world = env_rank()
    if world == 1:
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if not dist.is_initialized():
        dist.init_process_group(backend, rank=rank, world_size=world)
    return dist


def max_over_ranks(dist, seconds, device):
    if dist is None:
        return seconds
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_pcm(dist, pcm_shard):
    """Final interleaved gather of every rank's PCM shard (all ranks get the whole batch).
    Shards may differ in length by one stream; they are padded to the longest."""
    if dist is None:
        return pcm_shard
    import torch
    dtype, tail = pcm_shard.dtype, tuple(pcm_shard.shape[1:])
    pcm_shard = pcm_shard.contiguous().view(pcm_shard.shape[0], -1).view(torch.uint8)  # bytes: every backend moves them
    n = torch.tensor([pcm_shard.shape[0]], dtype=torch.int64, device=pcm_shard.device)
    sizes = [torch.zeros_like(n) for _ in range(dist.get_world_size())]
    dist.all_gather(sizes, n)
    longest = int(max(int(s.item()) for s in sizes))
    pad = torch.zeros((longest,) + tuple(pcm_shard.shape[1:]), dtype=pcm_shard.dtype, device=pcm_shard.device)
    pad[: pcm_shard.shape[0]] = pcm_shard
    parts = [torch.zeros_like(pad) for _ in sizes]
    dist.all_gather(parts, pad)
    whole = torch.cat([p[: int(s.item())] for p, s in zip(parts, sizes)], 0)
    return whole.view(dtype).view((whole.shape[0],) + tail)
