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


def gather_pcm(dist, pcm_shard, dst=0):
    """The north star's "final interleaved gather": every rank's PCM shard into rank `dst` (a true gather: one ingress of
    (N - 1) shards at `dst`, nothing received elsewhere -- RCCL runs it as point-to-point sends over xGMI).  Returns the
    whole batch on `dst`, None on the other ranks.  Shards may differ in length by one stream; they travel padded to the
    longest and are cut back on arrival."""
    if dist is None:
        return pcm_shard
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    dtype, tail = pcm_shard.dtype, tuple(pcm_shard.shape[1:])
    pcm_shard = pcm_shard.contiguous().view(pcm_shard.shape[0], -1).view(torch.uint8)  # bytes: every backend moves them
    n = torch.tensor([pcm_shard.shape[0]], dtype=torch.int64, device=pcm_shard.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)                                # 8 bytes per rank: the shard lengths
    longest = int(max(int(s.item()) for s in sizes))
    pad = torch.zeros((longest,) + tuple(pcm_shard.shape[1:]), dtype=pcm_shard.dtype, device=pcm_shard.device)
    pad[: pcm_shard.shape[0]] = pcm_shard
    parts = [torch.zeros_like(pad) for _ in sizes] if rank == dst else None
    dist.gather(pad, parts, dst=dst)
    if rank != dst:
        return None
    whole = torch.cat([p[: int(s.item())] for p, s in zip(parts, sizes)], 0)
    return whole.view(dtype).view((whole.shape[0],) + tail)


def post_run_report(dist, pcm, own_frames_per_s, device, barrier=None):
    """What a multi-rank bench run adds behind its timed region (bench.py --gpus N; tests/test_dist_cpu.py runs the same
    code under gloo): every rank's own rate (load balance) and one timed gather_pcm of the ranks' last PCM batch into rank
    0, checked there against the ranks' own shards' checksums.  Returns (per_rank_rates, gather_info); the latter's
    "ok" is None on ranks other than 0."""
    import time
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    barrier = barrier or dist.barrier
    t = torch.tensor([own_frames_per_s], dtype=torch.float64, device=device)
    parts = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    per_rank = [round(float(p.item()), 1) for p in parts]
    # a checksum of every shard, so that rank 0 can tell that what arrived is what was sent
    own = pcm.contiguous().view(-1).view(torch.uint8).to(torch.int64).sum().reshape(1).to(device)
    sums = [torch.zeros_like(own) for _ in range(world)]
    dist.all_gather(sums, own)
    gather_pcm(dist, pcm[: min(64, pcm.shape[0])])           # warm the communicator
    barrier()
    t0 = time.perf_counter()
    whole = gather_pcm(dist, pcm)
    barrier()
    seconds = max_over_ranks(dist, time.perf_counter() - t0, device)
    ok = None
    if rank == 0:
        n = pcm.shape[0]
        ok = whole.shape[0] == world * n and torch.equal(whole[:n], pcm) and all(
            int(whole[r * n:(r + 1) * n].contiguous().view(-1).view(torch.uint8).to(torch.int64).sum().item()) == int(sums[r].item())
            for r in range(world))
    nbytes = int(pcm.numel() * pcm.element_size())
    info = {"ms": round(seconds * 1e3, 3), "bytes_per_rank": nbytes, "ok": ok,
            "what": "dist.gather_pcm: gather of every rank's last PCM batch into rank 0 (%d streams' frames), after the "
                    "timed region" % (world * pcm.shape[0]),
            "GBps_into_rank0": round((world - 1) * nbytes / seconds / 1e9, 2)}
    return per_rank, info
