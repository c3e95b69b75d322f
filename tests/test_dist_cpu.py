"""The N>1 path on CPU: two gloo processes shard a batch of streams, each decodes its
shard (the oracle stands in for the GPU here -- tests may use it), PCM is gathered and
must equal the single-process result; timing is max-reduced."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from libxaac_amd import dist as xdist  # noqa: E402


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 8192, 65536, 65537):
        for world in (1, 2, 3, 8):
            spans = [xdist.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, n_streams, q):
    import oracle_lib
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    dist = xdist.init("gloo")
    orc = oracle_lib.load_oracle()
    rng = np.random.default_rng(99)                 # every rank generates the same global batch ...
    spec, ovl = oracle_lib.random_case(rng, n_streams, mag=17, ovl_mag=15)
    ics = np.stack([rng.integers(0, 4, n_streams), rng.integers(0, 2, n_streams)], 1).astype(np.uint8)
    st = np.stack([rng.integers(0, 4, n_streams), rng.integers(0, 2, n_streams)], 1).astype(np.uint8)
    lo, hi = xdist.shard_range(n_streams, rank, world)   # ... and decodes only its own streams
    r = orc.imdct_batch(spec[lo:hi], ics[lo:hi], ovl[lo:hi], st[lo:hi])
    dist.barrier()
    pcm = xdist.gather_pcm(dist, torch.from_numpy(r["pcm16"]))
    t = xdist.max_over_ranks(dist, 1.0 + rank, torch.device("cpu"))
    if rank == 0:
        q.put((pcm.numpy(), t))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_shard_and_gather(oracle):
    import oracle_lib
    n = 37                                           # odd: shards of 19 and 18 streams
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    [p.start() for p in procs]
    pcm, t = q.get(timeout=120)
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    rng = np.random.default_rng(99)
    spec, ovl = oracle_lib.random_case(rng, n, mag=17, ovl_mag=15)
    ics = np.stack([rng.integers(0, 4, n), rng.integers(0, 2, n)], 1).astype(np.uint8)
    st = np.stack([rng.integers(0, 4, n), rng.integers(0, 2, n)], 1).astype(np.uint8)
    want = oracle.imdct_batch(spec, ics, ovl, st)["pcm16"]
    assert np.array_equal(pcm, want)
    assert t == 2.0                                  # max over ranks of (1.0, 2.0)
