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


def _c4_steps(lib, lo, hi, steps):
    """the HE-AACv2 chain (oracle) on chains lo..hi-1 of tests/golden/sbr_chains.npz for `steps` frame-steps with the SBR
    and PS state carried; returns the PCM of every step [steps, hi - lo, 4096] and the final states' bytes"""
    import ctypes
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from make_golden_sbr_chains import chain_pcm
    ch = np.load(os.path.join(ROOT, "tests", "golden", "sbr_chains.npz"))
    P16 = ctypes.POINTER(ctypes.c_int16)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    st = np.ascontiguousarray(ch["hq_st0"][lo:hi]).copy()
    ps = np.ascontiguousarray(ch["hq_ps0"][lo:hi]).copy()
    out = np.zeros((steps, hi - lo, 4096), np.int16)
    for s in range(steps):
        for i, c in enumerate(range(lo, hi)):
            pin = np.ascontiguousarray(chain_pcm(1, c, s))
            h, f, pf = (np.ascontiguousarray(ch[k][c, s]) for k in ("hq_header", "hq_frame", "hq_ps_frame"))
            rc = lib.xo_sbr_dec_hq(vp(h), vp(f), vp(st[i]), vp(pf), vp(ps[i]), pin.ctypes.data_as(P16), 1,
                                   out[s, i].ctypes.data_as(P16), 2)
            assert rc == 0
    return out, st, ps


def _worker_c4(rank, world, port, n_streams, steps, q):
    import oracle_lib
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    dist = xdist.init("gloo")
    assert dist.get_world_size() == world
    lib = oracle_lib.load_oracle().lib
    lo, hi = xdist.shard_range(n_streams, rank, world)     # this rank's streams: state stays with the rank
    out, st, ps = _c4_steps(lib, lo, hi, steps)
    dist.barrier()
    gathered = [xdist.gather_pcm(dist, torch.from_numpy(out[s])) for s in range(steps)]   # the whole batch on rank 0, None elsewhere
    assert all((g is None) == (rank != 0) for g in gathered)
    t = xdist.max_over_ranks(dist, 0.5 + rank, torch.device("cpu"))
    # the code bench.py --gpus N runs behind its timed region (per-rank rates, the timed and checked gather), equal shards
    m = hi - lo if n_streams % world == 0 else min(xdist.shard_range(n_streams, r, world)[1] - xdist.shard_range(n_streams, r, world)[0] for r in range(world))
    per_rank, info = xdist.post_run_report(dist, torch.from_numpy(out[-1][:m]), 1000.0 * (rank + 1), torch.device("cpu"))
    assert per_rank == [1000.0 * (r + 1) for r in range(world)] and info["bytes_per_rank"] == out[-1][:m].nbytes
    assert info["ok"] is (True if rank == 0 else None)
    if rank == 0:
        q.put((np.stack([g.numpy() for g in gathered]), t))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_c4_chain_with_carried_state(oracle):
    """C5 by construction: 2 ranks shard 23 HE-AACv2 streams (12 + 11), each decodes 3 frame-steps of its shard with
    the SBR / PS state resident on the rank, the PCM of every step is gathered: identical to one process doing all"""
    n, steps = 23, 3
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_c4, args=(r, 2, port, n, steps, q)) for r in range(2)]
    [p.start() for p in procs]
    pcm, t = q.get(timeout=300)
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    want, _, _ = _c4_steps(oracle.lib, 0, n, steps)
    assert pcm.shape == want.shape and np.array_equal(pcm, want)
    assert t == 1.5


def _bench(*argv, env=None, timeout=300):
    import subprocess
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), env=e, capture_output=True, text=True,
                          timeout=timeout)


@pytest.mark.skipif(torch.cuda.is_available() and torch.cuda.device_count() >= 2, reason="box has the GPUs: the run would start")
def test_bench_gpus_2_without_a_launcher_stops_at_the_device_count():
    """`python bench.py --gpus 2` (the driver's bare form) must not die at a world-size assert: with no launcher it starts
    its own ranks, and on a box without two GPUs the only complaint is the missing devices"""
    r = _bench("--gpus", "2", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0
    assert "needs 2 GPUs" in r.stderr and "AssertionError" not in r.stderr and "nproc-per-node" not in r.stderr


@pytest.mark.skipif(torch.cuda.is_available() and torch.cuda.device_count() >= 2, reason="box has the GPUs: the run would start")
def test_bench_rank_under_a_launcher_stops_at_the_device_count():
    """a rank started the driver's way (torchrun environment) on a box without its GPU says so"""
    r = _bench("--gpus", "2", "--steps", "1", "--warmup", "0",
               env=dict(RANK="1", LOCAL_RANK="1", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533"))
    assert r.returncode != 0 and "needs 2 GPUs" in r.stderr


def test_bench_world_size_mismatch_is_reported():
    r = _bench("--gpus", "4", env=dict(RANK="0", LOCAL_RANK="0", WORLD_SIZE="2"))
    assert r.returncode != 0 and "WORLD_SIZE is 2" in r.stderr


def test_bench_self_launch_two_ranks_rendezvous_and_gather():
    """the same launcher path with --launch-check: bench.py starts two ranks through torch.distributed.run, they meet over
    gloo on 127.0.0.1 and run dist.post_run_report; rank 0's line carries what an N > 1 line carries"""
    import json
    r = _bench("--gpus", "2", "--launch-check")
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["launch_check"] is True and line["n_gpus"] == 2 and line["value"] is None
    assert line["max_over_ranks_s"] == 2.0 and line["per_rank_frames_per_s"] == [1000.0, 1001.0]
    assert line["gather"]["ok"] is True and line["gather"]["bytes_per_rank"] == 64 * 4096 * 2
