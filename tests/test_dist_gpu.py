"""The multi-GPU plumbing (libxaac_amd/dist.py) over RCCL on the one GPU this box has: a single-rank "nccl" process group on
cuda:0 -- communicator set-up, barrier, the max-over-ranks all-reduce, the all-gathers and the PCM gather of
post_run_report run on device tensors through RCCL (world 1: no peer, but every call the N > 1 bench makes is made).
The sharding arithmetic and a real two-rank exchange are covered on CPU over gloo (tests/test_dist_cpu.py); an N > 1 run on
hardware is the driver's."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys, json
sys.path.insert(0, %(root)r)
import numpy as np, torch
import torch.distributed as dist
import libxaac_amd
from libxaac_amd import dist as xdist
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"
# PCM from the HIP path: one AAC-LC batch through the IMDCT kernel
ctx = libxaac_amd.XaacContext(0, torch.cuda.current_stream().cuda_stream)
n = 256
g = torch.Generator(device="cpu").manual_seed(7)
spec = torch.randint(-(1 << 17), 1 << 17, (n, 1024), dtype=torch.int32, generator=g).to(dev)
ics = torch.zeros((n, 2), dtype=torch.uint8, device=dev)
ovl = torch.zeros((n, 512), dtype=torch.int32, device=dev)
st = torch.zeros((n, 2), dtype=torch.uint8, device=dev)
pcm = torch.zeros((n, 1024), dtype=torch.int16, device=dev)
status = torch.zeros(n, dtype=torch.int32, device=dev)
ctx.imdct_process_batch(spec, ics, ovl, st, pcm16=pcm, status=status)
torch.cuda.synchronize()
assert int(status.abs().sum().item()) == 0 and int(pcm.abs().sum().item()) > 0
dist.barrier()
t = xdist.max_over_ranks(dist, 1.25, dev)
whole = xdist.gather_pcm(dist, pcm)
per_rank, info = xdist.post_run_report(dist, pcm, 123.0, dev)
dist.barrier()
dist.destroy_process_group()
print("RESULT " + json.dumps({"t": t, "same": bool(torch.equal(whole, pcm)), "per_rank": per_rank, "ok": info["ok"],
                  "bytes": info["bytes_per_rank"]}))
'''


@pytest.mark.gpu
def test_single_rank_nccl_group_runs_every_collective_of_the_bench():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
    assert lines, (r.stdout[-1000:], r.stderr[-1000:])
    out = json.loads(lines[-1][7:])
    assert out["t"] == 1.25 and out["same"] and out["ok"] is True and out["per_rank"] == [123.0]
    assert out["bytes"] == 256 * 1024 * 2


@pytest.mark.gpu
def test_two_ranks_of_the_bench_decode_on_one_gpu():
    """The N > 1 path of the product on hardware: `bench.py --gpus 2 --share-device` starts two ranks the way the driver's
    command does (torch.distributed.run, one process per rank); each decodes its own 8192 streams with the HIP library on
    the box's one GPU, the ranks meet over gloo (barriers, max-over-ranks timing, the per-rank rates, the final PCM gather
    into rank 0) and rank 0 prints the line.  Rank 0's output is checked against the oracle, the gathered PCM of both against the
    ranks' checksums.  (RCCL itself runs at world 1 above; with two ranks on one device it refuses.)"""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-device", "--steps", "4", "--warmup", "2",
                        "--no-secondary", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert lines, (r.stdout[-1000:], r.stderr[-1000:])
    out = json.loads(lines[-1])
    assert out["n_gpus"] == 2 and "shared_device" in out and out["scaling"] == "weak"
    assert out["bit_exact_vs_oracle"] is True and out["refused_frac"] == 0.0
    assert len(out["per_rank_frames_per_s"]) == 2 and all(v > 0 for v in out["per_rank_frames_per_s"])
    assert out["gather"]["ok"] is True
    # the whole job's frames over the slower rank's time
    assert abs(out["value"] - 2 * 8192 * 4 / (out["ms_per_step"] * 4e-3)) / out["value"] < 1e-3
