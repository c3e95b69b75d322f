"""xaacdec_amd -gpus:G / -device:k: the native host's split of a batch over the devices of one node, checked without a GPU
(-plan prints the split and exits before the first HIP call).  The split is libxaac_amd/dist.py's shard_range -- the one
bench.py --gpus N makes over ranks: contiguous ranges whose sizes differ by at most one."""
import json
import os
import subprocess

import pytest

from libxaac_amd import dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "libxaac_amd", "xaacdec_amd")
AAC = os.path.join(ROOT, "tests", "golden", "streams", "mix_aot29_32k.aac")


def plan(*args):
    p = subprocess.run([CLI, "-plan"] + list(args), capture_output=True, text=True)
    return p.returncode, (json.loads(p.stdout) if p.returncode == 0 else None), p.stderr


@pytest.mark.parametrize("copies,gpus,device", [(1, 1, 0), (10, 3, 0), (8192, 8, 0), (7, 8, 0), (65536, 8, 0), (5, 2, 3)])
def test_split_is_shard_range(copies, gpus, device):
    rc, d, err = plan("-ifile:" + AAC, "-copies:%d" % copies, "-gpus:%d" % gpus, "-device:%d" % device)
    assert rc == 0, err
    used = min(gpus, copies)          # a shard without streams is not started
    assert d["streams"] == copies and d["gpus"] == used and len(d["shards"]) == used
    for r, s in enumerate(d["shards"]):
        lo, hi = dist.shard_range(copies, r, used)
        assert (s["device"], s["lo"], s["n"]) == (device + r, lo, hi - lo)
    assert sum(s["n"] for s in d["shards"]) == copies
    assert d["sbr"] == 1 and d["channels"] == 1      # an HE-AACv2 stream: the probe of frame 0 ran (CPU front end)


def test_list_mode_split(tmp_path):
    names = ["mix_aot29_32k", "mix_aot29_32k", "mix_aot29_32k"]
    lst = tmp_path / "list.txt"
    lst.write_text("".join(os.path.join(ROOT, "tests", "golden", "streams", n + ".aac") + "\n" for n in names))
    rc, d, err = plan("-ilist:" + str(lst), "-odir:" + str(tmp_path), "-gpus:2")
    assert rc == 0, err
    assert [(s["lo"], s["n"]) for s in d["shards"]] == [(0, 2), (2, 1)]


@pytest.mark.parametrize("bad", [["-gpus:0"], ["-device:-1"], ["-copies:0"]])
def test_bad_arguments_are_usage_errors(bad):
    p = subprocess.run([CLI, "-plan", "-ifile:" + AAC] + bad, capture_output=True, text=True)
    assert p.returncode == 1 and "usage" in p.stderr


def test_more_devices_than_the_node_has_is_refused_before_any_work():
    """without -plan the device count is checked first: on a box without a GPU that is HIP's own error, on a GPU box
    '-gpus:64' names the devices it lacks; either way nothing is decoded and the exit code is 2"""
    p = subprocess.run([CLI, "-ifile:" + AAC, "-ofile:/dev/null", "-copies:64", "-gpus:64"], capture_output=True, text=True)
    assert p.returncode == 2 and p.stdout == ""
