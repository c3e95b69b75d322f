"""libxaac_amd/xaacdec_amd, the native command line decoder (host front end + GPU back end, HIP runtime only: no Python, no
torch, no reference code in the process), on the committed ADTS streams: its WAV payload must equal what the reference
decoder writes, with the same flags -- none (the reference's default -esbr:1: SBR streams through Path A) and -esbr:0 -- live
against `oracle/_ref/xaacdec` and against the committed CRCs of its output, also when it decodes a batch of copies."""
import json
import os
import subprocess
import wave
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STREAMS = os.path.join(ROOT, "tests", "golden", "streams")
CLI = os.path.join(ROOT, "libxaac_amd", "xaacdec_amd")
GOLD_ORDER = ["mix_aot2_64k", "mix_aot5_48k", "mono_aot5_32k", "harm_aot5_48k", "mix_aot29_32k", "synth_lc_a", "synth_lc_b",
              "synth_lc_mono", "lc_aot2_16k_mono", "he_aot5_44k"]      # tools/make_golden_parser.py NAMES
NAMES = GOLD_ORDER

pytestmark = pytest.mark.gpu


def run(name, tmp_path, *flags):
    assert os.path.exists(CLI), "libxaac_amd/xaacdec_amd is not built (make -C libxaac_amd/host)"
    out = str(tmp_path / (name + ".wav"))
    p = subprocess.run([CLI, "-ifile:" + os.path.join(STREAMS, name + ".aac"), "-ofile:" + out, *flags], capture_output=True,
                       text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-500:]
    with wave.open(out) as w:
        assert w.getsampwidth() == 2
        return w.readframes(w.getnframes()), w.getframerate(), json.loads(p.stdout.strip().splitlines()[-1]), w.getnchannels()


@pytest.mark.parametrize("flags", [(), ("-esbr:0",)], ids=["default", "esbr0"])
@pytest.mark.parametrize("name", NAMES)
def test_wav_equals_the_reference_decoders(name, flags, tmp_path):
    pcm, rate, info, channels = run(name, tmp_path, *flags)
    gold = np.load(os.path.join(ROOT, "tests", "golden", "decoder_ref.npz"))
    k = GOLD_ORDER.index(name)
    samples, crc = ("samples", "crc") if flags else ("samples_esbr", "crc_esbr")
    assert (len(pcm) // (2 * channels), rate) == (int(gold[samples][k]), int(gold["rate"][k]))
    assert zlib.crc32(pcm) & 0xffffffff == int(gold[crc][k])
    ref = os.path.join(ROOT, "oracle", "_ref", "xaacdec")
    if not os.path.exists(ref):
        pytest.fail("oracle/_ref/xaacdec missing: the reference binary did not travel with the snapshot")
    want = str(tmp_path / "ref.wav")
    subprocess.run([ref, "-ifile:" + os.path.join(STREAMS, name + ".aac"), "-ofile:" + want, *flags], check=True, capture_output=True)
    with wave.open(want) as w:
        assert w.getnchannels() == channels and w.readframes(w.getnframes()) == pcm


@pytest.mark.parametrize("flags", [(), ("-esbr:0",)], ids=["default", "esbr0"])
@pytest.mark.parametrize("name", ["mix_aot29_32k", "mix_aot5_48k", "mix_aot2_64k"])
def test_a_batch_of_copies(name, flags, tmp_path):
    one, _, _, _ = run(name, tmp_path, *flags)
    many, _, info, _ = run(name, tmp_path, "-copies:64", "-verify", *flags)
    assert many == one and info["streams"] == 64 and info["mismatched_copies"] == 0


@pytest.mark.parametrize("flags", [(), ("-esbr:0",)], ids=["default", "esbr0"])
@pytest.mark.parametrize("names", [("synth_lc_a", "mix_aot2_64k", "synth_lc_b"), ("mix_aot5_48k", "harm_aot5_48k"),
                                   ("mix_aot29_32k",)], ids=["lc", "he", "hev2"])
def test_a_list_of_different_streams_in_one_batch(names, flags, tmp_path):
    """-ilist: streams of one kind but different content and length decoded in lock step, each to its own WAV; a stream that
    ends drops out (the AAC-LC limiter's delay line is taken where the stream ends)"""
    lst = tmp_path / "list.txt"
    lst.write_text("\n".join(os.path.join(STREAMS, n + ".aac") for n in names) + "\n")
    out = tmp_path / "out"
    out.mkdir()
    p = subprocess.run([CLI, "-ilist:" + str(lst), "-odir:" + str(out), *flags], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-500:]
    gold = np.load(os.path.join(ROOT, "tests", "golden", "decoder_ref.npz"))
    for n in names:
        k = GOLD_ORDER.index(n)
        with wave.open(str(out / (n + ".wav"))) as w:
            pcm = w.readframes(w.getnframes())
            samples, crc = ("samples", "crc") if flags else ("samples_esbr", "crc_esbr")
            assert (w.getnframes(), w.getframerate()) == (int(gold[samples][k]), int(gold["rate"][k])), n
            assert zlib.crc32(pcm) & 0xffffffff == int(gold[crc][k]), n


@pytest.mark.gpu
def test_a_damaged_stream_of_a_list_ends_alone(tmp_path):
    """-ilist: one file with garbage behind its 20th frame -- that stream ends there (its 20 frames are written), the other
    streams of the batch come out as if it had not been in the list"""
    good = open(os.path.join(STREAMS, "mix_aot29_32k.aac"), "rb").read()
    pos = 0
    for _ in range(20):   # ADTS: the frame length is 13 bits from bit 30 of the header
        pos += ((good[pos + 3] & 3) << 11) | (good[pos + 4] << 3) | (good[pos + 5] >> 5)
    bad = tmp_path / "damaged.aac"
    bad.write_bytes(good[:pos] + bytes(range(7, 200)))
    lst = tmp_path / "list.txt"
    lst.write_text(os.path.join(STREAMS, "mix_aot29_32k.aac") + "\n" + str(bad) + "\n")
    out = tmp_path / "out"
    out.mkdir()
    p = subprocess.run([CLI, "-ilist:" + str(lst), "-odir:" + str(out), "-esbr:0"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-500:]
    assert "the stream ends here" in p.stderr
    gold = np.load(os.path.join(ROOT, "tests", "golden", "decoder_ref.npz"))
    k = GOLD_ORDER.index("mix_aot29_32k")
    with wave.open(str(out / "mix_aot29_32k.wav")) as w:
        whole = w.readframes(w.getnframes())
        assert w.getnframes() == int(gold["samples"][k]) and zlib.crc32(whole) & 0xffffffff == int(gold["crc"][k])
    with wave.open(str(out / "damaged.wav")) as w:
        part = w.readframes(w.getnframes())
        assert w.getnframes() == 20 * 2048 and part == whole[:len(part)]


def _adts_frames(data):
    pos, out = 0, []
    while pos + 7 <= len(data):
        n = ((data[pos + 3] & 3) << 11) | (data[pos + 4] << 3) | (data[pos + 5] >> 5)
        out.append(data[pos:pos + n])
        pos += n
    return out


@pytest.mark.parametrize("flags", [("-esbr:0",), ()], ids=["esbr0", "default"])
def test_a_batch_mixing_streams_with_and_without_parametric_stereo(flags, tmp_path):
    """-ilist (-esbr:0 and the default -esbr:1): an HE-AAC (mono) stream, an HE-AACv2 one and two streams whose parametric stereo starts at different
    frames (HE-AAC frames with HE-AACv2 frames behind them) decoded in lock step -- every step sees frames with and without PS.
    Each WAV must equal what the reference decoder writes for that stream alone (`oracle/_ref/xaacdec`, same flags)."""
    ref = os.path.join(ROOT, "oracle", "_ref", "xaacdec")
    if not os.path.exists(ref):
        pytest.fail("oracle/_ref/xaacdec missing: the reference binary did not travel with the snapshot")
    v1 = _adts_frames(open(os.path.join(STREAMS, "mono_aot5_32k.aac"), "rb").read())
    v2 = _adts_frames(open(os.path.join(STREAMS, "mix_aot29_32k.aac"), "rb").read())
    files = {"mono_aot5_32k": b"".join(v1), "mix_aot29_32k": b"".join(v2), "late_ps_a": b"".join(v1[:7] + v2),
             "late_ps_b": b"".join(v1[:16] + v2[:20])}
    lst = tmp_path / "list.txt"
    for n, d in files.items():
        (tmp_path / (n + ".aac")).write_bytes(d)
    lst.write_text("\n".join(str(tmp_path / (n + ".aac")) for n in files) + "\n")
    out = tmp_path / "out"
    out.mkdir()
    p = subprocess.run([CLI, "-ilist:" + str(lst), "-odir:" + str(out), *flags], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-500:]
    for n in files:
        want = str(tmp_path / (n + "_ref.wav"))
        subprocess.run([ref, "-ifile:" + str(tmp_path / (n + ".aac")), "-ofile:" + want, *flags], check=True, capture_output=True)
        with wave.open(want) as a, wave.open(str(out / (n + ".wav"))) as b:
            assert (a.getnchannels(), a.getframerate(), a.getnframes()) == (b.getnchannels(), b.getframerate(), b.getnframes()), n
            assert a.readframes(a.getnframes()) == b.readframes(b.getnframes()), n


@pytest.mark.parametrize("flags", [(), ("-esbr:0",)], ids=["default", "esbr0"])
@pytest.mark.parametrize("name", ["mix_aot29_32k", "mix_aot5_48k", "mix_aot2_64k"])
def test_gpus_flag_shards_the_batch(name, flags, tmp_path):
    """-gpus:G / -device:k (the native multi-GPU host): with one device the output is today's; the sharded host path -- one
    thread, HIP context, stream and set of resident states per shard, the parser team shared -- runs here with three shards
    wrapped onto the box's one device (-wrap_devices) and must write the same PCM for every copy of every shard; asking for
    devices the node lacks is refused before any work"""
    one, _, _, _ = run(name, tmp_path, *flags)
    same, _, info, _ = run(name, tmp_path, "-gpus:1", "-device:0", "-copies:7", "-verify", *flags)
    assert same == one and info["gpus"] == 1 and info["mismatched_copies"] == 0
    many, _, info, _ = run(name, tmp_path, "-gpus:3", "-wrap_devices", "-copies:64", "-verify", *flags)
    assert many == one and info["gpus"] == 3 and info["streams"] == 64 and info["mismatched_copies"] == 0
    p = subprocess.run([CLI, "-ifile:" + os.path.join(STREAMS, name + ".aac"), "-ofile:" + str(tmp_path / "x.wav"), "-copies:64",
                        "-gpus:64", *flags], capture_output=True, text=True, timeout=300)
    assert p.returncode == 2 and "asks for devices" in p.stderr


def test_gpus_flag_with_a_list(tmp_path):
    """-ilist + -gpus: every stream's WAV comes out of the shard that owns it"""
    names = ("mix_aot5_48k", "harm_aot5_48k", "mix_aot5_48k")
    src = []
    for i, n in enumerate(names):             # distinct file names for the same stream
        dst = tmp_path / ("s%d_%s.aac" % (i, n))
        dst.write_bytes(open(os.path.join(STREAMS, n + ".aac"), "rb").read())
        src.append(str(dst))
    lst = tmp_path / "list.txt"
    lst.write_text("\n".join(src) + "\n")
    out = tmp_path / "out"
    out.mkdir()
    p = subprocess.run([CLI, "-ilist:" + str(lst), "-odir:" + str(out), "-gpus:2", "-wrap_devices", "-esbr:0"], capture_output=True,
                       text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-500:]
    gold = np.load(os.path.join(ROOT, "tests", "golden", "decoder_ref.npz"))
    for i, n in enumerate(names):
        k = GOLD_ORDER.index(n)
        with wave.open(str(out / ("s%d_%s.wav" % (i, n)))) as w:
            assert zlib.crc32(w.readframes(w.getnframes())) & 0xffffffff == int(gold["crc"][k]), n   # ("crc": the -esbr:0 output)
