"""Whole streams through the repo's own decoder -- the host front end of libxaac_amd/host (ADTS, AAC-LC syntax, SBR / PS side
info: no reference code in the process) feeding the GPU entry points -- against the unmodified reference decoder
(oracle/_ref/xaacdec -esbr:0) on the committed ADTS streams: the PCM must be identical sample for sample.  The same
comparison against committed CRCs of the reference's output (tests/golden/decoder_ref.npz, tools/make_golden_parser.py)."""
import os
import subprocess
import wave
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STREAMS = os.path.join(ROOT, "tests", "golden", "streams")
XAACDEC = os.path.join(ROOT, "oracle", "_ref", "xaacdec")
NAMES = ["mix_aot2_64k", "mix_aot5_48k", "mono_aot5_32k", "harm_aot5_48k", "mix_aot29_32k", "synth_lc_a", "synth_lc_b", "synth_lc_mono", "lc_aot2_16k_mono", "he_aot5_44k"]
GOLD_ORDER = ["mix_aot2_64k", "mix_aot5_48k", "mono_aot5_32k", "harm_aot5_48k", "mix_aot29_32k", "synth_lc_a", "synth_lc_b",
              "synth_lc_mono", "lc_aot2_16k_mono", "he_aot5_44k"]      # tools/make_golden_parser.py NAMES

pytestmark = pytest.mark.gpu


def reference_pcm(name, tmp_path, flags=("-esbr:0",)):
    if not os.path.exists(XAACDEC):
        pytest.fail("oracle/_ref/xaacdec missing: run __graft_entry__.build() where /root/reference exists")
    out = str(tmp_path / (name + ".wav"))
    subprocess.run([XAACDEC, "-ifile:" + os.path.join(STREAMS, name + ".aac"), "-ofile:" + out] + list(flags), check=True,
                   capture_output=True)
    with wave.open(out) as w:
        assert w.getsampwidth() == 2
        return np.frombuffer(w.readframes(w.getnframes()), np.int16).reshape(-1, w.getnchannels()), w.getframerate()


@pytest.mark.parametrize("name", NAMES)
def test_stream_equals_reference_decoder(name, tmp_path):
    from libxaac_amd import decoder
    want, rate = reference_pcm(name, tmp_path)
    data = open(os.path.join(STREAMS, name + ".aac"), "rb").read()
    got, got_rate = decoder.decode_streams([data])
    assert got_rate == rate
    assert got[0].shape == want.shape, (got[0].shape, want.shape)
    bad = np.nonzero(np.any(got[0] != want, axis=1))[0]
    assert bad.size == 0, "first differing sample %d of %d" % (bad[0], len(want))


@pytest.mark.parametrize("name", NAMES)
def test_stream_equals_reference_decoder_with_its_default_flags(name, tmp_path):
    """-esbr:1, the reference's default: SBR streams through Path A (float eSBR tools, QMF transposer, float PS)"""
    from libxaac_amd import decoder
    want, rate = reference_pcm(name, tmp_path, flags=())
    data = open(os.path.join(STREAMS, name + ".aac"), "rb").read()
    got, got_rate = decoder.decode_streams([data], esbr=True)
    assert got_rate == rate
    assert got[0].shape == want.shape, (got[0].shape, want.shape)
    bad = np.nonzero(np.any(got[0] != want, axis=1))[0]
    assert bad.size == 0, "first differing sample %d of %d" % (bad[0], len(want))


def test_esbr_streams_against_committed_crcs_and_in_batches():
    from libxaac_amd import decoder
    gold = np.load(os.path.join(ROOT, "tests", "golden", "decoder_ref.npz"))
    for name in ("mix_aot5_48k", "mono_aot5_32k", "harm_aot5_48k", "mix_aot29_32k", "he_aot5_44k"):
        k = GOLD_ORDER.index(name)
        data = open(os.path.join(STREAMS, name + ".aac"), "rb").read()
        got, rate = decoder.decode_streams([data] * 3, esbr=True)
        for g in got:
            assert (len(g), rate) == (int(gold["samples_esbr"][k]), int(gold["rate"][k])), name
            assert zlib.crc32(np.ascontiguousarray(g).tobytes()) & 0xffffffff == int(gold["crc_esbr"][k]), name


def test_streams_against_committed_crcs():
    from libxaac_amd import decoder
    gold = np.load(os.path.join(ROOT, "tests", "golden", "decoder_ref.npz"))
    for name in NAMES:
        k = GOLD_ORDER.index(name)
        data = open(os.path.join(STREAMS, name + ".aac"), "rb").read()
        got, rate = decoder.decode_streams([data])
        assert (len(got[0]), rate) == (int(gold["samples"][k]), int(gold["rate"][k])), name
        assert zlib.crc32(np.ascontiguousarray(got[0]).tobytes()) & 0xffffffff == int(gold["crc"][k]), name


def test_a_batch_of_streams_decodes_like_each_alone():
    """lock-step batches: N copies of a stream (and the states of N streams side by side on the device) give N times the PCM"""
    from libxaac_amd import decoder
    for name in ("mix_aot5_48k", "mix_aot29_32k", "mix_aot2_64k"):
        data = open(os.path.join(STREAMS, name + ".aac"), "rb").read()
        one, _ = decoder.decode_streams([data])
        many, _ = decoder.decode_streams([data] * 5)
        for m in many:
            assert np.array_equal(m, one[0]), name


def test_streams_of_different_lengths_share_a_batch():
    """a stream that ends early drops out of the steps; the others go on"""
    from libxaac_amd import decoder
    lib = decoder.load_host_library()
    import ctypes
    for name, per_frame in (("mix_aot5_48k", 2048), ("mix_aot2_64k", 1024)):
        data = open(os.path.join(STREAMS, name + ".aac"), "rb").read()
        hdr, pos, cuts = decoder.AdtsHeader(), 0, []
        while pos + 7 < len(data) and lib.xaac_adts_parse_header(data[pos:pos + 16], 16, ctypes.byref(hdr)) == 0:
            pos += hdr.frame_bytes
            cuts.append(pos)
        full, _ = decoder.decode_streams([data])
        got, _ = decoder.decode_streams([data, data[:cuts[9]], data[:cuts[20]]])
        assert np.array_equal(got[0], full[0])
        if per_frame == 2048:
            assert np.array_equal(got[1], full[0][:10 * 2048]) and np.array_equal(got[2], full[0][:21 * 2048])
        else:   # AAC-LC: the limiter's delay line is flushed behind the stream's last frame: a cut stream ends like the cut
            alone_1, _ = decoder.decode_streams([data[:cuts[9]]])   # stream decoded alone
            alone_2, _ = decoder.decode_streams([data[:cuts[20]]])
            assert np.array_equal(got[1], alone_1[0]) and np.array_equal(got[2], alone_2[0]) and len(got[1]) == 10 * 1024


@pytest.mark.gpu
@pytest.mark.parametrize("esbr", [False, True], ids=["esbr0", "default"])
def test_the_pipelined_loop_equals_the_plain_one(esbr):
    """overlap=True (the parser library's own threads parse step k + 1, copies up on their own stream into two sets of device
    inputs, copy down of step k - 1 beside both) against overlap=False (parse, copy, run, copy, one after the other) on batches
    with streams of different lengths, and a thread count that does not divide the batch"""
    from libxaac_amd import decoder
    for name in ("mix_aot29_32k", "mix_aot5_48k", "mix_aot2_64k"):
        data = open(os.path.join(STREAMS, name + ".aac"), "rb").read()
        lens, pos = [], 0
        while pos + 7 <= len(data):
            n = ((data[pos + 3] & 3) << 11) | (data[pos + 4] << 3) | (data[pos + 5] >> 5)
            lens.append(n)
            pos += n
        batch = [data, data[:sum(lens[:12])], data, data[:sum(lens[:3])], data]
        plain, rate_a = decoder.decode_streams(batch, overlap=False, esbr=esbr, threads=3)
        piped, rate_b = decoder.decode_streams(batch, overlap=True, esbr=esbr, threads=3)
        assert rate_a == rate_b
        for a, b in zip(plain, piped):
            assert np.array_equal(a, b), name
        assert len(plain[1]) < len(plain[0]) and len(plain[3]) < len(plain[1])


@pytest.mark.gpu
def test_adts_frames_with_several_raw_data_blocks_decode_like_the_plain_stream():
    """number_of_raw_data_blocks_in_frame > 0 end to end: the committed streams regrouped into ADTS frames of 1..4 blocks
    (tests/test_parser.py: the parser delivers one block per call; the reference decoder writes the same file for both
    layouts) next to their originals in one batch, both flag settings, PCM against the committed CRCs of the reference"""
    from libxaac_amd import decoder
    from test_parser import _remux_several_blocks_per_frame
    gold = np.load(os.path.join(ROOT, "tests", "golden", "decoder_ref.npz"))
    for name in ("mix_aot2_64k", "mix_aot5_48k", "mix_aot29_32k"):
        k = GOLD_ORDER.index(name)
        data = open(os.path.join(STREAMS, name + ".aac"), "rb").read()
        remux, prot = _remux_several_blocks_per_frame(data), _remux_several_blocks_per_frame(data, groups=(3, 4), protected=True)
        for esbr, key in ((False, "crc"), (True, "crc_esbr")):
            if esbr and name == "mix_aot2_64k":
                continue                       # (AAC-LC has no second reading)
            got, _ = decoder.decode_streams([remux, data, prot], esbr=esbr)
            for g in got:
                assert zlib.crc32(np.ascontiguousarray(g).tobytes()) & 0xffffffff == int(gold[key][k]), (name, esbr)
