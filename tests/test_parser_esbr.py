"""The host front end's -esbr:1 mode (xaac_parser_set_esbr, include/xaac_parse.h): the reference's default reading of the SBR
payload of HE-AAC / HE-AACv2 streams ("Path A": payload one frame late, ENHSBR extension, float scale factors), against
committed CRCs of what the REAL reference decoder holds at every ixheaacd_sbr_dec call when run with its default flags
(tests/golden/parser_ref.npz: *_esbr / *_hbe, made by tools/make_golden_parser.py through XAAC_ESBR_SIDE_FILE of
oracle/ref_capture.c): header tables, frame data, the xaac_esbr_side members, the PS frame, and the QMF transposer's
parameters the host derives at a reset (xaac_hbe_state_reinit).  Plus the new-stream values of the Path A states against a
capture of what the reference's first calls find, where oracle/_ref exists.  CPU only."""
import ctypes
import os
import subprocess
import sys
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from libxaac_amd import decoder, ESBR_PS_STATE_BYTES, ESBR_STATE_BYTES, HBE_STATE_BYTES  # noqa: E402
from esbr_structs import EsbrPsState, EsbrState  # noqa: E402
from hbe_structs import HbeState  # noqa: E402

STREAMS = os.path.join(ROOT, "tests", "golden", "streams")
NAMES = ["mix_aot5_48k", "mono_aot5_32k", "harm_aot5_48k", "mix_aot29_32k", "he_aot5_44k"]
CAPTURE = os.path.join(ROOT, "oracle", "_ref", "xaacdec_capture")


def crc(b):
    return zlib.crc32(bytes(b)) & 0xffffffff


def stream(name):
    return open(os.path.join(STREAMS, name + ".aac"), "rb").read()


def hbe_params(st):
    return [st.synth_size, st.k_start, st.start_band, st.end_band] + list(st.x_over_qmf) + [st.max_stretch]


@pytest.mark.parametrize("name", NAMES)
def test_side_info_in_esbr_mode_equals_the_references(name):
    lib = decoder.load_host_library()
    gold = np.load(os.path.join(ROOT, "tests", "golden", "parser_ref.npz"))
    ref, ref_hbe = gold[name + "_esbr"], gold[name + "_hbe"]
    frames = decoder.parse_stream(stream(name), esbr=True)
    assert len(frames) == ref.shape[0]
    n_ch = ref.shape[1]
    hbe = [HbeState(), HbeState()]
    for st in hbe:
        lib.xaac_hbe_state_init(ctypes.byref(st))
    applied = ps_frames = 0
    for f, (_, _, _, side, eside) in enumerate(frames):
        assert len(eside) == n_ch
        if side.reset:   # ixheaacd_sbr_dec_reset: the transposer's parameters follow the new band tables
            for c in range(side.reset_channels):
                assert lib.xaac_hbe_state_reinit(ctypes.byref(hbe[c]), ctypes.byref(side, decoder.SbrSide.header.offset)) == 0
        for c in range(n_ch):
            assert crc(side.header) == ref[f, c, 0], "frame %d: SBR header tables" % f
            assert crc(side.frame[c]) == ref[f, c, 1], "frame %d channel %d: SBR frame data" % (f, c)
            assert crc(eside[c]) == ref[f, c, 2], "frame %d channel %d: eSBR side info" % (f, c)
            if ref[f, c, 3]:
                assert side.ps and crc(side.ps_frame) == ref[f, c, 3], "frame %d: PS frame" % f
                ps_frames += 1
            else:
                assert not side.ps
            assert hbe_params(hbe[c]) == ref_hbe[f, c].tolist(), "frame %d channel %d: transposer parameters" % (f, c)
        applied += side.apply
    assert frames[0][3].apply == 0 and applied == len(frames) - 1     # the payload runs one frame late: frame 0 has none
    assert (ps_frames > 0) == (name == "mix_aot29_32k")


def test_esbr_mode_is_a_different_reading_of_the_same_payload():
    a = decoder.parse_stream(stream("mix_aot5_48k"))
    b = decoder.parse_stream(stream("mix_aot5_48k"), esbr=True)
    assert a[0][3].apply == 1 and b[0][3].apply == 0
    # frame k of the -esbr:0 reading and frame k + 1 of the -esbr:1 one come from the same payload: the same grid
    from sbr_capture import Frame
    fa = Frame.from_buffer_copy(bytes(a[3][3].frame[0]))
    fb = Frame.from_buffer_copy(bytes(b[4][3].frame[0]))
    assert fa.num_env == fb.num_env and list(fa.border_vec) == list(fb.border_vec)


def test_new_stream_states():
    lib = decoder.load_host_library()
    es, ps, hb = EsbrState(), EsbrPsState(), HbeState()
    assert (ctypes.sizeof(es), ctypes.sizeof(ps), ctypes.sizeof(hb)) == (ESBR_STATE_BYTES, ESBR_PS_STATE_BYTES, HBE_STATE_BYTES)
    for st in (es, ps, hb):
        ctypes.memset(ctypes.byref(st), 0xa5, ctypes.sizeof(st))
    lib.xaac_esbr_state_init(ctypes.byref(es))
    lib.xaac_esbr_ps_state_init(ctypes.byref(ps))
    lib.xaac_hbe_state_init(ctypes.byref(hb))
    assert es.esbr_start_up == 1
    es.esbr_start_up = 0
    assert not any(bytes(es)) and not any(bytes(hb))
    h = np.ctypeslib.as_array(ps.h_prev).copy()
    assert np.all(h[:2] == 1.0) and not h[2:].any()
    ctypes.memset(ctypes.byref(ps, EsbrPsState.h_prev.offset), 0, 2 * 20 * 4)
    assert not any(bytes(ps))


@pytest.mark.skipif(not os.path.exists(CAPTURE), reason="oracle/_ref/xaacdec_capture is not built")
@pytest.mark.parametrize("name", ["mix_aot5_48k", "mix_aot29_32k"])
def test_new_stream_states_equal_what_the_references_first_calls_find(name, tmp_path):
    lib = decoder.load_host_library()
    side, init = str(tmp_path / "side.bin"), str(tmp_path / "init.bin")
    subprocess.run([CAPTURE, "-ifile:" + os.path.join(STREAMS, name + ".aac"), "-ofile:" + str(tmp_path / "o.wav")],
                   env=dict(os.environ, XAAC_ESBR_SIDE_FILE=side, XAAC_ESBR_INIT_FILE=init), check=True, capture_output=True)
    raw = open(init, "rb").read()
    rs = ESBR_STATE_BYTES + HBE_STATE_BYTES + ESBR_PS_STATE_BYTES
    assert len(raw) == 2 * rs
    es, ps, hb = EsbrState(), EsbrPsState(), HbeState()
    lib.xaac_esbr_state_init(ctypes.byref(es))
    lib.xaac_esbr_ps_state_init(ctypes.byref(ps))
    lib.xaac_hbe_state_init(ctypes.byref(hb))
    mono = name == "mix_aot29_32k"
    for call in range(2):    # the initialisation pass over frame 0, then (mono) frame 0 again after the re-initialisation / (pair) channel 1
        o = call * rs
        assert raw[o:o + ESBR_STATE_BYTES] == bytes(es), "call %d: xaac_esbr_state" % call
        assert raw[o + ESBR_STATE_BYTES:o + ESBR_STATE_BYTES + HBE_STATE_BYTES] == bytes(hb), "call %d: xaac_hbe_state" % call
        if mono:
            assert raw[o + ESBR_STATE_BYTES + HBE_STATE_BYTES:o + rs] == bytes(ps), "call %d: xaac_esbr_ps_state" % call
