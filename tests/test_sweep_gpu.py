"""A parity sweep beyond the committed streams (tools/sweep_streams.py): the reference's encoder (oracle/_ref/xaacenc) makes
ADTS streams at six sampling rates (16 .. 48 kHz), two bit rates, mono and stereo, AAC-LC / HE-AAC / HE-AACv2; each is decoded
by the reference's decoder and by the repo's native one (own front end + GPU), with the reference's default flags and with
-esbr:0, and the WAV payloads must be identical; plus digital silence, full-scale noise and a click train at 44.1 kHz:
126 decodes (24 of them of streams with ENHSBR elements: harmonic patching, pre-flattening, inter-TES)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_streams_made_on_the_spot_decode_like_the_reference(tmp_path):
    for exe in ("oracle/_ref/xaacenc", "oracle/_ref/xaacdec", "libxaac_amd/xaacdec_amd"):
        if not os.path.exists(os.path.join(ROOT, exe)):
            pytest.fail(exe + " missing: it did not travel with the snapshot / was not built")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sweep_streams.py")], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, SWEEP_TMP=str(tmp_path)))
    lines = p.stdout.strip().splitlines()
    assert lines and lines[-1].startswith("cases "), p.stdout[-600:] + p.stderr[-600:]
    total, bad = int(lines[-1].split()[1]), int(lines[-1].split()[3])
    assert total >= 120 and bad == 0 and p.returncode == 0, "\n".join(l for l in lines if "identical" not in l)


def test_spliced_streams_with_sbr_header_changes_in_the_middle(tmp_path):
    """tools/splice_check.py: streams of different bit rates (different SBR ranges: a new SBR header, i.e. a reset of the SBR
    decoder, at every splice) joined frame-wise, with and without ENHSBR elements, and mono HE-AAC joined with HE-AACv2 (parametric stereo starting and stopping); the native decoder and decode_streams
    against the reference with both flag settings (Path A: the reset-time transposer runs on device rows)."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "splice_check.py")], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, SWEEP_TMP=str(tmp_path)))
    lines = p.stdout.strip().splitlines()
    assert lines and lines[-1] == "bad 0" and sum("identical" in l for l in lines) == 16, p.stdout[-1200:] + p.stderr[-600:]


def test_a_batch_of_streams_whose_sbr_headers_change_at_different_frames(tmp_path):
    """tools/splice_list_check.py: five files per kind (HE-AAC stereo, mono, HE-AACv2, HE-AAC with ENHSBR elements), parts of
    different bit rates joined at different frames, decoded as one -ilist batch with -esbr:0 and with the default flags: most
    steps see no reset of the SBR decoder, some see one or two streams reset while the others go on (Path A: the reset-time
    transposer runs on those streams' rows alone); a part that starts a file brings frames without SBR processing whose
    header still holds the parser's defaults.  Every WAV equals the reference decoder's for that file alone."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "splice_list_check.py")], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, SWEEP_TMP=str(tmp_path)))
    lines = p.stdout.strip().splitlines()
    assert lines and lines[-1] == "bad 0" and sum("identical" in l for l in lines) == 40, p.stdout[-1500:] + p.stderr[-600:]


def test_the_sweep_through_the_drop_in(tmp_path):
    """The same streams (and SBR at the three low core rates too) through the reference's own decoder with its seams served by
    the library (oracle/_ref/xaacdec_dropin), with the default flags, -esbr:0, -dsample:1 and -dsample:1 -esbr:0, at three of the six sampling rates (the whole sweep: profiles/r06_k_dropin_sweep.txt): 192 decodes of 0.8 s streams.
    (This is the sweep that found the reset-time rows the drop-in's eSBR seam had left stale: a 32 kHz stream with ENHSBR
    elements whose first SBR header arrives behind nine frames of audio.)"""
    for exe in ("oracle/_ref/xaacenc", "oracle/_ref/xaacdec", "oracle/_ref/xaacdec_dropin"):
        if not os.path.exists(os.path.join(ROOT, exe)):
            pytest.fail(exe + " missing: it did not travel with the snapshot / was not built")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sweep_streams.py")], capture_output=True, text=True, timeout=1500,
                       env=dict(os.environ, SWEEP_TMP=str(tmp_path), SWEEP_DECODER="dropin", SWEEP_ALL_SBR_RATES="1", SWEEP_SECONDS="0.8", SWEEP_RATES="22050,32000,48000"))
    lines = p.stdout.strip().splitlines()
    assert lines and lines[-1].startswith("cases "), p.stdout[-600:] + p.stderr[-600:]
    total, bad = int(lines[-1].split()[1]), int(lines[-1].split()[3])
    assert total >= 180 and bad == 0 and p.returncode == 0, "\n".join(l for l in lines if "identical" not in l)


def test_the_sweep_with_the_dft_transposer(tmp_path):
    """-esbr_hq:1 through the drop-in (ixheaacd_dft_hbe_apply on the GPU) over the sweep's HE-AAC / HE-AACv2 streams at all six
    sampling rates, with and without ENHSBR elements: every sample within 1 LSB of the reference decoder's (the float tolerance of
    BASELINE.json's north_star) wherever the reference decodes the stream with this flag: 50 of 66, 45 of them identical when this was written."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sweep_streams.py")], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, SWEEP_TMP=str(tmp_path), SWEEP_DECODER="dropin", SWEEP_ALL_SBR_RATES="1", SWEEP_SECONDS="0.8",
                                SWEEP_FLAGS="-esbr_hq:1", SWEEP_TOL="1"))
    lines = p.stdout.strip().splitlines()
    assert lines and lines[-1].startswith("cases "), p.stdout[-600:] + p.stderr[-600:]
    total, bad = int(lines[-1].split()[1]), int(lines[-1].split()[3])
    assert total >= 60 and bad == 0 and p.returncode == 0, "\n".join(l for l in lines if "identical" not in l)
    assert sum("identical" in l and "within" not in l for l in lines) >= 30


def test_usac_streams_made_on_the_spot_through_the_drop_in(tmp_path):
    """tools/sweep_usac.py at one sampling rate (the three-rate run: profiles/r06_s_usac_sweep.txt, 168 streams): the reference
    encoder's USAC modes -- FD / switched / TD cores, 2:1, 8:3 and 4:1 SBR and none, harmonic SBR, PVC, inter-TES, complex
    prediction, noise filling -- mono and stereo at two bit rates, decoded by the reference decoder with ixheaacd_fd_frm_dec and
    ixheaacd_sbr_dec served by the library: byte-identical, and no SBR call left to the reference."""
    for exe in ("oracle/_ref/xaacenc", "oracle/_ref/xaacdec", "oracle/_ref/xaacdec_dropin"):
        if not os.path.exists(os.path.join(ROOT, exe)):
            pytest.fail(exe + " missing: it did not travel with the snapshot / was not built")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sweep_usac.py")], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, SWEEP_TMP=str(tmp_path), SWEEP_SECONDS="0.8", SWEEP_RATES="44100"))
    lines = p.stdout.strip().splitlines()
    assert lines and lines[-1].startswith("cases "), p.stdout[-600:] + p.stderr[-600:]
    w = lines[-1].split()
    total, bad, gpu_calls = int(w[1]), int(w[3]), int(w[7])
    assert total >= 50 and bad == 0 and gpu_calls > 2000 and p.returncode == 0, "\n".join(l for l in lines if "identical" not in l)
    assert all(l.endswith("reference 0") for l in lines[:-1] if "identical" in l)


def test_the_repos_own_decoder_with_the_dft_transposer(tmp_path):
    """xaacdec_amd -esbr_hq:1 (own front end, xaac_hbe_dft_state_reinit on the host, the reset-time runs and the frames' chain on
    the GPU) over the sweep's HE-AAC / HE-AACv2 streams: within 1 LSB of the reference decoder wherever that one decodes the
    stream with this flag (it writes nothing for transposer sizes it has no transforms for: a fifth of the sweep)."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sweep_streams.py")], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, SWEEP_TMP=str(tmp_path), SWEEP_ALL_SBR_RATES="1", SWEEP_SECONDS="0.8", SWEEP_FLAGS="-esbr_hq:1", SWEEP_TOL="1"))
    lines = p.stdout.strip().splitlines()
    assert lines and lines[-1].startswith("cases "), p.stdout[-600:] + p.stderr[-600:]
    total, bad = int(lines[-1].split()[1]), int(lines[-1].split()[3])
    compared = sum("identical" in l for l in lines)
    assert total >= 60 and bad == 0 and compared >= 40 and p.returncode == 0, "\n".join(l for l in lines if "identical" not in l)
