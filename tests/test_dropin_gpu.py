"""The drop-in, end to end: the REAL reference decoder decodes whole .aac streams with its frame-level seams
(ixheaacd_imdct_process, ixheaacd_sbr_dec, ixheaacd_peak_limiter_process) diverted to libxaac_amd on the GPU (oracle/_ref/xaacdec_dropin, built
from oracle/ref_dropin.c by oracle/Makefile.ref); the output file must be byte-identical to what the unmodified
reference decoder (oracle/_ref/xaacdec) writes.  Needs the prebuilt oracle/_ref binaries next to the repo."""
import glob
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
STREAMS = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "streams", "*.aac")))


def _decode(binary, aac, out, extra=("-esbr:0",), env=None):
    p = subprocess.run([os.path.join(REF, binary), "-ifile:" + aac, "-ofile:" + out, *extra], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    return p.stderr.decode(errors="replace")


@pytest.mark.parametrize("aac", STREAMS, ids=[os.path.basename(s) for s in STREAMS])
def test_reference_decoder_with_gpu_back_end_is_byte_identical(aac, tmp_path):
    if not (os.path.exists(os.path.join(REF, "xaacdec")) and os.path.exists(os.path.join(REF, "xaacdec_dropin"))):
        pytest.fail("oracle/_ref/xaacdec[_dropin] missing: the reference binaries (built by oracle/Makefile.ref where /root/reference exists, git-ignored) did not travel with the snapshot -- the drop-in evidence must not vanish silently")
    ref_wav, gpu_wav = str(tmp_path / "ref.wav"), str(tmp_path / "gpu.wav")
    _decode("xaacdec", aac, ref_wav)
    log = _decode("xaacdec_dropin", aac, gpu_wav)
    m = re.search(r"(\d+) imdct_process and (\d+) sbr_dec calls ran on the GPU", log)
    assert m, log[-400:]
    n_imdct, n_sbr = int(m.group(1)), int(m.group(2))
    assert n_imdct > 30
    if "aot2_" not in aac and "synth_lc" not in aac:   # AAC-LC streams have no SBR calls
        assert n_sbr > 30
    else:   # plain AAC-LC with default flags: the peak limiter is on (api.c:3663-3669) and ran on the GPU too
        m = re.search(r"(\d+) peak_limiter_process calls ran on the GPU", log)
        assert m and int(m.group(1)) > 30, log[-400:]
    a, b = open(ref_wav, "rb").read(), open(gpu_wav, "rb").read()
    assert len(a) > 100000 and a == b, (len(a), len(b), n_imdct, n_sbr)


@pytest.mark.parametrize("aac", [s for s in STREAMS if "aot5_" in s or "aot29_" in s], ids=lambda s: os.path.basename(s))
def test_default_flags_he_aac_takes_the_esbr_path_on_the_gpu(aac, tmp_path):
    """HE-AAC v1 and v2 with the reference's DEFAULT flags (-esbr:1): ixheaacd_sbr_dec's Path A branch -- 32-bit analysis
    bank, float HF generator and envelope adjuster, [float parametric stereo,] 64-band synthesis bank(s) -- runs on the GPU
    (xaac_esbr_sbr_process_batch) and the decoded file is byte-identical to the unmodified reference decoder's."""
    if not (os.path.exists(os.path.join(REF, "xaacdec")) and os.path.exists(os.path.join(REF, "xaacdec_dropin"))):
        pytest.fail("oracle/_ref/xaacdec[_dropin] missing: the reference binaries (built by oracle/Makefile.ref where /root/reference exists, git-ignored) did not travel with the snapshot -- the drop-in evidence must not vanish silently")
    ref_wav, gpu_wav = str(tmp_path / "ref.wav"), str(tmp_path / "gpu.wav")
    _decode("xaacdec", aac, ref_wav, extra=())
    log = _decode("xaacdec_dropin", aac, gpu_wav, extra=())
    m = re.search(r"(\d+) sbr_dec calls took the eSBR \(Path A\) branch on the GPU", log)
    assert m and int(m.group(1)) > 30, log[-600:]
    a, b = open(ref_wav, "rb").read(), open(gpu_wav, "rb").read()
    assert len(a) > 100000 and a == b, (len(a), len(b), int(m.group(1)))
    mh = re.search(r"(\d+) of them with harmonic patching", log)
    if "harm_" in aac:   # encoded with -harmonic_sbr:1: sbr_patching_mode 0 frames, the HF generator fed by the QMF transposer
        assert mh and int(mh.group(1)) > 30, log[-600:]
    # and it is a different decode from the fixed-point path the other tests pin
    fix_wav = str(tmp_path / "fix.wav")
    _decode("xaacdec", aac, fix_wav)
    assert open(fix_wav, "rb").read() != a


@pytest.mark.parametrize("aac", [s for s in STREAMS if "aot5_" in s or "aot29_" in s], ids=lambda s: os.path.basename(s))
def test_down_sampled_sbr_through_the_gpu(aac, tmp_path):
    """-dsample:1 -esbr:0: the 32-channel synthesis bank (1024 samples a frame out; sbrdec_initfuncs.c:622) behind the fixed-point SBR
    chain -- xaac_sbr_lp / _hq_process_batch with down_sample -- byte-identical; HE-AACv2 streams, where the reference itself gives
    the right bank half a slot with this flag (qmf_dec.c:1117), are left to it and counted so.  (A survey of decoder flags late in
    round 6 found this combination decoded by the 64-channel bank in the drop-in: the hook had not looked at the bank's size.)"""
    if not (os.path.exists(os.path.join(REF, "xaacdec")) and os.path.exists(os.path.join(REF, "xaacdec_dropin"))):
        pytest.fail("oracle/_ref/xaacdec[_dropin] missing: the reference binaries did not travel with the snapshot")
    ref_wav, gpu_wav = str(tmp_path / "ref.wav"), str(tmp_path / "gpu.wav")
    extra = ("-esbr:0", "-dsample:1")
    _decode("xaacdec", aac, ref_wav, extra=extra)
    log = _decode("xaacdec_dropin", aac, gpu_wav, extra=extra)
    n_sbr = int(re.search(r"and (\d+) sbr_dec calls ran on the GPU", log).group(1))
    n_ds = int(re.search(r"(\d+) of the sbr_dec calls with the down-sampled synthesis bank", log).group(1))
    n_left = int(re.search(r"(\d+) sbr_dec calls left to the reference", log).group(1))
    if "aot29_" in aac:
        assert n_sbr == 0 and n_left > 30, log[-600:]
    else:
        assert n_sbr == n_ds and n_ds > 30 and n_left == 0, log[-600:]
    a, b = open(ref_wav, "rb").read(), open(gpu_wav, "rb").read()
    assert len(a) > 50000 and a == b, (len(a), len(b))
    full = str(tmp_path / "full.wav")
    _decode("xaacdec", aac, full)
    assert len(open(full, "rb").read()) > 1.8 * (len(a) - 44)      # half the samples of the plain decode


@pytest.mark.parametrize("aac", [s for s in STREAMS if "aot5_" in s or "aot29_" in s], ids=lambda s: os.path.basename(s))
def test_down_sampled_esbr_through_the_gpu(aac, tmp_path):
    """-dsample:1 with the reference's default flags: the eSBR branch with its 32-channel synthesis bank(s) (sbr_dec.c:605-628;
    xaac_esbr_sbr_batch.down_sample -> xaac_esbr_synthesis_ds_kernel), HE-AAC and HE-AACv2 (both banks): every call on the GPU,
    byte-identical (the reference's USAC decoder does not take the flag)"""
    if not (os.path.exists(os.path.join(REF, "xaacdec")) and os.path.exists(os.path.join(REF, "xaacdec_dropin"))):
        pytest.fail("oracle/_ref/xaacdec[_dropin] missing: the reference binaries did not travel with the snapshot")
    meta = aac[:-4] + ".txt"
    extra = ("-dsample:1",) + (("-mp4:1", "-imeta:" + meta) if os.path.exists(meta) else ())
    ref_wav, gpu_wav = str(tmp_path / "ref.wav"), str(tmp_path / "gpu.wav")
    _decode("xaacdec", aac, ref_wav, extra=extra)
    log = _decode("xaacdec_dropin", aac, gpu_wav, extra=extra)
    n_esbr = int(re.search(r"(\d+) sbr_dec calls took the eSBR \(Path A\) branch on the GPU", log).group(1))
    n_ds = int(re.search(r"(\d+) of the eSBR calls with the down-sampled synthesis bank", log).group(1))
    n_left = int(re.search(r"(\d+) sbr_dec calls left to the reference", log).group(1))
    assert n_esbr > 30 and n_ds == n_esbr and n_left == 0, log[-900:]
    a, b = open(ref_wav, "rb").read(), open(gpu_wav, "rb").read()
    assert len(a) > 50000 and a == b, (len(a), len(b))


def test_streams_present():
    assert len(STREAMS) >= 3


STREAMS_LD = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "streams_ld", "*.aac")))


@pytest.mark.parametrize("aac", STREAMS_LD, ids=[os.path.basename(s) for s in STREAMS_LD])
@pytest.mark.parametrize("flags", [("-esbr:0",), ()], ids=["esbr0", "default"])
def test_other_frame_lengths_through_the_gpu(aac, flags, tmp_path):
    """AAC-LC with 960-line frames, AAC-LD and AAC-ELD with 512- and 480-line frames (made by the reference encoder,
    tools/make_golden_streams.py): the real reference decoder with ixheaacd_imdct_process served by
    xaac_imdct960_process_batch / xaac_imdct_ld_process_batch writes the same bytes as the unmodified one; the ELD streams'
    low-delay SBR runs whole on the GPU (xaac_sbr_eld_process_batch: LD analysis bank, core, LD synthesis bank), and with
    the core left to the reference its two complex QMF banks alone (xaac_qmf_analysis_eld_batch / xaac_qmf_synthesis_eld_batch)."""
    if not (os.path.exists(os.path.join(REF, "xaacdec")) and os.path.exists(os.path.join(REF, "xaacdec_dropin"))):
        pytest.fail("oracle/_ref/xaacdec[_dropin] missing: the reference binaries (built by oracle/Makefile.ref where /root/reference exists, git-ignored) did not travel with the snapshot -- the drop-in evidence must not vanish silently")
    meta = aac[:-4] + ".txt"
    extra = tuple(flags) + (("-mp4:1", "-imeta:" + meta) if os.path.exists(meta) else ())
    ref_wav, gpu_wav = str(tmp_path / "ref.wav"), str(tmp_path / "gpu.wav")
    _decode("xaacdec", aac, ref_wav, extra=extra)
    log = _decode("xaacdec_dropin", aac, gpu_wav, extra=extra)
    m = re.search(r"(\d+) imdct_process calls of 960-line frames and (\d+) of AAC-LD / ELD frames ran on the GPU", log)
    assert m, log[-600:]
    n960, nld = int(m.group(1)), int(m.group(2))
    assert (n960 > 100 and nld == 0) if "lc960" in aac else (nld > 100 and n960 == 0), (n960, nld)
    mb = re.search(r"(\d+) LD / ELD analysis-bank and (\d+) synthesis-bank calls ran on the GPU", log)
    assert mb, log[-600:]
    if "eld" in os.path.basename(aac):   # low-delay SBR: every ixheaacd_sbr_dec call whole -- LD banks (16 / 15 slots) + core -- on the GPU
        me = re.search(r"(\d+) whole low-delay SBR calls", log)
        assert me and int(me.group(1)) > 100, log[-600:]
        # ... and with the core left to the reference (XAAC_DROPIN_NO_ELD_SBR) its two banks alone, as before
        bank_wav = str(tmp_path / "banks.wav")
        log2 = _decode("xaacdec_dropin", aac, bank_wav, extra=extra, env=dict(os.environ, XAAC_DROPIN_NO_ELD_SBR="1"))
        mb2 = re.search(r"(\d+) LD / ELD analysis-bank and (\d+) synthesis-bank calls ran on the GPU", log2)
        assert mb2 and int(mb2.group(1)) > 100 and int(mb2.group(2)) > 100, log2[-600:]
        assert open(bank_wav, "rb").read() == open(ref_wav, "rb").read()
    a, b = open(ref_wav, "rb").read(), open(gpu_wav, "rb").read()
    assert len(a) > 100000 and a == b, (len(a), len(b), n960, nld)


STREAMS_USAC = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "streams_usac", "*.aac")))


@pytest.mark.parametrize("aac", STREAMS_USAC, ids=[os.path.basename(s) for s in STREAMS_USAC])
def test_usac_streams_esbr_through_the_gpu(aac, tmp_path):
    """USAC (xHE-AAC) streams made by the reference encoder (tools/make_golden_streams.py: -aot:42; stereo 2:1 eSBR without
    and with harmonic SBR, a switched FD / LPD core, mono streams whose LPD frames use PVC, 4:1 eSBR): the reference's own
    USAC front end -- arithmetic decoder, LPD, stereo tools, all CPU -- hands its core samples and SBR side info to the same
    ixheaacd_sbr_dec seam, and every ORIG_SBR 2:1 call of it runs on the GPU (xaac_esbr_sbr_process_batch with the
    XAAC_ESBR_USAC / _NO_X_DELAY / _SKIP_ADJUST side flags), PVC frames included (the PVC decoder and the envelope adjuster's
    PVC branch run inside the same call, xaac_esbr_pvc_side / _state).  Streams at the other two SBR ratios -- 8:3 (768-line
    core, 24-channel analysis bank) and 4:1 (16-channel bank, 64 QMF slots), plain and with PVC frames -- go the same way
    (xaac_esbr_sbr_batch.sbr_ratio).  The decoded file is byte-identical to the unmodified decoder's."""
    if not (os.path.exists(os.path.join(REF, "xaacdec")) and os.path.exists(os.path.join(REF, "xaacdec_dropin"))):
        pytest.fail("oracle/_ref/xaacdec[_dropin] missing: the reference binaries did not travel with the snapshot")
    extra = ("-mp4:1", "-imeta:" + aac[:-4] + ".txt")
    ref_wav, gpu_wav = str(tmp_path / "ref.wav"), str(tmp_path / "gpu.wav")
    _decode("xaacdec", aac, ref_wav, extra=extra)
    log = _decode("xaacdec_dropin", aac, gpu_wav, extra=extra)
    m = re.search(r"(\d+) of them for USAC channels, (\d+) sbr_dec calls left to the reference", log)
    assert m, log[-600:]
    on_gpu, left = int(m.group(1)), int(m.group(2))
    name = os.path.basename(aac)
    # the USAC frequency-domain seam: ixheaacd_fd_frm_dec through xaac_usac_imdct_process_batch (1024- and 768-line frames; FAC
    # signal and bass post filter of LPD -> FD transitions stay the LPD decoder's, oracle/ref_dropin_usac.c)
    mi = re.search(r"(\d+) USAC fd_frm_dec calls ran on the GPU, (\d+) with a FAC signal, (\d+) behind an LPD frame", log)
    assert mi, log[-600:]
    n_fd, n_fac, n_lpd = (int(v) for v in mi.groups())
    md = re.search(r"(\d+) FAC signals made on the device", log)
    assert md and int(md.group(1)) == n_fac, log[-600:]     # ixheaacd_cal_fac_data itself on the GPU (xaac_usac_fac_in), not handed over
    if "td" in name:
        assert n_fd == 0                # LPD frames only
    elif "sw" in name:
        assert n_fd > (2 if "pvc" in name else 10) and n_lpd > 0 and (n_fac > 0 or "pvc" in name), (n_fd, n_fac, n_lpd)   # a switched core: LPD -> FD transitions (u21sw: with a FAC signal)
    else:
        assert n_fd > 30, n_fd
    mr = re.search(r"(\d+) of the USAC calls at 8:3 SBR \(24-channel bank\), (\d+) at 4:1", log)
    assert mr, log[-600:]
    n83, n41 = int(mr.group(1)), int(mr.group(2))
    assert (n83 == on_gpu and n41 == 0) if "83" in name else (n41 == on_gpu and n83 == 0) if "41" in name else (n83 == 0 and n41 == 0)
    if "pvc" in name:                 # PVC frames too: the PVC decoder and the adjuster's PVC branch inside the same call
        mp = re.search(r"(\d+) of the USAC calls were PVC frames", log)
        few = "41" in name              # a 4:1 frame is 4096 samples: half as many calls
        assert on_gpu > (15 if few else 30) and left == 0 and mp and int(mp.group(1)) > ((15 if few else 30) if "td" in name else 5), (on_gpu, left, log[-300:])
    else:
        assert on_gpu > 30 and left == 0, (on_gpu, left)
    if "harm" in name:
        mh = re.search(r"(\d+) of them with harmonic patching", log)
        assert mh and int(mh.group(1)) > 20, log[-600:]
    a, b = open(ref_wav, "rb").read(), open(gpu_wav, "rb").read()
    assert len(a) > 100000 and a == b, (len(a), len(b), on_gpu, left)


STREAMS_WIDE = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "streams_wide", "*.aac")))


@pytest.mark.parametrize("aac", STREAMS_WIDE, ids=[os.path.basename(s) for s in STREAMS_WIDE])
@pytest.mark.parametrize("flags", [("-esbr:0",), ()], ids=["esbr0", "default"])
def test_wider_streams_through_the_same_seams(aac, flags, tmp_path):
    """Streams wider than the scope table, made by the reference encoder (tools/make_golden_streams.py: streams_wide).
    5.1 AAC-LC and HE-AAC: the reference's channel-element loop calls the same three seams once per channel / element, so every
    IMDCT, every SBR call (fixed-point with -esbr:0, Path A with the default flags) and the limiter's six-channel calls run on
    the GPU unchanged.  HE-AAC / HE-AACv2 with 960-line frames (the DAB+ profile): the 960-line IMDCT is the library's, the SBR
    calls of 15 time slots (30 QMF slots) are left to the reference, counted as such.  Byte-identical output either way."""
    if not (os.path.exists(os.path.join(REF, "xaacdec")) and os.path.exists(os.path.join(REF, "xaacdec_dropin"))):
        pytest.fail("oracle/_ref/xaacdec[_dropin] missing: the reference binaries did not travel with the snapshot")
    meta = aac[:-4] + ".txt"
    extra = tuple(flags) + (("-mp4:1", "-imeta:" + meta) if os.path.exists(meta) else ())
    ref_wav, gpu_wav = str(tmp_path / "ref.wav"), str(tmp_path / "gpu.wav")
    _decode("xaacdec", aac, ref_wav, extra=extra)
    log = _decode("xaacdec_dropin", aac, gpu_wav, extra=extra)
    num = lambda pat: [int(v) for v in re.search(pat, log).groups()]
    n_imdct, n_sbr = num(r"(\d+) imdct_process and (\d+) sbr_dec calls ran on the GPU")
    n960, _ = num(r"(\d+) imdct_process calls of 960-line frames and (\d+) of AAC-LD")
    n_lim, = num(r"(\d+) peak_limiter_process calls ran on the GPU")
    n_esbr, = num(r"(\d+) sbr_dec calls took the eSBR \(Path A\) branch on the GPU")
    _, n_left = num(r"(\d+) of them for USAC channels, (\d+) sbr_dec calls left to the reference")
    name = os.path.basename(aac)
    if name.startswith("mc6_aot2"):
        assert n_imdct > 250 and n_lim > 40 and n_sbr == n_esbr == n_left == 0, log[-900:]
    elif name.startswith("mc6_aot5"):        # three channel pair / single elements + LFE a frame: one SBR call per element
        assert n_imdct > 140 and n_left == 0 and (n_sbr if flags else n_esbr) > 140, log[-900:]
    else:                                    # 960-line HE-AAC: the transform on the GPU, the 15-time-slot SBR calls the reference's
        assert n960 > 40 and n_imdct == n_sbr == n_esbr == 0 and (not flags or n_left > 40), log[-900:]
    a, b = open(ref_wav, "rb").read(), open(gpu_wav, "rb").read()
    assert len(a) > 100000 and a == b, (len(a), len(b))


@pytest.mark.parametrize("aac", [s for s in STREAMS if "aot5_" in s or "aot29_" in s], ids=lambda s: os.path.basename(s))
def test_dft_harmonic_transposer_behind_the_reference_decoder(aac, tmp_path):
    """-esbr_hq:1: every ixheaacd_dft_hbe_apply call of the reference decoder (hbe_dft_trans.c:771; one per channel and frame, two
    more where a header resets the SBR decoder) served by the library: inside xaac_esbr_sbr_process_batch (hbe_dft_state) for the
    frames' calls, by xaac_hbe_dft_apply_batch_run for the reset-time ones.  This path is float with libm calls and
    transforms of the library's own: the decoded file is held to the tolerance BASELINE.json's north_star gives float SBR --
    every 16-bit sample within 1 LSB of the unmodified decoder's -- and in fact most streams come out identical (the
    transposer's rows are only read where the stream asks for harmonic patching: harm_aot5_48k, 0.1 % of its samples off by one)."""
    import wave
    import numpy as np
    if not (os.path.exists(os.path.join(REF, "xaacdec")) and os.path.exists(os.path.join(REF, "xaacdec_dropin"))):
        pytest.fail("oracle/_ref/xaacdec[_dropin] missing: the reference binaries did not travel with the snapshot")
    ref_wav, gpu_wav = str(tmp_path / "ref.wav"), str(tmp_path / "gpu.wav")
    _decode("xaacdec", aac, ref_wav, extra=("-esbr_hq:1",))
    log = _decode("xaacdec_dropin", aac, gpu_wav, extra=("-esbr_hq:1",))
    m = re.search(r"(\d+) dft_hbe_apply calls .* ran on the GPU, (\d+) left to the reference", log)
    mc = re.search(r"(\d+) of the eSBR calls with the DFT transposer inside the chain", log)
    ml = re.search(r"(\d+) sbr_dec calls left to the reference", log)
    # the frames' calls run inside the Path A chain (xaac_esbr_sbr_batch.hbe_dft_state), the reset-time ones of ixheaacd_applysbr through
    # the seam on the function itself; nothing of such a stream's SBR is left to the reference
    assert m and mc and ml and int(mc.group(1)) > 30 and int(m.group(1)) >= 2 and int(m.group(2)) == 0 and int(ml.group(1)) == 0, log[-900:]
    if "XAAC_DROPIN_NO_DFT_CHAIN" not in os.environ:   # ... and with the chain held back every call goes through the function's seam
        log2 = _decode("xaacdec_dropin", aac, str(tmp_path / "gpu2.wav"), extra=("-esbr_hq:1",), env=dict(os.environ, XAAC_DROPIN_NO_DFT_CHAIN="1"))
        m2 = re.search(r"(\d+) dft_hbe_apply calls .* ran on the GPU, (\d+) left to the reference", log2)
        assert m2 and int(m2.group(1)) > 30 and int(m2.group(2)) == 0, log2[-800:]
        assert open(gpu_wav, "rb").read() == open(str(tmp_path / "gpu2.wav"), "rb").read()   # the same kernels either way

    def samples(path):
        with wave.open(path) as w:
            return w.getnchannels(), w.getframerate(), np.frombuffer(w.readframes(w.getnframes()), np.int16).astype(np.int32)
    a, b = samples(ref_wav), samples(gpu_wav)
    assert a[:2] == b[:2] and a[2].size == b[2].size and a[2].size > 50000
    d = np.abs(a[2] - b[2])
    assert d.max() <= 1 and (d != 0).mean() < 0.005, (int(d.max()), float((d != 0).mean()))
    # and the flag does change the decode where the stream uses harmonic patching
    if "harm_" in aac:
        plain = str(tmp_path / "plain.wav")
        _decode("xaacdec", aac, plain, extra=())
        assert not np.array_equal(samples(plain)[2], a[2])
