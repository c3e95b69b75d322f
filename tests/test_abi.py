"""The C-ABI library loads on a GPU-less box and exports every symbol that
include/xaac_amd.h declares; argument errors follow the reference's error-code
convention.  No compute calls here."""
import ctypes
import os
import subprocess
import sys
import re

import pytest

import libxaac_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


HOST_HEADERS = ("xaac_parse.h",)  # the CPU front end's boundary: libxaac_amd/libxaac_host.so


def _declared_functions(host=False):
    names = set()
    for h in sorted(os.listdir(os.path.join(ROOT, "include"))):  # every header of the boundary
        if (h in HOST_HEADERS) != host:
            continue
        txt = open(os.path.join(ROOT, "include", h)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        names |= set(re.findall(r"\b(xaac_[a-z0-9_]+)\s*\(", txt))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    lib = libxaac_amd.load_library()
    names = _declared_functions()
    assert "xaac_imdct_process_batch" in names and "xaac_create" in names and "xaac_hbe_cplx_anal_batch" in names
    for n in names:
        assert hasattr(lib, n), n


def _exported(path):
    out = subprocess.check_output(["nm", "-D", "--defined-only", path], text=True)
    return sorted(line.split()[-1] for line in out.splitlines() if line.strip())


def test_libraries_export_nothing_but_the_declared_symbols():
    """-fvisibility=hidden + XAAC_API (and libxaac_amd/csrc/exports.map for what the HIP front end forces visible): the
    dynamic symbol tables are the headers' function lists, no launch helpers, kernel handles or C++ names beside them"""
    from libxaac_amd import decoder
    libxaac_amd.load_library()
    decoder.load_host_library()
    assert _exported(libxaac_amd.library_path()) == _declared_functions()
    assert _exported(decoder.host_library_path()) == _declared_functions(host=True)


def test_host_library_exports_every_declared_symbol():
    """include/xaac_parse.h <-> libxaac_amd/libxaac_host.so (CPU only: loads and parses without a GPU)"""
    from libxaac_amd import decoder
    lib = decoder.load_host_library()
    names = _declared_functions(host=True)
    assert "xaac_parse_adts_frame" in names and "xaac_parse_sbr_side" in names and "xaac_sbr_state_init" in names
    for n in names:
        assert hasattr(lib, n), n


def test_host_struct_layouts_match_header(tmp_path):
    """the ctypes mirrors of libxaac_amd/decoder.py against what a C compiler makes of include/xaac_parse.h"""
    from libxaac_amd import decoder
    src, exe = tmp_path / "h.c", tmp_path / "h"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "xaac_parse.h"\nint main(void) { printf("%zu %zu %zu %zu %zu %zu %zu\\n", '
                   'sizeof(xaac_adts_header), sizeof(xaac_core_frame), offsetof(xaac_core_frame, spec), offsetof(xaac_core_frame, sbr), '
                   'sizeof(xaac_sbr_side), offsetof(xaac_sbr_side, frame), offsetof(xaac_sbr_side, ps_frame)); return 0; }\n')
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    want = [ctypes.sizeof(decoder.AdtsHeader), ctypes.sizeof(decoder.CoreFrame), decoder.CoreFrame.spec.offset,
            decoder.CoreFrame.sbr.offset, ctypes.sizeof(decoder.SbrSide), decoder.SbrSide.frame.offset,
            decoder.SbrSide.ps_frame.offset]
    assert got == want


def test_version_string():
    assert b"gfx950" in libxaac_amd.load_library().xaac_version()


def test_null_and_bad_arguments_are_fatal_codes():
    lib = libxaac_amd.load_library()
    assert lib.xaac_create(None, 0, None) & 0x80000000
    assert lib.xaac_destroy(None) & 0x80000000
    assert lib.xaac_sync(None) & 0x80000000
    assert lib.xaac_imdct_process_batch(None, None) & 0x80000000
    # every batch entry point turns a missing context or descriptor into a fatal code before it looks at anything else
    import re
    text = "".join(open(os.path.join(ROOT, "include", h)).read() for h in ("xaac_amd.h", "xaac_esbr.h", "xaac_hbe.h", "xaac_pvc.h"))
    names = set(re.findall(r"XAAC_API int32_t (xaac_\w+)\(xaac_ctx \*\w+, const \w+ \*\w+\);", text))
    assert len(names) >= 25 and "xaac_hbe_dft_apply_batch_run" in names and "xaac_esbr_sbr_process_batch" in names
    for n in sorted(names):
        fn = getattr(lib, n)
        fn.restype = ctypes.c_int32
        fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        assert fn(None, None) & 0x80000000, n


def _eld_state():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import sbr_capture
    assert ctypes.sizeof(sbr_capture.EldState) == libxaac_amd.SBR_ELD_STATE_BYTES
    return sbr_capture.EldState


def test_struct_layout_matches_header(tmp_path):
    """every batch descriptor's ctypes mirror against what a C compiler makes of include/xaac_amd.h"""
    import subprocess
    pairs = [("xaac_imdct_batch", libxaac_amd._ImdctBatch, "status"), ("xaac_qmf_ana_batch", libxaac_amd._QmfAnaBatch, "qmf"),
             ("xaac_qmf_syn_batch", libxaac_amd._QmfSynBatch, "pcm"), ("xaac_sbr_lp_batch", libxaac_amd._SbrLpBatch, "workspace_bytes"),
             ("xaac_sbr_hq_batch", libxaac_amd._SbrHqBatch, "max_band_hint"),
             ("xaac_sbr_eld_batch", libxaac_amd._SbrEldBatch, "qmf_handed_on"), ("xaac_sbr_eld_state", _eld_state(), "harm_flags_prev")]
    body = "".join('printf("%%zu %%zu\\n", sizeof(%s), offsetof(%s, %s));' % (c, c, last) for c, _, last in pairs)
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "xaac_amd.h"\nint main(void) { %s return 0; }\n' % body)
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    want = []
    for _, cls, last in pairs:
        want += [ctypes.sizeof(cls), getattr(cls, last).offset]
    assert got == want
    assert ctypes.sizeof(libxaac_amd._ImdctBatch) == 80   # 2 x int32 + 7 pointers + int32 (+ pad) + status pointer on LP64


def test_round2_struct_layouts_match_headers(tmp_path):
    """the round-2 descriptors and states (eSBR / Path A, float PS, USAC IMDCT, hand-over): ctypes mirrors of the binding
    and of tests/esbr_structs.py against what a C compiler makes of include/xaac_amd.h and include/xaac_esbr.h"""
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import esbr_structs as es
    pairs = [("xaac_esbr_ana_batch", libxaac_amd._EsbrAnaBatch, "qmf_im"), ("xaac_esbr_ana_nb_batch", libxaac_amd._EsbrAnaNbBatch, "qmf_im"),
             ("xaac_esbr_syn_batch", libxaac_amd._EsbrSynBatch, "out"),
             ("xaac_usac_imdct_batch", libxaac_amd._UsacImdctBatch, "fac_work"), ("xaac_sbr_handover_batch", libxaac_amd._HandoverBatch, "ps_state"),
             ("xaac_sbr_apply_side_batch", libxaac_amd._ApplySideBatch, "ps_state"),
             ("xaac_esbr_sbr_batch", libxaac_amd._EsbrSbrBatch, "down_sample"), ("xaac_esbr_side", es.EsbrSide, "pitch_in_bins"),
             ("xaac_esbr_pvc_side", es.EsbrPvcSide, "pvc"), ("xaac_esbr_pvc_state", es.EsbrPvcState, "esbr_start_up_pvc"),
             ("xaac_esbr_state", es.EsbrState, "ph_im"), ("xaac_esbr_ps_state", es.EsbrPsState, "syn_r"),
             ("xaac_esbr_ana_state", es.EsbrAna, "win_off"), ("xaac_esbr_syn_state", es.EsbrSyn, "filt_off")]
    body = "".join('printf("%%zu %%zu\\n", sizeof(%s), offsetof(%s, %s));' % (c, c, last) for c, _, last in pairs)
    src = tmp_path / "layout2.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "xaac_esbr.h"\nint main(void) { %s return 0; }\n' % body)
    exe = tmp_path / "layout2"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    want = []
    for _, cls, last in pairs:
        want += [ctypes.sizeof(cls), getattr(cls, last).offset]
    assert got == want
    assert ctypes.sizeof(es.EsbrAna) == 4 * libxaac_amd.ESBR_ANA_STATE_WORDS and ctypes.sizeof(es.EsbrSyn) == 4 * libxaac_amd.ESBR_SYN_STATE_WORDS
    assert (ctypes.sizeof(es.EsbrSide), ctypes.sizeof(es.EsbrState), ctypes.sizeof(es.EsbrPsState)) == \
        (libxaac_amd.ESBR_SIDE_BYTES, libxaac_amd.ESBR_STATE_BYTES, libxaac_amd.ESBR_PS_STATE_BYTES)
    assert (ctypes.sizeof(es.EsbrPvcSide), ctypes.sizeof(es.EsbrPvcState)) == (libxaac_amd.ESBR_PVC_SIDE_BYTES, libxaac_amd.ESBR_PVC_STATE_BYTES)


def test_hbe_struct_layouts_match_header(tmp_path):
    """the harmonic transposer's state and batch descriptors against include/xaac_hbe.h"""
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import hbe_structs as hs
    pairs = [("xaac_hbe_state", hs.HbeState, "max_stretch"), ("xaac_hbe_synth_batch", libxaac_amd._HbeSynthBatch, "status"),
             ("xaac_hbe_anal_batch", libxaac_amd._HbeAnalBatch, "status"),
             ("xaac_hbe_apply_batch_desc", libxaac_amd._HbeApplyBatch, "status"),
             ("xaac_hbe_dft_anal_batch", libxaac_amd._HbeDftAnalBatch, "status"), ("xaac_hbe_dft_anal_state", hs.HbeDftState, "a_start"),
             ("xaac_hbe_dft_state", hs.HbeDftFullState, "last_status"), ("xaac_hbe_dft_cfg", hs.HbeDftCfg, "fd_win"),
             ("xaac_hbe_dft_apply_batch", libxaac_amd._HbeDftApplyBatch, "rows32")]
    body = "".join('printf("%%zu %%zu\\n", sizeof(%s), offsetof(%s, %s));' % (c, c, last) for c, _, last in pairs)
    src = tmp_path / "layout3.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "xaac_hbe.h"\nint main(void) { %s return 0; }\n' % body)
    exe = tmp_path / "layout3"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    want = []
    for _, cls, last in pairs:
        want += [ctypes.sizeof(cls), getattr(cls, last).offset]
    assert got == want
    assert ctypes.sizeof(hs.HbeState) == libxaac_amd.HBE_STATE_BYTES
    assert (ctypes.sizeof(hs.HbeDftFullState), ctypes.sizeof(hs.HbeDftCfg)) == (libxaac_amd.HBE_DFT_FULL_STATE_BYTES, libxaac_amd.HBE_DFT_CFG_BYTES)
    src2 = tmp_path / "layout4.c"
    src2.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "xaac_amd.h"\nint main(void) { printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", '
                    'sizeof(xaac_qmf_ana_eld_state), offsetof(xaac_qmf_ana_eld_state, fp), sizeof(xaac_qmf_ana_eld_batch), '
                    'offsetof(xaac_qmf_ana_eld_batch, status), sizeof(xaac_qmf_syn_eld_state), offsetof(xaac_qmf_syn_eld_state, sixty4), '
                    'sizeof(xaac_qmf_syn_eld_batch), offsetof(xaac_qmf_syn_eld_batch, status)); '
                    'printf("%zu %zu %zu\\n", sizeof(xaac_imdct_ld_batch), offsetof(xaac_imdct_ld_batch, spec), offsetof(xaac_imdct_ld_batch, status)); return 0; }\n')
    exe2 = tmp_path / "layout4"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src2), "-o", str(exe2)])
    B = libxaac_amd._QmfAnaEldBatch
    S = libxaac_amd._QmfSynEldBatch
    assert [int(v) for v in subprocess.check_output([str(exe2)]).split()] == [
        2 * libxaac_amd.QMF_ANA_ELD_STATE_WORDS, 646, ctypes.sizeof(B), B.status.offset,
        2 * libxaac_amd.QMF_SYN_ELD_STATE_WORDS, 2566, ctypes.sizeof(S), S.status.offset,
        ctypes.sizeof(libxaac_amd._ImdctLdBatch), libxaac_amd._ImdctLdBatch.spec.offset, libxaac_amd._ImdctLdBatch.status.offset]


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(libxaac_amd.XaacError) as e:
        libxaac_amd.XaacContext(0)
    assert e.value.code == 0xFFFF8002


def test_limiter_structs_match_header(tmp_path):
    """the ctypes mirrors of the limiter boundary against what a C compiler makes of include/xaac_amd.h"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "xaac_amd.h"\n'
                   'int main(void) { printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(xaac_limiter_state), '
                   'offsetof(xaac_limiter_state, pre_smoothed_gain), offsetof(xaac_limiter_state, max_buf), '
                   'offsetof(xaac_limiter_state, delayed_input), sizeof(xaac_limiter_batch), '
                   'offsetof(xaac_limiter_batch, stride), offsetof(xaac_limiter_batch, pcm16), '
                   'offsetof(xaac_limiter_batch, workspace_bytes)); return 0; }\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    S, B = libxaac_amd.LimiterState, libxaac_amd._LimiterBatch
    assert got == [ctypes.sizeof(S), S.pre_smoothed_gain.offset, S.max_buf.offset, S.delayed_input.offset,
                   ctypes.sizeof(B), B.stride.offset, B.pcm16.offset, B.workspace_bytes.offset]
    assert libxaac_amd.LIMITER_STATE_BYTES == got[0]


def test_pvc_struct_layouts_match_header(tmp_path):
    """the PVC decoder's frame, state and batch descriptor against include/xaac_pvc.h"""
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import pvc_structs as ps
    pairs = [("xaac_pvc_frame", ps.PvcFrame, "pvc_id"), ("xaac_pvc_state", ps.PvcState, "prev_pvc_rate"), ("xaac_pvc_batch", libxaac_amd._PvcBatch, "status")]
    body = "".join('printf("%%zu %%zu\\n", sizeof(%s), offsetof(%s, %s));' % (c, c, last) for c, _, last in pairs)
    src = tmp_path / "layout_pvc.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "xaac_pvc.h"\nint main(void) { %s return 0; }\n' % body)
    exe = tmp_path / "layout_pvc"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    want = []
    for _, cls, last in pairs:
        want += [ctypes.sizeof(cls), getattr(cls, last).offset]
    assert got == want
    assert (ctypes.sizeof(ps.PvcFrame), ctypes.sizeof(ps.PvcState)) == (libxaac_amd.PVC_FRAME_BYTES, libxaac_amd.PVC_STATE_BYTES)


def test_parse_batch_descriptor_matches_header(tmp_path):
    """libxaac_amd/decoder.py's mirror of struct xaac_parse_batch (members appended in round 4: pos, frames, lines) against what
    a C compiler makes of include/xaac_parse.h"""
    import subprocess
    from libxaac_amd import decoder
    src = tmp_path / "pb.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "xaac_parse.h"\nint main(void) { printf("%zu %zu %zu %zu %zu\\n", '
                   'sizeof(xaac_parse_batch), offsetof(xaac_parse_batch, parser), offsetof(xaac_parse_batch, pos), '
                   'offsetof(xaac_parse_batch, frames), offsetof(xaac_parse_batch, lines)); return 0; }\n')
    exe = tmp_path / "pb"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    P = decoder._ParseBatch
    assert got == [ctypes.sizeof(P), P.parser.offset, P.pos.offset, P.frames.offset, P.lines.offset]
