"""oracle/Makefile.ref adds ONE flag to the reference's own build flags: -fno-strict-aliasing (gcc 11 otherwise
miscompiles the WORD32* stores into WORD16 arrays of ixheaacd_apply_rot_dec, ps_dec.c:929-943, and HE-AACv2 decodes to
different PCM).  Cross-check of the pin: the same sources built by clang with the reference's PLAIN flags
(oracle/_ref/xaacdec_clang) decode every golden stream to the same bytes as the gcc -fno-strict-aliasing build
(oracle/_ref/xaacdec) that every fixture in tests/golden/ was made with."""
import glob
import hashlib
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
STREAMS = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "streams", "*.aac")))


@pytest.mark.parametrize("aac", STREAMS, ids=[os.path.basename(s) for s in STREAMS])
@pytest.mark.parametrize("esbr", ["0", "1"])
def test_clang_plain_flags_build_decodes_identically(aac, esbr, tmp_path):
    if not all(os.path.exists(os.path.join(REF, b)) for b in ("xaacdec", "xaacdec_clang")):
        pytest.skip("oracle/_ref/xaacdec[_clang] missing (built by oracle/Makefile.ref where /root/reference exists)")
    md5 = {}
    for b in ("xaacdec", "xaacdec_clang"):
        out = str(tmp_path / (b + ".wav"))
        subprocess.run([os.path.join(REF, b), "-ifile:" + aac, "-ofile:" + out, "-esbr:" + esbr], stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL, timeout=600, check=False)
        md5[b] = hashlib.md5(open(out, "rb").read()).hexdigest()
    assert md5["xaacdec"] == md5["xaacdec_clang"]
