#!/usr/bin/env python3
"""debugging aid: the device-resident Path A states of the own decoder in front of every frame against what the reference's
ixheaacd_sbr_dec calls find (XAAC_ESBR_INIT_FILE + XAAC_ESBR_INIT_ALL of oracle/ref_capture.c)"""
import ctypes, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from libxaac_amd import decoder, ESBR_STATE_BYTES, HBE_STATE_BYTES, ESBR_PS_STATE_BYTES
from esbr_structs import EsbrState
from hbe_structs import HbeState
name = sys.argv[1]
src = name if os.path.exists(name) else os.path.join(ROOT, "tests/golden/streams", name + ".aac")
subprocess.run([os.path.join(ROOT, "oracle/_ref/xaacdec_capture"), "-ifile:" + src, "-ofile:/tmp/o.wav"], check=True, capture_output=True,
               env=dict(os.environ, XAAC_ESBR_SIDE_FILE="/tmp/side.bin", XAAC_ESBR_INIT_FILE="/tmp/init.bin", XAAC_ESBR_INIT_ALL="1"))
raw = open("/tmp/init.bin", "rb").read()
rs = ESBR_STATE_BYTES + HBE_STATE_BYTES + ESBR_PS_STATE_BYTES + 4096
calls = len(raw) // rs
import sbr_capture as sc
from esbr_structs import EsbrSide
sraw = open("/tmp/side.bin", "rb").read()
SRS = 32 + 336 + 1072 + ctypes.sizeof(EsbrSide) + 972 + 44
step = [0]
n_ch = None
def diff(T, a, b, label):
    A, B = T.from_buffer_copy(a), T.from_buffer_copy(b)
    out = []
    def walk(x, y, prefix):
        for n, _ in type(x)._fields_:
            u, v = getattr(x, n), getattr(y, n)
            if hasattr(u, "_fields_"):
                walk(u, v, prefix + n + ".")
                continue
            ua = np.ctypeslib.as_array(u).ravel() if hasattr(u, "__len__") else np.array([u])
            va = np.ctypeslib.as_array(v).ravel() if hasattr(v, "__len__") else np.array([v])
            if ua.tobytes() != va.tobytes():
                w = np.nonzero(ua.view(np.uint8).reshape(ua.size, -1).any(1) | True)[0]
                bad = np.nonzero((ua != va) | (np.isnan(ua.astype(np.float64)) != np.isnan(va.astype(np.float64))))[0]
                out.append("%s%s: %d of %d differ, first %s mine %s ref %s" % (prefix, n, len(bad), ua.size, bad[:4].tolist(), ua[bad[:4]].tolist(), va[bad[:4]].tolist()))
    walk(A, B, "")
    for o in out[:8]:
        print("   ", label, o)
    return len(out)
def trace(t):
    k = step[0]; step[0] += 1
    st = t["state"].cpu().numpy(); hb = t["hbe"].cpu().numpy()
    nch = st.shape[0]
    if k < int(os.environ.get("DBG_FIRST", "0")): return
    for c in range(nch):
        call = nch + k * nch + c           # the initialisation pass comes first
        if call >= calls: return
        o = call * rs
        nb = diff(EsbrState, st[c].tobytes(), raw[o:o + ESBR_STATE_BYTES], "frame %d ch %d state" % (k, c))
        nb += diff(HbeState, hb[c].tobytes(), raw[o + ESBR_STATE_BYTES:o + ESBR_STATE_BYTES + HBE_STATE_BYTES], "frame %d ch %d hbe" % (k, c))
        so = call * SRS + 32
        nb += diff(sc.Header, t["header"].cpu().numpy()[c].tobytes(), sraw[so:so + 336], "frame %d ch %d header" % (k, c))
        nb += diff(sc.Frame, t["frame"].cpu().numpy()[c].tobytes(), sraw[so + 336:so + 1408], "frame %d ch %d frame" % (k, c))
        nb += diff(EsbrSide, t["side"].cpu().numpy()[c].tobytes(), sraw[so + 1408:so + 1408 + ctypes.sizeof(EsbrSide)], "frame %d ch %d side" % (k, c))
        core = t["core"].cpu().numpy()[c]
        ref_core = np.frombuffer(raw[o + rs - 4096:o + rs], np.float32)
        bad = np.nonzero(core != ref_core)[0]
        if bad.size: print("frame", k, "ch", c, "core differs at", bad.size, "first", bad[:6].tolist(), core[bad[:6]].tolist(), ref_core[bad[:6]].tolist())
        if nb: print("frame", k, "ch", c, "members differing", nb)
    if k > int(os.environ.get("DBG_LAST", "6")): raise SystemExit
decoder.decode_streams([open(src, "rb").read()], esbr=True, overlap=False, _trace=trace)
