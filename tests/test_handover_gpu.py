"""xaac_sbr_state_handover: the state copies the reference makes when a mono stream turns into a parametric-stereo or a
stereo one (decoder/ixheaacd_sbrdecoder.c:762-775 and :777-806), checked field by field against those two memcpy blocks
written out on the struct mirrors of tests/sbr_capture.py; everything the reference does not copy (ring positions, the
rest of channel 1's state, every stream not listed) must stay as it was."""
import ctypes

import numpy as np
import pytest

import sbr_capture as c


def _arr(obj, name):
    return np.ctypeslib.as_array(getattr(obj, name)).reshape(-1)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["ps", "stereo"])
def test_handover(mode):
    import torch
    import libxaac_amd
    dev = torch.device("cuda:0")
    ctx = libxaac_amd.XaacContext(0, None)
    rng = np.random.default_rng(5)
    n = 12
    sb, pb = ctypes.sizeof(c.State), ctypes.sizeof(c.PsState)
    assert sb == libxaac_amd.SBR_STATE_BYTES and pb == libxaac_amd.PS_STATE_BYTES
    st0 = rng.integers(0, 256, (n, sb), dtype=np.uint8)
    ps0 = rng.integers(0, 256, (n, pb), dtype=np.uint8)
    src = np.array([0, 3, 4, 9], np.int32)
    dst = np.array([0, 3, 4, 9], np.int32) if mode == "ps" else np.array([1, 2, 5, 11], np.int32)
    st, ps = torch.from_numpy(st0).to(dev), torch.from_numpy(ps0).to(dev)
    ctx.sbr_state_handover(libxaac_amd.HANDOVER_PS_START if mode == "ps" else libxaac_amd.HANDOVER_STEREO_START,
                           torch.from_numpy(src).to(dev), torch.from_numpy(dst).to(dev), st, ps if mode == "ps" else None)
    ctx.sync()
    want_st, want_ps = st0.copy(), ps0.copy()
    for s, d in zip(src, dst):
        a = c.State.from_buffer_copy(st0[s].tobytes())
        if mode == "ps":
            b = c.PsState.from_buffer_copy(ps0[d].tobytes())
            _arr(b, "syn_ring_r")[:] = _arr(a, "syn_ring")
            b.st_syn_scale_r = a.st_syn_scale
            want_ps[d] = np.frombuffer(bytes(b), np.uint8)
        else:
            b = c.State.from_buffer_copy(st0[d].tobytes())
            _arr(b, "syn_ring")[:] = _arr(a, "syn_ring")
            _arr(b, "ana_ring")[:] = _arr(a, "ana_ring")
            _arr(b, "overlap")[:384] = _arr(a, "overlap")[:384]
            b.st_syn_scale, b.st_lb_scale = a.st_syn_scale, a.st_lb_scale
            b.ov_lb_scale, b.ov_hb_scale = a.ov_lb_scale, a.ov_hb_scale
            want_st[d] = np.frombuffer(bytes(b), np.uint8)
    assert np.array_equal(st.cpu().numpy(), want_st)
    assert np.array_equal(ps.cpu().numpy(), want_ps)
    assert not np.array_equal(want_st, st0) or mode == "ps"


@pytest.mark.gpu
@pytest.mark.parametrize("ch_fac", [1, 2])
def test_apply_side_batch_equals_the_host_functions(ch_fac):
    """xaac_sbr_state_apply_side_batch (the words ixheaacd_sbr_dec_reset / ixheaacd_prepare_upsamp rewrite, sbrdecoder.c:103-276,
    on the resident arrays) against the host library's xaac_sbr_state_apply_side / xaac_ps_state_apply_side on copies of the
    same random states: every flag combination, streams without a flag untouched."""
    import torch
    import libxaac_amd
    from libxaac_amd import decoder
    lib = decoder.load_host_library()
    dev = torch.device("cuda:0")
    ctx = libxaac_amd.XaacContext(0, None)
    rng = np.random.default_rng(11 + ch_fac)
    n = 300   # more than one 256-thread workgroup of channels
    nc = n * ch_fac
    sb, pb, hb = libxaac_amd.SBR_STATE_BYTES, libxaac_amd.PS_STATE_BYTES, libxaac_amd.SBR_HEADER_BYTES
    st0 = rng.integers(0, 256, (nc, sb), dtype=np.uint8)
    ps0 = rng.integers(0, 256, (n, pb), dtype=np.uint8)
    hdr = rng.integers(0, 256, (nc, hb), dtype=np.uint8)
    flags = np.zeros((n, 8), np.int32)
    flags[:, 1] = rng.integers(0, 2, n)          # reset
    flags[:, 2] = rng.integers(0, 3, n)          # reset_channels 0..2
    flags[:, 3] = rng.integers(0, 4, n) == 0     # upsampling
    flags[:, 0], flags[:, 4:] = 1, rng.integers(0, 2, (n, 4))
    st, ps = torch.from_numpy(st0).to(dev), torch.from_numpy(ps0).to(dev)
    ctx.sbr_state_apply_side_batch(torch.from_numpy(hdr).to(dev), torch.from_numpy(flags).to(dev), st, ch_fac,
                                   ps_state=ps if ch_fac == 1 else None)
    ctx.sync()
    want_st, want_ps = st0.copy(), ps0.copy()
    side = decoder.SbrSide()
    for i in range(n):
        for name, v in zip(("apply", "reset", "reset_channels", "upsampling", "stereo", "ps", "ps_start", "frame_ok"), flags[i]):
            setattr(side, name, int(v))
        ctypes.memmove(ctypes.addressof(side) + decoder.SbrSide.header.offset, hdr[i * ch_fac].ctypes.data, hb)
        for ch in range(ch_fac):
            row = np.ascontiguousarray(want_st[i * ch_fac + ch])
            lib.xaac_sbr_state_apply_side(row.ctypes.data, ctypes.byref(side), ch)
            want_st[i * ch_fac + ch] = row
        if ch_fac == 1:
            row = np.ascontiguousarray(want_ps[i])
            lib.xaac_ps_state_apply_side(row.ctypes.data, ctypes.byref(side))
            want_ps[i] = row
    assert np.array_equal(st.cpu().numpy(), want_st)
    assert np.array_equal(ps.cpu().numpy(), want_ps)
    assert not np.array_equal(want_st, st0) and (ch_fac == 2 or not np.array_equal(want_ps, ps0))
    quiet = (flags[:, 1] == 0) & (flags[:, 3] == 0)
    assert quiet.any() and np.array_equal(want_st.reshape(n, -1)[quiet], st0.reshape(n, -1)[quiet])
