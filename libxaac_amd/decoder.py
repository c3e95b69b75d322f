"""libxaac_amd.decoder -- whole-stream decoding with no reference code in the process: the host-side bitstream front end
(libxaac_amd/libxaac_host.so, include/xaac_parse.h: ADTS, AAC-LC syntax, SBR / PS side info, all on the CPU) feeding the
GPU entry points of include/xaac_amd.h (IMDCT, low-power / HQ SBR with parametric stereo, peak limiter), with every stream's
state resident in device memory.  Mirrors what the reference's command line decoder does for `xaacdec -esbr:0` on ADTS
AAC-LC / HE-AAC / HE-AACv2 streams of one or two channels (decoder/ixheaacd_api.c:2624-3788: frame loop, core -> SBR
hand-off :353-366, peak limiter and its delay :3666-3692, the flush of its delay line :2824-2866, mono duplicated to
stereo :3639-3660), so the PCM equals the reference's byte for byte (tests/test_decoder_gpu.py).

`parse_stream` is the CPU half alone (used by the CPU tests and the parser-rate measurement); `decode_streams` decodes
N streams in lock step, one batch of frames per GPU call.  There is no CPU fallback for the GPU half.
"""
import ctypes
import os

import numpy as np

from . import (LIMITER_STATE_BYTES, PCM_LC, PCM_SBR, PS_FRAME_BYTES, PS_STATE_BYTES, SBR_FRAME_BYTES, SBR_HEADER_BYTES,
               SBR_STATE_BYTES, LimiterState, XaacContext, peak_limiter_init)

_HERE = os.path.dirname(os.path.abspath(__file__))
_host = None

TOOL_MS, TOOL_INTENSITY, TOOL_PNS, TOOL_TNS, TOOL_PULSE, TOOL_SHORT, TOOL_ESCAPE = 1, 2, 4, 8, 16, 32, 64
HANDOVER_PS_START = 1


class AdtsHeader(ctypes.Structure):
    # struct xaac_adts_header
    _fields_ = [(n, ctypes.c_int32) for n in ("id", "layer", "protection_absent", "profile", "sr_index", "sampling_rate",
                                              "channel_config", "frame_bytes", "raw_blocks", "header_bytes")]


class CoreFrame(ctypes.Structure):
    # struct xaac_core_frame
    _fields_ = [("n_ch", ctypes.c_int32), ("element_id", ctypes.c_int32), ("common_window", ctypes.c_int32),
                ("sbr_ext_type", ctypes.c_int32), ("sbr_bytes", ctypes.c_int32), ("tools", ctypes.c_int32),
                ("ics", (ctypes.c_int16 * 4) * 2), ("spec", (ctypes.c_int32 * 1024) * 2), ("sbr", ctypes.c_uint8 * 272)]


class SbrSide(ctypes.Structure):
    # struct xaac_sbr_side
    _fields_ = [(n, ctypes.c_int32) for n in ("apply", "reset", "reset_channels", "upsampling", "stereo", "ps", "ps_start",
                                              "frame_ok")] + \
               [("header", ctypes.c_uint8 * SBR_HEADER_BYTES), ("frame", (ctypes.c_uint8 * SBR_FRAME_BYTES) * 2),
                ("ps_frame", ctypes.c_uint8 * PS_FRAME_BYTES)]


def host_library_path():
    return os.path.join(_HERE, "libxaac_host.so")


def load_host_library():
    """the CPU front end; raises when it has not been built (make -C libxaac_amd/host)"""
    global _host
    if _host is None:
        path = host_library_path()
        if not os.path.exists(path):
            raise RuntimeError("libxaac_host.so is not built: make -C libxaac_amd/host")
        lib = ctypes.CDLL(path)
        lib.xaac_parser_create.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        lib.xaac_parser_destroy.argtypes = [ctypes.c_void_p]
        lib.xaac_parser_destroy.restype = None
        lib.xaac_adts_parse_header.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(AdtsHeader)]
        lib.xaac_parse_adts_frame.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int32,
                                              ctypes.POINTER(CoreFrame), ctypes.POINTER(ctypes.c_size_t)]
        lib.xaac_parse_sbr_side.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(SbrSide)]
        for fn in ("xaac_sbr_state_init", "xaac_ps_state_init"):
            getattr(lib, fn).argtypes = [ctypes.c_void_p]
            getattr(lib, fn).restype = None
        lib.xaac_sbr_state_apply_side.argtypes = [ctypes.c_void_p, ctypes.POINTER(SbrSide), ctypes.c_int32]
        lib.xaac_sbr_state_apply_side.restype = None
        lib.xaac_ps_state_apply_side.argtypes = [ctypes.c_void_p, ctypes.POINTER(SbrSide)]
        lib.xaac_ps_state_apply_side.restype = None
        _host = lib
    return _host


class ParseError(RuntimeError):
    def __init__(self, code, frame):
        RuntimeError.__init__(self, "host parser: error %d in frame %d" % (code, frame))
        self.code, self.frame = code, frame


class StreamParser:
    """One ADTS stream through the host front end, frame by frame, the way the reference's decoder walks it: the first
    frame is decoded once while the decoder initialises (ixheaacd_dec_init, api.c:2097: only the PNS random seed
    survives that pass) and then again as the first output frame."""

    def __init__(self, data, with_sbr=None, stage=2):
        self.lib = load_host_library()
        self.data = bytes(data)
        self.buf = (ctypes.c_uint8 * len(self.data)).from_buffer_copy(self.data)
        self.h = ctypes.c_void_p()
        if self.lib.xaac_parser_create(ctypes.byref(self.h)):
            raise RuntimeError("xaac_parser_create failed")
        self.pos, self.frame_no, self.stage = 0, 0, stage
        self.core, self.side, self.used = CoreFrame(), SbrSide(), ctypes.c_size_t()
        hdr = AdtsHeader()
        rc = self.lib.xaac_adts_parse_header(self.buf, len(self.data), ctypes.byref(hdr))
        if rc:
            raise ParseError(rc, 0)
        self.core_rate = hdr.sampling_rate
        rc = self.lib.xaac_parse_adts_frame(self.h, self.buf, len(self.data), stage, ctypes.byref(self.core),
                                            ctypes.byref(self.used))      # the initialisation pass over frame 0
        if rc:
            raise ParseError(rc, 0)
        self.n_ch = self.core.n_ch
        # an SBR decoder exists for streams that carry SBR data or run at 24 kHz and below (implicit signalling, api.c:2160)
        self.sbr = bool(self.core.sbr_bytes > 0 or self.core_rate <= 24000) if with_sbr is None else bool(with_sbr)

    def close(self):
        if self.h:
            self.lib.xaac_parser_destroy(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def next(self):
        """-> False at the end of the stream, True with self.core (and self.side for SBR streams) filled"""
        left = len(self.data) - self.pos
        if left < 7:
            return False
        rc = self.lib.xaac_parse_adts_frame(self.h, ctypes.byref(self.buf, self.pos), left, self.stage, ctypes.byref(self.core),
                                            ctypes.byref(self.used))
        if rc == 1:      # a truncated last frame
            return False
        if rc:
            raise ParseError(rc, self.frame_no)
        self.pos += self.used.value
        if self.sbr:
            rc = self.lib.xaac_parse_sbr_side(self.h, 1, ctypes.byref(self.side))
            if rc:
                raise ParseError(rc, self.frame_no)
        self.frame_no += 1
        return True


def parse_stream(data, stage=2):
    """the CPU half alone: [(spec int32[n_ch, 1024], ics int16[n_ch, 4], tools, side or None)] of every frame"""
    p = StreamParser(data, stage=stage)
    out = []
    while p.next():
        n = p.core.n_ch
        spec = np.ctypeslib.as_array(p.core.spec)[:n].copy()
        ics = np.ctypeslib.as_array(p.core.ics)[:n].copy()
        side = None
        if p.sbr:
            side = SbrSide()
            ctypes.memmove(ctypes.byref(side), ctypes.byref(p.side), ctypes.sizeof(SbrSide))
        out.append((spec, ics, int(p.core.tools), side))
    p.close()
    return out


def _struct_bytes(fn, size):
    raw = (ctypes.c_uint8 * size)()
    fn(raw)
    return np.frombuffer(raw, np.uint8).copy()


def decode_streams(streams, ctx=None, device="cuda:0"):
    """Decodes N ADTS streams of the same kind (all AAC-LC stereo, all HE-AAC stereo, or all HE-AAC / HE-AACv2 mono) in
    lock step: per step one frame of every stream is parsed on the CPU and the whole batch runs through the GPU entry
    points.  -> (list of int16 [samples, 2] arrays, output sampling rate).  Streams that end early drop out of the batch."""
    import torch
    lib = load_host_library()
    dev = torch.device(device)
    own = ctx is None
    if own:   # the context launches on torch's current stream, so that its kernels and torch's copies stay in order
        ctx = XaacContext(dev.index or 0, torch.cuda.current_stream(dev).cuda_stream)
    ps = [StreamParser(d) for d in streams]
    n = len(ps)
    n_ch, sbr, rate = ps[0].n_ch, ps[0].sbr, ps[0].core_rate
    for p in ps:
        if (p.n_ch, p.sbr, p.core_rate) != (n_ch, sbr, rate):
            raise ValueError("decode_streams takes streams of one kind")
    nc = n * n_ch

    def dz(*shape, dtype=torch.uint8):
        return torch.zeros(*shape, dtype=dtype, device=dev)

    overlap, ovl_state = dz(nc, 512, dtype=torch.int32), dz(nc, 2)
    spec_h = torch.zeros(nc, 1024, dtype=torch.int32).pin_memory()
    ics_h = torch.zeros(nc, 2, dtype=torch.uint8).pin_memory()
    spec_d, ics_d = dz(nc, 1024, dtype=torch.int32), dz(nc, 2)
    out = [[] for _ in range(n)]
    if not sbr:
        # AAC-LC: IMDCT -> WORD32 + qshift_adj -> peak limiter -> round16 (api.c:3662-3692)
        out32, qadj = dz(n * 1024 * 2, dtype=torch.int32), dz(n * 2, dtype=torch.int8)
        lim0, delay = peak_limiter_init(2, rate)
        lim = torch.from_numpy(np.tile(np.frombuffer(bytes(lim0), np.uint8), (n, 1)).copy()).to(dev)
        ws = dz(max(ctx.peak_limiter_workspace_bytes(n), 16))
        pcm = dz(n * 1024 * 2, dtype=torch.int16)
    else:
        sbr_state0 = _struct_bytes(lib.xaac_sbr_state_init, SBR_STATE_BYTES)
        state = torch.from_numpy(np.tile(sbr_state0, (nc, 1)).copy()).to(dev)
        core16 = dz(nc * 1024, dtype=torch.int16)
        hdr_h = torch.zeros(nc, SBR_HEADER_BYTES, dtype=torch.uint8).pin_memory()
        frm_h = torch.zeros(nc, SBR_FRAME_BYTES, dtype=torch.uint8).pin_memory()
        hdr_d, frm_d = dz(nc, SBR_HEADER_BYTES), dz(nc, SBR_FRAME_BYTES)
        status = dz(nc, dtype=torch.int32)
        if n_ch == 2:
            ws = dz(ctx.sbr_lp_workspace_bytes(nc))
            pcm = dz(nc * 2048, dtype=torch.int16)
        else:
            ps_state0 = _struct_bytes(lib.xaac_ps_state_init, PS_STATE_BYTES)
            ps_state = torch.from_numpy(np.tile(ps_state0, (n, 1)).copy()).to(dev)
            psf_h = torch.zeros(n, PS_FRAME_BYTES, dtype=torch.uint8).pin_memory()
            psf_d = dz(n, PS_FRAME_BYTES)
            ws = dz(ctx.sbr_hq_workspace_bytes(n, True))
            pcm_ps, pcm_mono = dz(n * 2048 * 2, dtype=torch.int16), dz(n * 2048, dtype=torch.int16)
    alive = list(range(n))
    first = True
    while alive:
        alive = [i for i in alive if ps[i].next()]
        if not alive:
            break
        if len(alive) != n:
            # the batch shrinks only at stream ends: simplest correct handling is to finish the others one by one
            raise NotImplementedError("streams of different lengths: decode them in separate calls")
        for i in alive:
            c = ps[i].core
            spec_h[i * n_ch:(i + 1) * n_ch] = torch.from_numpy(np.ctypeslib.as_array(c.spec)[:n_ch])
            ics_np = np.ctypeslib.as_array(c.ics)[:n_ch, :2].astype(np.uint8)
            ics_h[i * n_ch:(i + 1) * n_ch] = torch.from_numpy(ics_np)
        spec_d.copy_(spec_h, non_blocking=True)
        ics_d.copy_(ics_h, non_blocking=True)
        if not sbr:
            ctx.imdct_process_batch(spec_d, ics_d, overlap, ovl_state, out32=out32, qshift_adj=qadj, ch_fac=n_ch)
            if n_ch == 1:
                raise NotImplementedError("mono AAC-LC without SBR")
            ctx.peak_limiter_process_batch(out32, qadj, lim, 2, ws, pcm16=pcm)
            ctx.sync()
            block = pcm.cpu().numpy().reshape(n, 1024, 2)
            for i in range(n):
                out[i].append(block[i, delay:] if first else block[i])
        else:
            ctx.imdct_process_batch(spec_d, ics_d, overlap, ovl_state, pcm16=core16, ch_fac=n_ch, pcm_mode=PCM_SBR)
            sides = [ps[i].side for i in alive]
            # frames that reset the SBR decoder or fall back to plain up-sampling change a few words of the resident state
            touched = [i for i in alive if ps[i].side.reset or ps[i].side.upsampling]
            if touched:
                ctx.sync()
                st_h = state.cpu().numpy()
                ps_h = ps_state.cpu().numpy() if n_ch == 1 else None
                for i in touched:
                    for c in range(n_ch):
                        row = np.ascontiguousarray(st_h[i * n_ch + c])
                        lib.xaac_sbr_state_apply_side(row.ctypes.data, ctypes.byref(ps[i].side), c)
                        st_h[i * n_ch + c] = row
                    if n_ch == 1:
                        row = np.ascontiguousarray(ps_h[i])
                        lib.xaac_ps_state_apply_side(row.ctypes.data, ctypes.byref(ps[i].side))
                        ps_h[i] = row
                state.copy_(torch.from_numpy(st_h))
                if n_ch == 1:
                    ps_state.copy_(torch.from_numpy(ps_h))
            for k, i in enumerate(alive):
                s = sides[k]
                hdr_np = np.frombuffer(bytes(s.header), np.uint8)
                for c in range(n_ch):
                    hdr_h[i * n_ch + c] = torch.from_numpy(hdr_np.copy())
                    frm_h[i * n_ch + c] = torch.from_numpy(np.frombuffer(bytes(s.frame[c]), np.uint8).copy())
                if n_ch == 1:
                    psf_h[i] = torch.from_numpy(np.frombuffer(bytes(s.ps_frame), np.uint8).copy())
            hdr_d.copy_(hdr_h, non_blocking=True)
            frm_d.copy_(frm_h, non_blocking=True)
            if n_ch == 2:
                ctx.sbr_lp_process_batch(core16, hdr_d, frm_d, state, pcm, ws, status=status, in_ch_fac=2, out_ch_fac=2)
                ctx.sync()
                block = pcm.cpu().numpy().reshape(n, 2048, 2)
                for i in range(n):
                    out[i].append(block[i].copy())
            else:
                with_ps = [bool(ps[i].side.ps) for i in alive]
                if any(with_ps) != all(with_ps):
                    raise NotImplementedError("a batch mixing PS and non-PS frames")
                if all(with_ps):
                    starts = [i for i in alive if ps[i].side.ps_start]
                    if starts:
                        idx = torch.tensor(starts, dtype=torch.int32, device=dev)
                        ctx.sbr_state_handover(HANDOVER_PS_START, idx, idx, state, ps_state)
                    psf_d.copy_(psf_h, non_blocking=True)
                    ctx.sbr_hq_process_batch(core16, hdr_d, frm_d, state, pcm_ps, ws, ps_frame=psf_d, ps_state=ps_state,
                                             status=status)
                    ctx.sync()
                    block = pcm_ps.cpu().numpy().reshape(n, 2048, 2)
                    for i in range(n):
                        out[i].append(block[i].copy())
                else:
                    ctx.sbr_hq_process_batch(core16, hdr_d, frm_d, state, pcm_mono, ws, status=status)
                    ctx.sync()
                    block = pcm_mono.cpu().numpy().reshape(n, 2048)
                    for i in range(n):
                        out[i].append(np.repeat(block[i][:, None], 2, axis=1))   # mono duplicated to stereo (api.c:3639)
            if int(status.min().item()) < 0:
                raise RuntimeError("the SBR kernels refused a frame")
        first = False
    if not sbr:
        # the limiter's delay line holds the last attack_time_samples samples: api.c:2824-2866
        ctx.sync()
        lim_h = lim.cpu().numpy()
        for i in range(n):
            st = LimiterState.from_buffer_copy(lim_h[i].tobytes())
            att, idx = st.attack_time_samples, st.delayed_input_index
            d = np.ctypeslib.as_array(st.delayed_input)[:att * 2].reshape(att, 2)
            tail = np.concatenate([d[idx:], d[:idx]]).astype(np.float64)
            v = tail.astype(np.int64)                         # (WORD32) of the float, then round16
            v = np.clip(v + 0x8000, -(1 << 31), (1 << 31) - 1) >> 16
            out[i].append(v.astype(np.int16))
    for p in ps:
        p.close()
    if own:
        ctx.close()
    return [np.concatenate(o) if o else np.zeros((0, 2), np.int16) for o in out], rate * (2 if sbr else 1)
