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

from . import (ESBR_PS_STATE_BYTES, ESBR_SIDE_BYTES, ESBR_STATE_BYTES, HBE_STATE_BYTES, LIMITER_STATE_BYTES, PCM_LC, PCM_SBR,
               PS_FRAME_BYTES, PS_STATE_BYTES, SBR_FRAME_BYTES, SBR_HEADER_BYTES, SBR_STATE_BYTES, LimiterState, XaacContext,
               peak_limiter_init)

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
    return os.environ.get("XAAC_HOST_LIBRARY") or os.path.join(_HERE, "libxaac_host.so")


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
        for fn in ("xaac_sbr_state_init", "xaac_ps_state_init", "xaac_esbr_state_init", "xaac_esbr_ps_state_init",
                   "xaac_hbe_state_init"):
            getattr(lib, fn).argtypes = [ctypes.c_void_p]
            getattr(lib, fn).restype = None
        lib.xaac_parser_set_esbr.argtypes = [ctypes.c_void_p, ctypes.c_int32]
        lib.xaac_parse_esbr_side.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]
        lib.xaac_hbe_state_reinit.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        lib.xaac_hbe_state_reinit_tails.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32]
        lib.xaac_hbe_state_reinit_tails.restype = ctypes.c_int32
        lib.xaac_parse_reset_pitch.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32)]
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
    """One ADTS stream through the host front end, frame by frame.  (The reference decodes the first frame twice, once
    while it initialises -- ixheaacd_dec_init, api.c:2097 -- and again as the first output frame; every state is set up
    afresh in between, api.c:2141-2170, so nothing of the first pass shows and one pass is all that is needed here.)"""

    def __init__(self, data, with_sbr=None, stage=2, esbr=False):
        self.lib = load_host_library()
        self.esbr = bool(esbr)     # the reference's default -esbr:1 reading of the SBR payload (xaac_parser_set_esbr)
        self.data = bytes(data)
        self.buf = (ctypes.c_uint8 * len(self.data)).from_buffer_copy(self.data)
        self.h = ctypes.c_void_p()
        if self.lib.xaac_parser_create(ctypes.byref(self.h)):
            raise RuntimeError("xaac_parser_create failed")
        if self.esbr and self.lib.xaac_parser_set_esbr(self.h, 1):
            raise RuntimeError("xaac_parser_set_esbr failed")
        self.pos, self.frame_no, self.stage = 0, 0, stage
        self.core, self.side, self.used = CoreFrame(), SbrSide(), ctypes.c_size_t()
        self.esbr_side = [(ctypes.c_uint8 * ESBR_SIDE_BYTES)(), (ctypes.c_uint8 * ESBR_SIDE_BYTES)()]
        hdr = AdtsHeader()
        rc = self.lib.xaac_adts_parse_header(self.buf, len(self.data), ctypes.byref(hdr))
        if rc:
            raise ParseError(rc, 0)
        self.core_rate = hdr.sampling_rate
        probe = ctypes.c_void_p()           # a look at frame 0 with a parser of its own: channels, SBR payload or not
        self.lib.xaac_parser_create(ctypes.byref(probe))
        rc = self.lib.xaac_parse_adts_frame(probe, self.buf, len(self.data), 1, ctypes.byref(self.core), ctypes.byref(self.used))
        self.lib.xaac_parser_destroy(probe)
        if rc:
            raise ParseError(rc, 0)
        self.n_ch = self.core.n_ch
        # the SBR tools run for the frames that carry an SBR payload (api.c:3369-3373: a stream at 24 kHz and below gets an SBR
        # decoder object by implicit signalling, api.c:2160, but without payloads it is never called: plain AAC-LC output)
        self.sbr = bool(self.core.sbr_bytes > 0) if with_sbr is None else bool(with_sbr)

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
            if self.esbr:
                for c in range(self.core.n_ch):
                    if self.lib.xaac_parse_esbr_side(self.h, c, self.esbr_side[c]):
                        raise ParseError(-2, self.frame_no)
        self.frame_no += 1
        return True


def parse_stream(data, stage=2, esbr=False):
    """the CPU half alone: [(spec int32[n_ch, 1024], ics int16[n_ch, 4], tools, side or None)] of every frame; esbr: the
    -esbr:1 reading of the SBR payload, and a fifth member [xaac_esbr_side bytes per channel]"""
    p = StreamParser(data, stage=stage, esbr=esbr)
    out = []
    while p.next():
        n = p.core.n_ch
        spec = np.ctypeslib.as_array(p.core.spec)[:n].copy()
        ics = np.ctypeslib.as_array(p.core.ics)[:n].copy()
        side = None
        if p.sbr:
            side = SbrSide()
            ctypes.memmove(ctypes.byref(side), ctypes.byref(p.side), ctypes.sizeof(SbrSide))
        if esbr:
            out.append((spec, ics, int(p.core.tools), side, [bytes(p.esbr_side[c]) for c in range(n)] if p.sbr else None))
        else:
            out.append((spec, ics, int(p.core.tools), side))
    p.close()
    return out


class _ParseBatch(ctypes.Structure):
    # struct xaac_parse_batch
    _fields_ = [(n, ctypes.c_int32) for n in ("n_streams", "n_ch", "with_sbr", "ps_enable", "stage", "threads")] + \
               [(n, ctypes.c_void_p) for n in ("parser", "data", "bytes", "spec", "ics", "header", "frame", "ps_frame", "flags",
                                               "tools", "consumed", "status", "esbr_side", "reset_pitch", "pos")] + \
               [("frames", ctypes.c_int32), ("lines", ctypes.c_void_p)]


F_APPLY, F_RESET, F_RESET_CHANNELS, F_UPSAMPLING, F_STEREO, F_PS, F_PS_START, F_FRAME_OK = range(8)


def usable_cores():
    """host cores this process may really use: the scheduler affinity mask cut by the cgroup CPU quota (v2 cpu.max or v1
    cfs_quota_us / cfs_period_us) when there is one"""
    try:
        aff = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        aff = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
            quota = None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, per = float(f.read()), float(g.read())
                quota = None if q <= 0 else q / per
        except (OSError, ValueError):
            pass
    return aff if quota is None else max(1, min(aff, int(quota + 0.5)))


def gpu_numa_cpus(device_index):
    """CPUs of the NUMA node the GPU hangs off (hipDeviceGetPCIBusId -> /sys/bus/pci/devices/<id>/numa_node -> the node's
    cpulist), cut by what the process may run on; None where that cannot be read (no sysfs, a single node, node -1)"""
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, int(device_index)) != 0:
            return None
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % buf.value.decode().lower()).read())
        if node < 0:
            return None
        cpus = []
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            a, _, b = part.partition("-")
            cpus += range(int(a), int(b or a) + 1)
        cpus = sorted(set(cpus) & os.sched_getaffinity(0))
        return cpus or None
    except (OSError, ValueError, AttributeError):
        return None


class _NearGpu:
    """While pinned staging memory is allocated and first touched: the calling thread on the GPU's NUMA node, so that the
    pages land there.  With the staging on the other socket the bus carries one direction at full rate but both at once
    (a step's spectra going up beside the PCM of the step before coming down) at 38 GiB/s in total instead of 63 (measured
    on a two-socket MI355X host with 64 MiB copies)."""

    def __init__(self, device_index):
        self.cpus = gpu_numa_cpus(device_index)

    def __enter__(self):
        self.before = None
        if self.cpus:
            try:
                self.before = os.sched_getaffinity(0)
                os.sched_setaffinity(0, self.cpus)
            except OSError:
                self.before = None
        return self

    def __exit__(self, *exc):
        if self.before is not None:
            os.sched_setaffinity(0, self.before)
        return False


class _TorchCpuThreads:
    """For the length of a decode: torch's intra-op CPU pool no larger than the cores the process is granted.  torch sizes
    the pool by the CPUs the machine lists; in a container that lists 256 and grants 16, one parallel fill of a staging
    array wakes a pool whose threads then spin through the cgroup's CPU quota, and the parser threads behind it are throttled:
    measured 1.1-1.6 x 10^6 frames/s end to end with the default pool, 3.6 x 10^6 with it capped (same box, same run)."""

    def __enter__(self):
        import torch
        self.torch, self.before = torch, torch.get_num_threads()
        cap = usable_cores()
        if self.before > cap:
            torch.set_num_threads(cap)
        return self

    def __exit__(self, *exc):
        if self.torch.get_num_threads() != self.before:
            self.torch.set_num_threads(self.before)
        return False


class BatchParser:
    """N ADTS streams of one kind through the host front end in lock step: xaac_parse_batch_run parses one frame of every
    stream on a team of CPU threads, straight into the (pinned) staging arrays handed to step()."""

    def __init__(self, streams, threads=0, stage=2, esbr=False):
        self.lib = load_host_library()
        self.lib.xaac_parse_batch_run.argtypes = [ctypes.c_void_p]       # (the original layout's symbols: no pos / frames / lines)
        self.lib.xaac_parse_batch_start.argtypes = [ctypes.c_void_p]
        self.lib.xaac_parse_batch_run_sized.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
        self.lib.xaac_parse_batch_start_sized.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
        self.lib.xaac_parse_batch_wait.argtypes = [ctypes.c_void_p]
        self.esbr = bool(esbr)
        self.n = n = len(streams)
        self.length = np.array([len(d) for d in streams], np.uint64)
        self.start = np.concatenate([[0], np.cumsum(self.length)[:-1]]).astype(np.uint64)
        self.blob = np.frombuffer(b"".join(bytes(d) for d in streams) + b"\0" * 16, np.uint8).copy()
        self.base = self.blob.ctypes.data
        self.pos = np.zeros(n, np.uint64)
        self._ptrs = (np.uint64(self.base) + self.start).astype(np.uint64)   # every stream's first byte
        self.parsers = (ctypes.c_void_p * n)()
        for i in range(n):
            h = ctypes.c_void_p()
            if self.lib.xaac_parser_create(ctypes.byref(h)):
                raise RuntimeError("xaac_parser_create failed")
            if self.esbr:
                self.lib.xaac_parser_set_esbr(h, 1)
            self.parsers[i] = h
        self.threads, self.stage = int(threads), int(stage)
        self.consumed, self.status = np.zeros(n, np.uint64), np.zeros(n, np.int32)
        self.reset_pitch = np.zeros(n, np.int32)   # at frames with a reset flag: xaac_parse_reset_pitch
        self.frames = np.zeros(n, np.int64)
        hdr = AdtsHeader()
        rc = self.lib.xaac_adts_parse_header(bytes(streams[0][:16]), min(16, len(streams[0])), ctypes.byref(hdr))
        if rc:
            raise ParseError(rc, 0)
        self.core_rate, self.n_ch = hdr.sampling_rate, (2 if hdr.channel_config == 2 else 1)
        core = CoreFrame()
        used = ctypes.c_size_t()
        probe = ctypes.c_void_p()
        self.lib.xaac_parser_create(ctypes.byref(probe))
        rc = self.lib.xaac_parse_adts_frame(probe, bytes(streams[0]), len(streams[0]), 1, ctypes.byref(core), ctypes.byref(used))
        self.lib.xaac_parser_destroy(probe)
        if rc:
            raise ParseError(rc, 0)
        self.sbr = bool(core.sbr_bytes > 0)

    def close(self):
        if getattr(self, "_in_flight", False):   # (a caller that gave up between start_step() and wait_step(): the team must get its batch back)
            self.lib.xaac_parse_batch_wait(None)
            self._in_flight = False
        for i in range(self.n):
            if self.parsers[i]:
                self.lib.xaac_parser_destroy(self.parsers[i])
                self.parsers[i] = None

    def _descriptor(self, spec, ics, hdr, frm, psf, flags, with_sbr, eside=None, status=None, reset_pitch=None, frames=1, lines=None):
        # data / bytes are the whole streams and stay as they are; the library moves self.pos (xaac_parse_batch::pos), so a
        # call costs this thread the filling of the descriptor and nothing per stream
        b = _ParseBatch()
        b.n_streams, b.n_ch, b.with_sbr, b.ps_enable, b.stage, b.threads = self.n, self.n_ch, int(with_sbr), 1, self.stage, self.threads
        b.parser, b.data, b.bytes = ctypes.addressof(self.parsers), self._ptrs.ctypes.data, self.length.ctypes.data
        ptr = lambda t: None if t is None else (t.data_ptr() if hasattr(t, "data_ptr") else t.ctypes.data)
        b.spec, b.ics, b.header, b.frame, b.ps_frame, b.flags = ptr(spec), ptr(ics), ptr(hdr), ptr(frm), ptr(psf), ptr(flags)
        b.tools, b.consumed = None, self.consumed.ctypes.data   # (which tools a frame used: nobody downstream asks)
        b.status = (self.status if status is None else status).ctypes.data
        b.esbr_side = ptr(eside)
        b.reset_pitch = (self.reset_pitch if reset_pitch is None else reset_pitch).ctypes.data
        b.pos = self.pos.ctypes.data
        b.frames = int(frames)
        b.lines = None if lines is None else lines.ctypes.data
        return b

    def _advance(self, ok, status=None):
        """one step's status words -> bool[n] (which streams delivered a frame); ok: the library call's return value, or None"""
        if ok is not None and ok < 0:
            raise RuntimeError("xaac_parse_batch: %d" % ok)
        status = self.status if status is None else status
        good = status == 0
        self.frames += good
        if not good.all():
            bad = ~good & (status != 1)
            if np.any(bad):
                i = int(np.nonzero(bad)[0][0])
                raise ParseError(int(status[i]), int(self.frames[i]))
        return good

    def step(self, spec, ics, hdr=None, frm=None, psf=None, flags=None, eside=None):
        """parses the next frame of every stream into the staging arrays; -> bool[n]: which streams delivered a frame
        (the others are at their end: their rows are left as they were)"""
        b = self._descriptor(spec, ics, hdr, frm, psf, flags, self.sbr, eside)
        return self._advance(self.lib.xaac_parse_batch_run_sized(ctypes.byref(b), ctypes.sizeof(b)))

    def start_step(self, spec, ics, hdr=None, frm=None, psf=None, flags=None, eside=None, status=None, reset_pitch=None, frames=1,
                   lines=None):
        """step() in two halves (xaac_parse_batch_start / _wait): the library's worker team parses while the caller does
        something else; the staging arrays are the team's until wait_step() returns.  status / reset_pitch: the caller's own
        int32[n] arrays for this step's results (a caller that starts the next step before it has looked at this one's).
        frames = T > 1: up to T consecutive frames of every stream in one call (xaac_parse_batch::frames), every array with
        a leading dimension T (status / reset_pitch int32[T, n]); finish_step is then called per step t with status[t].
        lines: int32[T, n] out, xaac_parse_batch::lines."""
        self._status_in_flight = status
        b = self._descriptor(spec, ics, hdr, frm, psf, flags, self.sbr, eside, status, reset_pitch, frames, lines)
        rc = self.lib.xaac_parse_batch_start_sized(ctypes.byref(b), ctypes.sizeof(b))
        if rc:
            raise RuntimeError("xaac_parse_batch_start: %d" % rc)
        self._in_flight = True

    def wait_step(self, check=True):
        """-> (bool[n] as step(), seconds the team parsed); check False: (parsed-ok count, seconds) -- the caller looks at its
        status array itself (finish_step) after it has started the next step"""
        busy = ctypes.c_double(0.0)
        ok = self.lib.xaac_parse_batch_wait(ctypes.byref(busy))
        self._in_flight = False
        if not check:
            return ok, busy.value
        return self._advance(ok, self._status_in_flight), busy.value

    def finish_step(self, ok, status):
        """what wait_step(check=True) does behind the library call, for a step waited for with check=False (ok: the call's
        return value, or None for one step of a call over several frames: the status words decide)"""
        return self._advance(ok, status)


def _struct_bytes(fn, size):
    raw = (ctypes.c_uint8 * size)()
    fn(raw)
    return np.frombuffer(raw, np.uint8).copy()


# float offsets of members of struct xaac_esbr_state (include/xaac_esbr.h; tests/test_parser_esbr.py checks them against the
# ctypes mirror of the header): qmf_re / qmf_im rows, and the transposer's last rows ph_re / ph_im
_ES_QMF_RE, _ES_QMF_IM, _ES_PH_RE, _ES_PH_IM = 1604, 4164, 8479, 8991


def decode_streams(streams, ctx=None, device="cuda:0", threads=0, keep_pcm=True, timing=None, overlap=True, esbr=False,
                   _trace=None, frames_per_parse=4):
    """Decodes N ADTS streams of the same kind (all AAC-LC stereo, all HE-AAC stereo, or all HE-AAC / HE-AACv2 mono) in
    lock step: per step one frame of every stream is parsed on CPU threads into pinned staging arrays, copied to the GPU
    (spectra + window info, SBR / PS side info: nothing else crosses the bus on the way in), run through the GPU entry
    points against the streams' device-resident states, and the PCM copied back.
    -> (list of int16 [samples, channels] arrays, output sampling rate).  keep_pcm False: the PCM still comes back to the host
    every step but is not collected (throughput measurements); timing: a dict that receives seconds per stage.
    overlap: the host parses the next steps (further sets of staging arrays; the parser library's own threads,
    xaac_parse_batch_start / _wait) while this thread queues these on the GPU.
    frames_per_parse: frames of every stream per parser call (xaac_parse_batch::frames): a stream's parser state and bytes are
    fetched once for T frames -- on the GPU box's host the parser ran 15-40 % faster with 2..8 than with 1 -- and the T steps go
    through the GPU one after the other.
    esbr: decode SBR streams the way the reference does with its default flags (-esbr:1, "Path A": the float eSBR tools of
    xaac_esbr_sbr_process_batch with the QMF harmonic transposer and float parametric stereo; the SBR payload runs one
    frame late, and the reference's command line decoder does not write the first frame's output,
    test/decoder/ixheaacd_main.c:2181-2186) instead of -esbr:0.  AAC-LC streams decode
    the same either way.  SBR header changes in the middle of a stream are followed as the reference follows them (the
    reset-time transposer runs, sbrdecoder.c:196-236, read 24 rows of the QMF history of the frame before: kept beside the
    state)."""
    with _TorchCpuThreads():
        return _decode_streams(streams, ctx, device, threads, keep_pcm, timing, overlap, esbr, _trace, frames_per_parse)


def _decode_streams(streams, ctx, device, threads, keep_pcm, timing, overlap, esbr, _trace, frames_per_parse):
    import time
    import torch
    lib = load_host_library()
    dev = torch.device(device)
    own = ctx is None
    if own:   # the context launches on torch's current stream, so that its kernels and torch's copies stay in order
        ctx = XaacContext(dev.index or 0, torch.cuda.current_stream(dev).cuda_stream)
    bp = BatchParser(streams, threads=threads, esbr=esbr)
    n, n_ch, sbr, rate = bp.n, bp.n_ch, bp.sbr, bp.core_rate
    esbr = bool(esbr) and sbr
    nc = n * n_ch
    t_parse = t_gpu = t_wait_parse = t_wait_down = 0.0

    def dz(*shape, dtype=torch.uint8):
        return torch.zeros(*shape, dtype=dtype, device=dev)

    near_gpu = _NearGpu(dev.index or 0)

    def pinned(*shape, dtype=torch.uint8):
        with near_gpu:         # allocated and first touched on the GPU's NUMA node
            t = torch.empty(*shape, dtype=dtype, pin_memory=True)
            t.numpy().fill(0)  # (numpy: one thread; torch.zeros would wake the whole intra-op pool for it)
        return t

    out_ch = 2 if sbr else n_ch     # SBR streams come out in stereo (PS, or the mono column twice); AAC-LC as coded
    ovl, ovl_state = dz(nc, 512, dtype=torch.int32), dz(nc, 2)
    # two sets of device input arrays: step k + 1 is copied up (its own stream) while step k's kernels read theirs
    spec_d2, ics_d2 = [dz(nc, 1024, dtype=torch.int32) for _ in range(2)], [dz(nc, 2) for _ in range(2)]
    hdr_d2 = frm_d2 = eside_d2 = psf_d2 = flags_d2 = None
    out = [[] for _ in range(n)]

    T = max(1, int(frames_per_parse))

    class Staging:    # what one parser call leaves for the GPU: pinned host arrays of T steps (one frame of every stream each)
        def __init__(self):
            self.spec, self.ics = pinned(T, nc, 1024, dtype=torch.int32), pinned(T, nc, 2)
            self.hdr = self.frm = self.psf = self.flags = self.eside = self.flags_pin = None
            if esbr:
                self.eside = pinned(T, nc, ESBR_SIDE_BYTES)
            if sbr:
                self.hdr, self.frm = pinned(T, nc, SBR_HEADER_BYTES), pinned(T, nc, SBR_FRAME_BYTES)
                self.flags = np.zeros((T, n, 8), np.int32)
                self.flags_pin = pinned(T, n, 8, dtype=torch.int32)   # the rows as they go up for xaac_sbr_state_apply_side_batch
                if n_ch == 1:
                    self.psf = pinned(T, n, PS_FRAME_BYTES)
            self.got, self.seconds = None, 0.0
            self.sent = [torch.cuda.Event() for _ in range(T)]   # step t's copies up are over (sent[T - 1]: the parser may write the set again)
            self.sent_once = False
            self.status, self.reset_pitch = np.zeros((T, n), np.int32), np.zeros((T, n), np.int32)   # this set's own
            self.lines = np.zeros((T, n), np.int32)   # leading spectral lines that may be non-zero, per step and stream

        def begin(self):    # the library's team parses into this set while the caller queues the steps before on the GPU
            if self.sent_once:
                self.sent[T - 1].synchronize()   # (the set's last copies up: long over when its turn comes again)
            bp.start_step(self.spec, self.ics, self.hdr, self.frm, self.psf, self.flags, self.eside, status=self.status,
                          reset_pitch=self.reset_pitch, frames=T, lines=self.lines)
            return self

        def end(self):      # back from the team; the results are looked at in finish(), once the next set is on its way
            self.ok, self.seconds = bp.wait_step(check=False)
            return self

        def finish(self):
            self.got = [bp.finish_step(None, self.status[t]) for t in range(T)]
            return self

        def step(self, t):  # what the loop below sees of step t
            pick = lambda a: None if a is None else a[t]
            v = _Step()
            v.spec, v.ics, v.hdr, v.frm, v.psf, v.eside = (pick(self.spec), pick(self.ics), pick(self.hdr), pick(self.frm),
                                                            pick(self.psf), pick(self.eside))
            v.flags, v.flags_pin, v.reset_pitch, v.got, v.sent, v.owner = (pick(self.flags), pick(self.flags_pin),
                                                                         self.reset_pitch[t], self.got[t], self.sent[t], self)
            v.lines = self.lines[t]
            return v

    class _Step:
        pass

    # three staging sets: the parse of step k + 1 | the copies up and kernels of step k | the copy down of step k - 1
    sets = [Staging(), Staging(), Staging()] if overlap else [Staging()]
    if not sbr:
        # AAC-LC: IMDCT -> WORD32 + qshift_adj -> peak limiter -> round16 (api.c:3662-3692)
        out32, qadj = dz(n * 1024 * n_ch, dtype=torch.int32), dz(n * n_ch, dtype=torch.int8)
        lim0, delay = peak_limiter_init(n_ch, rate)
        lim = torch.from_numpy(np.tile(np.frombuffer(bytes(lim0), np.uint8), (n, 1)).copy()).to(dev)
        lim_at_end, lim_taken = {}, np.zeros(n, bool)
        ws = dz(max(ctx.peak_limiter_workspace_bytes(n), 16))
        pcm2 = [dz(n * 1024 * n_ch, dtype=torch.int16) for _ in range(2)]
        pcm_h2 = [pinned(n * 1024 * n_ch, dtype=torch.int16) for _ in range(2)]
        status2 = status_h2 = None
    elif esbr:
        # Path A: IMDCT (16-bit core PCM) -> xaac_esbr_core_from_pcm16_batch -> the eSBR chain -> xaac_esbr_pcm16_from_float_batch
        # (saturate / truncate to 16 bit: ixheaacd_samples_sat, decode_main.c:82-107); every state member stays on the device
        state = torch.from_numpy(np.tile(_struct_bytes(lib.xaac_esbr_state_init, ESBR_STATE_BYTES), (nc, 1)).copy()).to(dev)
        hbe = dz(nc, HBE_STATE_BYTES)
        core16 = dz(nc * 1024, dtype=torch.int16)
        hdr_d2, frm_d2 = [dz(nc, SBR_HEADER_BYTES) for _ in range(2)], [dz(nc, SBR_FRAME_BYTES) for _ in range(2)]
        eside_d2 = [dz(nc, ESBR_SIDE_BYTES) for _ in range(2)]
        status2 = [dz(nc, dtype=torch.int32) for _ in range(2)]
        status_h2 = [pinned(nc, dtype=torch.int32) for _ in range(2)]
        ws = dz(ctx.esbr_workspace_bytes(nc))
        out_l, out_r = dz(nc, 2048, dtype=torch.float32), None
        core = dz(nc, 1024, dtype=torch.float32)
        pcm2 = [dz(n * 2048 * 2, dtype=torch.int16) for _ in range(2)]
        pcm_h2 = [pinned(n * 2048 * 2, dtype=torch.int16) for _ in range(2)]
        if n_ch == 1:
            ps_state = torch.from_numpy(np.tile(_struct_bytes(lib.xaac_esbr_ps_state_init, ESBR_PS_STATE_BYTES), (n, 1)).copy()).to(dev)
            psf_d2 = [dz(n, PS_FRAME_BYTES) for _ in range(2)]
            out_r = dz(n, 2048, dtype=torch.float32)
        older = dz(nc, 2, 24 * 64, dtype=torch.float32)   # rows 8..31 of the QMF history as the frame before found them
        hbe_tail = np.zeros((nc, 48), np.uint8)            # the transposers' integers (struct xaac_hbe_state from synth_size on)

        def hbe_hint():   # the largest transposer bank of the batch, as the ABI's LDS hint takes it (8, or 0 = any)
            return 8 if int(hbe_tail.view(np.int32)[:, 0].max()) <= 8 else 0
    else:
        state = torch.from_numpy(np.tile(_struct_bytes(lib.xaac_sbr_state_init, SBR_STATE_BYTES), (nc, 1)).copy()).to(dev)
        core16 = dz(nc * 1024, dtype=torch.int16)
        hdr_d2, frm_d2 = [dz(nc, SBR_HEADER_BYTES) for _ in range(2)], [dz(nc, SBR_FRAME_BYTES) for _ in range(2)]
        flags_d2 = [dz(n, 8, dtype=torch.int32) for _ in range(2)]
        status2 = [dz(nc, dtype=torch.int32) for _ in range(2)]
        status_h2 = [pinned(nc, dtype=torch.int32) for _ in range(2)]
        pcm_h2 = [pinned(n * 2048 * 2, dtype=torch.int16) for _ in range(2)]
        pcm2 = [dz(n * 2048 * 2, dtype=torch.int16) for _ in range(2)]
        if n_ch == 2:
            ws = dz(ctx.sbr_lp_workspace_bytes(nc))
        else:
            ps_state = torch.from_numpy(np.tile(_struct_bytes(lib.xaac_ps_state_init, PS_STATE_BYTES), (n, 1)).copy()).to(dev)
            psf_d2 = [dz(n, PS_FRAME_BYTES) for _ in range(2)]
            ws = dz(ctx.sbr_hq_workspace_bytes(n, True))
            pcm_mono = dz(n * 2048, dtype=torch.int16)
    first = True
    cur_set, t_in_set, pending = None, 0, None
    if overlap:
        # (the parse of the next steps runs on the parser library's own threads, xaac_parse_batch_start / _wait: a Python helper
        # thread would have to win the interpreter lock from this one, which gives it up only for microseconds at a time
        # while it queues copies and launches -- measured, its parse began when this thread blocked on the result)
        pending = sets[0].begin()
    which = 0
    # The copy down of step k runs on a second stream beside the copies up and kernels of step k + 1 (two PCM / status sets);
    # the host takes a step's PCM one step later.
    main_stream, down, up = torch.cuda.current_stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    done = [torch.cuda.Event(), torch.cuda.Event()]
    computed = [torch.cuda.Event(), torch.cuda.Event()]
    lines_held = [0, 0]   # per device input set: the leading spectral lines that may be non-zero there
    hip_rt = ctypes.CDLL("libamdhip64.so")
    hip_rt.hipMemcpy2DAsync.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t,
                                        ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    waiting = None    # (slot, got, shape, cut, drop) of the step whose PCM is on its way

    def consume():
        nonlocal waiting, t_wait_down
        if waiting is None:
            return
        slot_, got_, shape_, cut_, drop_ = waiting
        waiting = None
        t_c = time.perf_counter()
        done[slot_].synchronize()
        t_wait_down += time.perf_counter() - t_c
        # (rows of streams that are over run idle on whatever their staging rows hold -- possibly nothing the kernels accept:
        # what they say about those is not looked at)
        if status_h2 is not None and int(status_h2[slot_].numpy().reshape(n, -1)[got_].min(initial=0)) < 0:
            raise RuntimeError("the SBR kernels refused a frame")
        if keep_pcm and not drop_:
            block = pcm_h2[slot_].numpy().reshape(shape_)
            for i in np.nonzero(got_)[0]:
                out[i].append(block[i, cut_:].copy())

    def hand_down(slot_, got_, shape_, cut_=0, drop_=False):
        nonlocal waiting
        ev = computed[slot_]
        ev.record(main_stream)   # this step's kernels are queued: its input set may be refilled, its PCM may go down
        with torch.cuda.stream(down):
            down.wait_event(ev)
            pcm_h2[slot_].copy_(pcm2[slot_], non_blocking=True)
            if status2 is not None:
                status_h2[slot_].copy_(status2[slot_], non_blocking=True)
            done[slot_].record(down)
        consume()
        waiting = (slot_, got_, shape_, cut_, drop_)
        if not overlap:   # one staging set: its copies up must be over before the next parse writes it
            consume()

    step_no = 0
    t_steps = time.perf_counter()
    try:
        while True:
            if t_in_set == T or cur_set is None:   # the next parser call's frames
                t_w = time.perf_counter()
                if not overlap:
                    pending = sets[0].begin()
                cur_set = pending.end()
                t_wait_parse += time.perf_counter() - t_w
                if overlap:    # the next T steps' frames are parsed while this thread looks at these and queues them on the GPU
                    which = (which + 1) % 3
                    pending = sets[which].begin()
                cur_set.finish()
                t_parse += cur_set.seconds
                t_in_set = 0
            cur = cur_set.step(t_in_set)
            t_in_set += 1
            got = cur.got
            if not got.any():
                if overlap:
                    pending.end()   # (every stream is over: that call found nothing to parse)
                break
            slot = step_no & 1
            step_no += 1
            pcm = pcm2[slot]
            status = status2[slot] if status2 is not None else None
            spec_h, ics_h, hdr_h, frm_h, psf_h, flags = cur.spec, cur.ics, cur.hdr, cur.frm, cur.psf, cur.flags
            overlap_buf = ovl
            t0 = time.perf_counter()
            pick = lambda two: None if two is None else two[slot]
            spec_d, ics_d, hdr_d, frm_d, eside_d, psf_d, flags_d = (pick(spec_d2), pick(ics_d2), pick(hdr_d2), pick(frm_d2),
                                                                    pick(eside_d2), pick(psf_d2), pick(flags_d2))
            ps_frames = bool(sbr and n_ch == 1 and (flags[got, F_PS] != 0).all())
            side_words = bool(sbr and not esbr and (got & ((flags[:, F_RESET] != 0) | (flags[:, F_UPSAMPLING] != 0))).any())
            if side_words:   # xaac_sbr_state_apply_side_batch's flag rows (streams without a frame: zero rows)
                cur.flags_pin.numpy()[:] = flags * got[:, None].astype(np.int32)
            with torch.cuda.stream(up):   # everything this step sends up, beside the kernels of the step before
                if step_no > 2:
                    up.wait_event(computed[slot])   # (the kernels that read this input set two steps ago)
                # the spectra: only the leading lines that are not zero in every delivered row (AAC + SBR streams code the lower
                # half of the spectrum or less; 16 of the 26 MB a step of 4096 HE-AACv2 streams sends up are spectra), and
                # what this device set still holds beyond them from two steps ago (the host rows are zero there)
                lines_now = min(1024, (int(cur.lines[got].max()) + 63) & ~63)
                width = max(lines_now, lines_held[slot])
                lines_held[slot] = lines_now
                if width >= 1024:
                    spec_d.copy_(spec_h, non_blocking=True)
                elif width > 0:
                    rc = hip_rt.hipMemcpy2DAsync(spec_d.data_ptr(), 4096, spec_h.data_ptr(), 4096, 4 * width, nc, 1, up.cuda_stream)
                    if rc != 0:
                        raise RuntimeError("hipMemcpy2DAsync: %d" % rc)
                ics_d.copy_(ics_h, non_blocking=True)
                if sbr:
                    hdr_d.copy_(hdr_h, non_blocking=True)
                    frm_d.copy_(frm_h, non_blocking=True)
                    if esbr:
                        eside_d.copy_(cur.eside, non_blocking=True)
                    if ps_frames:
                        psf_d.copy_(psf_h, non_blocking=True)
                    if side_words:
                        flags_d.copy_(cur.flags_pin, non_blocking=True)
                cur.sent.record(up)
            cur.owner.sent_once = True
            main_stream.wait_event(cur.sent)
            if not sbr:
                # a stream that ended with the step before: its limiter state as its last frame left it (the rows of ended
                # streams go on running idle through the kernels; the delay line flushed behind a stream is the one it ended on)
                for i in np.nonzero(~got & ~lim_taken)[0]:
                    lim_at_end[int(i)] = lim[int(i)].cpu().numpy()     # (waits for the kernels of the step before)
                    lim_taken[i] = True
                ctx.imdct_process_batch(spec_d, ics_d, overlap_buf, ovl_state, out32=out32, qshift_adj=qadj, ch_fac=n_ch)
                ctx.peak_limiter_process_batch(out32, qadj, lim, n_ch, ws, pcm16=pcm)
                hand_down(slot, got, (n, 1024, n_ch), cut_=delay if first else 0)   # the limiter's delay is cut from the first frame
            elif esbr:
                # (interleaved as the reference holds it: its in-place 32 -> 16 bit conversion of a pair leaves traces of channel
                # 0 in channel 1, api.c:353-366, which the IMDCT's PCM_SBR hand-off restates for ch_fac 2)
                ctx.imdct_process_batch(spec_d, ics_d, overlap_buf, ovl_state, pcm16=core16, ch_fac=n_ch, pcm_mode=PCM_SBR)
                touched = np.nonzero(got & (flags[:, F_RESET] != 0))[0]
                if touched.size:
                    # ixheaacd_sbr_dec_reset for Path A (sbrdecoder.c:175-236): new transposer parameters from the header's band
                    # tables (its two delay lines cleared), then two transposer runs over rows 8..39 and 40..71 of the QMF buffer
                    # as the frame before left it: rows 8..31 are what that frame found as rows 8..31 of its history (`older`),
                    # rows 32..71 are the state's history (the codec bank's num_time_slots is 32 here).  The second run's last eight output rows become the
                    # state's ph rows (bands outside the transposer's range keep what they held).
                    k = touched.size * n_ch
                    rows_h = (touched[:, None] * n_ch + np.arange(n_ch)[None, :]).ravel()
                    rows = torch.from_numpy(rows_h).to(dev)
                    hb = hbe.index_select(0, rows)
                    tail_off = HBE_STATE_BYTES - 48
                    tails = np.ascontiguousarray(hbe_tail[rows_h])            # (one call for all of them: every stream's first
                    heads = np.ascontiguousarray(hdr_h.numpy()[rows_h])       #  frame is a reset frame)
                    bad = lib.xaac_hbe_state_reinit_tails(tails.ctypes.data, heads.ctypes.data, len(rows_h))
                    if bad >= 0:
                        raise RuntimeError("the QMF transposer refused the SBR band tables of stream %d" % (rows_h[bad] // n_ch))
                    hbe_tail[rows_h] = tails
                    hb[:, tail_off:] = torch.from_numpy(tails).to(dev)
                    hb32 = hb.view(torch.float32)
                    hb32[:, 1088:1088 + 1280 + 640] = 0.0          # synth_buf, analy_buf (behind input_buf[1024 + 64])
                    pitch = torch.from_numpy(np.repeat(cur.reset_pitch[touched], n_ch).astype(np.int32)).to(dev)
                    st32 = state.view(torch.float32)
                    hist, old = st32.index_select(0, rows), older.index_select(0, rows)
                    q_re, q_im = dz(k, 32, 64, dtype=torch.float32), dz(k, 32, 64, dtype=torch.float32)
                    pv_re, pv_im = dz(k, 32, 64, dtype=torch.float32), dz(k, 32, 64, dtype=torch.float32)
                    rst = dz(k, dtype=torch.int32)
                    q_re[:, :24] = old[:, 0].view(k, 24, 64)
                    q_im[:, :24] = old[:, 1].view(k, 24, 64)
                    q_re[:, 24:] = hist[:, _ES_QMF_RE:_ES_QMF_RE + 8 * 64].view(k, 8, 64)
                    q_im[:, 24:] = hist[:, _ES_QMF_IM:_ES_QMF_IM + 8 * 64].view(k, 8, 64)
                    ctx.hbe_apply_batch(q_re, q_im, hb, pv_re, pv_im, status=rst, pitch_in_bins=pitch, max_synth_size=hbe_hint())
                    q_re[:] = hist[:, _ES_QMF_RE + 8 * 64:_ES_QMF_RE + 40 * 64].view(k, 32, 64)
                    q_im[:] = hist[:, _ES_QMF_IM + 8 * 64:_ES_QMF_IM + 40 * 64].view(k, 32, 64)
                    pv_re[:, 24:] = hist[:, _ES_PH_RE:_ES_PH_RE + 512].view(k, 8, 64)
                    pv_im[:, 24:] = hist[:, _ES_PH_IM:_ES_PH_IM + 512].view(k, 8, 64)
                    ctx.hbe_apply_batch(q_re, q_im, hb, pv_re, pv_im, status=rst, pitch_in_bins=pitch, max_synth_size=hbe_hint())
                    hist[:, _ES_PH_RE:_ES_PH_RE + 512] = pv_re[:, 24:].reshape(k, 512)
                    hist[:, _ES_PH_IM:_ES_PH_IM + 512] = pv_im[:, 24:].reshape(k, 512)
                    st32.index_copy_(0, rows, hist)
                    hbe.index_copy_(0, rows, hb)
                st32 = state.view(torch.float32)
                older[:, 0] = st32[:, _ES_QMF_RE + 8 * 64:_ES_QMF_RE + 32 * 64]   # for the reset a later frame may bring
                older[:, 1] = st32[:, _ES_QMF_IM + 8 * 64:_ES_QMF_IM + 32 * 64]
                ctx.esbr_core_from_pcm16(core16, core, ch_fac=n_ch)
                if _trace is not None:   # debugging: the device states in front of the chain call
                    _trace(dict(state=state, hbe=hbe, ps_state=ps_state if n_ch == 1 else None, core=core, side=eside_d, header=hdr_d,
                                frame=frm_d))
                with_ps = (flags[got, F_PS] != 0) if n_ch == 1 else np.zeros(1, bool)
                if with_ps.any() != with_ps.all():
                    raise NotImplementedError("a batch mixing PS and non-PS frames")
                if with_ps.all():
                    ctx.esbr_sbr_process_batch(core, hdr_d, frm_d, eside_d, state, out_l, ws, status=status, ps_frame=psf_d,
                                               ps_state=ps_state, out_r=out_r, hbe_state=hbe, hbe_max_synth_size=hbe_hint())
                    ctx.esbr_pcm16_from_float(out_l, out_r, pcm)
                elif n_ch == 1:
                    ctx.esbr_sbr_process_batch(core, hdr_d, frm_d, eside_d, state, out_l, ws, status=status, hbe_state=hbe, hbe_max_synth_size=hbe_hint())
                    ctx.esbr_pcm16_from_float(out_l, out_l, pcm)                                # mono twice (api.c:3639-3660)
                else:
                    ctx.esbr_sbr_process_batch(core, hdr_d, frm_d, eside_d, state, out_l, ws, status=status, hbe_state=hbe, hbe_max_synth_size=hbe_hint())
                    ctx.esbr_pcm16_from_float(out_l, out_l[1:], pcm, stride=4096)
                hand_down(slot, got, (n, 2048, 2), drop_=first)      # the first frame's output is not written in this mode
            else:
                ctx.imdct_process_batch(spec_d, ics_d, overlap_buf, ovl_state, pcm16=core16, ch_fac=n_ch, pcm_mode=PCM_SBR)
                # frames that reset the SBR decoder or fall back to plain up-sampling change a few words of the resident state:
                # on the device, from the flag rows
                if side_words:
                    ctx.sbr_state_apply_side_batch(hdr_d, flags_d, state, n_ch, ps_state=ps_state if n_ch == 1 else None)
                if n_ch == 2:
                    ctx.sbr_lp_process_batch(core16, hdr_d, frm_d, state, pcm, ws, status=status, in_ch_fac=2, out_ch_fac=2)
                else:
                    with_ps = flags[got, F_PS] != 0
                    if with_ps.any() != with_ps.all():
                        raise NotImplementedError("a batch mixing PS and non-PS frames")
                    if with_ps.all():
                        starts = np.nonzero(got & (flags[:, F_PS_START] != 0))[0]
                        if starts.size:
                            idx = torch.from_numpy(starts.astype(np.int32)).to(dev)
                            ctx.sbr_state_handover(HANDOVER_PS_START, idx, idx, state, ps_state)
                        ctx.sbr_hq_process_batch(core16, hdr_d, frm_d, state, pcm, ws, ps_frame=psf_d, ps_state=ps_state, status=status)
                    else:
                        ctx.sbr_hq_process_batch(core16, hdr_d, frm_d, state, pcm_mono, ws, status=status)
                        # mono duplicated to stereo (api.c:3639-3660)
                        pcm.view(n, 2048, 2).copy_(pcm_mono.view(n, 2048, 1).expand(n, 2048, 2))
                hand_down(slot, got, (n, 2048, 2))
            t_gpu += time.perf_counter() - t0
            first = False
    except BaseException:
        bp.close()   # (also takes back a batch the parser team still holds)
        raise
    consume()
    t_steps = time.perf_counter() - t_steps
    if not sbr and keep_pcm:
        # the limiter's delay line holds the last attack_time_samples samples: api.c:2824-2866
        ctx.sync()
        lim_h = lim.cpu().numpy()
        for i in range(n):
            st = LimiterState.from_buffer_copy((lim_at_end[i] if i in lim_at_end else lim_h[i]).tobytes())
            att, idx = st.attack_time_samples, st.delayed_input_index
            d = np.ctypeslib.as_array(st.delayed_input)[:att * n_ch].reshape(att, n_ch)
            tail = np.concatenate([d[idx:], d[:idx]]).astype(np.float64)
            inside = (tail > -2147483649.0) & (tail < 2147483648.0)      # (WORD32) of a float as x86 converts it: what does
            v = np.where(inside, np.trunc(np.where(inside, tail, 0.0)), -2147483648.0).astype(np.int64)   # not fit is INT_MIN
            v = np.clip(v + 0x8000, -(1 << 31), (1 << 31) - 1) >> 16   # round16
            out[i].append(v.astype(np.int16))
    frames = int(bp.frames.sum())
    bp.close()
    if own:
        ctx.close()
    if timing is not None:
        # parse_s: inside the parser calls; wait_parse_s: what the loop waited for them; gpu_s: the loop's GPU section (enqueue
        # + wait_down_s, the wait for the previous step's PCM)
        timing.update(parse_s=t_parse, gpu_s=t_gpu, steps_s=t_steps, frames=frames, wait_parse_s=t_wait_parse, wait_down_s=t_wait_down)
    return [np.concatenate(o) if o else np.zeros((0, out_ch), np.int16) for o in out], rate * (2 if sbr else 1)
