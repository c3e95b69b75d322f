"""libxaac_amd -- MI355X (gfx950) back-end for the transform hot path of the
libxaac decoder: host-side mirror of the reference's frame-level seam.

The product is ``libxaac_amd/libxaac_amd.so`` (hand-written HIP + a C ABI,
``include/xaac_amd.h``).  This module is the thin Python host layer used by the
tests and ``bench.py``: it binds the C ABI with ctypes and uses PyTorch only to
own device memory and streams.  There is no CPU fallback: importing works
anywhere, but creating a context without the built library or without a HIP
device raises.

Reference seam mirrored (names, argument meaning, error behaviour):
``ixheaacd_imdct_process`` -- decoder/ixheaacd_lpfuncs.c:347 (decl
decoder/ixheaacd_block.h:132), here batched over N channel-frames.
"""
import ctypes
import os

__all__ = [
    "ONLY_LONG_SEQUENCE", "LONG_START_SEQUENCE", "EIGHT_SHORT_SEQUENCE", "LONG_STOP_SEQUENCE",
    "PCM_LC", "PCM_SBR", "BAD_WINDOW_SEQ", "XaacError", "XaacContext", "load_library", "library_path",
]

# decoder/ixheaacd_cnst.h:100-103
ONLY_LONG_SEQUENCE, LONG_START_SEQUENCE, EIGHT_SHORT_SEQUENCE, LONG_STOP_SEQUENCE = 0, 1, 2, 3
PCM_LC, PCM_SBR = 0, 1
BAD_WINDOW_SEQ = -0x7FFC   # XAAC_FATAL_BAD_WINDOW_SEQ (0xFFFF8004) as the int32 a status word holds

_HERE = os.path.dirname(os.path.abspath(__file__))


def library_path():
    # XAAC_AMD_LIBRARY: developer override to time an experimental build of the same library
    return os.environ.get("XAAC_AMD_LIBRARY") or os.path.join(_HERE, "libxaac_amd.so")


class XaacError(RuntimeError):
    """Non-zero IA_ERRORCODE-style return (bit 31 set = fatal)."""

    def __init__(self, code, where):
        self.code = code & 0xFFFFFFFF
        super().__init__("%s failed: 0x%08X" % (where, self.code))


class _ImdctBatch(ctypes.Structure):
    # struct xaac_imdct_batch, include/xaac_amd.h
    _fields_ = [("n_ch", ctypes.c_int32), ("ch_fac", ctypes.c_int32), ("spec", ctypes.c_void_p),
                ("ics", ctypes.c_void_p), ("overlap", ctypes.c_void_p), ("state", ctypes.c_void_p),
                ("out32", ctypes.c_void_p), ("pcm16", ctypes.c_void_p), ("qshift_adj", ctypes.c_void_p),
                ("pcm_mode", ctypes.c_int32), ("status", ctypes.c_void_p)]


class _QmfAnaBatch(ctypes.Structure):
    # struct xaac_qmf_ana_batch
    _fields_ = [("n_ch", ctypes.c_int32), ("ch_fac", ctypes.c_int32), ("low_pow", ctypes.c_int32),
                ("usb", ctypes.c_int32), ("slot_stride", ctypes.c_int32), ("pcm", ctypes.c_void_p),
                ("state", ctypes.c_void_p), ("qmf", ctypes.c_void_p)]


class _QmfSynBatch(ctypes.Structure):
    # struct xaac_qmf_syn_batch
    _fields_ = [("n_ch", ctypes.c_int32), ("ch_fac", ctypes.c_int32), ("low_pow", ctypes.c_int32),
                ("lsb", ctypes.c_int32), ("usb", ctypes.c_int32), ("split", ctypes.c_int32),
                ("slot_stride", ctypes.c_int32), ("down_sample", ctypes.c_int32), ("qmf", ctypes.c_void_p),
                ("scale", ctypes.c_void_p),
                ("state", ctypes.c_void_p), ("pcm", ctypes.c_void_p)]


class _EsbrSbrBatch(ctypes.Structure):
    # struct xaac_esbr_sbr_batch (include/xaac_esbr.h)
    _fields_ = [("n_ch", ctypes.c_int32), ("core", ctypes.c_void_p), ("header", ctypes.c_void_p),
                ("frame", ctypes.c_void_p), ("side", ctypes.c_void_p), ("state", ctypes.c_void_p),
                ("out", ctypes.c_void_p), ("ps_frame", ctypes.c_void_p), ("ps_state", ctypes.c_void_p),
                ("out_r", ctypes.c_void_p), ("status", ctypes.c_void_p), ("workspace", ctypes.c_void_p),
                ("workspace_bytes", ctypes.c_uint64), ("hbe_state", ctypes.c_void_p), ("hbe_max_synth_size", ctypes.c_int32),
                ("pvc_side", ctypes.c_void_p), ("pvc_state", ctypes.c_void_p), ("sbr_ratio", ctypes.c_int32),
                ("down_sample", ctypes.c_int32), ("hbe_dft_state", ctypes.c_void_p), ("hbe_dft_cfg_tab", ctypes.c_void_p),
                ("hbe_dft_coef_re", ctypes.c_void_p), ("hbe_dft_coef_im", ctypes.c_void_p), ("hbe_dft_cfg", ctypes.c_void_p)]


ESBR_RATIO_2_1, ESBR_RATIO_8_3, ESBR_RATIO_4_1 = 0, 1, 2   # xaac_esbr.h: XAAC_ESBR_RATIO_*


class _EsbrCoreInBatch(ctypes.Structure):
    # struct xaac_esbr_core_in_batch
    _fields_ = [("n_ch", ctypes.c_int32), ("ch_fac", ctypes.c_int32), ("pcm", ctypes.c_void_p), ("core", ctypes.c_void_p)]


class _EsbrPcmOutBatch(ctypes.Structure):
    # struct xaac_esbr_pcm_out_batch
    _fields_ = [("n", ctypes.c_int32), ("stride", ctypes.c_int32), ("left", ctypes.c_void_p), ("right", ctypes.c_void_p),
                ("pcm", ctypes.c_void_p)]


class _HandoverBatch(ctypes.Structure):
    # struct xaac_sbr_handover_batch
    _fields_ = [("n", ctypes.c_int32), ("mode", ctypes.c_int32), ("src", ctypes.c_void_p), ("dst", ctypes.c_void_p),
                ("state", ctypes.c_void_p), ("ps_state", ctypes.c_void_p)]


HANDOVER_PS_START, HANDOVER_STEREO_START = 1, 2

class _ApplySideBatch(ctypes.Structure):
    # struct xaac_sbr_apply_side_batch
    _fields_ = [("n_streams", ctypes.c_int32), ("ch_fac", ctypes.c_int32), ("header", ctypes.c_void_p), ("flags", ctypes.c_void_p),
                ("state", ctypes.c_void_p), ("ps_state", ctypes.c_void_p)]



class _UsacImdctBatch(ctypes.Structure):
    # struct xaac_usac_imdct_batch
    _fields_ = [("n_ch", ctypes.c_int32), ("ccfl", ctypes.c_int32), ("coef", ctypes.c_void_p), ("ics", ctypes.c_void_p),
                ("overlap", ctypes.c_void_p), ("shape_prev", ctypes.c_void_p), ("out32", ctypes.c_void_p),
                ("time", ctypes.c_void_p), ("status", ctypes.c_void_p), ("lpd_flags", ctypes.c_void_p), ("fac", ctypes.c_void_p),
                ("fac_in", ctypes.c_void_p), ("fac_work", ctypes.c_void_p)]


USAC_FAC_IN_WORDS = 129 + 17 + 256   # struct xaac_usac_fac_in: fac_data int32[129], lpc_prev float[17], acelp_in float[256]


class _QmfAnaEldBatch(ctypes.Structure):
    # struct xaac_qmf_ana_eld_batch
    _fields_ = [("n_ch", ctypes.c_int32), ("n_slots", ctypes.c_int32), ("usb", ctypes.c_int32), ("slot_stride", ctypes.c_int32),
                ("pcm", ctypes.c_void_p), ("state", ctypes.c_void_p), ("qmf", ctypes.c_void_p), ("status", ctypes.c_void_p)]


QMF_ANA_ELD_STATE_WORDS = 324   # struct xaac_qmf_ana_eld_state: ring[320], wr, f1, f2, fp (int16)


class _ImdctLdBatch(ctypes.Structure):
    # struct xaac_imdct_ld_batch
    _fields_ = [("n_ch", ctypes.c_int32), ("ch_fac", ctypes.c_int32), ("frame_length", ctypes.c_int32), ("eld", ctypes.c_int32),
                ("spec", ctypes.c_void_p), ("window_shape", ctypes.c_void_p), ("overlap", ctypes.c_void_p),
                ("shape_prev", ctypes.c_void_p), ("pcm16", ctypes.c_void_p), ("status", ctypes.c_void_p)]


class _QmfSynEldBatch(ctypes.Structure):
    # struct xaac_qmf_syn_eld_batch
    _fields_ = [("n_ch", ctypes.c_int32), ("n_slots", ctypes.c_int32), ("lsb", ctypes.c_int32), ("usb", ctypes.c_int32),
                ("split", ctypes.c_int32), ("slot_stride", ctypes.c_int32), ("qmf", ctypes.c_void_p), ("scale", ctypes.c_void_p),
                ("state", ctypes.c_void_p), ("pcm", ctypes.c_void_p), ("status", ctypes.c_void_p), ("qmf_scaled", ctypes.c_void_p)]


QMF_SYN_ELD_STATE_WORDS = 1284   # struct xaac_qmf_syn_eld_state: ring[1280], drc_offset, phase, fp, sixty4 (int16)


class _EsbrAnaBatch(ctypes.Structure):
    # struct xaac_esbr_ana_batch
    _fields_ = [("n_ch", ctypes.c_int32), ("core", ctypes.c_void_p), ("state", ctypes.c_void_p),
                ("qmf_re", ctypes.c_void_p), ("qmf_im", ctypes.c_void_p)]


class _EsbrAnaNbBatch(ctypes.Structure):
    # struct xaac_esbr_ana_nb_batch
    _fields_ = [("n_ch", ctypes.c_int32), ("n_bands", ctypes.c_int32), ("n_slots", ctypes.c_int32),
                ("core_stride", ctypes.c_int32), ("core", ctypes.c_void_p), ("state", ctypes.c_void_p),
                ("qmf_re", ctypes.c_void_p), ("qmf_im", ctypes.c_void_p)]


class _HbeSynthBatch(ctypes.Structure):
    # struct xaac_hbe_synth_batch
    _fields_ = [("n_ch", ctypes.c_int32), ("num_columns", ctypes.c_int32), ("qmf_re", ctypes.c_void_p),
                ("qmf_im", ctypes.c_void_p), ("state", ctypes.c_void_p), ("status", ctypes.c_void_p)]


class _HbeApplyBatch(ctypes.Structure):
    # struct xaac_hbe_apply_batch_desc
    _fields_ = [("n_ch", ctypes.c_int32), ("qmf_re", ctypes.c_void_p), ("qmf_im", ctypes.c_void_p),
                ("pitch_in_bins", ctypes.c_void_p), ("state", ctypes.c_void_p), ("pv_re", ctypes.c_void_p),
                ("pv_im", ctypes.c_void_p), ("status", ctypes.c_void_p), ("max_synth_size", ctypes.c_int32)]


class _HbeDftAnalBatch(ctypes.Structure):
    # struct xaac_hbe_dft_anal_batch
    _fields_ = [("n_ch", ctypes.c_int32), ("no_bins", ctypes.c_int32), ("time_in", ctypes.c_void_p), ("in_stride", ctypes.c_int32),
                ("coef_re", ctypes.c_void_p), ("coef_im", ctypes.c_void_p), ("cfg", ctypes.c_void_p), ("state", ctypes.c_void_p),
                ("qmf_re", ctypes.c_void_p), ("qmf_im", ctypes.c_void_p), ("status", ctypes.c_void_p)]


HBE_DFT_STATE_BYTES = 4 * 642   # struct xaac_hbe_dft_anal_state


class _HbeDftApplyBatch(ctypes.Structure):
    # struct xaac_hbe_dft_apply_batch
    _fields_ = [("n_ch", ctypes.c_int32), ("qmf_re", ctypes.c_void_p), ("qmf_im", ctypes.c_void_p), ("pitch_in_bins", ctypes.c_void_p),
                ("oversampling", ctypes.c_void_p), ("cfg_tab", ctypes.c_void_p), ("coef_re", ctypes.c_void_p), ("coef_im", ctypes.c_void_p),
                ("cfg", ctypes.c_void_p), ("state", ctypes.c_void_p), ("pv_re", ctypes.c_void_p), ("pv_im", ctypes.c_void_p),
                ("status", ctypes.c_void_p), ("rows32", ctypes.c_int32)]


HBE_DFT_FULL_STATE_BYTES = 4 * (1024 + 3072 + 1280 + 642 + 12)   # struct xaac_hbe_dft_state
HBE_DFT_CFG_BYTES = 4 * (512 + 768 + 3 * 2 * 772)               # struct xaac_hbe_dft_cfg


class _PvcBatch(ctypes.Structure):
    # struct xaac_pvc_batch (include/xaac_pvc.h)
    _fields_ = [("n_ch", ctypes.c_int32), ("frame", ctypes.c_void_p), ("qmf_re", ctypes.c_void_p), ("qmf_im", ctypes.c_void_p),
                ("qmf_stride", ctypes.c_int32), ("state", ctypes.c_void_p), ("out", ctypes.c_void_p), ("status", ctypes.c_void_p)]


PVC_FRAME_BYTES = 40    # struct xaac_pvc_frame
PVC_STATE_BYTES = 188   # struct xaac_pvc_state


class _HbeAnalBatch(ctypes.Structure):
    # struct xaac_hbe_anal_batch
    _fields_ = [("n_ch", ctypes.c_int32), ("state", ctypes.c_void_p), ("status", ctypes.c_void_p)]


HBE_STATE_BYTES = 4 * (1088 + 1280 + 640 + 32 * 128 + 64 * 128 + 12)   # struct xaac_hbe_state


class _EsbrSynBatch(ctypes.Structure):
    # struct xaac_esbr_syn_batch
    _fields_ = [("n_ch", ctypes.c_int32), ("qmf_re", ctypes.c_void_p), ("qmf_im", ctypes.c_void_p),
                ("state", ctypes.c_void_p), ("out", ctypes.c_void_p)]


ESBR_SIDE_BYTES, ESBR_STATE_BYTES, ESBR_PS_STATE_BYTES = 2036, 38012, 23048   # include/xaac_esbr.h (tests/test_abi.py checks them)
ESBR_PVC_SIDE_BYTES, ESBR_PVC_STATE_BYTES = 84, 12656                          # xaac_esbr_pvc_side / xaac_esbr_pvc_state
ESBR_ANA_STATE_WORDS = 322   # struct xaac_esbr_ana_state: ring[320], pos, win_off (int32)
ESBR_SYN_STATE_WORDS = 1282  # struct xaac_esbr_syn_state: ring[1280], drc_offset, filt_off (int32)


class _SbrLpBatch(ctypes.Structure):
    # struct xaac_sbr_lp_batch
    _fields_ = [("n_ch", ctypes.c_int32), ("in_ch_fac", ctypes.c_int32), ("out_ch_fac", ctypes.c_int32),
                ("down_sample", ctypes.c_int32),
                ("pcm_in", ctypes.c_void_p), ("header", ctypes.c_void_p), ("frame", ctypes.c_void_p),
                ("state", ctypes.c_void_p), ("pcm_out", ctypes.c_void_p), ("status", ctypes.c_void_p),
                ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_uint64)]


class _SbrHqBatch(ctypes.Structure):
    # struct xaac_sbr_hq_batch
    _fields_ = [("n_ch", ctypes.c_int32), ("in_ch_fac", ctypes.c_int32), ("out_ch_fac", ctypes.c_int32),
                ("down_sample", ctypes.c_int32), ("pcm_in", ctypes.c_void_p), ("header", ctypes.c_void_p),
                ("frame", ctypes.c_void_p), ("state", ctypes.c_void_p), ("ps_frame", ctypes.c_void_p),
                ("ps_state", ctypes.c_void_p), ("pcm_out", ctypes.c_void_p), ("status", ctypes.c_void_p),
                ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_uint64), ("max_band_hint", ctypes.c_int32)]


class _SbrEldBatch(ctypes.Structure):
    # struct xaac_sbr_eld_batch
    _fields_ = [("n_ch", ctypes.c_int32), ("n_slots", ctypes.c_int32), ("in_ch_fac", ctypes.c_int32), ("out_ch_fac", ctypes.c_int32),
                ("pcm_in", ctypes.c_void_p), ("header", ctypes.c_void_p), ("frame", ctypes.c_void_p), ("state", ctypes.c_void_p),
                ("pcm_out", ctypes.c_void_p), ("status", ctypes.c_void_p), ("workspace", ctypes.c_void_p),
                ("workspace_bytes", ctypes.c_uint64), ("qmf_handed_on", ctypes.c_void_p)]


class _LimiterBatch(ctypes.Structure):
    # struct xaac_limiter_batch
    _fields_ = [("n_streams", ctypes.c_int32), ("frame_len", ctypes.c_int32), ("samples", ctypes.c_void_p),
                ("stride", ctypes.c_int64), ("qshift_adj", ctypes.c_void_p), ("state", ctypes.c_void_p),
                ("num_channels", ctypes.c_int32), ("planar", ctypes.c_int32), ("pcm16", ctypes.c_void_p),
                ("status", ctypes.c_void_p), ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_uint64)]


LIM_MAX_ATTACK, LIM_MAX_CH = 480, 8


class LimiterState(ctypes.Structure):
    # struct xaac_limiter_state (ia_peak_limiter_struct with its buffers inline)
    _fields_ = [("attack_constant", ctypes.c_float), ("release_constant", ctypes.c_float),
                ("num_channels", ctypes.c_uint32), ("attack_time_samples", ctypes.c_uint32),
                ("limiter_on", ctypes.c_uint32), ("gain_modified", ctypes.c_float), ("min_gain", ctypes.c_float),
                ("delayed_input_index", ctypes.c_uint32), ("pre_smoothed_gain", ctypes.c_double),
                ("max_idx", ctypes.c_int32), ("cir_buf_pnt", ctypes.c_int32),
                ("max_buf", ctypes.c_float * LIM_MAX_ATTACK),
                ("delayed_input", ctypes.c_float * (LIM_MAX_ATTACK * LIM_MAX_CH))]


LIMITER_STATE_BYTES = ctypes.sizeof(LimiterState)   # 17328
SBR_HEADER_BYTES, SBR_FRAME_BYTES, SBR_STATE_BYTES = 336, 1072, 7300   # include/xaac_sbr.h
SBR_ELD_STATE_BYTES = 4236                                             # xaac_sbr_eld_state (include/xaac_amd.h)
PS_FRAME_BYTES, PS_STATE_BYTES = 972, 7764
QMF_ANA_STATE_WORDS = 322    # int16 words of struct xaac_qmf_ana_state: ring[320], wr, phase
QMF_SYN_STATE_WORDS = 1282   # int16 words of struct xaac_qmf_syn_state: ring[1280], drc_offset, phase

_lib = None


def load_library():
    """dlopen the product library; loud failure when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: PyTorch bundles its own libamdhip64 and owns the
    # streams / device memory handed to the C ABI, so it must be the copy that
    # libxaac_amd.so binds to -- import torch first whenever it is installed.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    path = library_path()
    if not os.path.exists(path):
        raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(or make -C libxaac_amd/csrc); there is no CPU fallback" % path)
    lib = ctypes.CDLL(path)
    lib.xaac_version.restype = ctypes.c_char_p
    lib.xaac_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int32, ctypes.c_void_p]
    lib.xaac_destroy.argtypes = [ctypes.c_void_p]
    lib.xaac_sync.argtypes = [ctypes.c_void_p]
    lib.xaac_warm_up.argtypes = [ctypes.c_void_p]
    lib.xaac_warm_up.restype = ctypes.c_int32
    lib.xaac_set_stream.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.xaac_imdct_process_batch.argtypes = [ctypes.c_void_p, ctypes.POINTER(_ImdctBatch)]
    lib.xaac_imdct_process_batch_host.argtypes = [ctypes.c_void_p, ctypes.POINTER(_ImdctBatch)]
    lib.xaac_imdct960_process_batch.argtypes = [ctypes.c_void_p, ctypes.POINTER(_ImdctBatch)]
    lib.xaac_imdct960_process_batch.restype = ctypes.c_int32
    lib.xaac_imdct_ld_process_batch.argtypes = [ctypes.c_void_p, ctypes.POINTER(_ImdctLdBatch)]
    lib.xaac_imdct_ld_process_batch.restype = ctypes.c_int32
    lib.xaac_last_launch.argtypes = [ctypes.c_void_p] + [ctypes.POINTER(ctypes.c_int32)] * 3
    lib.xaac_qmf_analysis_batch.argtypes = [ctypes.c_void_p, ctypes.POINTER(_QmfAnaBatch)]
    lib.xaac_qmf_synthesis_batch.argtypes = [ctypes.c_void_p, ctypes.POINTER(_QmfSynBatch)]
    lib.xaac_esbr_workspace_bytes.argtypes = [ctypes.c_int32]
    lib.xaac_esbr_workspace_bytes.restype = ctypes.c_uint64
    lib.xaac_esbr_workspace_bytes_ratio.argtypes = [ctypes.c_int32, ctypes.c_int32]
    lib.xaac_esbr_workspace_bytes_ratio.restype = ctypes.c_uint64
    lib.xaac_esbr_sbr_process_batch.argtypes = [ctypes.c_void_p, ctypes.POINTER(_EsbrSbrBatch)]
    lib.xaac_esbr_sbr_process_batch.restype = ctypes.c_int32
    lib.xaac_sbr_state_handover.argtypes = [ctypes.c_void_p, ctypes.POINTER(_HandoverBatch)]
    lib.xaac_sbr_state_handover.restype = ctypes.c_int32
    lib.xaac_sbr_state_apply_side_batch.argtypes = [ctypes.c_void_p, ctypes.POINTER(_ApplySideBatch)]
    lib.xaac_sbr_state_apply_side_batch.restype = ctypes.c_int32
    lib.xaac_usac_imdct_process_batch.argtypes = [ctypes.c_void_p, ctypes.POINTER(_UsacImdctBatch)]
    lib.xaac_usac_imdct_process_batch.restype = ctypes.c_int32
    lib.xaac_hbe_real_synth_batch.argtypes = [ctypes.c_void_p, ctypes.POINTER(_HbeSynthBatch)]
    lib.xaac_hbe_real_synth_batch.restype = ctypes.c_int32
    lib.xaac_hbe_dft_anal_batch_run.argtypes = [ctypes.c_void_p, ctypes.POINTER(_HbeDftAnalBatch)]
    lib.xaac_hbe_dft_anal_batch_run.restype = ctypes.c_int32
    lib.xaac_pvc_process_batch.argtypes = [ctypes.c_void_p, ctypes.POINTER(_PvcBatch)]
    lib.xaac_pvc_process_batch.restype = ctypes.c_int32
    lib.xaac_hbe_dft_apply_batch_run.argtypes = [ctypes.c_void_p, ctypes.POINTER(_HbeDftApplyBatch)]
    lib.xaac_hbe_dft_apply_batch_run.restype = ctypes.c_int32
    lib.xaac_hbe_apply_batch.argtypes = [ctypes.c_void_p, ctypes.POINTER(_HbeApplyBatch)]
    lib.xaac_hbe_apply_batch.restype = ctypes.c_int32
    lib.xaac_hbe_cplx_anal_batch.argtypes = [ctypes.c_void_p, ctypes.POINTER(_HbeAnalBatch)]
    lib.xaac_hbe_cplx_anal_batch.restype = ctypes.c_int32
    lib.xaac_qmf_analysis_eld_batch.argtypes = [ctypes.c_void_p, ctypes.POINTER(_QmfAnaEldBatch)]
    lib.xaac_qmf_analysis_eld_batch.restype = ctypes.c_int32
    lib.xaac_qmf_synthesis_eld_batch.argtypes = [ctypes.c_void_p, ctypes.POINTER(_QmfSynEldBatch)]
    lib.xaac_qmf_synthesis_eld_batch.restype = ctypes.c_int32
    lib.xaac_esbr_qmf_analysis_batch.argtypes = [ctypes.c_void_p, ctypes.POINTER(_EsbrAnaBatch)]
    lib.xaac_esbr_qmf_analysis_batch.restype = ctypes.c_int32
    lib.xaac_esbr_qmf_analysis_nb_batch.argtypes = [ctypes.c_void_p, ctypes.POINTER(_EsbrAnaNbBatch)]
    lib.xaac_esbr_qmf_analysis_nb_batch.restype = ctypes.c_int32
    lib.xaac_esbr_qmf_synthesis_batch.argtypes = [ctypes.c_void_p, ctypes.POINTER(_EsbrSynBatch)]
    lib.xaac_esbr_qmf_synthesis_batch.restype = ctypes.c_int32
    lib.xaac_esbr_qmf_synthesis_ds_batch.argtypes = [ctypes.c_void_p, ctypes.POINTER(_EsbrSynBatch)]
    lib.xaac_esbr_qmf_synthesis_ds_batch.restype = ctypes.c_int32
    lib.xaac_sbr_lp_process_batch.argtypes = [ctypes.c_void_p, ctypes.POINTER(_SbrLpBatch)]
    lib.xaac_sbr_lp_process_batch.restype = ctypes.c_int32
    lib.xaac_sbr_lp_workspace_bytes.argtypes = [ctypes.c_int32]
    lib.xaac_sbr_lp_workspace_bytes.restype = ctypes.c_uint64
    lib.xaac_sbr_hq_process_batch.argtypes = [ctypes.c_void_p, ctypes.POINTER(_SbrHqBatch)]
    lib.xaac_sbr_hq_process_batch.restype = ctypes.c_int32
    lib.xaac_sbr_eld_workspace_bytes.argtypes = [ctypes.c_int32]
    lib.xaac_sbr_eld_workspace_bytes.restype = ctypes.c_uint64
    lib.xaac_sbr_eld_process_batch.argtypes = [ctypes.c_void_p, ctypes.POINTER(_SbrEldBatch)]
    lib.xaac_sbr_hq_workspace_bytes.argtypes = [ctypes.c_int32, ctypes.c_int32]
    lib.xaac_sbr_hq_workspace_bytes.restype = ctypes.c_uint64
    lib.xaac_peak_limiter_init.argtypes = [ctypes.POINTER(LimiterState), ctypes.c_uint32, ctypes.c_uint32]
    lib.xaac_peak_limiter_init.restype = ctypes.c_int32
    lib.xaac_peak_limiter_process_batch.argtypes = [ctypes.c_void_p, ctypes.POINTER(_LimiterBatch)]
    lib.xaac_peak_limiter_process_batch.restype = ctypes.c_int32
    lib.xaac_peak_limiter_workspace_bytes.argtypes = [ctypes.c_int32]
    lib.xaac_peak_limiter_workspace_bytes.restype = ctypes.c_uint64
    for f in ("xaac_create", "xaac_destroy", "xaac_sync", "xaac_set_stream", "xaac_imdct_process_batch",
              "xaac_imdct_process_batch_host", "xaac_last_launch", "xaac_qmf_analysis_batch",
              "xaac_qmf_synthesis_batch"):
        getattr(lib, f).restype = ctypes.c_int32
    _lib = lib
    return lib


def _ptr(t, dtype_name, numel=None, allow_none=False, device_ok=None):
    """data pointer of a torch tensor or numpy array after shape/dtype checks"""
    if t is None:
        if allow_none:
            return None
        raise ValueError("required buffer is None")
    if hasattr(t, "data_ptr"):  # torch
        if not t.is_contiguous():
            raise ValueError("buffer must be contiguous")
        if str(t.dtype).replace("torch.", "") != dtype_name:
            raise TypeError("expected %s, got %s" % (dtype_name, t.dtype))
        if device_ok is not None and t.is_cuda != device_ok:
            raise ValueError("buffer is on the wrong side of the PCIe bus")
        n, p = t.numel(), t.data_ptr()
    else:  # numpy
        if not t.flags["C_CONTIGUOUS"]:
            raise ValueError("buffer must be contiguous")
        if t.dtype.name != dtype_name:
            raise TypeError("expected %s, got %s" % (dtype_name, t.dtype))
        if device_ok:
            raise ValueError("host array passed where a device tensor is required")
        n, p = t.size, t.ctypes.data
    if numel is not None and n != numel:
        raise ValueError("buffer has %d elements, expected %d" % (n, numel))
    return p


def peak_limiter_init(num_channels, sample_rate):
    """ixheaacd_peak_limiter_init: -> (LimiterState, delay in samples)"""
    st = LimiterState()
    rc = load_library().xaac_peak_limiter_init(ctypes.byref(st), int(num_channels), int(sample_rate))
    if rc < 0:
        raise XaacError(rc, "xaac_peak_limiter_init")
    return st, rc


class XaacContext:
    """One context per GPU / stream (xaac_create ... xaac_destroy)."""

    def __init__(self, device=0, stream_handle=None):
        """stream_handle: a hipStream_t as an int (e.g. torch.cuda.Stream().cuda_stream; 0 = the
        legacy null stream); None lets the library create and own a stream."""
        self._lib = load_library()
        h = ctypes.c_void_p()
        rc = self._lib.xaac_create(ctypes.byref(h), int(device), None)
        if rc != 0:
            raise XaacError(rc, "xaac_create")
        self._h = h
        self.device = device
        if stream_handle is not None:
            self.set_stream(stream_handle)

    def set_stream(self, stream_handle):
        rc = self._lib.xaac_set_stream(self._h, ctypes.c_void_p(int(stream_handle)))
        if rc != 0:
            raise XaacError(rc, "xaac_set_stream")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.xaac_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        rc = self._lib.xaac_sync(self._h)
        if rc != 0:
            raise XaacError(rc, "xaac_sync")

    def warm_up(self):
        """every kernel's code object onto the device now (otherwise: at each module's first launch)"""
        rc = self._lib.xaac_warm_up(self._h)
        if rc != 0:
            raise XaacError(rc, "xaac_warm_up")

    def last_launch(self):
        g, b, l = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        self._lib.xaac_last_launch(self._h, ctypes.byref(g), ctypes.byref(b), ctypes.byref(l))
        return {"grid": g.value, "block": b.value, "lds_bytes": l.value}

    def _batch(self, n_ch, spec, ics, overlap, state, out32, pcm16, qshift_adj, ch_fac, pcm_mode, on_device, status=None,
               frame=1024):
        b = _ImdctBatch()
        b.n_ch, b.ch_fac, b.pcm_mode = int(n_ch), int(ch_fac), int(pcm_mode)
        b.spec = _ptr(spec, "int32", n_ch * frame, device_ok=on_device)
        b.ics = _ptr(ics, "uint8", n_ch * 2, device_ok=on_device)
        b.overlap = _ptr(overlap, "int32", n_ch * frame // 2, device_ok=on_device)
        b.state = _ptr(state, "uint8", n_ch * 2, device_ok=on_device)
        b.out32 = _ptr(out32, "int32", n_ch * frame, allow_none=True, device_ok=on_device)
        b.pcm16 = _ptr(pcm16, "int16", n_ch * frame, allow_none=True, device_ok=on_device)
        b.qshift_adj = _ptr(qshift_adj, "int8", n_ch, allow_none=True, device_ok=on_device)
        b.status = _ptr(status, "int32", n_ch, allow_none=True, device_ok=on_device)
        return b

    def imdct_process_batch(self, spec, ics, overlap, state, out32=None, pcm16=None, qshift_adj=None,
                            ch_fac=1, pcm_mode=PCM_LC, status=None):
        """Batched ixheaacd_imdct_process on device tensors (asynchronous).

        spec int32[N,1024]; ics uint8[N,2] = (window_sequence, window_shape);
        overlap int32[N,512] in/out; state uint8[N,2] in/out (previous
        window_sequence, window_shape); optional outputs out32 int32[N*1024],
        pcm16 int16[N*1024] (interleaved at stride ch_fac), qshift_adj int8[N], status int32[N] (0, or
        BAD_WINDOW_SEQ for a channel-frame whose window bytes no bitstream can carry: left untouched)."""
        n_ch = spec.shape[0] if spec.dim() == 2 else spec.numel() // 1024
        b = self._batch(n_ch, spec, ics, overlap, state, out32, pcm16, qshift_adj, ch_fac, pcm_mode, True, status)
        rc = self._lib.xaac_imdct_process_batch(self._h, ctypes.byref(b))
        if rc != 0:
            raise XaacError(rc, "xaac_imdct_process_batch")

    def imdct960_process_batch(self, spec, ics, overlap, state, out32=None, pcm16=None, qshift_adj=None, ch_fac=1,
                               pcm_mode=PCM_LC, status=None):
        """Batched ixheaacd_imdct_process for frame_length 960 on device tensors (asynchronous): spec int32[N,960],
        overlap int32[N,480] in/out, out32 / pcm16 [N*960]; everything else as imdct_process_batch."""
        n_ch = spec.shape[0]
        b = self._batch(n_ch, spec, ics, overlap, state, out32, pcm16, qshift_adj, ch_fac, pcm_mode, True, status, frame=960)
        rc = self._lib.xaac_imdct960_process_batch(self._h, ctypes.byref(b))
        if rc != 0:
            raise XaacError(rc, "xaac_imdct960_process_batch")

    def imdct_ld_process_batch(self, spec, window_shape, overlap, shape_prev, pcm16, frame_length, eld, ch_fac=1, status=None):
        """Batched ixheaacd_imdct_process for AAC-LD (eld = 0) / AAC-ELD (eld = 1) frames on device tensors: spec
        int32[N, frame_length] (512 or 480), window_shape uint8[N], overlap int32[N, frame_length / 2] (LD) or
        [N, 3 * frame_length] (ELD) in/out, shape_prev uint8[N] in/out, pcm16 int16[N * frame_length] at stride ch_fac."""
        n_ch = spec.shape[0]
        b = _ImdctLdBatch()
        b.n_ch, b.ch_fac, b.frame_length, b.eld = n_ch, int(ch_fac), int(frame_length), int(eld)
        b.spec = _ptr(spec, "int32", n_ch * frame_length, device_ok=True)
        b.window_shape = _ptr(window_shape, "uint8", n_ch, device_ok=True)
        b.overlap = _ptr(overlap, "int32", n_ch * (3 * frame_length if eld else frame_length // 2), device_ok=True)
        b.shape_prev = _ptr(shape_prev, "uint8", n_ch, device_ok=True)
        b.pcm16 = _ptr(pcm16, "int16", n_ch * frame_length, device_ok=True)
        b.status = _ptr(status, "int32", n_ch, allow_none=True, device_ok=True)
        rc = self._lib.xaac_imdct_ld_process_batch(self._h, ctypes.byref(b))
        if rc != 0:
            raise XaacError(rc, "xaac_imdct_ld_process_batch")

    def imdct_process_batch_host(self, spec, ics, overlap, state, out32=None, pcm16=None, qshift_adj=None,
                                 ch_fac=1, pcm_mode=PCM_LC, status=None):
        """Same on host numpy arrays (copies over PCIe, synchronous)."""
        n_ch = spec.shape[0]
        b = self._batch(n_ch, spec, ics, overlap, state, out32, pcm16, qshift_adj, ch_fac, pcm_mode, False, status)
        rc = self._lib.xaac_imdct_process_batch_host(self._h, ctypes.byref(b))
        if rc != 0:
            raise XaacError(rc, "xaac_imdct_process_batch_host")

    def qmf_analysis_batch(self, pcm, state, qmf, low_pow, usb=32, slot_stride=None, ch_fac=1):
        """Batched ixheaacd_cplx_anal_qmffilt: one frame (1024 PCM16 -> 32 slots x 32 bands) per channel.
        pcm int16[n_ch*1024] interleaved at ch_fac; state int16[n_ch, 322] in/out; qmf int32[n_ch, 32, slot_stride]
        (real bands at +0, imaginary at +64 in HQ mode)."""
        n_ch = state.shape[0]
        if slot_stride is None:
            slot_stride = 64 if low_pow else 128
        b = _QmfAnaBatch()
        b.n_ch, b.ch_fac, b.low_pow, b.usb, b.slot_stride = n_ch, int(ch_fac), int(bool(low_pow)), int(usb), slot_stride
        b.pcm = _ptr(pcm, "int16", n_ch * 1024, device_ok=True)
        b.state = _ptr(state, "int16", n_ch * QMF_ANA_STATE_WORDS, device_ok=True)
        b.qmf = _ptr(qmf, "int32", n_ch * 32 * slot_stride, device_ok=True)
        rc = self._lib.xaac_qmf_analysis_batch(self._h, ctypes.byref(b))
        if rc != 0:
            raise XaacError(rc, "xaac_qmf_analysis_batch")

    def qmf_synthesis_batch(self, qmf, scale, state, pcm, low_pow, lsb, usb, split=6, slot_stride=None, ch_fac=1,
                            down_sample=False):
        """Batched ixheaacd_cplx_synt_qmffilt (no PS): 32 slots x 64 bands -> 2048 PCM16 per channel.
        qmf int32[n_ch, 32, slot_stride]; scale int16[n_ch, 4] = lb_scale, ov_lb_scale, hb_scale, st_syn_scale;
        state int16[n_ch, 1282] in/out; pcm int16[n_ch*2048] interleaved at ch_fac.  down_sample: the 32-channel
        bank (1024 samples per channel out)."""
        n_ch = state.shape[0]
        if slot_stride is None:
            slot_stride = 64 if low_pow else 128
        b = _QmfSynBatch()
        b.n_ch, b.ch_fac, b.low_pow, b.lsb, b.usb, b.split = n_ch, int(ch_fac), int(bool(low_pow)), int(lsb), int(usb), int(split)
        b.slot_stride = slot_stride
        b.down_sample = int(bool(down_sample))
        b.qmf = _ptr(qmf, "int32", n_ch * 32 * slot_stride, device_ok=True)
        b.scale = _ptr(scale, "int16", n_ch * 4, device_ok=True)
        b.state = _ptr(state, "int16", n_ch * QMF_SYN_STATE_WORDS, device_ok=True)
        b.pcm = _ptr(pcm, "int16", n_ch * (1024 if down_sample else 2048), device_ok=True)
        rc = self._lib.xaac_qmf_synthesis_batch(self._h, ctypes.byref(b))
        if rc != 0:
            raise XaacError(rc, "xaac_qmf_synthesis_batch")

    def esbr_workspace_bytes(self, n_ch, sbr_ratio=0):
        return int(self._lib.xaac_esbr_workspace_bytes_ratio(int(n_ch), int(sbr_ratio)))

    def esbr_sbr_process_batch(self, core, header, frame, side, state, out, workspace, status=None, ps_frame=None,
                               ps_state=None, out_r=None, hbe_state=None, hbe_max_synth_size=0, pvc_side=None, pvc_state=None,
                               sbr_ratio=0, down_sample=False, hbe_dft=None):
        """One frame of every channel through the Path A (eSBR, -esbr:1) branch of ixheaacd_sbr_dec, mono / stereo
        channels without PS: core float32[n_ch, 1024]; header / frame / side / state uint8 views of the xaac_sbr_header,
        xaac_sbr_frame, xaac_esbr_side, xaac_esbr_state arrays; out float32[n_ch, 2048].  With ps_frame / ps_state (uint8
        views of xaac_ps_frame / xaac_esbr_ps_state arrays) / out_r: HE-AACv2 streams, float parametric stereo, out = left.
        hbe_state (uint8[n_ch, HBE_STATE_BYTES]): the harmonic transposer runs on every frame and frames with harmonic_sbr
        set take its output.  sbr_ratio ESBR_RATIO_8_3: 768 samples of a core row through the 24-channel bank;
        ESBR_RATIO_4_1: the 16-channel bank, 64 slots, out float32[n_ch, 4096], workspace of esbr_workspace_bytes(n_ch, ratio).
        hbe_dft = (state uint8[n_ch, HBE_DFT_FULL_STATE_BYTES], cfg_tab uint8[n_cfg, HBE_DFT_CFG_BYTES], coef_re, coef_im
        float32[n_cfg, 64, 128], cfg int32[n_ch] or None) instead of hbe_state: -esbr_hq:1, the DFT transposer."""
        n_ch = out.shape[0]
        b = _EsbrSbrBatch()
        b.sbr_ratio = int(sbr_ratio)
        b.down_sample = int(bool(down_sample))    # the 32-channel synthesis bank(s): half the samples at the same row pitch
        b.n_ch = n_ch
        b.core = _ptr(core, "float32", n_ch * 1024, device_ok=True)
        b.header = _ptr(header, "uint8", n_ch * SBR_HEADER_BYTES, device_ok=True)
        b.frame = _ptr(frame, "uint8", n_ch * SBR_FRAME_BYTES, device_ok=True)
        b.side = _ptr(side, "uint8", n_ch * ESBR_SIDE_BYTES, device_ok=True)
        b.state = _ptr(state, "uint8", n_ch * ESBR_STATE_BYTES, device_ok=True)
        b.out = _ptr(out, "float32", n_ch * (4096 if b.sbr_ratio == ESBR_RATIO_4_1 else 2048), device_ok=True)
        b.ps_frame = _ptr(ps_frame, "uint8", n_ch * PS_FRAME_BYTES, allow_none=True, device_ok=True)
        b.ps_state = _ptr(ps_state, "uint8", n_ch * ESBR_PS_STATE_BYTES, allow_none=True, device_ok=True)
        b.out_r = _ptr(out_r, "float32", n_ch * 2048, allow_none=True, device_ok=True)
        b.status = _ptr(status, "int32", n_ch, allow_none=True, device_ok=True)
        b.workspace = _ptr(workspace, "uint8", device_ok=True)
        b.workspace_bytes = workspace.numel()
        b.hbe_state = _ptr(hbe_state, "uint8", n_ch * HBE_STATE_BYTES, allow_none=True, device_ok=True)
        b.hbe_max_synth_size = int(hbe_max_synth_size)   # 4 / 8: no larger transposer bank in the batch (less LDS per channel); 0: any
        # USAC channels with PVC frames: uint8 views of the xaac_esbr_pvc_side / xaac_esbr_pvc_state arrays (both or neither)
        b.pvc_side = _ptr(pvc_side, "uint8", n_ch * ESBR_PVC_SIDE_BYTES, allow_none=True, device_ok=True)
        b.pvc_state = _ptr(pvc_state, "uint8", n_ch * ESBR_PVC_STATE_BYTES, allow_none=True, device_ok=True)
        if hbe_dft is not None:
            d_state, d_cfg_tab, d_cre, d_cim, d_cfg = hbe_dft
            b.hbe_dft_state = _ptr(d_state, "uint8", n_ch * HBE_DFT_FULL_STATE_BYTES, device_ok=True)
            b.hbe_dft_cfg_tab = _ptr(d_cfg_tab, "uint8", device_ok=True)
            b.hbe_dft_coef_re = _ptr(d_cre, "float32", device_ok=True)
            b.hbe_dft_coef_im = _ptr(d_cim, "float32", device_ok=True)
            b.hbe_dft_cfg = _ptr(d_cfg, "int32", n_ch, allow_none=True, device_ok=True)
        rc = self._lib.xaac_esbr_sbr_process_batch(self._h, ctypes.byref(b))
        if rc != 0:
            raise XaacError(rc, "xaac_esbr_sbr_process_batch")

    def esbr_core_from_pcm16(self, pcm, core, ch_fac=1):
        """the core decoder's PCM16 (int16[n_ch * 1024], channels of an element interleaved at ch_fac) as float planes
        core float32[n_ch, 1024] (api.c:3385-3432)"""
        b = _EsbrCoreInBatch()
        b.n_ch, b.ch_fac = int(core.shape[0]), int(ch_fac)
        b.pcm = _ptr(pcm, "int16", b.n_ch * 1024, device_ok=True)
        b.core = _ptr(core, "float32", b.n_ch * 1024, device_ok=True)
        rc = self._lib.xaac_esbr_core_from_pcm16_batch(self._h, ctypes.byref(b))
        if rc != 0:
            raise XaacError(rc, "xaac_esbr_core_from_pcm16_batch")

    def esbr_pcm16_from_float(self, left, right, pcm, stride=2048):
        """ixheaacd_samples_sat for two channels: stream i's float planes left / right (device tensors; element offset
        i * stride) -> pcm int16[n * 2048 * 2]; left is right for a duplicated mono channel"""
        n = pcm.numel() // 4096
        b = _EsbrPcmOutBatch()
        b.n, b.stride = int(n), int(stride)
        b.left = _ptr(left, "float32", device_ok=True)
        b.right = _ptr(right, "float32", device_ok=True)
        b.pcm = _ptr(pcm, "int16", n * 4096, device_ok=True)
        rc = self._lib.xaac_esbr_pcm16_from_float_batch(self._h, ctypes.byref(b))
        if rc != 0:
            raise XaacError(rc, "xaac_esbr_pcm16_from_float_batch")

    def sbr_state_handover(self, mode, src, dst, state, ps_state=None):
        """ixheaacd_sbrdecoder.c:762-806 for the listed streams: mode HANDOVER_PS_START (dst indexes ps_state) or
        HANDOVER_STEREO_START (dst indexes state); src / dst int32 device tensors; state / ps_state uint8 views of the
        xaac_sbr_state / xaac_ps_state arrays."""
        b = _HandoverBatch()
        b.n, b.mode = int(src.numel()), int(mode)
        b.src = _ptr(src, "int32", device_ok=True)
        b.dst = _ptr(dst, "int32", b.n, device_ok=True)
        b.state = _ptr(state, "uint8", device_ok=True)
        b.ps_state = _ptr(ps_state, "uint8", allow_none=True, device_ok=True)
        rc = self._lib.xaac_sbr_state_handover(self._h, ctypes.byref(b))
        if rc != 0:
            raise XaacError(rc, "xaac_sbr_state_handover")

    def sbr_state_apply_side_batch(self, header, flags, state, ch_fac, ps_state=None):
        """ixheaacd_sbr_dec_reset / ixheaacd_prepare_upsamp (sbrdecoder.c:103-276) on the resident states of the streams whose
        flag rows say so: header uint8[n * ch_fac, SBR_HEADER_BYTES] (this frame's), flags int32[n, 8] (the parser's rows;
        zero for streams without a frame), state / ps_state the uint8 views of the xaac_sbr_state / xaac_ps_state arrays."""
        b = _ApplySideBatch()
        b.n_streams, b.ch_fac = int(flags.shape[0]), int(ch_fac)
        b.header = _ptr(header, "uint8", b.n_streams * b.ch_fac * SBR_HEADER_BYTES, device_ok=True)
        b.flags = _ptr(flags, "int32", b.n_streams * 8, device_ok=True)
        b.state = _ptr(state, "uint8", b.n_streams * b.ch_fac * SBR_STATE_BYTES, device_ok=True)
        b.ps_state = _ptr(ps_state, "uint8", allow_none=True, device_ok=True)
        rc = self._lib.xaac_sbr_state_apply_side_batch(self._h, ctypes.byref(b))
        if rc != 0:
            raise XaacError(rc, "xaac_sbr_state_apply_side_batch")

    def usac_imdct_process_batch(self, coef, ics, overlap, shape_prev, out32=None, time=None, status=None, ccfl=1024,
                                 lpd_flags=None, fac=None, fac_in=None, fac_work=None):
        """Batched ixheaacd_fd_frm_dec (USAC FD frame after an FD frame, ccfl 1024 or 768, no FAC): coef int32[n_ch, ccfl];
        ics uint8[n_ch, 2] (window_sequence 0..4, window_shape); overlap int32[n_ch, ccfl] in/out; shape_prev uint8[n_ch]
        in/out; out32 int32[n_ch, ccfl] (Q15) and / or time float32[n_ch, ccfl]; status int32[n_ch]."""
        n_ch = overlap.shape[0]
        b = _UsacImdctBatch()
        b.n_ch, b.ccfl = n_ch, int(ccfl)
        b.coef = _ptr(coef, "int32", n_ch * ccfl, device_ok=True)
        b.ics = _ptr(ics, "uint8", n_ch * 2, device_ok=True)
        b.overlap = _ptr(overlap, "int32", n_ch * ccfl, device_ok=True)
        b.shape_prev = _ptr(shape_prev, "uint8", n_ch, device_ok=True)
        b.out32 = _ptr(out32, "int32", n_ch * ccfl, allow_none=True, device_ok=True)
        b.time = _ptr(time, "float32", n_ch * ccfl, allow_none=True, device_ok=True)
        b.status = _ptr(status, "int32", n_ch, allow_none=True, device_ok=True)
        # LPD -> FD transitions: lpd_flags uint8[n_ch] (bit 0 td_frame_prev, bit 1 fac_data_present), fac int32[n_ch, 257]
        # (struct xaac_usac_fac: q, data[256])
        b.lpd_flags = _ptr(lpd_flags, "uint8", n_ch, allow_none=True, device_ok=True)
        b.fac = _ptr(fac, "int32", n_ch * 257, allow_none=True, device_ok=True)
        # ... or the signal made on the device (ixheaacd_cal_fac_data): fac_in int32 view of [n_ch] xaac_usac_fac_in, fac_work
        # int32[n_ch, 257] scratch
        b.fac_in = _ptr(fac_in, "int32", n_ch * USAC_FAC_IN_WORDS, allow_none=True, device_ok=True)
        b.fac_work = _ptr(fac_work, "int32", n_ch * 257, allow_none=True, device_ok=True)
        rc = self._lib.xaac_usac_imdct_process_batch(self._h, ctypes.byref(b))
        if rc != 0:
            raise XaacError(rc, "xaac_usac_imdct_process_batch")

    def hbe_real_synth_batch(self, qmf_re, qmf_im, state, status=None, num_columns=32):
        """Batched ixheaacd_real_synth_filt (the harmonic transposer's real synthesis bank): qmf_re / qmf_im
        float32[n_ch, num_columns, 64]; state uint8[n_ch, HBE_STATE_BYTES] in/out (struct xaac_hbe_state);
        status int32[n_ch] or None."""
        n_ch = state.shape[0]
        b = _HbeSynthBatch()
        b.n_ch, b.num_columns = n_ch, num_columns
        b.qmf_re = _ptr(qmf_re, "float32", n_ch * num_columns * 64, device_ok=True)
        b.qmf_im = _ptr(qmf_im, "float32", n_ch * num_columns * 64, device_ok=True)
        b.state = _ptr(state, "uint8", n_ch * HBE_STATE_BYTES, device_ok=True)
        b.status = _ptr(status, "int32", n_ch, device_ok=True) if status is not None else None
        rc = self._lib.xaac_hbe_real_synth_batch(self._h, ctypes.byref(b))
        if rc != 0:
            raise XaacError(rc, "xaac_hbe_real_synth_batch")

    def hbe_apply_batch(self, qmf_re, qmf_im, state, pv_re, pv_im, status=None, pitch_in_bins=None, max_synth_size=0):
        """Batched ixheaacd_qmf_hbe_apply (the QMF-domain harmonic transposer, frames without a pitch): qmf_re / qmf_im
        float32[n_ch, 32, 64]; state uint8[n_ch, HBE_STATE_BYTES] in/out; pv_re / pv_im float32[n_ch, 32, 64] (bands
        start_band..end_band-1 written); status int32[n_ch] or None; pitch_in_bins int32[n_ch] or None."""
        n_ch = state.shape[0]
        b = _HbeApplyBatch()
        b.n_ch = n_ch
        b.qmf_re = _ptr(qmf_re, "float32", n_ch * 2048, device_ok=True)
        b.qmf_im = _ptr(qmf_im, "float32", n_ch * 2048, device_ok=True)
        b.pitch_in_bins = _ptr(pitch_in_bins, "int32", n_ch, device_ok=True) if pitch_in_bins is not None else None
        b.state = _ptr(state, "uint8", n_ch * HBE_STATE_BYTES, device_ok=True)
        b.pv_re = _ptr(pv_re, "float32", n_ch * 2048, device_ok=True)
        b.pv_im = _ptr(pv_im, "float32", n_ch * 2048, device_ok=True)
        b.status = _ptr(status, "int32", n_ch, device_ok=True) if status is not None else None
        b.max_synth_size = int(max_synth_size)
        rc = self._lib.xaac_hbe_apply_batch(self._h, ctypes.byref(b))
        if rc != 0:
            raise XaacError(rc, "xaac_hbe_apply_batch")

    def hbe_dft_anal_batch(self, time_in, coef_re, coef_im, state, qmf_re, qmf_im, cfg=None, status=None, no_bins=32):
        """Batched ixheaacd_dft_hbe_cplx_anal_filt (the DFT transposer's analysis bank): time_in float32[n_ch, stride];
        coef_re / coef_im float32[n_cfg, 64, 128]; state uint8[n_ch, HBE_DFT_STATE_BYTES] in/out; qmf_re / qmf_im
        float32[n_ch, no_bins + 2, 64] in/out; cfg int32[n_ch] or None."""
        n_ch = state.shape[0]
        b = _HbeDftAnalBatch()
        b.n_ch, b.no_bins, b.in_stride = n_ch, no_bins, int(time_in.shape[1])
        b.time_in = _ptr(time_in, "float32", n_ch * b.in_stride, device_ok=True)
        b.coef_re = _ptr(coef_re, "float32", device_ok=True)
        b.coef_im = _ptr(coef_im, "float32", device_ok=True)
        b.cfg = _ptr(cfg, "int32", n_ch, device_ok=True) if cfg is not None else None
        b.state = _ptr(state, "uint8", n_ch * HBE_DFT_STATE_BYTES, device_ok=True)
        b.qmf_re = _ptr(qmf_re, "float32", n_ch * (no_bins + 2) * 64, device_ok=True)
        b.qmf_im = _ptr(qmf_im, "float32", n_ch * (no_bins + 2) * 64, device_ok=True)
        b.status = _ptr(status, "int32", n_ch, device_ok=True) if status is not None else None
        rc = self._lib.xaac_hbe_dft_anal_batch_run(self._h, ctypes.byref(b))
        if rc != 0:
            raise XaacError(rc, "xaac_hbe_dft_anal_batch_run")

    def hbe_dft_apply_batch(self, qmf_re, qmf_im, cfg_tab, coef_re, coef_im, state, pv_re, pv_im, status, pitch_in_bins=None,
                            oversampling=None, cfg=None, rows32=False):
        """Batched ixheaacd_dft_hbe_apply (the DFT harmonic transposer, -esbr_hq:1): qmf_re / qmf_im float32[n_ch, 32, 64];
        cfg_tab uint8[n_cfg, HBE_DFT_CFG_BYTES]; coef_re / coef_im float32[n_cfg, 64, 128]; state
        uint8[n_ch, HBE_DFT_FULL_STATE_BYTES] in/out; pv_re / pv_im float32[n_ch, 34, 64] in/out; status int32[n_ch];
        pitch_in_bins / oversampling / cfg int32[n_ch] or None."""
        n_ch = state.shape[0]
        b = _HbeDftApplyBatch()
        b.n_ch = n_ch
        b.qmf_re = _ptr(qmf_re, "float32", n_ch * 2048, device_ok=True)
        b.qmf_im = _ptr(qmf_im, "float32", n_ch * 2048, device_ok=True)
        b.pitch_in_bins = _ptr(pitch_in_bins, "int32", n_ch, device_ok=True) if pitch_in_bins is not None else None
        b.oversampling = _ptr(oversampling, "int32", n_ch, device_ok=True) if oversampling is not None else None
        b.cfg_tab = _ptr(cfg_tab, "uint8", device_ok=True)
        b.coef_re = _ptr(coef_re, "float32", device_ok=True)
        b.coef_im = _ptr(coef_im, "float32", device_ok=True)
        b.cfg = _ptr(cfg, "int32", n_ch, device_ok=True) if cfg is not None else None
        b.state = _ptr(state, "uint8", n_ch * HBE_DFT_FULL_STATE_BYTES, device_ok=True)
        b.rows32 = int(bool(rows32))      # pv blocks of 32 rows, written whole (see include/xaac_hbe.h)
        b.pv_re = _ptr(pv_re, "float32", n_ch * (32 if rows32 else 34) * 64, device_ok=True)
        b.pv_im = _ptr(pv_im, "float32", n_ch * (32 if rows32 else 34) * 64, device_ok=True)
        b.status = _ptr(status, "int32", n_ch, device_ok=True)
        rc = self._lib.xaac_hbe_dft_apply_batch_run(self._h, ctypes.byref(b))
        if rc != 0:
            raise XaacError(rc, "xaac_hbe_dft_apply_batch_run")

    def pvc_process_batch(self, frame, qmf_re, qmf_im, state, out, status=None):
        """Batched PVC envelope decoder (ixheaacd_qmf_enrg_calc + ixheaacd_pvc_process): frame uint8[n_ch, PVC_FRAME_BYTES];
        qmf_re / qmf_im float32[n_ch, rows, 64] (row 2 of the QMF buffers onwards; rows >= 32, and >= 64 for batches that hold
        pvc_rate 4 frames: on fewer rows such a frame is refused with status -1); state uint8[n_ch, PVC_STATE_BYTES]
        in/out; out float32[n_ch, 16, 64]; status int32[n_ch] or None."""
        n_ch = state.shape[0]
        b = _PvcBatch()
        if qmf_re.dim() != 3 or qmf_re.shape[2] != 64 or qmf_re.shape[1] < 32 or tuple(qmf_im.shape) != tuple(qmf_re.shape):
            raise ValueError("qmf_re / qmf_im must be [n_ch, rows >= 32, 64] and alike")
        b.n_ch, b.qmf_stride = n_ch, int(qmf_re.shape[1]) * int(qmf_re.shape[2])
        b.frame = _ptr(frame, "uint8", n_ch * PVC_FRAME_BYTES, device_ok=True)
        b.qmf_re = _ptr(qmf_re, "float32", n_ch * b.qmf_stride, device_ok=True)
        b.qmf_im = _ptr(qmf_im, "float32", n_ch * b.qmf_stride, device_ok=True)
        b.state = _ptr(state, "uint8", n_ch * PVC_STATE_BYTES, device_ok=True)
        b.out = _ptr(out, "float32", n_ch * 16 * 64, device_ok=True)
        b.status = _ptr(status, "int32", n_ch, device_ok=True) if status is not None else None
        rc = self._lib.xaac_pvc_process_batch(self._h, ctypes.byref(b))
        if rc != 0:
            raise XaacError(rc, "xaac_pvc_process_batch")

    def hbe_cplx_anal_batch(self, state, status=None):
        """Batched ixheaacd_complex_anal_filt (the harmonic transposer's complex analysis bank): state
        uint8[n_ch, HBE_STATE_BYTES] in/out; status int32[n_ch] or None."""
        n_ch = state.shape[0]
        b = _HbeAnalBatch()
        b.n_ch = n_ch
        b.state = _ptr(state, "uint8", n_ch * HBE_STATE_BYTES, device_ok=True)
        b.status = _ptr(status, "int32", n_ch, device_ok=True) if status is not None else None
        rc = self._lib.xaac_hbe_cplx_anal_batch(self._h, ctypes.byref(b))
        if rc != 0:
            raise XaacError(rc, "xaac_hbe_cplx_anal_batch")

    def qmf_analysis_eld_batch(self, pcm, state, qmf, n_slots, usb, status=None):
        """Batched LD / ELD complex analysis bank: pcm int16[n_ch, 32 * n_slots]; state int16[n_ch, 324] in/out (ring, wr,
        f1, f2, fp; a new stream: zeros with f2 = 32); qmf int32[n_ch, n_slots, slot_stride >= 96]."""
        n_ch = state.shape[0]
        b = _QmfAnaEldBatch()
        b.n_ch, b.n_slots, b.usb, b.slot_stride = n_ch, n_slots, usb, int(qmf.shape[2])
        b.pcm = _ptr(pcm, "int16", n_ch * 32 * n_slots, device_ok=True)
        b.state = _ptr(state, "int16", n_ch * QMF_ANA_ELD_STATE_WORDS, device_ok=True)
        b.qmf = _ptr(qmf, "int32", n_ch * n_slots * b.slot_stride, device_ok=True)
        b.status = _ptr(status, "int32", n_ch, device_ok=True) if status is not None else None
        rc = self._lib.xaac_qmf_analysis_eld_batch(self._h, ctypes.byref(b))
        if rc != 0:
            raise XaacError(rc, "xaac_qmf_analysis_eld_batch")

    def qmf_synthesis_eld_batch(self, qmf, scale, state, pcm, n_slots, lsb, usb, split, status=None, qmf_scaled=None):
        """Batched LD / ELD complex synthesis bank: qmf int32[n_ch, n_slots, slot_stride >= 128]; scale int16[n_ch, 4] (lb,
        ov_lb, hb, st_syn); state int16[n_ch, 1284] in/out (ring, drc_offset, phase, fp, sixty4; a new stream: zeros with
        sixty4 = 64); pcm int16[n_ch, 64 * n_slots]."""
        n_ch = state.shape[0]
        b = _QmfSynEldBatch()
        b.n_ch, b.n_slots, b.lsb, b.usb, b.split, b.slot_stride = n_ch, n_slots, lsb, usb, split, int(qmf.shape[2])
        b.qmf = _ptr(qmf, "int32", n_ch * n_slots * b.slot_stride, device_ok=True)
        b.scale = _ptr(scale, "int16", n_ch * 4, device_ok=True)
        b.state = _ptr(state, "int16", n_ch * QMF_SYN_ELD_STATE_WORDS, device_ok=True)
        b.pcm = _ptr(pcm, "int16", n_ch * 64 * n_slots, device_ok=True)
        b.status = _ptr(status, "int32", n_ch, device_ok=True) if status is not None else None
        b.qmf_scaled = _ptr(qmf_scaled, "int32", n_ch * n_slots * b.slot_stride, allow_none=True, device_ok=True)
        rc = self._lib.xaac_qmf_synthesis_eld_batch(self._h, ctypes.byref(b))
        if rc != 0:
            raise XaacError(rc, "xaac_qmf_synthesis_eld_batch")

    def esbr_qmf_analysis_batch(self, core, state, qmf_re, qmf_im):
        """Batched ixheaacd_esbr_analysis_filt_block (eSBR / Path A, 32 channels): core float32[n_ch, 1024];
        state int32[n_ch, 322] in/out (ring, pos, win_off); qmf_re / qmf_im float32[n_ch, 32, 64], bands 0..31 written."""
        n_ch = state.shape[0]
        b = _EsbrAnaBatch()
        b.n_ch = n_ch
        b.core = _ptr(core, "float32", n_ch * 1024, device_ok=True)
        b.state = _ptr(state, "int32", n_ch * ESBR_ANA_STATE_WORDS, device_ok=True)
        b.qmf_re = _ptr(qmf_re, "float32", n_ch * 2048, device_ok=True)
        b.qmf_im = _ptr(qmf_im, "float32", n_ch * 2048, device_ok=True)
        rc = self._lib.xaac_esbr_qmf_analysis_batch(self._h, ctypes.byref(b))
        if rc != 0:
            raise XaacError(rc, "xaac_esbr_qmf_analysis_batch")

    def esbr_qmf_analysis_nb_batch(self, n_bands, n_slots, core, state, qmf_re, qmf_im):
        """The 24- / 16-channel analysis banks of 8:3 / 4:1 SBR (sbr_dec.c:213-236): core float32[n_ch, core_stride];
        state int32[n_ch, 322] in/out; qmf_re / qmf_im float32[n_ch, n_slots, 64]."""
        n_ch = state.shape[0]
        b = _EsbrAnaNbBatch()
        b.n_ch, b.n_bands, b.n_slots = n_ch, n_bands, n_slots
        b.core_stride = int(core.shape[1])
        b.core = _ptr(core, "float32", n_ch * b.core_stride, device_ok=True)
        b.state = _ptr(state, "int32", n_ch * ESBR_ANA_STATE_WORDS, device_ok=True)
        b.qmf_re = _ptr(qmf_re, "float32", n_ch * n_slots * 64, device_ok=True)
        b.qmf_im = _ptr(qmf_im, "float32", n_ch * n_slots * 64, device_ok=True)
        rc = self._lib.xaac_esbr_qmf_analysis_nb_batch(self._h, ctypes.byref(b))
        if rc != 0:
            raise XaacError(rc, "xaac_esbr_qmf_analysis_nb_batch")

    def esbr_qmf_synthesis_batch(self, qmf_re, qmf_im, state, out):
        """Batched synthesis bank of ixheaacd_esbr_synthesis_filt_block (eSBR / Path A, 64 channels):
        qmf_re / qmf_im float32[n_ch, 32, 64]; state int32[n_ch, 1282] in/out (ring, drc_offset, filt_off);
        out float32[n_ch, 2048]."""
        n_ch = state.shape[0]
        b = _EsbrSynBatch()
        b.n_ch = n_ch
        b.qmf_re = _ptr(qmf_re, "float32", n_ch * 2048, device_ok=True)
        b.qmf_im = _ptr(qmf_im, "float32", n_ch * 2048, device_ok=True)
        b.state = _ptr(state, "int32", n_ch * ESBR_SYN_STATE_WORDS, device_ok=True)
        b.out = _ptr(out, "float32", n_ch * 2048, device_ok=True)
        rc = self._lib.xaac_esbr_qmf_synthesis_batch(self._h, ctypes.byref(b))
        if rc != 0:
            raise XaacError(rc, "xaac_esbr_qmf_synthesis_batch")

    def esbr_qmf_synthesis_ds_batch(self, qmf_re, qmf_im, state, out):
        """The same bank down-sampled (32 synthesis channels, sbr_dec.c:605-628): out float32[n_ch, 1024]."""
        n_ch = state.shape[0]
        b = _EsbrSynBatch()
        b.n_ch = n_ch
        b.qmf_re = _ptr(qmf_re, "float32", n_ch * 2048, device_ok=True)
        b.qmf_im = _ptr(qmf_im, "float32", n_ch * 2048, device_ok=True)
        b.state = _ptr(state, "int32", n_ch * ESBR_SYN_STATE_WORDS, device_ok=True)
        b.out = _ptr(out, "float32", n_ch * 1024, device_ok=True)
        rc = self._lib.xaac_esbr_qmf_synthesis_ds_batch(self._h, ctypes.byref(b))
        if rc != 0:
            raise XaacError(rc, "xaac_esbr_qmf_synthesis_ds_batch")

    def sbr_eld_workspace_bytes(self, n_ch):
        return int(self._lib.xaac_sbr_eld_workspace_bytes(int(n_ch)))

    def sbr_eld_process_batch(self, pcm_in, header, frame, state, pcm_out, workspace, n_slots, status=None, in_ch_fac=1,
                              out_ch_fac=1, qmf_handed_on=None):
        """Batched ixheaacd_sbr_dec for AAC-ELD channels (low-delay SBR): pcm_in int16[n_ch * 32 n_slots], state
        uint8[n_ch, SBR_ELD_STATE_BYTES] (xaac_sbr_eld_state), pcm_out int16[n_ch * 64 n_slots]; n_slots 16 or 15"""
        n_ch = state.shape[0]
        b = _SbrEldBatch()
        b.n_ch, b.n_slots, b.in_ch_fac, b.out_ch_fac = n_ch, int(n_slots), int(in_ch_fac), int(out_ch_fac)
        b.pcm_in = _ptr(pcm_in, "int16", n_ch * 32 * int(n_slots), device_ok=True)
        b.header = _ptr(header, "uint8", n_ch * SBR_HEADER_BYTES, device_ok=True)
        b.frame = _ptr(frame, "uint8", n_ch * SBR_FRAME_BYTES, device_ok=True)
        b.state = _ptr(state, "uint8", n_ch * SBR_ELD_STATE_BYTES, device_ok=True)
        b.pcm_out = _ptr(pcm_out, "int16", n_ch * 64 * int(n_slots), device_ok=True)
        b.status = _ptr(status, "int32", n_ch, allow_none=True, device_ok=True)
        b.workspace = _ptr(workspace, "uint8", device_ok=True)
        b.workspace_bytes = workspace.numel()
        b.qmf_handed_on = _ptr(qmf_handed_on, "int32", n_ch * int(n_slots) * 128, allow_none=True, device_ok=True)
        rc = self._lib.xaac_sbr_eld_process_batch(self._h, ctypes.byref(b))
        if rc != 0:
            raise XaacError(rc, "xaac_sbr_eld_process_batch")

    def sbr_hq_workspace_bytes(self, n_ch, with_ps=True):
        return int(self._lib.xaac_sbr_hq_workspace_bytes(int(n_ch), int(bool(with_ps))))

    def sbr_hq_process_batch(self, pcm_in, header, frame, state, pcm_out, workspace, ps_frame=None, ps_state=None,
                             status=None, in_ch_fac=1, out_ch_fac=1, down_sample=False, max_band_hint=0):
        """Batched ixheaacd_sbr_dec, HQ mode: one frame per stream.  With ps_frame / ps_state (uint8[n, PS_*_BYTES])
        the parametric-stereo tool runs too (HE-AACv2) and pcm_out is int16[n*2048*2] of L,R pairs; without them
        pcm_out is int16[n*2048] (HE-AAC mono, HQ)."""
        n_ch = state.shape[0]
        with_ps = ps_frame is not None
        b = _SbrHqBatch()
        b.n_ch, b.in_ch_fac, b.out_ch_fac = n_ch, int(in_ch_fac), int(out_ch_fac)
        b.down_sample = int(bool(down_sample))
        b.pcm_in = _ptr(pcm_in, "int16", n_ch * 1024, device_ok=True)
        b.header = _ptr(header, "uint8", n_ch * SBR_HEADER_BYTES, device_ok=True)
        b.frame = _ptr(frame, "uint8", n_ch * SBR_FRAME_BYTES, device_ok=True)
        b.state = _ptr(state, "uint8", n_ch * SBR_STATE_BYTES, device_ok=True)
        b.ps_frame = _ptr(ps_frame, "uint8", n_ch * PS_FRAME_BYTES, allow_none=True, device_ok=True)
        b.ps_state = _ptr(ps_state, "uint8", n_ch * PS_STATE_BYTES, allow_none=True, device_ok=True)
        b.pcm_out = _ptr(pcm_out, "int16", n_ch * (1024 if down_sample else 2048) * (2 if with_ps else 1), device_ok=True)
        b.status = _ptr(status, "int32", n_ch, allow_none=True, device_ok=True)
        b.workspace = _ptr(workspace, "uint8", device_ok=True)
        b.workspace_bytes = workspace.numel()
        b.max_band_hint = int(max_band_hint)   # 48: no stream of the batch reaches above QMF band 48 (xaac_amd.h); 0: no assertion
        rc = self._lib.xaac_sbr_hq_process_batch(self._h, ctypes.byref(b))
        if rc != 0:
            raise XaacError(rc, "xaac_sbr_hq_process_batch")

    def peak_limiter_workspace_bytes(self, n_streams):
        return int(self._lib.xaac_peak_limiter_workspace_bytes(int(n_streams)))

    def peak_limiter_process_batch(self, samples, qshift_adj, state, num_channels, workspace, frame_len=1024,
                                   pcm16=None, stride=None, status=None, planar=False):
        """Batched ixheaacd_peak_limiter_process (+ the round16 hand-off): one frame of every stream.
        samples int32[n_streams * stride] in/out, frame_len x num_channels interleaved per stream (what
        imdct_process_batch leaves in out32); qshift_adj int8[n_streams * num_channels]; state
        uint8[n_streams, LIMITER_STATE_BYTES] in/out (peak_limiter_init() makes one); pcm16 optional
        int16[n_streams * frame_len * num_channels]; workspace uint8[>= peak_limiter_workspace_bytes(n_streams)]."""
        n = state.shape[0]
        if stride is None:
            stride = frame_len * num_channels
        b = _LimiterBatch()
        b.n_streams, b.frame_len, b.num_channels, b.stride = n, int(frame_len), int(num_channels), int(stride)
        b.planar = int(bool(planar))
        b.samples = _ptr(samples, "int32", n * stride, device_ok=True)
        b.qshift_adj = _ptr(qshift_adj, "int8", n * num_channels, device_ok=True)
        b.state = _ptr(state, "uint8", n * LIMITER_STATE_BYTES, device_ok=True)
        b.pcm16 = _ptr(pcm16, "int16", n * frame_len * num_channels, allow_none=True, device_ok=True)
        b.status = _ptr(status, "int32", n, allow_none=True, device_ok=True)
        b.workspace = _ptr(workspace, "uint8", device_ok=True)
        b.workspace_bytes = workspace.numel()
        rc = self._lib.xaac_peak_limiter_process_batch(self._h, ctypes.byref(b))
        if rc != 0:
            raise XaacError(rc, "xaac_peak_limiter_process_batch")

    def sbr_lp_workspace_bytes(self, n_ch):
        return int(self._lib.xaac_sbr_lp_workspace_bytes(int(n_ch)))

    def sbr_lp_process_batch(self, pcm_in, header, frame, state, pcm_out, workspace, status=None, in_ch_fac=1,
                             out_ch_fac=1, down_sample=False):
        """Batched ixheaacd_sbr_dec, low-power mode (HE-AACv1): one frame per channel.
        pcm_in int16[n_ch*1024]; header/frame/state uint8[n_ch, SBR_*_BYTES] (structs of include/xaac_sbr.h,
        state in/out); pcm_out int16[n_ch*2048]; workspace uint8[>= sbr_lp_workspace_bytes(n_ch)];
        status optional int32[n_ch]."""
        n_ch = state.shape[0]
        b = _SbrLpBatch()
        b.n_ch, b.in_ch_fac, b.out_ch_fac = n_ch, int(in_ch_fac), int(out_ch_fac)
        b.down_sample = int(bool(down_sample))
        b.pcm_in = _ptr(pcm_in, "int16", n_ch * 1024, device_ok=True)
        b.header = _ptr(header, "uint8", n_ch * SBR_HEADER_BYTES, device_ok=True)
        b.frame = _ptr(frame, "uint8", n_ch * SBR_FRAME_BYTES, device_ok=True)
        b.state = _ptr(state, "uint8", n_ch * SBR_STATE_BYTES, device_ok=True)
        b.pcm_out = _ptr(pcm_out, "int16", n_ch * (1024 if down_sample else 2048), device_ok=True)
        b.status = _ptr(status, "int32", n_ch, allow_none=True, device_ok=True)
        b.workspace = _ptr(workspace, "uint8", device_ok=True)
        b.workspace_bytes = workspace.numel()
        rc = self._lib.xaac_sbr_lp_process_batch(self._h, ctypes.byref(b))
        if rc != 0:
            raise XaacError(rc, "xaac_sbr_lp_process_batch")
