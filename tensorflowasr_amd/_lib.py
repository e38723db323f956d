"""ctypes binding of libmi355asr.so (include/mi355asr.h).  There is no fallback: if the HIP library is
missing or a call fails, this module raises -- the product path never computes on the CPU."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MI355ASR_LIB: another build of the same library (kernel-variant experiments, tools/build_variant.py); still a HIP library
LIB_PATH = os.environ.get("MI355ASR_LIB") or os.path.join(_HERE, "libmi355asr.so")


class Config(ctypes.Structure):
    """mirror of `mi355asr_config` (include/mi355asr.h)."""
    _fields_ = [
        ("dmodel", ctypes.c_int32), ("num_blocks", ctypes.c_int32), ("head_size", ctypes.c_int32),
        ("num_heads", ctypes.c_int32), ("kernel_size", ctypes.c_int32), ("fc_factor", ctypes.c_float),
        ("reduction_factor", ctypes.c_int32), ("n_mels", ctypes.c_int32), ("sample_rate", ctypes.c_int32),
        ("stride_ms", ctypes.c_int32), ("n_dft", ctypes.c_int32), ("chunk_size", ctypes.c_int32),
        ("has_encoder", ctypes.c_int32), ("num_classes", ctypes.c_int32), ("ctc_num_blocks", ctypes.c_int32),
        ("ctc_kernel_size", ctypes.c_int32), ("ctc_fc_factor", ctypes.c_float), ("gemm_dtype", ctypes.c_int32),
        ("mel_layer_type", ctypes.c_int32), ("add_wav_info", ctypes.c_int32),
    ]


class ChunkConfig(ctypes.Structure):
    """mirror of `mi355asr_chunk_config` (include/mi355asr.h)."""
    _fields_ = [(n, ctypes.c_int32) for n in ("dmodel", "head_size", "num_heads", "kernel_size")] + \
        [("fc_factor", ctypes.c_float)] + \
        [(n, ctypes.c_int32) for n in (
            "n_mels", "sample_rate", "stride_ms", "n_dft", "reduction_factor",
            "enc_num_blocks", "enc_win_front", "enc_win_back",
            "picker_num_classes", "picker_num_blocks", "picker_win_front", "picker_win_back",
            "helper_num_blocks", "helper_win_front", "helper_win_back",
            "decoder_num_classes", "decoder_num_blocks", "decoder_win_front", "decoder_win_back")]


class TranslatorConfig(ctypes.Structure):
    """mirror of `mi355asr_translator_config`."""
    _fields_ = [(n, ctypes.c_int32) for n in ("dmodel", "num_blocks", "head_size", "num_heads", "kernel_size")] + \
        [("fc_factor", ctypes.c_float), ("inp_classes", ctypes.c_int32), ("tar_classes", ctypes.c_int32)]


class ChunkOutputs(ctypes.Structure):
    """mirror of `mi355asr_chunk_outputs`."""
    _fields_ = [(n, ctypes.c_void_p) for n in ("front_out", "enc_out", "picker_logits", "picker_hidden", "picked",
                                               "helper_out", "text_logits", "text_argmax")]


class Mi355AsrError(RuntimeError):
    pass


_P = ctypes.c_void_p
_I = ctypes.c_int32
_SZ = ctypes.c_size_t

# name -> (restype, argtypes): every symbol include/mi355asr.h declares
SIGNATURES = {
    "mi355asr_last_error": (ctypes.c_char_p, []),
    "mi355asr_version": (ctypes.c_char_p, []),
    "mi355asr_create": (ctypes.c_int, [ctypes.POINTER(Config), ctypes.POINTER(_P)]),
    "mi355asr_destroy": (ctypes.c_int, [_P]),
    "mi355asr_load_weight": (ctypes.c_int, [_P, ctypes.c_char_p, _P, _I, ctypes.POINTER(ctypes.c_int64)]),
    "mi355asr_load_weight_typed": (ctypes.c_int, [_P, ctypes.c_char_p, _P, _I, _I, ctypes.POINTER(ctypes.c_int64)]),
    "mi355asr_ctc_prefix_beam_workspace_bytes": (ctypes.c_int, [_I, _I, _I, _I, _I, ctypes.POINTER(ctypes.c_size_t)]),
    "mi355asr_frame_argmax": (ctypes.c_int, [_P, _I, _I, _P, _P]),
    "mi355asr_feature_pick_count": (ctypes.c_int, [_P, _I, _I, _I, _P, _P, _P, _P]),
    "mi355asr_feature_pick_gather": (ctypes.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "mi355asr_num_weights": (ctypes.c_int, [_P]),
    "mi355asr_stft_mode": (ctypes.c_int, [_P]),
    "mi355asr_weight_name": (ctypes.c_char_p, [_P, _I]),
    "mi355asr_weight_shape": (ctypes.c_int, [_P, _I, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int64), _I]),
    "mi355asr_finalize_weights": (ctypes.c_int, [_P, _P]),
    "mi355asr_set_expected_rows": (ctypes.c_int, [_P, ctypes.c_int64]),
    "mi355asr_out_frames": (ctypes.c_int, [_P, _I, ctypes.POINTER(_I), ctypes.POINTER(_I)]),
    "mi355asr_workspace_bytes": (ctypes.c_int, [_P, _I, _I, ctypes.POINTER(_SZ)]),
    "mi355asr_ctc_workspace_bytes": (ctypes.c_int, [_P, _I, _I, ctypes.POINTER(_SZ)]),
    "mi355asr_encoder_forward": (ctypes.c_int, [_P, _P, _I, _I, _P, _P, _SZ, _P]),
    "mi355asr_ctc_forward": (ctypes.c_int, [_P, _P, _I, _I, _P, _P, _P, _SZ, _P]),
    "mi355asr_ctc_greedy": (ctypes.c_int, [_P, _P, _I, _I, _I, _P, _P, _P]),
    "mi355asr_ctc_prefix_beam_host": (ctypes.c_int, [_P, _P, _I, _I, _I, _I, ctypes.c_double, _I, _I, _I, _P, _P, _P, _P]),
    "mi355asr_ctc_prefix_beam": (ctypes.c_int, [_P, _I, _P, _I, _I, _I, _I, ctypes.c_double, _I, _I, _I, _P, _P, _P, _P,
                                                _P, _SZ, _P]),
    "mi355asr_beam_math_eval": (ctypes.c_int, [_I, _P, _P, _I, _P]),
    "mi355asr_beam_create": (ctypes.c_int, [_I, _I, ctypes.c_double, _I, ctypes.POINTER(_P)]),
    "mi355asr_beam_decode": (ctypes.c_int, [_P, _P, _I, _I, _P, _P, _P, _P]),
    "mi355asr_beam_reset": (ctypes.c_int, [_P]),
    "mi355asr_beam_destroy": (ctypes.c_int, [_P]),
    "mi355asr_recognize": (ctypes.c_int, [_P, _P, _I, _I, _P, _P, _P, _P, _SZ, _P]),
    "mi355asr_melspectrogram": (ctypes.c_int, [_P, _P, _I, _I, _P, _P, _SZ, _P]),
    "mi355asr_conv_subsampling": (ctypes.c_int, [_P, _P, _I, _I, _P, _P, _SZ, _P]),
    "mi355asr_conformer_block": (ctypes.c_int, [_P, _I, _I, _P, _I, _I, _P, _P, _SZ, _P]),
    "mi355asr_chunk_create": (ctypes.c_int, [ctypes.POINTER(ChunkConfig), ctypes.POINTER(_P)]),
    "mi355asr_chunk_out_frames": (ctypes.c_int, [_P, _I, ctypes.POINTER(_I), ctypes.POINTER(_I)]),
    "mi355asr_chunk_workspace_bytes": (ctypes.c_int, [_P, _I, _I, ctypes.POINTER(_SZ)]),
    "mi355asr_chunk_predict": (ctypes.c_int, [_P, _P, _I, _I, ctypes.POINTER(ChunkOutputs), _P, _P, _P, _SZ, _P]),
    "mi355asr_chunk_front_stream_shape": (ctypes.c_int, [_P, _I, _I, _I, ctypes.POINTER(_I), ctypes.POINTER(_I)]),
    "mi355asr_chunk_stream_workspace_bytes": (ctypes.c_int, [_P, _I, _I, _I, _I, ctypes.POINTER(_SZ)]),
    "mi355asr_chunk_front_stream": (ctypes.c_int, [_P, _P, _I, _P, _I, _I, _P, _P, _P, _SZ, _P]),
    "mi355asr_chunk_stack_stream": (ctypes.c_int, [_P, _I, _P, _I, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "mi355asr_translator_create": (ctypes.c_int, [ctypes.POINTER(TranslatorConfig), ctypes.POINTER(_P)]),
    "mi355asr_translator_workspace_bytes": (ctypes.c_int, [_P, _I, _I, _I, ctypes.POINTER(_SZ)]),
    "mi355asr_translator_forward": (ctypes.c_int, [_P, _P, _P, _I, _I, _I, _P, _P, _P, _SZ, _P]),
    "mi355asr_profile_enable": (ctypes.c_int, [_P, _I]),
    "mi355asr_profile_schemes": (ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_int32), _I]),
    "mi355asr_profile_read": (ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64), _I, _I]),
}

KERNEL_NAMES = ["stft", "utt_max", "mel", "subconv", "sublinear", "ffn", "qkv", "attention", "attn_out", "pw1_glu",
                "dwconv", "conv_tail", "ctc_project", "ctc_head", "collapse", "ff1_qkv", "out_glu", "tail_ff2", "tail_ff1", "enc_stack"]

_lib = None


def lib():
    """Load libmi355asr.so once.  Raises if it has not been built (python -m tensorflowasr_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Mi355AsrError(
                "libmi355asr.so not found at %s -- build it with `python -m tensorflowasr_amd.build` "
                "(there is no CPU fallback)" % LIB_PATH)
        # torch first: its wheel bundles its own libamdhip64; loading ours before it would bring a second HIP runtime
        # into the process (the one under /opt/rocm) and device memory from one is invisible to the other
        import torch  # noqa: F401
        h = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


def check(rc):
    if rc != 0:
        raise Mi355AsrError("mi355asr error %d: %s" % (rc, lib().mi355asr_last_error().decode()))
