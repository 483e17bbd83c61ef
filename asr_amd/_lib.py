"""ctypes binding of libasrk.so (the C ABI declared in include/asrk.h).

The library is built in-tree by ``asr_amd/csrc/Makefile`` (``__graft_entry__.build()``).  There is
no fallback: if the shared object is missing or an entry point fails, the product raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# ASRK_LIB_PATH: A/B measurements of two builds inside one GPU call (device-to-device spread is 1-3 %)
LIB_PATH = os.environ.get("ASRK_LIB_PATH") or os.path.join(_HERE, "libasrk.so")

ASRK_F32, ASRK_BF16, ASRK_I16 = 0, 1, 2
(ASRK_OK, ASRK_ERR_NULL, ASRK_ERR_SHAPE, ASRK_ERR_ALIGN, ASRK_ERR_HIP, ASRK_ERR_WORKSPACE, ASRK_ERR_UNSUPPORTED,
 ASRK_ERR_LENGTH) = 0, -1, -2, -3, -4, -5, -6, -7


class AsrkError(RuntimeError):
    pass


class EncoderDims(Structure):
    _fields_ = [("d_model", c_int), ("n_layers", c_int), ("n_heads", c_int), ("ffn_dim", c_int), ("n_mels", c_int)]


class EncoderLayerWeights(Structure):
    _fields_ = [(n, c_void_p) for n in (
        "q_w", "q_b", "k_w", "v_w", "v_b", "o_w", "o_b", "ln1_w", "ln1_b",
        "fc1_w", "fc1_b", "fc2_w", "fc2_b", "ln2_w", "ln2_b")]


class EncoderWeights(Structure):
    _fields_ = [(n, c_void_p) for n in ("conv1_w", "conv1_b", "conv2_w", "conv2_b", "pos", "ln_w", "ln_b")] + [
        ("layers", POINTER(EncoderLayerWeights))]


class ProjectorDims(Structure):
    _fields_ = [("audio_dim", c_int), ("text_dim", c_int)]


class ProjectorParams(Structure):
    _fields_ = [(n, c_void_p) for n in ("gamma", "w1", "b1", "w2", "b2", "packed")]


class ProjectorGrads(Structure):
    _fields_ = [(n, c_void_p) for n in ("gamma", "w1", "b1", "w2", "b2")]


_SIGNATURES = {
    # name: (restype, argtypes)
    "asrk_version": (c_int, []),
    "asrk_strerror": (c_char_p, [c_int]),
    "asrk_logmel_plan_create": (c_int, [c_int, c_void_p, POINTER(c_void_p)]),
    "asrk_logmel_plan_destroy": (None, [c_void_p]),
    "asrk_logmel_workspace_bytes": (c_size_t, [c_int, c_int]),
    "asrk_logmel_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "asrk_logmel_f32_noalloc": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "asrk_encoder_create": (c_int, [POINTER(EncoderDims), POINTER(EncoderWeights), c_void_p, POINTER(c_void_p)]),
    "asrk_encoder_destroy": (None, [c_void_p]),
    "asrk_encoder_workspace_bytes": (c_size_t, [c_void_p, c_int]),
    "asrk_encoder_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "asrk_projector_workspace_bytes": (c_size_t, [POINTER(ProjectorDims), c_int]),
    "asrk_projector_saved_bytes": (c_size_t, [POINTER(ProjectorDims), c_int]),
    "asrk_projector_packed_bytes": (c_size_t, [POINTER(ProjectorDims)]),
    "asrk_projector_pack": (c_int, [POINTER(ProjectorDims), POINTER(ProjectorParams), c_void_p, c_void_p]),
    "asrk_encoder_projector_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t,
                                               POINTER(ProjectorDims), POINTER(ProjectorParams), c_void_p, c_int, c_void_p,
                                               c_void_p, c_size_t, c_void_p]),
    "asrk_projector_forward": (c_int, [POINTER(ProjectorDims), POINTER(ProjectorParams), c_void_p, c_int, c_void_p,
                                       c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "asrk_projector_backward": (c_int, [POINTER(ProjectorDims), POINTER(ProjectorParams), c_void_p, c_int, c_int,
                                        c_void_p, POINTER(ProjectorGrads), c_void_p, c_void_p, c_void_p, c_size_t,
                                        c_void_p]),
    "asrk_gemm_bf16": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int64,
                               c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "asrk_layernorm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    "asrk_collate_pcm": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "asrk_find_token_i64": (c_int, [c_void_p, c_int, c_int, ctypes.c_int64, c_void_p, c_void_p]),
    "asrk_embed_splice": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                  c_void_p, c_void_p]),
    "asrk_embed_splice_backward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "asrk_add_layernorm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float,
                                   c_void_p]),
    "asrk_attention_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "asrk_attention_log2_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "asrk_profile_enable": (c_int, [c_int]),
    "asrk_profile_read": (c_int, [c_int, POINTER(ctypes.c_double), POINTER(ctypes.c_double), POINTER(c_int), c_int]),
}

PROF_GEMM, PROF_ATTENTION, PROF_LOGMEL, PROF_NORM, PROF_SPLICE, PROF_GEMM256, PROF_GEMM128 = 0, 1, 2, 3, 4, 5, 6


def profile_read(cat: int, reset: bool = True):
    """(total_ms, total_work, launches) of one kernel category since the last reset (stream must be idle)."""
    ms, work, n = ctypes.c_double(), ctypes.c_double(), c_int()
    check(load().asrk_profile_read(cat, ctypes.byref(ms), ctypes.byref(work), ctypes.byref(n), int(reset)))
    return ms.value, work.value, n.value

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None
_owner_pid = None  # the process that loaded the library (and, through it, initialised HIP)

_FORK_MSG = (
    "asr_amd was called from a forked child process (pid {pid}, library loaded in pid {owner}): a HIP context does "
    "not survive fork(), so the MI355X kernels cannot run here.  The reference runs its feature extractor inside "
    "DataLoader workers (src/train.py:172-180 with dataloader_num_workers 6/16): set dataloader_num_workers=0, or keep "
    "the workers on the CPU-only host steps (LogMelFeatureExtractor.prepare_host / asr_amd.collate.pack_ragged) and "
    "run the extractor in the training process through asr_amd.collate.FeaturePrefetcher (INTEGRATION.md, section 2)."
)


def assert_not_forked() -> None:
    """the wrapper must detect use from a forked DataLoader worker instead of hanging or
    faulting inside HIP.  Two cases: this process is a fork of the one that loaded the library, or it is a fork of a
    process that had already initialised torch's GPU runtime (torch tracks that itself)."""
    pid = os.getpid()
    if _owner_pid is not None and pid != _owner_pid:
        raise AsrkError(_FORK_MSG.format(pid=pid, owner=_owner_pid))
    try:
        import torch

        bad = torch.cuda._is_in_bad_fork()
    except Exception:  # torch absent or without that probe: the pid check above is what remains
        bad = False
    if bad:
        raise AsrkError(_FORK_MSG.format(pid=pid, owner="the parent, which had initialised the GPU"))


def load() -> ctypes.CDLL:
    """Load libasrk.so once; raise (never fall back) if it is absent or incomplete, or if this is a forked child of
    the process that loaded it."""
    global _lib, _owner_pid
    if _lib is not None:
        if os.getpid() != _owner_pid:
            assert_not_forked()
        return _lib
    assert_not_forked()
    if not os.path.exists(LIB_PATH):
        raise AsrkError(
            f"{LIB_PATH} not found: build it with `make -C asr_amd/csrc` (or __graft_entry__.build()). "
            "asr_amd has no CPU fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.asrk_version() < 100:
        raise AsrkError("libasrk.so is older than this Python package")
    _lib, _owner_pid = lib, os.getpid()
    return lib


def check(code: int, what: str = "asrk") -> None:
    if code != 0:
        msg = load().asrk_strerror(code).decode()
        raise AsrkError(f"{what} failed: {msg} (code {code})")


def ptr(t) -> c_void_p:
    """data_ptr of a torch tensor (or None) as c_void_p."""
    return c_void_p(0 if t is None else t.data_ptr())


def current_stream():
    import torch

    return c_void_p(torch.cuda.current_stream().cuda_stream)
