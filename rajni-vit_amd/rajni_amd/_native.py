"""ctypes binding of librajni_hip.so (C ABI: include/rajni_hip.h).

PyTorch is plumbing here: it owns device memory and the stream; every kernel is ours.  There is NO
fallback: if the library is missing or the tensors are not on a ROCm device the call raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RAJNI_HIP_LIB") or os.path.join(_HERE, "lib", "librajni_hip.so")

RAJNI_F32, RAJNI_BF16 = 0, 1
EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RESID = 0, 1, 2
NUM_KCLASS = 17

c_void_p, c_int, c_long, c_float, c_size_t = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_size_t


class NativeError(RuntimeError):
    pass


class LinearArgs(C.Structure):
    _fields_ = [("x", c_void_p), ("lda", c_long), ("w", c_void_p), ("ldw", c_long),
                ("bias", c_void_p), ("gamma", c_void_p), ("resid", c_void_p), ("ldr", c_long),
                ("r_idx", c_void_p), ("r_np", c_int), ("r_nsrc", c_int),
                ("y", c_void_p), ("ldc", c_long), ("M", c_int), ("N", c_int), ("K", c_int),
                ("epilogue", c_int), ("dtype", c_int), ("stream_f32", c_int), ("w_scale", c_void_p),
                ("x_scale", c_void_p), ("y_scale", c_void_p)]


class Block(C.Structure):
    _fields_ = [("norm1_w", c_void_p), ("norm1_b", c_void_p),
                ("qkv_w", c_void_p), ("qkv_b", c_void_p),
                ("proj_w", c_void_p), ("proj_b", c_void_p), ("ls1", c_void_p),
                ("norm2_w", c_void_p), ("norm2_b", c_void_p),
                ("fc1_w", c_void_p), ("fc1_b", c_void_p),
                ("fc2_w", c_void_p), ("fc2_b", c_void_p), ("ls2", c_void_p),
                ("keep", c_int), ("update", c_int),
                ("keep_idx", c_void_p), ("scores", c_void_p), ("next_scores", c_void_p),
                ("forced_keep_idx", c_void_p),
                ("qkv_s", c_void_p), ("proj_s", c_void_p), ("fc1_s", c_void_p), ("fc2_s", c_void_p),
                ("fc1_rownorm_max", c_float), ("fc1_bias_absmax", c_float), ("attn_out_scale", c_float)]


class VitPlan(C.Structure):
    _fields_ = [("dtype", c_int), ("B", c_int), ("in_chans", c_int), ("img_size", c_int),
                ("patch_size", c_int), ("C", c_int), ("H", c_int), ("D", c_int), ("depth", c_int),
                ("hidden", c_int), ("num_classes", c_int), ("ln_eps", c_float),
                ("attn_scale", c_float), ("pos_has_cls", c_int),
                ("patch_w", c_void_p), ("patch_b", c_void_p), ("cls_token", c_void_p),
                ("pos_embed", c_void_p), ("blocks", C.POINTER(Block)),
                ("norm_w", c_void_p), ("norm_b", c_void_p), ("head_w", c_void_p), ("head_b", c_void_p),
                ("workspace", c_void_p), ("workspace_bytes", c_size_t),
                ("token_counts", C.POINTER(C.c_int32)), ("logits_ld", c_int), ("cls_only_last_block", c_int),
                ("resid_bf16", c_int), ("act_fp8", c_int)]


_SIGS = {
    "rajni_abi_version": (c_int, []),
    "rajni_last_error": (C.c_char_p, []),
    "rajni_device_check": (c_int, []),
    "rajni_importance": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "rajni_select_topk": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "rajni_score_select": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p, c_void_p,
                                   c_void_p, c_int, c_void_p]),
    "rajni_gather_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "rajni_attention": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float,
                                c_int, c_void_p]),
    "rajni_attention_fp8": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float,
                                    c_void_p]),
    "rajni_layernorm": (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float,
                                c_int, c_int, c_void_p]),
    "rajni_layernorm_fp8": (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float,
                                    c_int, c_int, c_float, c_int, c_void_p]),
    "rajni_linear": (c_int, [C.POINTER(LinearArgs), c_void_p]),
    "rajni_debug_force_gemm_tiling": (None, [c_int]),
    "rajni_debug_force_f8_tiling": (None, [c_int]),
    "rajni_debug_set_resid_stagger": (None, [c_int]),
    "rajni_debug_set_gemm_nblock_bytes": (None, [c_int]),
    "rajni_debug_force_attention": (None, [c_int]),
    "rajni_debug_force_score_two_pass": (None, [c_int]),
    "rajni_debug_set_gemm_stamps": (None, [c_void_p]),
    "rajni_patch_embed_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "rajni_patch_embed": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int,
                                  c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "rajni_vit_workspace_bytes": (c_size_t, [C.POINTER(VitPlan)]),
    "rajni_vit_forward": (c_int, [C.POINTER(VitPlan), c_void_p, c_void_p, c_void_p]),
    "rajni_profile_enable": (None, [C.c_uint]),
    "rajni_profile_class_name": (C.c_char_p, [c_int]),
    "rajni_profile_collect": (c_int, [C.POINTER(C.c_longlong), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                      C.POINTER(C.c_double)]),
    "rajni_profile_reset": (None, []),
}

EXPORTED_SYMBOLS = tuple(_SIGS.keys())
ABI_VERSION = 8     # include/rajni_hip.h; bumped whenever a struct or an entry point changes
_lib: Optional[C.CDLL] = None


def load_library(path: str = LIB_PATH) -> C.CDLL:
    """Load the C-ABI library (no GPU needed to load it).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise NativeError(
            f"librajni_hip.so not found at {path}. Build it with `python rajni-vit_amd/build.py` "
            "(hipcc --offload-arch=gfx950). rajni_amd has no CPU or PyTorch fallback by design.")
    lib = C.CDLL(path)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.rajni_abi_version() != ABI_VERSION:   # struct layouts below would not match: refuse, do not guess
        raise NativeError(f"{path} has ABI version {lib.rajni_abi_version()}, this package binds version {ABI_VERSION}: "
                          "rebuild it with `python rajni-vit_amd/build.py --force`")
    _lib = lib
    return lib


def lib() -> C.CDLL:
    return load_library()


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().rajni_last_error().decode("utf-8", "replace")
        if rc == 2:
            raise NotImplementedError(f"{what}: {msg}")
        raise NativeError(f"{what} failed (code {rc}): {msg}")


def stream_ptr(device=None) -> int:
    return torch.cuda.current_stream(device).cuda_stream


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def device_guard(device):
    """Context manager that makes `device` the current HIP device for the native calls inside it.  The C ABI takes
    raw pointers and a stream; kernel launches, hipFuncSetAttribute and the per-device CU count all go to the
    CURRENT device, so a model on cuda:1 while cuda:0 is current must switch first.  Free when already current."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if torch.cuda.current_device() == idx:
        return _NO_GUARD
    return torch.cuda.device(idx)


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.bfloat16:
        return RAJNI_BF16
    if dt == torch.float32:
        return RAJNI_F32
    raise NotImplementedError(f"rajni_amd: dtype {dt} is not supported (bfloat16 is the built compute type)")


def require_device(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise NativeError(
            f"{name} is on {t.device}: rajni_amd runs on a ROCm (MI355X / gfx950) device only and has "
            "no CPU fallback. Move the model and inputs to 'cuda'.")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


# ---- measurement hooks -------------------------------------------------------------------------

def profile_enable(mask: int) -> None:
    lib().rajni_profile_enable(mask)


def profile_reset() -> None:
    lib().rajni_profile_reset()


def profile_collect():
    """{kernel class name: dict(launches, ms, flops, bytes)} accumulated since the last reset."""
    n = NUM_KCLASS
    la = (C.c_longlong * n)()
    ms = (C.c_double * n)()
    fl = (C.c_double * n)()
    by = (C.c_double * n)()
    check(lib().rajni_profile_collect(la, ms, fl, by), "rajni_profile_collect")
    out = {}
    for i in range(n):
        if la[i]:
            out[lib().rajni_profile_class_name(i).decode()] = dict(
                kclass=i, launches=int(la[i]), ms=float(ms[i]), flops=float(fl[i]), bytes=float(by[i]))
    return out
