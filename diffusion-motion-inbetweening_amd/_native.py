"""ctypes binding of libcondmdi_hip.so (C-ABI declared in include/condmdi.h).

There is NO fallback: if the HIP library is missing or fails to load, every entry point raises.
``torch`` is imported first on purpose — PyTorch-ROCm ships its own ``libamdhip64.so`` (SONAME
``libamdhip64.so.7``); loading ours afterwards makes the dynamic linker bind our library to that
already-loaded runtime, so torch's streams / device pointers and our kernels share one HIP context.
"""
from __future__ import annotations

import ctypes
import ctypes as C
from pathlib import Path

import torch  # noqa: F401  (must precede the CDLL below, see module docstring)

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "csrc" / "libcondmdi_hip.so"
if __import__("os").environ.get("CMDI_PROBES_LIB") == "1":   # tools/ only: the instrumented build (build.py --probes)
    LIB_PATH = _PKG / "csrc" / "libcondmdi_hip_probes.so"
if __import__("os").environ.get("CMDI_LIB_VARIANT"):        # tools/ only: experiment builds (tools/variant_build.sh, tools/ab_variants.sh)
    _variant = __import__("os").environ["CMDI_LIB_VARIANT"]
    if not __import__("re").fullmatch(r"[A-Za-z0-9_]+", _variant):   # a name, never a path
        raise ValueError(f"CMDI_LIB_VARIANT must match [A-Za-z0-9_]+, got {_variant!r}")
    LIB_PATH = _PKG / "csrc" / ("libcondmdi_hip_" + _variant + ".so")

CMDI_MEAN_START_X, CMDI_MEAN_EPSILON = 0, 1
CMDI_SAMPLER_DDPM, CMDI_SAMPLER_DDIM = 0, 1
CMDI_PREC_DEFAULT, CMDI_PREC_F32, CMDI_PREC_F16X3, CMDI_PREC_BF16X6 = 0, 1, 2, 3
CMDI_ARCH_TRANS_ENC, CMDI_ARCH_UNET = 0, 1
PRECISIONS = {None: CMDI_PREC_DEFAULT, "default": CMDI_PREC_DEFAULT, "f32": CMDI_PREC_F32,
              "f16x3": CMDI_PREC_F16X3, "bf16x6": CMDI_PREC_BF16X6}
PRECISION_NAMES = {CMDI_PREC_F32: "f32", CMDI_PREC_F16X3: "f16x3", CMDI_PREC_BF16X6: "bf16x6"}


class NativeError(RuntimeError):
    """A libcondmdi_hip.so call returned a negative status."""


class RangeError(NativeError):
    """f16x3 only: a weight or an activation left the f16 range (|x| >= 65504).  The samplers answer by re-running the
    chain on a bf16x6 engine (exact three-plane operands, fp32's exponent range) — never on a CPU."""


CMDI_E_RANGE = -6


class ModelDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "n_layers", "d_model", "d_ff", "n_heads", "n_feats", "max_frames", "max_batch", "pe_rows",
        "text_cond", "want_grad", "precision", "arch", "unet_added")] + [("unet_mults", C.c_int32 * 4),
                                                                            ("unet_attention", C.c_int32)]


class ClipDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("vocab_size", "width", "heads", "layers", "context", "embed_dim", "max_batch")]


class Schedule(C.Structure):
    _fields_ = [("n_steps", C.c_int32), ("mean_type", C.c_int32)] + [
        (n, C.POINTER(C.c_float)) for n in (
            "post_coef1", "post_coef2", "sigma", "sqrt_ab", "sqrt_1mab", "sqrt_recip_ab",
            "sqrt_recipm1_ab", "ab", "ab_prev")] + [("timestep_map", C.POINTER(C.c_int64)), ("clip_x0", C.c_float)]


class Condition(C.Structure):
    _fields_ = [
        ("batch", C.c_int32), ("n_frames", C.c_int32), ("cfg", C.c_int32),
        ("d_enc_text", C.c_void_p), ("d_text_scale", C.c_void_p),
        ("d_inpaint_mask", C.c_void_p), ("d_inpaint_motion", C.c_void_p),
        ("imputate", C.c_int32), ("stop_imputation_at", C.c_int32),
        ("recon_guidance", C.c_int32), ("stop_recguidance_at", C.c_int32),
        ("recon_w", C.POINTER(C.c_float)),
        ("d_obs_x0", C.c_void_p), ("d_obs_mask", C.c_void_p),
    ]


# name -> (restype, argtypes); mirrors include/condmdi.h one-to-one
_VP, _I32, _I64, _U64, _F = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_float
SIGNATURES = {
    "cmdi_create": (C.c_int, [C.POINTER(ModelDesc), C.POINTER(_VP)]),
    "cmdi_destroy": (C.c_int, [_VP]),
    "cmdi_last_error": (C.c_char_p, []),
    "cmdi_version": (C.c_char_p, []),
    "cmdi_load_weight": (C.c_int, [_VP, C.c_char_p, _VP, _I64, _VP]),
    "cmdi_finalize_weights": (C.c_int, [_VP, _I32, _VP]),
    "cmdi_set_schedule": (C.c_int, [_VP, C.POINTER(Schedule)]),
    "cmdi_set_condition": (C.c_int, [_VP, C.POINTER(Condition), _VP]),
    "cmdi_mdm_forward": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP]),
    "cmdi_mdm_vjp": (C.c_int, [_VP, _VP, _VP, _VP]),
    "cmdi_step": (C.c_int, [_VP, _I32, _I32, _F, _VP, _VP, _VP, _U64, _I64, _VP]),
    "cmdi_sample_loop": (C.c_int, [_VP, _I32, _I32, _I32, _F, _VP, _VP, _U64, _I64, _VP]),
    "cmdi_set_graph": (C.c_int, [_VP, _I32]),
    "cmdi_sampler_update": (C.c_int, [_VP, _I32, _I32, _F, _VP, _VP, _VP, _VP, _VP, _U64, _I64, _VP]),
    "cmdi_q_sample": (C.c_int, [_VP, _I32, _VP, _VP, _VP, _I64, _VP]),
    "cmdi_randn": (C.c_int, [_VP, _VP, _I32, _I64, _U64, _I64, _I32, _VP]),
    "cmdi_recover_xyz": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _I32, _VP]),
    "cmdi_gemm_nt": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _I32, _VP]),
    "cmdi_attention_fwd": (C.c_int, [_VP, _VP, _I32, _I32, _I32, _VP]),
    "cmdi_conv_rows_h3": (C.c_int, [_VP, _I32, _VP, _VP, _VP, _VP, _VP] + [_I32] * 12 + [_VP]),
    "cmdi_conv_rows_x6": (C.c_int, [_VP, _I32, _VP, _VP, _VP, _VP, _VP, _I32] + [_I32] * 12 + [_VP]),
    "cmdi_gemm_h3_ln": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I32, _I32, _I32, _VP]),
    "cmdi_attention_fwd_h3": (C.c_int, [_VP, _VP, _I32, _I32, _I32, _VP]),
    "cmdi_attention_vjp_h3": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _I32, _VP]),
    "cmdi_precision": (C.c_int, [_VP]),
    "cmdi_range_status": (C.c_int, [_VP, C.POINTER(_I32), _VP]),
    "cmdi_range_clear": (C.c_int, [_VP, _VP]),
    "cmdi_split_f16": (C.c_int, [_VP, _VP, _I64, _I32, _VP]),
    "cmdi_gemm_h3": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _I32, _VP]),
    "cmdi_clip_create": (C.c_int, [C.POINTER(ClipDesc), C.POINTER(_VP)]),
    "cmdi_clip_destroy": (C.c_int, [_VP]),
    "cmdi_clip_load_weight": (C.c_int, [_VP, C.c_char_p, _VP, _I64, _VP]),
    "cmdi_clip_encode_text": (C.c_int, [_VP, _VP, _I32, _VP, _VP]),
    "cmdi_pack_x6": (C.c_int, [_VP, _VP, _I64, _I32, _VP]),
    "cmdi_gemm_x6": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _I32, _VP]),
    "cmdi_philox4x32_10": (None, [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "cmdi_workspace_bytes": (_I64, [_VP]),
    "cmdi_pipeline_parts": (C.c_int, [_VP]),
    "cmdi_profile_enable": (C.c_int, [_VP, _I32]),
    "cmdi_profile_select": (C.c_int, [_VP, _I32]),
    "cmdi_profile_read": (C.c_int, [_VP, C.POINTER(C.c_double), C.POINTER(_I64), C.POINTER(_I32),
                                    C.POINTER(_I32), C.POINTER(_I32)]),
    "cmdi_profile_kernel": (C.c_char_p, [_VP]),
}

_lib = None


def load() -> ctypes.CDLL:
    """Load libcondmdi_hip.so (built in-tree by build.py); raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise NativeError(
            f"{LIB_PATH} is missing: the CondMDI sampling path has no CPU / PyTorch fallback. "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc).")
    lib = ctypes.CDLL(str(LIB_PATH), mode=ctypes.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library drift
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        msg = load().cmdi_last_error()
        kind = RangeError if rc == CMDI_E_RANGE else NativeError
        raise kind(f"libcondmdi_hip: error {rc}: {msg.decode() if msg else '?'}")


def ptr(t) -> int:
    """Device pointer of a torch tensor (None -> NULL)."""
    return 0 if t is None else t.data_ptr()


def current_stream(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream
