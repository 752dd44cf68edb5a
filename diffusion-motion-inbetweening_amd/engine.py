"""Python handle on the native engine (libcondmdi_hip.so).

`Engine` owns one ``cmdi_handle``: the packed weights of an MDM ``trans_enc`` denoiser, the
respaced diffusion schedule and the per-call conditioning, all resident in HBM.  Tensors cross the
boundary as raw device pointers on torch's current HIP stream; no call synchronises.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _native as N


def _as_f32_ptr(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _require_device_f32(t: torch.Tensor, name: str, device) -> torch.Tensor:
    if t.device != device:
        t = t.to(device)
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


class Engine:
    """One native engine bound to one HIP device."""

    def __init__(self, *, n_layers: int, d_model: int, d_ff: int, n_heads: int, n_feats: int,
                 max_frames: int, max_batch: int, pe_rows: int = 5000, text_cond: bool = False,
                 want_grad: bool = False, precision: Optional[str] = None, arch: str = "trans_enc",
                 unet_added: int = 0, unet_mults=(2, 2, 2, 2), unet_attention: bool = False, device="cuda"):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise N.NativeError(
                "the CondMDI engine runs on a HIP device only (there is no CPU path); got "
                f"device={self.device}")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.lib = N.load()
        self.desc = N.ModelDesc(n_layers, d_model, d_ff, n_heads, n_feats, max_frames, max_batch,
                                pe_rows, int(text_cond), int(want_grad), N.PRECISIONS[precision],
                                N.CMDI_ARCH_UNET if arch == "unet" else N.CMDI_ARCH_TRANS_ENC, int(unet_added),
                                (C.c_int32 * 4)(*[int(m) for m in unet_mults]), int(unet_attention))
        self.arch = arch
        self.n_feats, self.max_frames, self.max_batch = n_feats, max_frames, max_batch
        self.want_grad = bool(want_grad)
        self.text_cond = bool(text_cond)
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            N.check(self.lib.cmdi_create(C.byref(self.desc), C.byref(self._h)))
        # "f32" (fp32 MFMA), "bf16x6" (exact three-plane bf16 operands, six MFMA products) or "f16x3" (22-bit split-f16
        # operands, three products); None = library default
        self.precision = N.PRECISION_NAMES.get(self.lib.cmdi_precision(self._h), "f32") \
            if (n_layers > 0 or arch == "unet") else "f32"
        self.n_steps = 0
        self.batch = 0
        self.n_frames = 0
        self.cfg = False
        self._schedule_key = None
        self._keep = []  # host arrays that must outlive a native call

    # -- lifetime -------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            try:
                torch.cuda.synchronize(self.device)
            except Exception:
                pass
            self.lib.cmdi_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def stream(self) -> int:
        return N.current_stream(self.device)

    def workspace_bytes(self) -> int:
        return int(self.lib.cmdi_workspace_bytes(self._h))

    def pipeline_parts(self) -> int:
        """How many independent batch pipelines sample_loop runs for the current condition."""
        return int(self.lib.cmdi_pipeline_parts(self._h))

    def check_range(self):
        """Raise if the device status flag was set since the last check: an activation left the f16 range
        (f16x3), or a timestep fell outside the time-embedding table (one 4-byte read-back, synchronises the
        stream — call once per sampling chain)."""
        flag = C.c_int32(0)
        with torch.cuda.device(self.device):
            N.check(self.lib.cmdi_range_status(self._h, C.byref(flag), self.stream))
        if flag.value & 2:
            raise IndexError("a timestep outside the time-embedding table reached the denoiser "
                             "(the reference raises at pe[timesteps], model/mdm.py:352)")
        if flag.value & 1 and self.precision == "f16x3":
            what = "an activation left the f16 range (|x| >= 65504 or non-finite) in the split-f16 GEMM path: results are invalid; "
            if self.arch == "unet" and self.desc.unet_attention:
                raise N.RangeError(what + "MDM_UNET(attention=True) is built for f16x3 only (no bf16x6 mode to fall back to): "
                                          "check the checkpoint and the normalisation of the inputs")
            raise N.RangeError(what + "re-run with precision='bf16x6' (CMDI_PRECISION=bf16x6)")

    def clear_range(self):
        """Drop a pending status flag without reading it (start of a sampling chain: report this chain's events only)."""
        with torch.cuda.device(self.device):
            N.check(self.lib.cmdi_range_clear(self._h, self.stream))

    def set_graph(self, on: bool):
        """hipGraph replay of whole denoising steps in sample_loop (bitwise identical results)."""
        N.check(self.lib.cmdi_set_graph(self._h, int(on)))

    def profile_enable(self, on: bool):
        N.check(self.lib.cmdi_profile_enable(self._h, int(on)))

    def profile_select(self, which: int):
        """0 = time the in_proj GEMM launches (default), 1 = the attention kernel launches."""
        N.check(self.lib.cmdi_profile_select(self._h, int(which)))

    def profile_read(self):
        """(total_ms, launches, (M, N, K)) of the in_proj GEMM launches recorded since enable."""
        ms, cnt = C.c_double(), C.c_int64()
        m, n, k = C.c_int32(), C.c_int32(), C.c_int32()
        N.check(self.lib.cmdi_profile_read(self._h, C.byref(ms), C.byref(cnt), C.byref(m),
                                           C.byref(n), C.byref(k)))
        return ms.value, cnt.value, (m.value, n.value, k.value)

    def profile_kernel(self) -> str:
        """Kernel family the profiled launches dispatched to (e.g. 'gemm_h3_kernel', 'gemm_h3p_kernel')."""
        return (self.lib.cmdi_profile_kernel(self._h) or b"").decode()

    # -- weights --------------------------------------------------------------------------------
    def load_state_dict(self, state_dict, n_time_rows: int = 1000):
        """Ingest the reference's MDM state dict (SURVEY.md §5.4 key names)."""
        with torch.cuda.device(self.device):
            for name, t in state_dict.items():
                if name.startswith("clip_model."):
                    continue
                t = _require_device_f32(t.detach(), name, self.device)
                N.check(self.lib.cmdi_load_weight(self._h, name.encode(), N.ptr(t), t.numel(),
                                                  self.stream))
            N.check(self.lib.cmdi_finalize_weights(self._h, int(n_time_rows), self.stream))

    # -- schedule -------------------------------------------------------------------------------
    def set_schedule(self, tables: dict, key=None):
        """`tables`: fp32 numpy arrays named like include/condmdi.h's cmdi_schedule fields."""
        if key is not None and key == self._schedule_key:
            return
        n = int(tables["n_steps"])
        arrs = {}
        for name in ("post_coef1", "post_coef2", "sigma", "sqrt_ab", "sqrt_1mab", "sqrt_recip_ab",
                     "sqrt_recipm1_ab", "ab", "ab_prev"):
            a = np.ascontiguousarray(tables[name], dtype=np.float32)
            assert a.shape == (n,), (name, a.shape)
            arrs[name] = a
        tmap = np.ascontiguousarray(tables["timestep_map"], dtype=np.int64)
        assert tmap.shape == (n,)
        sc = N.Schedule(n, int(tables["mean_type"]), *[_as_f32_ptr(arrs[k]) for k in (
            "post_coef1", "post_coef2", "sigma", "sqrt_ab", "sqrt_1mab", "sqrt_recip_ab",
            "sqrt_recipm1_ab", "ab", "ab_prev")], tmap.ctypes.data_as(C.POINTER(C.c_int64)),
                        float(tables.get("clip_x0", 0.0)))
        N.check(self.lib.cmdi_set_schedule(self._h, C.byref(sc)))
        self.n_steps = n
        self._schedule_key = key

    # -- condition ------------------------------------------------------------------------------
    def set_condition(self, *, batch: int, n_frames: int, cfg: bool = False,
                      enc_text: Optional[torch.Tensor] = None,
                      text_scale: Optional[torch.Tensor] = None,
                      inpaint_mask: Optional[torch.Tensor] = None,
                      inpaint_motion: Optional[torch.Tensor] = None,
                      imputate: bool = False, stop_imputation_at: int = 0,
                      recon_guidance: bool = False, stop_recguidance_at: int = 0,
                      recon_w: Optional[np.ndarray] = None,
                      obs_x0: Optional[torch.Tensor] = None, obs_mask: Optional[torch.Tensor] = None):
        dev = self.device
        keep = []
        if enc_text is not None:
            enc_text = _require_device_f32(enc_text, "enc_text", dev)
            assert enc_text.shape == (batch, 512), enc_text.shape
            keep.append(enc_text)
        if text_scale is not None:
            text_scale = _require_device_f32(torch.as_tensor(text_scale).reshape(-1), "text_scale", dev)
            assert text_scale.numel() == batch
            keep.append(text_scale)
        if inpaint_mask is not None:
            inpaint_mask = inpaint_mask.to(dev).to(torch.uint8).contiguous()
            assert inpaint_mask.numel() == batch * self.n_feats * n_frames, inpaint_mask.shape
            keep.append(inpaint_mask)
        if inpaint_motion is not None:
            inpaint_motion = _require_device_f32(inpaint_motion, "inpaint_motion", dev)
            assert inpaint_motion.numel() == batch * self.n_feats * n_frames
            keep.append(inpaint_motion)
        if obs_x0 is not None:
            obs_x0 = _require_device_f32(obs_x0, "obs_x0", dev)
            assert obs_x0.numel() == batch * self.n_feats * n_frames, obs_x0.shape
            keep.append(obs_x0)
        if obs_mask is not None:
            obs_mask = obs_mask.to(dev).to(torch.uint8).contiguous()
            assert obs_mask.numel() == batch * self.n_feats * n_frames, obs_mask.shape
            keep.append(obs_mask)
        rw = None
        if recon_w is not None:
            rw = np.ascontiguousarray(recon_w, dtype=np.float32)
            assert rw.shape == (self.n_steps,), (rw.shape, self.n_steps)
        cond = N.Condition(batch, n_frames, int(cfg), N.ptr(enc_text), N.ptr(text_scale),
                           N.ptr(inpaint_mask), N.ptr(inpaint_motion), int(imputate),
                           int(stop_imputation_at), int(recon_guidance), int(stop_recguidance_at),
                           _as_f32_ptr(rw) if rw is not None else None, N.ptr(obs_x0), N.ptr(obs_mask))
        with torch.cuda.device(dev):
            N.check(self.lib.cmdi_set_condition(self._h, C.byref(cond), self.stream))
        self.batch, self.n_frames, self.cfg = batch, n_frames, bool(cfg)
        self._keep = keep  # inputs are copied on the stream: keep them alive until it drains

    # -- denoiser -------------------------------------------------------------------------------
    def _check_x(self, x: torch.Tensor):
        if x.device != self.device or x.dtype != torch.float32 or not x.is_contiguous():
            raise ValueError("x must be a contiguous fp32 tensor on the engine's device")
        if x.numel() != self.batch * self.n_feats * self.n_frames:
            raise ValueError(f"x has {x.numel()} elements, condition was set for "
                             f"[{self.batch},{self.n_feats},1,{self.n_frames}]")

    def mdm_forward(self, x: torch.Tensor, t: torch.Tensor, split: bool = False):
        """Denoiser output for x [B,J,1,T] at ORIGINAL timesteps t [B] (int64)."""
        self._check_x(x)
        t = t.to(device=self.device, dtype=torch.int64).contiguous()
        assert t.numel() == self.batch
        out = torch.empty_like(x)
        out_u = torch.empty_like(x) if split else None
        with torch.cuda.device(self.device):
            N.check(self.lib.cmdi_mdm_forward(self._h, N.ptr(x), N.ptr(t), N.ptr(out), N.ptr(out_u),
                                              self.stream))
        return (out, out_u) if split else out

    def mdm_vjp(self, gout: torch.Tensor) -> torch.Tensor:
        self._check_x(gout)
        gx = torch.empty_like(gout)
        with torch.cuda.device(self.device):
            N.check(self.lib.cmdi_mdm_vjp(self._h, N.ptr(gout), N.ptr(gx), self.stream))
        return gx

    # -- sampler --------------------------------------------------------------------------------
    def step(self, x: torch.Tensor, step: int, *, sampler: int = N.CMDI_SAMPLER_DDPM,
             eta: float = 0.0, noise: Optional[torch.Tensor] = None,
             pred_xstart: Optional[torch.Tensor] = None, seed: int = 0, first_sample: int = 0):
        """In place: x_t -> x_{t-1} at respaced index `step`."""
        self._check_x(x)
        if noise is not None:
            self._check_x(noise)
        if pred_xstart is not None:
            self._check_x(pred_xstart)
        with torch.cuda.device(self.device):
            N.check(self.lib.cmdi_step(self._h, sampler, int(step), float(eta), N.ptr(x),
                                       N.ptr(pred_xstart), N.ptr(noise), int(seed) & (2**64 - 1),
                                       int(first_sample), self.stream))
        return x

    def sample_loop(self, x: torch.Tensor, first_step: int, last_step: int = 0, *,
                    sampler: int = N.CMDI_SAMPLER_DDPM, eta: float = 0.0,
                    noise_stream: Optional[torch.Tensor] = None, seed: int = 0,
                    first_sample: int = 0):
        self._check_x(x)
        if noise_stream is not None:
            n_draws = first_step - last_step + 1
            if (noise_stream.device != self.device or noise_stream.dtype != torch.float32
                    or not noise_stream.is_contiguous() or noise_stream.numel() != n_draws * x.numel()):
                raise ValueError("noise_stream must be contiguous fp32 [n_draws, *x.shape] on device")
        with torch.cuda.device(self.device):
            N.check(self.lib.cmdi_sample_loop(self._h, sampler, int(first_step), int(last_step),
                                              float(eta), N.ptr(x), N.ptr(noise_stream),
                                              int(seed) & (2**64 - 1), int(first_sample), self.stream))
        return x

    def sampler_update(self, x: torch.Tensor, model_out: torch.Tensor, step: int, *,
                       sampler: int = N.CMDI_SAMPLER_DDPM, eta: float = 0.0,
                       recon_grad: Optional[torch.Tensor] = None,
                       noise: Optional[torch.Tensor] = None,
                       pred_xstart: Optional[torch.Tensor] = None, seed: int = 0,
                       first_sample: int = 0):
        self._check_x(x)
        self._check_x(model_out)
        with torch.cuda.device(self.device):
            N.check(self.lib.cmdi_sampler_update(
                self._h, sampler, int(step), float(eta), N.ptr(model_out), N.ptr(recon_grad),
                N.ptr(x), N.ptr(pred_xstart), N.ptr(noise), int(seed) & (2**64 - 1),
                int(first_sample), self.stream))
        return x

    def q_sample(self, x0: torch.Tensor, noise: torch.Tensor, step: int) -> torch.Tensor:
        out = torch.empty_like(x0)
        with torch.cuda.device(self.device):
            N.check(self.lib.cmdi_q_sample(self._h, int(step), N.ptr(x0.contiguous()),
                                           N.ptr(noise.contiguous()), N.ptr(out), x0.numel(),
                                           self.stream))
        return out

    def randn(self, shape, *, seed: int, first_sample: int = 0, step: int = -1) -> torch.Tensor:
        out = torch.empty(tuple(shape), dtype=torch.float32, device=self.device)
        batch = int(shape[0])
        per = out.numel() // batch
        with torch.cuda.device(self.device):
            N.check(self.lib.cmdi_randn(self._h, N.ptr(out), batch, per, int(seed) & (2**64 - 1),
                                        int(first_sample), int(step), self.stream))
        return out


def gemm_nt(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None,
            tile: int = 0, epi: int = 0, resid: Optional[torch.Tensor] = None,
            out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """C = epi(A[M,K] · W[N,K]ᵀ + bias) through the engine's fp32 MFMA GEMM (test / bench hook);
    epi 0 = bias, 1 = bias + GELU, 3 = bias + residual."""
    lib = N.load()
    assert a.is_cuda and w.is_cuda and a.dtype == torch.float32 and w.dtype == torch.float32
    a, w = a.contiguous(), w.contiguous()
    m, k = a.shape
    n = w.shape[0]
    assert w.shape[1] == k
    c = out if out is not None else torch.empty((m, n), dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device):
        N.check(lib.cmdi_gemm_nt(N.ptr(a), N.ptr(w), N.ptr(bias), N.ptr(resid), N.ptr(c), m, n, k,
                                 int(epi), int(tile), N.current_stream(a.device)))
    return c


def pack_x6(w: torch.Tensor) -> torch.Tensor:
    """fp32 [rows, cols] -> three bf16 planes [rows, cols/32, 3, 32] with w = p0 + p1 + p2 exactly (test hook)."""
    lib = N.load()
    assert w.is_cuda and w.dtype == torch.float32 and w.dim() == 2 and w.shape[1] % 32 == 0
    w = w.contiguous()
    out = torch.empty((w.shape[0], w.shape[1] // 32, 3, 32), dtype=torch.bfloat16, device=w.device)
    with torch.cuda.device(w.device):
        N.check(lib.cmdi_pack_x6(N.ptr(w), N.ptr(out), w.shape[0], w.shape[1], N.current_stream(w.device)))
    return out


def gemm_x6(a: torch.Tensor, w_packed: torch.Tensor, bias: Optional[torch.Tensor] = None, epi: int = 0,
            resid: Optional[torch.Tensor] = None, variant: int = 1, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """C = epi(A[M,K] · Wᵀ + bias) on the bf16 pipe with exact three-plane operands (test / bench hook);
    w_packed = pack_x6(W[N,K]); epi 0 = bias, 1 = bias + GELU, 3 = bias + residual."""
    lib = N.load()
    assert a.is_cuda and a.dtype == torch.float32 and w_packed.dtype == torch.bfloat16
    a = a.contiguous()
    m, k = a.shape
    n = w_packed.shape[0]
    assert w_packed.shape[1] * 32 == k
    c = out if out is not None else torch.empty((m, n), dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device):
        N.check(lib.cmdi_gemm_x6(N.ptr(a), N.ptr(w_packed), N.ptr(bias), N.ptr(resid), N.ptr(c), m, n, k, int(epi),
                                 int(variant), N.current_stream(a.device)))
    return c


def split_f16(x: torch.Tensor) -> torch.Tensor:
    """fp32 [rows, cols] -> split rows [rows, 2*cols] float16; per 32-column chunk 32 hi values then
    32 values of (x - hi) * 2^11 (test hook)."""
    lib = N.load()
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2
    x = x.contiguous()
    out = torch.empty((x.shape[0], 2 * x.shape[1]), dtype=torch.float16, device=x.device)
    with torch.cuda.device(x.device):
        N.check(lib.cmdi_split_f16(N.ptr(x), N.ptr(out), x.shape[0], x.shape[1],
                                   N.current_stream(x.device)))
    return out


def gemm_h3(a_split: torch.Tensor, w_split: torch.Tensor, bias: Optional[torch.Tensor] = None,
            tile: int = 0, epi: int = 0, resid: Optional[torch.Tensor] = None,
            split_out: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """C = epi(A · Wᵀ + bias) on the split-f16 path (test / bench hook); operands from split_f16().
    epi 0 = bias, 1 = bias + GELU (always split output), 3 = bias + fp32 residual, 4 = bias + residual given
    as split rows."""
    lib = N.load()
    assert a_split.dtype == torch.float16 and w_split.dtype == torch.float16
    m, k2 = a_split.shape
    n = w_split.shape[0]
    assert w_split.shape[1] == k2 and k2 % 2 == 0
    k = k2 // 2
    split_out = split_out or epi == 1
    if out is not None:
        c = out
    elif split_out:
        c = torch.empty((m, 2 * n), dtype=torch.float16, device=a_split.device)
    else:
        c = torch.empty((m, n), dtype=torch.float32, device=a_split.device)
    with torch.cuda.device(a_split.device):
        N.check(lib.cmdi_gemm_h3(N.ptr(a_split), N.ptr(w_split), N.ptr(bias), N.ptr(resid),
                                 0 if split_out else N.ptr(c), N.ptr(c) if split_out else 0, m, n, k,
                                 int(epi), int(tile), N.current_stream(a_split.device)))
    return c


def gemm_h3_ln(a_split, w_split, bias, resid, gamma, beta, want_split: bool = False):
    """LayerNorm((A · Wᵀ + bias) + resid) with the normalisation fused into the GEMM epilogue (test hook)."""
    lib = N.load()
    m, k2 = a_split.shape
    n = w_split.shape[0]
    y = torch.empty((m, n), dtype=torch.float32, device=a_split.device)
    ys = torch.empty((m, 2 * n), dtype=torch.float16, device=a_split.device) if want_split else None
    with torch.cuda.device(a_split.device):
        N.check(lib.cmdi_gemm_h3_ln(N.ptr(a_split), N.ptr(w_split), N.ptr(bias), N.ptr(resid),
                                    N.ptr(gamma), N.ptr(beta), N.ptr(y), N.ptr(ys), m, n, k2 // 2,
                                    N.current_stream(a_split.device)))
    return (y, ys) if want_split else y


def unsplit_f16(s: torch.Tensor) -> torch.Tensor:
    """Inverse of split_f16 (exact in float64, returned as fp32)."""
    c = s.view(s.shape[0], -1, 2, 32).double()
    return (c[:, :, 0] + c[:, :, 1] / 2048.0).reshape(s.shape[0], -1).float()


def attention_fwd(qkv: torch.Tensor, n_seq: int, seq_len: int, n_heads: int) -> torch.Tensor:
    """softmax(QKᵀ/sqrt(128))V per (sequence, head) on a packed [n_seq*S, 3*H*128] tensor."""
    lib = N.load()
    assert qkv.is_cuda and qkv.dtype == torch.float32 and qkv.is_contiguous()
    assert qkv.shape == (n_seq * seq_len, 3 * n_heads * 128)
    out = torch.empty((n_seq * seq_len, n_heads * 128), dtype=torch.float32, device=qkv.device)
    with torch.cuda.device(qkv.device):
        N.check(lib.cmdi_attention_fwd(N.ptr(qkv), N.ptr(out), n_seq, seq_len, n_heads,
                                       N.current_stream(qkv.device)))
    return out


def attention_fwd_h3(qkv: torch.Tensor, n_seq: int, seq_len: int, n_heads: int) -> torch.Tensor:
    """attention_fwd on the split-f16 path: qkv (fp32) is split on the device first (test hook)."""
    lib = N.load()
    assert qkv.shape == (n_seq * seq_len, 3 * n_heads * 128)
    qs = split_f16(qkv)
    out = torch.empty((n_seq * seq_len, n_heads * 128), dtype=torch.float32, device=qkv.device)
    with torch.cuda.device(qkv.device):
        N.check(lib.cmdi_attention_fwd_h3(N.ptr(qs), N.ptr(out), n_seq, seq_len, n_heads,
                                          N.current_stream(qkv.device)))
    return out


def attention_fwd_h3_split(qkv_split: torch.Tensor, n_seq: int, seq_len: int, n_heads: int) -> torch.Tensor:
    """attention_fwd_h3 on rows that are ALREADY split ([>= n_seq*S, 6*H*128] f16; rows past n_seq*S are not part of the
    launch — the test that poisons the memory behind the tensor uses them)."""
    lib = N.load()
    assert qkv_split.dtype == torch.float16 and qkv_split.is_contiguous() and qkv_split.shape[1] == 6 * n_heads * 128
    assert qkv_split.shape[0] >= n_seq * seq_len
    out = torch.empty((n_seq * seq_len, n_heads * 128), dtype=torch.float32, device=qkv_split.device)
    with torch.cuda.device(qkv_split.device):
        N.check(lib.cmdi_attention_fwd_h3(N.ptr(qkv_split), N.ptr(out), n_seq, seq_len, n_heads,
                                          N.current_stream(qkv_split.device)))
    return out


def attention_vjp_h3(qkv: torch.Tensor, dout: torch.Tensor, n_seq: int, seq_len: int, n_heads: int) -> torch.Tensor:
    """d qkv [n_seq*S, 3*H*128] (fp32, un-split) of the split-f16 attention core for the output gradient `dout` (test hook)."""
    lib = N.load()
    d = n_heads * 128
    M = n_seq * seq_len
    assert qkv.shape == (M, 3 * d) and dout.shape == (M, d) and dout.dtype == torch.float32 and dout.is_contiguous()
    qs = split_f16(qkv)
    dqs = torch.zeros((M, 6 * d), dtype=torch.float16, device=qkv.device)
    work = torch.empty(2 * M * d + n_seq * n_heads * (2 * seq_len + 96 * ((seq_len + 31) // 32)) + 4, dtype=torch.float32,
                       device=qkv.device)
    with torch.cuda.device(qkv.device):
        N.check(lib.cmdi_attention_vjp_h3(N.ptr(qs), N.ptr(dout), N.ptr(dqs), N.ptr(work), n_seq, seq_len, n_heads,
                                          N.current_stream(qkv.device)))
    return unsplit_f16(dqs)


def conv_weight_k_order(w: torch.Tensor, taps: int) -> torch.Tensor:
    """[n, taps * cin] tap-major GEMM weights of a convolution (column tap * cin + c) -> the chunk-major K order the
    convolution GEMMs walk (column (chunk * taps + tap) * 32 + c32, cin % 32 == 0; include/condmdi.h cmdi_conv_rows_h3)."""
    n, k = w.shape
    cin = k // taps
    assert cin * taps == k and cin % 32 == 0
    return w.reshape(n, taps, cin // 32, 32).permute(0, 2, 1, 3).reshape(n, k).contiguous()


def philox4x32_10(counter, key):
    """Host Philox4x32-10 block (known-answer tests)."""
    lib = N.load()
    c = (C.c_uint32 * 4)(*[int(v) & 0xFFFFFFFF for v in counter])
    k = (C.c_uint32 * 2)(*[int(v) & 0xFFFFFFFF for v in key])
    o = (C.c_uint32 * 4)()
    lib.cmdi_philox4x32_10(c, k, o)
    return [int(v) for v in o]
