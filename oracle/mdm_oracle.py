"""numpy fp32 restatement of the MDM ``trans_enc`` denoiser and of its input-VJP.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Follows, line by line:
    token assembly   reference model/mdm.py:244-280   (embed_timestep :351-353, embed_text +
                     mask_cond :188-198,248-251, InputProcess :366-372, PositionalEncoding :332-335)
    encoder layers   torch nn.TransformerEncoderLayer post-norm math,
                     constructed at model/mdm.py:107-114, called at :284 with NO padding mask
    output           model/mdm.py:284 ([1:]), OutputProcess :409-423
    CFG              model/cfg_sampler.py:25-35
    VJP              what torch.autograd.grad(loss, z) computes at
                     diffusion/gaussian_diffusion.py:411-416 (hand-derived backward of the above)
Pinned by tests/golden (outputs of the real reference run on CPU, see tests/golden/make_golden.py).
"""
from __future__ import annotations

import numpy as np
from scipy.special import erf

F32 = np.float32
LN_EPS = F32(1e-5)


def _linear(x, w, b=None):
    y = x @ w.T
    return y if b is None else y + b


def _gelu(x):
    return (F32(0.5) * x * (F32(1.0) + erf(x * F32(0.7071067811865476)).astype(F32))).astype(F32)


def _gelu_grad(x):
    cdf = F32(0.5) * (F32(1.0) + erf(x * F32(0.7071067811865476)).astype(F32))
    pdf = F32(0.3989422804014327) * np.exp(F32(-0.5) * x * x)
    return (cdf + x * pdf).astype(F32)


def _silu(x):
    return (x / (F32(1.0) + np.exp(-x))).astype(F32)


def _layernorm(x, g, b):
    mean = x.mean(axis=-1, keepdims=True, dtype=F32)
    xc = x - mean
    var = (xc * xc).mean(axis=-1, keepdims=True, dtype=F32)
    rstd = F32(1.0) / np.sqrt(var + LN_EPS)
    xhat = xc * rstd
    return (xhat * g + b).astype(F32), xhat.astype(F32), rstd.astype(F32)


def _layernorm_bwd(dy, xhat, rstd, g):
    gg = dy * g
    m1 = gg.mean(axis=-1, keepdims=True, dtype=F32)
    m2 = (gg * xhat).mean(axis=-1, keepdims=True, dtype=F32)
    return (rstd * (gg - m1 - xhat * m2)).astype(F32)


def _softmax(s):
    m = s.max(axis=-1, keepdims=True)
    e = np.exp(s - m)
    return (e / e.sum(axis=-1, keepdims=True, dtype=F32)).astype(F32)


class MDMOracle:
    """sd: dict name -> numpy fp32 array (reference state-dict names)."""

    def __init__(self, sd: dict, n_heads: int = 4):
        self.sd = {k: np.asarray(v, dtype=F32) for k, v in sd.items()}
        self.H = n_heads
        self.d = self.sd["input_process.poseEmbedding.weight"].shape[0]
        self.L = 1 + max(int(k.split(".")[2]) for k in self.sd if k.startswith("seqTransEncoder.layers."))
        self.pe = self.sd["sequence_pos_encoder.pe"].reshape(-1, self.d)
        self.text = "embed_text.weight" in self.sd

    # ---- pieces -------------------------------------------------------------------------------
    def timestep_embedding(self, t):
        sd = self.sd
        h = _silu(_linear(self.pe[np.asarray(t)], sd["embed_timestep.time_embed.0.weight"],
                          sd["embed_timestep.time_embed.0.bias"]))
        return _linear(h, sd["embed_timestep.time_embed.2.weight"],
                       sd["embed_timestep.time_embed.2.bias"]).astype(F32)

    def _tokens(self, x, t, enc_text, uncond):
        sd = self.sd
        B, J, Fd, T = x.shape
        emb = self.timestep_embedding(t)                                   # [B, d]
        if self.text:
            c = np.zeros((B, sd["embed_text.weight"].shape[1]), dtype=F32) \
                if (uncond or enc_text is None) else np.asarray(enc_text, dtype=F32)
            emb = emb + _linear(c, sd["embed_text.weight"], sd["embed_text.bias"])
        frames = x.reshape(B, J * Fd, T).transpose(0, 2, 1)                 # [B, T, C]
        tok = _linear(frames, sd["input_process.poseEmbedding.weight"],
                      sd["input_process.poseEmbedding.bias"])               # [B, T, d]
        seq = np.concatenate([emb[:, None, :], tok], axis=1)                # [B, S, d]
        return (seq + self.pe[None, :T + 1, :]).astype(F32)

    def _layer(self, l, h, keep):
        sd, H, d = self.sd, self.H, self.d
        p = f"seqTransEncoder.layers.{l}."
        B, S, _ = h.shape
        dh = d // H
        qkv = _linear(h, sd[p + "self_attn.in_proj_weight"], sd[p + "self_attn.in_proj_bias"])
        q, k, v = (qkv[..., i * d:(i + 1) * d].reshape(B, S, H, dh).transpose(0, 2, 1, 3)
                   for i in range(3))                                       # [B, H, S, dh]
        scale = F32(1.0 / np.sqrt(dh))
        P = _softmax((q * scale) @ k.transpose(0, 1, 3, 2))                 # [B, H, S, S]
        o = (P @ v).transpose(0, 2, 1, 3).reshape(B, S, d)
        a = _linear(o, sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"])
        pre1 = (h + a).astype(F32)
        h1, xh1, r1 = _layernorm(pre1, sd[p + "norm1.weight"], sd[p + "norm1.bias"])
        u = _linear(h1, sd[p + "linear1.weight"], sd[p + "linear1.bias"]).astype(F32)
        ff = _linear(_gelu(u), sd[p + "linear2.weight"], sd[p + "linear2.bias"])
        pre2 = (h1 + ff).astype(F32)
        out, xh2, r2 = _layernorm(pre2, sd[p + "norm2.weight"], sd[p + "norm2.bias"])
        if keep is not None:
            keep.append(dict(q=q, k=k, v=v, P=P, xh1=xh1, r1=r1, u=u, xh2=xh2, r2=r2, scale=scale))
        return out

    # ---- forward ------------------------------------------------------------------------------
    def forward(self, x, t, enc_text=None, uncond=False, keep=None):
        """x [B, J, F, T] fp32, t [B] ORIGINAL timesteps -> [B, J, F, T]."""
        x = np.asarray(x, dtype=F32)
        B, J, Fd, T = x.shape
        h = self._tokens(x, t, enc_text, uncond)
        for l in range(self.L):
            h = self._layer(l, h, keep)
        out = _linear(h[:, 1:, :], self.sd["output_process.poseFinal.weight"],
                      self.sd["output_process.poseFinal.bias"])             # [B, T, C]
        return out.transpose(0, 2, 1).reshape(B, J, Fd, T).astype(F32)

    def forward_cfg(self, x, t, enc_text, text_scale):
        """out_u + s * (out_c - out_u), s broadcast over [B, 1, 1, 1]."""
        oc = self.forward(x, t, enc_text, uncond=False)
        ou = self.forward(x, t, enc_text, uncond=True)
        s = np.asarray(text_scale, dtype=F32).reshape(-1, 1, 1, 1)
        return (ou + (s * (oc - ou))).astype(F32), oc, ou

    # ---- input VJP ----------------------------------------------------------------------------
    def vjp(self, x, t, gout, enc_text=None, uncond=False):
        """(d out / d x)ᵀ · gout for one pass: the x-gradient autograd would produce."""
        sd, H, d = self.sd, self.H, self.d
        keep = []
        x = np.asarray(x, dtype=F32)
        B, J, Fd, T = x.shape
        self.forward(x, t, enc_text, uncond, keep=keep)
        g = np.asarray(gout, dtype=F32).reshape(B, J * Fd, T).transpose(0, 2, 1)  # [B, T, C]
        dh_ = np.zeros((B, T + 1, d), dtype=F32)
        dh_[:, 1:, :] = g @ sd["output_process.poseFinal.weight"]
        for l in reversed(range(self.L)):
            st = keep[l]
            p = f"seqTransEncoder.layers.{l}."
            dpre2 = _layernorm_bwd(dh_, st["xh2"], st["r2"], sd[p + "norm2.weight"])
            du = (dpre2 @ sd[p + "linear2.weight"]) * _gelu_grad(st["u"])
            dh1 = dpre2 + du @ sd[p + "linear1.weight"]
            dpre1 = _layernorm_bwd(dh1, st["xh1"], st["r1"], sd[p + "norm1.weight"])
            do = (dpre1 @ sd[p + "self_attn.out_proj.weight"])
            S = T + 1
            do = do.reshape(B, S, H, d // H).transpose(0, 2, 1, 3)                 # [B, H, S, dh]
            P, q, k, v, scale = st["P"], st["q"], st["k"], st["v"], st["scale"]
            dv = P.transpose(0, 1, 3, 2) @ do
            dP = do @ v.transpose(0, 1, 3, 2)
            dS = P * (dP - (dP * P).sum(axis=-1, keepdims=True, dtype=F32))
            dq = (dS @ k) * scale
            dk = (dS.transpose(0, 1, 3, 2) @ q) * scale
            dqkv = np.concatenate([m.transpose(0, 2, 1, 3).reshape(B, S, d) for m in (dq, dk, dv)],
                                  axis=-1)
            dh_ = (dpre1 + dqkv @ sd[p + "self_attn.in_proj_weight"]).astype(F32)
        gx = dh_[:, 1:, :] @ sd["input_process.poseEmbedding.weight"]              # [B, T, C]
        return gx.transpose(0, 2, 1).reshape(B, J, Fd, T).astype(F32)

    def vjp_cfg(self, x, t, gout, enc_text, text_scale):
        s = np.asarray(text_scale, dtype=F32).reshape(-1, 1, 1, 1)
        g = np.asarray(gout, dtype=F32)
        sg = (s * g).astype(F32)
        return (self.vjp(x, t, sg, enc_text, uncond=False)
                + self.vjp(x, t, (g - sg).astype(F32), enc_text, uncond=True)).astype(F32)
