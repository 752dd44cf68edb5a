"""numpy fp32 restatement of the MDM_UNET denoiser (keyframe-conditioned, AdaGN).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Follows, piece by piece:
    observation merge    reference model/mdm_unet.py:778-783  (x = obs*mask + x*~mask ; cat mask as channels)
    conditioning         :794-805 (TimestepEmbedder :881-895, embed_text + mask_cond :727-737)
    frame padding        :806-823 (permute to [frames, bs, C], right-pad to 224) and :824-848 (crop, reshape)
    TemporalUnet         :214-358 (time_mlp :236-241, downs/mid/ups :243-305, final_conv :307-311,
                         forward :318-358 with the skip stack and channel concat)
    ResidualTemporalBlock :165-212 (time_mlp = Mish -> Linear; AdaGN on the first conv only; 1x1 residual conv
                         when the channel counts differ)
    Conv1dBlock / Conv1dAdaGNBlock :33-99 (Conv1d k=5 pad 2 -> GroupNorm(8) -> [x*(1+scale)+shift] -> Mish)
    Downsample1d / Upsample1d :15-30 (Conv1d(3, stride 2, pad 1) / ConvTranspose1d(4, stride 2, pad 1))
    CFG                  model/cfg_sampler.py:25-35
    LinearAttention      :102-156 (Residual(PreNorm(LayerNorm over channels, LinearAttention)): to_qkv 1x1 without
                         bias, 4 heads x 32, q * 32^-0.5, softmax of k over the FRAMES, context = k v^T, to_out 1x1);
                         present iff the state dict holds "unet.downs.0.2.fn.fn.to_qkv.weight" (attention=True)
Pinned by tests/golden/unet_fwd.npz and unet_attn_fwd.npz (outputs of the real reference run on CPU,
tests/golden/make_golden_unet.py).
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
GN_EPS = F32(1e-5)
GN_GROUPS = 8
PAD_FRAMES = 224


def _mish(x):
    # x * tanh(softplus(x)); softplus with torch's threshold of 20 (beyond it softplus(x) = x)
    sp = np.where(x > F32(20.0), x, np.log1p(np.exp(np.minimum(x, F32(20.0))))).astype(F32)
    return (x * np.tanh(sp)).astype(F32)


def _silu(x):
    return (x / (F32(1.0) + np.exp(-x))).astype(F32)


def _linear(x, w, b):
    return (x @ w.T + b).astype(F32)


def conv1d(x, w, b, stride=1, pad=0):
    """x [B, Cin, L], w [Cout, Cin, k] -> [B, Cout, (L + 2 pad - k)/stride + 1]  (cross-correlation, as torch)."""
    B, Cin, L = x.shape
    Cout, _, k = w.shape
    xp = np.zeros((B, Cin, L + 2 * pad), dtype=F32)
    xp[:, :, pad:pad + L] = x
    Lo = (L + 2 * pad - k) // stride + 1
    # columns [B, Lo, Cin * k] (tap fastest within a channel, matching w.reshape(Cout, Cin*k))
    idx = (np.arange(Lo) * stride)[:, None] + np.arange(k)[None, :]
    cols = xp[:, :, idx]                                   # [B, Cin, Lo, k]
    cols = cols.transpose(0, 2, 1, 3).reshape(B, Lo, Cin * k)
    y = cols @ w.reshape(Cout, Cin * k).T + b
    return np.ascontiguousarray(y.transpose(0, 2, 1), dtype=F32)


def conv_transpose1d(x, w, b, stride=2, pad=1):
    """x [B, Cin, L], w [Cin, Cout, k] -> [B, Cout, (L-1) stride - 2 pad + k]  (torch ConvTranspose1d)."""
    B, Cin, L = x.shape
    _, Cout, k = w.shape
    full = np.zeros((B, Cout, (L - 1) * stride + k), dtype=F32)
    for j in range(k):
        # every input frame i scatters x[:, :, i] @ w[:, :, j] to output frame i*stride + j
        full[:, :, j:j + (L - 1) * stride + 1:stride] += np.einsum("bcl,co->bol", x, w[:, :, j]).astype(F32)
    Lo = (L - 1) * stride - 2 * pad + k
    return np.ascontiguousarray(full[:, :, pad:pad + Lo] + b[None, :, None], dtype=F32)


def group_norm(x, g, b, groups=GN_GROUPS):
    B, C, L = x.shape
    xg = x.reshape(B, groups, (C // groups) * L)
    mean = xg.mean(axis=-1, keepdims=True, dtype=F32)
    xc = xg - mean
    var = (xc * xc).mean(axis=-1, keepdims=True, dtype=F32)
    xh = (xc / np.sqrt(var + GN_EPS)).reshape(B, C, L)
    return (xh * g[None, :, None] + b[None, :, None]).astype(F32)


class UnetOracle:
    """sd: dict name -> numpy fp32 array with the reference's MDM_UNET state-dict names."""

    def __init__(self, sd: dict):
        self.sd = {k: np.asarray(v, dtype=F32) for k, v in sd.items()}
        self.n_levels = 1 + max(int(k.split(".")[2]) for k in self.sd if k.startswith("unet.downs."))
        self.pe = self.sd["sequence_pos_encoder.pe"].reshape(-1, self.sd["sequence_pos_encoder.pe"].shape[-1])

    # ---- blocks ------------------------------------------------------------------------------------
    def _conv_block(self, p, x, ss=None):
        sd = self.sd
        pre = "block1" if ss is not None else "block"
        w = sd[f"{p}.{pre}.0.weight"]
        h = conv1d(x, w, sd[f"{p}.{pre}.0.bias"], pad=w.shape[2] // 2)
        h = group_norm(h, sd[f"{p}.{pre}.2.weight"], sd[f"{p}.{pre}.2.bias"])
        if ss is not None:
            scale, shift = np.split(ss, 2, axis=1)                     # c.chunk(2, dim=1): scale first
            h = h * (F32(1.0) + scale[:, :, None]) + shift[:, :, None]
        return _mish(h.astype(F32))

    def _res_block(self, p, x, c):
        sd = self.sd
        ss = _linear(_mish(c), sd[f"{p}.time_mlp.1.weight"], sd[f"{p}.time_mlp.1.bias"])
        h = self._conv_block(f"{p}.blocks.0", x, ss)
        h = self._conv_block(f"{p}.blocks.1", h)
        if f"{p}.residual_conv.weight" in sd:
            x = conv1d(x, sd[f"{p}.residual_conv.weight"], sd[f"{p}.residual_conv.bias"])
        return (h + x).astype(F32)

    def _attention(self, p, x):
        """Residual(PreNorm(dim, LinearAttention(dim))) (reference :102-156); identity when the block is nn.Identity."""
        sd = self.sd
        if f"{p}.fn.fn.to_qkv.weight" not in sd:
            return x
        heads, dh = 4, 32
        var = x.var(axis=1, keepdims=True, dtype=F32)                  # torch.var(unbiased=False) over the channels
        mean = x.mean(axis=1, keepdims=True, dtype=F32)
        xn = ((x - mean) / np.sqrt(var + GN_EPS) * sd[f"{p}.fn.norm.g"] + sd[f"{p}.fn.norm.b"]).astype(F32)
        qkv = conv1d(xn, sd[f"{p}.fn.fn.to_qkv.weight"], np.zeros(3 * heads * dh, dtype=F32))
        B, _, n = qkv.shape
        q, k, v = [t.reshape(B, heads, dh, n) for t in np.split(qkv, 3, axis=1)]
        q = q * F32(dh ** -0.5)
        k = np.exp(k - k.max(axis=-1, keepdims=True))
        k = (k / k.sum(axis=-1, keepdims=True, dtype=F32)).astype(F32)
        context = np.einsum("bhdn,bhen->bhde", k, v).astype(F32)
        out = np.einsum("bhde,bhdn->bhen", context, q).astype(F32).reshape(B, heads * dh, n)
        return (conv1d(out, sd[f"{p}.fn.fn.to_out.weight"], sd[f"{p}.fn.fn.to_out.bias"]) + x).astype(F32)

    def temporal_unet(self, x, cond):
        """x [B, C, 224] ; cond [B, d] -> [B, C_out, 224]."""
        sd = self.sd
        c = _linear(_mish(_linear(cond, sd["unet.time_mlp.0.weight"], sd["unet.time_mlp.0.bias"])),
                    sd["unet.time_mlp.2.weight"], sd["unet.time_mlp.2.bias"])
        skips = []
        n = self.n_levels
        for l in range(n):
            x = self._res_block(f"unet.downs.{l}.0", x, c)
            x = self._res_block(f"unet.downs.{l}.1", x, c)
            x = self._attention(f"unet.downs.{l}.2", x)
            skips.append(x)
            if l < n - 1:
                x = conv1d(x, sd[f"unet.downs.{l}.3.conv.weight"], sd[f"unet.downs.{l}.3.conv.bias"], stride=2, pad=1)
        x = self._res_block("unet.mid_block1", x, c)
        x = self._attention("unet.mid_attn", x)
        x = self._res_block("unet.mid_block2", x, c)
        for u in range(n - 1):
            x = np.concatenate([x, skips.pop()], axis=1)
            x = self._res_block(f"unet.ups.{u}.0", x, c)
            x = self._res_block(f"unet.ups.{u}.1", x, c)
            x = self._attention(f"unet.ups.{u}.2", x)
            x = conv_transpose1d(x, sd[f"unet.ups.{u}.3.conv.weight"], sd[f"unet.ups.{u}.3.conv.bias"])
        x = self._conv_block("unet.final_conv.0", x)
        return conv1d(x, sd["unet.final_conv.1.weight"], sd["unet.final_conv.1.bias"])

    # ---- MDM_UNET.forward ---------------------------------------------------------------------------
    def timestep_embedding(self, t):
        sd = self.sd
        h = _silu(_linear(self.pe[np.asarray(t)], sd["embed_timestep.time_embed.0.weight"],
                          sd["embed_timestep.time_embed.0.bias"]))
        return _linear(h, sd["embed_timestep.time_embed.2.weight"], sd["embed_timestep.time_embed.2.bias"])

    def forward(self, x, t, enc_text=None, uncond=False, obs_x0=None, obs_mask=None):
        sd = self.sd
        B, J, Fd, T = x.shape
        x = np.asarray(x, dtype=F32)
        if obs_mask is not None:
            m = np.asarray(obs_mask, dtype=bool)
            x = np.where(m, np.asarray(obs_x0, dtype=F32), x)
            x = np.concatenate([x, m.astype(F32)], axis=1)
        emb = self.timestep_embedding(t)
        if "embed_text.weight" in sd:
            ctext = np.zeros((B, sd["embed_text.weight"].shape[1]), dtype=F32) \
                if (uncond or enc_text is None) else np.asarray(enc_text, dtype=F32)
            emb = emb + _linear(ctext, sd["embed_text.weight"], sd["embed_text.bias"])
        frames = np.zeros((B, x.shape[1] * Fd, PAD_FRAMES), dtype=F32)
        frames[:, :, :T] = x.reshape(B, x.shape[1] * Fd, T)
        y = self.temporal_unet(frames, emb.astype(F32))[:, :, :T]
        return y.reshape(B, J, Fd, T).astype(F32)

    def forward_cfg(self, x, t, enc_text, text_scale, obs_x0=None, obs_mask=None):
        oc = self.forward(x, t, enc_text, False, obs_x0, obs_mask)
        ou = self.forward(x, t, enc_text, True, obs_x0, obs_mask)
        s = np.asarray(text_scale, dtype=F32).reshape(-1, 1, 1, 1)
        return (ou + s * (oc - ou)).astype(F32), oc, ou
