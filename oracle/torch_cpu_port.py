"""torch-CPU restatement of the reference denoiser, for bench.py's cpu_baseline leg.

TEST / MEASUREMENT INFRASTRUCTURE ONLY (see oracle/__init__.py): never imported by the product.

The reference's MDM trans_enc is a thin wrapper around torch's own nn.TransformerEncoder (constructed at
model/mdm.py:107-114, called at :284), so timing THIS module on the host cores times the same third-party
CPU kernels (oneDNN / native matmul, softmax, LayerNorm, GELU) the reference's CPU path runs — a closer
stand-in for "the reference on this host" than the numpy oracle, whose small batched matmuls thread badly.
Follows: token assembly model/mdm.py:244-280 (embed_timestep :351-353, embed_text + mask_cond :188-198,248-251,
InputProcess :366-372, PositionalEncoding :332-335), encoder :284, OutputProcess :409-423, classifier-free
guidance as TWO sequential passes model/cfg_sampler.py:25-35, posterior update gaussian_diffusion.py:696-711.
Pinned by tests/test_oracle_golden.py::test_torch_cpu_port_matches_golden.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn


class TorchCpuMDM(nn.Module):
    def __init__(self, sd: dict, n_heads: int = 4):
        super().__init__()
        g = lambda k: torch.from_numpy(np.ascontiguousarray(sd[k], dtype=np.float32))
        d = sd["input_process.poseEmbedding.weight"].shape[0]
        f = sd["seqTransEncoder.layers.0.linear1.weight"].shape[0]
        L = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("seqTransEncoder.layers."))
        layer = nn.TransformerEncoderLayer(d_model=d, nhead=n_heads, dim_feedforward=f, dropout=0.1,
                                           activation="gelu")
        self.enc = nn.TransformerEncoder(layer, num_layers=L, enable_nested_tensor=False)
        self.enc.load_state_dict({k[len("seqTransEncoder."):]: g(k) for k in sd
                                  if k.startswith("seqTransEncoder.")})
        self.pe = g("sequence_pos_encoder.pe").reshape(-1, d)
        self.w_in, self.b_in = g("input_process.poseEmbedding.weight"), g("input_process.poseEmbedding.bias")
        self.w_out, self.b_out = g("output_process.poseFinal.weight"), g("output_process.poseFinal.bias")
        self.t1 = (g("embed_timestep.time_embed.0.weight"), g("embed_timestep.time_embed.0.bias"))
        self.t2 = (g("embed_timestep.time_embed.2.weight"), g("embed_timestep.time_embed.2.bias"))
        self.txt = (g("embed_text.weight"), g("embed_text.bias")) if "embed_text.weight" in sd else None
        self.eval()

    @torch.no_grad()
    def forward(self, x, t, enc_text=None, uncond=False):
        """x [B, J, 1, T] fp32 tensor, t [B] long (original timesteps) -> [B, J, 1, T]."""
        B, J, Fd, T = x.shape
        lin = nn.functional.linear
        emb = lin(nn.functional.silu(lin(self.pe[t], *self.t1)), *self.t2)           # [B, d]
        if self.txt is not None:
            c = torch.zeros(B, self.txt[0].shape[1]) if (uncond or enc_text is None) else enc_text
            emb = emb + lin(c, *self.txt)
        frames = x.reshape(B, J * Fd, T).permute(2, 0, 1)                              # [T, B, C]
        xseq = torch.cat([emb[None], lin(frames, self.w_in, self.b_in)], dim=0)       # [S, B, d]
        xseq = xseq + self.pe[:T + 1, None, :]
        out = lin(self.enc(xseq)[1:], self.w_out, self.b_out)                          # [T, B, C]
        return out.permute(1, 2, 0).reshape(B, J, Fd, T)

    @torch.no_grad()
    def forward_cfg(self, x, t, enc_text, text_scale):
        oc = self.forward(x, t, enc_text, uncond=False)
        ou = self.forward(x, t, enc_text, uncond=True)
        return ou + text_scale.view(-1, 1, 1, 1) * (oc - ou), oc, ou


class TorchCpuUNET(nn.Module):
    """MDM_UNET (keyframe-conditioned, AdaGN) on torch's CPU kernels — the reference's TemporalUnet IS a stack of
    nn.Conv1d / nn.GroupNorm / nn.Mish / nn.ConvTranspose1d (model/mdm_unet.py:15-99,165-358), so the functional
    calls below run the same oneDNN / native kernels as the reference's CPU path.  Follows MDM_UNET.forward /
    forward_core (:767-849) and TemporalUnet.forward (:318-358); weights by the reference's state-dict names.
    Pinned by tests/test_oracle_golden.py::test_torch_cpu_unet_matches_golden."""

    def __init__(self, sd: dict):
        super().__init__()
        self.w = {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)) if isinstance(v, np.ndarray)
                  else v.detach().float().cpu() for k, v in sd.items()}
        self.n_levels = 1 + max(int(k.split(".")[2]) for k in self.w if k.startswith("unet.downs."))
        self.pe = self.w["sequence_pos_encoder.pe"].reshape(-1, self.w["sequence_pos_encoder.pe"].shape[-1])

    def _block(self, p, x, ss=None):
        F, w = nn.functional, self.w
        pre = "block1" if ss is not None else "block"
        cw = w[f"{p}.{pre}.0.weight"]
        h = F.conv1d(x, cw, w[f"{p}.{pre}.0.bias"], padding=cw.shape[2] // 2)
        h = F.group_norm(h, 8, w[f"{p}.{pre}.2.weight"], w[f"{p}.{pre}.2.bias"])
        if ss is not None:
            scale, shift = ss.chunk(2, dim=1)
            h = h * (1 + scale[:, :, None]) + shift[:, :, None]
        return F.mish(h)

    def _res(self, p, x, c):
        F, w = nn.functional, self.w
        ss = F.linear(F.mish(c), w[f"{p}.time_mlp.1.weight"], w[f"{p}.time_mlp.1.bias"])
        h = self._block(f"{p}.blocks.1", self._block(f"{p}.blocks.0", x, ss))
        if f"{p}.residual_conv.weight" in w:
            x = F.conv1d(x, w[f"{p}.residual_conv.weight"], w[f"{p}.residual_conv.bias"])
        return h + x

    @torch.no_grad()
    def forward(self, x, t, enc_text=None, uncond=False, obs_x0=None, obs_mask=None):
        return self.forward_impl(x, t, enc_text, uncond, obs_x0, obs_mask)

    def forward_impl(self, x, t, enc_text=None, uncond=False, obs_x0=None, obs_mask=None):
        """The same without the no_grad guard: torch autograd through it is the test reference of the native input-VJP."""
        F, w = nn.functional, self.w
        B, J, Fd, T = x.shape
        if obs_mask is not None:
            x = torch.cat([obs_x0 * obs_mask + x * (~obs_mask), obs_mask.float()], dim=1)
        emb = F.linear(F.silu(F.linear(self.pe[t], w["embed_timestep.time_embed.0.weight"],
                                       w["embed_timestep.time_embed.0.bias"])),
                       w["embed_timestep.time_embed.2.weight"], w["embed_timestep.time_embed.2.bias"])
        if "embed_text.weight" in w:
            ctext = torch.zeros(B, w["embed_text.weight"].shape[1]) if (uncond or enc_text is None) else enc_text
            emb = emb + F.linear(ctext, w["embed_text.weight"], w["embed_text.bias"])
        h = F.pad(x.reshape(B, -1, T), (0, 224 - T))
        c = F.linear(F.mish(F.linear(emb, w["unet.time_mlp.0.weight"], w["unet.time_mlp.0.bias"])),
                     w["unet.time_mlp.2.weight"], w["unet.time_mlp.2.bias"])
        skips, n = [], self.n_levels
        for l in range(n):
            h = self._res(f"unet.downs.{l}.1", self._res(f"unet.downs.{l}.0", h, c), c)
            skips.append(h)
            if l < n - 1:
                h = F.conv1d(h, w[f"unet.downs.{l}.3.conv.weight"], w[f"unet.downs.{l}.3.conv.bias"], stride=2, padding=1)
        h = self._res("unet.mid_block2", self._res("unet.mid_block1", h, c), c)
        for u in range(n - 1):
            h = torch.cat((h, skips.pop()), dim=1)
            h = self._res(f"unet.ups.{u}.1", self._res(f"unet.ups.{u}.0", h, c), c)
            h = F.conv_transpose1d(h, w[f"unet.ups.{u}.3.conv.weight"], w[f"unet.ups.{u}.3.conv.bias"], stride=2, padding=1)
        h = F.conv1d(self._block("unet.final_conv.0", h), w["unet.final_conv.1.weight"], w["unet.final_conv.1.bias"])
        return h[:, :, :T].reshape(B, J, Fd, T)

    @torch.no_grad()
    def forward_cfg(self, x, t, enc_text, text_scale, obs_x0=None, obs_mask=None):
        oc = self.forward(x, t, enc_text, False, obs_x0, obs_mask)
        ou = self.forward(x, t, enc_text, True, obs_x0, obs_mask)
        return ou + text_scale.view(-1, 1, 1, 1) * (oc - ou), oc, ou
