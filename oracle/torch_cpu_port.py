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
