"""Deterministic synthetic MDM weights (no checkpoints exist offline — SURVEY.md §8d).

`make_state_dict` draws every tensor of the reference's MDM trans_enc state dict (key names and
shapes of SURVEY.md §5.4) from numpy's PCG64 stream, so the golden-fixture generator (which loads
them into the REAL reference model), the numpy oracle, the GPU tests and bench.py all see the same
weights without shipping 70 MB.  Scales follow torch's default inits (uniform ±1/sqrt(fan_in)),
with non-trivial LayerNorm affine terms and biases so that every term of the path is exercised.
"""
from __future__ import annotations

import numpy as np


def positional_table(max_len: int, d: int) -> np.ndarray:
    """fp32 sinusoidal table, same arithmetic as the reference (model/mdm.py:322-327):
    float32 position * float32 exp(arange(0,d,2) * (-ln(1e4)/d))."""
    pos = np.arange(max_len, dtype=np.float32)[:, None]
    div = np.exp(np.arange(0, d, 2, dtype=np.float32) * np.float32(-np.log(10000.0) / d)).astype(np.float32)
    pe = np.zeros((max_len, d), dtype=np.float32)
    ang = (pos * div).astype(np.float32)
    pe[:, 0::2] = np.sin(ang)
    pe[:, 1::2] = np.cos(ang)
    return pe


def make_state_dict(seed: int = 0, *, n_layers: int = 8, d: int = 512, f: int = 1024,
                    n_feats: int = 263, text: bool = True, clip_dim: int = 512,
                    pe_rows: int = 5000, pe: np.ndarray | None = None) -> dict:
    rng = np.random.default_rng(seed)

    def lin(out_f, in_f):
        k = 1.0 / np.sqrt(in_f)
        w = rng.uniform(-k, k, size=(out_f, in_f)).astype(np.float32)
        b = rng.uniform(-k, k, size=(out_f,)).astype(np.float32)
        return w, b

    sd = {}
    sd["input_process.poseEmbedding.weight"], sd["input_process.poseEmbedding.bias"] = lin(d, n_feats)
    table = positional_table(pe_rows, d) if pe is None else np.asarray(pe, dtype=np.float32)
    sd["sequence_pos_encoder.pe"] = table.reshape(pe_rows, 1, d)
    for l in range(n_layers):
        p = f"seqTransEncoder.layers.{l}."
        sd[p + "self_attn.in_proj_weight"], sd[p + "self_attn.in_proj_bias"] = lin(3 * d, d)
        sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"] = lin(d, d)
        sd[p + "linear1.weight"], sd[p + "linear1.bias"] = lin(f, d)
        sd[p + "linear2.weight"], sd[p + "linear2.bias"] = lin(d, f)
        for nm in ("norm1", "norm2"):
            sd[p + nm + ".weight"] = (1.0 + 0.1 * rng.standard_normal(d)).astype(np.float32)
            sd[p + nm + ".bias"] = (0.1 * rng.standard_normal(d)).astype(np.float32)
    sd["embed_timestep.sequence_pos_encoder.pe"] = sd["sequence_pos_encoder.pe"]
    sd["embed_timestep.time_embed.0.weight"], sd["embed_timestep.time_embed.0.bias"] = lin(d, d)
    sd["embed_timestep.time_embed.2.weight"], sd["embed_timestep.time_embed.2.bias"] = lin(d, d)
    if text:
        sd["embed_text.weight"], sd["embed_text.bias"] = lin(d, clip_dim)
    sd["output_process.poseFinal.weight"], sd["output_process.poseFinal.bias"] = lin(n_feats, d)
    return sd


def to_torch(sd: dict):
    import torch
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}


def fill_like(shapes: dict, seed: int = 0) -> dict:
    """Deterministic synthetic values for ANY state dict given as {name: shape} (used for MDM_UNET, whose
    reference initialisation zeroes half of its convolutions — useless for parity tests): names are visited
    in sorted order; '*.weight' of rank >= 2 ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in)), rank-1 '*.weight'
    (GroupNorm) ~ 1 + 0.1 N(0,1), biases ~ 0.1 N(0,1); positional tables ('*.pe') are not produced."""
    rng = np.random.default_rng(seed)
    out = {}
    for name in sorted(shapes):
        shape = tuple(shapes[name])
        if name.endswith(".pe"):
            continue
        if name.endswith(".weight") and len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            k = 1.0 / np.sqrt(fan_in)
            out[name] = rng.uniform(-k, k, size=shape).astype(np.float32)
        elif name.endswith(".weight") or name.endswith(".norm.g"):   # (norm.g: the U-Net's channel LayerNorm gain [1, C, 1])
            out[name] = (1.0 + 0.1 * rng.standard_normal(shape)).astype(np.float32)
        else:
            out[name] = (0.1 * rng.standard_normal(shape)).astype(np.float32)
    return out
