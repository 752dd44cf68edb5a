"""A-priori bounds on every tensor the default precision (f16x3) carries as split f16 rows — computed from the WEIGHTS alone.

Why: the f16x3 engine carries activations as (hi, lo) f16 pairs, |x| < 65504; a value outside sets the
device range flag and the chain is re-run on bf16x6 at 1.7-2.1 x the cost.  The guard is always there; this module answers
how close a given checkpoint can come to it — not from a sample of inputs but for EVERY input inside stated assumptions.

The reference's denoiser is torch's post-norm ``nn.TransformerEncoder`` (reference model/mdm.py:107-114:
``nn.TransformerEncoderLayer`` with its defaults, norm_first False), so every sublayer's output passes a LayerNorm before it is
used again.  A LayerNorm output y = xhat * gamma + beta has
|xhat_i| <= sqrt(d - 1), ||xhat||_2 <= sqrt(d) WHATEVER its input was, hence for any linear map behind it

    |W_j . y + b_j|  <=  min( sqrt(d) ||W_j * gamma||_2 ,  sqrt(d - 1) ||W_j * gamma||_1 )  +  |W_j . beta + b_j| .

Everything else follows by interval arithmetic per channel: attention output = a convex combination of v rows (<= the bound of
v), GELU(z) in [-0.17, max(z, 0)], residual sums add bounds, the time embedding is evaluated exactly for all timesteps.  Only
layer 0 sees un-normalised rows: the tokens of the input projection, bounded by ||W_in[j]||_1 * x_bound, and the embedding
token, bounded through ``text_l2_bound`` (the CLIP feature's 2-norm; uncond rows use c = 0).

The bound is rigorous for exact arithmetic; fp32 rounding moves values by ~1e-6 relative, covered by ``SLACK``.  What it does
NOT cover: the input-VJP of reconstruction guidance (its gradients depend on the guidance weight and the data; the device
rescales them by a power of two — tests run gout x 1e-12 ... 1e9 — and the guard watches them), and MDM_UNET (GroupNorm statistics over 128 x T values give sqrt(n) ~ 160 per site and un-normalised
residual paths between levels: no useful bound without data) — for those the run-time guard and bf16x6 remain the answer.

Host-side numpy on the state dict; no device, no oracle.  ``MDM.range_certificate()`` is the entry point for a loaded module,
``tools/range_certificate.py`` the one for a checkpoint file.
"""
from __future__ import annotations

import math

import numpy as np

F16_LIMIT = 65504.0
SLACK = 1.001          # fp32 rounding of the computed values against the exact-arithmetic bound


def _ln_out_bound(gamma, beta, d):
    return math.sqrt(d - 1.0) * np.abs(gamma) + np.abs(beta)


def _after_ln_bound(W, b, gamma, beta, d):
    """Per output channel j: |W_j . LN(x) + b_j| for ANY x."""
    Wg = W * gamma[None, :]
    var = np.minimum(math.sqrt(d) * np.sqrt((Wg * Wg).sum(axis=1)), math.sqrt(d - 1.0) * np.abs(Wg).sum(axis=1))
    return var + np.abs(W @ beta + (b if b is not None else 0.0))


def _silu(x):
    return x / (1.0 + np.exp(-x))


def trans_enc_range_certificate(sd: dict, *, x_bound: float = 16.0, text_l2_bound: float = 32.0, n_frames: int = 196,
                                n_timesteps: int = 1000) -> dict:
    """Bounds for MDM(arch='trans_enc') from its state dict (reference names, numpy or torch values).

    Assumptions (returned with the result): |x_t| <= ``x_bound`` on every feature of every frame (x_T ~ N(0, 1) over 1.6M draws
    peaks near 5.3; HumanML3D features are normalised), ||CLIP text feature||_2 <= ``text_l2_bound`` (ViT-B/32: ~ 7-12), at most
    ``n_frames`` frames, timesteps 0 <= t < ``n_timesteps``.
    """
    g = lambda k: np.asarray(sd[k].detach().cpu().numpy() if hasattr(sd[k], "detach") else sd[k], dtype=np.float64)
    W_in, b_in = g("input_process.poseEmbedding.weight"), g("input_process.poseEmbedding.bias")
    d = W_in.shape[0]
    pe = g("sequence_pos_encoder.pe").reshape(-1, d)
    if pe.shape[0] < n_frames + 1:
        raise ValueError(f"positional table has {pe.shape[0]} rows, need {n_frames + 1}")
    L = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("seqTransEncoder.layers."))
    # token 0: time embedding (exact over every timestep) + text projection (+ pe[0])
    h = _silu(pe[:min(n_timesteps, pe.shape[0])] @ g("embed_timestep.time_embed.0.weight").T + g("embed_timestep.time_embed.0.bias"))
    emb_fixed = h @ g("embed_timestep.time_embed.2.weight").T + g("embed_timestep.time_embed.2.bias") + pe[0]   # [t, d]: exact
    Wt = None
    if "embed_text.weight" in sd:                 # embed_text(mask_cond(c)): c = 0 for unconditional rows leaves the bias
        Wt = g("embed_text.weight")
        emb_fixed = emb_fixed + g("embed_text.bias")
    u_emb = np.abs(emb_fixed).max(axis=0) + (np.sqrt((Wt * Wt).sum(axis=1)) * text_l2_bound if Wt is not None else 0.0)
    tok_fixed = b_in[None, :] + pe[1:n_frames + 1]                                                              # [p, d]: exact
    u_tok = np.abs(W_in).sum(axis=1) * x_bound + np.abs(tok_fixed).max(axis=0)
    u_src = np.maximum(u_emb, u_tok)
    tensors = {"frames": float(x_bound), "tokens": float(u_src.max())}
    layers = []
    u_res, prev = u_src, None             # the layer input as the residual reads it; (gamma, beta) of the LayerNorm that made it
    for l in range(L):
        p = f"seqTransEncoder.layers.{l}."
        Wqkv, bqkv = g(p + "self_attn.in_proj_weight"), g(p + "self_attn.in_proj_bias")
        Wo, bo = g(p + "self_attn.out_proj.weight"), g(p + "self_attn.out_proj.bias")
        W1, b1, W2, b2 = g(p + "linear1.weight"), g(p + "linear1.bias"), g(p + "linear2.weight"), g(p + "linear2.bias")
        g1, be1, g2, be2 = g(p + "norm1.weight"), g(p + "norm1.bias"), g(p + "norm2.weight"), g(p + "norm2.bias")
        if prev is None:
            # layer 0 reads un-normalised rows: bound the COMPOSITE maps x -> qkv (cancellation inside W_qkv W_in counts) instead
            # of chaining two interval products
            via_tok = np.abs(Wqkv @ W_in).sum(axis=1) * x_bound + np.abs(tok_fixed @ Wqkv.T + bqkv).max(axis=0)
            via_emb = np.abs(emb_fixed @ Wqkv.T + bqkv).max(axis=0)
            if Wt is not None:
                WW = Wqkv @ Wt
                via_emb = via_emb + np.sqrt((WW * WW).sum(axis=1)) * text_l2_bound
            u_qkv = np.maximum(via_tok, via_emb)
        else:
            u_qkv = _after_ln_bound(Wqkv, bqkv, prev[0], prev[1], d)
        u_attn = u_qkv[2 * d:]                                  # softmax rows are convex weights over the v rows
        u_pre1 = u_res + np.abs(Wo) @ u_attn + np.abs(bo)
        u_ln1 = _ln_out_bound(g1, be1, d)
        u_ffn = np.maximum(_after_ln_bound(W1, b1, g1, be1, d), 0.17)   # |GELU(z)| <= max(|z|, 0.17)
        u_pre2 = u_ln1 + np.abs(W2) @ u_ffn + np.abs(b2)
        u_ln2 = _ln_out_bound(g2, be2, d)
        layers.append({"qkv": float(u_qkv.max()), "attention": float(u_attn.max()), "pre_norm1": float(u_pre1.max()),
                       "norm1": float(u_ln1.max()), "ffn_hidden": float(u_ffn.max()), "pre_norm2": float(u_pre2.max()),
                       "norm2": float(u_ln2.max())})
        u_res, prev = u_ln2, (g2, be2)
    for name in layers[0]:
        tensors[name] = max(lay[name] for lay in layers)
    worst = max(tensors.values()) * SLACK
    return {"arch": "trans_enc", "limit": F16_LIMIT, "max_bound": worst, "certified": bool(worst < F16_LIMIT),
            "headroom_bits": math.log2(F16_LIMIT / worst) if worst > 0 else float("inf"),
            "tensors": tensors, "layers": layers,
            "assumptions": {"x_bound": x_bound, "text_l2_bound": text_l2_bound, "n_frames": n_frames, "n_timesteps": n_timesteps,
                            "covers": "forward evaluations (plain sampling, CFG, imputation); not the guidance VJP, not MDM_UNET"}}
