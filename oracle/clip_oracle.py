"""torch-CPU restatement of the CLIP TEXT tower — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference's text encoder is the third-party ``clip`` package (openai/CLIP, installed un-pinned from git by the
reference's README.md:54 / requirements.txt:25-26 and ABSENT from /root/reference and from this image; weights download
at run time).  **Parity unpinned**: there is no golden vector to pin this restatement against; it follows the published
architecture (clip/model.py: CLIP.encode_text, Transformer, ResidualAttentionBlock, QuickGELU, build_attention_mask) and is
anchored on the reference's call sites model/mdm.py:173-186 (clip.load) and :211-237 (tokenize with 20 + 2 tokens padded to
77, encode_text(...).float()).  Layer arithmetic is torch's own (F.multi_head_attention_forward, F.layer_norm), fp32."""
import torch
import torch.nn.functional as F


def encode_text(sd: dict, text: torch.Tensor, heads: int = 8) -> torch.Tensor:
    """sd: openai/CLIP state-dict names -> fp32 CPU tensors; text [B, context] integer ids -> [B, embed_dim]."""
    x = sd["token_embedding.weight"][text.long()] + sd["positional_embedding"]          # [B, S, d]
    B, S, d = x.shape
    mask = torch.full((S, S), float("-inf")).triu_(1)                                    # build_attention_mask
    x = x.permute(1, 0, 2)                                                                # NLD -> LND
    layers = len({k.split(".")[2] for k in sd if k.startswith("transformer.resblocks.")})
    for l in range(layers):
        p = f"transformer.resblocks.{l}."
        h = F.layer_norm(x, (d,), sd[p + "ln_1.weight"], sd[p + "ln_1.bias"])
        a, _ = F.multi_head_attention_forward(
            h, h, h, d, heads, sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"], None, None, False, 0.0,
            sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"], training=False, need_weights=False, attn_mask=mask)
        x = x + a
        h = F.layer_norm(x, (d,), sd[p + "ln_2.weight"], sd[p + "ln_2.bias"])
        u = F.linear(h, sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"])
        u = u * torch.sigmoid(1.702 * u)                                                  # QuickGELU
        x = x + F.linear(u, sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"])
    x = F.layer_norm(x.permute(1, 0, 2), (d,), sd["ln_final.weight"], sd["ln_final.bias"])
    return x[torch.arange(B), text.argmax(dim=-1)] @ sd["text_projection"]
