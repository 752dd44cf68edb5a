"""CLIP text tower: the byte-pair tokenizer and the torch oracle on CPU — the oracle pinned against
an INDEPENDENT implementation of the same published architecture (HuggingFace transformers' CLIPTextModelWithProjection,
random weights) since the reference's own `clip` package and its weights are absent offline — and the native tower vs
the oracle on the GPU."""
import gzip

import numpy as np
import pytest
import torch

from conftest import rel_l2, sub
from oracle import clip_oracle


def random_clip_sd(seed, vocab, width, layers, ctx=77, embed=None):
    g = torch.Generator().manual_seed(seed)
    embed = embed or width
    r = lambda *s, std=0.02: torch.randn(*s, generator=g) * std
    sd = {"token_embedding.weight": r(vocab, width), "positional_embedding": r(ctx, width, std=0.01),
          "ln_final.weight": 1 + r(width, std=0.1), "ln_final.bias": r(width, std=0.1),
          "text_projection": r(width, embed, std=width ** -0.5)}
    for l in range(layers):
        p = f"transformer.resblocks.{l}."
        sd.update({p + "ln_1.weight": 1 + r(width, std=0.1), p + "ln_1.bias": r(width, std=0.1),
                   p + "attn.in_proj_weight": r(3 * width, width, std=width ** -0.5), p + "attn.in_proj_bias": r(3 * width),
                   p + "attn.out_proj.weight": r(width, width, std=width ** -0.5), p + "attn.out_proj.bias": r(width),
                   p + "ln_2.weight": 1 + r(width, std=0.1), p + "ln_2.bias": r(width, std=0.1),
                   p + "mlp.c_fc.weight": r(4 * width, width, std=width ** -0.5), p + "mlp.c_fc.bias": r(4 * width),
                   p + "mlp.c_proj.weight": r(width, 4 * width, std=(4 * width) ** -0.5), p + "mlp.c_proj.bias": r(width)})
    return sd


def random_tokens(seed, B, vocab, ctx=77):
    """clip.tokenize layout: <sot> ids <eot> then zeros; <eot> = vocab - 1 is the largest id."""
    rng = np.random.default_rng(seed)
    t = np.zeros((B, ctx), dtype=np.int64)
    for b in range(B):
        n = int(rng.integers(1, 21))
        t[b, 0] = vocab - 2
        t[b, 1:1 + n] = rng.integers(1, vocab - 2, n)
        t[b, 1 + n] = vocab - 1
    return torch.from_numpy(t)


def test_oracle_matches_an_independent_clip_implementation():
    tr = pytest.importorskip("transformers")
    width, layers, heads, vocab = 256, 3, 4, 1000
    cfg = tr.CLIPTextConfig(vocab_size=vocab, hidden_size=width, intermediate_size=4 * width, num_hidden_layers=layers,
                            num_attention_heads=heads, max_position_embeddings=77, hidden_act="quick_gelu",
                            projection_dim=width, eos_token_id=2, attn_implementation="eager")
    hf = tr.CLIPTextModelWithProjection(cfg).eval()
    sd = random_clip_sd(3, vocab, width, layers)
    m = {"text_model.embeddings.token_embedding.weight": sd["token_embedding.weight"],
         "text_model.embeddings.position_embedding.weight": sd["positional_embedding"],
         "text_model.final_layer_norm.weight": sd["ln_final.weight"], "text_model.final_layer_norm.bias": sd["ln_final.bias"],
         "text_projection.weight": sd["text_projection"].T.contiguous()}
    for l in range(layers):
        p, q = f"transformer.resblocks.{l}.", f"text_model.encoder.layers.{l}."
        wq, wk, wv = sd[p + "attn.in_proj_weight"].chunk(3)
        bq, bk, bv = sd[p + "attn.in_proj_bias"].chunk(3)
        m.update({q + "self_attn.q_proj.weight": wq, q + "self_attn.k_proj.weight": wk, q + "self_attn.v_proj.weight": wv,
                  q + "self_attn.q_proj.bias": bq, q + "self_attn.k_proj.bias": bk, q + "self_attn.v_proj.bias": bv,
                  q + "self_attn.out_proj.weight": sd[p + "attn.out_proj.weight"], q + "self_attn.out_proj.bias": sd[p + "attn.out_proj.bias"],
                  q + "layer_norm1.weight": sd[p + "ln_1.weight"], q + "layer_norm1.bias": sd[p + "ln_1.bias"],
                  q + "layer_norm2.weight": sd[p + "ln_2.weight"], q + "layer_norm2.bias": sd[p + "ln_2.bias"],
                  q + "mlp.fc1.weight": sd[p + "mlp.c_fc.weight"], q + "mlp.fc1.bias": sd[p + "mlp.c_fc.bias"],
                  q + "mlp.fc2.weight": sd[p + "mlp.c_proj.weight"], q + "mlp.fc2.bias": sd[p + "mlp.c_proj.bias"]})
    missing, unexpected = hf.load_state_dict(m, strict=False)
    assert not unexpected and all("position_ids" in k for k in missing), (missing, unexpected)
    tokens = random_tokens(5, 6, vocab)
    with torch.no_grad():
        want = hf(input_ids=tokens).text_embeds
    got = clip_oracle.encode_text(sd, tokens, heads=heads)
    assert rel_l2(got.numpy(), want.numpy()) <= 2e-6, rel_l2(got.numpy(), want.numpy())


def make_bpe_file(path):
    """A miniature merges file in the format of bpe_simple_vocab_16e6.txt.gz (header line, then 'a b' merges)."""
    merges = ["w a", "l k</w>", "wa lk</w>", "p e", "pe r", "per s", "o n</w>", "pers on</w>"]
    with gzip.open(path, "wt", encoding="utf-8") as fh:
        fh.write('"bpe_simple_vocab_16e6.txt#version: 0.2\\n'.replace("\\n", "\n"))
        fh.write("\n".join(merges) + "\n")
    return merges


def test_simple_tokenizer(tmp_path):
    ct = sub("model.clip_text")
    merges = make_bpe_file(tmp_path / "bpe.txt.gz")
    tok = ct.SimpleTokenizer(str(tmp_path / "bpe.txt.gz"))
    n = 512 + len(merges)
    assert tok.encoder["<|startoftext|>"] == n and tok.encoder["<|endoftext|>"] == n + 1
    ids = tok.encode("A  person   WALKS")
    assert tok.decode(ids).strip() == "a person walks"
    assert tok.encoder["person</w>"] in ids and tok.encoder["walk</w>"] not in ids        # 'walks' != 'walk'
    t = ct.tokenize(tok, ["a person walks", "walk"], context_length=22)
    assert t.shape == (2, 22) and t.dtype == torch.int32
    assert t[0, 0] == n and t[1, 0] == n and t[1, 1] == tok.encoder["walk</w>"] and t[1, 2] == n + 1 and t[1, 3:].sum() == 0
    assert int(t[0].argmax()) == len(ids) + 1                                               # <eot> is the largest id
    with pytest.raises(RuntimeError):
        ct.tokenize(tok, ["walk " * 40], context_length=22)
    tt = ct.tokenize(tok, ["walk " * 40], context_length=22, truncate=True)
    assert tt[0, -1] == n + 1 and tt[0, 0] == n


def test_tokenizer_splits_unicode_like_openai_clip(tmp_path):
    """openai/CLIP's pattern uses \\p{L} / \\p{N} (third-party `regex` module), not ASCII classes: non-ASCII letters and
    digits must stay inside / form their own tokens.  The scanner used when `regex` is absent gives the same boundaries."""
    ct = sub("model.clip_text")
    strings = ["a person walks forward, then don't stop!!'s 12 steps", "café größe ３ ½ x² naïve—it's", "<|startoftext|>hello<|endoftext|>",
               "tab\tand\nnewline  'll 've", "日本語のテキスト123 ok", "don''t !'s"]
    want = [["café", "größe", "３", "½", "x", "²", "naïve", "—", "it", "'s"]]
    assert ct._split_unicode(strings[1]) == want[0]
    try:
        import regex
    except ImportError:
        return
    pat = regex.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+", regex.IGNORECASE)
    for t in strings:
        assert pat.findall(t) == ct._split_unicode(t), t
    make_bpe_file(tmp_path / "bpe.txt.gz")
    tok = ct.SimpleTokenizer(str(tmp_path / "bpe.txt.gz"))
    assert tok.pat is not None and tok.decode(tok.encode("Café  ３")).strip() == "café ３"


def test_tower_state_dict_names_are_openai_clips():
    ct = sub("model.clip_text")
    tower = ct.CLIPTextTower(vocab_size=1000, transformer_width=256, transformer_heads=4, transformer_layers=2, embed_dim=256)
    names = set(tower.state_dict())
    want = set(random_clip_sd(0, 1000, 256, 2))
    assert names == want, names ^ want
    t2 = ct.CLIPTextTower.from_state_dict(dict(random_clip_sd(1, 1000, 256, 2), **{"visual.proj": torch.zeros(3)}))
    assert (t2.width, t2.heads, t2.layers, t2.vocab_size) == (256, 4, 2, 1000)
    with pytest.raises(sub("_native").NativeError):
        t2.encode_text(torch.zeros(1, 77, dtype=torch.long))                                   # no CPU path


@pytest.mark.gpu
@pytest.mark.parametrize("geom", ["vit_b32", "small"])
def test_native_text_tower_vs_oracle(geom):
    """csrc/clip_text.hip vs the torch restatement, random weights: the ViT-B/32 text geometry (12 x 512, 8 heads,
    vocab 49408, context 77 — what clip.load('ViT-B/32') gives the reference) and a small one."""
    ct = sub("model.clip_text")
    vocab, width, layers = (49408, 512, 12) if geom == "vit_b32" else (1000, 256, 2)
    sd = random_clip_sd(7, vocab, width, layers)
    tower = ct.CLIPTextTower.from_state_dict(sd).to("cuda:0")
    tokens = random_tokens(9, 5, vocab)
    want = clip_oracle.encode_text(sd, tokens, heads=width // 64)
    got = tower.encode_text(tokens.to("cuda:0"))
    assert got.shape == (5, width) and rel_l2(got.cpu().numpy(), want.numpy()) <= 2e-5, rel_l2(got.cpu().numpy(), want.numpy())
    again = tower.encode_text(tokens[:2])                       # smaller batch on the same engine, rows independent
    assert torch.equal(again, got[:2])


@pytest.mark.gpu
def test_mdm_encode_text_through_the_native_tower(tmp_path):
    """MDM.encode_text (reference model/mdm.py:211-237: 20 + 2 tokens, zero-padded to 77) with a CLIPTextTower attached as
    clip_model, and p_sample_loop taking y['text'] (no precomputed embedding) end to end."""
    from types import SimpleNamespace
    from oracle import weights
    ct, mu = sub("model.clip_text"), sub("utils.model_util")
    merges = make_bpe_file(tmp_path / "bpe.txt.gz")
    vocab = 512 + len(merges) + 2
    sd = random_clip_sd(11, vocab, 512, 2)
    model, _ = mu.create_model_and_diffusion(SimpleNamespace(dataset="humanml", layers=1), None)
    mu.load_model_wo_clip(model, weights.to_torch(weights.make_state_dict(2, text=True, n_layers=1)))
    model.clip_model = ct.CLIPTextTower.from_state_dict(sd, bpe_path=str(tmp_path / "bpe.txt.gz"))
    model.to("cuda:0").eval()
    texts = ["a person walks", "walk"]
    emb = model.encode_text(texts)
    tok = ct.tokenize(model.clip_model.tokenizer, texts, context_length=22, truncate=True)
    tok = torch.cat([tok, torch.zeros(2, 55, dtype=tok.dtype)], dim=1)
    want = clip_oracle.encode_text(sd, tok, heads=8)
    assert rel_l2(emb.cpu().numpy(), want.numpy()) <= 2e-5
    x, t = torch.randn(2, 263, 1, 24, device="cuda:0"), torch.tensor([10, 500], device="cuda:0")
    a = model(x, t, y={"text": texts})                           # CLIP(text) inside the call, as the reference does
    b = model(x, t, y={"text_embed": emb})
    assert torch.equal(a, b)
