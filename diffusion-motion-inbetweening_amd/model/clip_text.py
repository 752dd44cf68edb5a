"""CLIP text tower, MI355X-native — the step in front of the sampling loop.

The reference encodes prompts with the third-party ``clip`` package (``clip.load('ViT-B/32')`` at model/mdm.py:173-186,
``clip.tokenize`` + ``clip_model.encode_text`` at :211-237).  This module provides the same two pieces without that
dependency:

* ``CLIPTextTower`` — an ``nn.Module`` that holds the text transformer's parameters under openai/CLIP's own state-dict
  names (so ``tower.load_state_dict(clip_state_dict, strict=False)`` takes a released checkpoint as is) and runs
  ``encode_text`` in libcondmdi_hip.so (csrc/clip_text.hip: embedding, 12 causal pre-LN blocks, ln_final, EOT row @
  text_projection; fp32 — the reference runs the tower in fp16);
* ``SimpleTokenizer`` — the byte-pair tokenizer of openai/CLIP (clip/simple_tokenizer.py) over a vocabulary file the user
  supplies (``bpe_simple_vocab_16e6.txt.gz`` ships with the clip package; it is not available offline here).

``MDM.encode_text`` uses a tower attached as ``model.clip_model`` exactly like the reference's ``clip_model``.
"""
from __future__ import annotations

import ctypes as C
import gzip
import html
import re
from functools import lru_cache
from typing import List, Union

import torch
import torch.nn as nn

from .. import _native as N


class _Block(nn.Module):
    """Parameter holder of clip.model.ResidualAttentionBlock (pre-LN; MLP = c_fc, QuickGELU, c_proj)."""

    def __init__(self, d: int, heads: int):
        super().__init__()
        self.attn = nn.MultiheadAttention(d, heads)
        self.ln_1 = nn.LayerNorm(d)
        self.mlp = nn.Sequential()
        self.mlp.add_module("c_fc", nn.Linear(d, 4 * d))
        self.mlp.add_module("gelu", nn.Identity())     # QuickGELU has no parameters; keeps the reference's key names
        self.mlp.add_module("c_proj", nn.Linear(4 * d, d))
        self.ln_2 = nn.LayerNorm(d)


class _Transformer(nn.Module):
    def __init__(self, width: int, layers: int, heads: int):
        super().__init__()
        self.width, self.layers = width, layers
        self.resblocks = nn.Sequential(*[_Block(width, heads) for _ in range(layers)])


class CLIPTextTower(nn.Module):
    def __init__(self, embed_dim=512, context_length=77, vocab_size=49408, transformer_width=512, transformer_heads=8,
                 transformer_layers=12, bpe_path=None):
        super().__init__()
        self.embed_dim, self.context_length, self.vocab_size = embed_dim, context_length, vocab_size
        self.width, self.heads, self.layers = transformer_width, transformer_heads, transformer_layers
        self.transformer = _Transformer(transformer_width, transformer_layers, transformer_heads)
        self.token_embedding = nn.Embedding(vocab_size, transformer_width)
        self.positional_embedding = nn.Parameter(torch.empty(context_length, transformer_width))
        self.ln_final = nn.LayerNorm(transformer_width)
        self.text_projection = nn.Parameter(torch.empty(transformer_width, embed_dim))
        nn.init.normal_(self.token_embedding.weight, std=0.02)
        nn.init.normal_(self.positional_embedding, std=0.01)
        nn.init.normal_(self.text_projection, std=transformer_width ** -0.5)
        self.tokenizer = SimpleTokenizer(bpe_path) if bpe_path else None
        self._h, self._key, self._max_batch = None, None, 0

    @classmethod
    def from_state_dict(cls, sd: dict, bpe_path=None) -> "CLIPTextTower":
        """Geometry from a released openai/CLIP state dict (clip/model.py build_model); visual.* entries are ignored."""
        width = sd["ln_final.weight"].shape[0]
        layers = len({k.split(".")[2] for k in sd if k.startswith("transformer.resblocks.")})
        tower = cls(embed_dim=sd["text_projection"].shape[1], context_length=sd["positional_embedding"].shape[0],
                    vocab_size=sd["token_embedding.weight"].shape[0], transformer_width=width,
                    transformer_heads=width // 64, transformer_layers=layers, bpe_path=bpe_path)
        own = tower.state_dict()
        tower.load_state_dict({k: v.float() for k, v in sd.items() if k in own}, strict=True)
        return tower.eval()

    # ---- native engine ------------------------------------------------------------------------------------------
    def _engine(self, device, batch):
        key = tuple((p.data_ptr(), p._version) for p in self.parameters()) + (str(device),)
        if self._h is not None and key == self._key and batch <= self._max_batch:
            return self._h
        self.close()
        lib = N.load()
        max_batch = max(batch, self._max_batch, 8)
        desc = N.ClipDesc(self.vocab_size, self.width, self.heads, self.layers, self.context_length, self.embed_dim, max_batch)
        h = C.c_void_p()
        with torch.cuda.device(device):
            N.check(lib.cmdi_clip_create(C.byref(desc), C.byref(h)))
            for name, t in self.state_dict().items():
                t = t.detach().to(device=device, dtype=torch.float32).contiguous()
                N.check(lib.cmdi_clip_load_weight(h, name.encode(), N.ptr(t), t.numel(), N.current_stream(device)))
            torch.cuda.synchronize(device)     # the staging copies above are freed when this returns
        self._h, self._key, self._max_batch = h, key, max_batch
        return h

    def close(self):
        if self._h is not None:
            N.load().cmdi_clip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def encode_text(self, text: torch.Tensor) -> torch.Tensor:
        """text: integer token ids [B, context_length] (clip.tokenize) -> [B, embed_dim] fp32."""
        device = self.positional_embedding.device
        if device.type != "cuda":
            raise N.NativeError("the CLIP text tower runs on a HIP device only (no CPU path): call .to('cuda')")
        assert text.dim() == 2 and text.shape[1] == self.context_length, text.shape
        tokens = text.to(device=device, dtype=torch.int32).contiguous()
        out = torch.empty((tokens.shape[0], self.embed_dim), dtype=torch.float32, device=device)
        h = self._engine(device, tokens.shape[0])
        with torch.cuda.device(device):
            N.check(N.load().cmdi_clip_encode_text(h, N.ptr(tokens), tokens.shape[0], N.ptr(out), N.current_stream(device)))
        return out

    def tokenize(self, texts: Union[str, List[str]], context_length: int = None, truncate: bool = False) -> torch.Tensor:
        if self.tokenizer is None:
            raise N.NativeError("no BPE vocabulary: construct the tower with bpe_path=<bpe_simple_vocab_16e6.txt.gz>")
        return tokenize(self.tokenizer, texts, context_length or self.context_length, truncate)

    def forward(self, text):
        return self.encode_text(text)


# ---- byte-pair tokenizer (openai/CLIP clip/simple_tokenizer.py; ftfy's fix_text is replaced by html.unescape only) ------
@lru_cache()
def bytes_to_unicode():
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    cs = bs[:]
    n = 0
    for b in range(2 ** 8):
        if b not in bs:
            bs.append(b)
            cs.append(2 ** 8 + n)
            n += 1
    return dict(zip(bs, [chr(c) for c in cs]))


def _pairs(word):
    return set(zip(word[:-1], word[1:]))


class SimpleTokenizer:
    def __init__(self, bpe_path: str):
        self.byte_encoder = bytes_to_unicode()
        self.byte_decoder = {v: k for k, v in self.byte_encoder.items()}
        opener = gzip.open if str(bpe_path).endswith(".gz") else open
        with opener(bpe_path, "rt", encoding="utf-8") as fh:
            merges = fh.read().split("\n")
        merges = [tuple(m.split()) for m in merges[1:49152 - 256 - 2 + 1] if len(m.split()) == 2]
        vocab = list(self.byte_encoder.values())
        vocab = vocab + [v + "</w>" for v in vocab]
        vocab += ["".join(m) for m in merges]
        vocab += ["<|startoftext|>", "<|endoftext|>"]
        self.encoder = dict(zip(vocab, range(len(vocab))))
        self.decoder = {v: k for k, v in self.encoder.items()}
        self.bpe_ranks = dict(zip(merges, range(len(merges))))
        self.cache = {"<|startoftext|>": "<|startoftext|>", "<|endoftext|>": "<|endoftext|>"}
        # openai/CLIP's pattern is r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+"
        # (third-party `regex` module, IGNORECASE): Unicode letters / numbers, not ASCII classes.  `regex` is used when it is
        # importable; otherwise _split_unicode below scans with unicodedata categories — the same token boundaries (ADVICE r2:
        # an ASCII pattern split "café", "größe" or "３" differently from the reference's tokenizer, silently)
        try:
            import regex
            self.pat = regex.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+",
                                     regex.IGNORECASE)
        except ImportError:
            self.pat = None

    def bpe(self, token):
        if token in self.cache:
            return self.cache[token]
        word = tuple(token[:-1]) + (token[-1] + "</w>",)
        pairs = _pairs(word)
        if not pairs:
            return token + "</w>"
        while True:
            bigram = min(pairs, key=lambda p: self.bpe_ranks.get(p, float("inf")))
            if bigram not in self.bpe_ranks:
                break
            first, second = bigram
            new, i = [], 0
            while i < len(word):
                try:
                    j = word.index(first, i)
                    new.extend(word[i:j])
                    i = j
                except ValueError:
                    new.extend(word[i:])
                    break
                if word[i] == first and i < len(word) - 1 and word[i + 1] == second:
                    new.append(first + second)
                    i += 2
                else:
                    new.append(word[i])
                    i += 1
            word = tuple(new)
            if len(word) == 1:
                break
            pairs = _pairs(word)
        out = " ".join(word)
        self.cache[token] = out
        return out

    def encode(self, text):
        text = re.sub(r"\s+", " ", html.unescape(html.unescape(text)).strip()).strip().lower()
        ids = []
        for token in (self.pat.findall(text) if self.pat is not None else _split_unicode(text)):
            token = "".join(self.byte_encoder[b] for b in token.encode("utf-8"))
            ids.extend(self.encoder[t] for t in self.bpe(token).split(" "))
        return ids

    def decode(self, tokens):
        text = "".join(self.decoder[t] for t in tokens)
        return bytearray(self.byte_decoder[c] for c in text).decode("utf-8", errors="replace").replace("</w>", " ")


def _split_unicode(text: str):
    """The token boundaries of openai/CLIP's pattern (see SimpleTokenizer.__init__) without the `regex` module: alternatives
    tried in its order at every position — special tokens, the seven contractions, a run of letters (category L*), ONE number
    character (N*), a run of anything else that is not whitespace."""
    import unicodedata
    kind = lambda ch: unicodedata.category(ch)[0]
    out, i, n = [], 0, len(text)
    while i < n:
        hit = next((sp for sp in ("<|startoftext|>", "<|endoftext|>", "'s", "'t", "'re", "'ve", "'m", "'ll", "'d")
                    if text[i:i + len(sp)].lower() == sp), None)
        if hit is not None:
            out.append(text[i:i + len(hit)])
            i += len(hit)
            continue
        ch = text[i]
        if ch.isspace():
            i += 1
        elif kind(ch) == "L":
            j = i + 1
            while j < n and kind(text[j]) == "L":
                j += 1
            out.append(text[i:j])
            i = j
        elif kind(ch) == "N":
            out.append(ch)
            i += 1
        else:
            j = i + 1
            while j < n and not text[j].isspace() and kind(text[j]) not in ("L", "N"):
                j += 1
            out.append(text[i:j])
            i = j
    return out


def tokenize(tokenizer: SimpleTokenizer, texts, context_length: int = 77, truncate: bool = False) -> torch.Tensor:
    """clip.tokenize: <|startoftext|> ids <|endoftext|>, zero-padded to context_length."""
    if isinstance(texts, str):
        texts = [texts]
    sot, eot = tokenizer.encoder["<|startoftext|>"], tokenizer.encoder["<|endoftext|>"]
    out = torch.zeros(len(texts), context_length, dtype=torch.int32)
    for i, text in enumerate(texts):
        ids = [sot] + tokenizer.encode(text) + [eot]
        if len(ids) > context_length:
            if not truncate:
                raise RuntimeError(f"Input {text} is too long for context length {context_length}")
            ids = ids[:context_length]
            ids[-1] = eot
        out[i, :len(ids)] = torch.tensor(ids, dtype=torch.int32)
    return out
