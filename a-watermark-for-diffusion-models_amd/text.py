"""Text context for the UNet: CLIP byte-pair tokenizer + CLIP text transformer (rows X2 / G1 of SURVEY.md section 8a: "prompt '' -> context").

Reference: extract.py:66 calls the stock pipeline with `prompt=""`; diffusers' `encode_prompt` tokenises with `CLIPTokenizer` (padding
'max_length' = 77, truncation) and takes `CLIPTextModel(input_ids)[0]` -- the last hidden state after the final layer norm, causal mask
only (SD configs do not pass an attention mask).  Both live in `transformers` (not vendored by the reference); the restatement below is
pinned against the transformers implementation installed in this image on randomly initialised models / a synthetic vocabulary
(tests/test_text_host.py) -- CLIP itself has not changed between the reference's pinned version and this one.

This runs ONCE per run (the empty prompt is one [1,77,D] constant for every image and every step), so it is plain torch on the device in
the checkpoint's dtype -- not part of the per-step hot path and deliberately not a hand-written kernel.
"""
from __future__ import annotations

import json
import os
from functools import lru_cache
from typing import List, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F


# ---------------------------------------------------------------------------------------------------------------------
# tokenizer (CLIP's lower-cased byte-level BPE with the '</w>' end-of-word marker)
# ---------------------------------------------------------------------------------------------------------------------
@lru_cache()
def _byte_alphabet():
    """The reversible byte -> printable-unicode table of byte-level BPE: printable latin-1 bytes map to themselves, the rest to 256+."""
    keep = list(range(ord("!"), ord("~") + 1)) + list(range(ord("\xa1"), ord("\xac") + 1)) + list(range(ord("\xae"), ord("\xff") + 1))
    table, extra = {}, 0
    for b in range(256):
        if b in keep:
            table[b] = chr(b)
        else:
            table[b] = chr(256 + extra)
            extra += 1
    return table


class ClipTokenizer:
    def __init__(self, vocab: dict, merges: Sequence[str], *, bos="<|startoftext|>", eos="<|endoftext|>", pad=None, max_length=77):
        import regex
        self.vocab = vocab
        self.rank = {tuple(m.split()): i for i, m in enumerate(merges)}
        self.bos_id, self.eos_id = vocab[bos], vocab[eos]
        self.pad_id = vocab[pad] if pad is not None else self.eos_id
        self.max_length = max_length
        self._split = regex.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+", regex.IGNORECASE)
        self._cache = {}
        # the tokenizer library cuts its special tokens out of the text before BPE -- including a pad token that is an ordinary character
        # (SD 2.x pads with '!', so every '!' of a prompt becomes the pad id): mirrored here
        specials = sorted({bos, eos} | ({pad} if pad is not None else set()), key=len, reverse=True)
        self._specials = regex.compile("(" + "|".join(regex.escape(t) for t in specials) + ")")

    @classmethod
    def from_dir(cls, tok_dir: str) -> "ClipTokenizer":
        with open(os.path.join(tok_dir, "vocab.json"), encoding="utf-8") as f:
            vocab = json.load(f)
        with open(os.path.join(tok_dir, "merges.txt"), encoding="utf-8") as f:
            lines = f.read().split("\n")
        merges = [ln for ln in lines[1:] if ln and not ln.startswith("#") and len(ln.split()) == 2]      # first line is the version header
        pad, max_length = None, 77
        for name in ("special_tokens_map.json", "tokenizer_config.json"):
            p = os.path.join(tok_dir, name)
            if os.path.exists(p):
                with open(p, encoding="utf-8") as f:
                    cfg = json.load(f)
                pt = cfg.get("pad_token")
                if pad is None and pt is not None:
                    pad = pt["content"] if isinstance(pt, dict) else pt
                if isinstance(cfg.get("model_max_length"), int) and cfg["model_max_length"] < 10_000:
                    max_length = cfg["model_max_length"]
        return cls(vocab, merges, pad=pad, max_length=max_length)

    def _bpe(self, word: str) -> List[str]:
        if word in self._cache:
            return self._cache[word]
        parts = list(word[:-1]) + [word[-1] + "</w>"]
        while len(parts) > 1:
            best, where = None, None
            for i in range(len(parts) - 1):
                r = self.rank.get((parts[i], parts[i + 1]))
                if r is not None and (best is None or r < best):
                    best, where = r, (parts[i], parts[i + 1])
            if best is None:
                break
            merged, i = [], 0
            while i < len(parts):
                if i < len(parts) - 1 and (parts[i], parts[i + 1]) == where:
                    merged.append(parts[i] + parts[i + 1])
                    i += 2
                else:
                    merged.append(parts[i])
                    i += 1
            parts = merged
        self._cache[word] = parts
        return parts

    def encode(self, text: str) -> List[int]:
        """Token ids WITHOUT the start / end markers."""
        text = " ".join(text.split()).strip().lower()
        table = _byte_alphabet()
        ids = []
        for span in self._specials.split(text):
            if span in self.vocab and self._specials.fullmatch(span):
                ids.append(self.vocab[span])
                continue
            for piece in self._split.findall(span):
                word = "".join(table[b] for b in piece.encode("utf-8"))
                ids += [self.vocab[t] for t in self._bpe(word)]
        return ids

    def __call__(self, prompts: Sequence[str]) -> torch.Tensor:
        """[B, max_length] int64: <start> tokens <end> padded with the pad token, truncated to max_length keeping <end> last."""
        rows = []
        for p in prompts:
            ids = [self.bos_id] + self.encode(p)[: self.max_length - 2] + [self.eos_id]
            rows.append(ids + [self.pad_id] * (self.max_length - len(ids)))
        return torch.tensor(rows, dtype=torch.int64)


# ---------------------------------------------------------------------------------------------------------------------
# text transformer
# ---------------------------------------------------------------------------------------------------------------------
def _act(name: str):
    if name == "quick_gelu":
        return lambda x: x * torch.sigmoid(1.702 * x)
    if name == "gelu":
        return F.gelu
    if name in ("gelu_new", "gelu_pytorch_tanh"):
        return lambda x: F.gelu(x, approximate="tanh")
    raise ValueError(f"text_encoder/config.json: hidden_act {name!r} is not supported")


class _SelfAttention(nn.Module):
    def __init__(self, d, heads):
        super().__init__()
        self.heads = heads
        self.q_proj, self.k_proj, self.v_proj, self.out_proj = (nn.Linear(d, d) for _ in range(4))

    def forward(self, x, mask):
        B, S, D = x.shape
        h = self.heads
        q, k, v = (p(x).view(B, S, h, D // h).transpose(1, 2) for p in (self.q_proj, self.k_proj, self.v_proj))
        w = torch.softmax((q * (D // h) ** -0.5) @ k.transpose(-1, -2) + mask, dim=-1, dtype=torch.float32).to(x.dtype)
        return self.out_proj((w @ v).transpose(1, 2).reshape(B, S, D))


class _MLP(nn.Module):
    def __init__(self, d, inner, act):
        super().__init__()
        self.fc1, self.fc2, self.act = nn.Linear(d, inner), nn.Linear(inner, d), _act(act)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class _Layer(nn.Module):
    def __init__(self, d, heads, inner, act, eps):
        super().__init__()
        self.layer_norm1, self.self_attn = nn.LayerNorm(d, eps=eps), _SelfAttention(d, heads)
        self.layer_norm2, self.mlp = nn.LayerNorm(d, eps=eps), _MLP(d, inner, act)

    def forward(self, x, mask):
        x = x + self.self_attn(self.layer_norm1(x), mask)
        return x + self.mlp(self.layer_norm2(x))


class _Embeddings(nn.Module):
    def __init__(self, vocab, positions, d):
        super().__init__()
        self.token_embedding, self.position_embedding = nn.Embedding(vocab, d), nn.Embedding(positions, d)


class _Encoder(nn.Module):
    def __init__(self, n, *a):
        super().__init__()
        self.layers = nn.ModuleList([_Layer(*a) for _ in range(n)])


class _TextModel(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        d = cfg["hidden_size"]
        eps = cfg.get("layer_norm_eps", 1e-5)
        self.embeddings = _Embeddings(cfg["vocab_size"], cfg.get("max_position_embeddings", 77), d)
        self.encoder = _Encoder(cfg["num_hidden_layers"], d, cfg["num_attention_heads"], cfg["intermediate_size"], cfg.get("hidden_act", "quick_gelu"), eps)
        self.final_layer_norm = nn.LayerNorm(d, eps=eps)


class ClipTextEncoder(nn.Module):
    """Parameter names follow the checkpoint (`text_model.embeddings...`, `text_model.encoder.layers.N...`) so it loads 1:1."""

    def __init__(self, cfg: dict):
        super().__init__()
        self.cfg = cfg
        self.text_model = _TextModel(cfg)

    @torch.no_grad()
    def forward(self, input_ids: torch.Tensor) -> torch.Tensor:
        """[B, S] ids -> [B, S, D] last hidden state (after the final layer norm)."""
        m = self.text_model
        S = input_ids.shape[1]
        x = m.embeddings.token_embedding(input_ids) + m.embeddings.position_embedding.weight[:S]
        mask = torch.full((S, S), torch.finfo(x.dtype).min, dtype=x.dtype, device=x.device).triu_(1)
        for layer in m.encoder.layers:
            x = layer(x, mask)
        return m.final_layer_norm(x)

    @classmethod
    def from_dir(cls, enc_dir: str) -> "ClipTextEncoder":
        from .checkpoint import load_component_state_dict
        with open(os.path.join(enc_dir, "config.json")) as f:
            cfg = json.load(f)
        model = cls(cfg)
        raw = load_component_state_dict(enc_dir, "text_encoder")       # model.safetensors / .fp16 / sharded / pytorch_model.bin
        sd = {k: v for k, v in raw.items() if not k.endswith("position_ids") and not k.startswith("text_projection")}
        sd = {(k if k.startswith("text_model.") else "text_model." + k): v for k, v in sd.items()}       # newer transformers drop the prefix
        missing, unexpected = model.load_state_dict(sd, strict=False)
        if missing or unexpected:
            raise RuntimeError(f"text encoder state dict mismatch: missing {missing[:5]}, unexpected {unexpected[:5]}")
        return model


_ENCODERS = {}


def encode_prompt_from_dir(model_dir: str, prompts: Sequence[str], device, dtype) -> torch.Tensor:
    """`<model_dir>/tokenizer` + `<model_dir>/text_encoder` -> [B, 77, D] context in `dtype` on `device` (diffusers encode_prompt without
    classifier-free guidance)."""
    key = (os.path.abspath(model_dir), str(device), dtype)
    if key not in _ENCODERS:
        tok = ClipTokenizer.from_dir(os.path.join(model_dir, "tokenizer"))
        enc = ClipTextEncoder.from_dir(os.path.join(model_dir, "text_encoder")).to(device, dtype).eval()
        _ENCODERS[key] = (tok, enc)
    tok, enc = _ENCODERS[key]
    return enc(tok(list(prompts)).to(device)).to(dtype)
