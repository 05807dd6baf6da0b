"""CPU: the CLIP tokenizer / text transformer restatement (text.py) against the `transformers` implementation installed in the image --
the third-party code the reference's pipeline call uses for `prompt=""` (extract.py:66).  Random weights and a synthetic vocabulary:
there is no checkpoint here; what is pinned is the algorithm (BPE merge order, padding, causal mask, activation, final layer norm)."""
import json
import os

import numpy as np
import pytest
import torch

import gswm_amd
from gswm_amd import text as T

transformers = pytest.importorskip("transformers")


def _write_tokenizer(d, pad):
    alphabet = list(T._byte_alphabet().values())
    vocab = {}
    for ch in alphabet:
        vocab[ch] = len(vocab)
    for ch in alphabet:
        vocab[ch + "</w>"] = len(vocab)
    merges = ["a t</w>", "c at</w>", "t h", "th e</w>", "d o", "do g</w>", "i n", "in g</w>", "p h", "ph o", "pho t", "phot o</w>", "o f</w>",
              "s u", "su n", "sun s", "e t</w>", "1 0", "! !</w>", "' s</w>"]
    for m in merges:
        vocab["".join(m.split())] = len(vocab)
    vocab["<|startoftext|>"] = len(vocab)
    vocab["<|endoftext|>"] = len(vocab)
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "vocab.json"), "w", encoding="utf-8") as f:
        json.dump(vocab, f, ensure_ascii=False)
    with open(os.path.join(d, "merges.txt"), "w", encoding="utf-8") as f:
        f.write("#version: 0.2\n" + "\n".join(merges) + "\n")
    with open(os.path.join(d, "special_tokens_map.json"), "w") as f:
        json.dump({"bos_token": "<|startoftext|>", "eos_token": "<|endoftext|>", "unk_token": "<|endoftext|>", "pad_token": pad}, f)
    with open(os.path.join(d, "tokenizer_config.json"), "w") as f:
        json.dump({"model_max_length": 77, "pad_token": pad, "tokenizer_class": "CLIPTokenizer"}, f)
    return vocab


PROMPTS = ["", "a photo of the cat", "The DOG's sunset, photo 10!!", "  thing   in the  cat  ", "photo " * 60, "x" * 200, "café at 3pm :-)"]


@pytest.mark.parametrize("pad", ["<|endoftext|>", "!"])          # SD 1.x pads with <end>, SD 2.x with '!' (id 0)
def test_tokenizer_matches_transformers(tmp_path, pad):
    d = str(tmp_path / "tokenizer")
    _write_tokenizer(d, pad)
    ref = transformers.CLIPTokenizer.from_pretrained(d)
    mine = T.ClipTokenizer.from_dir(d)
    want = ref(PROMPTS, padding="max_length", max_length=ref.model_max_length, truncation=True, return_tensors="pt").input_ids
    got = mine(PROMPTS)
    assert got.shape == (len(PROMPTS), 77)
    assert torch.equal(got, want)
    assert got[0, 0] == mine.bos_id and got[0, 1] == mine.eos_id and (got[0, 2:] == mine.pad_id).all()       # the reference's prompt ""


@pytest.mark.parametrize("act,layers,heads", [("quick_gelu", 2, 4), ("gelu", 3, 2)])
def test_text_encoder_matches_transformers(tmp_path, act, layers, heads):
    cfg = transformers.CLIPTextConfig(vocab_size=300, hidden_size=64, intermediate_size=128, num_hidden_layers=layers, num_attention_heads=heads,
                                      max_position_embeddings=77, hidden_act=act, projection_dim=32)
    torch.manual_seed(0)
    ref = transformers.CLIPTextModel(cfg).eval()
    with torch.no_grad():
        for p in ref.parameters():
            p.copy_(torch.randn_like(p) * (0.3 if p.dim() > 1 else 0.1) + (1.0 if "norm" in "" else 0.0))
    d = str(tmp_path / "text_encoder")
    ref.save_pretrained(d, safe_serialization=True)
    mine = T.ClipTextEncoder.from_dir(d).eval()
    ids = torch.randint(0, 300, (3, 77))
    ids[:, 0] = 298
    ids[0, 1:] = 299                                                                                        # an "empty prompt" row
    with torch.no_grad():
        want = ref(input_ids=ids)[0]
    got = mine(ids)
    assert got.shape == (3, 77, 64)
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=2e-5, atol=2e-5)


def test_encode_prompt_from_dir(tmp_path):
    root = str(tmp_path)
    vocab = _write_tokenizer(os.path.join(root, "tokenizer"), "!")
    cfg = transformers.CLIPTextConfig(vocab_size=len(vocab), hidden_size=32, intermediate_size=64, num_hidden_layers=1, num_attention_heads=2,
                                      max_position_embeddings=77, hidden_act="gelu")
    torch.manual_seed(1)
    ref = transformers.CLIPTextModel(cfg).eval()
    ref.save_pretrained(os.path.join(root, "text_encoder"), safe_serialization=True)
    tok = transformers.CLIPTokenizer.from_pretrained(os.path.join(root, "tokenizer"))
    ids = tok(["", "a photo of the cat"], padding="max_length", max_length=77, truncation=True, return_tensors="pt").input_ids
    with torch.no_grad():
        want = ref(input_ids=ids)[0]
    got = T.encode_prompt_from_dir(root, ["", "a photo of the cat"], "cpu", torch.float32)
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=2e-5, atol=2e-5)
