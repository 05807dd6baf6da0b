"""GPU: the CLIP text transformer (text.py) in fp16 on the device against its own fp32 evaluation on the host -- the context of prompt "" is what
every UNet call of the inversion consumes (extract.py:66)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_text_encoder_fp16_on_device_matches_fp32_on_host(tmp_path):
    import gswm_amd
    from gswm_amd import text as T
    from test_text_host import _write_tokenizer
    root = str(tmp_path)
    vocab = _write_tokenizer(os.path.join(root, "tokenizer"), "!")
    cfg = {"vocab_size": len(vocab), "hidden_size": 128, "intermediate_size": 512, "num_hidden_layers": 3, "num_attention_heads": 4,
           "max_position_embeddings": 77, "hidden_act": "gelu", "layer_norm_eps": 1e-5}
    torch.manual_seed(3)
    enc = T.ClipTextEncoder(cfg).eval()
    ids = T.ClipTokenizer.from_dir(os.path.join(root, "tokenizer"))(["", "a photo of the cat"])
    ref = enc(ids)
    got = enc.half().cuda()(ids.cuda())
    assert got.dtype == torch.float16 and got.shape == (2, 77, 128)
    assert (got.float().cpu() - ref).abs().max().item() <= 2e-2 * max(1.0, ref.abs().max().item())
