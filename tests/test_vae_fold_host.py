"""CPU: the 1x1 quant_conv / post_quant_conv of diffusers' AutoencoderKL (behind extract.py:41 `vae.encode`) folded into the neighbouring 3x3
convolutions (vae._pw_out_folded / _pw_in_folded) equal the two-convolution form -- at the image border too."""
import torch
import torch.nn as nn
import torch.nn.functional as F

import gswm_amd  # noqa: F401
from gswm_amd import vae as V


def _unpack(wp, cin):
    n = wp.shape[0]
    return wp.view(n, 3, 3, cin).permute(0, 3, 1, 2).contiguous()


def test_quant_conv_folds_into_conv_out():
    torch.manual_seed(0)
    conv_out, post = nn.Conv2d(64, 8, 3, padding=1).double(), nn.Conv2d(8, 8, 1).double()
    x = torch.randn(2, 64, 6, 5, dtype=torch.float64)
    ref = post(conv_out(x))
    wp, b = V._pw_out_folded(conv_out, post)
    assert wp.shape == (64, 9 * 64) and b.shape == (64,)
    y = F.conv2d(x, _unpack(wp, 64), b, padding=1)
    assert (y[:, :8] - ref).abs().max().item() < 1e-5          # the fold is composed in fp32
    assert y[:, 8:].abs().max().item() == 0


def test_post_quant_conv_folds_into_conv_in_with_a_ones_channel():
    torch.manual_seed(1)
    pre, conv_in = nn.Conv2d(4, 4, 1).double(), nn.Conv2d(4, 128, 3, padding=1).double()
    z = torch.randn(2, 4, 5, 7, dtype=torch.float64)
    ref = conv_in(pre(z))
    wp, b = V._pw_in_folded(conv_in, pre)
    assert wp.shape == (128, 9 * 64)
    zin = torch.zeros(2, 64, 5, 7, dtype=torch.float64)
    zin[:, :4] = z
    zin[:, 4] = 1.0                                             # the constant channel that carries the 1x1 bias (zero in the padding)
    y = F.conv2d(zin, _unpack(wp, 64), b, padding=1)
    assert (y - ref).abs().max().item() < 1e-5


def test_encoder_token_constraint_sends_odd_lattices_to_the_torch_path():
    enc = V.Encoder((64, 64, 64, 64))
    assert enc._pf_tokens_ok(torch.empty(1, 3, 512, 512)) and not enc._pf_tokens_ok(torch.empty(1, 3, 520, 520))
