"""CPU: the load-time weight packing (auralis_amd/weights.py) reproduces conv1d / conv_transpose1d when evaluated
the way conv1d_mfma_kernel evaluates it (emulate_packed_conv mirrors the kernel's index arithmetic)."""
import pytest
import torch
import torch.nn.functional as F

from auralis_amd import weights as Wt


@pytest.mark.parametrize("cout,cin,k,d", [(64, 16, 3, 1), (64, 8, 7, 3), (32, 8, 11, 5), (128, 16, 7, 1)])
def test_packed_conv_equals_conv1d(cout, cin, k, d):
    torch.manual_seed(k * 10 + d)
    w = torch.randn(cout, cin, k)
    b = torch.randn(cout)
    x = torch.randn(cin, 50)
    ref = F.conv1d(F.leaky_relu(x, 0.1)[None], w, b, dilation=d, padding=(k - 1) // 2 * d)[0]
    got = Wt.emulate_packed_conv(x, Wt.pack_conv(w), b, k, d, (k - 1) // 2 * d, 0.1)
    assert (got - ref).abs().max().item() < 1e-4


@pytest.mark.parametrize("cin,cout,s,k", [(16, 8, 8, 16), (8, 32, 2, 4)])
def test_polyphase_equals_conv_transpose(cin, cout, s, k):
    torch.manual_seed(s)
    w = torch.randn(cin, cout, k)
    b = torch.randn(cout)
    x = torch.randn(cin, 37)
    ref = F.conv_transpose1d(F.leaky_relu(x, 0.1)[None], w, b, stride=s, padding=(k - s) // 2)[0]
    wp = Wt.pack_conv(Wt.polyphase_convT(w, s))
    got = Wt.emulate_packed_conv(x, wp, b, 2, 1, 1, 0.1, ups_s=s, ups_p=(k - s) // 2)
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() < 1e-4


def test_pack_all_names_and_shapes(dims, xtts_sd, gpt_sd_small):
    packed = Wt.pack_all(gpt_sd_small, xtts_sd)
    assert packed["voc.conv_pre.wp"].shape == (8, 1024, 7, 64)
    assert packed["voc.ups.0.wp"].shape == (32, 512, 2, 64)       # 256*8 virtual channels
    assert packed["voc.ups.3.wp"].shape == (1, 64, 2, 64)
    assert packed["voc.rb.11.c2.2.wp"].shape == (1, 32, 11, 32)
    assert packed["voc.conv_post.w"].shape == (32, 7)
    assert packed["mel_head.wT"].shape == (1024, 1088)
    assert packed["gpt.h.2.mlp.c_proj.w"].shape == (4096, 1024)
    assert all(v.dtype.name == "float32" for v in packed.values())
