"""CPU: the load-time weight packing (auralis_amd/weights.py) reproduces conv1d / conv_transpose1d when evaluated
the way conv1d_mfma_kernel evaluates it (emulate_packed_conv mirrors the kernel's index arithmetic)."""
import numpy as np
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


# ---- speaker-conditioning tensors (weights.pack_conditioning -> csrc/cond_net.h); host-side emulation of what the kernels compute
def _cond_pack(dims):
    from auralis_amd.checkpoint import make_synthetic_conditioning_weights
    from auralis_amd.weights import pack_conditioning
    sd = make_synthetic_conditioning_weights(dims, seed=3)
    sd["mel_stats"] = torch.ones(80)
    return sd, pack_conditioning(sd)


def test_pack_conditioning_conv_im2col_matches_conv2d(dims):
    """NHWC im2col rows (ky, kx, cin) x packed [9*Cin_pad][Cout_pad] weights == F.conv2d, stride 1 and 2, with the 32-channel stage
    padded to 64 channels; folded BatchNorm == F.batch_norm (eval)."""
    import torch.nn.functional as F
    sd, d = _cond_pack(dims)
    s = "hifigan_decoder.speaker_encoder."
    g = torch.Generator().manual_seed(0)
    for name, packed, stride, cin, cin_pad, cout in ((s + "layer1.0.conv1.weight", "cond.spk.layer1.0.conv1.w", 1, 32, 64, 32),
                                                      (s + "layer2.0.conv1.weight", "cond.spk.layer2.0.conv1.w", 2, 32, 64, 64),
                                                      (s + "layer3.1.conv2.weight", "cond.spk.layer3.1.conv2.w", 1, 128, 128, 128)):
        x = torch.randn(1, cin, 9, 13, generator=g)
        ref = F.conv2d(x, sd[name], None, stride=stride, padding=1)[0]               # [cout][Ho][Wo]
        Ho, Wo = ref.shape[1:]
        xn = np.zeros((9, 13, cin_pad), np.float32)
        xn[:, :, :cin] = x[0].permute(1, 2, 0).numpy()
        xp = np.pad(xn, ((1, 1), (1, 1), (0, 0)))
        cols = np.zeros((Ho * Wo, 9 * cin_pad), np.float32)
        for ho in range(Ho):
            for wo in range(Wo):
                patch = xp[ho * stride: ho * stride + 3, wo * stride: wo * stride + 3, :]   # [ky][kx][c]
                cols[ho * Wo + wo] = patch.reshape(-1)
        W = d[packed]
        assert W.shape[0] >= 9 * cin_pad and W.shape[0] % 16 == 0 and W.shape[1] % 64 == 0
        got = cols @ W[: 9 * cin_pad]
        assert np.abs(got[:, :cout] - ref.permute(1, 2, 0).reshape(-1, cout).numpy()).max() < 1e-4
        assert W.shape[1] == cout or np.abs(got[:, cout:]).max() == 0.0                 # padded output channels stay zero
    y = torch.randn(5, 64, generator=g)
    p = s + "layer2.0.bn1."
    ref = F.batch_norm(y, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"], False, 0.0, 1e-5)
    got = y.numpy() * d["cond.spk.layer2.0.bn1.scale"][None, :] + d["cond.spk.layer2.0.bn1.shift"][None, :]
    assert np.abs(got - ref.numpy()).max() < 1e-5


def test_pack_conditioning_front_end_tables(dims):
    """The fixed tables the HIP mel front-ends use: frames x DFT matrix -> |.|^2 equals torch.stft's power spectrum; the polyphase
    resampling table reproduces conditioning.resample; filterbank / window padding as the kernels index them."""
    from auralis_amd import conditioning as Cn
    _, d = _cond_pack(dims)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 6000, generator=g) * 0.1
    for tag, n_fft, win, hop, kind in (("gpt", 2048, 1024, 256, "hann"), ("spk", 512, 400, 160, "hamming")):
        w = (torch.hann_window if kind == "hann" else torch.hamming_window)(win, periodic=True)
        spec = torch.stft(x, n_fft, hop_length=hop, win_length=win, window=w, center=True, pad_mode="reflect", return_complex=True)[0]
        power = (spec.real ** 2 + spec.imag ** 2).T.numpy()                                # [T][bins]
        xp = np.pad(x[0].numpy(), n_fft // 2, mode="reflect")
        T = 1 + x.shape[1] // hop
        frames = np.stack([xp[t * hop: t * hop + n_fft] for t in range(T)]) * d["cond.win_" + tag][None, :]
        sp = frames.astype(np.float64) @ d["cond.dft_" + tag].astype(np.float64)
        bins = n_fft // 2 + 1
        got = sp[:, 0: 2 * bins: 2] ** 2 + sp[:, 1: 2 * bins: 2] ** 2
        assert got.shape == power.shape and np.abs(got - power).max() <= 2e-4 * power.max()
        assert np.abs(sp[:, 2 * bins:]).max() == 0.0 and d["cond.dft_" + tag].shape[1] % 128 == 0
        fb = d["cond.fb_" + tag]
        assert fb.shape[0] % 16 == 0 and fb.shape[1] == 128 and np.abs(fb[bins:]).max() == 0.0
    ref = Cn.resample(x, 22050, 16000)[0].numpy()
    tab = d["cond.rs_22050_16000"]
    assert tab.shape == (320, 459)
    xin = np.pad(x[0].numpy().astype(np.float64), (9, 9 + 441))
    n_out = ref.shape[0]
    got = np.array([np.dot(xin[(o // 320) * 441: (o // 320) * 441 + 459], tab[o % 320]) for o in range(n_out)])
    assert np.abs(got - ref).max() < 1e-5
