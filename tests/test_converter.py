"""CPU: checkpoint converter — same expectations the reference's own converter tests pin
(/root/reference/tests/integration/test_checkpoint_converter.py:140-347: architecture inference 1026 -> start 1024 /
stop 1025 / max 605, key renames, final_norm in both files, file names), on a random Coqui-shaped state dict."""
import json
import os

import torch

from auralis_amd.checkpoint import load_checkpoint
from auralis_amd.tools.convert_checkpoint import convert_checkpoint, infer_architecture, split_state_dict


def _coqui_state(hidden=128, layers=2, n_audio=1026, n_text=6153, mel_pos=608):
    s = {"gpt.text_embedding.weight": torch.randn(n_text, hidden), "gpt.text_pos_embedding.emb.weight": torch.randn(404, hidden),
         "gpt.mel_embedding.weight": torch.randn(n_audio, hidden), "gpt.mel_pos_embedding.emb.weight": torch.randn(mel_pos, hidden),
         "gpt.final_norm.weight": torch.randn(hidden), "gpt.final_norm.bias": torch.randn(hidden),
         "gpt.mel_head.weight": torch.randn(n_audio, hidden), "gpt.mel_head.bias": torch.randn(n_audio),
         "gpt.gpt.ln_f.weight": torch.randn(hidden), "gpt.gpt.ln_f.bias": torch.randn(hidden),
         "gpt.conditioning_encoder.init.weight": torch.randn(hidden, 80, 1),
         "hifigan_decoder.waveform_decoder.conv_post.weight": torch.randn(1, 32, 7),
         "mel_stats": torch.ones(80),
         "dvae.encoder.weight": torch.randn(4, 4), "torch_mel_spectrogram_dvae.mel_stft.window": torch.randn(8)}
    for i in range(layers):
        p = f"gpt.gpt.h.{i}."
        for n, shape in (("ln_1.weight", (hidden,)), ("ln_1.bias", (hidden,)), ("attn.c_attn.weight", (hidden, 3 * hidden)),
                         ("attn.c_attn.bias", (3 * hidden,)), ("attn.c_proj.weight", (hidden, hidden)), ("attn.c_proj.bias", (hidden,)),
                         ("ln_2.weight", (hidden,)), ("ln_2.bias", (hidden,)), ("mlp.c_fc.weight", (hidden, 4 * hidden)),
                         ("mlp.c_fc.bias", (4 * hidden,)), ("mlp.c_proj.weight", (4 * hidden, hidden)), ("mlp.c_proj.bias", (hidden,))):
            s[p + n] = torch.randn(*shape)
    return {("xtts." + k): v for k, v in s.items()}


def test_split_and_renames():
    gpt, xtts = split_state_dict(_coqui_state())
    for k in ("gpt.wte.weight", "gpt.wpe.emb.weight", "gpt.h.0.attn.c_attn.weight", "gpt.h.1.mlp.c_proj.bias", "gpt.ln_f.weight",
              "mel_head.weight", "mel_head.bias", "final_norm.weight", "final_norm.bias"):
        assert k in gpt, k
    assert "final_norm.weight" in xtts and torch.equal(xtts["final_norm.weight"], gpt["final_norm.weight"])
    assert "text_embedding.weight" in xtts and "conditioning_encoder.init.weight" in xtts and "mel_stats" in xtts
    assert "hifigan_decoder.waveform_decoder.conv_post.weight" in xtts
    assert not any("dvae" in k for k in list(gpt) + list(xtts))
    assert not any(k.startswith("gpt.gpt.") for k in gpt)


def test_architecture_inference():
    gpt, xtts = split_state_dict(_coqui_state(hidden=128, layers=2))
    a = infer_architecture(gpt, xtts)
    assert (a["num_audio_tokens"], a["start_audio_token"], a["stop_audio_token"], a["max_audio_tokens"]) == (1026, 1024, 1025, 605)
    assert (a["hidden_size"], a["num_hidden_layers"], a["num_attention_heads"], a["number_text_tokens"]) == (128, 2, 2, 6153)


def test_convert_writes_loadable_directory(tmp_path):
    ck = tmp_path / "model.pth"
    torch.save({"model": _coqui_state()}, ck)
    arch = convert_checkpoint(str(ck), str(tmp_path / "out"))
    assert os.path.isfile(tmp_path / "out" / "gpt" / "gpt2_model.safetensors")
    assert os.path.isfile(tmp_path / "out" / "core_xttsv2" / "xtts-v2.safetensors")
    assert json.load(open(tmp_path / "out" / "core_xttsv2" / "config.json"))["model_type"] == "xtts"
    gpt, xtts = load_checkpoint(str(tmp_path / "out"))
    assert gpt["gpt.wte.weight"].shape == (1026, 128) and "text_embedding.weight" in xtts and arch["num_hidden_layers"] == 2


def test_missing_gpt_tensors_are_reported():
    s = _coqui_state()
    del s["xtts.gpt.mel_head.weight"]
    try:
        split_state_dict(s)
    except ValueError as e:
        assert "mel_head.weight" in str(e)
    else:
        raise AssertionError("expected ValueError")
