"""CPU: the checkpoint's two config.json are consumed (reference: XTTSv2.py:276-277 builds XTTSGPTConfig / XTTSConfig from them;
class defaults xttsv2_gpt_config.py:133-186, converter output checkpoint_converter.py:176-223) and checked against what the
HIP kernels are compiled for."""
import json
import os

import pytest

from auralis_amd.checkpoint import (CheckpointConfigError, make_synthetic_gpt, make_synthetic_xtts, read_checkpoint_config,
                                    save_checkpoint)
from auralis_amd.config import XTTSDims


@pytest.fixture()
def ckpt(tmp_path):
    dims = XTTSDims()
    gpt_sd = make_synthetic_gpt(dims.gpt, seed=1, n_layer=1)
    xtts_sd = make_synthetic_xtts(dims, seed=1, gpt_sd=gpt_sd)
    root = str(tmp_path / "ck")
    save_checkpoint(root, gpt_sd, xtts_sd, dims, synthetic_tokenizer=True)
    return root, gpt_sd


def _edit(root, rel, **kv):
    p = os.path.join(root, rel)
    d = json.load(open(p))
    for k, v in kv.items():
        if v is None:
            d.pop(k, None)
        else:
            d[k] = v
    json.dump(d, open(p, "w"))


def test_converter_style_config_is_accepted(ckpt):
    root, gpt_sd = ckpt
    c = read_checkpoint_config(root, gpt_sd)
    assert c.n_layer == 1 and c.activation == "gelu_new" and not c.gelu_erf and c.gpt_max_audio_tokens == 605


def test_activation_gelu_selects_the_erf_kernels_and_absent_key_means_class_default(ckpt):
    root, gpt_sd = ckpt
    _edit(root, "gpt/config.json", activation_function="gelu")
    assert read_checkpoint_config(root, gpt_sd).gelu_erf
    _edit(root, "gpt/config.json", activation_function=None)          # XTTSGPTConfig default is "gelu" (erf)
    assert read_checkpoint_config(root, gpt_sd).activation == "gelu"
    _edit(root, "gpt/config.json", activation_function="gelu_pytorch_tanh")
    assert read_checkpoint_config(root, gpt_sd).activation == "gelu_new"
    _edit(root, "gpt/config.json", activation_function="relu")
    with pytest.raises(CheckpointConfigError, match="activation_function"):
        read_checkpoint_config(root, gpt_sd)


@pytest.mark.parametrize("key,value", [("num_audio_tokens", 1025), ("start_audio_token", 1023), ("stop_audio_token", 1026),
                                       ("hidden_size", 768), ("num_attention_heads", 12), ("n_inner", 3072),
                                       ("layer_norm_epsilon", 1e-6), ("max_audio_tokens", 800), ("gpt_max_audio_tokens", 700),
                                       ("max_text_tokens", 500), ("num_hidden_layers", 2)])
def test_mismatch_with_the_compiled_model_is_refused_naming_the_key(ckpt, key, value):
    root, gpt_sd = ckpt
    _edit(root, "gpt/config.json", **{key: value})
    with pytest.raises(CheckpointConfigError, match=key):
        read_checkpoint_config(root, gpt_sd)


def test_generation_length_and_languages_come_from_the_files(ckpt):
    root, gpt_sd = ckpt
    _edit(root, "gpt/config.json", gpt_max_audio_tokens=300)
    _edit(root, "core_xttsv2/config.json", languages=["en", "fr"], output_sample_rate=24000)
    c = read_checkpoint_config(root, gpt_sd)
    assert c.gpt_max_audio_tokens == 300 and c.languages == ["en", "fr"]
    _edit(root, "core_xttsv2/config.json", output_sample_rate=22050)
    with pytest.raises(CheckpointConfigError, match="output_sample_rate"):
        read_checkpoint_config(root, gpt_sd)


def test_missing_gpt_config_is_an_error(ckpt):
    root, gpt_sd = ckpt
    os.remove(os.path.join(root, "gpt", "config.json"))
    with pytest.raises(FileNotFoundError):
        read_checkpoint_config(root, gpt_sd)
