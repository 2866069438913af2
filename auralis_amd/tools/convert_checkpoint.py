"""Offline converter: Coqui XTTSv2 training checkpoint (`model.pth`, state["model"]) -> the two-safetensors layout the
engines load (same on-disk format as the reference's converter, src/auralis/models/xttsv2/utils/checkpoint_converter.py:
225-347; pinned by the expectations of the reference's tests/integration/test_checkpoint_converter.py:140-347).

    python -m auralis_amd.tools.convert_checkpoint model.pth out_dir

Rules (first match wins, after dropping training-only modules and a leading "xtts."):
    gpt.mel_embedding.weight          -> gpt file  gpt.wte.weight
    gpt.mel_pos_embedding.emb.weight  -> gpt file  gpt.wpe.emb.weight
    gpt.mel_head.*                    -> gpt file  mel_head.*
    gpt.gpt.h.N.* / gpt.gpt.ln_f.*    -> gpt file  gpt.h.N.* / gpt.ln_f.*
    gpt.final_norm.*                  -> BOTH files final_norm.*   (the reference applies it in both engines)
    anything else                     -> xtts file, "gpt." prefix removed (text_embedding, conditioning_*, hifigan_decoder.*)
"""
from __future__ import annotations

import argparse
import json
import os
import re
from typing import Any, Dict, Tuple

import torch

TRAINING_ONLY = ("dvae", "torch_mel_spectrogram_style_encoder", "torch_mel_spectrogram_dvae")
_BLOCK = re.compile(r"^gpt\.gpt\.(h\.\d+\..+|ln_f\.(weight|bias))$")
REQUIRED_GPT = ("gpt.wte.weight", "gpt.wpe.emb.weight", "gpt.ln_f.weight", "mel_head.weight", "final_norm.weight",
                "gpt.h.0.attn.c_attn.weight", "gpt.h.0.mlp.c_fc.weight")


def split_state_dict(model_state: Dict[str, torch.Tensor]) -> Tuple[Dict[str, torch.Tensor], Dict[str, torch.Tensor]]:
    gpt: Dict[str, torch.Tensor] = {}
    xtts: Dict[str, torch.Tensor] = {}
    for key, t in model_state.items():
        if any(p in key for p in TRAINING_ONLY):
            continue
        k = key[5:] if key.startswith("xtts.") else key
        if k == "gpt.mel_embedding.weight":
            gpt["gpt.wte.weight"] = t
        elif k == "gpt.mel_pos_embedding.emb.weight":
            gpt["gpt.wpe.emb.weight"] = t
        elif k.startswith("gpt.mel_head."):
            gpt[k[4:]] = t
        elif _BLOCK.match(k):
            gpt["gpt." + k[len("gpt.gpt."):]] = t
        elif k.startswith("gpt.final_norm."):
            gpt[k[4:]] = t
            xtts[k[4:]] = t
        else:
            xtts[k[4:] if k.startswith("gpt.") else k] = t
    missing = [r for r in REQUIRED_GPT if r not in gpt]
    if missing:
        raise ValueError(f"checkpoint lacks GPT tensors: {missing}")
    return gpt, xtts


def infer_architecture(gpt: Dict[str, torch.Tensor], xtts: Dict[str, torch.Tensor]) -> Dict[str, Any]:
    n_audio, hidden = gpt["mel_head.weight"].shape
    layers = 1 + max(int(k.split(".")[2]) for k in gpt if k.startswith("gpt.h."))
    arch = {
        "hidden_size": int(hidden), "num_hidden_layers": layers,
        "num_attention_heads": int(hidden) // 64 if hidden % 64 == 0 else 1,
        "n_inner": int(gpt["gpt.h.0.mlp.c_fc.weight"].shape[1]),
        "num_audio_tokens": int(n_audio), "start_audio_token": int(n_audio) - 2, "stop_audio_token": int(n_audio) - 1,
        "max_audio_tokens": int(gpt["gpt.wpe.emb.weight"].shape[0]) - 3,
    }
    if "text_embedding.weight" in xtts:
        arch["number_text_tokens"] = int(xtts["text_embedding.weight"].shape[0])
    if "text_pos_embedding.emb.weight" in xtts:
        arch["max_text_tokens"] = int(xtts["text_pos_embedding.emb.weight"].shape[0]) - 2
    return arch


def convert_checkpoint(checkpoint_path: str, output_dir: str) -> Dict[str, Any]:
    from safetensors.torch import save_file
    ckpt = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
    state = ckpt["model"] if isinstance(ckpt, dict) and "model" in ckpt else ckpt
    gpt, xtts = split_state_dict(state)
    arch = infer_architecture(gpt, xtts)
    os.makedirs(os.path.join(output_dir, "gpt"), exist_ok=True)
    os.makedirs(os.path.join(output_dir, "core_xttsv2"), exist_ok=True)
    save_file({k: v.contiguous() for k, v in gpt.items()}, os.path.join(output_dir, "gpt", "gpt2_model.safetensors"))
    save_file({k: v.contiguous() for k, v in xtts.items()}, os.path.join(output_dir, "core_xttsv2", "xtts-v2.safetensors"))
    with open(os.path.join(output_dir, "gpt", "config.json"), "w") as f:
        json.dump({"model_type": "xtts_gpt", "activation_function": "gelu_new", **arch}, f, indent=1)
    with open(os.path.join(output_dir, "core_xttsv2", "config.json"), "w") as f:
        json.dump({"model_type": "xtts", "gpt_config": arch}, f, indent=1)
    return arch


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("checkpoint")
    ap.add_argument("output_dir")
    a = ap.parse_args()
    print(json.dumps(convert_checkpoint(a.checkpoint, a.output_dir), indent=1))


if __name__ == "__main__":
    main()
