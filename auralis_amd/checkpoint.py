"""On-disk weight format of the reference (two safetensors + configs) and a seeded synthetic writer.

Format defined by src/auralis/models/xttsv2/utils/checkpoint_converter.py:225-284 and consumed at
src/auralis/models/xttsv2/XTTSv2.py:288-301 / components/vllm_mm_gpt.py:714-733:

  <dir>/gpt/gpt2_model.safetensors        gpt.wte.weight, gpt.wpe.emb.weight, gpt.h.N.*, gpt.ln_f.*,
                                          final_norm.*, mel_head.*   (HF Conv1D [in,out] layout)
  <dir>/core_xttsv2/xtts-v2.safetensors   text_embedding, text_pos_embedding, final_norm,
                                          hifigan_decoder.waveform_decoder.* (weight-norm g/v pairs), ...

Real checkpoints (AstraMindAI/xttsv2, AstraMindAI/xtts2-gpt) are not available offline; every test
and the bench use `make_synthetic_weights` (seed 1234).
"""
from __future__ import annotations

import json
import math
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch

from .config import GPTDims, VocoderDims, XTTSDims

Tensor = torch.Tensor
VOC_PREFIX = "hifigan_decoder.waveform_decoder."


def _conv_default_init(gen: torch.Generator, shape, fan_in: int) -> Tensor:
    b = 1.0 / math.sqrt(fan_in)
    return (torch.rand(shape, generator=gen, dtype=torch.float32) * 2.0 - 1.0) * b


def make_synthetic_gpt(dims: GPTDims, seed: int = 1234, n_layer: Optional[int] = None) -> Dict[str, Tensor]:
    """gpt2_model.safetensors contents: N(0, 0.02) matrices/embeddings, LayerNorm gamma=1 beta=0."""
    g = torch.Generator().manual_seed(seed)
    H, F_ = dims.hidden, dims.n_inner
    L = dims.n_layer if n_layer is None else n_layer

    def rn(*shape):
        return torch.randn(*shape, generator=g, dtype=torch.float32) * 0.02

    sd: Dict[str, Tensor] = {
        "gpt.wte.weight": rn(dims.mel_vocab, H),
        "gpt.wpe.emb.weight": rn(dims.mel_positions, H),
    }
    for i in range(L):
        p = f"gpt.h.{i}."
        sd[p + "ln_1.weight"] = torch.ones(H)
        sd[p + "ln_1.bias"] = torch.zeros(H)
        sd[p + "attn.c_attn.weight"] = rn(H, 3 * H)
        sd[p + "attn.c_attn.bias"] = rn(3 * H)
        sd[p + "attn.c_proj.weight"] = rn(H, H)
        sd[p + "attn.c_proj.bias"] = rn(H)
        sd[p + "ln_2.weight"] = torch.ones(H)
        sd[p + "ln_2.bias"] = torch.zeros(H)
        sd[p + "mlp.c_fc.weight"] = rn(H, F_)
        sd[p + "mlp.c_fc.bias"] = rn(F_)
        sd[p + "mlp.c_proj.weight"] = rn(F_, H)
        sd[p + "mlp.c_proj.bias"] = rn(H)
    sd["gpt.ln_f.weight"] = torch.ones(H)
    sd["gpt.ln_f.bias"] = torch.zeros(H)
    # non-trivial affine so that final_norm∘final_norm is actually exercised
    sd["final_norm.weight"] = 1.0 + rn(H) * 5.0
    sd["final_norm.bias"] = rn(H)
    sd["mel_head.weight"] = rn(dims.mel_vocab, H)
    sd["mel_head.bias"] = rn(dims.mel_vocab)
    return sd


def make_synthetic_xtts(dims: XTTSDims, seed: int = 1234, gpt_sd: Optional[Dict[str, Tensor]] = None) -> Dict[str, Tensor]:
    """xtts-v2.safetensors contents needed by the hot path (text embeddings + vocoder).

    Vocoder convs use torch's default conv init bound 1/sqrt(fan_in) (N(0,0.02) gives a near-silent
    waveform); weight-norm g = ||v|| so the effective weight equals v at init.
    """
    gd, vd = dims.gpt, dims.voc
    g = torch.Generator().manual_seed(seed + 1)
    H = gd.hidden
    sd: Dict[str, Tensor] = {
        "mel_stats": torch.ones(80),
        "text_embedding.weight": torch.randn(gd.text_vocab, H, generator=g) * 0.02,
        "text_pos_embedding.emb.weight": torch.randn(gd.text_positions, H, generator=g) * 0.02,
    }
    if gpt_sd is not None:
        sd["final_norm.weight"] = gpt_sd["final_norm.weight"].clone()
        sd["final_norm.bias"] = gpt_sd["final_norm.bias"].clone()
    p = VOC_PREFIX
    C0 = vd.initial_channel
    sd[p + "conv_pre.weight"] = _conv_default_init(g, (C0, vd.in_dim, 7), vd.in_dim * 7)
    sd[p + "conv_pre.bias"] = _conv_default_init(g, (C0,), vd.in_dim * 7)
    sd[p + "cond_layer.weight"] = _conv_default_init(g, (C0, vd.d_vector, 1), vd.d_vector)
    sd[p + "cond_layer.bias"] = _conv_default_init(g, (C0,), vd.d_vector)
    chans = vd.stage_channels()
    cin = C0
    for i, (s, k, c) in enumerate(zip(vd.upsample_rates, vd.upsample_kernels, chans)):
        v = _conv_default_init(g, (cin, c, k), c * k)   # ConvTranspose1d: fan_in computed on dim 1
        sd[p + f"ups.{i}.parametrizations.weight.original1"] = v
        sd[p + f"ups.{i}.parametrizations.weight.original0"] = v.flatten(1).norm(dim=1).view(cin, 1, 1).clone()
        sd[p + f"ups.{i}.bias"] = _conv_default_init(g, (c,), c * k)
        sd[p + f"conds.{i}.weight"] = _conv_default_init(g, (c, vd.d_vector, 1), vd.d_vector)
        sd[p + f"conds.{i}.bias"] = _conv_default_init(g, (c,), vd.d_vector)
        for j, rk in enumerate(vd.resblock_kernels):
            rb = p + f"resblocks.{i * len(vd.resblock_kernels) + j}."
            for grp in ("convs1", "convs2"):
                for q in range(len(vd.resblock_dilations)):
                    v = _conv_default_init(g, (c, c, rk), c * rk)
                    sd[rb + f"{grp}.{q}.parametrizations.weight.original1"] = v
                    # perturb g a little so folding g*v/||v|| is really exercised
                    gn = v.flatten(1).norm(dim=1).view(c, 1, 1)
                    sd[rb + f"{grp}.{q}.parametrizations.weight.original0"] = gn * (
                        1.0 + 0.1 * (torch.rand(c, 1, 1, generator=g) - 0.5))
                    sd[rb + f"{grp}.{q}.bias"] = _conv_default_init(g, (c,), c * rk)
        cin = c
    sd[p + "conv_post.weight"] = _conv_default_init(g, (1, cin, 7), cin * 7)
    sd.update(make_synthetic_conditioning_weights(dims, seed=seed + 2))
    return sd


def make_loud_vocoder(xtts_sd: Dict[str, Tensor], up_gain: float = 3.0, post_gain: float = 1.5) -> Dict[str, Tensor]:
    """The synthetic vocoder at speech amplitude.  Default conv init gives a quiet waveform (RMS ~0.04), where the "1 % of the
    signal" bar binds before the north-star 1e-3 absolute one; here the weight-norm gains of the four transposed convs are raised
    (every stage's activations grow by `up_gain`) and `conv_post` by `post_gain`, which brings the output to the RMS of real
    speech (0.12 with the defaults, peaks 0.5) and the fp16-stored residual stream / MRF sums to correspondingly larger magnitudes.  Returns a copy."""
    sd = {k: v.clone() for k, v in xtts_sd.items()}
    for i in range(4):
        sd[VOC_PREFIX + f"ups.{i}.parametrizations.weight.original0"] *= up_gain
    sd[VOC_PREFIX + "conv_post.weight"] *= post_gain
    return sd


def conditioning_param_shapes(dims: XTTSDims) -> Dict[str, tuple]:
    """Names/shapes of the once-per-speaker modules inside xtts-v2.safetensors:
    conditioning_encoder (Conv1d 80->1024 + 6 attention blocks), conditioning_perceiver (32 latents, 2 x (cross-attn
    8x64 + GEGLU FF 2730)), hifigan_decoder.speaker_encoder (ResNet-SE [3,4,6,3] x [32,64,128,256], ASP, fc 512)."""
    H = dims.gpt.hidden
    out: Dict[str, tuple] = {"conditioning_encoder.init.weight": (H, 80, 1), "conditioning_encoder.init.bias": (H,)}
    for i in range(6):
        p = f"conditioning_encoder.attn.{i}."
        out.update({p + "norm.weight": (H,), p + "norm.bias": (H,), p + "qkv.weight": (3 * H, H, 1),
                    p + "qkv.bias": (3 * H,), p + "proj_out.weight": (H, H, 1), p + "proj_out.bias": (H,)})
    out["conditioning_perceiver.latents"] = (32, H)
    inner = int(H * 4 * 2 / 3)
    for i in range(2):
        p = f"conditioning_perceiver.layers.{i}."
        out.update({p + "0.to_q.weight": (512, H), p + "0.to_kv.weight": (1024, H), p + "0.to_out.weight": (H, 512),
                    p + "1.0.weight": (2 * inner, H), p + "1.0.bias": (2 * inner,), p + "1.2.weight": (H, inner),
                    p + "1.2.bias": (H,)})
    out["conditioning_perceiver.norm.gamma"] = (H,)
    s = "hifigan_decoder.speaker_encoder."

    def bn(prefix, c):
        out.update({prefix + "weight": (c,), prefix + "bias": (c,), prefix + "running_mean": (c,),
                    prefix + "running_var": (c,), prefix + "num_batches_tracked": ()})
    out[s + "torch_spec.0.filter"] = (1, 1, 2)     # PreEmphasis buffer; the torchaudio mel buffers are recomputed
    out[s + "conv1.weight"] = (32, 1, 3, 3)
    out[s + "conv1.bias"] = (32,)
    bn(s + "bn1.", 32)
    inpl = 32
    for li, (planes, blocks) in enumerate(zip((32, 64, 128, 256), (3, 4, 6, 3)), start=1):
        for bi in range(blocks):
            p = s + f"layer{li}.{bi}."
            out[p + "conv1.weight"] = (planes, inpl if bi == 0 else planes, 3, 3)
            bn(p + "bn1.", planes)
            out[p + "conv2.weight"] = (planes, planes, 3, 3)
            bn(p + "bn2.", planes)
            out.update({p + "se.fc.0.weight": (planes // 8, planes), p + "se.fc.0.bias": (planes // 8,),
                        p + "se.fc.2.weight": (planes, planes // 8), p + "se.fc.2.bias": (planes,)})
            if bi == 0 and (li > 1):
                out[p + "downsample.0.weight"] = (planes, inpl, 1, 1)
                bn(p + "downsample.1.", planes)
        inpl = planes
    out.update({s + "attention.0.weight": (128, 2048, 1), s + "attention.0.bias": (128,)})
    bn(s + "attention.2.", 128)
    out.update({s + "attention.3.weight": (2048, 128, 1), s + "attention.3.bias": (2048,),
                s + "fc.weight": (512, 4096), s + "fc.bias": (512,)})
    return out


def make_synthetic_conditioning_weights(dims: XTTSDims, seed: int = 1236) -> Dict[str, Tensor]:
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, Tensor] = {}
    for name, shape in conditioning_param_shapes(dims).items():
        if name.endswith("num_batches_tracked"):
            sd[name] = torch.tensor(0, dtype=torch.long)
        elif name.endswith("torch_spec.0.filter"):
            sd[name] = torch.tensor([-0.97, 1.0]).view(1, 1, 2)
        elif name.endswith("running_var"):
            sd[name] = 0.5 + torch.rand(shape, generator=g)
        elif name.endswith(("norm.weight", "bn1.weight", "bn2.weight", "downsample.1.weight", "attention.2.weight", "norm.gamma")):
            sd[name] = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif len(shape) <= 1:
            sd[name] = 0.05 * torch.randn(shape, generator=g)
        else:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            sd[name] = torch.randn(shape, generator=g) / math.sqrt(fan_in)
    return sd


def make_synthetic_conditioning(dims: XTTSDims, seed_cond: int = 7, seed_spk: int = 8) -> Tuple[Tensor, Tensor]:
    """gpt_cond_latent [1,32,1024] ~ N(0,1)*0.02 (seed 7); speaker_embedding = L2-normalised N(0,1) [1,512,1] (seed 8)."""
    g1 = torch.Generator().manual_seed(seed_cond)
    cond = torch.randn(1, dims.gpt.perceiver_latents, dims.gpt.hidden, generator=g1) * 0.02
    g2 = torch.Generator().manual_seed(seed_spk)
    spk = torch.randn(1, dims.voc.d_vector, 1, generator=g2)
    spk = spk / spk.norm()
    return cond, spk


def make_synthetic_text_ids(dims: XTTSDims, n_text: int = 70, seed: int = 11):
    """[START] + ids ~ U[0, text_vocab) + [STOP]  (start/stop text ids 261/0 per XTTSv2.py:520-521)."""
    g = torch.Generator().manual_seed(seed)
    body = torch.randint(0, dims.gpt.text_vocab, (n_text - 2,), generator=g).tolist()
    return [261] + body + [0]


def save_checkpoint(root: str, gpt_sd: Dict[str, Tensor], xtts_sd: Dict[str, Tensor], dims: XTTSDims,
                    synthetic_tokenizer: bool = False, gpt_max_audio_tokens: Optional[int] = None) -> None:
    """Write the two safetensors + configs.  synthetic_tokenizer=True marks the directory as a seeded synthetic checkpoint
    whose text ids come from the stand-in vocabulary (api/text.py); a real checkpoint must carry tokenizer.json instead.
    gpt_max_audio_tokens: the generation length XTTSConfig carries (max_tokens of every request, XTTSv2.py:735; default 605)."""
    from safetensors.torch import save_file
    os.makedirs(os.path.join(root, "gpt"), exist_ok=True)
    os.makedirs(os.path.join(root, "core_xttsv2"), exist_ok=True)
    save_file({k: v.contiguous() for k, v in gpt_sd.items()}, os.path.join(root, "gpt", "gpt2_model.safetensors"))
    save_file({k: v.contiguous() for k, v in xtts_sd.items()}, os.path.join(root, "core_xttsv2", "xtts-v2.safetensors"))
    n_layer = 1 + max(int(k.split(".")[2]) for k in gpt_sd if k.startswith("gpt.h."))
    with open(os.path.join(root, "gpt", "config.json"), "w") as f:
        json.dump({"model_type": "xtts_gpt", "hidden_size": dims.gpt.hidden, "num_hidden_layers": n_layer,
                   "num_attention_heads": dims.gpt.n_head, "n_inner": dims.gpt.n_inner,
                   "num_audio_tokens": dims.gpt.mel_vocab, "start_audio_token": dims.gpt.start_token,
                   "stop_audio_token": dims.gpt.stop_token, "max_audio_tokens": dims.gpt.max_audio_tokens,
                   "activation_function": dims.gpt.activation,
                   **({"gpt_max_audio_tokens": int(gpt_max_audio_tokens)} if gpt_max_audio_tokens is not None else {})}, f)
    with open(os.path.join(root, "core_xttsv2", "config.json"), "w") as f:
        json.dump({"model_type": "xtts", "gpt_config": {"num_hidden_layers": n_layer},
                   **({"synthetic_tokenizer": True} if synthetic_tokenizer else {})}, f)


@dataclass
class CheckpointConfig:
    """What the engine takes from the checkpoint's two config.json (the reference builds XTTSGPTConfig / XTTSConfig from them,
    XTTSv2.py:276-277; defaults are the config CLASS defaults, xttsv2_gpt_config.py:133-186, xttsv2_config.py:212-260)."""
    n_layer: int = 30
    activation: str = "gelu"        # XTTSGPTConfig default; checkpoint_converter.py:197 writes "gelu_new"
    gpt_max_audio_tokens: int = 605  # max_tokens of every generation (XTTSv2.py:735)
    max_text_tokens: int = 402
    languages: Optional[List[str]] = None

    @property
    def gelu_erf(self) -> bool:
        return self.activation == "gelu"


class CheckpointConfigError(ValueError):
    """The checkpoint describes a model the compiled kernels cannot run (AUR_E_INVALID at the Python boundary)."""


def _load_json(path: str) -> Optional[dict]:
    if not os.path.isfile(path):
        return None
    with open(path) as f:
        return json.load(f)


def read_checkpoint_config(root: str, gpt_sd: Optional[Dict[str, Tensor]] = None) -> CheckpointConfig:
    """gpt/config.json (XTTSGPTConfig) and core_xttsv2/config.json (XTTSConfig) -> CheckpointConfig.  Every dimension the HIP
    kernels are compiled for is checked against the file; a mismatch raises CheckpointConfigError naming the key (the kernels
    would otherwise run a different model silently).  The activation is honoured: "gelu_new" / "gelu_pytorch_tanh" = tanh
    form, "gelu" = erf form (aur_config.gelu_erf); anything else is refused."""
    g = GPTDims()
    gj = _load_json(os.path.join(root, "gpt", "config.json"))
    xj = _load_json(os.path.join(root, "core_xttsv2", "config.json"))
    if gj is None:
        raise FileNotFoundError(f"{os.path.join(root, 'gpt', 'config.json')}: the GPT config of the checkpoint is required "
                                "(activation_function, token ids and sizes are read from it)")
    nested = (xj or {}).get("gpt_config") or {}

    def get(key, default):
        v = gj.get(key, nested.get(key))
        return default if v is None else v   # an explicit JSON null means "not given"

    def need(key, default, want, what):
        v = get(key, default)
        if v is None:
            v = want
        if v != want:
            raise CheckpointConfigError(f"gpt/config.json: {key} = {v!r}, but the MI355X kernels are built for {what} = {want!r}")

    need("hidden_size", 1024, g.hidden, "hidden size")
    need("num_attention_heads", 16, g.n_head, "attention heads")
    n_inner = get("n_inner", 4096)
    if (n_inner if n_inner is not None else 4 * g.hidden) != g.n_inner:
        raise CheckpointConfigError(f"gpt/config.json: n_inner = {n_inner!r}, but the MI355X kernels are built for {g.n_inner}")
    need("num_audio_tokens", 1026, g.mel_vocab, "mel vocabulary")
    need("start_audio_token", 1024, g.start_token, "start id")
    need("stop_audio_token", 1025, g.stop_token, "stop id")
    eps = float(get("layer_norm_epsilon", 1e-5))
    if abs(eps - g.ln_eps) > 1e-12:
        raise CheckpointConfigError(f"gpt/config.json: layer_norm_epsilon = {eps!r}, kernels use {g.ln_eps!r}")
    act = get("activation_function", None)
    if act is None:   # neither config names it: XTTSGPTConfig's class default (xttsv2_gpt_config.py:184), and say so
        import logging
        logging.getLogger("auralis_amd").warning("gpt/config.json has no activation_function: using the XTTSGPTConfig default 'gelu' "
                                                 "(erf form); checkpoints written by the reference's converter say 'gelu_new'")
        act = "gelu"
    if act in ("gelu_new", "gelu_pytorch_tanh"):
        act = "gelu_new"
    elif act != "gelu":
        raise CheckpointConfigError(f"gpt/config.json: activation_function = {act!r}; supported: 'gelu_new' (tanh form) and 'gelu' (erf form)")
    max_audio = int(get("max_audio_tokens", 605))
    gen_max = int(get("gpt_max_audio_tokens", max_audio))
    if max_audio + 3 > g.mel_positions or gen_max > g.max_audio_tokens:
        raise CheckpointConfigError(f"gpt/config.json: max_audio_tokens = {max_audio} / gpt_max_audio_tokens = {gen_max}; the engine "
                                    f"keeps {g.mel_positions} mel positions and {g.max_audio_tokens} latent rows per sequence")
    max_text = int(get("max_text_tokens", 402))
    if max_text + 2 > g.text_positions:
        raise CheckpointConfigError(f"gpt/config.json: max_text_tokens = {max_text}; the engine keeps {g.text_positions} text positions")
    n_layer = int(get("num_hidden_layers", 30))
    if gpt_sd is not None:
        have = 1 + max(int(k.split(".")[2]) for k in gpt_sd if k.startswith("gpt.h."))
        if have != n_layer:
            raise CheckpointConfigError(f"gpt/config.json: num_hidden_layers = {n_layer}, gpt2_model.safetensors holds {have} blocks")
    if xj is not None:
        v = VocoderDims()
        for key, want in (("input_sample_rate", v.input_sample_rate), ("output_sample_rate", v.output_sample_rate),
                          ("output_hop_length", v.output_hop_length), ("decoder_input_dim", v.in_dim), ("d_vector_dim", v.d_vector),
                          ("gpt_code_stride_len", v.ar_mel_length_compression)):
            if key in xj and xj[key] != want:
                raise CheckpointConfigError(f"core_xttsv2/config.json: {key} = {xj[key]!r}, but the vocoder kernels are built for {want!r}")
    return CheckpointConfig(n_layer=n_layer, activation=act, gpt_max_audio_tokens=gen_max, max_text_tokens=max_text,
                            languages=(xj or {}).get("languages"))


def load_checkpoint(root: str) -> Tuple[Dict[str, Tensor], Dict[str, Tensor]]:
    from safetensors.torch import load_file
    gpt_sd = load_file(os.path.join(root, "gpt", "gpt2_model.safetensors"))
    xtts_sd = load_file(os.path.join(root, "core_xttsv2", "xtts-v2.safetensors"))
    return gpt_sd, xtts_sd
