"""Run the reference's own GPT glue (vllm_mm_gpt.py: GPT2Model.forward, LearnedPositionEmbeddings,
_insert_conditioning_into_hidden_states) and its repetition-penalty processor (vllm/hijack.py) UNMODIFIED from
/root/reference, in the build container.

TEST INFRASTRUCTURE ONLY.  Both files import vllm (0.6.4.post1, not installed, not vendored), networkx and triton at module
level.  Those imports are satisfied by stub modules; everything the reference *defines* runs as written.  The one thing the
reference takes from vllm that does arithmetic on this path is `vllm.model_executor.models.gpt2.GPT2Block`; it is replaced by
`StandInGPT2Block`, the textbook GPT-2 block assembled from transformers' own Conv1D / gelu_new modules (the same arithmetic
oracle/xtts_oracle.py restates and tests/test_oracle_gpt.py checks against transformers.GPT2Model).  So the fixtures made from
this module pin the oracle's *glue* (embedding sums, start-token position, conditioning splice order, ln_f, penalty rule) to
the reference's code; the block arithmetic stays "cross-checked against transformers", as the header of xtts_oracle.py says.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import torch
import torch.nn as nn

REF_ROOT = os.environ.get("AURALIS_REFERENCE", "/root/reference")
_GPT = "src/auralis/models/xttsv2/components/vllm_mm_gpt.py"
_HIJACK = "src/auralis/models/xttsv2/components/vllm/hijack.py"


def reference_gpt_available() -> bool:
    return os.path.isfile(os.path.join(REF_ROOT, _GPT))


class StandInGPT2Block(nn.Module):
    """x += c_proj(attn(split(c_attn(ln_1 x)))); x += mlp.c_proj(gelu_new(mlp.c_fc(ln_2 x))), causal softmax(QK^T/8)V over the
    rows it is given (one sequence).  Signature of vllm's GPT2Block: (config, cache_config, quant_config, prefix) /
    forward(hidden_states, kv_cache, attn_metadata)."""

    def __init__(self, config, cache_config=None, quant_config=None, prefix: str = ""):
        super().__init__()
        from transformers.activations import ACT2FN
        from transformers.pytorch_utils import Conv1D
        H = config.hidden_size
        inner = config.n_inner if config.n_inner is not None else 4 * H
        self.n_head = config.num_attention_heads
        self.ln_1 = nn.LayerNorm(H, eps=config.layer_norm_epsilon)
        self.attn = nn.Module()
        self.attn.c_attn = Conv1D(3 * H, H)
        self.attn.c_proj = Conv1D(H, H)
        self.ln_2 = nn.LayerNorm(H, eps=config.layer_norm_epsilon)
        self.mlp = nn.Module()
        self.mlp.c_fc = Conv1D(inner, H)
        self.mlp.c_proj = Conv1D(H, inner)
        self.act = ACT2FN[config.activation_function]

    def forward(self, hidden_states, kv_cache=None, attn_metadata=None):
        T, H = hidden_states.shape
        d = H // self.n_head
        q, k, v = self.attn.c_attn(self.ln_1(hidden_states)).split(H, dim=-1)
        q, k, v = (t.view(T, self.n_head, d).transpose(0, 1) for t in (q, k, v))
        a = torch.nn.functional.scaled_dot_product_attention(q, k, v, is_causal=True)
        hidden_states = hidden_states + self.attn.c_proj(a.transpose(0, 1).reshape(T, H))
        return hidden_states + self.mlp.c_proj(self.act(self.mlp.c_fc(self.ln_2(hidden_states))))


class _Anything:
    """Stands in for every vllm name the module only mentions (type annotations, registries, base classes)."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        if len(a) == 1 and not k and isinstance(a[0], type):
            return a[0]          # registry decorators applied to the model class
        return _Anything()

    def __getattr__(self, name):
        return _Anything()

    def __mro_entries__(self, bases):
        return ()


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Anything()


def _install_stubs():
    names = ["networkx", "networkx.algorithms", "networkx.algorithms.clique", "triton", "triton.language", "vllm",
             "vllm.attention", "vllm.config", "vllm.distributed", "vllm.inputs", "vllm.model_executor",
             "vllm.model_executor.layers", "vllm.model_executor.layers.logits_processor",
             "vllm.model_executor.layers.quantization", "vllm.model_executor.layers.sampler",
             "vllm.model_executor.layers.vocab_parallel_embedding", "vllm.model_executor.model_loader",
             "vllm.model_executor.model_loader.weight_utils", "vllm.model_executor.models",
             "vllm.model_executor.models.gpt2", "vllm.model_executor.models.utils",
             "vllm.model_executor.models.interfaces", "vllm.model_executor.sampling_metadata", "vllm.multimodal",
             "vllm.multimodal.inputs", "vllm.multimodal.utils", "vllm.sequence", "vllm.utils"]
    added = []
    for n in names:
        if n not in sys.modules:
            m = _StubModule(n)
            m.__path__ = []
            sys.modules[n] = m
            added.append(n)
    sys.modules["vllm.model_executor.models.gpt2"].GPT2Block = StandInGPT2Block

    class VocabParallelEmbedding(nn.Embedding):
        def __init__(self, num, dim, **kw):
            super().__init__(num, dim)

    class ParallelLMHead(nn.Module):
        def __init__(self, num, dim, bias=False, **kw):
            super().__init__()
            self.weight = nn.Parameter(torch.zeros(num, dim))
            self.bias = nn.Parameter(torch.zeros(num)) if bias else None

    vpe = sys.modules["vllm.model_executor.layers.vocab_parallel_embedding"]
    vpe.VocabParallelEmbedding = VocabParallelEmbedding
    vpe.ParallelLMHead = ParallelLMHead

    def make_layers(n, fn, prefix=""):
        return 0, n, nn.ModuleList([fn(prefix=f"{prefix}.{i}") for i in range(n)])

    mu = sys.modules["vllm.model_executor.models.utils"]
    mu.make_layers = make_layers
    mu.make_empty_intermediate_tensors_factory = lambda *a, **k: (lambda *a2, **k2: None)

    class _PP:
        is_first_rank = True
        is_last_rank = True

    sys.modules["vllm.distributed"].get_pp_group = lambda: _PP()
    sys.modules["vllm.sequence"].VLLM_TOKEN_ID_ARRAY_TYPE = "l"
    sys.modules["vllm.utils"].is_list_of = lambda v, t, **k: isinstance(v, list) and all(isinstance(x, t) for x in v)

    class SamplingParams:          # hijack.py subclasses it with kw_only=True (msgspec.Struct in vllm)
        def __init_subclass__(cls, **kw):
            super().__init_subclass__()

    sys.modules["vllm"].SamplingParams = SamplingParams
    return added


def _load(name: str, rel: str):
    if name in sys.modules:
        return sys.modules[name]
    for p in ("auralis", "auralis.common", "auralis.common.logging", "auralis.models", "auralis.models.xttsv2",
              "auralis.models.xttsv2.components", "auralis.models.xttsv2.components.vllm"):
        if p not in sys.modules:
            m = types.ModuleType(p)
            m.__path__ = []
            sys.modules[p] = m
    if "auralis.common.logging.logger" not in sys.modules:
        lg = types.ModuleType("auralis.common.logging.logger")
        import logging
        lg.setup_logger = lambda *a, **k: logging.getLogger("auralis-ref")
        sys.modules["auralis.common.logging.logger"] = lg
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF_ROOT, rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference_gpt_module():
    """vllm_mm_gpt.py executed from /root/reference with the stubs above; returns the module."""
    if not reference_gpt_available():
        raise FileNotFoundError(f"{REF_ROOT} not present")
    added = _install_stubs()
    try:
        return _load("auralis.models.xttsv2.components.vllm_mm_gpt", _GPT)
    finally:
        for n in added:            # leave no spec-less stubs behind (transformers probes optional packages with find_spec)
            sys.modules.pop(n, None)


def load_reference_hijack_module():
    if not reference_gpt_available():
        raise FileNotFoundError(f"{REF_ROOT} not present")
    added = _install_stubs()
    try:
        _load("auralis.models.xttsv2.components.vllm.hidden_state_collector",
              "src/auralis/models/xttsv2/components/vllm/hidden_state_collector.py")
        return _load("auralis.models.xttsv2.components.vllm.hijack", _HIJACK)
    finally:
        for n in added:
            sys.modules.pop(n, None)


def build_reference_gpt2model(gpt_sd, n_layer: int):
    """The reference's GPT2Model (its own __init__ and forward) on the synthetic checkpoint's tensors."""
    from transformers import GPT2Config
    mod = load_reference_gpt_module()
    cfg = GPT2Config(vocab_size=1026, n_positions=608, n_embd=1024, n_layer=n_layer, n_head=16, n_inner=4096,
                     activation_function="gelu_new", layer_norm_epsilon=1e-5, resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0)
    cfg.num_audio_tokens = 1026
    cfg.max_audio_tokens = 605
    cfg.decoder_input_dim = 1024
    model = mod.GPT2Model(cfg)
    model.audio_start_generation_token = 1024
    sd = {"wte.weight": gpt_sd["gpt.wte.weight"], "wpe.emb.weight": gpt_sd["gpt.wpe.emb.weight"],
          "ln_f.weight": gpt_sd["gpt.ln_f.weight"], "ln_f.bias": gpt_sd["gpt.ln_f.bias"]}
    for i in range(n_layer):
        for k in ("ln_1.weight", "ln_1.bias", "ln_2.weight", "ln_2.bias", "attn.c_attn.weight", "attn.c_attn.bias",
                  "attn.c_proj.weight", "attn.c_proj.bias", "mlp.c_fc.weight", "mlp.c_fc.bias", "mlp.c_proj.weight",
                  "mlp.c_proj.bias"):
            sd[f"h.{i}.{k}"] = gpt_sd[f"gpt.h.{i}.{k}"]
    missing, unexpected = model.load_state_dict({k: v.clone().float() for k, v in sd.items()}, strict=True)
    model.eval()
    return model
