"""Import the reference's own vocoder class UNMODIFIED from /root/reference (build container only).

TEST INFRASTRUCTURE ONLY.  `import auralis` fails here (vllm, torchaudio, librosa … are absent), so the
single file hifigan_decoder.py is loaded through stub parent packages plus a stub `torchaudio`
(only touched by ResNetSpeakerEncoder.__init__, hifigan_decoder.py:537-548).  Nothing is copied:
the module is executed from where it lies.  /root/reference does not exist on the GPU box, so this
is used only to (a) validate oracle/xtts_oracle.py and (b) generate tests/golden/ fixtures.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

REF_ROOT = os.environ.get("AURALIS_REFERENCE", "/root/reference")
_REL = "src/auralis/models/xttsv2/components/tts/layers/xtts/hifigan_decoder.py"


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REF_ROOT, _REL))


def load_reference_hifigan():
    """Return the reference module object for hifigan_decoder.py."""
    if not reference_available():
        raise FileNotFoundError(f"{REF_ROOT} not present")
    _name = "auralis.models.xttsv2.components.tts.layers.xtts.hifigan_decoder"
    if _name in sys.modules:
        return sys.modules[_name]
    stubbed = False
    if "torchaudio" not in sys.modules:
        stubbed = True
        ta = types.ModuleType("torchaudio")
        tr = types.ModuleType("torchaudio.transforms")

        class _Mel:  # placeholder; the speaker encoder front-end is never executed by the oracle
            def __init__(self, *a, **k):
                pass

            def __call__(self, x):
                raise RuntimeError("stub")

        import torch

        class MelSpectrogram(torch.nn.Module):
            def __init__(self, *a, **k):
                super().__init__()

        tr.MelSpectrogram = MelSpectrogram
        ta.transforms = tr
        sys.modules["torchaudio"] = ta
        sys.modules["torchaudio.transforms"] = tr
    pkgs = ["auralis", "auralis.common", "auralis.models", "auralis.models.xttsv2",
            "auralis.models.xttsv2.components", "auralis.models.xttsv2.components.tts",
            "auralis.models.xttsv2.components.tts.layers", "auralis.models.xttsv2.components.tts.layers.xtts"]
    for p in pkgs:
        if p not in sys.modules:
            m = types.ModuleType(p)
            m.__path__ = []  # mark as package
            sys.modules[p] = m
    if "auralis.common.utilities" not in sys.modules:
        u = types.ModuleType("auralis.common.utilities")
        u.load_fsspec = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("stub"))
        sys.modules["auralis.common.utilities"] = u
    name = "auralis.models.xttsv2.components.tts.layers.xtts.hifigan_decoder"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF_ROOT, _REL))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    if stubbed:   # do not leave a spec-less stub behind (transformers probes torchaudio with find_spec)
        sys.modules.pop("torchaudio", None)
        sys.modules.pop("torchaudio.transforms", None)
    return mod


def build_reference_decoder(xtts_sd):
    """Instantiate the reference HifiDecoder and load the waveform_decoder.* tensors strictly."""
    import torch
    mod = load_reference_hifigan()
    dec = mod.HifiDecoder()
    pref = "hifigan_decoder.waveform_decoder."
    sub = {k[len(pref):]: v for k, v in xtts_sd.items() if k.startswith(pref)}
    missing, unexpected = dec.waveform_decoder.load_state_dict(sub, strict=True), None
    dec.eval()
    return dec


def load_reference_xtts_layer(name: str):
    """Load another single-file module of the reference's xtts layers (latent_encoder, perceiver_encoder) unmodified."""
    load_reference_hifigan()   # installs the stub parent packages
    full = "auralis.models.xttsv2.components.tts.layers.xtts." + name
    if full in sys.modules:
        return sys.modules[full]
    path = os.path.join(REF_ROOT, os.path.dirname(_REL), name + ".py")
    spec = importlib.util.spec_from_file_location(full, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[full] = mod
    spec.loader.exec_module(mod)
    return mod
