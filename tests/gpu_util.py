"""Helpers shared by the -m gpu tests (engine construction on the synthetic checkpoint)."""
import functools

import numpy as np
import torch

SPK_KEY = 42


@functools.lru_cache(maxsize=4)
def packed_weights(n_layer: int):
    from auralis_amd.checkpoint import make_synthetic_gpt, make_synthetic_xtts
    from auralis_amd.config import XTTSDims
    from auralis_amd.weights import pack_all
    dims = XTTSDims()
    gpt_sd = make_synthetic_gpt(dims.gpt, seed=1234, n_layer=n_layer)
    xtts_sd = make_synthetic_xtts(dims, seed=1234, gpt_sd=gpt_sd)
    return pack_all(gpt_sd, xtts_sd), gpt_sd, xtts_sd


def make_engine(n_layer: int, max_seqs: int = 4, **kw):
    from auralis_amd._lib import NativeEngine
    from auralis_amd.checkpoint import make_synthetic_conditioning
    from auralis_amd.config import XTTSDims
    packed, gpt_sd, xtts_sd = packed_weights(n_layer)
    eng = NativeEngine(n_layer=n_layer, max_seqs=max_seqs, **kw)
    eng.load_weights(packed)
    cond, spk = make_synthetic_conditioning(XTTSDims())
    eng.set_conditioning(SPK_KEY, cond.numpy(), spk.numpy())
    return eng, gpt_sd, xtts_sd, cond, spk


def rms(a):
    a = np.asarray(a, dtype=np.float64)
    return float(np.sqrt(np.mean(a * a)))
