"""Generate tests/golden/gpt_glue_L2.npz by running the REFERENCE's own GPT glue (build container only).

TEST INFRASTRUCTURE ONLY.  Usage: python -m oracle.make_golden_gpt
What runs from /root/reference, unmodified (oracle/ref_gpt_import.py): `GPT2Model.__init__/forward` and
`LearnedPositionEmbeddings` of vllm_mm_gpt.py (embedding sums, start-token embed at mel position 0, conditioning splice,
block loop, ln_f) and `LogitsRepetitionPenalizer` of vllm/hijack.py.  vllm's GPT2Block is the documented stand-in.
Weights are the seeded synthetic checkpoint (seed 1234, 2 layers), so only ids and reference outputs are stored.
The prompt is assembled the way XTTSv2.py does it: text_embedding(ids) + text_pos_embedding(ids) (:528, the reference's
LearnedPositionEmbeddings class) appended to the 32 speaker latents (_merge_conditioning, :330-345); the generated ids are
teacher-forced (`input_ids` = tokens, `position_ids` = 1..k, exactly what the decode steps feed one by one).
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from auralis_amd.checkpoint import (make_synthetic_conditioning, make_synthetic_gpt, make_synthetic_text_ids,  # noqa: E402
                                    make_synthetic_xtts)
from auralis_amd.config import XTTSDims  # noqa: E402
from oracle.ref_gpt_import import (build_reference_gpt2model, load_reference_gpt_module,  # noqa: E402
                                   load_reference_hijack_module)

N_LAYER, N_TEXT, TOKENS = 2, 9, [5, 77, 1000, 3, 3, 641]


def main():
    torch.manual_seed(0)
    dims = XTTSDims()
    gpt_sd = make_synthetic_gpt(dims.gpt, seed=1234, n_layer=N_LAYER)
    xtts_sd = make_synthetic_xtts(dims, seed=1234, gpt_sd=gpt_sd)
    mod = load_reference_gpt_module()
    model = build_reference_gpt2model(gpt_sd, N_LAYER)
    gpt_cond, _ = make_synthetic_conditioning(dims)                      # [1, 32, 1024]
    ids = list(make_synthetic_text_ids(dims, n_text=N_TEXT))
    # text side with the reference's position-embedding class
    text_emb = torch.nn.Embedding(*xtts_sd["text_embedding.weight"].shape)
    text_emb.weight.data.copy_(xtts_sd["text_embedding.weight"])
    text_pos = mod.LearnedPositionEmbeddings(xtts_sd["text_pos_embedding.emb.weight"].shape[0], 1024)
    text_pos.emb.weight.data.copy_(xtts_sd["text_pos_embedding.emb.weight"])
    t = torch.tensor(ids).unsqueeze(0)
    with torch.no_grad():
        text_rows = text_emb(t) + text_pos(t)                             # [1, n, 1024]
        cond = torch.cat([gpt_cond, text_rows], dim=1)                    # XTTSv2.py:343
        hidden = model(input_ids=torch.tensor(TOKENS), position_ids=torch.arange(1, len(TOKENS) + 1),
                       kv_caches=[None] * N_LAYER, attn_metadata=None, intermediate_tensors=None,
                       input_embeds=[cond], starting_sequence_start_ids=[0], is_profiling_run=False,
                       is_logit_only=torch.tensor([False]))               # [32 + n + 1 + k, 1024] = ln_f(h)
        # compute_logits (vllm_mm_gpt.py:671, 688): final_norm, then mel_head with bias
        fn = torch.nn.functional.layer_norm(hidden, (1024,), gpt_sd["final_norm.weight"], gpt_sd["final_norm.bias"], 1e-5)
        logits = fn @ gpt_sd["mel_head.weight"].t() + gpt_sd["mel_head.bias"]
    hj = load_reference_hijack_module()
    g = torch.Generator().manual_seed(5)
    raw = torch.randn(1026, generator=g) * 3.0
    prompt_ids, out_ids = [1] * (N_TEXT + 2) + [1024], [7, 7, 9, 1000, 9]   # placeholders + start token; generated so far
    pen = hj.LogitsRepetitionPenalizer(5.0)(prompt_ids, out_ids, raw.clone())
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    n_cond = 32 + N_TEXT
    np.savez_compressed(os.path.join(out_dir, "gpt_glue_L2.npz"), weights_seed=np.int64(1234), n_layer=np.int64(N_LAYER),
                        text_ids=np.asarray(ids, np.int64), tokens=np.asarray(TOKENS, np.int64),
                        cond_text_rows=text_rows[0].numpy(), ln_f_rows=hidden.numpy(),
                        logits_gen_rows=logits[n_cond:].numpy(),
                        pen_logits=raw.numpy(), pen_prompt_ids=np.asarray(prompt_ids, np.int64),
                        pen_output_ids=np.asarray(out_ids, np.int64), pen_result=pen.numpy())
    print("ln_f rows", tuple(hidden.shape), "logits rows", tuple(logits[n_cond:].shape))


if __name__ == "__main__":
    main()
