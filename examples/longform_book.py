"""Book-length, mixed-language synthesis (BASELINE config 5), one process per GPU.

    single GPU :  python examples/longform_book.py ckpt_dir voice.wav book.txt
    one node   :  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
                      examples/longform_book.py ckpt_dir voice.wav book.txt

Paragraphs (blank-line separated) become one request each with per-paragraph language detection; every rank synthesises
the paragraphs dealt to it, the speaker conditioning is computed once on rank 0 and broadcast over RCCL, rank 0 writes
book.wav."""
import asyncio
import os
import sys

import torch

from auralis_amd import TTS
from auralis_amd.longform import build_requests, split_paragraphs, synthesize_sharded
from auralis_amd.parallel import COND_ELEMS, SPK_ELEMS, pack_conditioning


def shared_voice(tts, voice, world, local_rank):
    """Speaker conditioning computed on rank 0 only, then ONE broadcast of 133 120 bytes (RCCL over xGMI)."""
    if world == 1:
        return voice
    dev = torch.device("cuda", local_rank)
    buf = torch.empty(COND_ELEMS + SPK_ELEMS, dtype=torch.float32, device=dev)
    if torch.distributed.get_rank() == 0:
        g, s = asyncio.run(tts.tts_engine.get_audio_conditioning([voice]))
        buf.copy_(pack_conditioning(torch.as_tensor(g), torch.as_tensor(s)))
    torch.distributed.broadcast(buf, src=0)
    flat = buf.cpu().numpy()
    return {"gpt_cond_latent": flat[:COND_ELEMS].reshape(1, 32, 1024), "speaker_embedding": flat[COND_ELEMS:].reshape(1, 512, 1)}


def main():
    ckpt, voice, book = sys.argv[1:4]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        torch.cuda.set_device(local_rank)
        torch.distributed.init_process_group("nccl")       # RCCL on ROCm
    tts = TTS(scheduler_max_concurrency=64).from_pretrained(ckpt, device=local_rank)
    try:
        paragraphs = split_paragraphs(open(book, encoding="utf-8").read())
        requests = build_requests(paragraphs, [shared_voice(tts, voice, world, local_rank)], seed=0)
        audio = synthesize_sharded(tts, requests, window=32)
        if audio is not None:
            audio.save("book.wav")
            print(f"{len(paragraphs)} paragraphs -> book.wav, {audio.get_info()[2] / 60:.1f} min of audio")
    finally:
        tts.close()
        if world > 1:
            torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
