"""Smallest end-to-end use on an MI355X: the reference's three-line recipe against the HIP engine.

    python examples/basic.py /path/to/checkpoint_dir voice.wav "Hello from an MI355X."

`checkpoint_dir` is the reference's on-disk format (config.json + gpt/gpt2_model.safetensors + core_xttsv2/xtts-v2.safetensors,
e.g. written by `python -m auralis_amd.tools.convert_checkpoint model.pth out_dir`); `voice.wav` is a few seconds of
reference speech (RIFF/WAVE or FLAC; the conditioning networks run on the GPU through aur_compute_conditioning) or a
precomputed-conditioning .npz."""
import sys

from auralis_amd import TTS, TTSRequest


def main():
    ckpt, voice, text = sys.argv[1], sys.argv[2], " ".join(sys.argv[3:]) or "Hello from an MI three fifty five X."
    tts = TTS(scheduler_max_concurrency=8).from_pretrained(ckpt)
    try:
        out = tts.generate_speech(TTSRequest(text=text, speaker_files=[voice], language="auto"))
        out.save("hello.wav")
        n, sr, dur = out.get_info()
        print(f"wrote hello.wav: {n} samples at {sr} Hz ({dur:.2f} s)")
    finally:
        tts.close()


if __name__ == "__main__":
    main()
