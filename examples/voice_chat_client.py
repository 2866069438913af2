"""Talk to the OpenAI-compatible server (python -m auralis_amd.entrypoints.oai_server --model <checkpoint_dir>):

    # plain speech synthesis, FLAC out (wav / pcm / flac are built in; mp3 / opus / aac need torchaudio or ffmpeg on the server)
    python examples/voice_chat_client.py speech http://127.0.0.1:8000 voice.flac "Hello there." out.flac

    # voice chat: the server forwards the messages to an upstream OpenAI-compatible endpoint with YOUR key, streams the text
    # deltas back and vocalises them every N words in the reference voice (the reference's /v1/chat/completions proxy)
    OPENAI_API_KEY=sk-... python examples/voice_chat_client.py chat http://127.0.0.1:8000 voice.wav \
        https://api.openai.com/v1/chat/completions gpt-4o-mini "Tell me a two-sentence story."
"""
import base64
import json
import os
import sys

import requests


def speech(server, voice, text, out_path):
    fmt = os.path.splitext(out_path)[1].lstrip(".") or "wav"
    body = {"input": text, "model": "xtts", "voice": [base64.b64encode(open(voice, "rb").read()).decode()],
            "response_format": fmt, "language": "auto"}
    r = requests.post(server + "/v1/audio/speech", json=body, timeout=600)
    r.raise_for_status()
    open(out_path, "wb").write(r.content)
    print(f"wrote {out_path} ({len(r.content)} bytes, {r.headers.get('content-type')})")


def chat(server, voice, upstream, model, prompt):
    body = {"model": model, "messages": [{"role": "user", "content": prompt}], "openai_api_url": upstream,
            "speaker_files": [base64.b64encode(open(voice, "rb").read()).decode()], "vocalize_at_every_n_words": 20,
            "modalities": ["text", "audio"]}
    hdr = {"Authorization": "Bearer " + os.environ["OPENAI_API_KEY"]}
    n_audio = 0
    with requests.post(server + "/v1/chat/completions", json=body, headers=hdr, stream=True, timeout=600) as r:
        r.raise_for_status()
        for line in r.iter_lines():
            if not line.startswith(b"data:"):
                continue
            payload = line[5:].strip()
            if payload == b"[DONE]":
                break
            ev = json.loads(payload)
            if ev.get("object") == "audio.chunk":
                n_audio += 1
                path = f"chat_{n_audio:03d}.wav"
                open(path, "wb").write(base64.b64decode(ev["data"]))
                print(f"\n[audio chunk -> {path}]")
            elif "error" in ev:
                print("\nerror:", ev["error"])
            else:
                print((ev.get("choices") or [{}])[0].get("delta", {}).get("content", ""), end="", flush=True)
    print()


if __name__ == "__main__":
    if len(sys.argv) >= 6 and sys.argv[1] == "speech":
        speech(*sys.argv[2:6])
    elif len(sys.argv) >= 8 and sys.argv[1] == "chat":
        chat(*sys.argv[2:8])
    else:
        print(__doc__)
