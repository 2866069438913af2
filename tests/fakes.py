"""Test doubles (CPU): a stand-in for NativeEngine with the same submit/step/poll surface and deterministic output."""
import numpy as np


class FakeNativeEngine:
    n_layer = 1

    def __init__(self, max_seqs=4, fail_on_step=None, conditioning_weights=None, fail_once_on_step=None, step_delay=0.0):
        self.conditioning_weights = conditioning_weights   # xtts-v2.safetensors tensors: enables compute_conditioning below
        self.max_seqs = max_seqs
        self.next_id = 1
        self.waiting, self.running, self.done = [], [], []
        self.speakers = {}
        self.submitted = []
        self.fail_on_step = fail_on_step              # every step from this one on raises (a lost device)
        self.fail_once_on_step = fail_once_on_step    # this step raises and fails what was running, as Engine::step / fail_in_flight
        self.steps = 0
        self.finished_total = 0
        self.polls = 0
        self.waiting_at_step = []   # sequences queued when each step began
        self.cancelled = []
        self.step_delay = step_delay   # seconds per step (a test that needs work still in flight when it acts)

    def compute_conditioning(self, pcm, max_ref_length=30, gpt_cond_len=6, gpt_cond_chunk_len=6, sound_norm_refs=False):
        """Stand-in for aur_compute_conditioning (mono float32 PCM at 22 050 Hz per reference): the PyTorch restatement of the
        conditioning networks in oracle/conditioning_oracle.py (XTTSv2.py:409-468).  Lives here, not in the product."""
        import torch

        from oracle import conditioning_oracle as Cn
        assert self.conditioning_weights is not None, "FakeNativeEngine(conditioning_weights=...) needed"
        sd = {k: v.float() for k, v in self.conditioning_weights.items()}
        embs, audios = [], []
        for x in pcm:
            a = torch.from_numpy(np.asarray(x, np.float32))[None, : 22050 * max_ref_length]
            if sound_norm_refs:
                a = (a / torch.abs(a).max()) * 0.75
            embs.append(Cn.speaker_embedding(sd, a, 22050))
            audios.append(a)
        cond = Cn.gpt_cond_latents(sd, torch.cat(audios, dim=-1), 22050, length=gpt_cond_len, chunk_length=gpt_cond_chunk_len)
        return cond.numpy(), torch.stack(embs).mean(dim=0).numpy()

    def set_conditioning(self, key, g, s):
        self.speakers[key] = (np.array(g), np.array(s))

    def submit(self, text_ids, speaker_key, temperature=0.75, top_p=0.85, top_k=50, repetition_penalty=5.0,
               max_tokens=605, seed=0, ignore_stop=False, priority=0):
        assert speaker_key in self.speakers
        sid = self.next_id
        self.next_id += 1
        n_steps = 1 + (sum(text_ids) % 5)          # finish out of submission order
        self.waiting.append({"seq_id": sid, "ids": list(text_ids), "left": n_steps, "seed": seed})
        self.submitted.append({"seq_id": sid, "text_ids": list(text_ids), "temperature": temperature, "seed": seed, "priority": priority})
        return sid

    def step(self):
        self.steps += 1
        if self.step_delay:
            import time
            time.sleep(self.step_delay)
        self.waiting_at_step.append(len(self.waiting))
        if self.fail_on_step is not None and self.steps >= self.fail_on_step:
            raise RuntimeError("injected engine failure")
        if self.fail_once_on_step is not None and self.steps == self.fail_once_on_step:
            for s in self.running:   # in flight -> reported through poll with error set; waiting sequences stay queued
                self.done.append({"seq_id": s["seq_id"], "tokens": np.zeros(0, np.int32), "wav": np.zeros(0, np.float32), "error": -2})
                self.finished_total += 1
            self.running = []
            raise RuntimeError("injected step failure")
        while self.waiting and len(self.running) < self.max_seqs:
            self.running.append(self.waiting.pop(0))
        for s in list(self.running):
            s["left"] -= 1
            if s["left"] <= 0:
                self.running.remove(s)
                n = len(s["ids"])
                wav = np.full(n * 10, float(s["seq_id"]), dtype=np.float32)
                self.done.append({"seq_id": s["seq_id"], "tokens": np.arange(n, dtype=np.int32), "wav": wav, "error": 0})
                self.finished_total += 1
        return len(self.waiting) + len(self.running), self.finished_total   # (n_live, n_finished_total) as aur_step

    def poll(self, cap=64, want_latents=True, copy=True):
        self.polls += 1
        out, self.done = self.done[:cap], self.done[cap:]
        return out

    def cancel(self, seq_id):
        """as aur_cancel: a waiting or running sequence stops and is reported with error -5 and no audio"""
        for q in (self.waiting, self.running):
            for s in list(q):
                if s["seq_id"] == seq_id:
                    q.remove(s)
                    self.done.append({"seq_id": seq_id, "tokens": np.zeros(0, np.int32), "wav": np.zeros(0, np.float32), "error": -5})
                    self.finished_total += 1
                    self.cancelled.append(seq_id)

    def release(self, seq_id):   # (poll(copy=False) contract of NativeEngine; the fake's arrays are always owned)
        pass

    def stats(self):
        return {"kv_blocks_total": 0}

    def close(self):
        pass
