/* auralis_amd.h — C ABI of the MI355X-native XTTSv2 generate_speech() hot path.
 *
 * Drop-in boundary for astramind-ai/Auralis (v0.2.8.post2).  The reference has no native code: its
 * engine plugin is the Python class XTTSv2Engine (src/auralis/models/xttsv2/XTTSv2.py:39) behind
 * BaseAsyncTTSEngine (src/auralis/models/base.py:57).  Each entry point below replaces the Python/vLLM
 * mechanism cited next to it; the ctypes binding a maintainer would add is shown in INTEGRATION.md and
 * shipped as auralis_amd/_lib.py.
 *
 * Conventions: every function returns 0 on success or a negative AUR_E_* code; the message of the last
 * failure on the calling thread is aur_last_error().  No exception crosses the boundary.  Handles are
 * opaque and owned by the library.  Pointers passed in are caller-owned HOST pointers borrowed for the
 * duration of the call unless the name says _device.  All floating-point tensors are fp32, row-major.
 * Threading: one driver thread per engine calls aur_step(); aur_submit()/aur_poll_finished()/
 * aur_release() may be called from other threads (internally serialised).
 */
#ifndef AURALIS_AMD_H
#define AURALIS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AUR_OK 0
#define AUR_E_INVALID (-1)   /* bad argument / unknown id / shape mismatch */
#define AUR_E_HIP (-2)       /* HIP runtime or kernel failure */
#define AUR_E_STATE (-3)     /* call not valid in the current state (e.g. weights not loaded) */
#define AUR_E_NOMEM (-4)
#define AUR_E_CANCELLED (-5) /* aur_result.error of a sequence that was cancelled with aur_cancel */

typedef struct aur_engine aur_engine;

/* Engine geometry.  Replaces the AsyncEngineArgs built in XTTSv2Engine.init_vllm_engine
 * (XTTSv2.py:198-232: max_model_len 1047, max_num_seqs = concurrency, block size 16). */
typedef struct aur_config {
    int32_t n_layer;          /* GPT blocks (30 for XTTSv2; tests may use fewer) */
    int32_t max_seqs;         /* concurrent sequences (continuous-batching slots).  A layer's K/V pool is addressed with 32-bit byte
                                 offsets, so max_seqs * 66 + 2 * max_speakers blocks must stay <= 32767 (fp32 pool; 65535 with
                                 kv_fp16): 494 / 991 slots at the default 64 speakers; aur_engine_create reports the limit it computed */
    int32_t max_prefill_rows; /* cap on prompt rows prefetched in one step (0 = default 8192) */
    int32_t max_speakers;     /* speaker-conditioning table entries (0 = default 64) */
    int32_t vocoder_min_batch;/* finished sequences wait until this many are ready for one vocoder pass -- at most 16 steps (~30 ms) counted
                                 from the step the oldest waiting one finished, plus, when a vocoder pass is already in flight, the rest of
                                 that pass (one pass at a time) -- and not at all when nothing else runs or waits.  0 = default
                                 max_seqs / 16 (4 at 64 slots); 1 = vocode at once */
    int32_t profile;          /* 1 = profile mode from the start (see aur_set_profile; every 64th decode step) */
    int32_t vocoder_fp16;     /* 1 = HiFi-GAN convs on fp16-input / fp32-accumulate MFMA (needs the voc16.* tensors);
                                 0 = exact-f32 MFMA parity mode */
    int32_t second_pass;      /* 1 = A/B mode: recompute the latents with the reference's literal second GPT pass
                                 (XTTSv2.py:617-687) instead of the decode-time stash */
    int32_t return_latents;   /* 1 = also copy each sequence's vocoder input latents to the host (aur_result.latents; parity
                                 tests).  0 = audio and tokens only, as the reference's TTSOutput (saves ~1.1 MB of D2H
                                 and host copies per 280-token utterance) */
    int32_t kv_fp16;          /* 1 = throughput mode: the paged K/V pool holds fp16 (rounded to nearest on write; scores, softmax
                                 and P.V stay fp32), which halves the bytes the decode attention streams.  NOT the parity mode:
                                 greedy ids can differ from the fp32 reference after a near-tie (measured rate: DESIGN.md §4).
                                 0 (default) = fp32 K/V, bit-exact contract */
    int32_t gemm_f32_exact;   /* arithmetic of BOTH GEMM regimes (decode rows: gemm_rows_kernel; prompt rows: gemm_tile_split_kernel /
                                 gemm_tile_kernel): 0 (default) = every fp32 operand is split exactly into three bf16 terms and a product
                                 runs as six bf16 MFMAs with fp32 accumulation (the accuracy of an fp32 dot product at 2.7x less
                                 matrix-pipe time); 1 = v_mfma_f32_*_f32, bitwise an fp32 fma chain.  The speaker-conditioning networks
                                 always use the exact-f32 kernels */
    int32_t gelu_erf;         /* MLP activation: 0 = tanh form ("gelu_new", what checkpoint_converter.py:197 writes), 1 = erf form
                                 ("gelu", the XTTSGPTConfig class default, xttsv2_gpt_config.py:184); from the checkpoint's
                                 gpt/config.json "activation_function" */
    int32_t admit_min_batch;  /* continuous batching under saturation: while MORE sequences wait than slots are free (and something
                                 is running), hold admission until this many slots are free, so that one prefill pass (a fixed
                                 ~4.5 ms of launches at 30 layers, whatever the rows) serves that many prompts.  A request that finds
                                 a free slot and no longer queue than free slots is admitted at once, as the reference's vLLM scheduler
                                 does (two_phase_scheduler.py:168-236 hands every request straight to it).  A hold ends after 32 steps (~60 ms) whatever is
                                 free by then.  0 = default max_seqs / 8 (8 at 64 slots); 1 = never hold */
    int32_t urgent_rows;      /* running sequences the engine fills up to while a latency-critical sequence (aur_seq_desc.priority > 0)
                                 waits or runs.  0 = default max_seqs / 4; max_seqs = no cap */
} aur_config;

/* One named fp32 tensor.  Names are the packed names produced by auralis_amd/weights.py from the
 * reference's on-disk keys (checkpoint_converter.py:225-284; loaders XTTSv2.py:288-301,
 * vllm_mm_gpt.py:714-733). */
typedef struct aur_tensor_desc {
    const char* name;
    const float* data;
    int64_t numel;
} aur_tensor_desc;

/* One sequence = one <=250-char text chunk.  Replaces TokensPrompt + ExtendedSamplingParams built in
 * XTTSv2Engine.get_generation_context (XTTSv2.py:727-756) and the text embedding of
 * prepare_text_tokens_async (XTTSv2.py:506-543). */
typedef struct aur_seq_desc {
    const int32_t* text_ids;  /* BPE ids incl. [START]/[STOP] (XTTSv2.py:520-521) */
    int32_t n_text;
    uint64_t speaker_key;     /* conditioning previously registered with aur_set_conditioning */
    float temperature;        /* < 1e-5 => greedy (vLLM _SAMPLING_EPS) */
    float top_p;
    int32_t top_k;            /* <= 0 disables */
    float repetition_penalty; /* hijack.py:49-88 semantics, 1.0 disables */
    int32_t max_tokens;       /* gpt_max_audio_tokens (605) */
    uint32_t seed;            /* per-sequence noise stream (new surface; reference has none) */
    int32_t ignore_stop;      /* 1 = fixed-length mode for timing (stop id does not end the sequence) */
    int32_t priority;         /* 0 = normal; > 0 = latency-critical (the chunk a listener is waiting for: the head of a stream; the
                                 reference has no counterpart, its streaming latency is whatever vLLM's FIFO gives,
                                 tests/integration/stream_ttfb.py:24-37).  It is admitted in front of the normal queue and at once,
                                 vocoded as soon as its tokens are done, and while it waits or runs the engine admits normal
                                 sequences only up to aur_config.urgent_rows running ones (a decode step's time grows with its
                                 rows).  Results do not depend on it */
} aur_seq_desc;

/* Finished sequence.  Pointers stay valid until aur_release(seq_id).  Replaces the RequestOutput consumed at
 * XTTSv2.py:785 and the TTSOutput array built at XTTSv2.py:804-811. */
typedef struct aur_result {
    uint64_t seq_id;
    int32_t n_tokens;
    const int32_t* tokens;    /* mel token ids, stop id included when emitted */
    int32_t n_samples;
    const float* wav;         /* 24 kHz mono PCM in [-1, 1] */
    int32_t n_latent_rows;
    const float* latents;     /* [n_tokens][1024] vocoder input (final_norm applied twice), may be NULL */
    int32_t error;            /* 0 or AUR_E_* if this sequence failed */
} aur_result;

typedef struct aur_stats {
    int64_t steps;                 /* aur_step calls that did work */
    int64_t prefill_rows;          /* prompt rows processed */
    int64_t decode_rows;           /* decode rows processed (one per live sequence per step) */
    int64_t tokens_generated;
    int64_t samples_generated;
    int64_t vocoder_batches;
    /* profile == 1 only: HIP-event time of the MFMA conv launches of the vocoder */
    int64_t conv_launches;
    double conv_ms;
    double conv_flops;             /* algorithmic FLOPs of those launches */
    double conv_bytes;             /* algorithmic (layer-granular) HBM bytes of those launches */
    int64_t gemm_launches;         /* profile mode: decode GEMM launches timed in the replay batches (aur_set_profile) */
    double gemm_ms;                /* event time of the batches minus one event-pair overhead per batch */
    double gemm_ms_raw;            /* event time of the batches */
    double event_pair_overhead_ms; /* measured cost of one empty HIP-event pair */
    double gemm_flops;
    double gemm_bytes;             /* algorithmic: weights once + activations + slabs */
    double vocoder_ms;             /* whole vocoder batches (interp .. conv_post), event-timed */
    double gpt_ms;                 /* prefill + decode + sampling, event-timed per step */
    int64_t kv_blocks_total;
    int64_t kv_blocks_free;
    /* profile mode, per-kernel splits.  GEMM kinds: 0 qkv, 1 attn proj, 2 fc, 3 mlp proj, 4 mel head.  After a profiled decode
     * step its launches are replayed kind by kind, the n_layer launches of one kind back to back between ONE HIP-event pair
     * (outputs redirected to scratch): *_ms are those intervals, *_launches the launches inside them.  Bytes are ALGORITHMIC
     * (weights + activations in + activations out, fp32 as stored; attention: K and V rows of every live sequence's whole
     * context + q + out). */
    int64_t gemm_kind_launches[5];
    double gemm_kind_ms[5];
    double gemm_kind_bytes[5];
    double gemm_kind_flops[5];
    int64_t attn_launches;
    double attn_ms;
    double attn_bytes;
    /* every decode step (not sampled): event time and algorithmic bytes of the whole step */
    int64_t decode_steps;
    double decode_ms;              /* embed .. sampler of the decode steps (gpt_ms = decode_ms + prefill_ms) */
    double prefill_ms;
    double decode_weight_bytes;    /* block + head weights, once per step, fp32 as stored */
    double decode_kv_bytes;        /* K/V rows read by attention, all layers, fp32 as stored */
    /* vocoder conv launches by class (profile mode): 0..3 = ResBlock convs of the 256- / 128- / 64- / 32-channel stage,
     * 4 = conv_pre and the four polyphase transposed convs.  The stages are bounded differently (the wide ones by the fp16
     * matrix pipe and LDS, the narrow ones by HBM), so the bench reports a roofline per class. */
    int64_t conv_class_launches[5];
    double conv_class_ms[5];
    double conv_class_bytes[5];
    double conv_class_flops[5];
    int64_t prefill_batches;       /* prefill passes (one per aur_step that admitted sequences; aur_config.admit_min_batch groups them) */
    /* resource gauges (not reset by aur_reset_stats): what a soak test watches next to hipMemGetInfo (the reference's only resource
     * assertion is tests/integration/memory_leak.py:42-51, VRAM delta over 100 generate_speech calls) */
    int64_t result_blocks;         /* pinned result blocks the engine has allocated (one per vocoder batch in flight or undelivered) */
    int64_t result_blocks_free;    /* ... of which no sequence holds a reference (reusable by the next vocoder batch) */
    int64_t result_block_bytes;    /* pinned host bytes of all result blocks */
    int64_t speakers;              /* voices registered in the speaker table (<= aur_config.max_speakers) */
    int64_t sequences_tracked;     /* sequences the engine still holds (waiting, running, vocoding, or finished and not yet released) */
} aur_stats;

const char* aur_last_error(void);
int aur_version(void);

int aur_engine_create(const aur_config* cfg, int device_id, aur_engine** out);
int aur_engine_destroy(aur_engine* e);

/* Upload packed weights (may be called several times; later names overwrite earlier ones). */
int aur_load_weights(aur_engine* e, const aur_tensor_desc* tensors, size_t n);

/* Register/replace the conditioning of one speaker: gpt_cond_latent [32][1024] and speaker embedding
 * [512] (outputs of XTTSv2Engine.get_conditioning_latents, XTTSv2.py:409-468).  Also precomputes the
 * vocoder's 1x1 conditioning convs (hifigan_decoder.py:243-251) for this speaker on the GPU. */
int aur_set_conditioning(aur_engine* e, uint64_t speaker_key, const float* gpt_cond, const float* spk_emb);
/* Same with DEVICE pointers (e.g. the receive buffer of an RCCL broadcast). */
int aur_set_conditioning_device(aur_engine* e, uint64_t speaker_key, const float* d_gpt_cond,
                                const float* d_spk_emb);

/* RCCL inside the boundary (the reference has no counterpart: it forwards tensor_parallel_size to vLLM,
 * XTTSv2.py:214-215).  One engine per GPU per process; the only exchange of the utterance-sharded path is the speaker
 * conditioning.  aur_comm_unique_id: rank 0 creates the 128-byte id and the launcher hands it to every rank (any channel);
 * aur_comm_init: collective, builds the communicator of this engine; aur_broadcast_conditioning: collective, the voice
 * registered on `root` is sent with ONE ncclBroadcast of 133 120 bytes over xGMI and registered on the other ranks from the
 * device receive buffer (same effect as aur_set_conditioning_device there).  RCCL is loaded at run time (dlopen). */
int aur_comm_unique_id(uint8_t* out128);
int aur_comm_init(aur_engine* e, const uint8_t* id128, int32_t rank, int32_t world_size);
int aur_broadcast_conditioning(aur_engine* e, uint64_t speaker_key, int32_t root);
/* What the communicator itself reports (ncclCommCount / ncclCommUserRank): *n_ranks = 0 before aur_comm_init. */
int aur_comm_info(aur_engine* e, int32_t* n_ranks, int32_t* rank);
/* 64-bit FNV-1a checksum over the registered conditioning of a voice as it sits in device memory (gpt_cond_latent [32][1024]
 * then speaker embedding [512], fp32 bytes): lets the ranks of a job verify that a broadcast delivered identical bytes. */
int aur_conditioning_checksum(aur_engine* e, uint64_t speaker_key, uint64_t* out);

/* *out = 1 if speaker_key is registered (and marks it most recently used), else 0.  The table holds
 * aur_config.max_speakers voices; when it is full, registering a new key evicts the least recently used voice that has
 * no undelivered sequences (replaces the reference's unbounded per-request conditioning, XTTSv2.py:409-468: callers
 * re-register an evicted voice). */
int aur_has_conditioning(aur_engine* e, uint64_t speaker_key, int32_t* out);

/* Speaker conditioning from reference audio on the GPU (replaces XTTSv2Engine.get_conditioning_latents,
 * models/xttsv2/XTTSv2.py:409-468, with get_speaker_embedding :312-328 and get_gpt_cond_latents :349-407 underneath:
 * ResNet-SE speaker encoder, ConditioningEncoder, PerceiverResampler and their mel front-ends).
 * pcm[r]: n_samples[r] mono float32 samples of reference r at 22 050 Hz in HOST memory (decoding / resampling of the files is
 * the loader's job, as in the reference's load_audio).  Outputs (host): gpt_cond_latent [32][1024], speaker embedding [512],
 * ready for aur_set_conditioning.  Needs the "cond.*" tensors (auralis_amd/weights.py: pack_conditioning). */
typedef struct aur_cond_params {
    int32_t max_ref_length;      /* seconds of each reference that are used (reference default 30) */
    int32_t gpt_cond_len;        /* seconds of the concatenated references that feed the latents (default 6) */
    int32_t gpt_cond_chunk_len;  /* seconds per chunk (default 6) */
    int32_t sound_norm_refs;     /* 1: each reference scaled to 0.75 of its peak */
} aur_cond_params;
int aur_compute_conditioning(aur_engine* e, const float* const* pcm, const int32_t* n_samples, int32_t n_refs,
                             const aur_cond_params* params, float* out_gpt_cond, float* out_spk_emb);

/* Queue a sequence (replaces llm_engine.generate, XTTSv2.py:752). */
int aur_submit(aur_engine* e, const aur_seq_desc* seq, uint64_t* seq_id);

/* One continuous-batching iteration: admit + prefill waiting sequences, one decode step for every live
 * sequence, fused sampling, and vocoding of sequences that finished (replaces vLLM's engine step, the
 * second pass of get_model_logits XTTSv2.py:617-687 and the HiFi-GAN call XTTSv2.py:804).
 * n_live = sequences still waiting/decoding/vocoding after the step. */
int aur_step(aur_engine* e, int32_t* n_live, int32_t* n_finished_total);

int aur_poll_finished(aur_engine* e, aur_result* out, size_t cap, size_t* n);
int aur_release(aur_engine* e, uint64_t seq_id);
/* Stop a sequence nobody is waiting for any more (a streaming consumer that disconnected; the reference has no counterpart: its
 * chunks decode to the end, two_phase_scheduler.py:279-291 only stops scheduling new ones).  A sequence still waiting for a slot is
 * dropped at once; a running one stops within two decode steps and is NOT vocoded; one whose tokens are done but whose vocoder batch has
 * not been launched is dropped; one already being vocoded (or finished) is left alone.  A cancelled sequence is reported by
 * aur_poll_finished like any other, with error = AUR_E_CANCELLED, the tokens generated so far and no audio; it still needs
 * aur_release.  Thread-safe, does not wait for a running aur_step.  No other sequence's output changes. */
int aur_cancel(aur_engine* e, uint64_t seq_id);

/* Standalone vocoder (HifiDecoder.forward, hifigan_decoder.py:776-802) for parity tests and for callers
 * that bring their own latents: latents [B][t_max][1024], n_lat[B] valid rows each; wav_out [B][wav_stride]. */
int aur_vocode(aur_engine* e, const float* latents, const int32_t* n_lat, int32_t B, int32_t t_max,
               uint64_t speaker_key, float* wav_out, int64_t wav_stride, int32_t* n_samples_out);

int aur_sync(aur_engine* e);
int aur_get_stats(aur_engine* e, aur_stats* out);
int aur_reset_stats(aur_engine* e);
/* Profile mode at run time (new surface; the reference has no counterpart): every > 0 -- HIP events around every vocoder conv
 * launch, and after every `every`-th decode step the per-kind replay batches described at aur_stats.gemm_kind_*; 0 = off (the
 * configuration the parity tests run).  Results of the sequences are the same either way. */
int aur_set_profile(aur_engine* e, int32_t every);

/* ---- per-kernel entry points used by the parity tests (host pointers) ------------------------------- */
/* Prefill-regime GEMM (gemm_tile_split_kernel, or gemm_tile_kernel under aur_config.gemm_f32_exact: the kernel every prompt-row
 * linear of vLLM's GPT2Block runs on here, vllm_mm_gpt.py:757-761 at M = prompt rows): out[M][N] = X[M][K] @ W[K][N]. */
int aur_dbg_gemm(aur_engine* e, const float* X, const float* W, float* out, int32_t M, int32_t N, int32_t K);
/* Decode-regime GEMM (gemm_rows_kernel: full-K workgroups, fused LayerNorm prologue and bias / gelu / residual epilogue;
 * replaces one GPT2Block linear of vLLM's GPT2Attention / GPT2MLP at M = live sequences, vllm_mm_gpt.py:757-761):
 * out[M][N] = epi(LN?(X[M][K]) @ W[K][N] + bias), epi 0 = bias, 1 = bias + gelu_new, 2 = out += (.. + bias).
 * ln != 0 applies LayerNorm(gamma, beta, eps 1e-5) to the rows of X first (K must be 1024).  K in {1024, 4096}. */
int aur_dbg_gemm_rows(aur_engine* e, const float* X, const float* W, const float* bias, const float* gamma,
                      const float* beta, float* out, int32_t M, int32_t N, int32_t K, int32_t epi, int32_t ln);
/* Stress of the decode GEMM's K-split form (the K = 4096 -> 1024 projection at M <= 32 live sequences, same reference linear): `iters`
 * back-to-back launches that ALTERNATE between two pseudo-random operand sets (activations and residual), so that consecutive
 * launches publish different partial tiles and a stale read cannot return the right bits, while a second stream streams 256 MiB
 * per launch through the chip for the whole test; each result is compared bitwise on the device with the unsplit kernel's result on
 * the matching operands; *mismatches_out = differing 32-bit words over all launches (0 = the cross-workgroup hand-off held every time). */
int aur_dbg_gemm_rows_ksplit_stress(aur_engine* e, int32_t M, int32_t iters, int64_t* mismatches_out);
/* The kernels' cross-lane helpers (lane_xor<J>, wave_sum, wave_max: DPP and gfx950 lane swaps, csrc/common.h) against the wavefront
 * shuffles they replace, on `blocks` workgroups of pseudo-random words; *mismatches_out = results that differ bitwise (0 expected).
 * Test support: the reference has no counterpart (its reductions are torch's). */
int aur_dbg_lane_xor_selftest(aur_engine* e, int32_t blocks, int64_t* mismatches_out);
/* The decode attention kernel alone (models/xttsv2/components/vllm_mm_gpt.py:757-761: GPT2Attention over vLLM's paged K/V).
 * q [M][1024]; row m attends ctx[m] >= 1 cached tokens whose keys / values are k, v [M][ctx_max][1024] (16 heads x 64 inside a token);
 * the first `shared` tokens (0, 16 or 32; <= every ctx[m]) are taken from row 0 and live in blocks every row's table points to, as the
 * speaker prefix does.  kv_half: the fp16 pool (values rounded to nearest on the way in).  out [M][1024].  Test support. */
int aur_dbg_paged_attention(aur_engine* e, const float* q, const float* k, const float* v, const int32_t* ctx, int32_t M,
                            int32_t ctx_max, int32_t shared, int32_t kv_half, float* out);
/* The prompt-row (prefill) attention kernel alone.  n_seq sequences with keys / values k, v [n_seq][ctx_max][1024]; M query rows
 * q [M][1024], row m belongs to sequence row_seq[m] and sits at position row_pos[m] (it attends that sequence's tokens 0..row_pos[m]);
 * consecutive rows of one sequence with positions ascending by one form the kernel's query blocks (cut every 32 rows), exactly as the
 * engine's prefill builds them.  The first `shared` tokens (0, 16 or 32) are sequence 0's and live in blocks every table points to.
 * out [M][1024].  Test support. */
int aur_dbg_prompt_attention(aur_engine* e, const float* q, const float* k, const float* v, const int32_t* row_seq, const int32_t* row_pos,
                             int32_t M, int32_t n_seq, int32_t ctx_max, int32_t shared, int32_t kv_half, float* out);
/* out[M][1024] = LayerNorm(h) rows */
int aur_dbg_layernorm(aur_engine* e, const float* h, const float* gamma, const float* beta, float* out,
                      int32_t M);
/* Generic masked conv on packed weights: x [B][Cin][L]; out [B][Cout][L*max(1,ups_s)]; see vocoder_kernels.h */
int aur_dbg_conv1d(aur_engine* e, const float* x, const float* wp, const float* bias, const float* res,
                   float* out, const int32_t* lens, int32_t B, int32_t Cin, int32_t Mtot, int32_t Cout,
                   int32_t L, int32_t KS, int32_t DIL, int32_t padl, float slope, int32_t ups_s, int32_t ups_p);
/* Same on the fp16-input MFMA kernel; wp16 = packed halves [Mtot/MT][Cin/16][KS][MT][16] (auralis_amd/weights.py). */
int aur_dbg_conv1d_f16(aur_engine* e, const float* x, const void* wp16, const float* bias, const float* res,
                       float* out, const int32_t* lens, int32_t B, int32_t Cin, int32_t Mtot, int32_t Cout,
                       int32_t L, int32_t KS, int32_t DIL, int32_t padl, float slope, int32_t ups_s, int32_t ups_p);
/* Prefill one prompt and return ln_f rows [n_rows][1024] and the penalised logits [1026] of the last row. */
int aur_dbg_prefill(aur_engine* e, const int32_t* text_ids, int32_t n_text, uint64_t speaker_key,
                    float repetition_penalty, float* lnf_rows_out, float* logits_out);
/* Run the fused sampler on given logits rows [B][1026] (greedy if temperature < 1e-5). */
int aur_dbg_sample(aur_engine* e, const float* logits, int32_t B, float temperature, float top_p, int32_t top_k,
                   float repetition_penalty, const uint8_t* seen /*[B][1026] or NULL*/, uint32_t seed,
                   int32_t step, int32_t* tokens_out);

#ifdef __cplusplus
}
#endif
#endif /* AURALIS_AMD_H */
