// Host runtime + C ABI (include/auralis_amd.h) of the MI355X-native XTTSv2 hot path.
//
// One Engine per GPU/process.  It owns: the packed weights, a paged KV pool with a block allocator, the
// per-slot device state of the continuous batcher, the latent stash that replaces the reference's second
// GPT pass (XTTSv2.py:617-687; equivalence argued in SURVEY.md §7 and tested in tests/), and the vocoder
// workspace.  The reference's counterparts are vLLM's AsyncLLMEngine/scheduler/block manager (un-vendored,
// XTTSv2.py:198-232), HiddenStatesCollector (components/vllm/hidden_state_collector.py) and the
// asyncio.to_thread HiFi-GAN call (XTTSv2.py:804).
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/auralis_amd.h"
#include "gpt_kernels.h"
#include "vocoder_kernels.h"

namespace aur {

static thread_local std::string g_last_error;

constexpr int kProfileEvery = 64;   // profile mode default: every 64th decode step is followed by the per-kind replay batches (profile_replay)
constexpr int kProj2Slabs = 4;   // split-K slabs of the prompt-row MLP projection (forward_rows)

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    void ensure(size_t n) {
        if (n <= bytes) return;
        if (p) HIP_CHECK(hipFree(p));
        p = nullptr;
        bytes = 0;
        HIP_CHECK(hipMalloc(&p, n));
        bytes = n;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <class T>
    T* as() const {
        return reinterpret_cast<T*>(p);
    }
};

struct PinBuf {
    void* p = nullptr;
    size_t bytes = 0;
    ~PinBuf() {
        if (p) (void)hipHostFree(p);
    }
    void ensure(size_t n) {
        if (n <= bytes) return;
        if (p) HIP_CHECK(hipHostFree(p));
        p = nullptr;
        HIP_CHECK(hipHostMalloc(&p, n, hipHostMallocDefault));
        bytes = n;
    }
    template <class T>
    T* as() const {
        return reinterpret_cast<T*>(p);
    }
};

}  // namespace aur
#include "cond_net.h"   // (uses DevBuf)
namespace aur {

// Pinned result block of one vocoder batch: the D2H copies land here and aur_result.wav / .latents point straight into
// it (no second host copy); the block returns to the pool when the last sequence of the batch is released.
struct PinBlock {
    PinBuf buf;
    int refs = 0;
};

struct Tensor {
    std::unique_ptr<DevBuf> buf;
    int64_t numel = 0;
    float* d() const { return buf->as<float>(); }
};

struct SlotInit {
    int slot;
    float temperature, top_p;
    int top_k;
    float rep_penalty;
    int max_tokens, ignore_stop;
    unsigned seed;
    int ngen;   // initial n_gen (0; dbg_sample injects a step)
};

__global__ __launch_bounds__(256) void init_slots_kernel(const SlotInit* __restrict__ init, int* slot_tok,
                                                         int* slot_pos, int* slot_kvpos, int* slot_ngen,
                                                         int* slot_finished, float* temperature, float* top_p,
                                                         int* top_k, float* rep, int* max_tokens, int* ignore_stop,
                                                         unsigned* seed, unsigned char* seen, int start_token) {
    const SlotInit s = init[blockIdx.x];
    unsigned char* row = seen + (long)s.slot * kSeenStride;
    for (int i = threadIdx.x; i < kSeenStride; i += 256) row[i] = (i == 1 || i == start_token) ? 1 : 0;
    if (threadIdx.x == 0) {
        slot_tok[s.slot] = start_token;
        slot_pos[s.slot] = 0;
        slot_kvpos[s.slot] = 0;
        slot_ngen[s.slot] = s.ngen;
        slot_finished[s.slot] = 0;
        temperature[s.slot] = s.temperature;
        top_p[s.slot] = s.top_p;
        top_k[s.slot] = s.top_k;
        rep[s.slot] = s.rep_penalty;
        max_tokens[s.slot] = s.max_tokens;
        ignore_stop[s.slot] = s.ignore_stop;
        seed[s.slot] = s.seed;
    }
}

// Test support (aur_dbg_gemm_rows_ksplit_stress): background HBM / L2 traffic on a second stream.  Every workgroup streams a
// 64 KiB-aligned share of `src` (>= 32 KiB per CU per launch, the guide's "uneven load" rule for cross-workgroup hand-off tests)
// and leaves one partial sum per workgroup so that the loads cannot be dropped.
__global__ __launch_bounds__(256) void stress_load_kernel(const float4* __restrict__ src, float* __restrict__ sink, long n_vec) {
    float acc = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n_vec; i += (long)gridDim.x * 256) {
        const float4 v = src[i];
        acc += v.x + v.y + v.z + v.w;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) atomicAdd(sink + blockIdx.x, acc);
}

// Row workspace of one GPT forward chain (prefill uses chain 0; decode can run two chains on two streams).
struct RowWs {
    hipStream_t st = nullptr;
    int rows_cap = 0;
    DevBuf h, xn, qbuf, att, act, P, P2, ybuf, stats, row_meta;   // row_meta: per-row K/V addressing of the current decode step   // stats: LayerNorm partials of the packed residual stream [rows][64] float2
    DevBuf i_row_slot, i_row_pos, i_desc, i_sample_row, i_sample_slot, i_next_kvpos, i_out_tok, i_qblk;
    std::vector<int2> h_qblk;   // query blocks of the prompt rows in flight (launch_prompt_attention)
    PinBuf pin;
    std::vector<int> sample_row, sample_slot;
};

enum class SeqState { WAITING, RUNNING, TOKENS_DONE, DONE, RELEASED };

struct Seq {
    uint64_t id = 0;
    SeqState state = SeqState::WAITING;
    std::vector<int> text_ids;
    int spk_row = -1;
    aur_seq_desc params{};
    int slot = -1;
    int n_prompt = 0;
    std::vector<int> blocks;
    std::vector<int32_t> tokens;
    const float* wav = nullptr;       // into `result_block`
    int n_samples = 0;
    const float* latents = nullptr;   // into `result_block` (aur_config.return_latents)
    int n_latent_rows = 0;
    PinBlock* result_block = nullptr;
    int pool_idx = -1;   // latent-pool entry once the tokens are done (vocoder stage)
    bool shared_prefix = false;
    int error = 0;
    bool cancel = false;   // aur_cancel: stop decoding at the next step, skip the vocoder
};

struct ConvLayer {
    const float* wp = nullptr;
    const float* bias = nullptr;
    const void* wp16 = nullptr;
};

class Engine {
public:
    Engine(const aur_config& c, int device) : cfg_(c), device_(device) {
        HIP_CHECK(hipSetDevice(device_));
        // GPT stream at the highest priority, vocoder stream at the lowest: when a vocoder batch overlaps the decode steps
        // of other sequences, the latency-bound decode kernels get the CUs first and the vocoder fills what is left
        int prio_lo = 0, prio_hi = 0;
        HIP_CHECK(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
        HIP_CHECK(hipStreamCreateWithPriority(&st_, hipStreamDefault, prio_hi));   // blocking w.r.t. the null stream on purpose
        AUR_REQUIRE(c.n_layer >= 1 && c.n_layer <= 64, "n_layer in [1,64]");
        AUR_REQUIRE(c.max_seqs >= 1 && c.max_seqs <= 4096, "max_seqs in [1,4096]");
        if (cfg_.max_prefill_rows <= 0) cfg_.max_prefill_rows = 8192;
        if (cfg_.max_speakers <= 0) cfg_.max_speakers = 64;
        const int S = cfg_.max_seqs;
        n_blocks_ = (long)S * kMaxBlocks + 2L * cfg_.max_speakers;   // + 2 shared prefix blocks per speaker
        kv_layer_stride_ = n_blocks_ * kKvBlockElems;
        kv_half_ = cfg_.kv_fp16 != 0;
        // paged_attention_kernel addresses a layer's pool with 32-bit byte offsets: 494 slots with the fp32 pool, 991 with fp16
        // (at 30 layers that is 120 GiB of K/V; HBM holds ~1000 slots' worth next to the weights)
        {
            const long cap = kv_half_ ? 2 * kMaxKvBlocksPerLayer + 1 : kMaxKvBlocksPerLayer;
            if (n_blocks_ > cap)
                throw InvalidArgument("requirement failed: K/V pool: a layer's pool must stay below 4 GiB; with max_speakers = " +
                                      std::to_string(cfg_.max_speakers) + (kv_half_ ? " and kv_fp16" : " and the fp32 pool") + " max_seqs <= " +
                                      std::to_string((cap - 2L * cfg_.max_speakers) / kMaxBlocks) + " (asked for " + std::to_string(S) + ")");
        }
        gemm_prec_ = cfg_.gemm_f32_exact ? 0 : 1;
        kv_.ensure((size_t)cfg_.n_layer * kv_layer_stride_ * (kv_half_ ? 2 : 4));
        for (int b = (int)n_blocks_ - 1; b >= 0; --b) free_blocks_.push_back(b);
        auto ints = [&](DevBuf& b, size_t n) {
            b.ensure(n * sizeof(int));
            HIP_CHECK(hipMemsetAsync(b.p, 0, n * sizeof(int), st_));
        };
        ints(slot_tok_, S); ints(slot_pos_, S); ints(slot_kvpos_, S); ints(slot_ngen_, S); ints(slot_finished_, S);
        ints(temperature_, S); ints(top_p_, S); ints(top_k_, S); ints(rep_, S); ints(max_tokens_, S);
        ints(ignore_stop_, S); ints(seed_, S);
        ints(block_tables_, (size_t)(S + 1) * kMaxBlocks);   // row S: pseudo-slot used to prefill a speaker prefix
        seen_.ensure((size_t)S * kSeenStride);
        HIP_CHECK(hipMemsetAsync(seen_.p, 0, (size_t)S * kSeenStride, st_));
        latents_.ensure((size_t)S * kMaxLatRows * kHidden * sizeof(float));
        spk_table_.ensure((size_t)cfg_.max_speakers * 32 * kHidden * sizeof(float));
        spk_emb_.ensure((size_t)cfg_.max_speakers * 512 * sizeof(float));
        voc_cond_.ensure((size_t)cfg_.max_speakers * kCondStride * sizeof(float));
        zero_bias_.ensure(4096 * sizeof(float));
        HIP_CHECK(hipMemsetAsync(zero_bias_.p, 0, 4096 * sizeof(float), st_));
        ksp_buf_.ensure((size_t)kGemmKspTiles * 16 * 256 * sizeof(float));   // K-split partial tiles of the small-M mlp projection
        ksp_cnt_.ensure((size_t)kGemmKspTiles * sizeof(unsigned));
        HIP_CHECK(hipMemsetAsync(ksp_cnt_.p, 0, (size_t)kGemmKspTiles * sizeof(unsigned), st_));
        h_block_tables_.assign((size_t)(S + 1) * kMaxBlocks, 0);
        spk_info_.assign(cfg_.max_speakers, SpeakerInfo{});
        if (const char* e = getenv("AUR_SHARE_PREFIX")) share_prefix_ = atoi(e) != 0;
        slot_owner_.assign(S, nullptr);
        HIP_CHECK(hipStreamCreateWithPriority(&st2_, hipStreamDefault, prio_hi));
        HIP_CHECK(hipStreamCreateWithPriority(&st_voc_, hipStreamDefault, prio_lo));
        HIP_CHECK(hipEventCreateWithFlags(&ev_lat_, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&ev_voc_done_, hipEventDisableTiming));
        HIP_CHECK(hipEventRecord(ev_lat_, st_));
        latpool_.ensure((size_t)2 * S * kMaxLatRows * kHidden * sizeof(float));
        for (int i = 2 * S - 1; i >= 0; --i) latpool_free_.push_back(i);
        ws_[0].st = st_;
        ws_[1].st = st2_;
        xt_f16_ = cfg_.vocoder_fp16 != 0;   // fp16 storage of the ResBlock intermediates and residual stream (see ConvArgs)
        if (const char* e = getenv("AUR_XT_F16")) xt_f16_ = xt_f16_ && atoi(e) != 0;
        if (const char* e = getenv("AUR_CONV_DMA")) conv_dma_ = atoi(e) != 0;   // 0: register-staged ResBlock convs (A/B only)
        if (const char* e = getenv("AUR_GEMM_PRESPLIT")) gemm_presplit_ = atoi(e) != 0; // 0: prompt-row GEMMs split their weights per tile instead of reading the planes packed at load time (A/B only)
        if (const char* e = getenv("AUR_DECODE_PIPELINE")) pipeline_ = atoi(e) != 0;
        if (const char* e = getenv("AUR_WAV_DIRECT")) wav_direct_ = atoi(e) != 0;   // 0: conv_post writes device memory, one D2H copy per sequence behind it (A/B only)
        if (const char* e = getenv("AUR_SAMPLER_FULL_SORT")) sampler_full_sort_ = atoi(e) != 0;
        if (const char* e = getenv("AUR_HOST_TRACE")) host_trace_ = atoi(e) != 0;
        if (const char* e = getenv("AUR_TEST_FAIL_STEP")) fail_at_step_ = atoi(e);
        if (const char* e = getenv("AUR_TEST_FAIL_VOC")) fail_at_voc_ = atoi(e);

        for (int i = 0; i < 2; ++i) {
            HIP_CHECK(hipEventCreateWithFlags(&ev_rb_[i], hipEventDisableTiming));
            HIP_CHECK(hipEventCreate(&ev_ds_[i]));
            HIP_CHECK(hipEventCreate(&ev_de_[i]));
        }
        HIP_CHECK(hipEventCreate(&ev_a_));
        HIP_CHECK(hipEventCreate(&ev_b_));
        HIP_CHECK(hipStreamSynchronize(st_));
    }
    ~Engine() {
        (void)hipSetDevice(device_);
        (void)hipStreamSynchronize(st_);
        (void)hipStreamSynchronize(st_voc_);
        for (auto* v : {&conv_events_, &gemm_events_})
            for (auto& e : *v) {
                (void)hipEventDestroy(e.a);
                (void)hipEventDestroy(e.b);
            }
        (void)hipEventDestroy(ev_a_);
        (void)hipEventDestroy(ev_b_);
        for (int i = 0; i < 2; ++i) {
            (void)hipEventDestroy(ev_rb_[i]);
            (void)hipEventDestroy(ev_ds_[i]);
            (void)hipEventDestroy(ev_de_[i]);
        }
        if (comm_) (void)Rccl::get().CommDestroy(comm_);
        (void)hipEventDestroy(ev_lat_);
        (void)hipEventDestroy(ev_voc_done_);
        (void)hipStreamDestroy(st_voc_);
        (void)hipStreamDestroy(st2_);
        (void)hipStreamDestroy(st_);
    }

    void use() { HIP_CHECK(hipSetDevice(device_)); }

    // ------------------------------------------------------------------ weights
    void load(const aur_tensor_desc* t, size_t n) {
        use();
        std::lock_guard<std::mutex> gl(gpu_mu_);   // not while a step is running
        std::lock_guard<std::mutex> lk(mu_);       // submit() reads the table sizes under mu_
        for (size_t i = 0; i < n; ++i) {
            AUR_REQUIRE(t[i].name && t[i].data && t[i].numel > 0, "bad tensor desc");
            Tensor& dst = w_[t[i].name];
            if (!dst.buf) dst.buf.reset(new DevBuf());
            dst.buf->ensure((size_t)t[i].numel * sizeof(float));
            dst.numel = t[i].numel;
            HIP_CHECK(hipMemcpy(dst.buf->p, t[i].data, (size_t)t[i].numel * sizeof(float), hipMemcpyHostToDevice));
        }
        voc_ready_ = false;
        gpt_ready_ = false;
    }
    const float* W(const std::string& name, int64_t numel = -1) const {
        auto it = w_.find(name);
        if (it == w_.end()) throw StateError("weight not loaded: " + name);
        if (numel >= 0 && it->second.numel != numel)
            throw InvalidArgument("weight " + name + " has " + std::to_string(it->second.numel) + " elements, expected " +
                                  std::to_string(numel));
        return it->second.d();
    }

    // ------------------------------------------------------------------ conditioning
    // Speaker table (caller holds mu_).  A row is pinned while any sequence submitted with it has not been delivered yet
    // (refs: aur_submit .. end of its vocoder batch, or its failure).  When the table is full the least recently used
    // unpinned voice is evicted; its two shared prefix KV blocks stay with the row and are overwritten by the newcomer.
    int speaker_row(uint64_t key, bool create) {
        auto it = spk_rows_.find(key);
        if (it != spk_rows_.end()) {
            spk_info_[it->second].last_use = ++spk_clock_;
            return it->second;
        }
        if (!create) return -1;
        int row = -1;
        if ((int)spk_rows_.size() < cfg_.max_speakers) {
            std::vector<char> used(cfg_.max_speakers, 0);
            for (auto& kv : spk_rows_) used[kv.second] = 1;
            for (int r = 0; r < cfg_.max_speakers && row < 0; ++r)
                if (!used[r]) row = r;
        } else {
            uint64_t victim = 0;
            for (auto& kv : spk_rows_) {
                const SpeakerInfo& si = spk_info_[kv.second];
                if (si.refs == 0 && (row < 0 || si.last_use < spk_info_[row].last_use)) {
                    row = kv.second;
                    victim = kv.first;
                }
            }
            if (row < 0) throw StateError("speaker table full: every registered voice has undelivered sequences (raise aur_config.max_speakers)");
            spk_rows_.erase(victim);
            spk_info_[row].ready = false;
        }
        spk_rows_[key] = row;
        spk_info_[row].last_use = ++spk_clock_;
        return row;
    }
    // ------------------------------------------------------------------ RCCL inside the boundary
    // The only exchange of the multi-GPU path is the speaker conditioning (133 120 B per voice).  RCCL is resolved at run time
    // (the copy PyTorch already loaded when there is one), so a single-GPU process never needs it.
    struct Rccl {
        typedef struct { char internal[128]; } UniqueId;
        int (*GetUniqueId)(UniqueId*) = nullptr;
        int (*CommInitRank)(void**, int, UniqueId, int) = nullptr;
        int (*Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
        int (*CommDestroy)(void*) = nullptr;
        int (*CommCount)(void*, int*) = nullptr;
        int (*CommUserRank)(void*, int*) = nullptr;
        const char* (*GetErrorString)(int) = nullptr;
        static Rccl& get() {
            static Rccl r = [] {
                Rccl x;
                void* h = nullptr;
                for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"})
                    if ((h = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
                if (!h) throw StateError("RCCL (librccl.so) not found: multi-GPU conditioning broadcast unavailable");
                x.GetUniqueId = reinterpret_cast<decltype(x.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
                x.CommInitRank = reinterpret_cast<decltype(x.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
                x.Broadcast = reinterpret_cast<decltype(x.Broadcast)>(dlsym(h, "ncclBroadcast"));
                x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
                x.CommCount = reinterpret_cast<decltype(x.CommCount)>(dlsym(h, "ncclCommCount"));
                x.CommUserRank = reinterpret_cast<decltype(x.CommUserRank)>(dlsym(h, "ncclCommUserRank"));
                x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
                if (!x.GetUniqueId || !x.CommInitRank || !x.Broadcast || !x.CommDestroy) throw StateError("RCCL symbols missing");
                return x;
            }();
            return r;
        }
        void check(int rc, const char* what) const {
            if (rc != 0) throw StateError(std::string("RCCL ") + what + " failed: " + (GetErrorString ? GetErrorString(rc) : "?"));
        }
    };
    static void comm_unique_id(uint8_t* out) {
        Rccl::UniqueId id;
        Rccl::get().check(Rccl::get().GetUniqueId(&id), "ncclGetUniqueId");
        memcpy(out, id.internal, 128);
    }
    void comm_init(const uint8_t* id_bytes, int rank, int world) {
        std::lock_guard<std::mutex> gl(gpu_mu_);
        use();
        AUR_REQUIRE(world >= 1 && rank >= 0 && rank < world, "comm_init: rank / world");
        if (comm_) throw StateError("communicator already initialised");
        Rccl::UniqueId id;
        memcpy(id.internal, id_bytes, 128);
        Rccl::get().check(Rccl::get().CommInitRank(&comm_, world, id, rank), "ncclCommInitRank");
        comm_rank_ = rank;
        comm_world_ = world;
    }
    // collective: every rank of the communicator calls it; `root` must have the voice registered, the others receive it over
    // xGMI into a device buffer and register it from there (no host bounce)
    void broadcast_conditioning(uint64_t key, int root) {
        constexpr int kCond = 32 * kHidden, kTot = kCond + 512;
        {
            std::lock_guard<std::mutex> gl(gpu_mu_);
            use();
            if (!comm_) throw StateError("aur_comm_init has not been called");
            AUR_REQUIRE(root >= 0 && root < comm_world_, "broadcast_conditioning: root");
            bcast_buf_.ensure((size_t)kTot * sizeof(float));
            float* buf = bcast_buf_.as<float>();
            if (comm_rank_ == root) {
                int row;
                {
                    std::lock_guard<std::mutex> lk(mu_);
                    row = speaker_row(key, false);
                }
                if (row < 0) throw StateError("broadcast_conditioning: the root has no such speaker_key");
                HIP_CHECK(hipMemcpyAsync(buf, spk_table_.as<float>() + (long)row * kCond, (size_t)kCond * sizeof(float), hipMemcpyDeviceToDevice, st_));
                HIP_CHECK(hipMemcpyAsync(buf + kCond, spk_emb_.as<float>() + (long)row * 512, 512 * sizeof(float), hipMemcpyDeviceToDevice, st_));
            }
            Rccl::get().check(Rccl::get().Broadcast(buf, buf, (size_t)kTot, /*ncclFloat32*/ 7, root, comm_, st_), "ncclBroadcast");
            HIP_CHECK(hipStreamSynchronize(st_));
        }
        if (comm_rank_ != root) set_conditioning(key, bcast_buf_.as<float>(), bcast_buf_.as<float>() + kCond, true);
    }

    // what the communicator itself reports (not the values comm_init was called with)
    void comm_info(int32_t* n_ranks, int32_t* rank) {
        std::lock_guard<std::mutex> gl(gpu_mu_);
        *n_ranks = 0;
        *rank = -1;
        if (!comm_) return;
        AUR_REQUIRE(Rccl::get().CommCount && Rccl::get().CommUserRank, "RCCL: ncclCommCount / ncclCommUserRank missing");
        int n = 0, r = -1;
        Rccl::get().check(Rccl::get().CommCount(comm_, &n), "ncclCommCount");
        Rccl::get().check(Rccl::get().CommUserRank(comm_, &r), "ncclCommUserRank");
        *n_ranks = n;
        *rank = r;
    }
    // FNV-1a over the voice's conditioning as registered in device memory
    uint64_t conditioning_checksum(uint64_t key) {
        constexpr int kCond = 32 * kHidden;
        std::vector<float> h((size_t)kCond + 512);
        {
            std::lock_guard<std::mutex> gl(gpu_mu_);
            use();
            int row;
            {
                std::lock_guard<std::mutex> lk(mu_);
                row = speaker_row(key, false);
            }
            if (row < 0) throw InvalidArgument("conditioning_checksum: unknown speaker_key");
            HIP_CHECK(hipMemcpyAsync(h.data(), spk_table_.as<float>() + (long)row * kCond, (size_t)kCond * 4, hipMemcpyDeviceToHost, st_));
            HIP_CHECK(hipMemcpyAsync(h.data() + kCond, spk_emb_.as<float>() + (long)row * 512, 512 * 4, hipMemcpyDeviceToHost, st_));
            HIP_CHECK(hipStreamSynchronize(st_));
        }
        uint64_t x = 1469598103934665603ull;
        const unsigned char* b = reinterpret_cast<const unsigned char*>(h.data());
        for (size_t i = 0; i < h.size() * 4; ++i) {
            x ^= b[i];
            x *= 1099511628211ull;
        }
        return x;
    }

    bool has_conditioning(uint64_t key) {
        std::lock_guard<std::mutex> lk(mu_);
        return speaker_row(key, false) >= 0;
    }
    // Reference audio (mono float32, 22 050 Hz, host memory) -> gpt_cond_latent [32][1024], speaker_embedding [512] on the GPU
    // (cond_net.h).  Needs the "cond.*" tensors of weights.py: pack_conditioning.
    void compute_conditioning(const float* const* pcm, const int32_t* n_samples, int n_refs, const aur_cond_params& cp, float* out_cond,
                              float* out_spk) {
        std::lock_guard<std::mutex> gl(gpu_mu_);   // shares the GPT stream; not while a step is running
        use();
        if (!cond_net_) {
            cond_net_.reset(new CondNet([this](const std::string& n, int64_t numel) { return W(n, numel); },
                                        [this](const std::string& n) { return w_.find(n) != w_.end(); }, st_));
        }
        AUR_REQUIRE(w_.find("cond.enc.init.w") != w_.end(), "conditioning weights not loaded (pack_conditioning)");
        CondParams p;
        p.max_ref_length = cp.max_ref_length;
        p.gpt_cond_len = cp.gpt_cond_len;
        p.gpt_cond_chunk_len = cp.gpt_cond_chunk_len;
        p.sound_norm_refs = cp.sound_norm_refs;
        AUR_REQUIRE(p.max_ref_length > 0 && p.gpt_cond_chunk_len > 0, "conditioning: lengths must be positive");
        std::vector<int> n(n_samples, n_samples + n_refs);
        cond_net_->run(pcm, n.data(), n_refs, p, out_cond, out_spk);
    }
    void set_conditioning(uint64_t key, const float* gpt_cond, const float* spk, bool device_ptrs) {
        std::lock_guard<std::mutex> gl(gpu_mu_);   // may be called while the driver thread is inside aur_step
        use();
        int row;
        {
            std::lock_guard<std::mutex> lk(mu_);
            row = speaker_row(key, true);
            if (spk_info_[row].refs != 0)
                throw StateError("speaker is in use by undelivered sequences: register the new voice under a new key");
        }
        const hipMemcpyKind kind = device_ptrs ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
        HIP_CHECK(hipMemcpyAsync(spk_table_.as<float>() + (long)row * 32 * kHidden, gpt_cond,
                                 32 * kHidden * sizeof(float), kind, st_));
        HIP_CHECK(hipMemcpyAsync(spk_emb_.as<float>() + (long)row * 512, spk, 512 * sizeof(float), kind, st_));
        // 1x1 conditioning convs of the vocoder on the speaker embedding (hifigan_decoder.py:243-251)
        float* dst = voc_cond_.as<float>() + (long)row * kCondStride;
        const float* g = spk_emb_.as<float>() + (long)row * 512;
        launch_gemv_rows(W("voc.cond_layer.w", 512 * 512), W("voc.cond_layer.b", 512), g, dst, 512, 512, 1, 0, 0, st_);
        const int C[4] = {256, 128, 64, 32};
        int off = 512;
        for (int i = 0; i < 4; ++i) {
            const std::string p = "voc.conds." + std::to_string(i);
            launch_gemv_rows(W(p + ".w", (int64_t)C[i] * 512), W(p + ".b", C[i]), g, dst + off, C[i], 512, 1, 0, 0, st_);
            off += C[i];
        }
        HIP_CHECK(hipStreamSynchronize(st_));
        prefill_speaker_prefix(row);
    }
    // Prefix sharing: the first 32 prompt rows (speaker latents, XTTSv2.py:345) are identical for every sequence of a
    // speaker and, under causal attention, so are their K/V in every layer.  They are computed once per speaker into two
    // KV blocks (32 tokens = 2 blocks of 16) that every sequence's block table points at; a prompt prefill then only
    // processes the text rows + start token.  Results are bitwise those of the unshared path (row-independent kernels).
    void prefill_speaker_prefix(int row) {
        SpeakerInfo& si = spk_info_[row];
        si.ready = false;
        if (!share_prefix_ || w_.find("gpt.wte") == w_.end()) return;   // GPT weights not loaded yet: unshared path
        ensure_gpt();
        {
            std::lock_guard<std::mutex> lk(mu_);
            if (si.blocks[0] < 0) {
                AUR_REQUIRE(free_blocks_.size() >= 2, "KV pool exhausted");
                for (int b = 0; b < 2; ++b) {
                    si.blocks[b] = free_blocks_.back();
                    free_blocks_.pop_back();
                }
            }
        }
        RowWs& w = ws_[0];
        last_active_.clear();
        const int S = cfg_.max_seqs;
        std::vector<int4> desc;
        std::vector<int> row_slot(32, S), row_pos(32);
        for (int i = 0; i < 32; ++i) {
            desc.push_back(make_int4(0, i, row, 0));
            row_pos[i] = i;
        }
        ensure_rows(w, 32);
        h_block_tables_[(size_t)S * kMaxBlocks + 0] = si.blocks[0];
        h_block_tables_[(size_t)S * kMaxBlocks + 1] = si.blocks[1];
        HIP_CHECK(hipMemcpyAsync(block_tables_.p, h_block_tables_.data(), h_block_tables_.size() * 4, hipMemcpyHostToDevice, w.st));
        HIP_CHECK(hipMemcpyAsync(w.i_desc.p, desc.data(), 32 * sizeof(int4), hipMemcpyHostToDevice, w.st));
        HIP_CHECK(hipMemcpyAsync(w.i_row_slot.p, row_slot.data(), 32 * 4, hipMemcpyHostToDevice, w.st));
        HIP_CHECK(hipMemcpyAsync(w.i_row_pos.p, row_pos.data(), 32 * 4, hipMemcpyHostToDevice, w.st));
        launch_embed_prompt(w.i_desc.as<int4>(), spk_table_.as<float>(), text_emb_, text_pos_, wte_, wpe_, w.h.as<float>(), 32, w.st);
        forward_rows(w, 32, w.i_row_slot.as<int>(), w.i_row_pos.as<int>(), upload_qblocks(w, row_slot, row_pos));
        HIP_CHECK(hipStreamSynchronize(w.st));
        si.ready = true;
    }

    // ------------------------------------------------------------------ submit / poll
    uint64_t submit(const aur_seq_desc& d) {
        AUR_REQUIRE(d.text_ids && d.n_text >= 1, "text_ids");
        AUR_REQUIRE(d.max_tokens >= 1 && d.max_tokens <= kMaxLatRows - 3, "max_tokens in [1,605]");
        const int n_prompt = 32 + d.n_text + 1;
        AUR_REQUIRE(n_prompt + d.max_tokens + 4 <= kMaxBlocks * kKvBlockTokens, "prompt + max_tokens exceeds max_model_len");
        AUR_REQUIRE(n_prompt <= cfg_.max_prefill_rows, "prompt longer than max_prefill_rows");
        AUR_REQUIRE(d.repetition_penalty > 0.f, "repetition_penalty > 0");
        std::lock_guard<std::mutex> lk(mu_);
        {   // reject bad text ids / positions here, not in the middle of a batched prefill
            auto te = w_.find("text_emb"), tp = w_.find("text_pos");
            if (te == w_.end() || tp == w_.end()) throw StateError("weights not loaded");
            const int vocab = (int)(te->second.numel / kHidden), npos = (int)(tp->second.numel / kHidden);
            AUR_REQUIRE(d.n_text <= npos, "text longer than the text position table");
            for (int i = 0; i < d.n_text; ++i) AUR_REQUIRE(d.text_ids[i] >= 0 && d.text_ids[i] < vocab, "text id out of range");
        }
        const int row = speaker_row(d.speaker_key, false);
        AUR_REQUIRE(row >= 0, "unknown speaker_key (call aur_set_conditioning first)");
        spk_info_[row].refs++;
        auto s = std::make_unique<Seq>();
        s->id = next_id_++;
        s->text_ids.assign(d.text_ids, d.text_ids + d.n_text);
        s->spk_row = row;
        s->params = d;
        s->params.text_ids = nullptr;
        s->n_prompt = n_prompt;
        const uint64_t id = s->id;
        if (d.priority > 0) {   // aur_seq_desc.priority: behind the urgent sequences already waiting, in front of everything else
            auto it = waiting_.begin();
            while (it != waiting_.end() && (*it)->params.priority > 0) ++it;
            waiting_.insert(it, s.get());
        } else {
            waiting_.push_back(s.get());
        }
        seqs_[id] = std::move(s);
        return id;
    }

    size_t poll(aur_result* out, size_t cap) {
        std::lock_guard<std::mutex> lk(mu_);
        size_t n = 0;
        while (n < cap && !done_.empty()) {
            Seq* s = done_.front();
            done_.pop_front();
            aur_result& r = out[n++];
            r.seq_id = s->id;
            r.n_tokens = (int)s->tokens.size();
            r.tokens = s->tokens.data();
            r.n_samples = s->n_samples;
            r.wav = s->wav;
            r.n_latent_rows = s->n_latent_rows;
            r.latents = s->latents;
            r.error = s->error;
        }
        return n;
    }
    void release(uint64_t id) {
        std::lock_guard<std::mutex> lk(mu_);
        auto it = seqs_.find(id);
        AUR_REQUIRE(it != seqs_.end(), "unknown seq_id");
        AUR_REQUIRE(it->second->state == SeqState::DONE, "sequence not finished");
        if (it->second->result_block) it->second->result_block->refs--;
        seqs_.erase(it);
    }

    // aur_cancel.  Takes only the queue lock: a WAITING sequence is dropped here; everything else is noted and handled by the driver
    // thread at the top of its next aur_step (process_cancels), so the caller -- the facade's event loop -- never waits for a step.
    void cancel(uint64_t id) {
        std::lock_guard<std::mutex> lk(mu_);
        auto it = seqs_.find(id);
        AUR_REQUIRE(it != seqs_.end(), "unknown seq_id");
        Seq* s = it->second.get();
        if (s->state == SeqState::WAITING) {
            waiting_.erase(std::remove(waiting_.begin(), waiting_.end(), s), waiting_.end());
            finish_cancelled(s);
        } else if (s->state == SeqState::RUNNING || s->state == SeqState::TOKENS_DONE) {
            if (!s->cancel) cancel_pending_.push_back(id);
        }
    }
    // (caller holds mu_; the sequence holds no slot, K/V block or pool entry any more)
    void finish_cancelled(Seq* s) {
        s->cancel = true;
        s->error = AUR_E_CANCELLED;
        s->wav = nullptr; s->n_samples = 0; s->latents = nullptr; s->n_latent_rows = 0;
        spk_info_[s->spk_row].refs--;
        s->state = SeqState::DONE;
        done_.push_back(s);
        finished_total_++;
    }
    // driver thread, inside aur_step (gpu_mu_ held): a running sequence gets max_tokens = 1 on the device -- the sampler then flags it
    // finished with the next token it draws, and finish_tokens() drops it instead of parking its latents; a sequence that waits in the
    // vocoder queue leaves it.  One already inside a vocoder batch is delivered normally.
    void process_cancels() {
        std::vector<uint64_t> ids;
        {
            std::lock_guard<std::mutex> lk(mu_);
            ids.swap(cancel_pending_);
        }
        for (uint64_t id : ids) {
            std::lock_guard<std::mutex> lk(mu_);
            auto it = seqs_.find(id);
            if (it == seqs_.end()) continue;
            Seq* s = it->second.get();
            if (s->state == SeqState::RUNNING && s->slot >= 0 && !s->cancel) {
                s->cancel = true;
                HIP_CHECK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(max_tokens_.as<int>() + s->slot), 1, 1, st_));
            } else if (s->state == SeqState::TOKENS_DONE) {
                auto q = std::find(voc_queue_.begin(), voc_queue_.end(), s);
                if (q != voc_queue_.end()) {
                    voc_queue_.erase(q);
                    if (s->pool_idx >= 0) latpool_free_.push_back(s->pool_idx);
                    s->pool_idx = -1;
                    finish_cancelled(s);
                }
            }
        }
    }

    // ------------------------------------------------------------------ one scheduler iteration
    // A failure inside a step (HIP error, exhausted pool) must not leave slots, KV blocks and pool entries half-updated:
    // every sequence that was in flight is failed (aur_result.error), its resources go back to the pools, and the
    // exception continues to the caller.  The engine stays usable if the device is.
    void step(int* n_live, int* n_finished_total) {
        std::lock_guard<std::mutex> gl(gpu_mu_);
        try {
            step_locked(n_live, n_finished_total);
        } catch (const InvalidArgument&) {
            fail_in_flight(AUR_E_INVALID);
            throw;
        } catch (const StateError&) {
            fail_in_flight(AUR_E_STATE);
            throw;
        } catch (...) {
            fail_in_flight(AUR_E_HIP);
            throw;
        }
    }
    void fail_in_flight(int code) {
        (void)hipStreamSynchronize(st_);
        (void)hipStreamSynchronize(st_voc_);
        (void)hipGetLastError();
        (void)hipMemsetAsync(ksp_cnt_.p, 0, (size_t)kGemmKspTiles * sizeof(unsigned), st_);   // an aborted K-split launch may have left tickets behind
        infl_.on = false;
        just_finished_.clear();
        last_active_.clear();
        n_gemm_events_ = 0;
        n_conv_events_ = 0;
        voc_timed_ = false;
        std::lock_guard<std::mutex> lk(mu_);
        auto fail = [&](Seq* s) {
            for (int b : s->blocks) free_blocks_.push_back(b);
            s->blocks.clear();
            if (s->pool_idx >= 0) latpool_free_.push_back(s->pool_idx);
            s->pool_idx = -1;
            if (s->slot >= 0) slot_owner_[s->slot] = nullptr;
            s->slot = -1;
            s->error = code;
            s->wav = nullptr; s->n_samples = 0; s->latents = nullptr; s->n_latent_rows = 0;
            spk_info_[s->spk_row].refs--;
            s->state = SeqState::DONE;
            done_.push_back(s);
            finished_total_++;
        };
        for (Seq* s : slot_owner_)
            if (s) fail(s);
        for (Seq* s : voc_queue_) fail(s);
        voc_queue_.clear();
        // the vocoder batch, whether it had been launched completely (voc_active_) or the throw came from inside voc_launch
        // after the sequences had left the queue
        for (Seq* s : voc_batch_)
            if (s->state != SeqState::DONE) fail(s);
        if (voc_block_held_ && voc_block_) voc_block_->refs--;
        voc_block_held_ = false;
        voc_batch_.clear();
        voc_active_ = false;
    }
    void step_locked(int* n_live, int* n_finished_total) {
        use();
        ensure_gpt();
        if (fail_at_step_ > 0 && --fail_at_step_ == 0) {   // AUR_TEST_FAIL_STEP=n: fault injection for the recovery test
            std::vector<Seq*> dummy;
            throw HipError("injected failure (AUR_TEST_FAIL_STEP)");
        }
        bool worked = false;
        process_cancels();
        // back-pressure: every running sequence must be able to park its latents when it finishes
        while ((int)latpool_free_.size() < cfg_.max_seqs) {
            if (voc_active_)
                voc_poll(true);
            else if (!voc_queue_.empty())
                voc_launch();
            else
                break;
        }
        // 1. admission + prefill
        std::vector<Seq*> admitted;
        {
            std::lock_guard<std::mutex> lk(mu_);
            int rows = 0;
            // aur_config.admit_min_batch: a prefill pass costs ~150 launches whatever its rows; with a queue longer than the free
            // slots, wait until a group of slots is free (the decode step costs the same at 56 rows as at 64)
            int free_slots = 0;
            for (int i = 0; i < cfg_.max_seqs; ++i)
                if (!slot_owner_[i]) ++free_slots;
            // (default: an eighth of the slots -- the group size that balances the pass's fixed cost against the idle slot-steps grows
            // with the slot count: g* ~ slots * sqrt(2 * pass_ms / (step_ms * tokens per sequence)) ~ slots / 7 at 30 layers)
            const int group = std::min(cfg_.admit_min_batch > 0 ? cfg_.admit_min_batch : std::max(1, cfg_.max_seqs / 8), cfg_.max_seqs);
            // ... but never for long: with few finishes in sight the free slots would idle while requests wait (kAdmitHoldSteps decode
            // steps ~ 60 ms at 30 layers, then whatever fits is admitted)
            // aur_seq_desc.priority (latency-critical sequences: the head chunk of a stream somebody is waiting to hear).  While one of
            // them waits or runs, (1) it is admitted at once, whatever the group logic says; (2) the others are admitted only up to
            // `urgent_rows` running sequences -- a decode step's time grows with its rows (attention above all: 1.6 ms at 64 rows, ~3 ms
            // at 192 and 30 layers), and every step of the urgent sequence pays it -- and only in groups (a prefill pass in front of
            // every step would stall the urgent sequence's steps for the pass's ~4.5 ms of launches each time).
            int urgent_waiting = 0, n_running = cfg_.max_seqs - free_slots;
            bool urgent_running = false;
            for (Seq* q : waiting_) urgent_waiting += q->params.priority > 0 ? 1 : 0;
            for (int i = 0; i < cfg_.max_seqs; ++i)
                if (slot_owner_[i] && slot_owner_[i]->params.priority > 0 && slot_owner_[i]->state == SeqState::RUNNING) urgent_running = true;
            const bool urgent = urgent_waiting > 0 || urgent_running;
            // (default a quarter of the slots; measured on the long-form stream at 192 slots, profiles/r06_c5s_urgent_sweep.log: first chunk
            // 0.55 s / 32.2 M samples/s without a cap, 0.43 / 32.3 at 128 rows, 0.42 / 31.3 at 96, 0.33 / 31.5 at 64, 0.30 / 31.5 at 48)
            const int urgent_cap = std::min(cfg_.max_seqs, cfg_.urgent_rows > 0 ? cfg_.urgent_rows : std::max(1, cfg_.max_seqs / 4));
            const int n_plain = (int)waiting_.size() - urgent_waiting;
            const int free_eff = urgent ? std::max(0, std::min(free_slots, urgent_cap - n_running)) : free_slots;
            bool hold = free_slots < cfg_.max_seqs && free_eff < std::min(group, n_plain);
            // (an urgent sequence running beside a trickle of arrivals: wait for a group or kUrgentHoldSteps, not a pass per step)
            if (!hold && urgent_running && n_plain > 0 && n_plain < group && free_eff > 0) hold = true;
            if (hold && free_eff > 0 && ++admit_hold_steps_ > (urgent_running ? kUrgentHoldSteps : kAdmitHoldSteps)) hold = false;
            if (!hold) admit_hold_steps_ = 0;
            while (!waiting_.empty()) {
                Seq* s = waiting_.front();
                const bool s_urgent = s->params.priority > 0;
                if (!s_urgent && (hold || (urgent && n_running + (int)admitted.size() >= urgent_cap))) break;
                const SpeakerInfo& si = spk_info_[s->spk_row];
                s->shared_prefix = si.ready && share_prefix_now_;
                const int skip = s->shared_prefix ? 2 : 0;   // table entries 0,1 = the speaker's shared prefix blocks
                const int extra = cfg_.second_pass ? 4 : 0;   // second pass appends 4 stop tokens after the generated ones
                const int need = (s->n_prompt + s->params.max_tokens + extra + kKvBlockTokens - 1) / kKvBlockTokens - skip;
                int slot = -1;
                for (int i = 0; i < cfg_.max_seqs; ++i)
                    if (!slot_owner_[i]) {
                        slot = i;
                        break;
                    }
                const int n_rows = s->n_prompt - (s->shared_prefix ? 32 : 0);
                if (slot < 0 || (int)free_blocks_.size() < need || rows + n_rows > cfg_.max_prefill_rows) break;
                waiting_.pop_front();
                s->slot = slot;
                slot_owner_[slot] = s;
                for (int b = 0; b < skip; ++b) h_block_tables_[(size_t)slot * kMaxBlocks + b] = si.blocks[b];
                for (int b = 0; b < need; ++b) {
                    s->blocks.push_back(free_blocks_.back());
                    free_blocks_.pop_back();
                    h_block_tables_[(size_t)slot * kMaxBlocks + skip + b] = s->blocks.back();
                }
                s->state = SeqState::RUNNING;
                rows += n_rows;
                admitted.push_back(s);
            }
        }
        if (!admitted.empty()) {
            HIP_CHECK(hipEventRecord(ev_a_, st_));
            prefill(admitted);   // synchronises the stream before it returns
            HIP_CHECK(hipEventRecord(ev_b_, st_));
            HIP_CHECK(hipEventSynchronize(ev_b_));
            float ms = 0.f;
            HIP_CHECK(hipEventElapsedTime(&ms, ev_a_, ev_b_));
            stats_.gpt_ms += ms;
            stats_.prefill_ms += ms;
            stats_.prefill_batches++;
            worked = true;
        }
        // 2. decode step for every running sequence (GPU time is accounted per collected step inside decode)
        std::vector<int> active;
        for (int i = 0; i < cfg_.max_seqs; ++i)
            if (slot_owner_[i] && slot_owner_[i]->state == SeqState::RUNNING) active.push_back(i);
        if (!active.empty()) {
            HIP_CHECK(hipEventRecord(ev_a_, st_));
            decode(active);
            worked = true;
        }
        {
            bool any_running = false;
            for (int i = 0; i < cfg_.max_seqs; ++i)
                if (slot_owner_[i] && slot_owner_[i]->state == SeqState::RUNNING) any_running = true;
            if (!any_running) drain_inflight();
        }
        if (worked) stats_.steps++;
        // 3. vocoder stage (asynchronous): finished sequences left their slots already (latents parked in the pool);
        //    one vocoder batch is in flight on its own stream while the next GPT steps run on the main stream.
        voc_poll(false);
        int running = 0;
        for (int i = 0; i < cfg_.max_seqs; ++i)
            if (slot_owner_[i]) ++running;
        size_t n_wait;
        {
            std::lock_guard<std::mutex> lk(mu_);
            n_wait = waiting_.size();
        }
        // aur_config.vocoder_min_batch: a vocoder pass of one utterance costs twice per utterance what a pass of eight does (and
        // slows the decode steps it runs beside); finished sequences wait for company, at most kVocHoldSteps steps
        const int minb = cfg_.vocoder_min_batch > 0 ? cfg_.vocoder_min_batch : std::max(1, cfg_.max_seqs / 16);
        // (the hold runs from the step the oldest queued sequence finished, also while another batch is in flight: reset only when
        // nothing waits, so a sequence queued behind an active batch is launched as soon as that batch is done once its hold is up)
        if (voc_queue_.empty()) voc_hold_steps_ = 0;
        else ++voc_hold_steps_;
        bool urgent_voc = false;   // an urgent sequence does not wait for company: its batch is whatever has finished with it
        for (Seq* q : voc_queue_) urgent_voc |= q->params.priority > 0;
        if (!voc_active_ && !voc_queue_.empty() &&
            ((int)voc_queue_.size() >= minb || urgent_voc || (running == 0 && n_wait == 0) || voc_hold_steps_ > kVocHoldSteps)) {
            voc_launch();
            voc_hold_steps_ = 0;
        }
        if (running == 0 && n_wait == 0 && voc_active_) voc_poll(true);   // nothing else to do: wait for the batch
        if (running == 0 && n_wait == 0 && !voc_active_ && !voc_queue_.empty()) {
            voc_launch();
            voc_poll(true);
        }
        {
            std::lock_guard<std::mutex> lk(mu_);
            int live = (int)waiting_.size() + (int)voc_queue_.size() + (voc_active_ ? (int)voc_batch_.size() : 0);
            for (int i = 0; i < cfg_.max_seqs; ++i)
                if (slot_owner_[i]) ++live;
            if (n_live) *n_live = live;
            if (n_finished_total) *n_finished_total = (int)finished_total_;
        }
    }

    // ------------------------------------------------------------------ standalone vocoder
    void vocode_host(const float* latents, const int* n_lat, int B, int t_max, uint64_t key, float* wav_out,
                     int64_t wav_stride, int* n_samples_out) {
        use();
        ensure_voc();
        const int row = speaker_row(key, false);
        AUR_REQUIRE(row >= 0, "unknown speaker_key");
        AUR_REQUIRE(B >= 1 && t_max >= 1, "B, t_max");
        AUR_REQUIRE(!voc_active_, "vocoder stage busy");
        std::vector<int> nl(n_lat, n_lat + B), cond(B, row);
        int max_samples = 0;
        for (int b = 0; b < B; ++b) {
            AUR_REQUIRE(nl[b] >= 1 && nl[b] <= t_max, "n_lat in [1,t_max]");
            max_samples = std::max(max_samples, frames_for(nl[b]) * 256);
        }
        AUR_REQUIRE(wav_stride >= max_samples, "wav_stride too small");
        tmp_lat_.ensure((size_t)B * t_max * kHidden * sizeof(float));
        HIP_CHECK(hipMemcpyAsync(tmp_lat_.p, latents, (size_t)B * t_max * kHidden * sizeof(float),
                                 hipMemcpyHostToDevice, st_voc_));
        tmp_wav_.ensure((size_t)B * max_samples * sizeof(float));
        run_vocoder(B, nl, tmp_lat_.as<float>(), (long)t_max * kHidden, nullptr, cond, tmp_wav_.as<float>(),
                    max_samples);
        HIP_CHECK(hipStreamSynchronize(st_voc_));
        collect_conv_events();
        for (int b = 0; b < B; ++b) {
            const int ns = frames_for(nl[b]) * 256;
            HIP_CHECK(hipMemcpy(wav_out + (long)b * wav_stride, tmp_wav_.as<float>() + (long)b * max_samples,
                                (size_t)ns * sizeof(float), hipMemcpyDeviceToHost));
            if (n_samples_out) n_samples_out[b] = ns;
        }
    }

    void sync() {
        use();
        HIP_CHECK(hipStreamSynchronize(st_));
        HIP_CHECK(hipStreamSynchronize(st_voc_));
    }
    aur_stats stats() {
        std::lock_guard<std::mutex> lk(mu_);
        aur_stats s = stats_;
        s.kv_blocks_total = n_blocks_;
        s.kv_blocks_free = (int64_t)free_blocks_.size();
        s.result_blocks = (int64_t)result_blocks_.size();
        s.result_blocks_free = 0;
        s.result_block_bytes = 0;
        for (auto& b : result_blocks_) {
            if (b->refs == 0) s.result_blocks_free++;
            s.result_block_bytes += (int64_t)b->buf.bytes;
        }
        s.speakers = (int64_t)spk_rows_.size();
        s.sequences_tracked = (int64_t)seqs_.size();
        return s;
    }
    void reset_stats() {
        std::lock_guard<std::mutex> lk(mu_);
        stats_ = aur_stats{};
    }
    void set_profile(int every) {
        std::lock_guard<std::mutex> gl(gpu_mu_);   // not while a step is running
        AUR_REQUIRE(every >= 0, "aur_set_profile: every >= 0");
        cfg_.profile = every > 0 ? 1 : 0;
        profile_every_ = every > 0 ? every : kProfileEvery;
        decode_step_count_ = 0;
    }

    // ------------------------------------------------------------------ debug entry points
    // prefill-regime GEMM (gemm_tile_kernel / gemm_tile_split_kernel, the arithmetic the engine was configured with) on host data: out = X @ W
    void dbg_gemm(const float* X, const float* Wm, float* out, int M, int N, int K) {
        use();
        DevBuf dx, dw, dp;
        dx.ensure((size_t)M * K * 4);
        dw.ensure((size_t)K * N * 4);
        dp.ensure((size_t)M * N * 4);
        HIP_CHECK(hipMemcpy(dx.p, X, (size_t)M * K * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(dw.p, Wm, (size_t)K * N * 4, hipMemcpyHostToDevice));
        DevBuf ds;   // the pre-split weight operand, as ensure_gpt() prepares it for the prompt-row GEMMs
        if (gemm_prec_ == 1 && gemm_presplit_ && N % 128 == 0) {
            ds.ensure((size_t)3 * K * N * 2);
            launch_pack_wsplit(dw.as<float>(), N, ds.p, K, N, st_);
        }
        launch_gemm_tile(dx.as<float>(), K, dw.as<float>(), dp.as<float>(), M, N, K, st_, nullptr, gemm_prec_, 1, ds.p);
        HIP_CHECK(hipStreamSynchronize(st_));
        HIP_CHECK(hipMemcpy(out, dp.p, (size_t)M * N * 4, hipMemcpyDeviceToHost));
    }
    // decode-regime GEMM on host data: out = epi(LN?(X) @ W + bias); W is the plain [K][N] matrix, packed on the device by
    // the same pack_wt16 the engine uses at load time.  epi: 0 bias, 1 bias + gelu_new, 2 out += (X @ W + bias).
    void dbg_gemm_rows(const float* X, const float* Wm, const float* bias, const float* gamma, const float* beta, float* out,
                       int M, int N, int K, int epi, bool ln) {
        use();
        AUR_REQUIRE(epi >= 0 && epi <= 2, "dbg_gemm_rows: epi in {0,1,2}");
        AUR_REQUIRE(!ln || (gamma && beta), "dbg_gemm_rows: LN needs gamma and beta");
        DevBuf dx, dw, dwt, db, dg, dbe, dout;
        const int mtt = (M + 63) / 64 * 4;                      // 16-row tiles of the packed activation buffers
        const bool out_packed = (epi == kEpiBiasGelu || epi == kEpiResidual);
        std::vector<float> hx((size_t)mtt * 16 * K, 0.f), ho((size_t)mtt * 16 * N, 0.f);
        for (int m = 0; m < M; ++m)
            for (int k = 0; k < K; ++k) hx[pk_off(m, k, mtt)] = X[(size_t)m * K + k];
        for (int m = 0; m < M; ++m)
            for (int n = 0; n < N; ++n) ho[out_packed ? pk_off(m, n, mtt) : (size_t)m * N + n] = out[(size_t)m * N + n];
        dx.ensure(hx.size() * 4);
        dw.ensure((size_t)K * N * 4);
        dwt.ensure((size_t)K * N * 4);
        dout.ensure(ho.size() * 4);
        HIP_CHECK(hipMemcpy(dx.p, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(dw.p, Wm, (size_t)K * N * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(dout.p, ho.data(), ho.size() * 4, hipMemcpyHostToDevice));
        if (bias) {
            db.ensure((size_t)N * 4);
            HIP_CHECK(hipMemcpy(db.p, bias, (size_t)N * 4, hipMemcpyHostToDevice));
        }
        DevBuf dst;
        if (ln) {
            // LayerNorm partials per 16-column tile, as the residual-stream producers emit them (GemmRowsArgs)
            std::vector<float2> hs((size_t)mtt * 16 * 64, make_float2(0.f, 0.f));
            for (int m = 0; m < M; ++m)
                for (int t = 0; t < 64; ++t) {
                    float sm = 0.f, m2 = 0.f;
                    for (int c = 0; c < 16; ++c) sm += X[(size_t)m * K + 16 * t + c];
                    const float mu = sm * (1.0f / 16.0f);
                    for (int c = 0; c < 16; ++c) {
                        const float d = X[(size_t)m * K + 16 * t + c] - mu;
                        m2 += d * d;
                    }
                    hs[(size_t)m * 64 + t] = make_float2(mu, m2);
                }
            dst.ensure(hs.size() * sizeof(float2));
            HIP_CHECK(hipMemcpy(dst.p, hs.data(), hs.size() * sizeof(float2), hipMemcpyHostToDevice));
            dg.ensure((size_t)K * 4);
            dbe.ensure((size_t)K * 4);
            HIP_CHECK(hipMemcpy(dg.p, gamma, (size_t)K * 4, hipMemcpyHostToDevice));
            HIP_CHECK(hipMemcpy(dbe.p, beta, (size_t)K * 4, hipMemcpyHostToDevice));
        }
        GemmRowsArgs a{};
        DevBuf dsc, dvec;
        if (ln) {   // LayerNorm folded into the weights, as ensure_gpt() does at load time
            dsc.ensure((size_t)K * N * 4);
            dvec.ensure((size_t)2 * N * 4);
            launch_fold_ln(dw.as<float>(), N, dg.as<float>(), dbe.as<float>(), bias ? db.as<float>() : nullptr, dsc.as<float>(),
                           dwt.as<float>(), dvec.as<float>(), dvec.as<float>() + N, K, N, st_);
            a.ln_c1 = dvec.as<float>();
            a.bias = dvec.as<float>() + N;
        } else {
            launch_pack_wt16(dw.as<float>(), N, dwt.as<float>(), K, N, st_);
            if (!bias) {   // the kernel loads its bias unconditionally
                db.ensure((size_t)N * 4);
                HIP_CHECK(hipMemsetAsync(db.p, 0, (size_t)N * 4, st_));
            }
            a.bias = db.as<float>();
        }
        a.X = dx.as<float>(); a.xmt = mtt; a.Wt = dwt.as<float>(); a.M = M; a.N = N; a.K = K;
        a.eps = 1e-5f; a.prec = gemm_prec_;
        if ((long)((M + 15) / 16) * (N / 16) <= kGemmKspTiles) { a.ksp_buf = ksp_buf_.as<float>(); a.ksp_cnt = ksp_cnt_.as<unsigned>(); }
        a.stats_in = ln ? dst.as<float2>() : nullptr;
        a.out = dout.as<float>(); a.ldo = N; a.omt = mtt;
        launch_gemm_rows(a, ln, (GemmRowsEpi)epi, st_);
        HIP_CHECK(hipStreamSynchronize(st_));
        HIP_CHECK(hipMemcpy(ho.data(), dout.p, ho.size() * 4, hipMemcpyDeviceToHost));
        for (int m = 0; m < M; ++m)
            for (int n = 0; n < N; ++n) out[(size_t)m * N + n] = ho[out_packed ? pk_off(m, n, mtt) : (size_t)m * N + n];
    }
    // Stress of the K-split projection's cross-workgroup protocol (gemm_rows_kernel.inc, KSP: write-through partial tiles, drained
    // vmcnt, device-scope ticket, sc0 sc1 read-back by the last arriver, counter reset for the next launch): `iters` back-to-back
    // launches of the K = 4096 -> 1024 residual GEMM at M rows, no host synchronisation in between, on two alternating operand
    // sets and beside a second stream that streams 256 MiB per launch, each compared word for word on the device with the UNSPLIT
    // kernel's result on the matching operands.  A visibility bug shows up as a non-zero count (dbg_gemm_rows_ksplit_stress).
    long long dbg_lane_xor_selftest(int blocks) {
        use();
        AUR_REQUIRE(blocks >= 1 && blocks <= 4096, "lane_xor selftest: 1..4096 workgroups");
        DevBuf dcnt;
        dcnt.ensure(8);
        HIP_CHECK(hipMemsetAsync(dcnt.p, 0, 8, st_));
        launch_lane_xor_selftest(0x9E3779B9u, blocks, dcnt.as<unsigned long long>(), st_);
        unsigned long long bad = 0;
        HIP_CHECK(hipMemcpyAsync(&bad, dcnt.p, 8, hipMemcpyDeviceToHost, st_));
        HIP_CHECK(hipStreamSynchronize(st_));
        return (long long)bad;
    }
    long long dbg_gemm_rows_ksplit_stress(int M, int iters) {
        use();
        const int K = 4 * kHidden, N = kHidden;
        AUR_REQUIRE(M >= 1 && M <= 32 && iters >= 1, "ksplit stress: the split is used for M <= 32 rows");
        AUR_REQUIRE(gemm_rows_shape(M, N, K, false).ksp > 1, "ksplit stress: the policy does not split at this M");
        // TWO operand sets, alternated launch by launch: consecutive launches publish DIFFERENT partial tiles into ksp_buf, so a
        // last arriver that read a stale partial (the previous launch's, left in a cache the sc0 sc1 read-back failed to bypass, or
        // a ticket that ran ahead of its data) produces the wrong words -- on identical operands it would have produced the right ones.
        const int mtt = 4;
        const size_t nx = (size_t)mtt * 16 * K, nh = (size_t)mtt * 16 * N;
        std::vector<float> hx(2 * nx), hw((size_t)K * N), hb(N), hh(2 * nh);
        unsigned sd = 12345u + (unsigned)M;
        auto rnd = [&] { sd = sd * 1664525u + 1013904223u; return (float)(sd >> 8) * (1.0f / 8388608.0f) - 1.0f; };
        for (auto& v : hx) v = rnd();
        for (auto& v : hw) v = 0.05f * rnd();
        for (auto& v : hb) v = 0.1f * rnd();
        for (auto& v : hh) v = rnd();
        DevBuf dx, dw, dwt, db, dh0, dh, dref, dcnt, dload, dsink;
        dx.ensure(hx.size() * 4); dw.ensure(hw.size() * 4); dwt.ensure(hw.size() * 4); db.ensure(hb.size() * 4);
        dh0.ensure(hh.size() * 4); dh.ensure(nh * 4); dref.ensure(hh.size() * 4); dcnt.ensure(8);
        HIP_CHECK(hipMemcpy(dx.p, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(dw.p, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(db.p, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(dh0.p, hh.data(), hh.size() * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemsetAsync(dcnt.p, 0, 8, st_));
        launch_pack_wt16(dw.as<float>(), N, dwt.as<float>(), K, N, st_);
        GemmRowsArgs a{};
        a.xmt = mtt; a.omt = mtt; a.Wt = dwt.as<float>(); a.M = M; a.N = N; a.K = K; a.bias = db.as<float>(); a.prec = gemm_prec_;
        // references: no scratch -> the UNSPLIT kernel, one per operand set
        HIP_CHECK(hipMemcpyAsync(dref.p, dh0.p, hh.size() * 4, hipMemcpyDeviceToDevice, st_));
        for (int s = 0; s < 2; ++s) {
            a.X = dx.as<float>() + s * nx; a.out = dref.as<float>() + s * nh;
            launch_gemm_rows(a, false, kEpiResidual, st_);
        }
        // background load on the low-priority stream for the whole test: 256 MiB streamed per launch by 1024 workgroups (256 KiB
        // each), a launch enqueued for every second split launch (~50 us of traffic against ~25 us per iteration), so that the
        // split kernel's workgroups start unevenly and its partials, tickets and read-backs share L2 and the fabric with other traffic
        const size_t load_bytes = (size_t)256 << 20;
        dload.ensure(load_bytes);
        dsink.ensure(1024 * 4);
        HIP_CHECK(hipMemsetAsync(dload.p, 0, load_bytes, st_voc_));
        HIP_CHECK(hipMemsetAsync(dsink.p, 0, 1024 * 4, st_voc_));
        HIP_CHECK(hipStreamSynchronize(st_));
        a.ksp_buf = ksp_buf_.as<float>(); a.ksp_cnt = ksp_cnt_.as<unsigned>();
        auto split_launch = [&](int s) {
            HIP_CHECK(hipMemcpyAsync(dh.p, dh0.as<float>() + s * nh, nh * 4, hipMemcpyDeviceToDevice, st_));
            a.X = dx.as<float>() + s * nx; a.out = dh.as<float>();
            launch_gemm_rows(a, false, kEpiResidual, st_);
        };
        for (int i = 0; i < iters; ++i) {
            const int s = i & 1;
            if ((i & 1) == 0)
                hipLaunchKernelGGL(stress_load_kernel, dim3(1024), dim3(256), 0, st_voc_, dload.as<float4>(), dsink.as<float>(), (long)(load_bytes / 16));
            if (i % 3 == 2) {   // two split launches on DIFFERENT data with nothing between them but the residual copy: the counter reset
                split_launch(s ^ 1);   // and the scratch reuse are on the path; only the second one is checked
            }
            split_launch(s);
            launch_count_mismatch(dh.p, dref.as<float>() + s * nh, (long)nh, dcnt.as<unsigned long long>(), st_);
        }
        HIP_CHECK(hipGetLastError());
        unsigned long long bad = 0;
        HIP_CHECK(hipMemcpyAsync(&bad, dcnt.p, 8, hipMemcpyDeviceToHost, st_));
        HIP_CHECK(hipStreamSynchronize(st_));
        HIP_CHECK(hipStreamSynchronize(st_voc_));
        return (long long)bad;
    }
    void dbg_paged_attention(const float* q, const float* k, const float* v, const int* ctx, int M, int ctx_max, int shared, bool half,
                             float* out) {
        use();
        AUR_REQUIRE(M >= 1 && M <= 256 && ctx_max >= 1 && ctx_max <= kMaxBlocks * kKvBlockTokens, "dbg_paged_attention: 1..256 rows, context 1..1056");
        AUR_REQUIRE(shared == 0 || shared == 16 || shared == 32, "dbg_paged_attention: shared prefix of 0, 16 or 32 tokens");
        const int n_shared = shared / kKvBlockTokens, per_row = (ctx_max + kKvBlockTokens - 1) / kKvBlockTokens;
        const long n_blocks = 1 + n_shared + (long)M * per_row;   // (block 0 stays empty: what a table entry past a row's blocks names)
        std::vector<int> rm((size_t)M * kRowMetaStride, 0);
        std::vector<float> pool((size_t)n_blocks * kKvBlockElems, 0.f);
        for (int m = 0; m < M; ++m) {
            AUR_REQUIRE(ctx[m] >= 1 && ctx[m] <= ctx_max && ctx[m] >= shared, "dbg_paged_attention: ctx[m] in [max(1, shared), ctx_max]");
            int* r = &rm[(size_t)m * kRowMetaStride];
            for (int b = 0; b < per_row; ++b) r[kRowMetaBt + b] = b < n_shared ? 1 + b : 1 + n_shared + m * per_row + b;
            r[0] = ctx[m] - 1;
            r[1] = m;
            r[kRowMetaWblk] = r[kRowMetaBt + (ctx[m] - 1) / kKvBlockTokens];
            for (int t = 0; t < ctx[m]; ++t) {
                const int src = t < shared ? 0 : m;   // the shared tokens are row 0's
                const int blk = r[kRowMetaBt + t / kKvBlockTokens];
                for (int kvi = 0; kvi < 2; ++kvi) {
                    const float* row = (kvi ? v : k) + ((size_t)src * ctx_max + t) * kHidden;
                    for (int h = 0; h < kHeads; ++h)
                        // ([block][K|V][head][token in block][64], gpt_kernels.hip: kv_offset)
                        std::memcpy(&pool[((((size_t)blk * 2 + kvi) * kHeads + h) * kKvBlockTokens + t % kKvBlockTokens) * kHeadDim],
                                    row + h * kHeadDim, kHeadDim * sizeof(float));
                }
            }
        }
        DevBuf dq, dpool, drm, dout;
        dq.ensure((size_t)M * kHidden * 4);
        drm.ensure(rm.size() * 4);
        dout.ensure((size_t)M * kHidden * 4);
        HIP_CHECK(hipMemcpy(dq.p, q, (size_t)M * kHidden * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(drm.p, rm.data(), rm.size() * 4, hipMemcpyHostToDevice));
        if (half) {
            std::vector<_Float16> ph(pool.size());
            for (size_t i = 0; i < pool.size(); ++i) ph[i] = (_Float16)pool[i];
            dpool.ensure(ph.size() * 2);
            HIP_CHECK(hipMemcpy(dpool.p, ph.data(), ph.size() * 2, hipMemcpyHostToDevice));
        } else {
            dpool.ensure(pool.size() * 4);
            HIP_CHECK(hipMemcpy(dpool.p, pool.data(), pool.size() * 4, hipMemcpyHostToDevice));
        }
        launch_paged_attention(dq.as<float>(), dpool.p, drm.as<int>(), kMaxBlocks, dout.as<float>(), M, st_, 0, half);
        HIP_CHECK(hipStreamSynchronize(st_));
        HIP_CHECK(hipMemcpy(out, dout.p, (size_t)M * kHidden * 4, hipMemcpyDeviceToHost));
    }
    void dbg_prompt_attention(const float* q, const float* k, const float* v, const int* row_seq, const int* row_pos, int M, int n_seq,
                              int ctx_max, int shared, bool half, float* out) {
        use();
        AUR_REQUIRE(M >= 1 && M <= 8192 && n_seq >= 1 && n_seq <= 64 && ctx_max >= 1 && ctx_max <= kMaxBlocks * kKvBlockTokens,
                    "dbg_prompt_attention: 1..8192 rows, 1..64 sequences, context 1..1056");
        AUR_REQUIRE(shared == 0 || shared == 16 || shared == 32, "dbg_prompt_attention: shared prefix of 0, 16 or 32 tokens");
        const int n_shared = shared / kKvBlockTokens, per_seq = (ctx_max + kKvBlockTokens - 1) / kKvBlockTokens;
        const long n_blocks = 1 + n_shared + (long)n_seq * per_seq;
        std::vector<int> bt((size_t)n_seq * kMaxBlocks, 0);
        std::vector<float> pool((size_t)n_blocks * kKvBlockElems, 0.f);
        for (int sq = 0; sq < n_seq; ++sq) {
            for (int b = 0; b < per_seq; ++b) bt[(size_t)sq * kMaxBlocks + b] = b < n_shared ? 1 + b : 1 + n_shared + sq * per_seq + b;
            for (int t = 0; t < ctx_max; ++t) {
                const int src = t < shared ? 0 : sq;
                const int blk = bt[(size_t)sq * kMaxBlocks + t / kKvBlockTokens];
                for (int kvi = 0; kvi < 2; ++kvi) {
                    const float* row = (kvi ? v : k) + ((size_t)src * ctx_max + t) * kHidden;
                    for (int h = 0; h < kHeads; ++h)
                        std::memcpy(&pool[((((size_t)blk * 2 + kvi) * kHeads + h) * kKvBlockTokens + t % kKvBlockTokens) * kHeadDim],
                                    row + h * kHeadDim, kHeadDim * sizeof(float));
                }
            }
        }
        std::vector<int> rs(row_seq, row_seq + M), rp(row_pos, row_pos + M);
        for (int m = 0; m < M; ++m)
            AUR_REQUIRE(rs[m] >= 0 && rs[m] < n_seq && rp[m] >= 0 && rp[m] < ctx_max, "dbg_prompt_attention: row_seq / row_pos out of range");
        RowWs w;
        w.st = st_;
        const int n_qblk = upload_qblocks(w, rs, rp);
        DevBuf dq, dpool, dbt, drs, drp, dout;
        dq.ensure((size_t)M * kHidden * 4);
        dout.ensure((size_t)M * kHidden * 4);
        dbt.ensure(bt.size() * 4);
        drs.ensure((size_t)M * 4);
        drp.ensure((size_t)M * 4);
        HIP_CHECK(hipMemcpy(dq.p, q, (size_t)M * kHidden * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(dbt.p, bt.data(), bt.size() * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(drs.p, rs.data(), (size_t)M * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(drp.p, rp.data(), (size_t)M * 4, hipMemcpyHostToDevice));
        if (half) {
            std::vector<_Float16> ph(pool.size());
            for (size_t i = 0; i < pool.size(); ++i) ph[i] = (_Float16)pool[i];
            dpool.ensure(ph.size() * 2);
            HIP_CHECK(hipMemcpy(dpool.p, ph.data(), ph.size() * 2, hipMemcpyHostToDevice));
        } else {
            dpool.ensure(pool.size() * 4);
            HIP_CHECK(hipMemcpy(dpool.p, pool.data(), pool.size() * 4, hipMemcpyHostToDevice));
        }
        launch_prompt_attention(dq.as<float>(), dpool.p, w.i_qblk.as<int2>(), n_qblk, drs.as<int>(), drp.as<int>(), dbt.as<int>(), kMaxBlocks,
                                dout.as<float>(), st_, half);
        HIP_CHECK(hipStreamSynchronize(st_));
        HIP_CHECK(hipMemcpy(out, dout.p, (size_t)M * kHidden * 4, hipMemcpyDeviceToHost));
    }
    void dbg_layernorm(const float* h, const float* gamma, const float* beta, float* out, int M) {
        use();
        DevBuf dh, dg, db, dout;
        dh.ensure((size_t)M * kHidden * 4);
        dg.ensure(kHidden * 4);
        db.ensure(kHidden * 4);
        dout.ensure((size_t)M * kHidden * 4);
        HIP_CHECK(hipMemcpy(dh.p, h, (size_t)M * kHidden * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(dg.p, gamma, kHidden * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(db.p, beta, kHidden * 4, hipMemcpyHostToDevice));
        launch_rows_ln(nullptr, 0, nullptr, dh.as<float>(), dg.as<float>(), db.as<float>(), dout.as<float>(), M, 1e-5f, st_);
        HIP_CHECK(hipStreamSynchronize(st_));
        HIP_CHECK(hipMemcpy(out, dout.p, (size_t)M * kHidden * 4, hipMemcpyDeviceToHost));
    }
    void dbg_conv1d(const float* x, const void* wp, const float* bias, const float* res, float* out,
                    const int* lens, int B, int Cin, int Mtot, int Cout, int L, int KS, int DIL, int padl,
                    float slope, int ups_s, int ups_p, bool f16) {
        use();
        const int Lout = L * std::max(1, ups_s);
        DevBuf dx, dw, db, dr, dout, dl;
        dx.ensure((size_t)B * Cin * L * 4);
        const size_t wbytes = (size_t)Mtot * Cin * KS * (f16 ? 2 : 4);
        dw.ensure(wbytes);
        dout.ensure((size_t)B * Cout * Lout * 4);
        dl.ensure((size_t)B * 4);
        HIP_CHECK(hipMemcpy(dx.p, x, (size_t)B * Cin * L * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(dw.p, wp, wbytes, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(dl.p, lens, (size_t)B * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemsetAsync(dout.p, 0, (size_t)B * Cout * Lout * 4, st_));
        if (bias) {
            db.ensure((size_t)Cout * 4);
            HIP_CHECK(hipMemcpy(db.p, bias, (size_t)Cout * 4, hipMemcpyHostToDevice));
        }
        if (res) {
            dr.ensure((size_t)B * Cout * Lout * 4);
            HIP_CHECK(hipMemcpy(dr.p, res, (size_t)B * Cout * Lout * 4, hipMemcpyHostToDevice));
        }
        ConvArgs a{};
        a.x = dx.as<float>();
        a.wp = dw.as<float>();
        a.wp16 = dw.p;
        a.bias = bias ? db.as<float>() : nullptr;
        a.res = res ? dr.as<float>() : nullptr;
        a.out = dout.as<float>();
        a.base_len = dl.as<int>();
        a.len_mul = 1;
        a.Cin = Cin; a.Mtot = Mtot; a.Cout = Cout;
        a.x_stride = L; a.o_stride = Lout;
        a.x_bstride = (long)Cin * L; a.o_bstride = (long)Cout * Lout;
        a.padl = padl; a.slope = slope; a.ups_s = ups_s; a.ups_p = ups_p;
        a.max_len = *std::max_element(lens, lens + B);
        a.B = B;
        if (f16)
            launch_conv1d_f16(a, KS, DIL, st_);
        else
            launch_conv1d(a, KS, DIL, st_);
        HIP_CHECK(hipStreamSynchronize(st_));
        HIP_CHECK(hipMemcpy(out, dout.p, (size_t)B * Cout * Lout * 4, hipMemcpyDeviceToHost));
    }
    void dbg_prefill(const int* text_ids, int n_text, uint64_t key, float rep_penalty, float* lnf_rows_out,
                     float* logits_out) {
        use();
        ensure_gpt();
        AUR_REQUIRE(idle(), "dbg_prefill needs an idle engine");
        aur_seq_desc d{};
        d.text_ids = text_ids; d.n_text = n_text; d.speaker_key = key; d.temperature = 0.f; d.top_p = 1.f; d.top_k = -1;
        d.repetition_penalty = rep_penalty; d.max_tokens = 1; d.seed = 0; d.ignore_stop = 1;
        const uint64_t id = submit(d);
        dbg_logits_.ensure((size_t)kMelVocab * 4);
        dbg_capture_ = true;
        share_prefix_now_ = false;
        step(nullptr, nullptr);
        share_prefix_now_ = true;
        dbg_capture_ = false;
        for (int guard = 0; guard < 1000 && seqs_.at(id)->state != SeqState::DONE; ++guard) step(nullptr, nullptr);
        Seq* s = seqs_.at(id).get();
        if (lnf_rows_out)
            HIP_CHECK(hipMemcpy(lnf_rows_out, dbg_lnf_.p, (size_t)s->n_prompt * kHidden * 4, hipMemcpyDeviceToHost));
        if (logits_out) HIP_CHECK(hipMemcpy(logits_out, dbg_logits_.p, (size_t)kMelVocab * 4, hipMemcpyDeviceToHost));
        aur_result r;
        while (poll(&r, 1) == 1) {
        }
        release(id);
    }
    void dbg_sample(const float* logits, int B, float temperature, float top_p, int top_k, float rep,
                    const uint8_t* seen, uint32_t seed, int step_idx, int* tokens_out) {
        use();
        AUR_REQUIRE(idle(), "dbg_sample needs an idle engine");
        AUR_REQUIRE(B >= 1 && B <= cfg_.max_seqs, "B <= max_seqs");
        std::vector<SlotInit> init(B);
        std::vector<int> slots(B);
        for (int b = 0; b < B; ++b) {
            init[b] = SlotInit{b, temperature, top_p, top_k, rep, 1 << 30, 1, seed + (unsigned)b, step_idx};
            slots[b] = b;
        }
        init_slots(init);
        if (seen) {
            for (int b = 0; b < B; ++b)
                HIP_CHECK(hipMemcpyAsync(seen_.as<unsigned char>() + (long)b * kSeenStride, seen + (long)b * kMelVocab,
                                         kMelVocab, hipMemcpyHostToDevice, st_));
        } else {
            HIP_CHECK(hipMemsetAsync(seen_.p, 0, (size_t)B * kSeenStride, st_));
        }
        DevBuf dl;
        dl.ensure((size_t)B * kMelVocab * 4);
        HIP_CHECK(hipMemcpyAsync(dl.p, logits, (size_t)B * kMelVocab * 4, hipMemcpyHostToDevice, st_));
        RowWs& w = ws_[0];
        w.i_sample_slot.ensure(B * 4);
        w.i_out_tok.ensure(B * 4);
        HIP_CHECK(hipMemcpyAsync(w.i_sample_slot.p, slots.data(), B * 4, hipMemcpyHostToDevice, st_));
        SamplerArgs a = sampler_args(w, dl.as<float>(), 1, B, kMelVocab, zero_bias_.as<float>(), nullptr);
        launch_sampler(a, st_);
        HIP_CHECK(hipStreamSynchronize(st_));
        HIP_CHECK(hipMemcpy(tokens_out, w.i_out_tok.p, B * 4, hipMemcpyDeviceToHost));
        for (int b = 0; b < B; ++b)
            if (tokens_out[b] >= 0) tokens_out[b] &= ~kTokFinishedBit;
    }

private:
    static constexpr int kMaxBlocks = 66;         // ceil(1047 / 16)
    static constexpr int kMaxLatRows = 608;
    static constexpr int kMelVocab = 1026;
    static constexpr int kHeadPad = 1088;         // mel_head columns padded to a multiple of 64
    static constexpr int kStartToken = 1024, kStopToken = 1025;
    static constexpr long kCondStride = 1024;

    bool idle() {
        std::lock_guard<std::mutex> lk(mu_);
        if (!waiting_.empty() || !voc_queue_.empty() || voc_active_ || infl_.on) return false;
        for (auto* s : slot_owner_)
            if (s) return false;
        return true;
    }

    static int frames_for(int n_lat) { return (int)std::floor((double)(4 * n_lat) * (24000.0 / 22050.0)); }

    // ------------------------------------------------------------------ GPT
    struct ConvEvent {
        hipEvent_t a, b;
        double flops, bytes;   // of ALL the launches between the two events
        bool used;
        int kind;   // decode profile events: 0..4 = GEMM kind (aur_stats.gemm_kind_*), 5 = paged attention
        int count;  // launches between the two events (decode profile batches; 1 for a vocoder conv)
    };
    struct LayerW {
        const float *ln1w, *ln1b, *wqkv, *bqkv, *wproj, *bproj, *ln2w, *ln2b, *wfc, *bfc, *wproj2, *bproj2;
        const void *sqkv, *sproj, *sfc, *sproj2;    // launch_pack_wsplit copies for the prefill-regime GEMM (the three bf16 planes of the split)
        const float *tqkv, *tproj, *tfc, *tproj2;   // pack_wt16 copies for the decode-regime GEMM (gemm_rows_kernel); tqkv / tfc
                                                    // have LayerNorm folded in (launch_fold_ln)
        const float *qkv_c1, *qkv_c2, *fc_c1, *fc_c2;   // ... with these epilogue vectors
    };
    // device-side repack of one [K][N] matrix into the decode GEMM's tile order (done once per load)
    const float* packed_copy(const float* Wm, int ldw, int K, int N) {
        packed_.emplace_back(new DevBuf());
        DevBuf& b = *packed_.back();
        b.ensure((size_t)K * N * sizeof(float));
        launch_pack_wt16(Wm, ldw, b.as<float>(), K, N, st_);
        return b.as<float>();
    }
    // pre-split copy of one [K][N] matrix for the prompt-row GEMM (only the split arithmetic uses it)
    const void* split_copy(const float* Wm, int K, int N) {
        if (gemm_prec_ != 1 || !gemm_presplit_) return nullptr;
        packed_.emplace_back(new DevBuf());
        DevBuf& b = *packed_.back();
        b.ensure((size_t)3 * K * N * 2);
        launch_pack_wsplit(Wm, N, b.p, K, N, st_);
        return b.p;
    }
    // LayerNorm-folded packed copy of a [K][N] matrix plus its two epilogue vectors (launch_fold_ln)
    const float* folded_copy(const float* Wm, const float* gamma, const float* beta, const float* bias, int K, int N,
                             const float** c1, const float** c2) {
        fold_scratch_.ensure((size_t)K * N * sizeof(float));
        packed_.emplace_back(new DevBuf());
        DevBuf& b = *packed_.back();
        b.ensure((size_t)K * N * sizeof(float));
        packed_.emplace_back(new DevBuf());
        DevBuf& v = *packed_.back();
        v.ensure((size_t)2 * N * sizeof(float));
        launch_fold_ln(Wm, N, gamma, beta, bias, fold_scratch_.as<float>(), b.as<float>(), v.as<float>(), v.as<float>() + N, K, N, st_);
        *c1 = v.as<float>();
        *c2 = v.as<float>() + N;
        return b.as<float>();
    }
    void ensure_gpt() {
        if (gpt_ready_) return;
        layers_.clear();
        const int H = kHidden;
        for (int i = 0; i < cfg_.n_layer; ++i) {
            const std::string p = "gpt.h." + std::to_string(i) + ".";
            LayerW l;
            l.ln1w = W(p + "ln_1.w", H); l.ln1b = W(p + "ln_1.b", H);
            l.wqkv = W(p + "attn.c_attn.w", (int64_t)H * 3 * H); l.bqkv = W(p + "attn.c_attn.b", 3 * H);
            l.wproj = W(p + "attn.c_proj.w", (int64_t)H * H); l.bproj = W(p + "attn.c_proj.b", H);
            l.ln2w = W(p + "ln_2.w", H); l.ln2b = W(p + "ln_2.b", H);
            l.wfc = W(p + "mlp.c_fc.w", (int64_t)H * 4 * H); l.bfc = W(p + "mlp.c_fc.b", 4 * H);
            l.wproj2 = W(p + "mlp.c_proj.w", (int64_t)4 * H * H); l.bproj2 = W(p + "mlp.c_proj.b", H);
            l.tqkv = l.tproj = l.tfc = l.tproj2 = nullptr;
            l.sqkv = l.sproj = l.sfc = l.sproj2 = nullptr;
            l.qkv_c1 = l.qkv_c2 = l.fc_c1 = l.fc_c2 = nullptr;
            layers_.push_back(l);
        }
        packed_.clear();
        {
            for (auto& l : layers_) {
                l.tqkv = folded_copy(l.wqkv, l.ln1w, l.ln1b, l.bqkv, H, 3 * H, &l.qkv_c1, &l.qkv_c2);
                l.tproj = packed_copy(l.wproj, H, H, H);
                l.tfc = folded_copy(l.wfc, l.ln2w, l.ln2b, l.bfc, H, 4 * H, &l.fc_c1, &l.fc_c2);
                l.tproj2 = packed_copy(l.wproj2, H, 4 * H, H);
                l.sqkv = split_copy(l.wqkv, H, 3 * H);
                l.sproj = split_copy(l.wproj, H, H);
                l.sfc = split_copy(l.wfc, H, 4 * H);
                l.sproj2 = split_copy(l.wproj2, 4 * H, H);
            }
        }
        wte_ = W("gpt.wte", (int64_t)kMelVocab * H);
        wpe_ = W("gpt.wpe", (int64_t)kMaxLatRows * H);
        lnfw_ = W("gpt.ln_f.w", H); lnfb_ = W("gpt.ln_f.b", H);
        fnw_ = W("final_norm.w", H); fnb_ = W("final_norm.b", H);
        headT_ = W("mel_head.wT", (int64_t)H * kHeadPad);
        headb_ = W("mel_head.b", kHeadPad);
        thead_ = packed_copy(headT_, kHeadPad, H, kHeadPad);
        HIP_CHECK(hipStreamSynchronize(st_));
        fold_scratch_.release();
        text_emb_ = W("text_emb");
        text_pos_ = W("text_pos");
        text_vocab_ = (int)(w_.at("text_emb").numel / H);
        text_positions_ = (int)(w_.at("text_pos").numel / H);
        ensure_voc();
        gpt_ready_ = true;
    }
    void ensure_rows(RowWs& w, int M) {
        if (M <= w.rows_cap) return;
        last_active_.clear();   // buffers move: the row indices have to be uploaded again
        const int cap = (std::max(M, 64) + 63) / 64 * 64;   // whole 64-row groups: the decode chain keeps its rows packed (pk_off)
        w.ybuf.ensure((size_t)cap * kHidden * 4);
        w.stats.ensure((size_t)cap * 64 * sizeof(float2));
        w.row_meta.ensure((size_t)cap * kRowMetaStride * sizeof(int));
        // embed_decode writes entries [0, kRowMetaBt + max_blocks) of the live rows only; the look-ahead entries behind them and the rows
        // >= M of the last tile are READ (their values discarded by selects): defined zeros, not whatever the allocation held
        HIP_CHECK(hipMemsetAsync(w.row_meta.p, 0, (size_t)cap * kRowMetaStride * sizeof(int), w.st));
        w.h.ensure((size_t)cap * kHidden * 4);
        w.xn.ensure((size_t)cap * kHidden * 4);
        w.qbuf.ensure((size_t)cap * kHidden * 4);
        w.att.ensure((size_t)cap * kHidden * 4);
        w.act.ensure((size_t)cap * 4 * kHidden * 4);
        w.P.ensure((size_t)cap * 4096 * 4);   // one GEMM output slab (prefill)
        w.i_row_slot.ensure((size_t)cap * 4);
        w.i_row_pos.ensure((size_t)cap * 4);
        w.i_desc.ensure((size_t)cap * sizeof(int4));
        w.rows_cap = cap;
    }
    // Profile mode, decode chain.  A HIP-event pair around ONE 5-12 us launch adds 2-5 us to what it measures (the pair serialises
    // the launch processing), so the per-kernel numbers of the bench line come from REPLAY BATCHES instead: after every
    // profile_every_-th decode step the step's own launches are issued once more, kind by kind -- the QKV GEMM of every layer back
    // to back between one event pair, then every layer's attention, proj, FC, proj2, then the head -- on the step's real operands
    // (each layer's weights, the live rows, the real contexts) with the OUTPUTS redirected to scratch buffers (the residual
    // stream, K/V pages and latents of the sequences are not touched).  One pair per n_layer launches: the interval is the
    // launch-to-launch period a kernel-trace reports per kernel.  Algorithmic bytes of a GEMM launch: weights once + the activation
    // rows once + the output tile once (the residual epilogue reads and writes it).
    ConvEvent& prof_event(int kind, double flops, double bytes, int count) {
        if (n_gemm_events_ == gemm_events_.size()) {
            ConvEvent e{};
            HIP_CHECK(hipEventCreate(&e.a));
            HIP_CHECK(hipEventCreate(&e.b));
            gemm_events_.push_back(e);
        }
        ConvEvent& ev = gemm_events_[n_gemm_events_++];
        ev.kind = kind;
        ev.flops = flops;
        ev.bytes = bytes;
        ev.count = count;
        return ev;
    }
    static double gemm_alg_bytes(const GemmRowsArgs& a, GemmRowsEpi epi) {
        return 4.0 * ((double)a.K * a.N + (double)a.M * a.K + (double)a.M * a.N * (epi == kEpiResidual ? 2.0 : 1.0));
    }
    // the four GEMM launches of layer l exactly as forward_decode issues them; `redirect` sends every output to the profile scratch
    GemmRowsArgs decode_gemm_args(RowWs& w, int l, int kind, int M, const int* d_row_slot, bool redirect) {
        const LayerW& L = layers_[l];
        const int mtt = w.rows_cap / 16;   // h, att, act are packed rows (pk_off) with this many 16-row tiles
        float* h = w.h.as<float>();
        GemmRowsArgs a{};
        a.M = M; a.prec = gemm_prec_; a.xmt = mtt;
        a.ksp_buf = ksp_buf_.as<float>(); a.ksp_cnt = ksp_cnt_.as<unsigned>();
        if (kind == 0) {
            a.eps = 1e-5f; a.X = h; a.Wt = L.tqkv; a.N = 3 * kHidden; a.K = kHidden; a.bias = L.qkv_c2;
            a.ln_c1 = L.qkv_c1; a.stats_in = w.stats.as<float2>(); a.out = redirect ? prof_q_.as<float>() : w.qbuf.as<float>(); a.ldo = kHidden;
            a.kv_layer = redirect ? prof_kv_.p : kv_layer(l); a.kv_half = kv_half_ ? 1 : 0; a.row_meta = w.row_meta.as<int>();
        } else if (kind == 1) {
            a.X = w.att.as<float>(); a.Wt = L.tproj; a.N = kHidden; a.K = kHidden; a.bias = L.bproj;
            a.out = redirect ? prof_h_.as<float>() : h; a.omt = mtt; a.stats_out = redirect ? prof_stats_.as<float2>() : w.stats.as<float2>();
        } else if (kind == 2) {
            a.eps = 1e-5f; a.X = h; a.Wt = L.tfc; a.N = 4 * kHidden; a.K = kHidden; a.bias = L.fc_c2;
            a.ln_c1 = L.fc_c1; a.stats_in = w.stats.as<float2>(); a.out = redirect ? prof_act_.as<float>() : w.act.as<float>(); a.omt = mtt;
            a.gelu_erf = cfg_.gelu_erf ? 1 : 0;
        } else {
            a.X = w.act.as<float>(); a.Wt = L.tproj2; a.N = kHidden; a.K = 4 * kHidden; a.bias = L.bproj2;
            a.out = redirect ? prof_h_.as<float>() : h; a.omt = mtt;
            // (ln_f computes its own statistics in final_rows_kernel)
            if (l + 1 < cfg_.n_layer) a.stats_out = redirect ? prof_stats_.as<float2>() : w.stats.as<float2>();
        }
        return a;
    }
    static void decode_gemm_kind(int kind, bool* ln, GemmRowsEpi* epi) {
        *ln = (kind == 0 || kind == 2);
        *epi = kind == 0 ? kEpiQkv : kind == 2 ? kEpiBiasGelu : kEpiResidual;
    }
    // One decode step through the blocks: 5 launches per layer (QKV GEMM with LN1 folded and KV page write, attention,
    // proj GEMM + residual, FC GEMM with LN2 folded + gelu, proj2 GEMM + residual), no split-K slabs in HBM.  Leaves the
    // residual stream in w.h (ln_f is applied by final_rows_kernel).
    void forward_decode(RowWs& w, int M, const int* d_row_slot) {
        const int mtt = w.rows_cap / 16;
        for (int l = 0; l < cfg_.n_layer; ++l) {
            for (int kind = 0; kind < 4; ++kind) {
                bool ln;
                GemmRowsEpi epi;
                decode_gemm_kind(kind, &ln, &epi);
                launch_gemm_rows(decode_gemm_args(w, l, kind, M, d_row_slot, false), ln, epi, w.st);
                if (kind == 0)
                    launch_paged_attention(w.qbuf.as<float>(), kv_layer(l), w.row_meta.as<int>(), kMaxBlocks, w.att.as<float>(), M, w.st, mtt, kv_half_);
            }
        }
    }
    GemmRowsArgs head_gemm_args(RowWs& w, int Ms, bool redirect) {
        GemmRowsArgs a{};
        a.M = Ms; a.prec = gemm_prec_; a.X = w.ybuf.as<float>(); a.xmt = w.rows_cap / 16; a.Wt = thead_; a.N = kHeadPad; a.K = kHidden; a.bias = headb_;
        a.out = redirect ? prof_act_.as<float>() : w.P2.as<float>(); a.ldo = kHeadPad;
        return a;
    }
    void profile_replay(RowWs& w, int M, const int* d_row_slot) {
        const int mtt = w.rows_cap / 16, nl = cfg_.n_layer;
        // (the residual replays READ their output scratch -- out += ... -- so a freshly allocated buffer is defined once)
        auto scratch = [&](DevBuf& buf, size_t n) {
            const void* before = buf.p;
            buf.ensure(n);
            if (buf.p != before) HIP_CHECK(hipMemsetAsync(buf.p, 0, n, w.st));
        };
        scratch(prof_q_, (size_t)w.rows_cap * kHidden * 4);
        scratch(prof_h_, (size_t)w.rows_cap * kHidden * 4);
        scratch(prof_stats_, (size_t)w.rows_cap * 64 * sizeof(float2));
        scratch(prof_act_, (size_t)w.rows_cap * 4 * kHidden * 4);
        prof_kv_.ensure((size_t)kv_layer_stride_ * (kv_half_ ? 2 : 4));
        for (int kind = 0; kind < 4; ++kind) {
            bool ln;
            GemmRowsEpi epi;
            decode_gemm_kind(kind, &ln, &epi);
            const GemmRowsArgs a0 = decode_gemm_args(w, 0, kind, M, d_row_slot, true);
            ConvEvent& ev = prof_event(kind, nl * 2.0 * a0.M * a0.N * a0.K, nl * gemm_alg_bytes(a0, epi), nl);
            HIP_CHECK(hipEventRecord(ev.a, w.st));
            for (int l = 0; l < nl; ++l) launch_gemm_rows(decode_gemm_args(w, l, kind, M, d_row_slot, true), ln, epi, w.st);
            HIP_CHECK(hipEventRecord(ev.b, w.st));
            if (kind == 0) {   // attention of every layer: the step's q rows against each layer's real K/V pages
                ConvEvent& ea = prof_event(5, nl * 4.0 * kHidden * step_kv_tokens_, nl * ((kv_half_ ? 4.0 : 8.0) * kHidden * step_kv_tokens_ + 8.0 * kHidden * M), nl);
                HIP_CHECK(hipEventRecord(ea.a, w.st));
                for (int l = 0; l < nl; ++l)
                    launch_paged_attention(w.qbuf.as<float>(), kv_layer(l), w.row_meta.as<int>(), kMaxBlocks, prof_h_.as<float>(), M, w.st, mtt, kv_half_);
                HIP_CHECK(hipEventRecord(ea.b, w.st));
            }
        }
        {   // the mel head has one weight matrix: its replays are L2 / Infinity-Cache warm after the first (1 launch in 152)
            const GemmRowsArgs a0 = head_gemm_args(w, M, true);
            ConvEvent& ev = prof_event(4, nl * 2.0 * a0.M * a0.N * a0.K, nl * gemm_alg_bytes(a0, kEpiBias), nl);
            HIP_CHECK(hipEventRecord(ev.a, w.st));
            for (int l = 0; l < nl; ++l) launch_gemm_rows(a0, false, kEpiBias, w.st);
            HIP_CHECK(hipEventRecord(ev.b, w.st));
        }
    }
    // Fixed cost of one HIP-event pair on an otherwise busy stream: recording events between back-to-back short
    // kernels serialises their launch processing, which inflates a ~12 us kernel by ~3 us.  Measured once as the
    // median elapsed time of 64 empty pairs and subtracted from every sampled GEMM interval (reported in aur_stats).
    float event_pair_overhead_ms() {
        if (event_overhead_ms_ >= 0.f) return event_overhead_ms_;
        std::vector<hipEvent_t> ev(128);
        for (auto& e : ev) HIP_CHECK(hipEventCreate(&e));
        for (int i = 0; i < 64; ++i) {
            HIP_CHECK(hipEventRecord(ev[2 * i], st_));
            HIP_CHECK(hipEventRecord(ev[2 * i + 1], st_));
        }
        HIP_CHECK(hipStreamSynchronize(st_));
        std::vector<float> d(64);
        for (int i = 0; i < 64; ++i) HIP_CHECK(hipEventElapsedTime(&d[i], ev[2 * i], ev[2 * i + 1]));
        for (auto& e : ev) (void)hipEventDestroy(e);
        std::sort(d.begin(), d.end());
        event_overhead_ms_ = d[32];
        return event_overhead_ms_;
    }
    void collect_gemm_events() {
        // the replay batches sit behind the step's token read-back on the stream (and, with the pipelined decode, the next step's
        // batches may already be queued behind them): wait for the last one recorded
        if (n_gemm_events_) HIP_CHECK(hipEventSynchronize(gemm_events_[n_gemm_events_ - 1].b));
        const float ovh = n_gemm_events_ ? event_pair_overhead_ms() : 0.f;
        if (n_gemm_events_) stats_.event_pair_overhead_ms = ovh;
        for (size_t i = 0; i < n_gemm_events_; ++i) {
            float ms = 0.f;
            HIP_CHECK(hipEventElapsedTime(&ms, gemm_events_[i].a, gemm_events_[i].b));
            const ConvEvent& ev = gemm_events_[i];
            if (ev.kind == 5) {
                stats_.attn_ms += ms;
                stats_.attn_bytes += ev.bytes;
                stats_.attn_launches += ev.count;
                continue;
            }
            if (ev.kind >= 0 && ev.kind < 5) {
                stats_.gemm_kind_ms[ev.kind] += ms;
                stats_.gemm_kind_bytes[ev.kind] += ev.bytes;
                stats_.gemm_kind_flops[ev.kind] += ev.flops;
                stats_.gemm_kind_launches[ev.kind] += ev.count;
            }
            stats_.gemm_ms_raw += ms;
            stats_.gemm_ms += std::max(0.f, ms - ovh);   // one pair per batch of ev.count launches
            stats_.gemm_flops += ev.flops;
            stats_.gemm_bytes += ev.bytes;
            stats_.gemm_launches += ev.count;
        }
        n_gemm_events_ = 0;
    }
    // Query blocks of a batch of prompt rows for launch_prompt_attention: runs of consecutive rows of one slot with positions
    // ascending by one, cut every 32 rows (from the run's first row: a sequence's blocks do not depend on what else is in the batch).
    int upload_qblocks(RowWs& w, const std::vector<int>& row_slot, const std::vector<int>& row_pos) {
        w.h_qblk.clear();
        const int M = (int)row_slot.size();
        for (int m = 0; m < M;) {
            int n = 1;
            while (m + n < M && n < 32 && row_slot[m + n] == row_slot[m] && row_pos[m + n] == row_pos[m] + n) ++n;
            w.h_qblk.push_back(make_int2(m, n));
            m += n;
        }
        w.i_qblk.ensure(w.h_qblk.size() * sizeof(int2));
        HIP_CHECK(hipMemcpyAsync(w.i_qblk.p, w.h_qblk.data(), w.h_qblk.size() * sizeof(int2), hipMemcpyHostToDevice, w.st));
        return (int)w.h_qblk.size();
    }
    // Prompt rows (prefill, speaker prefix, literal second pass): explicit row positions, LDS-tiled GEMMs in the configured arithmetic
    // (gemm_prec_: bf16 x 3 split by default, exact-f32 MFMA under aur_config.gemm_f32_exact; k ascending for every M => a prompt's rows do not depend on what else was admitted), separate LayerNorm launches.
    void forward_rows(RowWs& w, int M, const int* d_row_slot, const int* d_row_pos, int n_qblk) {
        float* h = w.h.as<float>();
        float* xn = w.xn.as<float>();
        float* P = w.P.as<float>();
        const int* bt = block_tables_.as<int>();
        const int* kvpos = slot_kvpos_.as<int>();
        AUR_REQUIRE(d_row_pos != nullptr, "forward_rows: prompt rows carry explicit positions");
        launch_rows_ln(nullptr, 0, nullptr, h, layers_[0].ln1w, layers_[0].ln1b, xn, M, 1e-5f, w.st);
        for (int l = 0; l < cfg_.n_layer; ++l) {
            const LayerW& L = layers_[l];
            void* kvl = kv_layer(l);
            if (gemm_prec_ == 1) {   // bias, q rows and the K / V page writes in the GEMM's epilogue (27.50 vs 27.62 ms per prefill)
                GemmGelu qe{L.bqkv, nullptr, 0};
                qe.qbuf = w.qbuf.as<float>(); qe.kv_layer = kvl; qe.row_slot = d_row_slot; qe.row_pos = d_row_pos; qe.block_tables = bt;
                qe.max_blocks = kMaxBlocks; qe.kv_half = kv_half_ ? 1 : 0;
                launch_gemm_tile(xn, kHidden, L.wqkv, P, M, 3 * kHidden, kHidden, w.st, &qe, gemm_prec_, 1, L.sqkv);
            } else {
                launch_gemm_tile(xn, kHidden, L.wqkv, P, M, 3 * kHidden, kHidden, w.st, nullptr, gemm_prec_, 1, L.sqkv);
                launch_qkv_epilogue(P, 1, L.bqkv, w.qbuf.as<float>(), kvl, d_row_slot, d_row_pos, kvpos, bt, kMaxBlocks, M, w.st, kv_half_);
            }
            launch_prompt_attention(w.qbuf.as<float>(), kvl, w.i_qblk.as<int2>(), n_qblk, d_row_slot, d_row_pos, bt, kMaxBlocks, w.att.as<float>(), w.st, kv_half_);
            // (two K-slabs for this GEMM -- 576 workgroups instead of 288 on 256 CUs: 27.55-27.68 vs 27.82 ms per prefill, four slabs
            // 27.83-27.93: not worth another slab sum)
            launch_gemm_tile(w.att.as<float>(), kHidden, L.wproj, P, M, kHidden, kHidden, w.st, nullptr, gemm_prec_, 1, L.sproj);
            launch_rows_ln(P, 1, L.bproj, h, L.ln2w, L.ln2b, xn, M, 1e-5f, w.st);
            const GemmGelu ge{L.bfc, w.act.as<float>(), cfg_.gelu_erf ? 1 : 0};
            launch_gemm_tile(xn, kHidden, L.wfc, P, M, 4 * kHidden, kHidden, w.st, &ge, gemm_prec_, 1, L.sfc);
            // K = 4096 against N = 1024: four K-slabs (4x the workgroups of a GEMM that otherwise fills one round of CUs with
            // 256 dependent k-steps); rows_ln sums them in a fixed order
            launch_gemm_tile(w.act.as<float>(), 4 * kHidden, L.wproj2, P, M, kHidden, 4 * kHidden, w.st, nullptr, gemm_prec_, kProj2Slabs, L.sproj2);
            const bool last = (l + 1 == cfg_.n_layer);
            launch_rows_ln(P, kProj2Slabs, L.bproj2, h, last ? lnfw_ : layers_[l + 1].ln1w, last ? lnfb_ : layers_[l + 1].ln1b,
                           xn, M, 1e-5f, w.st);
        }
    }
    SamplerArgs sampler_args(RowWs& w, const float* P, int S, int Ms, int Npad, const float* bias, const int* next_kvpos) {
        SamplerArgs a{};
        a.P = P; a.S = S; a.Ms = Ms; a.Npad = Npad; a.V = kMelVocab; a.bias = bias;
        a.sample_slot = w.i_sample_slot.as<int>();
        a.next_kvpos = next_kvpos;
        a.seen = seen_.as<unsigned char>();
        a.slot_tok = slot_tok_.as<int>(); a.slot_pos = slot_pos_.as<int>(); a.slot_kvpos = slot_kvpos_.as<int>();
        a.slot_ngen = slot_ngen_.as<int>(); a.slot_finished = slot_finished_.as<int>();
        a.temperature = temperature_.as<float>(); a.top_p = top_p_.as<float>(); a.top_k = top_k_.as<int>();
        a.rep_penalty = rep_.as<float>(); a.max_tokens = max_tokens_.as<int>(); a.ignore_stop = ignore_stop_.as<int>();
        a.seed = seed_.as<unsigned>();
        a.out_tok = w.i_out_tok.as<int>();
        a.dbg_logits = dbg_capture_ ? dbg_logits_.as<float>() : nullptr;
        a.stop_token = kStopToken;
        a.force_full_sort = sampler_full_sort_ ? 1 : 0;
        return a;
    }
    // final_norm -> latent stash -> mel_head GEMM -> fused sampler -> read back tokens/finished flags
    // final_norm -> latent stash -> mel_head GEMM -> fused sampler, in three parts so that the kernel part can be
    // captured in a hipGraph: (1) host->device index uploads, (2) kernels only, (3) token / finished-flag read-back.
    void sample_upload(RowWs& w, const std::vector<int>& sample_row, const std::vector<int>& sample_slot,
                       const std::vector<int>* next_kvpos) {
        const int Ms = (int)sample_slot.size();
        w.pin.ensure(((size_t)cfg_.max_seqs * 2 + 16) * sizeof(int));
        w.i_sample_row.ensure((size_t)std::max(Ms, 64) * 4);
        w.i_sample_slot.ensure((size_t)std::max(Ms, 64) * 4);
        w.i_next_kvpos.ensure((size_t)std::max(Ms, 64) * 4);
        w.i_out_tok.ensure((size_t)std::max(Ms, 64) * 4);
        w.ybuf.ensure((size_t)std::max(Ms, 64) * kHidden * 4);
        w.P2.ensure((size_t)std::max(Ms, 64) * kHeadPad * 4);
        HIP_CHECK(hipMemcpyAsync(w.i_sample_row.p, sample_row.data(), (size_t)Ms * 4, hipMemcpyHostToDevice, w.st));
        HIP_CHECK(hipMemcpyAsync(w.i_sample_slot.p, sample_slot.data(), (size_t)Ms * 4, hipMemcpyHostToDevice, w.st));
        if (next_kvpos)
            HIP_CHECK(hipMemcpyAsync(w.i_next_kvpos.p, next_kvpos->data(), (size_t)Ms * 4, hipMemcpyHostToDevice, w.st));
    }
    void sample_kernels(RowWs& w, int Ms, bool has_next_kvpos) {
        launch_final_norm(w.xn.as<float>(), w.i_sample_row.as<int>(), w.i_sample_slot.as<int>(), fnw_, fnb_, w.ybuf.as<float>(),
                          latents_.as<float>(), (long)kMaxLatRows * kHidden, slot_ngen_.as<int>(), kMaxLatRows, Ms, 1e-5f, w.st);
        launch_gemm_tile(w.ybuf.as<float>(), kHidden, headT_, w.P2.as<float>(), Ms, kHeadPad, kHidden, w.st, nullptr, gemm_prec_);
        SamplerArgs a = sampler_args(w, w.P2.as<float>(), 1, Ms, kHeadPad, headb_, has_next_kvpos ? w.i_next_kvpos.as<int>() : nullptr);
        launch_sampler(a, w.st);
    }
    // decode tail of the gemm_rows chain: rows are the live sequences in order (sample_row = identity), w.h holds the
    // residual stream: ln_f + final_norm (+ second final_norm into the latent stash) -> mel_head GEMM (+ bias) -> sampler
    void sample_kernels_decode(RowWs& w, int Ms) {
        const int mtt = w.rows_cap / 16;
        launch_final_rows(w.h.as<float>(), mtt, w.i_sample_slot.as<int>(), lnfw_, lnfb_, fnw_, fnb_, w.ybuf.as<float>(),
                          latents_.as<float>(), (long)kMaxLatRows * kHidden, slot_ngen_.as<int>(), kMaxLatRows, Ms, 1e-5f, w.st);
        launch_gemm_rows(head_gemm_args(w, Ms, false), false, kEpiBias, w.st);
        SamplerArgs sa = sampler_args(w, w.P2.as<float>(), 1, Ms, kHeadPad, zero_bias_.as<float>(), nullptr);
        launch_sampler(sa, w.st);
    }
    void sample_readback(RowWs& w, int Ms, hipStream_t st, int* pin = nullptr) {
        if (!pin) pin = w.pin.as<int>();
        HIP_CHECK(hipMemcpyAsync(pin, w.i_out_tok.p, (size_t)Ms * 4, hipMemcpyDeviceToHost, st));   // token | finished bit
    }
    void sample_launch(RowWs& w, const std::vector<int>& sample_row, const std::vector<int>& sample_slot,
                       const std::vector<int>* next_kvpos) {
        sample_upload(w, sample_row, sample_slot, next_kvpos);
        sample_kernels(w, (int)sample_slot.size(), next_kvpos != nullptr);
        sample_readback(w, (int)sample_slot.size(), w.st);
    }
    // after the stream has been synchronised: append tokens, retire finished sequences
    void sample_collect(RowWs& w, const std::vector<int>& sample_slot) {
        const int Ms = (int)sample_slot.size();
        const int* pin = w.pin.as<int>();
        for (int j = 0; j < Ms; ++j) {
            Seq* s = slot_owner_[sample_slot[j]];
            s->tokens.push_back(pin[j] & ~kTokFinishedBit);
            stats_.tokens_generated++;
            if (pin[j] & kTokFinishedBit) just_finished_.push_back(s);
        }
    }
    // sequences whose tokens completed in this step: (optional literal second pass) -> latent pool -> vocoder queue
    void retire_finished() {
        if (just_finished_.empty()) return;
        if (cfg_.second_pass) {
            std::vector<Seq*> wanted;
            for (Seq* s : just_finished_)
                if (!s->cancel) wanted.push_back(s);
            if (!wanted.empty()) second_pass(wanted);
        }
        for (Seq* s : just_finished_) finish_tokens(s);
        just_finished_.clear();
    }
    // A/B mode (aur_config.second_pass): the reference's get_model_logits (XTTSv2.py:617-687) restated on the GPU: one
    // prefill over [cond ; 1024 ; tokens ; 1025 x 4] with mel positions 0..N+4, collect ln_f rows of positions
    // n_cond .. n_cond+N-1 (i.e. drop the last 5), apply final_norm twice.  The sequence still owns its slot and KV
    // blocks, which are simply overwritten.  The default path (latent stash) never runs this.
    void second_pass(const std::vector<Seq*>& seqs) {
        RowWs& w = ws_[0];
        last_active_.clear();
        std::vector<int4> desc;
        std::vector<int> row_slot, row_pos, lat_row0;
        for (Seq* s : seqs) {
            const int first = s->shared_prefix ? 32 : 0;
            const int n_cond = 32 + (int)s->text_ids.size();
            for (int i = first; i < 32; ++i) desc.push_back(make_int4(0, i, s->spk_row, 0));
            for (int i = 0; i < (int)s->text_ids.size(); ++i) desc.push_back(make_int4(1, s->text_ids[i], i, 0));
            lat_row0.push_back((int)desc.size());
            const int N = (int)s->tokens.size();
            for (int j = 0; j < N + 5; ++j) {
                const int id = (j == 0) ? kStartToken : (j <= N ? s->tokens[j - 1] : kStopToken);
                desc.push_back(make_int4(2, id, std::min(j, kMaxLatRows - 1), 0));   // trailing rows are dropped; keep wpe in range
            }
            for (int i = first; i < n_cond + N + 5; ++i) {
                row_slot.push_back(s->slot);
                row_pos.push_back(i);
            }
        }
        const int M = (int)row_slot.size();
        ensure_rows(w, M);
        HIP_CHECK(hipMemcpyAsync(w.i_desc.p, desc.data(), (size_t)M * sizeof(int4), hipMemcpyHostToDevice, w.st));
        HIP_CHECK(hipMemcpyAsync(w.i_row_slot.p, row_slot.data(), (size_t)M * 4, hipMemcpyHostToDevice, w.st));
        HIP_CHECK(hipMemcpyAsync(w.i_row_pos.p, row_pos.data(), (size_t)M * 4, hipMemcpyHostToDevice, w.st));
        launch_embed_prompt(w.i_desc.as<int4>(), spk_table_.as<float>(), text_emb_, text_pos_, wte_, wpe_, w.h.as<float>(), M, w.st);
        forward_rows(w, M, w.i_row_slot.as<int>(), w.i_row_pos.as<int>(), upload_qblocks(w, row_slot, row_pos));
        for (size_t k = 0; k < seqs.size(); ++k) {
            Seq* s = seqs[k];
            if (latpool_free_.empty()) throw StateError("latent pool exhausted");
            s->pool_idx = latpool_free_.back();
            latpool_free_.pop_back();
            launch_double_norm_rows(w.xn.as<float>() + (long)lat_row0[k] * kHidden,
                                    latpool_.as<float>() + (long)s->pool_idx * kMaxLatRows * kHidden, (int)s->tokens.size(),
                                    fnw_, fnb_, 1e-5f, w.st);
        }
        HIP_CHECK(hipStreamSynchronize(w.st));
        stats_.prefill_rows += M;
    }
    void finish_tokens(Seq* s) {
        if (s->cancel) {   // aur_cancel: nobody wants the audio -- free slot and blocks, report it, no vocoder
            std::lock_guard<std::mutex> lk(mu_);
            for (int b : s->blocks) free_blocks_.push_back(b);
            s->blocks.clear();
            slot_owner_[s->slot] = nullptr;
            s->slot = -1;
            finish_cancelled(s);
            return;
        }
        // park the stashed latents in the pool (D2D on the main stream, ordered before any later prefill that reuses
        // the slot) and release slot + KV blocks at once: the vocoder stage no longer occupies a batcher slot
        if (s->pool_idx < 0) {   // (the second-pass mode has already written the pool entry)
            if (latpool_free_.empty()) throw StateError("latent pool exhausted");
            s->pool_idx = latpool_free_.back();
            latpool_free_.pop_back();
            const size_t n = s->tokens.size() * (size_t)kHidden * sizeof(float);
            HIP_CHECK(hipMemcpyAsync(latpool_.as<float>() + (long)s->pool_idx * kMaxLatRows * kHidden,
                                     latents_.as<float>() + (long)s->slot * kMaxLatRows * kHidden, n, hipMemcpyDeviceToDevice, st_));
        }
        HIP_CHECK(hipEventRecord(ev_lat_, st_));
        std::lock_guard<std::mutex> lk(mu_);
        s->state = SeqState::TOKENS_DONE;   // (under the lock: aur_cancel reads the state from another thread)
        for (int b : s->blocks) free_blocks_.push_back(b);
        s->blocks.clear();
        slot_owner_[s->slot] = nullptr;
        s->slot = -1;
        voc_queue_.push_back(s);
    }
    void init_slots(const std::vector<SlotInit>& init) {
        i_init_.ensure(init.size() * sizeof(SlotInit));
        HIP_CHECK(hipMemcpyAsync(i_init_.p, init.data(), init.size() * sizeof(SlotInit), hipMemcpyHostToDevice, st_));
        trace_launch("init_slots_kernel");
        hipLaunchKernelGGL(init_slots_kernel, dim3((unsigned)init.size()), dim3(256), 0, st_, i_init_.as<SlotInit>(),
                           slot_tok_.as<int>(), slot_pos_.as<int>(), slot_kvpos_.as<int>(), slot_ngen_.as<int>(),
                           slot_finished_.as<int>(), temperature_.as<float>(), top_p_.as<float>(), top_k_.as<int>(),
                           rep_.as<float>(), max_tokens_.as<int>(), ignore_stop_.as<int>(), seed_.as<unsigned>(),
                           seen_.as<unsigned char>(), kStartToken);
        HIP_CHECK(hipGetLastError());
    }
    void prefill(const std::vector<Seq*>& seqs) {
        RowWs& w = ws_[0];
        last_active_.clear();   // prefill reuses the index buffers of the decode chain
        std::vector<int4> desc;
        std::vector<int> row_slot, row_pos, sample_row, sample_slot, next_kvpos;
        std::vector<SlotInit> init;
        for (Seq* s : seqs) {
            const aur_seq_desc& p = s->params;
            init.push_back(SlotInit{s->slot, p.temperature, p.top_p, p.top_k, p.repetition_penalty, p.max_tokens,
                                    p.ignore_stop, p.seed, 0});
            const int first = s->shared_prefix ? 32 : 0;     // rows 0..31 live in the speaker's shared KV blocks
            for (int i = first; i < 32; ++i) desc.push_back(make_int4(0, i, s->spk_row, 0));
            for (int i = 0; i < (int)s->text_ids.size(); ++i) {
                const int id = s->text_ids[i];
                AUR_REQUIRE(id >= 0 && id < text_vocab_ && i < text_positions_, "text id / position out of range");
                desc.push_back(make_int4(1, id, i, 0));
            }
            desc.push_back(make_int4(2, kStartToken, 0, 0));
            for (int i = first; i < s->n_prompt; ++i) {
                row_slot.push_back(s->slot);
                row_pos.push_back(i);
            }
            sample_row.push_back((int)row_slot.size() - 1);
            sample_slot.push_back(s->slot);
            next_kvpos.push_back(s->n_prompt);
        }
        const int M = (int)row_slot.size();
        ensure_rows(w, M);
        init_slots(init);
        HIP_CHECK(hipMemcpyAsync(block_tables_.p, h_block_tables_.data(), h_block_tables_.size() * 4, hipMemcpyHostToDevice, w.st));
        HIP_CHECK(hipMemcpyAsync(w.i_desc.p, desc.data(), (size_t)M * sizeof(int4), hipMemcpyHostToDevice, w.st));
        HIP_CHECK(hipMemcpyAsync(w.i_row_slot.p, row_slot.data(), (size_t)M * 4, hipMemcpyHostToDevice, w.st));
        HIP_CHECK(hipMemcpyAsync(w.i_row_pos.p, row_pos.data(), (size_t)M * 4, hipMemcpyHostToDevice, w.st));
        if (debug_sync())
            fprintf(stderr, "[aur] prefill M=%d desc=%p spk=%p temb=%p tpos=%p wte=%p wpe=%p h=%p d0=(%d,%d,%d) dl=(%d,%d,%d)\n", M,
                    w.i_desc.p, spk_table_.p, (const void*)text_emb_, (const void*)text_pos_, (const void*)wte_, (const void*)wpe_, w.h.p,
                    desc[0].x, desc[0].y, desc[0].z, desc[M - 1].x, desc[M - 1].y, desc[M - 1].z);
        launch_embed_prompt(w.i_desc.as<int4>(), spk_table_.as<float>(), text_emb_, text_pos_, wte_, wpe_, w.h.as<float>(), M, w.st);
        forward_rows(w, M, w.i_row_slot.as<int>(), w.i_row_pos.as<int>(), upload_qblocks(w, row_slot, row_pos));
        if (dbg_capture_) {
            dbg_lnf_.ensure((size_t)M * kHidden * 4);
            HIP_CHECK(hipMemcpyAsync(dbg_lnf_.p, w.xn.p, (size_t)M * kHidden * 4, hipMemcpyDeviceToDevice, w.st));
        }
        stats_.prefill_rows += M;
        sample_launch(w, sample_row, sample_slot, &next_kvpos);
        HIP_CHECK(hipStreamSynchronize(w.st));
        sample_collect(w, sample_slot);
        retire_finished();
    }
    // One decode step: embed -> 30 x (QKV GEMM, attention, proj, FC, proj2) -> final norms -> head GEMM -> sampler, 154
    // launches that depend only on device-resident state (hipGraph replay of the chain was measured within noise in round 2:
    // the launches are already back to back).
    void decode_kernels(RowWs& w, int Mk) {
        launch_embed_decode(w.i_row_slot.as<int>(), slot_tok_.as<int>(), slot_pos_.as<int>(), wte_, wpe_, w.h.as<float>(), Mk, w.st,
                            w.rows_cap / 16, w.stats.as<float2>(), w.row_meta.as<int>(), slot_kvpos_.as<int>(),
                            block_tables_.as<int>(), kMaxBlocks);
        forward_decode(w, Mk, w.i_row_slot.as<int>());
        sample_kernels_decode(w, Mk);
    }
    // ---- pipelined decode (default): the step's kernel chain depends only on device-resident state, so step s+1 is
    // enqueued BEFORE the host waits for the token / finished-flag read-back of step s (otherwise the GPU idles for the
    // host round trip, ~80 us of a 2.8 ms step).  A sequence that turns out to have finished in step s rides along in
    // step s+1 as a ghost: the sampler leaves a finished slot untouched and reports token -1, its K/V write lands in
    // a block the sequence had reserved, and the GEMMs are bitwise batch-invariant, so the other rows are unaffected.
    struct InFlight {
        bool on = false;
        std::vector<int> slots;   // live slots at launch, in row order
        int buf = 0;              // read-back buffer / event index
        bool profiled = false;
    };
    InFlight launch_decode_step(const std::vector<int>& active) {
        RowWs& w = ws_[0];
        const int M = (int)active.size();
        if (active != last_active_) {
            w.sample_slot = active;
            w.sample_row.resize(M);
            for (int i = 0; i < M; ++i) w.sample_row[i] = i;
            ensure_rows(w, M);
            HIP_CHECK(hipMemcpyAsync(w.i_row_slot.p, w.sample_slot.data(), (size_t)M * 4, hipMemcpyHostToDevice, st_));
            sample_upload(w, w.sample_row, w.sample_slot, nullptr);
            HIP_CHECK(hipStreamSynchronize(st_));   // pageable sources: the vectors may change before the copies run
        }
        InFlight f;
        f.on = true;
        f.slots = active;
        f.buf = rb_next_;
        rb_next_ ^= 1;
        f.profiled = cfg_.profile != 0 && (decode_step_count_++ % profile_every_ == 0);
        pin_rb_[f.buf].ensure(((size_t)cfg_.max_seqs * 2 + 16) * sizeof(int));
        // algorithmic bytes of this step: every live sequence's context (prompt + tokens so far, + 1 if the previous step is
        // still in flight) in K and V, all layers; the weights once
        step_kv_tokens_ = 0.0;
        for (int slot : active) {
            const Seq* sq = slot_owner_[slot];
            step_kv_tokens_ += (double)(sq->n_prompt + (int)sq->tokens.size() + (infl_.on ? 1 : 0));
        }
        stats_.decode_kv_bytes += (kv_half_ ? 4.0 : 8.0) * kHidden * step_kv_tokens_ * cfg_.n_layer;
        stats_.decode_weight_bytes += 4.0 * ((double)cfg_.n_layer * 12.0 * kHidden * kHidden + (double)kHidden * kMelVocab);
        HIP_CHECK(hipEventRecord(ev_ds_[f.buf], st_));
        decode_kernels(w, M);
        HIP_CHECK(hipEventRecord(ev_de_[f.buf], st_));
        // (the sampler writing the tokens straight into the pinned block instead of this 256-byte copy: 621.3 vs 621.8 ms per bench
        // step, 1.193 vs 1.194 ms per single-utterance decode step -- the copy is not on the critical path of the pipelined loop)
        sample_readback(w, M, st_, pin_rb_[f.buf].as<int>());
        HIP_CHECK(hipEventRecord(ev_rb_[f.buf], st_));
        if (f.profiled) profile_replay(w, M, w.i_row_slot.as<int>());   // behind the read-back: the step's tokens do not wait for it
        last_active_ = active;
        return f;
    }
    void collect_decode_step(InFlight& f) {
        HIP_CHECK(hipEventSynchronize(ev_rb_[f.buf]));
        const int* pin = pin_rb_[f.buf].as<int>();
        int rows = 0;
        for (size_t j = 0; j < f.slots.size(); ++j) {
            const int raw = pin[j];
            if (raw < 0) continue;   // ghost row of a sequence that had already finished
            const int tok = raw & ~kTokFinishedBit;
            Seq* s = slot_owner_[f.slots[j]];
            AUR_REQUIRE(s && s->state == SeqState::RUNNING, "decode read-back for a slot without a running sequence");
            s->tokens.push_back(tok);
            stats_.tokens_generated++;
            ++rows;
            if (raw & kTokFinishedBit) just_finished_.push_back(s);
        }
        float ms = 0.f;
        HIP_CHECK(hipEventElapsedTime(&ms, ev_ds_[f.buf], ev_de_[f.buf]));
        stats_.gpt_ms += ms;
        stats_.decode_ms += ms;
        stats_.decode_steps++;
        stats_.decode_rows += rows;
        if (f.profiled) collect_gemm_events();
        f.on = false;
    }
    // launch the next step ahead of the read-back only if some sequence is certain to need it
    bool worth_speculating(const std::vector<int>& active, const InFlight& pending) const {
        if (!pipeline_ || cfg_.second_pass || debug_sync()) return false;
        for (int slot : active) {
            const Seq* s = slot_owner_[slot];
            const bool in_pending = std::find(pending.slots.begin(), pending.slots.end(), slot) != pending.slots.end();
            const int n_after = (int)s->tokens.size() + (in_pending ? 1 : 0);
            if (n_after < s->params.max_tokens) return true;
        }
        return false;
    }
    void decode_pipelined(const std::vector<int>& active) {
        if (!infl_.on) infl_ = launch_decode_step(active);
        InFlight next;
        const auto t0 = std::chrono::steady_clock::now();
        if (worth_speculating(active, infl_)) next = launch_decode_step(active);
        const auto t1 = std::chrono::steady_clock::now();
        collect_decode_step(infl_);
        const auto t2 = std::chrono::steady_clock::now();
        if (host_trace_) {
            ht_launch_ns_ += std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count();
            ht_wait_ns_ += std::chrono::duration_cast<std::chrono::nanoseconds>(t2 - t1).count();
            if (++ht_n_ % 256 == 0)
                fprintf(stderr, "[aur host trace] rows %zu: launch_decode_step %.1f us of host time per step, wait for the read-back %.1f us\n", active.size(),
                        ht_launch_ns_ / 256e3, ht_wait_ns_ / 256e3), ht_launch_ns_ = ht_wait_ns_ = 0;
        }
        retire_finished();
        infl_ = next;
    }
    void drain_inflight() {   // only ghosts can be left once no sequence is running
        if (infl_.on) collect_decode_step(infl_);
        retire_finished();
    }
    void decode(const std::vector<int>& active) { decode_pipelined(active); }

    // ------------------------------------------------------------------ vocoder
    void ensure_voc() {
        if (voc_ready_) return;
        const bool f16 = cfg_.vocoder_fp16 != 0;
        auto layer = [&](const std::string& n) {
            ConvLayer l;
            l.wp = W("voc." + n + ".wp");
            l.bias = W("voc." + n + ".bias");
            if (f16) l.wp16 = W("voc16." + n + ".wp");
            return l;
        };
        v_pre_ = layer("conv_pre");
        for (int i = 0; i < 4; ++i) {
            v_ups_[i] = layer("ups." + std::to_string(i));
            for (int j = 0; j < 3; ++j)
                for (int c = 0; c < 3; ++c) {
                    const std::string p = "rb." + std::to_string(i * 3 + j) + ".";
                    v_c1_[i][j][c] = layer(p + "c1." + std::to_string(c));
                    v_c2_[i][j][c] = layer(p + "c2." + std::to_string(c));
                }
        }
        v_post_ = W("voc.conv_post.w");
        voc_ready_ = true;
    }
    void conv(ConvArgs& a, int KS, int DIL, double tot_in, double tot_out) {
        const bool prof = cfg_.profile != 0;
        ConvEvent* ev = nullptr;
        if (prof) {
            if (n_conv_events_ == conv_events_.size()) {
                ConvEvent e{};
                HIP_CHECK(hipEventCreate(&e.a));
                HIP_CHECK(hipEventCreate(&e.b));
                conv_events_.push_back(e);
            }
            ev = &conv_events_[n_conv_events_++];
            // class for the per-stage roofline: ResBlock convs by channel count (256 / 128 / 64 / 32 -> 0..3), everything
            // else (conv_pre, the four transposed convs) -> 4
            ev->kind = (a.ups_s == 0 && a.Cin == a.Cout) ? (a.Cout == 256 ? 0 : a.Cout == 128 ? 1 : a.Cout == 64 ? 2 : a.Cout == 32 ? 3 : 4) : 4;
            ev->flops = 2.0 * a.Cin * KS * a.Mtot * tot_in;
            // bytes in the dtype each tensor is actually stored in (the c1 -> c2 intermediate may be fp16)
            const double mrf_b = a.mrf_f16 ? 2.0 : 4.0;
            const double o_b = a.out_act_f16 ? 2.0 : 4.0;
            const double out_b = a.mrf_mode == 0 ? o_b : a.mrf_mode == 1 ? mrf_b : a.mrf_mode == 2 ? 2.0 * mrf_b : mrf_b + o_b;
            ev->bytes = (a.x_f16 ? 2.0 : 4.0) * a.Cin * tot_in + a.Cout * tot_out * (out_b + (a.res ? (a.res_f16 ? 2.0 : 4.0) : 0.0));
            HIP_CHECK(hipEventRecord(ev->a, st_voc_));
        }
        if (cfg_.vocoder_fp16)
            launch_conv1d_f16(a, KS, DIL, st_voc_);
        else
            launch_conv1d(a, KS, DIL, st_voc_);
        if (prof) HIP_CHECK(hipEventRecord(ev->b, st_voc_));
    }
    void round_conv(RoundArgs& a, int KS, int DIL, double tot) {
        const bool prof = cfg_.profile != 0;
        ConvEvent* ev = nullptr;
        if (prof) {
            if (n_conv_events_ == conv_events_.size()) {
                ConvEvent e{};
                HIP_CHECK(hipEventCreate(&e.a));
                HIP_CHECK(hipEventCreate(&e.b));
                conv_events_.push_back(e);
            }
            ev = &conv_events_[n_conv_events_++];
            ev->kind = a.C == 64 ? 2 : 3;
            ev->flops = 2.0 * (2.0 * a.C * KS * a.C * tot);   // conv1 + conv2
            // stream in, then: stream out (mode 0) | running sum out (1) | in and out (2) | running sum in, stage output out (3)
            const double io = a.mrf_mode == 0 ? 2.0 : a.mrf_mode == 1 ? 2.0 : 4.0;
            ev->bytes = a.C * tot * (2.0 + io);
            HIP_CHECK(hipEventRecord(ev->a, st_voc_));
        }
        launch_resblock_round_f16(a, KS, DIL, st_voc_);
        if (prof) HIP_CHECK(hipEventRecord(ev->b, st_voc_));
    }
    void collect_conv_events() {
        for (size_t i = 0; i < n_conv_events_; ++i) {
            float ms = 0.f;
            HIP_CHECK(hipEventElapsedTime(&ms, conv_events_[i].a, conv_events_[i].b));
            stats_.conv_ms += ms;
            stats_.conv_flops += conv_events_[i].flops;
            stats_.conv_bytes += conv_events_[i].bytes;
            stats_.conv_launches++;
            const int k = conv_events_[i].kind;
            if (k >= 0 && k < 5) {
                stats_.conv_class_ms[k] += ms;
                stats_.conv_class_flops[k] += conv_events_[i].flops;
                stats_.conv_class_bytes[k] += conv_events_[i].bytes;
                stats_.conv_class_launches[k]++;
            }
        }
        n_conv_events_ = 0;
        if (voc_timed_) {
            float ms = 0.f;
            HIP_CHECK(hipEventElapsedTime(&ms, ev_va_, ev_vb_));
            stats_.vocoder_ms += ms;
            voc_timed_ = false;
        }
    }
    // latents: d_lat + lat_row[b]*lat_bstride (lat_row nullable => b); wav: d_wav + b*wav_bstride
    void run_vocoder(int B, const std::vector<int>& n_lat, const float* d_lat, long lat_bstride,
                     const std::vector<int>* lat_row, const std::vector<int>& cond_row, float* d_wav, long wav_bstride) {
        ensure_voc();
        if (!ev_va_) {
            HIP_CHECK(hipEventCreate(&ev_va_));
            HIP_CHECK(hipEventCreate(&ev_vb_));
        }
        std::vector<int>& meta = h_meta_;   // member: the async H2D copy below must not outlive its source
        meta.assign(4 * (size_t)B, 0);
        int maxT = 0;
        double totT = 0;
        for (int b = 0; b < B; ++b) {
            meta[b] = n_lat[b];
            meta[B + b] = frames_for(n_lat[b]);
            meta[2 * B + b] = cond_row[b];
            meta[3 * B + b] = lat_row ? (*lat_row)[b] : b;
            maxT = std::max(maxT, meta[B + b]);
            totT += meta[B + b];
        }
        v_meta_.ensure(meta.size() * 4);
        HIP_CHECK(hipMemcpyAsync(v_meta_.p, meta.data(), meta.size() * 4, hipMemcpyHostToDevice, st_voc_));
        const int* d_nlat = v_meta_.as<int>();
        const int* d_len = d_nlat + B;
        const int* d_cond = d_nlat + 2 * B;
        const int* d_latrow = d_nlat + 3 * B;
        const size_t T = (size_t)maxT, Bz = (size_t)B;
        v_z_.ensure(Bz * 1024 * T * 4);
        v_s0_.ensure(Bz * 512 * T * 4);
        for (auto* b : {&v_B_, &v_D_, &v_E_}) b->ensure(Bz * 8192 * T * 4);
        if (!v_zero_.p) {
            v_zero_.ensure(256);
            HIP_CHECK(hipMemsetAsync(v_zero_.p, 0, 256, st_voc_));
        }
        if (xt_f16_) {   // the residual stream as activated halves (stage input, running value)
            v_Ah_.ensure(Bz * 8192 * T * 2);
            v_Ch_.ensure(Bz * 8192 * T * 2);
            v_Ch2_.ensure(Bz * 8192 * T * 2);
        } else {
            v_A_.ensure(Bz * 8192 * T * 4);
            v_C_.ensure(Bz * 8192 * T * 4);
        }
        HIP_CHECK(hipEventRecord(ev_va_, st_voc_));
        // fp16 vocoder: z leaves the interpolation as interleaved halves (what conv_pre's staging rounded it to anyway) and conv_pre
        // runs on the LDS-DMA kernel like every other wide conv (round 5: the register-staged kernel, with 32 spilled VGPRs)
        const bool z_f16 = xt_f16_ && conv_dma_;
        launch_interp2(d_lat, lat_bstride, d_latrow, d_nlat, d_len, v_z_.as<float>(), (long)T, (long)(1024 * T), 1024, B, maxT, st_voc_, z_f16);
        const float* condt = voc_cond_.as<float>();
        ConvArgs a{};
        a.base_len = d_len; a.B = B; a.cond_row = d_cond; a.cond_stride = kCondStride;
        // conv_pre (+ cond_layer)
        a.x = v_z_.as<float>(); a.x_f16 = z_f16 ? 1 : 0; a.zeros = z_f16 ? v_zero_.p : nullptr;
        a.wp = v_pre_.wp; a.wp16 = v_pre_.wp16; a.bias = v_pre_.bias; a.cond = condt; a.res = nullptr; a.mrf = nullptr;
        a.out = v_s0_.as<float>(); a.len_mul = 1; a.Cin = 1024; a.Mtot = 512; a.Cout = 512;
        a.x_stride = (long)T; a.o_stride = (long)T; a.x_bstride = (long)(1024 * T); a.o_bstride = (long)(512 * T);
        a.padl = 3; a.slope = 1.0f; a.ups_s = 0; a.ups_p = 0; a.mrf_mode = 0; a.max_len = maxT;
        // fp16 vocoder: every stage input (conv_pre's output, the MRF means) is stored the way its consumer stages it,
        // fp16(lrelu(x, 0.1)) interleaved -- 0.01 for the last one, which feeds conv_post
        if (xt_f16_) { a.out_act_f16 = 1; a.out_slope = 0.1f; }
        conv(a, 7, 1, totT, totT);
        const int rates[4] = {8, 8, 2, 2}, kern[4] = {16, 16, 4, 4}, chans[4] = {256, 128, 64, 32};
        const int rk[3] = {3, 7, 11}, rd[3] = {1, 3, 5};
        const float* in = v_s0_.as<float>();
        int Cin = 512, mul = 1, cond_off = 512;
        float *A = v_A_.as<float>(), *Bb = v_B_.as<float>(), *Cb = v_C_.as<float>(), *D = v_D_.as<float>(), *E = v_E_.as<float>();
        void *Ah = v_Ah_.p, *Ch = v_Ch_.p, *Ch2 = v_Ch2_.p;   // fp16 vocoder: the activated residual stream (stage input / after rounds 0, 1)
        for (int i = 0; i < 4; ++i) {
            const int s = rates[i], C = chans[i], mul_out = mul * s;
            const long Lin = (long)T * mul, Lout = (long)T * mul_out;
            // transposed conv as 2-tap polyphase conv over virtual channels co*s + r
            a = ConvArgs{};
            a.base_len = d_len; a.B = B; a.cond_row = d_cond; a.cond_stride = kCondStride;
            a.x = in; a.x_f16 = xt_f16_ ? 1 : 0; a.zeros = (xt_f16_ && conv_dma_) ? v_zero_.p : nullptr; a.wp = v_ups_[i].wp; a.wp16 = v_ups_[i].wp16; a.bias = v_ups_[i].bias; a.cond = condt + cond_off; a.out = A;
            a.len_mul = mul; a.Cin = Cin; a.Mtot = C * s; a.Cout = C;
            a.x_stride = Lin; a.x_bstride = (long)Cin * Lin; a.o_stride = Lout; a.o_bstride = (long)C * Lout;
            a.padl = 1; a.slope = 0.1f; a.ups_s = s; a.ups_p = (kern[i] - s) / 2; a.mrf_mode = 0; a.max_len = maxT * mul;
            if (xt_f16_) { a.out = reinterpret_cast<float*>(Ah); a.out_act_f16 = 1; a.out_slope = 0.1f; }
            conv(a, 2, 1, totT * mul, totT * mul_out);
            cond_off += C;
            // the 64- and 32-channel stages (HBM-bound as two launches per round): one fused launch per ResBlock round, the
            // conv1 -> conv2 intermediate stays in LDS and the residual comes from the staged window (vocoder_kernels.h, RoundArgs);
            // the stream ping-pongs Ah -> Ch -> Ch2 because neighbouring tiles read each other's halo
            const bool fused = xt_f16_ && conv_dma_ && C <= 64;
            for (int j = 0; j < 3 && fused; ++j)
                for (int c = 0; c < 3; ++c) {
                    RoundArgs ra{};
                    ra.y = (c == 0) ? Ah : (c == 1) ? Ch : Ch2;
                    ra.out = (c == 0) ? Ch : (c == 1) ? Ch2 : nullptr;
                    ra.w1 = v_c1_[i][j][c].wp16; ra.w2 = v_c2_[i][j][c].wp16; ra.b1 = v_c1_[i][j][c].bias; ra.b2 = v_c2_[i][j][c].bias;
                    ra.base_len = d_len; ra.len_mul = mul_out; ra.stride = Lout; ra.bstride = (long)C * Lout; ra.C = C;
                    ra.B = B; ra.max_len = maxT * mul_out; ra.zeros = v_zero_.p;
                    if (c == 2) {
                        ra.mrf = D; ra.e_out = E; ra.mrf_mode = (j == 0) ? 1 : (j == 1) ? 2 : 3;
                        ra.e_slope = (i == 3) ? 0.01f : 0.1f;
                    }
                    round_conv(ra, rk[j], rd[c], totT * mul_out);
                }
            for (int j = 0; j < 3 && !fused; ++j)
                for (int c = 0; c < 3; ++c) {
                    // fp16 vocoder: the residual stream of a ResBlock lives in HBM as activated interleaved halves, fp16(lrelu(x)):
                    // Ah (the transposed conv's output, shared by the stage's three ResBlocks) in round 0, Ch (updated in place by
                    // the residual convs) afterwards.  First convs stage it as is; residual convs read it back and undo the
                    // activation (ConvArgs::res_f16).  fp32 vocoder: A / Cb, fp32.
                    const float* r = xt_f16_ ? reinterpret_cast<const float*>(c == 0 ? Ah : Ch) : (c == 0 ? A : Cb);
                    ConvArgs b1{};
                    b1.base_len = d_len; b1.B = B;
                    b1.x = r; b1.x_f16 = xt_f16_ ? 1 : 0; b1.zeros = conv_dma_ ? v_zero_.p : nullptr;
                    b1.wp = v_c1_[i][j][c].wp; b1.wp16 = v_c1_[i][j][c].wp16; b1.bias = v_c1_[i][j][c].bias; b1.out = Bb;
                    b1.len_mul = mul_out; b1.Cin = C; b1.Mtot = C; b1.Cout = C;
                    b1.x_stride = Lout; b1.o_stride = Lout; b1.x_bstride = (long)C * Lout; b1.o_bstride = (long)C * Lout;
                    b1.padl = (rk[j] - 1) / 2 * rd[c]; b1.slope = 0.1f; b1.max_len = maxT * mul_out;
                    if (xt_f16_) { b1.out_act_f16 = 1; b1.out_slope = 0.1f; }   // c1 -> c2 intermediate: fp16(lrelu(.)), what c2 stages anyway
                    conv(b1, rk[j], rd[c], totT * mul_out, totT * mul_out);
                    ConvArgs b2 = b1;
                    b2.out_act_f16 = 0;
                    b2.x = Bb; b2.wp = v_c2_[i][j][c].wp; b2.wp16 = v_c2_[i][j][c].wp16; b2.bias = v_c2_[i][j][c].bias;
                    b2.res = r;
                    if (xt_f16_) { b2.res_f16 = 1; b2.res_unact = 1.0f / 0.1f; }
                    b2.padl = (rk[j] - 1) / 2;
                    if (c < 2) {
                        if (xt_f16_) { b2.out = reinterpret_cast<float*>(Ch); b2.out_act_f16 = 1; b2.out_slope = 0.1f; }
                        else b2.out = Cb;
                    } else {
                        b2.mrf = D; b2.out = E; b2.mrf_mode = (j == 0) ? 1 : (j == 1) ? 2 : 3;
                        b2.mrf_f16 = xt_f16_ ? 1 : 0;   // D: running sum of the three ResBlock outputs (halves in fp16 mode)
                        if (xt_f16_ && j == 2) { b2.out_act_f16 = 1; b2.out_slope = (i == 3) ? 0.01f : 0.1f; }   // E: the next stage's input
                    }
                    conv(b2, rk[j], 1, totT * mul_out, totT * mul_out);
                }
            in = E; Cin = C; mul = mul_out;
        }
        launch_conv_post(E, v_post_, d_wav, d_len, 256, 32, (long)T * 256, (long)32 * T * 256, wav_bstride, 0.01f, B, maxT * 256, st_voc_, xt_f16_);
        HIP_CHECK(hipEventRecord(ev_vb_, st_voc_));
        voc_timed_ = true;
        stats_.vocoder_batches++;
    }
    void voc_launch() {
        const int B = (int)std::min(voc_queue_.size(), (size_t)cfg_.max_seqs);
        voc_batch_.assign(voc_queue_.begin(), voc_queue_.begin() + B);
        voc_queue_.erase(voc_queue_.begin(), voc_queue_.begin() + B);
        voc_nl_.resize(B);
        std::vector<int> rows(B), cond(B);
        voc_max_samples_ = 0;
        for (int b = 0; b < B; ++b) {
            voc_nl_[b] = (int)voc_batch_[b]->tokens.size();
            rows[b] = voc_batch_[b]->pool_idx;
            cond[b] = voc_batch_[b]->spk_row;
            voc_max_samples_ = std::max(voc_max_samples_, frames_for(voc_nl_[b]) * 256);
        }
        if (!wav_direct_) tmp_wav_.ensure((size_t)B * voc_max_samples_ * 4);
        {
            std::lock_guard<std::mutex> lk(mu_);
            voc_block_ = nullptr;
            for (auto& blk : result_blocks_)
                if (blk->refs == 0) voc_block_ = blk.get();
            if (!voc_block_) {
                result_blocks_.emplace_back(new PinBlock());
                voc_block_ = result_blocks_.back().get();
            }
            voc_block_->refs = 1;   // held by the in-flight batch
            voc_block_held_ = true;
        }
        if (fail_at_voc_ > 0 && --fail_at_voc_ == 0)   // AUR_TEST_FAIL_VOC=n: fault injection inside the n-th vocoder launch
            throw HipError("injected failure (AUR_TEST_FAIL_VOC)");
        voc_block_->buf.ensure((size_t)B * voc_max_samples_ * 4 + (cfg_.return_latents ? (size_t)B * kMaxLatRows * kHidden * 4 : 0));
        HIP_CHECK(hipStreamWaitEvent(st_voc_, ev_lat_, 0));   // latents parked by the main stream
        // The waveform goes straight into the pinned result block: conv_post (Cin -> 1, + tanh: 0.4 ms of HBM-bound work per 64
        // utterances) stores its samples over PCIe -- coalesced 1 KiB per workgroup -- instead of into device memory with one D2H copy
        // per sequence behind it (80 MB in 64 copies: 3.2 ms of the 64-utterance step with nothing to overlap, the batch's last
        // decode step being over).  hipHostMalloc memory is device-visible and coherent; the event below orders the host's reads.
        float* hw = voc_block_->buf.as<float>();
        float* hl = hw + (size_t)B * voc_max_samples_;
        run_vocoder(B, voc_nl_, latpool_.as<float>(), (long)kMaxLatRows * kHidden, &rows, cond, wav_direct_ ? hw : tmp_wav_.as<float>(),
                    voc_max_samples_);
        for (int b = 0; b < B; ++b) {
            const int ns = frames_for(voc_nl_[b]) * 256;
            if (!wav_direct_)
                HIP_CHECK(hipMemcpyAsync(hw + (size_t)b * voc_max_samples_, tmp_wav_.as<float>() + (long)b * voc_max_samples_,
                                         (size_t)ns * 4, hipMemcpyDeviceToHost, st_voc_));
            if (cfg_.return_latents)
                HIP_CHECK(hipMemcpyAsync(hl + (size_t)b * kMaxLatRows * kHidden,
                                         latpool_.as<float>() + (long)rows[b] * kMaxLatRows * kHidden,
                                         (size_t)voc_nl_[b] * kHidden * 4, hipMemcpyDeviceToHost, st_voc_));
        }
        HIP_CHECK(hipEventRecord(ev_voc_done_, st_voc_));
        voc_active_ = true;
    }
    // finalize the in-flight vocoder batch if it has completed (or wait for it)
    void voc_poll(bool wait) {
        if (!voc_active_) return;
        if (wait) {
            HIP_CHECK(hipEventSynchronize(ev_voc_done_));
        } else {
            const hipError_t q = hipEventQuery(ev_voc_done_);
            if (q == hipErrorNotReady) return;
            HIP_CHECK(q);
        }
        collect_conv_events();
        const int B = (int)voc_batch_.size();
        const float* hw = voc_block_->buf.as<float>();
        const float* hl = hw + (size_t)B * voc_max_samples_;
        for (int b = 0; b < B; ++b) {
            Seq* s = voc_batch_[b];
            const int ns = frames_for(voc_nl_[b]) * 256;
            s->wav = hw + (size_t)b * voc_max_samples_;
            s->n_samples = ns;
            if (cfg_.return_latents) {
                s->latents = hl + (size_t)b * kMaxLatRows * kHidden;
                s->n_latent_rows = voc_nl_[b];
            }
            stats_.samples_generated += ns;
            std::lock_guard<std::mutex> lk(mu_);
            s->result_block = voc_block_;
            voc_block_->refs++;
            latpool_free_.push_back(s->pool_idx);
            s->pool_idx = -1;
            spk_info_[s->spk_row].refs--;
            s->state = SeqState::DONE;
            done_.push_back(s);
            finished_total_++;
        }
        {
            std::lock_guard<std::mutex> lk(mu_);
            voc_block_->refs--;   // the batch's own hold
            voc_block_held_ = false;
        }
        voc_batch_.clear();
        voc_active_ = false;
    }

    aur_config cfg_;
    int device_;
    hipStream_t st_ = nullptr;
    hipEvent_t ev_a_ = nullptr, ev_b_ = nullptr, ev_va_ = nullptr, ev_vb_ = nullptr;
    bool voc_timed_ = false;
    std::unordered_map<std::string, Tensor> w_;
    bool gpt_ready_ = false, voc_ready_ = false;
    std::vector<LayerW> layers_;
    std::vector<std::unique_ptr<DevBuf>> packed_;   // pack_wt16 copies (decode GEMM layout)
    DevBuf fold_scratch_;                           // load-time scratch of launch_fold_ln
    int gemm_prec_ = 1;                             // GemmRowsArgs.prec of every decode GEMM (aur_config.gemm_f32_exact)
    const float* thead_ = nullptr;
    int fail_at_step_ = 0;              // AUR_TEST_FAIL_STEP=n: throw inside the n-th aur_step (recovery test)
    int fail_at_voc_ = 0;               // AUR_TEST_FAIL_VOC=n: throw inside the n-th vocoder launch, after the batch left the queue
    bool voc_block_held_ = false;       // the vocoder batch holds a reference on voc_block_
    const float *wte_ = nullptr, *wpe_ = nullptr, *lnfw_ = nullptr, *lnfb_ = nullptr, *fnw_ = nullptr, *fnb_ = nullptr,
                *headT_ = nullptr, *headb_ = nullptr, *text_emb_ = nullptr, *text_pos_ = nullptr;
    int text_vocab_ = 0, text_positions_ = 0;
    // KV pool
    DevBuf kv_;
    long n_blocks_ = 0, kv_layer_stride_ = 0;   // stride in elements
    bool kv_half_ = false;                        // aur_config.kv_fp16
    void* kv_layer(int l) const { return (char*)kv_.p + (size_t)l * kv_layer_stride_ * (kv_half_ ? 2 : 4); }
    std::vector<int> free_blocks_;
    std::vector<int> h_block_tables_;
    // per-slot device state
    DevBuf slot_tok_, slot_pos_, slot_kvpos_, slot_ngen_, slot_finished_, temperature_, top_p_, top_k_, rep_, max_tokens_,
        ignore_stop_, seed_, block_tables_, seen_, latents_;
    DevBuf spk_table_, spk_emb_, voc_cond_, zero_bias_;
    std::map<uint64_t, int> spk_rows_;
    struct SpeakerInfo {
        int blocks[2] = {-1, -1};
        bool ready = false;
        int refs = 0;             // sequences submitted with this voice and not delivered yet
        uint64_t last_use = 0;    // LRU clock
    };
    uint64_t spk_clock_ = 0;
    std::vector<SpeakerInfo> spk_info_;
    bool share_prefix_ = true;        // AUR_SHARE_PREFIX=0 disables
    bool share_prefix_now_ = true;    // dbg_prefill turns it off to return every prompt row
    std::mutex gpu_mu_;
    std::unique_ptr<CondNet> cond_net_;
    void* comm_ = nullptr;   // ncclComm_t
    int comm_rank_ = 0, comm_world_ = 1;
    DevBuf bcast_buf_;
    // row workspace
    RowWs ws_[2];
    // vocoder stage
    hipStream_t st_voc_ = nullptr;
    hipEvent_t ev_lat_ = nullptr, ev_voc_done_ = nullptr;
    DevBuf latpool_;
    std::vector<int> latpool_free_;
    std::deque<Seq*> voc_queue_;
    std::vector<Seq*> voc_batch_;
    std::vector<int> voc_nl_;
    int voc_max_samples_ = 0;
    bool voc_active_ = false;
    std::vector<std::unique_ptr<PinBlock>> result_blocks_;   // pool; a block is free when refs == 0 (guarded by mu_)
    PinBlock* voc_block_ = nullptr;                          // block of the in-flight vocoder batch
    InFlight infl_;                     // decode step launched but not yet collected
    PinBuf pin_rb_[2];
    hipEvent_t ev_rb_[2] = {nullptr, nullptr}, ev_ds_[2] = {nullptr, nullptr}, ev_de_[2] = {nullptr, nullptr};
    int rb_next_ = 0;
    bool sampler_full_sort_ = false;    // AUR_SAMPLER_FULL_SORT=1: disable the sampler's top-k fast path (A/B)
    bool host_trace_ = false;           // AUR_HOST_TRACE=1: host time of enqueueing a decode step vs waiting for the previous one (stderr, every 256 steps)
    long ht_launch_ns_ = 0, ht_wait_ns_ = 0, ht_n_ = 0;
    bool pipeline_ = true;              // AUR_DECODE_PIPELINE=0: wait for every read-back before launching the next step
    bool conv_dma_ = true;              // ResBlock convs of the fp16 vocoder on the LDS-DMA staged kernel
    bool xt_f16_ = false;               // set in the constructor: fp16 vocoder => fp16 c1 -> c2 intermediate (AUR_XT_F16=0 disables)
    std::vector<int> last_active_;      // live slots whose row indices are resident in the decode chain's index buffers
    hipStream_t st2_ = nullptr;
    DevBuf i_init_;
    DevBuf dbg_lnf_, dbg_logits_;
    bool dbg_capture_ = false;
    // vocoder
    ConvLayer v_pre_, v_ups_[4], v_c1_[4][3][3], v_c2_[4][3][3];
    const float* v_post_ = nullptr;
    DevBuf v_meta_, v_z_, v_s0_, v_A_, v_B_, v_C_, v_D_, v_E_, v_Ch_, v_Ch2_, v_Ah_, v_zero_, tmp_lat_, tmp_wav_;
    std::vector<ConvEvent> conv_events_;
    std::vector<ConvEvent> gemm_events_;
    size_t n_gemm_events_ = 0;
    int profile_every_ = kProfileEvery;
    DevBuf prof_q_, prof_h_, prof_stats_, prof_act_, prof_kv_;   // output scratch of profile_replay
    DevBuf ksp_buf_, ksp_cnt_;   // GemmRowsArgs::ksp_buf / ksp_cnt
    bool gemm_presplit_ = true;  // prompt-row GEMMs read their weights as the three bf16 planes packed at load time (launch_pack_wsplit)
    double step_kv_tokens_ = 0.0;       // sum of context lengths of the step being launched (profile accounting)
    long decode_step_count_ = 0;
    float event_overhead_ms_ = -1.f;
    std::vector<int> h_meta_;
    size_t n_conv_events_ = 0;
    // sequences
    std::mutex mu_;
    uint64_t next_id_ = 1;
    std::unordered_map<uint64_t, std::unique_ptr<Seq>> seqs_;
    std::deque<Seq*> waiting_, done_;
    std::vector<uint64_t> cancel_pending_;   // aur_cancel requests for sequences past the waiting queue (guarded by mu_)
    static constexpr int kAdmitHoldSteps = 32;   // longest hold of an admissible request by aur_config.admit_min_batch, in aur_steps
    static constexpr int kUrgentHoldSteps = 8;   // ... while an urgent sequence runs (arrivals are grouped, not kept waiting long)
    int admit_hold_steps_ = 0;
    static constexpr int kVocHoldSteps = 16;     // longest wait of a finished sequence for a fuller vocoder batch, in aur_steps (~30 ms)
    int voc_hold_steps_ = 0;
    bool wav_direct_ = true;   // AUR_WAV_DIRECT
    std::vector<Seq*> slot_owner_;
    std::vector<Seq*> just_finished_;
    int64_t finished_total_ = 0;
    aur_stats stats_{};
};

}  // namespace aur

// ================================================================================================ C ABI
struct aur_engine {
    aur::Engine impl;
    aur_engine(const aur_config& c, int dev) : impl(c, dev) {}
};

template <class F>
static int guarded(F&& f) {
    try {
        f();
        return AUR_OK;
    } catch (const aur::InvalidArgument& e) {
        aur::g_last_error = e.what();
        return AUR_E_INVALID;
    } catch (const aur::StateError& e) {
        aur::g_last_error = e.what();
        return AUR_E_STATE;
    } catch (const aur::HipError& e) {
        aur::g_last_error = e.what();
        return AUR_E_HIP;
    } catch (const std::bad_alloc&) {
        aur::g_last_error = "out of host memory";
        return AUR_E_NOMEM;
    } catch (const std::exception& e) {
        aur::g_last_error = e.what();
        return AUR_E_HIP;
    }
}
#define CHECK_PTR(p)                                    \
    if (!(p)) {                                         \
        aur::g_last_error = "null argument: " #p;       \
        return AUR_E_INVALID;                           \
    }

extern "C" {

const char* aur_last_error(void) { return aur::g_last_error.c_str(); }
int aur_version(void) { return 1; }

int aur_engine_create(const aur_config* cfg, int device_id, aur_engine** out) {
    CHECK_PTR(cfg);
    CHECK_PTR(out);
    *out = nullptr;
    return guarded([&] {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
            throw aur::StateError("no HIP device visible: the MI355X path has no CPU fallback");
        AUR_REQUIRE(device_id >= 0 && device_id < n, "device_id out of range");
        *out = new aur_engine(*cfg, device_id);
    });
}
int aur_engine_destroy(aur_engine* e) {
    CHECK_PTR(e);
    return guarded([&] { delete e; });
}
int aur_load_weights(aur_engine* e, const aur_tensor_desc* t, size_t n) {
    CHECK_PTR(e);
    CHECK_PTR(t);
    return guarded([&] { e->impl.load(t, n); });
}
int aur_set_conditioning(aur_engine* e, uint64_t key, const float* gpt_cond, const float* spk) {
    CHECK_PTR(e);
    CHECK_PTR(gpt_cond);
    CHECK_PTR(spk);
    return guarded([&] { e->impl.set_conditioning(key, gpt_cond, spk, false); });
}
int aur_set_conditioning_device(aur_engine* e, uint64_t key, const float* d_gpt_cond, const float* d_spk) {
    CHECK_PTR(e);
    CHECK_PTR(d_gpt_cond);
    CHECK_PTR(d_spk);
    return guarded([&] { e->impl.set_conditioning(key, d_gpt_cond, d_spk, true); });
}
int aur_has_conditioning(aur_engine* e, uint64_t key, int32_t* out) {
    CHECK_PTR(e);
    CHECK_PTR(out);
    return guarded([&] { *out = e->impl.has_conditioning(key) ? 1 : 0; });
}
int aur_comm_unique_id(uint8_t* out128) {
    CHECK_PTR(out128);
    return guarded([&] { aur::Engine::comm_unique_id(out128); });
}
int aur_comm_init(aur_engine* e, const uint8_t* id128, int32_t rank, int32_t world) {
    CHECK_PTR(e);
    CHECK_PTR(id128);
    return guarded([&] { e->impl.comm_init(id128, rank, world); });
}
int aur_broadcast_conditioning(aur_engine* e, uint64_t key, int32_t root) {
    CHECK_PTR(e);
    return guarded([&] { e->impl.broadcast_conditioning(key, root); });
}
int aur_comm_info(aur_engine* e, int32_t* n_ranks, int32_t* rank) {
    CHECK_PTR(e);
    CHECK_PTR(n_ranks);
    CHECK_PTR(rank);
    return guarded([&] { e->impl.comm_info(n_ranks, rank); });
}
int aur_conditioning_checksum(aur_engine* e, uint64_t key, uint64_t* out) {
    CHECK_PTR(e);
    CHECK_PTR(out);
    return guarded([&] { *out = e->impl.conditioning_checksum(key); });
}
int aur_compute_conditioning(aur_engine* e, const float* const* pcm, const int32_t* n_samples, int32_t n_refs, const aur_cond_params* p,
                             float* out_gpt_cond, float* out_spk_emb) {
    CHECK_PTR(e);
    CHECK_PTR(pcm);
    CHECK_PTR(n_samples);
    CHECK_PTR(p);
    CHECK_PTR(out_gpt_cond);
    CHECK_PTR(out_spk_emb);
    return guarded([&] { e->impl.compute_conditioning(pcm, n_samples, n_refs, *p, out_gpt_cond, out_spk_emb); });
}
int aur_submit(aur_engine* e, const aur_seq_desc* seq, uint64_t* seq_id) {
    CHECK_PTR(e);
    CHECK_PTR(seq);
    CHECK_PTR(seq_id);
    return guarded([&] { *seq_id = e->impl.submit(*seq); });
}
int aur_step(aur_engine* e, int32_t* n_live, int32_t* n_finished_total) {
    CHECK_PTR(e);
    return guarded([&] { e->impl.step(n_live, n_finished_total); });
}
int aur_poll_finished(aur_engine* e, aur_result* out, size_t cap, size_t* n) {
    CHECK_PTR(e);
    CHECK_PTR(out);
    CHECK_PTR(n);
    return guarded([&] { *n = e->impl.poll(out, cap); });
}
int aur_release(aur_engine* e, uint64_t seq_id) {
    CHECK_PTR(e);
    return guarded([&] { e->impl.release(seq_id); });
}
int aur_cancel(aur_engine* e, uint64_t seq_id) {
    CHECK_PTR(e);
    return guarded([&] { e->impl.cancel(seq_id); });
}
int aur_vocode(aur_engine* e, const float* latents, const int32_t* n_lat, int32_t B, int32_t t_max,
               uint64_t speaker_key, float* wav_out, int64_t wav_stride, int32_t* n_samples_out) {
    CHECK_PTR(e);
    CHECK_PTR(latents);
    CHECK_PTR(n_lat);
    CHECK_PTR(wav_out);
    return guarded([&] { e->impl.vocode_host(latents, n_lat, B, t_max, speaker_key, wav_out, wav_stride, n_samples_out); });
}
int aur_sync(aur_engine* e) {
    CHECK_PTR(e);
    return guarded([&] { e->impl.sync(); });
}
int aur_get_stats(aur_engine* e, aur_stats* out) {
    CHECK_PTR(e);
    CHECK_PTR(out);
    return guarded([&] { *out = e->impl.stats(); });
}
int aur_reset_stats(aur_engine* e) {
    CHECK_PTR(e);
    return guarded([&] { e->impl.reset_stats(); });
}
int aur_set_profile(aur_engine* e, int32_t every) {
    CHECK_PTR(e);
    return guarded([&] { e->impl.set_profile(every); });
}
int aur_dbg_gemm(aur_engine* e, const float* X, const float* W, float* out, int32_t M, int32_t N, int32_t K) {
    CHECK_PTR(e);
    return guarded([&] { e->impl.dbg_gemm(X, W, out, M, N, K); });
}
int aur_dbg_gemm_rows(aur_engine* e, const float* X, const float* W, const float* bias, const float* gamma, const float* beta,
                      float* out, int32_t M, int32_t N, int32_t K, int32_t epi, int32_t ln) {
    CHECK_PTR(e);
    CHECK_PTR(X);
    CHECK_PTR(W);
    CHECK_PTR(out);
    return guarded([&] { e->impl.dbg_gemm_rows(X, W, bias, gamma, beta, out, M, N, K, epi, ln != 0); });
}
int aur_dbg_gemm_rows_ksplit_stress(aur_engine* e, int32_t M, int32_t iters, int64_t* mismatches_out) {
    CHECK_PTR(e);
    CHECK_PTR(mismatches_out);
    return guarded([&] { *mismatches_out = e->impl.dbg_gemm_rows_ksplit_stress(M, iters); });
}
int aur_dbg_lane_xor_selftest(aur_engine* e, int32_t blocks, int64_t* mismatches_out) {
    CHECK_PTR(e);
    CHECK_PTR(mismatches_out);
    return guarded([&] { *mismatches_out = e->impl.dbg_lane_xor_selftest(blocks); });
}
int aur_dbg_paged_attention(aur_engine* e, const float* q, const float* k, const float* v, const int32_t* ctx, int32_t M, int32_t ctx_max,
                            int32_t shared, int32_t kv_half, float* out) {
    CHECK_PTR(e);
    CHECK_PTR(q);
    CHECK_PTR(k);
    CHECK_PTR(v);
    CHECK_PTR(ctx);
    CHECK_PTR(out);
    return guarded([&] { e->impl.dbg_paged_attention(q, k, v, ctx, M, ctx_max, shared, kv_half != 0, out); });
}
int aur_dbg_prompt_attention(aur_engine* e, const float* q, const float* k, const float* v, const int32_t* row_seq, const int32_t* row_pos,
                             int32_t M, int32_t n_seq, int32_t ctx_max, int32_t shared, int32_t kv_half, float* out) {
    CHECK_PTR(e);
    CHECK_PTR(q);
    CHECK_PTR(k);
    CHECK_PTR(v);
    CHECK_PTR(row_seq);
    CHECK_PTR(row_pos);
    CHECK_PTR(out);
    return guarded([&] { e->impl.dbg_prompt_attention(q, k, v, row_seq, row_pos, M, n_seq, ctx_max, shared, kv_half != 0, out); });
}
int aur_dbg_layernorm(aur_engine* e, const float* h, const float* gamma, const float* beta, float* out, int32_t M) {
    CHECK_PTR(e);
    return guarded([&] { e->impl.dbg_layernorm(h, gamma, beta, out, M); });
}
int aur_dbg_conv1d(aur_engine* e, const float* x, const float* wp, const float* bias, const float* res, float* out,
                   const int32_t* lens, int32_t B, int32_t Cin, int32_t Mtot, int32_t Cout, int32_t L, int32_t KS,
                   int32_t DIL, int32_t padl, float slope, int32_t ups_s, int32_t ups_p) {
    CHECK_PTR(e);
    return guarded([&] { e->impl.dbg_conv1d(x, wp, bias, res, out, lens, B, Cin, Mtot, Cout, L, KS, DIL, padl, slope, ups_s, ups_p, false); });
}
int aur_dbg_conv1d_f16(aur_engine* e, const float* x, const void* wp16, const float* bias, const float* res, float* out,
                       const int32_t* lens, int32_t B, int32_t Cin, int32_t Mtot, int32_t Cout, int32_t L, int32_t KS,
                       int32_t DIL, int32_t padl, float slope, int32_t ups_s, int32_t ups_p) {
    CHECK_PTR(e);
    return guarded([&] { e->impl.dbg_conv1d(x, wp16, bias, res, out, lens, B, Cin, Mtot, Cout, L, KS, DIL, padl, slope, ups_s, ups_p, true); });
}
int aur_dbg_prefill(aur_engine* e, const int32_t* text_ids, int32_t n_text, uint64_t speaker_key, float repetition_penalty,
                    float* lnf_rows_out, float* logits_out) {
    CHECK_PTR(e);
    CHECK_PTR(text_ids);
    return guarded([&] { e->impl.dbg_prefill(text_ids, n_text, speaker_key, repetition_penalty, lnf_rows_out, logits_out); });
}
int aur_dbg_sample(aur_engine* e, const float* logits, int32_t B, float temperature, float top_p, int32_t top_k,
                   float repetition_penalty, const uint8_t* seen, uint32_t seed, int32_t step, int32_t* tokens_out) {
    CHECK_PTR(e);
    CHECK_PTR(logits);
    CHECK_PTR(tokens_out);
    return guarded([&] { e->impl.dbg_sample(logits, B, temperature, top_p, top_k, repetition_penalty, seen, seed, step, tokens_out); });
}

}  // extern "C"
