// Host side of the C ABI (include/ppgs_amd.h): chunk planner, weight packing,
// plan cache, launch sequence of one encode, frontend tables, event timing.
#include "ppg_launch.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <limits.h>
#include <array>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace {

thread_local std::string g_error;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_error = buf;
    return code;
}

#define HIP_OK(expr)                                                              \
    do {                                                                          \
        hipError_t e_ = (expr);                                                   \
        if (e_ != hipSuccess)                                                     \
            return fail(PPG_EDEVICE, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline int round_up(int v, int a) { return (v + a - 1) / a * a; }

uint16_t host_bf16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// fp32 -> IEEE half, round to nearest even (subnormals and overflow to inf included)
uint16_t host_f16(float f) {
    return __builtin_bit_cast(uint16_t, (_Float16)f);
}
float host_f16_to_f32(uint16_t h) {
    return (float)__builtin_bit_cast(_Float16, h);
}

// ----------------------------------------------------------------------------
// Chunk planner (reference ppgs/model/transformer.py:49-64)
// ----------------------------------------------------------------------------
// A group = a contiguous run of computed windows that is executed as one
// independent pipeline on its own HIP stream (windows never interact), with
// its own slice of the workspace and token rows numbered from 0.
struct PlanGroup {
    std::vector<PpgWindow> windows;   // tok_off / vt_off relative to the group
    std::vector<int> blk_win;
    std::vector<AttnItem> items;
    int tokens = 0, vt_tokens = 0;
    size_t ws_offset = 0;
    PpgWindow* d_win = nullptr;
    int* d_blk = nullptr;
    AttnItem* d_items = nullptr;
};

struct Plan {
    std::vector<PpgWindow> all;       // every window, skipped ones with tok_off = -1
    std::vector<PpgWindow> windows;   // computed windows (valid > 0), absolute offsets
    std::vector<PlanGroup> groups;
    PpgPlanInfo info{};
};

void split_groups(Plan* plan, int ngroups, int qtile, int xcd_heads, int narrow_tiles) {
    plan->groups.clear();
    const int total = plan->info.tokens;
    size_t w = 0;
    for (int gi = 0; gi < ngroups && w < plan->windows.size(); ++gi) {
        PlanGroup grp;
        const int base_tok = plan->windows[w].tok_off, base_vt = plan->windows[w].vt_off;
        const long long target = (long long)total * (gi + 1) / ngroups;
        while (w < plan->windows.size() &&
               (grp.windows.empty() || gi == ngroups - 1 || plan->windows[w].tok_off + round_up(plan->windows[w].frames, 16) / 2 < target)) {
            PpgWindow win = plan->windows[w++];
            win.tok_off -= base_tok;
            win.vt_off -= base_vt;
            const int wi = (int)grp.windows.size();
            for (int k = 0; k < round_up(win.frames, 16) / 16; ++k) grp.blk_win.push_back(wi);
            grp.tokens = win.tok_off + round_up(win.frames, 16);
            grp.vt_tokens = win.vt_off + round_up(win.frames, 32);
            grp.windows.push_back(win);
        }
        // Query tiles.  A window that fits half a tile, or whose keys are at most half the longest window's, gets
        // tiles of half the width (attn_mixed_kernel): the latter run last, on a chip the long items no longer
        // fill, and a wave's time is its queries x the window's keys.
        int longest = 0;
        for (const PpgWindow& win : grp.windows) longest = std::max(longest, win.valid);
        for (int wi = 0; wi < (int)grp.windows.size(); ++wi) {
            const PpgWindow& win = grp.windows[wi];
            const int narrow = narrow_tiles && (narrow_tiles == 2 || 2 * win.valid <= longest || win.frames <= qtile / 2);
            for (int q0 = 0; q0 < win.frames; q0 += narrow ? qtile / 2 : qtile)
                grp.items.push_back(AttnItem{wi, q0, win.tok_off, win.vt_off, win.frames, win.valid, narrow, 0});
        }
        // Launch order = item order: longest first (keys actually visited; the
        // causal flag only shortens early query tiles, which keeps this order a
        // good proxy).  Workgroups are handed to CU slots in order, so a long item
        // dispatched late would run alone at the end of the kernel (batch 32 x
        // 1000 frames: 500/500/250-frame windows in utterance order finish at 2.0
        // long-item times, sorted at 1.5).
        std::stable_sort(grp.items.begin(), grp.items.end(), [&](const AttnItem& x, const AttnItem& y) {
            return grp.windows[x.window].valid > grp.windows[y.window].valid;
        });
        // XCD affinity: workgroup b of the 1-D attention grid runs (item b / heads, head b % heads) and goes to
        // XCD b % 8, each XCD with its own L2.  Deal the windows (in sorted order) into S = 8 / gcd(8, heads)
        // lanes and interleave the lanes, so that all query tiles of one (window, head) -- which stream the same
        // K and V^T rows -- sit S items apart and land on one XCD instead of four.
        if (xcd_heads > 0) {
            int g = xcd_heads; for (int b = 8; b; ) { const int t = g % b; g = b; b = t; }   // gcd(heads, 8)
            const int S = 8 / g;
            if (S > 1) {
                std::vector<std::vector<AttnItem>> lane(S);
                int rank = -1, last = -1;
                for (const AttnItem& it : grp.items) {
                    if (it.window != last) { ++rank; last = it.window; }
                    lane[rank % S].push_back(it);
                }
                std::vector<size_t> at(S, 0);
                size_t out = 0;
                while (out < grp.items.size())
                    for (int j = 0; j < S; ++j)
                        if (at[j] < lane[j].size()) grp.items[out++] = lane[j][at[j]++];
            }
        }
        plan->groups.push_back(std::move(grp));
    }
}

int build_plan(int chunk, int overlap, int max_positions, int batch, int frames,
               const int64_t* lengths, int legacy, int qtile, Plan* plan) {
    if (batch <= 0 || frames <= 0 || !lengths) return fail(PPG_EINVAL, "empty batch (batch=%d frames=%d)", batch, frames);
    for (int b = 0; b < batch; ++b)
        if (lengths[b] < 0 || lengths[b] > frames)
            return fail(PPG_EINVAL, "lengths[%d]=%lld outside [0, %d]", b, (long long)lengths[b], frames);
    if (legacy && frames >= max_positions)
        return fail(PPG_ELENGTH, "legacy_mode needs frames < %d, got %d", max_positions, frames);
    const int stride = chunk - 2 * overlap;
    const bool chunked = !legacy && frames > chunk;
    const int nchunks = chunked ? (frames + stride - 1) / stride : 1;
    int tok = 0, vt = 0;
    for (int b = 0; b < batch; ++b) {
        int64_t rem = lengths[b];
        for (int i = 0; i < nchunks; ++i) {
            PpgWindow w{};
            w.item = b;
            w.chunked = chunked ? 1 : 0;
            if (chunked) {
                w.start = i * stride;
                const int stop = std::min(w.start + chunk, frames + overlap);
                w.frames = stop - w.start;
                int64_t cl = std::min<int64_t>(std::max<int64_t>(rem + overlap, 0), chunk);
                if (cl == overlap) cl = 0;
                rem = std::max<int64_t>(rem - stride, 0);
                w.valid = (int)cl;
                w.keep_lo = overlap;
                w.keep_hi = std::min(chunk - overlap, w.frames);
                w.out_frame = i * stride;
            } else {
                w.start = 0;
                w.frames = frames;
                w.valid = (int)lengths[b];
                w.keep_lo = 0;
                w.keep_hi = frames;
                w.out_frame = 0;
            }
            // The reference broadcasts a (B, max(valid)) mask against (B, C, Tc):
            // positions >= valid are masked, valid never exceeds the window.
            w.valid = std::min(w.valid, w.frames);
            if (w.valid > 0) {
                w.tok_off = tok;
                w.vt_off = vt;
                tok += round_up(w.frames, 16);
                vt += round_up(w.frames, 32);
                plan->info.processed_frames += w.frames;
                plan->info.attention_pairs += (int64_t)w.frames * w.frames;
                plan->windows.push_back(w);
            } else {
                w.tok_off = -1;
                w.vt_off = -1;
                plan->info.skipped_windows++;
            }
            plan->all.push_back(w);
        }
    }
    plan->info.num_windows = (int)plan->windows.size();
    plan->info.tokens = tok;
    plan->info.vt_tokens = vt;
    return PPG_OK;
}

// ----------------------------------------------------------------------------
// Engine
// ----------------------------------------------------------------------------
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
};

// The events that fork a batch's pipelines from the caller's stream and join them into it order kernels of ONE device
// that read and write device memory: no system-scope fence (a record otherwise writes the caches back for the host and
// for peer devices to see -- every kernel already ends with the device-scope release that makes its results visible to
// the other XCDs, which is all the other pipeline's kernels need; a copy to the host or a collective behind the join
// brings its own fences).  Step 0.6750 -> 0.6700 ms, four alternations of three builds on one box
// (profiles/r6_fork_join_event_flags_ab.txt; hipEventReleaseToDevice alone: 0.6740).
constexpr unsigned kForkJoinEventFlags = hipEventDisableTiming | hipEventDisableSystemFence;
// ... and the timing events around a launch (the profiling getters) bracket the kernel, not a cache write-back for the host
constexpr unsigned kTimingEventFlags = hipEventDisableSystemFence;

struct DevLayer {
    char* wqkv; float* bqkv;
    char* wo; float* bo;
    char* w1; float* b1;
    char* w2; char* w2p; float* b2;
    char* wqkvk;          // W_qkv for the Q/K/V tail of the previous layer's FFN kernel: fp32 columns in paired order (= wqkv in bf16 mode)
    char* w1k;            // W1 for the out-proj-fused FFN: fp32 columns in paired order (= w1 in bf16 mode)
    float *g1, *e1, *g2, *e2;
    // fragment images of the feature-split layer kernel (ppg_layer32.hip), 16-bit modes with hidden 256
    char* wo_img = nullptr; char* w1_img = nullptr; char* w2_img = nullptr; char* wq_img = nullptr;
    // hi + lo fragment images of the fp16x2 mode's feature-split FFN kernel (ppg_ffn32x2.hip), hidden 256
    char* w1x_img = nullptr; char* w2x_img = nullptr; char* wox_img = nullptr; char* wqx_img = nullptr;
};

struct DevPlan {
    Plan host;
    void* buf = nullptr;
    size_t cap = 0;            // bytes of buf
    uint64_t stamp = 0;
    bool pinned = false;       // used under stream capture: a HIP graph holds its device pointers, never evicted
    hipEvent_t uploaded = nullptr;   // recorded behind the asynchronous upload of the tables
    hipStream_t upload_stream = nullptr;
    bool upload_done = false;
    std::vector<hipStream_t> users;  // every stream a launch reading the tables was queued on (encodes run on several)
    ~DevPlan() { if (uploaded) (void)hipEventDestroy(uploaded); }
};
// a device buffer whose last uses were ordered before `ready` (one event per stream that used it)
struct RetiredBuf { void* buf; size_t cap; std::vector<hipEvent_t> ready; };
// pinned host staging slot of the plan uploads: busy until `done`
struct StageSlot { void* host = nullptr; size_t cap = 0; hipEvent_t done = nullptr; bool used = false; };

struct Workspace {
    size_t xw, x, xb, qk, vt, ao, hid, part, total;
    int ffn_nt, ffn_splits;
    int vt_ld, qk_rows;
};

struct EventPair { hipEvent_t a, b; };

}  // namespace

struct PpgEngine {
    PpgConfig cfg{};
    int device = 0;
    int sz = 4;           // element bytes of the GEMM operands
    int KG = 16;          // elements per 64-byte K-group
    int Cp = 0;           // padded input channels of the gathered features
    int in_groups_per_tap = 0, in_total_groups = 0;
    int out_groups_per_tap = 0, out_total_groups = 0;
    int head_dim = 0;
    int ffn_nt = 0;       // 0 = pick per launch (choose_nt); 1..3 = forced
    int lin_nt = 0;       // same for the linear/conv kernels
    int num_cus = 256;
    bool ffn_fused = true;
    bool qkv_fused = true;   // next layer's Q/K/V projection as the tail of the fused FFN kernel (PPGS_AMD_QKV_FUSED=0: own kernel)
    int ffn_split_max = 0;   // PPGS_AMD_FFN_SPLIT_MAX: cap on the hidden splits (0: half the chunks)
    int ffn_splits_forced = 0;   // PPGS_AMD_FFN_SPLITS: this many hidden splits whatever the tile count (experiments)
    bool ffn_mixed = true;   // allow the mixed 3/3/2/2-block tiling of the fused layer kernel (PPGS_AMD_FFN_MIXED=0 disables)
    bool op_fused = true;    // attention out-projection + LN1 inside the FFN kernel (PPGS_AMD_OP_FUSED=0: own kernel)
    bool attn_xcd = true;    // attention items interleaved so that the query tiles of one (window, head) share an XCD's L2 (PPGS_AMD_ATTN_XCD=0: plain longest-first order)
    bool outconv = true;     // output convolution with LDS-resident weights where it applies (ppg_outconv.hip; PPGS_AMD_OUTCONV=0: linear_kernel)
    bool head32 = true;      // gather + input convolution + layer 0's Q/K/V in one kernel where it applies (with layer32, hidden 256, <= 96 input channels; PPGS_AMD_HEAD32=0: three launches)
    char* win_img = nullptr; // the input convolution as fragment images (ppg_head32.hip)
    int attn_narrow = 1;     // half-width query tiles for the short windows of a batch (PPGS_AMD_ATTN_NARROW=0: one width; 2: half-width tiles for every window)
    unsigned* d_overflow = nullptr;   // sticky device flag: a launch produced a non-finite logit for a valid frame (ppg_engine_nonfinite)
    int ffn32x2 = 3;         // fp16x2 mode, hidden 256, batches of >= half a chip of 96-token tiles: 3 = out-proj + LN1 + FFN + LN2 + the next layer's Q/K/V in ONE feature-split launch per layer (ppg_ffn32x2.hip), 2 = without the Q/K/V tail, 1 = the FFN block only, 0 = the token-split kernels always (PPGS_AMD_FFN32X2)
    bool split = false;      // PPG_PRECISION_FP16X2: operands as fp16 hi + lo planes in the fp32 path's byte layout (PrecX2)
    bool subtile = true;     // layer32 path, hidden 256: workgroups of two token blocks (three per 160-token tile) when whole tiles would leave two thirds of the CUs idle (PPGS_AMD_SUBTILE=0: whole tiles always)
    bool x16 = false;        // layer32 path: the residual stream between two layer kernels is stored as fp16 (X16 order) instead of fp32 -- default in the bf16 mode (PPGS_AMD_X16=0 / 1 overrides)
    bool layer32 = true;     // feature-split 32x32x16 layer kernel where it applies (16-bit modes, hidden 256, batches that fill the chip; PPGS_AMD_LAYER32=0: token-split kernels everywhere)
    bool ffn_split = true;   // split-hidden FFN for small token counts (PPGS_AMD_FFN_SPLIT=0 disables)
    int num_streams = 2;    // pipelines (HIP streams) a batch of >= 128 x CUs token rows is split into (PPGS_AMD_STREAMS;
                            // 2 = +4..6.5 % at C2 over one pipeline, the same bits there: the half-batches' kernels run beside each
                            // other, every launch on the CUs its one-per-CU workgroups occupy)
    std::vector<hipStream_t> side_streams;
    bool stream_one_pass = false; // PPGS_AMD_STREAM_ONE_PASS=1: KV-cached streams run the split-hidden FFN's reduce + LayerNorm inside the FFN launch (last workgroup of a tile by ticket) -- measured slower: its 64 rows are 4 dependent round trips on 4 waves, 43 us against 18 + 18..30
    int stream_min_rows = 128; // PPGS_AMD_STREAMS_MIN_ROWS: token rows per CU from which a batch is split into pipelines
    int stream_offset_us = 0;  // PPGS_AMD_STREAM_OFFSET_US: pipeline i of a split batch starts i * this late
    hipEvent_t ev_fork = nullptr;
    std::vector<hipEvent_t> ev_join;
    int l32_debug = 0, h32_debug = 0;         // PPGS_AMD_L32_DEBUG / PPGS_AMD_H32_DEBUG: phase-skipping switches of the timing experiments (wrong results), read once
    unsigned long long* ffn_dbg = nullptr;
    unsigned long long* head_dbg = nullptr;
    unsigned long long* attn_dbg = nullptr;  // PPGS_AMD_ATTN_TIMING (PPG_ATTN_TIMING builds)
    unsigned long long* lin_dbg = nullptr;   // PPGS_AMD_LIN_TIMING=<kernel class> (PPG_LIN_TIMING builds): stamps of layer 0
    int lin_dbg_class = -1;
    std::vector<void*> allocs;
    float* pe = nullptr;
    char* w_in = nullptr; float* b_in = nullptr;
    char* w_out = nullptr; float* b_out = nullptr;
    std::vector<DevLayer> layers;
    std::map<std::string, std::unique_ptr<DevPlan>> plans;
    uint64_t plan_stamp = 0;
    std::vector<RetiredBuf> retired;           // evicted plans' buffers: reused (or freed) once their event has passed
    StageSlot stage[4];
    int stage_next = 0;
    std::mutex mu;
    // profiling
    unsigned profiling = 0;          // bitmask of kernel classes to time
    std::vector<EventPair> events[PPG_K_COUNT];
    size_t events_used[PPG_K_COUNT] = {0};
    int profile_stride = 1;                    // time every stride-th launch of a class
    size_t launch_seq[PPG_K_COUNT] = {0};

    ~PpgEngine() {
        (void)hipSetDevice(device);
        if (d_overflow) (void)hipFree(d_overflow);
        if (head_dbg) {
            unsigned long long h[64];
            (void)hipDeviceSynchronize();
            if (hipMemcpy(h, head_dbg, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess)
                for (int w = 0; w < 4; ++w) {
                    const unsigned long long* t = h + w * 16;
                    fprintf(stderr, "head32 wave %d: gather %llu  bias+meta %llu  conv0 %llu  conv1 %llu  emit %llu  wq %llu  tail %llu | total %llu\n",
                            w, t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[6] - t[5], t[7] - t[6], t[7] - t[0]);
                }
            (void)hipFree(head_dbg);
        }
        if (ffn_dbg) {
            unsigned long long h[256];
            (void)hipDeviceSynchronize();
            if (hipMemcpy(h, ffn_dbg, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
                for (int i = 0; i < 6; ++i) {
                    const unsigned long long* t = h + 192 + i * 8;
                    if (t[0]) fprintf(stderr, "ffn qkv tail tile %d: wait %llu barrier %llu mfma %llu stores %llu dma %llu | total %llu\n", i + 4,
                                      t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[5] - t[0]);
                }
                for (int w = 0; w < 4; ++w) {
                    const unsigned long long* t = h + 128 + w * 16;
                    fprintf(stderr, "ffn prologue wave %d:", w);
                    for (int k = 1; k < 15; ++k) if (t[k]) fprintf(stderr, " [%d] %llu", k, t[k] - t[0]);
                    fprintf(stderr, "\n");
                }
                if (layer32 || split) {
                    for (int w = 0; w < 4; ++w) {
                        const unsigned long long* t = h + w * 8;
                        if (split)
                            fprintf(stderr, "ffn32x2 wave %d chunk 4: A1 %llu  wait %llu  A2 %llu  hand-over + barrier + wait %llu  B1 %llu  B2 (+ wait) %llu | total %llu\n",
                                    w, t[1] - t[0], t[6] - t[1], t[7] - t[6], t[2] - t[7], t[3] - t[2], t[5] - t[3], t[5] - t[0]);
                        else
                        fprintf(stderr, "layer32 wave %d chunk 4 (hidden 256: one stream): A blocks 0-2 %llu  A blocks 3,4 + h writes %llu  B blocks 0-2 + h writes %llu  B blocks 3,4 %llu | total %llu\n",
                                w, t[1] - t[0], t[2] - t[1], t[3] - t[2], t[5] - t[3], t[5] - t[0]);
                    }
                } else
                for (int w = 0; w < 4; ++w)
                    for (int c = 0; c < 4; ++c) {
                        const unsigned long long* t = h + (w * 4 + c) * 8;
                        fprintf(stderr, "ffn timing wave %d chunk %d: dma-issue %llu  A+pack %llu  B %llu  vmcnt %llu  barrier %llu | total %llu\n",
                                w, c + 8, t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[5] - t[0]);
                    }
            }
            (void)hipFree(ffn_dbg);
        }
        if (attn_dbg) {
            unsigned long long h[64];
            (void)hipDeviceSynchronize();
            if (hipMemcpy(h, attn_dbg, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess)
                for (int w = 0; w < 4; ++w)
                    for (int c = 0; c < 2; ++c) {
                        const unsigned long long* t = h + (w * 2 + c) * 8;
                        fprintf(stderr, "attn timing wave %d tile %d: dma-issue %llu  scores+softmax %llu  PV %llu  vmcnt %llu  barrier %llu | total %llu\n",
                                w, c + 2, t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[5] - t[0]);
                    }
            {   // per-workgroup records: start, end, valid keys, HW_ID -> PPGS_AMD_ATTN_TIMING_OUT (tools/attn_timeline.py)
                std::vector<unsigned long long> rec(4096 * 4);
                const char* path = getenv("PPGS_AMD_ATTN_TIMING_OUT");   // (PPG_ATTN_TIMING builds only: attn_dbg is null otherwise)
                if (path && hipMemcpy(rec.data(), attn_dbg + 64, rec.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
                    FILE* f = fopen(path, "wb");
                    if (f) { fwrite(rec.data(), 8, rec.size(), f); fclose(f); }
                }
            }
            (void)hipFree(attn_dbg);
        }
        if (lin_dbg) {
            const size_t n = 16 * 8192;
            std::vector<unsigned long long> h(n);
            (void)hipDeviceSynchronize();
            const char* path = getenv("PPGS_AMD_LIN_TIMING_OUT");   // (PPG_LIN_TIMING builds only)
            FILE* f = fopen(path ? path : "/tmp/lin_timing.bin", "wb");
            if (f && hipMemcpy(h.data(), lin_dbg, n * 8, hipMemcpyDeviceToHost) == hipSuccess) fwrite(h.data(), 8, n, f);
            if (f) fclose(f);
            (void)hipFree(lin_dbg);
        }
        for (auto& kv : plans) if (kv.second->buf) (void)hipFree(kv.second->buf);
        for (RetiredBuf& r : retired) { (void)hipFree(r.buf); for (hipEvent_t ev : r.ready) (void)hipEventDestroy(ev); }
        for (StageSlot& st : stage) { if (st.host) (void)hipHostFree(st.host); if (st.done) (void)hipEventDestroy(st.done); }
        for (hipStream_t st : side_streams) (void)hipStreamDestroy(st);
        for (hipEvent_t ev : ev_join) (void)hipEventDestroy(ev);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        for (void* p : allocs) (void)hipFree(p);
        for (auto& v : events) for (auto& e : v) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    }
};

namespace {

int upload(PpgEngine* e, const void* src, size_t bytes, void** dst) {
    void* p = nullptr;
    HIP_OK(hipMalloc(&p, std::max<size_t>(bytes, 16)));
    e->allocs.push_back(p);
    HIP_OK(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
    *dst = p;
    return PPG_OK;
}

int upload_f32(PpgEngine* e, const float* src, size_t n, size_t n_pad, float** dst) {
    std::vector<float> tmp(std::max(n, n_pad), 0.f);
    memcpy(tmp.data(), src, n * sizeof(float));
    return upload(e, tmp.data(), tmp.size() * sizeof(float), reinterpret_cast<void**>(dst));
}

// dst[r][c] (rows_pad x cols_pad, zero padded) = get(r, c), in the engine's element type
template <class F>
int upload_matrix(PpgEngine* e, int rows, int cols, int rows_pad, int cols_pad, F get, char** dst) {
    const size_t n = (size_t)rows_pad * cols_pad;
    if (e->split) {
        // fp16x2: every 32 elements of a row as [32 hi halves | 32 lo halves] (PrecX2, ppg_device.h)
        if (cols_pad % 32) return fail(PPG_EINVAL, "split-precision operand rows are multiples of 32 elements (%d)", cols_pad);
        std::vector<uint16_t> tmp(2 * n, 0);
        for (int r = 0; r < rows; ++r)
            for (int c = 0; c < cols; ++c) {
                const float v = get(r, c);
                const uint16_t hi = host_f16(v);
                const size_t at = ((size_t)r * cols_pad + (c / 32) * 32) * 2 + (c % 32);
                tmp[at] = hi;
                tmp[at + 32] = host_f16(v - host_f16_to_f32(hi));
            }
        return upload(e, tmp.data(), n * 4, reinterpret_cast<void**>(dst));
    }
    if (e->sz == 2) {
        std::vector<uint16_t> tmp(n, 0);
        for (int r = 0; r < rows; ++r)
            for (int c = 0; c < cols; ++c) tmp[(size_t)r * cols_pad + c] = e->cfg.precision == PPG_PRECISION_FP16 ? host_f16(get(r, c)) : host_bf16(get(r, c));
        return upload(e, tmp.data(), n * 2, reinterpret_cast<void**>(dst));
    }
    std::vector<float> tmp(n, 0.f);
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) tmp[(size_t)r * cols_pad + c] = get(r, c);
    return upload(e, tmp.data(), n * 4, reinterpret_cast<void**>(dst));
}

// Token blocks (of 16) per wave for the token-tiled kernels: a workgroup
// covers 64*nt tokens; pick the nt that minimises (rounds over the CUs) x
// (per-round cost ~ nt, larger tiles being slightly more efficient because
// each weight fragment read from LDS feeds nt MFMAs).
int choose_nt(const PpgEngine* e, int M, int max_nt) {
    if (e->ffn_nt >= 1) return std::min(e->ffn_nt, max_nt);
    static const double eff[4] = {0, 0.7, 1.0, 1.1};
    int best = 1;
    double best_cost = 1e30;
    for (int nt = 1; nt <= max_nt; ++nt) {
        const int blocks = (M + 64 * nt - 1) / (64 * nt);
        const int rounds = (blocks + e->num_cus - 1) / e->num_cus;
        const double cost = rounds * nt / eff[nt];
        if (cost < best_cost) { best_cost = cost; best = nt; }
    }
    return best;
}

// Fused-FFN tiling for a group of `M` token rows: tokens per wave (16*nt) and,
// when the tiles cannot fill the CUs, how many workgroups share one tile by
// splitting the hidden chunks between them (partial sums + a reduce/LN pass).
// A workgroup streams all of W1/W2 through its CU whatever its tile size, so
// few large tiles x several hidden splits beats many small tiles.
void choose_ffn_tiling(const PpgEngine* e, int M, int* nt_out, int* splits_out) {
    const int max_nt = (e->sz == 2 && e->cfg.hidden_channels == 256) ? 3
                       : (e->cfg.hidden_channels == 256 ? 2 : 1);
    int nt = choose_nt(e, M, max_nt);
    int splits = 1;
    const int chunks = e->cfg.ffn_channels / (32768 / (e->cfg.hidden_channels * e->sz));
    if (e->ffn_split && e->ffn_fused && e->ffn_nt == 0) {
        const int tiles_max_nt = (M + 64 * max_nt - 1) / (64 * max_nt);
        // only when the tiles would leave 7/8 of the chip idle: the partial-sum
        // round trip costs about as much as it saves above that (measured: 64 x 160
        // frames, 54 tiles: 500 us/step unsplit vs 576 us split)
        if (tiles_max_nt * 8 <= e->num_cus) {
            nt = max_nt;
            const int cap = e->ffn_split_max > 0 ? e->ffn_split_max : chunks / 2;
            while (splits * 2 <= cap && tiles_max_nt * splits * 2 <= e->num_cus) splits *= 2;
        }
    }
    // 4-byte operand modes (fp32, fp16x2) cannot hold more than 128 tokens in LDS, so a launch whose tiles
    // need a fraction over a whole number of rounds of the chip (C2: 320 tiles on 256 CUs = 2 rounds, the
    // second a quarter full) is cut into hidden splits instead: ceil(tiles * s / CUs) / s rounds of full tiles
    if (splits == 1 && e->sz == 4 && e->ffn_split && e->ffn_fused && e->ffn_nt == 0 && !e->op_fused) {
        const int tiles = (M + 64 * max_nt - 1) / (64 * max_nt);
        auto rounds = [&](int sp) { return (double)((tiles * sp + e->num_cus - 1) / e->num_cus) / sp + 0.08 * (sp > 1 ? 1 + 0.5 * sp : 0); };
        int best = 1;
        for (int sp = 2; sp <= 4 && sp <= chunks / 2; sp *= 2)
            if (rounds(sp) < rounds(best)) best = sp;
        if (best > 1) { nt = max_nt; splits = best; }
    }
    if (e->ffn_splits_forced > 0 && e->ffn_fused) { nt = max_nt; splits = e->ffn_splits_forced; }
    // mixed tiling (160-token workgroups, ppg_kernels.hip ffn_mixed_kernel): 2.5 blocks of
    // MFMA work per wave and chunk instead of nt; worth it when it saves a round or
    // shortens the one round there is (C2: 256 workgroups on 256 CUs instead of 214 larger ones)
    if (splits == 1 && e->ffn_mixed && e->ffn_fused && e->op_fused && e->ffn_nt == 0 &&
        e->sz == 2 && e->cfg.hidden_channels == 256) {
        const int blocks_nt = (M + 64 * nt - 1) / (64 * nt), blocks_mixed = (M + 159) / 160;
        const double rounds_nt = (blocks_nt + e->num_cus - 1) / e->num_cus;
        const double rounds_mixed = (blocks_mixed + e->num_cus - 1) / e->num_cus;
        if (rounds_mixed * 2.5 < rounds_nt * nt) nt = ppg::kFfnMixedTiling;
    }
    *nt_out = nt;
    *splits_out = splits;
}

// One wave that holds its stream for `ticks` of the 100 MHz real-time counter: the phase offset between the two
// pipelines of a split batch (PPGS_AMD_STREAM_OFFSET_US) -- every workgroup of a layer kernel reads its inputs at the
// launch's start and writes Q / K / V at its end, so two pipelines in phase hit the memory system together.
__global__ void phase_delay_kernel(unsigned long long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

Workspace layout(const PpgEngine* e, int tokens, int vt_tokens) {
    Workspace w{};
    const size_t M = tokens;
    const int H = e->cfg.hidden_channels;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    w.qk_rows = (int)M + 64;
    w.vt_ld = vt_tokens + 64;
    w.xw = take(M * e->Cp * e->sz);
    const size_t Ttile = layer32_tile_tokens(H);        // whole tiles of the layer32 kernel (X32 / AO32 layouts)
    const size_t Mt = (M + Ttile - 1) / Ttile * Ttile;
    w.x = take(Mt * H * 4);
    w.xb = take(e->sz == 2 ? M * H * 2 : (e->split ? M * H * 4 : 0));
    w.qk = take((size_t)w.qk_rows * 2 * H * e->sz);
    w.vt = take((size_t)H * w.vt_ld * e->sz);
    w.ao = take(Mt * H * e->sz);
    w.hid = take(e->ffn_fused ? 0 : M * e->cfg.ffn_channels * e->sz);
    choose_ffn_tiling(e, tokens, &w.ffn_nt, &w.ffn_splits);
    w.part = take(w.ffn_splits > 1 ? (size_t)w.ffn_splits * M * H * 4 : 0);
    w.total = off;
    return w;
}

// Pipelines (HIP streams) a batch is split into: windows are independent, so
// two half-batches on two streams fill the CUs one kernel's last, partly
// filled round of workgroups leaves idle (+8 % at C2).
int group_count(const PpgEngine* e, int tokens) {
    if (e->num_streams <= 1) return 1;
    return tokens >= e->stream_min_rows * e->num_cus ? e->num_streams : 1;
}

// Queries per attention workgroup of the encoder's whole-batch launches
int plan_qtile(const PpgEngine* e) { return ppg::attn_query_tile(e->head_dim); }

size_t finish_plan(const PpgEngine* e, Plan* p) {
    split_groups(p, group_count(e, p->info.tokens), plan_qtile(e), e->attn_xcd ? e->cfg.heads : 0, e->head_dim == 128 ? e->attn_narrow : 0);
    size_t off = 0;
    for (PlanGroup& grp : p->groups) {
        grp.ws_offset = off;
        off = align_up(off + layout(e, grp.tokens, grp.vt_tokens).total, 256);
    }
    p->info.workspace_bytes = off;
    return off;
}

// The cached (or new) plan of a batch shape, its tables on the device.  A miss costs no device-wide
// synchronisation: the tables go up by hipMemcpyAsync on `stream` from a pinned staging ring, device
// buffers of evicted plans are recycled behind an event.  Under stream capture (Engine.graphed)
// nothing may allocate or copy: the plan must already be cached (graphed() warms it up), and it is
// pinned from then on -- the graph holds its device pointers.
int get_plan(PpgEngine* e, int batch, int frames, const int64_t* lengths, int legacy, hipStream_t stream, DevPlan** out) {
    hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(stream, &capture);
    const bool capturing = capture != hipStreamCaptureStatusNone;
    std::string key;
    key.reserve(16 + 8 * (size_t)batch);
    const int hdr[3] = {batch, frames, legacy};
    key.append(reinterpret_cast<const char*>(hdr), sizeof(hdr));
    key.append(reinterpret_cast<const char*>(lengths), sizeof(int64_t) * (size_t)batch);
    auto it = e->plans.find(key);
    if (it != e->plans.end()) {
        DevPlan* hit = it->second.get();
        hit->stamp = ++e->plan_stamp;
        if (capturing) hit->pinned = true;
        if (std::find(hit->users.begin(), hit->users.end(), stream) == hit->users.end()) hit->users.push_back(stream);
        // the tables went up asynchronously on the stream of the first use: another stream waits for them
        if (!hit->upload_done) {
            if (hipEventQuery(hit->uploaded) == hipSuccess) hit->upload_done = true;
            else if (stream != hit->upload_stream) {
                if (capturing) HIP_OK(hipEventSynchronize(hit->uploaded));
                else HIP_OK(hipStreamWaitEvent(stream, hit->uploaded, 0));
            }
        }
        *out = hit;
        return PPG_OK;
    }
    if (capturing)
        return fail(PPG_EINVAL, "ppg_encode under stream capture needs a cached plan: run the same (batch, frames, lengths) once before capturing");
    auto dp = std::make_unique<DevPlan>();
    int rc = build_plan(e->cfg.chunk_length, e->cfg.chunk_overlap, e->cfg.max_positions, batch, frames,
                        lengths, legacy, plan_qtile(e), &dp->host);
    if (rc) return rc;
    Plan& p = dp->host;
    finish_plan(e, &p);
    // one device buffer: per group windows | blk_win | attention items
    struct Off { size_t win, blk, item; };
    std::vector<Off> offs;
    size_t total = 0;
    for (const PlanGroup& grp : p.groups) {
        Off o;
        o.win = total;
        o.blk = align_up(o.win + std::max<size_t>(grp.windows.size(), 1) * sizeof(PpgWindow), 256);
        o.item = align_up(o.blk + std::max<size_t>(grp.blk_win.size(), 1) * sizeof(int), 256);
        total = align_up(o.item + std::max<size_t>(grp.items.size(), 1) * sizeof(AttnItem), 256);
        offs.push_back(o);
    }
    total = std::max<size_t>(total, 256);
    std::vector<char> staging(total, 0);
    for (size_t gi = 0; gi < p.groups.size(); ++gi) {
        const PlanGroup& grp = p.groups[gi];
        memcpy(staging.data() + offs[gi].win, grp.windows.data(), grp.windows.size() * sizeof(PpgWindow));
        memcpy(staging.data() + offs[gi].blk, grp.blk_win.data(), grp.blk_win.size() * sizeof(int));
        memcpy(staging.data() + offs[gi].item, grp.items.data(), grp.items.size() * sizeof(AttnItem));
    }
    // bound the cache: evict the least recently used plan that no graph refers to; its buffer is retired behind
    // one event per stream that ever used it (kernels queued on ANY of them may still read the tables)
    if (e->plans.size() >= 64) {
        auto victim = e->plans.end();
        for (auto jt = e->plans.begin(); jt != e->plans.end(); ++jt)
            if (!jt->second->pinned && (victim == e->plans.end() || jt->second->stamp < victim->second->stamp)) victim = jt;
        if (victim != e->plans.end()) {
            if (victim->second->buf) {
                RetiredBuf r{victim->second->buf, victim->second->cap, {}};
                std::vector<hipStream_t> streams = victim->second->users;
                if (std::find(streams.begin(), streams.end(), stream) == streams.end()) streams.push_back(stream);
                for (hipStream_t user : streams) {
                    hipEvent_t ev = nullptr;
                    HIP_OK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
                    r.ready.push_back(ev);
                    HIP_OK(hipEventRecord(ev, user));
                }
                e->retired.push_back(std::move(r));
            }
            e->plans.erase(victim);
        }
    }
    // a retired buffer that is large enough and no longer in use, else a new one
    for (size_t i = 0; i < e->retired.size(); ++i) {
        RetiredBuf& r = e->retired[i];
        bool idle = true;
        for (hipEvent_t ev : r.ready) idle = idle && hipEventQuery(ev) == hipSuccess;
        if (!idle) continue;
        if (r.cap >= total && dp->buf == nullptr) {
            dp->buf = r.buf; dp->cap = r.cap;
        } else if (e->retired.size() > 16) {
            (void)hipFree(r.buf);                 // the pool stays small
        } else {
            continue;
        }
        for (hipEvent_t ev : r.ready) (void)hipEventDestroy(ev);
        e->retired.erase(e->retired.begin() + i);
        --i;
    }
    if (dp->buf == nullptr) {
        dp->cap = align_up(total, 4096);
        HIP_OK(hipMalloc(&dp->buf, dp->cap));
    }
    // upload through a pinned staging slot, asynchronously on the encode stream
    StageSlot& slot = e->stage[e->stage_next];
    e->stage_next = (e->stage_next + 1) % 4;
    if (slot.used) HIP_OK(hipEventSynchronize(slot.done));      // four uploads in flight at most
    if (slot.cap < total) {
        if (slot.host) (void)hipHostFree(slot.host);
        slot.cap = align_up(total * 2, 4096);
        HIP_OK(hipHostMalloc(&slot.host, slot.cap, hipHostMallocDefault));
    }
    if (!slot.done) HIP_OK(hipEventCreateWithFlags(&slot.done, hipEventDisableTiming));
    memcpy(slot.host, staging.data(), total);
    HIP_OK(hipMemcpyAsync(dp->buf, slot.host, total, hipMemcpyHostToDevice, stream));
    HIP_OK(hipEventRecord(slot.done, stream));
    slot.used = true;
    HIP_OK(hipEventCreateWithFlags(&dp->uploaded, hipEventDisableTiming));
    HIP_OK(hipEventRecord(dp->uploaded, stream));
    dp->upload_stream = stream;
    dp->users.push_back(stream);
    for (size_t gi = 0; gi < p.groups.size(); ++gi) {
        char* base = static_cast<char*>(dp->buf);
        p.groups[gi].d_win = reinterpret_cast<PpgWindow*>(base + offs[gi].win);
        p.groups[gi].d_blk = reinterpret_cast<int*>(base + offs[gi].blk);
        p.groups[gi].d_items = reinterpret_cast<AttnItem*>(base + offs[gi].item);
    }
    dp->stamp = ++e->plan_stamp;
    *out = dp.get();
    e->plans.emplace(std::move(key), std::move(dp));
    return PPG_OK;
}

struct Timed {
    PpgEngine* e; int cls; hipStream_t s; EventPair ev{}; bool on = false;
    Timed(PpgEngine* e_, int cls_, hipStream_t s_, bool live = true) : e(e_), cls(cls_), s(s_) {
        if (!live || !(e->profiling & (1u << cls))) return;
        if (e->launch_seq[cls]++ % (size_t)e->profile_stride) return;
        auto& pool = e->events[cls];
        size_t& used = e->events_used[cls];
        if (used == pool.size()) {
            EventPair p;
            if (hipEventCreateWithFlags(&p.a, kTimingEventFlags) != hipSuccess || hipEventCreateWithFlags(&p.b, kTimingEventFlags) != hipSuccess) return;
            pool.push_back(p);
        }
        ev = pool[used++];
        on = hipEventRecord(ev.a, s) == hipSuccess;
    }
    ~Timed() { if (on) (void)hipEventRecord(ev.b, s); }
};

// ----------------------------------------------------------------------------
// Frontend tables, one set per device
// ----------------------------------------------------------------------------
struct Frontend {
    bool ready = false;
    ppg::FrontendTables tb{};
    std::vector<void*> allocs;
    bool profiling = false;
    std::vector<EventPair> events;
    size_t events_used = 0;
};
std::mutex g_front_mu;
std::map<int, Frontend> g_front;

double slaney_hz_to_mel(double f) {
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp;
    const double logstep = log(6.4) / 27.0;
    return f >= min_log_hz ? min_log_mel + log(f / min_log_hz) / logstep : f / f_sp;
}
double slaney_mel_to_hz(double m) {
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp;
    const double logstep = log(6.4) / 27.0;
    return m >= min_log_mel ? min_log_hz * exp(logstep * (m - min_log_mel)) : f_sp * m;
}

// Slaney-scale, area-normalised triangular filterbank, the algorithm of
// librosa.filters.mel(sr=16000, n_fft=1024, n_mels=80) called at reference
// ppgs/preprocess/mel.py:61-64; float64 arithmetic, cast to float32.
void mel_filterbank(std::vector<float>* dense) {
    const int n_mels = 80, n_bins = 513;
    const double sr = 16000.0;
    std::vector<double> mel_f(n_mels + 2);
    const double lo = slaney_hz_to_mel(0.0), hi = slaney_hz_to_mel(sr / 2);
    for (int i = 0; i < n_mels + 2; ++i) mel_f[i] = slaney_mel_to_hz(lo + (hi - lo) * i / (n_mels + 1));
    dense->assign((size_t)n_mels * n_bins, 0.f);
    for (int i = 0; i < n_mels; ++i) {
        const double enorm = 2.0 / (mel_f[i + 2] - mel_f[i]);
        for (int k = 0; k < n_bins; ++k) {
            const double f = k * sr / 1024.0;
            const double lower = (f - mel_f[i]) / (mel_f[i + 1] - mel_f[i]);
            const double upper = (mel_f[i + 2] - f) / (mel_f[i + 2] - mel_f[i + 1]);
            const double v = std::max(0.0, std::min(lower, upper)) * enorm;
            (*dense)[(size_t)i * n_bins + k] = (float)v;
        }
    }
}

int frontend_for(int device, Frontend** out) {
    std::lock_guard<std::mutex> lock(g_front_mu);
    Frontend& f = g_front[device];
    if (!f.ready) {
        HIP_OK(hipSetDevice(device));
        std::vector<float> hann(1024);
        for (int n = 0; n < 1024; ++n) hann[n] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * n / 1024.0));
        std::vector<float2> tw(1024);
        for (int j = 0; j < 1024; ++j) {
            const double ang = -2.0 * M_PI * j / 1024.0;
            tw[j] = make_float2((float)cos(ang), (float)sin(ang));
        }
        std::vector<float> dense;
        mel_filterbank(&dense);
        // Banded filterbank image (ppg_launch.h, FrontendTables): per block of 16 filters the 32-bin
        // steps from the block's first non-zero bin (rounded down to 32) to its last.
        struct Block { int index, first, steps; };
        std::vector<Block> blocks;
        for (int mb = 0; mb < 5; ++mb) {
            int first = 513, last = -1;
            for (int m = 16 * mb; m < 16 * mb + 16; ++m)
                for (int k = 0; k < 513; ++k)
                    if (dense[(size_t)m * 513 + k] != 0.f) { first = std::min(first, k); last = std::max(last, k); }
            if (last < 0) { first = 0; last = 0; }
            first &= ~31;
            blocks.push_back({mb, first, (last - first) / 32 + 1});
        }
        // A wave runs kMelSteps steps in two segments (kMelSegment + the rest) and can finish a block only at
        // the end of a segment: longest block first, a block longer than the first segment takes a whole
        // wave, the others the smallest free segment they fit (of the wave with the fewest steps so far).
        std::stable_sort(blocks.begin(), blocks.end(), [](const Block& a, const Block& b) { return a.steps > b.steps; });
        std::vector<uint16_t> img;
        auto half_bits = [](float v) { const _Float16 h = (_Float16)v; uint16_t u; memcpy(&u, &h, 2); return u; };
        auto half_value = [](uint16_t u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; };
        auto add_fragments = [&](int mb, int first) {          // -> index of the high fragment
            const int frag = (int)(img.size() / 512);
            img.resize(img.size() + 1024, 0);
            for (int lane = 0; lane < 64 && mb >= 0; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int m = 16 * mb + (lane & 15), k = first + 8 * (lane >> 4) + j;
                    const float w = k < 513 ? dense[(size_t)m * 513 + k] * 65536.0f : 0.f;
                    const uint16_t hi = half_bits(w);
                    img[(size_t)frag * 512 + lane * 8 + j] = hi;
                    img[(size_t)(frag + 1) * 512 + lane * 8 + j] = half_bits(w - half_value(hi));
                }
            return frag;
        };
        const int zero_frag = add_fragments(-1, 0);
        const int NS = ppg::kMelSteps, seg_first[2] = {0, ppg::kMelSegment}, seg_size[2] = {ppg::kMelSegment, NS - ppg::kMelSegment};
        std::vector<int> prog(4 * NS * 4, 0);
        for (int i = 0; i < 4 * NS; ++i) { prog[i * 4] = zero_frag; prog[i * 4 + 2] = -1; }
        int used[4] = {0, 0, 0, 0};
        bool taken[4][2] = {};
        for (const Block& blk : blocks) {
            int wave = -1, seg = -1;
            for (int w = 0; w < 4; ++w) {
                if (blk.steps > seg_size[0]) {               // whole wave
                    if (!taken[w][0] && !taken[w][1] && blk.steps <= NS && wave < 0) { wave = w; seg = 2; }
                    continue;
                }
                for (int g = 0; g < 2; ++g) {
                    if (taken[w][g] || blk.steps > seg_size[g]) continue;
                    const bool better = wave < 0 || used[w] < used[wave] || (used[w] == used[wave] && w == wave && seg_size[g] < seg_size[seg]);
                    if (better) { wave = w; seg = g; }
                }
            }
            if (wave < 0 || blk.first + 32 * blk.steps > 544)
                return fail(PPG_EINVAL, "mel filter block %d: %d steps from bin %d do not fit the frontend's program", blk.index, blk.steps, blk.first);
            // the block's steps END at its segment's end (the steps before them stay zero fragments)
            const int last = seg == 2 ? NS - 1 : seg_first[seg] + seg_size[seg] - 1;
            for (int st = 0; st < blk.steps; ++st) {
                int* e = &prog[(wave * NS + last - (blk.steps - 1) + st) * 4];
                e[0] = add_fragments(blk.index, blk.first + 32 * st);
                e[1] = (blk.first + 32 * st) * 2;
            }
            prog[(wave * NS + last) * 4 + 2] = blk.index;
            if (seg == 2) taken[wave][0] = taken[wave][1] = true; else taken[wave][seg] = true;
            used[wave] += blk.steps;
        }
        auto up = [&](const void* src, size_t bytes, const void** dst) -> int {
            void* p = nullptr;
            HIP_OK(hipMalloc(&p, bytes));
            f.allocs.push_back(p);
            HIP_OK(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
            *dst = p;
            return PPG_OK;
        };
        int rc;
        if ((rc = up(hann.data(), hann.size() * 4, (const void**)&f.tb.hann))) return rc;
        if ((rc = up(tw.data(), tw.size() * 8, (const void**)&f.tb.twiddle))) return rc;
        if ((rc = up(img.data(), img.size() * 2, (const void**)&f.tb.mel_img))) return rc;
        if ((rc = up(prog.data(), prog.size() * 4, (const void**)&f.tb.mel_prog))) return rc;
        f.tb.dbg = nullptr;
#ifdef PPG_FE_TIMING
        if (getenv("PPGS_AMD_FE_TIMING")) {
            std::vector<unsigned long long> zeros(64, 0);
            if ((rc = up(zeros.data(), 512, (const void**)&f.tb.dbg))) return rc;
        }
#endif
        f.ready = true;
    }
    *out = &f;
    return PPG_OK;
}

}  // namespace

namespace ppg {
// error reporting for the other translation units of the library (ppg_resample.hip)
int fail_message(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    return fail(code, "%s", buf);
}
}  // namespace ppg

// ============================================================================
// C ABI
// ============================================================================
extern "C" {

const char* ppg_last_error(void) { return g_error.c_str(); }
int ppg_abi_version(void) { return PPG_ABI_VERSION; }

int ppg_engine_create(const PpgConfig* cfg, const PpgWeights* wts, int device, PpgEngine** out) {
    if (!cfg || !wts || !out) return fail(PPG_EINVAL, "null argument");
    const int H = cfg->hidden_channels, F = cfg->ffn_channels, L = cfg->num_layers, C = cfg->input_channels;
    if (cfg->kernel_size != 5) return fail(PPG_EINVAL, "kernel_size %d unsupported (5 only)", cfg->kernel_size);
    if (L < 0 || L > PPG_MAX_LAYERS) return fail(PPG_EINVAL, "num_layers %d outside [0,%d]", L, PPG_MAX_LAYERS);
    if (H != 256 && H != 512) return fail(PPG_EINVAL, "hidden_channels %d unsupported (256 or 512)", H);
    if (cfg->heads < 1 || H % cfg->heads) return fail(PPG_EINVAL, "heads %d does not divide hidden %d", cfg->heads, H);
    const int dh = H / cfg->heads;
    if (dh != 128 && dh != 256) return fail(PPG_EINVAL, "head dim %d unsupported (128 or 256)", dh);
    if (F % 64 || F < 64) return fail(PPG_EINVAL, "ffn_channels %d must be a multiple of 64", F);
    if (cfg->output_channels < 1 || cfg->output_channels > 48) return fail(PPG_EINVAL, "output_channels %d outside [1,48]", cfg->output_channels);
    if (C < 1) return fail(PPG_EINVAL, "input_channels %d", C);
    if (cfg->chunk_length <= 2 * cfg->chunk_overlap || cfg->chunk_length > 512)
        return fail(PPG_EINVAL, "chunk_length %d / overlap %d unsupported", cfg->chunk_length, cfg->chunk_overlap);
    if (cfg->precision != PPG_PRECISION_FP32 && cfg->precision != PPG_PRECISION_BF16 && cfg->precision != PPG_PRECISION_FP16 &&
        cfg->precision != PPG_PRECISION_FP16X2)
        return fail(PPG_EINVAL, "precision %d", cfg->precision);
    if (cfg->precision == PPG_PRECISION_FP16X2 && !((H == 256 && dh == 128) || (H == 512 && dh == 256)))
        return fail(PPG_EINVAL, "the fp16x2 mode covers hidden 256 with head dimension 128 and hidden 512 with head dimension 256 (hidden %d, head dimension %d)", H, dh);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(PPG_EDEVICE, "no HIP device: the PPG engine has no CPU path");
    if (device < 0 || device >= ndev) return fail(PPG_EDEVICE, "device %d of %d", device, ndev);
    HIP_OK(hipSetDevice(device));

    std::unique_ptr<PpgEngine> e(new PpgEngine());
    e->cfg = *cfg;
    e->device = device;
    e->split = cfg->precision == PPG_PRECISION_FP16X2;
    e->sz = (cfg->precision == PPG_PRECISION_FP32 || e->split) ? 4 : 2;
    e->KG = 64 / e->sz;
    e->head_dim = dh;
    e->Cp = round_up(C, e->split ? 32 : e->KG);      // (split operands: whole [32 hi | 32 lo] blocks)
    e->in_groups_per_tap = e->Cp / e->KG;
    e->in_total_groups = round_up(5 * e->in_groups_per_tap, 2);
    e->out_groups_per_tap = H / e->KG;
    e->out_total_groups = round_up(5 * e->out_groups_per_tap, 2);
    using ppg::env_switch;
    using ppg::env_experiment;
    e->ffn_nt = env_experiment("PPGS_AMD_FFN_NT", e->ffn_nt);
    e->lin_nt = env_experiment("PPGS_AMD_LIN_NT", e->lin_nt);
    e->ffn_fused = env_switch("PPGS_AMD_FFN_UNFUSED", !e->ffn_fused) == 0;
    e->ffn_split = env_experiment("PPGS_AMD_FFN_SPLIT", e->ffn_split) != 0;
    e->ffn32x2 = env_switch("PPGS_AMD_FFN32X2", e->ffn32x2);
    e->num_streams = std::max(1, std::min(env_switch("PPGS_AMD_STREAMS", e->num_streams), 4));
    e->stream_one_pass = env_experiment("PPGS_AMD_STREAM_ONE_PASS", e->stream_one_pass) != 0;
    e->stream_min_rows = std::max(1, env_experiment("PPGS_AMD_STREAMS_MIN_ROWS", e->stream_min_rows));
    e->stream_offset_us = std::max(0, env_experiment("PPGS_AMD_STREAM_OFFSET_US", e->stream_offset_us));
    HIP_OK(hipEventCreateWithFlags(&e->ev_fork, kForkJoinEventFlags));
    for (int i = 1; i < e->num_streams; ++i) {
        hipStream_t st;
        hipEvent_t ev;
        HIP_OK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        HIP_OK(hipEventCreateWithFlags(&ev, kForkJoinEventFlags));
        e->side_streams.push_back(st);
        e->ev_join.push_back(ev);
    }
    e->op_fused = env_switch("PPGS_AMD_OP_FUSED", e->op_fused) != 0;
    e->ffn_mixed = env_switch("PPGS_AMD_FFN_MIXED", e->ffn_mixed) != 0;
    e->ffn_split_max = env_experiment("PPGS_AMD_FFN_SPLIT_MAX", e->ffn_split_max);
    e->ffn_splits_forced = env_experiment("PPGS_AMD_FFN_SPLITS", e->ffn_splits_forced);
    e->qkv_fused = env_switch("PPGS_AMD_QKV_FUSED", e->qkv_fused) != 0;
    e->layer32 = env_switch("PPGS_AMD_LAYER32", e->layer32) != 0;
    e->attn_xcd = env_experiment("PPGS_AMD_ATTN_XCD", e->attn_xcd) != 0;
    e->attn_narrow = std::max(0, std::min(env_switch("PPGS_AMD_ATTN_NARROW", e->attn_narrow), 2));
    e->head32 = env_switch("PPGS_AMD_HEAD32", e->head32) != 0;
    e->subtile = env_switch("PPGS_AMD_SUBTILE", e->subtile) != 0;
    HIP_OK(hipMalloc(reinterpret_cast<void**>(&e->d_overflow), 256));
    HIP_OK(hipMemset(e->d_overflow, 0, 256));
    e->x16 = env_experiment("PPGS_AMD_X16", cfg->precision == PPG_PRECISION_BF16) != 0;
    e->outconv = env_switch("PPGS_AMD_OUTCONV", e->outconv) != 0;
#ifdef PPG_DEBUG_MODES
    e->l32_debug = env_switch("PPGS_AMD_L32_DEBUG", 0);
    e->h32_debug = env_switch("PPGS_AMD_H32_DEBUG", 0);
#endif
    if (e->split) {
        // the unfused launch sequence: Q/K/V, attention, out-projection + LayerNorm, FFN as one launch each
        e->op_fused = false; e->qkv_fused = false; e->ffn_mixed = false; e->ffn_fused = true;
        // hidden 512: the FFN as two GEMMs through a [tokens][ffn] buffer of [32 hi | 32 lo] rows (a chunk of the fused
        // kernel cannot hold a 32-wide hidden group of both weight tiles: ppg_kernels.hip, launch_linear_x2_nt)
        if (H == 512) e->ffn_fused = false;
    }
    if (!e->layer32 || H != 256 || e->Cp != 96 || !e->qkv_fused) e->head32 = false;
    if (e->sz != 2 || (H != 256 && H != 512) || F % 128 || F > 6656) e->layer32 = false;
#ifdef PPG_LIN_TIMING
    if (const char* v = getenv("PPGS_AMD_LIN_TIMING")) {
        e->lin_dbg_class = atoi(v);
        HIP_OK(hipMalloc(reinterpret_cast<void**>(&e->lin_dbg), 16 * 8192 * 8));
        HIP_OK(hipMemset(e->lin_dbg, 0, 16 * 8192 * 8));
    }
#endif
#ifdef PPG_ATTN_TIMING
    if (getenv("PPGS_AMD_ATTN_TIMING")) {
        HIP_OK(hipMalloc(reinterpret_cast<void**>(&e->attn_dbg), 512 + 4096 * 32));
        HIP_OK(hipMemset(e->attn_dbg, 0, 512 + 4096 * 32));
    }
#endif
#ifdef PPG_H32_TIMING
    if (getenv("PPGS_AMD_H32_TIMING")) {
        HIP_OK(hipMalloc(reinterpret_cast<void**>(&e->head_dbg), 512));
        HIP_OK(hipMemset(e->head_dbg, 0, 512));
    }
#endif
#ifdef PPG_FFN_TIMING
    if (getenv("PPGS_AMD_FFN_TIMING")) {
        HIP_OK(hipMalloc(reinterpret_cast<void**>(&e->ffn_dbg), 2048));
        HIP_OK(hipMemset(e->ffn_dbg, 0, 2048));
    }
#endif
    if (e->ffn_nt < 0 || e->ffn_nt > 3) e->ffn_nt = 0;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
            e->num_cus = prop.multiProcessorCount;
    }

    int rc;
    PpgEngine* E = e.get();
    if ((rc = upload_f32(E, wts->position_encoding, (size_t)cfg->max_positions * H, 0, &e->pe))) return rc;
    {   // in-conv  W'[h][tap*Cp + c] = w[h][c][tap]
        const float* w = wts->input_weight;
        const int Cp = e->Cp;
        rc = upload_matrix(E, H, 5 * Cp, H, e->in_total_groups * e->KG,
                           [&](int row, int k) { const int h = pair_row(row), tap = k / Cp, c = k % Cp; return c < C ? w[((size_t)h * C + c) * 5 + tap] : 0.f; },
                           &e->w_in);
        if (rc) return rc;
        if ((rc = upload_f32(E, wts->input_bias, H, 0, &e->b_in))) return rc;
    }
    if (e->head32) {
        // the same convolution as A fragments for ppg_head32.hip: [wave][rb][K-step = tap x 16-channel block],
        // lane (row l & 31 of the block in accumulator order phi, channels 8 (l >> 5) .. + 7), one zero fragment behind
        const float* w = wts->input_weight;
        const int RB = H / 128, KSI = 5 * e->Cp / 16, frags = 4 * RB * KSI;
        auto phi = [](int rho) { return 16 * ((rho >> 2) & 1) + 4 * (rho >> 3) + (rho & 3); };
        rc = upload_matrix(E, (frags + 1) * 64, 8, (frags + 1) * 64, 8,
                           [&](int r, int j) {
                               const int f = r >> 6, ln = r & 63;
                               if (f >= frags) return 0.f;
                               const int ks = f % KSI, rb = (f / KSI) % RB, wv = f / (KSI * RB);
                               const int tap = ks / (e->Cp / 16), c = 16 * (ks % (e->Cp / 16)) + 8 * (ln >> 5) + j;
                               const int h = 32 * RB * wv + 32 * rb + phi(ln & 31);
                               return c < C ? w[((size_t)h * C + c) * 5 + tap] : 0.f;
                           },
                           &e->win_img);
        if (rc) return rc;
    }
    {   // out-conv W'[n][tap*H + c] = w[n][c][tap], rows padded to 48
        const float* w = wts->output_weight;
        rc = upload_matrix(E, cfg->output_channels, 5 * H, 48, e->out_total_groups * e->KG,
                           [&](int n, int k) { const int tap = k / H, c = k % H; return w[((size_t)n * H + c) * 5 + tap]; },
                           &e->w_out);
        if (rc) return rc;
        if ((rc = upload_f32(E, wts->output_bias, cfg->output_channels, 48, &e->b_out))) return rc;
    }
    e->layers.resize(L);
    // The attention kernel takes its Q rows already multiplied by log2(e) / sqrt(head_dim) (attn_body's softmax
    // is a bare exp2 of the score accumulator): the factor goes into the Q rows of W_qkv and b_qkv here, before
    // the weights are rounded to the operand type.
    const float qscale = (float)(1.4426950408889634 / sqrt((double)dh));
    std::vector<float> in_w((size_t)3 * H * H), in_b((size_t)3 * H);
    for (int l = 0; l < L; ++l) {
        DevLayer& d = e->layers[l];
        for (size_t i = 0; i < in_w.size(); ++i) in_w[i] = wts->in_proj_weight[l][i] * (i < (size_t)H * H ? qscale : 1.0f);
        for (int i = 0; i < 3 * H; ++i) in_b[i] = wts->in_proj_bias[l][i] * (i < H ? qscale : 1.0f);
        auto plain = [&](const float* w, int rows, int cols, char** dst) {
            return upload_matrix(E, rows, cols, rows, cols, [&](int r, int c) { return w[(size_t)r * cols + c]; }, dst);
        };
        // output features in paired-block order (pair_row, ppg_device.h): tile row r computes feature pair_row(r)
        auto paired = [&](const float* w, int rows, int cols, char** dst) {
            return upload_matrix(E, rows, cols, rows, cols, [&](int r, int c) { return w[(size_t)pair_row(r) * cols + c]; }, dst);
        };
        if ((rc = paired(in_w.data(), 3 * H, H, &d.wqkv))) return rc;
        if ((rc = paired(wts->out_proj_weight[l], H, H, &d.wo))) return rc;
        if ((rc = plain(wts->linear1_weight[l], F, H, &d.w1))) return rc;
        if ((rc = paired(wts->linear2_weight[l], H, F, &d.w2))) return rc;
        d.wqkvk = d.wqkv;
        if (e->sz == 4 && !e->split) {
            const float* w = in_w.data();
            rc = upload_matrix(E, 3 * H, H, 3 * H, H, [&](int r, int c) { return w[(size_t)pair_row(r) * H + pair_row(c)]; }, &d.wqkvk);
            if (rc) return rc;
        }
        d.w1k = d.w1;
        if (e->sz == 4 && !e->split) {   // the fused prologue hands LN1's fp32 accumulators to phase A in paired K order
            const float* w = wts->linear1_weight[l];
            rc = upload_matrix(E, F, H, F, H, [&](int r, int c) { return w[(size_t)r * H + pair_row(c)]; }, &d.w1k);
            if (rc) return rc;
        }
        {   // pack_w2: k-slot order of the fused FFN's phase-B fragments.
            // bf16: inside each 32-wide hidden group, slot 8g + 4e + r holds
            // hidden 16e + 4g + r (the two phase-A accumulators e of lane
            // group g, concatenated).  fp32: natural order.
            const float* w = wts->linear2_weight[l];
            const bool bf = e->sz == 2 || e->split;
            rc = upload_matrix(E, H, F, H, F,
                               [&](int row, int c) {
                                   const int r = pair_row(row);
                                   if (!bf) return w[(size_t)r * F + c];
                                   const int grp = c / 32, s = c % 32, g = s / 8, ee = (s % 8) / 4, rr = s % 4;
                                   return w[(size_t)r * F + grp * 32 + 16 * ee + 4 * g + rr];
                               },
                               &d.w2p);
            if (rc) return rc;
        }
        if (e->layer32) {
            // Fragment images (ppg_layer32.hip): fragment = 64 lanes x 8 elements, lane l = (row l & 31 of
            // the 32-row block, K slots 8 (l >> 5) .. +7 of the 16-wide K-step).  Wave w owns features
            // 32 RB w .. of a H-wide result (RB = H / 128).  Output rows of a block sit in the order phi
            // the accumulator layout hands a lane 16 consecutive features in; where a GEMM's K is the
            // previous accumulator (x1, h, x2) the K order follows it.
            const int RB = H / 128, KS = H / 16;
            auto phi = [](int rho) { return 16 * ((rho >> 2) & 1) + 4 * (rho >> 3) + (rho & 3); };
            auto panel_k = [](int ks, int ln, int j) { return 32 * (ks >> 1) + 16 * (ln >> 5) + 8 * (ks & 1) + j; };   // feature in slot j of K-step ks
            auto image = [&](int frags, auto get, char** dst) {          // get(frag, lane, j)
                return upload_matrix(E, frags * 64, 8, frags * 64, 8,
                                     [&](int r, int j) { return get(r >> 6, r & 63, j); }, dst);
            };
            const float* wo = wts->out_proj_weight[l];
            rc = image(4 * RB * KS, [&](int f, int ln, int j) {          // [w][rb][ks], natural K (the attention output's)
                const int ks = f % KS, rb = (f / KS) % RB, w = f / (KS * RB);
                return wo[(size_t)(32 * RB * w + 32 * rb + phi(ln & 31)) * H + 16 * ks + 8 * (ln >> 5) + j];
            }, &d.wo_img);
            if (rc) return rc;
            const float* w1 = wts->linear1_weight[l];
            rc = image(F / 128 * 4 * KS, [&](int f, int ln, int j) {     // [chunk][w][ks]
                const int ks = f % KS, w = (f / KS) & 3, ch = f / (KS * 4);
                return w1[(size_t)(ch * 128 + 32 * w + (ln & 31)) * H + panel_k(ks, ln, j)];
            }, &d.w1_img);
            if (rc) return rc;
            const float* w2 = wts->linear2_weight[l];
            rc = image(F / 128 * 4 * RB * 8, [&](int f, int ln, int j) { // [chunk][w][rb][ks8], K = the chunk's h in accumulator order
                const int ks = f & 7, rb = (f >> 3) % RB, w = (f / (8 * RB)) & 3, ch = f / (8 * RB * 4);
                return w2[(size_t)(32 * RB * w + 32 * rb + phi(ln & 31)) * F + ch * 128 + 32 * (ks >> 1) + 16 * (ks & 1) +
                          8 * (j >> 2) + 4 * (ln >> 5) + (j & 3)];
            }, &d.w2_img);
            if (rc) return rc;
            // W_qkv of THIS layer (the previous layer's kernel runs it as its tail): per wave 3 RB steps of
            // 32 rows: Q rows 32 RB w + 32 rb + phi, K likewise, V rows in attn_kernel's tile order
            // (V^T row r = natural feature pair_row(r)); K order = the x2 panel's (as W1)
            const float* wq = in_w.data();
            rc = image(4 * 3 * RB * KS, [&](int f, int ln, int j) {      // [w][step][ks]
                const int ks = f % KS, st = (f / KS) % (3 * RB), w = f / (KS * 3 * RB);
                const int rb = st % RB, kind = st / RB;
                const int row = kind < 2 ? kind * H + 32 * RB * w + 32 * rb + phi(ln & 31)
                                         : 2 * H + pair_row(32 * RB * w + 32 * rb + (ln & 31));
                return wq[(size_t)row * H + panel_k(ks, ln, j)];
            }, &d.wq_img);
            if (rc) return rc;
        }
        if (e->split && e->ffn32x2 && ppg::ffn32x2_supported(H, F)) {      // (an engine outside it -- F > 3328 -- keeps the token-split kernels)
            // ppg_ffn32x2.hip: every A fragment twice, as the fp16 hi plane and the fp16 lo plane of the fp32 weight.
            // W1: [chunk][wave][plane][ks], rows natural, K natural (the panel is loaded in natural order);
            // W2: [chunk][wave][plane][rb][ks8], rows in the order phi, K = the chunk's h in accumulator order
            auto phi = [](int rho) { return 16 * ((rho >> 2) & 1) + 4 * (rho >> 3) + (rho & 3); };
            auto image2 = [&](int groups, auto get, char** dst) {      // get(group, fragment 0..15, lane, j) -> fp32 weight
                std::vector<uint16_t> tmp((size_t)groups * 32 * 512);
                for (int g = 0; g < groups; ++g)
                    for (int f = 0; f < 16; ++f)
                        for (int ln = 0; ln < 64; ++ln)
                            for (int j = 0; j < 8; ++j) {
                                const float v = get(g, f, ln, j);
                                const uint16_t hi = host_f16(v);
                                tmp[((size_t)g * 32 + f) * 512 + ln * 8 + j] = hi;
                                tmp[((size_t)g * 32 + 16 + f) * 512 + ln * 8 + j] = host_f16(v - host_f16_to_f32(hi));
                            }
                return upload(E, tmp.data(), tmp.size() * 2, reinterpret_cast<void**>(dst));
            };
            const float* w1 = wts->linear1_weight[l];
            rc = image2(F / 128 * 4, [&](int g, int ks, int ln, int j) {
                const int w = g & 3, ch = g >> 2;
                return w1[(size_t)(ch * 128 + 32 * w + (ln & 31)) * H + 16 * ks + 8 * (ln >> 5) + j];
            }, &d.w1x_img);
            if (rc) return rc;
            const float* w2 = wts->linear2_weight[l];
            rc = image2(F / 128 * 4, [&](int g, int f, int ln, int j) {
                const int w = g & 3, ch = g >> 2, rb = f >> 3, ks = f & 7;
                return w2[(size_t)(64 * w + 32 * rb + phi(ln & 31)) * F + ch * 128 + 32 * (ks >> 1) + 16 * (ks & 1) +
                          8 * (j >> 2) + 4 * (ln >> 5) + (j & 3)];
            }, &d.w2x_img);
            if (rc) return rc;
            // Wo: [wave][K half][plane][rb][ks8], K natural (the attention output's)
            const float* wo = wts->out_proj_weight[l];
            rc = image2(4 * 2, [&](int g, int f, int ln, int j) {
                const int kh = g & 1, w = g >> 1, rb = f >> 3, ks = f & 7;
                return wo[(size_t)(64 * w + 32 * rb + phi(ln & 31)) * H + 16 * (8 * kh + ks) + 8 * (ln >> 5) + j];
            }, &d.wox_img);
            if (rc) return rc;
            // W_qkv of THIS layer (the previous layer's launch runs it as its tail): [wave][kind][K half][plane][rb][ks8];
            // Q and K rows in the order phi, V rows in attn_kernel's tile order (V^T row r = natural feature pair_row(r))
            const float* wq = in_w.data();
            rc = image2(4 * 3 * 2, [&](int g, int f, int ln, int j) {
                const int kh = g & 1, kind = (g >> 1) % 3, w = g / 6, rb = f >> 3, ks = f & 7;
                const int row = kind < 2 ? kind * H + 64 * w + 32 * rb + phi(ln & 31) : 2 * H + pair_row(64 * w + 32 * rb + (ln & 31));
                return wq[(size_t)row * H + 16 * (8 * kh + ks) + 8 * (ln >> 5) + j];
            }, &d.wqx_img);
            if (rc) return rc;
        }
        if ((rc = upload_f32(E, in_b.data(), 3 * H, 0, &d.bqkv))) return rc;
        if ((rc = upload_f32(E, wts->out_proj_bias[l], H, 0, &d.bo))) return rc;
        if ((rc = upload_f32(E, wts->linear1_bias[l], F, 0, &d.b1))) return rc;
        if ((rc = upload_f32(E, wts->linear2_bias[l], H, 0, &d.b2))) return rc;
        if ((rc = upload_f32(E, wts->norm1_weight[l], H, 0, &d.g1))) return rc;
        if ((rc = upload_f32(E, wts->norm1_bias[l], H, 0, &d.e1))) return rc;
        if ((rc = upload_f32(E, wts->norm2_weight[l], H, 0, &d.g2))) return rc;
        if ((rc = upload_f32(E, wts->norm2_bias[l], H, 0, &d.e2))) return rc;
    }
    *out = e.release();
    return PPG_OK;
}

void ppg_engine_destroy(PpgEngine* engine) { delete engine; }

int ppg_plan_windows(const PpgEngine* engine, int batch, int frames, const int64_t* lengths,
                     int legacy_mode, PpgWindow* windows, int max_windows, PpgPlanInfo* info) {
    Plan plan;
    const int chunk = engine ? engine->cfg.chunk_length : 500;
    const int overlap = engine ? engine->cfg.chunk_overlap : 50;
    const int maxpos = engine ? engine->cfg.max_positions : 5000;
    const int qt = engine ? plan_qtile(engine) : ppg::attn_query_tile(128);
    int rc = build_plan(chunk, overlap, maxpos, batch, frames, lengths, legacy_mode, qt, &plan);
    if (rc) return rc;
    if (engine) finish_plan(engine, &plan);
    if (info) *info = plan.info;
    const int n = (int)plan.all.size();
    if (windows) for (int i = 0; i < n && i < max_windows; ++i) windows[i] = plan.all[i];
    return n;
}

int ppg_plan_attention_items(const PpgEngine* engine, int batch, int frames, const int64_t* lengths,
                             int legacy_mode, int heads, PpgAttentionItem* items, int max_items) {
    Plan plan;
    const int chunk = engine ? engine->cfg.chunk_length : 500;
    const int overlap = engine ? engine->cfg.chunk_overlap : 50;
    const int maxpos = engine ? engine->cfg.max_positions : 5000;
    const int head_dim = engine ? engine->head_dim : 128;
    const int qt = engine ? plan_qtile(engine) : ppg::attn_query_tile(head_dim);
    if (engine) heads = engine->cfg.heads;
    if (heads <= 0) return fail(PPG_EINVAL, "heads=%d", heads);
    int rc = build_plan(chunk, overlap, maxpos, batch, frames, lengths, legacy_mode, qt, &plan);
    if (rc) return rc;
    if (engine) finish_plan(engine, &plan);
    else split_groups(&plan, 1, qt, heads, head_dim == 128);
    int n = 0, wbase = 0;
    for (const PlanGroup& grp : plan.groups) {
        for (const AttnItem& it : grp.items) {
            if (items && n < max_items)
                items[n] = PpgAttentionItem{wbase + it.window, it.q0, it.narrow ? qt / 2 : qt, it.frames, it.valid, it.narrow};
            ++n;
        }
        wbase += (int)grp.windows.size();
    }
    return n;
}

int ppg_workspace_bytes(const PpgEngine* engine, int batch, int frames, const int64_t* lengths,
                        int legacy_mode, size_t* bytes) {
    if (!engine || !bytes) return fail(PPG_EINVAL, "null argument");
    PpgPlanInfo info;
    int rc = ppg_plan_windows(engine, batch, frames, lengths, legacy_mode, nullptr, 0, &info);
    if (rc < 0) return rc;
    *bytes = info.workspace_bytes;
    return PPG_OK;
}

int ppg_encode(PpgEngine* e, const void* features, int feature_dtype, const int64_t* lengths,
               int batch, int frames, int softmax, int legacy_mode, float* out,
               void* workspace, size_t workspace_bytes, void* stream_) {
    if (!e || !features || !lengths || !out) return fail(PPG_EINVAL, "null argument");
    if (feature_dtype != PPG_DTYPE_F16 && feature_dtype != PPG_DTYPE_F32) return fail(PPG_EINVAL, "feature dtype %d", feature_dtype);
    std::lock_guard<std::mutex> lock(e->mu);
    HIP_OK(hipSetDevice(e->device));
    hipStream_t s = static_cast<hipStream_t>(stream_);
    DevPlan* dp = nullptr;
    int rc = get_plan(e, batch, frames, lengths, legacy_mode, s, &dp);
    if (rc) return rc;
    const Plan& plan = dp->host;
    const PpgConfig& c = e->cfg;
    const int H = c.hidden_channels, F = c.ffn_channels, M = plan.info.tokens;
    const int prec = c.precision;
    const size_t out_elems = (size_t)batch * c.output_channels * frames;

    // exhausted windows (all keys masked): reference output there is
    // logits 0 -> uniform posteriors; nothing to compute.
    if (plan.info.skipped_windows > 0 || M == 0) {
        hipError_t he = ppg::launch_fill(out, out_elems, softmax ? 1.0f / c.output_channels : 0.f, s);
        if (he != hipSuccess) return fail(PPG_EDEVICE, "fill: %s", hipGetErrorString(he));
    }
    if (M == 0) return PPG_OK;

    if (!workspace || workspace_bytes < plan.info.workspace_bytes)
        return fail(PPG_EWORKSPACE, "workspace %zu bytes < required %zu", workspace_bytes, plan.info.workspace_bytes);
    if (reinterpret_cast<uintptr_t>(workspace) % 256) return fail(PPG_EINVAL, "workspace not 256-byte aligned");

    // (a launch outside the segment a call of run_group is asked for is skipped: see `live` there)
#define LAUNCH_OK(expr, what)                                                        \
    do {                                                                             \
        if (live) {                                                                  \
            hipError_t he_ = (expr);                                                 \
            if (he_ != hipSuccess) return fail(PPG_EDEVICE, "%s: %s", what, hipGetErrorString(he_)); \
        }                                                                            \
    } while (0)

    // one independent pipeline per plan group, each on its own stream.  The launch sequence of a group comes in SEGMENTS
    // -- 0: the head (gather, input convolution, layer 0's Q/K/V), 1 + l: layer l, 1 + layers: the output convolution --
    // and a call enqueues one of them (seg < 0: all): the groups' segments are enqueued alternately below.
    const int nseg = c.num_layers + 2;
    auto run_group = [&](const PlanGroup& grp, hipStream_t s, const int seg) -> int {
    bool live = seg < 0 || seg == 0;
    const int M = grp.tokens;
    const Workspace ws = layout(e, grp.tokens, grp.vt_tokens);
    char* base = static_cast<char*>(workspace) + grp.ws_offset;
    char* xw = base + ws.xw;
    float* X = reinterpret_cast<float*>(base + ws.x);
    char* Xb = (e->sz == 2 || e->split) ? base + ws.xb : nullptr;
    char* qk = base + ws.qk;
    char* vt = base + ws.vt;
    char* ao = base + ws.ao;
    char* hid = base + ws.hid;
    const char* act_x = (e->sz == 2 || e->split) ? Xb : reinterpret_cast<const char*>(X);

    const int nt = choose_nt(e, M, e->sz == 2 ? 3 : 2);     // linear / conv kernels
    // linear / conv kernels: measured best at C2 (two 256-register workgroups
    // per CU): 32-token waves for the wide projections, 16-token waves where
    // the epilogue dominates (LayerNorm, softmax scatter)
    const bool forced = e->lin_nt >= 1 && e->lin_nt <= 3;
    const int lnt = forced ? e->lin_nt : std::min(nt, 2);
    const int lnt_ln = forced ? e->lin_nt : 1;

    const bool use32 = e->layer32 && e->ffn_fused && ws.ffn_splits == 1;
    // (one 160-token tile per workgroup: below half a chip of tiles the three launches, with their smaller workgroups, are as fast)
    const int tiles32 = (M + ppg::layer32_tokens(H) - 1) / ppg::layer32_tokens(H);
    // sub-tile workgroups (two token blocks, three per tile) when whole tiles would leave two thirds of the CUs idle
    const bool sub32 = use32 && e->subtile && H == 256 && (F / 128) % 2 == 0 && 3 * tiles32 <= e->num_cus;
    const bool head = use32 && e->head32 && (2 * tiles32 >= e->num_cus || sub32);
    if (head) {
        Timed t(e, PPG_K_INCONV, s, live);
        Head32Args a{};
        a.feats = features; a.dtype = feature_dtype; a.C = c.input_channels; a.T = frames; a.overlap = c.chunk_overlap;
        a.win_img = e->win_img; a.b_in = e->b_in; a.pe = e->pe; a.X = X;
        a.wq_img = e->layers[0].wq_img; a.bq = e->layers[0].bqkv; a.qk_out = qk; a.vt_out = vt; a.vt_ld = ws.vt_ld;
        a.blk_win = grp.d_blk; a.win = grp.d_win; a.M = M; a.H = H;
        a.tiles = (M + ppg::layer32_tokens(H) - 1) / ppg::layer32_tokens(H);
        a.nwin = (int)grp.windows.size(); a.vt_rows = H; a.vt_tokens = grp.vt_tokens;
        a.qk_slack = qk + (size_t)M * 2 * H * e->sz; a.qk_slack_bytes = (int)(64 * 2 * H * e->sz);
        a.debug_mode = e->h32_debug;
        a.x_half = e->x16;
        a.sub_tiles = sub32;
        a.dbg = e->head_dbg;
        LAUNCH_OK(ppg::launch_head32(prec, a, s), "head32");
    }
    if (!head) {
        Timed t(e, PPG_K_GATHER, s, live);
        GatherArgs g{};
        g.feats = features; g.dtype = feature_dtype; g.C = c.input_channels; g.T = frames;
        g.overlap = c.chunk_overlap; g.xw = xw; g.Cp = e->Cp;
        g.blk_win = grp.d_blk; g.win = grp.d_win; g.M = M;
        // V^T padding columns and the K rows past the last token are read (masked)
        // by the attention tiles: the same launch keeps them finite
        g.vt = vt; g.vt_ld = ws.vt_ld; g.vt_rows = H; g.vt_tokens = grp.vt_tokens; g.nwin = (int)grp.windows.size();
        g.qk_slack = qk + (size_t)M * 2 * H * e->sz; g.qk_slack_bytes = (int)(64 * 2 * H * e->sz);
        LAUNCH_OK(ppg::launch_gather(prec, g, s), "gather");
    }
    auto base_args = [&]() {
        LinearArgs a{};
        a.blk_win = grp.d_blk; a.win = grp.d_win; a.M = M; a.H = H;
        a.X = X; a.Xb = Xb; a.v_start = INT_MAX; a.taps = 1;
        return a;
    };
    if (!head) {
        Timed t(e, PPG_K_INCONV, s, live);
        LinearArgs a = base_args();
        a.x_tiled = use32 ? (e->x16 ? 2 : 1) : 0;
        a.act = xw; a.lda_bytes = e->Cp * e->sz; a.taps = 5;
        a.groups_per_tap = e->in_groups_per_tap; a.real_groups = 5 * e->in_groups_per_tap;
        a.total_groups = e->in_total_groups;
        a.W = e->w_in; a.bias = e->b_in; a.N = H; a.pe = e->pe;
        if (e->lin_dbg_class == PPG_K_INCONV) a.dbg = e->lin_dbg;
        LAUNCH_OK(ppg::launch_linear(prec, EPI_INCONV, 16, lnt, a, H / 256, s), "in-conv");
    }
    const int hg = H / e->KG;   // K-groups of a hidden-wide row
    bool qkv_done = head;    // this layer's Q/K/V came out of the previous layer's FFN kernel (layer 0's: out of the head kernel)
    for (int l = 0; l < c.num_layers; ++l) {
        const DevLayer& d = e->layers[l];
        live = seg < 0 || seg == 1 + l;
        if (!qkv_done) {
            Timed t(e, PPG_K_QKV, s, live);
            LinearArgs a = base_args();
            a.act = act_x; a.lda_bytes = H * e->sz;
            a.groups_per_tap = hg; a.real_groups = hg; a.total_groups = hg;
            a.W = d.wqkv; a.bias = d.bqkv; a.N = 3 * H;
            a.out_rows = qk; a.out_ld = 2 * H; a.vt = vt; a.vt_ld = ws.vt_ld; a.v_start = 2 * H;
            if (l == 0 && e->lin_dbg_class == PPG_K_QKV) a.dbg = e->lin_dbg;
            LAUNCH_OK(ppg::launch_linear(prec, EPI_QKV, 16, lnt, a, 3 * H / 256, s), "qkv");
        }
        {
            Timed t(e, PPG_K_ATTENTION, s, live);
            AttnArgs a{};
            a.qk = qk; a.qk_ld_bytes = 2 * H * e->sz; a.vt = vt; a.vt_ld_bytes = ws.vt_ld * e->sz;
            a.ao = ao; a.H = H; a.causal = c.is_causal;
            a.items = grp.d_items; a.win = grp.d_win; a.M = M; a.ao_tiled = use32; a.heads = c.heads;
            a.dbg = l == 0 ? e->attn_dbg : nullptr;
            LAUNCH_OK(ppg::launch_attn(prec, a, (int)grp.items.size(), c.heads, e->head_dim, s), "attention");
        }
        if (use32) {
            Timed t(e, PPG_K_FFN, s, live);
            Layer32Args a{};
            a.ao = ao; a.wo_img = d.wo_img; a.w1_img = d.w1_img; a.w2_img = d.w2_img;
            a.bo = d.bo; a.g1 = d.g1; a.e1 = d.e1; a.b1 = d.b1; a.b2 = d.b2; a.g2 = d.g2; a.e2 = d.e2;
            a.X = X; a.Xb = Xb; a.M = M; a.F = F; a.H = H; a.dbg = l == 0 ? e->ffn_dbg : nullptr;
            a.debug_mode = e->l32_debug;
            a.x_half = e->x16;
            a.sub_tiles = sub32;
                qkv_done = e->qkv_fused && l + 1 < c.num_layers;
            a.write_x = l + 1 < c.num_layers;
            if (qkv_done) {
                const DevLayer& nx = e->layers[l + 1];
                a.wq_img = nx.wq_img; a.bq = nx.bqkv; a.qk_out = qk; a.vt_out = vt; a.vt_ld = ws.vt_ld;
                a.blk_win = grp.d_blk; a.win = grp.d_win;
                a.Xb = nullptr;              // nobody reads the 16-bit copy: x2 goes straight into the tail
            }
            LAUNCH_OK(ppg::launch_layer32(prec, a, s), "layer32");
            continue;
        }
        const bool fuse_op = e->ffn_fused && e->op_fused && ws.ffn_splits == 1;
        const bool x2_layer = e->split && e->ffn32x2 > 0 && d.w1x_img && 2 * ((M + ppg::ffn32x2_tokens() - 1) / ppg::ffn32x2_tokens()) >= e->num_cus;
        if (!fuse_op && !(x2_layer && e->ffn32x2 >= 2)) {
            Timed t(e, PPG_K_OUTPROJ_LN, s, live);
            LinearArgs a = base_args();
            a.act = ao; a.lda_bytes = H * e->sz;
            a.groups_per_tap = hg; a.real_groups = hg; a.total_groups = hg;
            a.W = d.wo; a.bias = d.bo; a.N = H; a.gamma = d.g1; a.beta = d.e1;
            if (l == 0 && e->lin_dbg_class == PPG_K_OUTPROJ_LN) a.dbg = e->lin_dbg;
            LAUNCH_OK(ppg::launch_linear(prec, EPI_RESLN, H / 16, lnt_ln, a, 1, s), "out-proj+LN");
        }
        {
            Timed t(e, PPG_K_FFN, s, live);
            if (x2_layer) {
                Ffn32X2Args a{};
                a.xb = Xb; a.X = X; a.xb_out = Xb; a.w1_img = d.w1x_img; a.w2_img = d.w2x_img;
                a.b1 = d.b1; a.b2 = d.b2; a.g2 = d.g2; a.e2 = d.e2; a.M = M; a.F = F; a.H = H;
                if (e->ffn32x2 >= 2) { a.ao = ao; a.wo_img = d.wox_img; a.bo = d.bo; a.g1 = d.g1; a.e1 = d.e1; }
                a.dbg = l == 0 ? e->ffn_dbg : nullptr;
                qkv_done = e->ffn32x2 >= 3 && l + 1 < c.num_layers;
                if (qkv_done) {
                    const DevLayer& nx = e->layers[l + 1];
                    a.wq_img = nx.wqx_img; a.bq = nx.bqkv; a.qk_out = qk; a.vt_out = vt; a.vt_ld = ws.vt_ld;
                    a.blk_win = grp.d_blk; a.win = grp.d_win;
                }
                LAUNCH_OK(ppg::launch_ffn32x2(a, s), "ffn32x2");
            } else if (e->ffn_fused) {
                FfnArgs a{};
                a.X = X; a.Xb = Xb; a.W1 = d.w1; a.b1 = d.b1; a.W2p = d.w2p; a.b2 = d.b2;
                a.gamma = d.g2; a.beta = d.e2; a.H = H; a.F = F; a.M = M; a.dbg = l == 0 ? e->ffn_dbg : nullptr;
                a.splits = ws.ffn_splits;
                a.partial = ws.ffn_splits > 1 ? reinterpret_cast<float*>(base + ws.part) : nullptr;
                if (fuse_op) { a.ao = ao; a.Wo = d.wo; a.bo = d.bo; a.g1 = d.g1; a.e1 = d.e1; a.W1 = d.w1k; }
                qkv_done = fuse_op && e->qkv_fused && l + 1 < c.num_layers;
                if (qkv_done) {
                    const DevLayer& nx = e->layers[l + 1];
                    a.Wq = nx.wqkvk; a.bq = nx.bqkv; a.qk_out = qk; a.vt_out = vt; a.vt_ld = ws.vt_ld;
                    a.blk_win = grp.d_blk; a.win = grp.d_win;
                }
                LAUNCH_OK(ppg::launch_ffn(prec, a, ws.ffn_nt, s), "ffn");
            } else {
                qkv_done = false;
                LinearArgs a = base_args();
                a.act = act_x; a.lda_bytes = H * e->sz;
                a.groups_per_tap = hg; a.real_groups = hg; a.total_groups = hg;
                a.W = d.w1; a.bias = d.b1; a.N = F; a.out_rows = hid; a.out_ld = F;
                LAUNCH_OK(ppg::launch_linear(prec, EPI_RELU, 16, lnt, a, F / 256, s), "ffn1");
                LinearArgs b = base_args();
                const int fg = F / e->KG;
                b.act = hid; b.lda_bytes = F * e->sz;
                b.groups_per_tap = fg; b.real_groups = fg; b.total_groups = fg;
                b.W = d.w2; b.bias = d.b2; b.N = H; b.gamma = d.g2; b.beta = d.e2;
                LAUNCH_OK(ppg::launch_linear(prec, EPI_RESLN, H / 16, lnt_ln, b, 1, s), "ffn2+LN");
            }
        }
    }
    live = seg < 0 || seg == 1 + c.num_layers;
    {
        Timed t(e, PPG_K_OUTCONV_SOFTMAX, s, live);
        LinearArgs a = base_args();
        a.act = act_x; a.lda_bytes = H * e->sz; a.taps = 5;
        a.groups_per_tap = e->out_groups_per_tap; a.real_groups = 5 * e->out_groups_per_tap;
        a.total_groups = e->out_total_groups;
        a.W = e->w_out; a.bias = e->b_out; a.N = 48;
        a.out = out; a.out_T = frames; a.out_C = c.output_channels; a.softmax = softmax;
        a.overflow = e->d_overflow;
        if (e->lin_dbg_class == PPG_K_OUTCONV_SOFTMAX) a.dbg = e->lin_dbg;
        if (e->outconv && ppg::outconv_supported(prec, a)) LAUNCH_OK(ppg::launch_outconv(prec, a, s), "out-conv+softmax");
        else LAUNCH_OK(ppg::launch_linear(prec, EPI_OUTCONV, 3, lnt_ln, a, 1, s), "out-conv+softmax");
    }
    return PPG_OK;
    };   // run_group

    const size_t ngroups = plan.groups.size();
    if (ngroups > 1) HIP_OK(hipEventRecord(e->ev_fork, s));
    for (size_t gi = 1; gi < ngroups; ++gi) {
        hipStream_t side = e->side_streams[gi - 1];
        HIP_OK(hipStreamWaitEvent(side, e->ev_fork, 0));
        if (e->stream_offset_us > 0)
            hipLaunchKernelGGL(phase_delay_kernel, dim3(1), dim3(64), 0, side, (unsigned long long)(100ull * e->stream_offset_us * gi));
    }
    // The pipelines' launches are enqueued segment by segment, alternately: enqueued one whole pipeline after the
    // other, the second one's first kernel reached its queue ~50 us (a dozen launches) behind the first one's -- nothing
    // in a loop of steps, where the host runs ahead of the device, but the first step of a short timed block (and a
    // latency-bound caller's only step) started one pipeline that much late (tools/block_overhead.py).
    if (ngroups == 1) {
        if ((rc = run_group(plan.groups[0], s, -1))) return rc;
    } else {
        for (int seg = 0; seg < nseg; ++seg)
            for (size_t gi = 0; gi < ngroups; ++gi)
                if ((rc = run_group(plan.groups[gi], gi == 0 ? s : e->side_streams[gi - 1], seg))) return rc;
    }
    for (size_t gi = 1; gi < ngroups; ++gi) {
        HIP_OK(hipEventRecord(e->ev_join[gi - 1], e->side_streams[gi - 1]));
        HIP_OK(hipStreamWaitEvent(s, e->ev_join[gi - 1], 0));
    }
#undef LAUNCH_OK
    return PPG_OK;
}

// ----------------------------------------------------------------------------
// Streaming causal mode: one utterance, frames arrive in
// chunks, the K / V^T rows and the residual rows of everything seen so far stay
// on the device.  The reference defines no streaming state
// (config/causal_transformer.py:18 only adds the causal mask): what this
// reproduces is the causal forward of the WHOLE utterance (one window, <= the
// chunk length), emitted incrementally.  With a causal mask, row t of every layer
// depends on rows <= t of the layer below -- except for the two k = 5 'same'
// convolutions (+-2 frames each).  After F frames:
//   rows < F - 2 of the residual stream and of every layer are final,
//   posteriors of rows < F - 4 are final (all of them after `flush`).
// A step recomputes from the last 16-row block boundary at or below the previous
// frontier (the kernels work on whole 16-row blocks; recomputing a row gives the
// same bits), on ROW RANGES of the token-split kernels: pointers advanced by
// row0, the window table's tok_off lowered by row0 so that window positions
// (PE, masks, V^T columns) stay absolute.
// ----------------------------------------------------------------------------
// A stream object holds `batch` utterances (ppg_stream_create: one), each ONE window of up to `cap` frames, in one
// token space: item b owns rows b R .. (b + 1) R - 1 of every buffer (R = cap rounded up to 32), its own window
// record (position b R, valid = its frontier) and its own K | Q / V^T rows in the per-layer caches.  A step advances
// all items by their own -- ragged, unaligned -- numbers of frames in ONE launch sequence: the launches of the
// token-split kernels take a ROW MAP (the 16-row blocks the step touches, of any items, one per wave) instead of a
// contiguous row range, the attention launch takes the items' query tiles.  Rows are recomputed from the last block
// boundary at or below an item's previous frontier; a recomputed row gives the same bits (the hidden chunks are
// walked in a fixed order under a row map).
struct StreamItemMeta { int row0; int received_before; int count; int pad; };
__global__ void stream_append_kernel(const StreamItemMeta* meta, const char* chunk, int nmax, int C, int R, int esz, char* feats) {
    const int b = blockIdx.y;
    const StreamItemMeta m = meta[b];
    const int total = C * m.count;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int c = i / m.count, t = i - c * m.count;
        const char* src = chunk + (((size_t)b * C + c) * nmax + t) * esz;
        char* dst = feats + (((size_t)b * C + c) * R + m.received_before + t) * esz;
        if (esz == 2) *reinterpret_cast<uint16_t*>(dst) = *reinterpret_cast<const uint16_t*>(src);
        else *reinterpret_cast<uint32_t*>(dst) = *reinterpret_cast<const uint32_t*>(src);
    }
}

struct PpgStream {
    PpgEngine* e = nullptr;
    int batch = 1, cap = 0, rows = 0, dtype = 0;      // rows = R, per item
    Workspace ws{};
    char* buf = nullptr;          // workspace
    char* feats = nullptr;        // (batch, C, rows) in `dtype`
    float* probs = nullptr;       // (batch, output_channels, rows)
    int* d_blk = nullptr;
    char* d_tables = nullptr;     // one step's tables (layout: Tables)
    int* d_tickets = nullptr;     // one-pass split-hidden FFN: a counter per token tile, zero between launches (FfnArgs::tickets)
    size_t qk_bytes = 0, vt_bytes = 0, cache_off = 0, part_off = 0;
    int max_splits = 1;           // hidden splits of the FFN launches (a step's few rows cannot stream the weights through few CUs fast enough)
    std::vector<int> received, x_valid, o_valid;
    std::vector<char> finished;
    std::vector<int> map_scratch;
    // per-step tables, uploaded with ONE copy from a pinned slot (asynchronous: the source must outlive the call)
    struct Tables {
        size_t win = 0, meta = 0, items = 0, maps = 0, bytes = 0;   // byte offsets: win[3][batch], meta[batch], items[max_items], maps[3][max_blocks]
        int max_items = 0, max_blocks = 0;
    } tb;
    static constexpr int kSlots = 8;
    char* staging = nullptr;
    hipEvent_t uploaded[kSlots] = {};
    unsigned step = 0;
    // A step's ~22 launches behind the table upload as ONE hipGraph replay (round 5).  The launches of a step are a
    // function of (row blocks per map, query tiles, softmax): a stream advanced at a steady cadence repeats a handful
    // of keys.  A key seen for the second time is captured (on the stream's own capture stream: the caller's may be
    // the legacy null stream, which cannot capture), from then on replayed.  OFF by default (PPGS_AMD_STREAM_GRAPH=1 turns
    // it on): on ROCm 7.2 the replay of these 22 short nodes plus the fork / join events adds 45 - 50 us to a synchronised
    // step (one stream 266 -> 315 us, 64 streams 361 -> 406 us): the step is bound by its dependent kernels, not by launches.
    struct StepGraph { hipGraphExec_t exec = nullptr; int seen = 0; };
    std::map<std::array<int, 6>, StepGraph> graphs;
    hipStream_t gstream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool graph_steps = false;     // (measured: a replay costs MORE than the launches it replaces, profiles/r5_stream_step_graph.txt)
    bool fused_layers = false;    // PPGS_AMD_STREAM_FUSED=2 at creation: every step's layers as ONE fused launch each (for streams pushed in whole chunks); fixed for the life of the stream
    ~PpgStream() {
        if (e) (void)hipSetDevice(e->device);
        for (auto& kv : graphs) if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
        if (gstream) (void)hipStreamDestroy(gstream);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        for (void* p : {(void*)buf, (void*)feats, (void*)probs, (void*)d_blk, (void*)d_tables, (void*)d_tickets}) if (p) (void)hipFree(p);
        if (staging) (void)hipHostFree(staging);
        for (hipEvent_t ev : uploaded) if (ev) (void)hipEventDestroy(ev);
    }
};

int ppg_stream_create_batch(PpgEngine* e, int batch, int max_frames, int feature_dtype, PpgStream** out) {
    if (!e || !out) return fail(PPG_EINVAL, "null argument");
    if (!e->cfg.is_causal) return fail(PPG_EINVAL, "streaming needs a causal engine (is_causal = 1)");
    if (batch < 1 || batch > 4096) return fail(PPG_EINVAL, "batch %d outside [1, 4096]", batch);
    if (max_frames < 1 || max_frames > e->cfg.chunk_length)
        return fail(PPG_ELENGTH, "max_frames %d outside [1, %d] (one window)", max_frames, e->cfg.chunk_length);
    if (feature_dtype != PPG_DTYPE_F16 && feature_dtype != PPG_DTYPE_F32) return fail(PPG_EINVAL, "feature dtype %d", feature_dtype);
    if (e->split && !e->ffn_fused) return fail(PPG_EINVAL, "the fp16x2 mode has no KV-cached stream at hidden %d", e->cfg.hidden_channels);
    std::lock_guard<std::mutex> lock(e->mu);
    HIP_OK(hipSetDevice(e->device));
    std::unique_ptr<PpgStream> st(new PpgStream);
    st->e = e; st->batch = batch; st->cap = max_frames; st->rows = round_up(max_frames, 32); st->dtype = feature_dtype;
    st->graph_steps = ppg::env_experiment("PPGS_AMD_STREAM_GRAPH", 0) != 0;
    st->fused_layers = ppg::env_switch("PPGS_AMD_STREAM_FUSED", 0) == 2;
    st->received.assign(batch, 0); st->x_valid.assign(batch, 0); st->o_valid.assign(batch, 0); st->finished.assign(batch, 0);
    const PpgConfig& c = e->cfg;
    const int R = st->rows, MT = batch * R;
    st->ws = layout(e, MT, MT);
    // K | Q rows and V^T per LAYER (the one-shot forward reuses one pair for all layers; here they are the cache)
    st->qk_bytes = align_up((size_t)st->ws.qk_rows * 2 * c.hidden_channels * e->sz, 256);
    st->vt_bytes = align_up((size_t)c.hidden_channels * st->ws.vt_ld * e->sz, 256);
    // partial sums of the split-hidden FFN launches: [splits][batch * R][H] fp32, at most 256 MiB
    {
        const int chunks = c.ffn_channels / (32768 / (c.hidden_channels * e->sz));
        // (under a row map a split's rows are slot-dense, ceil(map_blocks / 4) * 64 of them -- ppg_kernels.hip, ffn_body's
        // epilogue and the reduce pass: up to MT rounded up to a whole 64-row workgroup, not MT)
        const size_t per_split = (size_t)round_up(MT, 64) * c.hidden_channels * 4;
        st->max_splits = 1;
        while (st->max_splits * 2 <= std::max(1, chunks / 2) && (size_t)(st->max_splits * 2) * per_split <= ((size_t)256 << 20)) st->max_splits *= 2;
        st->part_off = align_up(st->ws.total, 256);
        st->cache_off = align_up(st->part_off + (st->max_splits > 1 ? st->max_splits * per_split : 0), 256);
    }
    const size_t bytes = st->cache_off + (size_t)c.num_layers * (st->qk_bytes + st->vt_bytes);
    HIP_OK(hipMalloc(reinterpret_cast<void**>(&st->buf), bytes));
    HIP_OK(hipMemset(st->buf, 0, bytes));
    const size_t esz = feature_dtype == PPG_DTYPE_F16 ? 2 : 4;
    HIP_OK(hipMalloc(reinterpret_cast<void**>(&st->feats), (size_t)batch * c.input_channels * R * esz));
    HIP_OK(hipMemset(st->feats, 0, (size_t)batch * c.input_channels * R * esz));
    HIP_OK(hipMalloc(reinterpret_cast<void**>(&st->probs), (size_t)batch * c.output_channels * R * 4));
    HIP_OK(hipMemset(st->probs, 0, (size_t)batch * c.output_channels * R * 4));
    // block -> window: block k belongs to item k / (R / 16)
    {
        std::vector<int> blk(MT / 16 + 8, -1);
        for (int k = 0; k < MT / 16; ++k) blk[k] = k / (R / 16);
        HIP_OK(hipMalloc(reinterpret_cast<void**>(&st->d_blk), blk.size() * sizeof(int)));
        HIP_OK(hipMemcpy(st->d_blk, blk.data(), blk.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    PpgStream::Tables& tb = st->tb;
    const int tile = e->head_dim == 128 ? ppg::attn_query_tile(e->head_dim) / 2 : ppg::attn_query_tile(e->head_dim);
    tb.max_items = batch * (R / tile + 1);
    tb.max_blocks = round_up(MT / 16, 4) + 4;
    size_t off = 0;
    tb.win = off; off = align_up(off + 3 * (size_t)batch * sizeof(PpgWindow), 64);
    tb.meta = off; off = align_up(off + (size_t)batch * sizeof(StreamItemMeta), 64);
    tb.items = off; off = align_up(off + (size_t)tb.max_items * sizeof(AttnItem), 64);
    tb.maps = off; off = align_up(off + 3 * align_up((size_t)tb.max_blocks * sizeof(int), 64), 64) + 256;   // (packed per step, each map 64-byte aligned)
    tb.bytes = off;
    HIP_OK(hipMalloc(reinterpret_cast<void**>(&st->d_tables), tb.bytes));
    HIP_OK(hipMemset(st->d_tables, 0, tb.bytes));
    if (e->stream_one_pass && c.hidden_channels == 256 && e->ffn_fused) {
        const size_t tiles = (size_t)tb.max_blocks / 4 + 4;
        HIP_OK(hipMalloc(reinterpret_cast<void**>(&st->d_tickets), tiles * sizeof(int)));
        HIP_OK(hipMemset(st->d_tickets, 0, tiles * sizeof(int)));
    }
    HIP_OK(hipHostMalloc(reinterpret_cast<void**>(&st->staging), PpgStream::kSlots * tb.bytes, hipHostMallocDefault));
    for (hipEvent_t& ev : st->uploaded) HIP_OK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    HIP_OK(hipDeviceSynchronize());
    *out = st.release();
    return PPG_OK;
}

int ppg_stream_create(PpgEngine* e, int max_frames, int feature_dtype, PpgStream** out) {
    return ppg_stream_create_batch(e, 1, max_frames, feature_dtype, out);
}

void ppg_stream_destroy(PpgStream* stream) { delete stream; }

int ppg_stream_rows(const PpgStream* st, int* rows, int* received, int* final_frames) {
    if (!st) return fail(PPG_EINVAL, "null argument");
    if (rows) *rows = st->rows;
    if (received) *received = st->received[0];
    if (final_frames) *final_frames = st->o_valid[0];
    return PPG_OK;
}

int ppg_stream_batch(const PpgStream* st) { return st ? st->batch : 0; }

const float* ppg_stream_posteriors(const PpgStream* st) { return st ? st->probs : nullptr; }

int ppg_stream_push_batch(PpgStream* st, const void* chunk, int nmax, const int* counts, const int* flush, int softmax,
                          int* first_final, int* num_final, void* stream_) {
    if (!st || !counts || nmax < 0) return fail(PPG_EINVAL, "bad argument");
    PpgEngine* e = st->e;
    const int B = st->batch, R = st->rows;
    bool any_frames = false;
    for (int b = 0; b < B; ++b) {
        const int n = counts[b];
        if (n < 0 || n > nmax) return fail(PPG_EINVAL, "item %d: %d frames outside [0, %d]", b, n, nmax);
        if (n > 0 || (flush && flush[b])) {
            if (st->finished[b]) return fail(PPG_EINVAL, "item %d: the stream was flushed", b);
            if (st->received[b] + n > st->cap) return fail(PPG_ELENGTH, "item %d: %d + %d frames > max_frames %d", b, st->received[b], n, st->cap);
        }
        any_frames = any_frames || n > 0;
    }
    if (any_frames && !chunk) return fail(PPG_EINVAL, "null chunk");
    std::lock_guard<std::mutex> lock(e->mu);
    HIP_OK(hipSetDevice(e->device));
    hipStream_t s = static_cast<hipStream_t>(stream_);
    const PpgConfig& c = e->cfg;
    const int H = c.hidden_channels, F = c.ffn_channels, prec = c.precision, MT = B * R;
    const int esz = st->dtype == PPG_DTYPE_F16 ? 2 : 4;
    const PpgStream::Tables& tb = st->tb;

    const unsigned slot_index = st->step++ % PpgStream::kSlots;
    char* stage = st->staging + (size_t)slot_index * tb.bytes;
    HIP_OK(hipEventSynchronize(st->uploaded[slot_index]));               // (its last use, kSlots steps ago, is long done)
    PpgWindow* win = reinterpret_cast<PpgWindow*>(stage + tb.win);       // [0] linear launches, [1] out-conv, [2] gather
    StreamItemMeta* meta = reinterpret_cast<StreamItemMeta*>(stage + tb.meta);
    AttnItem* items = reinterpret_cast<AttnItem*>(stage + tb.items);
    // the three row maps ([0] residual rows, [1] posterior rows, [2] gathered rows) are collected here and packed
    // right behind the items in use: the step uploads what it filled, not the tables' capacity
    std::vector<int>& maps = st->map_scratch;
    maps.assign(3 * (size_t)tb.max_blocks, 0);
    int nmap[3] = {0, 0, 0}, nitems = 0, max_count = 0;
    const int tile = e->head_dim == 128 ? ppg::attn_query_tile(e->head_dim) / 2 : ppg::attn_query_tile(e->head_dim);
    // the items' new frontiers are worked out on copies and committed only once every launch of the step is queued:
    // a step that fails half way (too many query tiles, a refused launch) leaves the stream where it was
    std::vector<int> received = st->received, x_valid = st->x_valid, o_valid = st->o_valid;
    std::vector<char> finished = st->finished;
    for (int b = 0; b < B; ++b) {
        const int n = counts[b];
        const bool fl = flush && flush[b];
        const bool active = n > 0 || fl;
        const int f_prev = received[b], x_prev = x_valid[b], o_prev = o_valid[b];
        if (active) received[b] += n;
        const int x_new = !active ? x_prev : (fl ? received[b] : std::max(received[b] - 2, 0));
        const int o_new = !active ? o_prev : (fl ? received[b] : std::max(x_new - 2, 0));
        if (first_final) first_final[b] = o_prev;
        if (num_final) num_final[b] = o_new - o_prev;
        if (fl) finished[b] = 1;
        x_valid[b] = x_new; o_valid[b] = o_new;
        // the item's window as the three kinds of launch see it
        PpgWindow w{};
        w.item = b; w.chunked = 0; w.start = 0; w.frames = R; w.valid = x_new;
        w.keep_lo = 0; w.keep_hi = R; w.out_frame = 0; w.vt_off = b * R; w.tok_off = b * R;
        win[b] = w;
        win[B + b] = w; win[B + b].frames = fl ? received[b] : R;    // out-conv: zero padding at the true end once it is known
        win[2 * B + b] = w;
        meta[b] = StreamItemMeta{b * R, f_prev, active ? n : 0, 0};
        max_count = std::max(max_count, meta[b].count);
        if (!active) continue;
        const int r0 = x_prev / 16 * 16, r1 = round_up(x_new, 16);          // rows of the residual stream to (re)compute
        const int g0 = f_prev / 16 * 16, g1 = round_up(received[b], 16);  // rows whose features changed
        const int o0 = o_prev / 16 * 16, o1 = round_up(o_new, 16);          // posterior rows
        for (int r = r0; r < r1; r += 16) maps[nmap[0]++] = b * R + r;
        for (int r = o0; r < o1; r += 16) maps[tb.max_blocks + nmap[1]++] = b * R + r;
        for (int r = g0; r < g1; r += 16) maps[2 * tb.max_blocks + nmap[2]++] = b * R + r;
        for (int q0 = r0 / tile * tile; q0 < r1; q0 += tile) {
            if (nitems == tb.max_items) return fail(PPG_EINVAL, "more than %d query tiles in one step", tb.max_items);
            items[nitems++] = AttnItem{b, q0, b * R, b * R, R, x_new, e->head_dim == 128 ? 1 : 0, 0};
        }
    }
    size_t map_off[3], used = align_up(tb.items + (size_t)nitems * sizeof(AttnItem), 64);
    for (int k = 0; k < 3; ++k) {
        const int padded = round_up(nmap[k], 4);                           // unused wave slots of the last workgroup: nothing to do
        for (int i = nmap[k]; i < padded; ++i) maps[k * tb.max_blocks + i] = MT;
        map_off[k] = used;
        memcpy(stage + used, maps.data() + (size_t)k * tb.max_blocks, (size_t)padded * sizeof(int));
        used = align_up(used + (size_t)padded * sizeof(int), 64);
    }
    // the launch stream: the caller's, or (graph mode) the stream's own behind a fork event
    hipStream_t caller = s;
    PpgStream::StepGraph* sg = nullptr;
    if (st->graph_steps) {
        if (!st->gstream) {
            HIP_OK(hipStreamCreateWithFlags(&st->gstream, hipStreamNonBlocking));
            HIP_OK(hipEventCreateWithFlags(&st->ev_fork, hipEventDisableTiming));
            HIP_OK(hipEventCreateWithFlags(&st->ev_join, hipEventDisableTiming));
        }
        if (st->graphs.size() >= 32) {               // (an irregular cadence: start over rather than grow)
            for (auto& kv : st->graphs) if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
            st->graphs.clear();
        }
        sg = &st->graphs[std::array<int, 6>{nmap[0], nmap[1], nmap[2], nitems, softmax, max_count > 0}];       // (the fused-layer choice follows nmap[0])
        HIP_OK(hipEventRecord(st->ev_fork, caller));
        HIP_OK(hipStreamWaitEvent(st->gstream, st->ev_fork, 0));
        s = st->gstream;
    }
    HIP_OK(hipMemcpyAsync(st->d_tables, stage, used, hipMemcpyHostToDevice, s));
    const PpgWindow* d_win = reinterpret_cast<const PpgWindow*>(st->d_tables + tb.win);
    const StreamItemMeta* d_meta = reinterpret_cast<const StreamItemMeta*>(st->d_tables + tb.meta);
    const AttnItem* d_items = reinterpret_cast<const AttnItem*>(st->d_tables + tb.items);

#define LAUNCH_OK(expr, what)                                                        \
    do {                                                                             \
        hipError_t he_ = (expr);                                                     \
        if (he_ != hipSuccess) return fail(PPG_EDEVICE, "%s: %s", what, hipGetErrorString(he_)); \
    } while (0)

    if (max_count > 0) {
        const int per_item = c.input_channels * max_count;
        hipLaunchKernelGGL(stream_append_kernel, dim3((per_item + 255) / 256, B), dim3(256), 0, s,
                           d_meta, static_cast<const char*>(chunk), nmax, c.input_channels, R, esz, st->feats);
        LAUNCH_OK(hipGetLastError(), "stream append");
    }
    // everything from here to the out-convolution depends on the key only: replay it, or capture it on its second
    // appearance (every kernel of the sequence has then run eagerly once: their one-time attribute calls are done)
    bool capturing = false;
    auto finish = [&](int rc) {                       // leave the capture (if any), join the caller's stream
        if (capturing) {
            hipGraph_t graph = nullptr;
            const hipError_t he = hipStreamEndCapture(s, &graph);
            capturing = false;
            if (rc == PPG_OK && he == hipSuccess && graph) {
                if (hipGraphInstantiate(&sg->exec, graph, nullptr, nullptr, 0) == hipSuccess) {
                    if (hipGraphLaunch(sg->exec, s) != hipSuccess) rc = fail(PPG_EDEVICE, "stream step: graph launch failed");
                } else {
                    sg->exec = nullptr;
                    rc = fail(PPG_EDEVICE, "stream step: graph instantiation failed");
                }
            } else if (rc == PPG_OK) {
                rc = fail(PPG_EDEVICE, "stream step: capture failed: %s", hipGetErrorString(he));
            }
            if (graph) (void)hipGraphDestroy(graph);
            if (rc != PPG_OK) sg->seen = -(1 << 30);        // (never try this key again)
        }
        return rc;
    };
#undef LAUNCH_OK
#define LAUNCH_OK(expr, what)                                                        \
    do {                                                                             \
        hipError_t he_ = (expr);                                                     \
        if (he_ != hipSuccess) return finish(fail(PPG_EDEVICE, "%s: %s", what, hipGetErrorString(he_))); \
    } while (0)
    const bool replay = sg && sg->exec;
    if (sg && !replay && ++sg->seen >= 2) {
        if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) capturing = true;
    }
    if (replay) {
        if (hipGraphLaunch(sg->exec, s) != hipSuccess) return fail(PPG_EDEVICE, "stream step: graph launch failed");
    } else {
    char* base = st->buf;
    const Workspace& ws = st->ws;
    char* xw = base + ws.xw;
    float* X = reinterpret_cast<float*>(base + ws.x);
    char* Xb = (e->sz == 2 || e->split) ? base + ws.xb : nullptr;
    char* ao = base + ws.ao;
    auto qk_of = [&](int l) { return base + st->cache_off + (size_t)l * (st->qk_bytes + st->vt_bytes); };
    auto vt_of = [&](int l) { return qk_of(l) + st->qk_bytes; };
    const char* act_x = (e->sz == 2 || e->split) ? Xb : reinterpret_cast<const char*>(X);

    if (nmap[2] > 0) {
        GatherArgs g{};
        g.feats = st->feats; g.dtype = st->dtype; g.C = c.input_channels; g.T = R; g.overlap = c.chunk_overlap;
        g.xw = xw; g.Cp = e->Cp;
        g.blk_win = st->d_blk; g.win = d_win + 2 * B; g.M = MT;
        g.rowmap = reinterpret_cast<const int*>(st->d_tables + map_off[2]); g.map_blocks = nmap[2];
        g.vt = vt_of(0); g.vt_ld = ws.vt_ld; g.vt_rows = H; g.vt_tokens = MT; g.nwin = 0;     // (no window tails to clear: R is a multiple of 32; the caches were zeroed at creation)
        g.qk_slack = qk_of(0) + (size_t)MT * 2 * H * e->sz; g.qk_slack_bytes = (int)(64 * 2 * H * e->sz);
        LAUNCH_OK(ppg::launch_gather(prec, g, s), "stream gather");
    }
    if (nmap[0] > 0) {
        auto base_args = [&]() {
            LinearArgs a{};
            a.blk_win = st->d_blk; a.win = d_win; a.M = MT; a.H = H;
            a.X = X; a.Xb = Xb; a.v_start = INT_MAX; a.taps = 1;
            a.rowmap = reinterpret_cast<const int*>(st->d_tables + map_off[0]); a.map_blocks = nmap[0];
            return a;
        };
        {
            LinearArgs a = base_args();
            a.act = xw; a.lda_bytes = e->Cp * e->sz; a.taps = 5;
            a.groups_per_tap = e->in_groups_per_tap; a.real_groups = 5 * e->in_groups_per_tap;
            a.total_groups = e->in_total_groups;
            a.W = e->w_in; a.bias = e->b_in; a.N = H; a.pe = e->pe;
            LAUNCH_OK(ppg::launch_linear(prec, EPI_INCONV, 16, 1, a, H / 256, s), "stream in-conv");
        }
        const int hg = H / e->KG;
        // hidden splits: as many workgroups per token tile as fill the chip (each streams its share of W1 / W2)
        int ffn_splits = 1;
        if (e->ffn_split) {
            const int wgs = (nmap[0] + 3) / 4;
            // (one-pass form: the tile's last workgroup sums the partial rows itself, 4 rows x <= 8 splits of loads in
            // flight per wave; four splits keep that to two round trips)
            int cap = e->ffn_split_max > 0 ? std::min(e->ffn_split_max, st->max_splits) : st->max_splits;
            if (st->d_tickets) cap = std::min(cap, e->ffn_split_max > 0 ? 8 : 4);
            while (ffn_splits * 2 <= cap && wgs * ffn_splits * 2 <= e->num_cus) ffn_splits *= 2;
        }
        // One launch per layer besides attention for BIG steps (round 5): out-projection + LayerNorm-1 + FFN + LayerNorm-2
        // + the NEXT layer's Q/K/V in the token-split fused kernel the one-shot forward uses for small batches, here under
        // the step's row map.  Its workgroups stream a whole layer's weights each (no hidden splits: a split would redo
        // the out-projection) -- ~85 us per layer however few they are, so only a step of >= 512 row blocks gains
        // (64 streams x 160 frames: 725 -> 639 us; 64 x 16: 363 -> 491 us, one stream 274 -> 432 us).  The form is a
        // property of the STREAM, fixed at creation (PPGS_AMD_STREAM_FUSED=2; default: the four launches): chosen per
        // step (round 5), a stream whose step sizes straddled the threshold computed a partially filled 16-row block
        // with two kernels of different accumulation order -- its cached K / V rows and the later recomputed X rows then
        // depended on the push cadence, against "a recomputed row gives the same bits wherever it lands".
        const bool fused = st->fused_layers && e->ffn_fused && e->op_fused && e->qkv_fused && !e->split;
        for (int l = 0; l < c.num_layers; ++l) {
            const DevLayer& d = e->layers[l];
            char* qk = qk_of(l);
            char* vt = vt_of(l);
            if (!fused || l == 0) {
                LinearArgs a = base_args();
                a.act = act_x; a.lda_bytes = H * e->sz;
                a.groups_per_tap = hg; a.real_groups = hg; a.total_groups = hg;
                a.W = d.wqkv; a.bias = d.bqkv; a.N = 3 * H;
                a.out_rows = qk; a.out_ld = 2 * H; a.vt = vt; a.vt_ld = ws.vt_ld; a.v_start = 2 * H;
                LAUNCH_OK(ppg::launch_linear(prec, EPI_QKV, 16, 1, a, 3 * H / 256, s), "stream qkv");
            }
            {
                AttnArgs a{};
                a.qk = qk; a.qk_ld_bytes = 2 * H * e->sz; a.vt = vt; a.vt_ld_bytes = ws.vt_ld * e->sz;
                a.ao = ao; a.H = H; a.causal = 1;
                a.items = d_items; a.win = d_win; a.M = MT; a.ao_tiled = 0; a.heads = c.heads;
                LAUNCH_OK(ppg::launch_attn(prec, a, nitems, c.heads, e->head_dim, s), "stream attention");
            }
            if (fused) {
                FfnArgs a{};
                a.X = X; a.Xb = Xb;
                a.W1 = d.w1k; a.b1 = d.b1; a.W2p = d.w2p; a.b2 = d.b2;
                a.gamma = d.g2; a.beta = d.e2; a.H = H; a.F = F; a.M = MT;
                a.splits = 1; a.partial = nullptr;
                a.rowmap = reinterpret_cast<const int*>(st->d_tables + map_off[0]); a.map_blocks = nmap[0];
                a.ao = ao; a.Wo = d.wo; a.bo = d.bo; a.g1 = d.g1; a.e1 = d.e1;
                if (l + 1 < c.num_layers) {
                    const DevLayer& nx = e->layers[l + 1];
                    a.Wq = nx.wqkvk; a.bq = nx.bqkv; a.qk_out = qk_of(l + 1); a.vt_out = vt_of(l + 1); a.vt_ld = ws.vt_ld;
                    a.blk_win = st->d_blk; a.win = d_win;
                }
                LAUNCH_OK(ppg::launch_ffn(prec, a, 1, s), "stream fused layer");
                continue;
            }
            {
                LinearArgs a = base_args();
                a.act = ao; a.lda_bytes = H * e->sz;
                a.groups_per_tap = hg; a.real_groups = hg; a.total_groups = hg;
                a.W = d.wo; a.bias = d.bo; a.N = H; a.gamma = d.g1; a.beta = d.e1;
                LAUNCH_OK(ppg::launch_linear(prec, EPI_RESLN, H / 16, 1, a, 1, s), "stream out-proj+LN");
            }
            {
                FfnArgs a{};
                a.X = X; a.Xb = Xb;
                a.W1 = d.w1; a.b1 = d.b1; a.W2p = d.w2p; a.b2 = d.b2;
                a.gamma = d.g2; a.beta = d.e2; a.H = H; a.F = F; a.M = MT;
                a.splits = ffn_splits; a.partial = ffn_splits > 1 ? reinterpret_cast<float*>(base + st->part_off) : nullptr;
                a.rowmap = reinterpret_cast<const int*>(st->d_tables + map_off[0]); a.map_blocks = nmap[0];
                a.tickets = ffn_splits > 1 ? st->d_tickets : nullptr;
                LAUNCH_OK(ppg::launch_ffn(prec, a, 1, s), "stream ffn");
            }
        }
    }
    if (nmap[1] > 0) {
        LinearArgs a{};
        a.blk_win = st->d_blk; a.win = d_win + B; a.M = MT; a.H = H; a.v_start = INT_MAX;
        a.act = act_x; a.lda_bytes = H * e->sz; a.taps = 5;
        a.groups_per_tap = e->out_groups_per_tap; a.real_groups = 5 * e->out_groups_per_tap;
        a.total_groups = e->out_total_groups;
        a.W = e->w_out; a.bias = e->b_out; a.N = 48;
        a.out = st->probs; a.out_T = R; a.out_C = c.output_channels; a.softmax = softmax;
        a.overflow = e->d_overflow;
        a.rowmap = reinterpret_cast<const int*>(st->d_tables + map_off[1]); a.map_blocks = nmap[1];
        LAUNCH_OK(ppg::launch_linear(prec, EPI_OUTCONV, 3, 1, a, 1, s), "stream out-conv+softmax");
    }
    }   // (eager or capturing)
    if (const int rc = finish(PPG_OK)) return rc;
#undef LAUNCH_OK
    HIP_OK(hipEventRecord(st->uploaded[slot_index], s));
    if (st->graph_steps) {
        HIP_OK(hipEventRecord(st->ev_join, s));
        HIP_OK(hipStreamWaitEvent(caller, st->ev_join, 0));
    }
    st->received.swap(received); st->x_valid.swap(x_valid); st->o_valid.swap(o_valid); st->finished.swap(finished);
    return PPG_OK;
}

int ppg_stream_push(PpgStream* st, const void* chunk, int n, int flush, int softmax,
                    int* first_final, int* num_final, void* stream_) {
    if (!st || st->batch != 1) return fail(PPG_EINVAL, "ppg_stream_push is the one-utterance form (use ppg_stream_push_batch)");
    if (n < 0 || (n > 0 && !chunk)) return fail(PPG_EINVAL, "bad argument");
    if (st->finished[0]) return fail(PPG_EINVAL, "the stream was flushed");
    return ppg_stream_push_batch(st, chunk, n, &n, &flush, softmax, first_final, num_final, stream_);
}

// ----------------------------------------------------------------------------
// wav2vec 2.0 feature encoder (w2v2fb representation)
// ----------------------------------------------------------------------------
namespace {
constexpr int kW2vLayers = 7;
const int kW2vKernel[kW2vLayers] = {10, 3, 3, 3, 3, 2, 2};      // transformers Wav2Vec2Config.conv_kernel
const int kW2vStride[kW2vLayers] = {5, 2, 2, 2, 2, 2, 2};       // .conv_stride
constexpr int kW2vChannels = 512;

// frames after each layer and the padded rows per item of each layer's token-major buffer:
// R[l-1] = 2 R[l], so that row m of layer l reads rows 2m + tap of layer l-1 for EVERY item
// (item b starts at row b * R[l]); R[6] = T[6] + 1 rounded up to 32 keeps every row a valid
// output reads inside its own item
struct W2vShape { long T[kW2vLayers]; long R[kW2vLayers]; };
bool w2v_shape(long samples, W2vShape* sh) {
    long t = samples;
    for (int l = 0; l < kW2vLayers; ++l) {
        if (t < kW2vKernel[l]) return false;
        t = (t - kW2vKernel[l]) / kW2vStride[l] + 1;
        sh->T[l] = t;
    }
    sh->R[kW2vLayers - 1] = (sh->T[kW2vLayers - 1] + 1 + 31) / 32 * 32;
    for (int l = kW2vLayers - 2; l >= 0; --l) sh->R[l] = 2 * sh->R[l + 1];
    return true;
}
}  // namespace

struct PpgW2v2 {
    PpgEngine eng;                 // device, operand size, upload bookkeeping
    float* w0 = nullptr;           // (512, 10)
    float* gamma = nullptr;
    float* beta = nullptr;
    char* w[kW2vLayers] = {};      // layers 1..6: [512 rows in paired order][taps * 512], GEMM operand type
    // PPGS_AMD_W2V2_CONV32=1 (experiment): layers 1..6 as plain GEMMs on ppg_gemm32.hip (fragment images of the same
    // weights).  Measured at 16 x 160 080 samples: 1.34 ms against 1.29 ms on linear_kernel<EPI_GELU> -- off.
    char* w_img[kW2vLayers] = {};
    float* zero_bias = nullptr;    // (the layers have no bias)
    bool conv32 = false;
};

int ppg_w2v2_create(const PpgW2v2Weights* wts, int precision, int device, PpgW2v2** out) {
    if (!wts || !out) return fail(PPG_EINVAL, "null argument");
    if (precision != PPG_PRECISION_FP32 && precision != PPG_PRECISION_BF16 && precision != PPG_PRECISION_FP16 && precision != PPG_PRECISION_FP16X2)
        return fail(PPG_EINVAL, "precision %d", precision);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(PPG_EDEVICE, "no HIP device: the wav2vec2 feature encoder has no CPU path");
    if (device < 0 || device >= ndev) return fail(PPG_EDEVICE, "device %d of %d", device, ndev);
    HIP_OK(hipSetDevice(device));
    std::unique_ptr<PpgW2v2> m(new PpgW2v2());
    PpgEngine* E = &m->eng;
    E->device = device;
    E->cfg.precision = precision;
    // (fp16x2: layers 1..6 with every operand an fp16 hi + lo pair in the fp32 path's byte layout -- PrecX2)
    E->split = precision == PPG_PRECISION_FP16X2;
    E->sz = (precision == PPG_PRECISION_FP32 || E->split) ? 4 : 2;
    E->KG = 64 / E->sz;
    hipDeviceProp_t prop;
    HIP_OK(hipGetDeviceProperties(&prop, device));
    E->num_cus = prop.multiProcessorCount;
    int rc;
    for (int l = 0; l < kW2vLayers; ++l) if (!wts->conv_weight[l]) return fail(PPG_EINVAL, "conv_weight[%d] is null", l);
    if (!wts->norm_weight || !wts->norm_bias) return fail(PPG_EINVAL, "group-norm parameters are null");
    if ((rc = upload_f32(E, wts->conv_weight[0], (size_t)kW2vChannels * 10, 0, &m->w0))) return rc;
    if ((rc = upload_f32(E, wts->norm_weight, kW2vChannels, 0, &m->gamma))) return rc;
    if ((rc = upload_f32(E, wts->norm_bias, kW2vChannels, 0, &m->beta))) return rc;
    for (int l = 1; l < kW2vLayers; ++l) {
        // torch Conv1d weight (out, in, k) -> [out (paired order)][tap * 512 + in]
        const float* w = wts->conv_weight[l];
        const int k = kW2vKernel[l], C = kW2vChannels;
        rc = upload_matrix(E, C, k * C, C, k * C,
                           [&](int r, int col) { const int tap = col / C, c = col - tap * C; return w[((size_t)pair_row(r) * C + c) * k + tap]; },
                           &m->w[l]);
        if (rc) return rc;
    }
    m->conv32 = ppg::env_experiment("PPGS_AMD_W2V2_CONV32", m->conv32) != 0;
    if (E->sz != 2) m->conv32 = false;
    if (m->conv32) {
        // layers 1..6 as plain GEMMs on the feature-split kernel: output row m reads the k input rows 2 m .. as ONE
        // contiguous run of K = k * 512 elements (rows of the input overlap: lda = 2 rows).  Images as the body's:
        // [N / 256][wave][K / 128][rb][8 K-steps], rows in accumulator order phi, K index = tap * 512 + channel.
        auto phi = [](int rho) { return 16 * ((rho >> 2) & 1) + 4 * (rho >> 3) + (rho & 3); };
        const int C = kW2vChannels;
        for (int l = 1; l < kW2vLayers; ++l) {
            const float* w = wts->conv_weight[l];
            const int k = kW2vKernel[l], K = k * C, chunks = K / 128, frags = (C / 256) * 4 * chunks * 16;
            rc = upload_matrix(E, frags * 64, 8, frags * 64, 8,
                               [&](int r, int j) {
                                   const int f = r >> 6, ln = r & 63;
                                   const int ks = f & 7, rb = (f >> 3) & 1, c = (f >> 4) % chunks, wv = ((f >> 4) / chunks) & 3, p = (f >> 4) / chunks / 4;
                                   const int n = 256 * p + 64 * wv + 32 * rb + phi(ln & 31), kk = 128 * c + 16 * ks + 8 * (ln >> 5) + j;
                                   const int tap = kk / C, ch = kk - tap * C;
                                   return w[((size_t)n * C + ch) * k + tap];
                               }, &m->w_img[l]);
            if (rc) return rc;
        }
        std::vector<float> zeros(C, 0.f);
        if ((rc = upload_f32(E, zeros.data(), C, 0, &m->zero_bias))) return rc;
    }
    *out = m.release();
    return PPG_OK;
}

void ppg_w2v2_destroy(PpgW2v2* model) { delete model; }

int64_t ppg_w2v2_frames(int64_t samples) {
    W2vShape sh;
    return w2v_shape(samples, &sh) ? sh.T[kW2vLayers - 1] : -1;
}

int ppg_w2v2_workspace_bytes(const PpgW2v2* model, int batch, int64_t samples, size_t* bytes) {
    if (!model || !bytes || batch <= 0) return fail(PPG_EINVAL, "bad argument");
    W2vShape sh;
    if (!w2v_shape(samples, &sh)) return fail(PPG_EINVAL, "%lld samples are too few for the conv stack", (long long)samples);
    const size_t row = (size_t)kW2vChannels * model->eng.sz;
    size_t total = align_up((size_t)batch * 65 * sizeof(double), 256);
    total += align_up((size_t)batch * kW2vChannels * sizeof(float2), 256);
    // (+ 1 row: the last output row of a 3-tap layer reads one row past its input -- a padding row nobody consumes)
    total += align_up(((size_t)batch * sh.R[0] + 1) * row, 256);
    total += align_up(((size_t)batch * sh.R[1] + 1) * row, 256);
    *bytes = total;
    return PPG_OK;
}

int ppg_w2v2_features(PpgW2v2* model, const float* audio, int batch, int64_t samples, float* out,
                      void* workspace, size_t workspace_bytes, void* stream) {
    if (!model || !audio || !out || !workspace) return fail(PPG_EINVAL, "null argument");
    size_t need = 0;
    int rc = ppg_w2v2_workspace_bytes(model, batch, samples, &need);
    if (rc) return rc;
    if (workspace_bytes < need) return fail(PPG_EWORKSPACE, "workspace %zu bytes < required %zu", workspace_bytes, need);
    PpgEngine* E = &model->eng;
    HIP_OK(hipSetDevice(E->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    W2vShape sh;
    w2v_shape(samples, &sh);
    const size_t row = (size_t)kW2vChannels * E->sz;
    char* base = static_cast<char*>(workspace);
    double* moments = reinterpret_cast<double*>(base);
    size_t off = align_up((size_t)batch * 65 * sizeof(double), 256);
    float2* scale_shift = reinterpret_cast<float2*>(base + off);
    off += align_up((size_t)batch * kW2vChannels * sizeof(float2), 256);
    char* bufs[2];
    bufs[0] = base + off;
    off += align_up(((size_t)batch * sh.R[0] + 1) * row, 256);
    bufs[1] = base + off;
    const int prec = E->cfg.precision;
    hipError_t he = ppg::launch_w2v2_layer0(prec, audio, batch, samples, sh.T[0], (int)sh.R[0], model->w0, model->gamma, model->beta,
                                            moments, scale_shift, bufs[0], s);
    if (he != hipSuccess) return fail(PPG_EDEVICE, "w2v2 layer 0: %s", hipGetErrorString(he));
    for (int l = 1; l < kW2vLayers; ++l) {
        if (model->conv32 && kW2vStride[l] == 2 && kW2vKernel[l] * kW2vChannels >= 384) {
            Gemm32Args g{};
            g.x = bufs[(l - 1) & 1]; g.lda_bytes = (int)(kW2vStride[l] * row); g.w_img = model->w_img[l]; g.bias = model->zero_bias;
            g.out16 = bufs[l & 1]; g.M = (int)(batch * sh.R[l]); g.N = kW2vChannels; g.K = kW2vKernel[l] * kW2vChannels; g.act_fn = 2;
            he = ppg::launch_gemm32(prec, g, s);
            if (he != hipSuccess) return fail(PPG_EDEVICE, "w2v2 conv layer %d: %s", l, hipGetErrorString(he));
            continue;
        }
        LinearArgs a{};
        a.v_start = INT_MAX;
        a.act = bufs[(l - 1) & 1]; a.lda_bytes = (int)row; a.taps = kW2vKernel[l];
        a.groups_per_tap = kW2vChannels / E->KG;
        a.real_groups = a.total_groups = a.taps * a.groups_per_tap;
        a.W = model->w[l]; a.N = kW2vChannels; a.H = kW2vChannels;
        a.out_rows = bufs[l & 1]; a.out_ld = kW2vChannels;
        a.M = (int)(batch * sh.R[l]); a.M_in = (int)(batch * sh.R[l - 1]); a.stride = kW2vStride[l];
        const int nt = choose_nt(E, a.M, E->sz == 2 ? 2 : 2);
        he = ppg::launch_linear(prec, EPI_GELU, 16, std::min(nt, 2), a, kW2vChannels / 256, s);
        if (he != hipSuccess) return fail(PPG_EDEVICE, "w2v2 conv layer %d: %s", l, hipGetErrorString(he));
    }
    he = ppg::launch_w2v2_output(prec, bufs[(kW2vLayers - 1) & 1], batch, (int)sh.R[kW2vLayers - 1], sh.T[kW2vLayers - 1], out, s);
    if (he != hipSuccess) return fail(PPG_EDEVICE, "w2v2 output: %s", hipGetErrorString(he));
    return PPG_OK;
}

// ----------------------------------------------------------------------------
// wav2vec 2.0 transformer body (include/ppgs_amd.h: ppg_w2v2_body_*): HF Wav2Vec2FeatureProjection,
// Wav2Vec2PositionalConvEmbedding and 12 post-norm encoder layers, one launch per GEMM (a fused layer at
// hidden 768 is bound by the weights a workgroup would stream).  16-bit modes: every projection (feature projection,
// Q | K | V, out-proj, FFN-1, FFN-2) on ppg_gemm32.hip with its epilogue fixed per use, the positional convolution on
// ppg_posconv.hip, LayerNorm-768 as a row kernel, attention as attn_kernel<.., 1, 64> (12 heads of 64).  fp32 mode (and
// the PPGS_AMD_W2V2_* = 0 switches): the same sequence on linear_kernel<EPI_QKV / EPI_GENERAL> (bias, GELU, residual
// in the epilogue; the positional convolution as 16 grouped k-tap GEMMs of one launch).  Token space: item b owns rows
// b R .. b R + frames - 1, R = frames rounded up to 32 (no half-written V^T groups); one attention window per item,
// keys limited to its valid frames.  Batches of >= 8 items run as two half-batches on two HIP streams.
// ----------------------------------------------------------------------------
struct PpgW2v2Body {
    PpgEngine eng;
    int hidden = 0, heads = 0, ffn = 0, layers = 0, taps = 0, groups = 0, gpt = 0;
    float eps = 1e-5f;
    float* pn_g = nullptr; float* pn_b = nullptr;
    char* proj_w = nullptr; float* proj_b = nullptr;
    char* proj_img = nullptr;      // the feature projection as gemm32 fragment images (16-bit modes)
    char* pos_w = nullptr; float* pos_b = nullptr;
    char* pos_img = nullptr;       // the positional convolution's fragment image (ppg_posconv.hip, 16-bit modes)
    bool posconv = true;           // PPGS_AMD_W2V2_POSCONV=0: the convolution as a k-tap GEMM on linear_kernel<EPI_GENERAL>
    float* en_g = nullptr; float* en_b = nullptr;
    struct Layer { char* wqkv; float* bqkv; char* wo; float* bo; float* g1; float* e1; char* w1; float* b1; char* w2; float* b2; float* g2; float* e2;
                   char* wo_img; char* w1_img; char* w2_img; char* wqkv_img; };   // fragment images for ppg_gemm32.hip (16-bit modes)
    bool gemm32 = true;            // PPGS_AMD_W2V2_GEMM32=0: linear_kernel<EPI_GENERAL> for every projection
    bool qkv32 = true;             // PPGS_AMD_W2V2_QKV32=0: Q/K/V on linear_kernel<EPI_QKV>
    std::vector<Layer> layer;
    // per pipeline (a batch of >= 8 items runs as two half-batches on two HIP streams, as the PPG network's engine does)
    struct Slot { char* staging = nullptr; size_t staging_bytes = 0; hipEvent_t uploaded = nullptr; };   // pinned tables of the call in flight
    Slot slot[2];
    int pipelines = 2;             // PPGS_AMD_W2V2_STREAMS
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    ~PpgW2v2Body() {
        (void)hipSetDevice(eng.device);
        for (Slot& sl : slot) {
            if (sl.staging) (void)hipHostFree(sl.staging);
            if (sl.uploaded) (void)hipEventDestroy(sl.uploaded);
        }
        if (side) (void)hipStreamDestroy(side);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
    }
};

int ppg_w2v2_body_create(const PpgW2v2BodyWeights* w, int precision, int device, PpgW2v2Body** out) {
    if (!w || !out) return fail(PPG_EINVAL, "null argument");
    if (precision != PPG_PRECISION_FP32 && precision != PPG_PRECISION_BF16 && precision != PPG_PRECISION_FP16 && precision != PPG_PRECISION_FP16X2)
        return fail(PPG_EINVAL, "precision %d", precision);
    const int H = w->hidden, F = w->ffn, L = w->num_layers;
    if (H != 768 || w->heads <= 0 || H / w->heads != 64 || H % w->heads) return fail(PPG_EINVAL, "hidden %d / heads %d: the body kernels are built for 768 = 12 x 64", H, w->heads);
    if (F <= 0 || F % 256 || L < 0 || L > PPG_W2V2_MAX_LAYERS) return fail(PPG_EINVAL, "ffn %d, layers %d", F, L);
    if (w->conv_groups != 16 || w->conv_kernel <= 0 || w->conv_kernel % 2 || H / w->conv_groups != 48)
        return fail(PPG_EINVAL, "positional convolution: kernel %d groups %d", w->conv_kernel, w->conv_groups);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(PPG_EDEVICE, "no HIP device: the wav2vec2 body has no CPU path");
    if (device < 0 || device >= ndev) return fail(PPG_EDEVICE, "device %d of %d", device, ndev);
    HIP_OK(hipSetDevice(device));
    std::unique_ptr<PpgW2v2Body> m(new PpgW2v2Body());
    PpgEngine* E = &m->eng;
    E->device = device; E->cfg.precision = precision;
    // fp16x2: every projection and the attention on fp16 hi + lo operand pairs (PrecX2: the fp32 path's launch sequence
    // and byte layout, three fp16 MFMAs per product); the positional convolution stays on f32-input MFMAs (its groups
    // of 48 channels are not whole [32 hi | 32 lo] blocks)
    E->split = precision == PPG_PRECISION_FP16X2;
    E->sz = (precision == PPG_PRECISION_FP32 || E->split) ? 4 : 2;
    E->KG = 64 / E->sz;
    hipDeviceProp_t prop;
    HIP_OK(hipGetDeviceProperties(&prop, device));
    E->num_cus = prop.multiProcessorCount;
    m->hidden = H; m->heads = w->heads; m->ffn = F; m->layers = L; m->taps = w->conv_kernel; m->groups = w->conv_groups;
    m->eps = w->layer_norm_eps;
    m->gemm32 = ppg::env_switch("PPGS_AMD_W2V2_GEMM32", m->gemm32) != 0;
    m->qkv32 = ppg::env_experiment("PPGS_AMD_W2V2_QKV32", m->qkv32) != 0;
    if (E->sz != 2 || H % 256 || F % 256 || H % 128 || F % 128) m->gemm32 = false;
    const int CG = H / w->conv_groups;                       // 48 channels per group
    m->gpt = (CG * E->sz + 63) / 64;                         // K-groups of 64 bytes per tap: 2 (16-bit, padded) or 3 (fp32)
    int rc;
#define NEED(ptr) if (!(ptr)) return fail(PPG_EINVAL, #ptr " is null")
    NEED(w->proj_norm_weight); NEED(w->proj_norm_bias); NEED(w->proj_weight); NEED(w->proj_bias);
    NEED(w->pos_conv_weight); NEED(w->pos_conv_bias); NEED(w->enc_norm_weight); NEED(w->enc_norm_bias);
    auto paired = [&](const float* src, int rows, int cols, char** dst) {
        return upload_matrix(E, rows, cols, rows, cols, [&](int r, int c) { return src[(size_t)pair_row(r) * cols + c]; }, dst);
    };
    if ((rc = upload_f32(E, w->proj_norm_weight, 512, 0, &m->pn_g))) return rc;
    if ((rc = upload_f32(E, w->proj_norm_bias, 512, 0, &m->pn_b))) return rc;
    if ((rc = paired(w->proj_weight, H, 512, &m->proj_w))) return rc;
    // [N / 256][wave][K / 128][rb][8 K-steps] fragments: lane l = (row phi(l & 31) of the block, k 8 (l >> 5) .. + 7)
    auto phi = [](int rho) { return 16 * ((rho >> 2) & 1) + 4 * (rho >> 3) + (rho & 3); };
    auto image = [&](const float* src, int N, int K, char** dst) {
        const int chunks = K / 128, frags = (N / 256) * 4 * chunks * 16;
        return upload_matrix(E, frags * 64, 8, frags * 64, 8,
                             [&](int r, int j) {
                                 const int f = r >> 6, ln = r & 63;
                                 const int ks = f & 7, rb = (f >> 3) & 1, c = (f >> 4) % chunks, wv = ((f >> 4) / chunks) & 3, p = (f >> 4) / chunks / 4;
                                 const int n = 256 * p + 64 * wv + 32 * rb + phi(ln & 31), k = 128 * c + 16 * ks + 8 * (ln >> 5) + j;
                                 return src[(size_t)n * K + k];
                             }, dst);
    };
    if (m->gemm32 && (rc = image(w->proj_weight, H, 512, &m->proj_img))) return rc;
    if ((rc = upload_f32(E, w->proj_bias, H, 0, &m->proj_b))) return rc;
    {   // W'[n][tap * gpt * KG + c] = w[n][c][tap] for c < 48 (n's own group), 0 for the pad channels; plain row order
        // (fp16x2: as plain fp32 -- this one GEMM runs on the f32-input MFMAs)
        E->split = false;
        const int taps = m->taps, kk = m->gpt * E->KG;
        const float* pw = w->pos_conv_weight;
        rc = upload_matrix(E, H, taps * kk, H, taps * kk,
                           [&](int n, int col) { const int tap = col / kk, c = col - tap * kk; return c < CG ? pw[((size_t)n * CG + c) * taps + tap] : 0.f; },
                           &m->pos_w);
        if (rc) return rc;
    }
    E->split = precision == PPG_PRECISION_FP16X2;
    if ((rc = upload_f32(E, w->pos_conv_bias, H, 0, &m->pos_b))) return rc;
    m->posconv = ppg::env_switch("PPGS_AMD_W2V2_POSCONV", m->posconv) != 0;
    if (E->sz != 2 || w->conv_groups != 16 || CG != 48 || m->taps != 128) m->posconv = false;
    if (m->posconv) {
        const float* pw = w->pos_conv_weight;
        const int frags = 16 * 4 * 32 * 6;
        rc = upload_matrix(E, frags * 64, 8, frags * 64, 8,
                           [&](int r, int j) {
                               const int f = r >> 6, ln = r & 63;
                               const int rb = f & 1, ks = (f % 6) >> 1, tl = (f / 6) & 31, wv = (f / 192) & 3, g = f / 768;
                               const int n = 32 * rb + phi(ln & 31), ch = 16 * ks + 8 * (ln >> 5) + j, tap = 32 * wv + tl;
                               return n < CG ? pw[((size_t)(g * CG + n) * CG + ch) * 128 + tap] : 0.f;
                           }, &m->pos_img);
        if (rc) return rc;
    }
    if ((rc = upload_f32(E, w->enc_norm_weight, H, 0, &m->en_g))) return rc;
    if ((rc = upload_f32(E, w->enc_norm_bias, H, 0, &m->en_b))) return rc;
    m->layer.resize(L);
    for (int l = 0; l < L; ++l) {
        const PpgW2v2LayerWeights& lw = w->layers[l];
        PpgW2v2Body::Layer& d = m->layer[l];
        const float qscale = (float)(1.4426950408889634 / sqrt(64.0));      // 12 heads of 64
        NEED(lw.q_weight); NEED(lw.q_bias); NEED(lw.k_weight); NEED(lw.k_bias); NEED(lw.v_weight); NEED(lw.v_bias);
        NEED(lw.out_weight); NEED(lw.out_bias); NEED(lw.norm1_weight); NEED(lw.norm1_bias);
        NEED(lw.ffn1_weight); NEED(lw.ffn1_bias); NEED(lw.ffn2_weight); NEED(lw.ffn2_bias); NEED(lw.norm2_weight); NEED(lw.norm2_bias);
        // in_proj = [q; k; v] rows, each block of 3H in paired order (linear_kernel<EPI_QKV>)
        rc = upload_matrix(E, 3 * H, H, 3 * H, H,
                           [&](int r, int c) {
                               const int rr = pair_row(r), which = rr / H, row = rr - which * H;
                               const float* src = which == 0 ? lw.q_weight : (which == 1 ? lw.k_weight : lw.v_weight);
                               return src[(size_t)row * H + c] * (which == 0 ? qscale : 1.0f);    // (see ppg_engine_create)
                           }, &d.wqkv);
        if (rc) return rc;
        std::vector<float> bq(3 * (size_t)H);
        for (int i = 0; i < H; ++i) bq[i] = lw.q_bias[i] * qscale;
        memcpy(bq.data() + H, lw.k_bias, H * sizeof(float));
        memcpy(bq.data() + 2 * H, lw.v_bias, H * sizeof(float));
        if ((rc = upload_f32(E, bq.data(), 3 * (size_t)H, 0, &d.bqkv))) return rc;
        if ((rc = paired(lw.out_weight, H, H, &d.wo))) return rc;
        if ((rc = upload_f32(E, lw.out_bias, H, 0, &d.bo))) return rc;
        if ((rc = upload_f32(E, lw.norm1_weight, H, 0, &d.g1))) return rc;
        if ((rc = upload_f32(E, lw.norm1_bias, H, 0, &d.e1))) return rc;
        if ((rc = paired(lw.ffn1_weight, F, H, &d.w1))) return rc;
        if ((rc = upload_f32(E, lw.ffn1_bias, F, 0, &d.b1))) return rc;
        if ((rc = paired(lw.ffn2_weight, H, F, &d.w2))) return rc;
        if ((rc = upload_f32(E, lw.ffn2_bias, H, 0, &d.b2))) return rc;
        if ((rc = upload_f32(E, lw.norm2_weight, H, 0, &d.g2))) return rc;
        if ((rc = upload_f32(E, lw.norm2_bias, H, 0, &d.e2))) return rc;
        d.wo_img = d.w1_img = d.w2_img = d.wqkv_img = nullptr;
        if (m->gemm32) {
            // Q | K | V as one image of 3H / 256 passes: Q (scaled as above) and K rows in accumulator order phi, the V
            // passes' rows in pair_row order (their accumulators come out transposed: ppg_gemm32.hip mode 3)
            const int chunks = H / 128, frags = (3 * H / 256) * 4 * chunks * 16;
            rc = upload_matrix(E, frags * 64, 8, frags * 64, 8,
                               [&](int r, int j) {
                                   const int f = r >> 6, ln = r & 63;
                                   const int ks = f & 7, rb = (f >> 3) & 1, c = (f >> 4) % chunks, wv = ((f >> 4) / chunks) & 3, p = (f >> 4) / chunks / 4;
                                   const int k = 128 * c + 16 * ks + 8 * (ln >> 5) + j;
                                   const int vp0 = 2 * H / 256;
                                   if (p >= vp0) return lw.v_weight[(size_t)pair_row(256 * (p - vp0) + 64 * wv + 32 * rb + (ln & 31)) * H + k];
                                   const int n = 256 * p + 64 * wv + 32 * rb + phi(ln & 31);
                                   return n < H ? lw.q_weight[(size_t)n * H + k] * qscale : lw.k_weight[(size_t)(n - H) * H + k];
                               }, &d.wqkv_img);
            if (rc) return rc;
            if ((rc = image(lw.out_weight, H, H, &d.wo_img))) return rc;
            if ((rc = image(lw.ffn1_weight, F, H, &d.w1_img))) return rc;
            if ((rc = image(lw.ffn2_weight, H, F, &d.w2_img))) return rc;
        }
    }
#undef NEED
    for (PpgW2v2Body::Slot& sl : m->slot) HIP_OK(hipEventCreateWithFlags(&sl.uploaded, hipEventDisableTiming));
    HIP_OK(hipStreamCreateWithFlags(&m->side, hipStreamNonBlocking));
    HIP_OK(hipEventCreateWithFlags(&m->ev_fork, kForkJoinEventFlags));
    HIP_OK(hipEventCreateWithFlags(&m->ev_join, kForkJoinEventFlags));
    m->pipelines = std::max(1, std::min(ppg::env_switch("PPGS_AMD_W2V2_STREAMS", m->pipelines), 2));
    *out = m.release();
    return PPG_OK;
}

void ppg_w2v2_body_destroy(PpgW2v2Body* body) { delete body; }

namespace {
struct BodyLayout { size_t win, blk, items, ln, x, p, xb, qk, vt, ao, hid, total; int R, M, nitems; };
BodyLayout body_layout(const PpgW2v2Body* m, int batch, int frames) {
    BodyLayout L{};
    const int H = m->hidden, sz = m->eng.sz;
    L.R = round_up(frames, 32);
    L.M = batch * L.R;
    L.nitems = batch * ((frames + 63) / 64);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    L.win = take((size_t)batch * sizeof(PpgWindow));
    L.blk = take((size_t)(L.M / 16) * sizeof(int));
    L.items = take((size_t)L.nitems * sizeof(AttnItem));
    L.ln = take(((size_t)L.M + 1) * 512 * sz);
    L.x = take((size_t)L.M * H * 4);
    L.p = take((size_t)L.M * H * 4);
    L.xb = take(((size_t)L.M + 1) * H * sz + 256);            // (the last group's pad channels read 32 bytes past a row)
    L.qk = take(((size_t)L.M + 64) * 2 * H * sz);
    L.vt = take((size_t)H * (L.M + 64) * sz);
    L.ao = take((size_t)L.M * H * sz);
    L.hid = take((size_t)L.M * m->ffn * sz);
    L.total = off;
    return L;
}
}  // namespace

namespace {
// Items of the first pipeline when the batch is split (0: one pipeline).  Items are independent (one attention
// window each); the projections' 128-row tiles of 8 192 rows are 192 workgroups on 256 CUs, and two half-batches on
// two streams run one half's GEMMs beside the other half's attention and LayerNorm launches.
int body_first_half(const PpgW2v2Body* m, int batch, int frames) {
    const int R = (frames + 31) / 32 * 32;
    if (m->pipelines < 2 || batch < 8 || (long)batch * R < 4096) return 0;
    return (batch + 1) / 2;
}
int body_forward_one(PpgW2v2Body* m, PpgW2v2Body::Slot& slot, const float* features, const int64_t* valid_frames, int batch, int frames,
                     float* out, void* workspace, size_t workspace_bytes, hipStream_t s);
}  // namespace

int ppg_w2v2_body_workspace_bytes(const PpgW2v2Body* body, int batch, int frames, size_t* bytes) {
    if (!body || !bytes || batch <= 0 || frames <= 0) return fail(PPG_EINVAL, "bad argument");
    const int h = body_first_half(body, batch, frames);
    *bytes = h ? align_up(body_layout(body, h, frames).total, 256) + body_layout(body, batch - h, frames).total
               : body_layout(body, batch, frames).total;
    return PPG_OK;
}

int ppg_w2v2_body_forward(PpgW2v2Body* m, const float* features, const int64_t* valid_frames, int batch, int frames,
                          float* out, void* workspace, size_t workspace_bytes, void* stream) {
    if (!m || !features || !valid_frames || !out || !workspace || batch <= 0 || frames <= 0) return fail(PPG_EINVAL, "bad argument");
    for (int b = 0; b < batch; ++b)
        if (valid_frames[b] < 1 || valid_frames[b] > frames) return fail(PPG_EINVAL, "valid_frames[%d]=%lld outside [1, %d]", b, (long long)valid_frames[b], frames);
    size_t need = 0;
    (void)ppg_w2v2_body_workspace_bytes(m, batch, frames, &need);
    if (workspace_bytes < need) return fail(PPG_EWORKSPACE, "workspace %zu bytes < required %zu", workspace_bytes, need);
    if (reinterpret_cast<uintptr_t>(workspace) % 256) return fail(PPG_EINVAL, "workspace not 256-byte aligned");
    HIP_OK(hipSetDevice(m->eng.device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int h = body_first_half(m, batch, frames);
    if (!h) return body_forward_one(m, m->slot[0], features, valid_frames, batch, frames, out, workspace, workspace_bytes, s);
    const size_t ws0 = align_up(body_layout(m, h, frames).total, 256);
    HIP_OK(hipEventRecord(m->ev_fork, s));
    HIP_OK(hipStreamWaitEvent(m->side, m->ev_fork, 0));
    int rc = body_forward_one(m, m->slot[1], features + (size_t)h * frames * 512, valid_frames + h, batch - h, frames,
                              out + (size_t)h * frames * m->hidden, static_cast<char*>(workspace) + ws0, workspace_bytes - ws0, m->side);
    if (rc) return rc;
    HIP_OK(hipEventRecord(m->ev_join, m->side));
    rc = body_forward_one(m, m->slot[0], features, valid_frames, h, frames, out, workspace, ws0, s);
    HIP_OK(hipStreamWaitEvent(s, m->ev_join, 0));
    return rc;
}

namespace {
int body_forward_one(PpgW2v2Body* m, PpgW2v2Body::Slot& slot, const float* features, const int64_t* valid_frames, int batch, int frames,
                     float* out, void* workspace, size_t workspace_bytes, hipStream_t s) {
    const BodyLayout L = body_layout(m, batch, frames);
    if (workspace_bytes < L.total) return fail(PPG_EWORKSPACE, "workspace %zu bytes < required %zu", workspace_bytes, L.total);
    PpgEngine* E = &m->eng;
    const int H = m->hidden, F = m->ffn, sz = E->sz, prec = E->cfg.precision, M = L.M, R = L.R;
    char* base = static_cast<char*>(workspace);

    // tables: one window per item, every 16-row block of item b -> window b, query tiles of 64
    const size_t table_bytes = L.ln;                           // win | blk | items are the first three regions
    if (slot.staging_bytes < table_bytes) {
        if (slot.staging) { HIP_OK(hipEventSynchronize(slot.uploaded)); (void)hipHostFree(slot.staging); slot.staging = nullptr; }
        HIP_OK(hipHostMalloc(reinterpret_cast<void**>(&slot.staging), table_bytes, hipHostMallocDefault));
        slot.staging_bytes = table_bytes;
    }
    HIP_OK(hipEventSynchronize(slot.uploaded));                  // the previous call's upload has left the staging buffer
    memset(slot.staging, 0, table_bytes);
    PpgWindow* hw = reinterpret_cast<PpgWindow*>(slot.staging + L.win);
    int* hb = reinterpret_cast<int*>(slot.staging + L.blk);
    AttnItem* hi = reinterpret_cast<AttnItem*>(slot.staging + L.items);
    int ni = 0;
    for (int b = 0; b < batch; ++b) {
        PpgWindow& w = hw[b];
        w.item = b; w.frames = frames; w.valid = (int)valid_frames[b]; w.keep_lo = 0; w.keep_hi = frames;
        w.tok_off = b * R; w.vt_off = b * R;
        for (int k = 0; k < R / 16; ++k) hb[b * (R / 16) + k] = b;
        for (int q0 = 0; q0 < frames; q0 += 64) hi[ni++] = AttnItem{b, q0, w.tok_off, w.vt_off, frames, w.valid, 0, 0};
    }
    // padding rows, slack rows / columns of the OPERAND buffers: finite (masked keys are still multiplied).  The fp32
    // residual buffers X and P (half of the bytes) are written in full by the projection / every GEMM epilogue.
    HIP_OK(hipMemsetAsync(base + L.ln, 0, L.x - L.ln, s));
    HIP_OK(hipMemsetAsync(base + L.xb, 0, L.hid - L.xb, s));
    HIP_OK(hipMemcpyAsync(base, slot.staging, table_bytes, hipMemcpyHostToDevice, s));
    HIP_OK(hipEventRecord(slot.uploaded, s));
    const PpgWindow* d_win = reinterpret_cast<const PpgWindow*>(base + L.win);
    const int* d_blk = reinterpret_cast<const int*>(base + L.blk);
    const AttnItem* d_items = reinterpret_cast<const AttnItem*>(base + L.items);
    char* ln = base + L.ln;
    float* X = reinterpret_cast<float*>(base + L.x);
    float* P = reinterpret_cast<float*>(base + L.p);
    char* Xb = base + L.xb;
    char* qk = base + L.qk;
    char* vt = base + L.vt;
    char* ao = base + L.ao;
    char* hid = base + L.hid;
    const int vt_ld = M + 64;
    // operand rows of the residual stream: the 16-bit copy (fp16x2: the [32 hi | 32 lo] copy), or X itself in fp32 mode
    const bool op_copy = sz == 2 || E->split;
    const char* act_x = op_copy ? Xb : reinterpret_cast<const char*>(X);
    char* xb_out = op_copy ? Xb : nullptr;

#define LAUNCH_OK(expr, what)                                                        \
    do {                                                                             \
        hipError_t he_ = (expr);                                                     \
        if (he_ != hipSuccess) return fail(PPG_EDEVICE, "%s: %s", what, hipGetErrorString(he_)); \
    } while (0)
    // tokens per wave (16 nt).  Measured at 16 x 499 frames, bf16: nt 1 4.41 ms, nt 2 4.82 ms, nt 3 6.35 ms
    int nt = std::min(choose_nt(E, M, 2), 2);
    nt = std::max(1, std::min(ppg::env_experiment("PPGS_AMD_W2V2_NT", nt), sz == 2 ? 3 : 2));
    auto general = [&](const char* act, int k_elems, const char* W, const float* bias, int N) {
        LinearArgs a{};
        a.blk_win = d_blk; a.win = d_win; a.M = M; a.H = H; a.v_start = INT_MAX; a.taps = 1;
        a.act = act; a.lda_bytes = k_elems * sz;
        a.groups_per_tap = k_elems / E->KG; a.real_groups = a.total_groups = a.groups_per_tap;
        a.W = W; a.bias = bias; a.N = N; a.out_ld32 = H;
        return a;
    };
    auto layer_norm = [&](const float* g, const float* b) {
        return ppg::launch_w2v2_layernorm(prec, H, P, nullptr, g, b, M, M, M, m->eps, X, xb_out, s);
    };
    // feature projection: LayerNorm(512) -> Linear, rows past the valid frames zeroed (HF: hidden_states[~mask] = 0)
    LAUNCH_OK(ppg::launch_w2v2_layernorm(prec, 512, features, nullptr, m->pn_g, m->pn_b, (long)batch * frames, frames, R, m->eps,
                                         op_copy ? nullptr : reinterpret_cast<float*>(ln), op_copy ? ln : nullptr, s), "w2v2 projection LayerNorm");
    if (m->gemm32) {
        // (linear_kernel's 16-token waves re-read the 768 x 512 weights per 64 rows: 205 us for 6.4 GFLOP)
        Gemm32Args g{};
        g.x = ln; g.w_img = m->proj_img; g.bias = m->proj_b; g.out32 = X; g.out16 = Xb; g.M = M; g.N = H; g.K = 512;
        g.win = d_win; g.rows_per_item = R;
        LAUNCH_OK(ppg::launch_gemm32(prec, g, s), "w2v2 projection");
    } else {
        LinearArgs a = general(ln, 512, m->proj_w, m->proj_b, H);
        a.zero_invalid = 1; a.out32 = X; a.out_rows = xb_out; a.out_ld = H;
        LAUNCH_OK(ppg::launch_linear(prec, EPI_GENERAL, 16, nt, a, H / 256, s), "w2v2 projection");
    }
    {   // positional convolution (+GELU) + residual -> P, then the encoder's LayerNorm
        if (m->posconv) {
            PosConvArgs pc{};
            pc.x16 = Xb; pc.ldx_bytes = H * 2; pc.w_img = m->pos_img; pc.bias = m->pos_b; pc.residual = X; pc.out32 = P;
            pc.M = M; pc.H = H; pc.rows_per_item = R; pc.frames = frames; pc.tiles_per_item = (R + 127) / 128;
            LAUNCH_OK(ppg::launch_posconv(prec, pc, batch, s), "w2v2 positional convolution");
            LAUNCH_OK(layer_norm(m->en_g, m->en_b), "w2v2 encoder LayerNorm");
        } else {
        // (fp16x2: fp32 rows of X against the fp32 weights, on the f32-input MFMAs)
        LinearArgs a = general(E->split ? reinterpret_cast<const char*>(X) : act_x, H, m->pos_w, m->pos_b, H);
        a.taps = m->taps; a.groups_per_tap = m->gpt; a.real_groups = a.total_groups = m->taps * m->gpt;
        a.act_y_stride = (H / m->groups) * sz; a.act_fn = 2; a.residual = X; a.out32 = P;
        // (32 or 48 tokens per wave -- fewer re-reads of a group's 590 KB of weights -- measured: no faster, 3.20 / 3.24
        // against 3.23 ms per forward: the launch is bound by the re-reads of the ACTIVATION rows, one pass per tap)
        int pos_nt = 1;
        pos_nt = std::max(1, std::min(ppg::env_experiment("PPGS_AMD_W2V2_POS_NT", pos_nt), sz == 2 ? 3 : 2));
        LAUNCH_OK(ppg::launch_linear(E->split ? PPG_PRECISION_FP32 : prec, EPI_GENERAL, 3, pos_nt, a, m->groups, s), "w2v2 positional convolution");
        LAUNCH_OK(layer_norm(m->en_g, m->en_b), "w2v2 encoder LayerNorm");
        }
    }
    for (int l = 0; l < m->layers; ++l) {
        const PpgW2v2Body::Layer& d = m->layer[l];
        {
            LinearArgs a = general(act_x, H, d.wqkv, d.bqkv, 3 * H);
            a.out_rows = qk; a.out_ld = 2 * H; a.vt = vt; a.vt_ld = vt_ld; a.v_start = 2 * H;
            if (m->gemm32 && m->qkv32) {
                Gemm32Args g{};
                g.x = Xb; g.w_img = d.wqkv_img; g.bias = d.bqkv; g.out16 = qk; g.M = M; g.N = 3 * H; g.K = H;
                g.vt = vt; g.vt_ld = vt_ld; g.ld_out = 2 * H; g.v_pass0 = 2 * H / 256; g.rows_per_item = R;
                LAUNCH_OK(ppg::launch_gemm32(prec, g, s), "w2v2 qkv");
            } else {
                LAUNCH_OK(ppg::launch_linear(prec, EPI_QKV, 16, nt, a, 3 * H / 256, s), "w2v2 qkv");
            }
        }
        {
            AttnArgs a{};
            a.qk = qk; a.qk_ld_bytes = 2 * H * sz; a.vt = vt; a.vt_ld_bytes = vt_ld * sz;
            a.ao = ao; a.H = H; a.causal = 0;
            a.items = d_items; a.win = d_win; a.M = M; a.ao_tiled = 0; a.heads = m->heads;
            LAUNCH_OK(ppg::launch_attn(prec, a, ni, m->heads, 64, s), "w2v2 attention");
        }
        if (m->gemm32) {
            Gemm32Args g{};
            g.x = ao; g.w_img = d.wo_img; g.bias = d.bo; g.residual = X; g.out32 = P; g.M = M; g.N = H; g.K = H;
            LAUNCH_OK(ppg::launch_gemm32(prec, g, s), "w2v2 out-proj");
            LAUNCH_OK(layer_norm(d.g1, d.e1), "w2v2 LayerNorm 1");
            Gemm32Args f1{};
            f1.x = Xb; f1.w_img = d.w1_img; f1.bias = d.b1; f1.out16 = hid; f1.M = M; f1.N = F; f1.K = H; f1.act_fn = 2;
            LAUNCH_OK(ppg::launch_gemm32(prec, f1, s), "w2v2 ffn 1");
            Gemm32Args f2{};
            f2.x = hid; f2.w_img = d.w2_img; f2.bias = d.b2; f2.residual = X; f2.out32 = P; f2.M = M; f2.N = H; f2.K = F;
            LAUNCH_OK(ppg::launch_gemm32(prec, f2, s), "w2v2 ffn 2");
            LAUNCH_OK(layer_norm(d.g2, d.e2), "w2v2 LayerNorm 2");
            continue;
        }
        {
            LinearArgs a = general(ao, H, d.wo, d.bo, H);
            a.residual = X; a.out32 = P;
            LAUNCH_OK(ppg::launch_linear(prec, EPI_GENERAL, 16, nt, a, H / 256, s), "w2v2 out-proj");
            LAUNCH_OK(layer_norm(d.g1, d.e1), "w2v2 LayerNorm 1");
        }
        {
            LinearArgs a = general(act_x, H, d.w1, d.b1, F);
            a.act_fn = 2; a.out_ld32 = F;
            if (op_copy) { a.out_rows = hid; a.out_ld = F; } else { a.out32 = reinterpret_cast<float*>(hid); }
            LAUNCH_OK(ppg::launch_linear(prec, EPI_GENERAL, 16, nt, a, F / 256, s), "w2v2 ffn 1");
            LinearArgs b = general(hid, F, d.w2, d.b2, H);
            b.residual = X; b.out32 = P;
            LAUNCH_OK(ppg::launch_linear(prec, EPI_GENERAL, 16, nt, b, H / 256, s), "w2v2 ffn 2");
            LAUNCH_OK(layer_norm(d.g2, d.e2), "w2v2 LayerNorm 2");
        }
    }
#undef LAUNCH_OK
    HIP_OK(hipMemcpy2DAsync(out, (size_t)frames * H * 4, X, (size_t)R * H * 4, (size_t)frames * H * 4, batch, hipMemcpyDeviceToDevice, s));
    return PPG_OK;
}
}  // namespace

int ppg_frontend(int device, const float* audio, int batch, int samples, void* spec, void* mel, void* stream) {
    if (!audio || (!spec && !mel)) return fail(PPG_EINVAL, "null argument");
    if (batch <= 0) return fail(PPG_EINVAL, "batch %d", batch);
    if (samples <= 432)
        return fail(PPG_EINVAL, "samples %d: reflect padding of 432 needs more than 432 samples", samples);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(PPG_EDEVICE, "no HIP device: the PPG frontend has no CPU path");
    Frontend* f = nullptr;
    int rc = frontend_for(device, &f);
    if (rc) return rc;
    HIP_OK(hipSetDevice(device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    EventPair ev{};
    bool on = false;
    if (f->profiling) {
        if (f->events_used == f->events.size()) {
            EventPair p;
            HIP_OK(hipEventCreateWithFlags(&p.a, kTimingEventFlags));
            HIP_OK(hipEventCreateWithFlags(&p.b, kTimingEventFlags));
            f->events.push_back(p);
        }
        ev = f->events[f->events_used++];
        on = hipEventRecord(ev.a, s) == hipSuccess;
    }
    if ((double)batch * 513.0 * (double)(samples / 160) >= 4294967296.0)
        return fail(PPG_EINVAL, "frontend: batch %d x 513 bins x %d frames does not fit the kernel's 32-bit output index", batch, samples / 160);
    hipError_t he = ppg::launch_frontend(f->tb, audio, batch, samples, spec, mel, s);
    if (on) (void)hipEventRecord(ev.b, s);
    if (f->tb.dbg) {
        static int dumps = 0;
        unsigned long long h[64];
        if (dumps++ < 2 && hipStreamSynchronize(s) == hipSuccess &&
            hipMemcpy(h, f->tb.dbg, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
            for (int w = 0; w < 4; ++w) {
                fprintf(stderr, "frontend wave %d (group 2 of workgroup 0):", w);
                for (int k = 1; k < 7; ++k) fprintf(stderr, " [%d] %lld", k, (long long)(h[w * 16 + k] - h[w * 16]));
                fprintf(stderr, "\n");
            }
        }
    }
    if (he != hipSuccess) return fail(PPG_EDEVICE, "frontend: %s", hipGetErrorString(he));
    return PPG_OK;
}

int ppg_engine_nonfinite(PpgEngine* e, int clear, int* flag) {
    if (!e || !flag) return fail(PPG_EINVAL, "null argument");
    if (!e->d_overflow) { *flag = 0; return PPG_OK; }
    std::lock_guard<std::mutex> lock(e->mu);
    HIP_OK(hipSetDevice(e->device));
    unsigned value = 0;
    HIP_OK(hipMemcpy(&value, e->d_overflow, sizeof(value), hipMemcpyDeviceToHost));      // (synchronises with the device)
    if (value && clear) HIP_OK(hipMemset(e->d_overflow, 0, sizeof(value)));
    *flag = (int)value;
    return PPG_OK;
}

int ppg_engine_pipelines(const PpgEngine* e, int tokens) {
    if (!e || tokens < 0) return 0;
    return group_count(e, tokens);
}

int ppg_engine_profile(PpgEngine* e, int enable) {
    if (!e) return fail(PPG_EINVAL, "null engine");
    e->profiling = (unsigned)enable;
    // a first pool of event pairs per enabled class, so that a short timed region does not pay for creating them
    if (enable) (void)hipSetDevice(e->device);
    for (int cls = 0; enable && cls < PPG_K_COUNT; ++cls) {
        if (!(e->profiling & (1u << cls))) continue;
        while (e->events[cls].size() < 32) {
            EventPair p;
            if (hipEventCreateWithFlags(&p.a, kTimingEventFlags) != hipSuccess || hipEventCreateWithFlags(&p.b, kTimingEventFlags) != hipSuccess) break;
            e->events[cls].push_back(p);
        }
    }
    return PPG_OK;
}

int ppg_engine_profile_reset(PpgEngine* e) {
    if (!e) return fail(PPG_EINVAL, "null engine");
    for (auto& u : e->events_used) u = 0;
    for (auto& q : e->launch_seq) q = 0;
    return PPG_OK;
}

int ppg_engine_profile_stride(PpgEngine* e, int stride) {
    if (!e || stride < 1) return fail(PPG_EINVAL, "profile stride must be >= 1");
    e->profile_stride = stride;
    return PPG_OK;
}

int ppg_engine_profile_read(PpgEngine* e, int cls, double* total_ms, int64_t* launches) {
    if (!e || cls < 0 || cls >= PPG_K_COUNT || !total_ms || !launches) return fail(PPG_EINVAL, "bad argument");
    double total = 0;
    for (size_t i = 0; i < e->events_used[cls]; ++i) {
        HIP_OK(hipEventSynchronize(e->events[cls][i].b));
        float ms = 0;
        HIP_OK(hipEventElapsedTime(&ms, e->events[cls][i].a, e->events[cls][i].b));
        total += ms;
    }
    *total_ms = total;
    *launches = (int64_t)e->events_used[cls];
    return PPG_OK;
}

int ppg_frontend_profile(int device, int enable) {
    Frontend* f = nullptr;
    int rc = frontend_for(device, &f);
    if (rc) return rc;
    f->profiling = enable != 0;
    f->events_used = 0;
    return PPG_OK;
}

int ppg_frontend_profile_read(int device, double* total_ms, int64_t* launches) {
    if (!total_ms || !launches) return fail(PPG_EINVAL, "null argument");
    Frontend* f = nullptr;
    int rc = frontend_for(device, &f);
    if (rc) return rc;
    double total = 0;
    for (size_t i = 0; i < f->events_used; ++i) {
        HIP_OK(hipEventSynchronize(f->events[i].b));
        float ms = 0;
        HIP_OK(hipEventElapsedTime(&ms, f->events[i].a, f->events[i].b));
        total += ms;
    }
    *total_ms = total;
    *launches = (int64_t)f->events_used;
    return PPG_OK;
}

}  // extern "C"
