// Output convolution + mask + softmax of the 16-bit modes at hidden 256 (reference ppgs/model/transformer.py:102-105:
// Conv1d k5 256 -> 40 'same' over each window's rows, x mask, slice of the kept rows, cat; ppgs/core.py:586 softmax):
//     logits[m][n] = sum_{tap, c} W[n][tap][c] x[m + tap - 2][c] + b[n]        n < 40 (48 MFMA rows), K = 1280
// A GEMM with N = 48: nothing to split over features, and linear_kernel<EPI_OUTCONV> (which this replaces for these
// shapes) re-stages the 123 KB of weights through LDS once per 64 tokens -- 24 us for 4 GFLOP.  Here a persistent
// workgroup per CU keeps ALL the weights in LDS (48 rows of 2592 bytes) for the whole launch and walks 64-token tiles:
// the tile's 68 rows of x (2 halo rows either side) go global -> registers -> LDS one tile ahead, a wave takes 16
// tokens, A fragments (weights) and B fragments (the token rows of one tap) are both ds_read_b128 and conflict-free:
// either pitch is 8 banks past a multiple of 64, so the 16 rows of a fragment read fall on different 16-byte bank
// groups for that instruction's lane groups.  Taps that leave a token's window are zeroed per lane (only in 16-token
// blocks that touch a window edge).  Epilogue as linear_kernel's: bias, mask, softmax over the 40 phonemes (per-lane
// partials + shuffles over the 4 lane groups), kept rows scattered to the (B, 40, T) output.
#include "ppg_lds.h"
#include "ppg_launch.h"

#include <utility>

#ifdef PPG_ONLY_BF16
#define PPG_OTHER_PRECISIONS 0
#else
#define PPG_OTHER_PRECISIONS 1
#endif

namespace {

constexpr int OT = 64;                  // tokens per tile (16 per wave)
constexpr int HALO = 2;
constexpr int NB = 3;                   // 16-feature blocks
constexpr int KTAP = 256;               // channels per tap
constexpr int KSTEPS = 5 * KTAP / 32;   // 40 MFMA K-steps
constexpr int WROW = 5 * KTAP * 2;      // bytes of one weight row
constexpr int WP = WROW + 32;           // pitch in LDS: 648 dwords = 8 mod 64
constexpr int TP = KTAP * 2 + 32;       // tile row pitch: 136 dwords = 8 mod 64
constexpr int L_TILE = NB * 16 * WP;
constexpr int LDS_BYTES = L_TILE + (OT + 2 * HALO) * TP;
static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");
constexpr int ROW_CHUNKS = KTAP * 2 / 16;                           // 16-byte pieces of a tile row
constexpr int STG = ((OT + 2 * HALO) * ROW_CHUNKS + 255) / 256;     // pieces per thread and tile

template <class P>
__global__ __launch_bounds__(256) void outconv_kernel(LinearArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int idx = lane & 15, g = lane >> 4;

    float bias[NB][4];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const float4 bv = *reinterpret_cast<const float4*>(a.bias + nb * 16 + 4 * g);
        bias[nb][0] = bv.x; bias[nb][1] = bv.y; bias[nb][2] = bv.z; bias[nb][3] = bv.w;
    }

    // rows m0 - 2 .. m0 + 65 of a tile, 16 bytes per thread and piece (rows outside [0, M) re-read an end row:
    // no token's window reaches them)
    const int ntiles = (a.M + OT - 1) / OT;
    u32x4 stg[STG];
    auto fetch = [&](int tile) {
#pragma unroll
        for (int j = 0; j < STG; ++j) {
            const int i = min(tid + 256 * j, (OT + 2 * HALO) * ROW_CHUNKS - 1);
            const int r = i / ROW_CHUNKS, piece = i - r * ROW_CHUNKS;
            const int m = min(max(tile * OT - HALO + r, 0), a.M - 1);
            stg[j] = *reinterpret_cast<const u32x4*>(a.act + (size_t)m * a.lda_bytes + piece * 16);
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int j = 0; j < STG; ++j) {
            const int i = tid + 256 * j;
            if (i < (OT + 2 * HALO) * ROW_CHUNKS) {
                const int r = i / ROW_CHUNKS, piece = i - r * ROW_CHUNKS;
                *reinterpret_cast<u32x4*>(smem + L_TILE + r * TP + piece * 16) = stg[j];
            }
        }
    };
    // a token's window record is two dependent loads (blk_win, then win[]): both are issued a phase ahead of their
    // use -- for the first tile around the weight staging, then one tile ahead
    struct Meta { TokMeta tm; int keep_lo, keep_hi, out_frame, item; };
    auto lookup_block = [&](int tile) {
        const int m = tile * OT + 16 * wave + idx;
        return m < a.M ? a.blk_win[m >> 4] : -1;
    };
    auto lookup_window = [&](int tile, int w) {
        Meta r;
        const PpgWindow win = a.win[max(w, 0)];
        r.tm.w = w;
        r.tm.tt = w >= 0 ? tile * OT + 16 * wave + idx - win.tok_off : 0;
        r.tm.frames = w >= 0 ? win.frames : 0;
        r.tm.valid = w >= 0 ? win.valid : 0;
        r.keep_lo = win.keep_lo; r.keep_hi = win.keep_hi; r.out_frame = win.out_frame; r.item = win.item;
        return r;
    };
    Meta cur{};
    int wcur = -1;
    if ((int)blockIdx.x < ntiles) { fetch(blockIdx.x); wcur = lookup_block(blockIdx.x); }

    // all weights, once: 30 16-byte pieces per thread, requested 15 at a time (a load -> store loop pays the L2
    // round trip per piece: 20 us of a 27 us launch)
    constexpr int WPIECES = NB * 16 * (WROW / 16), WBATCH = 15;
    static_assert(WPIECES == 2 * WBATCH * 256, "two batches");
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        u32x4 wv[WBATCH];
#pragma unroll
        for (int j = 0; j < WBATCH; ++j) {
            const int i = tid + 256 * (half * WBATCH + j);
            const int row = i / (WROW / 16), piece = i - row * (WROW / 16);
            wv[j] = *reinterpret_cast<const u32x4*>(a.W + (size_t)row * a.total_groups * 64 + piece * 16);
        }
        if (half == 1 && (int)blockIdx.x < ntiles) cur = lookup_window(blockIdx.x, wcur);
#pragma unroll
        for (int j = 0; j < WBATCH; ++j) {
            const int i = tid + 256 * (half * WBATCH + j);
            const int row = i / (WROW / 16), piece = i - row * (WROW / 16);
            *reinterpret_cast<u32x4*>(smem + row * WP + piece * 16) = wv[j];
        }
    }
    const uint32_t lds0 = lds_addr(smem);
    uint32_t waddr[NB];                                                                   // + ks * 64
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) waddr[nb] = lds0 + (uint32_t)((nb * 16 + idx) * WP + g * 16);
    const uint32_t xaddr = lds0 + (uint32_t)(L_TILE + (16 * wave + idx) * TP + g * 16);   // + tap * TP + (ks % 8) * 64
    // Order of a tile's memory operations (vmcnt retires in order, loads and stores alike): the rows of tile i + 1
    // are written to LDS right after the MFMAs of tile i and BEFORE its output stores, the rows of tile i + 2 are
    // requested after them -- a wait for rows then only ever waits for stores that are a whole tile old.
    const int step = gridDim.x;
    if ((int)blockIdx.x < ntiles) {
        stash();
        if ((int)blockIdx.x + step < ntiles) fetch(blockIdx.x + step);
    }
    __syncthreads();                           // weights and the first tile are in place
    for (int tile = blockIdx.x; tile < ntiles; tile += step) {
        const bool more = tile + step < ntiles;
        const int wnext = more ? lookup_block(tile + step) : -1;
        const TokMeta tm = cur.tm;
        // taps inside the token's window, as a mask; blocks without an edge skip the per-lane selects
        unsigned taps_ok = 0;
#pragma unroll
        for (int d = 0; d < 5; ++d) {
            const int st = tm.tt + d - HALO;
            if (tm.w >= 0 && st >= 0 && st < tm.frames) taps_ok |= 1u << d;
        }
        const bool edge = __any(taps_ok != 0x1fu);

        f32x4 acc[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
        // 160 fragment reads (per K-step the token rows' B fragment, then the three weight blocks' A fragments)
        // through a ring of RING registers, issued that far ahead of the MFMA that consumes them: left to the
        // compiler every MFMA waited for the read issued just before it -- one wave per SIMD, nothing to hide it
        constexpr int NFRAG = 4 * KSTEPS, RING = 12;
        u32x4 ring[RING];
        auto issue = [&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int ks = i / 4, j = i % 4, per_tap = KTAP / 32;
            if constexpr (j == 0) ds_read_b128_asm<(ks / per_tap) * TP + (ks % per_tap) * 64>(ring[i % RING], xaddr);
            else ds_read_b128_asm<ks * 64>(ring[i % RING], waddr[j - 1]);
        };
        u32x4 bcur = u32x4{0u, 0u, 0u, 0u};
        [&]<int... I>(std::integer_sequence<int, I...>) { (issue(std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, RING>{});
        [&]<int... I>(std::integer_sequence<int, I...>) {
            ([&] {
                constexpr int ks = I / 4, j = I % 4, tap = ks / (KTAP / 32);
                constexpr int behind = NFRAG - 1 - I < RING - 1 ? NFRAG - 1 - I : RING - 1;
                lgkm_wait<behind>(ring[I % RING]);
                if constexpr (j == 0) {
                    bcur = ring[I % RING];
                    if (edge && !((taps_ok >> tap) & 1u)) bcur = u32x4{0u, 0u, 0u, 0u};
                } else {
                    P::mma(acc[j - 1], ring[I % RING], bcur);
                }
                if constexpr (I + RING < NFRAG) issue(std::integral_constant<int, I + RING>{});
            }(), ...);
        }(std::make_integer_sequence<int, NFRAG>{});

        __syncthreads();                       // every wave has read this tile
        if (more) stash();
        Meta nxt{};
        if (more) nxt = lookup_window(tile + step, wnext);
        // logits = (conv + bias) * mask; per-frame softmax over the out_C phonemes
        const bool live = tm.w >= 0 && tm.tt < tm.frames;
        const bool valid = live && tm.tt < tm.valid;
        float v[NB][4];
        float mx = -INFINITY, chk = 0.f;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = nb * 16 + 4 * g + r;
                v[nb][r] = valid ? acc[nb][r] + bias[nb][r] : 0.f;
                if (n < a.out_C) mx = fmaxf(mx, v[nb][r]);
                if (n < a.out_C) chk = fmaf(v[nb][r], 0.f, chk);         // NaN as soon as one logit is NaN or infinite
            }
        if (chk != chk && a.overflow) atomicOr(a.overflow, 1u);          // the engine's sticky flag (see LinearArgs)
        if (a.softmax) {
            mx = wave_max_g(mx);
            float sum = 0.f;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = nb * 16 + 4 * g + r;
                    const float e = (n < a.out_C) ? expf(v[nb][r] - mx) : 0.f;
                    v[nb][r] = e;
                    sum += e;
                }
            sum = wave_sum_g(sum);
            const float inv = 1.0f / sum;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[nb][r] *= inv;
        }
        if (live) {
            if (tm.tt >= cur.keep_lo && tm.tt < cur.keep_hi) {
                const int frame = cur.out_frame + (tm.tt - cur.keep_lo);
                float* o = a.out + (size_t)cur.item * a.out_C * a.out_T + frame;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int n = nb * 16 + 4 * g + r;
                        if (n < a.out_C) o[(size_t)n * a.out_T] = v[nb][r];
                    }
            }
        }
        if (tile + 2 * step < ntiles) fetch(tile + 2 * step);
        __syncthreads();                       // the next tile is in place
        cur = nxt;
    }
}

}  // namespace

namespace ppg {

bool outconv_supported(int precision, const LinearArgs& a) {
    return precision != PPG_PRECISION_FP32 && a.taps == 5 && a.lda_bytes == KTAP * 2 && a.real_groups == 5 * KTAP / 32 &&
           a.total_groups * 64 >= WROW && a.N == NB * 16 && a.out_C <= NB * 16;
}

hipError_t launch_outconv(int precision, const LinearArgs& a, hipStream_t s) {
    if (!outconv_supported(precision, a)) return hipErrorInvalidValue;
    static LdsLimit limit[2];
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
    }
    const int ntiles = (a.M + OT - 1) / OT;
    // every workgroup the same number of tiles, +-1
    const int rounds = (ntiles + cus - 1) / cus;
    const int grid = (ntiles + rounds - 1) / rounds;
    if (precision == PPG_PRECISION_BF16) {
        const hipError_t e = limit[0].ensure(reinterpret_cast<const void*>(outconv_kernel<PrecBF16>), LDS_BYTES);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(outconv_kernel<PrecBF16>, dim3(grid), dim3(256), LDS_BYTES, s, a);
    } else {
#if PPG_OTHER_PRECISIONS
        const hipError_t e = limit[1].ensure(reinterpret_cast<const void*>(outconv_kernel<PrecF16>), LDS_BYTES);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(outconv_kernel<PrecF16>, dim3(grid), dim3(256), LDS_BYTES, s, a);
#else
        return hipErrorInvalidValue;
#endif
    }
    return hipGetLastError();
}

}  // namespace ppg
