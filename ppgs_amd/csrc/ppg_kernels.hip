// gfx950 kernels of the PPG encoder: window gather, generic "weights-from-LDS"
// linear/conv kernel with fused epilogues, fused FFN (attention: ppg_attn.hip).
// See ppg_device.h for the operand orientation shared by all of them.
#include "ppg_device.h"
#include "ppg_launch.h"

#include <stdlib.h>
#include <string.h>
#include "ppg_lds.h"

#include <limits.h>
#include <type_traits>

namespace {

constexpr float kLnEps = 1e-5f;

// ---------------------------------------------------------------------------
// acc <- LayerNorm(acc + bias + X[m][:]), acc in the transposed C layout
// (lane: token = tok0 + 16t + idx, features pair_feature(nb, g) + r).
// MODE LN_EPILOGUE: result to X (fp32) and Xb (bf16 copy, bf16 mode).
// MODE LN_KEEP: result stays in acc, nothing is stored (the fused FFN's LN1:
//   x1 is both the next GEMM's operand and, left in the accumulators that the
//   FFN then adds to, the second residual).
// MODE LN_NO_RESIDUAL: acc already contains the residual; stores like LN_EPILOGUE.
// MODE LN_NO_RESIDUAL_KEEP: same input, result to X only and kept in acc (the
//   next layer's Q/K/V projection follows in the same kernel, no bf16 copy needed).
// lnp = [bias | gamma | beta], H floats each, in LDS: read from global per
// (feature block, token block) these were 60 % of the epilogue's memory
// instructions, all queueing behind the residual loads and the stores in the
// CU's one address unit.  Loops run feature-block-outer so each parameter
// vector is fetched once for all NT token blocks.
// ---------------------------------------------------------------------------
enum { LN_EPILOGUE = 0, LN_KEEP = 1, LN_NO_RESIDUAL = 2, LN_NO_RESIDUAL_KEEP = 3 };
template <class P, int NB, int NT, int MODE>
__device__ __forceinline__ void resln(
    f32x4 (&acc)[NB][NT], const float* lnp, float* X, char* Xb, int H,
    int tok0, int M, int idx, int g)
{
    bool ok[NT];
    float* xrow[NT];
    float sum[NT], mean[NT], rstd[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int m = tok0 + 16 * t + idx;
        ok[t] = m < M;
        xrow[t] = X + (size_t)(ok[t] ? m : 0) * H;
        sum[t] = 0.f;
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int n = pair_feature(nb, g);
        const float4 bv = *reinterpret_cast<const float4*>(lnp + n);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float4 rv = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (MODE != LN_NO_RESIDUAL && MODE != LN_NO_RESIDUAL_KEEP) {
                if (ok[t]) rv = *reinterpret_cast<const float4*>(xrow[t] + n);
            }
            acc[nb][t][0] += bv.x + rv.x;
            acc[nb][t][1] += bv.y + rv.y;
            acc[nb][t][2] += bv.z + rv.z;
            acc[nb][t][3] += bv.w + rv.w;
            sum[t] += (acc[nb][t][0] + acc[nb][t][1]) + (acc[nb][t][2] + acc[nb][t][3]);
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        mean[t] = wave_sum_g(sum[t]) / (float)H;
        float sq = 0.f;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float d = acc[nb][t][r] - mean[t];
                sq += d * d;
            }
        }
        const float var = wave_sum_g(sq) / (float)H;
        rstd[t] = 1.0f / sqrtf(var + kLnEps);
    }
    PairStore<P> pair[NT];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int n = pair_feature(nb, g);
        const float4 gv = *reinterpret_cast<const float4*>(lnp + H + n);
        const float4 ev = *reinterpret_cast<const float4*>(lnp + 2 * H + n);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float y0 = (acc[nb][t][0] - mean[t]) * rstd[t] * gv.x + ev.x;
            const float y1 = (acc[nb][t][1] - mean[t]) * rstd[t] * gv.y + ev.y;
            const float y2 = (acc[nb][t][2] - mean[t]) * rstd[t] * gv.z + ev.z;
            const float y3 = (acc[nb][t][3] - mean[t]) * rstd[t] * gv.w + ev.w;
            if constexpr (MODE == LN_KEEP || MODE == LN_NO_RESIDUAL_KEEP) acc[nb][t] = f32x4{y0, y1, y2, y3};
            if constexpr (MODE != LN_KEEP) {
                if (ok[t]) {
                    *reinterpret_cast<float4*>(xrow[t] + n) = make_float4(y0, y1, y2, y3);
                    if constexpr ((P::kIsBF16 || P::kSplit) && MODE != LN_NO_RESIDUAL_KEEP)
                        pair[t].put(Xb + (size_t)(tok0 + 16 * t + idx) * H * P::kBytes + P::row_byte(n & ~7), nb & 1, y0, y1, y2, y3);
                }
            }
        }
    }
}

// Cooperative copy of `count` float vectors of H floats each (global, given
// as pointers) into consecutive LDS rows; the caller synchronises.
__device__ __forceinline__ void stage_params(float* lds, int H, int tid, const float* p0, const float* p1, const float* p2) {
    for (int i = tid; i < 3 * H / 4; i += 256) {
        const int v = i / (H / 4), j = i - v * (H / 4);
        const float* src = v == 0 ? p0 : (v == 1 ? p1 : p2);
        reinterpret_cast<float4*>(lds)[i] = reinterpret_cast<const float4*>(src)[j];
    }
}

// The fused FFN's LN1: acc <- LayerNorm(acc + bias + X[m][:]) in registers,
// nothing stored, plus the result packed as the next GEMM's B fragments xf
// (paired feature blocks 2p, 2p+1 = K-group p; fp32: block nb = K-group nb in
// the column order W1 was uploaded in).  The accumulators are MFMA results in
// AGPRs and must be back there for the chunk loop, whose register budget has
// no slack: one 16-token block at a time moves to VGPRs for the LayerNorm
// arithmetic, and the asm pins on the accumulators on either side keep the
// allocator from blending this code's live ranges with the GEMMs' around it
// (unpinned, it spills most of the accumulators to scratch while they are
// being computed and the chunk loop runs 15 % slower; pinning xf as well costs
// 12 % -- measured, PPG_FFN_TIMING).
// Residual row of the wave's token block t (lane: token idx, features of lane group g)
template <int NB>
__device__ __forceinline__ void load_residual_rows(float4 (&rv)[NB], const float* X, int H, int m, int M, int g) {
    const float* xrow = X + (size_t)(m < M ? m : 0) * H;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        rv[nb] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (m < M) rv[nb] = *reinterpret_cast<const float4*>(xrow + pair_feature(nb, g));
    }
}

template <class P, int NB, int NT, int XG, int NTX>
__device__ __forceinline__ void ln_keep(
    f32x4 (&acc)[NB][NT], u32x4 (&xf)[XG][NTX], const float* lnp, const float* X, int H,
    int tok0, int M, int idx, int g, float4 (&rcur)[NB] /* rows of block 0, loaded by the caller */)
{
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int t = 0; t < NT; ++t) asm volatile("" : "+a"(acc[nb][t]));
    // residual rows: block t+1 is loaded while block t is normalised
    auto load_rows = [&](int t, float4 (&rv)[NB]) { load_residual_rows<NB>(rv, X, H, tok0 + 16 * t + idx, M, g); };
    float4 rnext[NB];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t + 1 < NT) load_rows(t + 1, rnext);
        f32x4 v[NB];
        float sum = 0.f;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int n = pair_feature(nb, g);
            const float4 bv = *reinterpret_cast<const float4*>(lnp + n);
            const float4 rv = rcur[nb];
            v[nb] = acc[nb][t] + f32x4{bv.x + rv.x, bv.y + rv.y, bv.z + rv.z, bv.w + rv.w};
            sum += (v[nb][0] + v[nb][1]) + (v[nb][2] + v[nb][3]);
        }
        const float mean = wave_sum_g(sum) / (float)H;
        float sq = 0.f;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float d = v[nb][r] - mean;
                sq += d * d;
            }
        }
        const float rstd = 1.0f / sqrtf(wave_sum_g(sq) / (float)H + kLnEps);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int n = pair_feature(nb, g);
            const float4 gv = *reinterpret_cast<const float4*>(lnp + H + n);
            const float4 ev = *reinterpret_cast<const float4*>(lnp + 2 * H + n);
            v[nb][0] = (v[nb][0] - mean) * rstd * gv.x + ev.x;
            v[nb][1] = (v[nb][1] - mean) * rstd * gv.y + ev.y;
            v[nb][2] = (v[nb][2] - mean) * rstd * gv.z + ev.z;
            v[nb][3] = (v[nb][3] - mean) * rstd * gv.w + ev.w;
        }
#pragma unroll
        for (int kg = 0; kg < XG; ++kg) {
            if constexpr (P::kIsBF16) {
                xf[kg][t] = u32x4{P::pack2(v[2 * kg][0], v[2 * kg][1]), P::pack2(v[2 * kg][2], v[2 * kg][3]),
                                  P::pack2(v[2 * kg + 1][0], v[2 * kg + 1][1]), P::pack2(v[2 * kg + 1][2], v[2 * kg + 1][3])};
            } else {
                xf[kg][t] = u32x4{__float_as_uint(v[kg][0]), __float_as_uint(v[kg][1]), __float_as_uint(v[kg][2]), __float_as_uint(v[kg][3])};
            }
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            acc[nb][t] = v[nb];
            asm volatile("" : "+a"(acc[nb][t]));
        }
        if (t + 1 < NT) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) rcur[nb] = rnext[nb];
        }
    }
}

// ---------------------------------------------------------------------------
// Window gather: (B, C, T) features -> token-major Xw[M][Cp] in P::elem.
// Applies the left replicate padding of chunked windows (reference
// transformer.py:54) and zero-fills padded channels / padded token rows.
// Tile = 32 channels x 64 tokens through LDS so both sides are coalesced.
// ---------------------------------------------------------------------------
template <class P>
__global__ __launch_bounds__(256) void gather_kernel(GatherArgs a) {
    if (blockIdx.y == gridDim.y - 1) {
        // Housekeeping row: zero what no kernel writes but the attention tiles read
        // (masked, p = 0 -- but 0 x NaN would poison the sum): the odd 16-column
        // tail of windows whose padded length is not a multiple of 32, the slack
        // behind the last V^T column, and the q|k rows behind the last token.
        constexpr int kBytes = P::kBytes;
        for (int job = blockIdx.x; job <= a.nwin + 1; job += gridDim.x) {
            if (job == a.nwin + 1) {
                for (int i = threadIdx.x; i < a.qk_slack_bytes / 16; i += 256)
                    reinterpret_cast<uint4*>(a.qk_slack)[i] = make_uint4(0u, 0u, 0u, 0u);
                continue;
            }
            int col, count;
            if (job < a.nwin) {
                const PpgWindow w = a.win[job];
                // a window with an odd number of 16-token blocks: its last 32-column group is only half
                // written by the projections -- in the 16-bit modes the columns are permuted inside the
                // group (position 8g + 4e + r), so the unwritten ones are NOT the linear tail: clear the
                // whole group (this launch runs before the projections fill in their half)
                const int r16 = (w.frames + 15) & ~15, r32 = (w.frames + 31) & ~31;
                col = w.vt_off + r32 - 32;
                count = r32 != r16 ? 32 : 0;
            } else {
                col = a.vt_tokens;
                count = a.vt_ld - a.vt_tokens;
            }
            const int chunks = count * kBytes / 16;      // 16-column steps: whole 16-byte pieces
            for (int i = threadIdx.x; i < a.vt_rows * chunks; i += 256) {
                const int r = i / chunks, c = i - r * chunks;
                *reinterpret_cast<uint4*>(a.vt + ((size_t)r * a.vt_ld + col) * kBytes + c * 16) = make_uint4(0u, 0u, 0u, 0u);
            }
        }
        return;
    }
    __shared__ float tile[32][65];
    const int tok0 = blockIdx.x * 64;
    const int c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 63;   // token within tile (read phase)
    const int ty = threadIdx.x >> 6;   // 0..3
    // token row of tile position tl (row map: every 16-row block of the tile has its own place)
    auto row_of = [&](int tl) { return a.rowmap ? a.rowmap[blockIdx.x * 4 + (tl >> 4)] + (tl & 15) : tok0 + tl; };
    {
        const int m = row_of(tx);
        const TokMeta tm = tok_meta(a.blk_win, a.win, m, a.M);
        int frame = -1;
        size_t base = 0;
        if (tm.w >= 0 && tm.tt < tm.frames) {
            const PpgWindow w = a.win[tm.w];
            frame = w.chunked ? max(w.start + tm.tt - a.overlap, 0) : tm.tt;
            base = (size_t)w.item * a.C * a.T;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = c0 + ty * 8 + i;
            float v = 0.f;
            if (frame >= 0 && c < a.C) {
                const size_t off = base + (size_t)c * a.T + frame;
                v = (a.dtype == PPG_DTYPE_F16)
                        ? __half2float(reinterpret_cast<const __half*>(a.feats)[off])
                        : reinterpret_cast<const float*>(a.feats)[off];
            }
            tile[ty * 8 + i][tx] = v;
        }
    }
    __syncthreads();
    {
        const int c = threadIdx.x & 31;     // channel within tile (write phase)
        const int tq = threadIdx.x >> 5;    // 0..7
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int tl = tq * 8 + i;
            const int m = row_of(tl);
            if (m < a.M && c0 + c < a.Cp) {
                const float v = tile[c][tl];
                if constexpr (P::kSplit) {
                    char* row = a.xw + (size_t)m * a.Cp * P::kBytes + P::row_byte(c0 + c);
                    uint32_t hi, lo;
                    P::split2(v, 0.f, hi, lo);
                    *reinterpret_cast<uint16_t*>(row) = (uint16_t)hi;
                    *reinterpret_cast<uint16_t*>(row + 64) = (uint16_t)lo;
                } else {
                    typename P::elem* dst = reinterpret_cast<typename P::elem*>(a.xw) + (size_t)m * a.Cp + c0 + c;
                    *dst = P::cvt1(v);
                }
            }
        }
    }
}

__global__ void fill_kernel(float* p, size_t n, float v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}

// ---------------------------------------------------------------------------
// Generic linear / k-tap conv:  out^T[n][tok] = sum_k W[n][k] act[tok][k].
// Workgroup = 4 waves; wave owns 16*NT tokens, all NB*16 features of this
// blockIdx.y pass.  W tiles [NB*16 rows][128 B = 2 K-groups] are staged
// global -> registers -> LDS (double buffered, one barrier per tile, loads
// of tile s+1 issued before the MFMAs of tile s).
// ---------------------------------------------------------------------------
// NT <= 2 (and the 16-feature-block passes): capped at 256 registers so two
// workgroups share a CU and cover each other's LDS / barrier / issue stalls.
template <class P, int NT, int NB, int EPI>
__global__ __launch_bounds__(256, (NT <= 2 && NB <= 16) ? 2 : 1) void linear_kernel(LinearArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TILE_BYTES = NB * 16 * 128;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int idx = lane & 15;
    const int g = lane >> 4;
    const int tok0 = a.rowmap ? __builtin_amdgcn_readfirstlane(a.rowmap[blockIdx.x * 4 + wave]) : (blockIdx.x * 4 + wave) * 16 * NT;
    const int n0 = blockIdx.y * NB * 16;
    const bool swap = (EPI == EPI_QKV) && (n0 >= a.v_start);

#ifdef PPG_LIN_TIMING
    auto stamp = [&](int k) {
        if (a.dbg && tid == 0 && k < 15)
            a.dbg[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 + k] = __builtin_amdgcn_s_memtime();
    };
#else
    auto stamp = [&](int) {};
#endif
    stamp(0);
    // Window lookups (two dependent global loads) only where window positions
    // matter: the k-tap convs.  The plain projections read and write every
    // row below M -- padding rows hold finite values nobody consumes.
    constexpr bool CONV = EPI == EPI_INCONV || EPI == EPI_OUTCONV || EPI == EPI_GENERAL;
    TokMeta tm[NT];
    if constexpr (CONV) {
#pragma unroll
        for (int t = 0; t < NT; ++t) tm[t] = tok_meta(a.blk_win, a.win, tok0 + 16 * t + idx, a.M);
    }
    // transposed-V pass: window and first column (lane group 0) of each of the
    // wave's 16-token blocks, wave-uniform; looked up here so the loads are
    // long done when the epilogue needs them
    int vw[NT], vcol[NT];
    if constexpr (EPI == EPI_QKV) {
        if (swap) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int mb = tok0 + 16 * t;
                vw[t] = mb < a.M ? a.blk_win[mb >> 4] : -1;
                vcol[t] = -1;
                if (vw[t] >= 0) {
                    const int ttb = mb - a.win[vw[t]].tok_off;   // multiple of 16
                    if constexpr (P::kIsBF16 || P::kSplit) vcol[t] = a.win[vw[t]].vt_off + (ttb >> 5) * 32 + 4 * ((ttb >> 4) & 1);
                    else vcol[t] = a.win[vw[t]].vt_off + ttb;
                }
            }
        }
    }

    const int pad = a.taps >> 1;
    const char* wbase = a.W + (size_t)n0 * a.total_groups * 64;
    const int w_row_bytes = a.total_groups * 64;

    auto stage_w = [&](int s_, char* buf) {
        stage_tile<NB * 16, 128, 4>(wbase + (size_t)s_ * 128, (size_t)w_row_bytes, buf, wave, lane);
    };
    // activation fragments of tile s (K-groups 2s, 2s+1)
    auto load_act = [&](int s, u32x4 (&af)[2][NT]) {
#pragma unroll
        for (int kg = 0; kg < 2; ++kg) {
            const int gk = 2 * s + kg;
            const int tap = gk / a.groups_per_tap;
            const int kgi = gk - tap * a.groups_per_tap;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                u32x4 v = u32x4{0u, 0u, 0u, 0u};
                if constexpr (EPI == EPI_GELU) {
                    const int m = tok0 + 16 * t + idx;
                    const int mi = a.stride * m + tap;
                    if (m < a.M && mi < a.M_in && gk < a.real_groups)
                        v = *reinterpret_cast<const u32x4*>(a.act + (size_t)mi * a.lda_bytes + kgi * 64 + g * 16);
                } else if constexpr (CONV) {
                    const int st = tm[t].tt + tap - pad;
                    if (tm[t].w >= 0 && gk < a.real_groups && st >= 0 && st < tm[t].frames) {
                        const int m = tok0 + 16 * t + idx + tap - pad;
                        const char* act = a.act;
                        if constexpr (EPI == EPI_GENERAL) act += (size_t)blockIdx.y * a.act_y_stride;
                        v = *reinterpret_cast<const u32x4*>(act + (size_t)m * a.lda_bytes + kgi * 64 + g * 16);
                    }
                } else {
                    const int m = tok0 + 16 * t + idx;
                    if (m < a.M) v = *reinterpret_cast<const u32x4*>(a.act + (size_t)m * a.lda_bytes + gk * 64 + g * 16);
                }
                af[kg][t] = v;
            }
        }
    };

    f32x4 acc[NB][NT];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[nb][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int steps = a.total_groups >> 1;
    u32x4 acur[2][NT], anext[2][NT];
    stage_w(0, smem);
    load_act(0, acur);
    stamp(1);
    dma_wait_barrier();
    stamp(2);

    // SWAP (V pass of the QKV projection) exchanges the MFMA operands so the
    // accumulator comes out token-major-transposed; compile-time per loop.
    auto main_loop = [&](auto swap_tag) {
        constexpr bool SWAP = decltype(swap_tag)::value;
        for (int s = 0; s < steps; ++s) {
            const bool more = s + 1 < steps;
            if (more) {
                stage_w(s + 1, smem + ((s + 1) & 1) * TILE_BYTES);
                load_act(s + 1, anext);
            }
            const char* buf = smem + (s & 1) * TILE_BYTES;
            // fragment i = (kg, nb) = (i / NB, i % NB)
            using L = FragLayout<128, NB>;
            uint32_t fb[L::VAR];
            L::bases(lds_addr(buf), idx, g, fb);
            lds_stream<L, 2 * NB, (NB >= 8 ? 6 : 3)>(
                fb,
                [&](auto ic, const u32x4& wf) {
                    constexpr int i = decltype(ic)::value;
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        mma_kg<P, i / NB, SWAP, 0>(acc[i % NB][t], wf, acur, t, acc[i % NB][t]);
                    }
                });
            if (more) {
#pragma unroll
                for (int kg = 0; kg < 2; ++kg)
#pragma unroll
                    for (int t = 0; t < NT; ++t) acur[kg][t] = anext[kg][t];
            }
            stamp(3 + 2 * s);
            dma_wait_barrier();
            stamp(4 + 2 * s);
        }
    };
    if constexpr (EPI == EPI_QKV) {
        if (swap) main_loop(std::true_type{});
        else main_loop(std::false_type{});
    } else {
        main_loop(std::false_type{});
    }

    // ----------------------------- epilogues --------------------------------
    if constexpr (EPI == EPI_RESLN) {
        // the weight tiles are done with (the loop ended on a barrier): LDS now holds the LN parameters
        float* lnp = reinterpret_cast<float*>(smem);
        stage_params(lnp, a.H, tid, a.bias, a.gamma, a.beta);
        __syncthreads();
        resln<P, NB, NT, LN_EPILOGUE>(acc, lnp, a.X, a.Xb, a.H, tok0, a.M, idx, g);
    } else if constexpr (EPI == EPI_INCONV) {
        // y = live ? PE[tt] + (valid ? conv + bias : 0) : 0.  No control flow around
        // the loads (PE row 0 stands in for dead rows): with the loads inside
        // `if (live)` every (feature block, token block) paid its own global-load
        // latency, 33 k of this kernel's 56 k cycles per workgroup.
        bool live[NT], valid[NT], inside[NT];
        const float* perow[NT];
        PairStore<P> pair[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            live[t] = tm[t].w >= 0 && tm[t].tt < tm[t].frames;
            valid[t] = live[t] && tm[t].tt < tm[t].valid;
            inside[t] = tok0 + 16 * t + idx < a.M;
            perow[t] = a.pe + (size_t)(live[t] ? tm[t].tt : 0) * a.H;
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int n = n0 + pair_feature(nb, g);
            const float4 bv = *reinterpret_cast<const float4*>(a.bias + n);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float4 pv = *reinterpret_cast<const float4*>(perow[t] + n);
                float4 y;
                y.x = live[t] ? pv.x + (valid[t] ? acc[nb][t][0] + bv.x : 0.f) : 0.f;
                y.y = live[t] ? pv.y + (valid[t] ? acc[nb][t][1] + bv.y : 0.f) : 0.f;
                y.z = live[t] ? pv.z + (valid[t] ? acc[nb][t][2] + bv.z : 0.f) : 0.f;
                y.w = live[t] ? pv.w + (valid[t] ? acc[nb][t][3] + bv.w : 0.f) : 0.f;
                if (inside[t]) {
                    const int m = tok0 + 16 * t + idx;
                    if (a.x_tiled == 2)
                        *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(a.X) + x16_index(m, n, a.H)) = make_uint2(pack_f16x2(y.x, y.y), pack_f16x2(y.z, y.w));
                    else
                        *reinterpret_cast<float4*>(a.X + (a.x_tiled ? x32_index(m, n, a.H) : (size_t)m * a.H + n)) = y;
                    if constexpr (P::kIsBF16 || P::kSplit) pair[t].put(a.Xb + (size_t)m * a.H * P::kBytes + P::row_byte(n & ~7), nb & 1, y.x, y.y, y.z, y.w);
                }
            }
        }
    } else if constexpr (EPI == EPI_QKV) {
        // The pass's bias values go through LDS (the weight tiles are done with: the loop ended on a barrier).  Read
        // from global memory inside the store loop, every load waited -- vmcnt(0), the compiler's count across the
        // `m < M` control flow -- for the previous block's STORE to be acknowledged (tools/wait_scan.py).
        float* lbias = reinterpret_cast<float*>(smem);
        if (tid < NB * 4) reinterpret_cast<float4*>(lbias)[tid] = reinterpret_cast<const float4*>(a.bias + n0)[tid];
        __syncthreads();
        if (!swap) {
            PairStore<P> pair[NT];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int n = n0 + pair_feature(nb, g);
                const float4 bv = *reinterpret_cast<const float4*>(lbias + pair_feature(nb, g));     // once for all token blocks
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int m = tok0 + 16 * t + idx;
                    if (m >= a.M) continue;
                    pair[t].put(a.out_rows + (size_t)m * a.out_ld * P::kBytes + P::row_byte(n & ~7), nb & 1,
                                acc[nb][t][0] + bv.x, acc[nb][t][1] + bv.y,
                                acc[nb][t][2] + bv.z, acc[nb][t][3] + bv.w);
                }
            }
        } else {
            // swapped operands: lane holds tokens 4g..4g+3 of block t for tile
            // row nb*16 + idx; V^T rows stay in tile order (attn_kernel's
            // output accumulator then owns 8 consecutive head features, see
            // pair_row).  bf16: columns are permuted inside 32-token groups
            // (position 8g + 4e + r, e = parity of the 16-token block) so that
            // the PV A-fragment of attn_kernel is one 16-byte read -- and an
            // (even, odd) block pair of this wave is one 16-byte store.
            bool done_with_previous = false;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (done_with_previous) { done_with_previous = false; continue; }
                if (vw[t] < 0) continue;
                constexpr int kLast = NT - 1;
                const int tn = t < kLast ? t + 1 : t;
                const bool paired = (P::kIsBF16 || P::kSplit) && t < kLast && vw[tn] == vw[t] && (vcol[t] & 4) == 0;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const int row = n0 + nb * 16 + idx;                       // tile row
                    const float bv = lbias[pair_row(nb * 16 + idx)];
                    char* dst = a.vt + (size_t)(row - a.v_start) * a.vt_ld * P::kBytes + P::row_byte(vcol[t] + ((P::kIsBF16 || P::kSplit) ? 8 : 4) * g);
                    if constexpr (P::kSplit) {
                        if (paired) {          // both planes of the 8 columns of the (even, odd) block pair
                            uint32_t h[4], l[4];
                            P::split2(acc[nb][t][0] + bv, acc[nb][t][1] + bv, h[0], l[0]);
                            P::split2(acc[nb][t][2] + bv, acc[nb][t][3] + bv, h[1], l[1]);
                            P::split2(acc[nb][tn][0] + bv, acc[nb][tn][1] + bv, h[2], l[2]);
                            P::split2(acc[nb][tn][2] + bv, acc[nb][tn][3] + bv, h[3], l[3]);
                            *reinterpret_cast<u32x4*>(dst) = u32x4{h[0], h[1], h[2], h[3]};
                            *reinterpret_cast<u32x4*>(dst + 64) = u32x4{l[0], l[1], l[2], l[3]};
                            continue;
                        }
                    }
                    if (paired) {
                        *reinterpret_cast<u32x4*>(dst) = u32x4{
                            P::pack2(acc[nb][t][0] + bv, acc[nb][t][1] + bv), P::pack2(acc[nb][t][2] + bv, acc[nb][t][3] + bv),
                            P::pack2(acc[nb][tn][0] + bv, acc[nb][tn][1] + bv), P::pack2(acc[nb][tn][2] + bv, acc[nb][tn][3] + bv)};
                    } else {
                        store4<P>(dst, acc[nb][t][0] + bv, acc[nb][t][1] + bv, acc[nb][t][2] + bv, acc[nb][t][3] + bv);
                    }
                }
                done_with_previous = paired;
            }
        }
    } else if constexpr (EPI == EPI_GELU) {
        // exact GELU (erf form, torch.nn.functional.gelu default); W rows in paired order: 16-byte stores
        PairStore<P> pair[NT];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int n = n0 + pair_feature(nb, g);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int m = tok0 + 16 * t + idx;
                if (m >= a.M) continue;
                const f32x2 lo = gelu_erf_pair(f32x2{acc[nb][t][0], acc[nb][t][1]}), hi = gelu_erf_pair(f32x2{acc[nb][t][2], acc[nb][t][3]});
                pair[t].put(a.out_rows + (size_t)m * a.out_ld * P::kBytes + P::row_byte(n & ~7), nb & 1, lo.x, lo.y, hi.x, hi.y);
            }
        }
    } else if constexpr (EPI == EPI_GENERAL) {
        // y = act_fn(acc + bias) [zeroed past the window's valid rows] [+ residual] -> fp32 rows and / or 16-bit rows.
        // NB == 16: W rows in paired order (16-byte stores of the 16-bit copy); other NB: plain order, fp32 output only.
        PairStore<P> pair[NT];
        float* lbias = reinterpret_cast<float*>(smem);       // (as EPI_QKV: the bias through LDS)
        if (tid < NB * 4) reinterpret_cast<float4*>(lbias)[tid] = reinterpret_cast<const float4*>(a.bias + n0)[tid];
        __syncthreads();
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int n = n0 + (NB == 16 ? pair_feature(nb, g) : nb * 16 + 4 * g);
            const float4 bv = *reinterpret_cast<const float4*>(lbias + (n - n0));
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int m = tok0 + 16 * t + idx;
                if (m >= a.M) continue;
                float y[4] = {acc[nb][t][0] + bv.x, acc[nb][t][1] + bv.y, acc[nb][t][2] + bv.z, acc[nb][t][3] + bv.w};
                if (a.act_fn == 2) {
                    const f32x2 lo = gelu_erf_pair(f32x2{y[0], y[1]}), hi = gelu_erf_pair(f32x2{y[2], y[3]});
                    y[0] = lo.x; y[1] = lo.y; y[2] = hi.x; y[3] = hi.y;
                } else if (a.act_fn == 1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) y[r] = fmaxf(y[r], 0.f);
                }
                if (a.zero_invalid && !(tm[t].w >= 0 && tm[t].tt < tm[t].valid)) y[0] = y[1] = y[2] = y[3] = 0.f;
                if (a.residual) {
                    const float4 rv = *reinterpret_cast<const float4*>(a.residual + (size_t)m * a.out_ld32 + n);
                    y[0] += rv.x; y[1] += rv.y; y[2] += rv.z; y[3] += rv.w;
                }
                if (a.out32) *reinterpret_cast<float4*>(a.out32 + (size_t)m * a.out_ld32 + n) = make_float4(y[0], y[1], y[2], y[3]);
                if constexpr (NB == 16) {
                    if (a.out_rows) {
                        if constexpr (P::kIsBF16 || P::kSplit) pair[t].put(a.out_rows + (size_t)m * a.out_ld * P::kBytes + P::row_byte(n & ~7), nb & 1, y[0], y[1], y[2], y[3]);
                        else store4<P>(a.out_rows + ((size_t)m * a.out_ld + n) * P::kBytes, y[0], y[1], y[2], y[3]);
                    }
                }
            }
        }
    } else if constexpr (EPI == EPI_RELU) {
        float* lbias = reinterpret_cast<float*>(smem);       // (as EPI_QKV: no global load between the stores)
        if (tid < NB * 4) reinterpret_cast<float4*>(lbias)[tid] = reinterpret_cast<const float4*>(a.bias + n0)[tid];
        __syncthreads();
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int m = tok0 + 16 * t + idx;
            if (m >= a.M) continue;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int n = n0 + nb * 16 + 4 * g;
                const float4 bv = *reinterpret_cast<const float4*>(lbias + nb * 16 + 4 * g);
                // (split operands: the hi plane's place of the element inside its [32 hi | 32 lo] block)
                char* dst = a.out_rows + (size_t)m * a.out_ld * P::kBytes;
                if constexpr (P::kSplit) dst += P::row_byte(n); else dst += (size_t)n * P::kBytes;
                store4<P>(dst,
                          fmaxf(acc[nb][t][0] + bv.x, 0.f), fmaxf(acc[nb][t][1] + bv.y, 0.f),
                          fmaxf(acc[nb][t][2] + bv.z, 0.f), fmaxf(acc[nb][t][3] + bv.w, 0.f));
            }
        }
    } else if constexpr (EPI == EPI_OUTCONV) {
        // logits = (conv + bias) * mask; per-frame softmax over the out_C
        // phonemes: per-lane partial over (nb, r), then shuffles over g.
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const bool live = tm[t].w >= 0 && tm[t].tt < tm[t].frames;
            const bool valid = live && tm[t].tt < tm[t].valid;
            float v[NB][4];
            float mx = -INFINITY, chk = 0.f;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const float4 bv = *reinterpret_cast<const float4*>(a.bias + nb * 16 + 4 * g);
                const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = nb * 16 + 4 * g + r;
                    v[nb][r] = valid ? acc[nb][t][r] + bb[r] : 0.f;
                    if (n < a.out_C) mx = fmaxf(mx, v[nb][r]);
                    if (n < a.out_C) chk = fmaf(v[nb][r], 0.f, chk);     // NaN as soon as one logit is NaN or infinite
                }
            }
            if (chk != chk && a.overflow) atomicOr(a.overflow, 1u);
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
                const PpgWindow w = a.win[tm[t].w];
                if (tm[t].tt >= w.keep_lo && tm[t].tt < w.keep_hi) {
                    const int frame = w.out_frame + (tm[t].tt - w.keep_lo);
                    float* o = a.out + (size_t)w.item * a.out_C * a.out_T + frame;
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int n = nb * 16 + 4 * g + r;
                            if (n < a.out_C) o[(size_t)n * a.out_T] = v[nb][r];
                        }
                }
            }
        }
    }
#ifdef PPG_LIN_TIMING
    stamp(13);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp(14);
#endif
}

// ---------------------------------------------------------------------------
// Fused FFN + residual + LayerNorm:
//   X <- LN(X + W2 relu(W1 x + b1) + b2)
// The 2048-wide hidden never leaves registers: per hidden chunk, phase A
// computes h^T[hid][tok] (C layout: 4 consecutive hid per lane), which after
// ReLU/bf16 packing IS the B fragment of phase B (k-slot permutation folded
// into the host-side packing of W2, see pack_w2 in ppg_engine.hip).
// W1/W2 chunk tiles (32 KiB each) are staged global->regs->LDS.
// ---------------------------------------------------------------------------
// The kernel body for one wave.  NTB = 16-token blocks the wave OWNS (their y
// accumulators, LayerNorms, Q/K/V tail), NTA = blocks whose phase A it computes.
// ROLE 0: NTA == NTB, the plain kernel.  ROLE 1 / 2 (ffn_mixed_kernel): an
// owner of 3 blocks lets its partner wave (owner of 2) compute the phase A of
// its third block, so that both run 5 block-phases per chunk -- h of that
// block travels partner -> owner through LDS once per chunk, its x1 fragments
// owner -> partner once per launch.
// Rows m0 .. m0 + R - 1 (dense partial rows slot0 ..) of the split-hidden FFN: b2 + residual + the partial sums in
// split order, LayerNorm-2, X and its operand copy -- ffn_reduce_ln_kernel's arithmetic, R rows of one wave with all
// their loads in flight together (hidden 256: a lane owns features 4 lane .. + 3 of every row).
template <class P, int R>
__device__ __forceinline__ void reduce_ln_rows(const FfnArgs& a, int splits, int slot0, int m0, int lane, size_t stride) {
    constexpr int H = 256;
    const int n = lane * 4;
    const float4 bv = *reinterpret_cast<const float4*>(a.b2 + n);
    const float4 gv = *reinterpret_cast<const float4*>(a.gamma + n);
    const float4 ev = *reinterpret_cast<const float4*>(a.beta + n);
    float4 xv[R], p[R][8];
    // the partial rows first: their addresses do not depend on the row map (m0 may still be in flight)
#pragma unroll
    for (int j = 0; j < R; ++j)
#pragma unroll
        for (int k = 0; k < 8; ++k)      // (unconditional loads: a split past the last re-reads the last one and is not added)
            p[j][k] = *reinterpret_cast<const float4*>(a.partial + (size_t)min(k, splits - 1) * stride + (size_t)(slot0 + j) * H + n);
#pragma unroll
    for (int j = 0; j < R; ++j) xv[j] = *reinterpret_cast<const float4*>(a.X + (size_t)min(m0 + j, a.M - 1) * H + n);
#pragma unroll
    for (int j = 0; j < R; ++j) {
        float4 acc = make_float4(bv.x + xv[j].x, bv.y + xv[j].y, bv.z + xv[j].z, bv.w + xv[j].w);
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k < splits) { acc.x += p[j][k].x; acc.y += p[j][k].y; acc.z += p[j][k].z; acc.w += p[j][k].w; }
        float sum = (acc.x + acc.y) + (acc.z + acc.w);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
        const float mean = sum / (float)H;
        const float d0 = acc.x - mean, d1 = acc.y - mean, d2 = acc.z - mean, d3 = acc.w - mean;
        float sq = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) sq += __shfl_xor(sq, off);
        const float rstd = 1.0f / sqrtf(sq / (float)H + kLnEps);
        const float y0 = (acc.x - mean) * rstd * gv.x + ev.x;
        const float y1 = (acc.y - mean) * rstd * gv.y + ev.y;
        const float y2 = (acc.z - mean) * rstd * gv.z + ev.z;
        const float y3 = (acc.w - mean) * rstd * gv.w + ev.w;
        const int m = m0 + j;
        if (m >= a.M) continue;
        *reinterpret_cast<float4*>(a.X + (size_t)m * H + n) = make_float4(y0, y1, y2, y3);
        if constexpr (P::kIsBF16 || P::kSplit) store4<P>(a.Xb + (size_t)m * H * P::kBytes + P::row_byte(n), y0, y1, y2, y3);
    }
}

template <class P, int NTA, int NTB, int NBH, bool OP, bool QKV, int ROLE>
__device__ __forceinline__ void ffn_body(const FfnArgs& a, char* smem, const int tok0) {
    constexpr int NT = NTB;                         // token blocks of everything outside the chunk loop
    constexpr int NTX = NTA > NTB ? NTA : NTB;      // operand fragment sets held
    constexpr int NTL = NTA < NTB ? NTA : NTB;      // blocks whose h is produced and consumed by this wave
    static_assert(ROLE == 0 ? NTA == NTB : (ROLE == 1 ? NTA + 1 == NTB : NTA == NTB + 1), "role");
    static_assert(ROLE == 0 || OP, "the mixed tiling exists for the fused out-projection form only");
    constexpr int H = NBH * 16;
    constexpr int ROW1 = H * P::kBytes;             // W1 tile row bytes
    constexpr int XG = ROW1 / 64;                   // K-groups of x
    constexpr int HC = 32768 / ROW1;                // hidden rows per chunk
    constexpr int HB = HC / 16;                     // hidden 16-blocks per chunk
    constexpr int ROW2 = HC * P::kBytes;            // W2 tile row bytes (128 or 64)
    constexpr int HG = ROW2 / 64;                   // K-groups of phase B per chunk
    // LDS: 2 x {W1 tile 32 KiB, W2 tile 32 KiB} (DMA double buffer) + b1
    char* ldsb1 = smem + 131072;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int idx = lane & 15;
    const int g = lane >> 4;
    // gridDim.y > 1: split-hidden mode for token counts that cannot fill the
    // chip -- this workgroup sums only its 1/gridDim.y of the hidden chunks
    // and writes fp32 partial sums; ffn_reduce_ln_kernel finishes the layer.
    const int NC = a.F / HC / gridDim.y;
    const int c_base = blockIdx.y * NC;

    // LDS: W1 tiles at 0 / 32 KiB, W2 tiles at 64 / 96 KiB, b1 at 128 KiB
    // The sum over hidden chunks is order-free: workgroup b walks the chunks
    // starting at its own offset, so that the CUs of one XCD (b, b+8, b+16, ...)
    // do not all pull the same 64 KiB of weights through the same L2 lines at
    // the same moment.
    const int rot = a.rowmap ? 0 : (blockIdx.x >> 3) % NC;     // (row map = streaming: a recomputed row must give the same bits wherever it lands)
    auto hidden_chunk = [&](int c) { const int r = c + rot; return c_base + (r >= NC ? r - NC : r); };
    auto stage_w1 = [&](int c) {
        stage_tile<HC, ROW1, 4>(a.W1 + (size_t)hidden_chunk(c) * 32768, (size_t)ROW1, smem + (c & 1) * 32768, wave, lane);
    };
    auto stage_w2 = [&](int c) {
        stage_tile<H, ROW2, 4>(a.W2p + (size_t)hidden_chunk(c) * ROW2, (size_t)a.F * P::kBytes, smem + 65536 + (c & 1) * 32768, wave, lane);
    };
    // QKV: transposed-V column (lane group 0) of each of the wave's 16-token
    // blocks, wave-uniform, -1 = padding block; see linear_kernel
    int vw[NT], vcol[NT];
    if constexpr (QKV) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int mb = tok0 + 16 * t;
            vw[t] = mb < a.M ? a.blk_win[mb >> 4] : -1;
            vcol[t] = -1;
            if (vw[t] >= 0) {
                const int ttb = mb - a.win[vw[t]].tok_off;
                if constexpr (P::kIsBF16) vcol[t] = a.win[vw[t]].vt_off + (ttb >> 5) * 32 + 4 * ((ttb >> 4) & 1);
                else vcol[t] = a.win[vw[t]].vt_off + ttb;
            }
        }
    }
    // OP: the W1 double buffer first carries the H/HC tiles of W_o
    constexpr int OT = H / HC;
    static_assert(OT % 2 == 0, "W_o tiles must leave the W1 buffers in phase");
    auto stage_wo = [&](int i) {
        stage_tile<HC, ROW1, 4>(a.Wo + (size_t)i * 32768, (size_t)ROW1, smem + (i & 1) * 32768, wave, lane);
    };
    if constexpr (OP) {
        stage_wo(0);
        stage_w2(0);
        stage_wo(1);
    } else {
        stage_w1(0);
        stage_w2(0);
        if (NC > 1) stage_w1(1);
    }
    for (int i = tid; i < a.F / 4; i += 256)
        reinterpret_cast<float4*>(ldsb1)[i] = reinterpret_cast<const float4*>(a.b1)[i];
    // LayerNorm parameters behind b1: [b2 | gamma2 | beta2] and, with OP, [bo | gamma1 | beta1]
    float* lnp2 = reinterpret_cast<float*>(ldsb1) + a.F;
    float* ldsbq = lnp2 + 3 * H;          // QKV: the next layer's in_proj bias, 3H floats
    float* lnp1 = ldsbq + 3 * H;          // last: dead after LN1, the mixed tiling's h hand-off overlays it
    stage_params(lnp2, H, tid, a.b2, a.gamma, a.beta);
    if constexpr (OP) stage_params(lnp1, H, tid, a.bo, a.g1, a.e1);
    if constexpr (QKV) stage_params(ldsbq, H, tid, a.bq, a.bq + H, a.bq + 2 * H);

    // B fragments of the wave's tokens: x (bf16 copy / fp32 X), or with OP the
    // attention output, replaced by LN1's result below
    const char* actp = OP ? a.ao : ((P::kIsBF16 || P::kSplit) ? a.Xb : reinterpret_cast<const char*>(a.X));
    u32x4 xf[XG][NTX];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int m = tok0 + 16 * t + idx;
#pragma unroll
        for (int kg = 0; kg < XG; ++kg) {
            u32x4 v = u32x4{0u, 0u, 0u, 0u};
            if (m < a.M) v = *reinterpret_cast<const u32x4*>(actp + (size_t)m * ROW1 + kg * 64 + g * 16);
            xf[kg][t] = v;
        }
    }
    // hand-off areas of the mixed tiling: per wave pair (wave & 1), x1 fragments once
    // (in the second W2 buffer, free until chunk 0 stages W2(1)) and the raw fp32 h
    // accumulators of one block per chunk (double buffered; from the LN1 parameters on,
    // which are dead by then)
    const uint32_t xfer_x = lds_addr(smem) + 98304 + (wave & 1) * (XG * 1024) + lane * 16;
    const uint32_t xfer_h = lds_addr(smem) + 131072 + (uint32_t)a.F * 4 + 6 * H * 4 + (wave & 1) * (2 * HB * 1024) + lane * 16;

    f32x4 yacc[NBH][NT];
    if constexpr (!OP) {
#pragma unroll
        for (int nb = 0; nb < NBH; ++nb)
#pragma unroll
            for (int t = 0; t < NT; ++t) yacc[nb][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    using LA = FragLayout<ROW1, HB>;
    using LB = FragLayout<ROW2, NBH>;
    constexpr int RA = XG * HB;          // fragments of phase A
    constexpr int UNITS = HB * NTB;      // pack units (one hidden 16-block of one token block the wave owns)
    constexpr int DEPTH = NTX >= 3 ? 6 : 8;
    const uint32_t lds0 = lds_addr(smem);

    // pack unit u = (hb, t): bias + ReLU on the phase-A accumulator, packed
    // straight into the phase-B B-fragment (see header comment)
    // (ROLE 1: the last owned block's accumulators come from the partner wave: hfar)
    // (the bias is already in hv: it was the C operand of the block's first MFMA; in the
    // 16-bit modes the ReLU runs on the packed pair, one v_pk_max_i16 per two values)
    auto pack_unit = [&](auto uc, f32x4 (&hsrc)[HB][NTA], f32x4 (&hfar)[HB], u32x4 (&hf)[HG][NTB]) {
        constexpr int u = decltype(uc)::value;
        constexpr int hb = u / NTB, t = u % NTB;
        const f32x4 hv = [&] { if constexpr (t < NTL) return hsrc[hb][t]; else return hfar[hb]; }();
        if constexpr (P::kSplit) {
            // (a chunk is ONE 32-wide hidden group: blocks 0, 1 are the two halves of the lane's 8 k-slots; hf[0] the hi
            // plane, hf[1] the lo plane -- the 16-bit path's slot order, so W2 is packed as for the 16-bit modes)
            static_assert(HB == 2 && HG == 2, "split operands: hidden 256 (one 32-wide hidden group per chunk)");
            uint32_t h01, l01, h23, l23;
            P::split2(fmaxf(hv[0], 0.f), fmaxf(hv[1], 0.f), h01, l01);
            P::split2(fmaxf(hv[2], 0.f), fmaxf(hv[3], 0.f), h23, l23);
            if constexpr (hb & 1) { hf[0][t].z = h01; hf[0][t].w = h23; hf[1][t].z = l01; hf[1][t].w = l23; }
            else                  { hf[0][t].x = h01; hf[0][t].y = h23; hf[1][t].x = l01; hf[1][t].y = l23; }
        } else if constexpr (P::kIsBF16) {
            const uint32_t lo = P::relu2(P::pack2(hv[0], hv[1])), hi = P::relu2(P::pack2(hv[2], hv[3]));
            if constexpr (hb & 1) { hf[hb >> 1][t].z = lo; hf[hb >> 1][t].w = hi; }
            else                  { hf[hb >> 1][t].x = lo; hf[hb >> 1][t].y = hi; }
        } else {
            hf[hb][t] = u32x4{__float_as_uint(fmaxf(hv[0], 0.f)), __float_as_uint(fmaxf(hv[1], 0.f)),
                              __float_as_uint(fmaxf(hv[2], 0.f)), __float_as_uint(fmaxf(hv[3], 0.f))};
        }
    };
    // phase A of chunk c: h^T = W1c x^T (fragment i = (kg, hb)); the VALU of
    // `filler(step)` is issued between the MFMAs so the matrix pipe stays fed
    // (destination: rows OFF .. OFF + HB of `dst`, an [..][NT] accumulator array)
    // `cinit`: null -> accumulate from zero; else cinit[hb] is the C operand of block hb's
    // first MFMA (the lane's 4 bias values of the block's rows)
    auto phase_a_into = [&](int c, auto& dst, auto off_tag, auto count_tag, auto cinit, auto filler) {
        constexpr int OFF = decltype(off_tag)::value;
        constexpr int COUNT = decltype(count_tag)::value;       // token blocks
        uint32_t fba[LA::VAR];
        LA::bases(lds0 + (c & 1) * 32768, idx, g, fba);
        lds_stream<LA, RA, DEPTH>(fba, [&](auto ic, const u32x4& wf) {
            constexpr int i = decltype(ic)::value;
#pragma unroll
            for (int t = 0; t < COUNT; ++t) {
                if constexpr (i / HB == 0) {
                    if constexpr (std::is_same_v<decltype(cinit), std::nullptr_t>) mma_kg<P, 0, false, 1>(dst[OFF + i % HB][t], wf, xf, t, dst[OFF + i % HB][t]);
                    else mma_kg<P, 0, false, 2>(dst[OFF + i % HB][t], wf, xf, t, cinit[i % HB]);
                } else mma_kg<P, i / HB, false, 0>(dst[OFF + i % HB][t], wf, xf, t, dst[OFF + i % HB][t]);
            }
            filler(ic);
        });
    };
    auto phase_a = [&](int c, f32x4 (&hdst)[HB][NTA], const f32x4 (&bias)[HB], auto filler) {
        phase_a_into(c, hdst, std::integral_constant<int, 0>{}, std::integral_constant<int, NTA>{}, &bias[0], filler);
    };
    // b1 of hidden chunk c as the lane's C operands (rows 4g .. 4g+3 of each 16-row block)
    auto load_b1 = [&](int c, f32x4 (&bias)[HB]) {
#pragma unroll
        for (int hb = 0; hb < HB; ++hb) {
            u32x4 raw;
            ds_read_b128_asm<0>(raw, lds_addr(ldsb1) + (hidden_chunk(c) * HC + hb * 16 + 4 * g) * 4);
            bias[hb] = __builtin_bit_cast(f32x4, raw);
        }
    };

#ifdef PPG_FFN_TIMING
    auto pstamp = [&](int k) {
        if (a.dbg && blockIdx.x == 0 && lane == 0) a.dbg[128 + wave * 16 + k] = __builtin_amdgcn_s_memtime();
    };
#else
    auto pstamp = [&](int) {};
#endif
    pstamp(0);
    if constexpr (OP) {
        // x1 = LN1(X + W_o ao + b_o): the out-projection runs like a phase A
        // per W_o tile, straight into the (not yet live) y accumulators; the
        // W_o rows are in paired order, so the accumulators of blocks 2p, 2p+1
        // are K-group p of the next GEMM's B operand (bf16; fp32: W1's columns
        // are uploaded in the matching order).  x1 also stays in the y
        // accumulators: the FFN sums on top of it, so the second residual
        // costs no memory traffic at all.
        float4 res0[NBH];        // LN1's residual rows of block 0: fetched under the last W_o tile's MFMAs
        [&]<int... I>(std::integer_sequence<int, I...>) {
            ([&] {
                dma_wait_barrier();
                pstamp(1 + 2 * I);
                if constexpr (I == OT - 1) load_residual_rows<NBH>(res0, a.X, H, tok0 + idx, a.M, g);
                phase_a_into(I, yacc, std::integral_constant<int, I * HB>{}, std::integral_constant<int, NTB>{}, nullptr, [](auto) {});
                __syncthreads();              // buffer I & 1 is free for the tile after next
                if constexpr (I + 2 < OT) stage_wo(I + 2);
                else if (I + 2 - OT < NC) stage_w1(I + 2 - OT);
                pstamp(2 + 2 * I);
            }(), ...);
        }(std::make_integer_sequence<int, OT>{});
        ln_keep<P, NBH, NT, XG, NTX>(yacc, xf, lnp1, a.X, H, tok0, a.M, idx, g, res0);
        pstamp(12);
        // yacc keeps x1: phase B accumulates W2 h on top of the residual
        if constexpr (ROLE == 1) {
            // the partner computes the phase A of this wave's last block: hand it the fragments
#pragma unroll
            for (int kg = 0; kg < XG; ++kg)
                asm volatile("ds_write_b128 %0, %1" :: "v"(xfer_x + kg * 1024), "v"(xf[kg][NTB - 1]) : "memory");
        }
    }

    // one chunk: [A(c+1) || pack(c)] -> B(c)
#ifdef PPG_FFN_TIMING
    auto stamp = [&](int c, int k) {
        if (a.dbg && blockIdx.x == 0 && lane == 0 && c >= 8 && c < 12)
            a.dbg[(wave * 4 + (c - 8)) * 8 + k] = __builtin_amdgcn_s_memtime();
    };
#else
    auto stamp = [&](int, int) {};
#endif
    // ROLE 2: the raw phase-A accumulators of the partner's block (index NTA - 1 of
    // this wave's phase-A set) for hidden chunk c go to hand-off buffer c & 1; the
    // owner applies bias, ReLU and packing among its own pack units
    auto send_foreign = [&](int c, f32x4 (&hsrc)[HB][NTA]) {
#pragma unroll
        for (int hb = 0; hb < HB; ++hb)
            asm volatile("ds_write_b128 %0, %1" :: "v"(xfer_h + ((c & 1) * HB + hb) * 1024), "v"(hsrc[hb][NTA - 1]) : "memory");
    };
    auto chunk = [&](int c, f32x4 (&hcur)[HB][NTA], f32x4 (&hnext)[HB][NTA]) {
        stamp(c, 0);
        f32x4 b1n[HB];                       // bias of the NEXT chunk: C operands of its phase A
        if (c + 1 < NC) load_b1(c + 1, b1n);
        u32x4 hf[HG][NTB];
        f32x4 hfar[HB];
        if constexpr (ROLE == 1) {
            // raw h of the last block for this chunk: written by the partner before the barrier that ended chunk c - 1
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) {
                u32x4 raw;
                ds_read_b128_asm<0>(raw, xfer_h + ((c & 1) * HB + hb) * 1024);
                hfar[hb] = __builtin_bit_cast(f32x4, raw);
            }
        }
        if (c + 2 < NC) stage_w1(c + 2);
        if (c + 1 < NC) stage_w2(c + 1);
        stamp(c, 1);
        if (c + 1 < NC) {
            // (stream step 0 waits on an LDS op younger than the b1 / hand-off reads above)
            phase_a(c + 1, hnext, b1n, [&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if constexpr (i == 0 && ROLE == 1) {
#pragma unroll
                    for (int hb = 0; hb < HB; ++hb) asm volatile("" : "+v"(hfar[hb]));
                }
                // spread the UNITS pack units evenly over the RA stream steps
                if constexpr ((i * UNITS) / RA != ((i + 1) * UNITS) / RA)
                    pack_unit(std::integral_constant<int, (i * UNITS) / RA>{}, hcur, hfar, hf);
            });
            if constexpr (ROLE == 2) send_foreign(c + 1, hnext);
        } else {
            if constexpr (ROLE == 1) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int hb = 0; hb < HB; ++hb) asm volatile("" : "+v"(hfar[hb]));
            }
            [&]<int... U>(std::integer_sequence<int, U...>) {
                (pack_unit(std::integral_constant<int, U>{}, hcur, hfar, hf), ...);
            }(std::make_integer_sequence<int, UNITS>{});
        }
        stamp(c, 2);
        // phase B: y^T += W2c h^T ; fragment i = (kg, nb)
        uint32_t fbb[LB::VAR];
        LB::bases(lds0 + 65536 + (c & 1) * 32768, idx, g, fbb);
        lds_stream<LB, HG * NBH, DEPTH>(fbb, [&](auto ic, const u32x4& wf) {
            constexpr int i = decltype(ic)::value;
#pragma unroll
            for (int t = 0; t < NT; ++t) mma_kg<P, i / NBH, false, 0>(yacc[i % NBH][t], wf, hf, t, yacc[i % NBH][t]);
        });
        stamp(c, 3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(c, 4);
        __syncthreads();
        stamp(c, 5);
    };

    f32x4 h0[HB][NTA], h1[HB][NTA];
    pstamp(13);
    dma_wait_barrier();
    pstamp(14);
    if constexpr (ROLE == 2) {
        // x1 fragments of the partner's block (written before the barrier above)
#pragma unroll
        for (int kg = 0; kg < XG; ++kg) ds_read_b128_asm<0>(xf[kg][NTA - 1], xfer_x + kg * 1024);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int kg = 0; kg < XG; ++kg) asm volatile("" : "+v"(xf[kg][NTA - 1]));
    }
    {
        f32x4 b10[HB];
        load_b1(0, b10);
        phase_a(0, h0, b10, [](auto) {});
    }
    if constexpr (ROLE == 2) send_foreign(0, h0);
    __syncthreads();                      // W1 buffer 0 is re-filled by chunk 0's DMA; hand-off 0 is written
    for (int c = 0; c < NC; ++c) {
        chunk(c, h0, h1);
#pragma unroll
        for (int hb = 0; hb < HB; ++hb)
#pragma unroll
            for (int t = 0; t < NTA; ++t) h0[hb][t] = h1[hb][t];
    }

    pstamp(9);
    if (a.partial != nullptr) {
        // split-hidden: raw partial sums, [split][token][feature] fp32.  Under a row map the rows are those of the
        // launch's slots, densely (the step's rows lie all over the token space: a split stride of the whole space put
        // every 16-row block of every split on a page of its own -- 30 us of address translation in the reduce pass)
        const int prows = a.rowmap ? (int)gridDim.x * 64 : a.M;
        float* part = a.partial + (size_t)blockIdx.y * prows * H;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int m = tok0 + 16 * t + idx;
            if (m >= a.M) continue;
            const int pm = a.rowmap ? (blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + idx : m;
#pragma unroll
            for (int nb = 0; nb < NBH; ++nb)
                *reinterpret_cast<float4*>(part + (size_t)pm * H + pair_feature(nb, g)) =
                    make_float4(yacc[nb][t][0], yacc[nb][t][1], yacc[nb][t][2], yacc[nb][t][3]);
        }
        if constexpr (NT == 1 && NBH == 16 && !OP) {
            if (a.tickets != nullptr) {
                // one-pass form: every workgroup of the tile publishes its partial rows (device-scope release) and draws
                // a ticket; the last one to arrive sees all of them (acquire) and finishes the tile's 64 rows
                __threadfence();
                __syncthreads();
                int* last = reinterpret_cast<int*>(smem);            // (the weight buffers are dead)
                if (tid == 0) {
                    const int drawn = atomicAdd(a.tickets + blockIdx.x, 1);
                    *last = drawn == (int)gridDim.y - 1;
                    if (*last) a.tickets[blockIdx.x] = 0;            // for the next launch
                }
                __syncthreads();
                if (!*last) return;
                __threadfence();
                const int slot0 = (blockIdx.x * 4 + wave) * 16;
                if (tok0 + 16 <= a.M) {
#pragma unroll 1
                    for (int r = 0; r < 16; r += 4) reduce_ln_rows<P, 4>(a, (int)gridDim.y, slot0 + r, tok0 + r, lane, (size_t)prows * H);
                } else {
                    for (int r = 0; r < 16 && tok0 + r < a.M; ++r) reduce_ln_rows<P, 1>(a, (int)gridDim.y, slot0 + r, tok0 + r, lane, (size_t)prows * H);
                }
            }
        }
        return;
    }
    // (OP: nothing derived from the lane coordinates before the chunk loop may stay live across it)
    int tok0e = tok0, idxe = idx, ge = g;
    if constexpr (OP) {
        asm volatile("" : "+v"(tok0e), "+v"(idxe), "+v"(ge));
        tok0e = __builtin_amdgcn_readfirstlane(tok0e);
    }
    if constexpr (!QKV) {
        resln<P, NBH, NT, OP ? LN_NO_RESIDUAL : LN_EPILOGUE>(yacc, lnp2, a.X, a.Xb, H, tok0e, a.M, idxe, ge);
    } else {
        // ---- the next layer's Q/K/V projection on x2 = LN2(...) ----------------
        // x2 is in the accumulators in paired feature order, i.e. (as in the OP
        // prologue) already the B fragments of a GEMM over the hidden dimension.
        // W_qkv streams through the four now idle 32 KiB weight buffers, three
        // tiles of HC rows ahead; every tile yields HC finished features that are
        // stored at once (Q/K rows: 16-byte stores; V tiles: swapped operands,
        // transposed store, see linear_kernel).  With one wave per SIMD the
        // stores are issue-bound (~280 cycles each, measured) and do not overlap
        // the MFMAs -- interleaving them into the next tile's stream changes
        // nothing -- but the tail still beats the stand-alone projection kernel:
        // no activation reload, no second pass over X as bf16.
        static_assert(OP, "the Q/K/V tail needs the residual-in-accumulator form");
        constexpr int QT = 3 * H / HC;                     // W_qkv tiles
        constexpr int PIECES = 32768 / 1024 / 4;           // DMA instructions per tile and wave
        // workgroup b walks the tiles from its own offset (the stores of the
        // lockstepped workgroups then spread over the Q|K row instead of all
        // hitting one 128-byte column)
        const int qrot = blockIdx.x % QT;
        auto tile_of = [&](int i) { const int r = i + qrot; return r >= QT ? r - QT : r; };
        auto stage_q = [&](int i) {
            stage_tile<HC, ROW1, 4>(a.Wq + (size_t)tile_of(i) * 32768, (size_t)ROW1, smem + (i & 3) * 32768, wave, lane);
        };
        // the chunk loop ended on a barrier: all four buffers are free
        stage_q(0); stage_q(1); stage_q(2);
        resln<P, NBH, NT, LN_NO_RESIDUAL_KEEP>(yacc, lnp2, a.X, nullptr, H, tok0e, a.M, idxe, ge);
        pstamp(10);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int kg = 0; kg < XG; ++kg) {
                if constexpr (P::kIsBF16) {
                    xf[kg][t] = u32x4{P::pack2(yacc[2 * kg][t][0], yacc[2 * kg][t][1]), P::pack2(yacc[2 * kg][t][2], yacc[2 * kg][t][3]),
                                      P::pack2(yacc[2 * kg + 1][t][0], yacc[2 * kg + 1][t][1]), P::pack2(yacc[2 * kg + 1][t][2], yacc[2 * kg + 1][t][3])};
                } else {
                    xf[kg][t] = u32x4{__float_as_uint(yacc[kg][t][0]), __float_as_uint(yacc[kg][t][1]),
                                      __float_as_uint(yacc[kg][t][2]), __float_as_uint(yacc[kg][t][3])};
                }
            }
        }
        bool full_rows = true, full_cols = true;           // every store below is really issued: counted waits are exact
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            full_rows = full_rows && tok0 + 16 * t < a.M;
            full_cols = full_cols && vcol[t] >= 0;
        }
        const int v_start = 2 * H;
#ifdef PPG_FFN_TIMING
        auto qstamp = [&](int i, int k) {
            if (a.dbg && blockIdx.x == 0 && lane == 0 && wave == 0 && i >= 4 && i < 10) a.dbg[192 + (i - 4) * 8 + k] = __builtin_amdgcn_s_memtime();
        };
#else
        auto qstamp = [&](int, int) {};
#endif
        for (int i = 0; i < QT; ++i) {
            qstamp(i, 0);
            // Tile i's DMA is older than: the DMA of tiles i+1, i+2 and the stores of
            // tiles i-2, i-1 (order per iteration: wait, barrier, MFMAs, stores, DMA
            // of tile i+3).  Q/K tiles store HB/2 * NT times per wave, V tiles at
            // least as often, so that many operations may stay in flight.
            constexpr int NSTORE = (HB / 2 > 0 ? HB / 2 : 1) * NT;
            if (full_rows && full_cols && i >= 2 && i + 2 < QT)
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * PIECES + 2 * NSTORE) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            qstamp(i, 1);
            __syncthreads();
            qstamp(i, 2);
            const int n0 = tile_of(i) * HC;
            const bool swap = n0 >= v_start;
            f32x4 acc[HB][NT];
            uint32_t fba[LA::VAR];
            LA::bases(lds0 + (i & 3) * 32768, idx, g, fba);
            auto mfmas = [&](auto swap_tag) {
                constexpr bool SWAP = decltype(swap_tag)::value;
                lds_stream<LA, RA, DEPTH>(fba, [&](auto ic, const u32x4& wf) {
                    constexpr int f = decltype(ic)::value;
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        if constexpr (f / HB == 0) {
                            if constexpr (SWAP) P::mma0(acc[f % HB][t], xf[0][t], wf);
                            else P::mma0(acc[f % HB][t], wf, xf[0][t]);
                        } else {
                            if constexpr (SWAP) P::mma(acc[f % HB][t], xf[f / HB][t], wf);
                            else P::mma(acc[f % HB][t], wf, xf[f / HB][t]);
                        }
                    }
                });
            };
            if (!swap) {
                mfmas(std::false_type{});
                qstamp(i, 3);
                PairStore<P> pair[NT];
#pragma unroll
                for (int nb = 0; nb < HB; ++nb) {
                    const int gb = n0 / 16 + nb;                     // 16-row block index in W_qkv
                    const int n = pair_feature(gb, g);
                    const float4 bv = *reinterpret_cast<const float4*>(ldsbq + n);
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const int m = tok0 + 16 * t + idx;
                        if (tok0 + 16 * t >= a.M) continue;          // wave-uniform
                        pair[t].put(a.qk_out + ((size_t)m * 2 * H + (n & ~7)) * P::kBytes, gb & 1,
                                    acc[nb][t][0] + bv.x, acc[nb][t][1] + bv.y,
                                    acc[nb][t][2] + bv.z, acc[nb][t][3] + bv.w);
                    }
                }
            } else {
                mfmas(std::true_type{});
                qstamp(i, 3);
                bool done_with_previous = false;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    if (done_with_previous) { done_with_previous = false; continue; }
                    if (vw[t] < 0) continue;
                    constexpr int kLast = NT - 1;
                    const int tn = t < kLast ? t + 1 : t;
                    const bool paired = P::kIsBF16 && t < kLast && vw[tn] == vw[t] && (vcol[t] & 4) == 0;
#pragma unroll
                    for (int nb = 0; nb < HB; ++nb) {
                        const int row = n0 + nb * 16 + idx;
                        const float bv = ldsbq[pair_row(row)];
                        char* dst = a.vt_out + ((size_t)(row - v_start) * a.vt_ld + vcol[t] + (P::kIsBF16 ? 8 : 4) * g) * P::kBytes;
                        if (paired) {
                            *reinterpret_cast<u32x4*>(dst) = u32x4{
                                P::pack2(acc[nb][t][0] + bv, acc[nb][t][1] + bv), P::pack2(acc[nb][t][2] + bv, acc[nb][t][3] + bv),
                                P::pack2(acc[nb][tn][0] + bv, acc[nb][tn][1] + bv), P::pack2(acc[nb][tn][2] + bv, acc[nb][tn][3] + bv)};
                        } else {
                            store4<P>(dst, acc[nb][t][0] + bv, acc[nb][t][1] + bv, acc[nb][t][2] + bv, acc[nb][t][3] + bv);
                        }
                    }
                    done_with_previous = paired;
                }
            }
            qstamp(i, 4);
            if (i + 3 < QT) stage_q(i + 3);       // into the buffer tile i-1 was read from (all waves passed this iteration's barrier)
            qstamp(i, 5);
        }
        pstamp(11);
    }
}

template <class P, int NT, int NBH, bool OP, bool QKV>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void ffn_kernel(FfnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tok0 = a.rowmap ? __builtin_amdgcn_readfirstlane(a.rowmap[blockIdx.x * 4 + wave]) : (blockIdx.x * 4 + wave) * 16 * NT;
    ffn_body<P, NT, NT, NBH, OP, QKV, 0>(a, smem, tok0);
}

// Mixed tiling: 160 tokens per workgroup = waves owning 3, 3, 2, 2 blocks of 16,
// with the 2-block waves computing the phase A of their partner's third block
// (ffn_body ROLE 1 / 2): every wave runs 5 block-phases per chunk instead of 6.
// At the benchmark shape (40 960 token rows on 256 CUs = 160 rows per CU) this is
// one workgroup on every CU where 192-token workgroups leave 42 CUs idle.
template <class P, int NBH, bool QKV>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void ffn_mixed_kernel(FfnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int base = blockIdx.x * 160;
    if (wave < 2) ffn_body<P, 2, 3, NBH, true, QKV, 1>(a, smem, base + wave * 48);
    else ffn_body<P, 3, 2, NBH, true, QKV, 2>(a, smem, base + 96 + (wave - 2) * 32);
}

// Second half of the split-hidden FFN: X <- LN(X + b2 + sum_s partial[s]).
// One wave per token row (64 lanes x float4 = 256 features; hidden 512 in two
// steps), HBM-bound elementwise.
template <class P>
__global__ __launch_bounds__(256) void ffn_reduce_ln_kernel(FfnArgs a, int splits) {
    const int lane = threadIdx.x & 63;
    // (wave-uniform by construction; said so, the row-map entry is a scalar load -- it does not queue behind the
    // vector loads of the partial sums, which return in issue order)
    const int slot = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int m = a.rowmap ? a.rowmap[slot >> 4] + (slot & 15) : slot;          // (row map: only the rows of the step)
    if (a.H == 256 && splits <= 8) {
        // (the row-map entry and the partial rows travel together; a row past M is computed and not stored)
        reduce_ln_rows<P, 1>(a, splits, a.rowmap ? slot : min(slot, a.M - 1), m, lane,
                             (size_t)(a.rowmap ? (a.map_blocks + 3) / 4 * 64 : a.M) * 256);
        return;
    }
    if (m >= a.M) return;
    const int H = a.H;
    float4 v[2];
    float sum = 0.f;
    for (int h = 0; h < H / 256; ++h) {
        const int n = h * 256 + lane * 4;
        const float4 bv = *reinterpret_cast<const float4*>(a.b2 + n);
        const float4 xv = *reinterpret_cast<const float4*>(a.X + (size_t)m * H + n);
        float4 acc = make_float4(bv.x + xv.x, bv.y + xv.y, bv.z + xv.z, bv.w + xv.w);
        // the partial sums are added in split order, loaded 8 / 4 / 1 at a time:
        // one load latency per batch instead of one per split
        const float* part = a.partial + (size_t)(a.rowmap ? slot : m) * H + n;
        const size_t stride = (size_t)(a.rowmap ? (a.map_blocks + 3) / 4 * 64 : a.M) * H;
        int sidx = 0;
        auto batch = [&](auto count) {
            constexpr int N = decltype(count)::value;
            for (; sidx + N <= splits; sidx += N) {
                float4 p[N];
#pragma unroll
                for (int j = 0; j < N; ++j) p[j] = *reinterpret_cast<const float4*>(part + (size_t)(sidx + j) * stride);
#pragma unroll
                for (int j = 0; j < N; ++j) { acc.x += p[j].x; acc.y += p[j].y; acc.z += p[j].z; acc.w += p[j].w; }
            }
        };
        batch(std::integral_constant<int, 8>{});
        batch(std::integral_constant<int, 4>{});
        batch(std::integral_constant<int, 1>{});
        v[h] = acc;
        sum += (acc.x + acc.y) + (acc.z + acc.w);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
    const float mean = sum / (float)H;
    float sq = 0.f;
    for (int h = 0; h < H / 256; ++h) {
        const float d0 = v[h].x - mean, d1 = v[h].y - mean, d2 = v[h].z - mean, d3 = v[h].w - mean;
        sq += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sq += __shfl_xor(sq, off);
    const float rstd = 1.0f / sqrtf(sq / (float)H + kLnEps);
    for (int h = 0; h < H / 256; ++h) {
        const int n = h * 256 + lane * 4;
        const float4 gv = *reinterpret_cast<const float4*>(a.gamma + n);
        const float4 ev = *reinterpret_cast<const float4*>(a.beta + n);
        const float y0 = (v[h].x - mean) * rstd * gv.x + ev.x;
        const float y1 = (v[h].y - mean) * rstd * gv.y + ev.y;
        const float y2 = (v[h].z - mean) * rstd * gv.z + ev.z;
        const float y3 = (v[h].w - mean) * rstd * gv.w + ev.w;
        *reinterpret_cast<float4*>(a.X + (size_t)m * H + n) = make_float4(y0, y1, y2, y3);
        if constexpr (P::kIsBF16 || P::kSplit) store4<P>(a.Xb + (size_t)m * H * P::kBytes + P::row_byte(n), y0, y1, y2, y3);
    }
}

template <class P, int NT, int NB, int EPI>
hipError_t launch_linear_t(const LinearArgs& a, int ypasses, hipStream_t s) {
    if (a.rowmap && NT != 1) return hipErrorInvalidValue;
    const int blocks = a.rowmap ? (a.map_blocks + 3) / 4 : (a.M + 64 * NT - 1) / (64 * NT);
    const size_t lds = 2 * NB * 16 * 128;
    auto kern = linear_kernel<P, NT, NB, EPI>;
    if (lds > 65536) {
        static ppg::LdsLimit limit;
        const hipError_t e = limit.ensure(reinterpret_cast<const void*>(kern), lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(blocks, ypasses), dim3(256), lds, s, a);
    return hipGetLastError();
}

template <class P, int NB, int EPI>
hipError_t launch_linear_nt(int nt, const LinearArgs& a, int ypasses, hipStream_t s) {
    if (nt == 1) return launch_linear_t<P, 1, NB, EPI>(a, ypasses, s);
    if (nt == 3) return launch_linear_t<P, 3, NB, EPI>(a, ypasses, s);
    return launch_linear_t<P, 2, NB, EPI>(a, ypasses, s);
}

template <class P>
hipError_t launch_linear_p(int epi, int nb, int nt, const LinearArgs& a, int ypasses, hipStream_t s) {
    switch (epi) {
    case EPI_INCONV: return launch_linear_nt<P, 16, EPI_INCONV>(nt, a, ypasses, s);
    case EPI_QKV:    return launch_linear_nt<P, 16, EPI_QKV>(nt, a, ypasses, s);
    case EPI_RELU:   return launch_linear_nt<P, 16, EPI_RELU>(nt, a, ypasses, s);
    case EPI_GELU:   return launch_linear_nt<P, 16, EPI_GELU>(nt, a, ypasses, s);
    case EPI_GENERAL: return nb == 3 ? launch_linear_nt<P, 3, EPI_GENERAL>(nt, a, ypasses, s) : launch_linear_nt<P, 16, EPI_GENERAL>(nt, a, ypasses, s);
    case EPI_OUTCONV:return launch_linear_nt<P, 3, EPI_OUTCONV>(nt, a, ypasses, s);
    case EPI_RESLN:
        // one 16-token block per wave: the LayerNorm epilogue holds a whole feature row per token in registers
        if (nb == 16) return launch_linear_t<P, 1, 16, EPI_RESLN>(a, ypasses, s);
        if (nb == 32) return launch_linear_t<P, 1, 32, EPI_RESLN>(a, ypasses, s);
        return hipErrorInvalidValue;
    }
    return hipErrorInvalidValue;
}

template <class P, int NT, int NBH, bool OP, bool QKV>
hipError_t launch_ffn_t(const FfnArgs& a, hipStream_t s) {
    if (a.rowmap && NT != 1) return hipErrorInvalidValue;
    const dim3 blocks(a.rowmap ? (a.map_blocks + 3) / 4 : (a.M + 64 * NT - 1) / (64 * NT), a.partial ? a.splits : 1);
    auto kern = ffn_kernel<P, NT, NBH, OP, QKV>;
    const size_t lds = 131072 + (size_t)a.F * 4 + ((OP || QKV) ? 9 : 6) * (size_t)a.H * 4;   // b1, [b2 g2 e2], [bq], [bo g1 e1]
    static ppg::LdsLimit limit;
    hipError_t e = limit.ensure(reinterpret_cast<const void*>(kern), lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, blocks, dim3(256), lds, s, a);
    e = hipGetLastError();
    if (e != hipSuccess || !a.partial) return e;
    if (a.tickets != nullptr) {
        // (the kernel's one-pass form exists for 16 tokens per wave, hidden 256, no fused out-projection, <= 8 splits)
        if (NT == 1 && NBH == 16 && !OP && a.splits <= 8) return e;
        return hipErrorInvalidValue;
    }
    hipLaunchKernelGGL(ffn_reduce_ln_kernel<P>, dim3(a.rowmap ? a.map_blocks * 4 : (a.M + 3) / 4), dim3(256), 0, s, a, a.splits);
    return hipGetLastError();
}

template <class P, bool QKV>
hipError_t launch_ffn_mixed(const FfnArgs& a, hipStream_t s) {
    if constexpr (!P::kIsBF16) {
        return hipErrorInvalidValue;
    } else {
        if (a.H != 256 || a.partial != nullptr) return hipErrorInvalidValue;
        auto kern = ffn_mixed_kernel<P, 16, QKV>;
        constexpr int HB = 4;                                  // 16-row blocks of a 64-hidden chunk (hidden 256, bf16)
        // hand-off: 2 wave pairs x 2 buffers x HB KiB of raw accumulators, from the LN1 parameters on
        const size_t lds = 131072 + (size_t)a.F * 4 + 6 * (size_t)a.H * 4 + 2 * 2 * HB * 1024;
        static ppg::LdsLimit limit;
        const hipError_t e = limit.ensure(reinterpret_cast<const void*>(kern), lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3((a.M + 159) / 160), dim3(256), lds, s, a);
        return hipGetLastError();
    }
}

template <class P, bool OP, bool QKV>
hipError_t launch_ffn_op(const FfnArgs& a, int nt, hipStream_t s) {
    if (a.H == 256) {
        if (nt == 1) return launch_ffn_t<P, 1, 16, OP, QKV>(a, s);
        if constexpr (P::kIsBF16) if (nt == 3) return launch_ffn_t<P, 3, 16, OP, QKV>(a, s);
        return launch_ffn_t<P, 2, 16, OP, QKV>(a, s);
    }
    if (a.H == 512) return launch_ffn_t<P, 1, 32, OP, QKV>(a, s);
    return hipErrorInvalidValue;
}

template <class P>
hipError_t launch_ffn_p(const FfnArgs& a, int nt, hipStream_t s) {
    if (a.Wo != nullptr) {
        if (a.partial != nullptr) return hipErrorInvalidValue;   // every split would redo the out-projection
        if (nt == ppg::kFfnMixedTiling) return a.Wq != nullptr ? launch_ffn_mixed<P, true>(a, s) : launch_ffn_mixed<P, false>(a, s);
        if (a.Wq != nullptr) return launch_ffn_op<P, true, true>(a, nt, s);
        return launch_ffn_op<P, true, false>(a, nt, s);
    }
    if (a.Wq != nullptr) return hipErrorInvalidValue;            // the Q/K/V tail comes with the fused out-projection only
    return launch_ffn_op<P, false, false>(a, nt, s);
}

// Split-precision operands (PrecX2): the token-split kernels of the unfused launch sequence.  Hidden 256: Q/K/V,
// attention, out-projection + LayerNorm, the fused FFN.  Hidden 512 (head dimension 256): the FFN as two GEMMs
// (linear-1 + ReLU into [32 hi | 32 lo] rows, linear-2 + residual + LayerNorm) -- a chunk of the fused FFN kernel
// would have to hold a whole 32-wide hidden group of both weight tiles, 2 x 2 x 64 KiB of LDS.
template <int NT>
hipError_t launch_linear_x2_nt(int epi, int nb, const LinearArgs& a, int ypasses, hipStream_t s) {
    switch (epi) {
    case EPI_INCONV:  return launch_linear_t<PrecX2, NT, 16, EPI_INCONV>(a, ypasses, s);
    case EPI_QKV:     return launch_linear_t<PrecX2, NT, 16, EPI_QKV>(a, ypasses, s);
    case EPI_RELU:    return launch_linear_t<PrecX2, NT, 16, EPI_RELU>(a, ypasses, s);
    case EPI_GELU:    return launch_linear_t<PrecX2, NT, 16, EPI_GELU>(a, ypasses, s);       // wav2vec2 feature encoder
    case EPI_GENERAL: return nb == 16 ? launch_linear_t<PrecX2, NT, 16, EPI_GENERAL>(a, ypasses, s) : hipErrorInvalidValue;   // wav2vec2 body
    case EPI_OUTCONV: return launch_linear_t<PrecX2, NT, 3, EPI_OUTCONV>(a, ypasses, s);
    case EPI_RESLN:
        if (nb == 16) return launch_linear_t<PrecX2, 1, 16, EPI_RESLN>(a, ypasses, s);
        if (nb == 32) return launch_linear_t<PrecX2, 1, 32, EPI_RESLN>(a, ypasses, s);
        return hipErrorInvalidValue;
    }
    return hipErrorInvalidValue;
}
hipError_t launch_linear_x2(int epi, int nb, int nt, const LinearArgs& a, int ypasses, hipStream_t s) {
    return nt == 1 ? launch_linear_x2_nt<1>(epi, nb, a, ypasses, s) : launch_linear_x2_nt<2>(epi, nb, a, ypasses, s);
}
hipError_t launch_ffn_x2(const FfnArgs& a, int nt, hipStream_t s) {
    if (a.H != 256 || a.Wo != nullptr || a.Wq != nullptr) return hipErrorInvalidValue;
    return nt == 1 ? launch_ffn_t<PrecX2, 1, 16, false, false>(a, s) : launch_ffn_t<PrecX2, 2, 16, false, false>(a, s);
}

}  // namespace

namespace ppg {


// -DPPG_ONLY_BF16: kernel experiments build only the bf16 instantiations (a third of the compile time)
#ifdef PPG_ONLY_BF16
#define PPG_OTHER_PRECISIONS 0
#else
#define PPG_OTHER_PRECISIONS 1
#endif
// (-DPPG_ONLY_BF16 -DPPG_WITH_X2: bf16 and the split-precision mode)
#if PPG_OTHER_PRECISIONS || defined(PPG_WITH_X2)
#define PPG_X2 1
#else
#define PPG_X2 0
#endif

hipError_t launch_gather(int precision, const GatherArgs& a, hipStream_t s) {
    dim3 grid(a.rowmap ? (a.map_blocks + 3) / 4 : (a.M + 63) / 64, (a.Cp + 31) / 32 + 1);       // + the housekeeping row
    if (precision == PPG_PRECISION_BF16) hipLaunchKernelGGL(gather_kernel<PrecBF16>, grid, dim3(256), 0, s, a);
#if PPG_X2
    else if (precision == PPG_PRECISION_FP16X2) hipLaunchKernelGGL(gather_kernel<PrecX2>, grid, dim3(256), 0, s, a);
#endif
#if PPG_OTHER_PRECISIONS
    else if (precision == PPG_PRECISION_FP16) hipLaunchKernelGGL(gather_kernel<PrecF16>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(gather_kernel<PrecF32>, grid, dim3(256), 0, s, a);
#else
    else return hipErrorInvalidValue;
#endif
    return hipGetLastError();
}

hipError_t launch_fill(float* p, size_t n, float v, hipStream_t s) {
    const int blocks = (int)std::min<size_t>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(fill_kernel, dim3(blocks), dim3(256), 0, s, p, n, v);
    return hipGetLastError();
}

hipError_t launch_linear(int precision, int epi, int nb, int nt, const LinearArgs& a, int ypasses, hipStream_t s) {
    if (precision == PPG_PRECISION_BF16) return launch_linear_p<PrecBF16>(epi, nb, nt, a, ypasses, s);
#if PPG_X2
    if (precision == PPG_PRECISION_FP16X2) return launch_linear_x2(epi, nb, nt, a, ypasses, s);
#endif
#if PPG_OTHER_PRECISIONS
    if (precision == PPG_PRECISION_FP16) return launch_linear_p<PrecF16>(epi, nb, nt, a, ypasses, s);
    return launch_linear_p<PrecF32>(epi, nb, nt, a, ypasses, s);
#else
    return hipErrorInvalidValue;
#endif
}

hipError_t launch_ffn(int precision, const FfnArgs& a, int nt, hipStream_t s) {
    if (precision == PPG_PRECISION_BF16) return launch_ffn_p<PrecBF16>(a, nt, s);
#if PPG_X2
    if (precision == PPG_PRECISION_FP16X2) return launch_ffn_x2(a, nt, s);
#endif
#if PPG_OTHER_PRECISIONS
    if (precision == PPG_PRECISION_FP16) return launch_ffn_p<PrecF16>(a, nt, s);
    return launch_ffn_p<PrecF32>(a, nt, s);
#else
    return hipErrorInvalidValue;
#endif
}

}  // namespace ppg
