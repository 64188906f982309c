// Attention kernels of the PPG encoder for gfx950 (flash-style, transposed orientation; see attn_body).
#include "ppg_device.h"
#include "ppg_launch.h"

#include <stdlib.h>
#include <string.h>
#include "ppg_lds.h"

#include <limits.h>
#include <type_traits>
#include <utility>

namespace {

// ---------------------------------------------------------------------------
// Attention for one (window, head, query tile).  Transposed orientation:
//   S^T[key][q] = K q^T   (A = K rows from LDS,   B = Q rows in registers)
//   O^T[d][q]  += V^T P^T (A = V^T rows from LDS, B = P^T = exp(S^T - m))
// The S^T accumulator (4 consecutive keys per lane for one query) is, after
// exponentiation and packing, directly the B fragment of the PV MFMA; the
// matching key order of V^T is produced by the QKV epilogue.  Row max/sum
// over keys = per-lane partials + shuffles over the 4 lane groups.
// ---------------------------------------------------------------------------
// Bytes of one K (and one V^T) tile of the attention kernels (four of them in LDS: two DMA double buffers)
template <class P, int DH>
constexpr int attn_tile_bytes() { return (P::kSplit && DH == 256) ? 32768 : ((P::kSplit && DH == 64) ? 8192 : 16384); }

template <class P, int NTQ, int DH, int NW>
__device__ __forceinline__ void attn_body(const AttnArgs& a, const AttnItem& item, const int head, char* smem) {
    constexpr int ROWK = DH * P::kBytes;            // K tile row bytes
    constexpr int DG = ROWK / 64;                   // K-groups over head dim
    // bytes of a K tile and of a V^T tile: 16 KiB; split operands: ONE [32 hi | 32 lo] group of keys per tile, so
    // 32 KiB at d = 256 (one workgroup per CU: 128 KiB of tile buffers), 16 KiB at d = 128, 8 KiB at d = 64
    constexpr int TB = attn_tile_bytes<P, DH>();
    constexpr int KT = TB / ROWK;                   // keys per tile
    constexpr int KB = KT / 16;                     // key 16-blocks per tile
    constexpr int ROWV = KT * P::kBytes;            // V^T tile row bytes (128 or 64)
    constexpr int PG = ROWV / 64;                   // K-groups of PV per tile
    constexpr int DB = DH / 16;                     // head-dim 16-blocks

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int idx = lane & 15;
    const int g = lane >> 4;
    const struct { int tok_off, vt_off, frames, valid; } w = {item.tok_off, item.vt_off, item.frames, item.valid};
#ifdef PPG_ATTN_TIMING
    const unsigned long long wg_t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz, one counter for the whole chip
#endif
    const int qw0 = item.q0 + wave * 16 * NTQ;      // first query of this wave

    // Q fragments
    u32x4 qf[DG][NTQ];
    // requested by hand (the compiler, blind to the tile DMAs issued behind them, would wait for its own loads with
    // vmcnt(0) -- for every tile of the prologue -- at the first score MFMA): unconditional, a block past the window's
    // padded rows reads the window's first row and is zeroed behind the wait
    bool q_ok[NTQ];
#pragma unroll
    for (int t = 0; t < NTQ; ++t) {
        q_ok[t] = (qw0 + 16 * t) < ((w.frames + 15) & ~15);
        const int m = w.tok_off + (q_ok[t] ? qw0 + 16 * t + idx : 0);
        const char* src = a.qk + (size_t)m * a.qk_ld_bytes + (size_t)head * ROWK + g * 16;
#pragma unroll
        for (int kg = 0; kg < DG; ++kg)
            asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(qf[kg][t]) : "v"(src), "n"(kg * 64) : "memory");
    }

    int kend = w.valid;                              // keys >= valid are masked
    if (a.causal) kend = min(kend, item.q0 + NW * 16 * NTQ);
    const int ntiles = (kend + KT - 1) / KT;

    const char* kbase = a.qk + (size_t)w.tok_off * a.qk_ld_bytes + ((size_t)a.H + (size_t)head * DH) * P::kBytes;
    const char* vbase = a.vt + (size_t)head * DH * a.vt_ld_bytes + (size_t)w.vt_off * P::kBytes;

    // LDS: K tiles at 0 / TB, V^T tiles at 2 TB / 3 TB (DMA double buffers).
    // K runs one tile ahead of V: the scores of tile kt+1 are computed while
    // the softmax of tile kt runs (see the loop).
    auto stage_k = [&](int kt) {
        stage_tile<KT, ROWK, NW>(kbase + (size_t)kt * KT * a.qk_ld_bytes, (size_t)a.qk_ld_bytes, smem + (kt & 1) * TB, wave, lane);
    };
    auto stage_v = [&](int kt) {
        stage_tile<DH, ROWV, NW>(vbase + (size_t)kt * ROWV, (size_t)a.vt_ld_bytes, smem + 2 * TB + (kt & 1) * TB, wave, lane);
    };

    f32x4 oacc[DB][NTQ];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int t = 0; t < NTQ; ++t) oacc[db][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Softmax with a per-query SHIFT instead of a running maximum.  The Q rows arrive scaled by
    // log2(e) / sqrt(d) (folded into W_q and b_q by the engine), the score accumulators start at -shift
    // (the MFMA's C operand), so a score comes out of the matrix pipe as s - shift and p = exp2(that) is
    // ONE v_exp_f32: no multiply-subtract, no row maximum, no rescale of O in the tile loop.  The shift
    // is the exact maximum of the query's first tile; softmax needs nothing more of it than that no p
    // overflows (the result O / l does not depend on it, and p's relative precision does not either):
    // a later tile where a lane's p sum exceeds P::kProbCeil (2^40 in bf16 / fp32, 2^10 in fp16) re-bases the shift
    // to the new maximum -- rescales l and O, redoes that tile's p -- which trained attention logits
    // do not do after the first tile (covered by tests/test_gpu_parity.py::test_attention_rebase).
    float shift[NTQ], lrun[NTQ];
    f32x4 cinit[NTQ];                           // {-shift x 4}: C operand of a tile's first MFMAs
#pragma unroll
    for (int t = 0; t < NTQ; ++t) { shift[t] = 0.f; lrun[t] = 0.f; cinit[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    const uint32_t lds0 = lds_addr(smem);
    constexpr int NSTEP = DG * KB;              // fragments of a K tile (TB / 1 KiB)
    // S^T = K q^T - shift of tile kt into s (fragment i = (kg, kb) = (i / KB, i % KB));
    // `filler(step)` is VALU work issued between the MFMAs
    auto scores = [&](int kt, f32x4 (&s)[KB][NTQ], auto filler) {
        using LK = FragLayout<ROWK, KB>;
        uint32_t fbk[LK::VAR];
        LK::bases(lds0 + (kt & 1) * TB, idx, g, fbk);
        lds_stream<LK, NSTEP, 6>(fbk, [&](auto ic, const u32x4& kf) {
            constexpr int i = decltype(ic)::value;
#pragma unroll
            for (int t = 0; t < NTQ; ++t) {
                mma_kg<P, i / KB, false, (i / KB == 0 ? 2 : 0)>(s[i % KB][t], kf, qf, t, cinit[t]);
            }
            filler(ic);
        });
    };

    // keys below key_limit[t] + 4 g count for the lane's query of block t (padding mask, causal diagonal)
    int key_limit[NTQ];
#pragma unroll
    for (int t = 0; t < NTQ; ++t)
        key_limit[t] = (a.causal ? min(w.valid, qw0 + 16 * t + idx + 1) : w.valid) - 4 * g;
    // the lane's keys of tile kt are 16 kb + e + (kt KT + 4 g): one subtraction, then constants against it
    // (written out per key, the compiler hoists the 16 key indices above the branch and every tile pays)
    auto mask_tile = [&](int t, int kt, f32x4 (&s)[KB][NTQ]) {
        const int rel = key_limit[t] - kt * KT;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (kb * 16 + e >= rel) s[kb][t][e] = -INFINITY;
    };
    auto tile_max = [&](int t, f32x4 (&s)[KB][NTQ]) {
        float mx = max3(s[0][t][0], s[0][t][1], s[0][t][2]);
        mx = fmaxf(mx, s[0][t][3]);
#pragma unroll
        for (int kb = 1; kb < KB; ++kb) {
            mx = max3(mx, s[kb][t][0], s[kb][t][1]);
            mx = max3(mx, s[kb][t][2], s[kb][t][3]);
        }
        return wave_max_g(mx);                  // over the 4 lane groups: all keys of the tile
    };
    // p of key block kb: exponentials, their sum (which is also what the ceiling is tested on: a lane's 4 KB
    // values of a tile are all below their sum), the PV B fragment
    float psum[NTQ];
    auto exp_block = [&](auto moved, int t, int kb, float d, f32x4 (&s)[KB][NTQ], u32x4 (&pf)[PG][NTQ]) {
        constexpr bool MOVED = decltype(moved)::value;      // re-basing: the scores are d above the new shift
        const float p0 = __builtin_amdgcn_exp2f(MOVED ? s[kb][t][0] - d : s[kb][t][0]);
        const float p1 = __builtin_amdgcn_exp2f(MOVED ? s[kb][t][1] - d : s[kb][t][1]);
        const float p2 = __builtin_amdgcn_exp2f(MOVED ? s[kb][t][2] - d : s[kb][t][2]);
        const float p3 = __builtin_amdgcn_exp2f(MOVED ? s[kb][t][3] - d : s[kb][t][3]);
        psum[t] += (p0 + p1) + (p2 + p3);
        if constexpr (P::kSplit) {
            // (a tile is ONE 32-key group: pf[0] the hi plane of its 8 k-slots per lane, pf[1] the lo plane)
            uint32_t h01, l01, h23, l23;
            P::split2(p0, p1, h01, l01);
            P::split2(p2, p3, h23, l23);
            if (kb & 1) { pf[0][t].z = h01; pf[0][t].w = h23; pf[1][t].z = l01; pf[1][t].w = l23; }
            else        { pf[0][t].x = h01; pf[0][t].y = h23; pf[1][t].x = l01; pf[1][t].y = l23; }
        } else if constexpr (P::kIsBF16) {
            const uint32_t lo = P::pack2(p0, p1), hi = P::pack2(p2, p3);
            if (kb & 1) { pf[kb >> 1][t].z = lo; pf[kb >> 1][t].w = hi; }
            else        { pf[kb >> 1][t].x = lo; pf[kb >> 1][t].y = hi; }
        } else {
            pf[kb][t] = u32x4{__float_as_uint(p0), __float_as_uint(p1), __float_as_uint(p2), __float_as_uint(p3)};
        }
    };
    // Softmax of tile kt in NPIECE pieces, issued between the MFMAs of the next tile's scores:
    // per query block t, piece 0 masks (tiles that reach past the valid keys / the causal diagonal only),
    // piece 1 + kb exponentiates key block kb
    constexpr int PPT = 1 + KB;
    constexpr int NPIECE = NTQ * PPT;
    auto softmax_piece = [&](auto jc, int kt, bool need_mask, f32x4 (&s)[KB][NTQ], u32x4 (&pf)[PG][NTQ]) {
        constexpr int j = decltype(jc)::value;
        constexpr int t = j / PPT, r = j % PPT;
        if constexpr (r == 0) {
            if (need_mask) mask_tile(t, kt, s);
            psum[t] = 0.f;
        } else {
            exp_block(std::false_type{}, t, r - 1, 0.f, s, pf);
        }
    };
    // some p of the tile is past the ceiling: move the shift to the tile's maximum
    auto rebase = [&](f32x4 (&s)[KB][NTQ], f32x4 (&nxt)[KB][NTQ], u32x4 (&pf)[PG][NTQ]) {
#pragma unroll
        for (int t = 0; t < NTQ; ++t) {
            const float d = fmaxf(tile_max(t, s), 0.f);       // (a fully masked tile: -inf -> 0)
            const float alpha = __builtin_amdgcn_exp2f(-d);
            lrun[t] *= alpha;
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                oacc[db][t][0] *= alpha; oacc[db][t][1] *= alpha;
                oacc[db][t][2] *= alpha; oacc[db][t][3] *= alpha;
            }
            psum[t] = 0.f;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                exp_block(std::true_type{}, t, kb, d, s, pf);
                // the next tile's scores were started from the old shift
                nxt[kb][t][0] -= d; nxt[kb][t][1] -= d; nxt[kb][t][2] -= d; nxt[kb][t][3] -= d;
            }
            shift[t] += d;
            cinit[t] = f32x4{-shift[t], -shift[t], -shift[t], -shift[t]};
        }
    };

    // Three-stage loop of the 16-bit modes (round 5): iteration kt computes the scores of tile kt + 2, exponentiates tile
    // kt + 1 and accumulates tile kt -- the exponentials of a tile no longer have to fit between the MFMAs of ONE
    // phase (they rode on the score MFMAs only, clumped ~14 instructions to a step, while the P V MFMAs ran bare), they
    // are spread as micro-operations (two exponentials, their sum, one pack) over BOTH phases, one to every other step.
    constexpr bool STAGE3 = P::kIsBF16 && !P::kSplit && NTQ == 1;      // (32 queries per wave: 256 registers do not hold a third tile's state)
    // micro-operation u = (t, kb, h): p of the scores 2 h, 2 h + 1 of key block kb for query block t
    constexpr int NUOP = NTQ * KB * 2;
    auto exp_half = [&](auto u_tag, f32x4 (&s)[KB][NTQ], u32x4 (&pf)[PG][NTQ]) {
        constexpr int u = decltype(u_tag)::value, t = u / (2 * KB), kb = (u / 2) % KB, h = u % 2;
        const float p0 = __builtin_amdgcn_exp2f(s[kb][t][2 * h]), p1 = __builtin_amdgcn_exp2f(s[kb][t][2 * h + 1]);
        psum[t] += p0 + p1;
        const uint32_t w = P::pack2(p0, p1);
        if constexpr ((kb & 1) == 0 && h == 0) pf[kb >> 1][t].x = w;
        else if constexpr ((kb & 1) == 0) pf[kb >> 1][t].y = w;
        else if constexpr (h == 0) pf[kb >> 1][t].z = w;
        else pf[kb >> 1][t].w = w;
    };
    // the micro-operations of global step G of NS: [ceil(G NUOP / NS), ceil((G + 1) NUOP / NS))
    auto exp_ops = [&]<int G, int NS>(f32x4 (&s)[KB][NTQ], u32x4 (&pf)[PG][NTQ]) {
        constexpr int lo = (G * NUOP + NS - 1) / NS, hi = ((G + 1) * NUOP + NS - 1) / NS;
        [&]<int... U>(std::integer_sequence<int, U...>) {
            (exp_half(std::integral_constant<int, lo + U>{}, s, pf), ...);
        }(std::make_integer_sequence<int, hi - lo>{});
    };

    if (ntiles > 0) { stage_k(0); stage_v(0); }
    if (ntiles > 1) stage_k(1);
    // the first scores need the Q rows and K tile 0 only: V^T tile 0 and K tile 1 (the younger requests: vector-memory
    // operations complete in issue order) travel on under them
    {
        constexpr int OPS = TB / 1024 / NW;                 // DMA instructions of a tile per wave
        static_assert(2 * OPS < 64, "vmcnt range");
        if (ntiles > 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * OPS) : "memory");
        else if (ntiles > 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(OPS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int t = 0; t < NTQ; ++t)
#pragma unroll
            for (int kg = 0; kg < DG; ++kg) {
                asm volatile("" : "+v"(qf[kg][t]));
                if (!q_ok[t]) qf[kg][t] = u32x4{0u, 0u, 0u, 0u};
            }
        __syncthreads();
    }

    f32x4 scur[KB][NTQ], snext[KB][NTQ];
    if (ntiles > 0) {
        scores(0, scur, [](auto) {});
        // the shift: the first tile's exact maximum per query (0 for a query without a valid key in it)
        const bool mask0 = KT > w.valid || (a.causal && KT > qw0);
#pragma unroll
        for (int t = 0; t < NTQ; ++t) {
            if (mask0) mask_tile(t, 0, scur);
            const float m0 = tile_max(t, scur);
            shift[t] = (m0 == -INFINITY) ? 0.f : m0;
            cinit[t] = f32x4{-shift[t], -shift[t], -shift[t], -shift[t]};
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                scur[kb][t][0] -= shift[t]; scur[kb][t][1] -= shift[t];
                scur[kb][t][2] -= shift[t]; scur[kb][t][3] -= shift[t];
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // V^T tile 0, K tile 1
    __syncthreads();                       // K buffer 0 is re-filled by iteration 0's DMA

#ifdef PPG_ATTN_TIMING
    auto stamp = [&](int kt, int k) {
        if (a.dbg && blockIdx.x == 0 && lane == 0 && wave < 4 && kt >= 2 && kt < 4)
            a.dbg[(wave * 2 + (kt - 2)) * 8 + k] = __builtin_amdgcn_s_memtime();
    };
#else
    auto stamp = [&](int, int) {};
#endif
#ifdef PPG_ATTN_TIMING
    const unsigned long long wg_t1 = __builtin_amdgcn_s_memrealtime();   // prologue done (Q, first tiles, first scores)
#endif
    if constexpr (STAGE3) {
        constexpr int NPV = PG * DB;                        // fragments of a V^T tile
        auto mask_init = [&](int kt, f32x4 (&s)[KB][NTQ]) {     // tile kt's padding / causal mask, its sums from zero
            const bool need = (kt + 1) * KT > w.valid || (a.causal && (kt + 1) * KT > qw0);
#pragma unroll
            for (int t = 0; t < NTQ; ++t) {
                if (need) mask_tile(t, kt, s);
                psum[t] = 0.f;
            }
        };
        auto pv = [&](int kt, u32x4 (&pf)[PG][NTQ], auto filler) {
            using LV = FragLayout<ROWV, DB>;
            uint32_t fbv[LV::VAR];
            LV::bases(lds0 + 2 * TB + (kt & 1) * TB, idx, g, fbv);
            lds_stream<LV, NPV, 6>(fbv, [&](auto ic, const u32x4& vf) {
                constexpr int i = decltype(ic)::value;
#pragma unroll
                for (int t = 0; t < NTQ; ++t) mma_kg<P, i / DB, false, 0>(oacc[i % DB][t], vf, pf, t, oacc[i % DB][t]);
                filler(ic);
            });
        };
        // a p of the tile just exponentiated past the ceiling (wave-uniform test; the branch is cold)
        auto settle = [&](f32x4 (&s)[KB][NTQ], f32x4 (&nxt)[KB][NTQ], u32x4 (&pf)[PG][NTQ]) {
            bool high = false;
#pragma unroll
            for (int t = 0; t < NTQ; ++t) high |= psum[t] > (a.rebase_always ? 1.0f : P::kProbCeil);
            if (__any(high)) rebase(s, nxt, pf);
#pragma unroll
            for (int t = 0; t < NTQ; ++t) lrun[t] += psum[t];
        };
        u32x4 pfa[PG][NTQ], pfb[PG][NTQ];
        // tile 0's p (under the scores of tile 1, when there is one): the state the loop starts from is
        // pfa = p of tile 0, scur = the raw scores of tile 1
        if (ntiles > 0) {
            if (ntiles > 2) stage_k(2);
            mask_init(0, scur);
            if (ntiles > 1) {
                scores(1, snext, [&](auto ic) { exp_ops.template operator()<decltype(ic)::value, NSTEP>(scur, pfa); });
            } else {
                [&]<int... U>(std::integer_sequence<int, U...>) { (exp_half(std::integral_constant<int, U>{}, scur, pfa), ...); }(std::make_integer_sequence<int, NUOP>{});
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                    for (int t = 0; t < NTQ; ++t) snext[kb][t] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            settle(scur, snext, pfa);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int t = 0; t < NTQ; ++t) scur[kb][t] = snext[kb][t];
        }
        // iteration kt: DMA K(kt+3), V(kt+1) | scores(kt+2) and O += V P(kt), the exponentials of tile kt+1 in the gaps of both
        for (int kt = 0; kt < ntiles; ++kt) {
            if (kt + 3 < ntiles) stage_k(kt + 3);
            if (kt + 1 < ntiles) stage_v(kt + 1);
            if (kt + 2 < ntiles) {
                constexpr int NS = NSTEP + NPV;
                mask_init(kt + 1, scur);
                scores(kt + 2, snext, [&](auto ic) { exp_ops.template operator()<decltype(ic)::value, NS>(scur, pfb); });
                pv(kt, pfa, [&](auto ic) { exp_ops.template operator()<NSTEP + decltype(ic)::value, NS>(scur, pfb); });
                settle(scur, snext, pfb);
            } else if (kt + 1 < ntiles) {
                mask_init(kt + 1, scur);
                pv(kt, pfa, [&](auto ic) { exp_ops.template operator()<decltype(ic)::value, NPV>(scur, pfb); });
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                    for (int t = 0; t < NTQ; ++t) snext[kb][t] = f32x4{0.f, 0.f, 0.f, 0.f};
                settle(scur, snext, pfb);
            } else {
                pv(kt, pfa, [](auto) {});
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int t = 0; t < NTQ; ++t) scur[kb][t] = snext[kb][t];
#pragma unroll
            for (int pg = 0; pg < PG; ++pg)
#pragma unroll
                for (int t = 0; t < NTQ; ++t) pfa[pg][t] = pfb[pg][t];
        }
    } else
    // iteration kt: DMA K(kt+2), V(kt+1) | scores(kt+1) with softmax(kt) in its MFMA gaps | O += V P(kt)
    for (int kt = 0; kt < ntiles; ++kt) {
        stamp(kt, 0);
        if (kt + 2 < ntiles) stage_k(kt + 2);
        if (kt + 1 < ntiles) stage_v(kt + 1);
        stamp(kt, 1);
        u32x4 pf[PG][NTQ];
        const bool need_mask = (kt + 1) * KT > w.valid || (a.causal && (kt + 1) * KT > qw0);
        if (kt + 1 < ntiles) {
            scores(kt + 1, snext, [&](auto ic) {
                constexpr int i = decltype(ic)::value;
                // pieces [(i * NPIECE) / NSTEP, ((i + 1) * NPIECE) / NSTEP) ride on stream step i
                constexpr int lo = (i * NPIECE) / NSTEP, hi = ((i + 1) * NPIECE) / NSTEP;
                if constexpr (hi > lo) softmax_piece(std::integral_constant<int, lo>{}, kt, need_mask, scur, pf);
                if constexpr (hi > lo + 1) softmax_piece(std::integral_constant<int, lo + 1>{}, kt, need_mask, scur, pf);
                static_assert(hi <= lo + 2, "at most two pieces per step");
            });
        } else {
            [&]<int... J>(std::integer_sequence<int, J...>) {
                (softmax_piece(std::integral_constant<int, J>{}, kt, need_mask, scur, pf), ...);
            }(std::make_integer_sequence<int, NPIECE>{});
        }
        // a p past the ceiling (wave-uniform test; the branch is cold)
        bool high = false;
#pragma unroll
        for (int t = 0; t < NTQ; ++t) high |= psum[t] > (a.rebase_always ? 1.0f : P::kProbCeil);
        if (__any(high)) rebase(scur, snext, pf);
#pragma unroll
        for (int t = 0; t < NTQ; ++t) lrun[t] += psum[t];
        stamp(kt, 2);
        using LV = FragLayout<ROWV, DB>;
        uint32_t fbv[LV::VAR];
        LV::bases(lds0 + 2 * TB + (kt & 1) * TB, idx, g, fbv);
        lds_stream<LV, PG * DB, 6>(
            fbv,
            [&](auto ic, const u32x4& vf) {
                constexpr int i = decltype(ic)::value;
#pragma unroll
                for (int t = 0; t < NTQ; ++t) mma_kg<P, i / DB, false, 0>(oacc[i % DB][t], vf, pf, t, oacc[i % DB][t]);
            });
        stamp(kt, 3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(kt, 4);
        __syncthreads();
        stamp(kt, 5);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int t = 0; t < NTQ; ++t) scur[kb][t] = snext[kb][t];
    }

#ifdef PPG_ATTN_TIMING
    const unsigned long long wg_t2 = __builtin_amdgcn_s_memrealtime();   // tile loop done
#endif
#pragma unroll
    for (int t = 0; t < NTQ; ++t) {
        if ((qw0 + 16 * t) >= ((w.frames + 15) & ~15)) continue;
        const int m = w.tok_off + qw0 + 16 * t + idx;
        const float l = wave_sum_g(lrun[t]);
        const float inv = l > 0.f ? 1.0f / l : 0.f;
        // V^T rows are in tile order: block db holds head features pair_feature(db, g) + r
        PairStore<P> pair;
#pragma unroll
        for (int db = 0; db < DB; ++db) {
            const int n = head * DH + pair_feature(db, g);
            pair.put(a.ao + ((a.ao_tiled && P::kIsBF16) ? ao32_byte(m, n & ~7, a.H) : (size_t)m * a.H * P::kBytes + P::row_byte(n & ~7)), db & 1,
                     oacc[db][t][0] * inv, oacc[db][t][1] * inv,
                     oacc[db][t][2] * inv, oacc[db][t][3] * inv);
        }
    }
#ifdef PPG_ATTN_TIMING
    if (a.dbg && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long* rec = a.dbg + 64 + 4 * (size_t)blockIdx.x;
        rec[0] = wg_t0; rec[1] = __builtin_amdgcn_s_memrealtime(); rec[2] = (unsigned long long)w.valid | ((wg_t1 - wg_t0) << 16) | ((wg_t2 - wg_t0) << 40);   // (10 ns ticks)
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        rec[3] = hwid;
    }
#endif
}

// One launch index over (item, head), heads innermost: the items arrive longest first, and with a
// 2-D grid the whole first head (long AND short items) was dispatched before the second head's
// long items -- 64 of those then started 18 us into a 38 us launch (tools/attn_timeline.py).
template <class P, int NTQ, int DH>
__global__ __launch_bounds__(256, 2) void attn_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int item_index = blockIdx.x / a.heads;
    const AttnItem item = a.items[item_index];
    attn_body<P, NTQ, DH, 4>(a, item, blockIdx.x - item_index * a.heads, smem);
}
// Head dimension 128 with query tiles of two widths: 128 queries (32 per wave), or 64 (16 per wave) for the windows
// the planner marks narrow: windows that fit one narrow tile, and the short windows of a batch, which run last
// (longest first) on a chip the long ones no longer fill -- a wave's time is its queries x the window's keys, so
// half the queries per wave on twice the workgroups shortens that tail (32 x 1000 frames: 10.4 -> 8.6 us of 32).
// (Workgroups of 8 waves / 256 queries, which halve the K and V^T bytes streamed, measured slower: 26 vs 24 us per
// long item -- the kernel is bound by the instructions its SIMDs issue, not by the tile traffic.)
template <class P>
__global__ __launch_bounds__(256, 2) void attn_mixed_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int item_index = blockIdx.x / a.heads;
    const AttnItem item = a.items[item_index];
    const int head = blockIdx.x - item_index * a.heads;
    if (item.narrow) attn_body<P, 1, 128, 4>(a, item, head, smem);
    else attn_body<P, 2, 128, 4>(a, item, head, smem);
}


template <class P>
hipError_t launch_attn_p(const AttnArgs& a, int nitems, int heads, int head_dim, hipStream_t s) {
    if (head_dim == 128) {
        hipLaunchKernelGGL(attn_mixed_kernel<P>, dim3(nitems * heads), dim3(256), 65536, s, a);
    } else if (head_dim == 256) {
        hipLaunchKernelGGL((attn_kernel<P, 1, 256>), dim3(nitems * heads), dim3(256), 65536, s, a);
    } else if (head_dim == 64) {             // wav2vec2 body: 12 heads of 64
        // (a kernel holding both tile widths, as at d = 128, allocates 256 registers with spills for the 128-query body
        // and runs the 64-query one 6 % slower; 128-query tiles themselves measured slower at 16 x 499 frames)
        hipLaunchKernelGGL((attn_kernel<P, 1, 64>), dim3(nitems * heads), dim3(256), 65536, s, a);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace

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

namespace ppg {

int attn_query_tile(int head_dim) { return head_dim == 128 ? 128 : 64; }   // (d = 64 and d = 256: 16 queries per wave)

hipError_t launch_attn(int precision, const AttnArgs& args, int nitems, int heads, int head_dim, hipStream_t s) {
    // PPGS_AMD_ATTN_REBASE=always (read per launch: the tests switch it): the classic online softmax's worth of
    // re-basing, to exercise the path trained logits never take
    AttnArgs a = args;
    const char* mode = getenv("PPGS_AMD_ATTN_REBASE");
    a.rebase_always = (mode && strcmp(mode, "always") == 0) ? 1 : 0;
    if (precision == PPG_PRECISION_BF16) return launch_attn_p<PrecBF16>(a, nitems, heads, head_dim, s);
#if PPG_X2
    if (precision == PPG_PRECISION_FP16X2) {
        if (head_dim == 256) {       // 32-key tiles of 32 KiB (attn_tile_bytes): one workgroup per CU
            auto kern = attn_kernel<PrecX2, 1, 256>;
            constexpr size_t lds = 4 * attn_tile_bytes<PrecX2, 256>();
            static ppg::LdsLimit limit;
            const hipError_t e = limit.ensure(reinterpret_cast<const void*>(kern), lds);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(kern, dim3(nitems * heads), dim3(256), lds, s, a);
            return hipGetLastError();
        }
        if (head_dim == 64) {        // wav2vec2 body: 32-key tiles of 8 KiB
            constexpr size_t lds64 = 4 * attn_tile_bytes<PrecX2, 64>();
            hipLaunchKernelGGL((attn_kernel<PrecX2, 1, 64>), dim3(nitems * heads), dim3(256), lds64, s, a);
            return hipGetLastError();
        }
        if (head_dim != 128) return hipErrorInvalidValue;
        hipLaunchKernelGGL(attn_mixed_kernel<PrecX2>, dim3(nitems * heads), dim3(256), 65536, s, a);
        return hipGetLastError();
    }
#endif
#if PPG_OTHER_PRECISIONS
    if (precision == PPG_PRECISION_FP16) return launch_attn_p<PrecF16>(a, nitems, heads, head_dim, s);
    return launch_attn_p<PrecF32>(a, nitems, heads, head_dim, s);
#else
    return hipErrorInvalidValue;
#endif
}

}  // namespace ppg
