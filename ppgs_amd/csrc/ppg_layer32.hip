// The feature-split layer kernel for gfx950: everything between two attention
// kernels of one encoder layer (reference: torch TransformerEncoderLayer._sa_block's
// out_proj, norm1, _ff_block, norm2 -- SURVEY.md 2.3 rows M5/M6 -- and the next
// layer's in_proj) for a 160-token workgroup (hidden 256; 96 tokens at hidden 512),
// on v_mfma_f32_32x32x16.
//
// Why this shape (measured, tools/dma_probe.hip + tools/ffn32_probe.hip):
//  * one wave issues v_mfma_f32_16x16x32_bf16 every ~21.6 cycles but
//    v_mfma_f32_32x32x16_bf16 every 32: the same FLOPs in 74 % of the cycles;
//  * a 1 KiB global_load_lds costs the issuing wave ~60 cycles during which its SIMD's
//    matrix pipe idles (one 512-register wave per SIMD), 21 % of the token-split
//    kernel's chunk loop -- a plain global_load_dwordx4 into registers between two
//    32x32 MFMAs costs next to nothing.
// So the operands trade places against ffn_body (ppg_kernels.hip): the four waves
// (one per SIMD) all work on ALL 160 tokens -- 5 blocks of 32, the MFMA's N -- and
// split the FEATURES: wave w owns output features 64w..64w+63 of every 256-wide
// result and hidden rows 32w..32w+31 of every 128-hidden chunk.  Weights never touch
// LDS: a wave loads only its own rows, as ready-made A fragments the host packed in
// consumption order (1 KiB = one coalesced global_load_dwordx4 per fragment).
// Activations are the B operands and live in LDS as fragments too (lane-linear, no
// swizzle needed): the 80 KiB token panel (attention output, then x1, then x2) and
// the chunk's 40 KiB of h.  The work is balanced by construction (no 3/3/2/2 roles),
// LayerNorm statistics cross the waves through 5 KiB of LDS.
//
// Accumulator layout (32x32 C): lane l holds token l & 31; register i = 4q + r of row
// block rb is tile row 8q + 4(l >> 5) + r, which the host maps to natural feature
//   64 w + 32 rb + 16 (l >> 5) + 4 q + r           ("phi": 16 consecutive features per lane)
// so a lane's 16 values of a block are 64 contiguous bytes of an fp32 row.  Packed to
// 16 bits, registers (q = 2s, 2s+1) are exactly the B fragment of K-step s: the
// accumulator of one GEMM is the operand of the next after ONE ds_write_b128.
#include "ppg_layer32.h"

namespace {

template <class P, int HIDT, bool QKV, int TBS = tile_blocks(HIDT)>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void layer32_kernel(Layer32Args a) {
    using G = Geo<HIDT, TBS>;
    constexpr int RB = G::RB, KS = G::KS, KH = G::KH, TB = G::TBN, TOKS = G::TOKS;
    // Sub-tile workgroups (TBS < the tile's token blocks): workgroup b takes blocks tb0 .. tb0 + nblk - 1 of tile
    // b / NSUB; X32 / X16 / AO32 keep the full tile's order.  A block past the tile's end (the last sub-tile of a
    // 5-block tile split 2 + 2 + 1) is computed on the tile's block 0 again and never stored.
    constexpr int TBT = tile_blocks(HIDT);
    constexpr bool SUB = TBS != TBT;
    constexpr int NSUB = (TBT + TBS - 1) / TBS;
    static_assert(!SUB || (HIDT == 256 && TBS == 2), "sub-tile workgroups: hidden 256, two token blocks");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tok = lane & 31, hh = lane >> 5;
    const int tile = SUB ? (int)blockIdx.x / NSUB : (int)blockIdx.x;
    const int tb0 = SUB ? ((int)blockIdx.x % NSUB) * TBS : 0;
    const int nblk = SUB ? min(TBS, TBT - tb0) : TBS;
    auto gblock = [&](int t) { return SUB ? tb0 + (t < nblk ? t : -tb0) : t; };      // the tile's block behind local block t
    const int m0 = tile * (32 * TBT) + 32 * tb0;
    const int NCH = a.F / HC;
    const uint32_t lds0 = lds_addr32(smem);
    const uint32_t voff = lane * 16;
    const uint32_t pb0 = lds0 + G::L_ACT + lane * 16, pb1 = pb0 + 65536;   // panel fragments
    const uint32_t hb0 = lds0 + G::L_H + lane * 16;                         // h fragments
    float* lnp1 = reinterpret_cast<float*>(smem + G::L_LNP1);
    float* lnp2 = reinterpret_cast<float*>(smem + G::L_LNP2);
    float* stats = reinterpret_cast<float*>(smem + G::L_STATS);
    const int fbase = 32 * RB * wave;               // first feature of this wave
    // the W_qkv fragments of the tail's first half-step (they travel into set 1 under the last FFN chunk)
    [[maybe_unused]] const char* wq_first = a.wq_img + ((size_t)wave * 3 * RB * KS) * 1024;

#ifdef PPG_FFN_TIMING
    auto pstamp = [&](int k) {
        if (a.dbg && blockIdx.x == 0 && lane == 0) a.dbg[128 + wave * 16 + k] = __builtin_amdgcn_s_memtime();
    };
#else
    auto pstamp = [&](int) {};
#endif
    pstamp(0);

    u32x4 w1f[16], w2f[16];
    // 16 fragments (1 KiB each, consecutive) -> one register set
    auto load16 = [&](u32x4 (&wf)[16], const char* base) {
        [&]<int... K>(std::integer_sequence<int, K...>) { (gload_frag<K>(wf[K], voff, base), ...); }(std::make_integer_sequence<int, 16>{});
    };
    // ---- the residual rows, when they are fp16 (X16 order: 20 KiB per wave = 80 registers): requested before anything
    // else, they travel under the attention-output DMA, the W_o fetch and the out-projection.  (Loaded inside
    // LayerNorm-1 the compiler can afford one token block's worth of registers at a time: five dependent round trips
    // of ~2.5 k cycles each while the whole chip does the same -- 12.9 k of the kernel's 200 k cycles.  As fp32 the rows
    // are 160 registers per lane, which the out-projection cannot spare.)  The W_o wait below (vmcnt(0)) covers them.
    constexpr bool XPRE = HIDT == 256;
    constexpr int NRES = XPRE ? TB * RB * 2 : 1;
    u32x4 res[NRES];
    const bool xpre = XPRE && a.x_half;
    if (xpre) {
        const char* xbase = reinterpret_cast<const char*>(a.X) + ((size_t)tile * 4 + wave) * (TBT * RB * 2 * 1024);
        if constexpr (!SUB) {
            [&]<int... K>(std::integer_sequence<int, K...>) { (gload_frag<K>(res[K], voff, xbase), ...); }(std::make_integer_sequence<int, NRES>{});
        } else {
            [&]<int... K>(std::integer_sequence<int, K...>) {
                (gload_frag<K % (RB * 2)>(res[K], voff, xbase + (size_t)gblock(K / (RB * 2)) * (RB * 2 * 1024)), ...);
            }(std::make_integer_sequence<int, NRES>{});
        }
    }
    // ---- attention output (AO32 order: this tile's fragments are one contiguous block) -> LDS panel
    {
        const char* src = a.ao + (size_t)tile * (TBT * KS * 1024);
#pragma unroll
        for (int i = 0; i < TB * KS / 4; ++i) {
            const int p = 4 * i + wave;               // fragment (local block p / KS, K-step p % KS)
            const int q = SUB ? gblock(p / KS) * KS + p % KS : p;
            glds16(src + (size_t)q * 1024, voff, lds0 + G::L_ACT + p * 1024);
        }
    }
    f32x16 yacc[RB][TB];
    // parameters: every global load first, the LDS stores afterwards (one memory round trip, not three).  The loads
    // are unconditional on clamped indices -- a load under `if` comes out as a branch with its own wait -- and only
    // the LDS stores are predicated.
    {
        constexpr int NP = (3 * HIDT / 4 + 255) / 256;
        float4 p1[NP], p2[NP], pq[NP];
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const int i = min(tid + 256 * u, 3 * HIDT / 4 - 1);
            const int v = i / (HIDT / 4), j = i - v * (HIDT / 4);
            p1[u] = reinterpret_cast<const float4*>(v == 0 ? a.bo : (v == 1 ? a.g1 : a.e1))[j];
            p2[u] = reinterpret_cast<const float4*>(v == 0 ? a.b2 : (v == 1 ? a.g2 : a.e2))[j];
            if constexpr (QKV) pq[u] = reinterpret_cast<const float4*>(a.bq)[i];
        }
        constexpr int B1MAX = 8;                 // F <= 8192
        float4 pb[B1MAX];
#pragma unroll
        for (int u = 0; u < B1MAX; ++u) pb[u] = reinterpret_cast<const float4*>(a.b1)[min(tid + 256 * u, a.F / 4 - 1)];
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const int i = tid + 256 * u;
            if (i < 3 * HIDT / 4) {
                reinterpret_cast<float4*>(lnp1)[i] = p1[u];
                reinterpret_cast<float4*>(lnp2)[i] = p2[u];
                if constexpr (QKV) reinterpret_cast<float4*>(smem + G::L_BQ)[i] = pq[u];
            }
        }
#pragma unroll
        for (int u = 0; u < B1MAX; ++u)
            if (tid + 256 * u < a.F / 4) reinterpret_cast<float4*>(smem + G::L_B1)[tid + 256 * u] = pb[u];
    }
    // ---- out-projection: y[rb][tb] = W_o[rows of this wave] ao.  W_o fragments go straight to
    // registers, two row blocks x 16 K-steps at a time (image order [wave][rb][ks]).  They are issued
    // only here: registers written by an asm load must not be moved or spilled before the data is in,
    // and the compiler does not know they are in flight -- nothing register-hungry may sit between
    // such a load and its wait.
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    [&]<int... R>(std::integer_sequence<int, R...>) {
        ([&] {
            constexpr int rbp = R / KH, kh = R % KH;             // row-block pair, K half
            const char* base = a.wo_img + ((size_t)wave * RB * KS) * 1024;
            load16(w1f, base + ((size_t)(2 * rbp) * KS + 16 * kh) * 1024);
            load16(w2f, base + ((size_t)(2 * rbp + 1) * KS + 16 * kh) * 1024);
            vm_wait_all(w1f);
            vm_wait_all(w2f);
            if constexpr (R == 0) {
#pragma unroll
                for (int k = 0; k < NRES; ++k) asm volatile("" : "+v"(res[k]));
                __syncthreads();                             // panel and parameters are in LDS
                pstamp(1);
            }
            if (!(PPG_DBG(a) & 1)) {
                stream<OffPanel<KS, 16 * kh, 0, TB>, 16 * TB, 6>(pb0, pb1, [&](auto ic, const u32x4& bf) {
                    constexpr int i = decltype(ic)::value;
                    constexpr int ks = i / TB, tb = i % TB;
                    if constexpr (ks == 0 && kh == 0) {
                        yacc[2 * rbp][tb] = P::mma32(w1f[0], bf, zero);
                        yacc[2 * rbp + 1][tb] = P::mma32(w2f[0], bf, zero);
                    } else {
                        yacc[2 * rbp][tb] = P::mma32(w1f[ks], bf, yacc[2 * rbp][tb]);
                        yacc[2 * rbp + 1][tb] = P::mma32(w2f[ks], bf, yacc[2 * rbp + 1][tb]);
                    }
                });
            }
        }(), ...);
    }(std::make_integer_sequence<int, (RB / 2) * KH>{});
    pstamp(2);

    // ---- LayerNorm over the HIDT features of a token, spread over the four waves ---------------
    // acc <- LN(acc + bias [+ X]) * gamma + beta; `emit(tb, rb, y)` gets the 16 results of a block.
    // One pass for the statistics (sum and sum of squares in fp32: the inputs are O(1) residual
    // sums, var = E[v^2] - mean^2 loses nothing the 16-bit operands have not lost already), one
    // exchange through LDS.
    float* xt = a.X + ((size_t)tile * 4 + wave) * (TBT * RB * 4 * 256) + lane * 4;   // this lane's X32 slots of the tile
    char* xh = reinterpret_cast<char*>(a.X) + ((size_t)tile * 4 + wave) * (TBT * RB * 2 * 1024) + lane * 16;   // ... X16 slots
    auto layer_norm = [&](auto residual_tag, auto xhalf_tag, const float* lnp, auto emit) {
        constexpr bool RES = decltype(residual_tag)::value;
        constexpr bool XH = decltype(xhalf_tag)::value;
        constexpr int STAMP = RES ? 7 : 9;
        // the lane's parameter quads: in registers for all token blocks at hidden 256 (2 x 4 quads),
        // re-read from LDS per block at hidden 512 (they would take 192 registers)
        constexpr bool KEEP = RB == 2;
        float4 bias4[KEEP ? RB : 1][4], gv[KEEP ? RB : 1][4], ev[KEEP ? RB : 1][4];
        auto quad = [&](int which, int rb, int q) { return *reinterpret_cast<const float4*>(lnp + which * HIDT + fbase + 32 * rb + 16 * hh + 4 * q); };
        if constexpr (KEEP) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int q = 0; q < 4; ++q) bias4[rb][q] = quad(0, rb, q);
        }
#pragma unroll
        for (int t = 0; t < TB; ++t) {
            f32x2 sum2 = {0.f, 0.f}, sq2 = {0.f, 0.f};
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    float r8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    [[maybe_unused]] uint32_t w4[4] = {0u, 0u, 0u, 0u};
                    if constexpr (RES && XH) {          // X16 order: one contiguous KiB of fp16 per load instruction
                        u32x4 rv;
                        if constexpr (XPRE) rv = res[(t * RB + rb) * 2 + s2];          // (requested at the top of the kernel)
                        else rv = *reinterpret_cast<const u32x4*>(xh + ((gblock(t) * RB + rb) * 2 + s2) * 1024);
                        w4[0] = rv.x; w4[1] = rv.y; w4[2] = rv.z; w4[3] = rv.w;
                    } else if constexpr (RES) {         // X32 order: one contiguous KiB per load instruction
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const float4 rv = *reinterpret_cast<const float4*>(xt + ((gblock(t) * RB + rb) * 4 + 2 * s2 + e) * 256);
                            r8[4 * e + 0] = rv.x; r8[4 * e + 1] = rv.y; r8[4 * e + 2] = rv.z; r8[4 * e + 3] = rv.w;
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int q = 2 * s2 + e;
                        float4 bv;
                        if constexpr (KEEP) bv = bias4[rb][q]; else bv = quad(0, rb, q);
                        if constexpr (RES && XH) {
                            // acc + bias + fp16 residual: the conversion rides in the add (v_fma_mix_f32: src0 an fp16 half
                            // selected by op_sel, times 1.0, plus an fp32)
                            const float bq4[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const float sum = yacc[rb][t][4 * q + k] + bq4[k];
                                float out;
                                if (k & 1) asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(out) : "v"(w4[2 * e + k / 2]), "v"(sum));
                                else asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(out) : "v"(w4[2 * e + k / 2]), "v"(sum));
                                yacc[rb][t][4 * q + k] = out;
                            }
                        } else
                        {
                        yacc[rb][t][4 * q + 0] += bv.x + r8[4 * e + 0]; yacc[rb][t][4 * q + 1] += bv.y + r8[4 * e + 1];
                        yacc[rb][t][4 * q + 2] += bv.z + r8[4 * e + 2]; yacc[rb][t][4 * q + 3] += bv.w + r8[4 * e + 3];
                        }
                        stat4(sum2, sq2, yacc[rb][t][4 * q + 0], yacc[rb][t][4 * q + 1], yacc[rb][t][4 * q + 2], yacc[rb][t][4 * q + 3]);
                    }
                }
            const float sum = pair_sum(hsum2(sum2));
            const float sq = pair_sum(hsum2(sq2));
            if (hh == 0) {
                stats[wave * TOKS + 32 * t + tok] = sum;
                stats[4 * TOKS + wave * TOKS + 32 * t + tok] = sq;
            }
        }
        if constexpr (KEEP) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int q = 0; q < 4; ++q) { gv[rb][q] = quad(1, rb, q); ev[rb][q] = quad(2, rb, q); }
        }
        pstamp(STAMP);
        __syncthreads();
        pstamp(STAMP + 1);
#pragma unroll
        for (int t = 0; t < TB; ++t) {
            const float* s = stats + 32 * t + tok;
            const float mean = ((s[0] + s[TOKS]) + (s[2 * TOKS] + s[3 * TOKS])) * (1.0f / HIDT);
            const float ex2 = ((s[4 * TOKS] + s[5 * TOKS]) + (s[6 * TOKS] + s[7 * TOKS])) * (1.0f / HIDT);
            // (v_rsq_f32, 1 ulp: the IEEE division and square root are ~30 instructions per token block of a phase
            // in which the matrix pipe idles)
            const float rstd = __builtin_amdgcn_rsqf(fmaxf(ex2 - mean * mean, 0.f) + kLnEps32);
            const float shift = -mean * rstd;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float4 g4, e4;
                    if constexpr (KEEP) { g4 = gv[rb][q]; e4 = ev[rb][q]; } else { g4 = quad(1, rb, q); e4 = quad(2, rb, q); }
                    yacc[rb][t][4 * q + 0] = fmaf(fmaf(yacc[rb][t][4 * q + 0], rstd, shift), g4.x, e4.x);
                    yacc[rb][t][4 * q + 1] = fmaf(fmaf(yacc[rb][t][4 * q + 1], rstd, shift), g4.y, e4.y);
                    yacc[rb][t][4 * q + 2] = fmaf(fmaf(yacc[rb][t][4 * q + 2], rstd, shift), g4.z, e4.z);
                    yacc[rb][t][4 * q + 3] = fmaf(fmaf(yacc[rb][t][4 * q + 3], rstd, shift), g4.w, e4.w);
                }
                emit(t, rb, yacc[rb][t]);
            }
        }
    };
    // the 16 results of block (tb, rb), packed: K-steps 2 RB w + 2 rb, + 1 of the token panel
    // (slot 4e + r of K-step s' = register 4 (2s' + e) + r = natural feature 32 (ks/2) + 16 hh + 8 (ks%2) + 4e + r)
    auto panel_write = [&](int t, int rb, const f32x16& y) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const u32x4 frag = u32x4{P::pack2(y[8 * s + 0], y[8 * s + 1]), P::pack2(y[8 * s + 2], y[8 * s + 3]),
                                     P::pack2(y[8 * s + 4], y[8 * s + 5]), P::pack2(y[8 * s + 6], y[8 * s + 7])};
            const uint32_t addr = pb0 + (uint32_t)((t * KS + 2 * RB * wave + 2 * rb + s) * 1024);
            asm volatile("ds_write_b128 %0, %1" :: "v"(addr), "v"(frag) : "memory");
        }
    };
    // LayerNorm-1: the panel holds x1 afterwards (every wave finished reading the attention
    // output a barrier ago), the accumulators keep x1 as the FFN's residual
    if (a.x_half) layer_norm(std::true_type{}, std::true_type{}, lnp1, panel_write);
    else layer_norm(std::true_type{}, std::false_type{}, lnp1, panel_write);
    // W1 fragments of chunk 0 (image order [chunk][wave][ks]); nothing register-hungry follows before their wait
    load16(w1f, a.w1_img + ((size_t)wave * KS) * 1024);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    pstamp(3);

    // ---- FFN: per 128-hidden chunk  h = relu(W1c x1 + b1c)  (phase A),  y += W2c h  (phase B) ---
    // b1 of the lane's 16 hidden rows of chunk c -> the C operand of the chunk's first MFMAs
    auto bias_read = [&](u32x4 (&braw)[4], int c) {
#pragma unroll
        for (int q = 0; q < 4; ++q) ds_read128<0>(braw[q], lds0 + G::L_B1 + (uint32_t)((c * HC + 32 * wave + 8 * q + 4 * hh) * 4));
    };
    auto bias_of = [&](u32x4 (&braw)[4]) {
        f32x16 bias;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            asm volatile("" : "+v"(braw[q]));
            bias[4 * q + 0] = __uint_as_float(braw[q].x); bias[4 * q + 1] = __uint_as_float(braw[q].y);
            bias[4 * q + 2] = __uint_as_float(braw[q].z); bias[4 * q + 3] = __uint_as_float(braw[q].w);
        }
        return bias;
    };
    if constexpr (SUB) {
        // Two token blocks per workgroup leave the registers for a software pipeline over the chunks: the accumulators
        // of phase A exist twice, h in LDS twice, W1 in two register sets.  Iteration c:
        //   first half   phase A of chunk c + 1 (32 MFMAs, panel fragments) with chunk c's h -- ReLU, pack, ds_write --
        //                between the MFMAs; W2 of chunk c and W1 of chunk c + 2 requested
        //   barrier      every wave's h of chunk c is in LDS (one barrier per chunk: buffer c & 1 is next written in
        //                iteration c + 2, behind iteration c + 1's barrier, which no wave passes before all finished c)
        //   second half  phase B of chunk c (32 MFMAs, h fragments); b1 of chunk c + 2 requested
        // Chunk parity is a template argument (register sets and buffers alternate): the loop runs two chunks per trip.
        u32x4 w1g[16], w2g[16];
        f32x16 hacc[2][TB];
        u32x4 braw[4];
        auto h_write = [&](auto par_tag, auto t_tag, auto s_tag) {
            constexpr int par = decltype(par_tag)::value, t = decltype(t_tag)::value, s2 = decltype(s_tag)::value;
            const f32x16& hv = hacc[par][t];
            const u32x4 frag = u32x4{P::relu2(P::pack2(hv[8 * s2 + 0], hv[8 * s2 + 1])), P::relu2(P::pack2(hv[8 * s2 + 2], hv[8 * s2 + 3])),
                                     P::relu2(P::pack2(hv[8 * s2 + 4], hv[8 * s2 + 5])), P::relu2(P::pack2(hv[8 * s2 + 6], hv[8 * s2 + 7]))};
            const uint32_t addr = hb0 + (uint32_t)(par * (TB * 8 * 1024) + (t * 8 + 2 * wave + s2) * 1024);
            asm volatile("ds_write_b128 %0, %1" :: "v"(addr), "v"(frag) : "memory");
        };
        auto w1_of = [&](int c) { return a.w1_img + (((size_t)c * 4 + wave) * KS) * 1024; };
        auto w2_of = [&](int c) { return a.w2_img + (((size_t)c * 4 + wave) * RB * 8) * 1024; };
        // Weight traffic: 128 KiB per chunk and CU against 64 MFMAs per wave -- at the vector memory path's ~64 B / clk the
        // fragments take as long to arrive as the MFMAs to issue, and a burst of requests stalls the wave that issues it
        // (32 requests in 16 steps: 4.2 - 4.4 k cycles per chunk; spread evenly: 3.1 k).  So requests are spread over the
        // whole iteration and waited for with COUNTED vmcnt (16 requests per half: vmcnt(16) = "all but the newest half"):
        //   first half of iteration c    requests W1 of chunk c + 2 (needed at the top of iteration c + 1)
        //   second half                  requests W2 of chunk c + 1 (needed behind iteration c + 1's barrier)
        // The two W2 sets live in the ACCUMULATION registers (loads write
        // either file, an MFMA reads its A operand from either): four sets in the 256 architectural registers made the
        // compiler park one in the other file, with copies right behind the asm loads -- of data not yet there (NaNs).
        const int nrun = (PPG_DBG(a) & 2) ? 0 : NCH;
        if (nrun > 0) {
            // chunk 0's phase A (its W1 fragments were requested above into set 0); W1 of chunk 1, then W2 of chunk 0
            bias_read(braw, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const f32x16 bias0 = bias_of(braw);
            vm_wait_all(w1f);
            const char* w1n = w1_of(nrun > 1 ? 1 : 0);
            const char* w20 = w2_of(0);
            stream<OffPanel<KS, 0, 0, TB>, 16 * TB, 6>(pb0, pb1, [&](auto ic, const u32x4& bf) {
                constexpr int i = decltype(ic)::value;
                constexpr int ks = i / TB, tb = i % TB;
                if constexpr (ks == 0) hacc[0][tb] = P::mma32(w1f[0], bf, bias0);
                else hacc[0][tb] = P::mma32(w1f[ks], bf, hacc[0][tb]);
                if constexpr (i < 16) gload_frag<i>(w1g[i], voff, w1n);
                else gload_frag_acc<i - 16>(w2f[i - 16], voff, w20);
                if constexpr (i == 20) bias_read(braw, nrun > 1 ? 1 : 0);
            });
        }
        auto iter = [&](auto par_tag, auto next_tag, int c) {
            constexpr int PAR = decltype(par_tag)::value;
            constexpr bool NEXT = decltype(next_tag)::value;          // chunk c + 1 exists
            u32x4 (&wa)[16] = *(PAR ? &w1f : &w1g);                   // W1 of chunk c + 1 (set (c + 1) & 1)
            u32x4 (&wn)[16] = *(PAR ? &w1g : &w1f);                   // free: W1 of chunk c + 2 goes here
            u32x4 (&w2c)[16] = *(PAR ? &w2g : &w2f);                  // W2 of chunk c
            u32x4 (&w2n)[16] = *(PAR ? &w2f : &w2g);                  // free: W2 of chunk c + 1 goes here
            // W1 of chunk c + 2; behind the last chunks the first half-step of the Q/K/V tail, which expects it in set 0
            // (parity 0 is the second to last chunk: NCH is even); anything else harmless
            const char* nx = c + 2 < NCH ? w1_of(c + 2) : (QKV ? wq_first : w1_of(c));
            const char* nx2 = w2_of(c + 1 < NCH ? c + 1 : c);
            const int cb = c + 2 < NCH ? c + 2 : c;
            f32x16 bias;
            if constexpr (NEXT) bias = bias_of(braw);
            // in flight: [W1 of chunk c + 1 (or its stand-in)] [W2 of chunk c], 16 requests each, in this order
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
#pragma unroll
            for (int k = 0; k < 16; ++k) asm volatile("" : "+v"(wa[k]));
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (NEXT) {
                stream<OffPanel<KS, 0, 0, TB>, 16 * TB, 6>(pb0, pb1, [&](auto ic, const u32x4& bf) {
                    constexpr int i = decltype(ic)::value;
                    constexpr int ks = i / TB, tb = i % TB;
                    if constexpr (ks == 0) hacc[PAR ^ 1][tb] = P::mma32(wa[0], bf, bias);
                    else hacc[PAR ^ 1][tb] = P::mma32(wa[ks], bf, hacc[PAR ^ 1][tb]);
                    if constexpr (i % 2 == 0) gload_frag<i / 2>(wn[i / 2], voff, nx);
                    if constexpr (i % 4 == 1 && i / 4 < 2 * TB)
                        h_write(std::integral_constant<int, PAR>{}, std::integral_constant<int, (i / 4) / 2>{}, std::integral_constant<int, (i / 4) % 2>{});
                });
            } else {
                [&]<int... U>(std::integer_sequence<int, U...>) {
                    (h_write(std::integral_constant<int, PAR>{}, std::integral_constant<int, U / 2>{}, std::integral_constant<int, U % 2>{}), ...);
                }(std::make_integer_sequence<int, 2 * TB>{});
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
            // in flight: [W2 of chunk c] [W1 of chunk c + 2: 16 requests, if this iteration made them]
            if constexpr (NEXT) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < 16; ++k) asm volatile("" : "+a"(w2c[k]));
            __builtin_amdgcn_sched_barrier(0);
            stream<OffH2<TB, PAR>, 8 * TB, 6>(hb0, hb0, [&](auto ic, const u32x4& bf) {
                constexpr int i = decltype(ic)::value;
                constexpr int ks = i / TB, tb = i % TB;
                yacc[0][tb] = P::mma32(w2c[ks], bf, yacc[0][tb]);
                yacc[1][tb] = P::mma32(w2c[8 + ks], bf, yacc[1][tb]);
                if constexpr (NEXT) gload_frag_acc<i>(w2n[i], voff, nx2);
                if constexpr (i == 2) bias_read(braw, cb);
            });
        };
        for (int c = 0; c + 2 < nrun; c += 2) {
            iter(std::integral_constant<int, 0>{}, std::true_type{}, c);
            iter(std::integral_constant<int, 1>{}, std::true_type{}, c + 1);
        }
        if (nrun > 0) {
            iter(std::integral_constant<int, 0>{}, std::true_type{}, nrun - 2);
            iter(std::integral_constant<int, 1>{}, std::false_type{}, nrun - 1);
        }
    } else if constexpr (HIDT == 256) {
        // ONE fragment stream per chunk, 120 steps / 160 MFMAs, nothing of the hand-over exposed:
        //   steps   0.. 47  phase A, token blocks 0..2 (panel fragment (tb, ks), ks outer); this chunk's W2 fragments requested
        //   steps  48.. 79  phase A, token blocks 3, 4; the h of blocks 0..2 packed and written between the MFMAs
        //   step   74       barrier X: every wave's h of blocks 0..2 is in LDS (and every wave is out of the previous chunk)
        //   steps  80..103  phase B, token blocks 0..2 (h fragment (tb, ks), two MFMAs per step); the h of blocks 3, 4
        //                   written; the next chunk's W1 fragments and b1 requested
        //   step   98       barrier Y: every wave's h of blocks 3, 4 is in LDS (and every wave has issued its reads of blocks 0..2)
        //   steps 104..119  phase B, token blocks 3, 4
        // The ring reads 6 fragments ahead: fragment 80 (the first of h) is requested after step 74's MFMA, fragment 104 after
        // step 98's -- the barriers sit exactly there.  A wave's own ds_writes are complete at its barrier: LDS operations
        // complete in order and the wave has since waited for ring reads it requested after them.  h of blocks 0..2 is
        // overwritten 70 steps after barrier Y of the previous chunk, h of blocks 3, 4 after barrier X of this one.
        // Register files (round 5).  A kernel that may use all 512 registers gets the ACCUMULATION-register form for every
        // builtin MFMA; h, whose consumers are VALU instructions (pack, ReLU), then cost one v_accvgpr_read per value and
        // one v_accvgpr_write per bias value: 96 of a chunk's 180 VALU instructions, all of them in the hand-over's
        // clumps.  Phase A therefore runs on inline-asm MFMAs whose accumulators are architectural registers, and this
        // chunk's W2 fragments live in accumulation registers instead (w2a: loaded there directly, read from there by the
        // builtin MFMAs of phase B) -- the 64 architectural registers of w2f are h's for the duration of the loop.
        // The compiler inserts no wait states around an asm MFMA: a block's h is first read (h_write) at least four
        // MFMAs after the MFMA that completed it, and b1 comes straight from LDS reads behind their wait.
        u32x4 braw[4];
        u32x4 w2a[16];
        bias_read(braw, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int c = 0; c < ((PPG_DBG(a) & 2) ? 0 : NCH); ++c) {
#ifdef PPG_FFN_TIMING
            auto cstamp = [&](int k) { if (c == 4 && a.dbg && blockIdx.x == 0 && lane == 0) a.dbg[wave * 8 + k] = __builtin_amdgcn_s_memtime(); };
#else
            auto cstamp = [&](int) {};
#endif
            cstamp(0);
            const f32x16 bias = bias_of(braw);
            vm_wait_all(w1f);
            const char* w2c = a.w2_img + (((size_t)c * 4 + wave) * RB * 8) * 1024;      // this chunk's W2 fragments [rb][ks8]
            // what goes into the W1 register set during phase B: the next chunk's fragments; after the last chunk the
            // first half-step of the Q/K/V tail (or, without a tail, this chunk's again: harmless)
            const char* next1 = c + 1 < NCH ? a.w1_img + (((size_t)(c + 1) * 4 + wave) * KS) * 1024
                                : (QKV ? wq_first : a.w1_img + (((size_t)c * 4 + wave) * KS) * 1024);
            const int cn = c + 1 < NCH ? c + 1 : c;
            f32x16 hacc[TB];
            // ReLU + pack of one finished block: the accumulator IS the next GEMM's B fragment layout
            auto h_write = [&](auto t_tag, auto s_tag) {
                constexpr int t = decltype(t_tag)::value, s2 = decltype(s_tag)::value;
                const u32x4 frag = u32x4{P::relu2(P::pack2(hacc[t][8 * s2 + 0], hacc[t][8 * s2 + 1])), P::relu2(P::pack2(hacc[t][8 * s2 + 2], hacc[t][8 * s2 + 3])),
                                         P::relu2(P::pack2(hacc[t][8 * s2 + 4], hacc[t][8 * s2 + 5])), P::relu2(P::pack2(hacc[t][8 * s2 + 6], hacc[t][8 * s2 + 7]))};
                const uint32_t addr = hb0 + (uint32_t)((t * 8 + 2 * wave + s2) * 1024);
                asm volatile("ds_write_b128 %0, %1" :: "v"(addr), "v"(frag) : "memory");
            };
            // ... in two halves on two consecutive single-MFMA steps of phase A (3, 4): a whole unit -- 4 packs, 4 ReLUs, the
            // write -- is 40 issue cycles behind an MFMA that hides 32
            uint32_t hlo0 = 0, hlo1 = 0;
            auto h_half = [&](auto t_tag, auto s_tag, auto part_tag) {
                constexpr int t = decltype(t_tag)::value, s2 = decltype(s_tag)::value, part = decltype(part_tag)::value;
                if constexpr (part == 0) {
                    hlo0 = P::relu2(P::pack2(hacc[t][8 * s2 + 0], hacc[t][8 * s2 + 1]));
                    hlo1 = P::relu2(P::pack2(hacc[t][8 * s2 + 2], hacc[t][8 * s2 + 3]));
                } else {
                    const u32x4 frag = u32x4{hlo0, hlo1, P::relu2(P::pack2(hacc[t][8 * s2 + 4], hacc[t][8 * s2 + 5])), P::relu2(P::pack2(hacc[t][8 * s2 + 6], hacc[t][8 * s2 + 7]))};
                    const uint32_t addr = hb0 + (uint32_t)((t * 8 + 2 * wave + s2) * 1024);
                    asm volatile("ds_write_b128 %0, %1" :: "v"(addr), "v"(frag) : "memory");
                }
            };
            stream<OffChunk256, 120, 6>(pb0, pb1, [&](auto ic, const u32x4& bf) {
                constexpr int i = decltype(ic)::value;
                if constexpr (i < 48) {
                    constexpr int ks = i / 3, tb = i % 3;
                    if constexpr (ks == 0) P::mma32v0(hacc[tb], w1f[0], bf, bias);
                    else P::mma32v(hacc[tb], w1f[ks], bf);
                    if constexpr (i % 3 == 1) gload_frag_acc<i / 3>(w2a[i / 3], voff, w2c);
                } else if constexpr (i < 80) {
                    constexpr int j = i - 48, ks = j / 2, tb = 3 + j % 2;
                    if constexpr (j == 0) cstamp(1);
                    if constexpr (ks == 0) P::mma32v0(hacc[tb], w1f[0], bf, bias);
                    else P::mma32v(hacc[tb], w1f[ks], bf);
                    // (b1 is the C operand of step 49's MFMA, which reads it over its passes: the registers stay b1's until
                    // that MFMA is done -- the compiler, blind to the asm, handed them to the next instruction otherwise)
                    if constexpr (j == 4) asm volatile("" :: "v"(bias));
                    // unit k = (block k / 2, half k % 2) as two half-units on the steps j = 3 k + 2, 3 k + 3.  The last write is at
                    // j = 18 (step 66): a wave's LDS operations complete in order, and by barrier X (step 74) it has waited for the
                    // ring read it issued behind that write (step 66's, awaited at step 72).
                    if constexpr (j >= 2 && (j - 2) % 3 < 2 && (j - 2) / 3 < 6)
                        h_half(std::integral_constant<int, ((j - 2) / 3) / 2>{}, std::integral_constant<int, ((j - 2) / 3) % 2>{}, std::integral_constant<int, (j - 2) % 3>{});
                    if constexpr (i == 74) asm volatile("s_barrier" ::: "memory");
                } else if constexpr (i < 104) {
                    constexpr int j = i - 80, ks = j / 3, tb = j % 3;
                    if constexpr (j == 0) {
                        cstamp(2);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this chunk's W2 fragments have landed
#pragma unroll
                        for (int k = 0; k < 16; ++k) asm volatile("" : "+a"(w2a[k]));
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    yacc[0][tb] = P::mma32(w2a[ks], bf, yacc[0][tb]);
                    yacc[1][tb] = P::mma32(w2a[8 + ks], bf, yacc[1][tb]);
                    if constexpr (j < 16) gload_frag<j>(w1f[j], voff, next1);
                    // the four units of blocks 3, 4 on the steps j = 1, 4, 7, 10 (two MFMAs each: a unit fits).  The last write
                    // is at step 90; barrier Y (step 98) comes behind the wait for step 90's ring read (awaited at step 96).
                    if constexpr (j % 3 == 1 && j / 3 < 4) h_write(std::integral_constant<int, 3 + (j / 3) / 2>{}, std::integral_constant<int, (j / 3) % 2>{});
                    if constexpr (j == 17) bias_read(braw, cn);
                    if constexpr (i == 98) asm volatile("s_barrier" ::: "memory");
                } else {
                    constexpr int j = i - 104, ks = j / 2, tb = 3 + j % 2;
                    if constexpr (j == 0) cstamp(3);
                    yacc[0][tb] = P::mma32(w2a[ks], bf, yacc[0][tb]);
                    yacc[1][tb] = P::mma32(w2a[8 + ks], bf, yacc[1][tb]);
                }
            });
            cstamp(5);
        }
    } else {
        for (int c = 0; c < ((PPG_DBG(a) & 2) ? 0 : NCH); ++c) {
#ifdef PPG_FFN_TIMING
            auto cstamp = [&](int k) { if (c == 4 && a.dbg && blockIdx.x == 0 && lane == 0) a.dbg[wave * 8 + k] = __builtin_amdgcn_s_memtime(); };
#else
            auto cstamp = [&](int) {};
#endif
            cstamp(0);
            f32x16 bias;            // C operand of the chunk's first MFMAs: b1 of the lane's 16 hidden rows
            {
                u32x4 braw[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) ds_read128<0>(braw[q], lds0 + G::L_B1 + (uint32_t)((c * HC + 32 * wave + 8 * q + 4 * hh) * 4));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    asm volatile("" : "+v"(braw[q]));
                    bias[4 * q + 0] = __uint_as_float(braw[q].x); bias[4 * q + 1] = __uint_as_float(braw[q].y);
                    bias[4 * q + 2] = __uint_as_float(braw[q].z); bias[4 * q + 3] = __uint_as_float(braw[q].w);
                }
            }
            vm_wait_all(w1f);
            const char* w1c = a.w1_img + (((size_t)c * 4 + wave) * KS) * 1024;          // this chunk's W1 fragments [ks]
            const char* w2c = a.w2_img + (((size_t)c * 4 + wave) * RB * 8) * 1024;      // this chunk's W2 fragments [rb][ks8]
            // what goes into the W1 register set after this chunk: the next chunk's first 16 fragments; after the
            // last chunk the first half-step of the Q/K/V tail (or, without a tail, this chunk's again: harmless)
            const char* next1 = c + 1 < NCH ? a.w1_img + (((size_t)(c + 1) * 4 + wave) * KS) * 1024
                                : (QKV ? wq_first : w1c);
            f32x16 hacc[TB];
            // ReLU + pack of one finished block: the accumulator IS the next GEMM's B fragment layout
            auto h_write = [&](auto t_tag, auto s_tag) {
                constexpr int t = decltype(t_tag)::value, s2 = decltype(s_tag)::value;
                const u32x4 frag = u32x4{P::relu2(P::pack2(hacc[t][8 * s2 + 0], hacc[t][8 * s2 + 1])), P::relu2(P::pack2(hacc[t][8 * s2 + 2], hacc[t][8 * s2 + 3])),
                                         P::relu2(P::pack2(hacc[t][8 * s2 + 4], hacc[t][8 * s2 + 5])), P::relu2(P::pack2(hacc[t][8 * s2 + 6], hacc[t][8 * s2 + 7]))};
                const uint32_t addr = hb0 + (uint32_t)((t * 8 + 2 * wave + s2) * 1024);
                asm volatile("ds_write_b128 %0, %1" :: "v"(addr), "v"(frag) : "memory");
            };
                // hidden 512: 32 W1 and 32 W2 fragments per chunk pass through the two register sets in halves:
                // A1 (W1 ks 0..15 in set 1) | A2 (ks 16..31 in set 2) | B1 (W2 row blocks 0, 1 in set 1) | B2 (2, 3 in set 2),
                // each half prefetching the other set
                stream<OffPanel<KS, 0, 0, TB>, 16 * TB, 6>(pb0, pb1, [&](auto ic, const u32x4& bf) {
                    constexpr int i = decltype(ic)::value;
                    constexpr int ks = i / TB, tb = i % TB;
                    if constexpr (ks == 0) hacc[tb] = P::mma32(w1f[0], bf, bias);
                    else hacc[tb] = P::mma32(w1f[ks], bf, hacc[tb]);
                    if constexpr (i % 3 == 1) gload_frag<16 + i / 3>(w2f[i / 3], voff, w1c);
                });
                vm_wait_all(w2f);
                cstamp(1);
                stream<OffPanel<KS, 16, 0, TB>, 16 * TB, 6>(pb0, pb1, [&](auto ic, const u32x4& bf) {
                    constexpr int i = decltype(ic)::value;
                    constexpr int ks = i / TB, tb = i % TB;
                    if constexpr (i == 0) __builtin_amdgcn_s_barrier();        // the previous chunk's h is read
                    hacc[tb] = P::mma32(w2f[ks], bf, hacc[tb]);
                    if constexpr (i % 3 == 1) gload_frag<i / 3>(w1f[i / 3], voff, w2c);
                });
                cstamp(2);
                [&]<int... U>(std::integer_sequence<int, U...>) {
                    (h_write(std::integral_constant<int, U / 2>{}, std::integral_constant<int, U % 2>{}), ...);
                }(std::make_integer_sequence<int, 2 * TB>{});
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                cstamp(3);
                __syncthreads();
                cstamp(4);
                vm_wait_all(w1f);
                stream<OffH<TB>, 8 * TB, 6>(hb0, hb0, [&](auto ic, const u32x4& bf) {
                    constexpr int i = decltype(ic)::value;
                    constexpr int ks = i / TB, tb = i % TB;
                    yacc[0][tb] = P::mma32(w1f[ks], bf, yacc[0][tb]);
                    yacc[1][tb] = P::mma32(w1f[8 + ks], bf, yacc[1][tb]);
                    if constexpr (i < 16) gload_frag<16 + i>(w2f[i], voff, w2c);
                });
                vm_wait_all(w2f);
                stream<OffH<TB>, 8 * TB, 6>(hb0, hb0, [&](auto ic, const u32x4& bf) {
                    constexpr int i = decltype(ic)::value;
                    constexpr int ks = i / TB, tb = i % TB;
                    yacc[2][tb] = P::mma32(w2f[ks], bf, yacc[2][tb]);
                    yacc[3][tb] = P::mma32(w2f[8 + ks], bf, yacc[3][tb]);
                    if constexpr (i < 16) gload_frag<i>(w1f[i], voff, next1);
                });
            cstamp(5);
        }
    }
    vm_wait_all(w1f);
    pstamp(4);

    // ---- LayerNorm-2 -> X (fp32 residual stream, X32 order: one KiB per store) and its row-major 16-bit copy
    layer_norm(std::false_type{}, std::false_type{}, lnp2, [&](int t, int rb, const f32x16& y) {
        if (SUB && t >= nblk) {
            // a token block past the tile's end: nothing of it is stored (the panel copy feeds the tail's MFMAs only)
        } else if (a.write_x && a.x_half) {                 // (the last layer's residual stream has no reader)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
                *reinterpret_cast<u32x4*>(xh + ((gblock(t) * RB + rb) * 2 + s2) * 1024) = u32x4{pack_f16x2(y[8 * s2 + 0], y[8 * s2 + 1]), pack_f16x2(y[8 * s2 + 2], y[8 * s2 + 3]),
                                                                                  pack_f16x2(y[8 * s2 + 4], y[8 * s2 + 5]), pack_f16x2(y[8 * s2 + 6], y[8 * s2 + 7])};
        } else if (a.write_x) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(xt + ((gblock(t) * RB + rb) * 4 + q) * 256) = make_float4(y[4 * q + 0], y[4 * q + 1], y[4 * q + 2], y[4 * q + 3]);
        }
        if constexpr (QKV) panel_write(t, rb, y);          // x2: the B operand of the Q/K/V tail
        const int m = m0 + 32 * t + tok;
        if (a.Xb && m < a.M && !(SUB && t >= nblk)) {
            char* brow = a.Xb + ((size_t)m * HIDT + fbase + 32 * rb + 16 * hh) * 2;
#pragma unroll
            for (int s = 0; s < 2; ++s)
                *reinterpret_cast<u32x4*>(brow + 16 * s) = u32x4{P::pack2(y[8 * s + 0], y[8 * s + 1]), P::pack2(y[8 * s + 2], y[8 * s + 3]),
                                                                  P::pack2(y[8 * s + 4], y[8 * s + 5]), P::pack2(y[8 * s + 6], y[8 * s + 7])};
        }
    });
    pstamp(5);

    if constexpr (QKV) {
        qkv_tail<P, HIDT, TBS>(a, smem, m0, w1f, w2f, nblk);
        pstamp(6);
    }
}

template <class P, int HIDT>
hipError_t launch_layer32_h(const Layer32Args& a, hipStream_t s) {
    if (a.F % HC || a.F < HC || a.M <= 0 || a.F > 8192) return hipErrorInvalidValue;
    auto launch = [&](auto kern, auto geo, int subs) {
        using G = decltype(geo);
        const size_t lds = (size_t)G::L_B1 + (size_t)a.F * 4;
        if (lds > 163840) return hipErrorInvalidValue;
        const int tiles = (a.M + 32 * tile_blocks(HIDT) - 1) / (32 * tile_blocks(HIDT));
        static ppg::LdsLimit limit;
        const hipError_t e = limit.ensure(reinterpret_cast<const void*>(kern), lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(tiles * subs), dim3(256), lds, s, a);
        return hipGetLastError();
    };
    if constexpr (HIDT == 256) {
        if (a.sub_tiles) {                 // two token blocks per workgroup, 3 workgroups per 160-token tile
            if ((a.F / HC) % 2) return hipErrorInvalidValue;
            if (a.wq_img != nullptr) return launch(layer32_kernel<P, HIDT, true, 2>, Geo<HIDT, 2>{}, 3);
            return launch(layer32_kernel<P, HIDT, false, 2>, Geo<HIDT, 2>{}, 3);
        }
    }
    if (a.wq_img != nullptr) return launch(layer32_kernel<P, HIDT, true>, Geo<HIDT>{}, 1);
    return launch(layer32_kernel<P, HIDT, false>, Geo<HIDT>{}, 1);
}

template <class P>
hipError_t launch_layer32_p(const Layer32Args& a, hipStream_t s) {
    if (a.H == 256) return launch_layer32_h<P, 256>(a, s);
    if (a.H == 512) return launch_layer32_h<P, 512>(a, s);
    return hipErrorInvalidValue;
}

}  // namespace

namespace ppg {

int layer32_tokens(int hidden) { return hidden == 512 ? Geo<512>::TOKS : Geo<256>::TOKS; }

hipError_t launch_layer32(int precision, const Layer32Args& a, hipStream_t s) {
    if (precision == PPG_PRECISION_BF16) return launch_layer32_p<PrecBF16>(a, s);
    if (precision == PPG_PRECISION_FP16) return launch_layer32_p<PrecF16>(a, s);
    return hipErrorInvalidValue;
}

}  // namespace ppg
