// The FFN block of an encoder layer in the fp16x2 mode (PPG_PRECISION_FP16X2: every GEMM operand an fp16 hi + lo
// pair, three fp16 MFMAs per product, fp32 accumulation -- DESIGN 4.6) on the feature-split machinery of
// ppg_layer32.hip:   X <- LayerNorm2(X + b2 + W2 relu(W1 x + b1)),   x = the 16-bit-pair copy of X
// (reference: torch TransformerEncoderLayer._ff_block + norm2 as called from ppgs/model/transformer.py:76-80,
// SURVEY.md 2.3 row M6).  It replaces the token-split ffn_kernel<PrecX2> (+ ffn_reduce_ln_kernel) for batches that
// fill the chip: that kernel stages the WEIGHTS through LDS by DMA and runs v_mfma_f32_16x16x32_f16; here
//   * a workgroup takes 96 token rows (3 blocks of 32 = the MFMA's N; two operand planes cap the tile: the token
//     panel is 96 x 256 x (hi + lo) x 2 B = 96 KiB, the chunk's h 48 KiB), its four waves split the FEATURES: wave w
//     owns hidden rows 32 w .. of every 128-hidden chunk and output features 64 w .. 64 w + 63;
//   * weights never touch LDS: host-packed A-fragment images, hi and lo plane, 16 fragments per register set, two
//     sets that alternate through the chunk's four half-phases, each half prefetching the other set
//         A1  hacc += W1hi x {x_hi, x_lo}     96 MFMAs   (set 1; set 2 <- W1lo)
//         A2  hacc += W1lo x  x_hi            48 MFMAs   (set 2; set 1 <- W2hi)
//         h = relu(hacc) split into hi + lo fragments -> LDS, barrier
//         B1  yacc += W2hi x {h_hi, h_lo}     96 MFMAs   (set 1; set 2 <- W2lo)
//         B2  yacc += W2lo x  h_hi            48 MFMAs   (set 2; set 1 <- the next chunk's W1hi)
//     (a_lo b_lo, 2^-22 of the product, is dropped as in every kernel of the mode);
//   * activations are B fragments in LDS (lane-linear KiB, no swizzle): the panel is filled by global -> LDS DMA
//     straight from the [32 hi | 32 lo] rows the other kernels of the mode write (per-lane source addresses, the
//     destination is the fragment), h by ds_write_b128 from the accumulators, whose layout IS the next GEMM's B
//     fragment once W2's K order is permuted on the host (as in ppg_layer32.hip).
// Epilogue: + b2 + the fp32 residual row, LayerNorm over the 256 features spread over the four waves (statistics
// through LDS), fp32 rows and the 16-bit-pair copy out.
// OP = true puts the attention block's tail in front (as ppg_layer32.hip does): the panel is first filled with the
// attention output, x1 = LayerNorm1(X + bo + Wo ao) comes out of 288 more MFMAs per wave, stays in the accumulators as
// the FFN's residual and goes, split into hi + lo, into the panel -- one launch per layer less (the token-split
// linear_kernel<PrecX2, EPI_RESLN>: 45 us of a 2.1 ms step, five times).
#include "ppg_layer32.h"

namespace {

constexpr int XH = 256;                 // hidden width
constexpr int XTB = 3;                  // token blocks of 32 per workgroup
constexpr int XTOK = 32 * XTB;
constexpr int XKS = XH / 16;            // K-steps of a panel row
// LDS map (bytes)
constexpr int X_PANEL = 0;                                  // fragments [tb][ks][plane] of 1 KiB
constexpr int X_H = XTB * XKS * 2 * 1024;                   // h of one chunk: fragments [tb][8][plane]
constexpr int X_LNP = X_H + XTB * 8 * 2 * 1024;             // [b2 | gamma2 | beta2]
constexpr int X_STATS = X_LNP + 3 * XH * 4;                 // LayerNorm partial sums [2][4 waves][XTOK]
constexpr int X_B1 = X_STATS + 2 * 4 * XTOK * 4;            // b1, F floats

struct OffA1 { static constexpr int at(int i) { return (((i % 3) * XKS + i / 6) * 2 + (i % 6) / 3) * 1024; } };   // (ks, plane, tb)
struct OffA2 { static constexpr int at(int i) { return (((i % 3) * XKS + i / 3) * 2) * 1024; } };                 // (ks, tb), hi plane
struct OffB1 { static constexpr int at(int i) { return (((i % 3) * 8 + i / 6) * 2 + (i % 6) / 3) * 1024; } };
struct OffB2 { static constexpr int at(int i) { return (((i % 3) * 8 + i / 3) * 2) * 1024; } };
// out-projection, K half KHALF (K-steps 8 KHALF .. + 7 of the panel): hi weights x both planes, lo weights x hi plane
template <int KHALF> struct OffO1 { static constexpr int at(int i) { return (((i % 3) * XKS + 8 * KHALF + i / 6) * 2 + (i % 6) / 3) * 1024; } };
template <int KHALF> struct OffO2 { static constexpr int at(int i) { return (((i % 3) * XKS + 8 * KHALF + i / 3) * 2) * 1024; } };

template <bool OP, bool QKV>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void ffn32x2_kernel(Ffn32X2Args a) {
    using P = PrecF16;                  // the planes are fp16: v_mfma_f32_32x32x16_f16
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tok = lane & 31, hh = lane >> 5;
    const int m0 = blockIdx.x * XTOK;
    const int NCH = a.F / HC;
    const uint32_t lds0 = lds_addr32(smem);
    const uint32_t voff = lane * 16;
    const uint32_t pb0 = lds0 + X_PANEL + lane * 16, pb1 = pb0 + 65536;
    const uint32_t hb0 = lds0 + X_H + lane * 16;
    float* lnp = reinterpret_cast<float*>(smem + X_LNP);
    float* lnp1 = reinterpret_cast<float*>(smem + X_H);      // [bo | gamma1 | beta1]: in the h region, which the prologue does not use
    float* stats = reinterpret_cast<float*>(smem + X_STATS);
    const int fbase = 64 * wave;

#ifdef PPG_FFN_TIMING
    auto pstamp = [&](int k) { if (a.dbg && blockIdx.x == 0 && lane == 0) a.dbg[128 + wave * 16 + k] = __builtin_amdgcn_s_memtime(); };
#else
    auto pstamp = [&](int) {};
#endif
    pstamp(0);
    u32x4 s1[16], s2[16];
    auto load16 = [&](u32x4 (&wf)[16], const char* base) {
        [&]<int... K>(std::integer_sequence<int, K...>) { (gload_frag<K>(wf[K], voff, base), ...); }(std::make_integer_sequence<int, 16>{});
    };
    // ---- the token panel: fragment (tb, ks, plane) <- rows m0 + 32 tb .. of the [32 hi | 32 lo] copy: lane (tok, hh)
    // takes the 16 bytes of elements 16 ks + 8 hh .. + 7 of its row's plane.  Rows past M re-read the last row.
    {
        uint32_t lane_off[XTB];
#pragma unroll
        for (int t = 0; t < XTB; ++t) lane_off[t] = (uint32_t)min(m0 + 32 * t + tok, a.M - 1) * (XH * 4) + hh * 16;
#pragma unroll
        for (int i = 0; i < XTB * XKS * 2 / 4; ++i) {
            const int p = 4 * i + wave;                       // fragment index ((tb * XKS + ks) * 2 + plane)
            const int plane = p & 1, ks = (p >> 1) % XKS, tb = (p >> 1) / XKS;
            glds16((OP ? a.ao : a.xb) + (ks >> 1) * 128 + plane * 64 + (ks & 1) * 32, lane_off[tb], lds0 + X_PANEL + p * 1024);
        }
    }
    // parameters: every global load first, the LDS stores afterwards
    {
        float4 p2[1], p1[1];
        const int i = min(tid, 3 * XH / 4 - 1);
        const int v = i / (XH / 4), j = i - v * (XH / 4);
        p2[0] = reinterpret_cast<const float4*>(v == 0 ? a.b2 : (v == 1 ? a.g2 : a.e2))[j];
        if constexpr (OP) p1[0] = reinterpret_cast<const float4*>(v == 0 ? a.bo : (v == 1 ? a.g1 : a.e1))[j];
        constexpr int B1MAX = 8;                 // F <= 8192
        float4 pb[B1MAX];
#pragma unroll
        for (int u = 0; u < B1MAX; ++u) pb[u] = reinterpret_cast<const float4*>(a.b1)[min(tid + 256 * u, a.F / 4 - 1)];
        if (tid < 3 * XH / 4) reinterpret_cast<float4*>(lnp)[tid] = p2[0];
        if constexpr (OP) { if (tid < 3 * XH / 4) reinterpret_cast<float4*>(lnp1)[tid] = p1[0]; }
#pragma unroll
        for (int u = 0; u < B1MAX; ++u)
            if (tid + 256 * u < a.F / 4) reinterpret_cast<float4*>(smem + X_B1)[tid + 256 * u] = pb[u];
    }
    // image order: [chunk][wave][W1hi ks 0..15 | W1lo ks 0..15], [chunk][wave][W2hi (rb, ks8) | W2lo (rb, ks8)]
    auto w1_of = [&](int c) { return a.w1_img + (((size_t)c * 4 + wave) * 32) * 1024; };
    auto w2_of = [&](int c) { return a.w2_img + (((size_t)c * 4 + wave) * 32) * 1024; };
    f32x16 yacc[2][XTB];
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // LayerNorm over the 256 features of a token, spread over the four waves: yacc <- LN(yacc + bias [+ X row]) * gamma
    // + beta (one pass for the statistics, exchanged through LDS); emit(t, rb, m, y) gets the 16 results of a block
    auto layer_norm = [&](auto residual_tag, const float* prm, auto emit) {
        constexpr bool RES = decltype(residual_tag)::value;
        auto quad = [&](int which, int rb, int q) { return *reinterpret_cast<const float4*>(prm + which * XH + fbase + 32 * rb + 16 * hh + 4 * q); };
#pragma unroll
        for (int t = 0; t < XTB; ++t) {
            const int m = min(m0 + 32 * t + tok, a.M - 1);
            f32x2 sum2 = {0.f, 0.f}, sq2 = {0.f, 0.f};
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float4 rv = make_float4(0.f, 0.f, 0.f, 0.f);
                    if constexpr (RES) rv = *reinterpret_cast<const float4*>(a.X + (size_t)m * XH + fbase + 32 * rb + 16 * hh + 4 * q);
                    const float4 bv = quad(0, rb, q);
                    yacc[rb][t][4 * q + 0] += bv.x + rv.x; yacc[rb][t][4 * q + 1] += bv.y + rv.y;
                    yacc[rb][t][4 * q + 2] += bv.z + rv.z; yacc[rb][t][4 * q + 3] += bv.w + rv.w;
                    stat4(sum2, sq2, yacc[rb][t][4 * q + 0], yacc[rb][t][4 * q + 1], yacc[rb][t][4 * q + 2], yacc[rb][t][4 * q + 3]);
                }
            const float sum = pair_sum(hsum2(sum2));
            const float sq = pair_sum(hsum2(sq2));
            if (hh == 0) {
                stats[wave * XTOK + 32 * t + tok] = sum;
                stats[4 * XTOK + wave * XTOK + 32 * t + tok] = sq;
            }
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < XTB; ++t) {
            const float* st = stats + 32 * t + tok;
            const float mean = ((st[0] + st[XTOK]) + (st[2 * XTOK] + st[3 * XTOK])) * (1.0f / XH);
            const float ex2 = ((st[4 * XTOK] + st[5 * XTOK]) + (st[6 * XTOK] + st[7 * XTOK])) * (1.0f / XH);
            const float rstd = 1.0f / sqrtf(fmaxf(ex2 - mean * mean, 0.f) + kLnEps32);
            const float shift = -mean * rstd;
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 g4 = quad(1, rb, q), e4 = quad(2, rb, q);
                    yacc[rb][t][4 * q + 0] = fmaf(fmaf(yacc[rb][t][4 * q + 0], rstd, shift), g4.x, e4.x);
                    yacc[rb][t][4 * q + 1] = fmaf(fmaf(yacc[rb][t][4 * q + 1], rstd, shift), g4.y, e4.y);
                    yacc[rb][t][4 * q + 2] = fmaf(fmaf(yacc[rb][t][4 * q + 2], rstd, shift), g4.z, e4.z);
                    yacc[rb][t][4 * q + 3] = fmaf(fmaf(yacc[rb][t][4 * q + 3], rstd, shift), g4.w, e4.w);
                }
                emit(t, rb, m0 + 32 * t + tok, yacc[rb][t]);
            }
        }
    };
    // the 16 results of block (t, rb), split into hi + lo, into the panel: a lane's 16 values are K-step 4 wave + 2 rb + hh
    // of its token -- both halves of that fragment's lane pair (tok, tok + 32)
    auto panel_write = [&](int t, int rb, const f32x16& y) {
        const uint32_t frag = lds0 + X_PANEL + (uint32_t)(((t * XKS + 4 * wave + 2 * rb + hh) * 2) * 1024 + tok * 16);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            u32x4 fh, fl;
            uint32_t* ph = reinterpret_cast<uint32_t*>(&fh);
            uint32_t* pl = reinterpret_cast<uint32_t*>(&fl);
#pragma unroll
            for (int j = 0; j < 4; ++j) PrecX2::split2(y[8 * half + 2 * j], y[8 * half + 2 * j + 1], ph[j], pl[j]);
            const uint32_t addr = frag + half * 512;
            asm volatile("ds_write_b128 %0, %1" :: "v"(addr), "v"(fh) : "memory");
            asm volatile("ds_write_b128 %0, %1 offset:1024" :: "v"(addr), "v"(fl) : "memory");
        }
    };
    if constexpr (OP) {
        // ---- out-projection: image order [wave][K half][Wo hi (rb, ks8) | Wo lo (rb, ks8)]; x1 = LN1(X + bo + Wo ao)
        const char* wo = a.wo_img + ((size_t)wave * 64) * 1024;
        load16(s1, wo);
        load16(s2, wo + 16 * 1024);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the panel's DMA, the parameters, K half 0 of Wo
#pragma unroll
        for (int k = 0; k < 16; ++k) asm volatile("" : "+v"(s1[k]), "+v"(s2[k]));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
        [&]<int... KHALF>(std::integer_sequence<int, KHALF...>) {
            ([&] {
                stream<OffO1<KHALF>, 8 * 6, 6>(pb0, pb1, [&](auto ic, const u32x4& bf) {
                    constexpr int i = decltype(ic)::value;
                    constexpr int ks = i / 6, tb = i % 3;
                    if constexpr (KHALF == 0 && i < 3) {
                        yacc[0][tb] = P::mma32(s1[0], bf, zero);
                        yacc[1][tb] = P::mma32(s1[8], bf, zero);
                    } else {
                        yacc[0][tb] = P::mma32(s1[ks], bf, yacc[0][tb]);
                        yacc[1][tb] = P::mma32(s1[8 + ks], bf, yacc[1][tb]);
                    }
                });
                if constexpr (KHALF == 0) load16(s1, wo + 32 * 1024);       // K half 1, hi (its turn comes after this half's lo)
                stream<OffO2<KHALF>, 8 * 3, 6>(pb0, pb1, [&](auto ic, const u32x4& bf) {
                    constexpr int i = decltype(ic)::value;
                    constexpr int ks = i / 3, tb = i % 3;
                    yacc[0][tb] = P::mma32(s2[ks], bf, yacc[0][tb]);
                    yacc[1][tb] = P::mma32(s2[8 + ks], bf, yacc[1][tb]);
                });
                if constexpr (KHALF == 0) {
                    load16(s2, wo + 48 * 1024);
                    vm_wait_all(s1);
                    vm_wait_all(s2);
                }
            }(), ...);
        }(std::integer_sequence<int, 0, 1>{});
        pstamp(1);
        // x1: stays in the accumulators (the FFN sums on top of its own residual) and goes into the panel
        layer_norm(std::true_type{}, lnp1, [&](int t, int rb, int, const f32x16& y) { panel_write(t, rb, y); });
        pstamp(2);
        load16(s1, w1_of(0));
        vm_wait_all(s1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();                                     // the panel holds x1
    } else {
        load16(s1, w1_of(0));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the panel's DMA, the parameters, W1hi of chunk 0
#pragma unroll
        for (int k = 0; k < 16; ++k) asm volatile("" : "+v"(s1[k]));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int t = 0; t < XTB; ++t) yacc[rb][t] = zero;
    }

    pstamp(3);
    for (int c = 0; c < NCH; ++c) {
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
            for (int q = 0; q < 4; ++q) ds_read128<0>(braw[q], lds0 + X_B1 + (uint32_t)((c * HC + 32 * wave + 8 * q + 4 * hh) * 4));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                asm volatile("" : "+v"(braw[q]));
                bias[4 * q + 0] = __uint_as_float(braw[q].x); bias[4 * q + 1] = __uint_as_float(braw[q].y);
                bias[4 * q + 2] = __uint_as_float(braw[q].z); bias[4 * q + 3] = __uint_as_float(braw[q].w);
            }
        }
        vm_wait_all(s1);
        const char* w1c = w1_of(c);
        const char* w2c = w2_of(c);
        const char* next1 = w1_of(c + 1 < NCH ? c + 1 : c);          // (after the last chunk: harmless)
        f32x16 hacc[XTB];
        // A1: W1hi x {x_hi, x_lo}
        stream<OffA1, 16 * 6, 6>(pb0, pb1, [&](auto ic, const u32x4& bf) {
            constexpr int i = decltype(ic)::value;
            constexpr int ks = i / 6, tb = i % 3;
            // (h in architectural registers -- inline-asm MFMAs, as ppg_layer32.hip's phase A: its consumers are VALU
            // instructions, and the builtin's accumulation registers cost them one v_accvgpr_read per value)
            if constexpr (i < 3) P::mma32v0(hacc[tb], s1[0], bf, bias);
            else P::mma32v(hacc[tb], s1[ks], bf);
            if constexpr (i % 6 == 1) gload_frag<16 + i / 6>(s2[i / 6], voff, w1c);
            if constexpr (i == 16 * 6 - 1) asm volatile("" :: "v"(bias));        // (b1 stays b1's until its last reader is done)
        });
        cstamp(1);
        vm_wait_all(s2);
        cstamp(6);
        // A2: W1lo x x_hi
        stream<OffA2, 16 * 3, 6>(pb0, pb1, [&](auto ic, const u32x4& bf) {
            constexpr int i = decltype(ic)::value;
            constexpr int ks = i / 3, tb = i % 3;
            if constexpr (i == 0) __builtin_amdgcn_s_barrier();        // every wave has read the previous chunk's h
            P::mma32v(hacc[tb], s2[ks], bf);
            if constexpr (i % 3 == 1) gload_frag<i / 3>(s1[i / 3], voff, w2c);
        });
        cstamp(7);
        // (the compiler inserts no wait states behind an asm MFMA: block 0's accumulators were last written three MFMAs
        // ago, the others are read a unit or two of VALU work later; the nops cover the pipe's drain)
        asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
        // h = relu(hacc), split: registers 8 s2 .. + 7 of block t are K-step 2 wave + s2 of the chunk's h
#pragma unroll
        for (int t = 0; t < XTB; ++t)
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                u32x4 fh, fl;
                uint32_t* ph = reinterpret_cast<uint32_t*>(&fh);
                uint32_t* pl = reinterpret_cast<uint32_t*>(&fl);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    PrecX2::split2(fmaxf(hacc[t][8 * h2 + 2 * j], 0.f), fmaxf(hacc[t][8 * h2 + 2 * j + 1], 0.f), ph[j], pl[j]);
                const uint32_t addr = hb0 + (uint32_t)(((t * 8 + 2 * wave + h2) * 2) * 1024);
                asm volatile("ds_write_b128 %0, %1" :: "v"(addr), "v"(fh) : "memory");
                asm volatile("ds_write_b128 %0, %1 offset:1024" :: "v"(addr), "v"(fl) : "memory");
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
        vm_wait_all(s1);
        cstamp(2);
        // B1: W2hi x {h_hi, h_lo}
        stream<OffB1, 8 * 6, 6>(hb0, hb0, [&](auto ic, const u32x4& bf) {
            constexpr int i = decltype(ic)::value;
            constexpr int ks = i / 6, tb = i % 3;
            yacc[0][tb] = P::mma32(s1[ks], bf, yacc[0][tb]);
            yacc[1][tb] = P::mma32(s1[8 + ks], bf, yacc[1][tb]);
            if constexpr (i < 16) gload_frag<16 + i>(s2[i], voff, w2c);
        });
        cstamp(3);
        vm_wait_all(s2);
        // B2: W2lo x h_hi
        stream<OffB2, 8 * 3, 6>(hb0, hb0, [&](auto ic, const u32x4& bf) {
            constexpr int i = decltype(ic)::value;
            constexpr int ks = i / 3, tb = i % 3;
            yacc[0][tb] = P::mma32(s2[ks], bf, yacc[0][tb]);
            yacc[1][tb] = P::mma32(s2[8 + ks], bf, yacc[1][tb]);
            if constexpr (i < 16) gload_frag<i>(s1[i], voff, next1);
        });
        cstamp(5);
    }
    vm_wait_all(s1);
    pstamp(4);

    // ---- + b2 (+ the residual row, unless the accumulators started from it), LayerNorm-2, fp32 rows and the operand copy out
    auto store = [&](int t, int rb, int m, const f32x16& y) {
        if constexpr (QKV) panel_write(t, rb, y);            // x2: the B operand of the next layer's Q/K/V below
        if (m >= a.M) return;
        float* xrow = a.X + (size_t)m * XH + fbase + 32 * rb + 16 * hh;
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(xrow + 4 * q) = make_float4(y[4 * q + 0], y[4 * q + 1], y[4 * q + 2], y[4 * q + 3]);
        if (QKV) return;                                     // (nobody reads the operand copy: x2 goes straight into the tail)
        // the 16 features 64 w + 32 rb + 16 hh .. of the [32 hi | 32 lo] copy: 32 bytes of the hi plane, lo 64 on
        char* brow = a.xb_out + (size_t)m * (XH * 4) + (2 * wave + rb) * 128 + hh * 32;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            u32x4 fh, fl;
            uint32_t* ph = reinterpret_cast<uint32_t*>(&fh);
            uint32_t* pl = reinterpret_cast<uint32_t*>(&fl);
#pragma unroll
            for (int j = 0; j < 4; ++j) PrecX2::split2(y[8 * s + 2 * j], y[8 * s + 2 * j + 1], ph[j], pl[j]);
            *reinterpret_cast<u32x4*>(brow + 16 * s) = fh;
            *reinterpret_cast<u32x4*>(brow + 64 + 16 * s) = fl;
        }
    };
    if constexpr (OP) layer_norm(std::false_type{}, lnp, store);
    else layer_norm(std::true_type{}, lnp, store);

    pstamp(5);
    if constexpr (QKV) {
        // ---- the next layer's Q / K / V from the x2 panel: three projections of 288 MFMAs per wave, image order
        // [wave][kind][K half][hi (rb, ks8) | lo (rb, ks8)].  Q and K leave as [32 hi | 32 lo] rows (a lane owns 16
        // consecutive features of its token), V with the MFMA operands swapped, so that the accumulator comes out
        // transposed: a lane owns one V^T row (tile order: natural feature pair_row(row)) and 16 tokens, whose columns
        // inside a window's 32-token group sit at position 8 g + 4 e + r for token 16 e + 4 g + r (attn_kernel's PV
        // fragment order), both planes.
        const char* wq = a.wq_img + ((size_t)wave * 3 * 64) * 1024;
        load16(s1, wq);
        // the in_proj bias -> LDS (the h region: every wave is past its last chunk, LayerNorm-2's barrier lies between)
        float* bql = reinterpret_cast<float*>(smem + X_H);
        if (tid < 3 * XH / 4) reinterpret_cast<float4*>(bql)[tid] = reinterpret_cast<const float4*>(a.bq)[tid];
        // transposed-V columns of the 16-token halves of every token block (as ppg_layer32.h qkv_tail)
        int vcol[XTB][2];
        bool valigned[XTB];
        bool regular = m0 + XTOK <= a.M;      // every row exists, every block is one 32-token group: the store counts below are exact
#pragma unroll
        for (int t = 0; t < XTB; ++t) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int mb = m0 + 32 * t + 16 * h;
                vcol[t][h] = -1;
                if (mb < a.M) {
                    const int w = a.blk_win[mb >> 4];
                    if (w >= 0) {
                        const int ttb = mb - a.win[w].tok_off;
                        vcol[t][h] = a.win[w].vt_off + (ttb >> 5) * 32 + 4 * ((ttb >> 4) & 1);
                    }
                }
            }
            valigned[t] = vcol[t][0] >= 0 && vcol[t][1] == vcol[t][0] + 4 && (vcol[t][0] & 31) == 0;
            regular = regular && valigned[t];
        }
        vm_wait_all(s1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();                                     // the panel holds x2, the bias is in LDS
        // Stage = (projection, K half).  Its hi fragments are in set 1; the lo fragments travel into set 2 under the
        // stage's first stream, the NEXT stage's hi fragments into set 1 under its second.  The projections alternate
        // between two accumulator sets: the epilogue of projection k -- bias, split, 24 stores of 16 bytes per lane --
        // rides between the MFMAs of projection k + 1's first stream, behind that stream's 16 fragment requests, so
        // that the wait for them is vmcnt(24) in a regular tile (vmcnt counts in order), not a wait for the stores'
        // acknowledgement: a CU retires ~10 bytes of stores per clock, 96 KiB per projection and workgroup -- as long
        // as the projection's MFMAs take (PPG_FFN_TIMING stamps: 85 k cycles for the tail with the epilogues exposed).
        f32x16 qacc[2][2][XTB];
        float4 b4[2][4];
        float bvv[2];
        auto bias_load = [&](auto kind_tag) {
            constexpr int KIND = decltype(kind_tag)::value;
            if constexpr (KIND < 2) {
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int q = 0; q < 4; ++q) b4[rb][q] = *reinterpret_cast<const float4*>(bql + XH * KIND + fbase + 32 * rb + 16 * hh + 4 * q);
            } else {
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) bvv[rb] = bql[2 * XH + pair_row(fbase + 32 * rb + tok)];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                asm volatile("" : "+v"(bvv[rb]));
#pragma unroll
                for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(b4[rb][q].x), "+v"(b4[rb][q].y), "+v"(b4[rb][q].z), "+v"(b4[rb][q].w));
            }
        };
        // epilogue unit U = (row block U / 3, token block U % 3) of projection KIND: 4 stores in a regular tile
        auto unit = [&](auto kind_tag, auto u_tag) {
            constexpr int KIND = decltype(kind_tag)::value, U = decltype(u_tag)::value, rb = U / 3, t = U % 3;
            const f32x16& c = qacc[KIND & 1][rb][t];
            if constexpr (KIND < 2) {
                const int m = m0 + 32 * t + tok;
                if (m >= a.M) return;
                const float y[16] = {c[0] + b4[rb][0].x, c[1] + b4[rb][0].y, c[2] + b4[rb][0].z, c[3] + b4[rb][0].w, c[4] + b4[rb][1].x, c[5] + b4[rb][1].y, c[6] + b4[rb][1].z, c[7] + b4[rb][1].w,
                                     c[8] + b4[rb][2].x, c[9] + b4[rb][2].y, c[10] + b4[rb][2].z, c[11] + b4[rb][2].w, c[12] + b4[rb][3].x, c[13] + b4[rb][3].y, c[14] + b4[rb][3].z, c[15] + b4[rb][3].w};
                u32x4 fh[2], fl[2];
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    uint32_t* ph = reinterpret_cast<uint32_t*>(&fh[s]);
                    uint32_t* pl = reinterpret_cast<uint32_t*>(&fl[s]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) PrecX2::split2(y[8 * s + 2 * j], y[8 * s + 2 * j + 1], ph[j], pl[j]);
                }
                if (regular) {
                    // (as ppg_layer32.h's tail: the two 16-byte halves of a plane traded across 16-lane rows, one
                    // v_permlane16_swap per dword -- an instruction writes 16 rows x 64 contiguous bytes of a plane
                    // instead of 32 rows x 2 pieces)
                    const int r16 = lane & 15, k16 = lane >> 4;
                    char* d16 = a.qk_out + (size_t)(m0 + 32 * t + r16) * (2 * XH * 4) + KIND * (XH * 4) + (2 * wave + rb) * 128 + 16 * k16;
#pragma unroll
                    for (int plane = 0; plane < 2; ++plane) {
                        u32x4 lo, hi;
#pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            const auto sw = plane ? __builtin_amdgcn_permlane16_swap(fl[0][d], fl[1][d], false, false)
                                                  : __builtin_amdgcn_permlane16_swap(fh[0][d], fh[1][d], false, false);
                            lo[d] = sw[0]; hi[d] = sw[1];
                        }
                        *reinterpret_cast<u32x4*>(d16 + 64 * plane) = lo;
                        *reinterpret_cast<u32x4*>(d16 + 64 * plane + (size_t)16 * (2 * XH * 4)) = hi;
                    }
                } else {
                    char* dst = a.qk_out + (size_t)m * (2 * XH * 4) + KIND * (XH * 4) + (2 * wave + rb) * 128 + hh * 32;
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        *reinterpret_cast<u32x4*>(dst + 16 * s) = fh[s];
                        *reinterpret_cast<u32x4*>(dst + 64 + 16 * s) = fl[s];
                    }
                }
            } else {
                // lane = V^T row fbase + 32 rb + tok, registers 4 q + r = token 8 q + 4 hh + r of the block
                const float bv = bvv[rb];
                char* rowp = a.vt_out + (size_t)(fbase + 32 * rb + tok) * a.vt_ld * 4;
                if (valigned[t]) {       // one 32-token group: registers (q, q + 2) are 8 consecutive positions
#pragma unroll
                    for (int s2i = 0; s2i < 2; ++s2i) {
                        uint32_t h[4], l[4];
                        PrecX2::split2(c[4 * s2i + 0] + bv, c[4 * s2i + 1] + bv, h[0], l[0]);
                        PrecX2::split2(c[4 * s2i + 2] + bv, c[4 * s2i + 3] + bv, h[1], l[1]);
                        PrecX2::split2(c[4 * (s2i + 2) + 0] + bv, c[4 * (s2i + 2) + 1] + bv, h[2], l[2]);
                        PrecX2::split2(c[4 * (s2i + 2) + 2] + bv, c[4 * (s2i + 2) + 3] + bv, h[3], l[3]);
                        char* dst = rowp + PrecX2::row_byte(vcol[t][0] + 16 * s2i + 8 * hh);
                        *reinterpret_cast<u32x4*>(dst) = u32x4{h[0], h[1], h[2], h[3]};
                        *reinterpret_cast<u32x4*>(dst + 64) = u32x4{l[0], l[1], l[2], l[3]};
                    }
                } else {
#pragma unroll
                    for (int s2i = 0; s2i < 2; ++s2i) {        // half s2i of the block: tokens 16 s2i .., registers q = 2 s2i, 2 s2i + 1
                        if (vcol[t][s2i] < 0) continue;
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const int q = 2 * s2i + e;
                            store4<PrecX2>(rowp + PrecX2::row_byte(vcol[t][s2i] + 8 * (2 * e + hh)),
                                           c[4 * q + 0] + bv, c[4 * q + 1] + bv, c[4 * q + 2] + bv, c[4 * q + 3] + bv);
                        }
                    }
                }
            }
        };
        [&]<int... STAGE>(std::integer_sequence<int, STAGE...>) {
            ([&] {
                constexpr int KIND = STAGE / 2, KHALF = STAGE % 2;
                constexpr bool SWAP = KIND == 2;
                constexpr bool LAST = STAGE == 5;
                constexpr bool EPI = KHALF == 0 && KIND > 0;         // the previous projection's epilogue rides along
                f32x16 (&acc)[2][XTB] = qacc[KIND & 1];
                const char* lo = wq + (size_t)(STAGE * 32 + 16) * 1024;
                const char* nxt = wq + (size_t)((LAST ? STAGE : STAGE + 1) * 32) * 1024;
                if constexpr (EPI) bias_load(std::integral_constant<int, KIND - 1>{});
                stream<OffO1<KHALF>, 8 * 6, 6>(pb0, pb1, [&](auto ic, const u32x4& bf) {
                    constexpr int i = decltype(ic)::value;
                    constexpr int ks = i / 6, tb = i % 3;
                    if constexpr (KHALF == 0 && i < 3) {
                        acc[0][tb] = SWAP ? P::mma32(bf, s1[0], zero) : P::mma32(s1[0], bf, zero);
                        acc[1][tb] = SWAP ? P::mma32(bf, s1[8], zero) : P::mma32(s1[8], bf, zero);
                    } else {
                        acc[0][tb] = SWAP ? P::mma32(bf, s1[ks], acc[0][tb]) : P::mma32(s1[ks], bf, acc[0][tb]);
                        acc[1][tb] = SWAP ? P::mma32(bf, s1[8 + ks], acc[1][tb]) : P::mma32(s1[8 + ks], bf, acc[1][tb]);
                    }
                    if constexpr (i < 16) gload_frag<i>(s2[i], voff, lo);
                    if constexpr (EPI && i >= 16 && (i - 16) % 5 == 0 && (i - 16) / 5 < 6)
                        unit(std::integral_constant<int, KIND - 1>{}, std::integral_constant<int, (i - 16) / 5>{});
                });
                if constexpr (STAGE == 2) pstamp(13);
                if (EPI && regular) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if constexpr (STAGE == 2) pstamp(14);
#pragma unroll
                for (int k = 0; k < 16; ++k) asm volatile("" : "+v"(s2[k]));
                __builtin_amdgcn_sched_barrier(0);
                stream<OffO2<KHALF>, 8 * 3, 6>(pb0, pb1, [&](auto ic, const u32x4& bf) {
                    constexpr int i = decltype(ic)::value;
                    constexpr int ks = i / 3, tb = i % 3;
                    acc[0][tb] = SWAP ? P::mma32(bf, s2[ks], acc[0][tb]) : P::mma32(s2[ks], bf, acc[0][tb]);
                    acc[1][tb] = SWAP ? P::mma32(bf, s2[8 + ks], acc[1][tb]) : P::mma32(s2[8 + ks], bf, acc[1][tb]);
                    if constexpr (!LAST && i < 16) gload_frag<i>(s1[i], voff, nxt);
                });
                if constexpr (!LAST) vm_wait_all(s1);
                pstamp(7 + STAGE);
            }(), ...);
        }(std::make_integer_sequence<int, 6>{});
        // the last projection's epilogue
        bias_load(std::integral_constant<int, 2>{});
        [&]<int... U>(std::integer_sequence<int, U...>) {
            (unit(std::integral_constant<int, 2>{}, std::integral_constant<int, U>{}), ...);
        }(std::make_integer_sequence<int, 6>{});
        pstamp(6);
    }
}

}  // namespace

namespace ppg {

int ffn32x2_tokens() { return XTOK; }
// the geometries launch_ffn32x2 takes (its own checks: hidden 256, whole chunks, b1 behind the fixed LDS map)
bool ffn32x2_supported(int H, int F) { return H == XH && F >= HC && F % HC == 0 && F <= 8192 && (size_t)X_B1 + (size_t)F * 4 <= 163840; }

hipError_t launch_ffn32x2(const Ffn32X2Args& a, hipStream_t s) {
    if (!ffn32x2_supported(a.H, a.F) || a.M <= 0) return hipErrorInvalidValue;
    const size_t lds = (size_t)X_B1 + (size_t)a.F * 4;
    static ppg::LdsLimit limit[3];
    const bool op = a.wo_img != nullptr, qkv = a.wq_img != nullptr;
    if (qkv && !op) return hipErrorInvalidValue;            // (the tail comes with the fused out-projection only)
    auto launch = [&](auto kern, int which) {
        const hipError_t e = limit[which].ensure(reinterpret_cast<const void*>(kern), lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3((a.M + XTOK - 1) / XTOK), dim3(256), lds, s, a);
        return hipGetLastError();
    };
    if (qkv) return launch(ffn32x2_kernel<true, true>, 2);
    if (op) return launch(ffn32x2_kernel<true, false>, 1);
    return launch(ffn32x2_kernel<false, false>, 0);
}

}  // namespace ppg
