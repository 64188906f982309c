// Attention at head dimension 128 in the 16-bit modes: one wave per SIMD, 64 queries per wave, v_mfma_f32_32x32x16.
//
// Replaces, for the encoder's whole-batch launches, the reference's SDPA as called through
// ppgs/model/transformer.py:74-81 (F.multi_head_attention_forward: softmax(q k^T / sqrt d + key-padding / causal mask) v,
// heads of 128).  Same memory layouts as attn_mixed_kernel (ppg_kernels.hip): Q | K rows [M + 64][2 H], V transposed
// [H][Mvt + 64] with the columns of every 32-token group in position order 8 g + 4 e + r <- token 16 e + 4 g + r and
// the rows in pair_row order, the output in AO32 order (or row-major).  Those orders were made for the 16x16x32
// accumulator; they fit the 32x32 one as they are: registers {4 s' + r, 8 + 4 s' + r} of a 32-key score block ARE the B
// fragment of K-step 2 kb + s' of the P V product, and registers (0-3, 8-11) / (4-7, 12-15) of an output block are two
// runs of 8 consecutive features.
//
// Why another kernel (round 5; the timeline of the 16x16x32 kernel at 32 x 1000 frames, profiles/r5_attn_timeline.txt):
// two workgroups of four waves per CU, 128 queries each: a 500-key workgroup takes 21.4 us = 4.2 us prologue (512
// workgroups pulling Q and three tiles at once) + 8 tiles of 1.94 us + 1.8 us of stores; the matrix pipes are busy for
// 0.85 us of a tile.  Two waves share a SIMD's issue port: MFMAs, the softmax's VALU work and the DMA issue of BOTH
// add up.  Here a wave has the SIMD to itself and TWICE the queries: every K / V^T fragment read from LDS feeds two
// 32-query blocks... of which there are two per wave, A and B, whose phases interleave --
//
//     MFMA   | S_A(t) x P V_A(t-1), alternating | S_B(t) x P V_B(t-1), alternating |      S = K q^T - shift, MFMAs of 32 cycles
//     VALU   | exp_B(t-1)                        | exp_A(t)                          |
//
// so the softmax of one block always runs in the gaps of the other block's MFMAs, a phase ahead of the product that
// needs it: 3 VALU slots per 32-cycle MFMA (two v_exp_f32, or a packed add and a convert).  The 64 MFMAs of a tile are
// ONE stream of LDS fragment reads, 8 in flight, that runs on across the phases and across the tile loop's back edge.
// K and V^T tiles of 64 keys arrive by global -> LDS DMA three tiles ahead into rings of 3 / 4 slots (112 KiB: the
// workgroup owns its CU), XOR-swizzled on the source address so that the 32-row fragment reads are conflict-free
// (K rows of 256 B: slot ^ (row & 15); V^T rows of 128 B: slot ^ ((row >> 1) & 7) -- the 16-lane service groups of
// ds_read_b128 hold 8 even and 8 odd rows); one barrier per tile.
//
// Softmax as in attn_body: the engine folds log2(e) / sqrt(d) into W_q, the score accumulators START at -shift (the
// MFMA's C operand), p = ONE v_exp_f32; the shift is the exact maximum of the query's first tile and re-bases only
// when a lane's p sum of a tile passes P::kProbCeil (cold path; `rebase_always`: the classic online softmax's amount).
#include "ppg_layer32.h"

#include <cmath>
#include <cstdlib>
#include <cstring>

#ifdef PPG_ONLY_BF16
#define PPG_OTHER_PRECISIONS 0
#else
#define PPG_OTHER_PRECISIONS 1
#endif

namespace {

#include "ppg_attn64_regs.inc"

constexpr int A64_KT = 64;                         // keys per tile
constexpr int A64_TILE = 16384;                    // bytes of a K tile and of a V^T tile
constexpr int A64_NK = 3, A64_NV = 4;              // ring slots
constexpr int A64_LV = A64_NK * A64_TILE;          // LDS offset of the V^T ring
constexpr int A64_LDS = (A64_NK + A64_NV) * A64_TILE;
constexpr int A64_D = 8;                           // LDS fragment reads in flight

template <int OFF>
__device__ __forceinline__ void a64_read(u32x4& dst, uint32_t addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}

// The MFMAs as asm statements: the register FILE of every operand is part of the design -- the score accumulators
// (read by the softmax's VALU code a phase later) and the P fragments in architectural VGPRs, the output accumulators
// and the Q fragments (touched by nothing but MFMAs for the whole tile loop; gfx950 MFMAs take A / B from either file)
// in the accumulation file.  Left to the compiler's own VGPR / AGPR split the scores land in AGPRs and every tile
// pays ~100 v_accvgpr_read copies at the loop head plus four per Q fragment use.  hipcc pads no hazard of an asm
// statement (cdna_hip_programming.md 5.7): the schedule keeps every reader of an accumulator >= 2 MFMA slots behind
// its last writer (the first exponentials of a block read key block 0, written by the phase's second-to-last MFMA),
// and the places where compiler code reads accumulators right behind the stream (first tile, masks, epilogue) carry
// their own s_nop runs.
// (s_first: the C operand may have been written by the instruction in front -- the compiler materialises the zero
// tuple of the first tile with v_mov_b64 right there, and a VALU write needs two wait states before an MFMA reads it:
// found as scores of key 0 and 1 made of whatever the registers held before, 0x5a5a5a5a on a fresh box)
template <class P> struct A64Op;
template <> struct A64Op<PrecBF16> {
    static __device__ __forceinline__ void s_first(f32x16& d, const u32x4& k, const u32x4& q, const f32x16& c) {
        asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(k), "a"(q), "v"(c));
    }
    static __device__ __forceinline__ void s_acc(f32x16& d, const u32x4& k, const u32x4& q) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(k), "a"(q));
    }
};
template <> struct A64Op<PrecF16> {
    static __device__ __forceinline__ void s_first(f32x16& d, const u32x4& k, const u32x4& q, const f32x16& c) {
        asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(d) : "v"(k), "a"(q), "v"(c));
    }
    static __device__ __forceinline__ void s_acc(f32x16& d, const u32x4& k, const u32x4& q) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d) : "v"(k), "a"(q));
    }
};
// wait states behind an asm MFMA before compiler code may read its result (8-pass: 12 states and more to spare)
__device__ __forceinline__ void a64_mfma_settle() { asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory"); }

template <class P, int NQB>
__device__ __forceinline__ void attn64_body(const AttnArgs& a, const AttnItem& item, const int head, char* smem) {
    static_assert(NQB == 2, "two 32-query blocks per wave");
    constexpr int DH = 128;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hi = lane >> 5;
#ifdef PPG_ATTN_TIMING
    const unsigned long long wg_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int frames16 = (item.frames + 15) & ~15;
    const int qw0 = item.q0 + wave * 32 * NQB;       // first query of this wave
    const uint32_t lds0 = lds_addr32(smem);

    // ---- tile DMA: coalesced KiB pieces, the swizzle on the source address ------------------------------------------
    // K piece p = rows 4 p .. + 3 (16 slots of 16 B each), V^T piece p = rows 8 p .. + 7 (8 slots); wave w takes pieces
    // w, w + 4, w + 8, w + 12 of each: p & 3 (K) and p & 1 (V^T) are constant per wave, so ONE lane offset per kind
    const uint32_t koff = (uint32_t)((lane >> 4) * a.qk_ld_bytes) + (uint32_t)((((lane & 15) ^ ((4 * wave + (lane >> 4)) & 15))) << 4);
    const uint32_t voff = (uint32_t)((lane >> 3) * a.vt_ld_bytes) + (uint32_t)((((lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7))) << 4);
    const char* kbase = a.qk + (size_t)item.tok_off * a.qk_ld_bytes + ((size_t)a.H + (size_t)head * DH) * 2;
    const char* vbase = a.vt + (size_t)head * DH * a.vt_ld_bytes + (size_t)item.vt_off * 2;
    auto stage = [&](int kt) {               // tile kt -> K slot kt % 3, V^T slot kt % 4
        const uint32_t kdst = lds0 + (uint32_t)((kt % A64_NK) * A64_TILE + wave * 1024);
        const uint32_t vdst = lds0 + (uint32_t)(A64_LV + (kt % A64_NV) * A64_TILE + wave * 1024);
        const char* ksrc = kbase + (size_t)(kt * A64_KT + 4 * wave) * a.qk_ld_bytes;
        const char* vsrc = vbase + (size_t)(8 * wave) * a.vt_ld_bytes + (size_t)kt * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            glds16(ksrc + (size_t)(16 * i) * a.qk_ld_bytes, koff, kdst + i * 4096);
            glds16(vsrc + (size_t)(32 * i) * a.vt_ld_bytes, voff, vdst + i * 4096);
        }
    };

    // ... one piece at a time (piece 0..3: K, 4..7: V^T), for the tile loop: a DMA instruction costs its wave tens of issue cycles
    auto stage_piece = [&](int kt, auto pc) {
        constexpr int pi = decltype(pc)::value, i = pi & 3;
        if constexpr (pi < 4) {
            glds16(kbase + (size_t)(kt * A64_KT + 4 * wave + 16 * i) * a.qk_ld_bytes, koff,
                   lds0 + (uint32_t)((kt % A64_NK) * A64_TILE + wave * 1024 + i * 4096));
        } else {
            glds16(vbase + (size_t)(8 * wave + 32 * i) * a.vt_ld_bytes + (size_t)kt * 128, voff,
                   lds0 + (uint32_t)(A64_LV + (kt % A64_NV) * A64_TILE + wave * 1024 + i * 4096));
        }
    };
    int kend = item.valid;
    if (a.causal) kend = min(kend, item.q0 + 4 * 32 * NQB);
    const int ntiles = (kend + A64_KT - 1) / A64_KT;

    // ---- Q fragments (B operands), straight into accumulation registers: lane (query r, half hi) holds d = 16 ks + 8 hi .. + 7.
    // A block past the window's padded rows loads the window's first rows instead (its results are never stored).
    u32x4 qf[NQB][8];
    const uint32_t qoff = (uint32_t)(r * a.qk_ld_bytes + 16 * hi);
#pragma unroll
    for (int b = 0; b < NQB; ++b) {
        const int qrow0 = (qw0 + 32 * b) < frames16 ? qw0 + 32 * b : 0;           // (+ <= 31 rows of slack behind a started block)
        const char* qbase = a.qk + (size_t)(item.tok_off + qrow0) * a.qk_ld_bytes + (size_t)head * DH * 2;
        const uint64_t qb = reinterpret_cast<uint64_t>(qbase);
        const char* qs = reinterpret_cast<const char*>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(qb >> 32)) << 32) |
                                                       (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)qb));
        [&]<int... KS>(std::integer_sequence<int, KS...>) { (gload128_acc<32 * KS>(qf[b][KS], qoff, qs), ...); }(std::make_integer_sequence<int, 8>{});
    }
    if (ntiles > 0) stage(0);
    if (ntiles > 1) stage(1);
    if (ntiles > 2) stage(2);

    // ---- fragment addresses (the XOR swizzle mixes lane bits into every K-step: one VGPR per K-step) -----------------
    // K fragment (kb, ks): row 32 kb + r, slot (2 ks + hi) ^ (r & 15);  V^T fragment (db, s): row 32 db + r, slot (2 s + hi) ^ ((r >> 1) & 7)
    uint32_t kaddr[8], vaddr[4];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) kaddr[ks] = lds0 + (uint32_t)(r * 256 + (((2 * ks + hi) ^ (r & 15)) << 4));
#pragma unroll
    for (int s = 0; s < 4; ++s) vaddr[s] = lds0 + (uint32_t)(A64_LV + r * 128 + (((2 * s + hi) ^ ((r >> 1) & 7)) << 4));

    // (the output accumulators are the asm-owned registers a[128:255]: A64O<P, 4 b + db>)
    f32x16 sacc[NQB][2], cinit[NQB];
    u32x4 pf[NQB][4];
    float shift[NQB], lrun[NQB];
    f32x2 psum2[NQB];
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    [&]<int... B>(std::integer_sequence<int, B...>) { (A64O<P, B>::zero(), ...); }(std::make_integer_sequence<int, 8>{});
#pragma unroll
    for (int b = 0; b < NQB; ++b) { shift[b] = 0.f; lrun[b] = 0.f; cinit[b] = zero16; psum2[b] = f32x2{0.f, 0.f}; }
    // keys below key_limit[b] count for the lane's query of block b: key index of register R of key block kb of tile t
    // is 64 t + 32 kb + 8 (R >> 2) + 4 hi + (R & 3)
    int key_limit[NQB];
#pragma unroll
    for (int b = 0; b < NQB; ++b)
        key_limit[b] = (a.causal ? min(item.valid, qw0 + 32 * b + r + 1) : item.valid) - 4 * hi;
    auto need_mask = [&](int b, int t) {
        return (t + 1) * A64_KT > item.valid || (a.causal && (t + 1) * A64_KT > qw0 + 32 * b);
    };
    auto mask_block = [&](int b, int t) {        // (callers sit right behind the block's last score MFMA: settle first)
        a64_mfma_settle();
        const int rel = key_limit[b] - t * A64_KT;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int R = 0; R < 16; ++R)
                if (32 * kb + 8 * (R >> 2) + (R & 3) >= rel) sacc[b][kb][R] = -INFINITY;
    };
    auto block_max = [&](int b) {          // over the tile's 64 keys of the lane's query (both halves)
        float mx = max3(sacc[b][0][0], sacc[b][0][1], sacc[b][0][2]);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int R = (kb == 0 ? 3 : 0); R < 16; R += 2) {
                if (R + 1 < 16) mx = max3(mx, sacc[b][kb][R], sacc[b][kb][R + 1]);
                else mx = max3(mx, sacc[b][kb][R], sacc[b][kb][R]);
            }
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
        return max3(__uint_as_float(sw[0]), __uint_as_float(sw[1]), __uint_as_float(sw[1]));
    };
    // softmax of block b, in 16 half-quads: half-quad j covers registers 2 (j & 1) .. + 1 of quad q = j >> 1 =
    // (kb, s', h) = (q >> 2, (q >> 1) & 1, q & 1), register base 4 s' + 8 h: two exponentials, their sum, one packed
    // pair = dword 2 h + (j & 1) of pf[b][2 kb + s']
    auto sm_exp = [&](auto bc, auto jc, f32x2& e) {
        constexpr int b = decltype(bc)::value, j = decltype(jc)::value;
        constexpr int q = j >> 1, kb = q >> 2, sp = (q >> 1) & 1, h = q & 1, R = 4 * sp + 8 * h + 2 * (j & 1);
        e.x = __builtin_amdgcn_exp2f(sacc[b][kb][R]);
        e.y = __builtin_amdgcn_exp2f(sacc[b][kb][R + 1]);
    };
    auto sm_pack = [&](auto bc, auto jc, const f32x2& e) {
        constexpr int b = decltype(bc)::value, j = decltype(jc)::value;
        constexpr int q = j >> 1, kb = q >> 2, sp = (q >> 1) & 1, h = q & 1;
        psum2[b] += e;
        pf[b][2 * kb + sp][2 * h + (j & 1)] = P::pack2(e.x, e.y);
    };
    // some p of the tile is past the ceiling: move block b's shift to the tile's maximum (cold)
    auto rebase = [&](auto bc) {
        constexpr int b = decltype(bc)::value;
        const float d = fmaxf(block_max(b), 0.f);            // (a fully masked tile: -inf -> 0)
        const float alpha = __builtin_amdgcn_exp2f(-d);
        lrun[b] *= alpha;
        [&]<int... DB>(std::integer_sequence<int, DB...>) { (A64O<P, 4 * b + DB>::scale(alpha), ...); }(std::make_integer_sequence<int, 4>{});
        psum2[b] = f32x2{0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int sp = 0; sp < 2; ++sp)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int w2 = 0; w2 < 2; ++w2) {
                        const int R = 4 * sp + 8 * h + 2 * w2;
                        const float e0 = __builtin_amdgcn_exp2f(sacc[b][kb][R] - d), e1 = __builtin_amdgcn_exp2f(sacc[b][kb][R + 1] - d);
                        psum2[b] += f32x2{e0, e1};
                        pf[b][2 * kb + sp][2 * h + w2] = P::pack2(e0, e1);
                    }
        shift[b] += d;
#pragma unroll
        for (int R = 0; R < 16; ++R) cinit[b][R] = -shift[b];
        asm volatile("" : "+v"(cinit[b]));          // (opaque: or the splat is re-made from one register in front of every use)
    };
    // end of block b's softmax of a tile: the ceiling test (wave-uniform, cold branch), the running sum
    auto sm_close = [&](auto bc) {
        constexpr int b = decltype(bc)::value;
        float ps = hsum2(psum2[b]);
        if (__any(ps > (a.rebase_always ? 1.0f : P::kProbCeil))) {
            rebase(bc);
            ps = hsum2(psum2[b]);
        }
        lrun[b] += ps;
        psum2[b] = f32x2{0.f, 0.f};
    };

    // ---- first tile: scores of both blocks from C = 0, the shift, block A's softmax --------------------------------
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(16) : "memory");        // Q and tile 0 (tiles 1, 2: 8 pieces each may be in flight)
    if (ntiles <= 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int b = 0; b < NQB; ++b)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) asm volatile("" : "+a"(qf[b][ks]));
    __syncthreads();
    u32x4 ring[A64_D];
    if (ntiles > 0) {
        // 16 K fragments (ks, kb), each feeding both blocks
        auto rd0 = [&](auto ic, u32x4& dst) {
            constexpr int i = decltype(ic)::value;
            a64_read<(i % 2) * 8192>(dst, kaddr[i / 2]);
        };
        [&]<int... I>(std::integer_sequence<int, I...>) { (rd0(std::integral_constant<int, I>{}, ring[I]), ...); }(std::make_integer_sequence<int, A64_D>{});
        [&]<int... I>(std::integer_sequence<int, I...>) {
            ([&] {
                constexpr int ks = I / 2, kb = I % 2;
                lgkm_wait32<(15 - I < A64_D - 1) ? (15 - I) : (A64_D - 1)>(ring[I % A64_D]);
#pragma unroll
                for (int b = 0; b < NQB; ++b) {
                    if constexpr (ks == 0) A64Op<P>::s_first(sacc[b][kb], ring[I % A64_D], qf[b][ks], zero16);
                    else A64Op<P>::s_acc(sacc[b][kb], ring[I % A64_D], qf[b][ks]);
                }
                if constexpr (I + A64_D < 16) rd0(std::integral_constant<int, I + A64_D>{}, ring[I % A64_D]);
                __builtin_amdgcn_sched_barrier(0);
            }(), ...);
        }(std::make_integer_sequence<int, 16>{});
        a64_mfma_settle();
#pragma unroll
        for (int b = 0; b < NQB; ++b)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) asm volatile("" : "+v"(sacc[b][kb]));
#pragma unroll
        for (int b = 0; b < NQB; ++b) {
            if (need_mask(b, 0)) mask_block(b, 0);
            const float m0 = block_max(b);
            shift[b] = (m0 == -INFINITY) ? 0.f : m0;
#pragma unroll
            for (int R = 0; R < 16; ++R) cinit[b][R] = -shift[b];
            asm volatile("" : "+v"(cinit[b]));
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int R = 0; R < 16; ++R) sacc[b][kb][R] -= shift[b];
        }
        // block A's softmax of tile 0, bare
        [&]<int... J>(std::integer_sequence<int, J...>) {
            ([&] { f32x2 e; sm_exp(std::integral_constant<int, 0>{}, std::integral_constant<int, J>{}, e);
                   sm_pack(std::integral_constant<int, 0>{}, std::integral_constant<int, J>{}, e); }(), ...);
        }(std::make_integer_sequence<int, 16>{});
        sm_close(std::integral_constant<int, 0>{});
        // (VALU-written P fragments -> MFMA operands: the packs may not sink to the MFMA that reads them)
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) asm volatile("" : "+v"(pf[0][sidx]));
        asm volatile("s_nop 1" ::: "memory");
    }
#ifdef PPG_ATTN_TIMING
    const unsigned long long wg_t1 = __builtin_amdgcn_s_memrealtime();
#endif

    // ---- the tile loop -----------------------------------------------------------------------------------------------
    // Iteration t = 1 .. T-1, 64 steps, the fragment of a step read 8 steps ahead of its MFMA:
    //    0..31  even i: S_A(t)  on K(t) fragment (ks, kb) = (n / 2, n % 2), n = i / 2        odd i: P V_A(t-1) on V^T(t-1) fragment
    //   32..63  even i: S_B(t)                                                               odd i: P V_B(t-1)    (s, db) = (n / 4, n % 4)
    //   gaps 0..31: [mask B(t-1)] exp_B(t-1), close B        gaps 32..63: [mask A(t)] exp_A(t), close A
    // Score and output MFMAs ALTERNATE: an accumulator is touched by every 4th (scores) / 8th (output) MFMA of the
    // stream.  With a phase of nothing but one block's scores -- two accumulators in turn -- every MFMA waited for the
    // one two slots ahead of it to write its result back: 60 - 78 cycles per MFMA against 43 in the output phases
    // (per-phase stamps of the first version, profiles/r5_attn64_phases.txt).
    // The price is one tile more of latency in the pipeline: the product of tile t-1 runs beside the scores of tile t.
    // Barrier at step 48 (tile t + 1 landed, issued a whole iteration ago), then the DMA of tile t + 2 into the slots of
    // K(t - 1) and V^T(t - 2), one piece per gap.
    int kdelta = 0, vdelta = 0;            // byte steps of the ring addresses into the next iteration's slots
    auto read_of = [&](auto jc, u32x4& dst) {      // the fragment of step j (j >= 64: of the next iteration)
        constexpr int j = decltype(jc)::value % 64, n = (j % 32) / 2;
        if constexpr (j % 2 == 0) a64_read<(n % 2) * 8192>(dst, kaddr[n / 2]);
        else a64_read<(n % 4) * 4096>(dst, vaddr[n / 4]);
    };
#if defined(PPG_ATTN_TIMING) && defined(A64_STAMPS)
    auto stamp = [&](int tt, int k) {
        if (a.dbg && blockIdx.x == 0 && lane == 0 && tt >= 2 && tt < 4) a.dbg[(wave * 2 + (tt - 2)) * 8 + k] = __builtin_amdgcn_s_memtime();
    };
#else
    auto stamp = [&](int, int) {};
#endif
    f32x2 ehold;
    // gap g of a 32-gap phase carries half-quad g / 2 of block b: the exponentials in the even gap, sum + pack in the odd one
    auto sm_fill = [&](auto bc, auto gc) {
        constexpr int g = decltype(gc)::value;
        if constexpr (g % 2 == 0) sm_exp(bc, std::integral_constant<int, g / 2>{}, ehold);
        else sm_pack(bc, std::integral_constant<int, g / 2>{}, ehold);
    };
    auto run32 = [&](auto halfc, auto filler) {
        constexpr int half = decltype(halfc)::value;       // 0: block A's scores + product, 1: block B's
        [&]<int... G>(std::integer_sequence<int, G...>) {
            ([&] {
                constexpr int i = 32 * half + G, n = G / 2;
                u32x4& frag = ring[i % A64_D];
                lgkm_wait32<A64_D - 1>(frag);
                if constexpr (G % 2 == 0) {
                    if constexpr (n / 2 == 0) A64Op<P>::s_first(sacc[half][n % 2], frag, qf[half][0], cinit[half]);
                    else A64Op<P>::s_acc(sacc[half][n % 2], frag, qf[half][n / 2]);
                } else {
                    A64O<P, 4 * half + n % 4>::acc(frag, pf[half][n / 4]);
                }
                if constexpr (i == 56) {          // the reads from here on are the next iteration's
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) kaddr[ks] += (uint32_t)kdelta;
#pragma unroll
                    for (int sidx = 0; sidx < 4; ++sidx) vaddr[sidx] += (uint32_t)vdelta;
                }
#ifndef A64_ABL_NOREAD
                read_of(std::integral_constant<int, i + A64_D>{}, frag);
#endif
                __builtin_amdgcn_sched_barrier(0);      // (the gap's VALU work BEHIND the MFMA)
                filler(std::integral_constant<int, G>{});
                __builtin_amdgcn_sched_barrier(0);
            }(), ...);
        }(std::make_integer_sequence<int, 32>{});
    };
    if (ntiles > 1) {
        // tile 1 landed for every wave; the ring starts on K(1) / V^T(0)
        if (ntiles > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) kaddr[ks] += A64_TILE;
        [&]<int... I>(std::integer_sequence<int, I...>) { (read_of(std::integral_constant<int, I>{}, ring[I]), ...); }(std::make_integer_sequence<int, A64_D>{});
    }
    for (int tt = 1; tt < ntiles; ++tt) {
        kdelta = ((tt + 1) % A64_NK == 0) ? -(A64_NK - 1) * A64_TILE : A64_TILE;       // K(tt) -> K(tt + 1)
        vdelta = (tt % A64_NV == 0) ? -(A64_NV - 1) * A64_TILE : A64_TILE;             // V^T(tt - 1) -> V^T(tt)
        const bool maskb = need_mask(1, tt - 1), maska = need_mask(0, tt);
        stamp(tt, 0);
        run32(std::integral_constant<int, 0>{}, [&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if constexpr (g == 0) { if (maskb) mask_block(1, tt - 1); }
#ifndef A64_ABL_NOSOFTMAX
            sm_fill(std::integral_constant<int, 1>{}, gc);
#endif
            if constexpr (g == 15) stamp(tt, 1);
            if constexpr (g == 31) sm_close(std::integral_constant<int, 1>{});
        });
        stamp(tt, 2);
        run32(std::integral_constant<int, 1>{}, [&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if constexpr (g == 0) { if (maska) mask_block(0, tt); }
            if constexpr (g == 16) {
                stamp(tt, 3);
#ifndef A64_ABL_NOBARRIER
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_barrier" ::: "memory");
#endif
                stamp(tt, 4);
            }
#ifndef A64_ABL_NODMA
            if constexpr (g >= 17 && g <= 24) { if (tt + 2 < ntiles) stage_piece(tt + 2, std::integral_constant<int, g - 17>{}); }
#endif
#ifndef A64_ABL_NOSOFTMAX
            sm_fill(std::integral_constant<int, 0>{}, gc);
#endif
            if constexpr (g == 31) sm_close(std::integral_constant<int, 0>{});
        });
        stamp(tt, 5);
    }
    // the reads primed for an iteration that does not exist
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < A64_D; ++i) asm volatile("" : "+v"(ring[i]));
    if (ntiles > 0) {
        // the last tile's products: P V_A(T-1) with exp_B(T-1) in its gaps (a half-quad per gap), then P V_B(T-1);
        // vaddr points at V^T(T-1) (the last iteration moved it there; with one tile it never moved)
        if (ntiles > 1) {
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx) asm volatile("" : "+v"(vaddr[sidx]));
        }
        const bool maskb = need_mask(1, ntiles - 1);
        if (maskb) mask_block(1, ntiles - 1);
        auto rdv = [&](auto ic, u32x4& dst) {
            constexpr int i = decltype(ic)::value % 16;
            a64_read<(i % 4) * 4096>(dst, vaddr[i / 4]);
        };
        [&]<int... I>(std::integer_sequence<int, I...>) { (rdv(std::integral_constant<int, I>{}, ring[I]), ...); }(std::make_integer_sequence<int, A64_D>{});
        [&]<int... I>(std::integer_sequence<int, I...>) {
            ([&] {
                lgkm_wait32<(31 - I < A64_D - 1) ? (31 - I) : (A64_D - 1)>(ring[I % A64_D]);
                if constexpr (I < 16) A64O<P, I % 4>::acc(ring[I % A64_D], pf[0][I / 4]);
                else A64O<P, 4 + I % 4>::acc(ring[I % A64_D], pf[1][(I - 16) / 4]);
                if constexpr (I + A64_D < 32) rdv(std::integral_constant<int, I + A64_D>{}, ring[I % A64_D]);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (I < 16) {
                    f32x2 e;
                    sm_exp(std::integral_constant<int, 1>{}, std::integral_constant<int, I>{}, e);
                    sm_pack(std::integral_constant<int, 1>{}, std::integral_constant<int, I>{}, e);
                    if constexpr (I == 15) {
                        sm_close(std::integral_constant<int, 1>{});
#pragma unroll
                        for (int sidx = 0; sidx < 4; ++sidx) asm volatile("" : "+v"(pf[1][sidx]));
                        asm volatile("s_nop 1" ::: "memory");
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }(), ...);
        }(std::make_integer_sequence<int, 32>{});
    }
    a64_mfma_settle();
#ifdef PPG_ATTN_TIMING
    const unsigned long long wg_t2 = __builtin_amdgcn_s_memrealtime();
#endif

    // ---- O / l -> AO: a lane's registers (0-3, 8-11) and (4-7, 12-15) of block db are features 32 db + 8 hi .. + 7 and
    // 32 db + 16 + 8 hi .. + 7 of the head (V^T rows in pair_row order): two 16-byte stores
    auto store_block = [&](auto bc) {
        constexpr int b = decltype(bc)::value;
        const int tq = qw0 + 32 * b + r;
        if (tq >= frames16) return;                         // (rows past the window's padded rows belong to the next window)
        const int m = item.tok_off + tq;
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(lrun[b]), __float_as_uint(lrun[b]), false, false);
        const float l = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        const float inv = l > 0.f ? 1.0f / l : 0.f;
        [&]<int... DB>(std::integer_sequence<int, DB...>) {
            ([&] {
                float o[16];
                A64O<P, 4 * b + DB>::read(o);
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int n = head * DH + 32 * DB + 16 * half + 8 * hi;
                    char* dst = a.ao + (a.ao_tiled ? ao32_byte(m, n, a.H) : (size_t)m * a.H * 2 + (size_t)n * 2);
                    *reinterpret_cast<u32x4*>(dst) = u32x4{
                        P::pack2(o[4 * half + 0] * inv, o[4 * half + 1] * inv), P::pack2(o[4 * half + 2] * inv, o[4 * half + 3] * inv),
                        P::pack2(o[8 + 4 * half + 0] * inv, o[8 + 4 * half + 1] * inv), P::pack2(o[8 + 4 * half + 2] * inv, o[8 + 4 * half + 3] * inv)};
                }
            }(), ...);
        }(std::make_integer_sequence<int, 4>{});
    };
    store_block(std::integral_constant<int, 0>{});
    store_block(std::integral_constant<int, 1>{});
#ifdef PPG_ATTN_TIMING
    if (a.dbg && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long* rec = a.dbg + 64 + 4 * (size_t)blockIdx.x;
        rec[0] = wg_t0; rec[1] = __builtin_amdgcn_s_memrealtime();
        rec[2] = (unsigned long long)item.valid | ((wg_t1 - wg_t0) << 16) | ((wg_t2 - wg_t0) << 40);
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        rec[3] = hwid;
    }
#endif
}

template <class P>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void attn64_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int item_index = blockIdx.x / a.heads;
    const AttnItem item = a.items[item_index];
    attn64_body<P, 2>(a, item, blockIdx.x - item_index * a.heads, smem);
}

}  // namespace

namespace ppg {

int attn64_query_tile() { return 256; }

hipError_t launch_attn64(int precision, const AttnArgs& args, int nitems, int heads, hipStream_t s) {
    AttnArgs a = args;
    const char* mode = getenv("PPGS_AMD_ATTN_REBASE");
    a.rebase_always = (mode && strcmp(mode, "always") == 0) ? 1 : 0;
    static ppg::LdsLimit limit_bf16;
#if PPG_OTHER_PRECISIONS
    static ppg::LdsLimit limit_f16;
#endif
    if (precision == PPG_PRECISION_BF16) {
        auto kern = attn64_kernel<PrecBF16>;
        const hipError_t e = limit_bf16.ensure(reinterpret_cast<const void*>(kern), A64_LDS);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(nitems * heads), dim3(256), A64_LDS, s, a);
        return hipGetLastError();
    }
#if PPG_OTHER_PRECISIONS
    if (precision == PPG_PRECISION_FP16) {
        auto kern = attn64_kernel<PrecF16>;
        const hipError_t e = limit_f16.ensure(reinterpret_cast<const void*>(kern), A64_LDS);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(nitems * heads), dim3(256), A64_LDS, s, a);
        return hipGetLastError();
    }
#endif
    return hipErrorInvalidValue;
}

}  // namespace ppg
