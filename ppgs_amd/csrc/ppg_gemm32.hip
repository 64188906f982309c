// Feature-split GEMM for the projections of the wav2vec 2.0 body (HF Wav2Vec2EncoderLayer: attention q / k / v /
// out_proj, feed_forward.intermediate_dense / output_dense, the feature projection):
//     Y[m][n] = act_fn(sum_k X[m][k] W[n][k] + bias[n]) [+ R[m][n]]      X row-major 16-bit, W as A-fragment images
// on v_mfma_f32_32x32x16, the machinery of ppg_layer32.hip / ppg_head32.hip: one workgroup = 128 or 160 token rows x
// 256 output features, wave w owns features 64 w .. + 63 (two row blocks), weights go from host-packed fragment
// images straight into registers (three sets of 16 fragments = one K chunk of 128 each: chunk c + 2's travel under
// chunks c and c + 1), activations are B fragments read from a token-major LDS tile (rows of 272 bytes = 17 x 16: the
// 32 rows of a fragment read fall into 16 different 16-byte bank groups), three tile buffers.  The K loop is ONE
// stream of LDS fragment reads that runs on across the chunks (a ring of 8 in flight), with the chunk's requests,
// the tile writes of the next chunk and the chunk's one barrier riding in its steps (see `chunk` below).
// Measured per workgroup at 16 x 499 frames (PPG_GEMM_TIMING stamps, cycles): prologue 5 - 7 k, a chunk 2.8 - 3.1 k
// (64 MFMAs per wave = 2.05 k; 2.45 k with the weights hot in L2; the requests' issue slots are the rest),
// epilogue 8 k (Q / K), 14 k (residual: the 25 MB of residual rows all workgroups read at once), 15 k (GELU: VALU).
// Two things that cost more than the arithmetic before they were found: (1) stores from the ACCUMULATOR layout
// (a lane owns 16 features of ONE token: 64 rows per instruction) -- the epilogue turns the block through LDS into
// row order; (2) the compiler's wait-count insertion across the `m < M` control flow around the stores (a
// vmcnt(0) per row group = the previous store's acknowledgement, 300 cycles per store instruction:
// tools/wait_scan.py audits the built library for the pattern).
#include "ppg_layer32.h"
#include <cstdio>
#include <map>
#include <mutex>

namespace {

constexpr int KC = 128;                // K per chunk
constexpr int KSC = KC / 16;           // K-steps per chunk
constexpr int GROW = KC * 2 + 16;      // bytes per tile row in LDS

// step i = (ks = i / GT, tb = i % GT): rows 32 tb .. of the tile, K-step ks
template <int GT>
struct OffTile { static constexpr int at(int i) { return (32 * (i % GT)) * GROW + (i / GT) * 32; } };

#ifdef PPG_GEMM_TIMING
__device__ unsigned long long g_gemm_stamps[2][32];
#define GSTAMP(k) do { if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1) && blockIdx.y == (blockIdx.x ? gridDim.y - 1 : 0)) g_gemm_stamps[blockIdx.x ? 1 : 0][k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define GSTAMP(k) do {} while (0)
#endif

// GT = token blocks of 32 per workgroup (4 or 5: the launcher takes the one that fills the chip's rounds better)
// MODE: what the epilogue does, fixed at compile time (with run-time flags every group of rows went through five
// wave-uniform branches, and the straight-line epilogue runs once per workgroup from a cold instruction cache):
//   0  + fp32 residual -> fp32 rows                       (attention out-projection, FFN output)
//   1  exact GELU -> 16-bit rows                          (FFN intermediate)
//   2  rows past the item's valid frames zeroed -> fp32 and 16-bit rows   (feature projection)
//   3  Q | K | V of an attention layer: passes below v_pass0 -> 16-bit rows of leading dimension ld_out (Q | K per
//      token); the V passes run with the MFMA operands SWAPped, so that the accumulator comes out transposed (a lane
//      owns one V^T row and 16 tokens), and store V^T in attn_kernel's layout (rows in tile order -- the image's rows
//      are pair_row()-permuted --, columns of a 32-token group at position 8 g + 4 e + r for token 16 e + 4 g + r)
template <class P, int GT, int MODE, bool SWAP>
__device__ __forceinline__ void gemm32_body(const Gemm32Args& a, char* smem) {
    constexpr bool RES = MODE == 0, GELU = MODE == 1, WIN = MODE == 2, OUT32 = MODE == 0 || MODE == 2, OUT16 = MODE != 0;
    constexpr int GTOK = 32 * GT, GBUF = GTOK * GROW;
    constexpr int SROWS = GTOK / 4, NST = SROWS / 4;                 // rows staged per wave, load instructions per wave and chunk
    constexpr int STEPS = KSC * GT;                                  // LDS fragments (of two MFMAs each) per chunk
    constexpr int RD = 8;                                            // fragment reads in flight (STEPS % RD == 0: the ring runs on across chunks)
    static_assert(STEPS % RD == 0, "ring slot of a step must not depend on the chunk");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tok = lane & 31, hh = lane >> 5;
    const int m0 = blockIdx.x * GTOK;
    const int pass = blockIdx.y;                                  // 256 output features
    const uint32_t lds0 = lds_addr32(smem);
    const uint32_t voff = lane * 16;
    const int chunks = a.K / KC;

    // activation rows of a chunk: wave w stages rows SROWS w .. + SROWS - 1, 4 rows (of 256 bytes) per load instruction.
    // All loads of the main loop are asm (the waits are counted by hand: a load the compiler sees gets a compiler-made
    // vmcnt at its use that does not know about the asm loads in flight).
    const int srow = SROWS * wave + (lane >> 4), scol = (lane & 15) * 16;
    u32x4 stg[NST];
    const char* xrow[NST];
    const size_t lda = a.lda_bytes ? (size_t)a.lda_bytes : (size_t)a.K * 2;
#pragma unroll
    for (int i = 0; i < NST; ++i) xrow[i] = a.x + (size_t)min(m0 + srow + 4 * i, a.M - 1) * lda + scol;   // rows past M re-read the last one (never stored)
    auto fetch = [&](int c) {
#pragma unroll
        for (int i = 0; i < NST; ++i) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(stg[i]) : "v"(xrow[i] + (size_t)c * KC * 2) : "memory");
    };
    const uint32_t stash0 = lds0 + (uint32_t)(srow * GROW + scol);

    // three register sets of 16 weight fragments; the third lives in the accumulation registers (an MFMA takes its
    // A operand from either file, a global load writes either): with all three in the 256 architectural registers
    // the compiler moved fragments between the files while their loads were still in flight
    u32x4 w1f[16], w2f[16], w3f[16];
    // image order [pass][wave][chunk][rb][ks]: 16 fragments per (wave, chunk); a chunk USES them in the order
    // (rb 0, ks 0), (rb 1, ks 0), (rb 0, ks 1) ..: fragment of use-position j
    const char* wimg = a.w_img + (((size_t)pass * 4 + wave) * chunks) * 16 * 1024;
    auto load_frag = [&]<int J, bool TO_ACC>(u32x4 (&wf)[16], const char* base) {
        constexpr int k = (J % 2) * 8 + J / 2;
        if constexpr (TO_ACC) gload128_acc<(k % 4) * 1024>(wf[k], voff, base + (k / 4) * 4096);
        else gload_frag<k>(wf[k], voff, base);
    };

    f32x16 acc[2][GT];
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const uint32_t rb0 = lds0 + (uint32_t)(tok * GROW + hh * 16);
    u32x4 ring[RD];

    // ---- prologue: everything chunk 0 and half of chunk 1 need is requested at once
    GSTAMP(0);
    fetch(0);
    [&]<int... J>(std::integer_sequence<int, J...>) { (load_frag.template operator()<J, false>(w1f, wimg), ...); }(std::make_integer_sequence<int, 16>{});
    const char* wimg1 = wimg + (size_t)(chunks > 1 ? 1 : 0) * 16 * 1024;
    [&]<int... J>(std::integer_sequence<int, J...>) { (load_frag.template operator()<J, false>(w2f, wimg1), ...); }(std::make_integer_sequence<int, 8>{});
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
#pragma unroll
    for (int i = 0; i < NST; ++i) {
        asm volatile("" : "+v"(stg[i]));
        asm volatile("ds_write_b128 %0, %1" :: "v"(stash0 + (uint32_t)(4 * i * GROW)), "v"(stg[i]) : "memory");
    }
    fetch(chunks > 1 ? 1 : 0);
    [&]<int... J>(std::integer_sequence<int, J...>) { (load_frag.template operator()<8 + J, false>(w2f, wimg1), ...); }(std::make_integer_sequence<int, 8>{});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    [&]<int... I>(std::integer_sequence<int, I...>) { (ds_read128<OffTile<GT>::at(I)>(ring[I], rb0), ...); }(std::make_integer_sequence<int, RD>{});
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(16 + NST) : "memory");
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(w1f[i]));
    __builtin_amdgcn_sched_barrier(0);

    // Chunk c (c % 3 = B: tile buffer B, fragments in `cur`) as STEPS steps of one LDS fragment and two MFMAs each;
    // the ring of RD fragment reads runs on into the next chunk's tile, so the matrix pipe never waits for a chunk
    // to start.  Riding in the steps:
    //   0 .. 7            fragments (use positions 0 .. 7) of chunk c + 2 -> the set chunk c - 1 used
    //   8                 s_waitcnt vmcnt(16): everything but the 16 youngest requests -- the second half of chunk
    //                     c + 1's fragments and the 8 just issued -- has landed: the rows of chunk c + 1 and the
    //                     first half of its fragments (the second half is covered by chunk c + 1's own wait)
    //   8 .. 8 + NST - 1  rows of chunk c + 1: registers -> tile buffer (B + 1) % 3, then the rows of chunk c + 2
    //                     requested into the same registers (an LDS write reads its data when it issues)
    //   .. + 8            fragments (use positions 8 .. 15) of chunk c + 2
    //   STEPS - RD        s_barrier: the next read is the first of the next chunk's tile; every wave's writes to it
    //                     are older than the RD - 1 LDS operations allowed in flight.  The same barrier orders
    //                     chunk c - 1's reads of buffer (B + 2) % 3 before chunk c + 1's writes to it.
    auto chunk = [&](int c, u32x4 (&cur)[16], u32x4 (&nxt2)[16], auto b_tag, auto first_tag, auto acc_file_tag) {
        constexpr int B = decltype(b_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value;
        constexpr bool TO_ACC = decltype(acc_file_tag)::value;      // nxt2 is the set kept in accumulation registers
        const int c2 = c + 2 < chunks ? c + 2 : c;
        const char* nbase = wimg + (size_t)c2 * 16 * 1024;
        const uint32_t tile = rb0 + (uint32_t)(B * GBUF), tile_next = rb0 + (uint32_t)(((B + 1) % 3) * GBUF);
        const uint32_t stash = stash0 + (uint32_t)(((B + 1) % 3) * GBUF);
        auto step = [&]<int I>() {
            constexpr int ks = I / GT, tb = I % GT;
            u32x4& bf = ring[I % RD];
            lgkm_wait32<RD - 1>(bf);
            if constexpr (FIRST && ks == 0) {
                acc[0][tb] = SWAP ? P::mma32(bf, cur[0], zero) : P::mma32(cur[0], bf, zero);
                acc[1][tb] = SWAP ? P::mma32(bf, cur[8], zero) : P::mma32(cur[8], bf, zero);
            } else {
                acc[0][tb] = SWAP ? P::mma32(bf, cur[ks], acc[0][tb]) : P::mma32(cur[ks], bf, acc[0][tb]);
                acc[1][tb] = SWAP ? P::mma32(bf, cur[8 + ks], acc[1][tb]) : P::mma32(cur[8 + ks], bf, acc[1][tb]);
            }
            static_assert(STEPS >= 16 + NST + RD, "requests and writes before the barrier step");
            if constexpr (I < 8) {
                load_frag.template operator()<I, TO_ACC>(nxt2, nbase);
            } else if constexpr (I < 8 + NST) {
                constexpr int k = I - 8;
                if constexpr (k == 0) {
                    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
#pragma unroll
                    for (int i = 0; i < NST; ++i) asm volatile("" : "+v"(stg[i]));
                }
                asm volatile("ds_write_b128 %0, %1" :: "v"(stash + (uint32_t)(4 * k * GROW)), "v"(stg[k]) : "memory");
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(stg[k]) : "v"(xrow[k] + (size_t)c2 * KC * 2) : "memory");
            } else if constexpr (I < 16 + NST) {
                load_frag.template operator()<I - NST, TO_ACC>(nxt2, nbase);
            }
            if constexpr (I == STEPS - RD) __builtin_amdgcn_s_barrier();
            if constexpr (I + RD < STEPS) ds_read128<OffTile<GT>::at(I + RD)>(bf, tile);
            else ds_read128<OffTile<GT>::at(I + RD - STEPS)>(bf, tile_next);
        };
        [&]<int... I>(std::integer_sequence<int, I...>) { (step.template operator()<I>(), ...); }(std::make_integer_sequence<int, STEPS>{});
    };
    GSTAMP(1);
    chunk(0, w1f, w3f, std::integral_constant<int, 0>{}, std::true_type{}, std::true_type{});
    GSTAMP(2);
    for (int c = 1; c < chunks; c += 3) {
        chunk(c, w2f, w1f, std::integral_constant<int, 1>{}, std::false_type{}, std::false_type{});
        if (c == 1) GSTAMP(3);
        if (c + 1 >= chunks) break;
        chunk(c + 1, w3f, w2f, std::integral_constant<int, 2>{}, std::false_type{}, std::false_type{});
        if (c == 1) GSTAMP(4);
        if (c + 2 >= chunks) break;
        chunk(c + 2, w1f, w3f, std::integral_constant<int, 0>{}, std::false_type{}, std::true_type{});
        if (c == 1) GSTAMP(5);
    }
    // the last chunks' clamped requests and the ring's reads past the end
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < RD; ++i) asm volatile("" : "+v"(ring[i]));
#pragma unroll
    for (int i = 0; i < NST; ++i) asm volatile("" : "+v"(stg[i]));
#pragma unroll
    for (int i = 0; i < 16; ++i) { asm volatile("" : "+v"(w1f[i])); asm volatile("" : "+v"(w2f[i])); asm volatile("" : "+a"(w3f[i])); }
    __builtin_amdgcn_sched_barrier(0);
    GSTAMP(6);

    // ---- epilogue.  The accumulators hold, per lane, 16 consecutive features of ONE token: stored from there, every
    // 16-byte access of an instruction lands in a different row (64 rows 3 KiB apart) and the address path takes 200
    // (loads) to 500 (stores) cycles per instruction -- 25 k of the out-projection's 50 k cycles per workgroup.  So
    // each wave turns its 32 tokens x 64 features of a token block through a private LDS scratch (rows of 272 bytes)
    // into ROW order -- lane l: token 4 i + (l >> 4), features 4 (l & 15) .. + 3 -- where an instruction covers 4 rows
    // of 256 contiguous bytes; bias, activation, residual and the stores happen there.  The residual rows are
    // requested up front (a load cannot move above a store it may alias).  LDS operations of one wave execute in
    // order: no wait between a block's reads and the next block's writes.
    if constexpr (SWAP) {
        // V^T: lane = row 64 wave + 32 rb + (l & 31) of this pass's 256, registers 4 q + r = token 8 q + 4 hh + r of the
        // block; (q, q + 2) are the 8 consecutive columns 16 (q & 1) + 8 hh .. + 7 of the 32-token group.  Item rows
        // start at multiples of 32 (launch_gemm32 checks rows_per_item): column = token row.
        const int vrow0 = 256 * (pass - a.v_pass0) + 64 * wave + tok;
        float bv[2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) bv[rb] = a.bias[256 * a.v_pass0 + pair_row(vrow0 + 32 * rb)];
        asm volatile("" : "+v"(bv[0]), "+v"(bv[1]));
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            char* rowp = a.vt + (size_t)(vrow0 + 32 * rb) * a.vt_ld * 2;
#pragma unroll
            for (int t = 0; t < GT; ++t) {
                if (m0 + 32 * t >= a.M) continue;
                const f32x16& c = acc[rb][t];
                // (as ppg_layer32.h qkv_tail: the block's two halves after one v_permlane16_swap per dword -- an
                // instruction writes 16 V^T rows x 64 contiguous bytes instead of 32 rows x 32)
                u32x4 half[2], lo, hi;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
                    half[s2] = u32x4{
                        P::pack2(c[4 * s2 + 0] + bv[rb], c[4 * s2 + 1] + bv[rb]), P::pack2(c[4 * s2 + 2] + bv[rb], c[4 * s2 + 3] + bv[rb]),
                        P::pack2(c[4 * (s2 + 2) + 0] + bv[rb], c[4 * (s2 + 2) + 1] + bv[rb]), P::pack2(c[4 * (s2 + 2) + 2] + bv[rb], c[4 * (s2 + 2) + 3] + bv[rb])};
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const auto sw = __builtin_amdgcn_permlane16_swap(half[0][d], half[1][d], false, false);
                    lo[d] = sw[0]; hi[d] = sw[1];
                }
                // lane rows 0 .. 3 hold the 16-byte pieces 0, 2, 1, 3 of the 32 columns of V^T row (l & 15) resp. 16 + (l & 15)
                const int k16 = lane >> 4;
                char* dst = rowp + ((ptrdiff_t)((lane & 15) - tok) * a.vt_ld + m0 + 32 * t) * 2 + 16 * (2 * (k16 & 1) + (k16 >> 1));
                *reinterpret_cast<u32x4*>(dst) = lo;
                *reinterpret_cast<u32x4*>(dst + (size_t)16 * a.vt_ld * 2) = hi;
            }
        }
        GSTAMP(7);
        return;
    }
    const int fbase = 256 * pass + 64 * wave;
    const int r4 = lane >> 4, c16 = lane & 15;
    float4 res[RES ? GT : 1][8];
    if constexpr (RES) {
#pragma unroll
        for (int t = 0; t < GT; ++t)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int m = min(m0 + 32 * t + 4 * i + r4, a.M - 1);
                res[t][i] = *reinterpret_cast<const float4*>(a.residual + (size_t)m * a.N + fbase + 4 * c16);
            }
    }
    float4 bias4 = *reinterpret_cast<const float4*>(a.bias + fbase + 4 * c16);
    GSTAMP(8);
    __syncthreads();                                                    // (the other waves' reads of the last tile)
    // Every load has landed before the first store is issued: the rows are stored under `m < M`, and across that
    // control flow the compiler's wait-count insertion no longer knows how many stores follow a load -- it waits
    // with vmcnt(0) before each group's first use of a loaded value, i.e. for the previous group's STORE to be
    // acknowledged: 300 cycles per store instruction, 20 k of the out-projection's 44 k cycles per workgroup.
    asm volatile("" : "+v"(bias4.x), "+v"(bias4.y), "+v"(bias4.z), "+v"(bias4.w));
    if constexpr (RES) {
#pragma unroll
        for (int t = 0; t < GT; ++t)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(res[t][i].x), "+v"(res[t][i].y), "+v"(res[t][i].z), "+v"(res[t][i].w));
    }
    GSTAMP(9);
    const uint32_t scr_w = lds0 + (uint32_t)(wave * 32 * GROW + tok * GROW + hh * 64);
    const uint32_t scr_r = lds0 + (uint32_t)(wave * 32 * GROW + r4 * GROW + c16 * 16);
#pragma unroll
    for (int t = 0; t < GT; ++t) {
#ifdef PPG_GEMM_TIMING
        if (t == 1) GSTAMP(13);
#endif
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const u32x4 v = {__float_as_uint(acc[rb][t][4 * q + 0]), __float_as_uint(acc[rb][t][4 * q + 1]),
                                 __float_as_uint(acc[rb][t][4 * q + 2]), __float_as_uint(acc[rb][t][4 * q + 3])};
                asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(scr_w), "v"(v), "n"(128 * rb + 16 * q) : "memory");
            }
        u32x4 rows[8];
        [&]<int... I>(std::integer_sequence<int, I...>) { (ds_read128<4 * I * GROW>(rows[I], scr_r), ...); }(std::make_integer_sequence<int, 8>{});
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef PPG_GEMM_TIMING
        if (t == 1) GSTAMP(14);
#endif
        // (all arithmetic of the block first, in straight-line code: the row groups' dependent chains -- the GELU
        // polynomial -- interleave; the stores' `m < M` control flow comes behind)
        float y[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            asm volatile("" : "+v"(rows[i]));
            y[i][0] = __uint_as_float(rows[i][0]) + bias4.x; y[i][1] = __uint_as_float(rows[i][1]) + bias4.y;
            y[i][2] = __uint_as_float(rows[i][2]) + bias4.z; y[i][3] = __uint_as_float(rows[i][3]) + bias4.w;
            if constexpr (GELU) {
                const f32x2 lo = gelu_erf_pair(f32x2{y[i][0], y[i][1]}), hi = gelu_erf_pair(f32x2{y[i][2], y[i][3]});
                y[i][0] = lo.x; y[i][1] = lo.y; y[i][2] = hi.x; y[i][3] = hi.y;
            }
            if constexpr (RES) { y[i][0] += res[t][i].x; y[i][1] += res[t][i].y; y[i][2] += res[t][i].z; y[i][3] += res[t][i].w; }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = m0 + 32 * t + 4 * i + r4;
            if (m >= a.M) continue;
            if constexpr (WIN) {
                const int item = m / a.rows_per_item;
                if (m - item * a.rows_per_item >= a.win[item].valid) y[i][0] = y[i][1] = y[i][2] = y[i][3] = 0.f;
            }
            const size_t at = (size_t)m * (MODE == 3 ? a.ld_out : a.N) + fbase + 4 * c16;
            if constexpr (OUT32) *reinterpret_cast<float4*>(a.out32 + at) = make_float4(y[i][0], y[i][1], y[i][2], y[i][3]);
            if constexpr (OUT16) *reinterpret_cast<uint2*>(a.out16 + at * 2) = make_uint2(P::pack2(y[i][0], y[i][1]), P::pack2(y[i][2], y[i][3]));
        }
#ifdef PPG_GEMM_TIMING
        if (t == 0) GSTAMP(10);
        if (t == 1) GSTAMP(11);
        if (t == GT - 1) GSTAMP(12);
#endif
    }
    GSTAMP(7);
}

template <class P, int GT, int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm32_kernel(Gemm32Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if constexpr (MODE == 3) {
        if ((int)blockIdx.y >= a.v_pass0) gemm32_body<P, GT, 3, true>(a, smem);
        else gemm32_body<P, GT, 3, false>(a, smem);
    } else {
        gemm32_body<P, GT, MODE, false>(a, smem);
    }
}

}  // namespace

namespace ppg {

hipError_t launch_gemm32(int precision, const Gemm32Args& a, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0 || a.N % 256 || a.K < KC || a.K % KC) return hipErrorInvalidValue;
    int cus = 256;
    {
        static int cached = 0;
        if (!cached) { hipDeviceProp_t prop; int dev = 0; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cached = prop.multiProcessorCount; else cached = 256; }
        cus = cached;
    }
    // rounds of workgroups over the CUs: the wider tile unless the narrower one wastes less of its last round
    auto cost = [&](int gt) {
        const long wgs = (long)((a.M + 32 * gt - 1) / (32 * gt)) * (a.N / 256);
        return (double)((wgs + cus - 1) / cus) * gt;                  // time ~ rounds x tokens per workgroup
    };
    const int gt = cost(4) < cost(5) ? 4 : 5;
#ifdef PPG_GEMM_TIMING
    struct Report { hipStream_t s; int M, N, K, act; ~Report() {
        static int calls = 0;
        if (++calls < 100 || calls > 110) return;
        unsigned long long h[2][32];
        if (hipStreamSynchronize(s) != hipSuccess || hipMemcpyFromSymbol(h, HIP_SYMBOL(g_gemm_stamps), sizeof(h)) != hipSuccess) return;
        for (int b = 0; b < 2; ++b)
            fprintf(stderr, "gemm32 M %d N %d K %d act %d, %s workgroup (cycles): prologue %llu, chunk0 %llu, c1 %llu, c2 %llu, c3 %llu, rest %llu, epilogue %llu (loads issued %llu, landed %llu, block 0 %llu, block 1 %llu, rb 0 done %llu; block 1: LDS turn %llu, rest %llu), total %llu\n", M, N, K, act, b ? "last" : "first",
                    h[b][1] - h[b][0], h[b][2] - h[b][1], h[b][3] - h[b][2], h[b][4] - h[b][3], h[b][5] - h[b][4], h[b][6] - h[b][5], h[b][7] - h[b][6],
                    h[b][8] - h[b][6], h[b][9] - h[b][6], h[b][10] - h[b][6], h[b][11] - h[b][6], h[b][12] - h[b][6], h[b][14] - h[b][13], h[b][11] - h[b][14], h[b][7] - h[b][0]);
    } } report{s, a.M, a.N, a.K, a.act_fn};
#endif
    // the three uses of the wav2vec2 body (anything else is refused: the epilogue is fixed at compile time)
    int mode = -1;
    if (a.vt) mode = -1;
    else if (a.act_fn == 0 && a.residual && a.out32 && !a.out16 && !a.win) mode = 0;
    else if (a.act_fn == 2 && !a.residual && !a.out32 && a.out16 && !a.win) mode = 1;
    else if (a.act_fn == 0 && !a.residual && a.out32 && a.out16 && a.win && a.rows_per_item > 0) mode = 2;
    if (a.vt && a.out16 && !a.residual && !a.out32 && !a.win && a.act_fn == 0 && a.ld_out >= 256 * a.v_pass0 && a.v_pass0 >= 0 &&
             a.v_pass0 <= a.N / 256 && a.rows_per_item > 0 && a.rows_per_item % 32 == 0 && a.M % 32 == 0) mode = 3;
    if (mode < 0) return hipErrorInvalidValue;
    auto launch = [&](auto kern, int g) {
        const size_t lds = 3 * (size_t)(32 * g) * GROW;
        const dim3 grid((a.M + 32 * g - 1) / (32 * g), a.N / 256);
        static std::mutex mu;
        static std::map<const void*, ppg::LdsLimit> limits;            // one per kernel instantiation
        hipError_t e;
        { std::lock_guard<std::mutex> lock(mu); e = limits[reinterpret_cast<const void*>(kern)].ensure(reinterpret_cast<const void*>(kern), lds); }
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
        return hipGetLastError();
    };
    auto pick = [&]<class P>(P) {
        switch (mode * 2 + (gt == 5)) {
        case 0: return launch(gemm32_kernel<P, 4, 0>, 4);
        case 1: return launch(gemm32_kernel<P, 5, 0>, 5);
        case 2: return launch(gemm32_kernel<P, 4, 1>, 4);
        case 3: return launch(gemm32_kernel<P, 5, 1>, 5);
        case 4: return launch(gemm32_kernel<P, 4, 2>, 4);
        case 5: return launch(gemm32_kernel<P, 5, 2>, 5);
        case 6: return launch(gemm32_kernel<P, 4, 3>, 4);
        default: return launch(gemm32_kernel<P, 5, 3>, 5);
        }
    };
    if (precision == PPG_PRECISION_BF16) return pick(PrecBF16{});
#ifndef PPG_ONLY_BF16
    if (precision == PPG_PRECISION_FP16) return pick(PrecF16{});
#endif
    return hipErrorInvalidValue;
}

}  // namespace ppg
