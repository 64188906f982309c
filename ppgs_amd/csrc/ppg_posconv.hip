// Grouped positional convolution of the wav2vec 2.0 body (HF Wav2Vec2PositionalConvEmbedding: Conv1d(768, 768,
// kernel 128, padding 64, groups 16) with the last output frame dropped, GELU, added to its input):
//     P[m][n] = X[m][n] + gelu(b[n] + sum_{tap < 128, ch < 48} w[n][ch][tap] x16[m + tap - 64][48 g + ch]),   g = n / 48
// rows outside the item's frames read as zeros.  One workgroup = 128 tokens of one item x one group (48 features):
//   * the group's 48 channels of the 256 rows the tile's taps reach are fetched ONCE into LDS (rows of 112 bytes =
//     7 x 16: conflict-free fragment reads) -- as a k-tap GEMM on linear_kernel every tap re-read its 64 rows from
//     global memory behind a barrier (128 stages of ~1.3 k cycles for 12 k cycles of MFMAs: 320 us per launch);
//   * the MFMA's K runs over (tap, channel): a B fragment (16 K x 32 tokens) of tap t, channels 16 ks .. is the window
//     read t rows further on -- the same trick as the head kernel's 5-tap convolution;
//   * the four waves split the TAPS (32 each, all 128 tokens, both row blocks of the 48 -> 64 padded features), the
//     weights go from a host-packed fragment image [group][wave][tap][ks 3][rb 2] straight into registers (four taps
//     in flight, hand-counted vmcnt), and the partial sums meet in LDS: wave w finishes token block w.
// 16-bit precisions; the fp32 mode keeps linear_kernel<EPI_GENERAL>.
#include "ppg_layer32.h"
#include <map>
#include <mutex>

namespace {

constexpr int PC_TAPS = 128, PC_CG = 48, PC_TOK = 128, PC_ROWB = 112;
constexpr int PC_WIN_ROWS = PC_TOK + PC_TAPS;                 // 256 window rows: tokens tt0 - 64 .. tt0 + 191
constexpr int PC_WIN_BYTES = PC_WIN_ROWS * PC_ROWB;           // 28 KiB
constexpr int PC_RED_BYTES = 4 * 3 * 8192;                    // per destination block: three waves' partial sums
constexpr int PC_LDS = PC_WIN_BYTES + PC_RED_BYTES;
constexpr int PC_AHEAD = 4;                                   // taps of weights in flight

template <class P>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void posconv_kernel(PosConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tok = lane & 31, hh = lane >> 5;
    const int item = blockIdx.x / a.tiles_per_item, tile = blockIdx.x - item * a.tiles_per_item;
    const int g = blockIdx.y;
    const int tt0 = PC_TOK * tile;
    const long row0 = (long)item * a.rows_per_item;           // the item's first row
    const int frames = a.frames;
    const uint32_t lds0 = lds_addr32(smem);
    const uint32_t voff = lane * 16;

    // weights of this wave's first taps (asm loads: counted by hand below)
    const char* wimg = a.w_img + ((size_t)(g * 4 + wave) * 32) * 6 * 1024;
    u32x4 wf[PC_AHEAD][6];
    auto load_tap = [&]<int SET>(int tl) {
        const char* base = wimg + (size_t)tl * 6 * 1024;
        [&]<int... K>(std::integer_sequence<int, K...>) { (gload_frag<K>(wf[SET][K], voff, base), ...); }(std::make_integer_sequence<int, 6>{});
    };
    load_tap.template operator()<0>(0);
    load_tap.template operator()<1>(1);
    load_tap.template operator()<2>(2);
    load_tap.template operator()<3>(3);

    // the window: 256 rows x 6 pieces of 16 bytes, rows outside the item's frames are zeros
    {
        u32x4 piece[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int p = tid + 256 * i, r = p / 6, c = p - 6 * r;
            const int tt = tt0 - PC_TAPS / 2 + r;
            const long m = row0 + min(max(tt, 0), frames - 1);
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(piece[i]) : "v"(a.x16 + (size_t)m * a.ldx_bytes + (size_t)(g * PC_CG + 8 * c) * 2) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            asm volatile("" : "+v"(piece[i]));
            const int p = tid + 256 * i, r = p / 6, c = p - 6 * r;
            const int tt = tt0 - PC_TAPS / 2 + r;
            const bool ok = tt >= 0 && tt < frames;
            const u32x4 v = u32x4{ok ? piece[i].x : 0u, ok ? piece[i].y : 0u, ok ? piece[i].z : 0u, ok ? piece[i].w : 0u};
            asm volatile("ds_write_b128 %0, %1" :: "v"(lds0 + (uint32_t)(r * PC_ROWB + c * 16)), "v"(v) : "memory");
        }
#pragma unroll
        for (int s = 0; s < PC_AHEAD; ++s)
#pragma unroll
            for (int k = 0; k < 6; ++k) asm volatile("" : "+v"(wf[s][k]));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
    }

    f32x16 acc[2][4];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[rb][t][i] = 0.f;

    // step i of a tap = (ks = i / 4, tb = i % 4): window row 32 tb + tok + tap, channels 16 ks + 8 hh .. + 7.
    // The ring of 6 fragment reads runs on across the taps (12 steps per tap).
    constexpr int RD = 6;
    u32x4 ring[RD];
    uint32_t rbase = lds0 + (uint32_t)((tok + 32 * wave) * PC_ROWB + hh * 16);     // this wave's first tap
    auto frag_off = [](int i) constexpr { return (i % 12 % 4) * 32 * PC_ROWB + (i % 12 / 4) * 32 + (i / 12) * PC_ROWB; };
    [&]<int... I>(std::integer_sequence<int, I...>) { (ds_read128<frag_off(I)>(ring[I], rbase), ...); }(std::make_integer_sequence<int, RD>{});
    // four taps per iteration (the register sets rotate); tap t's weights are the oldest 6 of the 24 in flight
    for (int tl = 0; tl < 32; tl += PC_AHEAD) {
        auto tap = [&]<int S>() {
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(6 * (PC_AHEAD - 1)) : "memory");
#pragma unroll
            for (int k = 0; k < 6; ++k) asm volatile("" : "+v"(wf[S][k]));
            __builtin_amdgcn_sched_barrier(0);
            auto step = [&]<int J>() {
                constexpr int I = 12 * S + J, ks = J / 4, tb = J % 4;
                u32x4& bf = ring[I % RD];
                lgkm_wait32<RD - 1>(bf);
                acc[0][tb] = P::mma32(wf[S][2 * ks], bf, acc[0][tb]);
                acc[1][tb] = P::mma32(wf[S][2 * ks + 1], bf, acc[1][tb]);
                // steps 48 .. of this iteration are the next iteration's first ones: four rows further on
                if constexpr (I + RD < 48) ds_read128<frag_off(I + RD)>(bf, rbase);
                else ds_read128<frag_off(I + RD - 48) + 4 * PC_ROWB>(bf, rbase);
            };
            [&]<int... J>(std::integer_sequence<int, J...>) { (step.template operator()<J>(), ...); }(std::make_integer_sequence<int, 12>{});
            // this set's registers are free: the tap four further on (past the end: the last tap again, unused)
            load_tap.template operator()<S>(min(tl + S + PC_AHEAD, 31));
        };
        tap.template operator()<0>();
        tap.template operator()<1>();
        tap.template operator()<2>();
        tap.template operator()<3>();
        rbase += 4 * PC_ROWB;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < RD; ++i) asm volatile("" : "+v"(ring[i]));
#pragma unroll
    for (int s = 0; s < PC_AHEAD; ++s)
#pragma unroll
        for (int k = 0; k < 6; ++k) asm volatile("" : "+v"(wf[s][k]));
    __builtin_amdgcn_sched_barrier(0);

    // ---- the partial sums of the other waves' token blocks go to LDS, lane-linear: [destination block][source][rb][q]
    const uint32_t red0 = lds0 + PC_WIN_BYTES + lane * 16;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        if (d == wave) continue;
        const uint32_t at = red0 + (uint32_t)((d * 3 + (wave < d ? wave : wave - 1)) * 8192);
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const u32x4 v = {__float_as_uint(acc[rb][d][4 * q + 0]), __float_as_uint(acc[rb][d][4 * q + 1]),
                                 __float_as_uint(acc[rb][d][4 * q + 2]), __float_as_uint(acc[rb][d][4 * q + 3])};
                asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(at), "v"(v), "n"((rb * 4 + q) * 1024) : "memory");
            }
    }
    // residual and bias of this wave's block: the lane's 16 consecutive features of its token, per row block (the
    // second row block holds features 32 .. 47 in its lanes hh = 0; hh = 1 is padding).  Requested before any store.
    const int tt = tt0 + 32 * wave + tok;
    const bool inside = tt < a.rows_per_item && row0 + tt < a.M;
    const long m = row0 + (inside ? tt : 0);
    float4 res[2][4], bias[2][4];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        const int n = g * PC_CG + (rb == 1 && hh == 1 ? 0 : 32 * rb + 16 * hh);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            res[rb][q] = *reinterpret_cast<const float4*>(a.residual + (size_t)m * a.H + n + 4 * q);
            bias[rb][q] = *reinterpret_cast<const float4*>(a.bias + n + 4 * q);
        }
    }
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            asm volatile("" : "+v"(res[rb][q].x), "+v"(res[rb][q].y), "+v"(res[rb][q].z), "+v"(res[rb][q].w));
            asm volatile("" : "+v"(bias[rb][q].x), "+v"(bias[rb][q].y), "+v"(bias[rb][q].z), "+v"(bias[rb][q].w));
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    float y[2][16];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int i = 0; i < 16; ++i) y[rb][i] = 0.f;
#pragma unroll
    for (int d = 0; d < 4; ++d) {                              // (compile-time register index: select this wave's block)
        if (d != wave) continue;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int i = 0; i < 16; ++i) y[rb][i] = acc[rb][d][i];
    }
    {
        u32x4 part[3][8];
        const uint32_t at = red0 + (uint32_t)(wave * 3 * 8192);
        [&]<int... I>(std::integer_sequence<int, I...>) { (ds_read128<(I / 8) * 8192 + (I % 8) * 1024>(part[I / 8][I % 8], at), ...); }(std::make_integer_sequence<int, 24>{});
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                asm volatile("" : "+v"(part[s][k]));
                y[k / 4][4 * (k % 4) + 0] += __uint_as_float(part[s][k][0]); y[k / 4][4 * (k % 4) + 1] += __uint_as_float(part[s][k][1]);
                y[k / 4][4 * (k % 4) + 2] += __uint_as_float(part[s][k][2]); y[k / 4][4 * (k % 4) + 3] += __uint_as_float(part[s][k][3]);
            }
    }
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x2 lo = gelu_erf_pair(f32x2{y[rb][4 * q + 0] + bias[rb][q].x, y[rb][4 * q + 1] + bias[rb][q].y});
            const f32x2 hi = gelu_erf_pair(f32x2{y[rb][4 * q + 2] + bias[rb][q].z, y[rb][4 * q + 3] + bias[rb][q].w});
            y[rb][4 * q + 0] = lo.x + res[rb][q].x; y[rb][4 * q + 1] = lo.y + res[rb][q].y;
            y[rb][4 * q + 2] = hi.x + res[rb][q].z; y[rb][4 * q + 3] = hi.y + res[rb][q].w;
        }
    }
    if (inside) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            if (rb == 1 && hh == 1) continue;
            float* dst = a.out32 + (size_t)m * a.H + g * PC_CG + 32 * rb + 16 * hh;
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(dst + 4 * q) = make_float4(y[rb][4 * q + 0], y[rb][4 * q + 1], y[rb][4 * q + 2], y[rb][4 * q + 3]);
        }
    }
}

}  // namespace

namespace ppg {

hipError_t launch_posconv(int precision, const PosConvArgs& a, int batch, hipStream_t s) {
    if (a.H != 16 * PC_CG || a.rows_per_item <= 0 || a.frames <= 0 || a.frames > a.rows_per_item || batch <= 0) return hipErrorInvalidValue;
    auto launch = [&](auto kern) {
        static std::mutex mu;
        static std::map<const void*, ppg::LdsLimit> limits;
        hipError_t e;
        { std::lock_guard<std::mutex> lock(mu); e = limits[reinterpret_cast<const void*>(kern)].ensure(reinterpret_cast<const void*>(kern), PC_LDS); }
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(batch * a.tiles_per_item, 16), dim3(256), PC_LDS, s, a);
        return hipGetLastError();
    };
    if (precision == PPG_PRECISION_BF16) return launch(posconv_kernel<PrecBF16>);
#ifndef PPG_ONLY_BF16
    if (precision == PPG_PRECISION_FP16) return launch(posconv_kernel<PrecF16>);
#endif
    return hipErrorInvalidValue;
}

}  // namespace ppg
