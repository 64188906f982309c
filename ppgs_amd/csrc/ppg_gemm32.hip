// Feature-split GEMM for the wide projections of the wav2vec 2.0 body (HF Wav2Vec2EncoderLayer: attention.out_proj,
// feed_forward.intermediate_dense / output_dense):
//     Y[m][n] = act_fn(sum_k X[m][k] W[n][k] + bias[n]) [+ R[m][n]]      X row-major 16-bit, W as A-fragment images
// on v_mfma_f32_32x32x16, the machinery of ppg_layer32.hip / ppg_head32.hip: one workgroup = 160 token rows x 256
// output features, wave w owns features 64 w .. + 63 (two row blocks), weights go from host-packed fragment images
// straight into registers (two sets of 16 fragments = one K chunk of 128 each, alternating: the next chunk's travel
// under this chunk's MFMAs), activations are B fragments read from a token-major LDS tile (rows of 272 bytes = 17 x
// 16: the 32 rows of a fragment read fall into 16 different 16-byte bank groups).  Three tile buffers: the rows of
// chunk c + 2 are requested (plain global loads, 32 - 40 registers per lane) at the end of chunk c's MFMA stream and
// written to LDS at the start of chunk c + 1 -- a whole stream to land in; one barrier per chunk.  linear_kernel, which this replaces for these GEMMs,
// stages the WEIGHTS through LDS by DMA (60 cycles per KiB on the issuing wave) and re-reads the activations from
// global memory per K group.
#include "ppg_layer32.h"

namespace {

constexpr int KC = 128;                // K per chunk
constexpr int KSC = KC / 16;           // K-steps per chunk
constexpr int GROW = KC * 2 + 16;      // bytes per tile row in LDS

// step i = (ks = i / GT, tb = i % GT): rows 32 tb .. of the tile, K-step ks
template <int GT>
struct OffTile { static constexpr int at(int i) { return (32 * (i % GT)) * GROW + (i / GT) * 32; } };

// GT = token blocks of 32 per workgroup (4 or 5: the launcher takes the one that fills the chip's rounds better)
template <class P, int GT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm32_kernel(Gemm32Args a) {
    constexpr int GTOK = 32 * GT, GBUF = GTOK * GROW;
    constexpr int SROWS = GTOK / 4, NST = SROWS / 4;                 // rows staged per wave, load instructions per wave and chunk
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tok = lane & 31, hh = lane >> 5;
    const int m0 = blockIdx.x * GTOK;
    const int pass = blockIdx.y;                                  // 256 output features
    const uint32_t lds0 = lds_addr32(smem);
    const uint32_t voff = lane * 16;
    const int chunks = a.K / KC;

    // activation rows of a chunk: wave w stages rows SROWS w .. + SROWS - 1, 4 rows (of 256 bytes) per load instruction
    const int srow = SROWS * wave + (lane >> 4), scol = (lane & 15) * 16;
    u32x4 stg[NST];
    auto fetch = [&](int c) {
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int m = min(m0 + srow + 4 * i, a.M - 1);            // rows past M re-read the last one (never stored)
            stg[i] = *reinterpret_cast<const u32x4*>(a.x + ((size_t)m * a.K + (size_t)c * KC) * 2 + scol);
        }
    };
    auto stash = [&](int c) {
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const uint32_t addr = lds0 + (uint32_t)((c % 3) * GBUF + (srow + 4 * i) * GROW + scol);
            asm volatile("ds_write_b128 %0, %1" :: "v"(addr), "v"(stg[i]) : "memory");
        }
    };

    // three register sets of 16 weight fragments; the third lives in the accumulation registers (an MFMA takes its
    // A operand from either file, a global load writes either): with all three in the 256 architectural registers
    // the compiler moved fragments between the files while their loads were still in flight
    u32x4 w1f[16], w2f[16], w3f[16];
    auto load16 = [&](u32x4 (&wf)[16], const char* base) {
        [&]<int... K>(std::integer_sequence<int, K...>) { (gload_frag<K>(wf[K], voff, base), ...); }(std::make_integer_sequence<int, 16>{});
    };
    // image order [pass][wave][chunk][rb][ks]: 16 fragments per (wave, chunk)
    const char* wimg = a.w_img + (((size_t)pass * 4 + wave) * chunks) * 16 * 1024;

    f32x16 acc[2][GT];
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const uint32_t rb0 = lds0 + (uint32_t)(tok * GROW + hh * 16);

    fetch(0);
    stash(0);
    if (chunks > 1) fetch(1);
    load16(w1f, wimg);
    load16(w2f, wimg + (size_t)(chunks > 1 ? 1 : 0) * 16 * 1024);
    vm_wait_all(w1f);
    vm_wait_all(w2f);
#pragma unroll
    for (int i = 0; i < NST; ++i) asm volatile("" : "+v"(stg[i]));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();

    // chunk c from tile buffer c % 3 with the fragments in `cur`.  Its first act is to put the rows of chunk c + 1
    // (requested a whole stream ago) into their buffer.  On the first steps of its stream it requests the rows of
    // chunk c + 2, then the fragments of chunk c + 2 into the register set chunk c - 1 used: both have a stream and
    // more to land -- with fragments only ONE chunk ahead (and the rows requested at the END of the stream) every
    // chunk waited out an L2 round trip: 2.6 us per 1 us of MFMAs at K = 3072.  vmcnt counts in order: at the end
    // everything but the 16 youngest loads (those fragments) has landed.
    auto chunk = [&](int c, u32x4 (&cur)[16], u32x4 (&nxt2)[16], auto first_tag, auto acc_file_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        constexpr bool TO_ACC = decltype(acc_file_tag)::value;      // nxt2 is the set kept in accumulation registers
        const bool more = c + 1 < chunks, more2 = c + 2 < chunks;
        if (more) stash(c + 1);     // (a store reads its data registers when it issues: the stream's first steps may reload them)
        const char* nbase = wimg + (size_t)(more2 ? c + 2 : c) * 16 * 1024;
        const char* xnext = a.x + (size_t)(more2 ? c + 2 : c) * KC * 2 + scol;
        const uint32_t tile = rb0 + (uint32_t)((c % 3) * GBUF);
        constexpr int STEPS = KSC * GT;
        stream<OffTile<GT>, STEPS, 6>(tile, tile, [&](auto ic, const u32x4& bf) {
            constexpr int i = decltype(ic)::value;
            constexpr int ks = i / GT, tb = i % GT;
            if constexpr (FIRST && ks == 0) {
                acc[0][tb] = P::mma32(cur[0], bf, zero);
                acc[1][tb] = P::mma32(cur[8], bf, zero);
            } else {
                acc[0][tb] = P::mma32(cur[ks], bf, acc[0][tb]);
                acc[1][tb] = P::mma32(cur[8 + ks], bf, acc[1][tb]);
            }
            static_assert(STEPS >= 16 + NST, "the row loads, then 16 fragment loads");
            if constexpr (i < NST) {
                // (asm like the fragment loads: a load the compiler sees gets a compiler-made vmcnt wait at its
                // use, counted without the asm loads in flight -- i.e. a wait for most of the fragments)
                const int m = min(m0 + srow + 4 * i, a.M - 1);
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(stg[i]) : "v"(xnext + (size_t)m * a.K * 2) : "memory");
            } else if constexpr (i < NST + 16) {
                constexpr int k = i - NST;
                if constexpr (TO_ACC)
                    gload128_acc<(k % 4) * 1024>(nxt2[k], voff, nbase + (k / 4) * 4096);
                else
                    gload_frag<k>(nxt2[k], voff, nbase);
            }
        });
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
#pragma unroll
        for (int i = 0; i < NST; ++i) asm volatile("" : "+v"(stg[i]));
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
    };
    chunk(0, w1f, w3f, std::true_type{}, std::true_type{});
    for (int c = 1; c < chunks; c += 3) {
        chunk(c, w2f, w1f, std::false_type{}, std::false_type{});
        if (c + 1 < chunks) chunk(c + 1, w3f, w2f, std::false_type{}, std::false_type{});
        if (c + 2 < chunks) chunk(c + 2, w1f, w3f, std::false_type{}, std::true_type{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the last chunks' clamped requests)

    // ---- epilogue: the lane's 16 consecutive features of its token, per (row block, token block)
    const int fbase = 256 * pass + 64 * wave;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        const int n = fbase + 32 * rb + 16 * hh;
        float4 bias[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) bias[q] = *reinterpret_cast<const float4*>(a.bias + n + 4 * q);
#pragma unroll
        for (int t = 0; t < GT; ++t) {
            const int m = m0 + 32 * t + tok;
            const bool inside = m < a.M;
            const size_t row = (size_t)(inside ? m : 0) * a.N + n;
            float y[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                y[4 * q + 0] = acc[rb][t][4 * q + 0] + bias[q].x; y[4 * q + 1] = acc[rb][t][4 * q + 1] + bias[q].y;
                y[4 * q + 2] = acc[rb][t][4 * q + 2] + bias[q].z; y[4 * q + 3] = acc[rb][t][4 * q + 3] + bias[q].w;
            }
            if (a.act_fn == 2) {
#pragma unroll
                for (int i = 0; i < 16; ++i) y[i] = gelu_erf(y[i]);
            } else if (a.act_fn == 1) {
#pragma unroll
                for (int i = 0; i < 16; ++i) y[i] = fmaxf(y[i], 0.f);
            }
            if (a.residual) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 rv = *reinterpret_cast<const float4*>(a.residual + row + 4 * q);
                    y[4 * q + 0] += rv.x; y[4 * q + 1] += rv.y; y[4 * q + 2] += rv.z; y[4 * q + 3] += rv.w;
                }
            }
            if (!inside) continue;
            if (a.win) {
                const int item = m / a.rows_per_item;
                if (m - item * a.rows_per_item >= a.win[item].valid) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) y[i] = 0.f;
                }
            }
            if (a.out32) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(a.out32 + row + 4 * q) = make_float4(y[4 * q + 0], y[4 * q + 1], y[4 * q + 2], y[4 * q + 3]);
            }
            if (a.out16) {
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    *reinterpret_cast<u32x4*>(a.out16 + row * 2 + 16 * s) = u32x4{P::pack2(y[8 * s + 0], y[8 * s + 1]), P::pack2(y[8 * s + 2], y[8 * s + 3]),
                                                                                  P::pack2(y[8 * s + 4], y[8 * s + 5]), P::pack2(y[8 * s + 6], y[8 * s + 7])};
            }
        }
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
    auto launch = [&](auto kern, int g) {
        const size_t lds = 3 * (size_t)(32 * g) * GROW;
        const dim3 grid((a.M + 32 * g - 1) / (32 * g), a.N / 256);
        static ppg::LdsLimit limit[2];
        const hipError_t e = limit[g - 4].ensure(reinterpret_cast<const void*>(kern), lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
        return hipGetLastError();
    };
    if (precision == PPG_PRECISION_BF16) return gt == 4 ? launch(gemm32_kernel<PrecBF16, 4>, 4) : launch(gemm32_kernel<PrecBF16, 5>, 5);
    if (precision == PPG_PRECISION_FP16) return gt == 4 ? launch(gemm32_kernel<PrecF16, 4>, 4) : launch(gemm32_kernel<PrecF16, 5>, 5);
    return hipErrorInvalidValue;
}

}  // namespace ppg
