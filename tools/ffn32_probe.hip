// Probe of the feature-split FFN core with v_mfma_f32_32x32x16_bf16 (design study for
// the layer kernel):  y[256][160] = W2 relu(W1 x1 + b1)  for one 160-token workgroup.
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 tools/ffn32_probe.hip -o tools/bin/ffn32_probe
// All four waves (one per SIMD) work on ALL 160 tokens (5 blocks of 32); wave w owns
// hidden rows 32w..32w+31 of every 128-hidden chunk in phase A and output features
// 64w..64w+63 in phase B.  Weights never touch LDS: every wave loads only its own
// rows as ready-made A fragments (global_load_dwordx4, 1 KiB each, host-packed in
// consumption order).  Activations are the B operands, read from LDS: x1 (80 KiB,
// resident) and the chunk's h (40 KiB, written by the four waves, one barrier pair).
// Prints cycles per 128-hidden chunk and checks the numbers against the host.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <cstring>
#include <vector>
#include <utility>
#include <type_traits>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) short s16x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

constexpr int H = 256, F = 2048, TOK = 160, TB = 5, HC = 128, NCH = F / HC;

__device__ __forceinline__ uint32_t lds_addr(const void* p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}
__device__ __forceinline__ uint32_t relu2(uint32_t p) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, p), s16x2{0, 0}));
}
template <int OFF>
__device__ __forceinline__ void ds_read128(u32x4& dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int OFF>
__device__ __forceinline__ void ds_write128(uint32_t addr, const u32x4& v) {
    asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(addr), "v"(v), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void lgkm_wait(u32x4& r) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(r) : "n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
template <int OFF>
__device__ __forceinline__ void gload128(u32x4& dst, uint32_t voff, const char* sbase) {
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
template <int N, int COUNT>
__device__ __forceinline__ void vm_wait(u32x4 (&r)[COUNT]) {
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
#pragma unroll
    for (int i = 0; i < COUNT; ++i) asm volatile("" : "+v"(r[i]));
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ f32x16 mfma32(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// Stream of N LDS fragments (byte offset OFFS(i) from one of two bases 64 KiB apart), D reads in flight
template <class OFFS, int I, int N, int D, class USE>
__device__ __forceinline__ void stream_step(u32x4 (&ring)[D], uint32_t b0, uint32_t b1, USE& use) {
    if constexpr (I < N) {
        lgkm_wait<(N - 1 - I < D - 1) ? (N - 1 - I) : (D - 1)>(ring[I % D]);
        use(std::integral_constant<int, I>{}, ring[I % D]);
        if constexpr (I + D < N) {
            constexpr int off = OFFS::at(I + D);
            if constexpr (off < 65536) ds_read128<off>(ring[I % D], b0); else ds_read128<off - 65536>(ring[I % D], b1);
        }
        stream_step<OFFS, I + 1, N, D>(ring, b0, b1, use);
    }
}
template <class OFFS, int I, int N, int D>
__device__ __forceinline__ void stream_prime(u32x4 (&ring)[D], uint32_t b0, uint32_t b1) {
    if constexpr (I < D && I < N) {
        constexpr int off = OFFS::at(I);
        if constexpr (off < 65536) ds_read128<off>(ring[I], b0); else ds_read128<off - 65536>(ring[I], b1);
        stream_prime<OFFS, I + 1, N, D>(ring, b0, b1);
    }
}
template <class OFFS, int N, int D, class USE>
__device__ __forceinline__ void stream(uint32_t b0, uint32_t b1, USE use) {
    u32x4 ring[D];
    stream_prime<OFFS, 0, N, D>(ring, b0, b1);
    stream_step<OFFS, 0, N, D>(ring, b0, b1, use);
}
// phase A: step i = ks * 5 + tb reads x1 fragment tb * 16 + ks;  phase B: step i = ks * 5 + tb reads h fragment tb * 8 + ks
struct OffA { static constexpr int at(int i) { return ((i % TB) * 16 + i / TB) * 1024; } };
struct OffB { static constexpr int at(int i) { return ((i % TB) * 8 + i / TB) * 1024; } };

// LDS map: x1 fragments [5][16] KiB at 0, h fragments [5][8] KiB at 80 KiB, b1 at 120 KiB (8 KiB)
constexpr int LDS_X1 = 0, LDS_H = 81920, LDS_B1 = 122880, LDS_BYTES = 131072;

template <int LOADS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void ffn32(const char* w1img, const char* w2img, const float* b1, const char* x1img, float* y, unsigned long long* cycles, int reps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tok = lane & 31, hh = lane >> 5;
    // stage x1 (already in fragment order) and b1
    for (int i = tid; i < TB * 16 * 64; i += 256)
        reinterpret_cast<u32x4*>(smem + LDS_X1)[i] = reinterpret_cast<const u32x4*>(x1img + (size_t)blockIdx.x * TB * 16 * 1024)[i];
    for (int i = tid; i < F / 4; i += 256) reinterpret_cast<f32x4*>(smem + LDS_B1)[i] = reinterpret_cast<const f32x4*>(b1)[i];
    __syncthreads();

    const uint32_t lds0 = lds_addr(smem);
    const uint32_t xb0 = lds0 + LDS_X1 + lane * 16, xb1 = xb0 + 65536;
    const uint32_t hb0 = lds0 + LDS_H + lane * 16;
    const uint32_t voff = lane * 16;

    f32x16 yacc[2][TB];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int t = 0; t < TB; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) yacc[rb][t][i] = 0.f;

    u32x4 w1f[16], w2f[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { w1f[i] = u32x4{0x3c003c00u + lane, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}; w2f[i] = w1f[i]; }
    auto load_w1 = [&](int c) {     // this wave's 16 KiB of chunk c: fragments ks = 0..15
        const char* base = w1img + ((size_t)c * 4 + wave) * 16384;
        [&]<int... K>(std::integer_sequence<int, K...>) {
            ((K < 4 ? gload128<K * 1024>(w1f[K], voff, base) : gload128<(K % 4) * 1024>(w1f[K], voff, base + (K / 4) * 4096)), ...);
        }(std::make_integer_sequence<int, 16>{});
    };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < reps; ++rep) {
        if constexpr (LOADS) load_w1(0);
        for (int c = 0; c < NCH; ++c) {
            // ---- phase A: hacc[tb] = b1 + W1[rows of this wave] x1 -------------------------------
            f32x16 bias;
            {
                u32x4 braw[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) ds_read128<0>(braw[q], lds0 + LDS_B1 + (c * HC + 32 * wave + 8 * q + 4 * hh) * 4);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    asm volatile("" : "+v"(braw[q]));
                    bias[4 * q + 0] = __uint_as_float(braw[q].x); bias[4 * q + 1] = __uint_as_float(braw[q].y);
                    bias[4 * q + 2] = __uint_as_float(braw[q].z); bias[4 * q + 3] = __uint_as_float(braw[q].w);
                }
            }
            if constexpr (LOADS) vm_wait<0>(w1f);
            const char* w2base = w2img + ((size_t)c * 4 + wave) * 16384;
            f32x16 hacc[TB];
            stream<OffA, 16 * TB, 6>(xb0, xb1, [&](auto ic, const u32x4& bf) {
                constexpr int i = decltype(ic)::value;
                constexpr int ks = i / TB, tb = i % TB;
                if constexpr (ks == 0) hacc[tb] = mfma32(w1f[0], bf, bias);
                else hacc[tb] = mfma32(w1f[ks], bf, hacc[tb]);
                // this chunk's W2 fragments: one load per 5 MFMAs
                if constexpr (LOADS && tb == 2) gload128<(ks % 4) * 1024>(w2f[ks], voff, w2base + (ks / 4) * 4096);
            });
            __syncthreads();                 // every wave is done reading the previous chunk's h
            // ---- ReLU + pack: the accumulator IS the next GEMM's B fragment layout -------------
#pragma unroll
            for (int t = 0; t < TB; ++t)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const u32x4 frag = u32x4{relu2(pack_bf16x2(hacc[t][8 * s + 0], hacc[t][8 * s + 1])), relu2(pack_bf16x2(hacc[t][8 * s + 2], hacc[t][8 * s + 3])),
                                             relu2(pack_bf16x2(hacc[t][8 * s + 4], hacc[t][8 * s + 5])), relu2(pack_bf16x2(hacc[t][8 * s + 6], hacc[t][8 * s + 7]))};
                    asm volatile("ds_write_b128 %0, %1" :: "v"(hb0 + (t * 8 + 2 * wave + s) * 1024), "v"(frag) : "memory");
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();
            // ---- phase B: y[rows of this wave] += W2 h ------------------------------------------
            if constexpr (LOADS) vm_wait<0>(w2f);
            const int cn = c + 1 < NCH ? c + 1 : c;
            const char* w1base = w1img + ((size_t)cn * 4 + wave) * 16384;
            stream<OffB, 8 * TB, 6>(hb0, hb0, [&](auto ic, const u32x4& bf) {
                constexpr int i = decltype(ic)::value;
                constexpr int ks = i / TB, tb = i % TB;
                yacc[0][tb] = mfma32(w2f[ks], bf, yacc[0][tb]);
                yacc[1][tb] = mfma32(w2f[8 + ks], bf, yacc[1][tb]);
                // the next chunk's W1 fragments: two loads per 5 steps (10 MFMAs)
                if constexpr (LOADS && (tb == 1 || tb == 3)) {
                    constexpr int k = 2 * ks + (tb == 3);
                    gload128<(k % 4) * 1024>(w1f[k], voff, w1base + (k / 4) * 4096);
                }
            });
        }
        if constexpr (LOADS) vm_wait<0>(w1f);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
    // y[feature][token] natural order: tile row 8q + 4hh + r of block rb
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int t = 0; t < TB; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int f = 64 * wave + 32 * rb + 8 * (i >> 2) + 4 * hh + (i & 3);
                y[((size_t)blockIdx.x * H + f) * TOK + 32 * t + tok] = yacc[rb][t][i];
            }
}

static uint16_t bf16(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float f32(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main() {
    const int WG = 256;
    std::vector<float> W1((size_t)F * H), W2((size_t)H * F), B1(F), X((size_t)TOK * H);
    uint32_t seed = 12345u;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return ((seed >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (auto& v : W1) v = f32(bf16(rnd() * 0.25f));
    for (auto& v : W2) v = f32(bf16(rnd() * 0.1f));
    for (auto& v : B1) v = rnd() * 0.5f;
    for (auto& v : X) v = f32(bf16(rnd() * 2.0f));
    // k-slot 8 hh + 4 e + r of a 16-wide K step <-> element 16 ks + 8 e + 4 hh + r (what the accumulator layout hands over)
    auto kelem = [](int ks, int hh, int j) { return 16 * ks + 8 * (j >> 2) + 4 * hh + (j & 3); };
    std::vector<uint16_t> w1img((size_t)NCH * 4 * 16 * 64 * 8), w2img((size_t)NCH * 4 * 16 * 64 * 8), x1img((size_t)TB * 16 * 64 * 8);
    for (int c = 0; c < NCH; ++c) for (int w = 0; w < 4; ++w) for (int ks = 0; ks < 16; ++ks) for (int l = 0; l < 64; ++l) for (int j = 0; j < 8; ++j) {
        const int row = c * HC + 32 * w + (l & 31);
        w1img[((((size_t)c * 4 + w) * 16 + ks) * 64 + l) * 8 + j] = bf16(W1[(size_t)row * H + kelem(ks, l >> 5, j)]);
    }
    for (int c = 0; c < NCH; ++c) for (int w = 0; w < 4; ++w) for (int rb = 0; rb < 2; ++rb) for (int ks = 0; ks < 8; ++ks) for (int l = 0; l < 64; ++l) for (int j = 0; j < 8; ++j) {
        const int row = 64 * w + 32 * rb + (l & 31);
        w2img[((((size_t)c * 4 + w) * 16 + rb * 8 + ks) * 64 + l) * 8 + j] = bf16(W2[(size_t)row * F + c * HC + kelem(ks, l >> 5, j)]);
    }
    for (int tb = 0; tb < TB; ++tb) for (int ks = 0; ks < 16; ++ks) for (int l = 0; l < 64; ++l) for (int j = 0; j < 8; ++j)
        x1img[(((size_t)tb * 16 + ks) * 64 + l) * 8 + j] = bf16(X[(size_t)(32 * tb + (l & 31)) * H + kelem(ks, l >> 5, j)]);
    // host reference for a few tokens
    std::vector<float> ref((size_t)H * TOK, 0.f);
    for (int t = 0; t < TOK; t += 7) {
        std::vector<float> h(F);
        for (int n = 0; n < F; ++n) {
            float a = B1[n];
            for (int k = 0; k < H; ++k) a += W1[(size_t)n * H + k] * X[(size_t)t * H + k];
            h[n] = f32(bf16(a > 0.f ? a : 0.f));
        }
        for (int f = 0; f < H; ++f) {
            float a = 0.f;
            for (int n = 0; n < F; ++n) a += W2[(size_t)f * F + n] * h[n];
            ref[(size_t)f * TOK + t] = a;
        }
    }
    char *dw1, *dw2, *dx; float *db1, *dy; unsigned long long* dc;
    hipMalloc(&dw1, w1img.size() * 2); hipMalloc(&dw2, w2img.size() * 2); hipMalloc(&dx, x1img.size() * 2 * WG);
    hipMalloc(&db1, F * 4); hipMalloc(&dy, (size_t)WG * H * TOK * 4); hipMalloc(&dc, WG * 8);
    hipMemcpy(dw1, w1img.data(), w1img.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dw2, w2img.data(), w2img.size() * 2, hipMemcpyHostToDevice);
    for (int b = 0; b < WG; ++b) hipMemcpy(dx + (size_t)b * x1img.size() * 2, x1img.data(), x1img.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(db1, B1.data(), F * 4, hipMemcpyHostToDevice);
    auto run = [&](auto kern, const char* name) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        hipMemset(dy, 0, (size_t)WG * H * TOK * 4);
        hipLaunchKernelGGL(kern, dim3(WG), dim3(256), LDS_BYTES, 0, dw1, dw2, db1, dx, dy, dc, 1);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int reps = 4;
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(WG), dim3(256), LDS_BYTES, 0, dw1, dw2, db1, dx, dy, dc, reps);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> cyc(WG); hipMemcpy(cyc.data(), dc, WG * 8, hipMemcpyDeviceToHost);
        unsigned long long sum = 0; for (auto v : cyc) sum += v;
        hipLaunchKernelGGL(kern, dim3(WG), dim3(256), LDS_BYTES, 0, dw1, dw2, db1, dx, dy, dc, 1);
        std::vector<float> got((size_t)WG * H * TOK); hipMemcpy(got.data(), dy, got.size() * 4, hipMemcpyDeviceToHost);
        double worst = 0, scale = 0;
        for (int b : {0, 255}) for (int t = 0; t < TOK; t += 7) for (int f = 0; f < H; ++f) {
            worst = std::fmax(worst, std::fabs(got[((size_t)b * H + f) * TOK + t] - ref[(size_t)f * TOK + t]));
            scale = std::fmax(scale, std::fabs(ref[(size_t)f * TOK + t]));
        }
        const double flops = 4.0 * H * F * TOK * WG * reps;
        printf("%-34s %8.1f us/layer-pass  %6.0f cycles per 128-hidden chunk  %7.1f TFLOP/s  max|err| %.3e (|ref| <= %.2f)  %s\n", name,
               ms * 1e3 / reps, (double)sum / WG / reps / NCH, flops / (ms * 1e-3) / 1e12, worst, scale, hipGetErrorString(hipGetLastError()));
    };
    run(ffn32<1>, "weights global->VGPR, x1/h in LDS");
    run(ffn32<0>, "same without the weight loads");
    return 0;
}
