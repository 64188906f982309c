// LDS-DMA issue-cost probe (one 512-register-class wave per SIMD, like the layer kernel):
//   hipcc --offload-arch=gfx950 -O3 tools/dma_probe.hip -o gpurun_out/dma_probe && gpurun_out/dma_probe
// Each workgroup (4 waves, one per CU) streams 64 KiB "chunks" of a 2 MiB weight
// blob into a 2 x 64 KiB LDS double buffer, 16 x 1 KiB global_load_lds_dwordx4
// per wave and chunk, with 160 v_mfma_f32_16x16x32_bf16 per wave and chunk in
// between (the layer kernel's ratio).  Variants of the ISSUE sequence:
//   0  per piece: s_mov m0 + s_nop + load, per-piece SGPR base (the current form)
//   1  linear image: m0 written once per 4 pieces, pieces addressed by the
//      instruction's immediate offset (0, 1024, 2048, 3072) -- the offset moves
//      the global AND the LDS address
//   2  no DMA at all (MFMA-only floor)
//   3  variant 1 with the loads spread between the MFMAs (one per 10 MFMAs)
//   4  variant 3 with the four waves out of phase: wave w issues piece j before MFMA 10 j + 2 w + 1
//      (the CU has ONE vector-memory path: four waves issuing together share its 64 B/clk)
//   5  all 16 pieces back to back, but wave w only after its first 40 w MFMAs
//   6  MFMA-only with v_mfma_f32_32x32x16_bf16 (80 per chunk = the same FLOPs)
// Also checks that variant 1 lands the bytes where variant 0 does.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <utility>
#include <type_traits>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

__device__ __forceinline__ uint32_t lds_addr(const void* p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}

template <int OFF>
__device__ __forceinline__ void glds_off(uint32_t lane_off, uint64_t base) {
    asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" :: "v"(lane_off), "s"(base), "n"(OFF) : "memory");
}
__device__ __forceinline__ void set_m0(uint32_t v) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" :: "s"(v) : "memory", "m0");
}
__device__ __forceinline__ void glds_m0(uint32_t lane_off, uint64_t base, uint32_t lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(lane_off), "s"(base), "s"(lds) : "memory", "m0");
}

template <int VARIANT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void probe(const char* w, int chunks, float* sink, unsigned long long* cycles, uint32_t* check) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane(lds_addr(smem));
    f32x4 acc[40];
#pragma unroll
    for (int i = 0; i < 40; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a = __builtin_bit_cast(bf16x8, u32x4{0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u});
    bf16x8 b = __builtin_bit_cast(bf16x8, u32x4{0x3f803f80u, 0x3f803f80u + wave, 0x3f803f80u, 0x3f803f80u});
    const uint32_t lane_off = lane * 16;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int c = 0; c < chunks; ++c) {
        // this wave's 16 pieces of chunk c: pieces 16*wave .. 16*wave+15 of the 64
        const uint64_t gbase = (uint64_t)(uintptr_t)(w + ((size_t)(c % 32) * 65536) + wave * 16384);
        const uint32_t lbase = lds0 + (c & 1) * 65536 + wave * 16384;
        auto mfmas = [&](int from, int to) {
#pragma unroll
            for (int i = from; i < to; ++i)
                acc[i % 40] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i % 40], 0, 0, 0);
        };
        if constexpr (VARIANT == 0) {
#pragma unroll
            for (int p = 0; p < 16; ++p) glds_m0(lane_off, gbase + p * 1024, lbase + p * 1024);
            mfmas(0, 160);
        } else if constexpr (VARIANT == 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                set_m0(lbase + q * 4096);
                const uint64_t gb = gbase + q * 4096;
                glds_off<0>(lane_off, gb); glds_off<1024>(lane_off, gb); glds_off<2048>(lane_off, gb); glds_off<3072>(lane_off, gb);
            }
            mfmas(0, 160);
        } else if constexpr (VARIANT == 2) {
            mfmas(0, 160);
        } else if constexpr (VARIANT == 6) {
            typedef __attribute__((ext_vector_type(16))) float f32x16;
            static f32x16 dummy;
            (void)dummy;
            f32x16* big = reinterpret_cast<f32x16*>(acc);     // 10 accumulators of 16 registers
#pragma unroll
            for (int i = 0; i < 80; ++i)
                big[i % 10] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, big[i % 10], 0, 0, 0);
        } else if constexpr (VARIANT == 4 || VARIANT == 5) {
            auto body = [&](auto wtag) {
                constexpr int W = decltype(wtag)::value;
                if constexpr (VARIANT == 4) {
                    int done = 0;
                    (void)done;
                    [&]<int... J>(std::integer_sequence<int, J...>) {
                        ([&] {
                            constexpr int from = J == 0 ? 0 : 10 * (J - 1) + 2 * W + 1;
                            constexpr int to = 10 * J + 2 * W + 1;
                            mfmas(from, to);
                            if constexpr (J % 4 == 0) set_m0(lbase + (J / 4) * 4096);
                            glds_off<(J % 4) * 1024>(lane_off, gbase + (J / 4) * 4096);
                        }(), ...);
                    }(std::make_integer_sequence<int, 16>{});
                    mfmas(10 * 15 + 2 * W + 1, 160);
                } else {
                    mfmas(0, 40 * W);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        set_m0(lbase + q * 4096);
                        const uint64_t gb = gbase + q * 4096;
                        glds_off<0>(lane_off, gb); glds_off<1024>(lane_off, gb); glds_off<2048>(lane_off, gb); glds_off<3072>(lane_off, gb);
                    }
                    mfmas(40 * W, 160);
                }
            };
            if (wave == 0) body(std::integral_constant<int, 0>{});
            else if (wave == 1) body(std::integral_constant<int, 1>{});
            else if (wave == 2) body(std::integral_constant<int, 2>{});
            else body(std::integral_constant<int, 3>{});
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                set_m0(lbase + q * 4096);
                const uint64_t gb = gbase + q * 4096;
                glds_off<0>(lane_off, gb); mfmas(q * 40, q * 40 + 10);
                glds_off<1024>(lane_off, gb); mfmas(q * 40 + 10, q * 40 + 20);
                glds_off<2048>(lane_off, gb); mfmas(q * 40 + 20, q * 40 + 30);
                glds_off<3072>(lane_off, gb); mfmas(q * 40 + 30, q * 40 + 40);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 40; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (sink) sink[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    if (check && blockIdx.x == 0) {
        // last chunk's LDS buffer: every 16-byte slot should hold the blob's bytes in order
        const int cl = chunks - 1;
        const uint32_t* l = reinterpret_cast<const uint32_t*>(smem + (cl & 1) * 65536);
        for (int i = threadIdx.x; i < 16384; i += 256) check[i] = l[i];
    }
}

template <int VARIANT>
void run(const char* name, const char* w, const std::vector<uint32_t>& host, int chunks) {
    float* sink; unsigned long long* cyc; uint32_t* check;
    hipMalloc(&sink, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8); hipMalloc(&check, 65536);
    hipMemset(check, 0, 65536);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<VARIANT>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<VARIANT>, dim3(256), dim3(256), 131072, 0, w, chunks, sink, cyc, check);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<VARIANT>, dim3(256), dim3(256), 131072, 0, w, chunks, sink, cyc, check);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(256); hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    std::vector<uint32_t> got(16384); hipMemcpy(got.data(), check, 65536, hipMemcpyDeviceToHost);
    int bad = 0;
    if (VARIANT != 2 && VARIANT != 6) {
        const size_t off = (size_t)((chunks - 1) % 32) * 16384;   // dwords
        for (int i = 0; i < 16384; ++i) bad += got[i] != host[off + i];
    }
    unsigned long long sum = 0; for (auto v : h) sum += v;
    printf("%-44s %8.1f us  %7.0f cycles/chunk (s_memtime, mean over CUs)  %d bad dwords  error=%s\n", name, ms * 1e3,
           (double)sum / 256 / chunks, bad, hipGetErrorString(hipGetLastError()));
    hipFree(sink); hipFree(cyc); hipFree(check);
}

int main() {
    const size_t bytes = 2u << 20;
    std::vector<uint32_t> host(bytes / 4);
    for (size_t i = 0; i < host.size(); ++i) host[i] = (uint32_t)(i * 2654435761u);
    char* w; hipMalloc(&w, bytes); hipMemcpy(w, host.data(), bytes, hipMemcpyHostToDevice);
    const int chunks = 64;
    run<2>("no DMA (160 MFMA + barrier per chunk)", w, host, chunks);
    run<0>("m0 + load per piece (current form)", w, host, chunks);
    run<1>("m0 per 4 pieces, immediate offsets", w, host, chunks);
    run<3>("immediate offsets, spread between MFMAs", w, host, chunks);
    run<4>("spread, waves out of phase", w, host, chunks);
    run<5>("16 back to back, wave w after 40 w MFMAs", w, host, chunks);
    run<6>("no DMA, 80 x 32x32x16 MFMA per chunk", w, host, chunks);
    return 0;
}
