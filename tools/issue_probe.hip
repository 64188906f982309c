// Do MFMA and VALU work overlap on one SIMD of gfx950?  (attention: softmax VALU vs QK^T / PV MFMAs)
//   hipcc --offload-arch=gfx950 -O3 tools/issue_probe.hip -o tools/bin/issue_probe && tools/bin/issue_probe
// One workgroup of 8 waves per CU (two waves per SIMD: waves w and w + 4 share SIMD w).  Per loop iteration a
// "matrix" wave issues 16 v_mfma_f32_16x16x32_bf16 (4 independent chains) and a "vector" wave issues NV VALU
// instructions (a mix of v_fma_f32 and v_exp_f32 like the softmax: 1 exp per 3 plain ops).
//   mode 0: all 8 waves matrix            mode 1: all 8 waves vector
//   mode 2: waves 0..3 matrix, 4..7 vector (one of each per SIMD)
//   mode 3: every wave does both, interleaved (4 MFMAs, then NV / 4 VALU, ...)
//   mode 4: every wave does both, phase separated (16 MFMAs, then NV VALU)
//   mode 5: mode 3 with v_mfma_f32_32x32x16_bf16 (8 per iteration = the same FLOPs)
//   mode 6: every wave does both, finely interleaved (1 MFMA, NV / 16 VALU, ...)
//   mode 7: mode 6 with v_mfma_f32_32x32x16_bf16 (1 MFMA, NV / 8 VALU, ...)
// and modes 3, 6, 7 again with one wave per SIMD (256-thread workgroups, twice the iterations).
// Prints cycles per iteration (s_memtime of wave 0) for each mode: if mode 2 ~ max(mode 0, mode 1) / 2 the pipes
// overlap across waves; mode 3 vs mode 4 shows whether one wave can hide VALU under its own MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

constexpr int NV = 96;       // VALU instructions per iteration of a vector wave (72 plain + 24 exp)

template <int N>
__device__ __forceinline__ void valu_block(float (&x)[8], float c) {
#pragma unroll
    for (int i = 0; i < N / 4; ++i) {
        float& v = x[i % 8];
        asm volatile("v_fma_f32 %0, %0, %1, %1\n\tv_exp_f32 %0, %0\n\tv_add_f32 %0, %0, %1\n\tv_max_f32 %0, %0, %1"
                     : "+v"(v) : "v"(c));
    }
}

template <int MODE>
__global__ __launch_bounds__(512) void probe(int iters, float* sink, unsigned long long* cycles) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x4 acc[4];
    f32x16 acc32[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc32[i][j] = 0.f;
    bf16x8 a = __builtin_bit_cast(bf16x8, u32x4{0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u});
    bf16x8 b = __builtin_bit_cast(bf16x8, u32x4{0x3f803f80u, 0x3f803f80u + wave, 0x3f803f80u, 0x3f803f80u});
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = 0.001f * (lane + i);
    const float c = 0.5f;
    const bool matrix = MODE == 0 || (MODE == 2 && wave < 4);
    const bool vector = MODE == 1 || (MODE == 2 && wave >= 4);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE <= 2) {
            if (matrix) {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i % 4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i % 4], 0, 0, 0);
            }
            if (vector) valu_block<NV>(x, c);
        } else if (MODE == 3) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
                valu_block<NV / 4>(x, c);
            }
        } else if (MODE == 4) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i % 4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i % 4], 0, 0, 0);
            valu_block<NV>(x, c);
        } else if (MODE == 5) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int i = 0; i < 2; ++i) acc32[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc32[i], 0, 0, 0);
                valu_block<NV / 4>(x, c);
            }
        } else if (MODE == 6) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                acc[i % 4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i % 4], 0, 0, 0);
                valu_block<NV / 16>(x, c);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc32[i % 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc32[i % 2], 0, 0, 0);
                valu_block<NV / 8>(x, c);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 2; ++i) s += acc32[i][0] + acc32[i][15];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    sink[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
    if (threadIdx.x == 256 && blockIdx.x == 0) cycles[1] = t1 - t0;
}

template <int MODE>
void run(int iters, float* sink, unsigned long long* cycles, int threads = 512) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(threads), 0, 0, iters, sink, cycles);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(threads), 0, 0, iters, sink, cycles);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2];
    hipMemcpy(h, cycles, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d, %d waves per SIMD: %.3f us / iteration\n", MODE, threads / 256, ms * 1e3 / iters);
}

int main() {
    float* sink; unsigned long long* cycles;
    hipMalloc(&sink, 256 * 512 * 4);
    hipMalloc(&cycles, 16);
    const int iters = 2000;
    run<0>(iters, sink, cycles);
    run<1>(iters, sink, cycles);
    run<2>(iters, sink, cycles);
    run<3>(iters, sink, cycles);
    run<4>(iters, sink, cycles);
    run<5>(iters, sink, cycles);
    run<6>(iters, sink, cycles);
    run<7>(iters, sink, cycles);
    run<0>(2 * iters, sink, cycles, 256);
    run<1>(2 * iters, sink, cycles, 256);
    run<3>(2 * iters, sink, cycles, 256);
    run<4>(2 * iters, sink, cycles, 256);
    run<6>(2 * iters, sink, cycles, 256);
    run<7>(2 * iters, sink, cycles, 256);
    return 0;
}
