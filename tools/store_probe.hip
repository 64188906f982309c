// Store-path microbenchmark: how fast can a CU retire the store patterns the
// projection epilogues use?   hipcc --offload-arch=gfx950 -O3 tools/store_probe.hip -o /tmp/store_probe
// Output tile per wave: 32 token rows x 512 B (256 bf16 features) of a
// [M][1536 B] row-major buffer, like one pass of the Q/K/V projection.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

template <int PATTERN>
__global__ __launch_bounds__(256, 2) void probe(char* out, int M, int ld, int reps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int idx = lane & 15, g = lane >> 4;
    const int tok0 = (blockIdx.x * 4 + wave) * 32;
    char* base = out + (size_t)tok0 * ld + blockIdx.y * 512;
    const uint32_t v = threadIdx.x + reps;
    for (int r = 0; r < reps; ++r) {
        if constexpr (PATTERN == 0) {          // accumulator layout, 8 B per lane: 16 rows x 32 B per instruction
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int nb = 0; nb < 16; ++nb)
                    *reinterpret_cast<u32x2*>(base + (size_t)(16 * t + idx) * ld + (nb * 16 + 4 * g) * 2) = u32x2{v, v};
        } else if constexpr (PATTERN == 1) {   // full rows: 2 rows x 512 B per instruction
#pragma unroll
            for (int j = 0; j < 16; ++j)
                *reinterpret_cast<u32x4*>(base + (size_t)(2 * j + (lane >> 5)) * ld + (lane & 31) * 16) = u32x4{v, v, v, v};
        } else if constexpr (PATTERN == 2) {   // accumulator layout, 16 B per lane: 16 rows x 64 B per instruction
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int nb = 0; nb < 8; ++nb)
                    *reinterpret_cast<u32x4*>(base + (size_t)(16 * t + idx) * ld + nb * 64 + g * 16) = u32x4{v, v, v, v};
        } else if constexpr (PATTERN == 3) {   // 4 rows x 256 B per instruction
#pragma unroll
            for (int j = 0; j < 16; ++j)
                *reinterpret_cast<u32x4*>(base + (size_t)(4 * (j >> 1) + (lane >> 4)) * ld + (j & 1) * 256 + (lane & 15) * 16) = u32x4{v, v, v, v};
        } else if constexpr (PATTERN == 5) {   // 32x32 accumulator layout (ppg_layer32.h qkv_tail): 32 rows x 2 pieces of 16 B, 32 B apart
#pragma unroll
            for (int fb = 0; fb < 8; ++fb)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
                    *reinterpret_cast<u32x4*>(base + (size_t)(lane & 31) * ld + fb * 64 + (lane >> 5) * 32 + s2 * 16) = u32x4{v, v, v, v};
        } else if constexpr (PATTERN == 6) {   // the same data after a v_permlane16_swap per dword: 16 rows x 64 B per instruction
#pragma unroll
            for (int fb = 0; fb < 8; ++fb)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    u32x4 w = u32x4{v, v, v, v};
#pragma unroll
                    for (int d = 0; d < 4; ++d) { auto r = __builtin_amdgcn_permlane16_swap(w[d], w[d] + 1, false, false); w[d] = r[s2]; }
                    *reinterpret_cast<u32x4*>(base + (size_t)(16 * s2 + (lane & 15)) * ld + fb * 64 + (lane >> 4) * 16) = w;
                }
        } else if constexpr (PATTERN == 7) {   // V^T of the 32x32 layout: 32 rows (one per lane & 31), 32 B per row and instruction, rows far apart
#pragma unroll
            for (int fb = 0; fb < 8; ++fb)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
                    *reinterpret_cast<u32x4*>(out + ((size_t)(blockIdx.y * 256 + fb * 32 + (lane & 31)) * (size_t)(M + 64) + tok0) * 2 + s2 * 32 + (lane >> 5) * 16) = u32x4{v, v, v, v};
        } else if constexpr (PATTERN == 4) {   // 8 rows x 128 B per instruction
#pragma unroll
            for (int j = 0; j < 16; ++j)
                *reinterpret_cast<u32x4*>(base + (size_t)(8 * (j >> 2) + (lane >> 3)) * ld + (j & 3) * 128 + (lane & 7) * 16) = u32x4{v, v, v, v};
        }
    }
}

template <int PATTERN>
void run(const char* name, char* out, int M, int ld, int passes) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    dim3 grid(M / 128, passes);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(probe<PATTERN>, grid, dim3(256), 0, 0, out, M, ld, 1);
    hipEventRecord(a);
    const int iters = 20;
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(probe<PATTERN>, grid, dim3(256), 0, 0, out, M, ld, 1);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double us = ms * 1e3 / iters, mb = (double)M * 512 * passes / 1e6;
    printf("%-44s %7.1f us  %6.2f TB/s\n", name, us, mb / us);
}

// one workgroup per CU (100 KiB of LDS each), 4 waves, like the fused layer kernel's tail
template <int PATTERN>
void run_sparse(const char* name, char* out, int M, int ld, int passes) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    dim3 grid(M / 128, passes);
    auto kern = probe<PATTERN>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 102400);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, grid, dim3(256), 102400, 0, out, M, ld, 1);
    hipEventRecord(a);
    const int iters = 20;
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, grid, dim3(256), 102400, 0, out, M, ld, 1);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double us = ms * 1e3 / iters, mb = (double)M * 512 * passes / 1e6;
    printf("%-44s %7.1f us  %6.2f TB/s\n", name, us, mb / us);
}

int main() {
    const int M = 40960, ld = 1536;
    char* out;
    hipMalloc(&out, (size_t)(M + 64) * ld);
    printf("M = %d rows x %d B, 3 passes of 512 B (62.9 MB), 4 waves x 32 rows per workgroup, 2 workgroups per CU\n", M, ld);
    run<0>("acc layout  8 B/lane (16 rows x 32 B)", out, M, ld, 3);
    run<2>("acc layout 16 B/lane (16 rows x 64 B)", out, M, ld, 3);
    run<4>("8 rows x 128 B per instruction", out, M, ld, 3);
    run<3>("4 rows x 256 B per instruction", out, M, ld, 3);
    run<1>("2 rows x 512 B per instruction", out, M, ld, 3);
    printf("one workgroup (4 waves) per CU:\n");
    run_sparse<0>("acc layout  8 B/lane (16 rows x 32 B)", out, M, ld, 3);
    run_sparse<2>("acc layout 16 B/lane (16 rows x 64 B)", out, M, ld, 3);
    run_sparse<4>("8 rows x 128 B per instruction", out, M, ld, 3);
    run_sparse<1>("2 rows x 512 B per instruction", out, M, ld, 3);
    run_sparse<5>("32x32 acc layout (32 rows x 2 x 16 B)", out, M, ld, 3);
    run_sparse<6>("... after permlane16_swap (16 rows x 64 B)", out, M, ld, 3);
    run_sparse<7>("V^T rows of the 32x32 layout (32 rows x 32 B)", out, M, ld, 3);
    run_sparse<5>("32x32 acc layout (32 rows x 2 x 16 B)", out, M, ld, 3);
    run_sparse<6>("... after permlane16_swap (16 rows x 64 B)", out, M, ld, 3);
    hipFree(out);
    return 0;
}
