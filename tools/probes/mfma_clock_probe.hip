// What clock does an MI355X deliver while every matrix pipe is busy?  The dense-MFMA peak (2.5 PFLOP/s bf16) is quoted
// at 2.4 GHz; every roofline fraction of this repository divides by it.  This stand-alone program (no PyTorch, nothing
// of the package) runs v_mfma_f32_32x32x16_bf16 back to back -- four independent accumulators per wave, one wave per
// SIMD on every CU (or two) -- for `iters` x 64 MFMAs and reads two counters around the loop on every workgroup:
// s_memtime (shader-clock cycles) and s_memrealtime (100 MHz, chip-wide).  cycles / microsecond = the clock the loop
// ran at; cycles per MFMA = the issue cadence (32 = one MFMA per 8 passes); FLOPs / wall time = what "peak" is on
// this chip under its power limit.  A second mode interleaves 3 VALU instructions per MFMA (the softmax's density).
//     hipcc --offload-arch=gfx950 -O3 -std=c++20 tools/probes/mfma_clock_probe.hip -o tools/bin/mfma_clock_probe
//     tools/bin/mfma_clock_probe [iters=400]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <int VALU>
__global__ __launch_bounds__(256) void dense_mfma(int iters, unsigned long long* rec, float* sink) {
    f32x16 acc[4];
    for (int k = 0; k < 4; ++k)
        for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x ^ i)); }
    float x = 0.5f + 0.001f * threadIdx.x, y = 0.25f;
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j & 3], 0, 0, 0);
            if constexpr (VALU > 0) {
#pragma unroll
                for (int v = 0; v < VALU; ++v) { x = __builtin_fmaf(x, y, 0.001f); }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        unsigned long long* o = rec + 2 * (blockIdx.x * 4 + (threadIdx.x >> 6));
        o[0] = c1 - c0; o[1] = r1 - r0;
    }
    float s = x;
    for (int k = 0; k < 4; ++k) s += acc[k][0];
    if (s == 12345.678f) sink[0] = s;
}

template <int VALU>
void run(const char* name, int grid, int iters, unsigned long long* d_rec, float* d_sink) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {          // (the first launches run on a chip that was idle)
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(dense_mfma<VALU>, dim3(grid), dim3(256), 0, 0, iters, d_rec, d_sink);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
    }
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> rec(2 * grid * 4);
    (void)hipMemcpy(rec.data(), d_rec, rec.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> ghz, cyc;
    for (int i = 0; i < grid * 4; ++i) {
        ghz.push_back((double)rec[2 * i] / ((double)rec[2 * i + 1] * 10.0));          // cycles per ns
        cyc.push_back((double)rec[2 * i] / (64.0 * iters));
    }
    std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
    const double flops = 2.0 * 32 * 32 * 16 * 64.0 * iters * grid * 4;
    printf("%-34s grid %4d  %8.1f us  clock GHz p10/50/90 %.3f %.3f %.3f  cycles per MFMA p50 %.1f  %.0f TFLOP/s (%.3f of 2500)\n", name, grid,
           ms * 1e3, ghz[ghz.size() / 10], ghz[ghz.size() / 2], ghz[ghz.size() * 9 / 10], cyc[cyc.size() / 2], flops / (ms * 1e-3) / 1e12,
           flops / (ms * 1e-3) / 2.5e15);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 400;
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("%s: %d CUs, iters %d (x 64 MFMAs of 32x32x16 bf16 per wave)\n", prop.name, cus, iters);
    unsigned long long* d_rec; float* d_sink;
    (void)hipMalloc(&d_rec, 2 * 8 * 4 * 2 * cus * 8); (void)hipMalloc(&d_sink, 64);
    run<0>("MFMA only, 1 wave per SIMD", cus, iters, d_rec, d_sink);
    run<0>("MFMA only, 2 waves per SIMD", 2 * cus, iters, d_rec, d_sink);
    run<3>("MFMA + 3 VALU, 1 wave per SIMD", cus, iters, d_rec, d_sink);
    run<6>("MFMA + 6 VALU, 1 wave per SIMD", cus, iters, d_rec, d_sink);
    run<0>("MFMA only, half the CUs", cus / 2, iters, d_rec, d_sink);
    run<0>("MFMA only, 1 wave per SIMD, 10x longer", cus, 10 * iters, d_rec, d_sink);
    return 0;
}
