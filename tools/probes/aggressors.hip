// Synthetic "aggressor" kernels for the co-residency forensics: round 2's 256-thread mel frontend
// (one 256-thread team per workgroup; the build was removed in round 6) computed a frame pair wrong about once per 300 pairs when an attention kernel's workgroup
// shares its CU.  Which INGREDIENT of an attention kernel does it?  Every kernel here is one 256-thread workgroup
// shape with a caller-chosen amount of dynamic LDS (so that it fits beside one 79.5 KiB frontend workgroup) that
// spins on a mix of instruction classes selected by a bit mask:
//    1 v_exp_f32 (transcendental unit)        2 v_mfma_f32_16x16x32_bf16       4 LDS write + read traffic
//    8 DPP cross-lane moves                   16 ds_bpermute_b32               32 v_pk_fma_f32
//   64 plain v_fma_f32                       128 global loads (L2 resident)    256 global -> LDS DMA
//  512 v_sqrt_f32 / v_rcp_f32 / v_log_f32   1024 v_fma_f64                   2048 s_setprio 3 around the loop
//  4096 v_mfma_f32_32x32x16_bf16            8192 workgroup barriers
// Built by tools/probes/build.sh into tools/bin/libaggressors.so, driven by tests/test_gpu_soak.py (and, until round 6, tools/coresidency_matrix.py).
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void aggressor_kernel(int mask, int iters, int lds_bytes, const float* __restrict__ src,
                                                        float* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    float x = 0.001f * (float)(tid + 1), y = 1.0f + 0.0001f * (float)lane;
    f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
    f32x16 acc16 = {};
    f32x2 pk = {x, y};
    double d = 1.0 + 1e-9 * tid;
    bf16x8 a8, b8;
    for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(0.01f * (float)(i + lane)); b8[i] = (__bf16)(0.02f * (float)(i + 1)); }
    const int words = lds_bytes / 4;
    float* lds = reinterpret_cast<float*>(smem);
    if (mask & 2048) __builtin_amdgcn_s_setprio(3);
    for (int it = 0; it < iters; ++it) {
        if (mask & 1) {
#pragma unroll
            for (int k = 0; k < 8; ++k) x = __builtin_amdgcn_exp2f(x) * 0.25f;
        }
        if (mask & 2) {
#pragma unroll
            for (int k = 0; k < 4; ++k) acc4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc4, 0, 0, 0);
        }
        if (mask & 4096) {
#pragma unroll
            for (int k = 0; k < 2; ++k) acc16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc16, 0, 0, 0);
        }
        if ((mask & 4) && words >= 4096) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int w = (it * 1031 + k * 257 + tid * 17) % (words - 1);
                lds[w] = x + (float)k;
                y += lds[(w + 64 * 17) % (words - 1)];
            }
        }
        if (mask & 8) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                y += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, y), 0xB1, 0xf, 0xf, true)) * 1e-3f;
        }
        if (mask & 16) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                y += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((lane ^ (k + 1)) & 63) << 2, __builtin_bit_cast(int, y))) * 1e-3f;
        }
        if (mask & 32) {
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("v_pk_fma_f32 %0, %0, %1, %1 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "+v"(pk) : "v"(f32x2{0.999f, 1.0001f}));
        }
        if (mask & 64) {
#pragma unroll
            for (int k = 0; k < 8; ++k) y = __builtin_fmaf(y, 0.9999f, 1e-4f);
        }
        if (mask & 128) {
#pragma unroll
            for (int k = 0; k < 4; ++k) y += src[(it * 4099 + k * 1024 + tid + blockIdx.x * 256) & 0xfffff] * 1e-6f;
        }
        if ((mask & 256) && lds_bytes >= 8192) {
            const uint32_t dst = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)smem + ((it & 3) * 4 + (tid >> 6)) * 256;
            const float* p = src + ((it * 8191 + tid + blockIdx.x * 256) & 0xfffff);
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off"
                         :: "v"(p), "s"(__builtin_amdgcn_readfirstlane(dst)) : "memory", "m0");
        }
        if (mask & 512) {
            x = __builtin_amdgcn_sqrtf(x * x + 1e-6f);
            x = __builtin_amdgcn_rcpf(x + 1.0f);
            x = __builtin_amdgcn_logf(x + 2.0f);
        }
        if (mask & 1024) {
#pragma unroll
            for (int k = 0; k < 4; ++k) d = __builtin_fma(d, 0.99999, 1e-5);
        }
        if (mask & 8192) __syncthreads();
    }
    if (mask & 256) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float r = x + y + acc4[0] + acc4[3] + acc16[0] + acc16[15] + pk.x + pk.y + (float)d;
    if (r == 12345.678f) sink[tid] = r;        // (never: keeps everything live)
}

// A VICTIM without LDS exchanges or memory traffic: the same arithmetic twice from the same registers, compared.
// Catches corruption of VALU results / registers by a co-resident wave.  mode bit 1: packed fp32, 2: plain fp32 + sqrt,
// 4: DPP + v_perm, 8: LDS exchange with reversed-lane reads (stride-17 layout of the frontend's buffer).
__global__ __launch_bounds__(256, 2) void victim_kernel(int mode, int iters, unsigned long long* __restrict__ counters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x2* buf = reinterpret_cast<f32x2*>(smem) + wave * (1024 + 64);
    unsigned long long bad[4] = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        const float seed = 0.37f + 1e-3f * (float)((it * 131 + tid * 7 + blockIdx.x) & 1023);
        if (mode & 1) {
            f32x2 r[2];
            for (int rep = 0; rep < 2; ++rep) {
                f32x2 v[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) v[k] = f32x2{seed + 0.01f * k, seed * 0.5f - 0.02f * k};
                asm volatile("" : "+v"(v[0]), "+v"(v[5]), "+v"(v[9]));
#pragma unroll
                for (int round = 0; round < 4; ++round)
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        f32x2 t;
                        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(v[k]), "v"(v[(k + 5) & 15]));
                        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(v[k]) : "v"(v[k]), "v"(v[(k + 5) & 15]), "v"(t));
                        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(v[k]) : "v"(v[k]), "v"(v[(k + 3) & 15]));
                        v[k] = v[k] * 0.25f;
                    }
                f32x2 s = {0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 16; ++k) s += v[k];
                r[rep] = s;
            }
            if (__builtin_bit_cast(uint32_t, r[0].x) != __builtin_bit_cast(uint32_t, r[1].x) ||
                __builtin_bit_cast(uint32_t, r[0].y) != __builtin_bit_cast(uint32_t, r[1].y)) ++bad[0];
        }
        if (mode & 2) {
            float r[2];
            for (int rep = 0; rep < 2; ++rep) {
                float v[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) v[k] = seed + 0.013f * k;
                asm volatile("" : "+v"(v[0]), "+v"(v[7]));
#pragma unroll
                for (int round = 0; round < 4; ++round)
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const float a = 0.5f * (v[k] + v[(k + 3) & 15]), b = 0.5f * (v[k] - v[(k + 7) & 15]);
                        v[k] = __builtin_amdgcn_sqrtf(a * a + b * b + 1e-6f);
                    }
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < 16; ++k) s += v[k];
                r[rep] = s;
            }
            if (__builtin_bit_cast(uint32_t, r[0]) != __builtin_bit_cast(uint32_t, r[1])) ++bad[1];
        }
        if (mode & 4) {
            const uint32_t sel = (lane & 1) ? 0x07060302u : 0x01000504u;
            uint32_t r[2];
            for (int rep = 0; rep < 2; ++rep) {
                uint32_t acc = 0;
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    uint32_t pkv = __builtin_bit_cast(uint32_t, seed + (float)k) ^ (uint32_t)(lane * 0x9e3779b9u);
                    asm volatile("" : "+v"(pkv));
                    const uint32_t other = (uint32_t)__builtin_amdgcn_mov_dpp((int)pkv, 0xB1, 0xf, 0xf, true);
                    acc += __builtin_amdgcn_perm(pkv, other, sel);
                }
                r[rep] = acc;
            }
            if (r[0] != r[1]) ++bad[2];
        }
        if (mode & 8) {
            // write 16 values per lane at pad(16 lane + k), read back index (1023 - (lane + 64 r)) : written by lane
            // (1023 - lane - 64 r) / 16 as its k = (1023 - lane - 64 r) % 16
#pragma unroll
            for (int k = 0; k < 16; ++k) buf[17 * lane + k] = f32x2{seed + (float)(16 * lane + k), (float)it};
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int idx = 1023 - (lane + 64 * r);
                const f32x2 got = buf[idx + (idx >> 4)];
                // the writer's seed: its tid differs from ours
                const int wl = idx >> 4;
                const float wseed = 0.37f + 1e-3f * (float)((it * 131 + (wave * 64 + wl) * 7 + blockIdx.x) & 1023);
                if (got.x != wseed + (float)idx || got.y != (float)it) ++bad[3];
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) if (bad[k]) atomicAdd(counters + k, bad[k]);
    if (tid == 0) atomicAdd(counters + 4, 1ull);
}

extern "C" int aggressor_launch(int mask, int grid, int iters, int lds_bytes, const float* src, float* sink, void* stream) {
    static int attr = 0;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(aggressor_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 163840) != hipSuccess) return -1;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(victim_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 163840) != hipSuccess) return -1;
        attr = 1;
    }
    hipLaunchKernelGGL(aggressor_kernel, dim3(grid), dim3(256), lds_bytes, static_cast<hipStream_t>(stream), mask, iters, lds_bytes, src, sink);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int victim_launch(int mode, int grid, int iters, int lds_bytes, unsigned long long* counters, void* stream) {
    if (aggressor_launch(0, 1, 0, 0, nullptr, nullptr, stream)) return -1;     // (attributes)
    hipLaunchKernelGGL(victim_kernel, dim3(grid), dim3(256), lds_bytes, static_cast<hipStream_t>(stream), mode, iters, counters);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
