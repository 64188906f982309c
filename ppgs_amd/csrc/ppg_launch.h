// Host-callable launchers implemented in ppg_kernels.hip / ppg_frontend.hip.
#pragma once

#include <algorithm>
#include <cstdlib>
#include <hip/hip_runtime.h>

#include "ppg_device.h"

namespace ppg {

// Kernel-selection switches of the environment.  The product library reads only the ones the parity tests drive
// (env_switch: each A/Bs a kernel against the path it replaced, tests/test_gpu_parity.py names them); tile-shape
// overrides, schedule experiments and the stamps of the timing builds exist in -DPPG_EXPERIMENT_SWITCHES builds only
// (make VARIANT=exp EXTRA=-DPPG_EXPERIMENT_SWITCHES).  INTEGRATION.md lists both sets.
inline int env_switch(const char* name, int dflt) {
    const char* s = getenv(name);
    return s ? atoi(s) : dflt;
}
#ifdef PPG_EXPERIMENT_SWITCHES
inline int env_experiment(const char* name, int dflt) { return env_switch(name, dflt); }
inline bool env_experiment_set(const char* name) { return getenv(name) != nullptr; }
#else
inline int env_experiment(const char*, int dflt) { return dflt; }
inline bool env_experiment_set(const char*) { return false; }
#endif

int attn_query_tile(int head_dim);   // queries per attention workgroup

hipError_t launch_gather(int precision, const GatherArgs& a, hipStream_t s);
hipError_t launch_fill(float* p, size_t n, float v, hipStream_t s);

hipError_t launch_linear(int precision, int epi, int nb, int nt, const LinearArgs& a, int ypasses, hipStream_t s);
// nt: 16-token blocks per wave (1..3), or kFfnMixedTiling = 160-token workgroups with waves of 3, 3, 2, 2
// blocks that balance their work through LDS (bf16, hidden 256, fused out-projection only)
constexpr int kFfnMixedTiling = 5;
hipError_t launch_ffn(int precision, const FfnArgs& a, int nt, hipStream_t s);
hipError_t launch_attn(int precision, const AttnArgs& a, int nitems, int heads, int head_dim, hipStream_t s);
// feature-split layer kernel (ppg_layer32.hip): 16-bit precisions, hidden 256, F a multiple of 128
hipError_t launch_layer32(int precision, const Layer32Args& a, hipStream_t s);
// gather + input convolution + layer 0's Q/K/V of a 160-token tile (ppg_head32.hip): 16-bit precisions, hidden 256, <= 96 input channels
hipError_t launch_head32(int precision, const Head32Args& a, hipStream_t s);
// the wide projections of the wav2vec2 body on the feature-split machinery (ppg_gemm32.hip): 16-bit precisions
hipError_t launch_gemm32(int precision, const Gemm32Args& a, hipStream_t s);
// grouped positional convolution + GELU + residual of the wav2vec2 body (ppg_posconv.hip): 16-bit precisions
hipError_t launch_posconv(int precision, const PosConvArgs& a, int batch, hipStream_t s);
hipError_t launch_ffn32x2(const Ffn32X2Args& a, hipStream_t s);
int ffn32x2_tokens();
bool ffn32x2_supported(int H, int F);     // what launch_ffn32x2 accepts: the engine builds the hi + lo images and dispatches on it
// output convolution + mask + softmax with the weights resident in LDS (ppg_outconv.hip): 16-bit precisions, hidden 256, whole batches
bool outconv_supported(int precision, const LinearArgs& a);
hipError_t launch_outconv(int precision, const LinearArgs& a, hipStream_t s);
int layer32_tokens(int hidden);       // token rows per workgroup (160 at hidden 256, 96 at hidden 512)
// wav2vec2 feature encoder, layer 0 (conv k10 s5 + GroupNorm + GELU; ppg_w2v2.hip) and the fp32 read-out of the last layer
hipError_t launch_w2v2_layer0(int precision, const float* audio, int batch, long samples, long frames, int rows_per_item,
                              const float* w0, const float* gamma, const float* beta, double* moments, float2* scale_shift,
                              char* out, hipStream_t s);
hipError_t launch_w2v2_layernorm(int precision, int H, const float* in32, const char* in16, const float* gamma, const float* beta,
                                 long rows, int T_in, int R_out, float eps, float* out32, char* out16, hipStream_t s);
hipError_t launch_w2v2_output(int precision, const char* rows, int batch, int rows_per_item, long frames, float* out, hipStream_t s);

constexpr int kFrontendFrames = 16;    // frames per frontend group (the filterbank MFMA's 16 columns)
constexpr int kMelSteps = 8;           // banded filterbank: 32-bin steps per wave of the frontend ...
constexpr int kMelSegment = 5;         // ... in two segments of 5 + 3: a filter block fills one segment or a whole wave

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device and not a stream
// operation (it must stay out of graph capture): done once per (kernel
// instantiation, device) -- again only if a later launch needs more.
struct LdsLimit {
    size_t bytes[32] = {};
    hipError_t ensure(const void* kernel, size_t lds) {
        int device = 0;
        if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= 32) device = 0;
        if (bytes[device] >= lds) return hipSuccess;
        const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) bytes[device] = lds;
        return e;
    }
};

struct FrontendTables {
    const float* hann;        // [1024]
    const float2* twiddle;    // [1024] exp(-2 pi i j / 1024)
    // Banded filterbank for v_mfma_f32_16x16x32_f16: A fragments of 16 filters x 32 bins, lane l holds
    // filter 16 block + (l & 15), bins first + 8 (l >> 4) ..+7 as fp16; fragment 2 i is the high part of
    // weight * 2^16, 2 i + 1 the low part.  mel_prog[wave][kMelSteps] = (high fragment, byte offset of the
    // step's first bin in a magnitude row, filter block the step completes or -1, 0); a block completes at
    // step kMelSegment - 1 or at the last step; unused steps point at a zero fragment pair.
    const void* mel_img;
    const int4* mel_prog;
    unsigned long long* dbg;  // PPG_FE_TIMING builds: s_memtime stamps of workgroup 0 (PPGS_AMD_FE_TIMING=1)
};

hipError_t launch_frontend(const FrontendTables& tb, const float* audio, int batch, int samples,
                           void* spec, void* mel, hipStream_t s);

}  // namespace ppg
