// wav2vec 2.0 feature encoder, first layer (reference
// ppgs/preprocess/w2v2fb/core.py:66 -> HF transformers Wav2Vec2FeatureEncoder,
// modeling_wav2vec2.py: Wav2Vec2GroupNormConvLayer = Conv1d(1, 512, k 10, s 5, no
// bias) -> GroupNorm(512 groups of 1 channel, eps 1e-5, affine) -> GELU):
//
//   y0[b][t][c] = gelu(a[b][c] * sum_j w[c][j] x[b][5t + j] + s[b][c])
//
// The GroupNorm statistics of channel c run over all frames of the (padded) row.
// The convolution is linear, so they follow from 65 moments of the audio alone --
// M1[j] = sum_t x[5t+j], M2[j][k] = sum_t x[5t+j] x[5t+k] -- accumulated in fp64:
// mean_c = w_c . M1 / T, E[y^2] = w_c' M2 w_c / T.  No pass over the 512-channel
// activation is needed for them, and the normalised, activated layer-0 output is
// written exactly once, token-major [batch * R0][512] in the GEMM operand type:
// what layers 1..6 (linear_kernel<EPI_GELU>: strided 3- / 2-tap convolutions as
// MFMA GEMMs) read as their B operand.
#include "ppg_device.h"
#include "ppg_launch.h"

#include <type_traits>

namespace {

constexpr int W2V_C = 512;

__device__ __forceinline__ float gelu_exact(float x) { return gelu_erf(x); }

// moments[b][0..9] = M1, [10..64] = upper triangle of M2 (j <= k, row-major)
__global__ __launch_bounds__(256) void w2v2_moments_kernel(const float* audio, long samples, long frames, double* moments) {
    const int b = blockIdx.y;
    const float* x = audio + (size_t)b * samples;
    double acc[65];
#pragma unroll
    for (int i = 0; i < 65; ++i) acc[i] = 0.0;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < frames; t += (long)gridDim.x * 256) {
        float v[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) v[j] = x[5 * t + j];
        int o = 10;
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            acc[j] += v[j];
#pragma unroll
            for (int k = j; k < 10; ++k) acc[o++] += (double)v[j] * (double)v[k];
        }
    }
    __shared__ double red[4][65];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 65; ++i) {
        double v = acc[i];
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < 65)
        atomicAdd(moments + (size_t)b * 65 + threadIdx.x,
                  (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
}

// scale_shift[b][c] = (a, s): gelu argument = a * conv + s
__global__ __launch_bounds__(256) void w2v2_norm_kernel(const double* moments, const float* w0, const float* gamma, const float* beta,
                                                        long frames, float2* scale_shift) {
    const int b = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (c >= W2V_C) return;
    const double* m = moments + (size_t)b * 65;
    double w[10];
#pragma unroll
    for (int j = 0; j < 10; ++j) w[j] = w0[c * 10 + j];
    double mean = 0.0, ey2 = 0.0;
    int o = 10;
#pragma unroll
    for (int j = 0; j < 10; ++j) {
        mean += w[j] * m[j];
#pragma unroll
        for (int k = j; k < 10; ++k) ey2 += (j == k ? 1.0 : 2.0) * w[j] * w[k] * m[o++];
    }
    mean /= (double)frames;
    ey2 /= (double)frames;
    const double var = fmax(ey2 - mean * mean, 0.0);
    const double a = (double)gamma[c] / sqrt(var + 1e-5);
    scale_shift[(size_t)b * W2V_C + c] = make_float2((float)a, (float)((double)beta[c] - mean * a));
}

// 64 frames x 512 channels per workgroup; thread = 2 channels, all 64 frames
template <class P>
__global__ __launch_bounds__(256) void w2v2_conv0_kernel(const float* audio, long samples, long frames, int rows_per_item,
                                                         const float* w0, const float2* scale_shift, char* out) {
    __shared__ float win[64 * 5 + 16];
    const int b = blockIdx.y;
    const long t0 = (long)blockIdx.x * 64;
    const float* x = audio + (size_t)b * samples;
    for (int i = threadIdx.x; i < 64 * 5 + 5; i += 256) {
        const long n = 5 * t0 + i;
        win[i] = n < samples ? x[n] : 0.f;
    }
    const int c = 2 * threadIdx.x;
    float w[2][10];
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int j = 0; j < 10; ++j) w[e][j] = w0[(c + e) * 10 + j];
    const float2 ss0 = scale_shift[(size_t)b * W2V_C + c], ss1 = scale_shift[(size_t)b * W2V_C + c + 1];
    __syncthreads();
    typename P::elem* dst = reinterpret_cast<typename P::elem*>(out) + ((size_t)b * rows_per_item + t0) * W2V_C + c;
    for (int t = 0; t < 64; ++t) {
        float y0 = 0.f, y1 = 0.f;
        if (t0 + t < frames) {
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                const float v = win[5 * t + j];
                y0 = fmaf(w[0][j], v, y0);
                y1 = fmaf(w[1][j], v, y1);
            }
            const f32x2 gy = gelu_erf_pair(f32x2{fmaf(ss0.x, y0, ss0.y), fmaf(ss1.x, y1, ss1.y)});
            y0 = gy.x; y1 = gy.y;
        }
        if (t0 + t < rows_per_item) {
            if constexpr (P::kSplit) {      // [32 hi | 32 lo] blocks: the pair's hi halves, its lo halves 64 bytes on
                char* row = out + (((size_t)b * rows_per_item + t0 + t) * W2V_C) * P::kBytes + P::row_byte(c);
                uint32_t h, l;
                P::split2(y0, y1, h, l);
                *reinterpret_cast<uint32_t*>(row) = h;
                *reinterpret_cast<uint32_t*>(row + 64) = l;
            } else if constexpr (P::kIsBF16) *reinterpret_cast<uint32_t*>(dst + (size_t)t * W2V_C) = P::pack2(y0, y1);
            else *reinterpret_cast<float2*>(dst + (size_t)t * W2V_C) = make_float2(y0, y1);
        }
    }
}

// rows t < frames of every item, as fp32 (batch, frames, 512)
template <class P>
__global__ __launch_bounds__(256) void w2v2_output_kernel(const char* rows, int rows_per_item, long frames, float* out) {
    const int b = blockIdx.y;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;        // float4 index inside the item
    if (i >= frames * (W2V_C / 4)) return;
    const long t = i / (W2V_C / 4);
    const int c = (int)(i - t * (W2V_C / 4)) * 4;
    const typename P::elem* src = reinterpret_cast<const typename P::elem*>(rows) + ((size_t)b * rows_per_item + t) * W2V_C + c;
    float4 v;
    if constexpr (P::kSplit) {
        const char* at = rows + (((size_t)b * rows_per_item + t) * W2V_C) * P::kBytes + P::row_byte(c);
        const uint2 h = *reinterpret_cast<const uint2*>(at), l = *reinterpret_cast<const uint2*>(at + 64);
        auto f = [](uint32_t w, int k) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(k ? w >> 16 : w & 0xffffu)); };
        v = make_float4(f(h.x, 0) + f(l.x, 0), f(h.x, 1) + f(l.x, 1), f(h.y, 0) + f(l.y, 0), f(h.y, 1) + f(l.y, 1));
    } else if constexpr (std::is_same_v<P, PrecF32>) {
        v = *reinterpret_cast<const float4*>(src);
    } else {
        const uint2 raw = *reinterpret_cast<const uint2*>(src);
        if constexpr (std::is_same_v<P, PrecBF16>) {
            v = make_float4(bf16_to_f32((uint16_t)raw.x), bf16_to_f32((uint16_t)(raw.x >> 16)),
                            bf16_to_f32((uint16_t)raw.y), bf16_to_f32((uint16_t)(raw.y >> 16)));
        } else {
            v = make_float4((float)__builtin_bit_cast(_Float16, (uint16_t)raw.x), (float)__builtin_bit_cast(_Float16, (uint16_t)(raw.x >> 16)),
                            (float)__builtin_bit_cast(_Float16, (uint16_t)raw.y), (float)__builtin_bit_cast(_Float16, (uint16_t)(raw.y >> 16)));
        }
    }
    *reinterpret_cast<float4*>(out + ((size_t)b * frames + t) * W2V_C + c) = v;
}

// Row LayerNorm for the transformer body (widths 512 and 768 do not fit the GEMM epilogues' register tiles):
// one wave per row.  Input fp32 [rows_in][H] (item b's rows b * T_in ..) or operand-type rows; output row
// (m / T_in) * R_out + m % T_in -- the token space pads every item to R_out rows -- as fp32 and / or operand type.
// A lane owns 4 consecutive features of every 256 (16-byte accesses).  gamma and beta are fetched with the row, ahead
// of the first store: fetched inside the store loop (behind `if (out32)`) every iteration waited with vmcnt(0) for
// the previous iteration's STORE to be acknowledged (the compiler's wait counts across control flow) -- a chain of
// 12 store round trips per row, which made this kernel latency-bound at 4.3 TB/s of its 62 MB.
template <class P, int H>
__global__ __launch_bounds__(256) void w2v2_layernorm_kernel(const float* in32, const char* in16, const float* gamma, const float* beta,
                                                             long rows, int T_in, int R_out, float eps, float* out32, char* out16) {
    constexpr int PER = H / 256;
    const int lane = threadIdx.x & 63;
    const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= rows) return;
    float4 v[PER], gv[PER], bv[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int n = i * 256 + lane * 4;
        if (in32) {
            v[i] = *reinterpret_cast<const float4*>(in32 + m * H + n);
        } else if constexpr (P::kIsBF16) {
            const uint2 raw = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(in16) + m * H + n);
            const uint16_t r[4] = {(uint16_t)(raw.x & 0xffffu), (uint16_t)(raw.x >> 16), (uint16_t)(raw.y & 0xffffu), (uint16_t)(raw.y >> 16)};
            float f[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if constexpr (std::is_same_v<P, PrecBF16>) f[k] = bf16_to_f32(r[k]);
                else f[k] = (float)__builtin_bit_cast(_Float16, r[k]);
            }
            v[i] = make_float4(f[0], f[1], f[2], f[3]);
        } else {
            v[i] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(in16) + m * H + n);
        }
        gv[i] = *reinterpret_cast<const float4*>(gamma + n);
        bv[i] = *reinterpret_cast<const float4*>(beta + n);
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
    const float mean = sum / H;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
        sq += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sq += __shfl_xor(sq, off);
    const float rstd = 1.0f / sqrtf(sq / H + eps);
    const long mo = (m / T_in) * R_out + m % T_in;
    float4 y[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        y[i] = make_float4((v[i].x - mean) * rstd * gv[i].x + bv[i].x, (v[i].y - mean) * rstd * gv[i].y + bv[i].y,
                           (v[i].z - mean) * rstd * gv[i].z + bv[i].z, (v[i].w - mean) * rstd * gv[i].w + bv[i].w);
    }
    if (out32) {
#pragma unroll
        for (int i = 0; i < PER; ++i) *reinterpret_cast<float4*>(out32 + mo * H + i * 256 + lane * 4) = y[i];
    }
    if (out16) {
#pragma unroll
        for (int i = 0; i < PER; ++i)      // (16-bit pairs, fp32, or both planes of the [32 hi | 32 lo] blocks)
            store4<P>(out16 + (size_t)mo * H * P::kBytes + P::row_byte(i * 256 + lane * 4), y[i].x, y[i].y, y[i].z, y[i].w);
    }
}

}  // namespace

namespace ppg {

hipError_t launch_w2v2_layernorm(int precision, int H, const float* in32, const char* in16, const float* gamma, const float* beta,
                                 long rows, int T_in, int R_out, float eps, float* out32, char* out16, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    const dim3 grid((unsigned)((rows + 3) / 4));
    auto go = [&](auto p_tag) {
        using P = decltype(p_tag);
        if (H == 512) hipLaunchKernelGGL((w2v2_layernorm_kernel<P, 512>), grid, dim3(256), 0, s, in32, in16, gamma, beta, rows, T_in, R_out, eps, out32, out16);
        else if (H == 768) hipLaunchKernelGGL((w2v2_layernorm_kernel<P, 768>), grid, dim3(256), 0, s, in32, in16, gamma, beta, rows, T_in, R_out, eps, out32, out16);
        else return hipErrorInvalidValue;
        return hipGetLastError();
    };
    if (precision == PPG_PRECISION_BF16) return go(PrecBF16{});
    if (precision == PPG_PRECISION_FP16) return go(PrecF16{});
    if (precision == PPG_PRECISION_FP16X2) return in16 ? hipErrorInvalidValue : go(PrecX2{});     // (fp32 rows in, split rows out)
    return go(PrecF32{});
}


hipError_t launch_w2v2_layer0(int precision, const float* audio, int batch, long samples, long frames, int rows_per_item,
                              const float* w0, const float* gamma, const float* beta, double* moments, float2* scale_shift,
                              char* out, hipStream_t s) {
    hipError_t e = hipMemsetAsync(moments, 0, (size_t)batch * 65 * sizeof(double), s);
    if (e != hipSuccess) return e;
    const int mblocks = (int)std::min<long>((frames + 2047) / 2048, 256);
    hipLaunchKernelGGL(w2v2_moments_kernel, dim3(std::max(mblocks, 1), batch), dim3(256), 0, s, audio, samples, frames, moments);
    hipLaunchKernelGGL(w2v2_norm_kernel, dim3(W2V_C / 256, batch), dim3(256), 0, s, moments, w0, gamma, beta, frames, scale_shift);
    const dim3 grid((rows_per_item + 63) / 64, batch);
    if (precision == PPG_PRECISION_BF16) hipLaunchKernelGGL(w2v2_conv0_kernel<PrecBF16>, grid, dim3(256), 0, s, audio, samples, frames, rows_per_item, w0, scale_shift, out);
    else if (precision == PPG_PRECISION_FP16) hipLaunchKernelGGL(w2v2_conv0_kernel<PrecF16>, grid, dim3(256), 0, s, audio, samples, frames, rows_per_item, w0, scale_shift, out);
    else if (precision == PPG_PRECISION_FP16X2) hipLaunchKernelGGL(w2v2_conv0_kernel<PrecX2>, grid, dim3(256), 0, s, audio, samples, frames, rows_per_item, w0, scale_shift, out);
    else hipLaunchKernelGGL(w2v2_conv0_kernel<PrecF32>, grid, dim3(256), 0, s, audio, samples, frames, rows_per_item, w0, scale_shift, out);
    return hipGetLastError();
}

hipError_t launch_w2v2_output(int precision, const char* rows, int batch, int rows_per_item, long frames, float* out, hipStream_t s) {
    const dim3 grid((unsigned)((frames * (W2V_C / 4) + 255) / 256), batch);
    if (precision == PPG_PRECISION_BF16) hipLaunchKernelGGL(w2v2_output_kernel<PrecBF16>, grid, dim3(256), 0, s, rows, rows_per_item, frames, out);
    else if (precision == PPG_PRECISION_FP16) hipLaunchKernelGGL(w2v2_output_kernel<PrecF16>, grid, dim3(256), 0, s, rows, rows_per_item, frames, out);
    else if (precision == PPG_PRECISION_FP16X2) hipLaunchKernelGGL(w2v2_output_kernel<PrecX2>, grid, dim3(256), 0, s, rows, rows_per_item, frames, out);
    else hipLaunchKernelGGL(w2v2_output_kernel<PrecF32>, grid, dim3(256), 0, s, rows, rows_per_item, frames, out);
    return hipGetLastError();
}

}  // namespace ppg
