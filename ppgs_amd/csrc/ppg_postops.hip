// PPG post-ops on the device: the per-frame arithmetic of
// reference ppgs.distance (ppgs/core.py:399-472) and ppgs.sparsify (:510-543),
// and the time-stretch gather of ppgs.edit.grid.sample (ppgs/edit/grid.py:13-45).
// 40 phonemes per frame: one thread per frame, everything in registers (all
// loops over the 40 channels are fully unrolled), reads and writes coalesced
// over frames.  HBM-trivial: 160 B in per posteriorgram frame.
#include "../../include/ppgs_amd.h"

#include <hip/hip_runtime.h>

namespace ppg {
int fail_message(int code, const char* fmt, ...);
}

namespace {

constexpr int NP = 40;   // phonemes (ppgs/phonemes.py)

// jsd[t] = sum_p sqrt(max(0, (KL(X'||m) + KL(Y'||m)) / 2)), X' = M x (or x), m = (X' + Y') / 2
__global__ __launch_bounds__(64) void distance_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                       int frames, const float* __restrict__ mix, float* __restrict__ out)
{
    __shared__ float m[NP * NP];
    if (mix) for (int i = threadIdx.x; i < NP * NP; i += 64) m[i] = mix[i];
    __syncthreads();
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t >= frames) return;
    float xv[NP], yv[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        xv[p] = fminf(fmaxf(x[(size_t)p * frames + t], 1e-8f), 1.f - 1e-8f);
        yv[p] = fminf(fmaxf(y[(size_t)p * frames + t], 1e-8f), 1.f - 1e-8f);
    }
    float jsd = 0.f;
#pragma unroll 1
    for (int p = 0; p < NP; ++p) {
        float a, b;
        if (mix) {
            a = 0.f; b = 0.f;
#pragma unroll
            for (int q = 0; q < NP; ++q) { a += m[p * NP + q] * xv[q]; b += m[p * NP + q] * yv[q]; }
        } else {
            // static register indexing needs a compile-time p: select without a dynamic index
            a = 0.f; b = 0.f;
#pragma unroll
            for (int q = 0; q < NP; ++q) { a = q == p ? xv[q] : a; b = q == p ? yv[q] : b; }
        }
        const float la = logf((a + b) * 0.5f);
        const float kx = a * (logf(a) - la), ky = b * (logf(b) - la);
        const float avg = fmaxf((kx + ky) * 0.5f, 0.f);
        jsd += sqrtf(avg);
    }
    out[t] = jsd;
}

// method 0: keep values > threshold; 1: > per-frame quantile(threshold) (linear
// interpolation between order statistics, torch.quantile); 2: the round(threshold)
// largest.  Then renormalise: (v + 1e-8) / sum(v + 1e-8).
__global__ __launch_bounds__(64) void sparsify_kernel(const float* __restrict__ ppg, int frames, int method,
                                                       float threshold, float* __restrict__ out)
{
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t >= frames) return;
    const float* src = ppg + (size_t)blockIdx.y * NP * frames + t;
    float v[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) v[p] = src[(size_t)p * frames];
    bool keep[NP];
    if (method == 0) {
#pragma unroll
        for (int p = 0; p < NP; ++p) keep[p] = v[p] > threshold;
    } else {
        // rank of every value in ascending order (ties by index): the order statistics
        int rank[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            int r = 0;
#pragma unroll
            for (int q = 0; q < NP; ++q) r += (v[q] < v[p] || (v[q] == v[p] && q < p)) ? 1 : 0;
            rank[p] = r;
        }
        if (method == 1) {
            const float pos = threshold * (NP - 1);
            const int lo = (int)floorf(pos), hi = (int)ceilf(pos);
            const float w = pos - (float)lo;
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int p = 0; p < NP; ++p) { a = rank[p] == lo ? v[p] : a; b = rank[p] == hi ? v[p] : b; }
            const float thr = w < 0.5f ? a + w * (b - a) : b - (b - a) * (1.f - w);      // torch.lerp
#pragma unroll
            for (int p = 0; p < NP; ++p) keep[p] = v[p] > thr;
        } else {
            const int k = (int)(threshold + 0.5f);
#pragma unroll
            for (int p = 0; p < NP; ++p) keep[p] = rank[p] >= NP - k;
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int p = 0; p < NP; ++p) { v[p] = (keep[p] ? v[p] : 0.f) + 1e-8f; sum += v[p]; }
    const float inv = 1.f / sum;
    float* dst = out + (size_t)blockIdx.y * NP * frames + t;
#pragma unroll
    for (int p = 0; p < NP; ++p) dst[(size_t)p * frames] = v[p] * inv;
}

// out[r, g] = (1 - w) * ppg[r, lo] + w * ppg[r, hi] for the fractional frame index
// grid[g] (ppgs/edit/grid.py:13-45): hi = #{frames <= grid[g]} (searchsorted
// 'right' over 0..frames-1), lo = hi - 1, w = grid - floor(grid); the PPG is
// extended by one copy of its last frame, and lo = -1 (a negative grid value)
// indexes from the end like the reference's tensor indexing does, i.e. that copy.
// One thread per output frame, looping over the rows it is given: consecutive
// lanes read (nearly) consecutive frames of a row for any monotone grid.
__global__ __launch_bounds__(256) void grid_sample_kernel(const float* __restrict__ ppg, int rows, int frames,
                                                          const float* __restrict__ grid, int length,
                                                          float* __restrict__ out) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= length) return;
    const float x = grid[g];
    const float fl = floorf(x);
    const float w = x - fl;
    int hi = x < 0.f ? 0 : (fl >= (float)frames ? frames : (int)fl + 1);
    if (!(x == x)) hi = frames;                       // NaN sorts after every frame
    int lo = hi - 1;
    if (lo < 0) lo = frames;                          // index -1 of the extended PPG
    const int last = frames - 1;
    lo = lo > last ? last : lo;                       // the extra frame is the last one again
    hi = hi > last ? last : hi;
    const float u = 1.f - w;
    for (int r = blockIdx.y; r < rows; r += gridDim.y) {
        const float* src = ppg + (size_t)r * frames;
        out[(size_t)r * length + g] = __fadd_rn(__fmul_rn(u, src[lo]), __fmul_rn(w, src[hi]));
    }
}

}  // namespace

extern "C" {

int ppg_distance(int device, const float* ppg_x, const float* ppg_y, int frames, const float* mix,
                 float* jsd, void* stream) {
    if (!ppg_x || !ppg_y || !jsd || frames <= 0) return ppg::fail_message(PPG_EINVAL, "distance: bad argument");
    if (hipSetDevice(device) != hipSuccess) return ppg::fail_message(PPG_EDEVICE, "no HIP device: the post-ops have no CPU path");
    hipLaunchKernelGGL(distance_kernel, dim3((frames + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream),
                       ppg_x, ppg_y, frames, mix, jsd);
    const hipError_t he = hipGetLastError();
    return he == hipSuccess ? PPG_OK : ppg::fail_message(PPG_EDEVICE, "distance: %s", hipGetErrorString(he));
}

int ppg_sparsify(int device, const float* ppg, int batch, int frames, int method, float threshold,
                 float* out, void* stream) {
    if (!ppg || !out || batch <= 0 || frames <= 0 || method < 0 || method > 2)
        return ppg::fail_message(PPG_EINVAL, "sparsify: bad argument");
    if (method == 2 && (threshold < 0.5f || threshold > 40.5f))
        return ppg::fail_message(PPG_EINVAL, "sparsify: topk needs 1 <= k <= 40");
    if (hipSetDevice(device) != hipSuccess) return ppg::fail_message(PPG_EDEVICE, "no HIP device: the post-ops have no CPU path");
    hipLaunchKernelGGL(sparsify_kernel, dim3((frames + 63) / 64, batch), dim3(64), 0, static_cast<hipStream_t>(stream),
                       ppg, frames, method, threshold, out);
    const hipError_t he = hipGetLastError();
    return he == hipSuccess ? PPG_OK : ppg::fail_message(PPG_EDEVICE, "sparsify: %s", hipGetErrorString(he));
}

int ppg_grid_sample(int device, const float* ppg, int rows, int frames, const float* grid, int length,
                    float* out, void* stream) {
    if (!ppg || rows <= 0 || frames <= 0 || length < 0 || (length > 0 && (!grid || !out)))
        return ppg::fail_message(PPG_EINVAL, "grid_sample: bad argument");
    if (hipSetDevice(device) != hipSuccess) return ppg::fail_message(PPG_EDEVICE, "no HIP device: the post-ops have no CPU path");
    if (length == 0) return PPG_OK;
    hipLaunchKernelGGL(grid_sample_kernel, dim3((length + 255) / 256, rows < 4096 ? rows : 4096), dim3(256), 0,
                       static_cast<hipStream_t>(stream), ppg, rows, frames, grid, length, out);
    const hipError_t he = hipGetLastError();
    return he == hipSuccess ? PPG_OK : ppg::fail_message(PPG_EDEVICE, "grid_sample: %s", hipGetErrorString(he));
}

}  // extern "C"
