// Mel frontend on gfx950: reflect pad -> 1024-sample frames at hop 160 ->
// periodic Hann -> 1024-point DFT -> sqrt(re^2+im^2+1e-6) -> fp16 ->
// Slaney mel (sparse triangular rows) -> log(max(.,1e-5)) -> fp16.
// Replaces reference ppgs/preprocess/spectrogram.py:14-50 and mel.py:56-76.
//
// HBM-bound by construction: 640 B of new audio in and 160 B of fp16 mel out
// per frame; everything between lives in LDS.  One workgroup handles FPB = 8
// consecutive frames of one batch row: the 2144 samples they span are staged
// once (6.4x reuse between overlapping frames), frames are transformed two at
// a time (frame A in the real part, frame B in the imaginary part of one
// complex radix-4 Stockham FFT, 5 passes of 256 butterflies = one per thread),
// and the spectra are separated with the conjugate-symmetry identity.
#include "ppg_launch.h"

#include <hip/hip_fp16.h>

namespace {

constexpr int NFFT = 1024;
constexpr int HOP = 160;
constexpr int NBINS = 513;
constexpr int NMELS = 80;
constexpr int PADR = (NFFT - HOP) / 2;   // 432
constexpr int FPB = 8;                   // frames per workgroup
constexpr int SEG = NFFT + (FPB - 1) * HOP;   // 2144 samples

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

__global__ __launch_bounds__(256) void frontend_kernel(
    ppg::FrontendTables tb, const float* __restrict__ audio, int samples, int frames,
    __half* __restrict__ spec, __half* __restrict__ mel)
{
    __shared__ float seg[SEG];
    __shared__ float2 bufa[NFFT];
    __shared__ float2 bufb[NFFT];
    __shared__ float mag[2][NBINS + 3];          // fp16-rounded magnitudes of the frame pair
    __shared__ __half melout[NMELS][FPB];

    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int f0 = blockIdx.x * FPB;
    const float* arow = audio + (size_t)b * samples;

    // stage the reflect-padded segment: padded index j -> source j - 432
    for (int i = tid; i < SEG; i += 256) {
        int src = f0 * HOP + i - PADR;
        if (src < 0) src = -src;
        if (src >= samples) src = 2 * (samples - 1) - src;
        float v = 0.f;
        if (src >= 0 && src < samples) v = arow[src];
        seg[i] = v;
    }
    __syncthreads();

    for (int pair = 0; pair < FPB / 2; ++pair) {
        const int fa = f0 + 2 * pair;
        if (fa >= frames) break;                     // uniform
        // windowed frames -> complex input
        for (int i = tid; i < NFFT; i += 256) {
            const float wv = tb.hann[i];
            bufa[i] = make_float2(seg[(2 * pair) * HOP + i] * wv, seg[(2 * pair + 1) * HOP + i] * wv);
        }
        __syncthreads();
        // radix-4 Stockham, decimation in frequency: pass with sub-length n,
        // stride s:  p = tid / s, q = tid % s, twiddle exp(-2 pi i p s / 1024)
        float2* x = bufa;
        float2* y = bufb;
#pragma unroll
        for (int pass = 0; pass < 5; ++pass) {
            const int s = 1 << (2 * pass);
            const int n1 = 256 >> (2 * pass);
            const int p = tid >> (2 * pass);
            const int q = tid & (s - 1);
            const float2 A = x[q + s * p];
            const float2 B = x[q + s * (p + n1)];
            const float2 C = x[q + s * (p + 2 * n1)];
            const float2 D = x[q + s * (p + 3 * n1)];
            const float2 apc = make_float2(A.x + C.x, A.y + C.y);
            const float2 amc = make_float2(A.x - C.x, A.y - C.y);
            const float2 bpd = make_float2(B.x + D.x, B.y + D.y);
            const float2 bmd = make_float2(B.x - D.x, B.y - D.y);
            const float2 jb = make_float2(bmd.y, -bmd.x);           // -i (B - D)
            const int ti = p * s;
            const float2 w1 = tb.twiddle[ti];
            const float2 w2 = tb.twiddle[2 * ti];
            const float2 w3 = tb.twiddle[3 * ti];
            y[q + s * (4 * p + 0)] = make_float2(apc.x + bpd.x, apc.y + bpd.y);
            y[q + s * (4 * p + 1)] = cmul(make_float2(amc.x + jb.x, amc.y + jb.y), w1);
            y[q + s * (4 * p + 2)] = cmul(make_float2(apc.x - bpd.x, apc.y - bpd.y), w2);
            y[q + s * (4 * p + 3)] = cmul(make_float2(amc.x - jb.x, amc.y - jb.y), w3);
            __syncthreads();
            float2* tmp = x; x = y; y = tmp;
        }
        // x holds Z = FFT(a + i b); split: A[k] = (Z[k] + conj Z[N-k]) / 2,
        // B[k] = (Z[k] - conj Z[N-k]) / (2i)
        for (int k = tid; k < NBINS; k += 256) {
            const float2 zk = x[k];
            const float2 zn = x[(NFFT - k) & (NFFT - 1)];
            const float ar = 0.5f * (zk.x + zn.x), ai = 0.5f * (zk.y - zn.y);
            const float br = 0.5f * (zk.y + zn.y), bi = -0.5f * (zk.x - zn.x);
            const __half ha = __float2half_rn(sqrtf(ar * ar + ai * ai + 1e-6f));
            const __half hb = __float2half_rn(sqrtf(br * br + bi * bi + 1e-6f));
            mag[0][k] = __half2float(ha);
            mag[1][k] = __half2float(hb);
            if (spec) {
                spec[((size_t)b * NBINS + k) * frames + fa] = ha;
                if (fa + 1 < frames) spec[((size_t)b * NBINS + k) * frames + fa + 1] = hb;
            }
        }
        __syncthreads();
        if (mel && tid < 2 * NMELS) {
            const int which = tid / NMELS;
            const int m = tid - which * NMELS;
            const int start = tb.mel_start[m], count = tb.mel_count[m];
            const float* wt = tb.mel_weight + tb.mel_offset[m];
            float acc = 0.f;
            for (int i = 0; i < count; ++i) acc += wt[i] * mag[which][start + i];
            melout[m][2 * pair + which] = __float2half_rn(logf(fmaxf(acc, 1e-5f)));
        }
        __syncthreads();
    }
    if (mel) {
        for (int i = tid; i < NMELS * FPB; i += 256) {
            const int m = i / FPB, j = i % FPB;
            if (f0 + j < frames) mel[((size_t)b * NMELS + m) * frames + f0 + j] = melout[m][j];
        }
    }
}

}  // namespace

namespace ppg {

hipError_t launch_frontend(const FrontendTables& tb, const float* audio, int batch, int samples,
                           void* spec, void* mel, hipStream_t s) {
    const int frames = samples / HOP;
    dim3 grid((frames + FPB - 1) / FPB, batch);
    hipLaunchKernelGGL(frontend_kernel, grid, dim3(256), 0, s, tb, audio, samples, frames,
                       reinterpret_cast<__half*>(spec), reinterpret_cast<__half*>(mel));
    return hipGetLastError();
}

}  // namespace ppg
