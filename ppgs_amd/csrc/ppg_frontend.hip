// Mel frontend on gfx950: reflect pad -> 1024-sample frames at hop 160 ->
// periodic Hann -> 1024-point DFT -> sqrt(re^2+im^2+1e-6) -> fp16 ->
// Slaney mel (sparse triangular rows) -> log(max(.,1e-5)) -> fp16.
// Replaces reference ppgs/preprocess/spectrogram.py:14-50 and mel.py:56-76.
//
// HBM traffic is trivial (640 B of new audio in, 160 B of fp16 mel out per
// frame); the work is LDS round trips and VALU.  Layout of the work:
//   * a workgroup (4 waves) takes 8 consecutive frames of one batch row at a
//     time -- the 2144 samples they span are staged once, 6.4x reuse between
//     overlapping frames -- and loops over such groups (persistent grid), so
//     the twiddle / window / filterbank tables are loaded into LDS once;
//   * ONE WAVE transforms one frame pair (frame A in the real part, frame B in
//     the imaginary part of a complex FFT) with 16 points per lane held in
//     registers: 1024 = 16 x 16 x 4, two radix-16 passes and one radix-4 pass
//     of a Stockham autosort, i.e. three LDS exchanges instead of five, no
//     workgroup barrier inside the transform (LDS operations of one wave are
//     executed in order, so the exchanges are in place in a per-wave buffer);
//   * the pair is separated with the conjugate-symmetry identity, magnitudes
//     are rounded to fp16 where the reference rounds, and the 80 filter rows
//     (x both frames) are dealt to the lanes longest first.
#include "ppg_launch.h"

#include <hip/hip_fp16.h>

namespace {

constexpr int NFFT = 1024;
constexpr int HOP = 160;
constexpr int NBINS = 513;
constexpr int NMELS = 80;
constexpr int PADR = (NFFT - HOP) / 2;        // 432
constexpr int FPB = 8;                        // frames per group (4 pairs = 4 waves)
constexpr int SEG = NFFT + (FPB - 1) * HOP;   // 2144 samples
constexpr int BUF = NFFT + NFFT / 16;         // exchange buffer entries: index i lives at i + (i >> 4)
constexpr int MAGLD = 520;                    // (frame A, frame B) magnitude pairs per wave

// LDS map (bytes)
constexpr int OFF_TW = 0;                                   // float2[1024]
constexpr int OFF_MELW = OFF_TW + NFFT * 8;                 // float[kMaxMelWeights]
constexpr int OFF_META = OFF_MELW + ppg::kMaxMelWeights * 4;    // int start[80], count[80], offset[80], task[160]
constexpr int OFF_SEG = OFF_META + (3 * NMELS + 2 * NMELS) * 4;
constexpr int OFF_FFT = OFF_SEG + SEG * 4;                  // float2[4][BUF]
constexpr int OFF_MAG = OFF_FFT + 4 * BUF * 8;              // float2[4][MAGLD]: both frames of a pair side by side
constexpr int OFF_OUT = OFF_MAG + 4 * 2 * MAGLD * 4;        // __half[80][8]
constexpr int LDS_BYTES = OFF_OUT + NMELS * FPB * 2;
static_assert(LDS_BYTES <= 81920, "two workgroups per CU");

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }   // a * (-i)

// forward 4-point DFT, in place: (A, B, C, D) -> (X0, X1, X2, X3)
__device__ __forceinline__ void radix4(float2& A, float2& B, float2& C, float2& D) {
    const float2 apc = cadd(A, C), amc = csub(A, C), bpd = cadd(B, D), jb = mul_mi(csub(B, D));
    A = cadd(apc, bpd);
    B = cadd(amc, jb);
    C = csub(apc, bpd);
    D = csub(amc, jb);
}

// W16^m = exp(-2 pi i m / 16) for the exponents r*q the 4 x 4 decomposition uses
template <int M>
__device__ __forceinline__ float2 w16() {
    constexpr float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, r = 0.70710678118654752f;
    if constexpr (M == 0) return make_float2(1.f, 0.f);
    else if constexpr (M == 1) return make_float2(c1, -s1);
    else if constexpr (M == 2) return make_float2(r, -r);
    else if constexpr (M == 3) return make_float2(s1, -c1);
    else if constexpr (M == 4) return make_float2(0.f, -1.f);
    else if constexpr (M == 6) return make_float2(-r, -r);
    else { static_assert(M == 9, "exponent"); return make_float2(-c1, s1); }
}

// forward 16-point DFT in registers, natural order in and out:
// n = r + 4j, k = q + 4k2:  X[k] = sum_r W16^(rq) W4^(r k2) sum_j x[r + 4j] W4^(jq)
__device__ __forceinline__ void dft16(float2 (&x)[16]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) radix4(x[r], x[r + 4], x[r + 8], x[r + 12]);      // x[r + 4q] = a[r][q]
    x[5] = cmul(x[5], w16<1>());  x[9] = cmul(x[9], w16<2>());   x[13] = cmul(x[13], w16<3>());
    x[6] = cmul(x[6], w16<2>());  x[10] = mul_mi(x[10]);         x[14] = cmul(x[14], w16<6>());
    x[7] = cmul(x[7], w16<3>());  x[11] = cmul(x[11], w16<6>()); x[15] = cmul(x[15], w16<9>());
#pragma unroll
    for (int q = 0; q < 4; ++q) radix4(x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]);   // x[4q + k2] = X[q + 4 k2]
    // transpose the 4 x 4 index so that x[k] = X[k]
    float2 t;
#define PPG_SWAP(a, b) t = x[a]; x[a] = x[b]; x[b] = t;
    PPG_SWAP(1, 4) PPG_SWAP(2, 8) PPG_SWAP(3, 12) PPG_SWAP(6, 9) PPG_SWAP(7, 13) PPG_SWAP(11, 14)
#undef PPG_SWAP
}

__device__ __forceinline__ int pad(int i) { return i + (i >> 4); }

// LDS exchanges between the lanes of one wave: hardware executes a wave's LDS
// operations in order; this only stops the compiler from moving them.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(256, 2) void frontend_kernel(
    ppg::FrontendTables tb, const float* __restrict__ audio, int samples, int frames,
    int groups_per_row, int total_groups, __half* __restrict__ spec, __half* __restrict__ mel)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* tw = reinterpret_cast<float2*>(smem + OFF_TW);
    float* melw = reinterpret_cast<float*>(smem + OFF_MELW);
    int* mstart = reinterpret_cast<int*>(smem + OFF_META);
    int* mcount = mstart + NMELS;
    int* moffset = mcount + NMELS;
    int* mtask = moffset + NMELS;
    float* seg = reinterpret_cast<float*>(smem + OFF_SEG);
    __half* melout = reinterpret_cast<__half*>(smem + OFF_OUT);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    float2* buf = reinterpret_cast<float2*>(smem + OFF_FFT) + wave * BUF;
    float2* mag = reinterpret_cast<float2*>(smem + OFF_MAG) + wave * MAGLD;

    for (int i = tid; i < NFFT; i += 256) tw[i] = tb.twiddle[i];
    if (lane < MAGLD - NBINS) mag[NBINS + lane] = make_float2(0.f, 0.f);
    for (int i = tid; i < tb.mel_weights; i += 256) melw[i] = tb.mel_weight[i];
    if (tid < NMELS) { mstart[tid] = tb.mel_start[tid]; mcount[tid] = tb.mel_count[tid]; moffset[tid] = tb.mel_offset[tid]; }
    if (tid < NMELS) mtask[tid] = tb.mel_task[tid];

    // per-lane constants of the persistent loop: the window at the lane's 16
    // sample positions and the pass-1 twiddles W^(l k) (LDS bandwidth is what
    // bounds the transform; these were 12 KiB of LDS reads per frame pair)
    float hreg[16];
    float2 t1[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) hreg[j] = tb.hann[lane + 64 * j];
#pragma unroll
    for (int k = 1; k < 16; ++k) t1[k] = tb.twiddle[lane * k];

    // samples of a group, SEG / 256 per thread: fetched one group ahead, so the
    // global latency hides behind the previous group's transforms
    constexpr int PER = (SEG + 255) / 256;
    auto fetch = [&](int grp, float (&pre)[PER]) {
        const int b = grp / groups_per_row;
        const int f0 = (grp - b * groups_per_row) * FPB;
        const float* arow = audio + (size_t)b * samples;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            // reflect-padded segment: padded index i -> source i - 432
            int src = f0 * HOP + tid + 256 * j - PADR;
            if (src < 0) src = -src;
            if (src >= samples) src = 2 * (samples - 1) - src;
            float v = 0.f;
            if (src >= 0 && src < samples) v = arow[src];
            pre[j] = v;
        }
    };
    float pre[PER];
    if ((int)blockIdx.x < total_groups) fetch(blockIdx.x, pre);

#ifdef PPG_FE_TIMING
    int stamp_group = 0;
    auto stamp = [&](int k) {
        if (tb.dbg && blockIdx.x == 0 && lane == 0 && stamp_group == 2) tb.dbg[wave * 16 + k] = __builtin_amdgcn_s_memtime();
    };
#else
    auto stamp = [&](int) {};
#endif
    for (int grp = blockIdx.x; grp < total_groups; grp += gridDim.x) {
        const int b = grp / groups_per_row;
        const int f0 = (grp - b * groups_per_row) * FPB;
#ifdef PPG_FE_TIMING
        ++stamp_group;
#endif
        stamp(0);
        __syncthreads();                 // tables ready / previous group's seg and melout consumed
#pragma unroll
        for (int j = 0; j < PER; ++j)
            if (tid + 256 * j < SEG) seg[tid + 256 * j] = pre[j];
        if (grp + (int)gridDim.x < total_groups) fetch(grp + gridDim.x, pre);
        stamp(1);
        __syncthreads();
        stamp(2);

        const int fa = f0 + 2 * wave;    // this wave's frame pair (fa, fa + 1)
        if (fa < frames) {
            float2 x[16];
            // pass 1: radix 16, stride 1 -- lane l owns samples l + 64 j of both frames
            {
                const float* sa = seg + (2 * wave) * HOP + lane;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    x[j] = make_float2(sa[64 * j] * hreg[j], sa[64 * j + HOP] * hreg[j]);
                }
                dft16(x);
#pragma unroll
                for (int k = 1; k < 16; ++k) x[k] = cmul(x[k], t1[k]);
#pragma unroll
                for (int k = 0; k < 16; ++k) buf[17 * lane + k] = x[k];         // pad(16 l + k)
            }
            stamp(3);
            wave_sync();
            // Exchange-buffer addressing: index i lives at pad(i) = i + (i >> 4).  All
            // strides below are multiples of 16, so every address is a lane base plus
            // a compile-time offset (ds immediates instead of VALU).
            const int p0 = lane + (lane >> 4);                     // pad(lane)
            // pass 2: radix 16, stride 16 -- p = l / 16, q = l % 16
            {
#pragma unroll
                for (int j = 0; j < 16; ++j) x[j] = buf[p0 + 68 * j];           // pad(l + 64 j)
                dft16(x);
                const int p = lane >> 4, q = lane & 15;
#pragma unroll
                for (int k = 1; k < 16; ++k) x[k] = cmul(x[k], tw[16 * p * k]);
                float2* dst = buf + q + 272 * p;                                  // pad(q + 256 p + 16 k) = q + 272 p + 17 k
#pragma unroll
                for (int k = 0; k < 16; ++k) dst[17 * k] = x[k];
            }
            stamp(4);
            wave_sync();
            // pass 3: radix 4, stride 256, no twiddles; lane l owns q = l + 64 m
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                float2* z = buf + p0 + 68 * m;                                    // pad(q + 256 j) = pad(q) + 272 j
                float2 A = z[0], B = z[272], C = z[544], D = z[816];
                radix4(A, B, C, D);
                z[0] = A; z[272] = B; z[544] = C; z[816] = D;
            }
            stamp(5);
            wave_sync();
            // buf holds Z = FFT(a + i b); split: A[k] = (Z[k] + conj Z[N-k]) / 2,
            // B[k] = (Z[k] - conj Z[N-k]) / (2i);  k = l + 64 r, N - k = (1024 - l) - 64 r
            const int nb = (NFFT - lane) + ((NFFT - lane) >> 4);   // pad(1024 - l); lane 0, r = 0 wraps to Z[0]
            auto split_bin = [&](int r) {
                const int k = lane + 64 * r;
                const float2 zk = buf[p0 + 68 * r];
                const float2 zn = buf[(r == 0 && lane == 0) ? 0 : nb - 68 * r];
                const float ar = 0.5f * (zk.x + zn.x), ai = 0.5f * (zk.y - zn.y);
                const float br = 0.5f * (zk.y + zn.y), bi = -0.5f * (zk.x - zn.x);
                // arguments are >= 1e-6: the bare v_sqrt_f32 (1 ulp) needs none of sqrtf's
                // denormal handling, and the result is rounded to fp16 next
                const __half ha = __float2half_rn(__builtin_amdgcn_sqrtf(ar * ar + ai * ai + 1e-6f));
                const __half hb = __float2half_rn(__builtin_amdgcn_sqrtf(br * br + bi * bi + 1e-6f));
                mag[k] = make_float2(__half2float(ha), __half2float(hb));
                if (spec) {
                    spec[((size_t)b * NBINS + k) * frames + fa] = ha;
                    if (fa + 1 < frames) spec[((size_t)b * NBINS + k) * frames + fa + 1] = hb;
                }
            };
#pragma unroll
            for (int r = 0; r < 8; ++r) split_bin(r);
            if (lane == 0) split_bin(8);                           // k = 512
            stamp(6);
            wave_sync();
            if (mel) {
                // 80 filters x both frames of the pair: lane l takes schedule entries l and 127 - l
                // (longest filters first, so the two rounds are 5 + 1 iterations of 8 bins)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int slot = s == 0 ? lane : 127 - lane;
                    if (slot < NMELS) {
                        const int m = mtask[slot];
                        const int count = mcount[m];
                        const float4* wt = reinterpret_cast<const float4*>(melw + moffset[m]);   // rows are 32-byte aligned
                        const float2* mg = mag + mstart[m];
                        // sequential sums (the order of the oracle's sparse rows); filter rows are
                        // zero-padded to multiples of 8 bins by the host so that 8 terms share one
                        // LDS round trip (magnitude rows end in 7 zeroed pad floats)
                        float acc0 = 0.f, acc1 = 0.f;
                        for (int i = 0; i < count; i += 8) {
                            const float4 wa = wt[i / 4], wb = wt[i / 4 + 1];
                            const float w[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
                            float2 gm[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u) gm[u] = mg[i + u];
#pragma unroll
                            for (int u = 0; u < 8; ++u) { acc0 += w[u] * gm[u].x; acc1 += w[u] * gm[u].y; }
                        }
                        melout[m * FPB + 2 * wave] = __float2half_rn(logf(fmaxf(acc0, 1e-5f)));
                        melout[m * FPB + 2 * wave + 1] = __float2half_rn(logf(fmaxf(acc1, 1e-5f)));
                    }
                }
            }
        }
        stamp(7);
        __syncthreads();
        stamp(8);
        if (mel) {
            for (int i = tid; i < NMELS * FPB; i += 256) {
                const int m = i / FPB, j = i % FPB;
                if (f0 + j < frames) mel[((size_t)b * NMELS + m) * frames + f0 + j] = melout[i];
            }
        }
        stamp(9);
    }
}

}  // namespace

namespace ppg {

hipError_t launch_frontend(const FrontendTables& tb, const float* audio, int batch, int samples,
                           void* spec, void* mel, hipStream_t s) {
    const int frames = samples / HOP;
    const int groups_per_row = (frames + FPB - 1) / FPB;
    const int total = groups_per_row * batch;
    static LdsLimit limit;
    const hipError_t e = limit.ensure(reinterpret_cast<const void*>(frontend_kernel), LDS_BYTES);
    if (e != hipSuccess) return e;
    static int slots = 0;                // resident workgroups (two per CU; the devices of a node are alike)
    if (slots == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
        slots = 2 * cus;
    }
    // persistent grid: every workgroup gets the same number of groups, +-1
    const int rounds = (total + slots - 1) / slots;
    const int grid = (total + rounds - 1) / rounds;
    hipLaunchKernelGGL(frontend_kernel, dim3(grid), dim3(256), LDS_BYTES, s, tb, audio, samples, frames,
                       groups_per_row, total, reinterpret_cast<__half*>(spec), reinterpret_cast<__half*>(mel));
    return hipGetLastError();
}

}  // namespace ppg
