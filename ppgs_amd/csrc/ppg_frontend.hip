// Mel frontend on gfx950: reflect pad -> 1024-sample frames at hop 160 ->
// periodic Hann -> 1024-point DFT -> sqrt(re^2+im^2+1e-6) -> fp16 ->
// Slaney mel (sparse triangular rows) -> log(max(.,1e-5)) -> fp16.
// Replaces reference ppgs/preprocess/spectrogram.py:14-50 and mel.py:56-76.
//
// HBM traffic is trivial (640 B of new audio in, 160 B of fp16 mel out per
// frame); the work is LDS round trips and VALU.  Layout of the work:
//   * a workgroup (4 waves) takes 16 consecutive frames of one batch row at a
//     time -- the 3424 samples they span are staged once (double-buffered, the
//     next group's samples are written while this group is transformed), 4.8x
//     reuse between overlapping frames -- and loops over such groups
//     (persistent grid);
//   * ONE WAVE transforms one frame pair (frame A in the real part, frame B in
//     the imaginary part of a complex FFT) with 16 points per lane held in
//     registers: 1024 = 16 x 16 x 4, two radix-16 passes and one radix-4 pass
//     of a Stockham autosort, i.e. three LDS exchanges instead of five, no
//     workgroup barrier inside the transform (LDS operations of one wave are
//     executed in order, so the exchanges are in place in a per-wave buffer);
//   * the pair is separated with the conjugate-symmetry identity, magnitudes
//     are rounded to fp16 where the reference rounds and kept as fp16 rows
//     [frame][bin] in LDS;
//   * the filterbank is a banded matrix product on the matrix cores: 16 filters
//     x 16 frames x 32 bins per v_mfma_f32_16x16x32_f16, only over the bins a
//     block of 16 filters touches (20 steps for the 80 Slaney filters instead of
//     5 x 17).  The magnitudes ARE fp16; a float32 weight is the sum of two fp16
//     values (w 2^16 = hi + lo, |error| <= 2^-22 w), so two MFMAs per step, fp32
//     accumulation; the weight fragments live in registers for the whole launch.
#include "ppg_lds.h"
#include "ppg_launch.h"

#include <hip/hip_fp16.h>

namespace {

constexpr int NFFT = 1024;
constexpr int HOP = 160;
constexpr int NBINS = 513;
constexpr int NMELS = 80;
constexpr int PADR = (NFFT - HOP) / 2;        // 432
constexpr int FPB = ppg::kFrontendFrames;     // frames per group: 8 pairs, two per wave
constexpr int SEG = NFFT + (FPB - 1) * HOP;   // 3424 samples
constexpr int BUF = NFFT + NFFT / 16;         // exchange buffer entries: index i lives at i + (i >> 4)
constexpr int MSEG = ppg::kMelSegment;      // ... a filter block ends after step MSEG - 1 or after the last step
constexpr int NS = ppg::kMelSteps;            // filterbank steps per wave
// Magnitude rows: 544 fp16 (513 bins + zeros) = 1088 bytes = 272 dwords.  Pair j of the group
// keeps its first frame in row j and its second in row j + 8 (MFMA column n = row).  A
// filterbank B fragment is one ds_read_b128 per lane (n = lane & 15, 8 bins from 8 (lane >> 4));
// that instruction is served in lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... , i.e.
// every n once with k-group 0 or 1: rows 16 banks apart (272 mod 64) put n & 3 on different
// quarters, and the 16-byte chunks of a row are stored at chunk ^ sigma(n >> 2) within their
// 64-byte step so that the four n >> 2 classes of a group fall on different chunks.
constexpr int MROW = 1088;
__device__ __forceinline__ int sigma(int c) { return (0x6c >> (2 * c)) & 3; }   // 0, 3, 2, 1

// LDS map (bytes)
constexpr int OFF_TW = 0;                                   // float2[64]: exp(-2 pi i j / 64)
constexpr int SEGP = (SEG + 255) / 256 * 256;                // the DMA writes whole 256-sample (1 KiB) pieces
constexpr int OFF_SEG = OFF_TW + 64 * 8;                    // float[2][SEGP]
constexpr int OFF_FFT = OFF_SEG + 2 * SEGP * 4;             // float2[4][BUF]
constexpr int OFF_MAG = OFF_FFT + 4 * BUF * 8;              // fp16 rows [16][MROW / 2]
constexpr int LDS_BYTES = OFF_MAG + FPB * MROW;
// Four-wave teams per workgroup.  The product is 2: a 512-thread workgroup that requests the CU's whole LDS, so no other
// kernel's workgroup can sit beside it (round 2's launch -- one team of 256 threads and 79.5 KiB per
// workgroup -- was where the gfx950 hazard of DESIGN 4.4 was found: tools/probes/pk_mfma_probe.hip reproduces it alone).
constexpr int TEAMS = 2;
constexpr int CU_LDS_BYTES = 163840;
static_assert(2 * LDS_BYTES <= CU_LDS_BYTES && LDS_BYTES % 16 == 0, "two teams per CU; a two-team workgroup = the CU's LDS");
static_assert(OFF_MAG % 16 == 0 && MROW % 16 == 0, "fragment reads are 16-byte aligned");

// Complex arithmetic on (re, im) register pairs with the packed fp32 instructions.  Swapping or negating
// a half of an operand is an instruction modifier (op_sel / neg_lo / neg_hi): written without them, every
// multiplication by -i and every complex product costs two v_mov_b32 to build the swapped pair.
//
// ONE operand position must never be re-selected: SOURCE 1 with its op_sel bit set (the low result half reading
// the HIGH half of source 1 -- a swap or a broadcast of the high half).  On gfx950 that form intermittently
// computes with the wrong half while ANOTHER wave of the same SIMD issues 16x16x32 MFMAs -- a wave of another
// kernel (round 2's 256-thread workgroups beside an attention kernel: one frame pair in ~300 wrong) or of the same
// workgroup; source 0, source 2 and every selection with op_sel[1] = 0 are unaffected
// (tools/probes/pk_mfma_probe.hip reproduces it without this package).  Add and multiply commute, so
// the operand that needs its high half first simply goes to source 0; a complex product needs the high halves of
// BOTH factors for its real part, so its second half is two plain FMAs.  tools/pk_scan.py --strict (run by
// tests/test_host.py) fails the build of any kernel that contains the form and can share a SIMD.
typedef float cplx __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return a + b; }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return a - b; }
// a + (-i) b = (b.y + a.x, -b.x + a.y): b, swapped, is source 0
__device__ __forceinline__ cplx cadd_mi(cplx a, cplx b) {
    cplx r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1] neg_hi:[1,0]" : "=v"(r) : "v"(b), "v"(a));
    return r;
}
// a - (-i) b = (-b.y + a.x, b.x + a.y)
__device__ __forceinline__ cplx csub_mi(cplx a, cplx b) {
    cplx r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[1,0]" : "=v"(r) : "v"(b), "v"(a));
    return r;
}
// a w = (a.x w.x - a.y w.y, a.x w.y + a.y w.x): (a.x, a.x) * w packed (a is source 0), then -a.y w.y and a.y w.x on top
__device__ __forceinline__ cplx cmul(cplx a, cplx w) {
    cplx t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_fma_f32 %0, -%1, %2, %3" : "=v"(r.x) : "v"(a.y), "v"(w.y), "v"(t.x));
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r.y) : "v"(a.y), "v"(w.x), "v"(t.y));
    return r;
}

// forward 4-point DFT, in place: (A, B, C, D) -> (X0, X1, X2, X3); CMI: the input C is (-i) C
template <bool CMI = false>
__device__ __forceinline__ void radix4(cplx& A, cplx& B, cplx& C, cplx& D) {
    const cplx apc = CMI ? cadd_mi(A, C) : cadd(A, C), amc = CMI ? csub_mi(A, C) : csub(A, C);
    const cplx bpd = cadd(B, D), bmd = csub(B, D);
    A = cadd(apc, bpd);
    B = cadd_mi(amc, bmd);
    C = csub(apc, bpd);
    D = csub_mi(amc, bmd);
}

// W16^m = exp(-2 pi i m / 16) for the exponents r*q the 4 x 4 decomposition uses
template <int M>
__device__ __forceinline__ cplx w16() {
    constexpr float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, r = 0.70710678118654752f;
    if constexpr (M == 1) return cplx{c1, -s1};
    else if constexpr (M == 2) return cplx{r, -r};
    else if constexpr (M == 3) return cplx{s1, -c1};
    else if constexpr (M == 6) return cplx{-r, -r};
    else { static_assert(M == 9, "exponent"); return cplx{-c1, s1}; }
}

// forward 16-point DFT in registers, natural order in and out:
// n = r + 4j, k = q + 4k2:  X[k] = sum_r W16^(rq) W4^(r k2) sum_j x[r + 4j] W4^(jq)
__device__ __forceinline__ void dft16(cplx (&x)[16]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) radix4(x[r], x[r + 4], x[r + 8], x[r + 12]);      // x[r + 4q] = a[r][q]
    x[5] = cmul(x[5], w16<1>());  x[9] = cmul(x[9], w16<2>());   x[13] = cmul(x[13], w16<3>());
    x[6] = cmul(x[6], w16<2>());  /* x[10] *= -i: in its butterfly */ x[14] = cmul(x[14], w16<6>());
    x[7] = cmul(x[7], w16<3>());  x[11] = cmul(x[11], w16<6>()); x[15] = cmul(x[15], w16<9>());
    radix4(x[0], x[1], x[2], x[3]);                                                // x[4q + k2] = X[q + 4 k2]
    radix4(x[4], x[5], x[6], x[7]);
    radix4<true>(x[8], x[9], x[10], x[11]);
    radix4(x[12], x[13], x[14], x[15]);
    // transpose the 4 x 4 index so that x[k] = X[k]
    cplx t;
#define PPG_SWAP(a, b) t = x[a]; x[a] = x[b]; x[b] = t;
    PPG_SWAP(1, 4) PPG_SWAP(2, 8) PPG_SWAP(3, 12) PPG_SWAP(6, 9) PPG_SWAP(7, 13) PPG_SWAP(11, 14)
#undef PPG_SWAP
}

__device__ __forceinline__ int pad(int i) { return i + (i >> 4); }

// LDS exchanges between the lanes of one wave.  The writes must have COMPLETED before another lane's read of them
// is issued: lgkmcnt(0).  (Round 2 relied on "a wave's LDS operations execute in order" and only stopped the
// compiler from moving them; that holds on a quiet CU and fails beside another kernel's LDS traffic -- a frame
// pair whose conjugate-symmetry split read the partner bins of the previous pass, HISTORY.md, round 2.)
__device__ __forceinline__ void wave_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// SPEC: also store the linear magnitudes (tests, ppg_frontend with a spectrogram pointer); its 18 extra row
// addresses per lane cost the product instantiation ~200 spilled registers when it was a run-time branch
//
// A workgroup is TWO such four-wave teams (512 threads, 2 x 79.5 KiB of LDS = the whole CU): the occupancy of two
// 256-thread workgroups per CU, and no other kernel's workgroup can share the CU.  History: as one-team workgroups
// (round 2), beside the encoder's kernels on another stream (attention's 64 KiB workgroups fit next to one 80 KiB
// workgroup), a frame pair's transform came out wrong about once per 300 pairs -- the gfx950 packed-fp32 / MFMA
// interaction of DESIGN 4.4, since removed at its root (no v_pk_*_f32 with op_sel on source 1 in any kernel:
// tools/pk_scan.py audits the built library).  The two teams share nothing but the barriers.
template <bool SPEC>
__global__ __launch_bounds__(256 * TEAMS, 2 / TEAMS) void frontend_kernel(
    ppg::FrontendTables tb, const float* __restrict__ audio, int samples, int frames,
    int groups_per_row, int total_groups, int wide_ok, __half* __restrict__ spec, __half* __restrict__ mel)
{
    extern __shared__ __attribute__((aligned(16))) char smem_all[];
    const int team = TEAMS == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
    char* smem = smem_all + team * LDS_BYTES;
    cplx* tw = reinterpret_cast<cplx*>(smem + OFF_TW);
    float* seg0 = reinterpret_cast<float*>(smem + OFF_SEG);

    const int tid = threadIdx.x & 255;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    cplx* buf = reinterpret_cast<cplx*>(smem + OFF_FFT) + wave * BUF;

    if (tid < 64) { const float2 w = tb.twiddle[16 * tid]; tw[tid] = cplx{w.x, w.y}; }
    // bins 512..543 of every magnitude row: bin 512 is rewritten per pair, the rest stays zero
    // (the filterbank steps read whole 32-bin chunks)
    reinterpret_cast<uint32_t*>(smem + OFF_MAG + (tid >> 4) * MROW + 1024)[tid & 15] = 0u;

    // per-lane constants of the persistent loop: the window at the lane's 16
    // sample positions and the pass-1 twiddles W^(l k) (LDS bandwidth is what
    // bounds the transform; these were 12 KiB of LDS reads per frame pair)
    float hreg[16];
    cplx t1[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) hreg[j] = tb.hann[lane + 64 * j];
#pragma unroll
    for (int k = 1; k < 16; ++k) { const float2 w = tb.twiddle[lane * k]; t1[k] = cplx{w.x, w.y}; }

    // this wave's share of the banded filterbank: NS steps of (weight fragment pair, first bin,
    // filter block finished by the step or -1), kept in scalar registers
    int step_frag[NS], step_off[NS], step_block[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const int4 e = tb.mel_prog[wave * NS + i];
        step_frag[i] = __builtin_amdgcn_readfirstlane(e.x);
        step_off[i] = __builtin_amdgcn_readfirstlane(e.y);
        step_block[i] = __builtin_amdgcn_readfirstlane(e.z);
    }

    // Samples of a group: global -> LDS DMA issued in the middle of the previous group's transforms into
    // the other half of the sample buffer; no registers, no LDS stores.  A group that lies inside its row
    // (all but the first and last few of a row) moves 1 KiB per wave instruction from a wave-uniform base;
    // a group that reaches into the reflect padding moves 64 samples per instruction, each lane with its
    // own source address -- the padding is an address computation, and a sample that does not exist even
    // after reflection reads hann[0] = 0.
    auto stage = [&](int grp, int into) {
        const int b = grp / groups_per_row;
        const int f0 = (grp - b * groups_per_row) * FPB;
        const float* arow = audio + (size_t)b * samples;
        const uint32_t dst = lds_addr(seg0) + into * (SEGP * 4);
        const int first = f0 * HOP - PADR;
        if ((wide_ok & 1) && first >= 0 && first + SEGP <= samples) {
#pragma unroll
            for (int j = 0; j < (SEGP / 256 + 3) / 4; ++j) {
                const int piece = wave + 4 * j;
                if (piece >= SEGP / 256) break;
                glds16_saddr(reinterpret_cast<const char*>(arow + first), (uint32_t)(piece * 1024 + lane * 16), dst + piece * 1024);
            }
            return;
        }
#pragma unroll 2
        for (int piece = wave; piece < SEGP / 64; piece += 4) {
            // reflect-padded segment: padded index i -> source i - 432
            int src = first + piece * 64 + lane;
            if (src < 0) src = -src;
            if (src >= samples) src = 2 * (samples - 1) - src;
            const float* p = (src >= 0 && src < samples) ? arow + src : tb.hann;
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off"
                         :: "v"(p), "s"(__builtin_amdgcn_readfirstlane(dst + piece * 256)) : "memory", "m0");
        }
    };
    const int stride = TEAMS * gridDim.x;
    const int first_group = TEAMS * blockIdx.x + team;
    if (first_group < total_groups) stage(first_group, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

#ifdef PPG_FE_TIMING
    int stamp_group = 0;
    auto stamp = [&](int k) {
        if (tb.dbg && blockIdx.x == 0 && lane == 0 && stamp_group == 2) tb.dbg[wave * 16 + k] = __builtin_amdgcn_s_memtime();
    };
#else
    auto stamp = [&](int) {};
#endif
    int half = 0;
    // (both teams make the same number of trips -- the barriers are the workgroup's; a team past its last group
    // runs an empty trip: every frame of it lies past the row's end, nothing is transformed or stored)
    const int trips = (total_groups - TEAMS * (int)blockIdx.x + stride - 1) / stride;
    int grp = first_group;
    for (int trip = 0; trip < trips; ++trip, grp += stride, half ^= 1) {
        const bool active = grp < total_groups;
        const int b = active ? grp / groups_per_row : 0;
        const int f0 = active ? (grp - b * groups_per_row) * FPB : frames;
        const float* seg = seg0 + half * SEGP;
        // (row pitch of the outputs, hidden from loop-invariant code motion: the compiler otherwise
        // keeps -- and spills -- one 64-bit offset per output row a lane may ever store to)
        int pitch = frames;
        asm volatile("" : "+s"(pitch));
#ifdef PPG_FE_TIMING
        ++stamp_group;
#endif
        stamp(0);
        __syncthreads();                 // this group's samples are in place; the previous group's magnitudes are consumed
        stamp(1);

        // one frame pair (fa, fa + 1) = pair j of the group
        auto transform = [&](const int j, const bool verify = false) {
            const int fa = f0 + 2 * j;
            if (fa >= frames) return;
            cplx x[16];
            // pass 1: radix 16, stride 1 -- lane l owns samples l + 64 j of both frames
            {
                const float* sa = seg + (2 * j) * HOP + lane;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    x[i] = cplx{sa[64 * i], sa[64 * i + HOP]} * hreg[i];
                }
                dft16(x);
#pragma unroll
                for (int k = 1; k < 16; ++k) x[k] = cmul(x[k], t1[k]);
#pragma unroll
                for (int k = 0; k < 16; ++k) buf[17 * lane + k] = x[k];         // pad(16 l + k)
            }
            wave_sync();
            // Exchange-buffer addressing: index i lives at pad(i) = i + (i >> 4).  All
            // strides below are multiples of 16, so every address is a lane base plus
            // a compile-time offset (ds immediates instead of VALU).
            const int p0 = lane + (lane >> 4);                     // pad(lane)
            // pass 2: radix 16, stride 16 -- p = l / 16, q = l % 16
            {
#pragma unroll
                for (int i = 0; i < 16; ++i) x[i] = buf[p0 + 68 * i];           // pad(l + 64 i)
                dft16(x);
                const int p = lane >> 4, q = lane & 15;
#pragma unroll
                for (int k = 1; k < 16; ++k) x[k] = cmul(x[k], tw[p * k]);       // W1024^(16 p k)
                cplx* dst = buf + q + 272 * p;                                  // pad(q + 256 p + 16 k) = q + 272 p + 17 k
#pragma unroll
                for (int k = 0; k < 16; ++k) dst[17 * k] = x[k];
            }
            wave_sync();
            // pass 3: radix 4, stride 256, no twiddles; lane l owns q = l + 64 m
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                cplx* z = buf + p0 + 68 * m;                                    // pad(q + 256 j) = pad(q) + 272 j
                cplx A = z[0], B = z[272], C = z[544], D = z[816];
                radix4(A, B, C, D);
                z[0] = A; z[272] = B; z[544] = C; z[816] = D;
            }
            wave_sync();
            // buf holds Z = FFT(a + i b); split: A[k] = (Z[k] + conj Z[N-k]) / 2,
            // B[k] = (Z[k] - conj Z[N-k]) / (2i);  k = l + 64 r, N - k = (1024 - l) - 64 r.
            // Magnitudes as fp16: lane pairs exchange (DPP) so that the even lane holds bins
            // (k, k + 1) of frame a and the odd lane bins (k - 1, k) of frame b -- one dword each.
            const int nb = (NFFT - lane) + ((NFFT - lane) >> 4);   // pad(1024 - l); lane 0, r = 0 wraps to Z[0]
            const int odd = lane & 1;
            const int row = j + 8 * odd;
            char* wdst = smem + OFF_MAG + row * MROW + ((((lane >> 3) ^ sigma(row >> 2)) << 2) + ((lane >> 1) & 3)) * 4;
            const uint32_t sel = odd ? 0x07060302u : 0x01000504u;
            auto split_bin = [&](int r, __half& ha, __half& hb) {
                const int k = lane + 64 * r;
                const cplx zk = buf[p0 + 68 * r];
                const cplx zn = buf[(r == 0 && lane == 0) ? 0 : nb - 68 * r];
                const float ar = 0.5f * (zk.x + zn.x), ai = 0.5f * (zk.y - zn.y);
                const float br = 0.5f * (zk.y + zn.y), bi = -0.5f * (zk.x - zn.x);
                // arguments are >= 1e-6: the bare v_sqrt_f32 (1 ulp) needs none of sqrtf's
                // denormal handling, and the result is rounded to fp16 next
                ha = __float2half_rn(__builtin_amdgcn_sqrtf(ar * ar + ai * ai + 1e-6f));
                hb = __float2half_rn(__builtin_amdgcn_sqrtf(br * br + bi * bi + 1e-6f));
                if constexpr (SPEC) {
                    if (verify) return;
                    spec[(uint32_t)(b * NBINS + k) * (uint32_t)pitch + (uint32_t)fa] = ha;
                    if (fa + 1 < frames) spec[(uint32_t)(b * NBINS + k) * (uint32_t)pitch + (uint32_t)fa + 1u] = hb;
                }
            };
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                __half ha, hb;
                split_bin(r, ha, hb);
                const uint32_t pk = (uint32_t)__half_as_ushort(ha) | ((uint32_t)__half_as_ushort(hb) << 16);
                const uint32_t other = (uint32_t)__builtin_amdgcn_mov_dpp((int)pk, 0xB1, 0xf, 0xf, true);   // lane ^ 1
                const uint32_t val = __builtin_amdgcn_perm(pk, other, sel);
                *reinterpret_cast<uint32_t*>(wdst + 128 * r) = val;
            }
            if (verify) return;
            if (lane == 0) {                                        // k = 512
                __half ha, hb;
                split_bin(8, ha, hb);
                *reinterpret_cast<uint32_t*>(smem + OFF_MAG + j * MROW + ((64 ^ sigma(j >> 2)) << 4)) = __half_as_ushort(ha);
                *reinterpret_cast<uint32_t*>(smem + OFF_MAG + (j + 8) * MROW + ((64 ^ sigma((j + 8) >> 2)) << 4)) = __half_as_ushort(hb);
            }
        };

        transform(wave);
        stamp(2);
        if (!(wide_ok & 4) && grp + stride < total_groups) stage(grp + stride, half ^ 1);    // free since the barrier above: it held the previous group
        stamp(3);
        transform(wave + 4);
        stamp(4);
        // the wave's weight fragments (40 KB for all four waves, L2 / L1 resident): requested here, they
        // arrive while the workgroup gathers at the barrier -- held across the transforms they would
        // push the kernel past the 256 registers of two workgroups per CU
        u32x4 whi[NS], wlo[NS];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's share of the next group's samples has landed
        if (mel) {
            uint32_t loff = lane * 16;
            asm volatile("" : "+v"(loff));       // (the loads are loop-invariant: keep them here)
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const char* src = reinterpret_cast<const char*>(tb.mel_img) + (size_t)step_frag[i] * 1024;
                whi[i] = *reinterpret_cast<const u32x4*>(src + loff);
                wlo[i] = *reinterpret_cast<const u32x4*>(src + 1024 + loff);
            }
        }
        __syncthreads();                 // all 16 magnitude rows written
        stamp(5);
        if (mel) {
            const int n = lane & 15, g = lane >> 4;
            const char* brow = smem + OFF_MAG + n * MROW + ((g ^ sigma(n >> 2)) << 4);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const int frame = f0 + (n < 8 ? 2 * n : 2 * (n - 8) + 1);
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const u32x4 bfi = *reinterpret_cast<const u32x4*>(brow + step_off[i]);
                PrecF16::mma(acc, whi[i], bfi);
                PrecF16::mma(acc, wlo[i], bfi);
                const int block = step_block[i];
                if ((i == MSEG - 1 || i == NS - 1) && block >= 0) {
                    // C layout: column n = lane & 15 (frame), rows 4 g + e (filters of the block)
                    if (frame < frames) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int m = 16 * block + 4 * g + e;
                            mel[(uint32_t)(b * NMELS + m) * (uint32_t)pitch + (uint32_t)frame] =
                                __float2half_rn(logf(fmaxf(acc[e] * (1.0f / 65536.0f), 1e-5f)));
                        }
                    }
                    acc = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
        stamp(6);
        if (wide_ok & 4) {               // (debug: the next group's samples staged with nothing else in flight)
            __syncthreads();
            if (grp + stride < total_groups) stage(grp + stride, half ^ 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
}

}  // namespace

namespace ppg {

hipError_t launch_frontend(const FrontendTables& tb, const float* audio, int batch, int samples,
                           void* spec, void* mel, hipStream_t s) {
    const int frames = samples / HOP;
    const int groups_per_row = (frames + FPB - 1) / FPB;
    if (mel && (!tb.mel_img || !tb.mel_prog)) return hipErrorInvalidValue;
    if ((double)batch * NBINS * frames >= 4294967296.0) return hipErrorInvalidValue;    // the kernel indexes its outputs with 32 bits
    const int total = groups_per_row * batch;
    static LdsLimit limit[2];
    const void* kernel = spec ? reinterpret_cast<const void*>(frontend_kernel<true>) : reinterpret_cast<const void*>(frontend_kernel<false>);
    const hipError_t e = limit[spec != nullptr].ensure(kernel, CU_LDS_BYTES);
    if (e != hipSuccess) return e;
    static int slots = 0;                // resident teams: two per workgroup, one workgroup per CU (the devices of a node are alike)
    if (slots == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
        slots = 2 * cus;
    }
    // persistent grid: every team gets the same number of groups, +-1
    const int rounds = (total + slots - 1) / slots;
    const int teams = (total + rounds - 1) / rounds;
    const int grid = (teams + TEAMS - 1) / TEAMS;
    // 16-byte DMA pieces need 16-byte aligned rows
    int wide_ok = (reinterpret_cast<uintptr_t>(audio) % 16 == 0 && samples % 4 == 0) ? 1 : 0;
    // the product workgroup asks for ALL of the CU's LDS (not just its two teams' 2 x 79.5 KiB): a kernel that needs
    // <= 1 KiB of LDS would otherwise still fit beside it
    size_t lds_bytes = TEAMS == 2 ? CU_LDS_BYTES : LDS_BYTES;
    static const int fe_debug = env_experiment("PPGS_AMD_FE_DEBUG", 0);   // 1: narrow DMA only, 4: serial staging, 16: the exact LDS request
    if (fe_debug & 1) wide_ok = 0;
    if (fe_debug & 4) wide_ok |= 4;
    if (fe_debug & 16) lds_bytes = TEAMS * LDS_BYTES;   // (small-LDS kernels may then co-reside)
    if (spec)
        hipLaunchKernelGGL(frontend_kernel<true>, dim3(grid), dim3(256 * TEAMS), lds_bytes, s, tb, audio, samples, frames,
                           groups_per_row, total, wide_ok, reinterpret_cast<__half*>(spec), reinterpret_cast<__half*>(mel));
    else
        hipLaunchKernelGGL(frontend_kernel<false>, dim3(grid), dim3(256 * TEAMS), lds_bytes, s, tb, audio, samples, frames,
                           groups_per_row, total, wide_ok, reinterpret_cast<__half*>(spec), reinterpret_cast<__half*>(mel));
    return hipGetLastError();
}

}  // namespace ppg
