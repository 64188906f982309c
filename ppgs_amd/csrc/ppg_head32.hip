// The head of the encoder for gfx950, one kernel per 160-token tile: window gather (reference
// ppgs/model/transformer.py:49-64: chunks with replicate padding on the left), input convolution k = 5 with the
// positional encoding and the padding mask (transformer.py:26-47 via SURVEY.md 2.3 rows M1/M2) and layer 0's
// in_proj (row M4) -- what used to be three launches (gather, linear_kernel<EPI_INCONV>, linear_kernel<EPI_QKV>)
// with the gathered rows, the 16-bit copy of x and its re-read in between.  Same machinery as ppg_layer32.hip:
// wave w owns features 64 w .. 64 w + 63, weights are A fragments the host packed in consumption order, the
// activations are B fragments from LDS:
//   1. the tile's rows (2 halo rows either side) of the (B, C, T) features -> LDS, token-major, 96 channels of 16 bits
//      in rows of 208 bytes (13 x 16: the 32 rows a fragment read touches fall in 16 different 16-byte bank groups);
//   2. x[feature][token] = sum over (tap, channel block) W fragment x rows (token + tap - 2): the B fragment of tap t
//      is the same LDS tile read one row further on; taps that leave the token's window are zeroed per lane (only
//      token blocks that contain a window edge pay for the select);
//   3. + bias, + PE, zero past the valid frames -> X (fp32, X32 order) and, packed, the token panel;
//   4. the Q/K/V projection of ppg_layer32.h on the panel.
// Workgroups past the tiles zero the scratch areas nobody writes but the attention tiles read (see GatherArgs).
#include "ppg_layer32.h"

namespace {

constexpr int HID = 256;
constexpr int CP = 96;                 // padded input channels
constexpr int ROWB = 208;              // bytes per gathered row in LDS
constexpr int HALO = 2;
constexpr int KSI = 5 * CP / 16;       // K-steps of the convolution: (tap, 16-channel block)
constexpr int KSH = KSI / 2;           // per register set of fragments

// step i = (ks = K0 + i / TBN, tb = i % TBN): rows 32 tb + tap .. of the gathered tile, channels 16 cb ..
template <int K0, int TBN>
struct OffIn {
    static constexpr int at(int i) {
        const int ks = K0 + i / TBN, tb = i % TBN;
        return (32 * tb + ks / (CP / 16)) * ROWB + (ks % (CP / 16)) * 32;
    }
};

template <class P, int TBS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void head32_kernel(Head32Args a) {
    using G = Geo<HID, TBS>;
    constexpr int RB = G::RB, KS = G::KS, TB = G::TBN, TOKS = G::TOKS;
    // sub-tile workgroups (TBS = 2: three per 160-token tile, as in ppg_layer32.hip): workgroup b takes token blocks
    // tb0 .. of tile b / NSUB; a block past the tile's end is the next tile's first rows, computed and not stored
    constexpr int TBT = tile_blocks(HID);
    constexpr bool SUB = TBS != TBT;
    constexpr int NSUB = (TBT + TBS - 1) / TBS;
    static_assert(RB == 2 && KSI % 2 == 0, "two row blocks per wave, two register sets");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= a.tiles * NSUB) {
        // scratch areas: the absent half of a window's last 32-column V^T group (columns are permuted inside a group:
        // position 8 g + 4 e + r for token 16 e + 4 g + r, so the absent 16 tokens are the pieces 8 g + 4 .. + 7; the
        // present half is written by the tiles, byte-disjoint), the slack behind the last V^T column, the q | k rows
        // behind the last token
        for (int job = blockIdx.x - a.tiles * NSUB; job <= a.nwin + 1; job += gridDim.x - a.tiles * NSUB) {
            if (job == a.nwin + 1) {
                for (int i = tid; i < a.qk_slack_bytes / 16; i += 256) reinterpret_cast<uint4*>(a.qk_slack)[i] = make_uint4(0u, 0u, 0u, 0u);
            } else if (job == a.nwin) {
                const int chunks = (a.vt_ld - a.vt_tokens) * 2 / 16;
                for (int i = tid; i < a.vt_rows * chunks; i += 256) {
                    const int r = i / chunks, c = i - r * chunks;
                    *reinterpret_cast<uint4*>(a.vt_out + ((size_t)r * a.vt_ld + a.vt_tokens) * 2 + c * 16) = make_uint4(0u, 0u, 0u, 0u);
                }
            } else {
                const PpgWindow w = a.win[job];
                const int r16 = (w.frames + 15) & ~15, r32 = (w.frames + 31) & ~31;
                if (r32 != r16) {
                    const int col = w.vt_off + r32 - 32;
                    for (int i = tid; i < a.vt_rows * 4; i += 256) {
                        const int r = i >> 2, g = i & 3;
                        *reinterpret_cast<uint2*>(a.vt_out + ((size_t)r * a.vt_ld + col + 8 * g + 4) * 2) = make_uint2(0u, 0u);
                    }
                }
            }
        }
        return;
    }
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tok = lane & 31, hh = lane >> 5;
    const int tile = SUB ? (int)blockIdx.x / NSUB : (int)blockIdx.x;
    const int tb0 = SUB ? ((int)blockIdx.x % NSUB) * TBS : 0;
    const int nblk = SUB ? min(TBS, TBT - tb0) : TBS;
    const int m0 = tile * (32 * TBT) + 32 * tb0;
    const uint32_t lds0 = lds_addr32(smem);
    const uint32_t voff = lane * 16;
    const uint32_t pb0 = lds0 + G::L_ACT + lane * 16;
    const int fbase = 32 * RB * wave;
    char* rows = smem + G::L_H;                     // gathered rows: (TOKS + 2 HALO) x ROWB bytes in the h region
    static_assert((TOKS + 2 * HALO) * ROWB <= TB * 8 * 1024, "the gathered tile fits the h region");
    float* bq_lds = reinterpret_cast<float*>(smem + G::L_BQ);

#ifdef PPG_H32_TIMING
    auto pstamp = [&](int k) {
        if (!SUB && a.dbg && blockIdx.x == 0 && lane == 0) a.dbg[wave * 16 + k] = __builtin_amdgcn_s_memtime();
    };
#else
    auto pstamp = [&](int) {};
#endif
    pstamp(0);
    u32x4 w1f[16], w2f[16];
    auto load16 = [&](u32x4 (&wf)[16], const char* base) {
        [&]<int... K>(std::integer_sequence<int, K...>) { (gload_frag<K>(wf[K], voff, base), ...); }(std::make_integer_sequence<int, 16>{});
    };
    const char* wimg = a.win_img + ((size_t)wave * RB * KSI) * 1024;
    const char* wq0 = a.wq_img + ((size_t)wave * 3 * RB * KS) * 1024;
    // the convolution's first-round weight fragments travel under the gather (nothing below touches the two register
    // sets before the round's wait: tools/asm_load_scan.py checks the built code)
    load16(w1f, wimg);
    load16(w2f, wimg + (size_t)KSI * 1024);

    // ---- 1. gather: thread r < TOKS + 2 HALO takes row r = token m0 - HALO + r, all channels (for one channel,
    // neighbouring threads read neighbouring frames)
    if (tid < TOKS + 2 * HALO && !(PPG_DBG(a) & 4)) {
        const int m = m0 - HALO + tid;
        int frame = -1;
        size_t base = 0;
        if (m >= 0) {
            const TokMeta tm = tok_meta(a.blk_win, a.win, m, a.M);
            if (tm.w >= 0 && tm.tt < tm.frames) {
                const PpgWindow w = a.win[tm.w];
                frame = w.chunked ? max(w.start + tm.tt - a.overlap, 0) : tm.tt;
                base = (size_t)w.item * a.C * a.T;
            }
        }
        // every load is issued unconditionally (a dead row reads frame 0 of item 0, a padding channel re-reads the last
        // one) and masked afterwards: predicated loads came out as 96 branches with a wait each, 51 k cycles of this
        // kernel's 138 k
        uint32_t* dst = reinterpret_cast<uint32_t*>(rows + tid * ROWB);
        const int C = a.C;
        const bool row_live = frame >= 0;
        const size_t off = base + (size_t)max(frame, 0);
        float v[CP];
        if (a.dtype == PPG_DTYPE_F16) {
            const __half* src = reinterpret_cast<const __half*>(a.feats) + off;
#pragma unroll
            for (int c = 0; c < CP; ++c) v[c] = __half2float(src[(size_t)min(c, C - 1) * a.T]);
        } else {
            const float* src = reinterpret_cast<const float*>(a.feats) + off;
#pragma unroll
            for (int c = 0; c < CP; ++c) v[c] = src[(size_t)min(c, C - 1) * a.T];
        }
#pragma unroll
        for (int cp = 0; cp < CP / 2; ++cp)
            dst[cp] = P::pack2(row_live && 2 * cp < C ? v[2 * cp] : 0.f, row_live && 2 * cp + 1 < C ? v[2 * cp + 1] : 0.f);
    }
    pstamp(1);
    // layer 0's in_proj bias for the tail
    for (int i = tid; i < 3 * HID / 4; i += 256) reinterpret_cast<float4*>(bq_lds)[i] = reinterpret_cast<const float4*>(a.bq)[i];

    // per token of this lane: window position, and the tap offsets (-2 .. 2) that stay inside the window
    bool live[TB], valid[TB];
    int tt[TB], dlo[TB], dhi[TB];
    unsigned edge[TB];                               // wave-uniform: bit tap = some lane's tap leaves its window
#pragma unroll
    for (int t = 0; t < TB; ++t) {
        const TokMeta tm = tok_meta(a.blk_win, a.win, m0 + 32 * t + tok, a.M);
        live[t] = tm.w >= 0 && tm.tt < tm.frames;
        valid[t] = live[t] && tm.tt < tm.valid;
        tt[t] = live[t] ? tm.tt : 0;
        dlo[t] = live[t] ? -tm.tt : -HALO;
        dhi[t] = live[t] ? tm.frames - 1 - tm.tt : HALO;
        unsigned bits = 0;
#pragma unroll
        for (int tap = 0; tap < 5; ++tap)
            if (__any(tap - HALO < dlo[t] || tap - HALO > dhi[t])) bits |= 1u << tap;
        edge[t] = bits;
    }

    pstamp(2);
    // ---- 2. input convolution: y[rb][tb] = W[rows of this wave] x rows, two rounds of KSH K-steps, the fragments
    // of the wave's two row blocks in the two register sets (image order [wave][rb][KSI])
    f32x16 yacc[RB][TB];
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const uint32_t rb0 = lds0 + G::L_H + (uint32_t)(tok * ROWB + hh * 16);
    // PE rows of the lane's tokens, features f0 .. f0 + 15 of row block rb: fetched a phase ahead of their use
    float4 pe[TB][4];
    auto load_pe = [&](int rb) {
#pragma unroll
        for (int t = 0; t < TB; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                pe[t][q] = *reinterpret_cast<const float4*>(a.pe + (size_t)tt[t] * HID + fbase + 32 * rb + 16 * hh + 4 * q);
    };
    auto round = [&](auto r_tag, auto after_wait) {
        constexpr int R = decltype(r_tag)::value;
        // (16 fragments are fetched where KSH = 15 are used: the image carries one pad fragment at its end)
        vm_wait_all(w1f);
        vm_wait_all(w2f);
        after_wait();
        if constexpr (R == 0) __syncthreads();               // the gathered rows are in LDS
        stream<OffIn<KSH * R, TB>, KSH * TB, 6>(rb0, rb0, [&](auto ic, const u32x4& bf) {
            constexpr int i = decltype(ic)::value;
            constexpr int ksl = i / TB, tb = i % TB, tap = (KSH * R + ksl) / (CP / 16);
            u32x4 b = bf;
            if constexpr (tap != HALO) {
                if (edge[tb] & (1u << tap)) {
                    const bool ok = tap - HALO >= dlo[tb] && tap - HALO <= dhi[tb];
                    b = u32x4{ok ? bf.x : 0u, ok ? bf.y : 0u, ok ? bf.z : 0u, ok ? bf.w : 0u};
                }
            }
            if constexpr (R == 0 && ksl == 0) {
                yacc[0][tb] = P::mma32(w1f[0], b, zero);
                yacc[1][tb] = P::mma32(w2f[0], b, zero);
            } else {
                yacc[0][tb] = P::mma32(w1f[ksl], b, yacc[0][tb]);
                yacc[1][tb] = P::mma32(w2f[ksl], b, yacc[1][tb]);
            }
            // the second round's fragment ksl is requested behind the first round's last use of its registers
            if constexpr (R == 0 && tb == TB - 1) {
                gload_frag<ksl>(w1f[ksl], voff, wimg + (size_t)KSH * 1024);
                gload_frag<ksl>(w2f[ksl], voff, wimg + (size_t)(KSI + KSH) * 1024);
            }
        });
    };
    if (!(PPG_DBG(a) & 1)) round(std::integral_constant<int, 0>{}, [] {});
    else __syncthreads();
    pstamp(3);
    // (the PE rows of the first row block are requested behind the round's weight wait and land under its MFMAs)
    if (!(PPG_DBG(a) & 1)) round(std::integral_constant<int, 1>{}, [&] { load_pe(0); });
    else load_pe(0);

    pstamp(4);
    // ---- 3. x = live ? PE[tt] + (valid ? y + bias : 0) : 0 (as linear_kernel<EPI_INCONV>) -> X32 and the panel
    float* xt = a.X + ((size_t)tile * 4 + wave) * (TBT * RB * 4 * 256) + lane * 4;
    char* xh = reinterpret_cast<char*>(a.X) + ((size_t)tile * 4 + wave) * (TBT * RB * 2 * 1024) + lane * 16;   // X16 order
    auto emit = [&](auto rb_tag) {
        constexpr int rb = decltype(rb_tag)::value;
        const int f0 = fbase + 32 * rb + 16 * hh;            // the lane's 16 consecutive features of the block
        float4 bias[4], pv[TB][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) bias[q] = *reinterpret_cast<const float4*>(a.b_in + f0 + 4 * q);
#pragma unroll
        for (int t = 0; t < TB; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) pv[t][q] = pe[t][q];
        if constexpr (rb + 1 < RB) load_pe(rb + 1);          // the next block's rows travel under this one's stores
#pragma unroll
        for (int t = 0; t < TB; ++t) {
            f32x16 y;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                y[4 * q + 0] = live[t] ? pv[t][q].x + (valid[t] ? yacc[rb][t][4 * q + 0] + bias[q].x : 0.f) : 0.f;
                y[4 * q + 1] = live[t] ? pv[t][q].y + (valid[t] ? yacc[rb][t][4 * q + 1] + bias[q].y : 0.f) : 0.f;
                y[4 * q + 2] = live[t] ? pv[t][q].z + (valid[t] ? yacc[rb][t][4 * q + 2] + bias[q].z : 0.f) : 0.f;
                y[4 * q + 3] = live[t] ? pv[t][q].w + (valid[t] ? yacc[rb][t][4 * q + 3] + bias[q].w : 0.f) : 0.f;
                if (!a.x_half && t < nblk)
                    *reinterpret_cast<float4*>(xt + (((tb0 + t) * RB + rb) * 4 + q) * 256) = make_float4(y[4 * q + 0], y[4 * q + 1], y[4 * q + 2], y[4 * q + 3]);
            }
            if (a.x_half && t < nblk) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
                    *reinterpret_cast<u32x4*>(xh + (((tb0 + t) * RB + rb) * 2 + s2) * 1024) = u32x4{pack_f16x2(y[8 * s2 + 0], y[8 * s2 + 1]), pack_f16x2(y[8 * s2 + 2], y[8 * s2 + 3]),
                                                                                      pack_f16x2(y[8 * s2 + 4], y[8 * s2 + 5]), pack_f16x2(y[8 * s2 + 6], y[8 * s2 + 7])};
            }
            panel_store<P, HID, TBS>(pb0, wave, t, rb, y);
        }
    };
    emit(std::integral_constant<int, 0>{});
    emit(std::integral_constant<int, 1>{});
    pstamp(5);
    // layer 0's W_qkv fragments of the tail's first half-step (not earlier: requested during the second round, hipcc
    // copies the registers into the accumulation file inside the epilogue above -- its code -- before the loads have
    // landed, and hands them to the LDS ring: tools/asm_load_scan.py shows it)
    load16(w1f, wq0);
    vm_wait_all(w1f);
    pstamp(6);

    // ---- 4. layer 0's Q/K/V
    Layer32Args la{};
    la.wq_img = a.wq_img; la.qk_out = a.qk_out; la.vt_out = a.vt_out; la.vt_ld = a.vt_ld;
    la.blk_win = a.blk_win; la.win = a.win; la.M = a.M; la.H = HID;
    if (!(PPG_DBG(a) & 2)) qkv_tail<P, HID, TBS>(la, smem, m0, w1f, w2f, nblk);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    pstamp(7);
}

}  // namespace

namespace ppg {

hipError_t launch_head32(int precision, const Head32Args& a, hipStream_t s) {
    if (a.H != HID || a.C > CP || a.M <= 0 || a.tiles != (a.M + 32 * tile_blocks(HID) - 1) / (32 * tile_blocks(HID))) return hipErrorInvalidValue;
    auto launch = [&](auto kern, auto geo, int subs) {
        using G = decltype(geo);
        const size_t lds = G::L_B1;
        const dim3 grid(a.tiles * subs + std::min(a.nwin + 2, 32));
        static ppg::LdsLimit limit;
        const hipError_t e = limit.ensure(reinterpret_cast<const void*>(kern), lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
        return hipGetLastError();
    };
    constexpr int T5 = tile_blocks(HID);
    if (precision == PPG_PRECISION_BF16) return a.sub_tiles ? launch(head32_kernel<PrecBF16, 2>, Geo<HID, 2>{}, 3) : launch(head32_kernel<PrecBF16, T5>, Geo<HID>{}, 1);
    if (precision == PPG_PRECISION_FP16) return a.sub_tiles ? launch(head32_kernel<PrecF16, 2>, Geo<HID, 2>{}, 3) : launch(head32_kernel<PrecF16, T5>, Geo<HID>{}, 1);
    return hipErrorInvalidValue;
}

}  // namespace ppg
