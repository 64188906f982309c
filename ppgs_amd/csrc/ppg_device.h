// Shared device-side definitions for the gfx950 kernels of the PPG engine.
//
// Orientation used by every matrix kernel here ("features on M, tokens on N"):
//   out^T[n][tok] = sum_k W[n][k] * act[tok][k]
// The weight is the MFMA A operand (16 feature rows per block, read from an
// LDS tile shared by the workgroup's waves), the activation is the B operand
// (16 tokens per block, held in registers by the wave that owns the tokens).
// Both operands are K-contiguous in memory (PyTorch Linear layout W[out][in]
// and token-major activations), so every fragment is ONE 16-byte read:
//
//   lane l:  idx = l & 15 (feature row of A / token of B),  g = l >> 4
//   fragment = the 16 bytes at  [idx][kgroup*64 + g*16]   ("K-group" = 64 B)
//
//   bf16: 8 elements = k-slots 8g..8g+7 of one v_mfma_f32_16x16x32_bf16
//   fp32: 4 elements, element s feeds v_mfma_f32_16x16x4_f32 number s, whose
//         k-slot g then stands for k = 4g+s (A and B use the same bijection,
//         so the 4 MFMAs together contract the 16 k of the group exactly once)
//
// The accumulator of either shape is  C[row = 4g + r][col = idx], r = 0..3:
// a lane holds 4 consecutive FEATURES of one TOKEN -> 8/16-byte row-major
// stores, and per-token reductions (LayerNorm, softmax over the 40 phonemes)
// are a per-lane partial + two wavefront shuffles (xor 16, 32).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "../../include/ppgs_amd.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;   // one 16-byte MFMA operand fragment
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

__device__ __forceinline__ uint16_t f32_to_bf16_rne(float f) {
    uint32_t u = __float_as_uint(f);
    // NaN stays NaN; everything else round-to-nearest-even on bit 16
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bf16_to_f32(uint16_t h) {
    return __uint_as_float(((uint32_t)h) << 16);
}
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
// two fp32 -> packed bf16 (round to nearest even): one v_cvt_pk_bf16_f32
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}

// max(x, 0) on two packed 16-bit floats (bf16 or fp16) in ONE VALU instruction
// (v_pk_max_i16): the sign bit makes every negative value, -0 included, a
// negative integer; NaNs with the sign bit clear pass through like fmaxf's.
typedef __attribute__((ext_vector_type(2))) short s16x2;
__device__ __forceinline__ uint32_t relu_packed16(uint32_t packed) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, packed), s16x2{0, 0}));
}

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
// two fp32 -> packed fp16 (round to nearest even): one v_cvt_pk_f16_f32
__device__ __forceinline__ uint32_t pack_f16x2(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, f16x2));
}

// kIsBF16 reads "16-bit operands" (bf16 or fp16): it selects the layouts and
// epilogues both share; pack2 / cvt1 are the only places the formats differ.
struct PrecBF16 {
    typedef uint16_t elem;
    static constexpr int kBytes = 2;
    static constexpr int KG = 32;   // elements per 64-byte K-group
    static constexpr bool kIsBF16 = true;
    static constexpr bool kSplit = false;
    static __device__ __forceinline__ int row_byte(int n) { return n * 2; }      // byte offset of element n in its row
    static constexpr float kProbCeil = 1.0995116e12f;   // 2^40: attention re-bases its softmax shift above this (attn_body)
    static __device__ __forceinline__ uint32_t pack2(float lo, float hi) { return pack_bf16x2(lo, hi); }
    static __device__ __forceinline__ uint16_t cvt1(float v) { return f32_to_bf16_rne(v); }
    static __device__ __forceinline__ void mma(f32x4& acc, const u32x4& a, const u32x4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
            __builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    }
    // first MFMA of an accumulation: C = 0 inline constant, no zero-init pass
    static __device__ __forceinline__ void mma0(f32x4& acc, const u32x4& a, const u32x4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
            __builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    }
    // ... or C = another register quad (a bias: D and C are separate MFMA operands)
    static __device__ __forceinline__ void mmac(f32x4& acc, const u32x4& a, const u32x4& b, const f32x4& c) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
            __builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
    // ReLU on two packed bf16: as signed 16-bit integers every negative value (and -0) is < 0
    static __device__ __forceinline__ uint32_t relu2(uint32_t packed) { return relu_packed16(packed); }
    // 32x32x16: A = 32 rows x 16 k (lane: row l & 31, k 8 (l >> 5) ..+7), B likewise with columns,
    // C[row = 8 (i >> 2) + 4 (l >> 5) + (i & 3)][col = l & 31], i = 0..15
    static __device__ __forceinline__ f32x16 mma32(const u32x4& a, const u32x4& b, const f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
    // ... with the accumulator in ARCHITECTURAL registers (see ppg_layer32.hip, phase A: a kernel that may use all 512
    // registers gets the accumulation-register form for every builtin MFMA, and VALU consumers of a result then pay one
    // v_accvgpr_read per value).  Inline asm: the compiler inserts no wait states around it -- the caller keeps a result's
    // first VALU read at least two other MFMAs behind the MFMA that wrote it.
    static __device__ __forceinline__ void mma32v0(f32x16& d, const u32x4& a, const u32x4& b, const f32x16& c) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
    }
    static __device__ __forceinline__ void mma32v(f32x16& d, const u32x4& a, const u32x4& b) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
    }
};

struct PrecF16 {
    typedef uint16_t elem;
    static constexpr int kBytes = 2;
    static constexpr int KG = 32;
    static constexpr bool kIsBF16 = true;
    static constexpr bool kSplit = false;
    static __device__ __forceinline__ int row_byte(int n) { return n * 2; }
    static constexpr float kProbCeil = 1024.0f;         // 2^10 (fp16 ends at 2^16)
    static __device__ __forceinline__ uint32_t pack2(float lo, float hi) { return pack_f16x2(lo, hi); }
    static __device__ __forceinline__ uint16_t cvt1(float v) { return __builtin_bit_cast(uint16_t, (_Float16)v); }
    static __device__ __forceinline__ void mma(f32x4& acc, const u32x4& a, const u32x4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(
            __builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
    }
    static __device__ __forceinline__ void mma0(f32x4& acc, const u32x4& a, const u32x4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(
            __builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    }
    static __device__ __forceinline__ void mmac(f32x4& acc, const u32x4& a, const u32x4& b, const f32x4& c) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(
            __builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ uint32_t relu2(uint32_t packed) { return relu_packed16(packed); }
    static __device__ __forceinline__ f32x16 mma32(const u32x4& a, const u32x4& b, const f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ void mma32v0(f32x16& d, const u32x4& a, const u32x4& b, const f32x16& c) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
    }
    static __device__ __forceinline__ void mma32v(f32x16& d, const u32x4& a, const u32x4& b) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
    }
};

struct PrecF32 {
    typedef float elem;
    static constexpr int kBytes = 4;
    static constexpr int KG = 16;
    static constexpr bool kIsBF16 = false;
    static constexpr bool kSplit = false;
    static __device__ __forceinline__ int row_byte(int n) { return n * 4; }
    static constexpr float kProbCeil = 1.0995116e12f;
    static __device__ __forceinline__ uint32_t pack2(float, float) { return 0u; }   // never used: 32-bit layouts
    static __device__ __forceinline__ float cvt1(float v) { return v; }
    static __device__ __forceinline__ void mma(f32x4& acc, const u32x4& a, const u32x4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
    }
    static __device__ __forceinline__ void mma0(f32x4& acc, const u32x4& a, const u32x4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
    }
    static __device__ __forceinline__ void mmac(f32x4& acc, const u32x4& a, const u32x4& b, const f32x4& c) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
    }
    static __device__ __forceinline__ uint32_t relu2(uint32_t packed) { return packed; }   // never used: 32-bit layouts
};

// fp32 values as TWO fp16 halves, x = hi + lo (hi = fp16(x), lo = fp16(x - hi): 22 significand bits down to fp16's
// subnormal floor of 6e-8 -- the matrix pipe keeps fp16 subnormals, tools/denorm_probe.hip), and a product as THREE
// fp16 MFMAs with fp32 accumulation: a b ~ a_hi b_hi + a_hi b_lo + a_lo b_hi (the dropped a_lo b_lo is 2^-22 of the
// product).  Operand precision of fp32 at a third of the fp16 MFMA rate -- 5 x the rate of the f32-input MFMAs.
// Layout: the byte arithmetic of the fp32 path (4 bytes per element, 64-byte K-groups of "16 elements"), but a PAIR of
// K-groups holds 32 elements as [32 hi halves | 32 lo halves]: lane group g's 16 bytes of the first K-group are the
// fp16 MFMA fragment of k-slots 8g .. 8g + 7 of the hi plane, of the second K-group the same slots of the lo plane.
// Every MFMA call site walks K-groups in order and knows its index: an even K-group (hi of the LDS-side operand)
// meets the register-side operand's hi AND lo fragment, an odd one (lo) its hi fragment only (mma_kg below).
// Handovers between GEMMs inside a kernel (FFN h, attention P) and the V^T columns use the 16-bit path's slot
// orders (8 slots per lane from two accumulator blocks), with both planes.
struct PrecX2 {
    typedef float elem;
    static constexpr int kBytes = 4;
    static constexpr int KG = 16;
    static constexpr bool kIsBF16 = false;
    static constexpr bool kSplit = true;
    static constexpr float kProbCeil = 1024.0f;         // fp16 operands
    // byte offset of element n (a multiple of 4 or 8) of a row: the hi plane of its 32-element block; lo is 64 further
    static __device__ __forceinline__ int row_byte(int n) { return (n >> 5) * 128 + (n & 31) * 2; }
    static __device__ __forceinline__ uint32_t pack2(float lo, float hi) { return pack_f16x2(lo, hi); }
    // (a, b) -> packed fp16 pair of the hi plane and of the lo plane
    static __device__ __forceinline__ void split2(float a, float b, uint32_t& hi, uint32_t& lo) {
        hi = pack_f16x2(a, b);
        const f16x2 h = __builtin_bit_cast(f16x2, hi);
        lo = pack_f16x2(a - (float)h[0], b - (float)h[1]);
    }
    static __device__ __forceinline__ float cvt1(float v) { return v; }   // (never used: elements are written through split2)
    static __device__ __forceinline__ void mma(f32x4& acc, const u32x4& a, const u32x4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
    }
    static __device__ __forceinline__ void mma0(f32x4& acc, const u32x4& a, const u32x4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    }
    static __device__ __forceinline__ void mmac(f32x4& acc, const u32x4& a, const u32x4& b, const f32x4& c) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ uint32_t relu2(uint32_t packed) { return relu_packed16(packed); }
};

// One MFMA step of a GEMM whose LDS-side operand fragment `w` belongs to K-group KGI and whose register-side operand
// is the array x[K-group][token block] (see PrecX2).  SWAP: the register-side operand is the MFMA's A operand.
// FIRST: 0 accumulate, 1 start from zero, 2 start from `c`.
template <class P, int KGI, bool SWAP, int FIRST, int NKG, int NTT>
__device__ __forceinline__ void mma_kg(f32x4& acc, const u32x4& w, const u32x4 (&x)[NKG][NTT], int t, const f32x4& c) {
    auto one = [&](const u32x4& xv, int first) {
        if (first == 1) { if constexpr (SWAP) P::mma0(acc, xv, w); else P::mma0(acc, w, xv); }
        else if (first == 2) { if constexpr (SWAP) P::mmac(acc, xv, w, c); else P::mmac(acc, w, xv, c); }
        else { if constexpr (SWAP) P::mma(acc, xv, w); else P::mma(acc, w, xv); }
    };
    if constexpr (!P::kSplit) {
        one(x[KGI][t], FIRST);
    } else if constexpr (KGI % 2 == 0) {
        one(x[KGI][t], FIRST);
        one(x[KGI + 1][t], 0);
    } else {
        static_assert(FIRST == 0, "a split GEMM starts on an even K-group");
        one(x[KGI - 1][t], 0);
    }
}

// Store 4 consecutive elements (fp32 values) as P::elem at dst (8/16 B aligned)
template <class P>
__device__ __forceinline__ void store4(void* dst, float a, float b, float c, float d) {
    if constexpr (P::kSplit) {       // dst = the hi plane's place of the 4 elements (row base + P::row_byte(n)); lo 64 bytes on
        uint32_t h0, l0, h1, l1;
        P::split2(a, b, h0, l0);
        P::split2(c, d, h1, l1);
        *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(static_cast<char*>(dst) + 64) = make_uint2(l0, l1);
    } else if constexpr (P::kIsBF16) {
        *reinterpret_cast<uint2*>(dst) = make_uint2(P::pack2(a, b), P::pack2(c, d));
    } else {
        *reinterpret_cast<float4*>(dst) = make_float4(a, b, c, d);
    }
}

// Paired feature blocks.  In the accumulator layout a lane owns 4 consecutive
// tile rows (4g + r) of every 16-row block, i.e. 8-byte bf16 stores scattered
// over 16 token rows -- the slowest pattern the store path has (tools/
// store_probe.hip: 3.5 TB/s against 5.4 TB/s for 16-byte pieces).  The engine
// therefore uploads weight rows so that tile row 16e + 4g + r of every 32-row
// group computes natural feature 8g + 4e + r (pair_row): the lane's values of
// blocks 2p and 2p+1 are then the 8 consecutive features 32p + 8g .. + 7, one
// 16-byte store.  Nothing else changes: outputs stay in natural feature order.
__host__ __device__ inline int pair_row(int j) {          // natural feature computed by tile row j
    const int s = j & 31;
    return (j & ~31) + 8 * ((s & 15) >> 2) + 4 * (s >> 4) + (s & 3);
}
__device__ __forceinline__ int pair_feature(int nb, int g) {   // first of the lane's 4 features of block nb
    return (nb >> 1) * 32 + 8 * g + 4 * (nb & 1);
}
// put() the lane's 4 values of block 2p (e = 0), then of block 2p+1 (e = 1);
// dst8 = address of feature 32p + 8g.  bf16: one 16-byte store at e = 1.
template <class P>
struct PairStore {
    uint32_t lo0, lo1, sl0, sl1;
    __device__ __forceinline__ void put(void* dst8, int e, float a, float b, float c, float d) {
        if constexpr (P::kSplit) {   // dst8 = the hi plane's place of feature 32p + 8g (row base + P::row_byte(n & ~7))
            uint32_t h0, l0, h1, l1;
            P::split2(a, b, h0, l0);
            P::split2(c, d, h1, l1);
            if (e == 0) { lo0 = h0; lo1 = h1; sl0 = l0; sl1 = l1; }
            else {
                *reinterpret_cast<u32x4*>(dst8) = u32x4{lo0, lo1, h0, h1};
                *reinterpret_cast<u32x4*>(static_cast<char*>(dst8) + 64) = u32x4{sl0, sl1, l0, l1};
            }
        } else if constexpr (P::kIsBF16) {
            if (e == 0) { lo0 = P::pack2(a, b); lo1 = P::pack2(c, d); }
            else *reinterpret_cast<u32x4*>(dst8) = u32x4{lo0, lo1, P::pack2(a, b), P::pack2(c, d)};
        } else {
            reinterpret_cast<float4*>(dst8)[e] = make_float4(a, b, c, d);
        }
    }
};

// Reductions over the 4 lane groups g (lanes l, l+16, l+32, l+48) with the
// gfx950 row-swap VALU instructions instead of LDS-routed shuffles:
// v_permlane16_swap(x, x) leaves (x0,x0,x2,x2) / (x1,x1,x3,x3) in its two
// results (xi = row i), v_permlane32_swap(x, x) leaves (x0,x1,x0,x1) /
// (x2,x3,x2,x3); one combine after each gives every lane the full result.
__device__ __forceinline__ float wave_sum_g(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
// max of three without the NaN-canonicalising copies fmaxf() drags in for
// values the compiler cannot prove canonical (MFMA results)
__device__ __forceinline__ float max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float wave_max_g(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = max3(__uint_as_float(a[0]), __uint_as_float(a[1]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return max3(__uint_as_float(b[0]), __uint_as_float(b[1]), __uint_as_float(b[1]));
}

// Exact (erf-form) GELU, torch.nn.functional.gelu's default: gelu(x) = x Phi(x).  Built for the packed fp32
// instructions (v_pk_mul / v_pk_fma: two values per issue slot) and ONE transcendental per value:
//     0.5 erfc(z) = 0.5 exp2(-z G(z)),   z = |x| / sqrt 2,
// G a degree-7 polynomial fitted here to -log2(erfc z) / z on [0, 4.3] (least squares re-weighted towards minimax,
// weight = the sensitivity z erfc z; z is clamped to 4.3: erfc(4.3) = 1e-9), and
//     gelu(x) = max(x, 0) - (1 / sqrt 2) z exp2(-z G(z))        (x >= 0: x - x 0.5 erfc z;  x < 0: x 0.5 erfc z)
// -- no 1 - (1 - ..) cancellation on the negative side, no division.  Max abs error against the exact function over
// [-8, 8], evaluated in fp32: 5.0e-7 (the Abramowitz & Stegun 7.1.26 form used until round 4: 5.2e-7, with a
// reciprocal and an exponential per value: 28 instructions per pair against 15).  At the wav2vec2 widths the
// activation costs as many issue cycles as the GEMM in front of it.
__device__ __forceinline__ f32x2 gelu_erf_pair(f32x2 x) {
    const f32x2 z = {fminf(fabsf(x.x) * 0.70710678118654752f, 4.3f), fminf(fabsf(x.y) * 0.70710678118654752f, 4.3f)};
    f32x2 g = z * 4.5357837683e-05f + -4.4550141416e-04f;
    g = g * z + 1.4894216193e-03f;
    g = g * z + 7.7466185625e-04f;
    g = g * z + -2.8253708586e-02f;
    g = g * z + 1.4848162721e-01f;
    g = g * z + 9.1841639080e-01f;
    g = g * z + 1.6279085932e+00f;
    const f32x2 arg = z * g;
    const f32x2 e = {__builtin_amdgcn_exp2f(-arg.x), __builtin_amdgcn_exp2f(-arg.y)};
    // max(x, 0) as x / 2 + |x| / 2: fmaxf() returns its non-NaN operand, which turned GELU(NaN) and GELU(-Inf) into a
    // finite value where torch returns NaN (-Inf / 2 + Inf / 2 = NaN here too); halved BEFORE the add, so that a finite
    // x above FLT_MAX / 2 stays finite (x + |x| overflowed there: ADVICE r5).  One multiply and one fma with a free
    // |.| modifier per value
    const f32x2 half = x * 0.5f;
    const f32x2 relu = {fmaf(fabsf(x.x), 0.5f, half.x), fmaf(fabsf(x.y), 0.5f, half.y)};
    return __builtin_elementwise_fma(z * e, f32x2{-0.70710678118654752f, -0.70710678118654752f}, relu);
}
__device__ __forceinline__ float gelu_erf(float x) {
    const float z = fminf(fabsf(x) * 0.70710678118654752f, 4.3f);
    float g = fmaf(z, 4.5357837683e-05f, -4.4550141416e-04f);
    g = fmaf(g, z, 1.4894216193e-03f);
    g = fmaf(g, z, 7.7466185625e-04f);
    g = fmaf(g, z, -2.8253708586e-02f);
    g = fmaf(g, z, 1.4848162721e-01f);
    g = fmaf(g, z, 9.1841639080e-01f);
    g = fmaf(g, z, 1.6279085932e+00f);
    return fmaf(z * __builtin_amdgcn_exp2f(-(z * g)), -0.70710678118654752f, fmaf(fabsf(x), 0.5f, 0.5f * x));   // (NaN / -Inf -> NaN, as torch; no overflow below FLT_MAX)
}

// Phase-skipping switches of the timing experiments (PPGS_AMD_L32_DEBUG / PPGS_AMD_H32_DEBUG: WRONG results by
// design) exist only in -DPPG_DEBUG_MODES builds; the product library neither reads the variables nor tests the field.
#ifdef PPG_DEBUG_MODES
#define PPG_DBG(a) ((a).debug_mode)
#else
#define PPG_DBG(a) 0
#endif

// Epilogue kinds of linear_kernel
enum {
    EPI_INCONV = 0,   // +bias, zero beyond valid, +PE  -> X (fp32) [+ Xb]
    EPI_QKV = 1,      // +bias -> Q,K row-major; V transposed (swapped MFMA operands)
    EPI_RESLN = 2,    // +bias +residual, LayerNorm -> X [+ Xb]
    EPI_RELU = 3,     // +bias, ReLU -> hidden (unfused FFN path)
    EPI_OUTCONV = 4,  // +bias, mask, softmax over 40, scatter into (B,40,T)
    EPI_GELU = 5,     // strided k-tap conv without bias + exact GELU -> out_rows (wav2vec2 feature encoder layers 1..6)
    EPI_GENERAL = 6,  // +bias, [exact GELU / ReLU], [zero rows past the window's valid], [+fp32 residual] -> fp32 and / or 16-bit rows
                      // (wav2vec2 transformer body: projections, grouped positional conv as taps x groups over blockIdx.y)
};

struct LinearArgs {
    const char* act;          // token-major activations
    int lda_bytes;            // bytes per activation row
    int taps;                 // 1 (linear) or 5 (k=5 'same' conv over window rows)
    int groups_per_tap;       // 64-byte K-groups per tap
    int real_groups;          // taps * groups_per_tap
    int total_groups;         // real_groups rounded up to even (W rows zero padded)
    const char* W;            // [N][total_groups*64 bytes]
    const float* bias;        // [N]
    int N;                    // padded output features
    // outputs
    float* X;                 // residual stream fp32 [M][H]
    char* Xb;                 // bf16 copy of X [M][H] (bf16 mode), else null
    int H;
    const float* pe;          // [max_pos][H]
    const float* gamma;
    const float* beta;
    char* out_rows;           // row-major output (QK / hidden), elements of P::elem
    int out_ld;               // its row stride in elements
    char* vt;                 // transposed V [H][vt_ld]
    int vt_ld;
    int v_start;              // first V feature (2H); INT_MAX when unused
    float* out;               // (B, C, T) fp32 final output
    int out_T;
    int out_C;
    int softmax;
    const int* blk_win;       // window of each 16-token block (-1 = padding)
    const PpgWindow* win;
    int M;                    // rows in the token-major buffers (multiple of 16)
    unsigned long long* dbg;  // PPG_LIN_TIMING builds: 16 s_memtime stamps per workgroup (tools/lin_timing.py)
    int x_tiled;              // EPI_INCONV: X is written in X32 order (1) or as fp16 in X16 order (2): the layer32 kernel follows
    int stride;               // EPI_GELU: output row m reads input rows stride * m + tap, tap = 0 .. taps - 1
    int M_in;                 // EPI_GELU: rows of the input buffer
    // EPI_GENERAL
    int act_fn;               // 0 none, 1 ReLU, 2 exact GELU
    int zero_invalid;         // rows at or past their window's valid length come out as zeros
    const float* residual;    // fp32 [M][out_ld32] added after the activation, or null
    float* out32;             // fp32 [M][out_ld32], or null
    int out_ld32;
    int act_y_stride;         // bytes added to `act` per blockIdx.y (grouped convolution: group y reads its own channels)
    const int* rowmap;        // see GatherArgs (16 tokens per wave only)
    int map_blocks;
    unsigned* overflow;       // EPI_OUTCONV: bit 0 is set when a logit of a valid frame is not finite (fp16 operands past
                              // 65504 upstream, non-finite input features ...): the engine's sticky flag, or null
};

struct FfnArgs {
    // fused attention out-projection + residual + LayerNorm (ffn_kernel<.., OP = true>); Wo == null: not fused
    const char* ao;           // attention output, token-major [M][H] elements
    const char* Wo;           // [H][H], rows in paired order
    const float* bo;
    const float* g1;          // norm1
    const float* e1;
    // fused Q/K/V projection of the NEXT layer on the kernel's own output (ffn_kernel<.., QKV = true>); Wq == null: not fused
    const char* Wq;           // [3H][H], rows in paired order (fp32: columns too)
    const float* bq;
    char* qk_out;             // [M][2H] (q | k), elements
    char* vt_out;             // transposed V [H][vt_ld]
    int vt_ld;
    const int* blk_win;       // window of each 16-token block (-1 = padding)
    const PpgWindow* win;
    float* X;                 // residual stream, in/out
    char* Xb;                 // act operand / bf16 copy (bf16 mode); null in fp32 mode
    const char* W1;           // [F][H]
    const float* b1;
    const char* W2p;          // [H][F], k-permuted for bf16 (see pack_w2)
    const float* b2;
    const float* gamma;
    const float* beta;
    int H;
    int F;
    int M;
    unsigned long long* dbg;  // PPG_FFN_TIMING builds: s_memtime stamps of workgroup 0
    float* partial;           // split-hidden mode: [splits][M][H] fp32 partial sums, else null
    int splits;
    const int* rowmap;        // see GatherArgs (16 tokens per wave, no hidden splits)
    int map_blocks;
    int* tickets;             // split-hidden mode, 16 tokens per wave, hidden 256, <= 8 splits: one zeroed counter per token tile --
                              // the LAST of a tile's workgroups to arrive sums the partial rows and applies LayerNorm-2
                              // itself (no ffn_reduce_ln_kernel launch); null: the two-pass form
};

// Memory laid out for the feature-split layer kernel (160-token workgroup tiles):
//  * X32: the fp32 residual stream in the kernel's accumulator order -- the float4 (token m,
//    features n..n+3) sits where lane (token, feature half) of wave n/64 loads/stores it, so every
//    residual load and every LayerNorm store is one contiguous KiB per wave instruction;
//  * AO32: the attention output as the kernel's B fragments (token panel: [tile][5 token blocks]
//    [16 K-steps] fragments of 1 KiB, slot 8 (l >> 5) + j of K-step ks = feature 16 ks + 8 (l >> 5) + j).
__host__ __device__ inline int layer32_tile_tokens(int hidden) { return hidden == 512 ? 96 : 160; }
__host__ __device__ inline size_t x32_index(int m, int n, int hidden) {        // float index of X[m][n], n % 4 == 0
    const int toks = layer32_tile_tokens(hidden), RB = hidden / 128, TB = toks / 32;
    const int tile = m / toks, r = m - tile * toks, tb = r >> 5, tok = r & 31;
    const int w = n / (32 * RB), rb = (n >> 5) % RB, hh = (n >> 4) & 1, q = (n >> 2) & 3;
    return ((((((size_t)tile * 4 + w) * TB + tb) * RB + rb) * 4 + q) * 64 + hh * 32 + tok) * 4 + (n & 3);
}
// X16: the same stream as fp16 (the bf16 mode's residual stream between two layer kernels: half the bytes of the
// layer kernel's two chip-wide bursts; 11 significand bits against the mode's 8-bit operands): the 8 halves
// (token m, features n..n+7 of a 16-feature run) are one 16-byte lane slot of a KiB per (token block, row block, half)
__host__ __device__ inline size_t x16_index(int m, int n, int hidden) {        // half index of X[m][n]
    const int toks = layer32_tile_tokens(hidden), RB = hidden / 128, TB = toks / 32;
    const int tile = m / toks, r = m - tile * toks, tb = r >> 5, tok = r & 31;
    const int w = n / (32 * RB), rb = (n >> 5) % RB, hh = (n >> 4) & 1, s = (n >> 3) & 1;
    return ((((((size_t)tile * 4 + w) * TB + tb) * RB + rb) * 2 + s) * 64 + hh * 32 + tok) * 8 + (n & 7);
}
__host__ __device__ inline size_t ao32_byte(int m, int n, int hidden) {        // byte offset of AO[m][n], n % 8 == 0, 16-bit elements
    const int toks = layer32_tile_tokens(hidden), KS = hidden / 16, TB = toks / 32;
    const int tile = m / toks, r = m - tile * toks, tb = r >> 5, tok = r & 31;
    return ((((size_t)tile * TB + tb) * KS + (n >> 4)) * 64 + ((n >> 3) & 1) * 32 + tok) * 16;
}

// The feature-split layer kernel (ppg_layer32.hip): out-projection + residual + LayerNorm-1,
// FFN + residual + LayerNorm-2 [, the next layer's Q/K/V projection] for 160-token workgroups
// on v_mfma_f32_32x32x16.  Weights come as host-packed fragment images (1 KiB per MFMA A
// fragment, in consumption order; ppg_engine.hip pack_layer32).
struct Layer32Args {
    const char* ao;           // attention output in AO32 order (see ao32_byte)
    const char* wo_img;       // [4 waves][RB row blocks][H/16 k-steps] fragments, RB = H/128
    const char* w1_img;       // [F/128 chunks][4 waves][H/16 k-steps] fragments
    const char* w2_img;       // [F/128 chunks][4 waves][RB row blocks][8 k-steps] fragments
    const float* bo; const float* g1; const float* e1;      // out-proj bias, norm1
    const float* b1; const float* b2; const float* g2; const float* e2;
    float* X;                 // residual stream fp32 in X32 order (see x32_index), in/out
    char* Xb;                 // row-major 16-bit copy of the result (operand of the kernels that follow); null: not written
    int M;
    int F;
    int H;                    // 256 (160-token workgroups) or 512 (96-token workgroups)
    // fused Q/K/V projection of the next layer (wq_img != null)
    const char* wq_img;       // [4 waves][3 RB steps][H/16 k-steps] fragments
    const float* bq;
    char* qk_out;             // [M][2H] (q | k)
    char* vt_out;             // transposed V [H][vt_ld]
    int vt_ld;
    const int* blk_win;
    const PpgWindow* win;
    unsigned long long* dbg;  // PPG_FFN_TIMING builds: s_memtime stamps of workgroup 0
    int debug_mode;           // PPGS_AMD_L32_DEBUG (bisecting): bit 0 skip the out-projection, bit 1 skip the FFN
    int write_x;              // 0: the fp32 result is not stored (last layer: only the 16-bit copy is read afterwards)
    int x_half;               // 1: X is the fp16 stream in X16 order (see x16_index) instead of fp32 in X32 order
    int sub_tiles;            // 1 (hidden 256, F / 128 even): workgroups of two token blocks, three per 160-token tile -- batches that cannot give every CU a tile
};

// One attention workgroup's work: a query tile of one window.  The window fields
// the kernel needs ride along (one 32-byte load instead of item -> window, two
// dependent round trips at the head of every workgroup).
struct AttnItem {
    int window;
    int q0;                   // first query (window-relative) of the block's tile
    int tok_off;              // copies of the PpgWindow fields of `window`
    int vt_off;
    int frames;
    int valid;
    int narrow;               // 1: the item is half a query tile wide (16 queries per wave instead of 32)
    int pad1;
};

struct AttnArgs {
    unsigned long long* dbg;  // PPG_ATTN_TIMING builds: s_memtime stamps of workgroup (0, 0)
    const char* qk;           // [M][2H] (q | k), elements
    int qk_ld_bytes;
    const char* vt;           // [H][vt_ld]
    int vt_ld_bytes;
    char* ao;                 // [M][H] attention output (elements)
    int H;
    int causal;
    const AttnItem* items;
    const PpgWindow* win;
    int M;
    int ao_tiled;             // the output goes out in AO32 order (the layer32 kernel follows)
    int heads;
    int rebase_always;        // tests (PPGS_AMD_ATTN_REBASE=always): re-base the softmax shift whenever a p exceeds 1
};

struct GatherArgs {
    const void* feats;        // (B, C, T) fp16 or fp32
    int dtype;                // PPG_DTYPE_*
    int C;
    int T;
    int overlap;
    char* xw;                 // [M][Cp] elements
    int Cp;
    const int* blk_win;
    const PpgWindow* win;
    int M;
    // scratch areas nobody writes but the attention tiles read (masked): zeroed by
    // the extra blockIdx.y row of the gather launch
    char* vt;                 // transposed V [vt_rows][vt_ld] elements
    int vt_ld;
    int vt_rows;              // H
    int vt_tokens;            // columns in use; the rest of a row is slack
    int nwin;
    char* qk_slack;           // the rows behind the last token of the q|k buffer
    int qk_slack_bytes;       // multiple of 16
    // Row map (batched streaming: a step touches a few 16-row blocks of every stream's window): slot k of the launch
    // works on token rows rowmap[k] .. + 15 instead of 16 k ..; slots past map_blocks hold M (nothing to do)
    const int* rowmap;
    int map_blocks;
};

// Window of a token row (the planner numbers rows window by window, every window padded to 16 rows)
struct TokMeta {
    int w;       // window index or -1
    int tt;      // window-relative position
    int frames;  // Tc
    int valid;   // mask length
};

__device__ __forceinline__ TokMeta tok_meta(const int* blk_win, const PpgWindow* win, int m, int M) {
    TokMeta t;
    t.w = -1; t.tt = 0; t.frames = 0; t.valid = 0;
    if (m < M) {
        const int w = blk_win[m >> 4];
        if (w >= 0) {
            t.w = w;
            t.tt = m - win[w].tok_off;
            t.frames = win[w].frames;
            t.valid = win[w].valid;
        }
    }
    return t;
}

// ppg_head32.hip: window gather + input convolution + layer 0's Q/K/V projection of a 160-token tile
struct Head32Args {
    const void* feats;        // (B, C, T) fp16 or fp32
    int dtype;                // PPG_DTYPE_*
    int C;                    // <= 96
    int T;
    int overlap;
    const char* win_img;      // input convolution as A fragments [wave][rb][30 K-steps = 5 taps x 6 channel blocks] (+ 1 pad)
    const float* b_in;
    const float* pe;          // [max_positions][H]
    float* X;                 // fp32 residual stream out, X32 order
    const char* wq_img;       // layer 0's in_proj (as Layer32Args::wq_img)
    const float* bq;
    char* qk_out;
    char* vt_out;
    int vt_ld;
    const int* blk_win;
    const PpgWindow* win;
    int M;
    int H;                    // 256
    int tiles;                // workgroups 0 .. tiles - 1 compute; the rest keep the scratch areas finite (see GatherArgs)
    int nwin;
    int vt_rows;
    int vt_tokens;
    char* qk_slack;
    int qk_slack_bytes;
    int x_half;               // 1: X is written as fp16 in X16 order (see x16_index)
    int sub_tiles;            // 1: workgroups of two token blocks, three per 160-token tile (see Layer32Args)
    int debug_mode;           // timing experiments (wrong results): 1 no convolution, 2 no Q/K/V, 4 no gather
    unsigned long long* dbg;  // PPGS_AMD_H32_TIMING=1: s_memtime stamps of workgroup 0, [wave][16]
};

// ppg_gemm32.hip: Y = act_fn(X W^T + bias) [+ residual], 160 rows x 256 features per workgroup
struct Gemm32Args {
    const char* x;            // [M][K] 16-bit, row-major (row m at x + m * lda_bytes; lda_bytes 0 = K * 2.  Rows may overlap:
                              // a strided k-tap convolution over [rows][C] is the GEMM with K = k * C, lda = stride * C)
    const char* w_img;        // A fragments [N / 256][wave][K / 128][rb][8 K-steps] of 1 KiB (rows in accumulator order phi)
    const float* bias;        // [N]
    const float* residual;    // fp32 [M][N] or null
    float* out32;             // fp32 [M][N] or null
    char* out16;              // 16-bit [M][N] or null
    int M, N, K;              // N % 256 == 0, K % 128 == 0
    int act_fn;               // 0 none, 1 ReLU, 2 GELU
    // rows past an item's valid frames come out as zeros (HF: hidden_states[~mask] = 0 behind the feature projection):
    // item b owns rows b * rows_per_item .., valid frames in win[b].valid; null: every row is kept
    const PpgWindow* win;
    int rows_per_item;
    // Q | K | V (vt != null): passes (of 256 features) below v_pass0 go to out16 rows of leading dimension ld_out, the
    // others to V^T [N - 256 v_pass0][vt_ld] in attn_kernel's layout (image rows of those passes in pair_row order);
    // items start at multiples of 32 rows (rows_per_item % 32 == 0)
    char* vt;
    int vt_ld, ld_out, v_pass0;
    int lda_bytes;
};

// ppg_posconv.hip: the wav2vec2 body's grouped positional convolution + GELU + residual (16 groups of 48, 128 taps)
struct PosConvArgs {
    const char* x16;          // [M][H] 16-bit operand rows (rows past an item's valid frames are zeros)
    int ldx_bytes;
    const char* w_img;        // A fragments [group 16][wave 4][tap 32][ks 3][rb 2] of 1 KiB: rows = features 32 rb + phi(l & 31)
                              // of the group (>= 48: zeros), K = channels 16 ks + 8 (l >> 5) .. + 7 of tap 32 wave + tap
    const float* bias;        // [H]
    const float* residual;    // fp32 [M][H]
    float* out32;             // fp32 [M][H]
    int M, H;
    int rows_per_item;        // item b owns rows b * rows_per_item ..; its first `frames` rows are the sequence
    int frames;
    int tiles_per_item;       // ceil(rows_per_item / 128)
};

// ppg_ffn32x2.hip: the FFN block in the fp16x2 mode on the feature-split machinery (96-token workgroups)
struct Ffn32X2Args {
    const char* xb;           // [M][H] as [32 hi | 32 lo] fp16 blocks (PrecX2 rows, 4 bytes per element): the operand copy of X
    float* X;                 // fp32 [M][H]: residual in, LayerNorm-2 output out
    char* xb_out;             // the operand copy of the result (may be xb)
    const char* w1_img;       // [F/128 chunks][4 waves][W1 hi: 16 K-steps | W1 lo: 16 K-steps] fragments of 1 KiB
    const char* w2_img;       // [F/128 chunks][4 waves][W2 hi: (2 row blocks, 8 K-steps) | W2 lo] fragments
    const float* b1; const float* b2; const float* g2; const float* e2;
    int M, F, H;
    // the attention block's tail in front (wo_img != null): x1 = LayerNorm1(X + bo + Wo ao), ao in the rows' format
    const char* ao;
    const char* wo_img;       // [4 waves][K half][Wo hi: (2 row blocks, 8 K-steps) | Wo lo] fragments
    const float* bo; const float* g1; const float* e1;
    // the NEXT layer's Q / K / V projection behind LayerNorm-2 (wq_img != null; with wo_img only)
    const char* wq_img;       // [4 waves][kind q, k, v][K half][hi: (2 row blocks, 8 K-steps) | lo] fragments
    const float* bq;
    char* qk_out;             // [M][2H] (q | k) as [32 hi | 32 lo] rows
    char* vt_out;             // transposed V [H][vt_ld], columns as [32 hi | 32 lo] blocks
    int vt_ld;
    const int* blk_win;
    const PpgWindow* win;
    unsigned long long* dbg;  // PPG_FFN_TIMING builds: s_memtime stamps of workgroup 0
};
