// Stand-alone reproducer of the co-residency failure behind DESIGN 4.4 (no PyTorch, no library of this package):
// a VICTIM kernel evaluates a chain of VALU instructions twice from the same register inputs and counts lanes whose
// two results differ; an AGGRESSOR kernel spins on matrix-core instructions on another HIP stream.  Which victim
// instruction classes break beside which aggressor classes, and does it take two different kernels?
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -o tools/bin/pk_mfma_probe tools/probes/pk_mfma_probe.hip
//   tools/bin/pk_mfma_probe [reps]
//
// Victim classes (template V):  0 v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 with op_sel modifiers (the frontend's
// complex arithmetic), 1 the same three without modifiers, 2 v_pk_mul_f32 only, 3 v_pk_add_f32 only, 4 v_pk_fma_f32
// only, 5 v_pk_mov_b32 only, 6 v_fma_f64, 7 v_pk_fma_f16, 8 plain v_fma_f32 (control), 9 v_mul_f32 + v_add_f32 (control)
// Aggressor classes (template A): 0 none, 1 v_mfma_f32_16x16x32_bf16, 2 v_mfma_f32_32x32x16_bf16,
// 3 v_mfma_f32_32x32x2_f32, 4 v_mfma_f64_16x16x4_f64, 5 v_mfma_i32_16x16x64_i8, 6 16x16x32 bf16 with the accumulator
// in AGPRs, 7 v_exp_f32 (control)
// Placement: `pair` = victim and aggressor are different kernels on two streams; `same-wg` = one 512-thread kernel
// whose waves 0-3 run the victim chain and waves 4-7 the aggressor loop (two waves per SIMD, one of each);
// `same-kernel` = one 256-thread kernel whose even workgroups are victims and odd ones aggressors.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int V>
__device__ __forceinline__ unsigned victim_chain(float seed) {
    // 16 register pairs, 4 rounds; returns a hash of the final registers
    f32x2 v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = f32x2{seed + 0.01f * k, seed * 0.5f - 0.02f * k};
#pragma unroll
    for (int k = 0; k < 16; ++k) asm volatile("" : "+v"(v[k]));
#pragma unroll
    for (int round = 0; round < 4; ++round)
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            f32x2& a = v[k];
            const f32x2 b = v[(k + 5) & 15], c = v[(k + 3) & 15];
            const f32x2 q = {0.25f, 0.25f};
            if constexpr (V == 0) {
                f32x2 t;
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(a) : "v"(a), "v"(b), "v"(t));
                asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(a) : "v"(a), "v"(c));
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(q));
            } else if constexpr (V == 1) {
                f32x2 t;
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(a), "v"(b));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(a) : "v"(a), "v"(b), "v"(t));
                asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(c));
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(q));
            } else if constexpr (V == 2) {
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(b));
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(q));
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(c));
            } else if constexpr (V == 3) {
                asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(b));
                asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(c));
                asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(a) : "v"(a), "v"(b));
            } else if constexpr (V == 4) {
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(a) : "v"(a), "v"(q), "v"(b));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(a) : "v"(a), "v"(q), "v"(c));
            } else if constexpr (V == 5) {
                f32x2 t;
                asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(t) : "v"(a), "v"(b));
                asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[0,1]" : "=v"(a) : "v"(t), "v"(c));
            } else if constexpr (V == 6) {
                double d = __builtin_bit_cast(double, a), e = __builtin_bit_cast(double, b);
                asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(d), "v"(0.999), "v"(e));
                asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(d), "v"(0.5), "v"(e));
                a = __builtin_bit_cast(f32x2, d);
            } else if constexpr (V == 7) {
                asm volatile("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(a.x) : "v"(a.x), "v"(b.x), "v"(c.x));
                asm volatile("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(a.y) : "v"(a.y), "v"(b.y), "v"(c.y));
                asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(a.x) : "v"(a.x), "v"(q.x));
            } else if constexpr (V == 8) {
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a.x) : "v"(a.x), "v"(q.x), "v"(b.x));
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a.y) : "v"(a.y), "v"(q.y), "v"(b.y));
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a.x) : "v"(a.x), "v"(q.x), "v"(c.y));
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a.y) : "v"(a.y), "v"(q.y), "v"(c.x));
            } else if constexpr (V == 10) {     // class 0 with every destination distinct from its sources (early clobber)
                f32x2 t, u, w;
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=&v"(t) : "v"(a), "v"(b));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=&v"(u) : "v"(a), "v"(b), "v"(t));
                asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=&v"(w) : "v"(u), "v"(c));
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=&v"(a) : "v"(w), "v"(q));
            } else if constexpr (V == 11) {     // the op_sel multiply alone (destination distinct)
                f32x2 t;
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=&v"(t) : "v"(a), "v"(b));
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(a) : "v"(t), "v"(q));
            } else if constexpr (V == 12) {     // the op_sel fma alone, destination = source 0
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(a) : "0"(a), "v"(q), "v"(b));
            } else if constexpr (V == 13) {     // the op_sel add alone, destination = source 0
                asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(a) : "0"(a), "v"(c));
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(q));
            } else if constexpr (V == 14) {     // the op_sel fma alone, destination distinct
                f32x2 u;
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=&v"(u) : "v"(a), "v"(q), "v"(b));
                a = u;
            } else if constexpr (V == 15) {     // negation modifiers only
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,1,0] neg_hi:[1,0,0]" : "=v"(a) : "0"(a), "v"(q), "v"(b));
                asm volatile("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(a) : "0"(a), "v"(c));
            } else if constexpr (V == 16) {     // broadcast selections (both halves read the same half), destination = source 0
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,0]" : "=v"(a) : "0"(a), "v"(b));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,1,1]" : "=v"(a) : "0"(a), "v"(q), "v"(c));
            } else if constexpr (V == 17) {     // swap selections on the add, destination distinct
                f32x2 u;
                asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=&v"(u) : "v"(a), "v"(c));
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(a) : "v"(u), "v"(q));
            } else if constexpr (V == 19 || V == 20 || V == 22) {     // class 0 with idle issue slots between the dependent instructions
#define PPG_GAP(n) asm volatile("s_nop " #n)
                f32x2 t;
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));
                if constexpr (V == 19) PPG_GAP(0); else if constexpr (V == 20) PPG_GAP(1); else PPG_GAP(7);
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(a) : "v"(a), "v"(b), "v"(t));
                if constexpr (V == 19) PPG_GAP(0); else if constexpr (V == 20) PPG_GAP(1); else PPG_GAP(7);
                asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(a) : "v"(a), "v"(c));
                if constexpr (V == 19) PPG_GAP(0); else if constexpr (V == 20) PPG_GAP(1); else PPG_GAP(7);
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(q));
                if constexpr (V == 19) PPG_GAP(0); else if constexpr (V == 20) PPG_GAP(1); else PPG_GAP(7);
            } else if constexpr (V == 21) {     // the three swizzled forms on INDEPENDENT data: no instruction reads its predecessor's result
                // (sources are the pairs 4 and 8 places on, last written a dozen instructions ago)
                f32x2& d0 = v[k];
                const f32x2 e = v[(k + 4) & 15], f = v[(k + 8) & 15], g2 = v[(k + 12) & 15];
                if ((k & 3) == 0) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(d0) : "v"(e), "v"(f));
                else if ((k & 3) == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(d0) : "v"(e), "v"(f), "v"(g2));
                else if ((k & 3) == 2) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d0) : "v"(e), "v"(f));
                else asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d0) : "v"(e), "v"(q));
            } else if constexpr (V == 30) {
                f32x2 u;
                asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=&v"(u) : "v"(a), "v"(b), "v"(c));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(a) : "v"(u), "v"(q), "v"(f32x2{0.31f, 0.17f}));
            } else if constexpr (V == 31) {
                f32x2 u;
                asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=&v"(u) : "v"(a), "v"(b), "v"(c));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(a) : "v"(u), "v"(q), "v"(f32x2{0.31f, 0.17f}));
            } else if constexpr (V == 32) {
                f32x2 u;
                asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=&v"(u) : "v"(a), "v"(b), "v"(c));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(a) : "v"(u), "v"(q), "v"(f32x2{0.31f, 0.17f}));
            } else if constexpr (V == 33) {
                f32x2 u;
                asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(u) : "v"(a), "v"(b), "v"(c));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(a) : "v"(u), "v"(q), "v"(f32x2{0.31f, 0.17f}));
            } else if constexpr (V == 34) {
                f32x2 u;
                asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=&v"(u) : "v"(a), "v"(b), "v"(c));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(a) : "v"(u), "v"(q), "v"(f32x2{0.31f, 0.17f}));
            } else if constexpr (V == 35) {
                f32x2 u;
                asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=&v"(u) : "v"(a), "v"(b), "v"(c));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(a) : "v"(u), "v"(q), "v"(f32x2{0.31f, 0.17f}));
            } else if constexpr (V == 39) {
                f32x2 u;
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=&v"(u) : "v"(a), "v"(b), "v"(c));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(a) : "v"(u), "v"(q), "v"(f32x2{0.31f, 0.17f}));
            } else if constexpr (V == 40) {
                f32x2 u;
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=&v"(u) : "v"(a), "v"(b), "v"(c));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(a) : "v"(u), "v"(q), "v"(f32x2{0.31f, 0.17f}));
            } else if constexpr (V == 41) {
                f32x2 u;
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=&v"(u) : "v"(a), "v"(b), "v"(c));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(a) : "v"(u), "v"(q), "v"(f32x2{0.31f, 0.17f}));
            } else if constexpr (V == 42) {
                f32x2 u;
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(u) : "v"(a), "v"(b), "v"(c));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(a) : "v"(u), "v"(q), "v"(f32x2{0.31f, 0.17f}));
            } else if constexpr (V == 43) {
                f32x2 u;
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=&v"(u) : "v"(a), "v"(b), "v"(c));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(a) : "v"(u), "v"(q), "v"(f32x2{0.31f, 0.17f}));
            } else if constexpr (V == 44) {
                f32x2 u;
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=&v"(u) : "v"(a), "v"(b), "v"(c));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(a) : "v"(u), "v"(q), "v"(f32x2{0.31f, 0.17f}));
            } else if constexpr (V == 48) {
                f32x2 u;
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "=&v"(u) : "v"(a), "v"(b), "v"(c));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(a) : "v"(u), "v"(q), "v"(f32x2{0.31f, 0.17f}));
            } else if constexpr (V == 49) {
                f32x2 u;
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=&v"(u) : "v"(a), "v"(b), "v"(c));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(a) : "v"(u), "v"(q), "v"(f32x2{0.31f, 0.17f}));
            } else if constexpr (V == 50) {
                f32x2 u;
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=&v"(u) : "v"(a), "v"(b), "v"(c));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(a) : "v"(u), "v"(q), "v"(f32x2{0.31f, 0.17f}));
            } else if constexpr (V == 51) {
                f32x2 u;
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=&v"(u) : "v"(a), "v"(b), "v"(c));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(a) : "v"(u), "v"(q), "v"(f32x2{0.31f, 0.17f}));
            } else if constexpr (V == 52) {
                f32x2 u;
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=&v"(u) : "v"(a), "v"(b), "v"(c));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(a) : "v"(u), "v"(q), "v"(f32x2{0.31f, 0.17f}));
            } else if constexpr (V == 53) {
                f32x2 u;
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=&v"(u) : "v"(a), "v"(b), "v"(c));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(a) : "v"(u), "v"(q), "v"(f32x2{0.31f, 0.17f}));
            } else if constexpr (V == 54) {
                f32x2 u;
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "=&v"(u) : "v"(a), "v"(b), "v"(c));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(a) : "v"(u), "v"(q), "v"(f32x2{0.31f, 0.17f}));
            } else if constexpr (V == 55) {
                f32x2 u;
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "=&v"(u) : "v"(a), "v"(b), "v"(c));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(a) : "v"(u), "v"(q), "v"(f32x2{0.31f, 0.17f}));
            } else if constexpr (V == 56) {
                f32x2 u;
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,1]" : "=&v"(u) : "v"(a), "v"(b), "v"(c));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(a) : "v"(u), "v"(q), "v"(f32x2{0.31f, 0.17f}));
            } else if constexpr (V == 18) {     // what the COMPILER writes for complex arithmetic on float2 (no inline asm)
                const f32x2 m = f32x2{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
                const f32x2 r = f32x2{m.x + c.y, m.y - c.x};
                a = r * 0.25f;
                asm volatile("" : "+v"(a));
            } else {
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a.x) : "v"(a.x), "v"(q.x));
                asm volatile("v_add_f32 %0, %1, %2" : "=v"(a.x) : "v"(a.x), "v"(b.y));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a.y) : "v"(a.y), "v"(q.y));
                asm volatile("v_add_f32 %0, %1, %2" : "=v"(a.y) : "v"(a.y), "v"(c.x));
            }
        }
    unsigned h = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) h = h * 31u + __builtin_bit_cast(unsigned, v[k].x) * 7u + __builtin_bit_cast(unsigned, v[k].y);
    return h;
}

template <int V>
__device__ __forceinline__ void victim_body(int iters, unsigned long long* counters) {
    unsigned long long bad = 0;
    for (int it = 0; it < iters; ++it) {
        const float seed = 0.37f + 1e-3f * (float)((it * 131 + threadIdx.x * 7 + blockIdx.x) & 1023);
        const unsigned r0 = victim_chain<V>(seed);
        const unsigned r1 = victim_chain<V>(seed);
        bad += r0 != r1;
    }
    if (bad) atomicAdd(counters, bad);
    if ((threadIdx.x & 63) == 0) atomicAdd(counters + 1, (unsigned long long)iters * 64);
}

template <int A>
__device__ __forceinline__ float aggressor_body(int iters) {
    const int lane = threadIdx.x & 63;
    bf16x8 a8, b8;
    for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(0.01f * (float)(i + lane)); b8[i] = (__bf16)(0.02f * (float)(i + 1)); }
    f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
    f32x16 acc16 = {};
    f64x4 accd = {0., 0., 0., 0.};
    i32x4 acci = {0, 0, 0, 0};
    float x = 0.001f * (float)(lane + 1);
    for (int it = 0; it < iters; ++it) {
        if constexpr (A == 1) {
#pragma unroll
            for (int k = 0; k < 8; ++k) acc4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc4, 0, 0, 0);
        } else if constexpr (A == 2) {
#pragma unroll
            for (int k = 0; k < 4; ++k) acc16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc16, 0, 0, 0);
        } else if constexpr (A == 3) {
#pragma unroll
            for (int k = 0; k < 2; ++k) acc16 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, 0.5f, acc16, 0, 0, 0);
        } else if constexpr (A == 4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) accd = __builtin_amdgcn_mfma_f64_16x16x4f64((double)x, 0.5, accd, 0, 0, 0);
        } else if constexpr (A == 5) {
            const i32x4 ai = {lane, lane + 1, lane + 2, lane + 3}, bi = {1, 2, 3, 4};
#pragma unroll
            for (int k = 0; k < 8; ++k) acci = __builtin_amdgcn_mfma_i32_16x16x64_i8(ai, bi, acci, 0, 0, 0);
        } else if constexpr (A == 6) {
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc4) : "v"(a8), "v"(b8));
        } else if constexpr (A == 7) {
#pragma unroll
            for (int k = 0; k < 8; ++k) x = __builtin_amdgcn_exp2f(x) * 0.25f;
        } else if constexpr (A == 8) {
            typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
            f16x8 h8;
            for (int i = 0; i < 8; ++i) h8[i] = (_Float16)(0.01f * (float)(i + lane));
#pragma unroll
            for (int k = 0; k < 8; ++k) acc4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h8, h8, acc4, 0, 0, 0);
        } else if constexpr (A == 9) {
            typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
            f16x4 h4;
            for (int i = 0; i < 4; ++i) h4[i] = (_Float16)(0.01f * (float)(i + lane));
#pragma unroll
            for (int k = 0; k < 8; ++k) acc4 = __builtin_amdgcn_mfma_f32_16x16x16f16(h4, h4, acc4, 0, 0, 0);
        } else if constexpr (A == 10) {     // 32x32x16 bf16 back to back on four accumulators (a dense stream)
            f32x16 c1 = acc16, c2 = acc16, c3 = acc16;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                acc16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc16, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, c3, 0, 0, 0);
            }
            acc16 += c1 + c2 + c3;
        } else if constexpr (A == 11) {     // 16x16x32 bf16 on four independent accumulators
            f32x4 c1 = acc4, c2 = acc4, c3 = acc4;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                acc4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc4, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, c3, 0, 0, 0);
            }
            acc4 += c1 + c2 + c3;
        }
    }
    return x + acc4[0] + acc4[3] + acc16[0] + acc16[15] + (float)accd[0] + (float)acci[0];
}

template <int V>
__global__ __launch_bounds__(256, 2) void victim_kernel(int iters, unsigned long long* counters) {
    extern __shared__ char smem[];
    victim_body<V>(iters, counters);
}
template <int A>
__global__ __launch_bounds__(256) void aggressor_kernel(int iters, float* sink) {
    extern __shared__ char smem[];
    const float r = aggressor_body<A>(iters);
    if (r == 12345.678f) sink[threadIdx.x] = r;
}
// one workgroup, two waves per SIMD: waves 0-3 victims, 4-7 aggressors
template <int V, int A>
__global__ __launch_bounds__(512, 1) void same_wg_kernel(int v_iters, int a_iters, unsigned long long* counters, float* sink) {
    if (threadIdx.x < 256) victim_body<V>(v_iters, counters);
    else { const float r = aggressor_body<A>(a_iters); if (r == 12345.678f) sink[threadIdx.x] = r; }
}
// one kernel, roles by workgroup parity
template <int V, int A>
__global__ __launch_bounds__(256, 2) void same_kernel_kernel(int v_iters, int a_iters, unsigned long long* counters, float* sink) {
    extern __shared__ char smem[];
    if (blockIdx.x & 1) { const float r = aggressor_body<A>(a_iters); if (r == 12345.678f) sink[threadIdx.x] = r; }
    else victim_body<V>(v_iters, counters);
}

static hipStream_t sa, sb;
static unsigned long long* d_counters;
static float* d_sink;
static int reps = 20;

template <int V, int A>
void run_pair(const char* vname, const char* aname, int lds_victim, int lds_aggr) {
    CHECK(hipMemset(d_counters, 0, 16));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(victim_kernel<V>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(aggressor_kernel<A>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    for (int rep = 0; rep < reps; ++rep) {
        if (A != 0) for (int k = 0; k < 2; ++k) hipLaunchKernelGGL(aggressor_kernel<A>, dim3(512), dim3(256), lds_aggr, sa, 3000, d_sink);
        for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(victim_kernel<V>, dim3(512), dim3(256), lds_victim, sb, 40, d_counters);
        CHECK(hipDeviceSynchronize());
    }
    unsigned long long c[2];
    CHECK(hipMemcpy(c, d_counters, 16, hipMemcpyDeviceToHost));
    printf("pair        victim %-34s aggressor %-30s lds %6d/%6d : %10llu of %llu lane-checks differ\n", vname, aname, lds_victim, lds_aggr, c[0], c[1]);
    fflush(stdout);
}
template <int V, int A>
void run_same_wg(const char* vname, const char* aname) {
    CHECK(hipMemset(d_counters, 0, 16));
    for (int rep = 0; rep < reps; ++rep) {
        for (int k = 0; k < 4; ++k) hipLaunchKernelGGL((same_wg_kernel<V, A>), dim3(256), dim3(512), 0, sb, 40, 400, d_counters, d_sink);
        CHECK(hipDeviceSynchronize());
    }
    unsigned long long c[2];
    CHECK(hipMemcpy(c, d_counters, 16, hipMemcpyDeviceToHost));
    printf("same-wg     victim %-34s aggressor %-30s                   : %10llu of %llu lane-checks differ\n", vname, aname, c[0], c[1]);
    fflush(stdout);
}
template <int V, int A>
void run_same_kernel(const char* vname, const char* aname) {
    CHECK(hipMemset(d_counters, 0, 16));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(same_kernel_kernel<V, A>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    for (int rep = 0; rep < reps; ++rep) {
        for (int k = 0; k < 4; ++k) hipLaunchKernelGGL((same_kernel_kernel<V, A>), dim3(1024), dim3(256), 65536, sb, 40, 400, d_counters, d_sink);
        CHECK(hipDeviceSynchronize());
    }
    unsigned long long c[2];
    CHECK(hipMemcpy(c, d_counters, 16, hipMemcpyDeviceToHost));
    printf("same-kernel victim %-34s aggressor %-30s                   : %10llu of %llu lane-checks differ\n", vname, aname, c[0], c[1]);
    fflush(stdout);
}

int main(int argc, char** argv) {
    if (argc > 1) reps = atoi(argv[1]);
    CHECK(hipStreamCreate(&sa));
    CHECK(hipStreamCreate(&sb));
    CHECK(hipMalloc(reinterpret_cast<void**>(&d_counters), 64));
    CHECK(hipMalloc(reinterpret_cast<void**>(&d_sink), 4096));
    const char* vn[] = {"pk fma/mul/add f32 + op_sel", "pk fma/mul/add f32 plain", "v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32",
                        "v_pk_mov_b32", "v_fma_f64", "v_pk_fma_f16", "v_fma_f32 (control)", "v_mul/add_f32 (control)",
                        "class 0, dst distinct from srcs", "op_sel mul alone, dst distinct", "op_sel fma alone, dst = src0",
                        "op_sel add alone, dst = src0", "op_sel fma alone, dst distinct", "neg modifiers only", "broadcast op_sel, dst = src0",
                        "op_sel add alone, dst distinct", "compiler complex math (no asm)", "class 0 + s_nop 0 between", "class 0 + s_nop 1 between",
                        "swizzled forms, independent data", "class 0 + s_nop 7 between",
                        "-", "-", "-", "-", "-", "-", "-", "add src0 swap", "add src0 bcast-lo", "add src0 bcast-hi", "add src1 swap", "add src1 bcast-lo", "add src1 bcast-hi", "-", "-", "-", "mul src0 swap", "mul src0 bcast-lo", "mul src0 bcast-hi", "mul src1 swap", "mul src1 bcast-lo", "mul src1 bcast-hi", "-", "-", "-", "fma src0 swap", "fma src0 bcast-lo", "fma src0 bcast-hi", "fma src1 swap", "fma src1 bcast-lo", "fma src1 bcast-hi", "fma src2 swap", "fma src2 bcast-lo", "fma src2 bcast-hi"};
    const char* an[] = {"none", "mfma 16x16x32 bf16", "mfma 32x32x16 bf16", "mfma 32x32x2 f32", "mfma f64 16x16x4", "mfma i8 16x16x64",
                        "mfma 16x16x32 bf16 (AGPR acc)", "v_exp_f32 (control)", "mfma 16x16x32 f16", "mfma 16x16x16 f16",
                        "mfma 32x32x16 bf16 x4 acc", "mfma 16x16x32 bf16 x4 acc"};
#define PAIR(V, A) run_pair<V, A>(vn[V], an[A], 65536, 65536)
    // every victim class quiet, then beside the bf16 MFMA aggressor
    PAIR(0, 0); PAIR(0, 1); PAIR(1, 1); PAIR(2, 1); PAIR(3, 1); PAIR(4, 1); PAIR(5, 1); PAIR(6, 1); PAIR(7, 1); PAIR(8, 1); PAIR(9, 1);
    // the op_sel victim beside every aggressor class
    PAIR(0, 2); PAIR(0, 3); PAIR(0, 4); PAIR(0, 5); PAIR(0, 6); PAIR(0, 7);
    PAIR(6, 4);
    // which modifier, and does the destination have to overlap a source?
    PAIR(10, 1); PAIR(11, 1); PAIR(12, 1); PAIR(13, 1); PAIR(14, 1); PAIR(15, 1); PAIR(16, 1); PAIR(17, 1); PAIR(18, 1); PAIR(18, 0);
    // more matrix instruction forms beside the op_sel victim
    PAIR(19, 1); PAIR(20, 1); PAIR(22, 1); PAIR(21, 1);
    // every (instruction, source, half selection) on its own, destination distinct
    PAIR(30, 1); PAIR(31, 1); PAIR(32, 1); PAIR(33, 1); PAIR(34, 1); PAIR(35, 1); PAIR(39, 1); PAIR(40, 1); PAIR(41, 1); PAIR(42, 1); PAIR(43, 1); PAIR(44, 1); PAIR(48, 1); PAIR(49, 1); PAIR(50, 1); PAIR(51, 1); PAIR(52, 1); PAIR(53, 1); PAIR(54, 1); PAIR(55, 1); PAIR(56, 1);
    PAIR(0, 8); PAIR(0, 9); PAIR(0, 10); PAIR(0, 11); PAIR(18, 10); PAIR(18, 11);
    // LDS sizes: can they still co-reside?  (victim 160 KiB: alone on its CU)
    run_pair<0, 1>(vn[0], an[1], 163840, 65536);
    run_pair<0, 1>(vn[0], an[1], 0, 0);
    run_pair<1, 2>(vn[1], an[2], 0, 0);
    // two waves per SIMD inside ONE workgroup / one kernel
    run_same_wg<0, 1>(vn[0], an[1]);
    run_same_wg<1, 2>(vn[1], an[2]);
    run_same_wg<0, 0>(vn[0], an[0]);
    run_same_kernel<0, 1>(vn[0], an[1]);
    run_same_kernel<1, 2>(vn[1], an[2]);
    return 0;
}
