// LDS staging shared by the gfx950 kernels: XOR-swizzled tiles filled by global -> LDS DMA, and the
// compiler-invisible ds_read_b128 / s_waitcnt pair the fragment streams are built from.
#pragma once

#include "ppg_device.h"

#include <type_traits>

namespace {

// XOR swizzle of a 16-byte slot index inside an LDS tile row; conflict-free
// for the 16-lane service groups of ds_read_b128 when rows are 128 B (mask 7)
// or >= 256 B (mask 15); rows of 64 B keep a 2-way conflict.
template <int ROW_BYTES>
__device__ __forceinline__ int swz(int row, int p) {
    if constexpr (ROW_BYTES >= 256) return p ^ (row & 15);
    else if constexpr (ROW_BYTES == 128) return p ^ (row & 7);
    else return p ^ ((row >> 2) & 3);
}


// LDS byte address (low 32 bits of the flat address of a __shared__ object)
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}

// 16-byte global -> LDS DMA (global_load_lds_dwordx4): the LDS address is the
// wave-uniform base (M0) + lane*16, the global address is per lane.  Issued from
// inline asm in the scalar-base form -- wave-uniform 64-bit base in SGPRs + a
// 32-bit per-lane offset that is never rewritten: hipcc expands the
// __builtin_amdgcn_global_load_lds builtin to a 64-bit VALU add into one VGPR
// pair per piece, reused by the next piece, so every piece waits for the
// previous one to have read its address (-4 % layer kernel, -8 % attention).
__device__ __forceinline__ void glds16_saddr(const char* uniform_base, uint32_t lane_off, uint32_t lds_wave_addr) {
    // (readfirstlane: free where the compiler already knows the value is wave-uniform, and the
    // only way to get SGPR operands where it does not -- both bodies of ffn_mixed_kernel)
    const uint64_t b = reinterpret_cast<uint64_t>(uniform_base);
    const uint64_t base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) |
                          (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(lane_off), "s"(base), "s"(__builtin_amdgcn_readfirstlane(lds_wave_addr)) : "memory", "m0");
}

// Stage a [ROWS][ROW_BYTES] tile (global row stride gstride) into LDS by DMA.
// LDS image: 16-byte slot (row, pp) holds global piece p = swz(row, pp); the
// image is lane-linear per wave-instruction (1 KiB pieces), the swizzle is
// applied on the SOURCE address and again (same involution) on the reads.
//
// Addressing: piece = i*NWAVES + wave.  The swizzle pattern of a piece repeats
// with PERIOD pieces (= 16 rows), so  source(piece) = gsrc
//     + (piece / PERIOD) * 16 rows * gstride            wave-uniform
//     + lane_off[piece % PERIOD]                        per lane, <= V variants per wave
// i.e. a handful of 32-bit lane offsets instead of one 64-bit address per
// piece (which the compiler hoists out of the tile loop and then spills).
template <int ROWS, int ROW_BYTES, int NWAVES>
struct TileDma {
    static constexpr int S = ROW_BYTES / 16;                       // 16-byte slots per row
    static constexpr int PIECES = ROWS * S / 64;
    static constexpr int PER_WAVE = (PIECES + NWAVES - 1) / NWAVES;
    static constexpr int PERIOD = ROW_BYTES >= 256 ? 16 * S / 64 : 1;
    static constexpr int ROWS_PER_PERIOD = PERIOD * 64 / S;        // 16, or rows per piece when PERIOD == 1
    static constexpr int V = PERIOD > NWAVES ? PERIOD / NWAVES : 1;
    static_assert((ROWS * S) % 64 == 0, "tile must be a whole number of 1 KiB pieces");
    static_assert(PERIOD <= NWAVES || PERIOD % NWAVES == 0, "pattern period vs wave count");
    static_assert(ROW_BYTES >= 256 ? ROWS_PER_PERIOD == 16 : true, "swizzle period");

    static __device__ __forceinline__ void run(const char* gsrc, size_t gstride, char* lds, int wave, int lane) {
        uint32_t lane_off[V];
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const int r = (v * NWAVES + wave) % PERIOD;            // pattern index of pieces i = v (mod V)
            const int slot = r * 64 + lane;
            const int row = slot / S, pp = slot % S;
            lane_off[v] = (uint32_t)(row * gstride) + (uint32_t)(swz<ROW_BYTES>(row, pp) << 4);
        }
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i) {
            const int piece = i * NWAVES + wave;
            if (PIECES % NWAVES == 0 || piece < PIECES) {
                const size_t base = (size_t)(piece / PERIOD) * ROWS_PER_PERIOD * gstride;
                glds16_saddr(gsrc + base, lane_off[i % V], lds_addr(lds) + piece * 1024);
            }
        }
    }
};
template <int ROWS, int ROW_BYTES, int NWAVES>
__device__ __forceinline__ void stage_tile(const char* gsrc, size_t gstride, char* lds, int wave, int lane) {
    TileDma<ROWS, ROW_BYTES, NWAVES>::run(gsrc, gstride, lds, wave, lane);
}

// All DMA of this wave landed + workgroup barrier.
__device__ __forceinline__ void dma_wait_barrier() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}


// ds_read_b128 the compiler does not schedule or count: it sinks ordinary LDS
// loads next to their first use and waits lgkmcnt(0) after every one or two
// (a single wave per SIMD then pays the full LDS latency per MFMA pair), and
// it drains in-flight LDS DMA (vmcnt(0)) before LDS loads it cannot prove
// disjoint.  These asm reads are waited for by lgkm_wait<N>() below.
template <int OFF>
__device__ __forceinline__ void ds_read_b128_asm(u32x4& dst, uint32_t addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
// Wait until at most N LDS operations issued after `r`'s read are outstanding;
// `r` is tied to the statement so that no consumer (or copy) of it can be
// placed above the wait.
template <int N>
__device__ __forceinline__ void lgkm_wait(u32x4& r) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(r) : "n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);   // register-only MFMAs stay below too
}

// Address of MFMA fragment i = (kg, blk) = (i / NBLK, i % NBLK) of a swizzled
// [rows][ROW_BYTES] LDS tile (row = blk*16 + idx, 16-byte slot kg*4 + g XOR
// the row swizzle), split into a lane-variable base -- one of VAR variants,
// selected by the low bits of kg that the XOR mixes with lane bits -- and a
// compile-time immediate for the ds_read offset field.
template <int ROW_BYTES, int NBLK>
struct FragLayout {
    static constexpr int VAR = ROW_BYTES >= 256 ? 4 : (ROW_BYTES == 128 ? 2 : 1);
    static constexpr int variant(int i) { return (i / NBLK) % VAR; }
    static constexpr int imm(int i) { return (i % NBLK) * 16 * ROW_BYTES + ((i / NBLK) / VAR) * VAR * 64; }
    static __device__ __forceinline__ uint32_t base(int idx, int g, int j) {
        if constexpr (ROW_BYTES >= 256) return idx * ROW_BYTES + ((((j * 4) ^ (idx & 12)) + ((g ^ idx) & 3)) << 4);
        else if constexpr (ROW_BYTES == 128) return idx * 128 + ((((j * 4) ^ (idx & 4)) + ((g ^ idx) & 3)) << 4);
        else return idx * 64 + ((g ^ ((idx >> 2) & 3)) << 4);
    }
    static __device__ __forceinline__ void bases(uint32_t tile, int idx, int g, uint32_t (&fb)[VAR]) {
#pragma unroll
        for (int j = 0; j < VAR; ++j) fb[j] = tile + base(idx, g, j);
    }
};

// Software-pipelined LDS fragment stream for one wave per SIMD: D reads are
// kept in flight; fragment I is consumed (its MFMAs issued) and its ring slot
// immediately re-armed with fragment I + D.
template <class L, int I, int R, int D, class USE>
__device__ __forceinline__ void lds_stream_step(u32x4 (&ring)[D], const uint32_t (&fb)[L::VAR], USE& use) {
    if constexpr (I < R) {
        lgkm_wait<(R - 1 - I < D - 1) ? (R - 1 - I) : (D - 1)>(ring[I % D]);
        use(std::integral_constant<int, I>{}, ring[I % D]);
        if constexpr (I + D < R) ds_read_b128_asm<L::imm(I + D)>(ring[I % D], fb[L::variant(I + D)]);
        lds_stream_step<L, I + 1, R, D>(ring, fb, use);
    }
}
template <class L, int I, int R, int D>
__device__ __forceinline__ void lds_stream_prime(u32x4 (&ring)[D], const uint32_t (&fb)[L::VAR]) {
    if constexpr (I < D && I < R) {
        ds_read_b128_asm<L::imm(I)>(ring[I], fb[L::variant(I)]);
        lds_stream_prime<L, I + 1, R, D>(ring, fb);
    }
}
// Stream over the NBLK fragments of one K-group kg (runtime-free: kg is a
// constant after unrolling at the call site) of a FragLayout<ROW, NBLK> tile.
template <class L, int KG, int I, int NBLK, int D, class USE>
__device__ __forceinline__ void lds_group_step(u32x4 (&ring)[D], const uint32_t (&fb)[L::VAR], USE& use) {
    if constexpr (I < NBLK) {
        lgkm_wait<(NBLK - 1 - I < D - 1) ? (NBLK - 1 - I) : (D - 1)>(ring[I % D]);
        use(std::integral_constant<int, I>{}, ring[I % D]);
        if constexpr (I + D < NBLK) ds_read_b128_asm<L::imm(KG * NBLK + I + D)>(ring[I % D], fb[L::variant(KG * NBLK + I + D)]);
        lds_group_step<L, KG, I + 1, NBLK, D>(ring, fb, use);
    }
}
template <class L, int KG, int I, int NBLK, int D>
__device__ __forceinline__ void lds_group_prime(u32x4 (&ring)[D], const uint32_t (&fb)[L::VAR]) {
    if constexpr (I < D && I < NBLK) {
        ds_read_b128_asm<L::imm(KG * NBLK + I)>(ring[I], fb[L::variant(KG * NBLK + I)]);
        lds_group_prime<L, KG, I + 1, NBLK, D>(ring, fb);
    }
}
template <class L, int NBLK, int D, class USE>
__device__ __forceinline__ void lds_stream_group(const uint32_t (&fb)[L::VAR], int kg, USE use) {
    u32x4 ring[D];
    // kg is 0 or 1 at every call site (fully unrolled loops)
    if (kg == 0) { lds_group_prime<L, 0, 0, NBLK, D>(ring, fb); lds_group_step<L, 0, 0, NBLK, D>(ring, fb, use); }
    else         { lds_group_prime<L, 1, 0, NBLK, D>(ring, fb); lds_group_step<L, 1, 0, NBLK, D>(ring, fb, use); }
}

template <class L, int R, int D, class USE>
__device__ __forceinline__ void lds_stream(const uint32_t (&fb)[L::VAR], USE use) {
    u32x4 ring[D];
    lds_stream_prime<L, 0, R, D>(ring, fb);
    lds_stream_step<L, 0, R, D>(ring, fb, use);
}

}  // namespace
