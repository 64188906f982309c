// Are LDS accesses past a workgroup's allocation dropped, or do they land in a neighbouring workgroup's LDS?
// victim: 80 KiB of LDS filled with a pattern, re-checked for ~300 us; aggressor (64 KiB allocated) writes past its
// allocation with ds_write and with global_load_lds.   hipcc --offload-arch=gfx950 -O2 lds_oob_probe.hip -o lds_oob_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void victim(unsigned long long* bad, int spins) {
    extern __shared__ unsigned int lds[];
    const int n = 80 * 1024 / 4;
    for (int i = threadIdx.x; i < n; i += blockDim.x) lds[i] = 0xA5000000u + i;
    __syncthreads();
    for (int s = 0; s < spins; ++s) {
        for (int i = threadIdx.x; i < n; i += blockDim.x)
            if (lds[i] != 0xA5000000u + i) { atomicAdd(bad, 1ull); lds[i] = 0xA5000000u + i; }
        __syncthreads();
    }
}

__global__ void aggressor(const unsigned int* src, int mode, int spins, unsigned long long* sink) {
    extern __shared__ unsigned int lds[];
    unsigned int acc = 0;
    for (int s = 0; s < spins; ++s) {
        if (mode == 1) {                 // ds_write past the 64 KiB allocation: offsets 64 KiB .. 160 KiB
            for (int i = threadIdx.x; i < 96 * 1024 / 4; i += blockDim.x) {
                const unsigned int addr = 64 * 1024 + i * 4;
                asm volatile("ds_write_b32 %0, %1" :: "v"(addr), "v"(0xDEAD0000u + i) : "memory");
            }
        } else if (mode == 2) {          // LDS-DMA past the allocation
            for (int piece = 0; piece < 96; ++piece) {
                const unsigned int m0 = 64 * 1024 + piece * 1024;
                const unsigned int* p = src + threadIdx.x % 64;
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" :: "v"(p), "s"(m0) : "memory", "m0");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {                         // in-bounds traffic only
            for (int i = threadIdx.x; i < 64 * 1024 / 4; i += blockDim.x) lds[i] = i + s;
        }
        __syncthreads();
        acc += lds[threadIdx.x];
    }
    if (acc == 0x12345678u) atomicAdd(sink, 1ull);
}

int main() {
    unsigned long long *bad, *sink;
    unsigned int* src;
    hipMalloc(&bad, 8); hipMalloc(&sink, 8); hipMalloc(&src, 4096);
    hipMemset(src, 0x77, 4096);
    hipStream_t a, b;
    hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    hipFuncSetAttribute(reinterpret_cast<const void*>(victim), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    for (int mode = 0; mode < 3; ++mode) {
        hipMemset(bad, 0, 8);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(victim, dim3(256), dim3(256), 80 * 1024, a, bad, 400);
        hipLaunchKernelGGL(aggressor, dim3(2048), dim3(256), 64 * 1024, b, src, mode, 20, sink);
        hipDeviceSynchronize();
        unsigned long long h = 0;
        hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
        printf("aggressor mode %d (%s): victim saw %llu changed dwords (%s)\n", mode,
               mode == 0 ? "in-bounds writes" : mode == 1 ? "ds_write past its allocation" : "global_load_lds past its allocation", h, hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
