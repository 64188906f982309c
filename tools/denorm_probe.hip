// Does the gfx950 matrix pipe keep fp16 subnormal OPERANDS (needed by a hi + lo split of fp32 values into two fp16
// halves: lo = x - fp16(x) is subnormal for |x| < ~0.12)?  One wave, v_mfma_f32_32x32x16_f16 and 16x16x32_f16:
// A = all `a`, B = all `b`  ->  every C element = 16 (or 32) * a * b.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
__global__ void probe(float a, float b, float* out) {
    h8 va, vb;
    for (int i = 0; i < 8; ++i) { va[i] = (_Float16)a; vb[i] = (_Float16)b; }
    f16v c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(va, vb, c, 0, 0, 0);
    f4v d = {0};
    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(va, vb, d, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = c[0]; out[1] = d[0]; out[2] = (float)va[0]; }
}
int main() {
    float* out; hipMalloc(&out, 16);
    const float as[] = {1.0f, 6.1035156e-5f, 3.0517578e-5f, 5.9604645e-8f, 1.1920929e-7f * 3};
    for (float a : as) {
        for (float b : {1.0f, 1024.0f}) {
            probe<<<1, 64>>>(a, b, out);
            float h[3]; hipMemcpy(h, out, 12, hipMemcpyDeviceToHost);
            printf("a=%.9g b=%g: mfma32x32x16 -> %.9g (exact %.9g)  mfma16x16x32 -> %.9g (exact %.9g)  a as fp16 %.9g\n", a, b, h[0], 16.0 * a * b, h[1], 32.0 * a * b, h[2]);
        }
    }
    return 0;
}
