// Probe: does v_mfma_f32_32x32x16_f16 on gfx950 honour f16 subnormal INPUTS?  (decides whether the
// lo plane of the f16x2-split f32 path needs scaling).  Build: hipcc --offload-arch=gfx950 -o probe this.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(float* out, float aval, float bval) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)aval; b[i] = (_Float16)bval; }
    f16v c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
    float* d; hipMalloc(&d, 4);
    const float tests[][2] = {{1.f, 1.f}, {9.5367431640625e-07f, 1.f}, {1.f, 9.5367431640625e-07f}, {5.9604644775390625e-08f, 1.f},
                              {6.103515625e-05f, 1.f}, {9.5367431640625e-07f, 9.5367431640625e-07f}};
    for (auto& t : tests) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, t[0], t[1]);
        float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("a=%g b=%g -> %g (exact %g)\n", t[0], t[1], h, 16.0 * (double)t[0] * (double)t[1]);
    }
    return 0;
}
