// Probe: operand / result layout of v_mfma_f32_16x16x32_f16 and the lanes v_permlane16_swap exchanges.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* A, const float* B, float* C, unsigned* P) {   // A[16][32], B[32][16], C[16][16]
    const int l = threadIdx.x, j = l & 15, g = l >> 4;
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)A[j * 32 + 8 * g + i]; b[i] = (_Float16)B[(8 * g + i) * 16 + j]; }
    f4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) C[(4 * g + i) * 16 + j] = c[i];
    auto r = __builtin_amdgcn_permlane16_swap((unsigned)(100 + l), (unsigned)(200 + l), false, false);
    P[l] = r[0]; P[64 + l] = r[1];
}
int main() {
    float hA[512], hB[512], hC[256], ref[256];
    srand(3);
    for (auto& x : hA) x = (float)(rand() % 7 - 3);
    for (auto& x : hB) x = (float)(rand() % 5 - 2);
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) { float s = 0; for (int kk = 0; kk < 32; ++kk) s += hA[m * 32 + kk] * hB[kk * 16 + n]; ref[m * 16 + n] = s; }
    float *dA, *dB, *dC; unsigned* dP; unsigned hP[128];
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dC, 1024); hipMalloc(&dP, 512);
    hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dP);
    hipMemcpy(hC, dC, 1024, hipMemcpyDeviceToHost); hipMemcpy(hP, dP, 512, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 256; ++i) bad += hC[i] != ref[i];
    printf("mfma 16x16x32 f16 layout (A row=l&15 k=8(l>>4)+i; B col=l&15; C row=4(l>>4)+i col=l&15): %s (%d mismatches)\n", bad ? "WRONG" : "ok", bad);
    printf("permlane16_swap(vdst=100+l, src=200+l): r0 lanes 0,16,32,48 = %u %u %u %u ; r1 = %u %u %u %u\n", hP[0], hP[16], hP[32], hP[48], hP[64], hP[80], hP[96], hP[112]);
    return 0;
}
