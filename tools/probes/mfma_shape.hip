// Probe: sustained FLOP/s and in-kernel clock of LDS-fed f16 MFMA loops, 32x32x16 vs 16x16x32, random data.
// One workgroup of 4 or 8 waves per CU; every MFMA operand is re-read from LDS with ds_read_b128.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
struct Out { unsigned long long cyc, rt; float v; };

template <int SHAPE>
__global__ __launch_bounds__(512) void k(const h8* src, Out* out, int iters) {
    extern __shared__ h8 sm[];   // 64 KB of operands
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) sm[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float res = 0.f;
    if (SHAPE == 32) {
        f16v acc[4] = {};
        for (int q = 0; q < 4; ++q) acc[q][0] = (float)(q + lane);   // distinct chains: nothing to fold
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const h8 a = sm[(lane + 64 * ((it + j) & 31)) & 4095], b = sm[(lane + 64 * ((it * 3 + j + 7) & 31) + 2048) & 4095];
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[q], 0, 0, 0);
            }
        }
        for (int q = 0; q < 4; ++q) res += acc[q][0];
    } else {
        f4v acc[16] = {};
        for (int q = 0; q < 16; ++q) acc[q][0] = (float)(q + lane);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const h8 a = sm[(lane + 64 * ((it + j) & 31)) & 4095], b = sm[(lane + 64 * ((it * 3 + j + 7) & 31) + 2048) & 4095];
                // same FLOPs per operand pair as above: 8 MFMAs of 16x16x32 = 4 of 32x32x16
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[q + 8 * (j & 1)] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[q + 8 * (j & 1)], 0, 0, 0);
            }
        }
        for (int q = 0; q < 16; ++q) res += acc[q][0];
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[blockIdx.x].cyc = t1 - t0; out[blockIdx.x].rt = r1 - r0; out[blockIdx.x].v = res; }
}

int main() {
    std::vector<_Float16> h(4096 * 8);
    srand(1);
    for (auto& x : h) x = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2.f);
    h8* d; Out* o;
    hipMalloc(&d, h.size() * 2); hipMalloc(&o, sizeof(Out) * 256);
    hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (int waves : {4, 8}) for (int shape : {32, 16, 32, 16}) {
        const int iters = 20000;
        for (int rep = 0; rep < 2; ++rep) {
            if (shape == 32) hipLaunchKernelGGL(k<32>, dim3(256), dim3(64 * waves), 65536, 0, d, o, iters);
            else hipLaunchKernelGGL(k<16>, dim3(256), dim3(64 * waves), 65536, 0, d, o, iters);
            hipDeviceSynchronize();
        }
        std::vector<Out> r(256);
        hipMemcpy(r.data(), o, sizeof(Out) * 256, hipMemcpyDeviceToHost);
        double cyc = 0, rt = 0;
        for (auto& x : r) { cyc += x.cyc; rt += x.rt; }
        cyc /= 256; rt /= 256;
        const double flops = 256.0 * waves * iters * 8 * 4 * 32768.0;
        printf("waves/CU %d  shape %s : %.1f us  clock %.2f GHz  %.0f TFLOP/s  cycles per 32x32x16-equivalent MFMA %.1f\n", waves,
               shape == 32 ? "32x32x16" : "16x16x32", rt / 100.0, cyc / rt * 0.1, flops / (rt / 100.0 * 1e-6) / 1e12, cyc / (iters * 32.0) / (waves / 4));
    }
    return 0;
}
