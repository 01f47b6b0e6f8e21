// Probe: GPU-side cost of a back-to-back dependent kernel launch on one stream, by grid / block / LDS size.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
struct Args { char pad[232]; };
__global__ void empty(Args a) { extern __shared__ char sm[]; if (a.pad[0] == 77) sm[threadIdx.x] = 1; }
int main() {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&empty), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    Args a = {};
    const int cfg[][3] = {{1, 64, 0}, {256, 256, 0}, {256, 256, 80384}, {256, 512, 80384}, {512, 256, 80384}, {2048, 256, 0}, {2048, 256, 80384}, {256, 256, 117248}};
    for (auto& c : cfg) {
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(empty, dim3(c[0]), dim3(c[1]), c[2], 0, a);
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        const int N = 4000;
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(empty, dim3(c[0]), dim3(c[1]), c[2], 0, a);
        hipDeviceSynchronize();
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
        printf("grid %5d block %4d lds %6d : %.2f us per launch\n", c[0], c[1], c[2], us);
    }
    return 0;
}
