// Probe: GPU-side cost of a dependent back-to-back kernel launch on one stream.
//  (1) empty kernels by grid / block / LDS size (may be bounded by the CPU's enqueue rate);
//  (2) kernels that spin ~10 us each, so the GPU is the bottleneck: (time of N launches - N x spin) / N is what the
//      device itself spends between two dependent kernels; the same through a captured hipGraph.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Args { char pad[232]; };
__global__ void empty(Args a) { extern __shared__ char sm[]; if (a.pad[0] == 77) sm[threadIdx.x] = 1; }
__global__ void spin(Args a, long long ticks) {   // ticks of the 100 MHz s_memrealtime counter
    extern __shared__ char sm[];
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < ticks) {}
    if (a.pad[0] == 77) sm[threadIdx.x] = 1;
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&empty), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&spin), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    Args a = {};
    const int cfg[][3] = {{1, 64, 0}, {256, 256, 0}, {256, 256, 80384}, {512, 256, 80384}, {2048, 256, 80384}};
    for (auto& c : cfg) {
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(empty, dim3(c[0]), dim3(c[1]), c[2], 0, a);
        hipDeviceSynchronize();
        const double t0 = now_us();
        const int N = 4000;
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(empty, dim3(c[0]), dim3(c[1]), c[2], 0, a);
        hipDeviceSynchronize();
        printf("empty  grid %5d block %4d lds %6d : %.2f us per launch\n", c[0], c[1], c[2], (now_us() - t0) / N);
    }
    const int N = 1000;
    for (long long ticks : {1000LL, 2000LL}) {   // 10 us, 20 us
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(spin, dim3(256), dim3(256), 80384, 0, a, ticks);
        hipDeviceSynchronize();
        double t0 = now_us();
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(spin, dim3(256), dim3(256), 80384, 0, a, ticks);
        hipDeviceSynchronize();
        const double per = (now_us() - t0) / N;
        printf("spin %2lld us x %d launches: %.2f us per launch -> %.2f us between dependent kernels\n", ticks / 100, N, per, per - ticks / 100.0);
        hipStream_t s; hipStreamCreate(&s);
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(spin, dim3(256), dim3(256), 80384, s, a, ticks);
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, s); hipStreamSynchronize(s);
        t0 = now_us();
        hipGraphLaunch(ge, s); hipStreamSynchronize(s);
        const double perg = (now_us() - t0) / N;
        printf("  same as one hipGraph: %.2f us per node -> %.2f us between dependent kernels\n", perg, perg - ticks / 100.0);
    }
    return 0;
}
