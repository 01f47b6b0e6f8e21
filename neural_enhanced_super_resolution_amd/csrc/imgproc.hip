// The pipeline's pre-filter that costs the time: cv2.fastNlMeansDenoisingColored(image, None, h, h, 7, 21)
// of SuperResolutionPipeline._preprocess_image (nesr/nesr.py:674) = COLOR_LBGR2Lab, non-local means on L and on (a, b), COLOR_Lab2LBGR.
// This file holds the non-local means itself (OpenCV's FastNlMeansDenoisingInvoker with DistSquared, restated in oracle/cv2_ref.py;
// PARITY UNPINNED against cv2, which is not installed): for every pixel and each of the 21 x 21 offsets of the search window the
// summed squared difference of the 7 x 7 template windows over all channels is binned (>> 6: OpenCV's "almost" distance), looked up
// in a table of integer weights round(M exp(-d / (h^2 C))) (0 below M / 1000), and the search window's pixels are averaged with these
// weights in integer arithmetic with a rounded division.  HBM-bound in principle (one read, one write of the plane); in practice
// LDS / barrier work: 441 offsets x (squared differences, 7-wide row sums, 7-tall column sums) per 16 x 16 tile.
#include "nesr_kernels.h"

namespace nesr {
namespace {

constexpr int NT = 16, TR = 3, SR = 10, NB = TR + SR, NS = NT + 2 * NB, ND = NT + 2 * TR;   // tile, template / search radius, halo, LDS side, template-extended side

__device__ __forceinline__ int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * (n - 1) - p;
    return p;
}

template <int C>
__global__ __launch_bounds__(256) void nl_means_kernel(const uint8_t* __restrict__ src, int H, int W, const int* __restrict__ lut, int nbins, int shift,
                                                      uint8_t* __restrict__ dst) {
    __shared__ uint8_t img[C][NS][NS + 2];
    __shared__ int D[ND][ND + 1];
    __shared__ int Hs[ND][NT];
    const int tid = threadIdx.x, oy = tid >> 4, ox = tid & 15;
    const int y0 = blockIdx.y * NT, x0 = blockIdx.x * NT;
    for (int i = tid; i < NS * NS; i += 256) {
        const int ly = i / NS, lx = i - ly * NS;
        const int gy = reflect101(y0 - NB + ly, H), gx = reflect101(x0 - NB + lx, W);     // BORDER_REFLECT_101, as copyMakeBorder in the invoker
#pragma unroll
        for (int c = 0; c < C; ++c) img[c][ly][lx] = src[((size_t)c * H + gy) * W + gx];
    }
    __syncthreads();
    unsigned acc[C], wsum = 0;
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 0;
    for (int dy = -SR; dy <= SR; ++dy)
        for (int dx = -SR; dx <= SR; ++dx) {
            for (int i = tid; i < ND * ND; i += 256) {       // squared differences over the template-extended tile
                const int ty = i / ND, tx = i - ty * ND;
                int d = 0;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const int v = (int)img[c][ty + SR][tx + SR] - (int)img[c][ty + SR + dy][tx + SR + dx];
                    d += v * v;
                }
                D[ty][tx] = d;
            }
            __syncthreads();
            for (int i = tid; i < ND * NT; i += 256) {       // 7-wide row sums
                const int ty = i >> 4, tx = i & 15;
                int sm = 0;
#pragma unroll
                for (int k = 0; k < 2 * TR + 1; ++k) sm += D[ty][tx + k];
                Hs[ty][tx] = sm;
            }
            __syncthreads();
            int dist = 0;
#pragma unroll
            for (int k = 0; k < 2 * TR + 1; ++k) dist += Hs[oy + k][ox];
            int bin = dist >> shift;
            bin = bin < nbins ? bin : nbins - 1;
            const unsigned w = (unsigned)lut[bin];
            wsum += w;
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] += w * (unsigned)img[c][oy + NB + dy][ox + NB + dx];
        }
    const int y = y0 + oy, x = x0 + ox;
    if (y < H && x < W) {
        const unsigned den = wsum ? wsum : 1u;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const unsigned v = (acc[c] + wsum / 2) / den;
            dst[((size_t)c * H + y) * W + x] = (uint8_t)(v > 255u ? 255u : v);
        }
    }
}

}  // namespace

hipError_t launch_nl_means(const uint8_t* src, int C, int H, int W, const int* lut, int nbins, int shift, uint8_t* dst, hipStream_t s) {
    if (H <= 0 || W <= 0) return hipSuccess;
    const dim3 grid((W + NT - 1) / NT, (H + NT - 1) / NT);
    if (C == 1) hipLaunchKernelGGL(nl_means_kernel<1>, grid, dim3(256), 0, s, src, H, W, lut, nbins, shift, dst);
    else if (C == 2) hipLaunchKernelGGL(nl_means_kernel<2>, grid, dim3(256), 0, s, src, H, W, lut, nbins, shift, dst);
    else if (C == 3) hipLaunchKernelGGL(nl_means_kernel<3>, grid, dim3(256), 0, s, src, H, W, lut, nbins, shift, dst);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace nesr
