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


// ---------------------------------------------------------------------------------------------------------------------------------
// cv2.createCLAHE(clipLimit = 2.0, tileGridSize = (8, 8)).apply(L) of SuperResolutionPipeline._preprocess_image (nesr/nesr.py:680-684),
// restated from OpenCV's clahe.cpp (CLAHE_CalcLut_Body / CLAHE_Interpolation_Body; oracle/cv2_ref.py: clahe_u8; PARITY UNPINNED against
// cv2).  Two launches: one workgroup per tile builds the tile's histogram in LDS (the image counts as padded by BORDER_REFLECT_101
// to a multiple of the grid -- both sides, only when one of them does not divide), clips it, hands the clipped counts back out
// (evenly, then one more at every `step`-th bin), integrates and scales to the look-up table; then every pixel blends the four
// surrounding tiles' tables.  The float arithmetic is the torch composition's it replaces, operation by operation (single-precision
// multiplies and adds that are never contracted, the division by the tile size as a multiplication by its float reciprocal, round
// half to even), so the two agree bit for bit (tests/test_gpu_filters.py).  Byte work: one read of the plane per launch, one write.
namespace {

// single-precision operations that round by themselves: hipcc contracts a * b + c into an fma also when it is spelled
// __fadd_rn(__fmul_rn(a, b), c) (and `#pragma clang fp contract(off)` does not reach those intrinsics), the torch
// operations this replaces are separate kernels and never do
__device__ __forceinline__ float mul_rn(float a, float b) { float r; asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float add_rn(float a, float b) { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float sub_rn(float a, float b) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

__global__ __launch_bounds__(256) void clahe_lut_kernel(const uint8_t* __restrict__ src, int h, int w, int th, int tw, int gx, int clip, float scale,
                                                       float* __restrict__ lut) {
    __shared__ int hist[256];
    __shared__ int red[256];
    const int t = threadIdx.x, tile = blockIdx.x, ty = tile / gx, tx = tile - ty * gx;
    hist[t] = 0;
    __syncthreads();
    const int area = th * tw;
    for (int i = t; i < area; i += 256) {
        const int yy = ty * th + i / tw, xx = tx * tw + i % tw;
        int sy = yy >= h ? 2 * (h - 1) - yy : yy, sx = xx >= w ? 2 * (w - 1) - xx : xx;      // BORDER_REFLECT_101 at the bottom / right
        sy = sy < 0 ? 0 : (sy > h - 1 ? h - 1 : sy);
        sx = sx < 0 ? 0 : (sx > w - 1 ? w - 1 : sx);
        atomicAdd(&hist[src[(size_t)sy * w + sx]], 1);
    }
    __syncthreads();
    int v = hist[t];
    const int over = v > clip ? v - clip : 0;
    v = v > clip ? clip : v;
    red[t] = over;
    __syncthreads();
    for (int s2 = 128; s2 > 0; s2 >>= 1) {
        if (t < s2) red[t] += red[t + s2];
        __syncthreads();
    }
    const int clipped = red[0];
    __syncthreads();
    const int batch = clipped / 256, residual = clipped - batch * 256;
    v += batch;
    if (residual > 0) {
        int step = 256 / residual;
        step = step < 1 ? 1 : step;
        if (t % step == 0 && t / step < residual) v += 1;
    }
    // inclusive prefix sum over the 256 bins
    red[t] = v;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const int add = t >= off ? red[t - off] : 0;
        __syncthreads();
        red[t] += add;
        __syncthreads();
    }
    float q = rintf(mul_rn((float)red[t], scale));
    q = q < 0.f ? 0.f : (q > 255.f ? 255.f : q);
    lut[(size_t)tile * 256 + t] = q;
}

__global__ __launch_bounds__(256) void clahe_apply_kernel(const uint8_t* __restrict__ src, int h, int w, float inv_th, float inv_tw, int gx, int gy,
                                                         const float* __restrict__ lut, uint8_t* __restrict__ dst) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const float ys = sub_rn(mul_rn((float)y, inv_th), 0.5f), xs = sub_rn(mul_rn((float)x, inv_tw), 0.5f);
    const float fy = floorf(ys), fx = floorf(xs);
    const float ya = sub_rn(ys, fy), xa = sub_rn(xs, fx);
    int ty1 = (int)fy, tx1 = (int)fx;
    int ty2 = ty1 + 1, tx2 = tx1 + 1;
    ty2 = ty2 < 0 ? 0 : (ty2 > gy - 1 ? gy - 1 : ty2);
    tx2 = tx2 < 0 ? 0 : (tx2 > gx - 1 ? gx - 1 : tx2);
    ty1 = ty1 < 0 ? 0 : (ty1 > gy - 1 ? gy - 1 : ty1);
    tx1 = tx1 < 0 ? 0 : (tx1 > gx - 1 ? gx - 1 : tx1);
    const int v = src[(size_t)y * w + x];
    const float l11 = lut[((size_t)ty1 * gx + tx1) * 256 + v], l12 = lut[((size_t)ty1 * gx + tx2) * 256 + v];
    const float l21 = lut[((size_t)ty2 * gx + tx1) * 256 + v], l22 = lut[((size_t)ty2 * gx + tx2) * 256 + v];
    const float ixa = sub_rn(1.f, xa), iya = sub_rn(1.f, ya);
    const float top = add_rn(mul_rn(l11, ixa), mul_rn(l12, xa)), bot = add_rn(mul_rn(l21, ixa), mul_rn(l22, xa));
    float r = rintf(add_rn(mul_rn(top, iya), mul_rn(bot, ya)));
    r = r < 0.f ? 0.f : (r > 255.f ? 255.f : r);
    dst[(size_t)y * w + x] = (uint8_t)r;
}

}  // namespace

hipError_t launch_clahe(const uint8_t* src, int h, int w, int gx, int gy, int th, int tw, int clip, float scale, float inv_th, float inv_tw, float* lut,
                        uint8_t* dst, hipStream_t s) {
    hipLaunchKernelGGL(clahe_lut_kernel, dim3((unsigned)(gx * gy)), dim3(256), 0, s, src, h, w, th, tw, gx, clip, scale, lut);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(clahe_apply_kernel, dim3((unsigned)((w + 63) / 64), (unsigned)((h + 3) / 4)), dim3(256), 0, s, src, h, w, inv_th, inv_tw, gx, gy, lut, dst);
    return hipGetLastError();
}

}  // namespace nesr
