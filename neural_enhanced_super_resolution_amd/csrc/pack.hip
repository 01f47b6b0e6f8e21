// Layout kernels at the edges of the network (HBM-bound, one pass each).
//
// pack_input: RRDBNet.forward's `pixel_unshuffle(x, scale)` (basicsr arch_util:
//   x.view(b,c,h,s,w,s).permute(0,1,3,5,2,4) -> channel index c*s*s + sy*s + sx) fused with the
//   NCHW -> NHWC change of layout and the zero padding of conv_first's input channels to the
//   kernel's K-group.  With a u8 HWC source it also performs RealESRGANer.enhance's
//   `img.astype(float32) / 255` and the BGR->RGB flip (cv2.cvtColor), or nesr/nesr.py:851-857
//   (no flip).
#include <hip/hip_bf16.h>

#include "nesr_kernels.h"

namespace nesr {
namespace {

__device__ inline uint16_t f2bf(float f) {
    __hip_bfloat16 b = __float2bfloat16(f);
    return *reinterpret_cast<uint16_t*>(&b);
}
__device__ inline float bf2f(uint16_t u) { return __uint_as_float(((unsigned)u) << 16); }
// f32 -> (hi, lo) half pair of the f16x2 path, x = hi + lo * 2^-11 (conv3x3_f16x2.hip); returns false if x
// does not fit (|x| > 65504, NaN, Inf: clamped, and the caller raises the context's sticky range flag)
__device__ inline bool f2hl(float f, uint16_t& hi, uint16_t& lo) {
    const float x = fminf(fmaxf(f, -65504.f), 65504.f);
    const _Float16 h = (_Float16)x;
    const _Float16 l = (_Float16)((x - (float)h) * 2048.f);
    hi = __builtin_bit_cast(uint16_t, h);
    lo = __builtin_bit_cast(uint16_t, l);
    return __builtin_fabsf(f) <= 65504.f;
}
__device__ inline float hl2f(uint16_t hi, uint16_t lo) {
    return fmaf((float)__builtin_bit_cast(_Float16, lo), 1.f / 2048.f, (float)__builtin_bit_cast(_Float16, hi));
}

__global__ __launch_bounds__(256) void pack_input_kernel(PackArgs a) {
    const int s = a.unshuffle;
    const int ho = a.hin / s, wo = a.win / s;
    const size_t total = (size_t)a.n * ho * wo;
    bool bad = false;
    for (size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x; pix < total; pix += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(pix % wo);
        const int y = (int)((pix / wo) % ho);
        const int n = (int)(pix / ((size_t)wo * ho));
        for (int co = 0; co < a.cp; ++co) {
            float v = 0.f;
            const int c = co / (s * s);
            if (c < a.c) {
                const int r = co - c * s * s;
                const int sy = r / s, sx = r - sy * s;
                const int Y = y * s + sy, X = x * s + sx;
                if (a.src_u8) {
                    const int cs = a.flip ? (a.c - 1 - c) : c;
                    const uint8_t* src = static_cast<const uint8_t*>(a.src);
                    v = (float)src[(((size_t)n * a.hin + Y) * a.win + X) * a.c + cs] / 255.0f;
                } else {
                    const float* src = static_cast<const float*>(a.src);
                    v = src[(((size_t)n * a.c + c) * a.hin + Y) * a.win + X];
                }
            }
            const int kg = a.bf16 ? 16 : 8;
            const size_t idx = (size_t)(co / kg) * a.dst_map.chunk + pix * a.dst_map.pix + (co % kg);
            if (a.bf16 == 2) {
                uint16_t hi, lo;
                bad |= !f2hl(v, hi, lo);
                static_cast<uint16_t*>(a.dst)[idx] = hi;
                static_cast<uint16_t*>(a.dst)[idx + 16] = lo;
            } else if (a.bf16)
                static_cast<uint16_t*>(a.dst)[idx] = f2bf(v);
            else
                static_cast<float*>(a.dst)[idx] = v;
        }
    }
    if (bad && a.status) __hip_atomic_store(a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const void* src, int bf16, Map map, int n, int c, int h, int w, float* dst) {
    const size_t total = (size_t)n * c * h * w;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % w);
        const int y = (int)((i / w) % h);
        const int ch = (int)((i / ((size_t)w * h)) % c);
        const int nn = (int)(i / ((size_t)w * h * c));
        const size_t pix = ((size_t)nn * h + y) * w + x;
        const int kg = bf16 ? 16 : 8;
        const size_t idx = (size_t)(ch / kg) * map.chunk + pix * map.pix + (ch % kg);
        if (bf16 == 2)
            dst[i] = hl2f(static_cast<const uint16_t*>(src)[idx], static_cast<const uint16_t*>(src)[idx + 16]);
        else
            dst[i] = bf16 ? bf2f(static_cast<const uint16_t*>(src)[idx]) : static_cast<const float*>(src)[idx];
    }
}

// ---- the tiles of a frame: RealESRGANer.tile_process's `input_tile = self.img[:, :, y0:y1, x0:x1]` for every tile at once, with
// enhance()'s `img / 255` and BGR->RGB in front (cut), and its paste of every tile's un-padded centre with enhance()'s
// clamp(0, 1), RGB->BGR, x255 and round behind (paste) -- the float canvas of the whole frame never exists.
__global__ __launch_bounds__(256) void cut_tiles_kernel(TileIo t) {
    const int n = blockIdx.z, y = blockIdx.y, x = blockIdx.x * 256 + threadIdx.x;
    if (x >= t.Ws) return;
    const int* d = t.desc + 8 * n;           // y0, x0, h, w of the window in the frame
    const bool in = y < d[2] && x < d[3];
    float* dst = t.tiles + (((size_t)n * 3) * t.Hs + y) * t.Ws + x;
    const uint8_t* src = t.frame + ((size_t)(d[0] + (in ? y : 0)) * t.frame_w + d[1] + (in ? x : 0)) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = in ? (float)src[t.flip ? 2 - c : c] / 255.0f : 0.f;
        if (t.round) v = (float)(_Float16)v;      // RealESRGANer(half=True): `self.img = self.img.half()` before the tiles are cut
        dst[(size_t)c * t.Hs * t.Ws] = v;
    }
}
__global__ __launch_bounds__(256) void paste_tiles_kernel(TileIo t) {
    const int n = blockIdx.z, y = blockIdx.y, x = blockIdx.x * 256 + threadIdx.x;
    const int* d = t.desc + 8 * n;           // crop origin (y, x) inside the tile's output, crop size (h, w), row pitch, offset lo / hi of the destination
    if (y >= d[2] || x >= d[3]) return;
    const float* src = t.tiles + (((size_t)n * 3) * t.Hs + d[0] + y) * t.Ws + d[1] + x;
    uint8_t* dst = t.frame + (((size_t)(unsigned)d[6] << 32) | (size_t)(unsigned)d[5]) + (size_t)y * d[4] + (size_t)x * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = src[(size_t)c * t.Hs * t.Ws];
        if (t.round & 2) v = (float)(_Float16)v;      // RealESRGANer(half=True): the network's output is an fp16 tensor upstream
        float q = fminf(fmaxf(v, 0.f), 1.f) * 255.0f;
        q = (t.round & 1) ? rintf(q) : truncf(q);
        dst[t.flip ? 2 - c : c] = (uint8_t)q;
    }
}

}  // namespace

hipError_t launch_cut_tiles(const TileIo& t, int n, int maxh, int maxw, hipStream_t s) {
    (void)maxh; (void)maxw;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(cut_tiles_kernel, dim3((t.Ws + 255) / 256, t.Hs, n), dim3(256), 0, s, t);
    return hipGetLastError();
}
hipError_t launch_paste_tiles(const TileIo& t, int n, int maxh, int maxw, hipStream_t s) {
    if (n <= 0 || maxh <= 0 || maxw <= 0) return hipSuccess;
    hipLaunchKernelGGL(paste_tiles_kernel, dim3((maxw + 255) / 256, maxh, n), dim3(256), 0, s, t);
    return hipGetLastError();
}

hipError_t launch_pack_input(const PackArgs& a, hipStream_t s) {
    const size_t total = (size_t)a.n * (a.hin / a.unshuffle) * (a.win / a.unshuffle);
    if (total == 0) return hipSuccess;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(pack_input_kernel, dim3(blocks), dim3(256), 0, s, a);
    return hipGetLastError();
}

// The f16-pair path's range word is scoped to one forward: what an earlier forward left in word 0 (nobody asked
// nesr_check_range about it) moves to word 3 before this forward's first kernel, so conv_last does not poison a valid frame
// with an older frame's NaN, and the older error is still reported once.
__global__ void status_latch_kernel(unsigned* st) {
    if (st[0]) { st[3] = 1u; st[0] = 0u; }
}
hipError_t launch_status_latch(unsigned* st, hipStream_t s) {
    hipLaunchKernelGGL(status_latch_kernel, dim3(1), dim3(1), 0, s, st);
    return hipGetLastError();
}

hipError_t launch_nhwc_to_nchw(const void* src, int bf16, Map map, int n, int c, int h, int w, float* dst, hipStream_t s) {
    const size_t total = (size_t)n * c * h * w;
    if (total == 0) return hipSuccess;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(blocks), dim3(256), 0, s, src, bf16, map, n, c, h, w, dst);
    return hipGetLastError();
}

}  // namespace nesr
