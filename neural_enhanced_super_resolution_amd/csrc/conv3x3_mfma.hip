// 3x3 convolution as im2col-free implicit GEMM on the gfx950 matrix cores.
//
//   f32  : v_mfma_f32_32x32x2_f32   (exact f32, bitwise a k-ordered fmaf chain; 157 TFLOP/s peak)
//   bf16 : v_mfma_f32_32x32x16_bf16 (bf16 operands, f32 accumulate; ~2.5 PFLOP/s dense peak)
//
// Stands behind every torch.nn.Conv2d(…, 3, 1, 1) of basicsr's RRDBNet.
// The reference runs it in fp32 (half=False: nesr/nesr.py:227, standalone/direct_esrgan.py:125).
//
// GEMM view: M = output pixels, N = output channels, K = 9 taps x Cin.
//   workgroup : 256 threads = 4 waves, output tile 8 rows x 16 cols x all N (32 or 64)
//   wave      : one 32-pixel M-tile (2 rows x 16 cols) x NT 32-wide N-tiles
//   K loop    : chunks of 32 bytes of input channels (8 f32 / 16 bf16); per chunk the
//               (8+2)x(16+2) input halo tile and the 9 x chunk x N weight slab are staged
//               global -> registers -> LDS, double-buffered, one barrier per chunk.
//   A operand : lane (m = lane&31, h = lane>>5) reads the 16 bytes [16h, 16h+16) of its pixel's
//               chunk with one ds_read_b128: f32 -> 4 channels, MFMA k-step j pairs channels
//               {j, 4+j}; bf16 -> 8 channels = exactly the 32x32x16 A fragment (k = 8h + j).
//   B operand : weights pre-packed on the host as [chunk][tap][h][n][16 B] so the matching
//               ds_read_b128 is lane-contiguous (conflict-free).
//   LDS image : activations [h][pixel][16 B] (two planes), so 16 consecutive pixels are 256
//               contiguous bytes.
//   epilogue  : bias, LeakyReLU(0.2), up to two scaled residuals (RDB: x5*0.2+x; RRDB: out*0.2+x),
//               NHWC channel-slice store (the concat-free dense block), optional planar NCHW or
//               clamped/quantised u8 HWC store for conv_last.
#include <hip/hip_bf16.h>

#include "nesr_kernels.h"

namespace nesr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int TH = 8, TW = 16;
constexpr int PH = TH + 2, PW = TW + 2;
constexpr int NPIX = PH * PW;  // 180

template <bool BF>
struct Elem;
template <>
struct Elem<false> {
    typedef float T;
    static constexpr int KG = 8;  // channels per 32-byte chunk
    __device__ static float ld(const T* p) { return *p; }
    __device__ static void st(T* p, float v) { *p = v; }
};
template <>
struct Elem<true> {
    typedef uint16_t T;
    static constexpr int KG = 16;
    __device__ static float ld(const T* p) { return __uint_as_float(((unsigned)*p) << 16); }
    __device__ static void st(T* p, float v) {
        __bf16 b = (__bf16)v;  // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN preserving
        *p = *reinterpret_cast<uint16_t*>(&b);
    }
};

template <bool BF, int NT>
__global__ __launch_bounds__(256) void conv3x3_mfma_kernel(ConvArgs a) {
    typedef typename Elem<BF>::T T;
    constexpr int KG = Elem<BF>::KG;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int IN_F4 = 2 * NPIX;
    constexpr int W_F4 = 9 * 2 * 32 * NT;
    constexpr int STAGE_F4 = IN_F4 + W_F4;
    constexpr int WL = (W_F4 + 255) / 256;
    f32x4* lds = reinterpret_cast<f32x4*>(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int tiles_x = (a.w_ + TW - 1) / TW;
    const int tiles_y = (a.h + TH - 1) / TH;
    int bid = blockIdx.x;
    const int n = bid / (tiles_x * tiles_y);
    bid -= n * tiles_x * tiles_y;
    const int ty = bid / tiles_x, tx = bid - ty * tiles_x;
    const int y0 = ty * TH, x0 = tx * TW;

    // ---- staging assignment, fixed across chunks
    const T* in = static_cast<const T*>(a.in);
    const T* isrc[2];
    bool ivalid[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = tid + 256 * i;
        const int half = e >= NPIX ? 1 : 0;
        const int p = e - half * NPIX;
        const int py = p / PW, px = p - py * PW;
        const int Y = y0 - 1 + py, X = x0 - 1 + px;
        ivalid[i] = (e < IN_F4) && Y >= 0 && Y < a.h && X >= 0 && X < a.w_;
        const int sy = ivalid[i] ? (Y >> a.up) : 0, sx = ivalid[i] ? (X >> a.up) : 0;
        isrc[i] = in + (((size_t)n * a.in_h + sy) * a.in_w + sx) * a.in_stride + half * (KG / 2);
    }
    const f32x4* wsrc = static_cast<const f32x4*>(a.w);

    f32x4 pin[2], pw[WL];
    auto load_chunk = [&](int c) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            pin[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (ivalid[i]) pin[i] = *reinterpret_cast<const f32x4*>(isrc[i] + c * KG);
        }
        const f32x4* ws = wsrc + (size_t)c * W_F4;
#pragma unroll
        for (int i = 0; i < WL; ++i) {
            const int k = tid + 256 * i;
            if (k < W_F4) pw[i] = ws[k];
        }
    };
    auto store_chunk = [&](int stage) {
        f32x4* st = lds + stage * STAGE_F4;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = tid + 256 * i;
            if (e < IN_F4) st[e] = pin[i];
        }
#pragma unroll
        for (int i = 0; i < WL; ++i) {
            const int k = tid + 256 * i;
            if (k < W_F4) st[IN_F4 + k] = pw[i];
        }
    };

    // ---- per-lane compute coordinates
    const int m = lane & 31, hh = lane >> 5;
    const int prow = 2 * wave + (m >> 4), pcol = m & 15;
    const int a_base = hh * NPIX + prow * PW + pcol;  // + dy*PW + dx
    const int b_base = IN_F4 + hh * (32 * NT) + m;    // + tap*2*(32*NT) + nt*32

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int nchunks = a.cin / KG;
    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const bool more = (c + 1) < nchunks;
        if (more) load_chunk(c + 1);
        const f32x4* st = lds + (c & 1) * STAGE_F4;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap - dy * 3;
            const f32x4 av = st[a_base + dy * PW + dx];
            f32x4 bv[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) bv[t] = st[b_base + tap * 2 * (32 * NT) + t * 32];
            if constexpr (BF) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av),
                                                                     __builtin_bit_cast(bf16x8, bv[t]), acc[t], 0, 0, 0);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[t][j], acc[t], 0, 0, 0);
            }
        }
        if (more) store_chunk((c + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue.  C/D map of the 32x32 MFMA: col = lane&31 (-> cout), row = (r&3)+8*(r>>2)+4*(lane>>5) (-> pixel)
    T* out = static_cast<T*>(a.out);
    T* out2 = static_cast<T*>(a.out2);
    const T* res1 = static_cast<const T*>(a.res1);
    const T* res2 = static_cast<const T*>(a.res2);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int co = t * 32 + m;
        const float bias = a.bias[co];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mm = (r & 3) + 8 * (r >> 2) + 4 * hh;
            const int Y = y0 + 2 * wave + (mm >> 4), X = x0 + (mm & 15);
            if (Y >= a.h || X >= a.w_) continue;
            const size_t pix = ((size_t)n * a.h + Y) * a.w_ + X;
            float v = acc[t][r] + bias;
            if (a.lrelu) v = v > 0.f ? v : v * 0.2f;
            if (res1) v = __fadd_rn(__fmul_rn(v, a.s1), Elem<BF>::ld(res1 + pix * a.res1_stride + co));
            if (res2) v = __fadd_rn(__fmul_rn(v, a.s2), Elem<BF>::ld(res2 + pix * a.res2_stride + co));
            if (out) Elem<BF>::st(out + pix * a.out_stride + a.out_coff + co, v);
            if (out2) Elem<BF>::st(out2 + pix * a.out2_stride + co, v);
            if (co < a.cout_real) {
                if (a.out_nchw) a.out_nchw[(((size_t)n * a.cout_real + co) * a.h + Y) * a.w_ + X] = v;
                if (a.out_u8) {
                    float q = fminf(fmaxf(v, 0.f), 1.f) * 255.0f;
                    q = a.u8_round ? rintf(q) : truncf(q);
                    const int ch = a.u8_flip ? (a.cout_real - 1 - co) : co;
                    a.out_u8[pix * a.cout_real + ch] = (uint8_t)q;
                }
            }
        }
    }
}

template <bool BF>
hipError_t launch(const ConvArgs& a, hipStream_t s) {
    const int tiles = ((a.w_ + TW - 1) / TW) * ((a.h + TH - 1) / TH) * a.n;
    if (tiles <= 0) return hipSuccess;
    if (a.cin % Elem<BF>::KG) return hipErrorInvalidValue;
    if (a.coutp == 64) {
        constexpr int NT = 2;
        const size_t shm = 2 * (2 * NPIX + 9 * 2 * 32 * NT) * sizeof(f32x4);
        hipLaunchKernelGGL((conv3x3_mfma_kernel<BF, NT>), dim3(tiles), dim3(256), shm, s, a);
    } else if (a.coutp == 32) {
        constexpr int NT = 1;
        const size_t shm = 2 * (2 * NPIX + 9 * 2 * 32 * NT) * sizeof(f32x4);
        hipLaunchKernelGGL((conv3x3_mfma_kernel<BF, NT>), dim3(tiles), dim3(256), shm, s, a);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

inline uint16_t host_f2bf(float f) {  // round-to-nearest-even, NaN stays NaN
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

}  // namespace

size_t packed_weight_elems_f32(int cin_p, int coutp) { return (size_t)cin_p * 9 * coutp; }
size_t packed_weight_elems_bf16(int cin_p, int coutp) { return (size_t)cin_p * 9 * coutp; }

// OIHW -> [chunk = ci/8][tap][half = (ci%8)/4][n][j = ci%4]; zero padded in both ci and n.
void pack_weights_f32(const float* oihw, int cout, int cin, int cin_p, int coutp, float* dst) {
    const size_t total = packed_weight_elems_f32(cin_p, coutp);
    for (size_t i = 0; i < total; ++i) dst[i] = 0.f;
    for (int o = 0; o < cout; ++o)
        for (int ci = 0; ci < cin; ++ci)
            for (int tap = 0; tap < 9; ++tap) {
                const int c = ci / 8, half = (ci % 8) / 4, j = ci % 4;
                const size_t idx = ((((size_t)c * 9 + tap) * 2 + half) * coutp + o) * 4 + j;
                dst[idx] = oihw[((size_t)o * cin + ci) * 9 + tap];
            }
}

// OIHW -> [chunk = ci/16][tap][half = (ci%16)/8][n][j = ci%8] bf16 (RNE); zero padded.
void pack_weights_bf16(const float* oihw, int cout, int cin, int cin_p, int coutp, uint16_t* dst) {
    const size_t total = packed_weight_elems_bf16(cin_p, coutp);
    for (size_t i = 0; i < total; ++i) dst[i] = 0;
    for (int o = 0; o < cout; ++o)
        for (int ci = 0; ci < cin; ++ci)
            for (int tap = 0; tap < 9; ++tap) {
                const int c = ci / 16, half = (ci % 16) / 8, j = ci % 8;
                const size_t idx = ((((size_t)c * 9 + tap) * 2 + half) * coutp + o) * 8 + j;
                dst[idx] = host_f2bf(oihw[((size_t)o * cin + ci) * 9 + tap]);
            }
}

hipError_t launch_conv3x3_f32(const ConvArgs& a, hipStream_t s) { return launch<false>(a, s); }
hipError_t launch_conv3x3_bf16(const ConvArgs& a, hipStream_t s) { return launch<true>(a, s); }

}  // namespace nesr
