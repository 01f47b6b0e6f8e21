// bf16 3x3 convolution, large-tile variant: the throughput kernel for 4K-class frames
// (BASELINE.json configs[2], [3]).  v_mfma_f32_32x32x16_bf16 runs 16x the f32 MFMA rate, so the
// f32 kernel's tiling (128 pixels per workgroup) is staging-bound in bf16; this kernel raises the
// MFMA work per staged byte 4x and reuses each pixel fragment across the three vertical taps.
//
//   workgroup : 256 threads = 4 waves, output tile 16 rows x 32 cols (512 px) x all Cout (32|64)
//   wave      : 4 output rows x 32 cols x NT 32-wide channel tiles -> 4*NT accumulators (f32x16)
//   K loop    : 16-channel (32-byte) chunks.  The (16+2)x(32+2) halo tile and the 9x16xCout weight
//               slab are copied global -> LDS by LDS-DMA (global_load_lds_dwordx4: per-lane source,
//               lane-linear destination, no staging registers), 2 stages, one barrier per chunk.
//               Out-of-image pixels (the conv's zero padding) read a device zero page.
//   operands  : for one horizontal tap dx the wave reads 6 pixel fragments (rows -1..4) once and
//               uses each for up to 3 vertical taps: 6 + 3*NT ds_read_b128 per 12*NT MFMAs.
//   schedule  : step (dx,dy) = 4*NT MFMAs; the weight fragments of the next step and the pixel
//               fragments of the next dx are read during the current step (sched_barrier-pinned).
//   placement : blockIdx -> tile through the bijective XCD remap, so each XCD's L2 serves a
//               contiguous run of tiles (shared halo columns/rows and the weights hit L2).
//   epilogue  : as the generic kernel (conv3x3_mfma.hip): weights are the MFMA A operand, pixels
//               the B operand; every lane owns runs of 4 consecutive channels of one pixel per row.
#include <hip/hip_bf16.h>

#include "nesr_kernels.h"

namespace nesr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int TH = 16, TW = 32;
constexpr int PH = TH + 2, PW = TW + 2;
constexpr int NPIX = PH * PW;          // 612
constexpr int IN_ITEMS = 1280;         // 2 planes x 612 = 1224 16-byte items, padded to 5 rounds of 256
constexpr int IN_ROUNDS = IN_ITEMS / 256;

template <int NT>
struct Geo {
    static constexpr int W_ITEMS = 9 * 2 * 32 * NT;
    static constexpr int ROUNDS = (IN_ITEMS + W_ITEMS + 255) / 256;
    static constexpr int STAGE_ITEMS = ROUNDS * 256;
    static constexpr int W_ROUNDS = ROUNDS - IN_ROUNDS;
};

typedef __attribute__((address_space(3))) char lds_char;
typedef const __attribute__((address_space(1))) void* gptr_t;

__device__ inline f32x4 ld4_bf16(const uint16_t* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return f32x4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                 __uint_as_float(u.y & 0xffff0000u)};
}
__device__ inline void st4_bf16(uint16_t* p, f32x4 v) {
    const bf16x4 b = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    *reinterpret_cast<uint2*>(p) = __builtin_bit_cast(uint2, b);
}

template <int NT>
__global__ __launch_bounds__(256, 2) void conv3x3_bf16_big_kernel(ConvArgs a) {
    typedef Geo<NT> G;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const f32x4* lds = reinterpret_cast<const f32x4*>(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- XCD-aware tile id (bijective for any tile count)
    const int tiles_x = (a.w_ + TW - 1) / TW;
    const int tiles_y = (a.h + TH - 1) / TH;
    const int ntiles = tiles_x * tiles_y * a.n;
    int tile;
    {
        const int bid = blockIdx.x, q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int n = tile / (tiles_x * tiles_y);
    tile -= n * tiles_x * tiles_y;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int y0 = ty * TH, x0 = tx * TW;

    // ---- LDS-DMA sources.  Item k of the input region = plane (k / 612), padded-tile pixel (k % 612).
    const char* zero_page = static_cast<const char*>(a.zeros);
    const char* isrc[IN_ROUNDS];
    unsigned live = 0;   // bit i: round i's source advances 32 bytes per chunk (real pixel)
    {
        const char* in = static_cast<const char*>(a.in);
#pragma unroll
        for (int i = 0; i < IN_ROUNDS; ++i) {
            const int k = tid + 256 * i;
            const int half = k >= NPIX ? 1 : 0;
            const int p = k - half * NPIX;
            const int py = p / PW, px = p - py * PW;
            const int Y = y0 - 1 + py, X = x0 - 1 + px;
            const bool ok = (k < 2 * NPIX) && Y >= 0 && Y < a.h && X >= 0 && X < a.w_;
            const int sy = ok ? (Y >> a.up) : 0, sx = ok ? (X >> a.up) : 0;
            isrc[i] = ok ? in + ((((size_t)n * a.in_h + sy) * a.in_w + sx) * a.in_stride) * 2 + half * 16 : zero_page;
            live |= ok ? (1u << i) : 0u;
        }
    }
    const char* wbase = static_cast<const char*>(a.w);

    auto stage = [&](int c, int buf) {
        lds_char* dst = (lds_char*)(smem) + (size_t)buf * (G::STAGE_ITEMS * 16) + wave * 1024;
#pragma unroll
        for (int i = 0; i < IN_ROUNDS; ++i) {
            const char* s = isrc[i] + (((live >> i) & 1u) ? c * 32 : 0);
            __builtin_amdgcn_global_load_lds((gptr_t)s, dst + i * 4096, 16, 0, 0);
        }
        const char* wc = wbase + (size_t)c * (G::W_ITEMS * 16);
#pragma unroll
        for (int i = 0; i < G::W_ROUNDS; ++i) {
            const int k = tid + 256 * i;
            const char* s = k < G::W_ITEMS ? wc + k * 16 : zero_page;
            __builtin_amdgcn_global_load_lds((gptr_t)s, dst + (IN_ROUNDS + i) * 4096, 16, 0, 0);
        }
    };

    // ---- per-lane operand coordinates
    const int m = lane & 31, hh = lane >> 5;
    const int p_base = hh * NPIX + (4 * wave) * PW + m;           // + r*PW + dx
    const int w_base = IN_ITEMS + hh * (32 * NT) + m;             // + tap*2*(32*NT) + t*32

    f32x16 acc[4][NT];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][t][e] = 0.f;

    const int nchunks = a.cin / 16;
    stage(0, 0);
    __syncthreads();   // vmcnt(0) + barrier: chunk 0 landed
    for (int c = 0; c < nchunks; ++c) {
        if (c + 1 < nchunks) stage(c + 1, (c + 1) & 1);
        const f32x4* st = lds + (c & 1) * G::STAGE_ITEMS;
        f32x4 P[2][6];
        f32x4 Wf[2][NT];
#pragma unroll
        for (int r = 0; r < 6; ++r) P[0][r] = st[p_base + r * PW];
#pragma unroll
        for (int t = 0; t < NT; ++t) Wf[0][t] = st[w_base + t * 32];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            const int dx = s / 3, dy = s - dx * 3;
            if (s + 1 < 9) {
                const int dx1 = (s + 1) / 3, dy1 = (s + 1) - dx1 * 3;
                const int tap1 = dy1 * 3 + dx1;
#pragma unroll
                for (int t = 0; t < NT; ++t) Wf[(s + 1) & 1][t] = st[w_base + tap1 * 2 * (32 * NT) + t * 32];
            }
            if (dy == 0 && dx < 2) {
#pragma unroll
                for (int r = 0; r < 6; ++r) P[(dx + 1) & 1][r] = st[p_base + r * PW + dx + 1];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Wf[s & 1][t]),
                                                                        __builtin_bit_cast(bf16x8, P[dx & 1][r + dy]),
                                                                        acc[r][t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();   // next chunk landed (vmcnt(0)) and every wave is done with this stage
    }

    // ---- epilogue: lane = pixel column m of rows 4*wave + r; regs = 4-channel runs 8g + 4hh (+32t)
    const int X = x0 + m;
    const bool xok = X < a.w_;
    const int Xc = xok ? X : 0;
    const uint16_t* res1 = static_cast<const uint16_t*>(a.res1);
    const uint16_t* res2 = static_cast<const uint16_t*>(a.res2);
    uint16_t* out = static_cast<uint16_t*>(a.out);
    uint16_t* out2 = static_cast<uint16_t*>(a.out2);
    f32x4 bs[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) bs[t][g] = *reinterpret_cast<const f32x4*>(a.bias + t * 32 + 8 * g + 4 * hh);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int Y = y0 + 4 * wave + r;
        const bool valid = xok && Y < a.h;
        const size_t pix = ((size_t)n * a.h + (Y < a.h ? Y : 0)) * a.w_ + Xc;
        f32x4 r1[NT][4], r2[NT][4];
        if (res1) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) r1[t][g] = ld4_bf16(res1 + pix * a.res1_stride + t * 32 + 8 * g + 4 * hh);
        }
        if (res2) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) r2[t][g] = ld4_bf16(res2 + pix * a.res2_stride + t * 32 + 8 * g + 4 * hh);
        }
        f32x4 v[NT][4];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float x = acc[r][t][4 * g + q] + bs[t][g][q];
                    if (a.lrelu) x = x > 0.f ? x : x * 0.2f;
                    if (res1) x = __fadd_rn(__fmul_rn(x, a.s1), r1[t][g][q]);
                    if (res2) x = __fadd_rn(__fmul_rn(x, a.s2), r2[t][g][q]);
                    v[t][g][q] = x;
                }
        if (valid) {
            if (out) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int g = 0; g < 4; ++g) st4_bf16(out + pix * a.out_stride + a.out_coff + t * 32 + 8 * g + 4 * hh, v[t][g]);
            }
            if (out2) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int g = 0; g < 4; ++g) st4_bf16(out2 + pix * a.out2_stride + t * 32 + 8 * g + 4 * hh, v[t][g]);
            }
            if (a.cout_real > 0 && hh == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (q >= a.cout_real) break;
                    const float x = v[0][0][q];
                    if (a.out_nchw) a.out_nchw[(((size_t)n * a.cout_real + q) * a.h + Y) * a.w_ + X] = x;
                    if (a.out_u8) {
                        float qv = fminf(fmaxf(x, 0.f), 1.f) * 255.0f;
                        qv = a.u8_round ? rintf(qv) : truncf(qv);
                        const int ch = a.u8_flip ? (a.cout_real - 1 - q) : q;
                        a.out_u8[pix * a.cout_real + ch] = (uint8_t)qv;
                    }
                }
            }
        }
    }
}

template <int NT>
hipError_t launch_big(const ConvArgs& a, hipStream_t s) {
    typedef Geo<NT> G;
    constexpr size_t shm = 2 * (size_t)G::STAGE_ITEMS * 16;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_bf16_big_kernel<NT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int tiles = ((a.w_ + TW - 1) / TW) * ((a.h + TH - 1) / TH) * a.n;
    hipLaunchKernelGGL((conv3x3_bf16_big_kernel<NT>), dim3(tiles), dim3(256), shm, s, a);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_conv3x3_bf16_big(const ConvArgs& a, hipStream_t s) {
    if (a.cin % 16 || !a.zeros) return hipErrorInvalidValue;
    if (a.coutp == 64) return launch_big<2>(a, s);
    if (a.coutp == 32) return launch_big<1>(a, s);
    return hipErrorInvalidValue;
}

}  // namespace nesr
