// 3x3 convolution as im2col-free implicit GEMM on the gfx950 matrix cores -- the DIRECT form.
// (The default f32 path is the Winograd kernel, conv3x3_wino_f32.hip; this kernel is the exact
// reference point `f32-direct`, the bf16 path for small frames, and the body of the opt-in
// persistent trunk kernel at the end of this file.)
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
//               global -> registers -> LDS into a 3-slot ring (two chunks ahead), one barrier per
//               chunk, first tap of the next chunk read before the barrier.
//   pixels    : (the MFMA's B operand) lane (m = lane&31, h = lane>>5) reads the 16 bytes
//               [16h, 16h+16) of its pixel's chunk with one ds_read_b128: f32 -> 4 channels, MFMA
//               k-step j pairs channels {j, 4+j}; bf16 -> 8 channels = exactly the 32x32x16
//               fragment (k = 8h + j).
//   weights   : (the MFMA's A operand) pre-packed on the host as [chunk][tap][h][n][16 B] so the
//               matching ds_read_b128 is lane-contiguous (conflict-free).
//   LDS image : activations [h][pixel][16 B] (two planes), so 16 consecutive pixels are 256
//               contiguous bytes.
//   epilogue  : weights are the MFMA A operand and pixels the B operand, so a lane owns runs of 4
//               consecutive channels of one pixel: bias, LeakyReLU(0.2), up to two scaled residuals
//               (RDB: x5*0.2+x; RRDB: out*0.2+x), 16-byte channel-slice stores through the Map
//               addressing (the concat-free dense block), optional planar NCHW or clamped/quantised
//               u8 HWC store for conv_last.
#include <hip/hip_bf16.h>

#include <cstdlib>

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
    __device__ static f32x4 ld4(const T* p) { return *reinterpret_cast<const f32x4*>(p); }
    // plain stores here: the write-through (sc1) form that helps the Winograd and bf16 XL kernels
    // measured +1.3 % (f32) / +4.5 % (bf16, 8-byte stores) slower in this kernel
    __device__ static void st4(T* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
};
template <>
struct Elem<true> {
    typedef uint16_t T;
    static constexpr int KG = 16;
    __device__ static f32x4 ld4(const T* p) {   // 4 bf16 (8 bytes) -> 4 f32
        const uint2 u = *reinterpret_cast<const uint2*>(p);
        return f32x4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                     __uint_as_float(u.y & 0xffff0000u)};
    }
    __device__ static void st4(T* p, f32x4 v) {   // plain casts -> v_cvt_pk_bf16_f32 (RNE, NaN preserving)
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        const bf16x4 b = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
        *reinterpret_cast<uint2*>(p) = __builtin_bit_cast(uint2, b);
    }
};

// One output tile (8 x 16 pixels at (y0, x0) of image n) of one convolution.  Shared by the
// per-layer kernel below and by the persistent trunk kernel (trunk_persist.hip).
template <bool BF, int NT, int WV = 4>
__device__ __forceinline__ void conv_tile(const ConvArgs& a, const int n, const int y0, const int x0, char* smem) {
    typedef typename Elem<BF>::T T;
    constexpr int KG = Elem<BF>::KG;
    // WV waves per workgroup, 2 output rows each: tile (2*WV) x 16 pixels
    constexpr int THREADS = 64 * WV;
    constexpr int NPIX = (2 * WV + 2) * PW;
    constexpr int IN_F4 = 2 * NPIX;
    constexpr int W_F4 = 9 * 2 * 32 * NT;
    constexpr int STAGE_F4 = IN_F4 + W_F4;
    f32x4* lds = reinterpret_cast<f32x4*>(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    // ---- staging assignment, fixed across chunks and branch-free.  One chunk's LDS image is
    // [input: 2 planes x 180 pixels | weights: 9 taps x 2 halves x 32*NT] 16-byte items; item k of
    // round i (k = tid + THREADS i, clamped to the last item: duplicates write identical bytes) has a
    // per-thread source pointer that advances by a per-thread stride per chunk.  Out-of-image
    // pixels (the conv's zero padding) load a valid dummy address and are zeroed by a select.
    constexpr int TOTAL_F4 = STAGE_F4;
    constexpr int R = (TOTAL_F4 + THREADS - 1) / THREADS;
    const char* src[R];
    long long cstride[R];
    int slot[R];
    bool zero[R];
    {
        const char* in = static_cast<const char*>(a.in);
        const char* wsrc = static_cast<const char*>(a.w);
        constexpr int ES = (int)sizeof(T);
#pragma unroll
        for (int i = 0; i < R; ++i) {
            int k = tid + THREADS * i;
            k = k < TOTAL_F4 ? k : TOTAL_F4 - 1;
            slot[i] = k;
            if (k < IN_F4) {
                const int half = k >= NPIX ? 1 : 0;
                const int p = k - half * NPIX;
                const int py = p / PW, px = p - py * PW;
                const int Y = y0 - 1 + py, X = x0 - 1 + px;
                const bool ok = Y >= 0 && Y < a.h && X >= 0 && X < a.w_;
                const int sy = ok ? (Y >> a.up) : 0, sx = ok ? (X >> a.up) : 0;
                src[i] = in + ((((size_t)n * a.in_h + sy) * a.in_w + sx) * a.in_map.pix) * ES + half * 16;
                cstride[i] = -1;   // input item: advances by the map's K-group stride
                zero[i] = !ok;
            } else {
                src[i] = wsrc + (size_t)(k - IN_F4) * 16;
                cstride[i] = W_F4 * 16;
                zero[i] = false;
            }
            if (cstride[i] < 0) cstride[i] = a.in_map.chunk * ES;
        }
    }

    f32x4 pv[R];
    auto load_chunk = [&](int c) {
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(src[i] + (long long)c * cstride[i]);
            pv[i] = zero[i] ? f32x4{0.f, 0.f, 0.f, 0.f} : v;
        }
    };
    auto store_chunk = [&](int stage) {
        f32x4* st = lds + stage * STAGE_F4;
#pragma unroll
        for (int i = 0; i < R; ++i) st[slot[i]] = pv[i];
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

    // bias is fetched now so its latency hides under the K loop
    f32x4 bs[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) bs[t][g] = *reinterpret_cast<const f32x4*>(a.bias + t * 32 + 8 * g + 4 * hh);

    const int nchunks = a.cin / KG;
    // operand registers: a 3-slot rotation, tap t lives in slot t % 3 and is read two taps ahead
    f32x4 opa[3];
    f32x4 opb[3][NT];
    auto read_tap = [&](const f32x4* st, int tap, int sl) {
        const int dy = tap / 3, dx = tap - dy * 3;
        opa[sl] = st[a_base + dy * PW + dx];
#pragma unroll
        for (int t = 0; t < NT; ++t) opb[sl][t] = st[b_base + tap * 2 * (32 * NT) + t * 32];
    };
    auto mfma_tap = [&](int sl) {
        if constexpr (BF) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, opb[sl][t]),
                                                                 __builtin_bit_cast(bf16x8, opa[sl]), acc[t], 0, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(opb[sl][t][j], opa[sl][j], acc[t], 0, 0, 0);
        }
    };

    // 3-slot LDS ring: chunk c+2 is staged while chunk c is computed, so chunk c+1 is already
    // visible (previous iteration's barrier) and the rolling operand prefetch runs straight across
    // the chunk boundary: taps 0,1 of chunk c+1 are read during taps 7,8 of chunk c, before the
    // barrier.  sched_barrier pins "reads for tap+2, then MFMAs of tap": left alone, hipcc sinks the
    // ds_reads to just before their s_waitcnt and the single-accumulator MFMA chain stalls on them.
    load_chunk(0);
    store_chunk(0);
    load_chunk(nchunks > 1 ? 1 : 0);
    store_chunk(1);
    __syncthreads();
    const f32x4* s_cur = lds;
    const f32x4* s_nxt = lds + STAGE_F4;
    int stage_fill = 2;
    read_tap(s_cur, 0, 0);
    read_tap(s_cur, 1, 1);
    for (int c = 0; c < nchunks; ++c) {
        load_chunk(c + 2 < nchunks ? c + 2 : nchunks - 1);   // clamped: a redundant reload is harmless
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (tap + 2 < 9)
                read_tap(s_cur, tap + 2, (tap + 2) % 3);
            else
                read_tap(s_nxt, tap + 2 - 9, (tap + 2) % 3);
            __builtin_amdgcn_sched_barrier(0);
            mfma_tap(tap % 3);
            __builtin_amdgcn_sched_barrier(0);
        }
        store_chunk(stage_fill);
        __syncthreads();
        s_cur = s_nxt;
        s_nxt = lds + stage_fill * STAGE_F4;
        stage_fill = stage_fill == 2 ? 0 : stage_fill + 1;
    }

    // ---- epilogue.  The weights are the MFMA's A operand and the pixels its B operand, so in the
    // 32x32 C/D map col = lane&31 is this lane's pixel and rows (r&3)+8*(r>>2)+4*(lane>>5) are output
    // channels: every lane owns 4 x NT runs of 4 consecutive channels of ONE pixel -> 16-byte
    // residual loads and stores, all loads issued before any use (no per-element branches).
    const int Y = y0 + prow, X = x0 + pcol;
    const bool valid = Y < a.h && X < a.w_;
    const size_t pix = ((size_t)n * a.h + (valid ? Y : 0)) * a.w_ + (valid ? X : 0);
    f32x4 r1[NT][4], r2[NT][4];
    // element offset of the 4-channel run starting at channel c (c % 4 == 0) of this lane's pixel
    auto at = [&](const Map& mp, int c) -> size_t { return (size_t)(c / KG) * mp.chunk + pix * mp.pix + (c % KG); };
    if (a.res1) {
        const T* rp = static_cast<const T*>(a.res1);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) r1[t][g] = Elem<BF>::ld4(rp + at(a.res1_map, t * 32 + 8 * g + 4 * hh));
    }
    if (a.res2) {
        const T* rp = static_cast<const T*>(a.res2);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) r2[t][g] = Elem<BF>::ld4(rp + at(a.res2_map, t * 32 + 8 * g + 4 * hh));
    }
    f32x4 v[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float x = acc[t][4 * g + q] + bs[t][g][q];
                if (a.lrelu) x = x > 0.f ? x : x * 0.2f;
                v[t][g][q] = x;
            }
        }
    if (a.res1) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int q = 0; q < 4; ++q) v[t][g][q] = __fadd_rn(__fmul_rn(v[t][g][q], a.s1), r1[t][g][q]);
    }
    if (a.res2) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int q = 0; q < 4; ++q) v[t][g][q] = __fadd_rn(__fmul_rn(v[t][g][q], a.s2), r2[t][g][q]);
    }
    if (valid) {
        if (a.out) {
            T* op = static_cast<T*>(a.out);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) Elem<BF>::st4(op + at(a.out_map, a.out_coff + t * 32 + 8 * g + 4 * hh), v[t][g]);
        }
        if (a.out2) {
            T* op = static_cast<T*>(a.out2);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) Elem<BF>::st4(op + at(a.out2_map, t * 32 + 8 * g + 4 * hh), v[t][g]);
        }
        if (a.cout_real > 0 && hh == 0) {
            // conv_last: channels 0..cout_real-1 (<= 4) live in v[0][0] of the lower half-wave
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

template <bool BF, int NT, int WV>
__global__ __launch_bounds__(64 * WV) void conv3x3_mfma_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TH_ = 2 * WV;
    const int tiles_x = (a.w_ + TW - 1) / TW;
    const int tiles_y = (a.h + TH_ - 1) / TH_;
    int bid = blockIdx.x;
    const int n = bid / (tiles_x * tiles_y);
    bid -= n * tiles_x * tiles_y;
    const int ty = bid / tiles_x, tx = bid - ty * tiles_x;
    conv_tile<BF, NT, WV>(a, n, ty * TH_, tx * TW, smem);
}

template <bool BF, int NT, int WV>
hipError_t launch_nt(const ConvArgs& a, hipStream_t s) {
    constexpr int TH_ = 2 * WV;
    constexpr size_t shm = 3 * (2 * (TH_ + 2) * PW + 9 * 2 * 32 * NT) * sizeof(f32x4);
    static unsigned long long attr_done = 0;
    {
        const hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&conv3x3_mfma_kernel<BF, NT, WV>), shm, attr_done);
        if (e != hipSuccess) return e;
    }
    const int tiles = ((a.w_ + TW - 1) / TW) * ((a.h + TH_ - 1) / TH_) * a.n;
    if (tiles <= 0) return hipSuccess;
    hipLaunchKernelGGL((conv3x3_mfma_kernel<BF, NT, WV>), dim3(tiles), dim3(64 * WV), shm, s, a);
    return hipGetLastError();
}

template <bool BF>
hipError_t launch(const ConvArgs& a, hipStream_t s) {
    if (a.cin % Elem<BF>::KG) return hipErrorInvalidValue;
    // NESR_NT1_WAVES=2: Cout=32 layers in 4x16-pixel tiles, 2 waves per workgroup (4 independent
    // workgroups per CU on a 256x256 frame instead of 2) -- measured 4.5 % slower in-process, kept for A/B
    static const int nt1_waves = [] { const char* e = getenv("NESR_NT1_WAVES"); return e ? atoi(e) : 4; }();
    if (a.coutp == 64) return launch_nt<BF, 2, 4>(a, s);
    if (a.coutp == 32) return nt1_waves == 2 ? launch_nt<BF, 1, 2>(a, s) : launch_nt<BF, 1, 4>(a, s);
    return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------------------------
// Persistent trunk kernel.  Layer L of tile t may start once its 8 neighbouring tiles have
// completed layers < L (their outputs are this tile's halo).  That single condition also covers
// every write-after-read on the rotating buffers: neighbouring tiles are never more than one
// layer apart, and no layer writes a channel slice the previous layer reads.
// Hand-off (cdna_hip_programming.md Guideline 16): producer = plain stores, every wave
// s_waitcnt vmcnt(0), workgroup barrier, one lane agent-scope release + vmcnt(0) + relaxed agent
// store of the counter; consumer = one wave polls relaxed, ONE agent-scope acquire, vmcnt(0),
// workgroup barrier, then plain loads.  Placement independent; every spin is bounded.
constexpr size_t TRUNK_RING_BYTES = 3 * (2 * NPIX + 9 * 2 * 32 * 2) * sizeof(f32x4);
constexpr size_t TRUNK_SHM = TRUNK_RING_BYTES + 16;

template <bool BF>
__global__ __launch_bounds__(256, 2) void trunk_persist_kernel(TrunkArgs t) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    volatile unsigned* s_abort = reinterpret_cast<volatile unsigned*>(smem + TRUNK_RING_BYTES);
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = (t.w + TW - 1) / TW;
    const int tiles_y = (t.h + TH - 1) / TH;
    const int per_img = tiles_x * tiles_y;
    const int ntiles = per_img * t.n;
    if (tid == 0) *s_abort = 0;
    __syncthreads();

    for (int L = 0; L < t.nlayers; ++L) {
        const TrunkLayer ly = t.layers[L];
        ConvArgs a;
        a.in = t.buf[ly.in_buf];
        a.in_map = t.map;
        a.in_h = t.h;
        a.in_w = t.w;
        a.up = 0;
        a.cin = ly.cin;
        a.w = ly.w;
        a.bias = ly.bias;
        a.coutp = ly.coutp;
        a.n = t.n;
        a.h = t.h;
        a.w_ = t.w;
        a.out = t.buf[ly.out_buf];
        a.out_map = t.map;
        a.out_coff = ly.out_coff;
        a.out2 = nullptr;
        a.out2_map = t.map;
        a.lrelu = ly.lrelu;
        a.res1 = ly.res1_buf >= 0 ? t.buf[ly.res1_buf] : nullptr;
        a.res1_map = t.map;
        a.s1 = ly.s1;
        a.res2 = ly.res2_buf >= 0 ? t.buf[ly.res2_buf] : nullptr;
        a.res2_map = t.map;
        a.s2 = ly.s2;
        a.out_nchw = nullptr;
        a.cout_real = 0;
        a.out_u8 = nullptr;
        a.u8_flip = a.u8_round = 0;
        a.zeros = t.zeros;

        for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
            const int n = tile / per_img;
            const int rem = tile - n * per_img;
            const int ty = rem / tiles_x, tx = rem - ty * tiles_x;

            // ---- wait for the 8 neighbours to have finished layer L-1 (bounded)
            if (L > 0) {
                if (wave == 0) {
                    const int lane = tid & 63;
                    const int dy = lane / 3 - 1, dx = lane - (lane / 3) * 3 - 1;
                    const int ny = ty + dy, nx = tx + dx;
                    const bool watch = lane < 9 && lane != 4 && ny >= 0 && ny < tiles_y && nx >= 0 && nx < tiles_x;
                    const unsigned* ctr = t.progress + (watch ? n * per_img + ny * tiles_x + nx : tile);
                    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                    bool aborted = false;
                    for (;;) {
                        const unsigned v = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (__all(!watch || v >= (unsigned)L)) break;
                        if (__hip_atomic_load(t.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u ||
                            __builtin_amdgcn_s_memrealtime() - t0 > 300000000ull) {   // 3 s at 100 MHz
                            aborted = true;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    if (aborted) {
                        if (lane == 0) {
                            __hip_atomic_store(t.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            *s_abort = 1u;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __syncthreads();
                if (*s_abort) return;   // uniform: every thread reads the same LDS word after the barrier
            }

            if (ly.coutp == 64)
                conv_tile<BF, 2>(a, n, ty * TH, tx * TW, smem);
            else
                conv_tile<BF, 1>(a, n, ty * TH, tx * TW, smem);

            // ---- publish: this tile has completed layer L
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(t.progress + tile, (unsigned)(L + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

template <bool BF>
hipError_t launch_trunk(const TrunkArgs& t, hipStream_t s) {
    static int max_blocks = -1;   // co-resident workgroups (all devices of a node are the same part)
    static unsigned long long attr_done = 0;
    {
        const hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&trunk_persist_kernel<BF>), TRUNK_SHM, attr_done);
        if (e != hipSuccess) return e;
    }
    if (max_blocks < 0) {
        hipError_t e;
        int per_cu = 0, dev = 0;
        hipDeviceProp_t prop;
        if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
        if ((e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) return e;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, trunk_persist_kernel<BF>, 256, TRUNK_SHM);
        if (e != hipSuccess) return e;
        if (per_cu > 2) per_cu = 2;   // LDS admits 2; never trust the API for more (MI355X_MICROARCH.md, residency)
        if (per_cu < 1) return hipErrorLaunchOutOfResources;
        max_blocks = per_cu * prop.multiProcessorCount;
    }
    const int ntiles = ((t.w + TW - 1) / TW) * ((t.h + TH - 1) / TH) * t.n;
    if (ntiles <= 0 || t.nlayers <= 0) return hipSuccess;
    const int grid = ntiles < max_blocks ? ntiles : max_blocks;
    TrunkArgs targ = t;
    void* params[] = {&targ};
    // cooperative launch: the runtime verifies that the whole grid is co-resident
    return hipLaunchCooperativeKernel(reinterpret_cast<const void*>(&trunk_persist_kernel<BF>), dim3(grid), dim3(256), params,
                                      TRUNK_SHM, s);
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

hipError_t launch_conv3x3_f32(const ConvArgs& a, hipStream_t s) {
    if (a.y_lo || a.y_hi) return hipErrorInvalidValue;   // row ranges: conv3x3_f16x2_kernel only
    return launch<false>(a, s);
}
hipError_t launch_trunk_persist(const TrunkArgs& t, bool bf16, hipStream_t s) {
    return bf16 ? launch_trunk<true>(t, s) : launch_trunk<false>(t, s);
}
// bf16: frames of more than 256x256 trunk pixels take the large-tile LDS-DMA kernel (a single
// 256x256 frame is only 64 of its tiles, too few for 256 CUs); the choice
// depends on the frame size only (never on the batch), so a tile's arithmetic is the same on every
// rank and in every batch.  NESR_BF16_KERNEL=small|big|xl overrides (tests, A/B timing).
hipError_t launch_conv3x3_bf16(const ConvArgs& a, hipStream_t s) {
    if (a.y_lo || a.y_hi) return hipErrorInvalidValue;   // row ranges: conv3x3_f16x2_kernel only
    static const int mode = [] {
        const char* e = getenv("NESR_BF16_KERNEL");
        if (!e) return 0;
        return e[0] == 's' ? 1 : (e[0] == 'b' ? 2 : (e[0] == 'x' ? 3 : 0));
    }();
    const bool large = (long)a.h * a.w_ > 256L * 256L;
    if (a.zeros) {
        if (mode == 3 || (mode == 0 && large) || a.rag_n || a.size_independent) return launch_conv3x3_bf16_xl(a, s);
    }
    if (a.rag_n) return hipErrorInvalidValue;   // only the large-tile kernel knows about ragged batches
    return launch<true>(a, s);
}

}  // namespace nesr
