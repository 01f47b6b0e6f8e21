// bf16 3x3 convolution, large-tile ("XL") variant: the throughput kernel for 4K-class frames
// (BASELINE.json configs[2], [3]).  v_mfma_f32_32x32x16_bf16 runs 16x the f32 MFMA rate, so the
// generic kernel's tiling (128 pixels per workgroup, register staging) is far from feeding it.
//
//   layout    : activations are channel-blocked, [C/16][pixels][16 bf16] (nesr_kernels.h, Map): one
//               K-chunk (16 channels = 32 bytes) of neighbouring pixels is contiguous in HBM, so one
//               LDS-DMA wave-instruction (64 lanes x 16 B) copies 32 pixels = 8 whole 128-byte
//               lines.  (With NHWC every lane touched its own line and the kernels were bound by the
//               L1's access rate at ~20 % of the MFMA peak: profiles/r01/bf16_tcp_bound.txt.)
//   workgroup : 512 threads = 8 waves, output tile 32 x 32 pixels x all Cout (32|64), 1 per CU
//   wave      : 4 output rows x 32 cols x NT 32-wide channel tiles -> 4*NT f32x16 accumulators
//   K loop    : 16-channel chunks; the (32+2)x(32+2) halo tile and the 9x16xCout weight slab are
//               copied global -> LDS by LDS-DMA (global_load_lds_dwordx4 from inline asm: per-lane
//               source, lane-linear destination, no staging registers).  3-slot input ring (two
//               chunks ahead: HBM / Infinity-Cache latency) + 2-slot weight ring (one chunk ahead:
//               L2 latency), a COUNTED s_waitcnt vmcnt and a raw s_barrier: one barrier per chunk,
//               the DMA never drains inside the loop.  Out-of-image pixels read a zero page.
//                 iteration c:  s_waitcnt vmcnt(nin)  -> this wave's w(c), in(c) have landed
//                               s_barrier             -> everyone's have; all finished compute(c-1)
//                               issue w(c+1), in(c+2) into the slots last read in iteration c-1
//                               compute(c)
//   LDS image : input slot = [padded pixel][32 B]; the two 16-byte halves of a pixel are swapped
//               when bit 3 of its padded column is set (applied on the DMA source address and on the
//               read), which makes the 16-lane ds_read_b128 groups conflict-free; the key depends on
//               the column only, so a lane needs one address per horizontal tap and the six rows
//               are immediate offsets.
//   operands  : for one horizontal tap dx the wave reads 6 pixel fragments (rows -1..4) once and
//               uses each for up to 3 vertical taps: 6 + 3*NT ds_read_b128 per 12*NT MFMAs; the next
//               step's fragments are read between the current step's MFMAs (sched_group_barrier).
//   placement : blockIdx -> tile through the bijective XCD remap (neighbouring tiles share an L2).
//   epilogue  : weights are the MFMA A operand, pixels the B operand -> a lane owns runs of 4
//               consecutive channels of one pixel; pairs of runs are exchanged between the two
//               half-waves (v_permlane32_swap) so every lane stores 16 contiguous bytes and a wave
//               instruction writes 1 KiB of whole lines.
#include <hip/hip_bf16.h>

#include <cstdlib>

#include "nesr_kernels.h"

namespace nesr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int TW = 32, PW = 34;

// Geometry of one instantiation: WAVES waves (4 output rows each), an input ring of ISLOTS slots.
//   <8 waves, 3 slots>: 32x32-pixel tiles, one workgroup per CU  (least staged bytes per FLOP)
//   <4 waves, 3|2 slots>: 16x32-pixel tiles, two independent workgroups per CU (each one's DMA
//                         prologue and store epilogue overlap the other's MFMAs)
template <int WAVES>
struct Geo {
    static constexpr int THREADS = 64 * WAVES;
    static constexpr int TH = 4 * WAVES;
    static constexpr int PH = TH + 2;
    static constexpr int NPIX = PH * PW;                                // padded pixels
    static constexpr int IN_ITEMS = 2 * NPIX;                           // 16-byte items per input slot
    static constexpr int IN_ROUNDS = (IN_ITEMS + THREADS - 1) / THREADS;
    static constexpr int IN_BYTES = IN_ITEMS * 16;
};

typedef __attribute__((address_space(3))) char lds_char;

__device__ inline f32x4 ld4_bf16(const uint16_t* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return f32x4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                 __uint_as_float(u.y & 0xffff0000u)};
}
// 16-byte feature-map store, write-through (sc1): see store_fm in conv3x3_wino_f32.hip
// (in-process A/B on 6 tiles of 266x266: -1.6 % forward time).
__device__ __forceinline__ void store16(uint16_t* p, uint4 v) {
#ifndef NESR_PLAIN_STORES
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(__builtin_bit_cast(f32x4, v)) : "memory");
#else
    *reinterpret_cast<uint4*>(p) = v;
#endif
}

__device__ inline uint2 pack4_bf16(f32x4 v) {   // plain casts -> v_cvt_pk_bf16_f32 (RNE, NaN preserving)
    const bf16x4 b = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    return __builtin_bit_cast(uint2, b);
}

// LDS-DMA from inline asm: hipcc does not count it, so it inserts no s_waitcnt vmcnt(0) before the
// ds_reads of OTHER ring slots (with the builtin it cannot prove the slots do not alias and drains
// the pipeline every chunk).  All waits for these DMAs are the hand-counted ones in the K loop.
// M0 (LDS destination base) is compiler-reserved: saved and restored inside the statement
// (cdna_hip_programming.md section 5.7).
__device__ __forceinline__ void glds16_asm(const char* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}

template <int NT, int WAVES, int ISLOTS>
__global__ __launch_bounds__(64 * WAVES, 2) void conv3x3_bf16_xl_kernel(ConvArgs a) {
    typedef Geo<WAVES> G;
    constexpr int THREADS = G::THREADS, TH = G::TH;
    constexpr int IN_ITEMS = G::IN_ITEMS, IN_ROUNDS = G::IN_ROUNDS, IN_BYTES = G::IN_BYTES;
    constexpr int W_ITEMS = 9 * 2 * 32 * NT;
    constexpr int W_ROUNDS = (W_ITEMS + THREADS - 1) / THREADS;
    constexpr int W_BYTES = W_ITEMS * 16;
    constexpr int WRING = ISLOTS * IN_BYTES;   // LDS: [input ring: ISLOTS x IN_BYTES][weight ring: 2 x W_BYTES]
    constexpr int AHEAD = ISLOTS - 1;          // input chunks in flight ahead of the one being computed
    extern __shared__ __attribute__((aligned(16))) char smem[];

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
    // this image's own size inside its h x w slot (ragged batches: a frame's tiles of different sizes in one launch)
    int vh = a.h, vw = a.w_;
    if (a.rag_n) {
        vh = (int)a.rag_h[n] << a.rag_shift;
        vw = (int)a.rag_w[n] << a.rag_shift;
        if (y0 >= vh || x0 >= vw) return;          // the whole workgroup: nothing of this tile belongs to the image
    }

    // ---- LDS-DMA sources.  LDS item k of an input slot = padded pixel k>>1, 16-byte slot k&1; it
    // holds channel half (k&1) ^ bit3(padded column) of that pixel's chunk.
    const char* zero_page = static_cast<const char*>(a.zeros);
    const char* isrc[IN_ROUNDS];
    unsigned live = 0, present = 0;
    {
        const char* in = static_cast<const char*>(a.in);
#pragma unroll
        for (int i = 0; i < IN_ROUNDS; ++i) {
            const int k = tid + THREADS * i;
            const int p = k >> 1;
            const int py = p / PW, px = p - py * PW;
            const int half = (k & 1) ^ ((px >> 3) & 1);
            const int Y = y0 - 1 + py, X = x0 - 1 + px;
            const bool has = k < IN_ITEMS;
            const bool ok = has && Y >= 0 && Y < vh && X >= 0 && X < vw;
            const int sy = ok ? (Y >> a.up) : 0, sx = ok ? (X >> a.up) : 0;
            isrc[i] = ok ? in + ((((size_t)n * a.in_h + sy) * a.in_w + sx) * a.in_map.pix) * 2 + half * 16 : zero_page;
            live |= ok ? (1u << i) : 0u;
            present |= has ? (1u << i) : 0u;
        }
    }
    const long long in_cstride = a.in_map.chunk * 2;   // bytes between K-chunks
    const char* wbase = static_cast<const char*>(a.w);

    // LDS-DMA is EXEC-masked: lanes without an item do not write, so the rings need no padding.
    // vmcnt counts WAVE instructions, so the number of DMA instructions a wave issues per call
    // must be known exactly: rounds in which the wave has no item at all are skipped by a
    // wave-uniform (scalar) branch, partial waves run the instruction with partial EXEC.
    const unsigned lds_base = (unsigned)(size_t)(lds_char*)(smem);
    const int nin = (IN_ITEMS - wave * 64 + THREADS - 1) / THREADS;   // input DMA instructions of this wave per chunk
    const int nwt = (W_ITEMS - wave * 64 + THREADS - 1) / THREADS;
    auto issue_in_round = [&](int c, int slot, int i) {   // i is a compile-time constant at every call site
        if (i < nin) {
            const unsigned dst = lds_base + slot * IN_BYTES + wave * 1024;
            const char* s = isrc[i] + (((live >> i) & 1u) ? (long long)c * in_cstride : 0ll);
            if ((present >> i) & 1u) glds16_asm(s, __builtin_amdgcn_readfirstlane(dst + i * (THREADS * 16)));
        }
    };
    auto issue_w_round = [&](int c, int slot, int i) {
        if (i < nwt) {
            const unsigned dst = lds_base + WRING + slot * W_BYTES + wave * 1024;
            const int k = tid + THREADS * i;
            if (k < W_ITEMS) glds16_asm(wbase + (size_t)c * W_BYTES + k * 16, __builtin_amdgcn_readfirstlane(dst + i * (THREADS * 16)));
        }
    };
    auto issue_in = [&](int c, int slot) {
#pragma unroll
        for (int i = 0; i < IN_ROUNDS; ++i) issue_in_round(c, slot, i);
    };
    auto issue_w = [&](int c, int slot) {
#pragma unroll
        for (int i = 0; i < W_ROUNDS; ++i) issue_w_round(c, slot, i);
    };
    // ---- per-lane operand coordinates
    const int m = lane & 31, hh = lane >> 5;
    // byte offset inside an input slot of this lane's fragment for horizontal tap dx, row 0 of the wave's 6
    int p_off[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) p_off[dx] = ((4 * wave) * PW + m + dx) * 32 + ((hh ^ (((m + dx) >> 3) & 1)) << 4);
    const int w_off = hh * (32 * NT) + m;             // item index inside a weight slot
    // waves whose 4 rows are all below the image do no arithmetic (edge tiles), but still stage and barrier
    const bool active = (y0 + 4 * wave) < vh;

    f32x16 acc[4][NT];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][t][e] = 0.f;

    const int nchunks = a.cin / 16;
    // prologue: in(0), w(0) [, in(1)] -- the loop's first wait leaves exactly the AHEAD-1 newest input
    // chunks in flight
    issue_in(0, 0);
    issue_w(0, 0);
    if (AHEAD >= 2 && nchunks > 1) issue_in(1, 1);
    int islot = 0, ifill = AHEAD >= 2 ? 2 : 1;
    for (int c = 0; c < nchunks; ++c) {
        // everything of this wave except the input chunks still ahead (AHEAD-1 of them, nin
        // instructions each) must have landed: w(c) and in(c)
        if (AHEAD < 2 || c + 1 >= nchunks)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (nin == IN_ROUNDS)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IN_ROUNDS) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IN_ROUNDS - 1) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (c + 1 < nchunks) issue_w(c + 1, (c + 1) & 1);
        if (c + AHEAD < nchunks) issue_in(c + AHEAD, ifill);
        if (active) {
            const char* st = smem + islot * IN_BYTES;
            const f32x4* sw = reinterpret_cast<const f32x4*>(smem + WRING + (c & 1) * W_BYTES);
            auto pix_frag = [&](int r, int dx) -> f32x4 {
                return *reinterpret_cast<const f32x4*>(st + p_off[dx] + r * (PW * 32));
            };
            f32x4 P[2][6];
            f32x4 Wf[2][NT];
#pragma unroll
            for (int r = 0; r < 6; ++r) P[0][r] = pix_frag(r, 0);
#pragma unroll
            for (int t = 0; t < NT; ++t) Wf[0][t] = sw[w_off + t * 32];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 9; ++s) {
                const int dx = s / 3, dy = s - dx * 3;
                if (s + 1 < 9) {
                    const int dx1 = (s + 1) / 3, dy1 = (s + 1) - dx1 * 3;
                    const int tap1 = dy1 * 3 + dx1;
#pragma unroll
                    for (int t = 0; t < NT; ++t) Wf[(s + 1) & 1][t] = sw[w_off + tap1 * 2 * (32 * NT) + t * 32];
                }
                if (dy == 0 && dx < 2) {
#pragma unroll
                    for (int r = 0; r < 6; ++r) P[(dx + 1) & 1][r] = pix_frag(r, dx + 1);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Wf[s & 1][t]),
                                                                            __builtin_bit_cast(bf16x8, P[dx & 1][r + dy]),
                                                                            acc[r][t], 0, 0, 0);
                // the next step's fragment reads ride between this step's MFMAs instead of in front of them
                // (in-process A/B: -1.6 % / -1.9 % forward time on 24 / 6 tiles of 532x532)
                {
                    constexpr int NRD = NT;           // weight fragments of the next tap
#pragma unroll
                    for (int i = 0; i < 4 * NT; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (dy == 0 && dx < 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                        else if (i < NRD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        islot = islot == ISLOTS - 1 ? 0 : islot + 1;
        ifill = ifill == ISLOTS - 1 ? 0 : ifill + 1;
    }
    if (!active) return;

    // ---- epilogue: lane = pixel column m of rows 4*wave + r; regs = 4-channel runs 8g + 4hh (+32t)
    const int X = x0 + m;
    const bool xok = X < vw;
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
        const bool valid = xok && Y < vh;
        const size_t pix = ((size_t)n * a.h + (Y < a.h ? Y : 0)) * a.w_ + Xc;
        auto at = [&](const Map& mp, int c) -> size_t { return (size_t)(c >> 4) * mp.chunk + pix * mp.pix + (c & 15); };
        f32x4 r1[NT][4], r2[NT][4];
        if (res1) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) r1[t][g] = ld4_bf16(res1 + at(a.res1_map, t * 32 + 8 * g + 4 * hh));
        }
        if (res2) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) r2[t][g] = ld4_bf16(res2 + at(a.res2_map, t * 32 + 8 * g + 4 * hh));
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
        // 16-byte stores: for the run pair (g = 2j, 2j+1) the lower half-wave ends up with channels
        // 16j..16j+7 of its pixel (own run 2j + the upper lane's run 2j) and the upper half-wave with
        // 16j+8..16j+15 (the lower lane's run 2j+1 + own run 2j+1).  permlane32_swap(x, y): lanes 32-63
        // of x <-> lanes 0-31 of y.  All lanes execute the swaps (no divergence before them).
        uint4 wide[NT][2];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const uint2 e = pack4_bf16(v[t][2 * j]), o = pack4_bf16(v[t][2 * j + 1]);
                const auto sx = __builtin_amdgcn_permlane32_swap(e.x, o.x, false, false);
                const auto sy = __builtin_amdgcn_permlane32_swap(e.y, o.y, false, false);
                wide[t][j] = uint4{sx[0], sy[0], sx[1], sy[1]};
            }
        if (valid) {
            // lane (pixel, hh) stores channels t*32 + 16j + 8hh .. +7
            if (out) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        store16(out + at(a.out_map, a.out_coff + t * 32 + 16 * j + 8 * hh), wide[t][j]);
            }
            if (out2) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        store16(out2 + at(a.out2_map, t * 32 + 16 * j + 8 * hh), wide[t][j]);
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

template <int NT, int WAVES, int ISLOTS>
hipError_t launch_xl(const ConvArgs& a, hipStream_t s) {
    typedef Geo<WAVES> G;
    constexpr size_t shm = (size_t)ISLOTS * G::IN_BYTES + 2 * (size_t)(9 * 2 * 32 * NT * 16);
    static unsigned long long attr_done = 0;
    {
        const hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&conv3x3_bf16_xl_kernel<NT, WAVES, ISLOTS>), shm, attr_done);
        if (e != hipSuccess) return e;
    }
    const int tiles = ((a.w_ + TW - 1) / TW) * ((a.h + G::TH - 1) / G::TH) * a.n;
    hipLaunchKernelGGL((conv3x3_bf16_xl_kernel<NT, WAVES, ISLOTS>), dim3(tiles), dim3(G::THREADS), shm, s, a);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_conv3x3_bf16_xl(const ConvArgs& a, hipStream_t s) {
    if (a.y_lo || a.y_hi) return hipErrorInvalidValue;   // row ranges: conv3x3_f16x2_kernel only
    if (a.cin % 16 || !a.zeros) return hipErrorInvalidValue;
    // the 16-byte accesses of the pair image / wide stores need pixel strides of whole 8-channel groups
    if (a.in_map.pix % 8 || (a.out && (a.out_map.pix % 8 || a.out_coff % 16))) return hipErrorInvalidValue;
    static const int geo = [] {   // NESR_XL_GEOMETRY=8 (32x32 tiles, 1 WG/CU) | 4 (16x32 tiles, 2 WGs/CU)
        const char* e = getenv("NESR_XL_GEOMETRY");
        return e ? atoi(e) : 4;
    }();
    if (geo == 8) {
        if (a.coutp == 64) return launch_xl<2, 8, 3>(a, s);
        if (a.coutp == 32) return launch_xl<1, 8, 3>(a, s);
    } else {
        if (a.coutp == 64) return launch_xl<2, 4, 2>(a, s);   // 2 x 19.6 + 2 x 18.4 = 76 KB -> 2 per CU
        if (a.coutp == 32) return launch_xl<1, 4, 3>(a, s);   // 3 x 19.6 + 2 x 9.2 = 77 KB -> 2 per CU
    }
    return hipErrorInvalidValue;
}

}  // namespace nesr
