// f32 3x3 convolution on the f16 matrix cores: every f32 operand travels as an exact-sum pair of
// halves, x = hi + lo * 2^-11 with hi = f16(x), lo = f16((x - hi) * 2^11), and the product is
// accumulated in f32 from three MFMAs
//
//     w*x  ~=  w_hi*x_hi + 2^-11 (w_hi*x_lo + w_lo*x_hi)        (dropped: w_lo*x_lo ~ 2^-22 |w x|)
//
// The two cross terms have their own f32 accumulator and are scaled once in the epilogue.  Precision of
// the pair: hi carries 11 significant bits and lo (kept normal by the 2^11 scale) the next 11, so
// |x - (hi + lo 2^-11)| <= 2^-22 |x| for 6.1e-5 <= |x| <= 65504; below 6.1e-5 hi is an f16 subnormal
// (absolute step 2^-24) and the scaled lo resolves the rest to an absolute 2^-35.  Range: |x| > 65504
// or a non-finite x does not fit; such a value is clamped AND reported -- a sticky device flag
// (ConvArgs::status) that nesr_check_status / nesr_check_range turn into NESR_ERR_RANGE and that makes
// conv_last write NaN instead of a saturated image (a diverged network gives NaN in the reference too).
//
// v_mfma_f32_32x32x16_f16 runs at 16x the rate of the f32 MFMA, so three of them cost 3/16 of the
// direct f32 kernel's matrix time and 27/64 of the Winograd kernel's.  Measured on the 23-block
// network against an f64 evaluation of the same weights (tools/probes/split_accuracy.py): max abs
// error 3.1e-6 for this scheme, 1.2e-6 for plain f32 (torch CPU), i.e. the same class of error
// as the Winograd kernel -- 300x inside the 1e-3 tolerance.  (The MFMA honours f16 subnormal inputs on
// gfx950, tools/probes/mfma_f16_denorm.hip: hi needs no special case.)
//
// Activations live pre-split in HBM, channel-blocked like the bf16 path: per 16-channel K-chunk a
// pixel owns 64 bytes = [16 hi halves | 16 scaled-lo halves] (Map: pix = 32, chunk = pixels * 32, in
// 2-byte units); the producing kernel's epilogue splits once, the consumers feed LDS by LDS-DMA
// without touching a VGPR.  Same 4 bytes per value as f32 storage.
//
//   workgroup : 4 MFMA waves (+ 4 DMA-only waves when a CU gets one workgroup), output tile 8 rows x 32 cols
//               x 32 output channels; wave w owns rows 2w, 2w+1.  Layers with 64 output channels run two
//               workgroups per tile (adjacent in the launch order).
//   K loop    : 16-channel chunks; input halo tile [(rows+2) x 34 pixels][64 B] and weight slab
//               [9 taps][hi|lo][k half][32 couts][16 B] by LDS-DMA into 2-slot rings, counted
//               vmcnt waits + one barrier per chunk (as conv3x3_bf16.hip).
//   LDS image : pixel p's 16-byte slot s = 2*plane + k-half sits at p*64 + (s ^ ((col>>2)&3))*16:
//               the 16-lane groups of ds_read_b128 (MI355X_MICROARCH.md, LDS) then cover all 16
//               slots of a 256-byte bank row for every horizontal tap.
//   epilogue  : bias / LeakyReLU / residuals in f32 (residuals re-joined hi + lo), split, 16-byte
//               write-through stores of 8 channels per plane; conv_last writes planar f32 / u8.
#include <cstdlib>
#include <type_traits>

#include "nesr_kernels.h"

namespace nesr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char lds_char;

namespace {

constexpr int TW = 32, PW = TW + 2;
constexpr int WAVES = 4;   // MFMA waves per workgroup
constexpr int RW = 2;      // rows per MFMA wave: output tile 8 rows x 32 cols x 32 output channels

// DMAW = extra waves that only issue the LDS-DMAs (0: the MFMA waves issue them themselves)
template <int DMAW>
struct Geo {
    static constexpr int LAUNCH_THREADS = 64 * (WAVES + DMAW);
    static constexpr int THREADS = 64 * (DMAW ? DMAW : WAVES);   // lanes that share one DMA round
    static constexpr int TH = WAVES * RW;
    static constexpr int PH = TH + 2;
    static constexpr int NPIX = PH * PW;
    static constexpr int IN_ITEMS = 4 * NPIX;   // 16-byte items per input slot
    static constexpr int IN_ROUNDS = (IN_ITEMS + THREADS - 1) / THREADS;
    static constexpr int IN_BYTES = IN_ITEMS * 16;
};
constexpr int W_ITEMS = 9 * 2 * 2 * 32;   // 16-byte items of one 32-cout weight slab (per K-chunk)
constexpr int W_BYTES = W_ITEMS * 16;
constexpr int ISLOTS = 2;                 // two-slot rings: every wait is vmcnt(0)

// LDS-DMA from inline asm (see conv3x3_bf16.hip): not counted by hipcc, waited for by hand; scalar 64-bit base and an
// unsigned 32-bit per-lane byte offset
__device__ __forceinline__ void glds16_s(const char* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_dst)
        : "memory");
}
// Plain (write-back) stores: every workgroup of a layer reaches its epilogue at about the same time, and
// the write-through form (sc1) that helps the Winograd kernel made this burst 3x longer here
// (in-kernel stamps: 12.4k -> 3.7k cycles per epilogue; -4 % / -6 % forward time on 1 / 6 tiles).
__device__ __forceinline__ void store16(uint16_t* p, uint4 v) {
#ifdef NESR_SC1_STORES
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(__builtin_bit_cast(f32x4, v)) : "memory");
#else
    *reinterpret_cast<uint4*>(p) = v;
#endif
}

constexpr float LO_SCALE = 2048.f, LO_INV = 1.f / 2048.f;   // lo is stored as f16((x - hi) * 2^11)

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// two values -> packed (hi, hi), (lo, lo) halves.  hi = f16(x) (round to nearest even), lo = f16((x - hi) * 2^11):
// x - hi is exact in f32 and so is the scaling, hence fma(hi, -2^11, x * 2^11) is that value bit for bit.  A value that
// does not fit (|x| >= 65520, Inf, NaN) makes hi Inf / NaN: `badbits` collects the all-ones exponent fields (bit 15 of a
// half after adding 0x0400 to its masked exponent), so nothing is clamped -- a non-finite half poisons the sums it enters,
// the sticky flag is raised and conv_last writes NaN.
__device__ __forceinline__ void split2(float x0, float x1, unsigned& hi, unsigned& lo, unsigned& badbits) {
    const f32x2 x = {x0, x1};
    const f16x2 h = __builtin_convertvector(x, f16x2);                        // v_cvt_pk_f16_f32
    const f32x2 hf = __builtin_convertvector(h, f32x2);
    const f32x2 d = __builtin_elementwise_fma(hf, f32x2{-LO_SCALE, -LO_SCALE}, x * LO_SCALE);   // packed f32 mul / fma
    const f16x2 l = __builtin_convertvector(d, f16x2);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
    badbits |= (hi & 0x7C007C00u) + 0x04000400u;
}
// x -> (hi, scaled lo) halves of 4 values; `bad` collects "does not fit the pair" (|x| >= 65520, NaN, Inf)
__device__ __forceinline__ void split4(f32x4 v, uint2& hi, uint2& lo, bool& bad) {
    unsigned bits = 0;
    split2(v[0], v[1], hi.x, lo.x, bits);
    split2(v[2], v[3], hi.y, lo.y, bits);
    bad |= (bits & 0x80008000u) != 0u;
}


// ---- whole-line feature-map access for the 16x16x32 epilogue.  After the cout exchange lane (pixel j16, k-group g4)
// holds 8 consecutive couts of ONE pixel: base 0 / 16 / 8 / 24 for g4 = 0 / 1 / 2 / 3, i.e. piece g4>>1 (8 channels)
// of chunk X (g4 even) or X + 1 (g4 odd), as a hi and a lo 16-byte piece.  Stored like that, a wave instruction
// would write 16-byte fragments of 64 different 64-byte slots (measured: ~4.6 us per epilogue, the stores crawl
// through the address coalescer).  One v_permlane16_swap per register (odd rows of `hi` <-> even rows of `lo`)
// regroups the pieces by CHUNK: afterwards `hi` holds, for every lane, a piece of chunk X -- piece index
// 2 (g4 & 1) + (g4 >> 1) of the pixel's 64-byte slot [hi 0-7 | hi 8-15 | lo 0-7 | lo 8-15] -- and `lo` the same piece
// of chunk X + 1: two store instructions of 16 pixels x 64 bytes = 1 KiB of whole lines each.  The swap is its own
// inverse, so residuals are loaded the same way.  All 64 lanes must execute these (no divergence around them).
__device__ __forceinline__ void regroup_pairs(uint4& hi, uint4& lo) {
    auto s0 = __builtin_amdgcn_permlane16_swap(hi.x, lo.x, false, false);
    auto s1 = __builtin_amdgcn_permlane16_swap(hi.y, lo.y, false, false);
    auto s2 = __builtin_amdgcn_permlane16_swap(hi.z, lo.z, false, false);
    auto s3 = __builtin_amdgcn_permlane16_swap(hi.w, lo.w, false, false);
    hi = uint4{s0[0], s1[0], s2[0], s3[0]};
    lo = uint4{s0[1], s1[1], s2[1], s3[1]};
}
// v0, v1 = this lane's 8 couts -> split, regrouped: `cx` goes to chunk X, `cx1` to chunk X + 1 (at piece offset)
__device__ __forceinline__ void split_regroup(f32x4 v0, f32x4 v1, uint4& cx, uint4& cx1, bool& bad) {
    uint2 h0, l0, h1, l1;
    split4(v0, h0, l0, bad);
    split4(v1, h1, l1, bad);
    cx = uint4{h0.x, h0.y, h1.x, h1.y};
    cx1 = uint4{l0.x, l0.y, l1.x, l1.y};
    regroup_pairs(cx, cx1);
}
// the inverse for residuals: p = the pixel's slot in chunk X (2-byte units), piece8 = this lane's piece offset
__device__ __forceinline__ void load_regrouped(const uint16_t* p, long long chunk_el, f32x4& q0, f32x4& q1) {
    uint4 cx = *reinterpret_cast<const uint4*>(p);
    uint4 cx1 = *reinterpret_cast<const uint4*>(p + chunk_el);
    regroup_pairs(cx, cx1);      // -> own hi, own lo
    const f16x8 h = __builtin_bit_cast(f16x8, cx), l = __builtin_bit_cast(f16x8, cx1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        q0[i] = fmaf((float)l[i], LO_INV, (float)h[i]);
        q1[i] = fmaf((float)l[4 + i], LO_INV, (float)h[4 + i]);
    }
}

#ifndef NESR_ABL
#define NESR_ABL 0   // timing ablations: 1 empty kernel, 2 stop after the prologue, 4 no epilogue, 8 no MFMA, 16 no K-loop DMA, 64 cycle stamps, 512 no per-chunk barrier
#endif
#if NESR_ABL & 64
__device__ unsigned long long g_stamps[256];
#define STAMP(i) do { if (stamping) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); if (lane == 0 && (i) < 200) g_stamps[(i)] = t_; \
    if ((i) == 0 || (i) == 4) { unsigned long long r_ = __builtin_amdgcn_s_memrealtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); if (lane == 0) g_stamps[200 + (i)] = r_; } } } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

template <int DMAW>
__global__ __launch_bounds__(64 * (WAVES + DMAW), DMAW ? (WAVES + DMAW) / 4 : 2) void conv3x3_f16x2_kernel(ConvArgs a) {
    if (NESR_ABL & 1) return;
    typedef Geo<DMAW> G;
    constexpr int THREADS = G::THREADS, TH = G::TH;
    constexpr int IN_ITEMS = G::IN_ITEMS, IN_ROUNDS = G::IN_ROUNDS, IN_BYTES = G::IN_BYTES;
    constexpr int W_ROUNDS = (W_ITEMS + THREADS - 1) / THREADS;
    constexpr int WRING = ISLOTS * IN_BYTES;   // LDS: [input ring][weight ring: 2 x W_BYTES]
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool is_dma = DMAW ? wave_all >= WAVES : true;     // issues the LDS-DMAs
    const bool is_cmp = wave_all < WAVES;                    // runs the MFMAs and the epilogue
    const int wave = is_cmp ? wave_all : wave_all - WAVES;   // index inside its role
    const int tid = wave * 64 + lane;
#if NESR_ABL & 64
    const bool stamping = blockIdx.x == 77 && wave_all == 1;
#endif
    STAMP(0);

    // ---- XCD-aware work index (bijective for any count); the cout groups of one tile are neighbours
    const int CG = a.coutp / 32;
    const int y_lo = a.y_lo, y_hi = a.y_hi > 0 ? a.y_hi : a.h;      // output rows of this launch
    const int tiles_x = (a.w_ + TW - 1) / TW;
    const int tiles_y = (y_hi - y_lo + TH - 1) / TH;
    const int total = tiles_x * tiles_y * a.n * CG;
    int idx;
    {
        const int bid = blockIdx.x, q = total >> 3, r = total & 7, xcd = bid & 7;
        idx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    STAMP(5);
    const int cg = idx % CG;
    int tile = idx / CG;
    const int n = tile / (tiles_x * tiles_y);
    tile -= n * tiles_x * tiles_y;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int y0 = y_lo + ty * TH, x0 = tx * TW;

    // ---- LDS-DMA plan.  LDS item k of an input slot = padded pixel k>>2, physical slot k&3; it holds
    // logical slot (k&3) ^ (bit 2 of the padded column << 1) of that pixel's 64 bytes.  Item k = tid + THREADS*i
    // belongs to this lane in round i; its source is (scalar base of image n and chunk c) + voff[i].
    // Out-of-image items (the conv's zero padding) are the same for every chunk: they are zeroed once in
    // every ring slot and left out of the DMAs (EXEC-masked lanes do not write).
    const unsigned lds_base = (unsigned)(size_t)(lds_char*)(smem);
    unsigned voff[IN_ROUNDS];
    unsigned okmask = 0;
    // scalar base = image n, first stored row this tile reads; the per-lane offsets below are relative to it and stay
    // below (tile rows + 2) * in_w * 64 bytes, so frames of any size address correctly (64-bit base, 32-bit offsets)
    const int row0 = (y0 > 0 ? y0 - 1 : 0) >> a.up;
    const char* in_img = static_cast<const char*>(a.in) + ((size_t)n * a.in_h + row0) * a.in_w * 64;
    const long long in_cstride = a.in_map.chunk * 2;   // bytes between K-chunks
    const char* wbase = static_cast<const char*>(a.w) + (size_t)cg * W_BYTES;
    const long long w_cstride = (long long)CG * W_BYTES;
    // DMA round j of chunk c: rounds [0, W_ROUNDS) move the weight slab, the rest the input tile
    constexpr int NDMA = W_ROUNDS + IN_ROUNDS;
    auto dma_round = [&](int c, int slot, int j) {   // j is a compile-time constant at every call site
        if (j < W_ROUNDS) {
            const int k = tid + THREADS * j;
            const unsigned dst = lds_base + WRING + slot * W_BYTES + j * (THREADS * 16) + wave * 1024;
            if (k < W_ITEMS) glds16_s(wbase + (long long)c * w_cstride, (unsigned)k * 16u, __builtin_amdgcn_readfirstlane(dst));
        } else if (j < NDMA) {
            const int i = j - W_ROUNDS;
            const unsigned dst = lds_base + slot * IN_BYTES + i * (THREADS * 16) + wave * 1024;
            if ((okmask >> i) & 1u) glds16_s(in_img + (long long)c * in_cstride, voff[i], __builtin_amdgcn_readfirstlane(dst));
        }
    };
    // the plan and chunk 0's DMAs together: the weight slab first (needs no plan), then every input round as
    // soon as its offsets exist, so the first bytes are under way while the rest is still being computed
    if (is_dma) {
#pragma unroll
        for (int j = 0; j < W_ROUNDS; ++j) dma_round(0, 0, j);
        int p = tid >> 2;
        int py = p / PW, px = p - py * PW;
        const int sl = tid & 3;
#pragma unroll
        for (int i = 0; i < IN_ROUNDS; ++i) {
            const int k = tid + THREADS * i;
            const int sg = sl ^ (((px >> 2) & 1) << 1);
            const int Y = y0 - 1 + py, X = x0 - 1 + px;
            const bool has = k < IN_ITEMS;
            const bool ok = has && Y >= 0 && Y < a.h && X >= 0 && X < a.w_;
            voff[i] = ((unsigned)((Y >> a.up) - row0) * (unsigned)a.in_w + (unsigned)(X >> a.up)) * 64u + sg * 16;
            okmask |= ok ? (1u << i) : 0u;
            dma_round(0, 0, W_ROUNDS + i);
            if (has && !ok) {
#pragma unroll
                for (int sl2 = 0; sl2 < ISLOTS; ++sl2) *reinterpret_cast<f32x4*>(smem + sl2 * IN_BYTES + k * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            px += (THREADS / 4) % PW;
            py += (THREADS / 4) / PW;
            if (px >= PW) { px -= PW; py += 1; }
        }
    }
    STAMP(6);

    const bool active = is_cmp && (y0 + RW * wave) < y_hi;

    // ---- operands.  The MFMA (v_mfma_f32_16x16x32_f16) has K = 32 = two units of 16 channels: lanes 0-31
    // (k-groups g = 0,1 = channel halves) feed unit 0, lanes 32-63 unit 1, and a unit is one (tap, plane)
    // choice -- just another per-lane LDS address.  Steps 0-2 pair the taps (dy 0 | dy 1) of column dx = step,
    // step 3 pairs (2,0) | (2,1); with WH/WL = [w(tap a) | w(tap b)] and XH/XL = [x(tap a) | x(tap b)] a step is
    // main += WH*XH, cross += WH*XL + WL*XH.  Step 4 is tap (2,2) alone: main += [w_hi | 0] * [x_hi | .] and
    // cross += [w_hi | w_lo] * [x_lo | x_hi], i.e. both cross terms in one MFMA.  14 MFMAs per 16x16 tile and
    // chunk instead of 13.5; the chip holds a ~20 % higher clock on this shape than on 32x32x16
    // (tools/probes/mfma_shape.hip).  Tiles of a wave: rows r, pixel halves nh (16 px), cout halves mt
    // (16 couts).  LDS slot swizzle for this lane order: physical slot = slot ^ (bit 2 of the padded column << 1)
    // (conflict-free for the ds_read_b128 lane groups).
    const int j16 = lane & 15, g4 = lane >> 4, un = g4 >> 1, kh = g4 & 1;
    int b16[5][2], a16[5];
#pragma unroll
    for (int st_ = 0; st_ < 5; ++st_) {
        const int dy = st_ < 3 ? un : 2;
        const int dx = st_ < 3 ? st_ : (st_ == 3 ? un : 2);
#pragma unroll
        for (int nh = 0; nh < 2; ++nh) {
            const int col = 16 * nh + j16 + dx;
            b16[st_][nh] = ((RW * wave + dy) * PW + col) * 64 + ((kh ^ (((col >> 2) & 1) << 1)) << 4);
        }
        a16[st_] = ((((dy * 3 + dx) * 2) * 2 + kh) * 32 + j16) * 16;
    }
    f32x4 acc16[RW][2][2][2];       // [row][pixel half][cout half][main | cross]
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int q = 0; q < 2; ++q) acc16[r][nh][mt][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    // this lane's 8 output channels after the epilogue's permlane16 exchange, and their bias (fetched now: the
    // latency hides under the K loop)
    const int cb16 = 32 * cg + (g4 & 1) * 16 + (g4 >> 1) * 8;
    const int piece8 = ((g4 & 1) * 2 + (g4 >> 1)) * 8;    // this lane's 16-byte piece of a 64-byte slot after regroup_pairs (2-byte units)
    const f32x4 bz0 = *reinterpret_cast<const f32x4*>(a.bias + cb16), bz1 = *reinterpret_cast<const f32x4*>(a.bias + cb16 + 4);

    const int nchunks = a.cin / 16;
    STAMP(7);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the zero padding is in LDS before the first barrier
    STAMP(1);
    if (NESR_ABL & 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        return;
    }
    for (int c = 0; c < nchunks; ++c) {
        // this wave's DMAs of chunk c (issued during chunk c-1) have landed; after the barrier, everybody's
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        STAMP(8 + 4 * c);
        if (!(NESR_ABL & 512)) __builtin_amdgcn_s_barrier();   // 512: timing only (races)
        asm volatile("" ::: "memory");
        STAMP(9 + 4 * c);
        // the next chunk's DMAs go out first (beside the MFMAs each costs 150+ cycles of issue instead of ~85:
        // measured, in-kernel stamps)
        if (is_dma && c + 1 < nchunks && !(NESR_ABL & 16)) {
#pragma unroll
            for (int j = 0; j < NDMA; ++j) dma_round(c + 1, (c + 1) & 1, j);
        }
        STAMP(10 + 4 * c);
        if (active) {
            const char* st = smem + (c & 1) * IN_BYTES;
            const char* swb = smem + WRING + (c & 1) * W_BYTES;
            f32x4 Af[2][2][2];        // [buffer][cout half][variant]
            f32x4 Bf[2][RW][2][2];    // [buffer][row][pixel half][variant]
            auto load_step = [&](int s_, int buf) {   // s_ is a compile-time constant at every call site
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    if (s_ < 4) {
                        Af[buf][mt][0] = *reinterpret_cast<const f32x4*>(swb + a16[s_] + mt * 256);          // WH
                        Af[buf][mt][1] = *reinterpret_cast<const f32x4*>(swb + a16[s_] + mt * 256 + 1024);   // WL
                    } else {
                        f32x4 hi = *reinterpret_cast<const f32x4*>(swb + a16[4] + mt * 256);
                        Af[buf][mt][1] = *reinterpret_cast<const f32x4*>(swb + a16[4] + mt * 256 + un * 1024);   // [w_hi | w_lo]
                        if (un) hi = f32x4{0.f, 0.f, 0.f, 0.f};
                        Af[buf][mt][0] = hi;                                                                      // [w_hi | 0]
                    }
                }
#pragma unroll
                for (int r = 0; r < RW; ++r)
#pragma unroll
                    for (int nh = 0; nh < 2; ++nh) {
                        const int o0 = b16[s_][nh];                                             // XH; last tap: [x_hi | x_hi]
                        const int o1 = s_ == 4 ? (b16[4][nh] ^ ((un ^ 1) << 5)) : (b16[s_][nh] ^ 32);   // XL; last tap: [x_lo | x_hi]
                        Bf[buf][r][nh][0] = *reinterpret_cast<const f32x4*>(st + o0 + r * (PW * 64));
                        Bf[buf][r][nh][1] = *reinterpret_cast<const f32x4*>(st + o1 + r * (PW * 64));
                    }
            };
            load_step(0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s_ = 0; s_ < 5; ++s_) {
                const int buf = s_ & 1;
                if (s_ + 1 < 5) load_step(s_ + 1, buf ^ 1);
                if (NESR_ABL & 8) {
                    acc16[0][0][0][0][0] += Af[buf][0][0][0] + Af[buf][1][1][0] + Bf[buf][0][0][0][0] + Bf[buf][RW - 1][1][1][0];
                    continue;
                }
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const f16x8 a0 = __builtin_bit_cast(f16x8, Af[buf][mt][0]), a1 = __builtin_bit_cast(f16x8, Af[buf][mt][1]);
#pragma unroll
                    for (int r = 0; r < RW; ++r)
#pragma unroll
                        for (int nh = 0; nh < 2; ++nh) {
                            const f16x8 x0_ = __builtin_bit_cast(f16x8, Bf[buf][r][nh][0]), x1_ = __builtin_bit_cast(f16x8, Bf[buf][r][nh][1]);
                            if (s_ < 4) {
                                acc16[r][nh][mt][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, x1_, acc16[r][nh][mt][1], 0, 0, 0);
                                acc16[r][nh][mt][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, x0_, acc16[r][nh][mt][0], 0, 0, 0);
                                acc16[r][nh][mt][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, x0_, acc16[r][nh][mt][1], 0, 0, 0);
                            } else {
                                acc16[r][nh][mt][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, x0_, acc16[r][nh][mt][0], 0, 0, 0);
                                acc16[r][nh][mt][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, x1_, acc16[r][nh][mt][1], 0, 0, 0);
                            }
                        }
                }
                // the next step's 4 + 4 RW fragment reads ride between this step's MFMAs (two MFMAs, one read)
                // instead of in front of them
                if (s_ + 1 < 5) {
#pragma unroll
                    for (int i = 0; i < 4 + 4 * RW; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // MFMA
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        STAMP(11 + 4 * c);
    }
    STAMP(2);
    if (!active) return;
    if (NESR_ABL & 4) {
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < RW; ++r)
#pragma unroll
            for (int e = 0; e < 16; ++e) sum += acc16[r][e & 1][(e >> 1) & 1][(e >> 2) & 1][e & 3];
        if (sum == 12345.678f) static_cast<float*>(a.out)[0] = sum;
        return;
    }

    // ---- epilogue.  C/D layout: lane (pixel j16, k-group g4) holds couts 16 mt + 4 g4 + i.
    // v_permlane16_swap of the mt = 0 / mt = 1 values leaves every lane with 8 consecutive couts of its
    // pixel: base 0 / 16 / 8 / 24 for g4 = 0 / 1 / 2 / 3 (tools/probes/mfma16_layout.hip).
    const int cb = cb16;
    const uint16_t* res1 = static_cast<const uint16_t*>(a.res1);
    const uint16_t* res2 = static_cast<const uint16_t*>(a.res2);
    uint16_t* out = static_cast<uint16_t*>(a.out);
    uint16_t* out2 = static_cast<uint16_t*>(a.out2);
    bool bad = false;
    // conv_last: a range failure anywhere upstream in this forward (sticky word, set by earlier launches on this
    // stream) turns the image into NaN instead of a saturated picture
    const bool poison = a.cout_real > 0 && a.status && __builtin_nontemporal_load(a.status) != 0u;
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int nh = 0; nh < 2; ++nh) {
            const int Y = y0 + RW * wave + r, X = x0 + 16 * nh + j16;
            const bool valid = X < a.w_ && Y < y_hi;
            const size_t pix = ((size_t)n * a.h + (Y < a.h ? Y : 0)) * a.w_ + (X < a.w_ ? X : 0);
            f32x4 v0, v1;   // couts cb .. cb+3, cb+4 .. cb+7
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float e = fmaf(acc16[r][nh][0][1][i], LO_INV, acc16[r][nh][0][0][i]);
                const float o = fmaf(acc16[r][nh][1][1][i], LO_INV, acc16[r][nh][1][0][i]);
                const auto sw_ = __builtin_amdgcn_permlane16_swap(__float_as_uint(e), __float_as_uint(o), false, false);
                v0[i] = __uint_as_float(sw_[0]);
                v1[i] = __uint_as_float(sw_[1]);
            }
            v0 += bz0;
            v1 += bz1;
            if (a.lrelu) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { v0[i] = fmaxf(v0[i], v0[i] * 0.2f); v1[i] = fmaxf(v1[i], v1[i] * 0.2f); }   // LeakyReLU(0.2), same values as the select form
            }
            // slot of this pixel in chunk X = (coff + 32 cg) / 16 of a map, at this lane's piece
            auto slot = [&](const Map& mp, int coff) -> size_t { return (size_t)((coff + 32 * cg) >> 4) * mp.chunk + pix * mp.pix + piece8; };
            if (res1) {
                f32x4 q0, q1;
                load_regrouped(res1 + slot(a.res1_map, 0), a.res1_map.chunk, q0, q1);
#pragma unroll
                for (int i = 0; i < 4; ++i) { v0[i] = __fadd_rn(__fmul_rn(v0[i], a.s1), q0[i]); v1[i] = __fadd_rn(__fmul_rn(v1[i], a.s1), q1[i]); }
            }
            if (res2) {
                f32x4 q0, q1;
                load_regrouped(res2 + slot(a.res2_map, 0), a.res2_map.chunk, q0, q1);
#pragma unroll
                for (int i = 0; i < 4; ++i) { v0[i] = __fadd_rn(__fmul_rn(v0[i], a.s2), q0[i]); v1[i] = __fadd_rn(__fmul_rn(v1[i], a.s2), q1[i]); }
            }
            if (out || out2) {
                uint4 cx, cx1;
                bool bad_here = false;
                split_regroup(v0, v1, cx, cx1, bad_here);
                bad |= bad_here && valid;
                if (valid && out) {
                    uint16_t* p = out + slot(a.out_map, a.out_coff);
                    store16(p, cx);
                    store16(p + a.out_map.chunk, cx1);
                }
                if (valid && out2) {
                    uint16_t* p = out2 + slot(a.out2_map, 0);
                    store16(p, cx);
                    store16(p + a.out2_map.chunk, cx1);
                }
            }
            if (!valid) continue;
            if (a.cout_real > 0 && cb == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (q >= a.cout_real) break;
                    const float x = poison ? __builtin_nanf("") : v0[q];
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
    if (bad && a.status) __hip_atomic_store(a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    STAMP(3);
#if NESR_ABL & 64
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(4);
#endif
}

// ======================================================================================================
// One residual dense block per launch (frames whose 8x32-pixel tiles fit the CUs, e.g. 512x512 x2plus: 256 tiles).
//
// A per-layer launch of such a frame is one round of 256-512 workgroups that all wait for their first bytes at the
// same time, all compute, all store: ~6.6 us of every ~20 us launch is not matrix work.  Here
// a workgroup keeps its tile through conv1..conv5 of the block: 52 K-chunk steps (4 + 6 + 8 + 10 + 2 x 12) in ONE
// software pipeline -- the DMA waves run one step ahead straight across the layer boundaries, the LDS-DMA plan is
// computed once, and a layer's stores drain while the next layer's first chunks are already being multiplied.
//
// The only thing a layer needs from other workgroups is the 1-pixel halo of the 32 channels the previous layer has
// just produced -- and in a dense block those are the LAST two of its 6..12 input chunks.  So the wait is placed there:
// the DMA waves poll the neighbours' progress words (one per tile: x1..x4 published) just before they fetch those two
// chunks, several microseconds after the neighbours stored them.  Protocol (MI355X_MICROARCH.md, inter-workgroup
// visibility): producer = write-through (sc1) 16-byte stores, every storing wave's s_waitcnt vmcnt(0), the
// workgroup barrier, one lane's relaxed agent-scope store of the progress word; consumer = relaxed agent-scope
// (sc1) polls by the wave that then issues the loads, loads that bypass L1 (sc1).  Nothing is read before it has
// been written inside one launch and a kernel boundary invalidates L1/L2, so no line can be stale.  Every tile has its
// own resident workgroup (grid <= CUs, one workgroup per CU): waits are bounded and set an abort word instead of hanging.
struct RdbArgs {
    const void* cur;         // the block's 192-channel buffer: x0 read, x1..x4 written then read
    long long chunk_bytes;   // bytes between its 16-channel chunks (pixels * 64)
    void* out;               // conv5's destination: the x0 slice of the next block's buffer (same geometry)
    const void* res2;        // second residual (the RRDB's input, x0 slice of its first buffer) or null
    float s1, s2;
    const void* w[5];        // packed weights of conv1..conv5 (pack_weights_f16x2)
    const float* bias[5];
    int n, h, w_;
    unsigned* progress;      // [tiles] epoch + layers published
    unsigned epoch;          // progress == epoch + j  <=>  x_j of this launch is visible
    unsigned* abort_flag;
    unsigned* status;        // sticky range word (ConvArgs::status)
    unsigned long long timeout_ticks;   // s_memrealtime ticks (100 MHz) a neighbour wait may take
};

__device__ __forceinline__ void glds16_s_sc1(const char* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2 sc1\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_dst)
        : "memory");
}
__device__ __forceinline__ void store16_wt_s(const char* sbase, unsigned voff, uint4 v) {   // write-through (visible device-wide once vmcnt retires it): wave-uniform base + 32-bit lane offset
    asm volatile("global_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" ::"v"(voff), "v"(__builtin_bit_cast(f32x4, v)), "s"(sbase) : "memory");
}

#ifndef NESR_RDB_ABL
#define NESR_RDB_ABL 0   // timing ablations (WRONG results): 1 no neighbour polling, 2 plain activation loads, 4 plain x1..x4 stores, 8 no MFMA, 16 no epilogue
#endif

// s_waitcnt vmcnt(k) for a wave-uniform run-time k (the count is an immediate)
__device__ __forceinline__ void wait_vmcnt_le(int k) {
    switch (k) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    }
}

#if NESR_RDB_ABL & 256
__device__ unsigned long long g_rdb_stamps[2][64][8];     // [role: MFMA wave 1 | DMA wave 0][step][event] of workgroup 77
__device__ unsigned long long g_rdb_taps[2][8][8];         // [MFMA wave 1 | 5 (one SIMD)][step - 20][start of tap-step 0..4, end] of workgroup 77
__device__ unsigned long long g_rdb_arrive[12][8][2];      // [wave][step - 20][arrival at | release from the step's barrier] of workgroup 77
#define RARRIVE(w, step, ev) do { if (blockIdx.x == 77 && lane == 0 && (step) >= 20 && (step) < 28) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); g_rdb_arrive[w][(step) - 20][ev] = t_; } } while (0)
#define RSTAMP(role, step, ev) do { if (blockIdx.x == 77 && lane == 0) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); g_rdb_stamps[role][step][ev] = t_; } } while (0)
#else
#define RSTAMP(role, step, ev) do { } while (0)
#define RARRIVE(w, step, ev) do { } while (0)
#endif

constexpr int RSLOTS = 4;   // ring slots of the fused kernel: the DMA waves run three steps ahead, step s + 1 has landed at barrier s
constexpr int MW = 8;       // MFMA waves of the fused kernel: one tile row each, two per SIMD
constexpr int DW = 4;       // DMA waves: LDS-DMA, neighbour polling, progress words
#ifndef NESR_RDB_PAR
#define NESR_RDB_PAR 0   // -1: the two MFMA waves of a SIMD run their epilogue halves at opposite ends of a step; 0 / 1: both at the start / end
#endif
#ifndef NESR_RDB_EPI_STEPS
#define NESR_RDB_EPI_STEPS 1
#endif
constexpr int EPI_STEPS = NESR_RDB_EPI_STEPS;   // a finished layer's epilogue rides in the first 1 or 2 steps of the next one
// The DMA waves fetch RSLOTS - 1 steps ahead: the first x1 chunk of conv2 (its step 4) is requested in conv2's step 1, and
// that request polls this tile's own progress word among the nine -- which goes out EPI_STEPS steps into the layer.
static_assert(EPI_STEPS <= 4 - (RSLOTS - 1), "the tile would wait for its own progress word");

// Roles.  MFMA waves (0..7): row w of the 8x32-pixel tile, 32 couts: LDS fragment reads, MFMAs, and the epilogue of
// the PREVIOUS (layer, cout group) -- its sums wait in 16 registers (main + cross / 2^11) and are finished (bias,
// LeakyReLU / residuals, split, whole-line stores) at the start of the next layer's first step, by both waves of a SIMD
// at the same point (NESR_RDB_PAR 0).  Measured alternatives: the epilogue in the DMA waves, or in
// four waves of its own -- a VALU instruction of a wave that shares its SIMD with two MFMA waves gets one issue slot
// per MFMA, ~16 cycles each: layer boundaries cost 23 % of the kernel; staggered between the two MFMA waves of a SIMD
// (one before, one after its MFMAs) -- a VALU block beside the partner's MFMA stream crawls just the same.
// DMA waves (8..11): one step's LDS-DMAs three steps ahead (4-slot ring), polling of the neighbours' progress words
// before the first chunk of each x_l, and this tile's own progress word EPI_STEPS steps into the next layer.
__global__ __launch_bounds__(64 * (MW + DW), 3) void rdb_f16x2_kernel(RdbArgs a) {
    typedef Geo<4> G;
    constexpr int THREADS = G::THREADS, TH = G::TH;      // THREADS = DMA lanes (256), TH = 8 rows
    constexpr int IN_ITEMS = G::IN_ITEMS, IN_ROUNDS = G::IN_ROUNDS, IN_BYTES = G::IN_BYTES;
    constexpr int W_ROUNDS = (W_ITEMS + THREADS - 1) / THREADS;
    constexpr int WRING = RSLOTS * IN_BYTES;             // LDS: [input ring][weight ring][bias table]
    constexpr int BIAS = RSLOTS * (IN_BYTES + W_BYTES);  // 6 x 32 f32: conv1..conv4, conv5 couts 0-31, conv5 couts 32-63
    static_assert(TH == MW, "one tile row per MFMA wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool is_dma = wave_all >= MW;
    const bool is_cmp = !is_dma;
    const int wave = is_cmp ? wave_all : wave_all - MW;
    const int tid = wave * 64 + lane;

    const int tiles_x = (a.w_ + TW - 1) / TW;
    const int tiles_y = (a.h + TH - 1) / TH;
    const int total = tiles_x * tiles_y * a.n;
    int tile;
    {
        const int bid = blockIdx.x, q = total >> 3, r = total & 7, xcd = bid & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int n = tile / (tiles_x * tiles_y);
    const int t2 = tile - n * tiles_x * tiles_y;
    const int ty = t2 / tiles_x, tx = t2 - ty * tiles_x;
    const int y0 = ty * TH, x0 = tx * TW;

    const int j16 = lane & 15, g4 = lane >> 4;
    // (layer, cout group, chunk) of the step after (l, cg, c); l == 5: past the end
    auto advance = [](int& l, int& cg, int& c) {
        const int nc = l == 4 ? 12 : 4 + 2 * l, ncg = l == 4 ? 2 : 1;
        if (++c == nc) { c = 0; if (++cg == ncg) { cg = 0; ++l; } }
    };

    // The two roles run separate loops over the same 52 steps (one s_barrier per step, one more before the first):
    // their register sets never coexist.
    if (is_dma) {
#ifdef NESR_RDB_DMAPRIO
        // (round 2 raised these four waves' priority -- "everything waits for what they issue".  Measured in round 3, on this
        // kernel and on rdb_bf16_strip_kernel: without it the frame is 0.5 % / 1.1 % faster; their scalar instructions otherwise
        // win every issue slot they ask for on the SIMD they share with the MFMA waves.)
        __builtin_amdgcn_s_setprio(NESR_RDB_DMAPRIO);
#endif
        // ---- the plan (once per block) and the neighbours' progress words
        const unsigned lds_base = (unsigned)(size_t)(lds_char*)(smem);
        unsigned voff[IN_ROUNDS];
        unsigned okmask = 0;       // per lane: rounds in which this lane has an in-image item
        unsigned wavemask = 0;     // wave-uniform: rounds in which any lane of the wave has one (= instructions issued)
        const int row0 = y0 > 0 ? y0 - 1 : 0;
        const char* in_img = static_cast<const char*>(a.cur) + ((size_t)n * a.h + row0) * a.w_ * 64;
        const unsigned* watch = nullptr;    // lanes 0..8: progress word of tile (ty + i/3 - 1, tx + i%3 - 1), if it exists
        int kdma = 0;                       // LDS-DMA instructions this wave issues per step (vmcnt counts wave instructions)
        if (lane < 9) {
            const int ny = ty + lane / 3 - 1, nx = tx + lane % 3 - 1;
            if (ny >= 0 && ny < tiles_y && nx >= 0 && nx < tiles_x) watch = a.progress + (n * tiles_y + ny) * tiles_x + nx;
        }
        {
            int p = tid >> 2;
            int py = p / PW, px = p - py * PW;
            const int sl = tid & 3;
#pragma unroll
            for (int i = 0; i < IN_ROUNDS; ++i) {
                const int k = tid + THREADS * i;
                const int sg = sl ^ (((px >> 2) & 1) << 1);
                const int Y = y0 - 1 + py, X = x0 - 1 + px;
                const bool has = k < IN_ITEMS;
                const bool ok = has && Y >= 0 && Y < a.h && X >= 0 && X < a.w_;
                voff[i] = ((unsigned)(Y - row0) * (unsigned)a.w_ + (unsigned)X) * 64u + sg * 16;
                okmask |= ok ? (1u << i) : 0u;
                if (__builtin_amdgcn_ballot_w64(ok) != 0ull) { wavemask |= 1u << i; ++kdma; }
                if (has && !ok) {
#pragma unroll
                    for (int sl2 = 0; sl2 < RSLOTS; ++sl2) *reinterpret_cast<f32x4*>(smem + sl2 * IN_BYTES + k * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                px += (THREADS / 4) % PW;
                py += (THREADS / 4) / PW;
                if (px >= PW) { px -= PW; py += 1; }
            }
#pragma unroll
            for (int j = 0; j < W_ROUNDS; ++j) kdma += (wave * 64 + THREADS * j) < W_ITEMS ? 1 : 0;
        }
        unsigned seen = 0;     // layers of this launch known to be published by all nine tiles
        // an abort word raised earlier in this forward (or by another workgroup): nothing is waited for any more -- the forward's
        // output is invalid either way and the host turns the word into NESR_ERR_HIP (nesr_check_range)
        bool gave_up = __hip_atomic_load(a.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
        // DMAs of one step: weight slab of (layer l, cout group cg, chunk c), then input chunk c -- after the nine tiles
        // have published the layer that produced it (chunks 4.. hold x1..: chunk c belongs to x_((c-4)/2+1)).
        // Exactly kdma wave instructions.
        auto dma_step = [&](int l, int cg, int c, int slot) {
            const int CG = l == 4 ? 2 : 1;
            const char* wsrc = static_cast<const char*>(a.w[l]) + ((size_t)c * CG + cg) * W_BYTES;
#pragma unroll
            for (int j = 0; j < W_ROUNDS; ++j) {
                const int k = tid + THREADS * j;
                const unsigned dst = lds_base + WRING + slot * W_BYTES + j * (THREADS * 16) + wave * 1024;
                if (k < W_ITEMS && !(NESR_RDB_ABL & 64)) glds16_s(wsrc, (unsigned)k * 16u, __builtin_amdgcn_readfirstlane(dst));
            }
            const unsigned need = c < 4 ? 0u : (unsigned)((c - 4) >> 1) + 1u;
            if (need > seen && !gave_up && !(NESR_RDB_ABL & 1)) {
                const unsigned target = a.epoch + need;
                unsigned long long t0 = 0;
                for (unsigned it = 0;; ++it) {
                    const unsigned v = watch ? __hip_atomic_load(watch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : target;
                    if (__builtin_amdgcn_ballot_w64((int)(v - target) < 0) == 0ull) break;
                    if ((it & 63u) == 0u) {
                        // bounded by wall clock: a neighbour that never arrives (workgroups not co-resident: another process's
                        // persistent kernel on the device) ends in an abort word within timeout_ticks, and once the word is up --
                        // here or in any other workgroup -- every later wait of the launch and of the forward is skipped
                        const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                        if (it == 0) t0 = now;
                        const unsigned ab = __hip_atomic_load(a.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (ab != 0u || now - t0 > a.timeout_ticks) {
                            if (ab == 0u && lane == 0) __hip_atomic_store(a.abort_flag, 1u | ((unsigned)c << 8) | ((unsigned)tile << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            gave_up = true;
                            break;
                        }
                    }
                    __builtin_amdgcn_s_sleep(4);
                }
                seen = need;
            }
            const char* isrc = in_img + (long long)c * a.chunk_bytes;
#pragma unroll
            for (int i = 0; i < IN_ROUNDS; ++i) {
                const unsigned dst = lds_base + slot * IN_BYTES + i * (THREADS * 16) + wave * 1024;
                if (((wavemask >> i) & 1u) && !(NESR_RDB_ABL & 32)) {          // wave-uniform: the instruction count is the same every step
                    if ((okmask >> i) & 1u) {
                        if (NESR_RDB_ABL & 2) glds16_s(isrc, voff[i], __builtin_amdgcn_readfirstlane(dst));
                        else glds16_s_sc1(isrc, voff[i], __builtin_amdgcn_readfirstlane(dst));
                    }
                }
            }
        };
        // The six bias vectors of the block sit in LDS (768 B, filled once here): a global load inside the MFMA waves'
        // step loop would cost a vmcnt wait per use.
        if (wave == 0 && lane < 48) {
            const int l = lane >> 3 > 4 ? 4 : lane >> 3, cgq = lane >> 3 > 4 ? 8 + (lane & 7) : (lane & 7);   // lanes 32..47: conv5's 64 biases
            *reinterpret_cast<f32x4*>(smem + BIAS + lane * 16) = *reinterpret_cast<const f32x4*>(a.bias[l] + 4 * cgq);
        }
        int fl = 0, fcg = 0, fc = 0;      // the next step to fetch
        dma_step(0, 0, 0, 0);
        advance(fl, fcg, fc);
        dma_step(fl, fcg, fc, 1);
        advance(fl, fcg, fc);
        dma_step(fl, fcg, fc, 2);
        advance(fl, fcg, fc);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the zero padding is in LDS before the first barrier
        int fill = 3;
        int cl = 0, ccg = 0, cc = 0;      // the current step
        // chunk 0 has landed: the MFMA waves request the launch's first fragments behind this barrier
        wait_vmcnt_le((NESR_RDB_ABL & 96) ? 0 : 2 * kdma);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        for (int step = 0; step < 52; ++step) {
            // this step's DMAs and the next step's have landed -- the MFMA waves read the next step's first fragments
            // before the next barrier -- (those of step + 2, issued one step ago, may stay in flight)
            if (wave == 0) RSTAMP(1, step, 0);
#if NESR_RDB_ABL & 256
            if ((step == 0 || step == 51) && wave == 0 && blockIdx.x == 77 && lane == 0) {      // shader clock = d memtime / d memrealtime x 100 MHz
                const unsigned long long r_ = __builtin_amdgcn_s_memrealtime(), t_ = __builtin_amdgcn_s_memtime();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                g_rdb_stamps[1][step == 0 ? 62 : 63][0] = r_;
                g_rdb_stamps[1][step == 0 ? 62 : 63][1] = t_;
            }
#endif
            wait_vmcnt_le((step >= 50 || (NESR_RDB_ABL & 96)) ? 0 : kdma);
            if (wave == 0) RSTAMP(1, step, 1);
            RARRIVE(MW + wave, step, 0);
            if (!(NESR_RDB_ABL & 512)) __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            RARRIVE(MW + wave, step, 1);
            if (wave == 0) RSTAMP(1, step, 2);
            // layer cl - 1 of this tile is in memory: its MFMA waves stored it during steps 0 .. EPI_STEPS - 1 of layer cl
            // and waited for those stores (vmcnt(0)) before this barrier
            if (cc == EPI_STEPS && ccg == 0 && cl >= 1 && wave == 0 && lane == 0)
                __hip_atomic_store(a.progress + tile, a.epoch + (unsigned)cl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (fl < 5) {      // three steps ahead, into the slot read one step ago
                dma_step(fl, fcg, fc, fill);
                advance(fl, fcg, fc);
                fill = fill == RSLOTS - 1 ? 0 : fill + 1;
            }
            if (wave == 0) RSTAMP(1, step, 3);
            advance(cl, ccg, cc);
        }
        return;
    }

    // ---- MFMA role: operand addresses as in conv3x3_f16x2_kernel, one row per wave
#ifdef NESR_RDB_YPRIO
    if (wave >= 4) __builtin_amdgcn_s_setprio(NESR_RDB_YPRIO);     // the later-dispatched wave of each SIMD loses the issue arbitration by age
#endif
    const bool active = (y0 + wave) < a.h;
    const int un = g4 >> 1, kh = g4 & 1;
    int b16[5][2], a16[5];
#pragma unroll
    for (int st_ = 0; st_ < 5; ++st_) {
        const int dy = st_ < 3 ? un : 2;
        const int dx = st_ < 3 ? st_ : (st_ == 3 ? un : 2);
#pragma unroll
        for (int nh = 0; nh < 2; ++nh) {
            const int col = 16 * nh + j16 + dx;
            b16[st_][nh] = ((wave + dy) * PW + col) * 64 + ((kh ^ (((col >> 2) & 1) << 1)) << 4);
        }
        a16[st_] = ((((dy * 3 + dx) * 2) * 2 + kh) * 32 + j16) * 16;
    }
    int slot = 0;       // ring slot of the current step
#if NESR_RDB_ABL & 256
    int mstep = 0;
#endif
    // Fragment addresses live in registers as absolute LDS addresses of the slot they are read from next; each is used
    // once per step and moved on to the next slot right away (one v_add per register and step; the pixel half and the
    // weight variants are immediate offsets) instead of slot base + offset per read (one v_add per ds_read).
    typedef const __attribute__((address_space(3))) f32x4* lds_f32x4;
    const unsigned lds0 = (unsigned)(size_t)(lds_char*)(smem);
    unsigned bcur[5][2], acur[5], acur4b;
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        bcur[t][0] = lds0 + (unsigned)b16[t][0];
        bcur[t][1] = lds0 + (unsigned)(t == 4 ? (b16[4][0] ^ ((un ^ 1) << 5)) : (b16[t][0] ^ 32));
        acur[t] = lds0 + (unsigned)(WRING + a16[t]);
    }
    acur4b = acur[4] + (unsigned)un * 1024u;
    // ---- the deferred epilogue
    const int par = (NESR_RDB_PAR >= 0) ? NESR_RDB_PAR : wave >> 2;                          // which end of a step this wave's epilogue half sits at
    const int cbl = (g4 & 1) * 16 + (g4 >> 1) * 8;      // a lane's 8 output channels inside a 32-cout group after the permlane16 exchange
    const int piece8 = ((g4 & 1) * 2 + (g4 >> 1)) * 8;  // its 16-byte piece of a 64-byte slot after regroup_pairs (2-byte units)
    const bool res2 = a.res2 != nullptr;
    f32x4 ep[2][2];                                     // [pixel half][cout half]: sums of the finished (layer, cout group)
#pragma unroll
    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) ep[nh][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    int ep_l = -1, ep_cg = 0;
    bool bad = false;
    auto unpack_res = [&](f32x4 rx, f32x4 rx1, f32x4& q0, f32x4& q1) {
        uint4 cx = __builtin_bit_cast(uint4, rx), cx1 = __builtin_bit_cast(uint4, rx1);
        regroup_pairs(cx, cx1);      // -> own hi, own lo
        const f16x8 h = __builtin_bit_cast(f16x8, cx), lo = __builtin_bit_cast(f16x8, cx1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            q0[i] = fmaf((float)lo[i], LO_INV, (float)h[i]);
            q1[i] = fmaf((float)lo[4 + i], LO_INV, (float)h[4 + i]);
        }
    };
    // pixel half nh of (layer l, cout group cg): conv1..4 -> LeakyReLU into cur's channels 64 + 32 l; conv5 ->
    // x5 * s1 + x0 (and * s2 + RRDB input) into `out`'s channels 32 cg.  All 64 lanes (v_permlane16_swap).
    // Addresses: a wave-uniform row base (SGPRs) plus a 32-bit lane offset -- pixel slot and 16-byte piece -- so that
    // the epilogue carries two VGPRs of addressing instead of 64-bit pointers.
    const long long row_bytes = (((long long)n * a.h + (y0 + wave)) * a.w_) * 64;
    unsigned lane_off[2];
    bool lane_ok[2];
#pragma unroll
    for (int nh = 0; nh < 2; ++nh) {
        const int X = x0 + 16 * nh + j16;
        lane_ok[nh] = X < a.w_;
        lane_off[nh] = (unsigned)(lane_ok[nh] ? X : 0) * 64u + (unsigned)piece8 * 2u;
    }
    auto uni = [](const char* p) -> const char* {      // tell the compiler the pointer is wave-uniform
        const unsigned long long v = (unsigned long long)p;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return (const char*)(((unsigned long long)hi << 32) | lo);
    };
    // pixel half nh of (layer l, cout group cg): conv1..4 -> LeakyReLU into cur's channels 64 + 32 l; conv5 ->
    // x5 * s1 + x0 (and * s2 + RRDB input) into `out`'s channels 32 cg.  All 64 lanes (v_permlane16_swap).
    auto epi_half = [&](int l, int cg, const f32x4& e0, const f32x4& e1, int nh) {
#if NESR_RDB_ABL & 256
        if (wave == 1) RSTAMP(0, mstep, 3);
#endif
        const bool valid = lane_ok[nh];
        const unsigned voff = lane_off[nh];
        const char* bsrc = smem + BIAS + (l * 32 + 32 * cg + cbl) * 4;      // conv5's second group follows its first
        f32x4 v0, v1;   // couts cbl .. cbl+3, cbl+4 .. cbl+7
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const auto sw_ = __builtin_amdgcn_permlane16_swap(__float_as_uint(e0[i]), __float_as_uint(e1[i]), false, false);
            v0[i] = __uint_as_float(sw_[0]);
            v1[i] = __uint_as_float(sw_[1]);
        }
        v0 += *reinterpret_cast<const f32x4*>(bsrc);
        v1 += *reinterpret_cast<const f32x4*>(bsrc + 16);
        const int dchunk = l == 4 ? 2 * cg : 4 + 2 * l;
        if (l < 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { v0[i] = fmaxf(v0[i], v0[i] * 0.2f); v1[i] = fmaxf(v1[i], v1[i] * 0.2f); }   // LeakyReLU(0.2), same values as the select form
        } else {
            const char* r1 = uni(static_cast<const char*>(a.cur) + (long long)(2 * cg) * a.chunk_bytes + row_bytes);      // x0's couts 32 cg ..
            const f32x4 r10 = *reinterpret_cast<const f32x4*>(r1 + voff), r11 = *reinterpret_cast<const f32x4*>(r1 + a.chunk_bytes + voff);
            f32x4 r20 = r10, r21 = r11;
            if (res2) {
                const char* r2 = uni(static_cast<const char*>(a.res2) + (long long)(2 * cg) * a.chunk_bytes + row_bytes);
                r20 = *reinterpret_cast<const f32x4*>(r2 + voff);
                r21 = *reinterpret_cast<const f32x4*>(r2 + a.chunk_bytes + voff);
            }
            f32x4 q0, q1;
            unpack_res(r10, r11, q0, q1);
#pragma unroll
            for (int i = 0; i < 4; ++i) { v0[i] = __fadd_rn(__fmul_rn(v0[i], a.s1), q0[i]); v1[i] = __fadd_rn(__fmul_rn(v1[i], a.s1), q1[i]); }
            if (res2) {
                unpack_res(r20, r21, q0, q1);
#pragma unroll
                for (int i = 0; i < 4; ++i) { v0[i] = __fadd_rn(__fmul_rn(v0[i], a.s2), q0[i]); v1[i] = __fadd_rn(__fmul_rn(v1[i], a.s2), q1[i]); }
            }
        }
#if NESR_RDB_ABL & 256
        asm volatile("" :: "v"(v0), "v"(v1));
        if (wave == 1) RSTAMP(0, mstep, 4);
#endif
        uint4 cx, cx1;
        bool bad_here = false;
        split_regroup(v0, v1, cx, cx1, bad_here);
        bad |= bad_here && valid;
#if NESR_RDB_ABL & 256
        asm volatile("" :: "v"(__builtin_bit_cast(f32x4, cx)), "v"(__builtin_bit_cast(f32x4, cx1)));
        if (wave == 1) RSTAMP(0, mstep, 5);
#endif
        if (valid) {
            const char* d0 = uni(static_cast<const char*>(l == 4 ? a.out : a.cur) + (long long)dchunk * a.chunk_bytes + row_bytes);
            if (l < 4 && !(NESR_RDB_ABL & 4)) {
                store16_wt_s(d0, voff, cx);
                store16_wt_s(d0 + a.chunk_bytes, voff, cx1);
            } else {
                *reinterpret_cast<uint4*>(const_cast<char*>(d0) + voff) = cx;
                *reinterpret_cast<uint4*>(const_cast<char*>(d0) + a.chunk_bytes + voff) = cx1;
            }
        }
#if NESR_RDB_ABL & 256
        if (wave == 1) RSTAMP(0, mstep, 6);
#endif
    };
    // the part of the pending epilogue that belongs to step c of the running layer
    auto epi_part = [&](int c) {
        if (NESR_RDB_ABL & 16) return;
        if (EPI_STEPS == 1) {
            epi_half(ep_l, ep_cg, ep[0][0], ep[0][1], 0);
            epi_half(ep_l, ep_cg, ep[1][0], ep[1][1], 1);
        } else if (c == 0) {
            epi_half(ep_l, ep_cg, ep[0][0], ep[0][1], 0);
        } else {
            epi_half(ep_l, ep_cg, ep[1][0], ep[1][1], 1);
        }
    };
    f32x4 acc16[2][2][2];       // [pixel half][cout half][main | cross]
    f32x4 Af[2][2][2];          // [buffer][cout half][variant]
    f32x4 Bf[2][2][2];          // [buffer][pixel half][variant]
    // fragments of tap-step s_ of the step in ring slot `sl` -> buffer `buf`; the addresses move on to slot sl + 1
    // (the second pixel half is 16 pixels = 1024 bytes on: the slot swizzle looks at bit 2 of the column only)
    auto load_step = [&](int sl, int s_, int buf) {
        const bool wrap = sl == RSLOTS - 1;
        const unsigned dB = wrap ? (unsigned)(-(RSLOTS - 1) * IN_BYTES) : (unsigned)IN_BYTES;
        const unsigned dA = wrap ? (unsigned)(-(RSLOTS - 1) * W_BYTES) : (unsigned)W_BYTES;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            if (s_ < 4) {
                Af[buf][mt][0] = *((lds_f32x4)(size_t)(acur[s_] + mt * 256));
                Af[buf][mt][1] = *((lds_f32x4)(size_t)(acur[s_] + mt * 256 + 1024));
            } else {
                f32x4 hi = *((lds_f32x4)(size_t)(acur[4] + mt * 256));
                Af[buf][mt][1] = *((lds_f32x4)(size_t)(acur4b + mt * 256));
                if (un) hi = f32x4{0.f, 0.f, 0.f, 0.f};
                Af[buf][mt][0] = hi;
            }
        }
#pragma unroll
        for (int nh = 0; nh < 2; ++nh) {
            Bf[buf][nh][0] = *((lds_f32x4)(size_t)(bcur[s_][0] + nh * 1024));
            Bf[buf][nh][1] = *((lds_f32x4)(size_t)(bcur[s_][1] + nh * 1024));
        }
        acur[s_] += dA;
        if (s_ == 4) acur4b += dA;
        bcur[s_][0] += dB;
        bcur[s_][1] += dB;
    };
    // One step = 5 tap-steps of the chunk in ring slot `slot`; P = buffer parity of its first tap-step (5 is odd: it
    // flips every step, and every layer has an even number of steps).  The fragments of tap-step 0 were requested in
    // the previous step (or just below, for the launch's first step); those of the next step's tap-step 0 are requested
    // beside this step's last MFMAs -- the next step's chunk landed before this step's barrier -- so that no wave sits
    // behind an LDS round trip after a barrier.
    auto step_body = [&](auto Pc, int c, bool last_of_all) {
        constexpr int P = decltype(Pc)::value;
        // the x_l stores of this wave have retired before the barrier after which the progress word goes out
        if (c == EPI_STEPS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if NESR_RDB_ABL & 256
        if (wave == 1) RSTAMP(0, mstep, 0);
        RARRIVE(wave, mstep, 0);
#endif
        if (!(NESR_RDB_ABL & 512)) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#if NESR_RDB_ABL & 256
        RARRIVE(wave, mstep, 1);
        if (wave == 1) RSTAMP(0, mstep, 1);
#endif
        const bool epi_now = active && ep_l >= 0 && c < EPI_STEPS;
        const int nslot = slot == RSLOTS - 1 ? 0 : slot + 1;
        if (active && !(NESR_RDB_ABL & 8)) {
            if (epi_now && par == 0) {
                epi_part(c);
                __builtin_amdgcn_sched_barrier(0);
            }
#if NESR_RDB_ABL & 256
            unsigned long long tt[6];
#endif
#pragma unroll
            for (int s_ = 0; s_ < 5; ++s_) {
                const int buf = (s_ + P) & 1;
#if NESR_RDB_ABL & 256
                tt[s_] = __builtin_amdgcn_s_memtime();      // no wait here: read at the end of the step
                __builtin_amdgcn_sched_barrier(0);
#endif
                if (s_ + 1 < 5) load_step(slot, s_ + 1, buf ^ 1);
                else if (!last_of_all) load_step(nslot, 0, buf ^ 1);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const f16x8 a0 = __builtin_bit_cast(f16x8, Af[buf][mt][0]), a1 = __builtin_bit_cast(f16x8, Af[buf][mt][1]);
#pragma unroll
                    for (int nh = 0; nh < 2; ++nh) {
                        const f16x8 x0_ = __builtin_bit_cast(f16x8, Bf[buf][nh][0]), x1_ = __builtin_bit_cast(f16x8, Bf[buf][nh][1]);
                        if (s_ < 4) {
                            acc16[nh][mt][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, x1_, acc16[nh][mt][1], 0, 0, 0);
                            acc16[nh][mt][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, x0_, acc16[nh][mt][0], 0, 0, 0);
                            acc16[nh][mt][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, x0_, acc16[nh][mt][1], 0, 0, 0);
                        } else {
                            acc16[nh][mt][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, x0_, acc16[nh][mt][0], 0, 0, 0);
                            acc16[nh][mt][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, x1_, acc16[nh][mt][1], 0, 0, 0);
                        }
                    }
                }
                // the next tap-step's 8 fragment reads ride between this one's MFMAs
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (epi_now && par == 1) {
                epi_part(c);
                __builtin_amdgcn_sched_barrier(0);
            }
#if NESR_RDB_ABL & 256
            tt[5] = __builtin_amdgcn_s_memtime();
            if (blockIdx.x == 77 && (wave == 1 || wave == 5) && lane == 0 && mstep >= 20 && mstep < 28) {
#pragma unroll
                for (int i = 0; i < 6; ++i) g_rdb_taps[wave == 5][mstep - 20][i] = tt[i];
            }
#endif
        }
        slot = nslot;
#if NESR_RDB_ABL & 256
        if (wave == 1) RSTAMP(0, mstep, 2);
        ++mstep;
#endif
    };
    // the launch's first fragments: chunk 0 is in LDS once every DMA wave has passed its first wait -- one extra barrier
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (active) load_step(0, 0, 0);
    for (int l = 0; l < 5; ++l) {
        const int nc = l == 4 ? 12 : 4 + 2 * l;
        const int ncg = l == 4 ? 2 : 1;
        for (int cg = 0; cg < ncg; ++cg) {
#pragma unroll
            for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int q = 0; q < 2; ++q) acc16[nh][mt][q] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int c = 0; c < nc; c += 2) {
                step_body(std::integral_constant<int, 0>{}, c, false);
                step_body(std::integral_constant<int, 1>{}, c + 1, l == 4 && cg == 1 && c + 2 == nc);
            }
            // the layer's sums (main + cross / 2^11) wait for the next layer's first steps
#pragma unroll
            for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) ep[nh][mt][i] = fmaf(acc16[nh][mt][1][i], LO_INV, acc16[nh][mt][0][i]);
            ep_l = l;
            ep_cg = cg;
        }
    }
    // the last cout group of conv5: nothing is left to overlap it with
    if (active && !(NESR_RDB_ABL & 16)) {
        epi_half(4, 1, ep[0][0], ep[0][1], 0);
        epi_half(4, 1, ep[1][0], ep[1][1], 1);
    }
    if (bad && a.status) __hip_atomic_store(a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int DMAW>
hipError_t launch_split(const ConvArgs& a, hipStream_t s) {
    typedef Geo<DMAW> G;
    constexpr size_t shm = (size_t)ISLOTS * G::IN_BYTES + 2 * (size_t)W_BYTES;
    static unsigned long long attr_done = 0;
    {
        const hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&conv3x3_f16x2_kernel<DMAW>), shm, attr_done);
        if (e != hipSuccess) return e;
    }
    const int rows = (a.y_hi > 0 ? a.y_hi : a.h) - a.y_lo;
    const long total = (long)((a.w_ + TW - 1) / TW) * ((rows + G::TH - 1) / G::TH) * a.n * (a.coutp / 32);
    if (total <= 0) return hipSuccess;
    if (total > 0x7fffffffL) return hipErrorInvalidValue;
    hipLaunchKernelGGL((conv3x3_f16x2_kernel<DMAW>), dim3((unsigned)total), dim3(G::LAUNCH_THREADS), shm, s, a);
    return hipGetLastError();
}

inline uint16_t f2h(float f) {
    const _Float16 h = (_Float16)f;
    return __builtin_bit_cast(uint16_t, h);
}
inline float h2f(uint16_t u) { return (float)__builtin_bit_cast(_Float16, u); }

}  // namespace

#if NESR_RDB_ABL & 256
extern "C" int nesr_debug_rdb_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rdb_stamps), sizeof(unsigned long long) * 2 * 64 * 8);
}
extern "C" int nesr_debug_rdb_taps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rdb_taps), sizeof(unsigned long long) * 2 * 8 * 8);
}
extern "C" int nesr_debug_rdb_arrive(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rdb_arrive), sizeof(unsigned long long) * 12 * 8 * 2);
}
#endif
#if NESR_ABL & 64
extern "C" int nesr_debug_stamps(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * (n < 256 ? n : 256));
}
#endif

size_t packed_weight_elems_f16x2(int cin_p, int coutp) { return (size_t)cin_p * 9 * coutp * 2; }

// OIHW f32 -> [chunk = ci/16][cout group = o/32][tap][plane hi|lo][k half = (ci%16)/8][o%32][ci%8] halves
void pack_weights_f16x2(const float* oihw, int cout, int cin, int cin_p, int coutp, uint16_t* dst) {
    const size_t total = packed_weight_elems_f16x2(cin_p, coutp);
    for (size_t i = 0; i < total; ++i) dst[i] = 0;
    const int groups = coutp / 32;
    for (int o = 0; o < cout; ++o)
        for (int ci = 0; ci < cin; ++ci)
            for (int tap = 0; tap < 9; ++tap) {
                // |w| <= 65504 and finite: nesr_finalize_weights rejects anything else (NESR_ERR_RANGE)
                const float wv = oihw[((size_t)o * cin + ci) * 9 + tap];
                const uint16_t hi = f2h(wv);
                const uint16_t lo = f2h((wv - h2f(hi)) * LO_SCALE);
                const int c = ci / 16, kh = (ci % 16) / 8, kk = ci % 8;
                const size_t slab = ((size_t)c * groups + o / 32) * (size_t)(W_ITEMS * 8);
                const size_t ihi = slab + ((((size_t)tap * 2 + 0) * 2 + kh) * 32 + o % 32) * 8 + kk;
                const size_t ilo = slab + ((((size_t)tap * 2 + 1) * 2 + kh) * 32 + o % 32) * 8 + kk;
                dst[ihi] = hi;
                dst[ilo] = lo;
            }
}

int rdb_f16x2_tiles(int n, int h, int w) { return ((w + TW - 1) / TW) * ((h + 7) / 8) * n; }

hipError_t launch_rdb_f16x2(const RdbLaunch& r, hipStream_t s) {
    typedef Geo<4> G;
    constexpr size_t shm = (size_t)RSLOTS * (G::IN_BYTES + (size_t)W_BYTES) + 768;
    static unsigned long long attr_done = 0;
    {
        const hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&rdb_f16x2_kernel), shm, attr_done);
        if (e != hipSuccess) return e;
    }
    if ((long long)r.w_ * 64 * 12 >= (1ll << 32)) return hipErrorInvalidValue;
    RdbArgs a;
    a.cur = r.cur; a.chunk_bytes = r.chunk_bytes; a.out = r.out; a.res2 = r.res2; a.s1 = r.s1; a.s2 = r.s2;
    for (int i = 0; i < 5; ++i) { a.w[i] = r.w[i]; a.bias[i] = r.bias[i]; }
    a.n = r.n; a.h = r.h; a.w_ = r.w_;
    a.progress = r.progress; a.epoch = r.epoch; a.abort_flag = r.abort_flag; a.status = r.status;
    a.timeout_ticks = r.timeout_ticks ? r.timeout_ticks : 20000000ull;
    int total = rdb_f16x2_tiles(r.n, r.h, r.w_);
    if (total <= 0) return hipSuccess;
    // every tile needs its own RESIDENT workgroup: what the device admits of this kernel (registers, LDS) times its compute units
    static int resident = -1;
    if (resident < 0) {
        int per_cu = 0, dev = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rdb_f16x2_kernel, 64 * (MW + DW), shm) != hipSuccess || hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            return hipErrorUnknown;
        resident = per_cu * cus;
    }
    if (total > resident) return hipErrorLaunchOutOfResources;
    total -= r.debug_drop;       // test hook (nesr_debug_fault): the last workgroups never start, their neighbours' waits must end in the abort word
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(rdb_f16x2_kernel, dim3((unsigned)total), dim3(64 * (MW + DW)), shm, s, a);
    return hipGetLastError();
}

hipError_t launch_conv3x3_f16x2(const ConvArgs& a, hipStream_t s) {
    if (a.cin % 16 || (a.coutp != 32 && a.coutp != 64)) return hipErrorInvalidValue;
    if (a.y_lo < 0 || (a.y_hi > 0 && (a.y_hi > a.h || a.y_hi <= a.y_lo)) || (a.y_hi == 0 && a.y_lo != 0)) return hipErrorInvalidValue;
    if ((long long)a.in_w * 64 * 12 >= (1ll << 32)) return hipErrorInvalidValue;   // 32-bit byte offsets inside one tile's rows
    if (a.in_map.pix != 32 || (a.out && (a.out_map.pix % 32 || a.out_coff % 16))) return hipErrorInvalidValue;
    if ((a.out_nchw || a.out_u8) && (a.coutp != 32 || a.cout_real < 1 || a.cout_real > 4)) return hipErrorInvalidValue;
    // 8x32-px tiles.  Launches that give a CU at most one workgroup (a 512x512 frame's 32-channel layers): four
    // extra waves issue the DMAs, so the MFMA waves never stall on LDS-DMA issue (~85 cycles each, 10 per chunk
    // and wave; in-kernel stamps: -15 % per K-chunk).  With two workgroups per CU the other workgroup already
    // fills those gaps and the extra waves only cost occupancy (measured +21 % on 6 tiles of 532x532).
    static const int cus = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
        return n;
    }();
    static const int dmaw = [] { const char* e = getenv("NESR_SPLIT_DMAW"); return e ? atoi(e) : -1; }();
    const long t1 = (long)((a.w_ + TW - 1) / TW) * (((a.y_hi > 0 ? a.y_hi : a.h) - a.y_lo + 7) / 8) * a.n * (a.coutp >> 5);
    // ... and only when this context has the device to itself: with frames in flight on other streams the 8-wave
    // workgroups (166 registers) keep a second kernel's workgroups off the CU (2 frames in flight: 158 -> 166 MP/s
    // without them; one frame alone: 134 -> 138 MP/s with them)
    const bool producer = dmaw >= 0 ? dmaw > 0 : (t1 <= cus && !a.shared_device);
    return producer ? launch_split<4>(a, s) : launch_split<0>(a, s);
}

}  // namespace nesr
