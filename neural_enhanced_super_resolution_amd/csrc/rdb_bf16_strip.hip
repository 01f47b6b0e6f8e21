// bf16 residual dense block with the whole 192-channel working set of a workgroup resident in LDS
// (basicsr ResidualDenseBlock.forward: x1..x4 = lrelu(conv_k(cat(x0..x_{k-1}))), x5 = conv5(cat) * 0.2 + x0 [* 0.2 + RRDB
// input]; reached through RealESRGANer.tile_process from standalone/direct_esrgan.py:118-127,148 -- BASELINE.json
// configs[2], [3]: the 4K frame as 40 tiles of <= 532 x 532).
//
// The per-layer bf16 kernels (conv3x3_bf16.hip) re-read every earlier map of the block for every layer: 40 chunk-tiles
// of LDS-DMA per block where 12 are distinct, ~325 GB per 4K frame, matrix pipe 29-44 % busy (profiles/r02).  Here a
// workgroup owns a STRIP of an image -- 16 columns, all rows -- and sweeps it top to bottom in positions of 12 rows,
// conv1..conv5 of the block at every position, layer m lagging m-1 rows behind layer 1:
//
//     position i, layer m (1..5) computes rows [12 i - (m-1), 12 i - (m-1) + 12)
//
// so that every layer finds the rows it needs (one above, one below its own) already computed by the layer before
// it, and nothing is ever recomputed.  x0..x4 live in LDS as circular row windows of 18 / 17 / 16 / 15 / 14 rows
// (x_l is read by layers l+1..5 over rows [12 i - 5, 12 i - l + 12]); 18 pixels wide: the strip plus one halo column a
// side.  x0 rows stream in from global memory by LDS-DMA one position ahead (halo columns included: x0 is a kernel
// input); x1..x4 never leave the CU except for the strip's two edge columns, which the neighbouring strips need as
// their halos: 12 rows x 64 B per layer and side, handed over through global memory as 8-byte {data, tag} granules
// (MI355X_MICROARCH.md, handoff-1to1: one sc1 16-byte store = two granules, the consumer polls the payload itself).
// Global traffic per position (192 pixels): 27 KB of x0 in, 24 KB of x5 out, 12 KB of edge columns -- against 590 KB
// of LDS-DMA for the same pixels in the per-layer form.
//
//   workgroup : 4 MFMA waves (3 rows x 16 pixels x 32 couts each, v_mfma_f32_16x16x32_bf16: K = 32 = one PAIR of
//               16-channel chunks) + 4 DMA waves (weight stream, x0 rows, halo import); one per CU (153 KB of LDS)
//   step      : one (layer, cout group, chunk pair): 3 horizontal taps x 18 MFMAs; 26 steps per position
//               (2 + 3 + 4 + 5 + 2 x 6; conv5's two cout groups alternate so that x0 is finished with after its first
//               four steps and the next position's rows can stream in under the other eight)
//   weights   : the block's weights are ONE stream of 78 slots of 6 KB in consumption order (step, tap column), the same
//               for every position; a 7-slot LDS ring, the DMA waves one step ahead, one s_barrier per step
//   operands  : per tap column a wave reads 5 pixel-row fragments (each used for up to 3 vertical taps) and 6 weight
//               fragments for 18 MFMAs; the next column's (or next step's) fragments are requested beside the MFMAs
//   schedule  : the strips of an image run in lockstep on as many workgroups; which workgroup runs which strips, in
//               which order, is a host-side packing (strip_schedule) -- every workgroup walks its list in one global
//               image order, so waits cannot form a cycle; grid <= compute units, every workgroup resident
#include <hip/hip_bf16.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <random>
#include <type_traits>
#include <vector>

#include "nesr_kernels.h"

namespace nesr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char lds_char;
typedef const __attribute__((address_space(3))) f32x4* lds_f32x4;

namespace {

constexpr int BH = STRIP_BH, BW = STRIP_BW, PWS = BW + 2;
constexpr int CHB = PWS * 32;                    // bytes of one 16-channel chunk of one padded row
constexpr int ROWB0 = 4 * CHB, ROWB1 = 2 * CHB;  // x0 rows hold 4 chunks, x1..x4 rows 2
constexpr int OFF1 = 18 * ROWB0;                 // x0: 18 rows
constexpr int ACT = OFF1 + (17 + 16 + 15 + 14) * ROWB1;   // 112,896
constexpr int WSLOT = 6144, NWS = 7;
constexpr int NSTEP = 26, WPER = 3 * NSTEP;      // weight slots per position
constexpr int WRING = ACT;
constexpr int BIASO = WRING + NWS * WSLOT;       // 192 f32: conv1..conv4 (32 each), conv5 (64)
constexpr int FLAGO = BIASO + 768;               // one word: abort seen at entry
constexpr int SCRO = FLAGO + 16;                // 256 B nobody reads: landing zone of the L2 prefetch touches
constexpr int LDSB = SCRO + 256;
constexpr int MW = 4, DW = 4;
static_assert(LDSB <= 160 * 1024, "LDS");
constexpr int XCH_LAYER = BH * 128;              // bytes of one (layer, side, parity) mailbox: 12 rows x 8 granule pairs
constexpr int XCH_STRIP = 2 * 2 * 4 * XCH_LAYER; // per strip: [side][position parity][layer 1..4]
static_assert(XCH_STRIP == STRIP_XCH_BYTES, "nesr_kernels.h");

// Window parameters of chunk pair p (0, 1: the two halves of x0; 2..5: x1..x4), branch-free from packed tables:
// offset / 1152, rows of the circular window, bytes between rows.
struct MapP { int off, rowb, R; };
__device__ __forceinline__ MapP pair_map(int p) {
    MapP m;
    m.off = (int)((0x5445352401ull * 256ull >> (8 * p)) & 255ull) * ROWB1;       // {0, 1, 36, 53, 69, 84} x 1152
    m.R = (int)((0x0e0f10111212ull >> (8 * p)) & 255ull);                        // {18, 18, 17, 16, 15, 14}
    m.rowb = p < 2 ? ROWB0 : ROWB1;
    return m;
}
__device__ __forceinline__ MapP map_of(int l) { return pair_map(l == 0 ? 0 : l + 1); }   // map l = x_l
// the five window bases (slot of row 12 pos - 5 of x0..x4) packed 8 bits each
__device__ __forceinline__ int base_of(unsigned long long wbp, int l) { return (int)((wbp >> (8 * l)) & 255ull); }
__device__ __forceinline__ unsigned long long advance_bases(unsigned long long wbp) {
    unsigned long long out = 0;
#pragma unroll
    for (int l = 0; l < 5; ++l) {
        int b = (int)((wbp >> (8 * l)) & 255ull) + BH;
        b -= b >= 18 - l ? 18 - l : 0;
        out |= (unsigned long long)b << (8 * l);
    }
    return out;
}
constexpr unsigned long long WB_INIT = 13ull | (12ull << 8) | (11ull << 16) | (10ull << 24) | (9ull << 32);   // (18 - l) - 5

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
// six consecutive 1-KiB pieces (one 6-KiB weight slot) by ONE wave: the instruction's immediate offset moves the global and the
// LDS address alike, so M0 and the scalar base are set up twice per slot instead of once per piece (the scalar address
// arithmetic of six separate pieces was ~100 of the ~125 cycles a piece cost the issuing wave)
__device__ __forceinline__ void glds16_slot6_s(const char* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    const char* sbase2 = sbase + 4096;
    const unsigned lds2 = lds_dst + 4096;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
        "global_load_lds_dwordx4 %1, %2 offset:2048\n\t"
        "global_load_lds_dwordx4 %1, %2 offset:3072\n\t"
        "s_mov_b32 m0, %5\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %4\n\t"
        "global_load_lds_dwordx4 %1, %4 offset:1024\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_dst), "s"(sbase2), "s"(lds2)
        : "memory");
}
// one row of the x0 window = three pieces with per-piece lane masks (columns outside the image, lanes past the row's 2304
// bytes): M0 set up once, the pieces' LDS offsets as immediates (the same immediate moves the global address: taken out of
// the scalar bases), EXEC narrowed per piece
__device__ __forceinline__ void glds16_row3_s(const char* sbase, unsigned v0, unsigned v1, unsigned v2, unsigned lds_dst,
                                              unsigned long long m0_, unsigned long long m1_, unsigned long long m2_) {
    unsigned keep;
    unsigned long long save;
    const char* sb1 = sbase - 1024;
    const char* sb2 = sbase - 2048;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b64 %1, exec\n\t"
        "s_mov_b32 m0, %6\n\t"
        "s_and_b64 exec, %1, %9\n\t"
        "global_load_lds_dwordx4 %2, %5\n\t"
        "s_and_b64 exec, %1, %10\n\t"
        "global_load_lds_dwordx4 %3, %7 offset:1024\n\t"
        "s_and_b64 exec, %1, %11\n\t"
        "global_load_lds_dwordx4 %4, %8 offset:2048\n\t"
        "s_mov_b64 exec, %1\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep), "=&s"(save)
        : "v"(v0), "v"(v1), "v"(v2), "s"(sbase), "s"(lds_dst), "s"(sb1), "s"(sb2), "s"(m0_), "s"(m1_), "s"(m2_)
        : "memory");
}
__device__ __forceinline__ void glds4_s(const char* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dword %1, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_dst)
        : "memory");
}
// 16-byte write-through store / L1-bypassing load of the edge-column mailboxes
__device__ __forceinline__ void store16_sc1(char* p, uint4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(__builtin_bit_cast(f32x4, v)) : "memory");
}
__device__ __forceinline__ uint4 load16_sc1_wait(const char* p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return __builtin_bit_cast(uint4, v);
}
__device__ __forceinline__ uint2 pack4_bf16(f32x4 v) {   // plain casts -> v_cvt_pk_bf16_f32 (RNE, NaN preserving)
    const bf16x4 b = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    return __builtin_bit_cast(uint2, b);
}
__device__ __forceinline__ f32x4 unpack4_bf16(uint2 u) {
    return f32x4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u)};
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
// max(v, 0.2 v), exact; v_max_f32 by hand: fmaxf() first canonicalises its operands (one more VALU instruction per value,
// in an epilogue that no MFMA covers)
__device__ __forceinline__ f32x4 lrelu4(f32x4 v) {
    const f32x2 lo = f32x2{v[0], v[1]} * 0.2f, hi = f32x2{v[2], v[3]} * 0.2f;
    f32x4 o;
    asm("v_max_f32 %0, %1, %2" : "=v"(o[0]) : "v"(v[0]), "v"(lo[0]));
    asm("v_max_f32 %0, %1, %2" : "=v"(o[1]) : "v"(v[1]), "v"(lo[1]));
    asm("v_max_f32 %0, %1, %2" : "=v"(o[2]) : "v"(v[2]), "v"(hi[0]));
    asm("v_max_f32 %0, %1, %2" : "=v"(o[3]) : "v"(v[3]), "v"(hi[1]));
    return o;
}

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for_n(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }
template <class F>
__device__ __forceinline__ void static_for3(F&& f) {
    f(std::integral_constant<int, 0>{});
    f(std::integral_constant<int, 1>{});
    f(std::integral_constant<int, 2>{});
}

struct StripArgs {
    const char* cur;          // the block's buffer: x0 = its chunks 0..3, channel-blocked bf16 [chunk][n*H*W pixels][16]
    long long chunk_bytes;
    char* out;                // conv5's destination (chunks 0..3 of the next block's buffer)
    const char* res2;         // RRDB input for the second residual, or null
    float s1, s2;
    const char* wstream;      // [78][6144 B]: the block's weights in consumption order (pack_strip_weights)
    const float* bias;        // [192]
    int H, W;                 // slot geometry: image n starts at pixel n*H*W, rows are W pixels apart
    const int4* items;        // {image, strip, image height, image width}
    const int* wg_first;      // [grid + 1]: a workgroup's items
    char* xch;                // edge-column mailboxes, XCH_STRIP bytes per (image, strip)
    int smax;                 // strips per image slot (mailbox index = image * smax + strip)
    unsigned epoch;           // tags of this launch are epoch + position * 8 + layer
    unsigned* abort_flag;
    unsigned long long timeout_ticks;   // s_memrealtime ticks (100 MHz) a halo wait may take
};

// wave priorities (s_setprio): measured on the 4K frame, the DMA waves at 3 (as in the f32 fused kernel) cost 1.1 % against 0 or 1 --
// their scalar instructions then win every issue slot they ask for on the SIMD they share with an MFMA wave
#ifndef NESR_STRIP_DMA_PRIO
#define NESR_STRIP_DMA_PRIO 0
#endif
#ifndef NESR_STRIP_MFMA_PRIO
#define NESR_STRIP_MFMA_PRIO 0
#endif
#ifndef NESR_STRIP_ABL
#define NESR_STRIP_ABL 0   // timing ablations (WRONG results): 1 no halo waits, 2 no MFMA, 4 no weight DMA, 8 no epilogues
#endif

#if NESR_STRIP_ABL & 256
__device__ unsigned long long g_strip_stamps[8][32][8];      // [wave][step of position 5 of the first item][event] of workgroup 77
#define SSTAMP(w, step, ev) do { if (blockIdx.x == 77 && lane == 0 && stamp_on) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); g_strip_stamps[w][step][ev] = t_; } } while (0)
#else
#define SSTAMP(w, step, ev) do { } while (0)
#endif

__global__ __launch_bounds__(64 * (MW + DW), 2) void rdb_bf16_strip_kernel(StripArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool is_dma = wave_all >= MW;
    const int wv = is_dma ? wave_all - MW : wave_all;
    const unsigned lds0 = (unsigned)(size_t)(lds_char*)(smem);
    const int it0 = a.wg_first[blockIdx.x], it1 = a.wg_first[blockIdx.x + 1];

    // an abort raised by an earlier launch of this forward (or by another workgroup a moment ago): everybody of this
    // workgroup sees the same answer and leaves before the first barrier-counted section
    if (threadIdx.x == 0) *reinterpret_cast<unsigned*>(smem + FLAGO) = __hip_atomic_load(a.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (*reinterpret_cast<volatile unsigned*>(smem + FLAGO) != 0u) return;

    if (is_dma) {
        // ======================================================= DMA role.  Waves 0..2: the weight stream (six 1-KiB pieces each per
        // step); waves 0 / 1 also import the left / right halo column; wave 3: the x0 rows.
        if (NESR_STRIP_DMA_PRIO) __builtin_amdgcn_s_setprio(NESR_STRIP_DMA_PRIO);
        const int d = wv;
        int wq7 = 4, wq78 = 4;     // weight stream position of q = 3 g + 4 (the first slot fetched after the barrier of step g)
        auto weight_task = [&](int q7, int q78, int kb) {
            if (NESR_STRIP_ABL & 4) return;
            glds16_s(a.wstream + (size_t)q78 * WSLOT + kb * 1024, (unsigned)lane * 16u,
                     __builtin_amdgcn_readfirstlane(lds0 + WRING + q7 * WSLOT + kb * 1024));
        };
        if (d < 3) {   // the ring's first four slots, once per workgroup
            for (int t = d; t < 24; t += 3) weight_task(t / 6, t / 6, t % 6);
        } else if (lane < 48) {
            *reinterpret_cast<f32x4*>(smem + BIASO + lane * 16) = *reinterpret_cast<const f32x4*>(a.bias + 4 * lane);
        }
        bool aborted = false;
        for (int it = it0; it < it1; ++it) {
            const int4 item = a.items[2 * it], seg = a.items[2 * it + 1];
            const int img = item.x & 0xffff, s = item.y, h = item.z, w = item.w;
            const int pos0 = seg.x, pos1 = seg.y;       // positions of this item (a row segment of the strip: strip_schedule)
            const int xs = s * BW, ns = (w + BW - 1) / BW;
            const int slot_id = (item.x >> 16) * a.smax + s;      // mailboxes belong to the (image, segment)
            const bool has_nb = d == 0 ? s > 0 : (d == 1 ? s + 1 < ns : false);
            // x0 loader plan (wave 3): item k = 64 i + lane of a row = chunk k / 36, padded pixel (k % 36) / 2, half k & 1
            unsigned xoff[3];
            unsigned xok = 0;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int k = 64 * i + lane;
                const int c = k / 36, rem = k - 36 * c, p = rem >> 1, hf = rem & 1;
                const int x = xs - 1 + p;
                const bool ok = k < 144 && x >= 0 && x < w;
                xoff[i] = (unsigned)c * (unsigned)a.chunk_bytes + (unsigned)(ok ? x : 0) * 32u + hf * 16;
                xok |= ok ? (1u << i) : 0u;
            }
            const char* img_base = a.cur + (size_t)img * a.H * a.W * 32;
            const unsigned long long xmask0 = __builtin_amdgcn_ballot_w64(xok & 1u), xmask1 = __builtin_amdgcn_ballot_w64((xok >> 1) & 1u),
                                     xmask2 = __builtin_amdgcn_ballot_w64((xok >> 2) & 1u);
            auto x0_row = [&](int y, int slot) {      // image row y -> row slot `slot` of the x0 window (three pieces)
                if (y < h) {
                    glds16_row3_s(img_base + (size_t)y * a.W * 32, xoff[0], xoff[1], xoff[2], __builtin_amdgcn_readfirstlane(lds0 + slot * ROWB0), xmask0, xmask1, xmask2);
                } else {
#pragma unroll
                    for (int i = 0; i < 3; ++i)
                        if (64 * i + lane < 144) *reinterpret_cast<f32x4*>(smem + slot * ROWB0 + i * 1024 + lane * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            };
            // ---- the windows start empty (zeros are the padding above the image and left / right of it)
            for (int k = threadIdx.x; k < ACT / 16; k += 64 * (MW + DW)) *reinterpret_cast<f32x4*>(smem + k * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            unsigned long long wbp = WB_INIT;
            for (int k = 0; k < pos0; ++k) wbp = advance_bases(wbp);
            if (d == 3)      // the x0 rows of the first position: 12 pos0 - 5 .. 12 pos0 + 12 (at position 0: rows 0..12 in slots 0..12)
                for (int k = 0; k < 18; ++k) {
                    const int y = BH * pos0 - 5 + k;
                    int slot = base_of(wbp, 0) + k;
                    slot -= slot >= 18 ? 18 : 0;
                    if (y >= 0 && y < h) x0_row(y, slot);
                }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            for (int pos = pos0; pos < pos1; ++pos) {
                int imp_m = 0;          // layer whose edge columns are being imported
                bool imp_done = true, pending = false;
                unsigned long long gr[3] = {0ull, 0ull, 0ull};     // the poll in flight: granules lane, 64 + lane, 128 + lane of the mailbox
#if NESR_STRIP_ABL & 256
                const bool stamp_on = it == it0 && pos == 5;
#endif
                for (int idx = 0; idx < NSTEP; ++idx) {
                    SSTAMP(MW + d, idx, 0);
                    // everything of this wave but the poll issued last step (its three loads are this wave's youngest) has
                    // landed; wave 3's x0 rows are only needed at the next position's first barrier
                    if (d == 3) { if (idx == 0 || idx == NSTEP - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                    else if (pending) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    SSTAMP(MW + d, idx, 1);
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    SSTAMP(MW + d, idx, 2);
                    // halo import: the neighbour's edge column of x_m, published as {data, tag} granules after its layer m at this
                    // position.  A poll is three 8-byte loads per lane (192 granules); two go out per step -- one right behind the
                    // barrier, looked at after the weight DMAs have been issued, one then, looked at behind the next barrier --
                    // and nothing waits for one inside a step until the step before the first one that reads x_m's halo.
                    const int m_now = idx < 2 ? 0 : idx < 5 ? 1 : idx < 9 ? 2 : idx < 14 ? 3 : 4;
                    if (m_now != imp_m) { imp_m = m_now; imp_done = !has_nb; pending = false; }
                    const bool polling = d < 2 && imp_m > 0 && !imp_done && !(NESR_STRIP_ABL & 1);
                    const unsigned tag = a.epoch + (unsigned)pos * 8u + (unsigned)imp_m;
                    const unsigned long long* src = reinterpret_cast<const unsigned long long*>(
                        a.xch + ((((size_t)(slot_id + (d ? 1 : -1)) * 2 + (1 - d)) * 2 + (pos & 1)) * 4 + (imp_m > 0 ? imp_m - 1 : 0)) * XCH_LAYER) + lane;
                    // The three loads of a poll are hand-written (`global_load_dwordx2 ... sc1` = the relaxed agent-scope load) and waited
                    // for by COUNT: the compiler would put `s_waitcnt vmcnt(0)` in front of the first use of a load it knows about, and
                    // that wait would also cover this wave's six weight pieces of the step (or, placed in front of them, hold them back
                    // for the rest of the poll's ~2500-cycle round trip -- measured: that was most of the barrier waits of a position).
                    auto poll_issue = [&]() {
                        asm volatile("global_load_dwordx2 %0, %3, off sc1\n\tglobal_load_dwordx2 %1, %4, off sc1\n\tglobal_load_dwordx2 %2, %5, off sc1"
                                     : "=&v"(gr[0]), "=&v"(gr[1]), "=&v"(gr[2])
                                     : "v"(src), "v"(src + 64), "v"(src + 128)
                                     : "memory");
                        pending = true;
                    };
                    auto poll_landed_behind_weights = [&]() {      // the poll is older than the six weight pieces just issued
                        asm volatile("s_waitcnt vmcnt(6)" : "+v"(gr[0]), "+v"(gr[1]), "+v"(gr[2])::"memory");
                    };
                    auto poll_landed = [&]() { asm volatile("s_waitcnt vmcnt(0)" : "+v"(gr[0]), "+v"(gr[1]), "+v"(gr[2])::"memory"); };
                    auto poll_take = [&]() -> bool {      // the poll in flight: complete? then into the window's halo column
                        const bool ok = aborted || ((unsigned)(gr[0] >> 32) == tag && (unsigned)(gr[1] >> 32) == tag && (unsigned)(gr[2] >> 32) == tag);
                        pending = false;
                        if (__builtin_amdgcn_ballot_w64(!ok) != 0ull) return false;
                        const int m = imp_m;
                        const MapP mp = map_of(m);
                        const int px = d ? PWS - 1 : 0;
#pragma unroll
                        for (int j = 0; j < 3; ++j) {
                            const int gi = lane + 64 * j, e = gi >> 1, row = e >> 3, q = e & 7;
                            int sl = base_of(wbp, m) + row + 6 - m;
                            sl -= sl >= mp.R ? mp.R : 0;
                            *reinterpret_cast<unsigned*>(smem + mp.off + sl * ROWB1 + (q >> 2) * CHB + px * 32 + (q & 3) * 8 + (gi & 1) * 4) =
                                aborted ? 0u : (unsigned)gr[j];
                        }
                        imp_done = true;
                        return true;
                    };
                    if (d < 3) {
                        // weights of q = 3 g + 4, + 5, + 6 into the slots step g - 1 has finished with: wave d the whole slot q + d
                        {
                            int q7 = wq7 + d, q78 = wq78 + d;
                            q7 -= q7 >= NWS ? NWS : 0;
                            q78 -= q78 >= WPER ? WPER : 0;
                            if (!(NESR_STRIP_ABL & 4))
                                glds16_slot6_s(a.wstream + (size_t)q78 * WSLOT, (unsigned)lane * 16u, __builtin_amdgcn_readfirstlane(lds0 + WRING + q7 * WSLOT));
                        }
                        wq7 += 3; wq7 -= wq7 >= NWS ? NWS : 0;
                        wq78 += 3; wq78 -= wq78 >= WPER ? WPER : 0;
                    } else if (pos + 1 < pos1) {
                        const int tk = idx == 1 ? 0 : idx == 4 ? 1 : idx == 8 ? 2 : idx == 13 ? 3 : idx == 16 ? 4 : -1;
                        if (idx >= 18 && idx < 25) {
                            // the next position's 12 x0 rows over seven steps (2, 2, 2, 2, 2, 1, 1), once conv5 has finished with x0 (its
                            // steps 0..3); they have landed before the position's LAST barrier: behind it the MFMA waves already request
                            // the next position's first fragments
                            const int first = idx < 23 ? 2 * (idx - 18) : 10 + (idx - 23), cnt = idx < 23 ? 2 : 1;
                            for (int k = 0; k < cnt; ++k) {
                                const int rr = first + k;
                                int slot = base_of(wbp, 0) + rr;
                                slot -= slot >= 18 ? 18 : 0;
                                x0_row(BH * pos + 13 + rr, slot);
                            }
                        } else if (tk >= 0 && !(NESR_STRIP_ABL & 32)) {
                            // ... and long before that, their lines into L2: one dword per 128 bytes by LDS-DMA into a scratch corner of
                            // LDS that nobody reads.  A load that goes out to HBM holds up whatever is queued behind it in this CU's
                            // memory pipeline for its whole latency (measured: the weight DMAs of the next step land ~1500 cycles
                            // late), so the touches go out in the steps that end with a layer's epilogue -- the one time the next
                            // step's weights are not waited for -- and the row DMAs above then return in L2 time.
                            const int t = lane + 64 * tk;
                            const int row = t / 24, c = (t % 24) / 6, piece = t % 6;
                            const int x_lo = xs > 0 ? xs - 1 : 0, x_hi = xs + BW + 1 < w ? xs + BW + 1 : w;
                            const int seg = (x_hi - x_lo) * 32;
                            const int off = piece * 128 < seg - 4 ? piece * 128 : seg - 4;
                            const int y0 = BH * pos + 13;
                            if (t < 288 && y0 + row < h)
                                glds4_s(img_base + (size_t)y0 * a.W * 32,
                                        (unsigned)c * (unsigned)a.chunk_bytes + (unsigned)(row * a.W + x_lo) * 32u + (unsigned)off,
                                        __builtin_amdgcn_readfirstlane(lds0 + SCRO));
                        }
                    }
                    SSTAMP(MW + d, idx, 3);
                    if (polling && pending) {      // last step's poll, looked at behind this step's weight pieces
                        if (NESR_STRIP_ABL & 4) poll_landed(); else poll_landed_behind_weights();
                        poll_take();
                    }
                    if (polling && !imp_done) {
                        const int m = imp_m;
                        const bool block = idx == (m == 1 ? 3 : m == 2 ? 7 : m == 3 ? 12 : 23);
                        unsigned long long t_start = 0;
                        for (unsigned spin = 0;; ++spin) {
                            poll_issue();
                            if (!block) break;
                            poll_landed();
                            if (poll_take()) break;
                            // bounded: a neighbour that never publishes (its workgroup not resident) ends in an abort word
                            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                            if (spin == 0) t_start = now;
                            if ((spin & 7u) == 7u) {
                                const unsigned ab = __hip_atomic_load(a.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                if (ab != 0u || now - t_start > a.timeout_ticks) {
                                    if (ab == 0u && lane == 0)
                                        __hip_atomic_store(a.abort_flag, 1u | ((unsigned)m << 8) | ((unsigned)blockIdx.x << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    aborted = true;
                                }
                            }
                        }
                    }
                }
                wbp = advance_bases(wbp);
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        return;
    }

    // =============================================================== MFMA role: wave wv = rows 3 wv .. 3 wv + 2 of the position, all 32 couts
    // of the step.  (Measured alternative: eight MFMA waves, two per SIMD, each half the couts -- same values, same time: the
    // pair of a SIMD finishes a step in ~1650 cycles against ~1300 for one wave doing all 54 MFMAs, because every pixel
    // fragment is then read twice; the step is bounded by MFMA time PLUS the time the fragment reads take to return.)
    if (NESR_STRIP_MFMA_PRIO) __builtin_amdgcn_s_setprio(NESR_STRIP_MFMA_PRIO);
    const int j16 = lane & 15, g4 = lane >> 4;
    const unsigned lane_b = lds0 + (g4 >> 1) * CHB + j16 * 32 + (g4 & 1) * 16;
    const unsigned lane_a = lds0 + WRING + (g4 >> 1) * 1024 + (g4 & 1) * 512 + j16 * 16;
    const int lane_e = (j16 + 1) * 32 + g4 * 8;     // a lane's 4 couts of its pixel inside a chunk row
    const float inv_s1 = 1.0f / a.s1;
    int aq = 0;                                     // ring slot of the current step's first tap column
    f32x4 acc[2][3][2];                             // [cout group (conv5) | 0][row][cout half]
    f32x4 Bf[3][5], Af[2][3][2];                       // pixel fragments of tap column i (order 1, 0, 2) in buffer i; weight fragments [dy][cout half]
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) acc[c][r][mt] = zero4;

    // fragment row addresses of the step (layer m, chunk pair p) for the packed window bases
    auto baddr_of = [&](unsigned long long bases, int m, int p, unsigned (&ba)[5]) {
        const MapP mp = pair_map(p);
        const int b = base_of(bases, p < 2 ? 0 : p - 1), t0 = 3 * wv + 5 - m;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            int sl = b + t0 + k;
            sl -= sl >= mp.R ? mp.R : 0;
            ba[k] = lane_b + mp.off + sl * mp.rowb;
        }
    };
    auto load_b1 = [&](int buf, int k, const unsigned (&ba)[5], int dx) {      // buf, k: compile-time constants at every call site
        if (NESR_STRIP_ABL & 16) { asm volatile("" : "+v"(Bf[buf][k])); return; }
        Bf[buf][k] = *((lds_f32x4)(size_t)(ba[k] + dx * 32));
    };
    auto load_a = [&](int buf, int dy, int slot) {      // buf, dy: compile-time constants at every call site
        if (NESR_STRIP_ABL & 16) { asm volatile("" : "+v"(Af[buf][dy][0]), "+v"(Af[buf][dy][1])); return; }
        const unsigned wa = lane_a + slot * WSLOT + dy * 2048;
        Af[buf][dy][0] = *((lds_f32x4)(size_t)(wa));
        Af[buf][dy][1] = *((lds_f32x4)(size_t)(wa + 256));
    };
    // One step: tap columns in the order 1, 0, 2 (column 1 touches no halo pixel, so its fragments may be requested before
    // the barrier behind which a freshly imported halo column becomes visible).  Column i's pixel fragments sit in buffer
    // i: the next column's five are requested at the head of a column, the next STEP's first five during the last
    // column (buffer 0 is free by then); the six weight fragments of a column are re-requested for the next column (or
    // the next step) right after their last MFMA -- 12 to 18 MFMAs before their first use.  (Measured: spreading the reads
    // one per MFMA gap with the weights of a column requested inside that column -- 6 to 12 MFMAs ahead -- is 14 % slower:
    // the LDS round trip under four waves' traffic is longer than that.)  No branch inside: a wave whose rows lie outside the
    // image multiplies too (its sums are masked in the epilogue; it would only wait at the barrier otherwise).
    // LM (pixel-fragment loads of this step): 0 all; 1 columns 0 and 2 only, buffer 0 is left alone; 2 only the next step's first
    // column.  conv5 runs the two cout groups of a chunk pair as steps (1, 2): the second one finds all three columns still in
    // the registers.  (The next step's fragment addresses -- scalar work + five adds -- are computed before the step's barrier, not
    // inside the MFMA stream: measured there, they cost the stream 350 cycles where they save 100 in front of the barrier.)
    // Weight fragments: two register sets, a tap column's in set (P + i) & 1 (P: parity of the step's first column; three columns a step,
    // so it flips every step -- 26 steps a position: P is the parity of the step's number in the position, a compile-time constant
    // of every call).  A set is re-requested right after a vertical tap's last MFMA for the column TWO ahead (this step's
    // third, or the next step's first: 36 MFMAs of lead instead of 18); only the step's second column cannot be asked for that early
    // -- its slot lands behind this step's barrier -- and is requested at the head of the first.
    auto step_body = [&](auto Cc, auto Lc, auto Pc, const unsigned (&ba)[5], const unsigned (&nba)[5]) {
        constexpr int CG = decltype(Cc)::value, LM = decltype(Lc)::value, P = decltype(Pc)::value;
        static_for3([&](auto Ic) {
            constexpr int i = decltype(Ic)::value, cb = (P + i) & 1;
            int sl1 = aq + 1, sl2 = aq + i + 2;
            sl1 -= sl1 >= NWS ? NWS : 0;
            sl2 -= sl2 >= NWS ? NWS : 0;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                if (i == 0) { if (LM != 2) load_b1(1, k, ba, 0); }
                else if (i == 1) { if (LM != 2) load_b1(2, k, ba, 2); }
                else if (LM != 1) load_b1(0, k, nba, 1);
            }
            if constexpr (i == 0) {
                load_a(cb ^ 1, 0, sl1); load_a(cb ^ 1, 1, sl1); load_a(cb ^ 1, 2, sl1);
            }
            static_for3([&](auto Dc) {
                constexpr int dy = decltype(Dc)::value;
                if (NESR_STRIP_ABL & 2) {
                    asm volatile("" ::"v"(Af[cb][dy][0]), "v"(Af[cb][dy][1]), "v"(Bf[i][dy]), "v"(Bf[i][dy + 2]));
                } else {
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        const bf16x8 wf = __builtin_bit_cast(bf16x8, Af[cb][dy][mt]);
#pragma unroll
                        for (int r = 0; r < 3; ++r)
                            acc[CG][r][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, __builtin_bit_cast(bf16x8, Bf[i][r + dy]), acc[CG][r][mt], 0, 0, 0);
                    }
                }
                if constexpr (i < 2) load_a(cb, dy, sl2);
                // pinned: the hipcc scheduler otherwise sinks every read to just before its first use and the MFMA stream
                // stops at a short lgkmcnt wait a dozen times per step
                __builtin_amdgcn_sched_barrier(0);
            });
        });
        aq += 3;
        aq -= aq >= NWS ? NWS : 0;
    };
    for (int it = it0; it < it1; ++it) {
        const int4 item = a.items[2 * it], seg = a.items[2 * it + 1];
        const int img = item.x & 0xffff, s = item.y, h = item.z, w = item.w;
        const int pos0 = seg.x, pos1 = seg.y, row_lo = seg.z, row_hi = seg.w;      // positions run, output rows stored
        const int xs = s * BW, ns = (w + BW - 1) / BW;
        const int slot_id = (item.x >> 16) * a.smax + s;
        const bool colok = xs + j16 < w;
        const bool edge = (j16 == 0 && s > 0) || (j16 == BW - 1 && s + 1 < ns);
        char* xch_mine = a.xch + (size_t)slot_id * XCH_STRIP + (j16 == BW - 1 ? 2 * 4 * XCH_LAYER : 0) + g4 * 16 + (3 * wv) * 128;
        const size_t img_px = (size_t)img * a.H * a.W;

        for (int k = threadIdx.x; k < ACT / 16; k += 64 * (MW + DW)) *reinterpret_cast<f32x4*>(smem + k * 16) = zero4;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        unsigned long long wbp = WB_INIT;
        for (int k = 0; k < pos0; ++k) wbp = advance_bases(wbp);
        __builtin_amdgcn_s_barrier();      // the first position's x0 rows and the weight ring's first slots have landed
        asm volatile("" ::: "memory");
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)      // conv1's sums start from its bias (the table is in LDS since the first barrier)
                acc[0][r][mt] = *reinterpret_cast<const f32x4*>(smem + BIASO + (16 * mt + 4 * g4) * 4);
        unsigned ba[5], nba[5];
        baddr_of(wbp, 1, 0, ba);
#pragma unroll
        for (int k = 0; k < 5; ++k) load_b1(0, k, ba, 1);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) load_a(0, dy, aq);

        for (int pos = pos0; pos < pos1; ++pos) {
            const unsigned tagbase = a.epoch + (unsigned)pos * 8u;
#if NESR_STRIP_ABL & 256
            const bool stamp_on = it == it0 && pos == 5;
            int sidx = 0;
            if (stamp_on && blockIdx.x == 77 && lane == 0) { unsigned long long r_ = __builtin_amdgcn_s_memrealtime(), t_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); g_strip_stamps[wv][30][0] = r_; g_strip_stamps[wv][30][1] = t_; }
#endif
            const unsigned long long wbn = advance_bases(wbp);
            const int yw = BH * pos + 3 * wv;       // this wave's first row of layer 1 at this position
            // ---- conv1..conv4 (the layer number is a compile-time constant of each copy)
            // bias vector (floats boff + 16 mt + 4 g4 .. + 3 of the table) for this lane's four couts
            auto bias4 = [&](int boff, int mt) { return *reinterpret_cast<const f32x4*>(smem + BIASO + (boff + 16 * mt + 4 * g4) * 4); };
            auto layer = [&](auto Mc) {
                constexpr int m = decltype(Mc)::value;
                constexpr int first = m == 1 ? 0 : m == 2 ? 2 : m == 3 ? 5 : 9;      // number of the layer's first step in the position
                static_for_n<m + 1>([&](auto Pp) {
                    constexpr int p = decltype(Pp)::value;
                    if (p < m) baddr_of(wbp, m, p + 1, nba);
                    else baddr_of(wbp, m + 1, 0, nba);
                    SSTAMP(wv, sidx, 0);
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    SSTAMP(wv, sidx, 1);
                    step_body(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, (first + p) & 1>{}, ba, nba);
                    SSTAMP(wv, sidx, 2);
#if NESR_STRIP_ABL & 256
                    ++sidx;
#endif
#pragma unroll
                    for (int k = 0; k < 5; ++k) ba[k] = nba[k];
                });
                if (!(NESR_STRIP_ABL & 8)) {
                    // ---- x_m = lrelu(acc + bias): into this strip's window, edge columns also to the neighbours
                    // (the sums started from the bias: a bias read here would stand, with its LDS round trip, between the last
                    // MFMA and the first store)
                    const MapP mp = map_of(m);
                    const unsigned tag = tagbase + (unsigned)m;
                    char* xdst = xch_mine + ((size_t)(pos & 1) * 4 + (m - 1)) * XCH_LAYER;
                    const int bm = base_of(wbp, m);
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        const int y = yw - (m - 1) + r;
                        const bool ok = colok && y >= 0 && y < h;
                        int sl = bm + 3 * wv + r + 6 - m;
                        sl -= sl >= mp.R ? mp.R : 0;
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt) {
                            uint2 pk = pack4_bf16(lrelu4(acc[0][r][mt]));
                            pk.x = ok ? pk.x : 0u;
                            pk.y = ok ? pk.y : 0u;
                            *reinterpret_cast<uint2*>(smem + mp.off + sl * ROWB1 + mt * CHB + lane_e) = pk;
                            if (edge) store16_sc1(xdst + r * 128 + mt * 64, uint4{pk.x, tag, pk.y, tag});
                            if (m < 4) acc[0][r][mt] = bias4(m * 32, mt);      // the next layer's sums start from its bias
                        }
                    }
                    if (m < 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    SSTAMP(wv, sidx - 1, 3);
                } else {
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt) { asm volatile("" ::"v"(acc[0][r][mt])); acc[0][r][mt] = bias4(m < 4 ? m * 32 : 0, mt); }
                }
            };
            layer(std::integral_constant<int, 1>{});
            layer(std::integral_constant<int, 2>{});
            layer(std::integral_constant<int, 3>{});
            layer(std::integral_constant<int, 4>{});
            // ---- conv5, its two cout groups alternating.  x5 * s1 + x0 = (x5 + x0 / s1) * s1: the first residual starts
            // the sums, while its rows are still in the window (the next position's x0 rows overwrite them from conv5's
            // fifth step on)
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                int sl = base_of(wbp, 0) + 3 * wv + r + 1;      // row 12 pos - 4 + 3 wv + r
                sl -= sl >= 18 ? 18 : 0;
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
                    {
                        const f32x4 u = unpack4_bf16(*reinterpret_cast<const uint2*>(smem + sl * ROWB0 + (2 * c2 + mt) * CHB + lane_e)), bz = bias4(128 + 32 * c2, mt);
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[c2][r][mt][i] = fmaf(u[i], inv_s1, bz[i]);
                    }
            }
            uint4 res2q[2][3];      // second residual, 16 bytes per lane in the regrouped layout of the output stores (below)
            for (int p = 0; p < 6; ++p) {
                SSTAMP(wv, sidx, 0);
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                SSTAMP(wv, sidx, 1);
                step_body(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, ba, ba);   // the second cout group finds these pixels in the registers
                SSTAMP(wv, sidx, 2);
#if NESR_STRIP_ABL & 256
                ++sidx;
#endif
                if (p < 5) baddr_of(wbp, 5, p + 1, nba);
                else baddr_of(wbn, 1, 0, nba);                             // (past the last position: rows nobody uses)
                if (p == 5 && a.res2) {
#pragma unroll
                    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                        for (int r = 0; r < 3; ++r) {
                            const int y = yw - 4 + r;
                            const bool valid = colok && y >= 0 && y < h;
                            const size_t pix = img_px + (size_t)(valid ? y : 0) * a.W + (valid ? xs + j16 : 0);
                            res2q[c2][r] = *reinterpret_cast<const uint4*>(a.res2 + (size_t)(2 * c2 + (g4 & 1)) * a.chunk_bytes + pix * 32 + (g4 >> 1) * 16);
                        }
                }
                SSTAMP(wv, sidx, 0);
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                SSTAMP(wv, sidx, 1);
                step_body(std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{}, ba, nba);
                SSTAMP(wv, sidx, 2);
#if NESR_STRIP_ABL & 256
                ++sidx;
#endif
#pragma unroll
                for (int k = 0; k < 5; ++k) ba[k] = nba[k];
            }
            if (!(NESR_STRIP_ABL & 8)) {
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const int y = yw - 4 + r;
                    // (a row segment stores its own rows only: the rows of its first position above row_lo are the warm-up whose
                    // x1..x4 inputs it never computed -- the segment above stores them)
                    const bool valid = colok && y >= row_lo && y < row_hi;
                    const size_t pix = img_px + (size_t)(valid ? y : 0) * a.W + (valid ? xs + j16 : 0);
#pragma unroll
                    for (int c2 = 0; c2 < 2; ++c2) {
                        // A lane holds 4 couts (8 bytes) of chunk 2 c2 and 4 of chunk 2 c2 + 1.  One v_permlane16_swap per register
                        // (odd 16-lane rows of the first <-> even rows of the second) leaves it with 8 consecutive couts of ONE chunk --
                        // chunk 2 c2 + (g4 & 1), byte 16 (g4 >> 1) of the pixel's 32 -- so that the block's output leaves in 6 store
                        // instructions per wave and position instead of 12, and the second residual arrives in 6 loads (the swap is its
                        // own inverse).  A vector-memory instruction issued by an MFMA wave beside the DMA waves' bursts costs that
                        // wave a few hundred cycles: these are the only ones it has.
                        f32x4 v[2] = {acc[c2][r][0] * a.s1, acc[c2][r][1] * a.s1};
                        if (a.res2) {
                            const uint4 q = res2q[c2][r];
                            const auto sx = __builtin_amdgcn_permlane16_swap(q.x, q.z, false, false);
                            const auto sy = __builtin_amdgcn_permlane16_swap(q.y, q.w, false, false);
                            const f32x4 q0 = unpack4_bf16(uint2{sx[0], sy[0]}), q1 = unpack4_bf16(uint2{sx[1], sy[1]});
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                v[0][i] = __fadd_rn(__fmul_rn(v[0][i], a.s2), q0[i]);
                                v[1][i] = __fadd_rn(__fmul_rn(v[1][i], a.s2), q1[i]);
                            }
                        }
                        const uint2 p0 = pack4_bf16(v[0]), p1 = pack4_bf16(v[1]);
                        const auto ox = __builtin_amdgcn_permlane16_swap(p0.x, p1.x, false, false);
                        const auto oy = __builtin_amdgcn_permlane16_swap(p0.y, p1.y, false, false);
                        if (valid)
                            *reinterpret_cast<uint4*>(a.out + (size_t)(2 * c2 + (g4 & 1)) * a.chunk_bytes + pix * 32 + (g4 >> 1) * 16) = uint4{ox[0], oy[0], ox[1], oy[1]};
                        if (c2 == 0) {
                            acc[0][r][0] = bias4(0, 0);      // conv1 of the next position
                            acc[0][r][1] = bias4(0, 1);
                        }
                    }
                }
            } else {
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt) { asm volatile("" ::"v"(acc[c2][r][mt])); acc[c2][r][mt] = bias4(0, mt); }
            }
            SSTAMP(wv, 25, 3);
#if NESR_STRIP_ABL & 256
            if (stamp_on && blockIdx.x == 77 && lane == 0) { unsigned long long r_ = __builtin_amdgcn_s_memrealtime(), t_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); g_strip_stamps[wv][31][0] = r_; g_strip_stamps[wv][31][1] = t_; }
#endif
            wbp = wbn;
        }
    }
}

inline uint16_t f2bf(float f) {  // round-to-nearest-even, NaN stays NaN
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

}  // namespace

#if NESR_STRIP_ABL & 256
extern "C" int nesr_debug_strip_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_strip_stamps), sizeof(unsigned long long) * 8 * 32 * 8);
}
#endif

size_t strip_weight_bytes() { return (size_t)WPER * WSLOT; }

// The five convs of one dense block (OIHW f32, conv_k: [32 | 64][64 + 32 (k-1)][3][3]) -> the kernel's weight stream:
// [step: layer m, chunk pair p, (conv5: cout group)][tap column in the order 1, 0, 2][dy][chunk of the pair][k half]
// [cout % 32][8 channels] bf16.
void pack_strip_weights(const float* const w[5], uint16_t* dst) {
    size_t slot = 0;
    for (int m = 1; m <= 5; ++m) {
        const int cin = 64 + 32 * (m - 1), npairs = m + 1, ncg = m == 5 ? 2 : 1;
        for (int p = 0; p < npairs; ++p)
            for (int cg = 0; cg < ncg; ++cg)
                for (int i = 0; i < 3; ++i, ++slot) {
                    const int dx = i == 0 ? 1 : (i == 1 ? 0 : 2);
                    uint16_t* d = dst + slot * (WSLOT / 2);
                    for (int dy = 0; dy < 3; ++dy)
                        for (int ck = 0; ck < 2; ++ck)
                            for (int kh = 0; kh < 2; ++kh)
                                for (int o = 0; o < 32; ++o)
                                    for (int kk = 0; kk < 8; ++kk) {
                                        const int co = 32 * cg + o, ci = 32 * p + 16 * ck + 8 * kh + kk;
                                        d[(((dy * 2 + ck) * 2 + kh) * 32 + o) * 8 + kk] = f2bf(w[m - 1][(((size_t)co * cin + ci) * 3 + dy) * 3 + dx]);
                                    }
                }
    }
}

// Which workgroup runs which strips, in which order.  The strips of an image start together on as many workgroups
// (they exchange edge columns position by position); an image therefore occupies ns workgroups for npos positions, and
// the images are packed onto `cus` workgroup time lines: best of a few hundred randomised greedy placements (an image
// goes to the workgroups that become free earliest and, among those, to the ones that waited least).  Every workgroup's
// list is in placement order -- one global order of the images -- so the waits between workgroups cannot form a cycle.
//
// A batch with fewer strips than the device has compute units (one rank's share of a sharded frame, a small frame) would
// leave most of them idle for the 23 positions of a tile.  An image may therefore be cut into ROW SEGMENTS that run as
// independent units: the segment that stores the rows of positions [a, b) of the image's own position grid starts one
// position early (position a - 1 is its warm-up: layer m there finds no rows of x_{m-1} from a position before, so the
// first 4 output rows it could store are wrong and are left to the segment above -- from row 12 (a - 1) + 4 on every value
// has exactly the operands of the unsegmented sweep, in the same order: the same bits).  One more position per extra
// segment, `seg_len` = most positions of the image's grid per segment (0: the packer tries several and keeps the
// shortest makespan; < 0: never cut).
namespace {
struct Unit { int img, vimg, ns, pos0, pos1, row_lo, row_hi; };

bool pack_units(const std::vector<Unit>& units, const int* hw, int cus, int trials, StripSchedule& best) {
    const int n = (int)units.size();
    std::mt19937 rng(12345);
    std::vector<int> order(n), load(cus), idx(cus);
    bool improved = false;
    for (int trial = 0; trial < trials; ++trial) {
        for (int i = 0; i < n; ++i) order[i] = i;
        // longest first; within one length a random order of the widths (trial 0: widest first)
        if (trial) std::shuffle(order.begin(), order.end(), rng);
        std::stable_sort(order.begin(), order.end(), [&](int x, int y) {
            const int lx = units[x].pos1 - units[x].pos0, ly = units[y].pos1 - units[y].pos0;
            if (lx != ly) return lx > ly;
            return trial == 0 && units[x].ns > units[y].ns;
        });
        std::fill(load.begin(), load.end(), 0);
        std::vector<std::vector<int>> lists(cus);
        for (int oi = 0; oi < n; ++oi) {
            const Unit& u = units[order[oi]];
            for (int g = 0; g < cus; ++g) idx[g] = g;
            std::sort(idx.begin(), idx.end(), [&](int x, int y) { return load[x] != load[y] ? load[x] < load[y] : x < y; });
            const int start = load[idx[u.ns - 1]];            // the earliest time ns workgroups are free
            int cnt = 0;
            while (cnt < cus && load[idx[cnt]] <= start) ++cnt;
            // of the cnt candidates take the ns that became free last (least idle time thrown away)
            for (int k = 0; k < u.ns; ++k) {
                const int g = idx[cnt - u.ns + k];
                load[g] = start + (u.pos1 - u.pos0);
                lists[g].push_back(order[oi] * 4096 + k);
            }
        }
        const int mk = *std::max_element(load.begin(), load.end());
        if (best.makespan < 0 || mk < best.makespan) {
            improved = true;
            best.makespan = mk;
            best.items.clear();
            best.wg_first.clear();
            int used = 0;
            for (int g = 0; g < cus; ++g) {
                if (lists[g].empty()) continue;
                best.wg_first.push_back((int)best.items.size() / 8);
                for (int code : lists[g]) {
                    const Unit& u = units[code / 4096];
                    const int s = code % 4096;
                    const int it[8] = {u.img | (u.vimg << 16), s, hw[2 * u.img], hw[2 * u.img + 1], u.pos0, u.pos1, u.row_lo, u.row_hi};
                    best.items.insert(best.items.end(), it, it + 8);
                }
                ++used;
            }
            best.wg_first.push_back((int)best.items.size() / 8);
            best.grid = used;
            best.nvimg = n;
        }
    }
    return improved;
}
}  // namespace

StripSchedule strip_schedule(int n, const int* hw, int cus, int seg_len) {
    StripSchedule best;
    best.makespan = -1;
    long work = 0;
    int smax = 1, pmax = 1;
    for (int i = 0; i < n; ++i) {
        const int ns = (hw[2 * i + 1] + BW - 1) / BW, npos = (hw[2 * i] + 4 + BH - 1) / BH;
        if (ns > cus) return best;     // an image wider than the device has workgroups: the per-layer kernels take it
        smax = std::max(smax, ns);
        pmax = std::max(pmax, npos);
        work += (long)ns * npos;
    }
    // images of at most `thr` positions are cut into segments of at most L positions of their grid (L = 0: not cut)
    auto units_for = [&](int L, int thr) {
        std::vector<Unit> units;
        for (int i = 0; i < n; ++i) {
            const int h = hw[2 * i], ns = (hw[2 * i + 1] + BW - 1) / BW, npos = (h + 4 + BH - 1) / BH;
            const int k = L > 0 && npos <= thr ? (npos + L - 1) / L : 1;
            int a = 0;
            for (int j = 0; j < k; ++j) {
                const int b = a + npos / k + (j < npos % k ? 1 : 0);
                Unit u;
                u.img = i; u.vimg = (int)units.size(); u.ns = ns;
                u.pos0 = a > 0 ? a - 1 : 0; u.pos1 = b;
                u.row_lo = a > 0 ? BH * a - 4 : 0;
                u.row_hi = b < npos ? BH * b - 4 : h;
                if (u.row_hi > h) u.row_hi = h;
                if (u.row_lo < u.row_hi || k == 1) units.push_back(u);
                a = b;
            }
        }
        return units;
    };
    pack_units(units_for(0, 0), hw, cus, 300, best);
    const int bound = (int)((work + cus - 1) / cus);
    auto consider = [&](int L, int thr, int trials) {
        if (L < 2 || best.makespan <= bound) return;
        const std::vector<Unit> u = units_for(L, thr);
        if (u.size() >= 32768) return;
        StripSchedule cut = best;      // pack_units replaces it only by a shorter makespan
        if (pack_units(u, hw, cus, trials, cut)) best = cut;
    };
    if (seg_len > 0) {
        StripSchedule cut;
        cut.makespan = -1;
        const std::vector<Unit> u = units_for(seg_len, pmax);
        if (u.size() < 32768 && pack_units(u, hw, cus, 100, cut)) best = cut;
    } else if (seg_len == 0 && pmax > 4) {
        // Cutting pays when whole images leave compute units idle.  (a) every image, in 2, 3, ... segments of the tallest
        // (batches with fewer strips than compute units); (b) only the images of at most p positions, halved or cut in three
        // (a full 4K frame: its 512 tall strips fill the device twice, the 128 strips of the short bottom tile row would
        // keep half of it busy for 6 more positions -- halved they are 3 or 4 positions on all of it: 52 -> 50).
        for (int k = 2; k <= 8; ++k) consider((pmax + k - 1) / k, pmax, 60);
        std::vector<int> classes;
        for (int i = 0; i < n; ++i) classes.push_back((hw[2 * i] + 4 + BH - 1) / BH);
        std::sort(classes.begin(), classes.end());
        classes.erase(std::unique(classes.begin(), classes.end()), classes.end());
        for (size_t ci = 0; ci + 1 < classes.size() && ci < 6; ++ci)
            for (int k = 2; k <= 3; ++k) consider((classes[ci] + k - 1) / k, classes[ci], 150);
    }
    best.smax = smax;
    best.efficiency = best.makespan > 0 ? (double)work / ((double)cus * best.makespan) : 0.0;
    return best;
}

hipError_t launch_rdb_bf16_strip(const StripLaunch& r, hipStream_t s) {
    static unsigned long long attr_done = 0;
    {
        const hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&rdb_bf16_strip_kernel), LDSB, attr_done);
        if (e != hipSuccess) return e;
    }
    if (r.grid <= 0) return hipSuccess;
    if (4 * r.chunk_bytes >= (1ll << 32)) return hipErrorInvalidValue;     // 32-bit lane offsets across the four x0 chunks
    StripArgs a;
    a.cur = static_cast<const char*>(r.cur);
    a.chunk_bytes = r.chunk_bytes;
    a.out = static_cast<char*>(r.out);
    a.res2 = static_cast<const char*>(r.res2);
    a.s1 = r.s1; a.s2 = r.s2;
    a.wstream = static_cast<const char*>(r.wstream);
    a.bias = r.bias;
    a.H = r.H; a.W = r.W;
    a.items = static_cast<const int4*>(r.items);
    a.wg_first = r.wg_first;
    a.xch = static_cast<char*>(r.xch);
    a.smax = r.smax;
    a.epoch = r.epoch;
    a.abort_flag = r.abort_flag;
    a.timeout_ticks = r.timeout_ticks;
    static int resident = -1;      // one workgroup per CU (LDS): the schedule's grid is at most the compute-unit count
    if (resident < 0) {
        int per_cu = 0, dev = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rdb_bf16_strip_kernel, 64 * (MW + DW), LDSB) != hipSuccess || hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            return hipErrorUnknown;
        resident = per_cu * cus;
    }
    if (r.grid > resident) return hipErrorLaunchOutOfResources;
    const int grid = r.grid - r.debug_drop;     // test hook (nesr_debug_fault): the last workgroups never start
    if (grid <= 0) return hipSuccess;
    hipLaunchKernelGGL(rdb_bf16_strip_kernel, dim3((unsigned)grid), dim3(64 * (MW + DW)), LDSB, s, a);
    return hipGetLastError();
}

}  // namespace nesr
