// f32 3x3 convolution by Winograd F(2x2, 3x3) on the f32 matrix cores.
//
// The direct implicit GEMM (conv3x3_mfma.hip) spends 9 MFMA-MACs per output pixel and channel pair
// and sits at ~76 % of the 157 TFLOP/s f32 MFMA peak; Winograd needs 16 per 2x2 outputs = 4 per
// pixel: 2.25x less matrix work for the same convolution (Lavin & Gray, "Fast Algorithms for
// Convolutional Neural Networks").  Numerically it is not the k-ordered fmaf chain of the direct
// kernel: the transforms add a few ulps per layer (measured against the oracle in
// tests/test_gpu_conv.py / test_gpu_rrdbnet.py; the north-star tolerance is 1e-3 on the output).
//
//   Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A        per 4x4 input patch d, 3x3 kernel g
//
//   GEMM view : 16 independent GEMMs (one per transform position p): M_p[cout][tile] +=
//               U_p[cout][cin] * V_p[cin][tile];  v_mfma_f32_16x16x4_f32 with A = U (16 couts x 4
//               k-slots), B = V (4 k-slots x 16 tiles).
//   workgroup : output tile 8 rows x 16 cols = 4 x 8 Winograd tiles; 4 waves.  Wave w: half = w>>1
//               (Winograd tile rows 2*half, 2*half+1 -> 16 tiles = the MFMA's 16 columns) and
//               channel group cgw = w&1; it owns NT groups of 16 output channels
//               (Cout 32: couts 16*cgw..+15; Cout 64: couts 32*cgw..+31).
//   lane      : j = lane&15 -> tile (row 2*half + (j>>3), col j&7); kq = lane>>4 -> K-slot: the lane
//               transforms input channels {2kq, 2kq+1} of its own tile's 4x4 patch in registers
//               (32 adds per channel, VALU beside the MFMA pipe) and feeds them as the B operand.
//   K loop    : chunks of 8 input channels into a ring of 3 (Cout 32) or 2 (Cout 64) LDS slots; the
//               input halo tile goes global -> registers -> LDS (branch-free, zero padding by
//               select), the weight slab by LDS-DMA.  Input image in LDS:
//               [kq plane][padded pixel][2 ch]; weights: host-transformed U, packed
//               [chunk][p][cout group][kq][16 couts][2 ch] so the A-operand ds_read_b64 of 32 lanes is
//               256 contiguous bytes.
//   epilogue  : the 16 M_p of a (tile, 4 couts) live in one lane -> output transform in registers,
//               then bias / LeakyReLU / residuals and 16-byte stores of 4 consecutive channels for
//               each of the tile's 2x2 pixels (same fused epilogue semantics as the direct kernel).
#include <cstdlib>

#include "nesr_kernels.h"

namespace nesr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) char lds_char;

namespace {

// LDS-DMA from inline asm (see conv3x3_bf16.hip): hipcc neither counts it nor waits for it; M0 is
// saved and restored inside the statement.
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

constexpr int TH = 8, TW = 16, PH = TH + 2, PW = TW + 2, NPIX = PH * PW;   // 180 padded pixels
constexpr int IN_ITEMS = 2 * NPIX;                                          // 16-byte staging items (4 channels each)
constexpr int IN_BYTES = 4 * NPIX * 8;                                      // 4 kq planes x pixel x 8 B = 5760

// NT = 16-channel groups per wave, CW = waves across the output channels (Cout = 16 * NT * CW); the
// workgroup has 2 * CW waves (two tile-halves).
// 16-byte feature-map store, write-through (sc1): the lines leave L2 while the kernel runs instead of
// as a dirty write-back at the kernel boundary (MI355X_MICROARCH.md, price list row "boundary":
// + B / 6 TB/s for B dirty bytes; 8-17 MB per launch here).  In-process A/B: -1.9 % frame time.
// Inline asm because no builtin carries the sc1 bit on a flat global store; the trailing s_nop keeps
// hipcc from reusing the data registers before the store has read them (cdna_hip_programming.md 5.7).
__device__ __forceinline__ void store_fm(float* p, f32x4 v) {
#ifdef NESR_PLAIN_STORES
    *reinterpret_cast<f32x4*>(p) = v;
#else
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#endif
}

#ifndef NESR_ABL
#define NESR_ABL 0   // timing ablations (tools/ablate.sh): 1 no MFMA, 2 no input transform, 4 no input staging, 8 no weight DMA, 16 no epilogue
#endif
template <int NT, int STAGES, int CW>
__global__ __launch_bounds__(128 * CW, 2) void conv3x3_wino_f32_kernel(ConvArgs a) {
    constexpr int THREADS = 128 * CW;
    constexpr int COUT = 16 * NT * CW;
    constexpr int W_BYTES = 16 * COUT * 32;          // per chunk: 16 positions x COUT x 8 ch x 4 B
    constexpr int W_ITEMS = W_BYTES / 16;            // 512 * NT * 2
    constexpr int STAGE_BYTES = IN_BYTES + W_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];

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

    // ---- staging.  Input halo tile: global -> registers -> LDS (the 16-byte items are split into the
    // two 8-byte kq planes; branch-free, zero padding by select).  Weight slab: a linear copy of
    // W_BYTES, done by LDS-DMA from inline asm (no staging registers, no ds_write issue slots).
    constexpr int RI = (IN_ITEMS + THREADS - 1) / THREADS;
    const char* src[RI];
    int dst[RI];
    bool zero[RI], has[RI];
    {
        const char* in = static_cast<const char*>(a.in);
#pragma unroll
        for (int i = 0; i < RI; ++i) {
            const int k0 = tid + THREADS * i;
            has[i] = k0 < IN_ITEMS;
            const int k = has[i] ? k0 : IN_ITEMS - 1;
            const int half = k >= NPIX ? 1 : 0;       // channels 4*half .. 4*half+3 -> planes 2*half, 2*half+1
            const int p = k - half * NPIX;
            const int py = p / PW, px = p - py * PW;
            const int Y = y0 - 1 + py, X = x0 - 1 + px;
            const bool ok = Y >= 0 && Y < a.h && X >= 0 && X < a.w_;
            const int sy = ok ? (Y >> a.up) : 0, sx = ok ? (X >> a.up) : 0;
            src[i] = in + ((((size_t)n * a.in_h + sy) * a.in_w + sx) * a.in_map.pix) * 4 + half * 16;
            dst[i] = (2 * half) * (NPIX * 8) + p * 8;
            zero[i] = !ok;
        }
    }
    const long long in_cstride = a.in_map.chunk * 4;
    f32x4 pv[RI];
    auto load_chunk = [&](int c) {
#pragma unroll
        for (int i = 0; i < RI; ++i) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(src[i] + (long long)c * in_cstride);
            pv[i] = zero[i] ? f32x4{0.f, 0.f, 0.f, 0.f} : v;
        }
    };
    auto store_chunk = [&](int stage) {
        char* st = smem + stage * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < RI; ++i) {
            if (has[i]) {   // channels (c0,c1) -> plane 2*half, (c2,c3) -> plane 2*half+1
                *reinterpret_cast<f32x2*>(st + dst[i]) = f32x2{pv[i][0], pv[i][1]};
                *reinterpret_cast<f32x2*>(st + dst[i] + NPIX * 8) = f32x2{pv[i][2], pv[i][3]};
            }
        }
    };
    const unsigned lds_base = (unsigned)(size_t)(lds_char*)(smem);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const char* wsrc = static_cast<const char*>(a.w) + tid * 16;
    constexpr int WR = W_ITEMS / THREADS;   // whole rounds: W_ITEMS = 512 * NT * CW
    auto dma_weights = [&](int c, int stage) {
        const unsigned d0 = lds_base + stage * STAGE_BYTES + IN_BYTES + wave_u * 1024;
        const char* s0 = wsrc + (size_t)c * W_BYTES;
#pragma unroll
        for (int i = 0; i < WR; ++i) glds16_asm(s0 + i * (THREADS * 16), __builtin_amdgcn_readfirstlane(d0 + i * (THREADS * 16)));
    };

    // ---- per-lane coordinates
    const int j = lane & 15, kq = lane >> 4;
    const int half = wave / CW, cgw = wave % CW;
    const int tr = 2 * half + (j >> 3), tc = j & 7;                      // Winograd tile inside the workgroup tile
    const int patch0 = kq * (NPIX * 8) + ((2 * tr) * PW + 2 * tc) * 8;    // byte offset of patch pixel (0,0), plane kq
    // weights: [p][cout group g = NT*cgw + t][kq][16][2 ch]: lane reads 8 B at ((p*NT*CW + g)*4 + kq)*128 + j*8
    const int w0 = IN_BYTES + ((NT * cgw) * 4 + kq) * 128 + j * 8;

    f32x4 acc[NT][16];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int p = 0; p < 16; ++p) acc[t][p] = f32x4{0.f, 0.f, 0.f, 0.f};

    // bias is fetched now so its latency hides under the K loop
    f32x4 bias_r[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bias_r[t] = *reinterpret_cast<const f32x4*>(a.bias + 16 * (NT * cgw + t) + 4 * kq);

    const int nchunks = a.cin / 8;
    // LDS ring of STAGES slots, STAGES-1 chunks staged ahead (3 slots for Cout=32; Cout=64 has room
    // for 2 per workgroup at two workgroups per CU)
    {   // all prologue chunks in flight together (one memory round trip)
        f32x4 pq[STAGES - 1][RI];
#pragma unroll
        for (int i = 0; i < STAGES - 1; ++i) {
            const int ci = i < nchunks ? i : nchunks - 1;
            dma_weights(ci, i);
            load_chunk(ci);
#pragma unroll
            for (int r = 0; r < RI; ++r) pq[i][r] = pv[r];
        }
#pragma unroll
        for (int i = 0; i < STAGES - 1; ++i) {
#pragma unroll
            for (int r = 0; r < RI; ++r) pv[r] = pq[i][r];
            store_chunk(i);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the prologue's weight DMAs have landed
    __syncthreads();

    // input transform helpers: V = B^T d B of this lane's tile for its two channels (f32x2 = both channels)
    auto read_row = [&](const char* st, int r, f32x2 (&d)[4][4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) d[r][q] = *reinterpret_cast<const f32x2*>(st + patch0 + (r * PW + q) * 8);
    };
    auto bt_cols = [&](const f32x2 (&d)[4][4], f32x2 (&t0)[4][4], int q) {   // column q of B^T d
        t0[0][q] = d[0][q] - d[2][q];
        t0[1][q] = d[1][q] + d[2][q];
        t0[2][q] = d[2][q] - d[1][q];
        t0[3][q] = d[1][q] - d[3][q];
    };
    auto b_row = [&](const f32x2 (&t0)[4][4], f32x2 (&V)[16], int r) {       // row r of (B^T d) B
        V[4 * r + 0] = t0[r][0] - t0[r][2];
        V[4 * r + 1] = t0[r][1] + t0[r][2];
        V[4 * r + 2] = t0[r][2] - t0[r][1];
        V[4 * r + 3] = t0[r][1] - t0[r][3];
    };
    // 16x16x4 f32 MFMA: 32-cycle issue, 40-cycle dependent latency -> the two k-steps of one accumulator
    // are separated by an MFMA on another accumulator.  Groups of 4 MFMAs; the weight fragments of the
    // next group are read while the current group runs, and (3-slot ring: the next chunk is already
    // visible) the NEXT chunk's patch reads and input transform are sliced into the groups as well, so
    // the VALU work runs beside the MFMA pipe instead of in front of it.  Everything is pinned with
    // sched_barrier: left alone hipcc sinks each ds_read to just before its s_waitcnt and re-pairs the
    // dependent MFMAs.
    constexpr int GROUPS = NT == 2 ? 16 : 8;   // NT=2: one position x 2 cout groups; NT=1: two positions
    constexpr bool PIPE = STAGES == 3;
    // conv_last (cout_real <= 16 real channels, planar / u8 output): the waves of the second
    // 16-channel group have nothing to compute -- they only help staging and keep the barriers
    const bool idle = NT == 1 && CW == 2 && a.cout_real > 0 && cgw == 1;

    int s_cur = 0, s_nxt = 1 % STAGES, s_fill = STAGES - 1;
    f32x2 V[16];
    if constexpr (PIPE) {
        f32x2 d[4][4], t0[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) read_row(smem, r, d);
#pragma unroll
        for (int q = 0; q < 4; ++q) bt_cols(d, t0, q);
#pragma unroll
        for (int r = 0; r < 4; ++r) b_row(t0, V, r);
    }
    for (int c = 0; c < nchunks; ++c) {
        const int cn = c + STAGES - 1 < nchunks ? c + STAGES - 1 : nchunks - 1;
        if (!(NESR_ABL & 8)) dma_weights(cn, s_fill);    // slot s_fill was last read in iteration c-1 (barrier passed)
        if (!(NESR_ABL & 4)) load_chunk(cn);
        const char* st = smem + s_cur * STAGE_BYTES;
        const char* sn = smem + s_nxt * STAGE_BYTES;
        f32x2 dn[4][4], tn[4][4], Vn[16];
        if constexpr (!PIPE) {
#pragma unroll
            for (int r = 0; r < 4; ++r) read_row(st, r, dn);
#pragma unroll
            for (int q = 0; q < 4; ++q) bt_cols(dn, tn, q);
#pragma unroll
            for (int r = 0; r < 4; ++r) b_row(tn, V, r);
        }
        auto U = [&](int p, int t) -> f32x2 { return *reinterpret_cast<const f32x2*>(st + w0 + (p * NT * CW + t) * 512); };
        f32x2 ua[2], ub[2];
        if constexpr (NT == 2) { ua[0] = U(0, 0); ub[0] = U(0, 1); } else { ua[0] = U(0, 0); ub[0] = U(1, 0); }
        __builtin_amdgcn_sched_barrier(0);
        if (!idle) {
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            const int cur = g & 1, nxt = cur ^ 1;
            if (g + 1 < GROUPS) {
                if constexpr (NT == 2) { ua[nxt] = U(g + 1, 0); ub[nxt] = U(g + 1, 1); }
                else { ua[nxt] = U(2 * g + 2, 0); ub[nxt] = U(2 * g + 3, 0); }
            }
            if constexpr (PIPE && !(NESR_ABL & 2)) {   // a slice of the next chunk's input transform (GROUPS == 8 here)
                if (g < 4) read_row(sn, g, dn);
                else if (g == 4) { bt_cols(dn, tn, 0); bt_cols(dn, tn, 1); }
                else if (g == 5) { bt_cols(dn, tn, 2); bt_cols(dn, tn, 3); }
                else if (g == 6) { b_row(tn, Vn, 0); b_row(tn, Vn, 1); }
                else { b_row(tn, Vn, 2); b_row(tn, Vn, 3); }
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (NESR_ABL & 1) {
                acc[0][g][0] += ua[cur][0] + ub[cur][0] + V[g][0];
            } else if constexpr (NT == 2) {
                acc[0][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[cur][0], V[g][0], acc[0][g], 0, 0, 0);
                acc[1][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(ub[cur][0], V[g][0], acc[1][g], 0, 0, 0);
                acc[0][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[cur][1], V[g][1], acc[0][g], 0, 0, 0);
                acc[1][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(ub[cur][1], V[g][1], acc[1][g], 0, 0, 0);
            } else {
                const int p = 2 * g;
                acc[0][p] = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[cur][0], V[p][0], acc[0][p], 0, 0, 0);
                acc[0][p + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ub[cur][0], V[p + 1][0], acc[0][p + 1], 0, 0, 0);
                acc[0][p] = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[cur][1], V[p][1], acc[0][p], 0, 0, 0);
                acc[0][p + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ub[cur][1], V[p + 1][1], acc[0][p + 1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        if constexpr (PIPE && !(NESR_ABL & 2)) {
            if (!idle) {
#pragma unroll
                for (int p = 0; p < 16; ++p) V[p] = Vn[p];
            }
        }
        // (carrying the staged chunk in registers across a whole iteration, so that nothing issued in an
        // iteration is waited for in it, measured no gain: the loop is not latency-bound)
        if (!(NESR_ABL & 4)) store_chunk(s_fill);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's weight DMA (issued a whole chunk ago) has landed
        __syncthreads();
        s_cur = s_cur == STAGES - 1 ? 0 : s_cur + 1;
        s_nxt = s_nxt == STAGES - 1 ? 0 : s_nxt + 1;
        s_fill = s_fill == STAGES - 1 ? 0 : s_fill + 1;
    }

    // ---- output transform + fused epilogue.  C/D map of 16x16x4: col = lane&15 (tile), rows 4*(lane>>4)+r
    // (couts); this lane: tile (tr, tc), channels 16*(NT*cgw + t) + 4*kq .. +3.
    if (idle) return;
    if (NESR_ABL & 16) {
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int p = 0; p < 16; ++p) sum += acc[t][p][0] + acc[t][p][1] + acc[t][p][2] + acc[t][p][3];
        if (sum == 12345.678f) static_cast<float*>(a.out)[0] = sum;
        return;
    }
    const float* res1 = static_cast<const float*>(a.res1);
    const float* res2 = static_cast<const float*>(a.res2);
    float* out = static_cast<float*>(a.out);
    float* out2 = static_cast<float*>(a.out2);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int co = 16 * (NT * cgw + t) + 4 * kq;
        const f32x4 bias = bias_r[t];
        // A^T M A with A^T = [1 1 1 0; 0 1 -1 -1]; positions p = 4*row + col
        f32x4 m0[4], m1[4];   // rows of A^T M (2 x 4)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            m0[q] = acc[t][0 + q] + acc[t][4 + q] + acc[t][8 + q];
            m1[q] = acc[t][4 + q] - acc[t][8 + q] - acc[t][12 + q];
        }
        f32x4 y[2][2];
        y[0][0] = m0[0] + m0[1] + m0[2];
        y[0][1] = m0[1] - m0[2] - m0[3];
        y[1][0] = m1[0] + m1[1] + m1[2];
        y[1][1] = m1[1] - m1[2] - m1[3];
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int Y = y0 + 2 * tr + dy, X = x0 + 2 * tc + dx;
                const bool valid = Y < a.h && X < a.w_;
                const size_t pix = ((size_t)n * a.h + (valid ? Y : 0)) * a.w_ + (valid ? X : 0);
                auto at = [&](const Map& mp, int ch) -> size_t { return (size_t)(ch / 8) * mp.chunk + pix * mp.pix + (ch % 8); };
                f32x4 v = y[dy][dx] + bias;
                if (a.lrelu) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = v[q] > 0.f ? v[q] : v[q] * 0.2f;
                }
                if (res1) {
                    const f32x4 r = *reinterpret_cast<const f32x4*>(res1 + at(a.res1_map, co));
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = __fadd_rn(__fmul_rn(v[q], a.s1), r[q]);
                }
                if (res2) {
                    const f32x4 r = *reinterpret_cast<const f32x4*>(res2 + at(a.res2_map, co));
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = __fadd_rn(__fmul_rn(v[q], a.s2), r[q]);
                }
                if (valid) {
                    if (out) store_fm(out + at(a.out_map, a.out_coff + co), v);
                    if (out2) store_fm(out2 + at(a.out2_map, co), v);
                    if (a.cout_real > 0 && co == 0) {
                        // conv_last: channels 0..cout_real-1 (<= 4) are this lane's run (cgw = 0, kq = 0)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            if (q >= a.cout_real) break;
                            if (a.out_nchw) a.out_nchw[(((size_t)n * a.cout_real + q) * a.h + Y) * a.w_ + X] = v[q];
                            if (a.out_u8) {
                                float qv = fminf(fmaxf(v[q], 0.f), 1.f) * 255.0f;
                                qv = a.u8_round ? rintf(qv) : truncf(qv);
                                const int ch = a.u8_flip ? (a.cout_real - 1 - q) : q;
                                a.out_u8[pix * a.cout_real + ch] = (uint8_t)qv;
                            }
                        }
                    }
                }
            }
    }
}

template <int NT, int STAGES, int CW>
hipError_t launch_wino(const ConvArgs& a, hipStream_t s) {
    constexpr size_t shm = STAGES * (size_t)(IN_BYTES + 16 * (16 * NT * CW) * 32);
    static unsigned long long attr_done = 0;
    {
        const hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&conv3x3_wino_f32_kernel<NT, STAGES, CW>), shm, attr_done);
        if (e != hipSuccess) return e;
    }
    const int tiles = ((a.w_ + TW - 1) / TW) * ((a.h + TH - 1) / TH) * a.n;
    if (tiles <= 0) return hipSuccess;
    hipLaunchKernelGGL((conv3x3_wino_f32_kernel<NT, STAGES, CW>), dim3(tiles), dim3(128 * CW), shm, s, a);
    return hipGetLastError();
}

}  // namespace

size_t packed_weight_elems_wino_f32(int cin_p, int coutp) { return (size_t)cin_p * 16 * coutp; }

// OIHW -> U = G g G^T per (cout, cin), packed [chunk = ci/8][p][cout group = o/16][kq = (ci%8)/2][o%16][ci%2].
void pack_weights_wino_f32(const float* oihw, int cout, int cin, int cin_p, int coutp, float* dst) {
    const size_t total = packed_weight_elems_wino_f32(cin_p, coutp);
    for (size_t i = 0; i < total; ++i) dst[i] = 0.f;
    static const float G[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
    const int groups = coutp / 16;
    for (int o = 0; o < cout; ++o)
        for (int ci = 0; ci < cin; ++ci) {
            const float* g = oihw + ((size_t)o * cin + ci) * 9;
            float tmp[4][3];
            for (int r = 0; r < 4; ++r)
                for (int q = 0; q < 3; ++q) tmp[r][q] = G[r][0] * g[0 * 3 + q] + G[r][1] * g[1 * 3 + q] + G[r][2] * g[2 * 3 + q];
            for (int r = 0; r < 4; ++r)
                for (int q = 0; q < 4; ++q) {
                    const float u = tmp[r][0] * G[q][0] + tmp[r][1] * G[q][1] + tmp[r][2] * G[q][2];
                    const int p = 4 * r + q, c = ci / 8, kq = (ci % 8) / 2, sidx = ci % 2;
                    const size_t idx = (((((size_t)c * 16 + p) * groups + o / 16) * 4 + kq) * 16 + o % 16) * 2 + sidx;
                    dst[idx] = u;
                }
        }
}

hipError_t launch_conv3x3_wino_f32(const ConvArgs& a, hipStream_t s) {
    if (a.y_lo || a.y_hi) return hipErrorInvalidValue;   // row ranges: conv3x3_f16x2_kernel only
    if (a.cin % 8) return hipErrorInvalidValue;
    if ((a.out_nchw || a.out_u8) && (a.coutp != 32 || a.cout_real < 1 || a.cout_real > 4)) return hipErrorInvalidValue;
    // NESR_WINO_NT1=wide: Cout=32 layers as 2-wave workgroups, each wave owning both 16-channel groups
    // (no redundant input transform between cout-group waves, but one wave per SIMD on a 256x256 frame):
    // measured 29 % slower in-process, kept for A/B
    static const bool wide = [] { const char* e = getenv("NESR_WINO_NT1"); return e && e[0] == 'w'; }();
    if (a.coutp == 64) return launch_wino<2, 2, 2>(a, s);   // 2 x 38.5 KB -> two workgroups per CU
    if (a.coutp == 32) {
        if (wide && !(a.out_nchw || a.out_u8)) return launch_wino<2, 3, 1>(a, s);
        return launch_wino<1, 3, 2>(a, s);                   // 3 x 22.1 KB -> two workgroups per CU
    }
    return hipErrorInvalidValue;
}

}  // namespace nesr
