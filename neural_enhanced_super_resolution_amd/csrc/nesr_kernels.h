// Internal launcher interface between the C-ABI layer (nesr_api.cpp) and the HIP kernels.
// Not part of the public ABI (that is include/nesr_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

namespace nesr {

// Activation addressing.  Channels are grouped in K-groups of KG channels (8 for f32, 16 for
// bf16 = 32 bytes, one MFMA K-chunk); element (pixel p, channel c) of a feature map lives at
//     base + (c / KG) * chunk + p * pix + (c % KG)          (in elements)
//   NHWC (f32 path)            : pix = C_total, chunk = KG      -> base + p*C + c
//   channel-blocked (bf16 path): pix = KG,      chunk = P*KG    -> [C/KG][P pixels][KG]
// The blocked form makes one K-chunk of neighbouring pixels contiguous in HBM, so the staging
// loads / LDS-DMA read whole 128-byte lines (NHWC gives every lane its own line: the bf16 kernels
// were L1-access-bound, profiles/r01/bf16_tcp_bound.txt).  SURVEY.md section 7's "one 192-channel
// buffer whose channel slices are x0|x1|x2|x3|x4" holds in both forms: a slice is a channel range.
struct Map {
    int pix;          // elements between consecutive pixels
    long long chunk;  // elements between consecutive K-groups
};

// More than 64 KiB of dynamic LDS needs an opt-in per kernel AND per device: a process may hold contexts on
// several GPUs (nesr_create's device_id), and the attribute belongs to the device's copy of the code object.
// `done` is the launcher's static bitmask of devices already set (a benign race: the call is idempotent).
inline hipError_t ensure_dynamic_lds(const void* kernel, size_t bytes, unsigned long long& done) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 64 && ((done >> dev) & 1ull)) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess && dev < 64) done |= 1ull << dev;
    return e;
}

constexpr int RAG_MAX = 64;   // images per ragged batch

// One fused 3x3 / stride 1 / zero-pad 1 convolution (torch.cat-free dense block: conv k reads
// channels [0, cin) and writes channels [out_coff, out_coff + coutp) of the same buffer).
struct ConvArgs {
    // input: channels [0, cin) are read
    const void* in;
    Map in_map;
    int in_h, in_w;        // stored (source) height/width
    int up;                // 1: the logical input is the nearest x2 upsample of the stored one
                           //    (F.interpolate(scale_factor=2, mode='nearest') folded into addressing)
    int cin;               // padded to a multiple of KG
    // packed weights / bias (device)
    const void* w;
    const float* bias;     // [coutp] f32
    int coutp;             // padded output channels: 32 or 64
    // geometry of the convolution (logical input == output spatial size)
    int n, h, w_;
    // output rows [y_lo, y_hi) only (0, 0 = all of [0, h)): the banded multi-GPU mode computes the rows its neighbours
    // wait for first.  The input is still the whole image: rows outside [0, h) are the zero padding, rows outside
    // [y_lo, y_hi) are real data.  conv3x3_f16x2_kernel only (the other launchers reject a row range).
    int y_lo, y_hi;
    // feature-map output (may be null when out_nchw / out_u8 is used)
    void* out;
    Map out_map;
    int out_coff;
    void* out2;            // optional second copy of the same values (channels [0,coutp))
    Map out2_map;
    // epilogue: v = acc + bias; if lrelu v = leaky(v, 0.2); if res1 v = v*s1 + res1; if res2 v = v*s2 + res2
    int lrelu;
    const void* res1; Map res1_map; float s1;
    const void* res2; Map res2_map; float s2;
    // planar f32 output [n][cout_real][h][w] (conv_last feeding RRDBNet.forward's NCHW result)
    float* out_nchw;
    int cout_real;
    // fused image output: u8 HWC [h][w][3], clamp(0,1) * 255, optional channel flip
    uint8_t* out_u8;
    int u8_flip, u8_round;
    // >= 16 bytes of device zeros: LDS-DMA source for out-of-image pixels (bf16 large-tile kernel)
    const void* zeros;
    // f16-pair path: sticky device word set when a value does not fit the (hi, lo) pair (|x| > 65504 or not
    // finite); conv_last reads it and writes NaN instead of a saturated image.  May be null.
    unsigned* status;
    // host-side hint, not read by kernels: other contexts' launches share the device with this one (frames or tile
    // groups in flight on several streams) -- prefer kernel forms whose workgroups leave room on a CU
    int shared_device;
    // Images of different sizes in one batch (nesr_forward_ragged: the tiles of a frame, whose edge tiles are smaller):
    // image i occupies the top-left (rag_h[i] << rag_shift) x (rag_w[i] << rag_shift) pixels of its h x w slot, the rest
    // of the slot is neither read (it counts as the zero padding) nor written.  rag_n = 0: every image is h x w.  The
    // sizes travel in the kernel arguments (no device table, no copy to order).  conv3x3_bf16_xl_kernel only.
    int rag_n, rag_shift;
    // host-side hint: choose the kernel by arithmetic only, never by image size (nesr_set_size_independent): a tile's bits
    // must not depend on whether it was evaluated alone or inside a ragged batch
    int size_independent;
    unsigned short rag_h[RAG_MAX], rag_w[RAG_MAX];
};

// f32 path: v_mfma_f32_32x32x2_f32 implicit GEMM (conv3x3_f32.hip)
hipError_t launch_conv3x3_f32(const ConvArgs& a, hipStream_t s);
// host-side weight repack for the f32 kernel: OIHW f32 -> [cin/8][tap][half][coutp][4]
size_t packed_weight_elems_f32(int cin_p, int coutp);
void pack_weights_f32(const float* oihw, int cout, int cin, int cin_p, int coutp, float* dst);

// f32 Winograd F(2x2,3x3) path: v_mfma_f32_16x16x4_f32, 2.25x less matrix work (conv3x3_wino_f32.hip);
// feature-map outputs only (conv_last keeps the direct kernel)
hipError_t launch_conv3x3_wino_f32(const ConvArgs& a, hipStream_t s);
size_t packed_weight_elems_wino_f32(int cin_p, int coutp);
void pack_weights_wino_f32(const float* oihw, int cout, int cin, int cin_p, int coutp, float* dst);

// bf16 path: v_mfma_f32_32x32x16_bf16 implicit GEMM (conv3x3_bf16.hip)
hipError_t launch_conv3x3_bf16(const ConvArgs& a, hipStream_t s);       // picks the variant by frame size
hipError_t launch_conv3x3_bf16_xl(const ConvArgs& a, hipStream_t s);    // conv3x3_bf16.hip: 32x32-px tiles, 3-deep LDS-DMA ring
size_t packed_weight_elems_bf16(int cin_p, int coutp);
void pack_weights_bf16(const float* oihw, int cout, int cin, int cin_p, int coutp, uint16_t* dst);

// f32 path on the f16 matrix cores (conv3x3_f16x2.hip): operands as (hi, lo) half pairs, x = hi + lo * 2^-11,
// three MFMAs per product, f32 accumulation.  Activations: channel-blocked [C/16][pixels][16 hi | 16 lo]
// (Map in 2-byte units: pix = 32, chunk = pixels * 32).
hipError_t launch_conv3x3_f16x2(const ConvArgs& a, hipStream_t s);
size_t packed_weight_elems_f16x2(int cin_p, int coutp);   // in 2-byte units
void pack_weights_f16x2(const float* oihw, int cout, int cin, int cin_p, int coutp, uint16_t* dst);

// One residual dense block (conv1..conv5) per launch, f16-pair form, for frames whose 8x32-pixel tiles all get their
// own resident workgroup (conv3x3_f16x2.hip, rdb_f16x2_kernel): tiles = rdb_f16x2_tiles(n, h, w) <= compute units.
struct RdbLaunch {
    const void* cur;          // the block's 192-channel buffer (f16-pair layout: [12 chunks][pixels][64 B])
    long long chunk_bytes;
    void* out;                // conv5 destination buffer (its channels 0..63)
    const void* res2;         // RRDB input for the second residual, or null
    float s1, s2;
    const void* w[5];
    const float* bias[5];
    int n, h, w_;
    unsigned* progress;       // [tiles] device words, monotonic across launches
    unsigned epoch;           // strictly increasing by >= 8 per launch
    unsigned* abort_flag;
    unsigned* status;
    unsigned long long timeout_ticks;   // 100 MHz ticks a neighbour wait may take (0: 200 ms)
    int debug_drop;           // test hook: this many workgroups of the launch never start
};
int rdb_f16x2_tiles(int n, int h, int w);
hipError_t launch_rdb_f16x2(const RdbLaunch& r, hipStream_t s);

// bf16 dense block with the working set resident in LDS (rdb_bf16_strip.hip): a workgroup owns a 16-column strip of an
// image and sweeps it top to bottom in positions of 12 rows; x1..x4 never leave the CU except for the strips' edge columns.
constexpr int STRIP_BH = 12, STRIP_BW = 16;
constexpr int STRIP_XCH_BYTES = 2 * 2 * 4 * STRIP_BH * 128;   // edge-column mailboxes of one strip
struct StripSchedule {
    std::vector<int> items;      // 8 ints per item: image | mailbox group << 16, strip, image height, image width, first position, end
                                 // position, first output row stored, end row (a row segment of the strip; whole images: 0, npos, 0, h)
    std::vector<int> wg_first;   // [grid + 1]
    int grid = 0, smax = 0, makespan = -1;
    int nvimg = 0;               // mailbox groups (image segments): the exchange buffer holds nvimg * smax * STRIP_XCH_BYTES
    double efficiency = 0.0;     // strip positions of work / (compute units x makespan)
};
// n images of hw[2i] x hw[2i+1] internal pixels -> the packing (makespan < 0: the kernel does not apply).  seg_len: positions per row
// segment at most (0: chosen by the packer, < 0: images are never cut)
StripSchedule strip_schedule(int n, const int* hw, int cus, int seg_len = 0);
size_t strip_weight_bytes();
void pack_strip_weights(const float* const w[5], uint16_t* dst);   // conv1..conv5 OIHW f32 of one dense block
struct StripLaunch {
    const void* cur; long long chunk_bytes; void* out; const void* res2;
    float s1, s2;
    const void* wstream; const float* bias;       // device: pack_strip_weights image, [192] f32 (conv1..4: 32 each, conv5: 64)
    int H, W;                                     // slot geometry of the batch buffers
    const void* items; const int* wg_first;       // device copies of the schedule
    int grid, smax;
    void* xch;                                    // nvimg * smax * STRIP_XCH_BYTES
    unsigned epoch;                               // strictly increasing by >= 2048 per launch
    unsigned* abort_flag;
    unsigned long long timeout_ticks;
    int debug_drop;                               // test hook: this many workgroups of the launch never start
};
hipError_t launch_rdb_bf16_strip(const StripLaunch& r, hipStream_t s);

// Persistent trunk (conv3x3_mfma.hip): all dense-block convs of the 23 RRDBs in ONE cooperative
// launch.  Workgroups keep their tiles from layer to layer and synchronise with their 8
// neighbouring tiles only (per-tile progress counters, agent-scope release/acquire), instead of 345
// kernel boundaries at which the whole chip drains and refills.
struct TrunkLayer {
    int in_buf, out_buf, res1_buf, res2_buf;   // 0..2 = the rotating 192-channel buffers P,Q,R; -1 = none
    int cin, coutp, out_coff, lrelu;
    float s1, s2;
    const void* w;
    const float* bias;
};
struct TrunkArgs {
    const TrunkLayer* layers;   // device array
    int nlayers;
    void* buf[3];
    Map map;                    // addressing of P,Q,R (192 channels each)
    int n, h, w;
    unsigned* progress;         // [tiles] layers completed per tile, zeroed before the launch
    unsigned* abort_flag;       // set by a workgroup whose bounded wait timed out
    const void* zeros;
};
// returns the workgroup count it would launch (0 = the persistent form does not apply)
hipError_t launch_trunk_persist(const TrunkArgs& t, bool bf16, hipStream_t s);

// input packers (pack.hip): NCHW f32 (+ pixel_unshuffle) -> NHWC with `cp` channels (zero padded)
struct PackArgs {
    const void* src;     // f32 NCHW [n][c][hin][win]   or u8 HWC [hin][win][3] (n == 1)
    int src_u8;          // 1: u8 HWC source, value/255, optional channel flip
    int flip;
    int n, c, hin, win;
    int unshuffle;       // 1, 2 or 4
    void* dst;           // feature map with cp channels (zero padded), addressed through dst_map
    Map dst_map;
    int cp;
    int bf16;            // destination layout: 0 f32 NHWC (KG = 8), 1 bf16 blocked (KG = 16), 2 f16 hi|lo blocked (KG = 16)
    unsigned* status;    // as ConvArgs::status (layout 2 only); may be null
};
hipError_t launch_pack_input(const PackArgs& a, hipStream_t s);
hipError_t launch_status_latch(unsigned* status, hipStream_t s);   // range word of an unchecked earlier forward: word 0 -> word 3

// all tiles of a frame at once (pack.hip): u8 HWC frame -> float NCHW tile slots (cut), float NCHW tile outputs -> u8 (paste)
constexpr int TILE_IO_MAX = 64;
struct TileIo {
    uint8_t* frame;       // cut: the frame (read); paste: base of the destination
    int frame_w;          // cut: pixels per frame row
    float* tiles;         // [n][3][Hs][Ws]
    int Hs, Ws;
    int flip, round;
    int desc[8 * TILE_IO_MAX];   // per tile: cut {y0, x0, h, w}; paste {crop y, crop x, h, w, row pitch (bytes), offset lo, offset hi}
};
hipError_t launch_cut_tiles(const TileIo& t, int n, int maxh, int maxw, hipStream_t s);
hipError_t launch_paste_tiles(const TileIo& t, int n, int maxh, int maxw, hipStream_t s);

// cv2.fastNlMeansDenoising (template 7, search 21) on [C][H][W] u8 planes taken as one C-channel image (imgproc.hip)
hipError_t launch_nl_means(const uint8_t* src, int C, int H, int W, const int* lut, int nbins, int shift, uint8_t* dst, hipStream_t s);

// cv2 CLAHE on one u8 plane (imgproc.hip): th x tw = tile size of the (virtually) padded image, clip = max(int(clipLimit * th * tw / 256), 1),
// scale = 255 / (th * tw), inv_th / inv_tw = 1 / th, 1 / tw as floats; lut: gx * gy * 256 floats of device scratch
hipError_t launch_clahe(const uint8_t* src, int h, int w, int gx, int gy, int th, int tw, int clip, float scale, float inv_th, float inv_tw, float* lut,
                        uint8_t* dst, hipStream_t s);

// feature map (channels [0,c)) -> planar f32 NCHW; used by the single-layer test hook
hipError_t launch_nhwc_to_nchw(const void* src, int kind /* as PackArgs::bf16 */, Map map, int n, int c, int h, int w, float* dst, hipStream_t s);

}  // namespace nesr
