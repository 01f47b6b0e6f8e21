/*
 * nesr_hip.h -- C ABI of libnesr_hip.so: the MI355X (gfx950) Real-ESRGAN / RRDBNet inference path.
 *
 * The reference (gddickinson/neural_enhanced_super_resolution) is pure Python and has no FFI of
 * its own; the arithmetic of this path lives in the un-vendored pip packages basicsr (RRDBNet)
 * and realesrgan (RealESRGANer).  Each entry point below names the reference interface it
 * stands behind (paths relative to the reference root).  The Python host side
 * (neural_enhanced_super_resolution_amd/) binds these with ctypes and presents the
 * RRDBNet / RealESRGANer objects the reference constructs at nesr/nesr.py:216-229 and
 * standalone/direct_esrgan.py:104-127.
 *
 * Conventions: every function returns 0 on success and a negative code on failure; the message
 * is available from nesr_last_error() (thread-local).  No exception crosses the ABI.  Device
 * pointers are caller-owned (e.g. torch tensors' data_ptr()); the context owns packed weights
 * and its workspace.  `stream` is a hipStream_t passed as void* (NULL = default stream); all
 * device work is enqueued on it asynchronously.  One in-flight call per context
 * (the reference calls from one thread at a time: main thread or one QThread,
 * nesr/gui/app.py:72,1732-1749); hipSetDevice is done on entry so any thread may call.
 */
#ifndef NESR_HIP_H
#define NESR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nesr_ctx nesr_ctx;

/* NESR_DTYPE_F32_WINOGRAD: f32 storage and f32 matrix-core arithmetic, the feature-map 3x3 convs
 * evaluated by Winograd F(2x2,3x3) (2.25x less matrix work; a few ulps per layer away from the
 * direct form, far inside the 1e-3 output tolerance). */
/* NESR_DTYPE_F32_SPLIT: f32 in, f32 out, f32 accumulation; every conv operand is carried as a pair of halves,
 * x = hi + lo * 2^-11 (hi = f16(x), lo = f16((x - hi) * 2^11)), and each product is three f16 MFMAs
 * (hi*hi + hi*lo + lo*hi), see conv3x3_f16x2.hip.  The pair holds 22 significant bits for 6.1e-5 <= |x| <= 65504
 * and an absolute 2^-35 below; a value outside +-65504 or a non-finite one does NOT fit: nesr_finalize_weights
 * rejects such a weight (NESR_ERR_RANGE), and such an input or activation raises the context's range flag --
 * the float output of THAT forward is then NaN and nesr_check_range / nesr_check_status return NESR_ERR_RANGE.  The flag is
 * scoped to one forward: a caller of nesr_forward that never asks does not get NaN for later, valid frames; the unreported
 * condition is latched and returned (once) by the next nesr_check_range.  Callers of nesr_forward MUST call nesr_check_range
 * (or nesr_check_status) before trusting an output.
 * Whole-network max abs error vs an f64 evaluation 3e-6 on the bench weights (plain f32: 1e-6). */
enum { NESR_DTYPE_F32 = 0, NESR_DTYPE_BF16 = 1, NESR_DTYPE_F32_WINOGRAD = 2, NESR_DTYPE_F32_SPLIT = 3 };
enum { NESR_ROUND_TRUNC = 0, NESR_ROUND_NEAREST = 1 };

enum {
    NESR_OK = 0,
    NESR_ERR_ARG = -1,      /* bad argument / unsupported shape */
    NESR_ERR_HIP = -2,      /* a HIP runtime call failed */
    NESR_ERR_STATE = -3,    /* weights missing / not finalized */
    NESR_ERR_NOMEM = -4,
    NESR_ERR_RANGE = -5     /* f16-pair fp32 form: a weight, input or activation was non-finite or beyond +-65504 */
};

/*
 * Replaces: RRDBNet.__init__ (basicsr archs/rrdbnet_arch.py) as called at nesr/nesr.py:216,
 * standalone/direct_esrgan.py:104, standalone/superres_project.py:69.
 *   conv_first_in_ch : input channels of conv_first (3 for x4plus; 12 for x2plus and for the
 *                      nesr 12-channel quirk)
 *   unshuffle        : 0 = none (upstream scale=4), 2 = pixel_unshuffle(2) folded into the input
 *                      load (upstream scale=2), 4 = pixel_unshuffle(4) (upstream scale=1).
 *                      forward() then expects C = conv_first_in_ch / unshuffle^2 input channels.
 *   dtype            : one of NESR_DTYPE_* -- storage and arithmetic of activations/weights (accumulation is always f32).
 */
int nesr_create(nesr_ctx** out, int device_id, int conv_first_in_ch, int unshuffle, int num_feat,
                int num_block, int num_grow_ch, int num_out_ch, int dtype);

/*
 * Replaces: model.load_state_dict(loadnet[keyname], strict=True) in RealESRGANer.__init__
 * (realesrgan utils.py), reached from nesr/nesr.py:220-229, standalone/direct_esrgan.py:118-127.
 * `key` is the upstream state_dict name ("conv_first.weight", "body.0.rdb1.conv1.bias", ...),
 * `data` a HOST pointer to f32 (weights OIHW, bias [O]); it is repacked and copied, the caller
 * keeps ownership.  Unknown key or shape mismatch -> NESR_ERR_ARG (strict=True semantics).
 */
int nesr_load_weight(nesr_ctx* ctx, const char* key, const float* data, const int64_t* shape, int ndim);

/* strict=True: fails with NESR_ERR_STATE (message lists the first missing keys) unless every
 * tensor of the architecture was loaded. Uploads packed weights to the device. */
int nesr_finalize_weights(nesr_ctx* ctx);

/* Number of state_dict tensors the architecture expects (702 for num_block=23). */
int nesr_num_tensors(const nesr_ctx* ctx);

/*
 * Replaces: RRDBNet.forward -- `self.model(self.img)` in RealESRGANer.process/tile_process and
 * `model(img_12ch)` at nesr/nesr.py:891,935.
 *   x_dev : NCHW f32 [N, C, H, W] contiguous device memory
 *   y_dev : NCHW f32 [N, num_out_ch, 4*H/u, 4*W/u] (u = unshuffle or 1) device memory
 * H and W must be multiples of the unshuffle factor (upstream asserts the same).
 */
int nesr_forward(nesr_ctx* ctx, const void* x_dev, int N, int C, int H, int W, void* y_dev, void* stream);

/*
 * Fused image path: replaces `img/255`, cv2 BGR<->RGB flips, HWC->CHW,
 * RRDBNet.forward, clamp(0,1), x255 and the quantiser of RealESRGANer.enhance
 * (round_mode NESR_ROUND_NEAREST, flip_rgb 1) or of nesr/nesr.py:851-857,894-901
 * (NESR_ROUND_TRUNC, flip_rgb 0).
 *   in_hwc_dev  : u8 [H, W, 3] device memory;  out_hwc_dev : u8 [4H/u, 4W/u, 3] device memory.
 * Requires conv_first_in_ch / unshuffle^2 == 3 and num_out_ch == 3.
 */
int nesr_forward_u8(nesr_ctx* ctx, const uint8_t* in_hwc_dev, int H, int W, uint8_t* out_hwc_dev,
                    int flip_rgb, int round_mode, void* stream);

/*
 * Forward pass on N images of DIFFERENT sizes in one batch: the tiles of one frame as realesrgan's tile_process cuts them
 * (`for y in range(tiles_y): for x in range(tiles_x): ... self.model(input_tile)`, called from
 * standalone/direct_esrgan.py:148 with tile=512, tile_pad=10 -- interior tiles 532 x 532, edge and corner tiles smaller).
 * Image i lies in the top-left hw[2i] x hw[2i+1] pixels of slot i of x_dev ([N, C, H, W] f32; the rest of a slot is
 * ignored) and its output in the top-left of slot i of y_dev ([N, num_out_ch, 4H/u, 4W/u]; the rest is not written).
 * Each image is evaluated as an image of its own -- its borders are the zero padding of every conv -- with the values
 * nesr_forward gives it alone (see nesr_set_size_independent).  hw: host array of N (height, width) pairs, multiples of
 * the unshuffle factor, 1 <= N <= 64.  compute dtype bf16 only (NESR_ERR_ARG otherwise: the f32 forms batch equal shapes).
 */
int nesr_forward_ragged(nesr_ctx* ctx, const void* x_dev, int N, int C, int H, int W, const int* hw, void* y_dev, void* stream);

/* Choose kernels by arithmetic only, never by image size or batch composition, so that an image's values are the same bits
 * alone, in an equal-shape batch and in a ragged batch -- and on any rank of a sharded frame, whatever its share.  bf16: the dense
 * blocks always run as the LDS-resident kernel (rdb_bf16_strip_kernel; a batch too small to fill the device is cut into row
 * segments, which does not change a bit), the other layers as the large-tile kernel also for small images.  RealESRGANer sets it
 * for a tiling wrapper.  Off by default: a plain context picks whichever form is faster for the batch at hand (the two bf16
 * forms agree to bf16 resolution, not bit for bit). */
int nesr_set_size_independent(nesr_ctx* ctx, int on);

/* Device bytes of activation workspace forward() needs for a batch of N frames of H x W input. */
size_t nesr_workspace_bytes(const nesr_ctx* ctx, int N, int H, int W);

/* Pre-allocates the workspace (forward() grows it on demand otherwise, which synchronises). */
int nesr_reserve(nesr_ctx* ctx, int N, int H, int W);

/* Batch size <= max_batch (frames of H x W input evaluated by one forward()) whose trunk launches
 * fill the device's compute units most evenly -- used by RealESRGANer.tile_process to group the
 * equal-shaped tiles of upstream's tile grid. */
int nesr_preferred_batch(const nesr_ctx* ctx, int H, int W, int max_batch);

/* Algorithmic FLOPs (2 x MACs) of one forward() on N frames of H x W input. */
double nesr_forward_flops(const nesr_ctx* ctx, int N, int H, int W);

/*
 * Banded evaluation: the exact (seamless) multi-GPU mode of SURVEY.md section 8(e).  Stands behind the same
 * reference call as nesr_forward -- `self.model(img)` on a whole frame, nesr/nesr.py:887-891 with tile=0
 * (nesr/nesr.py:224) -- when the frame is split into row bands, one per rank.  A rank holds its band plus `apron`
 * rows of its neighbours (N = 1, input [1,C,H,W] NCHW f32 including the apron rows) and runs the stages in order:
 *     nesr_band_begin                       pixel_unshuffle + conv_first
 *     nesr_band_rdb(i), i = 0 .. 3*num_block-1   the five convs of RDB i (RRDB i/3, dense block i%3)
 *     nesr_band_tail                        conv_body .. conv_last  ->  [1,num_out_ch,4h,4w] f32
 * Every 3x3 conv spoils one more row at a band edge that is not a frame edge, so before a stage the caller
 * overwrites the apron rows of the feature map the stage reads with the neighbours' band rows: buffer i%3 before
 * nesr_band_rdb(i) (5 rows), buffers 0 and 3 before nesr_band_tail (3 rows).  nesr_band_rows moves rows
 * [row0, row0+nrows) of the num_feat-channel slice of a buffer (0,1,2 = the rotating dense-block buffers, 3 = the
 * conv_first output kept for the trunk skip) to/from a contiguous staging buffer of nrows * nesr_band_row_bytes
 * bytes, in the context's own element layout (opaque: only ever handed to another rank's nesr_band_rows).
 * neural_enhanced_super_resolution_amd/banded.py is the reference-side protocol (RCCL point-to-point).
 */
/*
 * The same stages with the exchange taken off the critical path (banded.py's default protocol):
 *   nesr_band_rdb_phase(i, 0, ...)   conv1..conv4 of RDB i, and conv5 on the `edge_rows` band rows next to each apron --
 *                                     the rows the neighbours wait for (top / bottom = apron rows of this band image)
 *   nesr_band_pack_edges             those rows of the buffer RDB i wrote -> two caller-owned staging buffers (one call;
 *                                     the caller sends them, e.g. RCCL point-to-point on a side stream)
 *   nesr_band_rdb_phase(i, 1, ...)   conv5 on the band rows in between, while the edge rows travel
 *   nesr_band_unpack_aprons          the neighbours' rows -> this band image's apron rows, before RDB i + 1
 * Values are those of nesr_band_rdb (row ranges of the same kernel).  Row-range launches exist for NESR_DTYPE_F32_SPLIT;
 * for the other dtypes phase 0 runs the whole block and phase 1 nothing (same protocol, no overlap).
 */
int nesr_band_rdb_phase(nesr_ctx* ctx, int index, int phase, int top, int bottom, int edge_rows, void* hip_stream);
int nesr_band_pack_edges(nesr_ctx* ctx, int buffer, int top, int bottom, int nrows, void* top_dst, void* bottom_dst, void* hip_stream);
int nesr_band_unpack_aprons(nesr_ctx* ctx, int buffer, int top, int bottom, int nrows, const void* top_src, const void* bottom_src,
                            void* hip_stream);
int nesr_band_begin(nesr_ctx* ctx, const void* x_dev, int C, int H, int W, void* hip_stream);
int nesr_band_rdb(nesr_ctx* ctx, int index, void* hip_stream);
int nesr_band_tail(nesr_ctx* ctx, void* y_dev, void* hip_stream);
size_t nesr_band_row_bytes(const nesr_ctx* ctx);
int nesr_band_rows(nesr_ctx* ctx, int buffer, int row0, int nrows, void* staging_dev, int write, void* hip_stream);

/* Hint: forwards of this context run while other contexts of the process use the same device (several frames or
 * tile groups in flight on different streams).  Changes kernel selection only, never a value. */
int nesr_set_concurrent(nesr_ctx* ctx, int concurrent);

/*
 * The dense blocks of small f32 frames (rdb_f16x2_kernel) and of bf16 tile batches (rdb_bf16_strip_kernel) run as PERSISTENT
 * launches whose workgroups wait for one another; they need every workgroup resident, i.e. the device to themselves.  Inside a
 * process such launches of different contexts / streams are serialised per device (an event wait, no host blocking).  Against
 * another process nothing can order them: every wait is bounded by wall clock (200 ms; NESR_FUSED_TIMEOUT_MS), a workgroup
 * that gives up raises an abort word that ends all other waits of that forward at once, nesr_check_range / nesr_check_status
 * then return NESR_ERR_HIP for it, and the context switches to per-layer launches for good (f32: the same values bit for bit;
 * bf16: the per-layer kernels' values) -- re-run the frame.  nesr_set_fused(ctx, 0 | 1) makes that choice by hand;
 * nesr_fused_state returns bit 0 = persistent launches enabled, bits 1.. = forwards that gave up so far.
 * nesr_debug_fault is a TEST HOOK: the next persistent launch leaves out its last `drop_workgroups` workgroups (their
 * neighbours' waits must end in the abort word within the time limit).
 * Stands behind the same reference calls as nesr_forward (nesr/nesr.py:887-891, standalone/direct_esrgan.py:148).
 */
int nesr_set_fused(nesr_ctx* ctx, int on);
int nesr_fused_state(const nesr_ctx* ctx);
int nesr_debug_fault(nesr_ctx* ctx, int drop_workgroups);

/*
 * Timing hook for bench.py's roofline leg: when enabled, forward() brackets the dominant kernel
 * family (the dense-block 3x3 convs) with hipEvents on the caller's stream; nesr_kernel_time_ms
 * returns the accumulated elapsed ms and launch count since the last call (synchronises those
 * events).  Off by default.
 */
int nesr_set_kernel_timing(nesr_ctx* ctx, int enable);
int nesr_kernel_time_ms(nesr_ctx* ctx, double* total_ms, int64_t* launches, double* flops);

/* Waits for the device and reports deferred failures of asynchronous work (the persistent trunk
 * kernel bounds every inter-workgroup wait and sets an abort word instead of hanging; the f16-pair
 * form's range flag, see nesr_check_range). */
int nesr_check_status(nesr_ctx* ctx);

/* Range / abort check of the forwards enqueued so far on `hip_stream` (NESR_DTYPE_F32_SPLIT, and contexts whose dense blocks
 * ran as persistent launches; NESR_OK at once otherwise): waits for that stream only, returns NESR_ERR_RANGE if an input or activation did not fit the
 * (hi, lo) pair -- in the latest forward, or in an earlier one nobody asked about (the message says which; the latest output is
 * valid in the second case) -- and clears the flag.  Where the reference would hand back NaN/Inf pixels
 * (`model(img)` on diverged data, nesr/nesr.py:891) this path hands back NaN (float output) plus this error; the
 * Python wrappers call it after every device-to-host copy. */
int nesr_check_range(nesr_ctx* ctx, void* hip_stream);

void nesr_destroy(nesr_ctx* ctx);

/*
 * Tiled frames without the float canvas (SURVEY.md section 8(f) row 1 for frames larger than a tile).  Replaces, for all tiles of a
 * frame at once, RealESRGANer.enhance's `img.astype(float32) / 255`, BGR->RGB and tile_process's
 * `input_tile = self.img[:, :, y0:y1, x0:x1]` (nesr_cut_tiles_u8), and tile_process's paste of every tile's un-padded centre followed
 * by enhance's clamp(0, 1), RGB->BGR, x255, round (nesr_paste_tiles_u8); called from standalone/direct_esrgan.py:148 with
 * tile=512, tile_pad=10.
 *   through_fp16 : the values pass through fp16 once: upstream's RealESRGANer(half=True) hands the network an fp16 tensor
 *             (`self.img = self.img.half()`) and gets one back
 *   windows : host array, n x (y0, x0, h, w): tile i = frame[y0:y0+h, x0:x0+w] into the top-left of slot i of tiles_nchw_dev
 *             ([n, 3, Hs, Ws] f32; the rest of a slot is zeroed) -- the layout nesr_forward_ragged takes
 *   desc    : host array, n x (crop y, crop x, h, w, destination byte offset, destination row pitch in bytes): the h x w pixels
 *             at (crop y, crop x) of tile i's output (slot i of tiles_nchw_dev, [n, 3, Hs, Ws] f32) go to dst_dev + offset as u8 HWC
 *             rows `pitch` bytes apart -- a window of the frame's output canvas, or a packed per-tile buffer (multi-GPU gather)
 * 1 <= n <= 64.
 */
int nesr_cut_tiles_u8(int device_id, const uint8_t* frame_hwc_dev, int H, int W, int flip_rgb, int through_fp16, const int* windows, int n, int Hs,
                      int Ws, float* tiles_nchw_dev, void* hip_stream);
int nesr_paste_tiles_u8(int device_id, const float* tiles_nchw_dev, int n, int Hs, int Ws, const int64_t* desc, uint8_t* dst_dev, size_t dst_bytes,
                        int flip_rgb, int round_mode, int through_fp16, void* hip_stream);

/*
 * Sharded frames (SURVEY.md section 8(b), 8(e) mode 1): one process per GPU, RCCL point to point over xGMI, no collective.  Stands
 * behind `upscaler.enhance(img)` of standalone/direct_esrgan.py:148 (RealESRGANer(tile=512, tile_pad=10), :118-127) when one frame is
 * evaluated by several GPUs: the tiles of upstream's grid are independent network evaluations, dealt to the ranks in contiguous
 * runs balanced by padded area; the uint8 frame is row-scattered (rank r holds rows [r H / N, (r + 1) H / N)) and a rank fetches only
 * the rows its tiles read beyond its band; quantised tile centres are gathered on rank 0.
 *   nesr_comm_unique_id : rank 0 fills 128 bytes the host distributes by its own means (the launcher's rendezvous)
 *   nesr_comm_init      : every rank, with its rank / the world size / those 128 bytes (ncclCommInitRank on the context's device);
 *                         librccl.so is loaded on the first of these calls, never before
 *   nesr_forward_sharded_u8 : band_dev = this rank's rows of the u8 HWC BGR frame (device memory); out_dev (rank 0 only) = the
 *                         [H s, W s, 3] u8 BGR result; enqueued on hip_stream.  Without nesr_comm_init it is the one-rank case.
 *                         through_fp16 as in nesr_cut_tiles_u8 / nesr_paste_tiles_u8.  bf16 contexts.
 *   nesr_shard_plan     : the plan by itself (host only, no device): tiles as 13 ints each (input window y0 y1 x0 x1, output window,
 *                         crop inside the tile's output, owner rank) and the row moves (src, dst, row lo, row hi); counts are always
 *                         returned, the arrays are filled when they are large enough.  `scale` = output / input size.
 */
int nesr_comm_unique_id(void* id128);
int nesr_comm_init(nesr_ctx* ctx, int rank, int nranks, const void* id128);
int nesr_comm_destroy(nesr_ctx* ctx);
int nesr_forward_sharded_u8(nesr_ctx* ctx, const uint8_t* band_dev, int H, int W, int tile, int tile_pad, int through_fp16, uint8_t* out_dev,
                            void* hip_stream);
int nesr_shard_plan(int H, int W, int scale, int tile, int tile_pad, int nranks, int* tiles13, int cap_tiles, int* ntiles, int* moves4,
                    int cap_moves, int* nmoves);

/*
 * the non-local means inside `cv2.fastNlMeansDenoisingColored(image, None, h, h, 7, 21)` of
 * SuperResolutionPipeline._preprocess_image (nesr/nesr.py:674), on [C, H, W] u8 planes taken as ONE C-channel image (C = 1: the L
 * plane, C = 2: the a and b planes).  weights_dev: int32 table over the binned ("almost", >> 6) template distance,
 * round(M exp(-d / (h^2 C))) with OpenCV's fixed-point M, 0 below M / 1000 (imgproc.nl_means_weights builds it).  The Lab
 * conversions around it stay torch operations (imgproc.py).  Parity unpinned against cv2 (absent): checked against
 * oracle/cv2_ref.py, a restatement of OpenCV's invoker.
 */
int nesr_nl_means_u8(int device_id, const uint8_t* planes_dev, int C, int H, int W, int template_size, int search_size, const int* weights_dev, int nbins,
                     uint8_t* out_dev, void* hip_stream);

/*
 * `cv2.createCLAHE(clipLimit=2.0, tileGridSize=(8, 8)).apply(l)` of SuperResolutionPipeline._preprocess_image (nesr/nesr.py:680-684) on one
 * [H, W] u8 plane: per-tile clipped and redistributed histograms -> look-up tables (the image counts as padded by BORDER_REFLECT_101 to
 * a multiple of the grid -- both sides, only when one does not divide, as clahe.cpp does), every pixel the bilinear blend of the four
 * surrounding tiles' tables.  lut_dev: grid_x * grid_y * 256 floats of device scratch.  Bit for bit imgproc.clahe_u8's torch
 * composition; parity unpinned against cv2 (absent): checked against oracle/cv2_ref.py.
 */
int nesr_clahe_u8(int device_id, const uint8_t* gray_dev, int H, int W, double clip_limit, int grid_x, int grid_y, float* lut_dev, uint8_t* out_dev,
                  void* hip_stream);

/*
 * Single-layer entry (test hook for the per-layer parity tests): one 3x3 stride-1 zero-pad-1
 * convolution + bias (+ LeakyReLU(0.2) if lrelu) (+ nearest x2 upsample of the input first if
 * upsample), i.e. torch.nn.Conv2d / F.leaky_relu / F.interpolate as composed in RRDBNet.forward.
 *   x_dev NCHW f32 [N,Cin,H,W];  w_host OIHW f32 [Cout,Cin,3,3];  b_host [Cout];
 *   y_dev NCHW f32 [N,Cout,H<<upsample,W<<upsample].  Synchronous.
 */
int nesr_conv3x3(int device_id, int dtype, const void* x_dev, int N, int Cin, int H, int W,
                 const float* w_host, const float* b_host, int Cout, int lrelu, int upsample,
                 void* y_dev, void* stream);

const char* nesr_last_error(void);
const char* nesr_version(void);

#ifdef __cplusplus
}
#endif
#endif /* NESR_HIP_H */
