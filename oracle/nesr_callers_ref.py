"""Oracle (test infrastructure): numpy restatement of the NESR pipeline's ESRGAN call sites,
nesr/nesr.py:311-475 (_process_with_tiling), :754-813 (_apply_esrgan), :845-903 (_apply_esrgan_12channel),
:905-945 (_apply_esrgan_3channel), with the network injected (oracle.rrdbnet_ref.RRDBNetRef).

PARITY UNPINNED twice over: the network is basicsr's (absent), and the two cv2 image ops the
reference calls here -- cv2.GaussianBlur(u8,(3,3),0) and cv2.resize(INTER_LANCZOS4) -- are restated
from OpenCV's documented algorithms because cv2 is not installed (no reference output exists to pin
them: the reference has no fixtures).  Pure numpy loops/vector ops, written independently of the
product's torch implementation (neural_enhanced_super_resolution_amd/nesr_adapter.py).
"""
import math

import numpy as np
import torch


def gaussian_blur3x3_u8(img):
    """cv2.GaussianBlur(img, (3,3), 0) on uint8 HWC: weights [1 2 1]x[1 2 1]/16, BORDER_REFLECT_101,
    fixed-point round-half-up."""
    h, w = img.shape[:2]
    mode = "reflect" if (h > 1 and w > 1) else "edge"
    p = np.pad(img.astype(np.int32), ((1, 1), (1, 1), (0, 0)), mode=mode)
    acc = np.zeros(img.shape, np.int32)
    k = (1, 2, 1)
    for dy in range(3):
        for dx in range(3):
            acc += k[dy] * k[dx] * p[dy:dy + h, dx:dx + w]
    return np.clip((acc + 8) >> 4, 0, 255).astype(np.uint8)


def build_12channel(image_rgb):
    """nesr.py:851-879 -> float32 [1,12,H,W] (numpy)."""
    bgr = image_rgb[:, :, ::-1]
    t = np.transpose(bgr, (2, 0, 1)).astype(np.float32) / np.float32(255.0)
    blurred = np.transpose(gaussian_blur3x3_u8(np.ascontiguousarray(bgr)), (2, 0, 1)).astype(np.float32) / np.float32(255.0)
    return np.concatenate([t, np.clip(t * np.float32(1.1), 0, 1), np.clip(t * np.float32(0.9), 0, 1), blurred], 0)[None]


def build_3channel_x4(image_rgb):
    bgr = image_rgb[:, :, ::-1]
    t = np.transpose(bgr, (2, 0, 1)).astype(np.float32) / np.float32(255.0)
    return np.concatenate([t, t, t, t], 0)[None]


def quantize_trunc_to_rgb(out_chw):
    """nesr.py:894-901: CHW float -> HWC uint8 RGB (truncation)."""
    o = np.transpose(out_chw, (1, 2, 0)) * 255.0
    o = np.clip(o, 0, 255).astype(np.uint8)
    return o[:, :, ::-1]


def apply_12channel(model, image_rgb):
    with torch.no_grad():
        y = model(torch.from_numpy(build_12channel(image_rgb))).squeeze().numpy()
    return np.ascontiguousarray(quantize_trunc_to_rgb(y))


def apply_3channel(model, image_rgb):
    with torch.no_grad():
        y = model(torch.from_numpy(build_3channel_x4(image_rgb))).squeeze().numpy()
    return np.ascontiguousarray(quantize_trunc_to_rgb(y))


def lanczos4_resize_u8(img, out_h, out_w):
    """cv2.resize(img, (out_w, out_h), interpolation=INTER_LANCZOS4): oracle/cv2_ref.py (OpenCV's 8-bit fixed point)."""
    from oracle import cv2_ref
    return cv2_ref.resize_lanczos4(np.ascontiguousarray(img), out_h, out_w)


def process_with_tiling(processor, image, tile_size, padding, upscale_factor):
    """nesr.py:311-475 without the probe tile and the bicubic fallbacks."""
    h, w, c = image.shape
    if h <= tile_size and w <= tile_size:
        return processor(image)
    out_h, out_w = int(h * upscale_factor), int(w * upscale_factor)
    output = np.zeros((out_h, out_w, c), np.uint8)
    for i in range(math.ceil(h / tile_size)):
        for j in range(math.ceil(w / tile_size)):
            y_start, y_end = max(0, i * tile_size - padding), min(h, (i + 1) * tile_size + padding)
            x_start, x_end = max(0, j * tile_size - padding), min(w, (j + 1) * tile_size + padding)
            tile = image[y_start:y_end, x_start:x_end]
            pt = processor(tile)
            oy0, oy1 = int(y_start * upscale_factor), int(y_end * upscale_factor)
            ox0, ox1 = int(x_start * upscale_factor), int(x_end * upscale_factor)
            if padding > 0:
                pu = int(padding * upscale_factor)
                oy0 += pu if y_start > 0 else 0
                oy1 -= pu if y_end < h else 0
                ox0 += pu if x_start > 0 else 0
                ox1 -= pu if x_end < w else 0
            th, tw = pt.shape[:2]
            sy, sx = th / tile.shape[0], tw / tile.shape[1]
            ty0 = 0 if y_start == 0 else int(padding * sy)
            ty1 = th if y_end == h else int(th - padding * sy)
            tx0 = 0 if x_start == 0 else int(padding * sx)
            tx1 = tw if x_end == w else int(tw - padding * sx)
            ty0 = max(0, min(ty0, th - 1)); ty1 = max(ty0 + 1, min(ty1, th))
            tx0 = max(0, min(tx0, tw - 1)); tx1 = max(tx0 + 1, min(tx1, tw))
            oh, ow = oy1 - oy0, ox1 - ox0
            if oh <= 0 or ow <= 0:
                continue
            region = pt[ty0:ty1, tx0:tx1]
            if region.shape[0] != oh or region.shape[1] != ow:
                region = lanczos4_resize_u8(region, oh, ow)
            output[oy0:oy1, ox0:ox1] = region
    return output


def apply_esrgan(model, image, config=None, device_kind="cuda", large_mp=16, trace=None):
    """nesr.py:754-813 (the dispatch; the exception ladder :815-843 is not restated: the oracle's network does not fail).
    `large_mp` is the literal 16 of nesr.py:787."""
    cfg = {"enable_tiling": True, "force_3channel": False, "max_tile_size": 512, "upscale_factor": 2.0}
    cfg.update(config or {})
    h, w, _ = image.shape
    image_megapixels = (h * w) / (1024 * 1024)                                   # nesr.py:762
    use_tiling = False
    if cfg["enable_tiling"]:                                                       # nesr.py:766-776
        if device_kind == "cpu":
            threshold = cfg.get("cpu_megapixel_threshold", 2)
        elif device_kind == "mps":
            threshold = cfg.get("mps_megapixel_threshold", 4)
        else:
            threshold = cfg.get("cuda_megapixel_threshold", 8)
        use_tiling = image_megapixels > threshold
    use_3channel = cfg["force_3channel"]                                           # nesr.py:779
    if device_kind == "mps" and image_megapixels > 1:                             # nesr.py:782-784
        use_3channel = True
    if image_megapixels > large_mp:                                                # nesr.py:787-790
        use_tiling = True
        use_3channel = True
    proc = (lambda t: apply_3channel(model, np.ascontiguousarray(t))) if use_3channel else \
           (lambda t: apply_12channel(model, np.ascontiguousarray(t)))
    if trace is not None:
        trace.append({"in_shape": (h, w), "tiled": use_tiling, "three_channel": use_3channel})
    if use_tiling:                                                                 # nesr.py:797-807
        return process_with_tiling(proc, image, cfg["max_tile_size"], 16, cfg["upscale_factor"])
    return proc(image)                                                             # nesr.py:810-813


def enhance_iterations(model, image, config=None, device_kind="cuda", large_mp=16, trace=None, filters=False):
    """nesr.py:516-633 with use_diffusion=False, segment_enhancement=False: per iteration
    current = _postprocess_image(_ensemble_results([_apply_esrgan(_preprocess_image(current))])), the ensemble of one model
    being the identity (nesr.py:1035-1036); filters=False leaves the cv2 pre / post filters (oracle/cv2_ref.py) out."""
    cfg = {"iterations": 3, "upscale_factor": 2.0, "denoise_level": 0.5, "adaptive_sharpening": True}
    cfg.update(config or {})
    current = image
    for _ in range(int(cfg["iterations"])):
        if filters:
            from oracle import cv2_ref
            current = cv2_ref.preprocess_image(current, cfg["denoise_level"])
        current = apply_esrgan(model, current, cfg, device_kind, large_mp, trace)
        if filters:
            from oracle import cv2_ref
            current = cv2_ref.postprocess_image(current, cfg["adaptive_sharpening"])
    return current
