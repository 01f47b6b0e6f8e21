"""GPU versions of the NESR pipeline's ESRGAN call sites (SURVEY.md section 8(a) rows a16-a19,
section 8(f) row 2): everything between "HWC uint8 RGB ndarray in" and "HWC uint8 RGB ndarray out"
runs on the device -- 12-channel synthesis, the network (``upscaler.model``, the reference's own way of
calling it), the truncating quantiser, and the NESR tiler with its Lanczos resize.

Reference (all in nesr/nesr.py):
  _apply_esrgan                 :754-843   dispatch by megapixels / device, 12-ch vs 3-ch, tiling
  _apply_esrgan_12channel       :845-903   [img, clamp(1.1 img), clamp(0.9 img), GaussianBlur3x3(img)] -> model
  _apply_esrgan_3channel        :905-945   img repeated 4x -> model
  _process_with_tiling          :311-475   ceil grid, +-padding windows, crop, Lanczos resize to the canvas

Differences, on purpose: no exception ladder (nesr.py:815-843, 448-473 turn any backend failure into a
bicubic result -- here failures raise), no MPS branches, no probe tile (nesr.py:349-357 runs the
processor on a 256x256 corner and discards the result).

cv2 is not available offline, so its two non-trivial image ops are restated, PARITY UNPINNED:
  * cv2.GaussianBlur(u8, (3,3), 0): kernel [1 2 1]/4 separable, BORDER_REFLECT_101, fixed-point with
    round-half-up -> (S + 8) >> 4 on the 16-weight integer sum.
  * cv2.resize(..., INTER_LANCZOS4): cv2's coefficient formula (interpolateLanczos4), sampling geometry and 8-bit
    fixed point (coefficients x2048 as shorts, integer passes, (v + 2^21) >> 22): imgproc.lanczos4_resize.
"""
from __future__ import annotations

import math

import numpy as np
import torch
from torch.nn import functional as F

from .realesrganer import normalize_u8_on_device


# ----------------------------------------------------------------------------- 12-channel builder
def gaussian_blur3x3_u8(img):
    """img: uint8 [H, W, C] tensor -> uint8, cv2.GaussianBlur(img, (3, 3), 0) restated (see module doc)."""
    x = img.permute(2, 0, 1).unsqueeze(0).to(torch.int32)
    h, w = x.shape[-2:]
    if h > 1 and w > 1:
        xp = F.pad(x.float(), (1, 1, 1, 1), mode="reflect").to(torch.int32)     # BORDER_REFLECT_101
    else:
        xp = F.pad(x.float(), (1, 1, 1, 1), mode="replicate").to(torch.int32)
    hs = xp[..., :, 0:-2] + 2 * xp[..., :, 1:-1] + xp[..., :, 2:]
    s = hs[..., 0:-2, :] + 2 * hs[..., 1:-1, :] + hs[..., 2:, :]
    return ((s + 8) >> 4).clamp_(0, 255).to(torch.uint8).squeeze(0).permute(1, 2, 0)


def _u8_on(image_rgb, device):
    """HWC uint8 ndarray or tensor -> tensor on `device` (a tensor already there is used as it is: the
    iteration loop keeps its frames on the GPU)."""
    if isinstance(image_rgb, torch.Tensor):
        return image_rgb.to(device)
    return torch.as_tensor(np.ascontiguousarray(image_rgb)).to(device)


def build_12channel(image_rgb, device):
    """nesr.py:851-882: RGB u8 -> [1, 12, H, W] float32 on device
    = [bgr/255, clamp(1.1 bgr/255), clamp(0.9 bgr/255), GaussianBlur3x3(bgr)/255]."""
    bgr = _u8_on(image_rgb, device).flip(2)                                                # cv2.COLOR_RGB2BGR
    t = normalize_u8_on_device(bgr.permute(2, 0, 1))                                       # /255.0
    blurred = normalize_u8_on_device(gaussian_blur3x3_u8(bgr).permute(2, 0, 1))
    return torch.cat([t, torch.clamp(t * 1.1, 0, 1), torch.clamp(t * 0.9, 0, 1), blurred], 0).unsqueeze(0)


def build_3channel_x4(image_rgb, device):
    """nesr.py:915-927: RGB u8 -> [1, 12, H, W] = the BGR/255 image repeated 4 times."""
    bgr = _u8_on(image_rgb, device).flip(2)
    t = normalize_u8_on_device(bgr.permute(2, 0, 1))
    return torch.cat([t, t, t, t], 0).unsqueeze(0)


def quantize_trunc_to_rgb(output):
    """nesr.py:894-901: [1,3,H,W] float -> HWC uint8 RGB: x*255, clip(0,255), truncating astype, BGR->RGB."""
    out = output.squeeze(0).float()
    q = (out.permute(1, 2, 0) * 255.0).clamp_(0, 255).to(torch.uint8)
    return q.flip(2)


def _to_host(upscaler, t):
    """Device-to-host copy + the range check of the f16-pair fp32 form (a NaN image is an error here; the
    reference's np.clip(...).astype(uint8) would turn NaN pixels into garbage silently, nesr.py:897-898)."""
    host = t.cpu().numpy()
    check = getattr(upscaler.model, "check_range", None)
    if check is not None:
        check()
    return host


@torch.no_grad()
def apply_esrgan_12channel(upscaler, image_rgb, as_numpy=True):
    """_apply_esrgan_12channel (nesr.py:845-903) with every step on the GPU."""
    model = upscaler.model
    model.eval()
    y = model(build_12channel(image_rgb, upscaler.device))
    q = quantize_trunc_to_rgb(y)
    return _to_host(upscaler, q) if as_numpy else q


@torch.no_grad()
def apply_esrgan_3channel(upscaler, image_rgb, as_numpy=True):
    """_apply_esrgan_3channel (nesr.py:905-945)."""
    model = upscaler.model
    model.eval()
    y = model(build_3channel_x4(image_rgb, upscaler.device))
    q = quantize_trunc_to_rgb(y)
    return _to_host(upscaler, q) if as_numpy else q


# ----------------------------------------------------------------------------- Lanczos-4 resize
from .imgproc import lanczos4_resize as lanczos4_resize_u8   # noqa: E402  cv2.resize(INTER_LANCZOS4), OpenCV's 8-bit fixed point


# ----------------------------------------------------------------------------- NESR tiler + dispatcher
@torch.no_grad()
def process_with_tiling(processor, image_rgb, tile_size, padding, upscale_factor, device, as_numpy=True):
    """_process_with_tiling (nesr.py:311-475): `processor(tile_rgb_u8 ndarray|tensor) -> uint8 tensor`.
    Returns an HWC uint8 RGB image of size int(h*uf) x int(w*uf) (ndarray, or the device tensor)."""
    h, w, c = image_rgb.shape
    if h <= tile_size and w <= tile_size:
        out = processor(image_rgb)
        return out.cpu().numpy() if as_numpy else out
    nth, ntw = math.ceil(h / tile_size), math.ceil(w / tile_size)
    out_h, out_w = int(h * upscale_factor), int(w * upscale_factor)
    canvas = torch.zeros((out_h, out_w, c), dtype=torch.uint8, device=device)
    for i in range(nth):
        for j in range(ntw):
            y0, y1 = max(0, i * tile_size - padding), min(h, (i + 1) * tile_size + padding)
            x0, x1 = max(0, j * tile_size - padding), min(w, (j + 1) * tile_size + padding)
            tile = image_rgb[y0:y1, x0:x1]
            pt = processor(tile)
            oy0, oy1 = int(y0 * upscale_factor), int(y1 * upscale_factor)
            ox0, ox1 = int(x0 * upscale_factor), int(x1 * upscale_factor)
            if padding > 0:
                pu = int(padding * upscale_factor)
                if y0 > 0:
                    oy0 += pu
                if y1 < h:
                    oy1 -= pu
                if x0 > 0:
                    ox0 += pu
                if x1 < w:
                    ox1 -= pu
            th, tw = pt.shape[:2]
            sy, sx = th / tile.shape[0], tw / tile.shape[1]
            ty0 = 0 if y0 == 0 else int(padding * sy)
            ty1 = th if y1 == h else int(th - padding * sy)
            tx0 = 0 if x0 == 0 else int(padding * sx)
            tx1 = tw if x1 == w else int(tw - padding * sx)
            ty0 = max(0, min(ty0, th - 1))
            ty1 = max(ty0 + 1, min(ty1, th))
            tx0 = max(0, min(tx0, tw - 1))
            tx1 = max(tx0 + 1, min(tx1, tw))
            oh, ow = oy1 - oy0, ox1 - ox0
            if oh <= 0 or ow <= 0:
                continue
            region = pt[ty0:ty1, tx0:tx1]
            if region.shape[0] != oh or region.shape[1] != ow:
                region = lanczos4_resize_u8(region, oh, ow)     # cv2.resize(..., INTER_LANCZOS4), nesr.py:438-443
            canvas[oy0:oy1, ox0:ox1] = region
    return canvas.cpu().numpy() if as_numpy else canvas


LARGE_IMAGE_MP = 16      # nesr.py:787: above this many "megapixels" (px / 1024^2) tiling and 3-channel mode are forced


def apply_esrgan(upscaler, image_rgb, config=None, device_kind="cuda", as_numpy=True, trace=None, large_mp=LARGE_IMAGE_MP):
    """_apply_esrgan (nesr.py:754-813): the reference's dispatch, minus its fallback ladder.
    config keys as the reference's: enable_tiling, force_3channel, max_tile_size, upscale_factor,
    cuda_megapixel_threshold (the reference's literal default for cuda is 8).  `large_mp` is the reference's
    literal 16 (a parameter only so that tests can reach that branch with small frames); `trace`, if a list,
    receives one dict describing the route taken."""
    cfg = {"enable_tiling": True, "force_3channel": False, "max_tile_size": 512, "upscale_factor": 2.0}
    cfg.update(config or {})
    h, w, _ = image_rgb.shape
    megapixels = (h * w) / (1024 * 1024)
    use_tiling = False
    if cfg["enable_tiling"]:
        thr = {"cpu": cfg.get("cpu_megapixel_threshold", 2), "mps": cfg.get("mps_megapixel_threshold", 4)}.get(
            device_kind, cfg.get("cuda_megapixel_threshold", 8))
        use_tiling = megapixels > thr
    use_3ch = cfg["force_3channel"]
    if megapixels > large_mp:
        use_tiling, use_3ch = True, True
    model = upscaler.model
    calls0 = getattr(model, "calls", None)
    net_px = [0]

    def one(t):
        net_px[0] += int(t.shape[0]) * int(t.shape[1])
        return apply_esrgan_3channel(upscaler, t, as_numpy=False) if use_3ch else apply_esrgan_12channel(upscaler, t, as_numpy=False)

    if use_tiling:
        out = process_with_tiling(one, image_rgb, cfg["max_tile_size"], 16, cfg["upscale_factor"], upscaler.device, as_numpy=False)
    else:
        out = one(image_rgb)
    if trace is not None:
        trace.append({"in_shape": (h, w), "out_shape": tuple(out.shape[:2]), "tiled": bool(use_tiling), "three_channel": bool(use_3ch),
                      "model_calls": None if calls0 is None else model.calls - calls0, "net_input_px": net_px[0]})
    if as_numpy:
        return _to_host(upscaler, out)
    return out


def enhance_iterations(upscaler, image_rgb, config=None, device_kind="cuda", preprocess=None, postprocess=None,
                       trace=None, large_mp=LARGE_IMAGE_MP, filters=False):
    """The iteration loop of SuperResolutionPipeline.enhance_image (nesr.py:516-633) around its ESRGAN stage:

        for iteration in range(config['iterations']):           nesr.py:516
            current = _preprocess_image(current)                 nesr.py:537   -> `preprocess` (NL-means + CLAHE; None = off)
            esrgan_result = _apply_esrgan(current)               nesr.py:566   -> apply_esrgan above
            current = _ensemble_results([esrgan_result])         nesr.py:596   one model: the identity (nesr.py:1035-1036)
            current = _postprocess_image(current)                nesr.py:616   -> `postprocess` (adaptive unsharp; None = off)

    with diffusion and segmentation off (BASELINE.json configs[4]: `--no_diffusion`; the SegFormer weights are a
    network fetch).  `filters=True` runs the reference's cv2 pre / post filters too (NL-means + CLAHE, adaptive unsharp:
    imgproc.py, OpenCV's algorithms restated -- parity unpinned).  Frames stay on the GPU between iterations; the final frame is returned as an HWC uint8 RGB
    ndarray.  A backend failure raises (the reference would hand back a bicubic resize, nesr.py:835-843); `trace`
    receives one dict per iteration with the route and the number of network evaluations, so a caller can assert
    that the network really ran."""
    cfg = {"iterations": 3, "upscale_factor": 2.0, "denoise_level": 0.5, "adaptive_sharpening": True}
    cfg.update(config or {})
    if filters:      # the reference's own pre / post filters (nesr.py:668-689, 1056-1084), on the device: imgproc.py (cv2 restated)
        from . import imgproc
        preprocess = preprocess or (lambda im: imgproc.preprocess_image(_u8_on(im, upscaler.device), cfg["denoise_level"]))
        postprocess = postprocess or (lambda im: imgproc.postprocess_image(_u8_on(im, upscaler.device), cfg["adaptive_sharpening"]))
    current = image_rgb
    for iteration in range(int(cfg["iterations"])):
        if preprocess is not None:
            current = preprocess(current)
        current = apply_esrgan(upscaler, current, cfg, device_kind, as_numpy=False, trace=trace, large_mp=large_mp)
        if trace is not None:
            trace[-1]["iteration"] = iteration
        if postprocess is not None:
            current = postprocess(current)
    return _to_host(upscaler, current) if isinstance(current, torch.Tensor) else current
