"""The OpenCV image operations the reference wraps around its ESRGAN stage, as device-side torch code
(SURVEY.md section 8(f) rows 3 and 4):

  RealESRGANer.enhance(outscale=...)        cv2.resize(INTER_LANCZOS4)                      [UPSTREAM realesrgan utils.py]
  RealESRGANer.enhance(alpha_upsampler=..)  cv2.resize(alpha, INTER_LINEAR)                 [UPSTREAM]
  SuperResolutionPipeline._preprocess_image cv2.fastNlMeansDenoisingColored + CLAHE on L    nesr/nesr.py:668-689
  SuperResolutionPipeline._postprocess_image variance-masked unsharp                        nesr/nesr.py:1056-1084
  (_process_with_tiling's Lanczos paste and the 12-channel builder's 3x3 blur use the same functions: nesr_adapter.py)

PARITY UNPINNED, all of it: cv2 is not installed here or on the GPU box and the reference holds no output of any of
these calls, so each function restates OpenCV's documented algorithm (8-bit paths in OpenCV's fixed point: 11-bit
resize coefficients, 8-bit Gaussian kernels, integer non-local-means weights; Lab conversions in float with rounding
where OpenCV uses lookup tables -- expect +-1 LSB there).  oracle/cv2_ref.py restates them again, independently, in numpy:
the tests compare two restatements, not this code with OpenCV.

Everything takes and returns HWC uint8 tensors on the caller's device (the frames of the iteration loop stay on the
GPU: a 16384x16384 frame through cv2's CPU non-local means would take minutes).
"""
from __future__ import annotations

import math

import torch
from torch.nn import functional as F


# ----------------------------------------------------------------------------------------------- resize
def _lanczos4_coeffs(frac):
    """cv2 interpolateLanczos4 for a float32 tensor of fractional offsets -> [..., 8] float32 weights."""
    s45 = 0.70710678118654752440084436210485
    cs = torch.tensor([[1, 0], [-s45, -s45], [0, 1], [s45, -s45], [-1, 0], [s45, s45], [0, -1], [-s45, s45]],
                      dtype=torch.float64, device=frac.device)
    x = frac.to(torch.float64)
    y0 = -(x + 3) * (math.pi * 0.25)
    s0, c0 = torch.sin(y0), torch.cos(y0)
    i = torch.arange(8, device=frac.device, dtype=torch.float64)
    y = -(x[..., None] + 3 - i) * (math.pi * 0.25)
    co = ((cs[:, 0] * s0[..., None] + cs[:, 1] * c0[..., None]) / (y * y)).to(torch.float32)
    co = co * (1.0 / co.sum(-1, keepdim=True))
    exact = (frac < 1.1920929e-07)[..., None]
    delta = torch.zeros(8, device=frac.device)
    delta[3] = 1.0
    return torch.where(exact, delta.expand_as(co), co)


def _axis_taps(n_in, n_out, device, taps, first):
    """Source indices [n_out, taps] (clamped: cv2 replicates the border) and fractional offsets [n_out] of cv2.resize."""
    scale = n_in / n_out
    pos = (torch.arange(n_out, device=device, dtype=torch.float64) + 0.5) * scale - 0.5
    pos = pos.to(torch.float32)
    i0 = torch.floor(pos)
    frac = pos - i0
    idx = (i0.long()[:, None] + torch.arange(first, first + taps, device=device)).clamp_(0, n_in - 1)
    return idx, frac, i0.long()


def lanczos4_resize(img, out_h, out_w):
    """cv2.resize(img, (out_w, out_h), interpolation=cv2.INTER_LANCZOS4) for HWC uint8 or uint16 (int32-held) tensors.
    uint8: OpenCV's fixed point -- coefficients rounded to 11 bits (x2048, short), integer horizontal pass, integer vertical
    pass, (v + 2^21) >> 22, saturate.  uint16: float32 coefficients and sums, round to nearest even, saturate."""
    h, w, c = img.shape
    is8 = img.dtype == torch.uint8
    ix, fx, _ = _axis_taps(w, out_w, img.device, 8, -3)
    iy, fy, _ = _axis_taps(h, out_h, img.device, 8, -3)
    wx, wy = _lanczos4_coeffs(fx), _lanczos4_coeffs(fy)
    if is8:
        ax = torch.round(wx * 2048.0).clamp_(-32768, 32767).to(torch.int64)     # saturate_cast<short>(c * INTER_RESIZE_COEF_SCALE)
        ay = torch.round(wy * 2048.0).clamp_(-32768, 32767).to(torch.int64)
        x = img.permute(2, 0, 1).to(torch.int64)                                  # [C, H, W]
        rows = (x[:, :, ix] * ax).sum(-1)                                         # [C, H, out_w]
        out = (rows[:, iy, :] * ay[None, :, :, None]).sum(2)                      # [C, out_h, out_w]
        out = (out + (1 << 21)) >> 22
        return out.clamp_(0, 255).to(torch.uint8).permute(1, 2, 0).contiguous()
    x = img.permute(2, 0, 1).float()
    rows = (x[:, :, ix] * wx).sum(-1)
    out = (rows[:, iy, :] * wy[None, :, :, None]).sum(2)
    return torch.round(out).clamp_(0, 65535).to(torch.int32).permute(1, 2, 0).contiguous()


def linear_resize_f32(img, out_h, out_w):
    """cv2.resize(img, (out_w, out_h), interpolation=cv2.INTER_LINEAR) for float32 HW or HWC tensors."""
    squeeze = img.dim() == 2
    x = (img[:, :, None] if squeeze else img).permute(2, 0, 1).float()
    h, w = x.shape[1:]

    def axis(n_in, n_out):
        scale = n_in / n_out
        pos = ((torch.arange(n_out, device=img.device, dtype=torch.float64) + 0.5) * scale - 0.5).to(torch.float32)
        i0 = torch.floor(pos)
        f = pos - i0
        i0 = i0.long()
        lo = i0 < 0
        hi = i0 >= n_in - 1
        f = torch.where(lo | hi, torch.zeros_like(f), f)
        i0 = torch.where(lo, torch.zeros_like(i0), torch.where(hi, torch.full_like(i0, n_in - 1), i0))
        return i0, (i0 + 1).clamp_(max=n_in - 1), f

    x0, x1, fx = axis(w, out_w)
    y0, y1, fy = axis(h, out_h)
    rows = x[:, :, x0] * (1.0 - fx) + x[:, :, x1] * fx
    out = rows[:, y0, :] * (1.0 - fy)[None, :, None] + rows[:, y1, :] * fy[None, :, None]
    out = out.permute(1, 2, 0)
    return out[:, :, 0].contiguous() if squeeze else out.contiguous()


# ----------------------------------------------------------------------------------------------- Gaussian blur
def gaussian_kernel_u8(sigma, ksize=0):
    """Fixed-point Gaussian kernel of OpenCV's 8-bit path: ksize = round(6 sigma + 1) | 1 when not given, float kernel
    exp(-x^2 / 2 sigma^2) normalised, x256 rounded, the centre adjusted so that the taps sum to 256."""
    if ksize <= 0:
        ksize = int(round(sigma * 6 + 1)) | 1
    r = ksize // 2
    small = {1: [1.0], 3: [0.25, 0.5, 0.25], 5: [0.0625, 0.25, 0.375, 0.25, 0.0625],
             7: [0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125]}
    if sigma <= 0 and ksize in small:              # OpenCV's tabulated small kernels
        k = torch.tensor(small[ksize], dtype=torch.float64)
    else:
        if sigma <= 0:
            sigma = 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8
        xs = torch.arange(-r, r + 1, dtype=torch.float64)
        k = torch.exp(-(xs * xs) / (2.0 * sigma * sigma))
        k = k / k.sum()
    q = torch.round(k * 256.0).to(torch.int64)
    q[r] += 256 - int(q.sum())
    return q


def gaussian_blur_u8(img, sigma, ksize=0):
    """cv2.GaussianBlur(img, (ksize, ksize) or (0, 0), sigma) on HWC (or HW) uint8: separable fixed-point filter,
    BORDER_REFLECT_101, one rounding at the end ((v + 2^15) >> 16)."""
    squeeze = img.dim() == 2
    x = (img[:, :, None] if squeeze else img).permute(2, 0, 1).unsqueeze(0)
    k = gaussian_kernel_u8(sigma, ksize).to(img.device)
    r = k.numel() // 2
    h, w = x.shape[-2:]
    xi = x.to(torch.int32)                                                # every partial sum < 2^24
    cols = _reflect101_index(w, r, img.device)
    rows = _reflect101_index(h, r, img.device)
    hp = xi[..., cols]                                                    # [1, C, H, W + 2r]
    hs = sum(int(k[t]) * hp[..., t:t + w] for t in range(2 * r + 1))
    vp = hs[..., rows, :]
    vs = sum(int(k[t]) * vp[..., t:t + h, :] for t in range(2 * r + 1))
    out = ((vs + (1 << 15)) >> 16).clamp_(0, 255).to(torch.uint8).squeeze(0).permute(1, 2, 0)
    return out[:, :, 0].contiguous() if squeeze else out.contiguous()


# ----------------------------------------------------------------------------------------------- colour
def rgb2gray_u8(img):
    """cv2.cvtColor(img, COLOR_RGB2GRAY) on uint8: (R 4899 + G 9617 + B 1868 + 2^13) >> 14."""
    x = img.to(torch.int64)
    return ((x[..., 0] * 4899 + x[..., 1] * 9617 + x[..., 2] * 1868 + (1 << 13)) >> 14).to(torch.uint8)


_D65 = (0.950456, 1.0, 1.088754)
_M = ((0.412453, 0.357580, 0.180423), (0.212671, 0.715160, 0.072169), (0.019334, 0.119193, 0.950227))


def _lab_f(t):
    return torch.where(t > 0.008856, torch.pow(t.clamp_min(1e-12), 1.0 / 3.0), 7.787 * t + 16.0 / 116.0)


def _srgb_to_linear(c):
    return torch.where(c <= 0.04045, c / 12.92, torch.pow((c + 0.055) / 1.055, 2.4))


def _linear_to_srgb(c):
    return torch.where(c <= 0.0031308, c * 12.92, 1.055 * torch.pow(c.clamp_min(1e-12), 1.0 / 2.4) - 0.055)


def rgb2lab_u8(img, linear=False, first_is_blue=False):
    """cv2.cvtColor(img, COLOR_RGB2Lab | COLOR_LRGB2Lab | COLOR_LBGR2Lab) on uint8 -> uint8 (L 255/100, a + 128, b + 128).
    linear=True: no sRGB gamma (the L* variants); first_is_blue=True: channel 0 is taken as blue (the *BGR* variants)."""
    c = img.float() / 255.0
    if not linear:
        c = _srgb_to_linear(c)
    r, g, b = (c[..., 2], c[..., 1], c[..., 0]) if first_is_blue else (c[..., 0], c[..., 1], c[..., 2])
    X = (_M[0][0] * r + _M[0][1] * g + _M[0][2] * b) / _D65[0]
    Y = _M[1][0] * r + _M[1][1] * g + _M[1][2] * b
    Z = (_M[2][0] * r + _M[2][1] * g + _M[2][2] * b) / _D65[2]
    fx, fy, fz = _lab_f(X), _lab_f(Y), _lab_f(Z)
    L = torch.where(Y > 0.008856, 116.0 * fy - 16.0, 903.3 * Y)
    A = 500.0 * (fx - fy) + 128.0
    B = 200.0 * (fy - fz) + 128.0
    out = torch.stack([L * 255.0 / 100.0, A, B], -1)
    return torch.round(out).clamp_(0, 255).to(torch.uint8)


def lab2rgb_u8(lab, linear=False, first_is_blue=False):
    """cv2.cvtColor(lab, COLOR_Lab2RGB | COLOR_Lab2LRGB | COLOR_Lab2LBGR) on uint8 -> uint8."""
    x = lab.float()
    L = x[..., 0] * 100.0 / 255.0
    a = x[..., 1] - 128.0
    b = x[..., 2] - 128.0
    fy = (L + 16.0) / 116.0
    Y = torch.where(L <= 8.0, L / 903.3, fy * fy * fy)
    fy = torch.where(L <= 8.0, 7.787 * Y + 16.0 / 116.0, fy)
    fx = fy + a / 500.0
    fz = fy - b / 200.0

    def inv(f):
        return torch.where(f <= 6.0 / 29.0, (f - 16.0 / 116.0) / 7.787, f * f * f)

    X = inv(fx) * _D65[0]
    Z = inv(fz) * _D65[2]
    r = 3.240479 * X - 1.537150 * Y - 0.498535 * Z
    g = -0.969256 * X + 1.875991 * Y + 0.041556 * Z
    bl = 0.055648 * X - 0.204043 * Y + 1.057311 * Z
    c = torch.stack([bl, g, r] if first_is_blue else [r, g, bl], -1).clamp_(0, 1)
    if not linear:
        c = _linear_to_srgb(c)
    return torch.round(c * 255.0).clamp_(0, 255).to(torch.uint8)


# ----------------------------------------------------------------------------------------------- CLAHE
def clahe_tile_size(h, w, grid=(8, 8)):
    """(tile height, tile width) cv2's CLAHE uses for an h x w image (clahe.cpp, CLAHE_Impl::apply)."""
    gx, gy = grid
    ph, pw = (gy - h % gy, gx - w % gx) if (h % gy or w % gx) else (0, 0)
    return (h + ph) // gy, (w + pw) // gx


def clahe_u8(gray, clip_limit=2.0, grid=(8, 8), use_hip=None):
    """cv2.createCLAHE(clipLimit, tileGridSize).apply(gray) on an HW uint8 tensor: the image is padded (REFLECT_101) to a
    multiple of the grid, every tile gets a clipped, redistributed histogram and a look-up table, and a pixel takes the
    bilinear blend of the four surrounding tiles' tables.  On a ROCm device the two HIP kernels of csrc/imgproc.hip
    (nesr_clahe_u8) unless use_hip=False selects the torch composition below -- the two agree bit for bit."""
    h, w = gray.shape
    gx, gy = grid
    if use_hip is None:
        use_hip = gray.device.type == "cuda" and gray.dtype == torch.uint8
    if use_hip:
        import ctypes
        from . import _lib
        src = gray.contiguous()
        out = torch.empty_like(src)
        lut = torch.empty((gy * gx * 256,), dtype=torch.float32, device=src.device)
        index = src.device.index if src.device.index is not None else torch.cuda.current_device()
        with torch.cuda.device(src.device):
            stream = torch.cuda.current_stream(src.device).cuda_stream
            _lib.check(_lib.load().nesr_clahe_u8(index, ctypes.c_void_p(src.data_ptr()), h, w, float(clip_limit), gx, gy, ctypes.c_void_p(lut.data_ptr()),
                                                 ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(stream)), "nesr_clahe_u8")
        return out
    # clahe.cpp: only when BOTH sides divide by the grid is the image used as it is; otherwise copyMakeBorder pads the bottom by
    # tilesY - h % tilesY and the right by tilesX - w % tilesX -- a side that does divide gets a whole extra tilesY / tilesX pixels
    ph, pw = (gy - h % gy, gx - w % gx) if (h % gy or w % gx) else (0, 0)
    src = gray
    if ph or pw:
        ry = torch.arange(h + ph, device=gray.device)
        rx = torch.arange(w + pw, device=gray.device)
        ry = torch.where(ry >= h, 2 * (h - 1) - ry, ry).clamp_(0, h - 1)              # BORDER_REFLECT_101 at the bottom / right
        rx = torch.where(rx >= w, 2 * (w - 1) - rx, rx).clamp_(0, w - 1)
        src = gray[ry][:, rx]
    H, W = src.shape
    th, tw = H // gy, W // gx
    area = th * tw
    clip = max(int(clip_limit * area / 256.0), 1)
    tiles = src.reshape(gy, th, gx, tw).permute(0, 2, 1, 3).reshape(gy * gx, area).long()
    hist = torch.zeros((gy * gx, 256), dtype=torch.int64, device=gray.device)
    hist.scatter_add_(1, tiles, torch.ones_like(tiles))
    clipped = (hist - clip).clamp_min(0).sum(1)
    hist = hist.clamp_max(clip)
    batch = clipped // 256
    residual = clipped - batch * 256
    hist = hist + batch[:, None]
    step = torch.where(residual > 0, (256 // residual.clamp_min(1)).clamp_min(1), torch.ones_like(residual))
    i = torch.arange(256, device=gray.device)[None, :]
    # one extra count at bins 0, step, 2 step, ... until `residual` of them have been handed out
    extra = ((i % step[:, None]) == 0) & ((i // step[:, None]) < residual[:, None])
    hist = hist + extra.long()
    lut = torch.round(hist.cumsum(1).float() * (255.0 / area)).clamp_(0, 255)            # [tiles, 256]
    lut = lut.reshape(gy, gx, 256)
    ys = torch.arange(h, device=gray.device).float() / th - 0.5
    xs = torch.arange(w, device=gray.device).float() / tw - 0.5
    ty1, tx1 = torch.floor(ys), torch.floor(xs)
    ya, xa = ys - ty1, xs - tx1
    ty1, tx1 = ty1.long(), tx1.long()
    ty2, tx2 = (ty1 + 1).clamp(0, gy - 1), (tx1 + 1).clamp(0, gx - 1)
    ty1, tx1 = ty1.clamp(0, gy - 1), tx1.clamp(0, gx - 1)
    v = gray.long()
    l11 = lut[ty1[:, None], tx1[None, :], v]
    l12 = lut[ty1[:, None], tx2[None, :], v]
    l21 = lut[ty2[:, None], tx1[None, :], v]
    l22 = lut[ty2[:, None], tx2[None, :], v]
    xa, ya = xa[None, :], ya[:, None]
    res = (l11 * (1 - xa) + l12 * xa) * (1 - ya) + (l21 * (1 - xa) + l22 * xa) * ya
    return torch.round(res).clamp_(0, 255).to(torch.uint8)


# ----------------------------------------------------------------------------------------------- non-local means
def _reflect101_index(n, r, device):
    """Source index of positions -r .. n + r - 1 under BORDER_REFLECT_101 (gfedcb|abcdefgh|gfedcba)."""
    idx = torch.arange(-r, n + r, device=device)
    if n == 1:
        return torch.zeros_like(idx)
    period = 2 * (n - 1)
    idx = idx % period
    return torch.where(idx >= n, period - idx, idx)


def _reflect101_pad(x, r):
    """[..., H, W] tensor -> padded by r on both axes with BORDER_REFLECT_101 (any r, also larger than the image)."""
    h, w = x.shape[-2:]
    return x[..., _reflect101_index(h, r, x.device), :][..., _reflect101_index(w, r, x.device)]


def nl_means_weights(C, h, template=7, search=21):
    """(int64 table of weights over the binned template distance, shift): OpenCV's almost_dist2weight -- round(M exp(-d / (h^2 C)))
    with the fixed-point multiplier M = INT_MAX // (search^2 * 255), 0 below 0.001 M; bins = distance >> shift, 2^shift the next
    power of two above template^2."""
    tsq = template * template
    shift = 0
    while (1 << shift) < tsq:
        shift += 1
    mult = (1 << shift) / tsq                                                        # almost_dist -> actual dist
    max_est = search * search * 255
    M = (2 ** 31 - 1) // max_est                                                     # fixed_point_mult: 19,096 for a 21x21 search window
    max_dist = 255 * 255 * C
    nbins = ((max_dist * tsq) >> shift) + 1                                          # bins of the 'almost' distance
    d = torch.arange(nbins, dtype=torch.float64) * mult
    wt = torch.round(M * torch.exp(-d / (h * h * C)))
    wt = torch.where(wt < 0.001 * M, torch.zeros_like(wt), wt).to(torch.int64)
    return wt, shift


def fast_nl_means_u8(planes, h, template=7, search=21, rows_per_block=0, use_hip=None):
    """cv2.fastNlMeansDenoising on [C, H, W] uint8 planes treated as ONE C-channel image (C = 1: the L plane, C = 2: the ab
    planes of fastNlMeansDenoisingColored): for every pixel and every offset of the search window the summed squared
    difference of the template windows (over all channels) is binned (>> 6 for the 7x7 template, as OpenCV's
    'almost' distance), turned into an integer weight round(M exp(-d / (h^2 C))) (0 below 0.001 M) and the pixels of the
    search window are averaged with these weights in integer arithmetic (rounded division).

    On the ROCm device with the reference's window sizes (7, 21: nesr/nesr.py:674) this is one HIP kernel
    (csrc/imgproc.hip, nesr_nl_means_u8); the torch composition below is kept for other sizes, the CPU and the tests
    (use_hip=False) -- the two agree bit for bit."""
    C, H, W = planes.shape
    wt, shift = nl_means_weights(C, h, template, search)
    nbins = wt.numel()
    if use_hip is None:
        use_hip = planes.device.type == "cuda" and template == 7 and search == 21 and 1 <= C <= 3
    if use_hip:
        import ctypes
        from . import _lib
        nz = int((wt != 0).sum().item())                                             # the weights fall monotonically to 0: a short table is enough
        lut = wt[:max(nz + 1, 1)].to(torch.int32).to(planes.device)
        src = planes.contiguous()
        out = torch.empty_like(src)
        index = src.device.index if src.device.index is not None else torch.cuda.current_device()
        with torch.cuda.device(src.device):
            stream = torch.cuda.current_stream(src.device).cuda_stream
            _lib.check(_lib.load().nesr_nl_means_u8(index, ctypes.c_void_p(src.data_ptr()), C, H, W, template, search, ctypes.c_void_p(lut.data_ptr()),
                                                    lut.numel(), ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(stream)), "nesr_nl_means_u8")
        return out
    tr, sr = template // 2, search // 2
    border = tr + sr
    x = _reflect101_pad(planes.long(), border)                                      # [C, H + 2b, W + 2b]
    wt = wt.to(planes.device)
    centre = x[:, sr:sr + H + 2 * tr, sr:sr + W + 2 * tr]                            # template-padded view of the image
    acc = torch.zeros((C, H, W), dtype=torch.int64, device=planes.device)
    wsum = torch.zeros((H, W), dtype=torch.int64, device=planes.device)
    box = torch.ones((1, 1, template, template), dtype=torch.float32, device=planes.device)   # sums < 2^24: exact in f32
    for dy in range(-sr, sr + 1):
        for dx in range(-sr, sr + 1):
            other = x[:, sr + dy:sr + dy + H + 2 * tr, sr + dx:sr + dx + W + 2 * tr]
            sq = ((centre - other) ** 2).sum(0).to(torch.float32)
            dist = F.conv2d(sq[None, None], box)[0, 0].to(torch.int64)               # [H, W] template sums
            wgt = wt[(dist >> shift).clamp_max(nbins - 1)]
            wsum += wgt
            acc += wgt[None] * x[:, border + dy:border + dy + H, border + dx:border + dx + W]
    out = (acc + (wsum // 2)[None]) // wsum.clamp_min(1)[None]
    return out.clamp_(0, 255).to(torch.uint8)


def fast_nl_means_colored_u8(img, h, h_color, template=7, search=21):
    """cv2.fastNlMeansDenoisingColored(img, None, h, hColor, template, search): COLOR_LBGR2Lab (no gamma, channel 0 taken
    as blue -- whatever order the caller's image is in: the reference hands it RGB, nesr/nesr.py:674), non-local means on
    L with h and on (a, b) with hColor, COLOR_Lab2LBGR."""
    lab = rgb2lab_u8(img, linear=True, first_is_blue=True)
    p = lab.permute(2, 0, 1).contiguous()
    L = fast_nl_means_u8(p[0:1], h, template, search)
    ab = fast_nl_means_u8(p[1:3], h_color, template, search)
    return lab2rgb_u8(torch.cat([L, ab], 0).permute(1, 2, 0), linear=True, first_is_blue=True)


# ----------------------------------------------------------------------------------------------- the pipeline's filters
def preprocess_image(img, denoise_level=0.5):
    """SuperResolutionPipeline._preprocess_image (nesr/nesr.py:668-689) on an HWC uint8 RGB device tensor."""
    if denoise_level > 0:
        strength = denoise_level * 10
        img = fast_nl_means_colored_u8(img, strength, strength, 7, 21)
    lab = rgb2lab_u8(img)                                                            # COLOR_RGB2LAB
    L = clahe_u8(lab[..., 0].contiguous(), 2.0, (8, 8))
    lab = torch.cat([L[..., None], lab[..., 1:]], -1)
    return lab2rgb_u8(lab)                                                           # COLOR_LAB2RGB


def postprocess_image(img, adaptive_sharpening=True):
    """SuperResolutionPipeline._postprocess_image (nesr/nesr.py:1056-1084): unsharp (1.5 img - 0.5 blur_3) where the local
    detail |gray - blur_2(gray)| exceeds 10, the image itself elsewhere."""
    if not adaptive_sharpening:
        return img
    gray = rgb2gray_u8(img)
    variance = (gray.to(torch.int16) - gaussian_blur_u8(gray, 2.0).to(torch.int16)).clamp_(0, 255)   # cv2.subtract saturates; convertScaleAbs keeps it
    blurred = gaussian_blur_u8(img, 3.0)
    sharpened = torch.round(img.float() * 1.5 - blurred.float() * 0.5).clamp_(0, 255).to(torch.uint8)   # addWeighted: saturate_cast<uchar>
    mask = (variance > 10)[..., None]
    return torch.where(mask, sharpened, img)
