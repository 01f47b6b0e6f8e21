"""Oracle (test infrastructure): numpy restatement of the OpenCV calls on the reference's path --

  cv2.resize(INTER_LANCZOS4 | INTER_LINEAR)         realesrgan utils.py enhance() [UPSTREAM]; nesr/nesr.py:438-443
  cv2.GaussianBlur, cv2.cvtColor(RGB2GRAY | RGB2LAB | LAB2RGB), cv2.subtract / addWeighted / threshold
                                                    nesr/nesr.py:1056-1084 (_postprocess_image), :872
  cv2.fastNlMeansDenoisingColored, cv2.createCLAHE  nesr/nesr.py:668-689 (_preprocess_image)

PARITY UNPINNED: cv2 is absent from this image and from the GPU box and the reference holds no fixture of these calls.
Each function follows OpenCV's published algorithm (imgproc resize.cpp / smooth / color_lab / clahe.cpp,
photo fast_nlmeans_denoising_invoker.hpp) in plain numpy with explicit loops, written independently of the product's torch
code (neural_enhanced_super_resolution_amd/imgproc.py); the Lab conversions are float formulas where OpenCV's 8-bit path
uses lookup tables, so both sides may sit +-1 LSB from real OpenCV output.  Small images only (pure-Python loops).
"""
import math

import numpy as np


def _reflect101(p, n):
    """cv2.borderInterpolate(p, n, BORDER_REFLECT_101)."""
    if n == 1:
        return 0
    while p < 0 or p >= n:
        p = -p if p < 0 else 2 * (n - 1) - p
    return p


def _lanczos_weights(x):
    if x < np.finfo(np.float32).eps:
        return [0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0]
    s45 = 0.70710678118654752440084436210485
    cs = [(1, 0), (-s45, -s45), (0, 1), (s45, -s45), (-1, 0), (s45, s45), (0, -1), (-s45, s45)]
    y0 = -(x + 3) * math.pi * 0.25
    s0, c0 = math.sin(y0), math.cos(y0)
    co = np.zeros(8, np.float32)
    for i in range(8):
        y = -(x + 3 - i) * math.pi * 0.25
        co[i] = np.float32((cs[i][0] * s0 + cs[i][1] * c0) / (y * y))
    return list((co * np.float32(1.0 / co.sum(dtype=np.float32))).astype(np.float32))


def _resize_axis(n_in, n_out, taps, first, weights):
    table = []
    scale = n_in / n_out
    for d in range(n_out):
        pos = np.float32((d + 0.5) * scale - 0.5)
        i0 = math.floor(pos)
        frac = np.float32(pos - np.float32(i0))
        idx = [min(max(i0 + first + k, 0), n_in - 1) for k in range(taps)]
        table.append((idx, weights(frac, i0)))
    return table


def resize_lanczos4(img, out_h, out_w):
    """cv2.resize(img, (out_w, out_h), interpolation=cv2.INTER_LANCZOS4), HWC uint8 (fixed point) or uint16 (float)."""
    h, w, c = img.shape
    fixed = img.dtype == np.uint8
    tx = _resize_axis(w, out_w, 8, -3, lambda f, i0: _lanczos_weights(f))
    ty = _resize_axis(h, out_h, 8, -3, lambda f, i0: _lanczos_weights(f))
    if fixed:
        q = lambda ws: [int(min(max(int(np.rint(np.float32(v) * np.float32(2048.0))), -32768), 32767)) for v in ws]   # noqa: E731
        rows = np.zeros((h, out_w, c), np.int64)
        for d, (idx, ws) in enumerate(tx):
            a = q(ws)
            for k in range(8):
                rows[:, d, :] += a[k] * img[:, idx[k], :].astype(np.int64)
        out = np.zeros((out_h, out_w, c), np.int64)
        for d, (idx, ws) in enumerate(ty):
            b = q(ws)
            for k in range(8):
                out[d] += b[k] * rows[idx[k]]
        return np.clip((out + (1 << 21)) >> 22, 0, 255).astype(np.uint8)
    rows = np.zeros((h, out_w, c), np.float32)
    for d, (idx, ws) in enumerate(tx):
        acc = np.zeros((h, c), np.float32)
        for k in range(8):
            acc = acc + np.float32(ws[k]) * img[:, idx[k], :].astype(np.float32)
        rows[:, d, :] = acc
    out = np.zeros((out_h, out_w, c), np.float32)
    for d, (idx, ws) in enumerate(ty):
        acc = np.zeros((out_w, c), np.float32)
        for k in range(8):
            acc = acc + np.float32(ws[k]) * rows[idx[k]]
        out[d] = acc
    return np.clip(np.rint(out), 0, 65535).astype(np.uint16)


def resize_linear_f32(img, out_h, out_w):
    """cv2.resize(img, (out_w, out_h), interpolation=cv2.INTER_LINEAR) on float32 HW."""
    h, w = img.shape

    def lin(f, i0, n):
        if i0 < 0:
            return (0, 0, np.float32(0))
        if i0 >= n - 1:
            return (n - 1, n - 1, np.float32(0))
        return (i0, i0 + 1, f)

    xs, ys = [], []
    for d in range(out_w):
        pos = np.float32((d + 0.5) * (w / out_w) - 0.5)
        i0 = math.floor(pos)
        xs.append(lin(np.float32(pos - np.float32(i0)), i0, w))
    for d in range(out_h):
        pos = np.float32((d + 0.5) * (h / out_h) - 0.5)
        i0 = math.floor(pos)
        ys.append(lin(np.float32(pos - np.float32(i0)), i0, h))
    rows = np.zeros((h, out_w), np.float32)
    for d, (a, b, f) in enumerate(xs):
        rows[:, d] = img[:, a] * (np.float32(1) - f) + img[:, b] * f
    out = np.zeros((out_h, out_w), np.float32)
    for d, (a, b, f) in enumerate(ys):
        out[d] = rows[a] * (np.float32(1) - f) + rows[b] * f
    return out


def gaussian_kernel_fixed(sigma, ksize=0):
    if ksize <= 0:
        ksize = int(round(sigma * 6 + 1)) | 1
    small = {1: [1.0], 3: [0.25, 0.5, 0.25], 5: [0.0625, 0.25, 0.375, 0.25, 0.0625],
             7: [0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125]}
    if sigma <= 0 and ksize in small:
        k = small[ksize]
    else:
        if sigma <= 0:
            sigma = 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8
        r = ksize // 2
        k = [math.exp(-(i - r) ** 2 / (2.0 * sigma * sigma)) for i in range(ksize)]
        t = sum(k)
        k = [v / t for v in k]
    q = [int(round(v * 256.0)) for v in k]
    q[len(q) // 2] += 256 - sum(q)
    return q


def gaussian_blur_u8(img, sigma, ksize=0):
    """cv2.GaussianBlur on HWC or HW uint8: 8-bit fixed-point taps, REFLECT_101, (v + 2^15) >> 16."""
    x = img[:, :, None] if img.ndim == 2 else img
    h, w, c = x.shape
    k = gaussian_kernel_fixed(sigma, ksize)
    r = len(k) // 2
    hs = np.zeros((h, w, c), np.int64)
    for col in range(w):
        for t in range(len(k)):
            hs[:, col, :] += k[t] * x[:, _reflect101(col + t - r, w), :].astype(np.int64)
    vs = np.zeros((h, w, c), np.int64)
    for row in range(h):
        for t in range(len(k)):
            vs[row] += k[t] * hs[_reflect101(row + t - r, h)]
    out = np.clip((vs + (1 << 15)) >> 16, 0, 255).astype(np.uint8)
    return out[:, :, 0] if img.ndim == 2 else out


def rgb2gray_u8(img):
    x = img.astype(np.int64)
    return ((x[..., 0] * 4899 + x[..., 1] * 9617 + x[..., 2] * 1868 + 8192) >> 14).astype(np.uint8)


_XN, _ZN = 0.950456, 1.088754


def _f(t):
    return np.where(t > 0.008856, np.cbrt(np.maximum(t, 1e-12)), 7.787 * t + 16.0 / 116.0)


def rgb2lab_u8(img, linear=False, first_is_blue=False):
    c = img.astype(np.float32) / np.float32(255.0)
    if not linear:
        c = np.where(c <= 0.04045, c / 12.92, np.power((c + 0.055) / 1.055, 2.4)).astype(np.float32)
    r, g, b = (c[..., 2], c[..., 1], c[..., 0]) if first_is_blue else (c[..., 0], c[..., 1], c[..., 2])
    X = (0.412453 * r + 0.357580 * g + 0.180423 * b) / _XN
    Y = 0.212671 * r + 0.715160 * g + 0.072169 * b
    Z = (0.019334 * r + 0.119193 * g + 0.950227 * b) / _ZN
    fx, fy, fz = _f(X), _f(Y), _f(Z)
    L = np.where(Y > 0.008856, 116.0 * fy - 16.0, 903.3 * Y)
    out = np.stack([L * 255.0 / 100.0, 500.0 * (fx - fy) + 128.0, 200.0 * (fy - fz) + 128.0], -1).astype(np.float32)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def lab2rgb_u8(lab, linear=False, first_is_blue=False):
    x = lab.astype(np.float32)
    L, a, b = x[..., 0] * np.float32(100.0 / 255.0), x[..., 1] - 128.0, x[..., 2] - 128.0
    fy = (L + 16.0) / 116.0
    Y = np.where(L <= 8.0, L / 903.3, fy ** 3)
    fy = np.where(L <= 8.0, 7.787 * Y + 16.0 / 116.0, fy)
    fx, fz = fy + a / 500.0, fy - b / 200.0
    inv = lambda f: np.where(f <= 6.0 / 29.0, (f - 16.0 / 116.0) / 7.787, f ** 3)   # noqa: E731
    X, Z = inv(fx) * _XN, inv(fz) * _ZN
    r = 3.240479 * X - 1.537150 * Y - 0.498535 * Z
    g = -0.969256 * X + 1.875991 * Y + 0.041556 * Z
    bl = 0.055648 * X - 0.204043 * Y + 1.057311 * Z
    c = np.clip(np.stack([bl, g, r] if first_is_blue else [r, g, bl], -1), 0, 1).astype(np.float32)
    if not linear:
        c = np.where(c <= 0.0031308, c * 12.92, 1.055 * np.power(np.maximum(c, 1e-12), 1.0 / 2.4) - 0.055).astype(np.float32)
    return np.clip(np.rint(c * np.float32(255.0)), 0, 255).astype(np.uint8)


def clahe_tile_size(h, w, grid=(8, 8)):
    """(tile height, tile width) cv2's CLAHE uses for an h x w image (clahe.cpp, CLAHE_Impl::apply)."""
    gx, gy = grid
    ph, pw = (gy - h % gy, gx - w % gx) if (h % gy or w % gx) else (0, 0)
    return (h + ph) // gy, (w + pw) // gx


def clahe_u8(gray, clip_limit=2.0, grid=(8, 8)):
    """cv2.createCLAHE(clip_limit, grid).apply(gray) (clahe.cpp: CLAHE_CalcLut_Body + CLAHE_Interpolation_Body)."""
    h, w = gray.shape
    gx, gy = grid
    # clahe.cpp: only when BOTH sides divide by the grid is the image used as it is; otherwise copyMakeBorder pads the bottom by
    # tilesY - h % tilesY and the right by tilesX - w % tilesX -- a side that does divide gets a whole extra tilesY / tilesX pixels
    ph, pw = (gy - h % gy, gx - w % gx) if (h % gy or w % gx) else (0, 0)
    src = gray
    if ph or pw:
        src = np.zeros((h + ph, w + pw), np.uint8)
        for y in range(h + ph):
            for x in range(w + pw):
                src[y, x] = gray[_reflect101(y, h), _reflect101(x, w)]
    th, tw = src.shape[0] // gy, src.shape[1] // gx
    area = th * tw
    clip = max(int(clip_limit * area / 256.0), 1)
    scale = np.float32(255.0) / np.float32(area)
    luts = np.zeros((gy, gx, 256), np.float32)
    for ty in range(gy):
        for tx in range(gx):
            hist = np.bincount(src[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw].ravel(), minlength=256).astype(np.int64)
            clipped = 0
            for i in range(256):
                if hist[i] > clip:
                    clipped += hist[i] - clip
                    hist[i] = clip
            batch, residual = clipped // 256, clipped % 256
            hist += batch
            if residual:
                step = max(256 // residual, 1)
                i = 0
                while i < 256 and residual > 0:
                    hist[i] += 1
                    residual -= 1
                    i += step
            s = 0
            for i in range(256):
                s += hist[i]
                luts[ty, tx, i] = np.clip(np.rint(np.float32(s) * scale), 0, 255)
    out = np.zeros((h, w), np.uint8)
    inv_th, inv_tw = np.float32(1.0 / th), np.float32(1.0 / tw)
    for y in range(h):
        tyf = np.float32(y) * inv_th - np.float32(0.5)
        ty1 = math.floor(tyf)
        ya = np.float32(tyf - ty1)
        ty2 = min(ty1 + 1, gy - 1)
        ty1 = max(ty1, 0)
        for x in range(w):
            txf = np.float32(x) * inv_tw - np.float32(0.5)
            tx1 = math.floor(txf)
            xa = np.float32(txf - tx1)
            tx2 = min(tx1 + 1, gx - 1)
            tx1 = max(tx1, 0)
            v = gray[y, x]
            res = (luts[ty1, tx1, v] * (1 - xa) + luts[ty1, tx2, v] * xa) * (1 - ya) + \
                  (luts[ty2, tx1, v] * (1 - xa) + luts[ty2, tx2, v] * xa) * ya
            out[y, x] = int(np.clip(np.rint(np.float32(res)), 0, 255))
    return out


def fast_nl_means_u8(planes, h, template=7, search=21):
    """cv2.fastNlMeansDenoising on a [C, H, W] uint8 image of C channels (FastNlMeansDenoisingInvoker, DistSquared)."""
    C, H, W = planes.shape
    tr, sr = template // 2, search // 2
    b = tr + sr
    ext = np.zeros((C, H + 2 * b, W + 2 * b), np.int64)
    for y in range(H + 2 * b):
        for x in range(W + 2 * b):
            ext[:, y, x] = planes[:, _reflect101(y - b, H), _reflect101(x - b, W)]
    tsq = template * template
    shift = 0
    while (1 << shift) < tsq:
        shift += 1
    mult = (1 << shift) / tsq
    M = (2 ** 31 - 1) // (search * search * 255)
    out = np.zeros((C, H, W), np.uint8)
    wcache = {}
    for y in range(H):
        for x in range(W):
            wsum, est = 0, np.zeros(C, np.int64)
            t0 = ext[:, y + b - tr:y + b + tr + 1, x + b - tr:x + b + tr + 1]
            for dy in range(-sr, sr + 1):
                for dx in range(-sr, sr + 1):
                    t1 = ext[:, y + b + dy - tr:y + b + dy + tr + 1, x + b + dx - tr:x + b + dx + tr + 1]
                    dist = int(((t0 - t1) ** 2).sum())
                    ad = dist >> shift
                    wgt = wcache.get(ad)
                    if wgt is None:
                        wv = math.exp(-(ad * mult) / (h * h * C))
                        wgt = int(round(M * wv))
                        if wgt < 0.001 * M:
                            wgt = 0
                        wcache[ad] = wgt
                    wsum += wgt
                    est += wgt * ext[:, y + b + dy, x + b + dx]
            out[:, y, x] = np.clip((est + wsum // 2) // max(wsum, 1), 0, 255)
    return out


def fast_nl_means_colored_u8(img, h, h_color, template=7, search=21):
    lab = rgb2lab_u8(img, linear=True, first_is_blue=True)
    p = np.transpose(lab, (2, 0, 1))
    L = fast_nl_means_u8(p[0:1], h, template, search)
    ab = fast_nl_means_u8(p[1:3], h_color, template, search)
    return lab2rgb_u8(np.transpose(np.concatenate([L, ab], 0), (1, 2, 0)), linear=True, first_is_blue=True)


def preprocess_image(img, denoise_level=0.5):
    """nesr/nesr.py:668-689."""
    if denoise_level > 0:
        s = denoise_level * 10
        img = fast_nl_means_colored_u8(img, s, s, 7, 21)
    lab = rgb2lab_u8(img)
    lab = np.concatenate([clahe_u8(np.ascontiguousarray(lab[..., 0]), 2.0, (8, 8))[..., None], lab[..., 1:]], -1)
    return lab2rgb_u8(lab)


def postprocess_image(img, adaptive_sharpening=True):
    """nesr/nesr.py:1056-1084."""
    if not adaptive_sharpening:
        return img
    gray = rgb2gray_u8(img)
    variance = np.clip(gray.astype(np.int32) - gaussian_blur_u8(gray, 2.0).astype(np.int32), 0, 255).astype(np.uint8)   # subtract saturates
    blurred = gaussian_blur_u8(img, 3.0)
    sharpened = np.clip(np.rint(img.astype(np.float32) * np.float32(1.5) - blurred.astype(np.float32) * np.float32(0.5)), 0, 255).astype(np.uint8)
    binary = np.where(variance > 10, 255, 0).astype(np.uint8)                       # threshold(variance, 10, 255, THRESH_BINARY)
    alpha = binary.astype(np.float32) / 255.0
    result = np.zeros_like(img)
    for c in range(3):
        result[:, :, c] = img[:, :, c] * (1 - alpha) + sharpened[:, :, c] * alpha      # float -> uint8 assignment truncates
    return result.astype(np.uint8)
