#!/usr/bin/env python3
"""Generates the committed golden vectors under tests/golden/ from the torch-CPU oracle.

PARITY UNPINNED: the reference has no tests, fixtures or golden vectors of its own
 and its RRDBNet / RealESRGANer implementations (basicsr, realesrgan) are
absent, so these vectors are outputs of the oracle's restatement, not of the reference itself.
They pin the oracle against drift and give the GPU path a fixed target.

Only data is written: seeds, small input arrays and expected outputs (npz).  The one
reference-held input is a 64x96 crop of images/test.jpeg (the reference's only test asset),
stored as decoded pixels; everything else is regenerated from numpy PCG64 seeds.

    python oracle/make_golden.py            # rewrites tests/golden/*.npz
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

from neural_enhanced_super_resolution_amd.synth import synthetic_frame, synthetic_state_dict  # noqa: E402
from oracle.realesrganer_ref import RealESRGANerRef  # noqa: E402
from oracle.rrdbnet_ref import RRDBNetRef  # noqa: E402

CONV_SHAPES = [(3, 64), (12, 64), (64, 32), (96, 32), (128, 32), (160, 32), (192, 64), (64, 64), (64, 3)]
FLOAT_CASES = ("tile0", "tile32_prepad10_odd", "bgra")
NET_MODES = {"x2plus": (3, 2), "x4plus": (3, 4), "nesr12": (12, 4)}


def conv_case(cin, cout, h=12, w=20):
    """Deterministic inputs of one conv layer case (numpy PCG64: identical on every host)."""
    rng = np.random.default_rng(cin * 1000 + cout)
    x = rng.standard_normal((1, cin, h, w), dtype=np.float32)
    wgt = (rng.standard_normal((cout, cin, 3, 3), dtype=np.float32) * np.float32((2.0 / (cin * 9)) ** 0.5))
    b = rng.standard_normal((cout,), dtype=np.float32) * np.float32(0.1)
    return x, wgt, b


def net_input(num_in_ch, h, w, seed):
    return np.random.default_rng(seed).random((1, num_in_ch, h, w), dtype=np.float32)


def wrapper_cases():
    """(name, ctor kwargs, input-kind) of the RealESRGANer.enhance golden cases."""
    return [
        ("tile0", dict(tile=0, tile_pad=10, pre_pad=0), "bgr"),
        ("tile32_pad10", dict(tile=32, tile_pad=10, pre_pad=0), "bgr"),
        ("tile0_prepad10", dict(tile=0, tile_pad=10, pre_pad=10), "bgr"),
        ("tile32_prepad10_odd", dict(tile=32, tile_pad=10, pre_pad=10), "bgr_odd"),
        ("tile0_odd", dict(tile=0, tile_pad=10, pre_pad=0), "bgr_odd"),
        ("gray", dict(tile=0, tile_pad=10, pre_pad=0), "gray"),
        ("bgra", dict(tile=32, tile_pad=10, pre_pad=0), "bgra"),
        ("u16", dict(tile=0, tile_pad=10, pre_pad=0), "u16"),
    ]


def wrapper_input(kind, crop):
    if kind == "bgr":
        return crop
    if kind == "bgr_odd":
        return np.ascontiguousarray(crop[:63, :95])
    if kind == "gray":
        return np.ascontiguousarray(crop[:, :, 1])
    if kind == "bgra":
        alpha = synthetic_frame(crop.shape[0], crop.shape[1], seed=5, channels=0)
        return np.concatenate([crop, alpha[:, :, None]], axis=2)
    if kind == "u16":
        return (crop.astype(np.uint16) * 257)
    raise ValueError(kind)


def load_test_crop():
    """64x96 crop of the reference's images/test.jpeg, BGR order (what cv2.imread would give)."""
    path = os.path.join(GOLDEN, "test_jpeg_crop_64x96_bgr.npy")
    ref_img = "/root/reference/images/test.jpeg"
    if os.path.exists(ref_img):
        from PIL import Image
        rgb = np.asarray(Image.open(ref_img).convert("RGB"))
        crop = np.ascontiguousarray(rgb[200:264, 180:276, ::-1])
        np.save(path, crop)
    return np.load(path)


def test_jpeg_full_case():
    """The reference's only asset, images/test.jpeg (512x512), through the oracle wrapper with
    standalone/direct_esrgan.py's settings (:104, :118-127: x2plus shape, tile=512, tile_pad=10, pre_pad=0, half=False),
    seeded synthetic 23-block weights.  Kept: the decoded input, every second pixel of the 1024x1024 result and a
    full-resolution 256x256 window of it."""
    path = os.path.join(GOLDEN, "test_jpeg_full.npz")
    ref_img = "/root/reference/images/test.jpeg"
    if not os.path.exists(ref_img):
        return np.load(path)
    from PIL import Image
    bgr = np.ascontiguousarray(np.asarray(Image.open(ref_img).convert("RGB"))[:, :, ::-1])
    sd = synthetic_state_dict(seed=0, num_in_ch=3, scale=2, num_block=23)
    up = RealESRGANerRef(scale=2, model_path={"params_ema": sd}, model=RRDBNetRef(3, 3, scale=2), tile=512, tile_pad=10, pre_pad=0)
    out, mode = up.enhance(bgr)
    assert out.shape == (1024, 1024, 3) and mode == "RGB"
    np.savez_compressed(path, input_bgr=bgr, out_strided=np.ascontiguousarray(out[::2, ::2]),
                        out_window=np.ascontiguousarray(out[384:640, 384:640]))
    return np.load(path)


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)

    # 1. per-layer conv cases
    out = {}
    for cin, cout in CONV_SHAPES:
        x, w, b = conv_case(cin, cout)
        y = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), padding=1)
        out[f"y_{cin}_{cout}"] = y.numpy()
        out[f"ylrelu_{cin}_{cout}"] = F.leaky_relu(y, 0.2).numpy()
    x, w, b = conv_case(64, 64)
    out["yup_64_64"] = F.leaky_relu(F.conv2d(F.interpolate(torch.from_numpy(x), scale_factor=2, mode="nearest"),
                                             torch.from_numpy(w), torch.from_numpy(b), padding=1), 0.2).numpy()
    np.savez_compressed(os.path.join(GOLDEN, "conv_layers.npz"), **out)

    # 2. two-block networks in the three modes, even and ragged sizes
    out = {}
    for name, (cin, scale) in NET_MODES.items():
        sd = synthetic_state_dict(seed=3, num_in_ch=cin, scale=scale, num_block=2)
        net = RRDBNetRef(cin, 3, scale=scale, num_block=2)
        net.load_state_dict(sd)
        for (h, w) in ((32, 48), (34, 46)):
            with torch.no_grad():
                out[f"{name}_{h}x{w}"] = net(torch.from_numpy(net_input(cin, h, w, seed=7))).numpy()
    np.savez_compressed(os.path.join(GOLDEN, "mininet.npz"), **out)

    # 3. full depth (23 blocks) x2plus and x4plus on 64x64 / 32x32
    out = {}
    for name, (cin, scale, hw) in {"x2plus": (3, 2, 64), "x4plus": (3, 4, 32)}.items():
        sd = synthetic_state_dict(seed=0, num_in_ch=cin, scale=scale, num_block=23)
        net = RRDBNetRef(cin, 3, scale=scale, num_block=23)
        net.load_state_dict(sd)
        with torch.no_grad():
            out[name] = net(torch.from_numpy(net_input(cin, hw, hw, seed=11))).numpy()
    np.savez_compressed(os.path.join(GOLDEN, "fulldepth.npz"), **out)

    # 4. RealESRGANer.enhance cases on the test.jpeg crop (2-block x2plus net: the wrapper logic
    #    does not depend on depth) + one full-depth case
    crop = load_test_crop()
    out = {}
    sd2 = synthetic_state_dict(seed=3, num_in_ch=3, scale=2, num_block=2)
    for name, kw, kind in wrapper_cases():
        up = RealESRGANerRef(scale=2, model_path={"params_ema": sd2}, model=RRDBNetRef(3, 3, scale=2, num_block=2), **kw)
        img = wrapper_input(kind, crop)
        q, mode = up.enhance(img)
        f, _, _ = up.enhance_float(img)
        out[f"{name}_q"] = q
        if name in FLOAT_CASES:   # float (pre-quantisation) image kept for a few cases only: size
            out[f"{name}_f"] = f.astype(np.float32)
        out[f"{name}_mode"] = np.array(mode)
    sd23 = synthetic_state_dict(seed=0, num_in_ch=3, scale=2, num_block=23)
    up = RealESRGANerRef(scale=2, model_path={"params_ema": sd23}, model=RRDBNetRef(3, 3, scale=2), tile=32, tile_pad=10, pre_pad=0)
    q, _ = up.enhance(crop)
    f, _, _ = up.enhance_float(crop)
    out["full23_tile32_q"] = q
    # x4plus wrapper (scale=4: no mod pad) on a 24x40 corner
    sd4 = synthetic_state_dict(seed=3, num_in_ch=3, scale=4, num_block=2)
    up = RealESRGANerRef(scale=4, model_path={"params": sd4}, model=RRDBNetRef(3, 3, scale=4, num_block=2), tile=16, tile_pad=4, pre_pad=3)
    q, _ = up.enhance(np.ascontiguousarray(crop[:24, :40]))
    out["x4_tile16_q"] = q
    np.savez_compressed(os.path.join(GOLDEN, "wrapper.npz"), **out)

    test_jpeg_full_case()

    for f in sorted(os.listdir(GOLDEN)):
        print(f"{f:40s} {os.path.getsize(os.path.join(GOLDEN, f)) / 1024:8.1f} KiB")


if __name__ == "__main__":
    main()
