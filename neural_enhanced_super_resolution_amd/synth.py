"""Seeded synthetic weights and frames.

No checkpoint ships with the reference (models/weights/RealESRGAN_x2plus.pth is stripped,
/root/reference/.MISSING_LARGE_BLOBS) and there is no network, so benches, smoke and parity
tests use weights drawn from a seeded numpy PCG64 stream (bit-reproducible across hosts)
with upstream's init statistics: basicsr ``default_init_weights(scale=0.1)`` = kaiming-normal
x 0.1 for the dense-block convs, torch's default Conv2d init elsewhere.  ``rdb_gain`` scales the
dense-block weights so that the 23-block trunk contributes visibly to the output (the parity
tests would be blind to trunk errors with an all-but-identity trunk).

Frames follow SURVEY.md section 8(d): uint8 noise smoothed by a 5x5 box.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch

from .rrdbnet import rrdbnet_state_dict_spec


def synthetic_state_dict(seed=0, num_in_ch=3, num_out_ch=3, scale=4, num_feat=64, num_block=23,
                         num_grow_ch=32, rdb_gain=0.1 * 4.0, bias_std=0.02):
    rng = np.random.default_rng(seed)
    sd = OrderedDict()
    spec = rrdbnet_state_dict_spec(num_in_ch, num_out_ch, scale, num_feat, num_block, num_grow_ch)
    for key, shape in spec.items():
        if key.endswith(".weight"):
            fan_in = shape[1] * 9
            if key.startswith("body."):
                std = rdb_gain * math.sqrt(2.0 / fan_in)          # kaiming_normal_(a=0) * scale
                w = rng.standard_normal(shape, dtype=np.float32) * np.float32(std)
            else:
                bound = 1.0 / math.sqrt(fan_in)                    # kaiming_uniform_(a=sqrt(5))
                w = rng.uniform(-bound, bound, size=shape).astype(np.float32)
            sd[key] = torch.from_numpy(w)
        else:
            sd[key] = torch.from_numpy((rng.standard_normal(shape, dtype=np.float32) * np.float32(bias_std)))
    if num_out_ch == 3:
        # centre the output in the displayable range so clamp/quantise paths see real variation
        sd["conv_last.bias"] = sd["conv_last.bias"] + 0.5
    return sd


def synthetic_frame(h, w, seed=0, channels=3):
    """uint8 HWC BGR frame: uniform noise smoothed by a 5x5 box (edge-replicated)."""
    rng = np.random.default_rng(1000 + seed)
    shape = (h, w, channels) if channels else (h, w)
    x = rng.integers(0, 256, size=shape, dtype=np.uint8).astype(np.float32)
    pad = [(2, 2), (2, 2)] + ([(0, 0)] if channels else [])
    xp = np.pad(x, pad, mode="edge")
    acc = np.zeros_like(x)
    for dy in range(5):
        for dx in range(5):
            acc += xp[dy:dy + h, dx:dx + w]
    # stretch back to a wide range: box-filtered noise has std ~ 74/5
    y = (acc / 25.0 - 127.5) * 4.0 + 127.5
    return np.clip(np.rint(y), 0, 255).astype(np.uint8)
