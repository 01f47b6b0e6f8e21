"""Oracle (test infrastructure): torch-CPU restatement of basicsr 1.4.2 ``RRDBNet``.

PARITY UNPINNED (see oracle/__init__.py): basicsr is not vendored in the reference.
Every function cites the upstream construct it restates and the reference call site
that fixes its arguments.

Reference call sites that pin the architecture:
  nesr/nesr.py:216                 RRDBNet(num_in_ch=12, num_out_ch=3, num_feat=64, num_block=23, num_grow_ch=32)
  standalone/direct_esrgan.py:104  RRDBNet(num_in_ch=3,  num_out_ch=3, num_feat=64, num_block=23, num_grow_ch=32)
  standalone/superres_project.py:69  same as direct_esrgan
(all without ``scale=``, so upstream's default scale=4 applies; the canonical x2plus net is
scale=2 -> pixel_unshuffle(2) -> conv_first with 12 input channels).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.2  # upstream: nn.LeakyReLU(negative_slope=0.2)


def conv_first_in_ch(num_in_ch: int, scale: int) -> int:
    """Upstream RRDBNet.__init__: scale==2 -> num_in_ch*4, scale==1 -> num_in_ch*16."""
    if scale == 2:
        return num_in_ch * 4
    if scale == 1:
        return num_in_ch * 16
    return num_in_ch


def state_dict_spec(num_in_ch=3, num_out_ch=3, scale=4, num_feat=64, num_block=23, num_grow_ch=32):
    """Ordered {state_dict key: shape} of upstream RRDBNet.

    conv_first, body.{i}.rdb{1,2,3}.conv{1..5}, conv_body, conv_up1, conv_up2, conv_hr,
    conv_last; each with .weight (OIHW) and .bias.
    """
    spec = OrderedDict()

    def conv(name, cin, cout):
        spec[name + ".weight"] = (cout, cin, 3, 3)
        spec[name + ".bias"] = (cout,)

    conv("conv_first", conv_first_in_ch(num_in_ch, scale), num_feat)
    for b in range(num_block):
        for r in (1, 2, 3):
            for k in range(1, 5):
                conv(f"body.{b}.rdb{r}.conv{k}", num_feat + (k - 1) * num_grow_ch, num_grow_ch)
            conv(f"body.{b}.rdb{r}.conv5", num_feat + 4 * num_grow_ch, num_feat)
    conv("conv_body", num_feat, num_feat)
    conv("conv_up1", num_feat, num_feat)
    conv("conv_up2", num_feat, num_feat)
    conv("conv_hr", num_feat, num_feat)
    conv("conv_last", num_feat, num_out_ch)
    return spec


def num_params(**kw) -> int:
    return sum(math.prod(s) for s in state_dict_spec(**kw).values())


def macs_per_internal_pixel(num_in_ch=3, num_out_ch=3, scale=4, num_feat=64, num_block=23, num_grow_ch=32) -> int:
    """MACs per trunk-resolution pixel (17,932,032 for x2plus)."""
    cin0 = conv_first_in_ch(num_in_ch, scale)
    rdb = sum(9 * (num_feat + k * num_grow_ch) * num_grow_ch for k in range(4)) + 9 * (num_feat + 4 * num_grow_ch) * num_feat
    m = 9 * cin0 * num_feat + num_block * 3 * rdb + 9 * num_feat * num_feat
    m += 4 * 9 * num_feat * num_feat            # conv_up1 at 2x
    m += 16 * 9 * num_feat * num_feat * 2       # conv_up2, conv_hr at 4x
    m += 16 * 9 * num_feat * num_out_ch         # conv_last at 4x
    return m


def pixel_unshuffle(x: torch.Tensor, scale: int) -> torch.Tensor:
    """Upstream basicsr.archs.arch_util.pixel_unshuffle: view(b,c,h,s,w,s).permute(0,1,3,5,2,4)."""
    b, c, hh, hw = x.size()
    assert hh % scale == 0 and hw % scale == 0
    h, w = hh // scale, hw // scale
    return x.view(b, c, h, scale, w, scale).permute(0, 1, 3, 5, 2, 4).reshape(b, c * scale * scale, h, w)


def _conv(x, sd, name):
    return F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"], stride=1, padding=1)


def _lrelu(x):
    return F.leaky_relu(x, LRELU_SLOPE)


def rdb_forward(x, sd, prefix):
    """Upstream ResidualDenseBlock.forward."""
    x1 = _lrelu(_conv(x, sd, prefix + ".conv1"))
    x2 = _lrelu(_conv(torch.cat((x, x1), 1), sd, prefix + ".conv2"))
    x3 = _lrelu(_conv(torch.cat((x, x1, x2), 1), sd, prefix + ".conv3"))
    x4 = _lrelu(_conv(torch.cat((x, x1, x2, x3), 1), sd, prefix + ".conv4"))
    x5 = _conv(torch.cat((x, x1, x2, x3, x4), 1), sd, prefix + ".conv5")
    return x5 * 0.2 + x


def rrdb_forward(x, sd, prefix):
    """Upstream RRDB.forward: rdb3(rdb2(rdb1(x))) * 0.2 + x (section 8 a5)."""
    out = rdb_forward(x, sd, prefix + ".rdb1")
    out = rdb_forward(out, sd, prefix + ".rdb2")
    out = rdb_forward(out, sd, prefix + ".rdb3")
    return out * 0.2 + x


def rrdbnet_forward(x, sd, scale=4, num_block=23):
    """Upstream RRDBNet.forward (section 3.2): x NCHW float -> NCHW float, 4x the trunk resolution."""
    if scale == 2:
        feat = pixel_unshuffle(x, 2)
    elif scale == 1:
        feat = pixel_unshuffle(x, 4)
    else:
        feat = x
    feat = _conv(feat, sd, "conv_first")
    trunk = feat
    for b in range(num_block):
        trunk = rrdb_forward(trunk, sd, f"body.{b}")
    feat = feat + _conv(trunk, sd, "conv_body")
    feat = _lrelu(_conv(F.interpolate(feat, scale_factor=2, mode="nearest"), sd, "conv_up1"))
    feat = _lrelu(_conv(F.interpolate(feat, scale_factor=2, mode="nearest"), sd, "conv_up2"))
    return _conv(_lrelu(_conv(feat, sd, "conv_hr")), sd, "conv_last")


class RRDBNetRef(torch.nn.Module):
    """nn.Module shell around :func:`rrdbnet_forward` with upstream's ctor signature and
    state_dict key names, so the wrapper oracle (and host-logic tests) can treat it exactly
    like ``basicsr.archs.rrdbnet_arch.RRDBNet``."""

    def __init__(self, num_in_ch, num_out_ch, scale=4, num_feat=64, num_block=23, num_grow_ch=32):
        super().__init__()
        self.scale = scale
        self.num_block = num_block
        self._spec = state_dict_spec(num_in_ch, num_out_ch, scale, num_feat, num_block, num_grow_ch)
        self._names = {}
        for i, (k, shape) in enumerate(self._spec.items()):
            pname = "p%d" % i
            self._names[k] = pname
            self.register_parameter(pname, torch.nn.Parameter(torch.zeros(shape), requires_grad=False))

    # upstream key names in, upstream key names out
    def state_dict(self, *a, **k):
        return OrderedDict((key, getattr(self, p).detach()) for key, p in self._names.items())

    def load_state_dict(self, sd, strict=True):
        missing = [k for k in self._spec if k not in sd]
        unexpected = [k for k in sd if k not in self._spec]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict: missing {missing[:3]} unexpected {unexpected[:3]}")
        for k, shape in self._spec.items():
            if k in sd:
                if tuple(sd[k].shape) != tuple(shape):
                    raise RuntimeError(f"size mismatch for {k}: {tuple(sd[k].shape)} vs {tuple(shape)}")
                getattr(self, self._names[k]).data.copy_(sd[k])

    def forward(self, x):
        return rrdbnet_forward(x, self.state_dict(), self.scale, self.num_block)
