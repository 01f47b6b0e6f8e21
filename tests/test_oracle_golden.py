"""CPU: pins the oracle (torch-CPU restatement) -- against the committed golden vectors and against
the structural facts the reference itself records.  PARITY UNPINNED beyond that: the reference has
no tests or golden vectors for this path."""
import math
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import rrdbnet_ref as R
from oracle.realesrganer_ref import RealESRGANerRef
from oracle.rrdbnet_ref import RRDBNetRef

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
import make_golden as G  # noqa: E402

from neural_enhanced_super_resolution_amd.synth import synthetic_state_dict  # noqa: E402

TOL = 2e-5   # oneDNN may pick different kernels on another host / thread count


def test_param_count_matches_checkpoint_size():
    """16,703,171 f32 parameters = 66,812,684 B, consistent with the 67,010,191-byte
    RealESRGAN_x2plus.pth the reference records (nesr/utils/downloader.py:25; the rest is pickle framing)."""
    n = R.num_params(num_in_ch=3, scale=2)
    assert n == 16_703_171
    assert 0 < 67_010_191 - 4 * n < 300_000
    assert R.num_params(num_in_ch=3, scale=4) == 16_697_987
    assert len(R.state_dict_spec(3, 3, 2)) == 702
    # the nesr quirk (nesr/nesr.py:216: num_in_ch=12 without scale=2) has the same tensors as x2plus
    assert R.state_dict_spec(12, 3, 4) == R.state_dict_spec(3, 3, 2)


def test_macs_per_pixel():
    assert R.macs_per_internal_pixel(scale=2) == 17_932_032
    assert R.macs_per_internal_pixel(scale=4) == 17_926_848


def test_pixel_unshuffle_is_torch_pixel_unshuffle():
    x = torch.arange(2 * 3 * 8 * 12, dtype=torch.float32).reshape(2, 3, 8, 12)
    assert torch.equal(R.pixel_unshuffle(x, 2), F.pixel_unshuffle(x, 2))
    assert torch.equal(R.pixel_unshuffle(x, 4), F.pixel_unshuffle(x, 4))


def test_conv_layer_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "conv_layers.npz"))
    for cin, cout in G.CONV_SHAPES:
        x, w, b = G.conv_case(cin, cout)
        y = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), padding=1)
        assert np.abs(y.numpy() - g[f"y_{cin}_{cout}"]).max() < TOL
        assert np.abs(F.leaky_relu(y, 0.2).numpy() - g[f"ylrelu_{cin}_{cout}"]).max() < TOL


@pytest.mark.parametrize("mode", sorted(G.NET_MODES))
def test_mininet_golden(golden_dir, mode):
    g = np.load(os.path.join(golden_dir, "mininet.npz"))
    cin, scale = G.NET_MODES[mode]
    net = RRDBNetRef(cin, 3, scale=scale, num_block=2)
    net.load_state_dict(synthetic_state_dict(seed=3, num_in_ch=cin, scale=scale, num_block=2))
    for (h, w) in ((32, 48), (34, 46)):
        with torch.no_grad():
            y = net(torch.from_numpy(G.net_input(cin, h, w, seed=7))).numpy()
        s = 2 if scale == 2 else 4
        assert y.shape == (1, 3, h * s, w * s)
        assert np.abs(y - g[f"{mode}_{h}x{w}"]).max() < TOL


def test_fulldepth_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "fulldepth.npz"))
    net = RRDBNetRef(3, 3, scale=2)
    net.load_state_dict(synthetic_state_dict(seed=0, num_in_ch=3, scale=2))
    with torch.no_grad():
        y = net(torch.from_numpy(G.net_input(3, 64, 64, seed=11))).numpy()
    assert np.abs(y - g["x2plus"]).max() < 1e-4


def test_wrapper_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "wrapper.npz"))
    crop = np.load(os.path.join(golden_dir, "test_jpeg_crop_64x96_bgr.npy"))
    sd2 = synthetic_state_dict(seed=3, num_in_ch=3, scale=2, num_block=2)
    for name, kw, kind in G.wrapper_cases():
        up = RealESRGANerRef(scale=2, model_path={"params_ema": sd2}, model=RRDBNetRef(3, 3, scale=2, num_block=2), **kw)
        img = G.wrapper_input(kind, crop)
        q, mode = up.enhance(img)
        want = g[f"{name}_q"]
        assert mode == str(g[f"{name}_mode"])
        assert q.shape == want.shape and q.dtype == want.dtype
        assert q.shape[0] == img.shape[0] * 2 and q.shape[1] == img.shape[1] * 2
        lsb = 257 if q.dtype == np.uint16 else 1
        assert np.abs(q.astype(np.int64) - want.astype(np.int64)).max() <= lsb   # at most a rounding tie


def test_tiling_changes_output_but_pad_is_seamless_for_shallow_net(golden_dir):
    """tile_process pastes un-padded centres: with a receptive field (2 blocks: ~31 px at trunk
    resolution) wider than tile_pad the tiled result differs from the untiled one -- which is why the
    build reproduces upstream's tile grid instead of inventing a seamless one."""
    g = np.load(os.path.join(golden_dir, "wrapper.npz"))
    assert g["tile0_q"].shape == g["tile32_pad10_q"].shape
    assert (g["tile0_q"] != g["tile32_pad10_q"]).any()


def test_strict_load_rejects_wrong_arch():
    sd = synthetic_state_dict(seed=0, num_in_ch=3, scale=2, num_block=1)
    with pytest.raises(RuntimeError):
        RRDBNetRef(3, 3, scale=4, num_block=1).load_state_dict(sd)   # 3-ch conv_first vs 12-ch weights


def test_oracle_reproduces_the_full_test_jpeg_fixture(golden_dir):
    """images/test.jpeg (the reference's only asset, 512x512) at standalone/direct_esrgan.py's settings: the committed
    fixture is the oracle's output on the decoded pixels, and the oracle still produces it bit for bit."""
    from oracle.realesrganer_ref import RealESRGANerRef
    from oracle.rrdbnet_ref import RRDBNetRef
    g = np.load(os.path.join(golden_dir, "test_jpeg_full.npz"))
    bgr = g["input_bgr"]
    assert bgr.shape == (512, 512, 3) and bgr.dtype == np.uint8 and abs(float(bgr.mean()) - 98.3) < 1.0
    sd = synthetic_state_dict(seed=0, num_in_ch=3, scale=2, num_block=23)
    up = RealESRGANerRef(scale=2, model_path={"params_ema": sd}, model=RRDBNetRef(3, 3, scale=2), tile=512, tile_pad=10, pre_pad=0)
    out, mode = up.enhance(bgr)
    assert mode == "RGB" and out.shape == (1024, 1024, 3)
    assert np.array_equal(out[::2, ::2], g["out_strided"]) and np.array_equal(out[384:640, 384:640], g["out_window"])
