"""GPU: the HIP path (through the C ABI) against the committed golden vectors and against the
oracle wrapper end to end.  fp32 tolerance = BASELINE.json's north-star criterion (1e-3 max abs on
the float image); the uint8 image may differ by one LSB where the float value sits on a rounding tie."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
import make_golden as G  # noqa: E402

from neural_enhanced_super_resolution_amd import RealESRGANer, RRDBNet, conv3x3  # noqa: E402
from neural_enhanced_super_resolution_amd.synth import synthetic_frame, synthetic_state_dict  # noqa: E402

TOL = 1e-3


def test_conv_layers_vs_golden(cuda_device, golden_dir):
    g = np.load(os.path.join(golden_dir, "conv_layers.npz"))
    for cin, cout in G.CONV_SHAPES:
        x, w, b = G.conv_case(cin, cout)
        xt, wt, bt = torch.from_numpy(x).to(cuda_device), torch.from_numpy(w), torch.from_numpy(b)
        assert np.abs(conv3x3(xt, wt, bt).cpu().numpy() - g[f"y_{cin}_{cout}"]).max() < 2e-5
        assert np.abs(conv3x3(xt, wt, bt, lrelu=True).cpu().numpy() - g[f"ylrelu_{cin}_{cout}"]).max() < 2e-5
    x, w, b = G.conv_case(64, 64)
    got = conv3x3(torch.from_numpy(x).to(cuda_device), torch.from_numpy(w), torch.from_numpy(b), lrelu=True, upsample=True)
    assert np.abs(got.cpu().numpy() - g["yup_64_64"]).max() < 2e-5


@pytest.mark.parametrize("mode", sorted(G.NET_MODES))
def test_mininet_vs_golden(cuda_device, golden_dir, mode):
    g = np.load(os.path.join(golden_dir, "mininet.npz"))
    cin, scale = G.NET_MODES[mode]
    net = RRDBNet(cin, 3, scale=scale, num_block=2)
    net.load_state_dict(synthetic_state_dict(seed=3, num_in_ch=cin, scale=scale, num_block=2))
    net.to(cuda_device)
    for (h, w) in ((32, 48), (34, 46)):
        y = net(torch.from_numpy(G.net_input(cin, h, w, seed=7)).to(cuda_device)).cpu().numpy()
        assert np.abs(y - g[f"{mode}_{h}x{w}"]).max() < 5e-5


@pytest.mark.parametrize("mode,hw", [("x2plus", 64), ("x4plus", 32)])
def test_fulldepth_vs_golden(cuda_device, golden_dir, mode, hw):
    g = np.load(os.path.join(golden_dir, "fulldepth.npz"))
    cin, scale = G.NET_MODES[mode]
    net = RRDBNet(cin, 3, scale=scale)
    net.load_state_dict(synthetic_state_dict(seed=0, num_in_ch=cin, scale=scale))
    net.to(cuda_device)
    y = net(torch.from_numpy(G.net_input(cin, hw, hw, seed=11)).to(cuda_device)).cpu().numpy()
    err = np.abs(y - g[mode]).max()
    print(f"{mode} full depth max abs err vs golden: {err:.3e}")
    assert err < TOL


def test_wrapper_cases_vs_golden(cuda_device, golden_dir):
    g = np.load(os.path.join(golden_dir, "wrapper.npz"))
    crop = np.load(os.path.join(golden_dir, "test_jpeg_crop_64x96_bgr.npy"))
    sd2 = synthetic_state_dict(seed=3, num_in_ch=3, scale=2, num_block=2)
    for name, kw, kind in G.wrapper_cases():
        up = RealESRGANer(scale=2, model_path={"params_ema": sd2}, model=RRDBNet(3, 3, scale=2, num_block=2),
                          half=False, device=cuda_device, **kw)
        img = G.wrapper_input(kind, crop)
        q, mode = up.enhance(img)
        want = g[f"{name}_q"]
        assert mode == str(g[f"{name}_mode"]) and q.shape == want.shape and q.dtype == want.dtype
        lsb = 257 if q.dtype == np.uint16 else 1
        diff = np.abs(q.astype(np.int64) - want.astype(np.int64))
        assert diff.max() <= lsb, (name, diff.max())
        # rounding flips only: an f32-level difference (~3e-6) moves a value across a rounding boundary of
        # the 8-bit grid (step 3.9e-3) about once in a thousand, of the 16-bit grid (step 1.5e-5) more often
        assert (diff > 0).mean() < (1e-2 if q.dtype == np.uint16 else 1e-3), (name, (diff > 0).mean())
        if f"{name}_f" in g.files:
            f, _, _ = up.enhance_float(img)
            assert np.abs(f - g[f"{name}_f"]).max() < TOL, name


def test_wrapper_full_depth_tiled_vs_golden(cuda_device, golden_dir):
    g = np.load(os.path.join(golden_dir, "wrapper.npz"))
    crop = np.load(os.path.join(golden_dir, "test_jpeg_crop_64x96_bgr.npy"))
    sd = synthetic_state_dict(seed=0, num_in_ch=3, scale=2)
    up = RealESRGANer(scale=2, model_path={"params_ema": sd}, model=RRDBNet(3, 3, scale=2), tile=32, tile_pad=10, pre_pad=0,
                      half=False, device=cuda_device)
    q, _ = up.enhance(crop)
    assert np.abs(q.astype(int) - g["full23_tile32_q"].astype(int)).max() <= 1


def test_x4plus_wrapper_vs_golden(cuda_device, golden_dir):
    g = np.load(os.path.join(golden_dir, "wrapper.npz"))
    crop = np.load(os.path.join(golden_dir, "test_jpeg_crop_64x96_bgr.npy"))
    sd4 = synthetic_state_dict(seed=3, num_in_ch=3, scale=4, num_block=2)
    up = RealESRGANer(scale=4, model_path={"params": sd4}, model=RRDBNet(3, 3, scale=4, num_block=2), tile=16, tile_pad=4,
                      pre_pad=3, device=cuda_device)
    q, _ = up.enhance(np.ascontiguousarray(crop[:24, :40]))
    assert q.shape == (96, 160, 3)
    assert np.abs(q.astype(int) - g["x4_tile16_q"].astype(int)).max() <= 1


def test_fused_u8_path_equals_float_path(cuda_device):
    """enhance() takes the fused u8 kernel path for plain 8-bit BGR frames; it must produce the
    bytes of the float path (same arithmetic, quantiser folded into conv_last's epilogue)."""
    sd = synthetic_state_dict(seed=3, num_in_ch=3, scale=2, num_block=2)
    up = RealESRGANer(scale=2, model_path={"params_ema": sd}, model=RRDBNet(3, 3, scale=2, num_block=2), tile=0, pre_pad=0,
                      device=cuda_device)
    img = synthetic_frame(40, 56, seed=9)
    assert up._fused_u8_ok(img)
    fused, _ = up.enhance(img)
    f, _, _ = up.enhance_float(img)
    assert np.array_equal(fused, (f * 255.0).round().astype(np.uint8))


def test_nesr_style_direct_model_call_and_truncating_quantiser(cuda_device):
    """nesr/nesr.py:845-903: builds a 12-channel input, calls upscaler.model(x) directly and
    quantises with clip(x*255).astype(uint8) (truncation)."""
    from oracle.rrdbnet_ref import RRDBNetRef
    sd = synthetic_state_dict(seed=2, num_in_ch=12, scale=4, num_block=2)
    model = RRDBNet(num_in_ch=12, num_out_ch=3, num_feat=64, num_block=2, num_grow_ch=32)
    up = RealESRGANer(scale=2, model_path={"params_ema": sd}, model=model, tile=0, tile_pad=0, pre_pad=0, half=False,
                      device="cuda")
    ref = RRDBNetRef(12, 3, scale=4, num_block=2)
    ref.load_state_dict(sd)
    rgb = synthetic_frame(24, 36, seed=4)
    t = torch.from_numpy(np.transpose(rgb[:, :, ::-1].copy(), (2, 0, 1))).float() / 255.0
    x12 = torch.cat([t, torch.clamp(t * 1.1, 0, 1), torch.clamp(t * 0.9, 0, 1), t], 0).unsqueeze(0)   # nesr.py:860-879 (blur slot = identity here)
    with torch.no_grad():
        m = up.model
        m.eval()
        out = m(x12.to("cuda")).squeeze().cpu().numpy()
        want = ref(x12).squeeze().numpy()
    assert out.shape == (3, 96, 144)
    assert np.abs(out - want).max() < TOL
    q = np.clip(np.transpose(out, (1, 2, 0)) * 255.0, 0, 255).astype(np.uint8)
    qw = np.clip(np.transpose(want, (1, 2, 0)) * 255.0, 0, 255).astype(np.uint8)
    assert np.abs(q.astype(int) - qw.astype(int)).max() <= 1


def test_linearity_property_at_full_size(cuda_device):
    """Size-independent property at BASELINE's full C2 size (the oracle takes seconds there, so
    this does not call it): one conv layer is linear, conv(a*x + b*y) == a*conv(x) + b*conv(y)
    (bias handled), on a 256x256x64 feature map."""
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 64, 256, 256, generator=g).to(cuda_device)
    y = torch.randn(1, 64, 256, 256, generator=g).to(cuda_device)
    w = torch.randn(64, 64, 3, 3, generator=g) * 0.05
    b = torch.zeros(64)
    lhs = conv3x3(1.5 * x - 0.5 * y, w, b)
    rhs = 1.5 * conv3x3(x, w, b) - 0.5 * conv3x3(y, w, b)
    assert (lhs - rhs).abs().max().item() < 1e-4


def test_linearity_property_at_full_size_f16_pair_form(cuda_device):
    """The same property for the default f32 form (operands as f16 pairs) at the full C2 layer size, plus
    homogeneity by a power of two (exact for the pair representation except where a lo half is an f16 subnormal)."""
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 64, 256, 256, generator=g).to(cuda_device)
    y = torch.randn(1, 64, 256, 256, generator=g).to(cuda_device)
    w = torch.randn(32, 64, 3, 3, generator=g) * 0.05
    b = torch.zeros(32)
    cx, cy = conv3x3(x, w, b, dtype="f32-split"), conv3x3(y, w, b, dtype="f32-split")
    lhs = conv3x3(1.5 * x - 0.5 * y, w, b, dtype="f32-split")
    assert (lhs - (1.5 * cx - 0.5 * cy)).abs().max().item() < 1e-4
    assert (conv3x3(4.0 * x, w, b, dtype="f32-split") - 4.0 * cx).abs().max().item() < 4e-6 * cx.abs().max().item()


@pytest.mark.parametrize("algo,shift", [("f32-direct", 2), ("f32", 2), ("f32-winograd", 4)])
def test_translation_property_full_net(cuda_device, algo, shift):
    """Away from borders the network commutes with translation: the direct and the f16-pair kernels'
    per-pixel arithmetic is position independent (any whole trunk pixel = 2 input px), the Winograd
    kernel's depends only on the position inside its 2x2 output tile (2 trunk px = 4 input px).  A shift
    by that period must reproduce the output bit for bit, shifted by 2x."""
    sd = synthetic_state_dict(seed=3, num_in_ch=3, scale=2, num_block=1)
    net = RRDBNet(3, 3, scale=2, num_block=1, compute_dtype=algo)
    net.load_state_dict(sd)
    net.to(cuda_device)
    x = torch.rand(1, 3, 128, 160, generator=torch.Generator().manual_seed(1)).to(cuda_device)
    y0 = net(x)
    y1 = net(torch.roll(x, shifts=(shift, shift), dims=(2, 3)))
    m = 88   # > receptive-field radius of the 1-block net at output resolution (17 trunk px * 4 + upsampler convs)
    o = 2 * shift
    assert torch.equal(y1[:, :, m + o:-m, m + o:-m], y0[:, :, m:-m - o, m:-m - o])


@pytest.mark.parametrize("kw", [dict(tile=32, tile_pad=10, pre_pad=0), dict(tile=32, tile_pad=10, pre_pad=10), dict(tile=0, tile_pad=10, pre_pad=10)])
@pytest.mark.parametrize("hw", [(64, 96), (63, 95)])
def test_u8_device_pipeline_equals_host_float_path(cuda_device, kw, hw):
    """enhance() keeps 8-bit frames on the GPU end to end; its bytes must equal the upstream-shaped
    host path (enhance_float + numpy quantisation)."""
    sd = synthetic_state_dict(seed=3, num_in_ch=3, scale=2, num_block=2)
    up = RealESRGANer(scale=2, model_path={"params_ema": sd}, model=RRDBNet(3, 3, scale=2, num_block=2), device=cuda_device, **kw)
    img = synthetic_frame(hw[0], hw[1], seed=6)
    assert up._u8_on_device_ok(img)
    got, mode = up.enhance(img)
    f, _, _ = up.enhance_float(img)
    assert mode == "RGB" and np.array_equal(got, (f * 255.0).round().astype(np.uint8))


@pytest.mark.parametrize("inflight", [2, 3])
def test_enhance_many_equals_enhance(cuda_device, inflight):
    """Frames in flight on separate streams / context replicas: same bytes as one at a time, in order."""
    sd = synthetic_state_dict(seed=3, num_in_ch=3, scale=2, num_block=2)
    up = RealESRGANer(scale=2, model_path={"params_ema": sd}, model=RRDBNet(3, 3, scale=2, num_block=2), tile=0, pre_pad=0,
                      half=False, device=cuda_device)
    frames = [synthetic_frame(64, 96, seed=s) for s in range(5)]
    want = [up.enhance(f) for f in frames]
    got = up.enhance_many(frames, inflight=inflight)
    assert len(got) == len(want)
    for (g, gm), (w, wm) in zip(got, want):
        assert gm == wm and np.array_equal(g, w)
    odd = [synthetic_frame(63, 95, seed=9)]                 # needs mod-padding: falls back to enhance()
    assert np.array_equal(up.enhance_many(odd)[0][0], up.enhance(odd[0])[0])


def test_full_test_jpeg_direct_esrgan_settings(cuda_device, golden_dir):
    """The reference's only asset, the whole 512x512 images/test.jpeg, through the drop-in exactly as
    standalone/direct_esrgan.py:104,118-127,148 drives it (tile=512, tile_pad=10, pre_pad=0, half=False), against the
    committed oracle output (every second pixel + a full-resolution window).  8-bit: at most 1 LSB, on rounding ties."""
    g = np.load(os.path.join(golden_dir, "test_jpeg_full.npz"))
    sd = synthetic_state_dict(seed=0, num_in_ch=3, scale=2, num_block=23)
    up = RealESRGANer(scale=2, model_path={"params_ema": sd}, model=RRDBNet(3, 3, scale=2), tile=512, tile_pad=10, pre_pad=0,
                      half=False, device=cuda_device)
    assert up.weights_provenance[0] == "unverified"        # synthetic weights: said so, not implied
    out, mode = up.enhance(g["input_bgr"])
    assert mode == "RGB" and out.shape == (1024, 1024, 3) and up.model.calls == 1
    for got, want in ((out[::2, ::2], g["out_strided"]), (out[384:640, 384:640], g["out_window"])):
        d = np.abs(got.astype(int) - want.astype(int))
        assert d.max() <= 1 and (d > 0).mean() < 1e-3, (d.max(), (d > 0).mean())
