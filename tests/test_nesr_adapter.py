"""The NESR caller-side glue: product torch code vs the oracle's numpy
restatement.  CPU tests run the product functions on CPU tensors with an injected CPU network
(host-logic parity); the GPU test runs them on the MI355X with the HIP network.
PARITY UNPINNED for the two cv2 ops (GaussianBlur, INTER_LANCZOS4): both sides restate OpenCV."""
import numpy as np
import pytest
import torch

from neural_enhanced_super_resolution_amd import nesr_adapter as A
from neural_enhanced_super_resolution_amd.synth import synthetic_frame, synthetic_state_dict
from oracle import nesr_callers_ref as O
from oracle.rrdbnet_ref import RRDBNetRef


class _Up:
    """Minimal stand-in for a RealESRGANer: the adapter touches only .model and .device."""

    def __init__(self, model, device):
        self.model, self.device = model, torch.device(device)


def _ref_model(num_block=1, seed=2):
    sd = synthetic_state_dict(seed=seed, num_in_ch=12, scale=4, num_block=num_block)
    m = RRDBNetRef(12, 3, scale=4, num_block=num_block)
    m.load_state_dict(sd)
    return m, sd


@pytest.mark.parametrize("hw", [(1, 7), (5, 1), (17, 23), (64, 48)])
def test_gaussian_blur_matches_oracle(hw):
    img = synthetic_frame(hw[0], hw[1], seed=hw[0] + hw[1])
    got = A.gaussian_blur3x3_u8(torch.from_numpy(img)).numpy()
    assert np.array_equal(got, O.gaussian_blur3x3_u8(img))


def test_gaussian_blur_known_values():
    img = np.zeros((5, 5, 1), np.uint8)
    img[2, 2, 0] = 160
    out = O.gaussian_blur3x3_u8(img)[:, :, 0]
    assert out[2, 2] == 40 and out[1, 2] == 20 and out[1, 1] == 10 and out[0, 0] == 0   # 160 * [1 2 1]x[1 2 1] / 16


def test_12channel_builder_matches_oracle_bitwise():
    img = synthetic_frame(33, 47, seed=1)
    got = A.build_12channel(img, "cpu").numpy()
    want = O.build_12channel(img)
    assert got.shape == (1, 12, 33, 47) and got.dtype == np.float32
    assert np.array_equal(got, want)
    assert np.array_equal(A.build_3channel_x4(img, "cpu").numpy(), O.build_3channel_x4(img))


def test_truncating_quantiser():
    y = torch.tensor([[[[-0.1, 0.0, 0.5, 0.999, 1.0, 1.7]]] * 3]).float()     # [1,3,1,6]
    q = A.quantize_trunc_to_rgb(y).numpy()
    assert q.tolist() == [[[0, 0, 0], [0, 0, 0], [127, 127, 127], [254, 254, 254], [255, 255, 255], [255, 255, 255]]]


@pytest.mark.parametrize("shape,out", [((40, 56), (20, 28)), ((37, 53), (19, 27)), ((16, 16), (33, 31))])
def test_lanczos_matches_oracle(shape, out):
    img = synthetic_frame(shape[0], shape[1], seed=3)
    got = A.lanczos4_resize_u8(torch.from_numpy(img), out[0], out[1]).numpy()
    want = O.lanczos4_resize_u8(img, out[0], out[1])
    assert got.shape == want.shape
    assert np.abs(got.astype(int) - want.astype(int)).max() <= 1


def test_lanczos_identity_and_constant():
    img = synthetic_frame(12, 9, seed=4)
    assert np.array_equal(A.lanczos4_resize_u8(torch.from_numpy(img), 12, 9).numpy(), img)
    flat = np.full((10, 10, 3), 77, np.uint8)
    assert (A.lanczos4_resize_u8(torch.from_numpy(flat), 5, 5).numpy() == 77).all()


def test_apply_12channel_and_tiler_cpu_vs_oracle():
    model, _ = _ref_model()
    up = _Up(model, "cpu")
    img = synthetic_frame(40, 52, seed=5)
    assert np.array_equal(A.apply_esrgan_12channel(up, img), O.apply_12channel(model, img))
    assert np.array_equal(A.apply_esrgan_3channel(up, img), O.apply_3channel(model, img))
    # tiler: 4x network, 2x canvas -> every tile goes through the Lanczos resize (nesr.py:437-443)
    got = A.process_with_tiling(lambda t: A.apply_esrgan_12channel(up, t, as_numpy=False), img, 24, 4, 2.0, "cpu")
    want = O.process_with_tiling(lambda t: O.apply_12channel(model, np.ascontiguousarray(t)), img, 24, 4, 2.0)
    assert got.shape == want.shape == (80, 104, 3)
    assert np.abs(got.astype(int) - want.astype(int)).max() <= 1


def test_dispatcher_thresholds():
    """nesr.py:762-790: cuda threshold 8 'MP' (px/1024^2); > 16 MP forces tiling + 3-channel."""
    calls = []

    class M(torch.nn.Module):
        def forward(self, x):
            calls.append(tuple(x.shape))
            return torch.zeros(1, 3, x.shape[2] * 4, x.shape[3] * 4)

    up = _Up(M(), "cpu")
    out = A.apply_esrgan(up, np.zeros((64, 64, 3), np.uint8), {"max_tile_size": 32})
    assert out.shape == (256, 256, 3) and calls == [(1, 12, 64, 64)]            # small: untiled, 4x (the quirk)
    calls.clear()
    out = A.apply_esrgan(up, np.zeros((64, 64, 3), np.uint8), {"max_tile_size": 32, "cuda_megapixel_threshold": 0.001})
    assert out.shape == (128, 128, 3) and len(calls) == 4                       # tiled: canvas from upscale_factor 2


def test_iteration_driver_cpu_vs_oracle():
    """nesr.py:516-633 around the ESRGAN stage: three iterations that take the three routes of _apply_esrgan
    (untiled 12-channel -> tiled 12-channel -> forced tiling + 3-channel).  The adapter's host logic with a torch
    CPU network injected, against the numpy oracle with the same network."""
    model, _ = _ref_model()
    up = _Up(model, "cpu")
    img = synthetic_frame(20, 24, seed=9)
    cfg = {"iterations": 3, "upscale_factor": 2.0, "max_tile_size": 48, "cuda_megapixel_threshold": 0.005}
    large = 0.02
    trace, otrace = [], []
    got = A.enhance_iterations(up, img, cfg, "cuda", trace=trace, large_mp=large)
    want = O.enhance_iterations(model, img, cfg, "cuda", large_mp=large, trace=otrace)
    assert got.shape == want.shape == (320, 384, 3)
    assert [(t["tiled"], t["three_channel"]) for t in trace] == [(t["tiled"], t["three_channel"]) for t in otrace] == \
        [(False, False), (True, False), (True, True)]
    assert [t["in_shape"] for t in trace] == [(20, 24), (80, 96), (160, 192)]
    d = np.abs(got.astype(int) - want.astype(int))
    assert d.max() <= 3 and (d > 0).mean() < 0.02      # float32 Lanczos in torch vs numpy: rounding ties, passed on


@pytest.mark.gpu
def test_three_channel_route_on_gpu_vs_oracle(cuda_device):
    """_apply_esrgan_3channel (nesr.py:905-945) on the HIP engine, and the forced branch of nesr.py:787-790
    (beyond the large-image literal: tiling + 3-channel) through the dispatcher."""
    from neural_enhanced_super_resolution_amd import RealESRGANer, RRDBNet
    model, sd = _ref_model(num_block=2, seed=6)
    up = RealESRGANer(scale=2, model_path={"params_ema": sd}, model=RRDBNet(12, 3, num_block=2), tile=0, tile_pad=0,
                      pre_pad=0, half=False, device="cuda")
    img = synthetic_frame(48, 72, seed=7)
    got = A.apply_esrgan_3channel(up, img)
    want = O.apply_3channel(model, img)
    assert got.shape == (192, 288, 3)
    d = np.abs(got.astype(int) - want.astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 2e-3
    calls = up.model.calls
    trace = []
    got = A.apply_esrgan(up, img, {"max_tile_size": 32}, trace=trace, large_mp=0.002)
    want = O.apply_esrgan(model, img, {"max_tile_size": 32}, large_mp=0.002)
    assert trace[0]["tiled"] and trace[0]["three_channel"] and up.model.calls - calls == 6
    assert got.shape == want.shape == (96, 144, 3)
    assert np.abs(got.astype(int) - want.astype(int)).max() <= 2


@pytest.mark.gpu
def test_adapter_on_gpu_vs_oracle(cuda_device):
    from neural_enhanced_super_resolution_amd import RealESRGANer, RRDBNet
    model, sd = _ref_model(num_block=2, seed=6)
    up = RealESRGANer(scale=2, model_path={"params_ema": sd}, model=RRDBNet(12, 3, num_block=2), tile=0, tile_pad=0,
                      pre_pad=0, half=False, device="cuda")                     # nesr/nesr.py:216-229
    img = synthetic_frame(48, 72, seed=7)
    got = A.apply_esrgan_12channel(up, img)
    want = O.apply_12channel(model, img)
    assert got.shape == (192, 288, 3)
    d = np.abs(got.astype(int) - want.astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 2e-3
    got = A.apply_esrgan(up, img, {"max_tile_size": 32, "cuda_megapixel_threshold": 0.001})
    want = O.process_with_tiling(lambda t: O.apply_12channel(model, np.ascontiguousarray(t)), img, 32, 16, 2.0)
    assert got.shape == want.shape == (96, 144, 3)
    assert np.abs(got.astype(int) - want.astype(int)).max() <= 2                # network LSB through the Lanczos taps
