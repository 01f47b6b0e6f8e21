"""GPU: the cv2-defined steps around the ESRGAN stage on the device (imgproc.py) -- enhance(outscale=...), the alpha
resize, and the pipeline's pre / post filters inside the iteration loop.  PARITY UNPINNED against OpenCV (not installed;
no reference fixture): the checker is oracle/cv2_ref.py, an independent numpy restatement."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_imgproc_on_the_device_equals_the_oracle(cuda_device):
    from neural_enhanced_super_resolution_amd import imgproc as P
    from neural_enhanced_super_resolution_amd.synth import synthetic_frame
    from oracle import cv2_ref as O
    img = synthetic_frame(44, 60, seed=2)[:, :, ::-1].copy()
    d = cuda_device
    assert np.array_equal(P.lanczos4_resize(_t(img, d), 66, 90).cpu().numpy(), O.resize_lanczos4(img, 66, 90))
    assert np.array_equal(P.gaussian_blur_u8(_t(img, d), 3.0).cpu().numpy(), O.gaussian_blur_u8(img, 3.0))
    assert np.array_equal(P.postprocess_image(_t(img, d)).cpu().numpy(), O.postprocess_image(img))
    small = np.ascontiguousarray(img[:14, :16])
    lab = O.rgb2lab_u8(small, True, True)
    planes = np.ascontiguousarray(np.transpose(lab, (2, 0, 1)))
    assert np.array_equal(P.fast_nl_means_u8(_t(planes[1:3], d), 5.0).cpu().numpy(), O.fast_nl_means_u8(planes[1:3], 5.0))
    g = np.ascontiguousarray(img[:, :, 1])
    dd = np.abs(P.clahe_u8(_t(g, d)).cpu().numpy().astype(int) - O.clahe_u8(g).astype(int))
    assert dd.max() <= 1 and (dd > 0).mean() < 0.01


@pytest.mark.parametrize("C,hw,h", [(1, (37, 53), 5.0), (2, (37, 53), 5.0), (1, (5, 7), 10.0), (2, (64, 64), 3.0), (1, (16, 33), 25.0)])
def test_nl_means_kernel_is_the_torch_composition_bit_for_bit(cuda_device, C, hw, h):
    """csrc/imgproc.hip (nesr_nl_means_u8: one launch) against the 441-offset torch composition it replaces, and on the small
    shapes against the oracle's per-pixel loops (oracle/cv2_ref.py, FastNlMeansDenoisingInvoker restated); nesr/nesr.py:674."""
    from neural_enhanced_super_resolution_amd import imgproc as P
    from oracle import cv2_ref as O
    g = torch.Generator().manual_seed(11)
    base = torch.randint(0, 256, (C, hw[0] // 4 + 2, hw[1] // 4 + 2), generator=g).float()
    planes = torch.nn.functional.interpolate(base[None], size=hw, mode="bilinear", align_corners=False)[0]
    planes = (planes + torch.randint(-6, 7, planes.shape, generator=g)).clamp(0, 255).to(torch.uint8)      # smooth image + noise: weights of every size occur
    dev = planes.to(cuda_device)
    got = P.fast_nl_means_u8(dev, h, 7, 21)
    want = P.fast_nl_means_u8(dev, h, 7, 21, use_hip=False)
    assert torch.equal(got, want)
    if h >= 5.0:
        assert not torch.equal(got, dev)                              # it denoises (with h = 3 every weight but the centre's rounds to 0)
    if hw[0] * hw[1] <= 600:
        assert np.array_equal(got.cpu().numpy(), O.fast_nl_means_u8(planes.numpy(), h, 7, 21))


@pytest.mark.parametrize("hw", [(64, 64), (64, 70), (50, 72), (37, 53), (9, 200), (512, 512), (1000, 1777)])
@pytest.mark.parametrize("clip", [2.0, 40.0])
def test_clahe_kernel_is_the_torch_composition_bit_for_bit(cuda_device, hw, clip):
    """csrc/imgproc.hip (nesr_clahe_u8: histogram / table launch + blend launch) against the torch composition it replaces, on sides
    that divide by the grid, sides of which exactly one divides (clahe.cpp pads BOTH then), odd sizes and tiles larger than 65535
    pixels; against the oracle (numpy, oracle/cv2_ref.py) to one grey level (float32 vs float64 blend weights); nesr/nesr.py:680-684."""
    from neural_enhanced_super_resolution_amd import imgproc as P
    from oracle import cv2_ref as O
    g = torch.Generator().manual_seed(5)
    base = torch.randint(0, 256, (1, 1, hw[0] // 8 + 2, hw[1] // 8 + 2), generator=g).float()
    img = torch.nn.functional.interpolate(base, size=hw, mode="bilinear", align_corners=False)[0, 0]
    img = (img + torch.randint(-10, 11, img.shape, generator=g)).clamp(0, 255).to(torch.uint8)
    dev = img.to(cuda_device)
    got = P.clahe_u8(dev, clip, (8, 8))
    want = P.clahe_u8(dev, clip, (8, 8), use_hip=False)
    assert torch.equal(got, want), (got.int() - want.int()).abs().max().item()
    assert not torch.equal(got, dev)
    if hw[0] * hw[1] <= 300000:
        dd = np.abs(got.cpu().numpy().astype(int) - O.clahe_u8(img.numpy(), clip, (8, 8)).astype(int))
        assert dd.max() <= 1 and (dd > 0).mean() < 0.01


def test_nl_means_kernel_time_on_a_large_plane(cuda_device):
    """2048 x 2048 (the second iteration's input of BASELINE.json configs[4]): seconds as torch operations, milliseconds as a kernel."""
    import time
    from neural_enhanced_super_resolution_amd import imgproc as P
    x = torch.randint(0, 256, (2, 2048, 2048), dtype=torch.uint8, device=cuda_device)
    P.fast_nl_means_u8(x[:, :64, :64], 5.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    y = P.fast_nl_means_u8(x, 5.0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"nl-means 2 x 2048 x 2048: {dt * 1e3:.1f} ms")
    assert y.shape == x.shape and dt < 1.0


def test_enhance_outscale_on_gpu(cuda_device):
    from neural_enhanced_super_resolution_amd import RealESRGANer, RRDBNet
    from neural_enhanced_super_resolution_amd.synth import synthetic_frame, synthetic_state_dict
    from oracle.realesrganer_ref import RealESRGANerRef
    from oracle.rrdbnet_ref import RRDBNetRef
    sd = synthetic_state_dict(seed=3, num_in_ch=3, scale=2, num_block=2)
    up = RealESRGANer(scale=2, model_path={"params_ema": sd}, model=RRDBNet(3, 3, scale=2, num_block=2), tile=0, pre_pad=0,
                      half=False, device=cuda_device)
    ref = RealESRGANerRef(scale=2, model_path={"params_ema": sd}, model=RRDBNetRef(3, 3, scale=2, num_block=2), tile=0, pre_pad=0)
    img = synthetic_frame(40, 52, seed=4)
    for outscale in (1.5, 3.5):
        a, _ = up.enhance(img, outscale=outscale)
        b, _ = ref.enhance(img, outscale=outscale)
        assert a.shape == b.shape == (int(40 * outscale), int(52 * outscale), 3)
        d = np.abs(a.astype(int) - b.astype(int))
        assert d.max() <= 2 and (d > 0).mean() < 5e-3       # a network LSB seen through the Lanczos taps


def test_iteration_with_the_reference_filters(cuda_device):
    """One iteration of nesr/nesr.py:516-633 with its own filters on (NL-means + CLAHE before, adaptive unsharp after):
    HIP network + device filters against the oracle chain."""
    from neural_enhanced_super_resolution_amd import RealESRGANer, RRDBNet, nesr_adapter as A
    from neural_enhanced_super_resolution_amd.synth import synthetic_frame, synthetic_state_dict
    from oracle import nesr_callers_ref as O
    from oracle.rrdbnet_ref import RRDBNetRef
    sd = synthetic_state_dict(seed=6, num_in_ch=12, scale=4, num_block=1)
    model = RRDBNetRef(12, 3, num_block=1)
    model.load_state_dict(sd, strict=True)
    up = RealESRGANer(scale=2, model_path={"params_ema": sd}, model=RRDBNet(12, 3, num_block=1), tile=0, tile_pad=0, pre_pad=0,
                      half=False, device=cuda_device)
    img = synthetic_frame(18, 22, seed=5)[:, :, ::-1].copy()
    cfg = {"iterations": 1, "upscale_factor": 2.0}
    trace = []
    got = A.enhance_iterations(up, img, cfg, "cuda", trace=trace, filters=True)
    want = O.enhance_iterations(model, img, cfg, "cuda", filters=True)
    assert got.shape == want.shape == (72, 88, 3) and trace[0]["model_calls"] == 1
    d = np.abs(got.astype(int) - want.astype(int))
    # float Lab / CLAHE stages differ by an LSB between the two restatements and the network and the unsharp mask pass that on
    assert np.median(d) == 0 and (d > 3).mean() < 0.05, (d.max(), (d > 3).mean())
    plain = A.enhance_iterations(up, img, cfg, "cuda")
    assert not np.array_equal(plain, got)            # the filters did something
