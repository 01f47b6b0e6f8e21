"""GPU: BASELINE.json configs[3] as bench.py times it -- RealESRGAN_x4plus (no pixel_unshuffle: the trunk runs at the input
resolution, two nearest x2 + conv stages), half=True (bf16 here), RealESRGANer(scale=4, tile=512, tile_pad=10) on a
1920 x 1080 frame: 23 blocks, upstream's 4 x 3 tile grid (532 x 532 interior, 522 / 394 wide and 522 / 66 tall borders), all
twelve tiles as one ragged batch through the LDS-resident dense-block kernel (rdb_bf16_strip.hip).

Reference: standalone/direct_esrgan.py:104,118-127,148 with the x4plus checkpoint (nesr/utils/downloader.py:29-36).
The CPU oracle (oracle/rrdbnet_ref.py, f32) runs on one interior tile and on the 66-row edge tile; at full size the properties
"tiled enhance() == per-tile model() calls pasted by hand" and "1-rank enhance_sharded == enhance" hold bit for bit.
bf16 is judged by PSNR against the f32 oracle (floor below; 1e-3 is not expected of 8-bit significands through 351 layers).
The small cases cover the bf16 forms nothing else compares with the oracle: x4plus and the nesr 12-channel quirk
(nesr/nesr.py:216: RRDBNet(num_in_ch=12) without scale=2), on both bf16 paths (per-layer kernels / strip kernel)."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

PSNR_FLOOR_BF16 = 40.0     # dB on the float image of a tile vs the f32 oracle (measured 45-48 dB on the bench weights)


def _net_input(bgr_u8):
    return torch.from_numpy(np.ascontiguousarray((bgr_u8[:, :, ::-1].astype(np.float32) / np.float32(255.0)).transpose(2, 0, 1))[None])


def _psnr(a, b):
    mse = float(((a.double() - b.double()) ** 2).mean())
    return 10 * math.log10(1.0 / max(mse, 1e-30))


@pytest.fixture(scope="module")
def c4():
    from neural_enhanced_super_resolution_amd import RRDBNet, RealESRGANer
    from neural_enhanced_super_resolution_amd.synth import synthetic_frame, synthetic_state_dict
    sd = synthetic_state_dict(seed=0, num_in_ch=3, scale=4)
    frame = synthetic_frame(1080, 1920, seed=0)
    up = RealESRGANer(scale=4, model_path={"params_ema": sd}, model=RRDBNet(3, 3, scale=4), tile=512, tile_pad=10, pre_pad=0,
                      half=True, device="cuda:0")
    out, mode = up.enhance(frame)
    return sd, frame, up, out, mode


def test_c4_shapes_tile_grid_and_kernel(c4):
    sd, frame, up, out, mode = c4
    assert up.model.compute_dtype == "bf16" and up.model.out_scale() == 4
    assert out.shape == (4320, 7680, 3) and out.dtype == np.uint8 and mode == "RGB"
    grid = up.tile_grid(1080, 1920)
    assert len(grid) == 12
    shapes = sorted({(g[0][1] - g[0][0], g[0][3] - g[0][2]) for g in grid})
    assert shapes == [(66, 394), (66, 522), (66, 532), (522, 394), (522, 522), (522, 532), (532, 394), (532, 522), (532, 532)]
    assert out.std() > 10
    assert up.model.strip_kernel_active()                 # the dense blocks ran as the LDS-resident kernel
    assert up.model.fused_state() == (True, 0)


def test_c4_tiled_enhance_equals_hand_pasted_tiles(c4, cuda_device):
    sd, frame, up, out, _ = c4
    from neural_enhanced_super_resolution_amd.realesrganer import normalize_u8_on_device
    x = torch.from_numpy(frame).to(cuda_device)
    img = normalize_u8_on_device(x.permute(2, 0, 1).flip(0)).unsqueeze(0).half()
    canvas = img.new_zeros((1, 3, 4320, 7680))
    for (py0, py1, px0, px1), (oy0, oy1, ox0, ox1), (cy0, cy1, cx0, cx1) in up.tile_grid(1080, 1920):
        t = up.model(img[:, :, py0:py1, px0:px1])
        canvas[:, :, oy0:oy1, ox0:ox1] = t[:, :, cy0:cy1, cx0:cx1]
    up.model.check_status()
    q = (canvas[0].float().clamp_(0, 1).flip(0).permute(1, 2, 0) * 255.0).round().to(torch.uint8).cpu().numpy()
    assert np.array_equal(q, out)


def test_c4_one_rank_sharded_equals_enhance(c4):
    from neural_enhanced_super_resolution_amd import sharded
    sd, frame, up, out, _ = c4
    got = sharded.enhance_sharded(up, frame, (1080, 1920))
    assert np.array_equal(got, out)


@pytest.mark.parametrize("which", ["interior_532x532", "edge_66x532"])
def test_c4_tile_vs_oracle_psnr(c4, cuda_device, which):
    from oracle.rrdbnet_ref import RRDBNetRef
    sd, frame, up, out, _ = c4
    grid = up.tile_grid(1080, 1920)
    g = grid[1 * 4 + 1] if which.startswith("interior") else grid[2 * 4 + 1]
    (py0, py1, px0, px1), (oy0, oy1, ox0, ox1), (cy0, cy1, cx0, cx1) = g
    assert f"{py1 - py0}x{px1 - px0}" == which.split("_")[1]
    x = _net_input(frame[py0:py1, px0:px1])
    ref = RRDBNetRef(3, 3, scale=4)
    ref.load_state_dict(sd, strict=True)
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    with torch.no_grad():
        want = ref(x)
    got = up.model(x.to(cuda_device).half()).float().cpu()
    psnr = _psnr(got, want)
    print(f"C4 tile {which}: bf16 x4plus vs f32 oracle PSNR {psnr:.1f} dB, max abs {(got - want).abs().max().item():.3e}")
    assert psnr > PSNR_FLOOR_BF16, psnr
    q = (got[0, :, cy0:cy1, cx0:cx1].half().float().clamp(0, 1).flip(0).permute(1, 2, 0) * 255.0).round().to(torch.uint8).numpy()
    assert np.array_equal(q, out[oy0:oy1, ox0:ox1])
    qref = (want[0, :, cy0:cy1, cx0:cx1].clamp(0, 1).flip(0).permute(1, 2, 0) * 255.0).round().to(torch.uint8).numpy()
    d = np.abs(q.astype(int) - qref.astype(int))
    print(f"   8-bit: max diff {d.max()} LSB, mean {d.mean():.3f}")
    assert d.mean() < 1.5


@pytest.mark.parametrize("strip", ["0", "1"])
@pytest.mark.parametrize("form", ["x4plus", "nesr-12ch"])
def test_bf16_x4_forms_against_the_oracle_at_mininet_size(cuda_device, form, strip):
    """2 blocks, 40 x 56 input, batch 2; bf16 through the per-layer kernels (NESR_STRIP=0) and the strip kernel (NESR_STRIP=1)."""
    from neural_enhanced_super_resolution_amd import RRDBNet
    from neural_enhanced_super_resolution_amd.synth import synthetic_state_dict
    from oracle.rrdbnet_ref import RRDBNetRef
    if form == "x4plus":
        sd = synthetic_state_dict(seed=0, num_in_ch=3, scale=4, num_block=2)
        kw = dict(num_in_ch=3, scale=4)
    else:
        sd = synthetic_state_dict(seed=0, num_in_ch=3, scale=2, num_block=2)     # x2plus tensor shapes: conv_first takes 12 channels
        kw = dict(num_in_ch=12, scale=4)
    old = os.environ.get("NESR_STRIP")
    os.environ["NESR_STRIP"] = strip
    try:
        net = RRDBNet(kw["num_in_ch"], 3, scale=kw["scale"], num_block=2, compute_dtype="bf16")
        net.load_state_dict(sd)
        net.eval().to(cuda_device)
        x = torch.rand(2, kw["num_in_ch"], 40, 56, generator=torch.Generator().manual_seed(5))
        got = net(x.to(cuda_device)).cpu()
        net.check_status()
    finally:
        if old is None:
            os.environ.pop("NESR_STRIP", None)
        else:
            os.environ["NESR_STRIP"] = old
    ref = RRDBNetRef(kw["num_in_ch"], 3, scale=kw["scale"], num_block=2)
    ref.load_state_dict(sd, strict=True)
    with torch.no_grad():
        want = ref(x)
    assert got.shape == want.shape == (2, 3, 160, 224)
    psnr = _psnr(got, want)
    print(f"bf16 {form} (NESR_STRIP={strip}): PSNR {psnr:.1f} dB vs the f32 oracle")
    assert psnr > 55.0, psnr          # two blocks only: far above the 23-block floor
