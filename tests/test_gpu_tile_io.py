"""GPU: nesr_cut_tiles_u8 / nesr_paste_tiles_u8 -- all tiles of a frame in one launch each -- against the torch operations of
RealESRGANer.enhance + tile_process they replace (standalone/direct_esrgan.py:148 with tile=512, tile_pad=10): `img / 255`,
BGR->RGB, the padded tile windows; the paste of the un-padded centres, clamp(0, 1), RGB->BGR, x255, round."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("half", [False, True])
def test_cut_and_paste_are_the_torch_operations(cuda_device, half):
    from neural_enhanced_super_resolution_amd.realesrganer import normalize_u8_on_device
    from neural_enhanced_super_resolution_amd.rrdbnet import cut_tiles_u8, paste_tiles_u8
    from neural_enhanced_super_resolution_amd.synth import synthetic_frame
    frame = torch.from_numpy(synthetic_frame(90, 130, seed=2)).to(cuda_device)
    img = normalize_u8_on_device(frame.permute(2, 0, 1).flip(0)).unsqueeze(0)
    if half:
        img = img.half()
    windows = [(0, 0, 40, 52), (30, 48, 60, 82), (50, 100, 40, 30), (89, 129, 1, 1)]
    x = cut_tiles_u8(frame, windows, (60, 82), flip_rgb=True, through_fp16=half)
    for i, (y0, x0, h, w) in enumerate(windows):
        assert torch.equal(x[i, :, :h, :w], img[0, :, y0:y0 + h, x0:x0 + w].float())
        assert float(x[i, :, h:, :].abs().sum()) == 0 and float(x[i, :, :, w:].abs().sum()) == 0
    # paste: values outside [0, 1], ties at .5, a canvas and a packed buffer
    g = torch.Generator().manual_seed(3)
    tiles = (torch.rand(3, 3, 64, 96, generator=g) * 1.4 - 0.2).to(cuda_device)
    tiles[0, :, :4, :4] = torch.tensor([0.5 / 255, 1.5 / 255, 2.5 / 255, 254.5 / 255], device=cuda_device)
    crops = [(2, 3, 40, 60), (0, 0, 64, 96), (10, 20, 1, 7)]
    canvas = torch.zeros((200, 300, 3), dtype=torch.uint8, device=cuda_device)
    dst = [(5, 7), (60, 100), (199, 290)]
    descs = [(cy, cx, h, w, (oy * 300 + ox) * 3, 300 * 3) for (cy, cx, h, w), (oy, ox) in zip(crops, dst)]
    paste_tiles_u8(tiles, descs, canvas, flip_rgb=True, round_nearest=True, through_fp16=half)
    want = torch.zeros_like(canvas)
    for i, ((cy, cx, h, w), (oy, ox)) in enumerate(zip(crops, dst)):
        t = tiles[i, :, cy:cy + h, cx:cx + w]
        t = (t.half() if half else t).float().clamp(0, 1)
        want[oy:oy + h, ox:ox + w] = (t.flip(0).permute(1, 2, 0) * 255.0).round().to(torch.uint8)
    assert torch.equal(canvas, want)
    packed = torch.zeros(40 * 60 * 3 + 7 * 3, dtype=torch.uint8, device=cuda_device)
    paste_tiles_u8(tiles[[0, 2]].contiguous(), [(2, 3, 40, 60, 0, 180), (10, 20, 1, 7, 40 * 60 * 3, 21)], packed, flip_rgb=False, round_nearest=False,
                   through_fp16=half)
    t0 = tiles[0, :, 2:42, 3:63]
    t0 = (t0.half() if half else t0).float().clamp(0, 1)
    assert torch.equal(packed[:7200].view(40, 60, 3), (t0.permute(1, 2, 0) * 255.0).trunc().to(torch.uint8))


def test_bad_descriptors_are_refused(cuda_device):
    from neural_enhanced_super_resolution_amd._lib import NesrHipError
    from neural_enhanced_super_resolution_amd.rrdbnet import cut_tiles_u8, paste_tiles_u8
    frame = torch.zeros((20, 30, 3), dtype=torch.uint8, device=cuda_device)
    with pytest.raises(NesrHipError, match="outside the frame"):
        cut_tiles_u8(frame, [(10, 10, 11, 5)], (16, 16))
    tiles = torch.zeros((1, 3, 8, 8), device=cuda_device)
    with pytest.raises(NesrHipError, match="destination outside"):
        paste_tiles_u8(tiles, [(0, 0, 8, 8, 0, 24)], torch.zeros(100, dtype=torch.uint8, device=cuda_device))
